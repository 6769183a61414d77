"""assemble_first timing (HIP events around the call, 10 repetitions): the one-launch row-block form against the width
bins, bit-identity checked.  ``af_bench.py N [box|delaunay] [refine]``: box = N^3 x 6 tetrahedra (bench size 128);
delaunay = a jittered (N+1)^3 lattice triangulated and refined uniformly (32 2 = the bench's unstructured leg)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oasisx_amd as ox
from oasisx_amd import mesh as M
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
kind = sys.argv[2] if len(sys.argv) > 2 else "box"
refine = int(sys.argv[3]) if len(sys.argv) > 3 else 0
box = [[-1., -1., -1.], [1., 1., 1.]]
mesh = M.create_box(None, box, [N, N, N]) if kind == "box" else M.create_delaunay_box(None, box, N, refine=refine)
bcs = [[ox.DirichletBC(0.0, ox.LocatorMethod.GEOMETRICAL, lambda x: np.isclose(np.abs(x[0]), 1.0))] for _ in range(3)]
out = {}
modes = [m == "1" for m in os.environ.get("MODES", "0 1").split()]
S = None
for blocks in modes:
    if S is None:
        S = ox.FractionalStep_AB_CN(mesh, ("Lagrange", 2), ("Lagrange", 1), bcs_u=bcs, bcs_p=[],
                                    options=dict({"low_memory_version": True, "assemble_row_blocks": blocks},
                                                 **({"spmv_windows": True} if os.environ.get("BRICK") == "1" else {})))
        print("brick order of the P2 numbering:", S._Vi[0][0].brick, flush=True)
        g = torch.Generator(device="cuda").manual_seed(1)
        S._U1.dev().copy_(torch.randn(S._U1.dev().shape, dtype=torch.float64, device="cuda", generator=g))
        S._U2.dev().copy_(torch.randn(S._U2.dev().shape, dtype=torch.float64, device="cuda", generator=g))
        P = S._A.pattern
        print(f"{kind} N={N} refine={refine}: {P.n_rows} rows, {P.n_slices} slices, {len(P.bin_width)} width bins "
              f"{list(map(int, P.bin_width))}, {P.n_row_blocks} row blocks of <= {P.row_blk_entries * 8 // 1024} KB", flush=True)
    S._row_blocks = blocks
    for _ in range(3): S.assemble_first(0.01, 0.01)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): S.assemble_first(0.01, 0.01)
    e1.record(); torch.cuda.synchronize()
    out[blocks] = (S._A.vals.clone(), S._BFIRST.dev().clone())
    print(f"row_blocks={blocks}: assemble_first {e0.elapsed_time(e1)/10:.3f} ms; "
          f"checksum {float(S._A.vals.sum()):.12e} {float(S._BFIRST.dev().sum()):.12e}", flush=True)
if len(out) == 2:
    print("bit-identical:", torch.equal(out[False][0], out[True][0]) and torch.equal(out[False][1], out[True][1]))
