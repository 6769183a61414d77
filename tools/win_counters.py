"""The LDS-window SpMV kernels (k_spmv_win) of the bench's velocity pattern for the PMC passes of tools/pmc_win.sh:
mass matrix with value codes and an f64-valued matrix, one and three right-hand sides, a few repetitions each."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oasisx_amd import _lib, fem
from oasisx_amd import mesh as M
from oasisx_amd.la import SellMatrix

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(os.environ.get("REPS", "3"))
lib = _lib.load()
mesh = M.create_box(None, [[-1., -1., -1.], [1., 1., 1.]], [N, N, N])
V = fem.FunctionSpace(mesh, 2)
V.build_windows()
P = V.pattern
Mm, Am = SellMatrix(P, symmetric=True), SellMatrix(P)
geom = V.native.nmesh.geom
cells = _lib.ox_cells(mesh.gdim, 0, int(geom.shape[0]), geom.data_ptr())
adj = V.adj.struct()
nb, bptr, bsl, bw = P.bins_args()
_lib.check(lib.ox_assemble_matrix(0, 2, C.byref(cells), _lib.ptr(V.cell_dofs), C.byref(adj), _lib.ptr(V.adj.adj_pos), V.adj.pw,
                                  Mm.ref(), nb, bptr, bsl, bw, _lib.current_stream()), "ox_assemble_matrix")
Mm.version += 1
print("dictionary:", Mm.freeze(pairs="never"), P.w_stats, flush=True)
Am.vals.copy_(Mm.vals * (1.0 + 0.25 * torch.sin(torch.arange(P.size, device="cuda", dtype=torch.float64))))
for nc in (1, 3):
    x = (torch.sin(torch.arange(P.n_cols * nc, device="cuda", dtype=torch.float64) * 1e-3) + 1).reshape(P.n_cols, nc).contiguous()
    y = torch.zeros_like(x)
    for name, A in (("M", Mm), ("A", Am)):
        for _ in range(2):
            A.mult(x, y, nc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            A.mult(x, y, nc)
        e1.record()
        torch.cuda.synchronize()
        print(f"{name} nc={nc}: {e0.elapsed_time(e1) * 1e3 / reps:.1f} us", flush=True)
