"""Why does the dictionary-off pressure SpMV of bench.py (128^3, 377 MB of f64 values + 16-bit columns: larger than the
256 MB Infinity Cache) time 64 us in one run and 74 us in the next?  One process: the same matrix
is re-created at different device addresses (a dummy allocation of varying size is made first and kept) and timed
(a) back to back, (b) interleaved with a pass over 140 MB of vectors, as inside a CG iteration."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oasisx_amd import _lib, fem
from oasisx_amd import mesh as M
from oasisx_amd.la import SellMatrix

lib = _lib.load()
mesh = M.create_box(None, [[-1., -1., -1.], [1., 1., 1.]], [128, 128, 128])
V = fem.FunctionSpace(mesh, 1)
P = V.pattern
geom = V.native.nmesh.geom
cells = _lib.ox_cells(3, 0, int(geom.shape[0]), geom.data_ptr())
adj = V.adj.struct()
nb, bptr, bsl, bw = P.bins_args()
x = (torch.sin(torch.arange(P.n_cols, device="cuda", dtype=torch.float64) * 1e-3) + 1).reshape(-1, 1).contiguous()
y = torch.zeros_like(x)
vec = torch.zeros(int(140e6 / 8), dtype=torch.float64, device="cuda")
keep = []
for trial, pad_mb in enumerate((0, 37, 101, 256, 300, 517, 1024, 2048)):
    if pad_mb:
        keep.append(torch.empty(pad_mb << 20, dtype=torch.uint8, device="cuda"))
    A = SellMatrix(P, symmetric=True)  # a fresh value array at a new address
    _lib.check(lib.ox_assemble_matrix(1, 1, C.byref(cells), _lib.ptr(V.cell_dofs), C.byref(adj), _lib.ptr(V.adj.adj_pos), V.adj.pw,
                                      A.ref(), nb, bptr, bsl, bw, _lib.current_stream()), "ox_assemble_matrix")
    A.version += 1
    out = []
    for interleave in (False, True):
        for _ in range(10):
            A.mult(x, y, 1)
        ts = []
        for _ in range(60):
            if interleave:
                vec.mul_(1.0000001)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            A.mult(x, y, 1)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        out.append((ts[len(ts) // 2], ts[0], ts[-1]))
    print(f"values at {A.vals.data_ptr():#x} (pad {pad_mb} MB): back to back median {out[0][0]:.1f} us (min {out[0][1]:.1f}, max {out[0][2]:.1f}) | "
          f"after a 140-MB vector pass {out[1][0]:.1f} us (min {out[1][1]:.1f}, max {out[1][2]:.1f})", flush=True)
    del A
