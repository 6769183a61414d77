"""Micro-benchmark of the pre-assembled rectangular operators' SpMV (ox_spmv_multi)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oasisx_amd import fem, mesh as M
from oasisx_amd.la import MultiSellMatrix
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
mesh = M.create_box(None, [[-1.,-1.,-1.],[1.,1.,1.]], [N,N,N])
V, Q = fem.FunctionSpace(mesh, 2), fem.FunctionSpace(mesh, 1)
pvq, _, _ = fem.build_rect_pattern(V, Q)
pqv, _, _ = fem.build_rect_pattern(Q, V)
for name, pat, v2s, nx, ny in (("V x Q (s2v)", pvq, False, 1, 3), ("Q x V (v2s)", pqv, True, 3, 1)):
    A = MultiSellMatrix(pat, 3); A.vals.uniform_(0.5, 1.5)
    x = torch.rand(pat.n_cols * nx, device="cuda", dtype=torch.float64)
    y = torch.zeros(pat.n_rows * ny, device="cuda", dtype=torch.float64)
    import ctypes as C
    xp, yp = C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr())
    for _ in range(10): A.mult(v2s, xp, None, 1.0, yp)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): A.mult(v2s, xp, None, 1.0, yp)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    B = pat.nnz * 28 + 8 * (x.numel() + y.numel())
    print(f"{name}: nnz={pat.nnz} slots={pat.size} {us:.1f} us  {B/us/1e3:.0f} GB/s")
