"""Storage and SpMV figures of an unstructured (Delaunay) tetrahedral mesh: SELL-64 padding, 16-bit
column coverage, pressure / velocity SpMV time and GB/s on bytes moved -- the numbers DESIGN.md quotes
for "what an unstructured mesh gets".   python tools/irregular_report.py [n]   (jittered (n+1)^3 lattice)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oasisx_amd import fem, mesh as M
from oasisx_amd.la import SellMatrix
from tests.helpers import delaunay_box_mesh
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
t0 = time.time(); pts, tets = delaunay_box_mesh(n, 3, seed=0); t1 = time.time()
mesh = M.from_arrays(pts, tets)
print(f"Delaunay mesh: {pts.shape[0]} vertices, {tets.shape[0]} tetrahedra ({t1 - t0:.1f} s in Qhull)")
for deg, nc in ((1, 1), (2, 1), (2, 3)):
    t0 = time.time(); V = fem.FunctionSpace(mesh, deg); torch.cuda.synchronize(); ts = time.time() - t0
    P = V.pattern
    A = SellMatrix(P); A.vals.uniform_(0.5, 1.5)
    rl = P.row_len.cpu().numpy()
    x = torch.rand(P.n_cols, nc, dtype=torch.float64, device="cuda"); y = torch.zeros_like(x)
    for _ in range(10): A.mult(x, y, nc)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): A.mult(x, y, nc)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    moved = P.size * (8 + 2 * P.frac16 + 4 * (1 - P.frac16)) + 8 * (P.size // 128) + nc * 16 * P.n_rows
    print(f"P{deg} x{nc}: set-up {ts:.2f} s, rows {P.n_rows}, nnz {P.nnz}, row length {rl.min()}..{rl.max()} (mean {rl.mean():.1f}), "
          f"SELL-64 padding {P.size / P.nnz - 1:.1%}, 16-bit column coverage {P.frac16:.3f}; SpMV {us:.1f} us = "
          f"{moved / us / 1e3:.0f} GB/s on bytes moved ({moved / us / 1e3 / 8000:.2f} of 8 TB/s)")
