"""Velocity-side SpMV kernels under the PMC passes of tools/pmc_u.sh: one process builds the P2 space of
the bench workload once and runs, a few repetitions each,
  k_spmv<3,0,3> / k_spmv<1,0,3>  on a matrix with f64 values (what A is: rebuilt every step, no dictionary),
  k_spmv<3,0,7> / k_spmv<1,0,7>  on the mesh's mass matrix with its value dictionary (M in velocity_update).
HIP-event times are printed beside (python tools/spmv_counters.py [N])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from oasisx_amd import fem, _lib, mesh as M
from oasisx_amd.la import SellMatrix

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(os.environ.get("REPS", "3"))
mesh = M.create_box(None, [[-1., -1., -1.], [1., 1., 1.]], [N, N, N])
V = fem.FunctionSpace(mesh, 2)
P = V.pattern
lib = _lib.load()
A = SellMatrix(P)
A.vals.uniform_(0.5, 1.5)
Mm = SellMatrix(P)
geom = V.native.nmesh.geom
cells = _lib.ox_cells(mesh.gdim, 0, int(geom.shape[0]), geom.data_ptr())
adj = V.adj.struct()
nb, bptr, bsl, bw = P.bins_args()
_lib.check(lib.ox_assemble_matrix(0, V.degree, C.byref(cells), _lib.ptr(V.cell_dofs), C.byref(adj), _lib.ptr(V.adj.adj_pos),
                                  V.adj.pw, Mm.ref(), nb, bptr, bsl, bw, _lib.current_stream()), "ox_assemble_matrix")
Mm.version += 1
print("mass matrix dictionary:", Mm.freeze(), Mm._struct.n_dict, flush=True)
for nc in (3, 1):
    x = (torch.sin(torch.arange(P.n_cols * nc, device="cuda", dtype=torch.float64) * 1e-3) + 1).reshape(P.n_cols, nc).contiguous()
    y = torch.zeros_like(x)
    for name, mat in (("A (f64 values)", A), ("M (value codes)", Mm)):
        for _ in range(2):
            mat.mult(x, y, nc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            mat.mult(x, y, nc)
        e1.record()
        torch.cuda.synchronize()
        print(f"{name} nc={nc}: {e0.elapsed_time(e1) * 1e3 / reps:.1f} us per launch", flush=True)
