"""Time per Jacobi-BiCGStab iteration on the 128^3 (or N^3) P2 velocity pattern, three right-hand sides in lockstep
(what velocity_tentative_solve runs): a fixed number of iterations (rtol 0), wall time of the solve / iterations; the
matrix is M/dt-like plus a non-symmetric perturbation.  With OX_LIB_PATH a tuning build of the library is timed.
    python tools/bcgs_iter_bench.py [N] [iterations]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oasisx_amd import fem, _lib, mesh as M
from oasisx_amd.fem import FieldStorage
from oasisx_amd.ksp import KSPSolver
from oasisx_amd.la import SellMatrix
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ITS = int(sys.argv[2]) if len(sys.argv) > 2 else 12
mesh = M.create_box(None, [[-1., -1., -1.], [1., 1., 1.]], [N, N, N])
V = fem.FunctionSpace(mesh, 2)
A = SellMatrix(V.pattern)
lib = _lib.load()
geom = V.native.nmesh.geom
cells = _lib.ox_cells(3, 0, int(geom.shape[0]), geom.data_ptr())
adj = V.adj.struct()
nb, bptr, bsl, bw = V.pattern.bins_args()
_lib.check(lib.ox_assemble_matrix(0, 2, C.byref(cells), _lib.ptr(V.cell_dofs), C.byref(adj), _lib.ptr(V.adj.adj_pos), V.adj.pw,
                                  A.ref(), nb, bptr, bsl, bw, _lib.current_stream()), "ox_assemble_matrix")
A.vals.mul_(1.0 + 0.05 * torch.sin(torch.arange(A.vals.numel(), device="cuda", dtype=torch.float64)))
A.version += 1
n = V.num_dofs
B, X = FieldStorage(n, 3, "cuda"), FieldStorage(n, 3, "cuda")
B.dev().copy_(torch.sin(torch.arange(3 * n, device="cuda", dtype=torch.float64) * 1e-3).reshape(n, 3))
for rnd in range(int(os.environ.get("ROUNDS", "3"))):
    ksp = KSPSolver(None, {"ksp_type": "bcgs", "pc_type": "jacobi", "ksp_rtol": 0.0, "ksp_atol": 0.0, "ksp_max_it": ITS})
    ksp.setOperators(A)
    ksp.solve_block(B, X)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ksp.solve_block(B, X)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"N={N}: {1e3 * (t1 - t0) / max(ksp.iterations[0], 1):.3f} ms per BiCGStab iteration ({ksp.iterations[0]} iterations, "
          f"|D^-1 r| {ksp.last_result.rnorm[0]:.6e})", flush=True)
