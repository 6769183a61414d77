"""Time per Jacobi-CG iteration on the 128^3 (or N^3) P1 pressure matrix, standard against merged-reduction recurrences
(OX_KSP_CG / OX_KSP_CG_MERGED): a fixed number of iterations (rtol 0), wall time of the solve / iterations.
    python tools/cg_iter_bench.py [N] [iterations]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oasisx_amd import fem, _lib, mesh as M
from oasisx_amd.fem import FieldStorage
from oasisx_amd.ksp import KSPSolver
from oasisx_amd.la import SellMatrix
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ITS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
mesh = M.create_box(None, [[-1., -1., -1.], [1., 1., 1.]], [N, N, N])
V = fem.FunctionSpace(mesh, 1)
A = SellMatrix(V.pattern, symmetric=True)
lib = _lib.load()
geom = V.native.nmesh.geom
cells = _lib.ox_cells(mesh.gdim, 0, int(geom.shape[0]), geom.data_ptr())
adj = V.adj.struct()
nb, bptr, bsl, bw = V.pattern.bins_args()
_lib.check(lib.ox_assemble_matrix(1, 1, C.byref(cells), _lib.ptr(V.cell_dofs), C.byref(adj), _lib.ptr(V.adj.adj_pos), V.adj.pw,
                                  A.ref(), nb, bptr, bsl, bw, _lib.current_stream()), "ox_assemble_matrix")
A.version += 1
print("dictionary:", A.freeze(), "pair stream:", A.ps_code is not None)
n = V.num_dofs
x0 = torch.sin(torch.arange(n, device="cuda", dtype=torch.float64) * 1e-3)
B = FieldStorage(n, 1, "cuda")
A.mult(x0.reshape(-1, 1).contiguous(), B.dev(), 1)
for rnd in range(int(os.environ.get("ROUNDS", "3"))):
    for merged in (False, True):
        ksp = KSPSolver(None, {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 0.0, "ksp_atol": 0.0, "ksp_max_it": ITS,
                               "ksp_cg_merged_reduction": merged, "ksp_cg_single_reduction": False})
        ksp.setOperators(A)
        X = FieldStorage(n, 1, "cuda")
        ksp.solve_block(B, X)  # warm-up (work space, check interval)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ksp.solve_block(B, X)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"N={N} merged={merged}: {1e6 * (t1 - t0) / max(ksp.iterations[0], 1):.1f} us per iteration ({ksp.iterations[0]} iterations, "
              f"reason {ksp.last_result.reason[0]}, |D^-1 r| {ksp.last_result.rnorm[0]:.3e})")
