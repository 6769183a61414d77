"""The partitioned pressure CG of rank 0 of a P-rank job at 128^3, alone on the GPU with self-loop plans
(parallel.SelfLoopComm): 400 forced iterations, three times -- the command to put behind `rocprofv3 --kernel-trace --stats --`
for the kernel budget of one partitioned iteration.
    python tools/selfloop_cg_trace.py [p2p|rccl] [P] [pressure|tentative|update|update1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oasisx_amd as ox
from oasisx_amd import mesh as M
from oasisx_amd.fem import FieldStorage
from oasisx_amd.ksp import KSPSolver
from oasisx_amd.parallel import SelfLoopComm
tr = sys.argv[1] if len(sys.argv) > 1 else "p2p"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
comm = SelfLoopComm(0, P, tr)
N = 128
mesh = M.create_box(comm, [[-1.0] * 3, [1.0] * 3], [N, N, N])
on = lambda x: np.isclose(np.abs(x[0]), 1.0) | np.isclose(np.abs(x[1]), 1.0) | np.isclose(np.abs(x[2]), 1.0)
bcs = [[ox.DirichletBC(0.0, ox.LocatorMethod.GEOMETRICAL, on)] for _ in range(3)]
KSP = {"pc_type": "jacobi", "ksp_rtol": 1e-8, "ksp_atol": 1e-14, "ksp_max_it": 10000, "ksp_initial_guess_nonzero": True}
S = ox.FractionalStep_AB_CN(mesh, ("Lagrange", 2), ("Lagrange", 1), bcs_u=bcs, bcs_p=[],
                            solver_options={"tentative": dict(KSP, ksp_type="bcgs"), "pressure": dict(KSP, ksp_type="cg"), "scalar": dict(KSP, ksp_type="cg")})
what = sys.argv[3] if len(sys.argv) > 3 else "pressure"
Q, A, nc, kind, its = {"pressure": (S._Q, S._Ap, 1, "cg", 400), "tentative": (S._Vi[0][0], S._A, 3, "bcgs", 40),
                       "update": (S._Vi[0][0], S._M, 3, "cg", 60), "update1": (S._Vi[0][0], S._M, 1, "cg", 120)}[what]
if what == "tentative":
    S.assemble_first(0.00125, 0.01)  # (A = M / dt + ... : a matrix BiCGStab can iterate on)
n = Q.n_local
B, X = FieldStorage(n, nc, "cuda"), FieldStorage(n, nc, "cuda")
B.dev().copy_(torch.randn(n, nc, dtype=torch.float64, device="cuda"))
ksp = KSPSolver(comm, {"ksp_type": kind, "pc_type": "jacobi", "ksp_rtol": 0.0, "ksp_atol": 0.0, "ksp_max_it": its})
ksp.setOperators(A)
for rep in range(3):
    X.dev().zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
    ksp.solve_block(B, X); torch.cuda.synchronize()
    print(f"{tr} P={P} {what}: {1e6 * (time.perf_counter() - t0) / its:.1f} us per iteration, {nc} column(s), rows {Q.n_owned} + {n - Q.n_owned} ghosts", flush=True)
