"""Diagnostic: run-to-run bit identity of two plain steps (no snapshots in between)."""
import sys
import torch
sys.path.insert(0, ".")
from tests.helpers import KRYLOV, make_hip_problem

opts = {k: dict(v, ksp_initial_guess_nonzero=True) for k, v in KRYLOV.items()}
dt, nu = 0.005, 0.01
NRUN = int(sys.argv[1]) if len(sys.argv) > 1 else 40
NSTEP = int(sys.argv[2]) if len(sys.argv) > 2 else 2


def run():
    S, clock, mesh = make_hip_problem(3, 5, u_deg=2, solver_options=opts)
    init = [S._M.vals.clone(), S._K.vals.clone(), S._Ap.vals.clone(), S._U1.rdev().clone(), S._U2.rdev().clone(), S._P.dev().clone()]
    its = []
    for k in range(NSTEP):
        clock["t"] += dt
        S.solve(dt, nu)
        its.append([list(S._solver_u.iterations), list(S._solver_p.iterations), list(S._solver_c.iterations)])
    return init, [S._U.rdev().clone(), S._P.dev().clone(), S._RHS1.dev().clone(), S._BFIRST.dev().clone()], its


ref = run()
nbad = 0
for r in range(NRUN):
    cur = run()
    for j, (a, b) in enumerate(zip(ref[0], cur[0])):
        if not torch.equal(a, b):
            print("run", r, "INIT differs", j, float((a - b).abs().max()), flush=True)
    bad = [j for j, (a, b) in enumerate(zip(ref[1], cur[1])) if not torch.equal(a, b)]
    if bad:
        nbad += 1
        print("run", r, "differs in", bad, [float((ref[1][j] - cur[1][j]).abs().max()) for j in bad], ref[2], cur[2], flush=True)
print("runs", NRUN, "differing", nbad)
