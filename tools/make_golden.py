"""Generate tests/golden/*.npz with the numpy oracle (oracle/ipcs_oracle.py).

SELF-GENERATED fixtures: the reference cannot run in this container (DOLFINx/PETSc absent), and
its own tests hold no numeric vectors for this path, so these files pin OUR oracle against
regressions and give the GPU tests a fixed target; they are not reference-captured outputs.
Mesh, dof numbering and every number inside come from the oracle alone (the files carry the
mesh and dof tables so any implementation can be compared through them).

    python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ipcs_oracle as O  # noqa: E402

LU = {k: {"ksp_type": "preonly", "pc_type": "lu"} for k in ("tentative", "pressure", "scalar")}


def run(dim, N, u_deg, steps, dt=0.005, nu=0.01):
    S, clock = O.taylor_green_problem(N, dim, u_deg=u_deg, nu=nu, dt=dt, solver_options=LU)
    out = {"coords": S.F.coords, "cells": S.F.cells, "vd": S.F.vd, "qd": S.F.qd, "x_v": S.x_v, "x_q": S.x_q,
           "dt": dt, "nu": nu, "u_deg": u_deg, "bc_dofs": S.bcs_u[0][0].dofs}
    t = 0.0
    for s in range(1, steps + 1):
        t += dt
        clock["t"] = t
        S.solve(dt, nu, max_iter=1)
        if s == 1:
            A = S.A.tocsr()
            out.update(A_indptr=A.indptr, A_indices=A.indices, A_data=A.data, rhs1=S.rhs1.copy(),
                       b_first=S.b_first.copy(), dp_1=S.dp.copy())
        if s in (1, steps):
            out[f"u_{s}"] = S.u1.copy()
            out[f"p_{s}"] = S.p.copy()
    return out


if __name__ == "__main__":
    gold = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    for name, args in (("tg2d_p2p1_n8", (2, 8, 2, 5)), ("tg2d_p1p1_n8", (2, 8, 1, 5)),
                       ("tg3d_p2p1_n3", (3, 3, 2, 3)), ("tg3d_p1p1_n4", (3, 4, 1, 3))):
        d = run(*args)
        np.savez_compressed(os.path.join(gold, name + ".npz"), **d)
        print(name, {k: np.asarray(v).shape for k, v in d.items() if k.startswith(("u_", "p_"))})
