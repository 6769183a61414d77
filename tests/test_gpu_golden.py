"""GPU: the HIP path against the committed golden fixtures (tests/golden, made by
tools/make_golden.py from the numpy oracle).  The product numbers its dofs differently from the
fixture, so fields are matched through the dof coordinates."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _match(xa, xb):
    """perm with xa[perm] == xb (coordinates, exact up to 1e-12)."""
    def key(x):
        q = np.round(x * 4096).astype(np.int64)
        k = q[:, 0]
        for j in range(1, q.shape[1]):
            k = k * (1 << 20) + q[:, j]
        return k
    ka, kb = key(xa), key(xb)
    oa, ob = np.argsort(ka), np.argsort(kb)
    assert (ka[oa] == kb[ob]).all()
    perm = np.empty_like(oa)
    perm[ob] = oa
    assert np.abs(xa[perm] - xb).max() < 1e-12
    return perm


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "*.npz"))))
def test_hip_path_reproduces_golden(hip, path):
    from tests.helpers import LU, make_hip_problem

    g = np.load(path)
    dim = g["coords"].shape[1]
    N = int(round((g["coords"].shape[0]) ** (1.0 / dim))) - 1
    dt, nu = float(g["dt"]), float(g["nu"])
    S, clock, mesh = make_hip_problem(dim, N, int(g["u_deg"]), nu=nu, dt=dt, solver_options=LU)
    pv = _match(S._Vi[0][0].x.cpu().numpy(), g["x_v"])  # product index of golden dof k
    pq = _match(S._Q.x.cpu().numpy(), g["x_q"])
    last = max(int(k[2:]) for k in g.files if k.startswith("u_") and k[2:].isdigit())
    t = 0.0
    for s in range(1, last + 1):
        t += dt
        clock["t"] = t
        S.solve(dt, nu, max_iter=1)
        if s == 1:
            rhs1 = np.stack([f.x.array for f in S._rhs1], axis=1)
            assert np.abs(rhs1[pv] - g["rhs1"]).max() < 1e-10 * np.abs(g["rhs1"]).max()
        if f"u_{s}" in g.files:
            u = S.u.x.array.reshape(-1, dim)
            assert np.abs(u[pv] - g[f"u_{s}"]).max() < 1e-8
            assert np.abs(S._p.x.array[pq] - g[f"p_{s}"]).max() < 1e-7
