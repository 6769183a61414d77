"""Shared set-up for the parity tests: the same Taylor-Green problem on the HIP path and on the
CPU oracle, with the oracle fed the product's mesh arrays and dof numbering so that fields and
matrices compare index by index."""
from __future__ import annotations

import numpy as np

from oracle import ipcs_oracle as O

KRYLOV = {
    "tentative": {"ksp_type": "bcgs", "pc_type": "jacobi", "ksp_rtol": 1e-11, "ksp_atol": 1e-30},
    "pressure": {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-11, "ksp_atol": 1e-30},
    "scalar": {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-11, "ksp_atol": 1e-30},
}
LU = {k: {"ksp_type": "preonly", "pc_type": "lu"} for k in ("tentative", "pressure", "scalar")}


def tg_mesh(dim, N, device=None):
    from oasisx_amd import mesh as M

    if dim == 2:
        return M.create_rectangle(None, [[-1.0, -1.0], [1.0, 1.0]], [N, N], device=device)
    return M.create_box(None, [[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]], [N, N, N], device=device)


def on_boundary(x):
    on = np.isclose(np.abs(x[0]), 1.0) | np.isclose(np.abs(x[1]), 1.0)
    return on


def on_boundary3(x):
    return on_boundary(x) | np.isclose(np.abs(x[2]), 1.0)


def make_hip_problem(dim, N, u_deg=2, nu=0.01, dt=0.005, t0=0.0, solver_options=None, window=256,
                     body_force=None, rotational=False, low_memory=True):
    """FractionalStep_AB_CN on the HIP path with the demo's set-up
    (reference demo/taylor_green.py:104-182)."""
    import oasisx_amd as ox

    mesh = tg_mesh(dim, N)
    clock = {"t": t0}
    marker = on_boundary if dim == 2 else on_boundary3
    fns = [O.tg_u, O.tg_v, O.tg_w][:dim]
    bcs_u = [[ox.DirichletBC(lambda x, f=f: f(x, clock["t"], nu), ox.LocatorMethod.GEOMETRICAL, marker)]
             for f in fns]
    S = ox.FractionalStep_AB_CN(mesh, ("Lagrange", u_deg), ("Lagrange", 1), bcs_u=bcs_u, bcs_p=[],
                                solver_options=solver_options or KRYLOV, body_force=body_force,
                                options={"sell_window": window, "low_memory_version": low_memory},
                                rotational=rotational)
    for i, f in enumerate(fns):
        S._u2[i].interpolate(lambda x, f=f: f(x, t0 - dt, nu))
        S._u1[i].interpolate(lambda x, f=f: f(x, t0, nu))
    S._p.interpolate(lambda x: O.tg_p(x, t0 - dt / 2.0, nu))
    return S, clock, mesh


def make_oracle_twin(S, mesh, dim, u_deg=2, nu=0.01, dt=0.005, t0=0.0, solver_options=None, rotational=False):
    """The CPU oracle on the product's mesh arrays and dof numbering."""
    Vi, Q = S._Vi[0][0], S._Q
    return O.taylor_green_problem(
        0, dim, u_deg=u_deg, p_deg=1, nu=nu, dt=dt, t0=t0, solver_options=solver_options or KRYLOV,
        mesh=(mesh.coords.cpu().numpy(), Vi.cells_in_kernel_order()),
        vd=Vi.cell_dofs.cpu().numpy(), qd=Q.cell_dofs.cpu().numpy(),
        x_v=Vi.x.cpu().numpy(), x_q=Q.x.cpu().numpy(), rotational=rotational)


def run_tg_pair(dim=2, N=8, u_deg=2, steps=2, nu=0.01, dt=0.005, hip_options=None, oracle_options=None,
                rotational=False, low_memory=True):
    S, clock, mesh = make_hip_problem(dim, N, u_deg, nu, dt, solver_options=hip_options, rotational=rotational,
                                      low_memory=low_memory)
    R, rclock = make_oracle_twin(S, mesh, dim, u_deg, nu, dt, solver_options=oracle_options, rotational=rotational)
    t = 0.0
    for _ in range(steps):
        t += dt
        clock["t"] = t
        rclock["t"] = t
        S.solve(dt, nu, max_iter=1)
        R.solve(dt, nu, max_iter=1)
    u = S.u.x.array.reshape(-1, dim)
    p = S._p.x.array
    return {"du": float(np.abs(u - R.u1).max()), "dp": float(np.abs(p - R.p).max()),
            "umax": float(np.abs(R.u1).max()), "its_hip": S.iteration_counts(), "its_oracle": dict(R.its),
            "S": S, "R": R}


def delaunay_box_mesh(n, dim=3, seed=0, jitter=0.35):
    """Arrays (points (nv, dim), cells (nc, dim+1) int64) of oasisx_amd.mesh.create_delaunay_box on [-1, 1]^dim:
    a genuinely unstructured simplicial mesh (Delaunay triangulation of a jittered lattice)."""
    from oasisx_amd import mesh as M

    m = M.create_delaunay_box(None, [[-1.0] * dim, [1.0] * dim], n, seed=seed, jitter=jitter, device="cpu")
    return m.coords.cpu().numpy().copy(), m.cells.cpu().numpy().astype(np.int64)
