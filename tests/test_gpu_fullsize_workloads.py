"""GPU: the non-headline BASELINE workloads at the metric's size, 128^3 x 6 tets P2-P1 (12 582 912 cells, 16 974 593
velocity dofs per component), against the C/OpenMP port on ITS OWN mesh, dof numbering, sparsity patterns and
operators (oracle/cpu_baseline.from_mesh): only the mesh definition and the state vectors are shared, matched through
the dof coordinates.

  * lid-driven cavity Re = 1000 from rest (BASELINE.json configs[3]'s problem on one GPU; reference
    fracstep.py:660-696 drives the step): three steps on the device, the fourth on both sides;
  * Ethier-Steinman Beltrami flow (all three components and the pressure non-trivial; SURVEY.md 8d's stronger field):
    two steps on the device, the third on both sides.

Checked: relative L2 difference of u and p after the common step (Krylov tolerance 1e-8 on both sides: the bound is the
solver tolerance times the conditioning, as in bench.py's cross-check), every converged reason > 0, Krylov iteration
counts within +-2 (or 3 %) of the port's, the Dirichlet data themselves (lid exactly 1 on its 257^2 P2 dofs, walls exactly 0)
and the tentative velocity on the identity rows (un-lifted rows inside the Krylov solve: the data to the solver
tolerance).  About 40 s of host work each.
"""
import gc

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(wname, gpu_steps):
    import bench
    import oasisx_amd as ox
    from oasisx_amd import mesh as M
    from oracle.cpu_baseline import run_cpu_baseline

    N = 128
    W = bench.make_workload(wname, N, np, torch)
    p0, p1 = W["box"]
    nu, dt, fns = W["nu"], W["dt"], W["fns"]
    clock = {"t": 0.0}

    def on_boundary(x):
        on = np.zeros(x.shape[1], dtype=bool)
        for k in range(3):
            on |= np.isclose(x[k], p0[k]) | np.isclose(x[k], p1[k])
        return on

    def bcv(f):
        def g(x):
            return f(x, clock["t"])
        g.supports_torch = True
        return g

    def at(f, t):
        def g(x):
            return f(x, t)
        g.supports_torch = True
        return g

    mesh = M.create_box(None, [p0, p1], [N, N, N])
    ksp = {"pc_type": "jacobi", "ksp_rtol": 1e-8, "ksp_atol": 1e-14, "ksp_max_it": 10000, "ksp_initial_guess_nonzero": True}
    tent = dict(ksp, ksp_type="bcgs")
    if wname == "cavity":
        tent["ksp_bcgs_restarts"] = 5  # a start from rest breaks BiCGStab down exactly (tests/test_gpu_configs.py)
    S = ox.FractionalStep_AB_CN(
        mesh, ("Lagrange", 2), ("Lagrange", 1),
        bcs_u=[[ox.DirichletBC(bcv(f), ox.LocatorMethod.GEOMETRICAL, on_boundary)] for f in fns], bcs_p=[],
        solver_options={"tentative": tent, "pressure": dict(ksp, ksp_type="cg"), "scalar": dict(ksp, ksp_type="cg")},
        options={"low_memory_version": False})
    assert mesh.num_cells == 12582912 and S._n_u == 16974593 and S._n_q == 2146689
    assert S._M.pattern.nnz == 484609025 and S._Ap.pattern.nnz == 31802497
    if W["analytic"]:
        for i, f in enumerate(fns):
            S._u2[i].interpolate(at(f, -dt))
            S._u1[i].interpolate(at(f, 0.0))
        S._p.interpolate(lambda x: W["p"](x, -dt / 2.0))

    # the Dirichlet values of the tentative velocity, read right after its solve (velocity_update does not
    # re-impose them: reference fracstep.py:607-658)
    bc = S._bcs_u[0][0]
    seen = {}
    solve_t = S.velocity_tentative_solve

    def tentative_and_record():
        r = solve_t()
        idx = bc._dofs_dev.to(torch.int64)
        seen["u_bc"] = S._U.dev()[idx].clone()
        seen["g"] = torch.stack([b[0]._g_dev for b in S._bcs_u], dim=1).clone()
        seen["errors"] = r[1]
        return r

    S.velocity_tentative_solve = tentative_and_record

    def step():
        clock["t"] += dt
        S.solve(dt, nu, max_iter=1)

    for _ in range(gpu_steps):
        step()
    out = run_cpu_baseline(S, clock, dt, nu, {"rtol": 1e-8, "atol": 1e-14, "max_it": 10000, "guess": True},
                           lambda X, t: np.stack([np.asarray(f(X, t), dtype=np.float64) for f in fns]), gpu_step=step,
                           mesh_def=(p0, p1, [N, N, N]), threads_1=False, scipy_check=False)
    errors, error_p, errors_c = S._last_errors
    res = {"out": out, "seen": seen, "reasons": (errors, error_p, errors_c), "umax": float(S._U.dev().abs().max())}
    del S, mesh
    gc.collect()
    torch.cuda.empty_cache()
    return res


def _common_checks(res):
    out = res["out"]
    errors, error_p, errors_c = res["reasons"]
    assert (np.asarray(errors) > 0).all() and int(error_p) > 0 and (np.asarray(errors_c) > 0).all(), res["reasons"]
    # rtol 1e-8 on both sides, different summation orders: solver tolerance x conditioning bounds the difference
    assert out["gpu_vs_cpu_rel_l2_u"] < 1e-8 and out["gpu_vs_cpu_rel_l2_p"] < 1e-6, out
    its_g, its_c = out["gpu_krylov_iterations"], out["krylov_iterations"]
    for k in ("tentative", "pressure", "update"):
        # (hundreds of CG iterations with two summation orders: a few iterations either way)
        assert abs(max(its_g[k]) - max(its_c[k])) <= max(2, 0.03 * max(its_c[k])), (k, its_g, its_c)
    seen = res["seen"]
    # identity rows (A[row] = e_row, un-lifted: fracstep.py:470-472) are solved by the Krylov method with all others:
    # the tentative velocity carries the Dirichlet values to the solver tolerance
    assert float((seen["u_bc"] - seen["g"]).abs().max()) < 1e-8


def test_cavity_128cubed_step_against_the_c_port_on_its_own_mesh(hip):
    res = _run("cavity", gpu_steps=3)
    _common_checks(res)
    g = res["seen"]["g"]
    # lid 1 (x component on z = 1, corners and edges included), walls 0 -- exactly
    assert set(torch.unique(g[:, 0]).tolist()) == {0.0, 1.0} and float(g[:, 1:].abs().max()) == 0.0
    assert int((g[:, 0] == 1.0).sum()) == 257 ** 2  # the P2 dofs of the lid
    assert 0.5 < res["umax"] < 1.5  # the lid drives the flow
    its = res["out"]["gpu_krylov_iterations"]
    assert max(its["pressure"]) > 100  # a from-rest pressure solve is a real one


def test_beltrami_128cubed_step_against_the_c_port_on_its_own_mesh(hip):
    res = _run("beltrami", gpu_steps=2)
    _common_checks(res)
    assert res["umax"] > 1.0
