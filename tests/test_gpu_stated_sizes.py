"""GPU: BASELINE.json's configurations at their STATED sizes, against checkers that share nothing
with the product but the mesh definition: the oracle / C port generate their own mesh, number their
own dofs, build their own sparsity patterns and assemble their own operators; fields are matched
through the dof coordinates (a wrong dof map, edge numbering or geometry in the product cannot
cancel out).

  C1  2-D Taylor-Green 32 x 32 on [-1,1]^2, P2-P1 (reference demo/taylor_green.py:126-182):
      numpy oracle, direct solves (scipy splu) against the device Krylov solves at 1e-12
  C1' the demo's convergence study N = 8, 16, 32 (reference .github/workflows/tests.yml:59,
      demo/taylor_green.py:225-241) on the device: space-time L2 errors fall at the expected rates
  C2  3-D Taylor-Green 64^3 x 6 tets, P1-P1: the C/OpenMP port (validated against the numpy oracle
      in tests/test_oracle_pins.py), same Krylov methods and tolerances
  C5  the defining feature of the 256^3 P2-P1 configuration -- a pattern with more than 2^31 storage
      slots (int64 offsets, blocked set-up): the smallest even N that crosses it (212), entry counts
      from SURVEY.md section 8's formulas, partition of unity, nullspace, one step against the
      analytic field.  (256^3 itself: profiles/, bench.py -N 256 --matrix-free.)
"""
import gc

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _match(xa, xb, lo, hi):
    from oracle.cpu_baseline import match_by_coordinates

    return match_by_coordinates(xa, xb, np.asarray(lo, dtype=float), np.asarray(hi, dtype=float))


def test_c1_taylor_green_32x32_p2p1_against_the_oracle_on_its_own_mesh(hip):
    from oracle import ipcs_oracle as O
    from tests.helpers import LU, make_hip_problem

    N, nu, dt, steps = 32, 0.01, 0.005, 5
    S, clock, mesh = make_hip_problem(2, N, 2, nu=nu, dt=dt, solver_options=LU, window=4096)
    R, rclock = O.taylor_green_problem(N, 2, u_deg=2, p_deg=1, nu=nu, dt=dt, solver_options=LU)  # own mesh, own dofs
    assert S._n_u == 4225 and S._n_q == 1089 and S._M.pattern.nnz == 47617 and S._Ap.pattern.nnz == 7361
    assert R.M.nnz == 47617 and R.Ap.nnz == 7361
    pv = _match(S._Vi[0][0].x.cpu().numpy(), R.F.x_v, [-1, -1], [1, 1])
    pq = _match(S._Q.x.cpu().numpy(), R.F.x_q, [-1, -1], [1, 1])
    # operators, entry by entry, through the coordinate matching
    for A_hip, A_or in ((S._M, R.M), (S._K, R.K)):
        Ah = A_hip.to_scipy()[pv][:, pv]
        assert abs(Ah - A_or).max() < 1e-13 * abs(A_or).max()
    Ah = S._Ap.to_scipy()[pq][:, pq]
    assert abs(Ah - R.Ap).max() < 1e-12 * abs(R.Ap).max()
    t = 0.0
    for s in range(steps):
        t += dt
        clock["t"] = rclock["t"] = t
        S.solve(dt, nu, max_iter=1)
        R.solve(dt, nu, max_iter=1)
        if s == 0:
            rhs1 = np.stack([f.x.array for f in S._rhs1], axis=1)
            assert np.abs(rhs1[pv] - R.rhs1).max() < 1e-10 * np.abs(R.rhs1).max()
    u = S.u.x.array.reshape(-1, 2)
    # direct solves on the host vs Krylov at rtol 1e-12 on the device: solver tolerance x conditioning
    assert np.abs(u[pv] - R.u1).max() < 1e-8
    assert np.abs(S._p.x.array[pq] - R.p).max() < 1e-7


def test_c1_demo_convergence_rates_on_the_device(hip):
    """The demo's study (N = 8, 16, 32; dt = 0.005, T = 0.1 here): the space-time L2 errors of u
    and p fall from one mesh to the next -- the analytic pin of the whole path on the device."""
    from demo.taylor_green_hip import run_taylor_green

    errs = [run_taylor_green(N, dt=0.005, T=0.1, nu=0.01, degree_u=2) for N in (8, 16, 32)]
    eu, ep = [e["error_u"] for e in errs], [e["error_p"] for e in errs]
    hs = [e["h"] for e in errs]
    ru = [np.log(eu[i] / eu[i + 1]) / np.log(hs[i] / hs[i + 1]) for i in range(2)]
    rp = [np.log(ep[i] / ep[i + 1]) / np.log(hs[i] / hs[i + 1]) for i in range(2)]
    # the bounds of the oracle's own pin (tests/test_oracle_pins.py): better than 2nd / 1.5th order
    assert min(ru) > 2.0 and min(rp) > 1.5, (eu, ep, ru, rp)
    assert eu[-1] < 5e-4 and ep[-1] < 2e-3, (eu, ep)


def test_c2_taylor_green_64cubed_p1p1_against_the_c_port_on_its_own_mesh(hip):
    from oracle import cpu_baseline as CB
    from oracle import ipcs_oracle as O
    from tests.helpers import make_hip_problem

    N, nu, steps = 64, 0.01, 2
    dt = 0.005 * 32 / N
    ksp = {"ksp_rtol": 1e-10, "ksp_atol": 1e-30, "pc_type": "jacobi"}
    opts = {"tentative": dict(ksp, ksp_type="bcgs"), "pressure": dict(ksp, ksp_type="cg"),
            "scalar": dict(ksp, ksp_type="cg")}
    S, clock, mesh = make_hip_problem(3, N, 1, nu=nu, dt=dt, solver_options=opts, window=4096)
    assert mesh.num_cells == 1572864 and S._n_u == S._n_q == 274625 and S._Ap.pattern.nnz == 4018753  # BASELINE C2
    coords, cells = O.create_box_mesh([-1, -1, -1], [1, 1, 1], [N, N, N])
    cpu, x_v, x_q = CB.from_mesh(coords, cells, 1, 1, {"rtol": 1e-10, "atol": 1e-30, "max_it": 10000, "guess": False})
    assert cpu.ci.shape[0] == 4018753
    pv = _match(S._Vi[0][0].x.cpu().numpy(), x_v, [-1, -1, -1], [1, 1, 1])
    X = np.zeros((3, x_v.shape[0]))
    X[:] = x_v.T
    for i, f in enumerate((O.tg_u, O.tg_v, O.tg_w)):
        cpu.u2[i] = f(X, -dt, nu)
        cpu.u1[i] = f(X, 0.0, nu)
    cpu.p[:] = O.tg_p(X, -dt / 2.0, nu)
    Xb = X[:, cpu.bc_dofs]
    t = 0.0
    for _ in range(steps):
        t += dt
        clock["t"] = t
        S.solve(dt, nu, max_iter=1)
        cpu.step(dt, nu, np.stack([f(Xb, t, nu) for f in (O.tg_u, O.tg_v, O.tg_w)]))
    u = S.u.x.array.reshape(-1, 3)
    du = np.abs(u[pv] - cpu.u1.T).max()
    dp = np.abs(S._p.x.array[pv] - cpu.p).max()
    # both sides: Jacobi-BiCGStab / Jacobi-CG at rtol 1e-10 from a zero guess
    assert du < 1e-8 and dp < 1e-6, (du, dp)
    its_h, its_c = S.iteration_counts(), cpu.its
    assert abs(max(its_h["pressure"]) - its_c["pressure"][0]) <= 2, (its_h, its_c)


def test_c5_pattern_beyond_2_to_31_storage_slots(hip):
    """BASELINE.json configs[4] at its stated size on one GPU: 256^3 P2-P1, matrix-free (low_memory_version),
    135 005 697 dofs per velocity component, 3 867 809 793 / 253 036 801 nonzeros (SURVEY.md 8's table), int64
    entry offsets; exactness identities of M and K and one time step against the analytic field.  ~184 GiB."""
    from oracle import ipcs_oracle as O
    from tests.helpers import make_hip_problem

    gc.collect()
    torch.cuda.empty_cache()
    N, nu = 256, 0.01
    dt = 0.005 * 32 / N
    opts = {k: {"ksp_type": t, "pc_type": "jacobi", "ksp_rtol": 1e-8, "ksp_atol": 1e-14,
                "ksp_initial_guess_nonzero": True}
            for k, t in (("tentative", "bcgs"), ("pressure", "cg"), ("scalar", "cg"))}
    S, clock, mesh = make_hip_problem(3, N, u_deg=2, nu=nu, dt=dt, solver_options=opts, window=4096, low_memory=True)
    try:
        P = S._M.pattern
        assert P.size > 2 ** 31, P.size  # int64 entry offsets are exercised
        assert int(P.slice_ptr[-1].item()) == P.size
        assert mesh.num_cells == 6 * N ** 3
        assert S._n_u == (2 * N + 1) ** 3 and S._n_q == (N + 1) ** 3
        assert P.nnz == 230 * N ** 3 + 138 * N ** 2 + 24 * N + 1 == 3867809793
        assert S._Ap.pattern.nnz == 15 * N ** 3 + 21 * N ** 2 + 9 * N + 1 == 253036801
        n = S._n_u
        one = torch.ones(n, 1, dtype=torch.float64, device="cuda")
        y = torch.zeros_like(one)
        S._M.mult(one, y, 1)
        assert abs(float(y.sum()) - 8.0) < 1e-9  # sum_ij M_ij = |Omega|
        S._K.mult(one, y, 1)
        assert float(y.abs().max()) < 1e-8  # K 1 = 0
        del one, y
        clock["t"] += dt
        S.solve(dt, nu, max_iter=1)
        X3 = S._Vi[0][0].x[:n].cpu().numpy().T
        U = S._U.dev()[:n].cpu().numpy()
        for c, f in enumerate((O.tg_u, O.tg_v, O.tg_w)):
            assert np.abs(U[:, c] - f(X3, clock["t"], nu)).max() < 5e-5
    finally:
        del S
        gc.collect()
        torch.cuda.empty_cache()


def test_3d_beltrami_convergence_against_the_analytic_solution(hip):
    """An oracle-free pin of the whole 3-D path: the Ethier-Steinman Beltrami flow (all three velocity
    components and the pressure non-trivial, exact Dirichlet data) on N = 8, 16, 32 with dt ~ h.  The
    nodal velocity error against the ANALYTIC field must fall towards second order;
    nothing in this test comes from the numpy / C restatements."""
    import math

    import oasisx_amd as ox
    from oasisx_amd import mesh as M

    import bench  # the workload definitions of bench.py (analytic fields as array-API callables)

    errs = []
    for N in (8, 16, 32):
        W = bench.make_workload("beltrami", N, np, torch)
        nu, fns = W["nu"], W["fns"]
        p0, p1 = W["box"]
        T, steps = 0.04, N // 2
        dt = T / steps
        clock = {"t": 0.0}

        def on_boundary(x):
            on = np.zeros(x.shape[1], dtype=bool)
            for k in range(3):
                on |= np.isclose(x[k], p0[k]) | np.isclose(x[k], p1[k])
            return on

        mesh = M.create_box(None, [p0, p1], [N, N, N])
        bcs_u = [[ox.DirichletBC(lambda x, f=f: f(x, clock["t"]), ox.LocatorMethod.GEOMETRICAL, on_boundary)] for f in fns]
        ksp = {"pc_type": "jacobi", "ksp_rtol": 1e-11, "ksp_atol": 1e-30, "ksp_max_it": 10000}
        so = {"tentative": dict(ksp, ksp_type="bcgs"), "pressure": dict(ksp, ksp_type="cg"), "scalar": dict(ksp, ksp_type="cg")}
        S = ox.FractionalStep_AB_CN(mesh, ("Lagrange", 2), ("Lagrange", 1), bcs_u=bcs_u, bcs_p=[], solver_options=so)
        for i, f in enumerate(fns):
            S._u2[i].interpolate(lambda x, f=f: f(x, -dt))
            S._u1[i].interpolate(lambda x, f=f: f(x, 0.0))
        S._p.interpolate(lambda x: W["p"](x, -dt / 2.0))
        for _ in range(steps):
            clock["t"] += dt
            S.solve(dt, nu, max_iter=1)
        X = S._Vi[0][0].tabulate_dof_coordinates().T
        err = max(float(np.abs(S._u1[i].x.array - fns[i](X, clock["t"])).max()) for i in range(3))
        errs.append(err)
        del S, mesh
        gc.collect()
    rates = [math.log(errs[k] / errs[k + 1], 2.0) for k in range(2)]
    assert errs[2] < errs[1] < errs[0] and rates[0] > 1.6 and rates[1] > 1.8, (errs, rates)
