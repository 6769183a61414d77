"""GPU: P3-P2 Taylor-Hood on triangles -- the reference demo's ``-u 3 -p 2`` (demo/taylor_green.py:82-83,111; spaces:
fracstep.py:163-184 with ``lagrange_variant=gll_warped``).  The library builds the degree-3 space (two dofs per edge
ordered along the edge's global direction, one per cell; edge nodes at the Gauss-Lobatto-Legendre points), the row
kernels run their <2, 3> instantiations on a degree-9 rule (csrc/fe_tables_h.h), the pressure space is P2.

Checked: the space against the oracle's OWN numbering through the dof coordinates; M, K, Ap, the convection-diffusion
matrix and the rectangular operators entry by entry; whole time steps (both ``low_memory_version`` branches) against the
oracle; the Taylor-Green errors against the ANALYTIC solution fall from mesh to mesh and undercut P2-P1's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problem(N, low_memory=True, solver_options=None, nu=0.01, dt=0.005, u_deg=3, p_deg=2):
    import oasisx_amd as ox
    from oracle import ipcs_oracle as O
    from tests.helpers import KRYLOV, on_boundary, tg_mesh

    mesh = tg_mesh(2, N)
    clock = {"t": 0.0}
    fns = [O.tg_u, O.tg_v]
    S = ox.FractionalStep_AB_CN(
        mesh, ("Lagrange", u_deg), ("Lagrange", p_deg), bcs_p=[], solver_options=solver_options or KRYLOV,
        bcs_u=[[ox.DirichletBC(lambda x, f=f: f(x, clock["t"], nu), ox.LocatorMethod.GEOMETRICAL, on_boundary)] for f in fns],
        options={"sell_window": 256, "low_memory_version": low_memory})
    for i, f in enumerate(fns):
        S._u2[i].interpolate(lambda x, f=f: f(x, -dt, nu))
        S._u1[i].interpolate(lambda x, f=f: f(x, 0.0, nu))
    S._p.interpolate(lambda x: O.tg_p(x, -dt / 2.0, nu))
    return S, clock, mesh


def _twin(S, mesh, nu, dt, solver_options=None, low_memory=True):
    from oracle import ipcs_oracle as O
    from tests.helpers import KRYLOV

    Vi, Q = S._Vi[0][0], S._Q
    return O.taylor_green_problem(0, 2, u_deg=Vi.degree, p_deg=Q.degree, nu=nu, dt=dt, t0=0.0,
                                  solver_options=solver_options or KRYLOV, low_memory=low_memory,
                                  mesh=(mesh.coords.cpu().numpy(), Vi.cells_in_kernel_order()), vd=Vi.cell_dofs.cpu().numpy(),
                                  qd=Q.cell_dofs.cpu().numpy(), x_v=Vi.x.cpu().numpy(), x_q=Q.x.cpu().numpy())


def test_p3_space_against_the_oracle_on_its_own_numbering(hip):
    from oasisx_amd import fem
    from oracle import ipcs_oracle as O
    from oracle.cpu_baseline import match_by_coordinates
    from tests.helpers import tg_mesh

    N = 7
    mesh = tg_mesh(2, N)
    V = fem.FunctionSpace(mesh, 3, window=128)
    coords, cells = O.create_rectangle_mesh([-1, -1], [1, 1], [N, N])
    F = O.Forms(coords, cells, 3, 2)
    nv, ne, nc = (N + 1) ** 2, 3 * N * N + 2 * N, 2 * N * N
    assert V.num_dofs == F.nv == nv + 2 * ne + nc and V.nd == 10
    perm = match_by_coordinates(V.x.cpu().numpy(), F.x_v, np.array([-1.0, -1.0]), np.array([1.0, 1.0]))  # same point set
    # a cubic is reproduced exactly by nodal interpolation: orientation of the edge dofs, node positions, basis
    f = lambda x: x[0] ** 3 - 2.0 * x[0] * x[1] ** 2 + x[1] - 0.5 * x[0] * x[1]  # noqa: E731
    u = fem.Function(V)
    u.interpolate(f)
    assert fem.assemble_l2_error_sq(u, f) < 1e-26
    # operators of the library (own numbering) against the oracle's (its numbering), matched through coordinates
    import ctypes as C

    from oasisx_amd import _lib
    from oasisx_amd.la import SellMatrix

    lib = _lib.load()
    geom = V.native.nmesh.geom
    cs = _lib.ox_cells(2, 0, int(geom.shape[0]), geom.data_ptr())
    adj = V.adj.struct()
    nb, bptr, bsl, bw = V.pattern.bins_args()
    for kind, ref in ((0, F.mass_v()), (1, F.stiffness_v())):
        A = SellMatrix(V.pattern)
        _lib.check(lib.ox_assemble_matrix(kind, 3, C.byref(cs), _lib.ptr(V.cell_dofs), C.byref(adj), _lib.ptr(V.adj.adj_pos),
                                          V.adj.pw, A.ref(), nb, bptr, bsl, bw, _lib.current_stream()), "ox_assemble_matrix")
        Ah = A.to_scipy()[perm][:, perm]
        assert Ah.nnz == ref.nnz and abs(Ah - ref).max() < 1e-13 * abs(ref).max()


@pytest.mark.parametrize("low_memory", [True, False])
def test_p3p2_operators_and_steps_against_the_oracle(hip, low_memory):
    nu, dt = 0.01, 0.005
    S, clock, mesh = _problem(6, low_memory)
    R, rclock = _twin(S, mesh, nu, dt, low_memory=low_memory)
    assert S._Vi[0][0].degree == 3 and S._Q.degree == 2
    for A_hip, A_or in ((S._M, R.M), (S._K, R.K), (S._Ap, R.Ap)):
        assert abs(A_hip.to_scipy() - A_or).max() < 1e-13 * abs(A_or).max()
    if not low_memory:
        for i in range(2):
            for Mat, ref in ((S._p_vdxi_Mat, R.P[i]), (S._grad_p_Mat, R.Gm[i]), (S._divu_Mat, R.D[i])):
                assert abs(Mat.to_scipy(i) - ref).max() < 1e-13 * max(abs(ref).max(), 1.0)
    t = 0.0
    for s in range(3):
        t += dt
        clock["t"] = rclock["t"] = t
        S.solve(dt, nu, max_iter=1)
        R.solve(dt, nu, max_iter=1)
        if s == 0:
            assert abs(S._A.to_scipy() - R.A).max() < 1e-12 * abs(R.A).max()  # convection included (degree-8 integrand)
            rhs1 = np.stack([f.x.array for f in S._rhs1], axis=1)
            assert np.abs(rhs1 - R.rhs1).max() < 1e-11 * np.abs(R.rhs1).max()
    u = S.u.x.array.reshape(-1, 2)
    assert np.abs(u - R.u1).max() < 1e-8 and np.abs(S._p.x.array - R.p).max() < 1e-7


def test_p3p2_taylor_green_errors_against_the_analytic_solution(hip):
    """Nothing of the numpy / C restatements here: the device's P3-P2 fields against the analytic Taylor-Green solution
    (space-time L2 norms as demo/taylor_green.py:225-226) on N = 4, 8, 16, and P2-P1 on the finest mesh beside it."""
    from oasisx_amd import fem
    from oracle import ipcs_oracle as O
    from tests.helpers import LU

    nu, dt, T = 0.01, 0.002, 0.02
    errs = {}
    for deg, Ns in (((3, 2), (4, 8, 16)), ((2, 1), (16,))):
        for N in Ns:
            S, clock, mesh = _problem(N, True, LU, nu, dt, deg[0], deg[1])
            t, eu2, ep2 = 0.0, 0.0, 0.0
            for _ in range(int(round(T / dt))):
                t += dt
                clock["t"] = t
                S.solve(dt, nu, max_iter=1)
                eu2 += dt * sum(fem.assemble_l2_error_sq(S._u[i], lambda x, f=f: f(x, t, nu)) for i, f in enumerate((O.tg_u, O.tg_v)))
                ep2 += dt * fem.assemble_l2_error_sq(S._p, lambda x: O.tg_p(x, t - dt / 2.0, nu))
            errs[(deg, N)] = (np.sqrt(eu2), np.sqrt(ep2))
    e4, e8, e16 = (errs[((3, 2), N)] for N in (4, 8, 16))
    assert e16[0] < e8[0] < e4[0] and e16[1] < e8[1] < e4[1], errs
    assert np.log2(e8[0] / e16[0]) > 1.8 and np.log2(e8[1] / e16[1]) > 2.0, errs
    assert e16[0] < 0.5 * errs[((2, 1), 16)][0] and e16[1] < 0.5 * errs[((2, 1), 16)][1], errs


def test_p3_fields_written_on_vtk_lagrange_triangles(hip, tmp_path):
    """VTXWriter on a degree-3 field (the demo's output with -u 3): VTK_LAGRANGE_TRIANGLE cells whose node order is
    VTK's -- corners, then the two nodes of edge 01 from 0 to 1, of 12 from 1 to 2, of 20 from 2 to 0, then the centroid."""
    from oasisx_amd import fem, io
    from tests.helpers import tg_mesh

    mesh = tg_mesh(2, 5)
    V = fem.FunctionSpace(mesh, 3, window=128)
    W = fem.VectorFunctionSpace(V, 2)
    u = fem.Function(W, name="u")
    u.interpolate(lambda x: np.stack([x[0] ** 3 - x[1], 2.0 * x[0] * x[1]]))
    w = io.VTXWriter(mesh.comm, str(tmp_path / "u.bp"), [u], engine="BP4")
    w.write(0.25)
    w.close()
    d = io.read_vtu(str(tmp_path / "u_000000.vtu"))
    assert int(d["types"][0]) == 69
    conn = d["connectivity"].reshape(-1, 10)
    assert conn.shape[0] == mesh.num_cells and np.unique(conn).shape[0] == V.num_dofs
    P = d["points"][:, :2]
    t0, t1 = fem.GLL3
    for j, (a, b) in enumerate(((0, 1), (1, 2), (2, 0))):
        for k, t in enumerate((t0, t1)):
            np.testing.assert_allclose(P[conn[:, 3 + 2 * j + k]], (1 - t) * P[conn[:, a]] + t * P[conn[:, b]], atol=1e-14)
    np.testing.assert_allclose(P[conn[:, 9]], P[conn[:, :3]].mean(axis=1), atol=1e-14)
    np.testing.assert_allclose(d["point_data"]["u"][:, 0], P[:, 0] ** 3 - P[:, 1], atol=1e-13)


def test_p3p2_steps_against_the_c_port_on_its_own_mesh(hip):
    """The second witness: oracle/ipcs_cpu.c (generic in the element tensors) set up from the mesh definition alone --
    its own P3 / P2 numbering, patterns and operators, Jacobi-BiCGStab / Jacobi-CG like the device -- against the device's
    P3-P2 steps through the dof coordinates."""
    from oracle import cpu_baseline as CB
    from oracle import ipcs_oracle as O
    from oracle.cpu_baseline import match_by_coordinates

    N, nu, dt, steps = 20, 0.01, 0.004, 3
    ksp = {"ksp_rtol": 1e-11, "ksp_atol": 1e-30, "pc_type": "jacobi"}
    opts = {"tentative": dict(ksp, ksp_type="bcgs"), "pressure": dict(ksp, ksp_type="cg"), "scalar": dict(ksp, ksp_type="cg")}
    S, clock, mesh = _problem(N, True, opts, nu, dt)
    coords, cells = O.create_rectangle_mesh([-1, -1], [1, 1], [N, N])
    cpu, x_v, x_q = CB.from_mesh(coords, cells, 3, 2, {"rtol": 1e-11, "atol": 1e-30, "max_it": 10000, "guess": False})
    lo, hi = np.array([-1.0, -1.0]), np.array([1.0, 1.0])
    pv = match_by_coordinates(S._Vi[0][0].x.cpu().numpy(), x_v, lo, hi)
    pq = match_by_coordinates(S._Q.x.cpu().numpy(), x_q, lo, hi)
    assert cpu.nu_ == S._n_u == (N + 1) ** 2 + 2 * (3 * N * N + 2 * N) + 2 * N * N and cpu.nq == S._n_q
    X = np.zeros((3, x_v.shape[0]))
    X[:2] = x_v.T
    Xq = np.zeros((3, x_q.shape[0]))
    Xq[:2] = x_q.T
    for i, f in enumerate((O.tg_u, O.tg_v)):
        cpu.u2[i] = f(X, -dt, nu)
        cpu.u1[i] = f(X, 0.0, nu)
    cpu.p[:] = O.tg_p(Xq, -dt / 2.0, nu)
    Xb = X[:, cpu.bc_dofs]
    t = 0.0
    for _ in range(steps):
        t += dt
        clock["t"] = t
        S.solve(dt, nu, max_iter=1)
        cpu.step(dt, nu, np.stack([f(Xb, t, nu) for f in (O.tg_u, O.tg_v)]))
    u = S.u.x.array.reshape(-1, 2)
    assert np.abs(u[pv] - cpu.u1.T).max() < 1e-8 and np.abs(S._p.x.array[pq] - cpu.p).max() < 1e-7
    its_h, its_c = S.iteration_counts(), cpu.its
    assert abs(max(its_h["pressure"]) - its_c["pressure"][0]) <= 2, (its_h, its_c)
