"""GPU: P3-P2 Taylor-Hood on TETRAHEDRA (round 5; reference fracstep.py:163-184 takes any Basix element, its demo's
``-u 3 -p 2``: demo/taylor_green.py:82-83,111).  The library builds the degree-3 space of a tetrahedral mesh -- 20 dofs per
cell: 4 vertices, two per edge ordered along the edge's global direction at the Gauss-Lobatto-Legendre points
(``gll_warped``), one per face at its centroid --; the row kernels run their <3, 3> instantiations on a 70-point
Grundmann-Moeller rule of degree 9 (csrc/fe_tables_h3.h); the convection rows are formed by quadrature at run time (the
tensor the P2 kernel keeps in LDS would take 192 KB); the pressure space is P2.

Checked: the space against the oracle's OWN numbering through the dof coordinates; exact reproduction of cubics; M, K, Ap,
the convection-diffusion matrix and the rectangular operators entry by entry; whole time steps (both
``low_memory_version`` branches) against the oracle; a step against the C port on its own mesh; the Beltrami errors
against the ANALYTIC solution."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problem(N, low_memory=True, solver_options=None, nu=0.01, dt=0.005, u_deg=3, p_deg=2):
    import oasisx_amd as ox
    from oracle import ipcs_oracle as O
    from tests.helpers import KRYLOV, on_boundary3, tg_mesh

    mesh = tg_mesh(3, N)
    clock = {"t": 0.0}
    fns = [O.tg_u, O.tg_v, O.tg_w]
    S = ox.FractionalStep_AB_CN(
        mesh, ("Lagrange", u_deg), ("Lagrange", p_deg), bcs_p=[], solver_options=solver_options or KRYLOV,
        bcs_u=[[ox.DirichletBC(lambda x, f=f: f(x, clock["t"], nu), ox.LocatorMethod.GEOMETRICAL, on_boundary3)] for f in fns],
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
    return O.taylor_green_problem(0, 3, u_deg=Vi.degree, p_deg=Q.degree, nu=nu, dt=dt, t0=0.0,
                                  solver_options=solver_options or KRYLOV, low_memory=low_memory,
                                  mesh=(mesh.coords.cpu().numpy(), Vi.cells_in_kernel_order()), vd=Vi.cell_dofs.cpu().numpy(),
                                  qd=Q.cell_dofs.cpu().numpy(), x_v=Vi.x.cpu().numpy(), x_q=Q.x.cpu().numpy())


def test_p3_space_on_tetrahedra_against_the_oracle_on_its_own_numbering(hip):
    import ctypes as C

    from oasisx_amd import _lib, fem
    from oasisx_amd.la import SellMatrix
    from oracle import ipcs_oracle as O
    from oracle.cpu_baseline import match_by_coordinates
    from tests.helpers import tg_mesh

    N = 3
    mesh = tg_mesh(3, N)
    V = fem.FunctionSpace(mesh, 3, window=128)
    coords, cells = O.create_box_mesh([-1, -1, -1], [1, 1, 1], [N, N, N])
    F = O.Forms(coords, cells, 3, 2)
    nv = (N + 1) ** 3
    ne = 3 * N * (N + 1) ** 2 + 3 * N * N * (N + 1) + N ** 3  # axis edges + face diagonals + the cubes' main diagonals
    nf = 12 * N ** 3 + 6 * N * N  # 6 tets x 4 faces = 24 per cube, interior faces shared: (24 N^3 + 2 * 6 N^2 ... ) / 2
    assert V.nd == 20 and V.num_dofs == F.nv == nv + 2 * ne + nf, (V.num_dofs, F.nv, nv, ne, nf)
    perm = match_by_coordinates(V.x.cpu().numpy(), F.x_v, np.array([-1.0] * 3), np.array([1.0] * 3))  # same point set
    # a cubic is reproduced exactly by nodal interpolation: orientation of the edge dofs, face dofs, node positions, basis
    f = lambda x: x[0] ** 3 - 2.0 * x[0] * x[1] * x[2] + x[2] ** 2 * x[1] - 0.5 * x[0] * x[1] + x[2]  # noqa: E731
    u = fem.Function(V)
    u.interpolate(f)
    assert fem.assemble_l2_error_sq(u, f) < 1e-25
    lib = _lib.load()
    geom = V.native.nmesh.geom
    cs = _lib.ox_cells(3, 0, int(geom.shape[0]), geom.data_ptr())
    adj = V.adj.struct()
    nb, bptr, bsl, bw = V.pattern.bins_args()
    for kind, ref in ((0, F.mass_v()), (1, F.stiffness_v())):
        for blocks in (False, True):
            A = SellMatrix(V.pattern)
            if blocks:
                nblk, bp, ent = V.pattern.blocks_args()
                _lib.check(lib.ox_assemble_matrix_blocks(kind, 3, C.byref(cs), _lib.ptr(V.cell_dofs), C.byref(adj),
                                                         _lib.ptr(V.adj.adj_pos), V.adj.pw, A.ref(), nblk, bp, ent,
                                                         _lib.current_stream()), "ox_assemble_matrix_blocks")
            else:
                _lib.check(lib.ox_assemble_matrix(kind, 3, C.byref(cs), _lib.ptr(V.cell_dofs), C.byref(adj), _lib.ptr(V.adj.adj_pos),
                                                  V.adj.pw, A.ref(), nb, bptr, bsl, bw, _lib.current_stream()), "ox_assemble_matrix")
            Ah = A.to_scipy()[perm][:, perm]
            assert Ah.nnz == ref.nnz and abs(Ah - ref).max() < 1e-12 * abs(ref).max(), (kind, blocks)
    # entity closures: the dofs on the boundary faces found topologically = those found geometrically
    from tests.helpers import on_boundary3

    from oasisx_amd import mesh as M

    x = V.x.cpu().numpy()
    geo = np.nonzero(on_boundary3(x.T))[0]
    for fdim in (2, 1, 0):  # faces carry vertex, edge AND face dofs; edges vertex and edge dofs; vertices themselves
        ents = M.locate_entities_boundary(mesh, fdim, on_boundary3)
        topo = V.entity_dofs(fdim, ents)
        assert np.isin(topo, geo).all()
        if fdim == 2:
            assert np.array_equal(np.sort(topo), np.sort(geo))
    cells_all = M.locate_entities(mesh, 3, lambda x_: np.full(x_.shape[1], True))
    assert np.array_equal(V.entity_dofs(3, cells_all), np.arange(V.num_dofs))  # the closure of all cells: every dof


@pytest.mark.parametrize("low_memory", [True, False])
def test_p3p2_tets_operators_and_steps_against_the_oracle(hip, low_memory):
    nu, dt = 0.01, 0.005
    S, clock, mesh = _problem(3, low_memory)
    R, rclock = _twin(S, mesh, nu, dt, low_memory=low_memory)
    assert S._Vi[0][0].degree == 3 and S._Q.degree == 2 and S._Vi[0][0].nd == 20
    for A_hip, A_or in ((S._M, R.M), (S._K, R.K), (S._Ap, R.Ap)):
        assert abs(A_hip.to_scipy() - A_or).max() < 1e-12 * abs(A_or).max()
    if not low_memory:
        for i in range(3):
            for Mat, ref in ((S._p_vdxi_Mat, R.P[i]), (S._grad_p_Mat, R.Gm[i]), (S._divu_Mat, R.D[i])):
                assert abs(Mat.to_scipy(i) - ref).max() < 1e-12 * max(abs(ref).max(), 1.0)
    t = 0.0
    for s in range(2):
        t += dt
        clock["t"] = rclock["t"] = t
        S.solve(dt, nu, max_iter=1)
        R.solve(dt, nu, max_iter=1)
        if s == 0:
            assert abs(S._A.to_scipy() - R.A).max() < 1e-11 * abs(R.A).max()  # convection included (degree-8 integrand)
            rhs1 = np.stack([f.x.array for f in S._rhs1], axis=1)
            assert np.abs(rhs1 - R.rhs1).max() < 1e-10 * np.abs(R.rhs1).max()
    u = S.u.x.array.reshape(-1, 3)
    assert np.abs(u - R.u1).max() < 1e-8 and np.abs(S._p.x.array - R.p).max() < 1e-7


def test_p3p2_tets_step_against_the_c_port_on_its_own_mesh(hip):
    """A second, independent witness: oracle/ipcs_cpu.c builds its own dof numbering, CSR patterns and operators from the
    mesh definition alone (reference tensors on the oracle's 125-point rule); fields matched through dof coordinates."""
    from oracle import cpu_baseline as CB
    from oracle import ipcs_oracle as O
    from tests.helpers import KRYLOV

    nu, dt, N, steps = 0.01, 0.005, 3, 2
    S, clock, mesh = _problem(N, True, KRYLOV, nu, dt)
    coords, cells = O.create_box_mesh([-1.0] * 3, [1.0] * 3, [N, N, N])
    cpu, x_v, x_q = CB.from_mesh(coords, cells, 3, 2, {"rtol": 1e-11, "atol": 1e-30, "max_it": 10000, "guess": False})
    lo, hi = np.array([-1.0] * 3), np.array([1.0] * 3)
    pv = CB.match_by_coordinates(S._Vi[0][0].x.cpu().numpy(), x_v, lo, hi)
    pq = CB.match_by_coordinates(S._Q.x.cpu().numpy(), x_q, lo, hi)
    assert cpu.nu_ == S._n_u and cpu.nq == S._n_q
    X, Xq = x_v.T.copy(), x_q.T.copy()
    fns = (O.tg_u, O.tg_v, O.tg_w)
    for i, f in enumerate(fns):
        cpu.u2[i] = f(X, -dt, nu)
        cpu.u1[i] = f(X, 0.0, nu)
    cpu.p[:] = O.tg_p(Xq, -dt / 2.0, nu)
    Xb = X[:, cpu.bc_dofs]
    t = 0.0
    for _ in range(steps):
        t += dt
        clock["t"] = t
        S.solve(dt, nu, max_iter=1)
        cpu.step(dt, nu, np.stack([f(Xb, t, nu) for f in fns]))
    u = S.u.x.array.reshape(-1, 3)
    assert np.abs(u[pv] - cpu.u1.T).max() < 1e-8 and np.abs(S._p.x.array[pq] - cpu.p).max() < 1e-7
    its_h, its_c = S.iteration_counts(), cpu.its
    assert abs(max(its_h["pressure"]) - its_c["pressure"][0]) <= 2, (its_h, its_c)


def test_p3p2_tets_beltrami_errors_against_the_analytic_solution(hip):
    """Nothing of the numpy / C restatements here: the device's P3-P2 fields on tetrahedra against the analytic
    Ethier-Steinman solution (nodal and L2 errors) on N = 2, 4, 8 with dt small enough for the splitting error to stay
    below the spatial one: the error falls from mesh to mesh and P3-P2 undercuts P2-P1 on the finest."""
    import math

    import oasisx_amd as ox
    from oasisx_amd import fem
    from tests.helpers import LU, on_boundary3, tg_mesh

    a, d_, nu = math.pi / 4.0, math.pi / 2.0, 0.01

    def comp(i, j, k):
        return lambda x, t: -a * (np.exp(a * x[i]) * np.sin(a * x[j] + d_ * x[k])
                                  + np.exp(a * x[k]) * np.cos(a * x[i] + d_ * x[j])) * math.exp(-nu * d_ * d_ * t)
    fns = [comp(0, 1, 2), comp(1, 2, 0), comp(2, 0, 1)]

    def pres(x, t):
        X, Y, Z = x[0], x[1], x[2]
        s = (np.exp(2 * a * X) + np.exp(2 * a * Y) + np.exp(2 * a * Z)
             + 2 * np.sin(a * X + d_ * Y) * np.cos(a * Z + d_ * X) * np.exp(a * (Y + Z))
             + 2 * np.sin(a * Y + d_ * Z) * np.cos(a * X + d_ * Y) * np.exp(a * (Z + X))
             + 2 * np.sin(a * Z + d_ * X) * np.cos(a * Y + d_ * Z) * np.exp(a * (X + Y)))
        return -0.5 * a * a * s * math.exp(-2.0 * nu * d_ * d_ * t)

    dt, steps = 1e-3, 5
    errs = {}
    for deg, Ns in (((3, 2), (2, 4, 8)), ((2, 1), (8,))):
        for N in Ns:
            mesh = tg_mesh(3, N)
            clock = {"t": 0.0}
            S = ox.FractionalStep_AB_CN(
                mesh, ("Lagrange", deg[0]), ("Lagrange", deg[1]), bcs_p=[], solver_options=LU,
                bcs_u=[[ox.DirichletBC(lambda x, f=f: f(x, clock["t"]), ox.LocatorMethod.GEOMETRICAL, on_boundary3)] for f in fns],
                options={"sell_window": 256})
            for i, f in enumerate(fns):
                S._u2[i].interpolate(lambda x, f=f: f(x, -dt))
                S._u1[i].interpolate(lambda x, f=f: f(x, 0.0))
            S._p.interpolate(lambda x: pres(x, -dt / 2.0))
            for _ in range(steps):
                clock["t"] += dt
                S.solve(dt, nu, max_iter=1)
            e2 = sum(fem.assemble_l2_error_sq(S._u[i], lambda x, f=fns[i]: f(x, clock["t"])) for i in range(3))
            errs[(deg, N)] = math.sqrt(e2)
    e = [errs[((3, 2), N)] for N in (2, 4, 8)]
    # (interpolation alone would give 16 per halving; from N = 4 on the splitting error of the scheme at this dt takes over)
    assert e[1] < e[0] / 6.0 and e[2] < e[1] / 2.5, (e, errs)
    assert errs[((3, 2), 8)] < 0.7 * errs[((2, 1), 8)], errs


def test_p3_tets_vtu_output(hip, tmp_path):
    """P3 fields on tetrahedra are written on VTK_LAGRANGE_TETRAHEDRON cells (type 71) with the P3 nodes as points: every
    cell's 20 points in VTK's order -- vertices, the edges 01, 12, 20, 03, 13, 23 (two points each, along the edge), the
    faces 013, 123, 023, 012."""
    import base64

    from oasisx_amd import fem, io
    from tests.helpers import tg_mesh

    mesh = tg_mesh(3, 2)
    V = fem.FunctionSpace(mesh, 3, window=128)
    u = fem.Function(V, "u")
    u.interpolate(lambda x: x[0] + 2.0 * x[1] - x[2])
    w = io.VTXWriter(None, str(tmp_path / "u.bp"), [u])
    w.write(0.0)
    w.close()
    X, conn, offsets, types = w._topology
    assert (types == 71).all() and conn.shape[0] == 20 * mesh.num_cells
    conn = conn.reshape(-1, 20)
    cv = mesh.coords.cpu().numpy()[mesh.cells[V.local_cells].cpu().numpy()]  # (nc, 4, 3) vertex coordinates, kernel cell order
    g0, g1 = 0.5 - 0.5 / math.sqrt(5.0), 0.5 + 0.5 / math.sqrt(5.0)
    P = X[conn]
    assert np.abs(P[:, :4] - cv).max() < 1e-14
    for k, (a, b) in enumerate(((0, 1), (1, 2), (2, 0), (0, 3), (1, 3), (2, 3))):
        assert np.abs(P[:, 4 + 2 * k] - ((1 - g0) * cv[:, a] + g0 * cv[:, b])).max() < 1e-14
        assert np.abs(P[:, 5 + 2 * k] - ((1 - g1) * cv[:, a] + g1 * cv[:, b])).max() < 1e-14
    for k, f in enumerate(((0, 1, 3), (1, 2, 3), (0, 2, 3), (0, 1, 2))):
        assert np.abs(P[:, 16 + k] - cv[:, list(f)].mean(axis=1)).max() < 1e-14
    assert base64  # (the file itself is exercised by tests/test_io.py's readers for the other elements)
