"""GPU: the reference's own projector test (test/test_projector.py:16-50) restated: the gradient of a P2 field
projected into the discontinuous P1 vector space, exact to 1e-12 in L2, and again after the source field has
changed and the right-hand side has been re-assembled (``assemble_rhs()`` + ``solve(assemble_rhs=False)``)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _l2_error(W, ph, exact):
    """sqrt(int |exact - ph|^2 dx) of a DG1 vector field: ph is linear per cell, exact a callable x -> (gdim, npts);
    degree-4 Gauss-Jacobi rule per cell (exact for the linear-minus-linear integrands of the test)."""
    from oasisx_amd.fem import _simplex_rule

    mesh = W.mesh
    d = mesh.gdim
    bary, w = _simplex_rule(d, 3)
    xc = mesh.coords[mesh.cells[W.local_cells]].cpu().numpy()  # (nc, d+1, d)
    vals = ph.x.array.reshape(xc.shape[0], d + 1, d)         # (nc, vertex, component)
    xq = np.einsum("qa,eak->eqk", bary, xc)
    pq = np.einsum("qa,eak->eqk", bary, vals)
    ex = np.stack(exact(xq.reshape(-1, d).T), axis=1).reshape(xq.shape)
    detj = np.abs(np.linalg.det(xc[:, 1:, :] - xc[:, :1, :]))
    return float(np.sqrt(np.einsum("e,q,eqk->", detj, w, (ex - pq) ** 2)))


@pytest.mark.parametrize("dim,N", [(2, 10), (3, 4)])
def test_projector(hip, dim, N):
    import oasisx_amd as ox
    from oasisx_amd import fem
    from oasisx_amd import mesh as M
    from oasisx_amd.function import grad

    mesh = M.create_unit_square(None, N, N) if dim == 2 else M.create_unit_cube(None, N, N, N)
    V = fem.FunctionSpace(mesh, 2)
    u = fem.Function(V)
    u.interpolate(lambda x: x[0] * x[0] + 3 * x[1] + 2 * x[1] * x[1] + (0.5 * x[2] * x[0] if dim == 3 else 0.0))
    W = fem.DGSpace(mesh, 1, shape=(mesh.geometry.dim,))
    petsc_options = {"ksp_type": "preonly", "pc_type": "lu", "pc_factor_mat_solver_type": "mumps"}
    gradient_projector = ox.Projector(grad(u), W, [], petsc_options=petsc_options)
    assert gradient_projector.solve() > 0
    ph = gradient_projector.x
    if dim == 2:
        u_ex = lambda x: (2 * x[0], 3 + 4 * x[1])  # noqa: E731
    else:
        u_ex = lambda x: (2 * x[0] + 0.5 * x[2], 3 + 4 * x[1], 0.5 * x[0])  # noqa: E731
    assert np.isclose(_l2_error(W, ph, u_ex), 0.0, atol=1e-12)

    u.interpolate(lambda x: x[0] + 2 * x[1] * x[1])
    # the projector still holds the old right-hand side until it is re-assembled
    assert not np.isclose(_l2_error(W, ph, (lambda x: (1 + 0 * x[0], 4 * x[1]) + ((0 * x[0],) if dim == 3 else ()))), 0.0,
                          atol=1e-6)
    gradient_projector.assemble_rhs()
    gradient_projector.solve(assemble_rhs=False)
    u_ex_new = (lambda x: (1 + 0 * x[0], 4 * x[1])) if dim == 2 else (lambda x: (1 + 0 * x[0], 4 * x[1], 0 * x[0]))
    assert np.isclose(_l2_error(W, ph, u_ex_new), 0.0, atol=1e-12)
    # M (M^-1 b) = b: the two directions of the block-diagonal mass matrix agree
    import torch
    from oasisx_amd import _lib
    import ctypes as C

    back = torch.zeros_like(gradient_projector._B.dev())
    _lib.check(_lib.load().ox_dg1_mass(0, C.byref(gradient_projector._cells), W.dim, gradient_projector._X.ptr(),
                                       _lib.ptr(back), _lib.current_stream()), "ox_dg1_mass")
    b = gradient_projector._B.dev()
    assert float((back - b).abs().max()) <= 1e-14 * max(float(b.abs().max()), 1.0)


def _mesh(dim, N):
    from oasisx_amd import mesh as M

    return M.create_unit_square(None, N, N) if dim == 2 else M.create_unit_cube(None, N, N, N)


@pytest.mark.parametrize("dim,N", [(2, 7), (3, 3)])
def test_projector_of_a_pointwise_expression_of_fields(hip, dim, N):
    """``Projector(Expression(fn, u, grad(u), w), V)``: what a UFL expression of Functions, their gradients and the
    coordinates is to the reference's projector (function.py:75).  The fields are interpolated polynomials, so the
    integrand is a known polynomial of x: the right-hand side and the solution must equal those of the same
    projector given that polynomial as a callable of x (an independent evaluation path)."""
    import oasisx_amd as ox
    from oasisx_amd import fem
    from oasisx_amd.function import Expression, grad

    mesh = _mesh(dim, N)
    V, V1 = fem.FunctionSpace(mesh, 2), fem.FunctionSpace(mesh, 1)
    u, w = fem.Function(V), fem.Function(V1)
    uf = lambda x: x[0] * x[0] + 3 * x[1] + 2 * x[1] * x[1] + 0.5 * x[2] * x[0]  # noqa: E731
    wf = lambda x: 1.0 - 0.25 * x[0] + 2.0 * x[1] + x[2]  # noqa: E731
    u.interpolate(uf)
    w.interpolate(wf)
    opts = {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-14, "ksp_atol": 1e-30}
    md = {"quadrature_degree": 8}
    ex = Expression(lambda x, u_, gu, w_: u_ * w_ + gu[0] - 2.0 * gu[1] * x[1], u, grad(u), w)
    an = lambda x: uf(x) * wf(x) + (2 * x[0] + 0.5 * x[2]) - 2.0 * (3 + 4 * x[1]) * x[1]  # noqa: E731
    pe = ox.Projector(ex, V, petsc_options=opts, metadata=md)
    pa = ox.Projector(an, V, petsc_options=opts, metadata=md)
    assert pe.solve() > 0 and pa.solve() > 0
    b, ba = pe._b.x.array, pa._b.x.array
    assert np.abs(b - ba).max() < 1e-13 * max(1.0, np.abs(ba).max())
    assert np.abs(pe.x.x.array - pa.x.x.array).max() < 1e-10
    # the fields change, the expression follows on re-assembly (as a UFL form of Functions does)
    u.interpolate(lambda x: 1.0 + x[1])
    pe.assemble_rhs()
    an2 = lambda x: (1.0 + x[1]) * wf(x) + 0.0 - 2.0 * 1.0 * x[1]  # noqa: E731
    pa2 = ox.Projector(an2, V, petsc_options=opts, metadata=md)
    pa2.assemble_rhs()
    assert np.abs(pe._b.x.array - pa2._b.x.array).max() < 1e-13
    with pytest.raises(TypeError):
        Expression(lambda x: x[0], 3.0)
    with pytest.raises(ValueError):  # a vector-valued fn for a scalar target
        ox.Projector(Expression(lambda x, g: g, grad(u)), V, petsc_options=opts).assemble_rhs()


@pytest.mark.parametrize("dim,N", [(2, 6), (3, 3)])
def test_projector_into_a_blocked_lagrange_space(hip, dim, N):
    """A blocked target (the reference's projector takes any space): one mass matrix, ``dim`` right-hand sides in one
    block solve.  Column by column it must equal the scalar projector; a blocked Function projects onto itself."""
    import oasisx_amd as ox
    from oasisx_amd import fem
    from oasisx_amd.function import Expression, grad

    mesh = _mesh(dim, N)
    V = fem.FunctionSpace(mesh, 2)
    W = fem.VectorFunctionSpace(V, dim)
    u = fem.Function(V)
    u.interpolate(lambda x: x[0] * x[0] + 3 * x[1] + 2 * x[1] * x[1] + 0.5 * x[2] * x[0])
    opts = {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-14, "ksp_atol": 1e-30}
    # grad(u) of a P2 field is P1 per cell but discontinuous: its continuous projection, against the scalar path
    pv = ox.Projector(Expression(lambda x, g: g, grad(u)), W, petsc_options=opts)
    assert pv.solve() > 0
    xv = pv.x.x.array.reshape(-1, dim)
    for c in range(dim):
        ps = ox.Projector(Expression(lambda x, g, c=c: g[c], grad(u)), V, petsc_options=opts)
        assert ps.solve() > 0
        assert np.abs(ps._b.x.array - pv._b.x.array.reshape(-1, dim)[:, c]).max() < 1e-14
        assert np.abs(ps.x.x.array - xv[:, c]).max() < 1e-11
    # interior of the domain aside, a globally quadratic field has a continuous linear gradient: reproduced exactly
    X = V.tabulate_dof_coordinates()
    exact = np.stack([2 * X[:, 0] + 0.5 * X[:, 2], 3 + 4 * X[:, 1], 0.5 * X[:, 0]], axis=1)[:, :dim]
    assert np.abs(xv - exact).max() < 1e-10
    # a sequence of rows works like a tensor; a blocked Function of the space projects onto itself
    pt = ox.Projector(Expression(lambda x, g: tuple(g[c] for c in range(dim)), grad(u)), W, petsc_options=opts)
    pt.assemble_rhs()
    assert np.array_equal(pt._b.x.array, pv._b.x.array)
    U = fem.Function(W)
    U.interpolate(lambda x: np.stack([x[0] * x[1], 1.0 - x[1] * x[1], x[0] + x[2]][:dim]))
    pu = ox.Projector(U, W, petsc_options=opts)
    assert pu.solve() > 0
    assert np.abs(pu.x.x.array - U.x.array).max() < 1e-11
    # the gradient of a blocked field: (dim, gdim, npts); its divergence into the scalar space
    pd = ox.Projector(Expression(lambda x, G: sum(G[c, c] for c in range(dim)), grad(U)), V, petsc_options=opts)
    pa = ox.Projector(lambda x: x[1] - 2.0 * x[1] + (1.0 if dim == 3 else 0.0), V, petsc_options=opts)
    pd.assemble_rhs(), pa.assemble_rhs()
    assert np.abs(pd._b.x.array - pa._b.x.array).max() < 1e-14


@pytest.mark.parametrize("dim,N", [(2, 6), (3, 3)])
def test_projector_of_an_expression_into_dg1(hip, dim, N):
    """DG1 targets: ``Expression(lambda x, g: g, grad(u))`` must give the right-hand side of the dedicated
    ``grad(u)`` kernel (ox_dg1_grad_rhs) and the reference test's exact answer; a scalar DG1 target reproduces a P1
    field vertex by vertex."""
    import oasisx_amd as ox
    from oasisx_amd import fem
    from oasisx_amd.function import Expression, grad

    mesh = _mesh(dim, N)
    V = fem.FunctionSpace(mesh, 2)
    u = fem.Function(V)
    u.interpolate(lambda x: x[0] * x[0] + 3 * x[1] + 2 * x[1] * x[1] + 0.5 * x[2] * x[0])
    W = fem.DGSpace(mesh, 1, shape=(dim,))
    pg = ox.Projector(grad(u), W, [])
    pe = ox.Projector(Expression(lambda x, g: g, grad(u)), W, [])
    assert pg.solve() > 0 and pe.solve() > 0
    bg, be = pg._B.rhost(), pe._B.rhost()
    assert np.abs(bg - be).max() < 1e-14 * max(1.0, np.abs(bg).max())
    u_ex = (lambda x: (2 * x[0], 3 + 4 * x[1])) if dim == 2 else (lambda x: (2 * x[0] + 0.5 * x[2], 3 + 4 * x[1], 0.5 * x[0]))
    assert np.isclose(_l2_error(W, pe.x, u_ex), 0.0, atol=1e-12)
    # scalar DG1 target, an expression of a P1 field and of x
    V1 = fem.FunctionSpace(mesh, 1)
    w = fem.Function(V1)
    w.interpolate(lambda x: 1.0 - 0.25 * x[0] + 2.0 * x[1])
    W0 = fem.DGSpace(mesh, 1)
    p0 = ox.Projector(Expression(lambda x, w_: 2.0 * w_ + x[0], w), W0, [])
    assert p0.solve() > 0
    Xd = W0.tabulate_dof_coordinates()
    assert np.abs(p0.x.x.array - (2.0 * (1.0 - 0.25 * Xd[:, 0] + 2.0 * Xd[:, 1]) + Xd[:, 0])).max() < 1e-12


def test_body_force_as_an_expression_of_a_field(hip):
    """fracstep.py:284-289: a force component may be any UFL expression -- here a buoyancy-like ``beta * T(x) * g`` of a
    P2 temperature field.  The assembled b0 must equal that of the same force written as a callable of x."""
    import oasisx_amd as ox
    from oasisx_amd import fem
    from oasisx_amd.function import Expression

    mesh = _mesh(2, 8)
    T = fem.Function(fem.FunctionSpace(mesh, 2))
    Tf = lambda x: 1.0 + x[0] * x[1] - 0.5 * x[1] * x[1]  # noqa: E731
    T.interpolate(Tf)

    def make(force):
        bcs_u = [[ox.DirichletBC(0.0, ox.LocatorMethod.GEOMETRICAL, lambda x: np.isclose(x[0], 0.0))] for _ in range(2)]
        return ox.FractionalStep_AB_CN(mesh, ("Lagrange", 2), ("Lagrange", 1), bcs_u, [], body_force=force,
                                       options={"body_force_quadrature_degree": 8})

    fe = make((0.0, Expression(lambda x, t: -9.81 * 0.1 * t, T)))
    fa = make((0.0, lambda x: -9.81 * 0.1 * Tf(x)))
    be, ba = fe._B0.rhost(), fa._B0.rhost()
    assert np.abs(ba[:, 1]).max() > 1e-4 and np.abs(be[:, 0]).max() == 0.0
    assert np.abs(be - ba).max() < 1e-15 + 1e-13 * np.abs(ba).max()
