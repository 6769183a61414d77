"""GPU: the other BASELINE.json configuration shapes at reduced size, HIP path vs oracle.
C4: lid-driven cavity (constant, component-wise different Dirichlet data, start from rest)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dim,N,deg", [(2, 10, 2), (3, 4, 2), (3, 5, 1)])
def test_lid_driven_cavity_steps_match_oracle(hip, dim, N, deg):
    import oasisx_amd as ox
    from oasisx_amd import mesh as M
    from oracle import ipcs_oracle as O
    from tests.helpers import KRYLOV

    nu, dt, steps = 1e-3 if dim == 3 else 1e-2, 1.0 / (4 * N), 3
    mesh = (M.create_unit_square(None, N, N) if dim == 2 else M.create_unit_cube(None, N, N, N))
    top = dim - 1  # the lid is the face x_top = 1

    def lid(x):
        return np.isclose(x[top], 1.0)

    def walls(x):
        on = np.zeros(x.shape[1], dtype=bool)
        for k in range(dim):
            on |= np.isclose(x[k], 0.0) | np.isclose(x[k], 1.0)
        return on & ~lid(x)

    def allb(x):
        return walls(x) | lid(x)

    G = ox.LocatorMethod.GEOMETRICAL
    bcs = [[ox.DirichletBC(1.0, G, lid), ox.DirichletBC(0.0, G, walls)]] + \
          [[ox.DirichletBC(0.0, G, allb)] for _ in range(dim - 1)]
    S = ox.FractionalStep_AB_CN(mesh, ("Lagrange", deg), ("Lagrange", 1), bcs_u=bcs, bcs_p=[],
                                solver_options=KRYLOV, options={"sell_window": 128})
    Vi, Q = S._Vi[0][0], S._Q
    F = O.Forms(mesh.coords.cpu().numpy(), Vi.cells_in_kernel_order(), deg, 1, vd=Vi.cell_dofs.cpu().numpy(),
                qd=Q.cell_dofs.cpu().numpy(), nv_dofs=Vi.num_dofs, nq_dofs=Q.num_dofs)
    xv = Vi.x.cpu().numpy()
    X = np.zeros((3, xv.shape[0]))
    X[:dim] = xv.T
    dl, dw, da = (np.nonzero(f(X))[0] for f in (lid, walls, allb))
    obcs = [[O.DirichletData(dl, 1.0), O.DirichletData(dw, 0.0)]] + [[O.DirichletData(da, 0.0)] for _ in range(dim - 1)]
    R = O.OracleFractionalStep(F, xv, Q.x.cpu().numpy(), obcs, solver_options=KRYLOV)
    # A cold start (u = 0) makes the first BiCGStab residual live on the boundary rows only and the
    # method breaks down exactly (rho = rhat.r = 0) -- in PETSc's KSPBCGS as well; see the test below.
    # Start from a small smooth interior field instead.
    bump = 0.05 * np.prod([np.sin(np.pi * X[k]) for k in range(dim)], axis=0)
    for i in range(dim):
        for a, b in ((S._u1, R.u1), (S._u2, R.u2)):
            a[i].x.array[:] = bump * (i + 1)
            b[:, i] = bump * (i + 1)
    for _ in range(steps):
        S.solve(dt, nu, max_iter=1)
        R.solve(dt, nu, max_iter=1)
    u = S.u.x.array.reshape(-1, dim)
    assert np.abs(R.u1).max() > 0.5  # the lid drives the flow
    assert np.abs(u - R.u1).max() < 1e-8 and np.abs(S._p.x.array - R.p).max() < 1e-6
    # inner iterations (max_iter > 1): same fixed point iteration, same diff
    d1 = S.solve(dt, nu, max_iter=3, max_error=1e-9)
    d2 = R.solve(dt, nu, max_iter=3, max_error=1e-9)
    assert abs(d1 - d2) < 1e-7 * max(1.0, abs(d2))
    assert np.abs(S.u.x.array.reshape(-1, dim) - R.u1).max() < 1e-7


def test_cold_start_bicgstab_breakdown_is_reported_like_petsc(hip):
    """u = 0, un-lifted identity-row BCs: r0 is supported on boundary rows, after one step
    rho = rhat.r = 0 exactly.  PETSc's KSPBCGS returns DIVERGED_BREAKDOWN (-5); so do the oracle
    and the HIP solver (solve() then raises through its assert, as the reference's would)."""
    import oasisx_amd as ox
    from oasisx_amd import _lib
    from oasisx_amd import mesh as M
    from tests.helpers import KRYLOV

    mesh = M.create_unit_square(None, 6, 6)
    G = ox.LocatorMethod.GEOMETRICAL
    lid = lambda x: np.isclose(x[1], 1.0)  # noqa: E731
    allb = lambda x: np.isclose(x[0], 0) | np.isclose(x[0], 1) | np.isclose(x[1], 0) | np.isclose(x[1], 1)  # noqa: E731
    bcs = [[ox.DirichletBC(0.0, G, allb), ox.DirichletBC(1.0, G, lid)], [ox.DirichletBC(0.0, G, allb)]]
    S = ox.FractionalStep_AB_CN(mesh, ("Lagrange", 2), ("Lagrange", 1), bcs_u=bcs, bcs_p=[], solver_options=KRYLOV)
    S.assemble_first(0.05, 0.01)
    S.velocity_tentative_assemble()
    diff, errors = S.velocity_tentative_solve()
    assert errors[0] == _lib.DIVERGED_BREAKDOWN and errors[1] == _lib.CONVERGED_ATOL
    with pytest.raises(AssertionError):
        S.solve(0.05, 0.01)


@pytest.mark.parametrize("dim,N", [(2, 8), (3, 4)])
def test_cold_start_with_direct_solver_options(hip, dim, N):
    """The reference demo's options (preonly + lu everywhere) from rest: a direct solver cannot break
    down, so its Krylov stand-in restarts instead (ksp_bcgs_restarts) and matches the oracle's LU."""
    import oasisx_amd as ox
    from oasisx_amd import mesh as M
    from oracle import ipcs_oracle as O
    from tests.helpers import LU

    nu, dt = 1e-2, 0.05
    mesh = (M.create_unit_square(None, N, N) if dim == 2 else M.create_unit_cube(None, N, N, N))
    top = dim - 1
    lid = lambda x: np.isclose(x[top], 1.0)  # noqa: E731

    def allb(x):
        on = np.zeros(x.shape[1], dtype=bool)
        for k in range(dim):
            on |= np.isclose(x[k], 0.0) | np.isclose(x[k], 1.0)
        return on

    G = ox.LocatorMethod.GEOMETRICAL
    bcs = [[ox.DirichletBC(0.0, G, allb), ox.DirichletBC(1.0, G, lid)]] + [[ox.DirichletBC(0.0, G, allb)] for _ in range(dim - 1)]
    S = ox.FractionalStep_AB_CN(mesh, ("Lagrange", 2), ("Lagrange", 1), bcs_u=bcs, bcs_p=[], solver_options=LU,
                                options={"sell_window": 128})
    Vi, Q = S._Vi[0][0], S._Q
    F = O.Forms(mesh.coords.cpu().numpy(), Vi.cells_in_kernel_order(), 2, 1, vd=Vi.cell_dofs.cpu().numpy(),
                qd=Q.cell_dofs.cpu().numpy(), nv_dofs=Vi.num_dofs, nq_dofs=Q.num_dofs)
    xv = Vi.x.cpu().numpy()
    X = np.zeros((3, xv.shape[0]))
    X[:dim] = xv.T
    da, dl = np.nonzero(allb(X))[0], np.nonzero(lid(X))[0]
    obcs = [[O.DirichletData(da, 0.0), O.DirichletData(dl, 1.0)]] + [[O.DirichletData(da, 0.0)] for _ in range(dim - 1)]
    R = O.OracleFractionalStep(F, xv, Q.x.cpu().numpy(), obcs, solver_options=LU)
    for _ in range(2):
        S.solve(dt, nu, max_iter=1)
        R.solve(dt, nu, max_iter=1)
    assert np.abs(R.u1).max() > 0.5
    assert np.abs(S.u.x.array.reshape(-1, dim) - R.u1).max() < 1e-7
    assert np.abs(S._p.x.array - R.p).max() < 1e-5 * max(1.0, np.abs(R.p).max())
