"""Parity of the HIP path against the CPU oracle -- every call goes through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _spaces(dim, N, deg, window=256):
    from oasisx_amd import fem
    from tests.helpers import tg_mesh

    mesh = tg_mesh(dim, N)
    return mesh, fem.FunctionSpace(mesh, deg, window=window)


@pytest.mark.parametrize("dim,N,deg", [(2, 9, 1), (2, 7, 2), (3, 4, 1), (3, 3, 2)])
@pytest.mark.parametrize("ncomp", [1, 2, 3])
def test_spmv_matches_scipy(hip, dim, N, deg, ncomp):
    """S1: Mat.mult on the SELL-64 layout == scipy CSR mat-vec of the same matrix."""
    import torch

    from oasisx_amd.la import SellMatrix
    from oracle import ipcs_oracle as O

    mesh, V = _spaces(dim, N, deg)
    F = O.Forms(mesh.coords.cpu().numpy(), V.cells_in_kernel_order(), deg, 1, vd=V.cell_dofs.cpu().numpy(),
                qd=V.cells_in_kernel_order(), nv_dofs=V.num_dofs, nq_dofs=mesh.num_vertices)
    rng = np.random.default_rng(0)
    Kc = F.stiffness_v() + 0.3 * F.convection(rng.standard_normal((V.num_dofs, dim)))
    A = SellMatrix(V.pattern)
    A.vals.copy_(V.pattern.values_from_csr(Kc))
    x = rng.standard_normal((V.num_dofs, ncomp))
    xd = torch.from_numpy(x).cuda()
    yd = torch.zeros_like(xd)
    A.mult(xd, yd, ncomp)
    ref = Kc @ x
    assert np.abs(yd.cpu().numpy() - ref).max() <= 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("dim,N,deg", [(2, 6, 1), (2, 5, 2), (3, 3, 1), (3, 3, 2)])
def test_preassembled_matrices(hip, dim, N, deg):
    """A1/A2/A3: mass, stiffness and pressure Laplacian equal the oracle's assembly."""
    from tests.helpers import make_hip_problem, make_oracle_twin

    S, clock, mesh = make_hip_problem(dim, N, deg)
    R, _ = make_oracle_twin(S, mesh, dim, deg)
    for name, mine, ref in (("M", S._M, R.M), ("K", S._K, R.K), ("Ap", S._Ap, R.Ap)):
        d = abs(mine.to_scipy() - ref).max()
        assert d <= 1e-13 * abs(ref).max(), (name, d)
    assert abs(S._vol - R.vol) < 1e-12
    assert np.abs(S._wQ.cpu().numpy() - R.wq).max() < 1e-14


@pytest.mark.parametrize("dim,N,deg", [(2, 6, 1), (2, 5, 2), (3, 3, 1), (3, 3, 2)])
@pytest.mark.parametrize("body_force", [None, (0.3, -0.1, 0.2)])
def test_assemble_first_and_tentative_rhs(hip, dim, N, deg, body_force):
    """a4/a5: A (with identity BC rows), b_first and rhs1 equal the oracle's
    (the restated reference test/test_tentative_velocity.py:231-235, matrices included)."""
    from tests.helpers import make_hip_problem, make_oracle_twin

    bf = None if body_force is None else body_force[:dim]
    dt, nu = 0.1, 0.5
    S, clock, mesh = make_hip_problem(dim, N, deg, nu=nu, dt=dt, body_force=bf)
    R, rc = make_oracle_twin(S, mesh, dim, deg, nu=nu, dt=dt)
    if bf is not None:
        R.b0[:] = np.stack([R.F.body_force_vec(float(f)) for f in bf], axis=1)
    rng = np.random.default_rng(1)
    ps = rng.standard_normal(S._n_q)
    S._ps.x.array[:] = ps
    R.ps[:] = ps
    clock["t"] = rc["t"] = dt
    for bcl in S._bcs_u:
        for bc in bcl:
            bc.update_bc()
    for bcl in R.bcs_u:
        for bc in bcl:
            bc.update(R.x_v)
    S.assemble_first(dt, nu)
    R.assemble_first(dt, nu)
    dA = abs(S._A.to_scipy() - R.A).max()
    assert dA <= 1e-12 * abs(R.A).max(), dA
    bfirst = np.stack([f.x.array for f in S._b_first], axis=1)
    assert np.abs(bfirst - R.b_first).max() <= 1e-12 * np.abs(R.b_first).max()
    S.velocity_tentative_assemble()
    R.velocity_tentative_assemble()
    for i in range(dim):
        for bc in S._bcs_u[i]:
            bc.apply(S._rhs1[i].x)
        for bc in R.bcs_u[i]:
            bc.apply(R.rhs1[:, i])
    rhs1 = np.stack([f.x.array for f in S._rhs1], axis=1)
    assert np.abs(rhs1 - R.rhs1).max() <= 1e-12 * np.abs(R.rhs1).max()


@pytest.mark.parametrize("ksp", ["cg", "bcgs"])
@pytest.mark.parametrize("ncomp", [1, 3])
def test_krylov_matches_oracle_iterations(hip, ksp, ncomp):
    """K1-K3: same algorithm, same conventions -> same iteration counts and solutions."""
    import torch

    from oasisx_amd import _lib
    from oasisx_amd.fem import FieldStorage
    from oasisx_amd.ksp import KSPSolver
    from oasisx_amd.la import SellMatrix
    from oracle import ipcs_oracle as O

    mesh, V = _spaces(3, 4, 2)
    F = O.Forms(mesh.coords.cpu().numpy(), V.cells_in_kernel_order(), 2, 1, vd=V.cell_dofs.cpu().numpy(),
                qd=V.cells_in_kernel_order(), nv_dofs=V.num_dofs, nq_dofs=mesh.num_vertices)
    rng = np.random.default_rng(3)
    Amat = F.mass_v() * 50.0 + F.stiffness_v()
    if ksp == "bcgs":
        Amat = Amat + 0.5 * F.convection(rng.standard_normal((V.num_dofs, 3)))
    A = SellMatrix(V.pattern, symmetric=(ksp == "cg"))
    A.vals.copy_(V.pattern.values_from_csr(Amat))
    B = FieldStorage(V.num_dofs, ncomp, "cuda")
    X = FieldStorage(V.num_dofs, ncomp, "cuda")
    b = rng.standard_normal((V.num_dofs, ncomp))
    B.host()[:] = b
    opts = {"ksp_type": ksp, "pc_type": "jacobi", "ksp_rtol": 1e-10, "ksp_atol": 1e-30}
    s = KSPSolver(None, opts)
    s.setOperators(A)
    reasons = s.solve_block(B, X)
    fn = O.jacobi_cg if ksp == "cg" else O.jacobi_bicgstab
    for c in range(ncomp):
        xr, reason, its, rn = fn(Amat.tocsr(), b[:, c], None, 1e-10, 1e-30, 10000)
        assert reasons[c] == reason == _lib.CONVERGED_RTOL
        assert abs(s.iterations[c] - its) <= 1, (s.iterations, its)
        assert np.abs(X.host()[:, c] - xr).max() <= 1e-8 * np.abs(xr).max()


@pytest.mark.parametrize("dim,N,deg", [(2, 8, 2), (2, 8, 1), (3, 4, 2), (3, 4, 1)])
def test_full_steps_match_oracle_krylov(hip, dim, N, deg):
    """a10: two full IPCS steps, HIP vs oracle with the same Krylov settings."""
    from tests.helpers import run_tg_pair

    r = run_tg_pair(dim, N, deg, steps=2)
    assert r["du"] < 1e-8 and r["dp"] < 1e-7, (r["du"], r["dp"], r["its_hip"], r["its_oracle"])


def test_full_steps_match_oracle_lu(hip):
    """The demo's configuration (LU everywhere, reference demo/taylor_green.py:117-121):
    HIP path (tight Krylov) vs oracle (sparse LU)."""
    from tests.helpers import LU, run_tg_pair

    r = run_tg_pair(2, 8, 2, steps=3, hip_options=LU, oracle_options=LU)
    assert r["du"] < 1e-8 and r["dp"] < 1e-7, (r["du"], r["dp"])


@pytest.mark.parametrize("dim,N", [(2, 8), (3, 4)])
def test_rotational_pressure_update(hip, dim, N):
    """rotational=True (reference fracstep.py:237-251,593-602): ps = Proj_Q(p + dp - xi nu div u)."""
    from tests.helpers import run_tg_pair

    r = run_tg_pair(dim, N, 2, steps=2, rotational=True)
    assert r["du"] < 1e-8 and r["dp"] < 1e-7, (r["du"], r["dp"])
    plain = run_tg_pair(dim, N, 2, steps=2, rotational=False)
    assert np.abs(r["R"].p - plain["R"].p).max() > 1e-6  # the rotational term does change p


def test_projector_of_a_function(hip):
    """Projector(Function, space): the L2 projection of a field of the space is the field
    (reference test/test_projector.py checks exactness for representable targets, atol 1e-12)."""
    from oasisx_amd import Projector, fem
    from tests.helpers import tg_mesh

    mesh = tg_mesh(2, 10)
    V = fem.FunctionSpace(mesh, 2, window=128)
    u = fem.Function(V)
    u.interpolate(lambda x: x[0] * x[0] + 3 * x[1] + 2 * x[1] * x[1])
    proj = Projector(u, V, [], petsc_options={"ksp_type": "preonly", "pc_type": "lu"})
    assert proj.solve() > 0
    assert np.abs(proj.x.x.array - u.x.array).max() < 1e-10
    u.interpolate(lambda x: x[0] + 2 * x[1] * x[1])
    proj.assemble_rhs()
    assert proj.solve(assemble_rhs=False) > 0
    assert np.abs(proj.x.x.array - u.x.array).max() < 1e-10


@pytest.mark.parametrize("dim,deg,on_device", [(2, 1, False), (2, 2, True), (3, 1, True), (3, 2, False)])
def test_projector_of_an_expression(hip, dim, deg, on_device):
    """Projector(callable, space): reference test/test_projector.py projects x[0]**degree onto the degree-`degree`
    Lagrange space and compares with the interpolant (atol 1e-12 there); also a non-polynomial integrand
    against the numpy oracle's load vector, and the same bits on a second run (ordered sums, no atomics)."""
    import torch

    from oasisx_amd import Projector, fem
    from tests.helpers import tg_mesh

    mesh = tg_mesh(dim, 6 if dim == 3 else 12)
    V = fem.FunctionSpace(mesh, deg, window=128)

    def poly(x):
        return x[0] ** deg + 2.0 * x[1] - (x[dim - 1] ** deg if deg == 2 else x[dim - 1])

    poly.supports_torch = on_device
    proj = Projector(poly, V, [], petsc_options={"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-14, "ksp_atol": 1e-30})
    assert proj.solve() > 0
    u = fem.Function(V)
    u.interpolate(lambda x: x[0] ** deg + 2.0 * x[1] - (x[dim - 1] ** deg if deg == 2 else x[dim - 1]))
    assert np.abs(proj.x.x.array - u.x.array).max() < 1e-10
    # a smooth non-polynomial integrand: the load vector against a host evaluation with a finer rule
    from oasisx_amd.function import load_vector

    def g(x):
        xp = torch if torch.is_tensor(x) else np
        return xp.sin(2.0 * x[0]) * xp.exp(x[1])

    g.supports_torch = on_device
    b1 = load_vector(V, g, proj._geom, 6)
    b2 = load_vector(V, g, proj._geom, 6)
    assert torch.equal(b1, b2)
    b_fine = load_vector(V, g, proj._geom, 10)
    assert float((b1 - b_fine).abs().max()) < 1e-9 * max(float(b_fine.abs().max()), 1e-30) + 1e-13
    # sum_i b_i = int g dx (partition of unity)
    ref = float(b_fine.sum())
    assert abs(float(b1.sum()) - ref) < 1e-10 * max(abs(ref), 1.0)


def test_dirichlet_values_evaluated_on_the_device(hip):
    """A callable marked supports_torch gets device coordinates; the imposed values equal the
    numpy path's (reference test/test_bcs.py: apply == set_bc with the interpolated function)."""
    import torch

    from oasisx_amd import DirichletBC, LocatorMethod, fem
    from tests.helpers import tg_mesh

    mesh = tg_mesh(3, 4)
    V = fem.FunctionSpace(mesh, 2, window=128)
    clock = {"t": 0.3}

    def f(x):
        xp = torch if torch.is_tensor(x) else np
        return xp.sin(x[0] + clock["t"]) * x[1] + x[2] ** 2

    def fdev(x):
        return f(x)

    fdev.supports_torch = True
    marker = lambda x: np.isclose(np.abs(x[2]), 1.0)  # noqa: E731
    a, b = DirichletBC(f, LocatorMethod.GEOMETRICAL, marker), DirichletBC(fdev, LocatorMethod.GEOMETRICAL, marker)
    a.create_bc(V)
    b.create_bc(V)
    for t in (0.1, 0.2):
        clock["t"] = t
        a.update_bc()
        b.update_bc()
        assert np.abs(a.values_host() - b.values_host()).max() < 1e-14
    S = fem.FieldStorage(V.num_dofs, 3, "cuda")
    u = fem.Function(V, "u", S, 1)
    b.apply(u.x)
    X = V.tabulate_dof_coordinates()
    exp = np.zeros(V.num_dofs)
    exp[b._dofs] = f(X[b._dofs].T)
    assert np.abs(S.host()[:, 1] - exp).max() < 1e-14 and np.abs(S.host()[:, 0]).max() == 0.0


@pytest.mark.parametrize("ksp", ["cg", "bcgs"])
@pytest.mark.parametrize("guess", [False, True])
def test_lockstep_solve_narrows_to_the_last_live_column(hip, ksp, guess):
    """Columns that converge at very different iteration counts: the solve continues on a compact
    single column once the others are done; every column still equals its own oracle solve."""
    from oasisx_amd import _lib
    from oasisx_amd.fem import FieldStorage
    from oasisx_amd.ksp import KSPSolver
    from oasisx_amd.la import SellMatrix
    from oracle import ipcs_oracle as O

    mesh, V = _spaces(3, 5, 2)
    F = O.Forms(mesh.coords.cpu().numpy(), V.cells_in_kernel_order(), 2, 1, vd=V.cell_dofs.cpu().numpy(),
                qd=V.cells_in_kernel_order(), nv_dofs=V.num_dofs, nq_dofs=mesh.num_vertices)
    rng = np.random.default_rng(5)
    Amat = (F.mass_v() * 20.0 + F.stiffness_v()).tocsr()
    if ksp == "bcgs":
        Amat = (Amat + 0.5 * F.convection(rng.standard_normal((V.num_dofs, 3)))).tocsr()
    A = SellMatrix(V.pattern, symmetric=(ksp == "cg"))
    A.vals.copy_(V.pattern.values_from_csr(Amat))
    n = V.num_dofs
    b = np.zeros((n, 3))
    b[:, 0] = Amat @ np.ones(n) * 1e-3      # easy: the solution is a constant
    b[:, 2] = rng.standard_normal(n)         # hard: rough right-hand side
    x0 = np.zeros((n, 3))
    if guess:
        x0[:, 0], x0[:, 2] = 1e-3, 0.1 * rng.standard_normal(n)
    B, X = FieldStorage(n, 3, "cuda"), FieldStorage(n, 3, "cuda")
    B.host()[:] = b
    X.host()[:] = x0
    s = KSPSolver(None, {"ksp_type": ksp, "pc_type": "jacobi", "ksp_rtol": 1e-10, "ksp_atol": 1e-12,
                         "ksp_initial_guess_nonzero": guess})
    s.setOperators(A)
    reasons = s.solve_block(B, X)
    fn = O.jacobi_cg if ksp == "cg" else O.jacobi_bicgstab
    its = s.iterations
    for c in range(3):
        xr, reason, it_ref, _ = fn(Amat, b[:, c], x0[:, c].copy() if guess else None, 1e-10, 1e-12, 10000)
        assert reasons[c] == reason and reasons[c] > 0, (c, reasons, reason)
        assert abs(its[c] - it_ref) <= 1, (c, its, it_ref)
        assert np.abs(X.host()[:, c] - xr).max() <= 1e-8 * max(1e-3, np.abs(xr).max())
    assert its[2] > its[0] + 8 and its[1] == 0  # the narrowing path was taken


@pytest.mark.parametrize("dim,N,deg", [(2, 6, 1), (2, 5, 2), (3, 3, 1), (3, 3, 2)])
def test_preassembled_rectangular_operators(hip, dim, N, deg):
    """low_memory_version=False (reference fracstep.py:311-315,332-336,348-352,392-404): the
    3*gdim rectangular operators equal the oracle's, and steps run through them."""
    from tests.helpers import make_hip_problem, make_oracle_twin, run_tg_pair

    S, clock, mesh = make_hip_problem(dim, N, deg, low_memory=False)
    R, _ = make_oracle_twin(S, mesh, dim, deg)
    for i in range(dim):
        for mine, ref in ((S._p_vdxi_Mat, R.F.p_vdxi_mat(i)), (S._grad_p_Mat, R.F.grad_p_mat(i)),
                          (S._divu_Mat, R.F.divu_mat(i))):
            d = abs(mine.to_scipy(i) - ref).max()
            assert d <= 1e-13 * max(1.0, abs(ref).max()), (mine.name, i, d)
    r = run_tg_pair(dim, N, deg, steps=2, low_memory=False)
    assert r["du"] < 1e-8 and r["dp"] < 1e-7, (r["du"], r["dp"])
    # and the two variants of the HIP path agree with each other to round-off x solver tolerance
    r2 = run_tg_pair(dim, N, deg, steps=2, low_memory=True)
    u1 = r["S"].u.x.array
    u2 = r2["S"].u.x.array
    assert np.abs(u1 - u2).max() < 1e-8


def test_kspsolver_public_solve_api(hip):
    """KSPSolver(comm, options, prefix).setOperators(A); solve(b.x.petsc_vec, x) -> reason, on a
    stand-alone field and on one column of an interleaved block (reference ksp.py:14-78)."""
    from oasisx_amd import _lib, fem
    from oasisx_amd.ksp import KSPSolver
    from oasisx_amd.la import SellMatrix
    from oracle import ipcs_oracle as O

    mesh, V = _spaces(2, 8, 2)
    F = O.Forms(mesh.coords.cpu().numpy(), V.cells_in_kernel_order(), 2, 1, vd=V.cell_dofs.cpu().numpy(),
                qd=V.cells_in_kernel_order(), nv_dofs=V.num_dofs, nq_dofs=mesh.num_vertices)
    Amat = (F.mass_v() * 10 + F.stiffness_v()).tocsr()
    A = SellMatrix(V.pattern, symmetric=True)
    A.vals.copy_(V.pattern.values_from_csr(Amat))
    import scipy.sparse.linalg as spla

    rng = np.random.default_rng(7)
    bvec = rng.standard_normal(V.num_dofs)
    ref = spla.spsolve(Amat.tocsc(), bvec)
    s = KSPSolver(mesh.comm, {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-12}, prefix="mine")
    s.setOperators(A)
    s.setOptions(A)
    b, x = fem.Function(V), fem.Function(V)
    b.x.array[:] = bvec
    assert s.solve(b.x.petsc_vec, x) == _lib.CONVERGED_RTOL
    assert np.abs(x.x.array - ref).max() < 1e-9
    # one column of a 2-component block
    S = fem.FieldStorage(V.num_dofs, 2, "cuda")
    B = fem.FieldStorage(V.num_dofs, 2, "cuda")
    x1, b1 = fem.Function(V, "x1", S, 1), fem.Function(V, "b1", B, 1)
    b1.x.array[:] = bvec
    s.updateOptions({"ksp_type": "preonly", "pc_type": "lu"})
    assert s.solve(b1.x.petsc_vec, x1) == _lib.CONVERGED_ITS
    assert np.abs(S.host()[:, 1] - ref).max() < 1e-9 and np.abs(S.host()[:, 0]).max() == 0.0
    assert len(s.iterations) == 4
