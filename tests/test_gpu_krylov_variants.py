"""GPU: the single-reduction CG (Chronopoulos-Gear recurrences, ONE merged reduction per iteration,
PETSc's -ksp_cg_single_reduction; the default of KSPSolver here) against the standard recurrences
(-ksp_cg_single_reduction false) and the oracle's PETSc-convention CG: same solution to solver
tolerance, iteration counts within +-2 (the two recurrences differ in rounding only), same converged
reasons, lock-step columns and the narrowed continuation included."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _system(dim, N, deg, mass=3.0):
    from oasisx_amd import fem
    from oasisx_amd import mesh as M
    from oasisx_amd.la import SellMatrix
    from oracle import ipcs_oracle as O

    mesh = (M.create_rectangle(None, [[-1.0, -1.0], [1.0, 1.0]], [N, N]) if dim == 2
            else M.create_box(None, [[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]], [N, N, N]))
    V = fem.FunctionSpace(mesh, deg, window=256)
    F = O.Forms(mesh.coords.cpu().numpy(), V.cells_in_kernel_order(), deg, 1, vd=V.cell_dofs.cpu().numpy(),
                qd=V.cells_in_kernel_order(), nv_dofs=V.num_dofs, nq_dofs=mesh.num_vertices)
    Acsr = (F.stiffness_v() + mass * F.mass_v()).tocsr()
    A = SellMatrix(V.pattern, symmetric=True)
    A.vals.copy_(V.pattern.values_from_csr(Acsr))
    A.version += 1
    return V, A, Acsr


@pytest.mark.parametrize("dim,N,deg,nc", [(2, 24, 2, 1), (3, 8, 2, 3), (3, 10, 1, 2)])
def test_single_reduction_cg_matches_standard_cg_and_oracle(hip, dim, N, deg, nc):
    from oasisx_amd.fem import FieldStorage
    from oasisx_amd.ksp import KSPSolver
    from oracle import ipcs_oracle as O

    V, A, Acsr = _system(dim, N, deg)
    n = V.num_dofs
    x = V.x.cpu().numpy()
    cols = [np.cos(2.0 * x[:, 0]) * (1.0 + x[:, 1]), 1e-3 * np.sin(5.0 * x[:, 0] * x[:, -1]), np.exp(x[:, 1])][:nc]
    B = FieldStorage(n, nc, "cuda")
    B.dev()[:] = torch.from_numpy(np.stack(cols, axis=1)).cuda()
    out = {}
    for single in (True, False):
        ksp = KSPSolver(None, {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-10, "ksp_atol": 1e-50,
                               "ksp_cg_single_reduction": single})
        ksp.setOperators(A)
        X = FieldStorage(n, nc, "cuda")
        reasons = ksp.solve_block(B, X)
        out[single] = (X.dev().cpu().numpy().copy(), ksp.iterations[:nc], reasons)
    for c in range(nc):
        sol, reason, its, _ = O.jacobi_cg(Acsr, cols[c], rtol=1e-10, atol=1e-50)
        for single in (True, False):
            xs, it, rs = out[single]
            assert rs[c] == reason == 2  # KSP_CONVERGED_RTOL
            assert abs(it[c] - its) <= (2 if single else 1), (single, it, its)
            assert np.abs(xs[:, c] - sol).max() < 1e-8 * max(np.abs(sol).max(), 1.0)
    assert np.abs(out[True][0] - out[False][0]).max() < 1e-8 * np.abs(out[False][0]).max()


def test_single_reduction_cg_nonzero_guess_and_max_it(hip):
    from oasisx_amd.fem import FieldStorage
    from oasisx_amd.ksp import KSPSolver
    from oracle import ipcs_oracle as O

    V, A, Acsr = _system(2, 16, 2)
    n = V.num_dofs
    b = np.sin(3.0 * V.x.cpu().numpy()[:, 0])
    B = FieldStorage(n, 1, "cuda")
    B.dev()[:, 0] = torch.from_numpy(b).cuda()
    sol, _, its0, _ = O.jacobi_cg(Acsr, b, rtol=1e-10, atol=1e-50)
    # a good initial guess needs fewer iterations and reaches the same answer
    ksp = KSPSolver(None, {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-10, "ksp_initial_guess_nonzero": True})
    ksp.setOperators(A)
    X = FieldStorage(n, 1, "cuda")
    X.dev()[:, 0] = torch.from_numpy(sol * (1.0 + 1e-4)).cuda()
    assert ksp.solve_block(B, X)[0] == 2 and ksp.iterations[0] < its0
    assert np.abs(X.dev()[:, 0].cpu().numpy() - sol).max() < 1e-8
    # iteration limit: DIVERGED_ITS after exactly max_it iterations
    ksp = KSPSolver(None, {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-14, "ksp_max_it": 5})
    ksp.setOperators(A)
    X = FieldStorage(n, 1, "cuda")
    assert ksp.solve_block(B, X)[0] == -3 and ksp.iterations[0] == 5
    # ... and x is the fifth iterate (its last update is applied by the host once the device reports the end)
    for single in (False, True):
        ksp = KSPSolver(None, {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-14, "ksp_max_it": 5,
                               "ksp_cg_single_reduction": single})
        ksp.setOperators(A)
        X = FieldStorage(n, 1, "cuda")
        assert ksp.solve_block(B, X)[0] == -3
        x5, r5, i5, _ = O.jacobi_cg(Acsr, b, rtol=1e-14, atol=1e-50, max_it=5)
        assert r5 == -3 and i5 == 5
        assert np.abs(X.dev()[:, 0].cpu().numpy() - x5).max() < 1e-12 * max(np.abs(x5).max(), 1.0)


@pytest.mark.parametrize("kind,single", [("cg", False), ("cg", True), ("bcgs", False)])
def test_supplied_first_matvec_gives_the_same_iterates(hip, kind, single):
    """ox_ksp_solve_ax0: with A x0 handed in (the velocity update has M u* from its right-hand side,
    the tentative solve gets A u1 from the fused assembly) the solver skips its first mat-vec and
    produces the same iterates bit for bit."""
    from oasisx_amd.fem import FieldStorage
    from oasisx_amd.ksp import KSPSolver

    V, A, _ = _system(3, 8, 2)
    n, nc = V.num_dofs, 3
    g = torch.Generator(device="cuda").manual_seed(11)
    rhs = torch.randn(n, nc, dtype=torch.float64, device="cuda", generator=g)
    x0 = torch.randn(n, nc, dtype=torch.float64, device="cuda", generator=g)
    res = []
    for give in (False, True):
        ks = KSPSolver(None, {"ksp_type": kind, "pc_type": "jacobi", "ksp_rtol": 1e-9, "ksp_atol": 1e-30,
                              "ksp_initial_guess_nonzero": True, "ksp_cg_single_reduction": single})
        ks.setOperators(A)
        B, X, AX = FieldStorage(n, nc, "cuda"), FieldStorage(n, nc, "cuda"), FieldStorage(n, nc, "cuda")
        B.dev().copy_(rhs)
        X.dev().copy_(x0)
        if give:
            A.mult(X.dev(), AX.dev(), nc)
        reasons = ks.solve_block(B, X, ax0=AX if give else None)
        assert all(r > 0 for r in reasons)
        res.append((X.dev().clone(), ks.iterations))
    assert res[0][1] == res[1][1] and torch.equal(res[0][0], res[1][0])


def test_fused_assembly_returns_the_first_matvec_of_the_tentative_solve(hip):
    """ox_assemble_first_au: (A @ u1) from the epilogue of the fused kernel equals A.mult(u1) bit for
    bit (identity rows included), and the time step that uses it equals the one that does not."""
    from tests.helpers import KRYLOV, make_hip_problem

    opts = {k: dict(v, ksp_initial_guess_nonzero=True) for k, v in KRYLOV.items()}
    S, clock, mesh = make_hip_problem(3, 6, u_deg=2, solver_options=opts)
    dt, nu = 0.005, 0.01
    for _ in range(2):  # the second step starts from u == u1
        clock["t"] += dt
        S.solve(dt, nu)
    clock["t"] += dt
    for bcu in S._bcs_u:
        for bc in bcu:
            bc.update_bc()
    S.assemble_first(dt, nu)
    assert S._AU1_valid and torch.equal(S._U.rdev(), S._U1.rdev())
    ref = torch.zeros_like(S._B3.dev())
    S._A.mult(S._U1.rdev(), ref, S._gdim)
    assert torch.equal(ref, S._B3.dev())
    # the whole step with and without the shortcut (read-only views: the token must survive until the solve)
    S.velocity_tentative_assemble()
    u_before = S._U.rdev().clone()
    assert S._u_is_u1 == (S._U.generation, S._U1.generation)
    spy = {}
    solve_block = S._solver_u.solve_block
    S._solver_u.solve_block = lambda B, X, ax0=None: (spy.update(ax0=ax0), solve_block(B, X, ax0=ax0))[1]
    _, err = S.velocity_tentative_solve()
    assert spy["ax0"] is S._B3  # the shortcut was really taken
    its_with, u_with = list(S._solver_u.iterations), S._U.dev().clone()
    S._U.dev().copy_(u_before)
    S._AU1_valid = False
    _, err2 = S.velocity_tentative_solve()
    assert (err > 0).all() and (err2 > 0).all()
    assert spy["ax0"] is None
    assert its_with == list(S._solver_u.iterations) and torch.equal(u_with, S._U.dev())


@pytest.mark.parametrize("how", ["bc.apply", "dev()", "host", "ksp.solve"])
def test_a_write_to_u_between_two_steps_drops_the_first_matvec_shortcut(hip, how):
    """A u1 from the fused assembly may stand in for the tentative solve's first mat-vec only while u
    still equals u1 bit for bit.  Every way of writing u (or u1) from outside between two steps -- DirichletBC.apply on
    the field, a device view, the host array, a KSPSolver.solve into it -- must drop the shortcut, and the step must
    equal the one computed without it."""
    from tests.helpers import KRYLOV, make_hip_problem

    opts = {k: dict(v, ksp_initial_guess_nonzero=True) for k, v in KRYLOV.items()}
    dt, nu = 0.005, 0.01
    out = []
    for force_off in (False, True):
        S, clock, mesh = make_hip_problem(3, 5, u_deg=2, solver_options=opts)
        for _ in range(2):
            clock["t"] += dt
            S.solve(dt, nu)
        # an outside write to u: a different initial guess for the next tentative solve (u != u1 from here on)
        if how == "bc.apply":
            clock["t"] += 7 * dt  # other Dirichlet values than the ones u carries
            S._bcs_u[0][0].update_bc()
            S._bcs_u[0][0].apply(S._u[0].x)
            clock["t"] -= 7 * dt
        elif how == "dev()":
            S._U.dev()[::3, 1] *= 1.25
        elif how == "host":
            S._u[2].x.array[::5] += 0.125
        else:
            S._solver_c.solve(S._rhs1[1], S._u[1])
        spy = {}
        solve_block = S._solver_u.solve_block
        S._solver_u.solve_block = lambda B, X, ax0=None, f=solve_block, spy=spy: (spy.update(ax0=ax0), f(B, X, ax0=ax0))[1]
        if force_off:
            af = S.assemble_first
            S.assemble_first = lambda *a, af=af, S=S: (af(*a), setattr(S, "_AU1_valid", False))[0]
        clock["t"] += dt
        S.solve(dt, nu)
        assert spy["ax0"] is None, how  # no stale A u1 as the product of another initial guess
        out.append((S._U.rdev().clone(), list(S._solver_u.iterations)))
    assert out[0][1] == out[1][1] and torch.equal(out[0][0], out[1][0])


@pytest.mark.parametrize("dim,N,deg,nc", [(2, 24, 2, 1), (3, 8, 2, 3), (3, 10, 1, 2), (3, 16, 2, 3)])
def test_merged_reduction_bicgstab_matches_standard_bicgstab_and_oracle(hip, dim, N, deg, nc):
    """OX_KSP_BCGS_MERGED (two synchronisation points per iteration: rhat.v, then {t.t, t.s, rhat.s, rhat.t, s.s}
    with |r| by recurrence) against the standard three-point BiCGStab and the oracle's PETSc-convention BiCGStab
    (reference ksp.py:76 at fracstep.py:521): same converged reason, iteration counts within +-2, same solution to
    solver tolerance; lock-step columns of very different scale, the narrowed continuation and a zero
    right-hand side column (converged at once: its deferred x update must add nothing) included."""
    from oasisx_amd.fem import FieldStorage
    from oasisx_amd.ksp import KSPSolver
    from oracle import ipcs_oracle as O

    # M/dt-dominated, mildly non-symmetric: the tentative-velocity matrix's character (10-30 iterations; on
    # ill-conditioned systems BiCGStab's count wanders by 10 % with the rounding of any inner product)
    V, A, Acsr = _system(dim, N, deg, mass=2000.0)
    A.vals.mul_(1.0 + 0.002 * torch.sin(torch.arange(A.vals.numel(), device="cuda", dtype=torch.float64)))
    A.version += 1
    Acsr = A.to_scipy()
    n = V.num_dofs
    x = V.x.cpu().numpy()
    cols = [np.cos(2.0 * x[:, 0]) * (1.0 + x[:, 1]), 1e-3 * np.sin(5.0 * x[:, 0] * x[:, -1]), np.zeros(n)][:nc]
    B = FieldStorage(n, nc, "cuda")
    B.dev()[:] = torch.from_numpy(np.stack(cols, axis=1)).cuda()
    out = {}
    for merged in (True, False):
        ksp = KSPSolver(None, {"ksp_type": "bcgs", "pc_type": "jacobi", "ksp_rtol": 1e-10, "ksp_atol": 1e-50,
                               "ksp_bcgs_merged_reduction": merged})
        ksp.setOperators(A)
        X = FieldStorage(n, nc, "cuda")
        X.dev().fill_(7.0)  # a zero initial guess is the solver's business
        reasons = ksp.solve_block(B, X)
        out[merged] = (X.dev().cpu().numpy().copy(), ksp.iterations[:nc], reasons)
    for c in range(nc):
        if not cols[c].any():  # b = 0: converged on the spot (atol), x = 0 from both
            for merged in (True, False):
                assert out[merged][2][c] > 0 and out[merged][1][c] == 0 and not out[merged][0][:, c].any()
            continue
        sol, reason, its, _ = O.jacobi_bicgstab(Acsr, cols[c], rtol=1e-10, atol=1e-50)
        for merged in (True, False):
            xs, it, rs = out[merged]
            assert rs[c] == reason == 2  # KSP_CONVERGED_RTOL
            assert abs(it[c] - its) <= (2 if merged else 1), (merged, it, its)
            assert np.abs(xs[:, c] - sol).max() < 1e-7 * max(np.abs(sol).max(), 1.0)
    assert np.abs(out[True][0] - out[False][0]).max() < 1e-7 * np.abs(out[False][0]).max()


@pytest.mark.parametrize("dim,N,deg,pairs", [(2, 24, 2, "never"), (3, 10, 1, "never"), (3, 16, 1, "always"), (3, 8, 2, "never")])
def test_merged_reduction_cg_matches_standard_cg_and_oracle(hip, dim, N, deg, pairs):
    """OX_KSP_CG_MERGED (ksp_cg_merged_reduction: one synchronisation point and three kernels per iteration -- alpha AND
    beta from the sums of the mat-vec's epilogue, r'.z' = r.z - 2 alpha q.z + alpha^2 q.D^-1 q) against the standard
    recurrences and the oracle's PETSc-convention CG: same converged reason, iteration counts within 1 (the convergence
    test sees the same true |D^-1 r|), same solution to solver tolerance; zero and nonzero initial guess, the iteration
    limit, a right-hand side that converges at once; the generic and the pair-slot SpMV kernel."""
    from oasisx_amd.fem import FieldStorage
    from oasisx_amd.ksp import KSPSolver
    from oracle import ipcs_oracle as O

    V, A, Acsr = _system(dim, N, deg)
    if pairs == "always":
        assert A.freeze(pairs="always") and A.ps_code is not None
    n = V.num_dofs
    x = V.x.cpu().numpy()
    b = np.cos(2.0 * x[:, 0]) * (1.0 + x[:, 1]) + 1e-3 * np.sin(5.0 * x[:, 0] * x[:, -1])
    B = FieldStorage(n, 1, "cuda")
    B.dev()[:, 0] = torch.from_numpy(b).cuda()
    sol, reason, its, _ = O.jacobi_cg(Acsr, b, rtol=1e-10, atol=1e-50)
    out = {}
    for merged in (True, False):
        ksp = KSPSolver(None, {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-10, "ksp_atol": 1e-50,
                               "ksp_cg_merged_reduction": merged, "ksp_cg_single_reduction": False})
        ksp.setOperators(A)
        X = FieldStorage(n, 1, "cuda")
        rs = ksp.solve_block(B, X)
        out[merged] = (X.dev()[:, 0].cpu().numpy().copy(), ksp.iterations[0], rs[0], float(ksp.last_result.rnorm[0]))
        assert rs[0] == reason == 2
        assert abs(ksp.iterations[0] - its) <= 1, (merged, ksp.iterations, its)
        assert np.abs(out[merged][0] - sol).max() < 1e-8 * max(np.abs(sol).max(), 1.0)
    assert abs(out[True][1] - out[False][1]) <= 1
    assert abs(out[True][3] - out[False][3]) <= 0.5 * max(out[True][3], out[False][3])  # the same (true) residual norm, up to one iteration
    # nonzero initial guess: fewer iterations, same answer; a guess that already satisfies the test: zero iterations
    opts = {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-10, "ksp_initial_guess_nonzero": True,
            "ksp_cg_merged_reduction": True, "ksp_cg_single_reduction": False}
    ksp = KSPSolver(None, opts)
    ksp.setOperators(A)
    X = FieldStorage(n, 1, "cuda")
    X.dev()[:, 0] = torch.from_numpy(sol * (1.0 + 1e-4)).cuda()
    assert ksp.solve_block(B, X)[0] == 2 and 0 < ksp.iterations[0] < its
    assert np.abs(X.dev()[:, 0].cpu().numpy() - sol).max() < 1e-8 * max(np.abs(sol).max(), 1.0)
    its_g = ksp.iterations[0]
    assert ksp.solve_block(B, X)[0] == 2 and ksp.iterations[0] <= 1 < its_g + 1
    # iteration limit: DIVERGED_ITS after exactly max_it iterations, x = the max_it-th iterate of the standard form
    xs = {}
    for merged in (True, False):
        ksp = KSPSolver(None, {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-14, "ksp_max_it": 5,
                               "ksp_cg_merged_reduction": merged, "ksp_cg_single_reduction": False})
        ksp.setOperators(A)
        X = FieldStorage(n, 1, "cuda")
        assert ksp.solve_block(B, X)[0] == -3 and ksp.iterations[0] == 5
        xs[merged] = X.dev()[:, 0].cpu().numpy().copy()
    assert np.abs(xs[True] - xs[False]).max() < 1e-12 * np.abs(xs[False]).max()
    # b = 0: converged at once (atol), x = 0
    ksp = KSPSolver(None, {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-10, "ksp_atol": 1e-30,
                           "ksp_cg_merged_reduction": True})
    ksp.setOperators(A)
    Z, X = FieldStorage(n, 1, "cuda"), FieldStorage(n, 1, "cuda")
    assert ksp.solve_block(Z, X)[0] in (2, 3) and ksp.iterations[0] == 0 and float(X.dev().abs().max()) == 0.0


def test_identical_runs_give_identical_bits_and_the_launch_schedule_does_not_follow_the_clock(hip):
    """Round 4: the number of iterations queued between two host reads used to be re-derived from the measured time of
    the first solves; with several columns it decides when a solve narrows to its last live column, whose 1-column
    kernels sum in another order -- so identical runs differed in the last bits whenever an iteration took about the
    threshold (seen at this size: 13 to 55 of 200 runs, none with kernels serialised).  The interval now is a function
    of the matrix alone: it stays what the first solve chose, and repeated runs agree bit for bit."""
    from tests.helpers import KRYLOV, make_hip_problem

    opts = {k: dict(v, ksp_initial_guess_nonzero=True) for k, v in KRYLOV.items()}
    dt, nu = 0.005, 0.01
    ref = None
    for run in range(12):
        S, clock, mesh = make_hip_problem(3, 5, u_deg=2, solver_options=opts)
        every0 = None
        for _ in range(2):  # (each step: up to 10 inner iterations = up to 10 tentative and pressure solves)
            clock["t"] += dt
            S.solve(dt, nu)
            every = {k: dict(s._every) for k, s in (("u", S._solver_u), ("p", S._solver_p), ("c", S._solver_c))}
            every0 = every0 or every
            assert every == every0
        got = (S._U.rdev().clone(), S._P.rdev().clone(), [list(s.iterations) for s in (S._solver_u, S._solver_p, S._solver_c)])
        if ref is None:
            ref = got
        else:
            assert got[2] == ref[2] and torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), run


@pytest.mark.parametrize("dim,N,deg,nc,dictionary", [(2, 24, 2, 1, False), (2, 25, 1, 1, False), (3, 16, 1, 1, True),
                                                     (3, 8, 2, 3, False), (2, 1, 1, 1, False), (3, 12, 1, 2, True)])
def test_folded_cg_matches_the_five_kernel_iteration_and_oracle(hip, dim, N, deg, nc, dictionary):
    """Round 4: one-column CG on one GPU runs its two synchronisation points inside the update kernels
    (k_cg_update1f / k_cg_update2f, 3 kernels per iteration; per-solver option ``ksp_cg_fold_blocks``).  Against the five-kernel form
    (fold off) and the oracle's PETSc-convention CG: same converged reasons, iteration counts within 1, solutions to
    solver tolerance; odd and even row counts (the pair loop's tail), a 4-row system, a value dictionary of the diagonal
    (1-byte codes in the update kernels), several columns (the folded kernels serve the narrowed tail of the lock-step
    solve), a right-hand side that needs no iteration, a solve cut by max_it, and block counts from 1 to many."""
    from oasisx_amd import _lib
    from oasisx_amd.fem import FieldStorage
    from oasisx_amd.ksp import KSPSolver
    from oracle import ipcs_oracle as O

    lib = _lib.load()
    V, A, Acsr = _system(dim, N, deg, mass=3.0)
    if dictionary:
        A.freeze()
    n = V.num_dofs
    x = V.x.cpu().numpy()
    cols = [np.cos(2.0 * x[:, 0]) * (1.0 + x[:, 1]), 1e-3 * np.sin(5.0 * x[:, 0] * x[:, -1]), np.exp(x[:, 1])][:nc]
    B = FieldStorage(n, nc, "cuda")
    B.dev()[:] = torch.from_numpy(np.stack(cols, axis=1)).cuda()
    default_blocks = lib.ox_ksp_default_fold_blocks()
    assert default_blocks > 0
    out = {}
    if True:
        for blocks in (0, 1, 7, default_blocks, 1024):
            fold = {"ksp_cg_fold_blocks": blocks}  # per solver: two solvers with different settings do not interfere
            ksp = KSPSolver(None, {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-10, "ksp_atol": 1e-50,
                                   "ksp_cg_single_reduction": False, "ksp_cg_merged_reduction": False, **fold})
            ksp.setOperators(A)
            assert ksp._cg_folded() == (blocks > 0)
            assert ksp._cg_kernels_per_iteration() == (3 if blocks > 0 else 5)
            X = FieldStorage(n, nc, "cuda")
            reasons = ksp.solve_block(B, X)
            sol = X.dev().cpu().numpy().copy()
            its = list(ksp.iterations[:nc])
            # the solution as the right-hand side's answer: a second solve from it needs no iteration
            ksp2 = KSPSolver(None, {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-6, "ksp_atol": 1e-50,
                                    "ksp_initial_guess_nonzero": True, "ksp_cg_single_reduction": False,
                                    "ksp_cg_merged_reduction": False, **fold})
            ksp2.setOperators(A)
            r2 = ksp2.solve_block(B, X)
            assert all(r == 2 for r in r2) and list(ksp2.iterations[:nc]) == [0] * nc
            assert np.array_equal(X.dev().cpu().numpy(), sol)  # untouched
            # cut by max_it: reason -3 after exactly that many iterations
            ksp3 = KSPSolver(None, {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-30, "ksp_atol": 1e-300,
                                    "ksp_max_it": 3, "ksp_cg_single_reduction": False, "ksp_cg_merged_reduction": False,
                                    **fold})
            ksp3.setOperators(A)
            X3 = FieldStorage(n, nc, "cuda")
            r3 = ksp3.solve_block(B, X3)
            out[blocks] = (sol, its, reasons, X3.dev().cpu().numpy().copy(), list(r3), list(ksp3.iterations[:nc]))
    ref = out[0]
    for c in range(nc):
        sol, reason, its, _ = O.jacobi_cg(Acsr, cols[c], rtol=1e-10, atol=1e-50)
        for blocks, (xs, it, rs, x3, r3, it3) in out.items():
            assert rs[c] == reason == 2, (blocks, rs)
            assert abs(it[c] - its) <= 1 and abs(it[c] - ref[1][c]) <= 1, (blocks, it, its, ref[1])
            assert np.abs(xs[:, c] - sol).max() < 1e-8 * max(np.abs(sol).max(), 1.0)
            if n > 4:  # (a 4-row system is solved exactly within 3 iterations)
                assert r3[c] == -3 and it3[c] == 3, (blocks, r3, it3)
            # three iterations of the same recurrences: the same iterate up to the rounding of the dot products
            assert np.abs(x3[:, c] - ref[3][:, c]).max() <= 1e-12 * max(np.abs(ref[3][:, c]).max(), 1e-300)


@pytest.mark.parametrize("dim,N,deg,dictionary", [(2, 24, 2, False), (3, 16, 1, True), (2, 25, 1, False), (2, 1, 1, False)])
def test_folded_merged_cg_matches_its_three_kernel_form(hip, dim, N, deg, dictionary):
    """Round 4: the merged-reduction CG with its one synchronisation point inside the update kernel (k_cgm_updatef, 2
    kernels per iteration; the state alternates between two blocks, so only even batches fold).  Fold on / off, even and
    odd batch lengths (an odd ``check_every`` takes the three-kernel form), warm start included: same reasons, iteration
    counts within 1 of each other and of the oracle, solutions to solver tolerance."""
    from oasisx_amd import _lib
    from oasisx_amd.fem import FieldStorage
    from oasisx_amd.ksp import KSPSolver
    from oracle import ipcs_oracle as O

    lib = _lib.load()
    V, A, Acsr = _system(dim, N, deg, mass=3.0)
    if dictionary:
        A.freeze()
    n = V.num_dofs
    x = V.x.cpu().numpy()
    b = np.cos(2.0 * x[:, 0]) * (1.0 + x[:, 1])
    B = FieldStorage(n, 1, "cuda")
    B.dev()[:, 0] = torch.from_numpy(b).cuda()
    sol, reason, its, _ = O.jacobi_cg(Acsr, b, rtol=1e-10, atol=1e-50)
    out = {}
    if True:
        for blocks in (0, -1):
            for every in (None, 3, 4):
                ksp = KSPSolver(None, {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-10, "ksp_atol": 1e-50,
                                       "ksp_cg_single_reduction": False, "ksp_cg_merged_reduction": True,
                                       "ksp_cg_fold_blocks": blocks})
                ksp.setOperators(A)
                ksp.check_every = every
                # (the library's own answer: the fold needs even batches, min(check interval, 8))
                even = (min(every, 8) if every else min(ksp._interval_for(1, _lib.KSP_CG_MERGED), 8)) % 2 == 0
                assert ksp._cg_merged() and ksp._cg_kernels_per_iteration() == (2 if blocks and even else 3)
                X = FieldStorage(n, 1, "cuda")
                r = ksp.solve_block(B, X)
                it = ksp.iterations[0]
                xs = X.dev()[:, 0].cpu().numpy().copy()
                # a warm start from the solution: nothing left to do, x untouched
                ksp.updateOptions({"ksp_initial_guess_nonzero": True, "ksp_rtol": 1e-6})
                r2 = ksp.solve_block(B, X)
                assert r2[0] == 2 and ksp.iterations[0] == 0 and np.array_equal(X.dev()[:, 0].cpu().numpy(), xs)
                out[(blocks, every)] = (xs, it, r[0])
    for key, (xs, it, r) in out.items():
        assert r == reason == 2, (key, r)
        assert abs(it - its) <= 2 and abs(it - out[(0, None)][1]) <= 1, (key, it, its)
        assert np.abs(xs - sol).max() < 1e-8 * max(np.abs(sol).max(), 1.0)
