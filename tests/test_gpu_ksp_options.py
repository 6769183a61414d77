"""GPU: PETSc options of ``KSP.solve`` (reference ksp.py:38-53,71-78; fracstep.py:570) that round 5 honours instead of
warning about -- ``ksp_divtol`` (KSP_DIVERGED_DTOL), ``ksp_error_if_not_converged`` (an exception) -- and the test on the
STORED residual behind the merged-reduction BiCGStab (the default on mesh-partitioned operators), whose loop tests a
recurrence norm.  The oracle's ``jacobi_cg`` / ``jacobi_bicgstab`` are extended the same way (``divtol``)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _space_and_forms(dim, N, deg):
    from oasisx_amd import fem
    from oasisx_amd import mesh as M
    from oracle import ipcs_oracle as O

    mesh = (M.create_rectangle(None, [[-1.0, -1.0], [1.0, 1.0]], [N, N]) if dim == 2
            else M.create_box(None, [[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]], [N, N, N]))
    V = fem.FunctionSpace(mesh, deg, window=256)
    F = O.Forms(mesh.coords.cpu().numpy(), V.cells_in_kernel_order(), deg, 1, vd=V.cell_dofs.cpu().numpy(),
                qd=V.cells_in_kernel_order(), nv_dofs=V.num_dofs, nq_dofs=mesh.num_vertices)
    return mesh, V, F


def _matrix(V, Acsr, symmetric):
    from oasisx_amd.la import SellMatrix

    A = SellMatrix(V.pattern, symmetric=symmetric)
    A.vals.copy_(V.pattern.values_from_csr(Acsr.tocsr()))
    A.version += 1
    return A


def _ill_scaled(dim, N, deg, decades, seed=0):
    """A nonsymmetric, badly COLUMN-scaled operator on the P`deg` pattern: (K + 3 M + convection-like skew part) D with
    D = diag(10^(decades * U(-1, 1))).  Left Jacobi removes a row scaling, not this one: BiCGStab then works with
    |s| >> |r| near convergence, where the recurrence s.s - 2 omega t.s + omega^2 t.t of the merged variant cancels."""
    import scipy.sparse as sp

    mesh, V, F = _space_and_forms(dim, N, deg)
    K, M = F.stiffness_v().tocsr(), F.mass_v().tocsr()
    rng = np.random.default_rng(seed)
    skew = sp.triu(K, 1)
    skew = (skew - skew.T) * 0.4
    d = 10.0 ** (decades * rng.uniform(-1.0, 1.0, V.num_dofs))
    Acsr = ((K + 3.0 * M + skew) @ sp.diags(d)).tocsr()
    return V, Acsr


def _true_rel_residual(Acsr, b, x):
    dinv = 1.0 / Acsr.diagonal()
    return np.linalg.norm(dinv * (b - Acsr @ x)) / np.linalg.norm(dinv * b)


@pytest.mark.parametrize("dim,N,deg,nc,decades", [(2, 20, 2, 1, 3.0), (2, 16, 2, 3, 3.0), (3, 6, 2, 3, 2.5), (2, 24, 1, 2, 4.0)])
def test_merged_bicgstab_ends_on_the_stored_residual(hip, dim, N, deg, nc, decades):
    """rtol 1e-12 on an ill-scaled operator: the merged-reduction BiCGStab must END with the residual it reports -- and
    the explicitly recomputed one -- inside the tolerance, like the three-point form, whatever its recurrence norm said on
    the way (VERDICT r04 "missing" 3).  Lock-step columns (different right-hand sides converge at different iterations)
    and the narrowed continuation included; ``resumed`` counts the re-openings."""
    from oasisx_amd.fem import FieldStorage
    from oasisx_amd.ksp import KSPSolver
    from oracle import ipcs_oracle as O

    V, Acsr = _ill_scaled(dim, N, deg, decades)
    A = _matrix(V, Acsr, symmetric=False)
    n = V.num_dofs
    x = V.x.cpu().numpy()
    rng = np.random.default_rng(1)
    cols = [Acsr @ (np.cos(2.0 * x[:, 0]) * (1.0 + x[:, 1])), Acsr @ rng.standard_normal(n), 1e-6 * np.exp(x[:, 1])][:nc]
    B = FieldStorage(n, nc, "cuda")
    B.dev()[:] = torch.from_numpy(np.stack(cols, axis=1)).cuda()
    rtol = 1e-12
    out = {}
    for merged in (False, True):
        ksp = KSPSolver(None, {"ksp_type": "bcgs", "pc_type": "jacobi", "ksp_rtol": rtol, "ksp_atol": 1e-300,
                               "ksp_max_it": 4000, "ksp_bcgs_merged_reduction": merged})
        ksp.setOperators(A)
        X = FieldStorage(n, nc, "cuda")
        reasons = ksp.solve_block(B, X)
        res = ksp.last_result
        out[merged] = dict(x=X.dev().cpu().numpy().copy(), reasons=reasons, its=list(res.its[:nc]),
                           rn=list(res.rnorm[:nc]), bn=list(res.bnorm[:nc]), resumed=list(res.resumed[:nc]))
    std, mrg = out[False], out[True]
    assert std["resumed"] == [0] * nc
    for c in range(nc):
        _, reason_o, its_o, _ = O.jacobi_bicgstab(Acsr, cols[c], rtol=rtol, atol=1e-300, max_it=4000)
        true_s, true_m = _true_rel_residual(Acsr, cols[c], std["x"][:, c]), _true_rel_residual(Acsr, cols[c], mrg["x"][:, c])
        if reason_o == 2 and std["reasons"][c] == 2:
            # the three-point form converged: so must the merged one, and on a stored residual inside the tolerance
            assert mrg["reasons"][c] == 2, (c, mrg, std)
            assert mrg["rn"][c] <= rtol * mrg["bn"][c]
            # explicit residual: what the stored one drifts to in ANY BiCGStab; no worse than the three-point form's
            assert true_m <= 10.0 * max(true_s, rtol), (c, true_m, true_s, mrg["resumed"])
        else:  # not reachable at this conditioning: both must say so (never a converged reason on a residual above tol)
            assert mrg["reasons"][c] <= 0 or mrg["rn"][c] <= rtol * mrg["bn"][c]
        if mrg["reasons"][c] > 0:
            assert mrg["rn"][c] <= max(rtol * mrg["bn"][c], 1e-300)


def test_merged_bicgstab_reopens_a_column_whose_recurrence_norm_undershoots(hip):
    """The mechanism itself.  The recurrence |r|^2 = s.s - 2 omega t.s + omega^2 t.t is s.s (1 - cos^2(s, t)): once one
    omega step reduces the residual by more than ~1e-8 it is rounding noise of either sign, and a negative value is
    clamped to 0 -- "converged" whatever the tolerance.  D^-1 A = I + 1e-9 E does that at its first iteration (|s| ~ 1e-9,
    |r| ~ 1e-18 of |b|); with rtol 1e-25 the stored residual then FAILS the test the recurrence norm passed: the column
    must be re-opened (``resumed``), run one more iteration and end on a stored residual inside the tolerance, exactly
    like the three-point form.  (The sign of the noise differs from right-hand side to right-hand side: several.)"""
    import scipy.sparse as sp

    from oasisx_amd.fem import FieldStorage
    from oasisx_amd.ksp import KSPSolver

    mesh, V, F = _space_and_forms(2, 16, 2)
    K = F.stiffness_v().tocsr()
    skew = sp.triu(K, 1)
    skew = (skew - skew.T) * 0.4
    n = V.num_dofs
    rng = np.random.default_rng(11)
    Acsr = (sp.diags(rng.uniform(1.0, 2.0, n)) + 1e-9 * (K + skew)).tocsr()
    A = _matrix(V, Acsr, symmetric=False)
    rtol = 1e-25
    resumed, nsolves = 0, 0
    for nc in (1, 3, 2, 3):
        cols = [rng.standard_normal(n) for _ in range(nc)]
        B = FieldStorage(n, nc, "cuda")
        B.dev()[:] = torch.from_numpy(np.stack(cols, axis=1)).cuda()
        out = {}
        for merged in (False, True):
            ksp = KSPSolver(None, {"ksp_type": "bcgs", "pc_type": "jacobi", "ksp_rtol": rtol, "ksp_atol": 1e-300,
                                   "ksp_max_it": 50, "ksp_bcgs_merged_reduction": merged})
            ksp.setOperators(A)
            X = FieldStorage(n, nc, "cuda")
            reasons = ksp.solve_block(B, X)
            res = ksp.last_result
            out[merged] = (reasons, list(res.its[:nc]), list(res.rnorm[:nc]), list(res.bnorm[:nc]), list(res.resumed[:nc]),
                           X.dev().cpu().numpy().copy())
        (rs, its_s, rn_s, bn_s, _, xs), (rm, its_m, rn_m, bn_m, resm, xm) = out[False], out[True]
        assert rs == [2] * nc, (rs, its_s, rn_s)
        for c in range(nc):
            assert rm[c] == 2 and rn_m[c] <= rtol * bn_m[c], (c, rm, rn_m, bn_m, resm)
            assert abs(its_m[c] - its_s[c]) <= 1, (its_m, its_s)
            assert np.abs(xm[:, c] - xs[:, c]).max() <= 1e-15 * np.abs(xs[:, c]).max()
        resumed += sum(resm)
        nsolves += nc
    assert 0 < resumed <= 2 * nsolves, resumed


@pytest.mark.parametrize("kind", ["cg", "bcgs", "bcgs_merged", "cg_single", "cg_merged"])
def test_divtol_ends_a_diverging_solve_like_the_oracle(hip, kind):
    """``ksp_divtol`` (PETSc default 1e4): |D^-1 r| >= divtol |D^-1 b| -> KSP_DIVERGED_DTOL = -4, at the iteration where
    the oracle's (KSPConvergedDefault's) test fires -- iteration 0 for a far-off initial guess, a later one for a
    threshold the (non-monotone) preconditioned residual crosses on its way."""
    from oasisx_amd.fem import FieldStorage
    from oasisx_amd.ksp import KSPSolver
    from oracle import ipcs_oracle as O

    mesh, V, F = _space_and_forms(2, 20, 2)
    Acsr = (F.stiffness_v() + 0.05 * F.mass_v()).tocsr()
    sym = not kind.startswith("bcgs")
    A = _matrix(V, Acsr, symmetric=sym)
    n = V.num_dofs
    # a smooth load: the PRECONDITIONED residual norm, which neither method minimises, grows for dozens of iterations
    # (CG: 5.4, 7.7, ... up to 47 |D^-1 b|; BiCGStab: 1.02, 1.08, 1.10, ... 1.9) before it falls
    b = np.cos(2.0 * V.x.cpu().numpy()[:, 0])
    fn = O.jacobi_cg if sym else O.jacobi_bicgstab
    opts = {"ksp_type": "cg" if sym else "bcgs", "pc_type": "jacobi", "ksp_rtol": 1e-10, "ksp_atol": 1e-300,
            "ksp_cg_single_reduction": kind == "cg_single", "ksp_cg_merged_reduction": kind == "cg_merged",
            "ksp_bcgs_merged_reduction": kind == "bcgs_merged"}
    B = FieldStorage(n, 1, "cuda")
    B.dev()[:, 0] = torch.from_numpy(b).cuda()
    # (a) the default leaves a converging solve alone
    ksp = KSPSolver(None, opts)
    ksp.setOperators(A)
    X = FieldStorage(n, 1, "cuda")
    assert ksp.solve_block(B, X)[0] == 2
    # (b) a threshold the residual history crosses: found on the oracle, reproduced on the device
    found = None
    for divtol in (1.05, 6.0, 20.0):
        _, reason, its, rn = fn(Acsr, b, rtol=1e-10, atol=1e-300, divtol=divtol)
        if reason == -4 and its >= 2:
            found = (divtol, its)
            break
    assert found is not None, "the test problem's residual never grows: pick another right-hand side"
    divtol, its_o = found
    ksp = KSPSolver(None, dict(opts, ksp_divtol=divtol))
    ksp.setOperators(A)
    X = FieldStorage(n, 1, "cuda")
    assert ksp.solve_block(B, X)[0] == -4
    slack = 0 if kind in ("cg", "bcgs") else 2  # (the reduced-synchronisation recurrences differ in rounding / test one point later)
    assert abs(ksp.iterations[0] - its_o) <= slack, (ksp.iterations, its_o)
    assert ksp.last_result.rnorm[0] >= divtol * ksp.last_result.bnorm[0]
    # (c) a far-off initial guess: flagged before the first iteration
    x0 = 1e6 * np.cos(7.0 * V.x.cpu().numpy()[:, 0])
    _, reason, its, _ = fn(Acsr, b, x0=x0, rtol=1e-10, atol=1e-300, divtol=1e4)
    assert reason == -4 and its == 0
    ksp = KSPSolver(None, dict(opts, ksp_initial_guess_nonzero=True))
    ksp.setOperators(A)
    X.dev()[:, 0] = torch.from_numpy(x0).cuda()
    assert ksp.solve_block(B, X)[0] == -4 and ksp.iterations[0] == 0
    assert np.array_equal(X.dev()[:, 0].cpu().numpy(), x0)  # x is left as it was handed in


def test_divtol_in_lockstep_columns(hip):
    """One column diverges (DTOL), the others converge: per-column reasons, and the converged columns' solutions are those
    of a solve without the diverging one."""
    from oasisx_amd.fem import FieldStorage
    from oasisx_amd.ksp import KSPSolver
    from oracle import ipcs_oracle as O

    mesh, V, F = _space_and_forms(2, 16, 2)
    Acsr = (F.stiffness_v() + 0.05 * F.mass_v()).tocsr()
    A = _matrix(V, Acsr, symmetric=True)
    n = V.num_dofs
    x = V.x.cpu().numpy()
    rng = np.random.default_rng(2)
    cols = [np.cos(40.0 * x[:, 0]) * np.cos(37.0 * x[:, 1]), np.cos(2.0 * x[:, 0]), rng.standard_normal(n)]
    # a threshold only the smooth load crosses (see above)
    for divtol in (3.0, 6.0, 20.0):
        rs = [O.jacobi_cg(Acsr, c, rtol=1e-10, atol=1e-300, divtol=divtol) for c in cols]
        if [r[1] for r in rs] == [2, -4, 2]:
            break
    else:
        pytest.skip("no threshold separates the columns on this mesh")
    B = FieldStorage(n, 3, "cuda")
    B.dev()[:] = torch.from_numpy(np.stack(cols, axis=1)).cuda()
    ksp = KSPSolver(None, {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-10, "ksp_atol": 1e-300, "ksp_divtol": divtol,
                           "ksp_cg_single_reduction": False})
    ksp.setOperators(A)
    X = FieldStorage(n, 3, "cuda")
    assert ksp.solve_block(B, X) == [2, -4, 2]
    its = ksp.iterations[:3]
    xs = X.dev().cpu().numpy()
    for c in (0, 2):
        assert abs(its[c] - rs[c][2]) <= 1
        assert np.abs(xs[:, c] - rs[c][0]).max() < 1e-8 * np.abs(rs[c][0]).max()
    assert its[1] == rs[1][2]


def test_error_if_not_converged_raises(hip):
    """``ksp_error_if_not_converged`` (the reference sets it for its pressure solver, fracstep.py:570): a negative reason
    becomes an exception that names it; without the option the reason is returned."""
    from oasisx_amd.fem import FieldStorage
    from oasisx_amd.ksp import KSPConvergenceError, KSPSolver
    from oracle import ipcs_oracle as O

    mesh, V, F = _space_and_forms(2, 12, 2)
    Acsr = (F.stiffness_v() + 3.0 * F.mass_v()).tocsr()
    A = _matrix(V, Acsr, symmetric=True)
    n = V.num_dofs
    b = np.sin(3.0 * V.x.cpu().numpy()[:, 0])
    B = FieldStorage(n, 1, "cuda")
    B.dev()[:, 0] = torch.from_numpy(b).cuda()
    base = {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-14, "ksp_max_it": 4}
    ksp = KSPSolver(None, base)
    ksp.setOperators(A)
    X = FieldStorage(n, 1, "cuda")
    assert ksp.solve_block(B, X)[0] == -3
    ksp = KSPSolver(None, dict(base, ksp_error_if_not_converged=1))
    ksp.setOperators(A)
    with pytest.raises(KSPConvergenceError, match="DIVERGED_ITS after 4 iterations") as e:
        ksp.solve_block(B, X)
    assert e.value.reasons == [-3] and e.value.iterations == [4]
    # the oracle's stand-in for PETSc raises at the same place
    ok = O.OracleKSP(dict(base, ksp_error_if_not_converged=1))
    ok.set_operator(Acsr)
    with pytest.raises(RuntimeError, match="has not converged"):
        ok.solve(b, np.zeros(n))
    # a converging solve is unaffected; "true"/"false" strings as PETSc options arrive
    ksp = KSPSolver(None, {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-8, "ksp_error_if_not_converged": "true"})
    ksp.setOperators(A)
    assert ksp.solve_block(B, X)[0] == 2
    ksp = KSPSolver(None, dict(base, ksp_error_if_not_converged="false"))
    ksp.setOperators(A)
    assert ksp.solve_block(B, X)[0] == -3


def test_pressure_solver_without_pressure_bcs_raises_on_failure(hip):
    """Reference fracstep.py:562-576: without pressure conditions the pressure solver runs with
    ``ksp_error_if_not_converged``; here the Krylov stand-in of its MUMPS solve inherits exactly that."""
    from oasisx_amd.ksp import KSPConvergenceError
    from tests.helpers import KRYLOV, make_hip_problem

    so = {k: dict(v) for k, v in KRYLOV.items()}
    so["pressure"].update(ksp_max_it=2, ksp_rtol=1e-14)
    S, clock, mesh = make_hip_problem(2, 8, 2, solver_options=so)
    clock["t"] = 0.005
    with pytest.raises(KSPConvergenceError, match="oasis_solver|DIVERGED_ITS"):
        S.solve(0.005, 0.01, max_iter=1)


def test_fold_setting_is_per_solver_not_per_process(hip):
    """VERDICT r04 hygiene: two solvers with different fold settings alive at once keep their own schedule and report their
    own kernels per iteration (the library holds no process-wide switch for it any more)."""
    from oasisx_amd.fem import FieldStorage
    from oasisx_amd.ksp import KSPSolver

    mesh, V, F = _space_and_forms(2, 24, 1)
    Acsr = (F.stiffness_v() + 3.0 * F.mass_v()).tocsr()
    A = _matrix(V, Acsr, symmetric=True)
    n = V.num_dofs
    B = FieldStorage(n, 1, "cuda")
    B.dev()[:, 0] = torch.from_numpy(np.cos(3.0 * V.x.cpu().numpy()[:, 1])).cuda()
    base = {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-10, "ksp_cg_single_reduction": False,
            "ksp_cg_merged_reduction": False}
    a, b = KSPSolver(None, dict(base, ksp_cg_fold_blocks=0)), KSPSolver(None, dict(base, ksp_cg_fold_blocks=16))
    a.setOperators(A)
    b.setOperators(A)
    assert (a._cg_folded(), b._cg_folded()) == (False, True)
    assert (a._cg_kernels_per_iteration(), b._cg_kernels_per_iteration()) == (5, 3)
    Xa, Xb = FieldStorage(n, 1, "cuda"), FieldStorage(n, 1, "cuda")
    ra, rb = a.solve_block(B, Xa)[0], b.solve_block(B, Xb)[0]
    ra2 = a.solve_block(B, Xa)[0]  # after b ran: a is still the five-kernel form, bit for bit what it was
    assert ra == rb == ra2 == 2 and abs(a.iterations[0] - b.iterations[0]) <= 1
    assert np.abs(Xa.dev().cpu().numpy() - Xb.dev().cpu().numpy()).max() < 1e-8


def test_nonzero_guess_with_a_tiny_right_hand_side_meets_divtol_at_iteration_zero(hip):
    """every entry point applies PETSc's default ``divtol = 1e4`` against |D^-1 b|.  With a NONZERO initial
    guess and a near-zero right-hand side (a near-steady pressure increment) the initial residual |D^-1 (b - A x0)| is
    more than 1e4 |D^-1 b|: KSP_DIVERGED_DTOL at iteration 0, as PETSc's KSPConvergedDefault reports it.  A caller that
    knows its guess may be far from a tiny right-hand side passes ``ksp_divtol``; the oracle's
    stand-in for PETSc behaves the same."""
    from oasisx_amd.fem import FieldStorage
    from oasisx_amd.ksp import KSPSolver
    from oracle import ipcs_oracle as O

    mesh, V, F = _space_and_forms(2, 12, 2)
    Acsr = (F.stiffness_v() + 3.0 * F.mass_v()).tocsr()
    A = _matrix(V, Acsr, symmetric=True)
    n = V.num_dofs
    x = V.x.cpu().numpy()
    b = 1e-12 * np.sin(3.0 * x[:, 0])
    x0 = 1.0 + 0.5 * np.cos(2.0 * x[:, 1])
    B, X = FieldStorage(n, 1, "cuda"), FieldStorage(n, 1, "cuda")
    B.dev()[:, 0] = torch.from_numpy(b).cuda()
    base = {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-8, "ksp_atol": 1e-50, "ksp_initial_guess_nonzero": True}
    ksp = KSPSolver(None, base)
    ksp.setOperators(A)
    X.dev()[:, 0] = torch.from_numpy(x0).cuda()
    assert ksp.solve_block(B, X)[0] == -4 and int(ksp.last_result.its[0]) == 0  # KSP_DIVERGED_DTOL before the first iteration
    assert np.array_equal(X.dev()[:, 0].cpu().numpy(), x0)  # the guess is handed back untouched
    _, reason_o, its_o, _ = O.jacobi_cg(Acsr, b, x0=x0.copy(), rtol=1e-8, atol=1e-50)
    assert reason_o == -4 and its_o == 0
    # with the bound raised the same solve converges to the tiny solution
    ksp = KSPSolver(None, dict(base, ksp_divtol=1e30))
    ksp.setOperators(A)
    X.dev()[:, 0] = torch.from_numpy(x0).cuda()
    assert ksp.solve_block(B, X)[0] == 2
    sol, reason_o, its_o, _ = O.jacobi_cg(Acsr, b, x0=x0.copy(), rtol=1e-8, atol=1e-50, divtol=1e30)
    assert reason_o == 2 and abs(int(ksp.last_result.its[0]) - its_o) <= 1
    # (from an O(1) guess towards an O(1e-11) solution the TRUE residual ends at rounding level, eps |A| |x0| ~ 1e-3 |b|,
    # on both sides; what is checked is that the iteration left the guess for the tiny solution, as the oracle's did)
    xg = X.dev()[:, 0].cpu().numpy()
    assert np.abs(xg).max() < 1e-9 and np.abs(sol).max() < 1e-9
    assert _true_rel_residual(Acsr, b, xg) < 1e-2 and _true_rel_residual(Acsr, b, sol) < 1e-2
