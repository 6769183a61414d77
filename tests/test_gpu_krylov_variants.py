"""GPU: the single-reduction CG (Chronopoulos-Gear recurrences, ONE merged reduction per iteration,
PETSc's -ksp_cg_single_reduction; the default of KSPSolver here) against the standard recurrences
(-ksp_cg_single_reduction false) and the oracle's PETSc-convention CG: same solution to solver
tolerance, iteration counts within +-2 (the two recurrences differ in rounding only), same converged
reasons, lock-step columns and the narrowed continuation included."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _system(dim, N, deg):
    from oasisx_amd import fem
    from oasisx_amd import mesh as M
    from oasisx_amd.la import SellMatrix
    from oracle import ipcs_oracle as O

    mesh = (M.create_rectangle(None, [[-1.0, -1.0], [1.0, 1.0]], [N, N]) if dim == 2
            else M.create_box(None, [[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]], [N, N, N]))
    V = fem.FunctionSpace(mesh, deg, window=256)
    F = O.Forms(mesh.coords.cpu().numpy(), V.cells_in_kernel_order(), deg, 1, vd=V.cell_dofs.cpu().numpy(),
                qd=V.cells_in_kernel_order(), nv_dofs=V.num_dofs, nq_dofs=mesh.num_vertices)
    Acsr = (F.stiffness_v() + 3.0 * F.mass_v()).tocsr()
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
