"""BASELINE.json's full size (C3: 3-D Taylor-Green 128^3 x 6 tetrahedra, P2-P1) through
size-independent properties -- the oracle cannot run at this size, exact identities can:
entry counts, partition of unity and polynomial exactness of the assembled
operators, symmetry, nullspaces, the discrete divergence of a representable solenoidal field,
bit-identity of the compressed SpMV streams, and one full step against the analytic solution."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N = 128


@pytest.fixture(scope="module")
def problem(hip):
    from tests.helpers import make_hip_problem

    opts = {k: {"ksp_type": t, "pc_type": "jacobi", "ksp_rtol": 1e-8, "ksp_atol": 1e-14,
                "ksp_initial_guess_nonzero": True}
            for k, t in (("tentative", "bcgs"), ("pressure", "cg"), ("scalar", "cg"))}
    S, clock, mesh = make_hip_problem(3, N, u_deg=2, nu=0.01, dt=0.005 * 32 / N, solver_options=opts, window=4096,
                                      low_memory=False)
    yield S, clock, mesh
    del S
    torch.cuda.empty_cache()


def _vec(n, nc=1, fill=None):
    x = torch.zeros(n, nc, dtype=torch.float64, device="cuda")
    if fill is not None:
        x[:] = fill
    return x


def test_entry_counts(problem):
    S, _, mesh = problem
    assert mesh.num_cells == 6 * N ** 3
    assert S._n_u == (2 * N + 1) ** 3 and S._n_q == (N + 1) ** 3
    assert S._M.pattern.nnz == 230 * N ** 3 + 138 * N ** 2 + 24 * N + 1
    assert S._Ap.pattern.nnz == 15 * N ** 3 + 21 * N ** 2 + 9 * N + 1
    if not S._low_memory:
        assert S._pat_vq.nnz == S._pat_qv.nnz == 65 * N ** 3 + 57 * N ** 2 + 15 * N + 1


def test_partition_of_unity_nullspaces_and_exactness(problem):
    S, _, _ = problem
    Vi, Q = S._Vi[0][0], S._Q
    nu_, nq = S._n_u, S._n_q
    one_u, one_q = _vec(nu_, 1, 1.0), _vec(nq, 1, 1.0)
    y = _vec(nu_)
    S._M.mult(one_u, y, 1)
    assert abs(float(y.sum()) - 8.0) < 1e-10  # sum_ij M_ij = |Omega| = 2^3
    S._K.mult(one_u, y, 1)
    assert float(y.abs().max()) < 1e-9  # K 1 = 0
    yq = _vec(nq)
    S._Ap.mult(one_q, yq, 1)
    assert float(yq.abs().max()) < 1e-9  # Ap 1 = 0 (no pressure Dirichlet condition)
    # quadratic exactness on f = x^2 + y z (in P2): f^T M 1 = int f = 8/3; f^T K f = int |grad f|^2
    X = Vi.x[:nu_]
    f = (X[:, 0] ** 2 + X[:, 1] * X[:, 2]).reshape(-1, 1).contiguous()
    S._M.mult(one_u, y, 1)
    assert abs(float((f * y).sum()) - 8.0 / 3.0) < 1e-10
    S._K.mult(f, y, 1)
    # |grad f|^2 = 4x^2 + z^2 + y^2 -> 8 * (4/3 + 1/3 + 1/3) = 16
    assert abs(float((f * y).sum()) - 16.0) < 1e-9


def test_symmetry_and_bit_identity_of_the_compressed_streams(problem):
    from oasisx_amd import _lib

    S, _, _ = problem
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(5)
    for A in (S._M, S._Ap):
        n = A.pattern.n_rows
        assert A.vcode is not None and A.pattern.frac16 > 0.9
        x = torch.randn(n, 1, dtype=torch.float64, device="cuda", generator=g)
        z = torch.randn(n, 1, dtype=torch.float64, device="cuda", generator=g)
        ys = []
        for var in (1, 3, 7, 15):  # int32 columns + f64 values; 16-bit columns; + 1-byte value codes; pair slots
            A.set_levels(var)
            y = _vec(n)
            A.mult(x, y, 1)
            ys.append(y)
        A.set_levels(None)
        assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2]) and torch.equal(ys[0], ys[3])
        if A is S._Ap:
            assert A.ps_code is not None, "the P1 stiffness matrix pairs up well: the pair-slot stream must exist"
        w = _vec(n)
        A.mult(z, w, 1)
        a, b = float((z * ys[0]).sum()), float((x * w).sum())
        assert abs(a - b) <= 1e-11 * max(abs(a), abs(b), 1.0)  # (z, A x) = (x, A z)


def test_divergence_of_a_representable_solenoidal_field_vanishes(problem):
    S, _, _ = problem
    n = S._n_u
    X = S._Vi[0][0].x[:n]
    U = S._U.dev()
    keep = U.clone()
    U[:n, 0], U[:n, 1], U[:n, 2] = X[:, 1] * X[:, 2], X[:, 0] * X[:, 2], -2.0 * X[:, 0] * X[:, 1]  # div = 0, in P2
    S.pressure_assemble(0.01)  # b2 = -(1/dt) int div(u) q
    b2 = S._B2.dev()[: S._n_q].clone()
    U.copy_(keep)
    assert float(b2.abs().max()) < 1e-9


def test_one_step_against_the_analytic_solution(problem):
    from oracle import ipcs_oracle as O

    S, clock, _ = problem
    nu, dt = 0.01, 0.005 * 32 / N
    clock["t"] += dt
    S.solve(dt, nu, max_iter=1)
    its = S.iteration_counts()
    assert all(r > 0 for r in its["pressure"][:1]) and max(its["pressure"]) < 5000
    n = S._n_u
    X3 = S._Vi[0][0].x[:n].cpu().numpy().T
    U = S._U.dev()[:n].cpu().numpy()
    for c, f in enumerate((O.tg_u, O.tg_v, O.tg_w)):
        ex = f(X3, clock["t"], nu)
        assert np.abs(U[:, c] - ex).max() < 5e-5, (c, float(np.abs(U[:, c] - ex).max()))
