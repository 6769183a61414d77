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
