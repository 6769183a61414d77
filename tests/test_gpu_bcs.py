"""GPU: the reference's DirichletBC tests (test/test_bcs.py:19-160) restated for P = 1..4: a time-dependent
function / Constant imposed through ``bc.apply`` (the ox_set_bc kernel) equals the interpolated function written
into the located dofs, with the dofs located geometrically and topologically (entity dimensions 0, 1, 2)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _locator(x):
    return np.isclose(x[0], 1)


class TimeDependentBC:
    def __init__(self, t):
        self.t = t

    def eval(self, x):
        return np.sin(x[0]) + x[1] * self.t


@pytest.mark.parametrize("P", [1, 2, 3, 4])
def test_function_geometrical(hip, P):
    from oasisx_amd import DirichletBC, LocatorMethod, fem
    from oasisx_amd import mesh as M

    mesh = M.create_unit_square(None, 10, 10)
    condition_0 = TimeDependentBC(0.1)
    bc = DirichletBC(condition_0.eval, LocatorMethod.GEOMETRICAL, _locator)
    V = fem.functionspace(mesh, ("Lagrange", int(P)))
    bc.create_bc(V)
    dofs = fem.locate_dofs_geometrical(V, _locator)
    assert dofs.shape[0] == 10 * P + 1
    for t in (0.1, 0.2, 0.3):
        u = fem.Function(V)
        u.interpolate(lambda x: np.sin(x[0]) + x[1] * t)
        u_bcx = np.zeros(V.num_dofs)
        u_bcx[dofs] = u.x.array[dofs]  # set_bc(u_bcx, dirichletbc(u, dofs))
        u_bc = fem.Function(V)
        condition_0.t = t
        bc.update_bc()
        bc.apply(u_bc.x.petsc_vec)
        assert np.allclose(u_bcx, u_bc.x.array)


@pytest.mark.parametrize("P", [1, 2, 3, 4])
@pytest.mark.parametrize("dim", [0, 1, 2])
def test_function_topological(hip, P, dim):
    from oasisx_amd import DirichletBC, LocatorMethod, fem
    from oasisx_amd import mesh as M

    mesh = M.create_unit_square(None, 10, 10)
    condition_0 = TimeDependentBC(0.1)
    entities = M.locate_entities(mesh, dim, _locator)
    value = np.int32(3)
    et = M.meshtags(mesh, dim, entities, np.full(len(entities), value, dtype=np.int32))
    bc = DirichletBC(condition_0.eval, LocatorMethod.TOPOLOGICAL, (et, value))
    V = fem.functionspace(mesh, ("Lagrange", int(P)))
    bc.create_bc(V)
    dofs = fem.locate_dofs_topological(V, dim, entities)
    # vertices on x = 1: 11; edges on x = 1: their closure = all 10 P + 1 points; no cell lies in x = 1
    assert dofs.shape[0] == {0: 11, 1: 10 * P + 1, 2: 0}[dim]
    for t in (0.1, 0.2, 0.3):
        u = fem.Function(V)
        u.interpolate(lambda x: np.sin(x[0]) + x[1] * t)
        u_bcx = np.zeros(V.num_dofs)
        u_bcx[dofs] = u.x.array[dofs]
        u_bc = fem.Function(V)
        condition_0.t = t
        bc.update_bc()
        bc.apply(u_bc.x.petsc_vec)
        assert np.allclose(u_bcx, u_bc.x.array)


@pytest.mark.parametrize("P", [1, 2, 3, 4])
def test_constant_geometrical(hip, P):
    from oasisx_amd import DirichletBC, LocatorMethod, fem
    from oasisx_amd import mesh as M

    mesh = M.create_unit_square(None, 10, 10)
    time = fem.Constant(mesh, 1.0)
    bc = DirichletBC(time, LocatorMethod.GEOMETRICAL, _locator)
    V = fem.functionspace(mesh, ("Lagrange", int(P)))
    bc.create_bc(V)
    dofs = fem.locate_dofs_geometrical(V, _locator)
    for t in (0.1, 0.2, 0.3):
        time.value += t
        u_bcx = np.zeros(V.num_dofs)
        u_bcx[dofs] = float(time.value)
        u_bc = fem.Function(V)
        bc.apply(u_bc.x.petsc_vec)
        assert np.allclose(u_bcx, u_bc.x.array)


@pytest.mark.parametrize("P", [1, 2, 3, 4])
@pytest.mark.parametrize("dim", [0, 1, 2])
def test_constant_topological(hip, P, dim):
    """Reference test/test_bcs.py:130-163: a time-dependent Constant on dofs located TOPOLOGICALLY from mesh tags
    of entity dimension 0, 1 and 2; the Constant is tracked by reference (its value changes between applications)."""
    from oasisx_amd import DirichletBC, LocatorMethod, fem
    from oasisx_amd import mesh as M

    mesh = M.create_unit_square(None, 10, 10)
    time = fem.Constant(mesh, 1.0)
    entities = M.locate_entities(mesh, dim, _locator)
    value = np.int32(3)
    et = M.meshtags(mesh, dim, entities, np.full(len(entities), value, dtype=np.int32))
    bc = DirichletBC(time, LocatorMethod.TOPOLOGICAL, (et, value))
    V = fem.functionspace(mesh, ("Lagrange", int(P)))
    bc.create_bc(V)
    dofs = fem.locate_dofs_topological(V, dim, entities)
    assert dofs.shape[0] == {0: 11, 1: 10 * P + 1, 2: 0}[dim]
    expected = 1.0
    for t in (0.1, 0.2, 0.3):
        time.value += t
        expected += t
        u_bcx = np.zeros(V.num_dofs)
        u_bcx[dofs] = float(time.value)  # set_bc(u_bcx, [dirichletbc(time, dofs, V)])
        u_bc = fem.Function(V)
        bc.apply(u_bc.x.petsc_vec)
        assert np.allclose(u_bcx, u_bc.x.array)
        assert dim == 2 or np.isclose(u_bc.x.array[dofs], expected).all()
