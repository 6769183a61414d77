"""Field output (oasisx_amd.io.VTXWriter, the stand-in for dolfinx.io.VTXWriter in the reference
demo, demo/taylor_green.py:183-184,211-215): files parse, values and times round-trip, and the
quadratic cells use VTK's node order."""
import os

import numpy as np
import pytest

from oasisx_amd import fem, io
from oasisx_amd import mesh as M

_EDGES = {22: [(0, 1), (1, 2), (2, 0)], 24: [(0, 1), (1, 2), (0, 2), (0, 3), (1, 3), (2, 3)]}


@pytest.mark.parametrize("dim,N,deg", [(2, 4, 2), (3, 2, 2), (3, 3, 1), (2, 5, 1)])
def test_vtu_series_round_trip(tmp_path, dim, N, deg):
    mesh = (M.create_rectangle(None, [[-1.0, -1.0], [1.0, 1.0]], [N, N], device="cpu") if dim == 2
            else M.create_box(None, [[0.0, 0.0, 0.0], [1.0, 2.0, 3.0]], [N, N, N], device="cpu"))
    V = fem.functionspace(mesh, ("Lagrange", deg))
    W = fem.VectorFunctionSpace(V, dim)
    p = fem.Function(V, name="p")
    u = fem.Function(W, name="u")
    w = io.VTXWriter(mesh.comm, str(tmp_path / "out" / "fields.bp"), [u, p], engine="BP4")
    times = [0.5, 0.75, 1.0]
    for t in times:
        p.interpolate(lambda x: t * (x[0] + 2 * x[1] - 3 * x[2]))
        u.interpolate(lambda x: np.stack([t + x[k] for k in range(dim)]))
        w.write(t)
    w.close()
    with pytest.raises(RuntimeError):
        w.write(2.0)
    pvd = (tmp_path / "out" / "fields.pvd").read_text()
    X = V.tabulate_dof_coordinates()
    for k, t in enumerate(times):
        name = f"fields_{k:06d}.vtu"
        assert name in pvd and repr(t) in pvd
        d = io.read_vtu(str(tmp_path / "out" / name))
        assert d["time"] == t
        np.testing.assert_array_equal(d["points"][:, :dim], X[:, :dim])
        np.testing.assert_allclose(d["point_data"]["p"], t * (X[:, 0] + 2 * X[:, 1] - 3 * X[:, 2]), atol=1e-14)
        np.testing.assert_allclose(d["point_data"]["u"][:, :dim], t + X[:, :dim], atol=1e-14)
        assert (d["point_data"]["u"][:, dim:] == 0).all()
        nper = {(2, 1): 3, (3, 1): 4, (2, 2): 6, (3, 2): 10}[(dim, deg)]
        conn = d["connectivity"].reshape(-1, nper)
        assert conn.shape[0] == mesh.num_cells and (d["offsets"] == nper * (1 + np.arange(mesh.num_cells))).all()
        vt = int(d["types"][0])
        assert vt == {(2, 1): 5, (3, 1): 10, (2, 2): 22, (3, 2): 24}[(dim, deg)]
        if deg == 2:  # mid-edge nodes sit between the end points VTK prescribes
            nv = dim + 1
            for j, (a, b) in enumerate(_EDGES[vt]):
                mid = 0.5 * (d["points"][conn[:, a]] + d["points"][conn[:, b]])
                np.testing.assert_allclose(d["points"][conn[:, nv + j]], mid, atol=1e-14)
        # every point is used, cells have positive size
        assert np.unique(conn).shape[0] == V.num_dofs


def test_functions_on_different_spaces_are_refused(tmp_path):
    mesh = M.create_unit_square(None, 3, 3, device="cpu")
    V1, V2 = fem.functionspace(mesh, ("Lagrange", 1)), fem.functionspace(mesh, ("Lagrange", 2))
    with pytest.raises(RuntimeError):
        io.VTXWriter(mesh.comm, str(tmp_path / "x.bp"), [fem.Function(V1), fem.Function(V2)])


MSH22 = """$MeshFormat
2.2 0 8
$EndMeshFormat
$PhysicalNames
2
1 7 "inlet"
2 9 "fluid"
$EndPhysicalNames
$Nodes
5
1 0 0 0
2 1 0 0
4 1 1 0
7 0 1 0
9 5 5 0
$EndNodes
$Elements
5
1 1 2 7 1 1 7
2 1 2 8 2 2 4
3 2 2 9 1 1 2 4
4 2 2 9 1 1 4 7
5 15 2 3 1 1
$EndElements
"""

MSH41 = """$MeshFormat
4.1 0 8
$EndMeshFormat
$Entities
4 4 1 0
1 0 0 0 0
2 1 0 0 0
3 1 1 0 0
4 0 1 0 0
1 0 0 0 1 0 0 0 2 1 -2
2 1 0 0 1 1 0 1 8 2 2 -3
3 0 1 0 1 1 0 0 2 3 -4
4 0 0 0 0 1 0 1 7 2 4 -1
1 0 0 0 1 1 0 1 9 4 1 2 3 4
$EndEntities
$Nodes
1 4 1 4
2 1 0 4
1
2
3
4
0 0 0
1 0 0
1 1 0
0 1 0
$EndNodes
$Elements
3 4 1 4
1 2 1 1
1 2 3
1 4 1 1
2 4 1
2 1 2 2
3 1 2 3
4 1 3 4
$EndElements
"""


@pytest.mark.parametrize("text", [MSH22, MSH41])
def test_gmsh_reader_with_physical_groups(tmp_path, text):
    """Gmsh ASCII .msh (2.2 and 4.1), the route the reference's users take into DOLFINx (gmshio): mesh, cell tags and
    facet tags; node ids with gaps and unused nodes; the tagged lines land on the right mesh facets."""
    from oasisx_amd import mesh as M

    path = str(tmp_path / "square.msh")
    open(path, "w").write(text)
    mesh, ct, ft = M.read_gmsh(path, device="cpu")
    assert mesh.gdim == 2 and mesh.num_vertices == 4 and mesh.num_cells == 2
    assert (ct.values == 9).all() and ct.dim == 2
    x = mesh.coords.numpy()
    ev, _ = mesh._entities(1)
    assert ft.dim == 1 and sorted(ft.values.tolist()) == [7, 8]
    for tag, xval in ((7, 0.0), (8, 1.0)):
        f = ft.find(tag)
        assert f.shape[0] == 1 and np.allclose(x[ev[f[0]], 0], xval)  # inlet on x = 0, the other line on x = 1
    assert (np.diff(ft.indices) > 0).all()
    # the area, and the same mesh through read_mesh / import_mesh
    J = x[mesh.cells.numpy()[:, 1:]] - x[mesh.cells.numpy()[:, :1]]
    assert abs(np.abs(np.linalg.det(J)).sum() / 2 - 1.0) < 1e-14
    assert M.read_mesh(path, device="cpu").num_cells == 2


def test_gmsh_reader_refuses_binary_files(tmp_path):
    from oasisx_amd import mesh as M

    path = str(tmp_path / "b.msh")
    open(path, "w").write("$MeshFormat\n4.1 1 8\n$EndMeshFormat\n")
    with pytest.raises(ValueError, match="binary"):
        M.read_gmsh(path)


MSH22_3D = """$MeshFormat
2.2 0 8
$EndMeshFormat
$Nodes
5
1 0 0 0
2 1 0 0
3 0 1 0
4 0 0 1
5 1 1 1
$EndNodes
$Elements
5
1 2 2 11 1 1 2 3
2 2 2 12 2 2 3 5
3 4 2 20 1 1 2 3 4
4 4 2 21 2 2 3 4 5
5 15 2 3 1 1
$EndElements
"""


def test_gmsh_reader_tetrahedra_with_tagged_triangles(tmp_path):
    """A 3-D .msh: tetrahedra become the cells (two volume tags), tagged triangles become facet tags on the mesh's own
    facet numbering (one exterior face of each tetrahedron)."""
    from oasisx_amd import mesh as M

    path = str(tmp_path / "two_tets.msh")
    open(path, "w").write(MSH22_3D)
    mesh, ct, ft = M.read_gmsh(path, device="cpu")
    assert mesh.gdim == 3 and mesh.num_vertices == 5 and mesh.num_cells == 2
    assert sorted(ct.values.tolist()) == [20, 21] and ct.dim == 3
    fv, _ = mesh._entities(2)
    x = mesh.coords.numpy()
    assert ft.dim == 2 and sorted(ft.values.tolist()) == [11, 12]
    f11, f12 = ft.find(11), ft.find(12)
    assert f11.shape[0] == 1 and f12.shape[0] == 1
    assert np.allclose(x[fv[f11[0]], 2], 0.0)  # the triangle (1, 2, 3) lies in z = 0
    assert {tuple(p) for p in x[fv[f12[0]]].tolist()} == {(1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (1.0, 1.0, 1.0)}
    # volumes: 1/6 and 1/3
    c = mesh.cells.numpy()
    vol = np.abs(np.linalg.det(x[c[:, 1:]] - x[c[:, :1]])) / 6.0
    assert np.allclose(sorted(vol.tolist()), [1.0 / 6.0, 1.0 / 3.0])
    # a P2 space on it: 5 vertices + 9 edges
    from oasisx_amd import fem
    assert fem.functionspace(mesh, ("Lagrange", 2)).num_dofs == 14
