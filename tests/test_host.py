"""CPU: host logic of the product (mesh, spaces, SELL layout, BC location) and the C ABI surface."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oasisx_amd import fem
from oasisx_amd import mesh as M
from oracle import ipcs_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads and exports every function include/oasisx_hip.h declares
    (no compute calls: there is no GPU here)."""
    from oasisx_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "oasisx_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ox_[a-z0-9_]+)\s*\(", hdr))
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = _lib.load()
    assert L.ox_version() >= 100 and L.ox_sell_kv() == fem.KV


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: a compute call on a machine without a GPU raises, it does not emulate."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from oasisx_amd import _lib
    from oasisx_amd.la import SellMatrix

    m = M.create_unit_square(None, 3, 3, device="cpu")
    V = fem.FunctionSpace(m, 1, window=64)
    A = SellMatrix(V.pattern)
    x = torch.zeros(V.num_dofs, dtype=torch.float64)
    with pytest.raises(_lib.OasisxHipError):
        A.mult(x, x.clone(), 1)


@pytest.mark.parametrize("dim", [2, 3])
def test_mesh_generators_match_oracle_layout(dim):
    if dim == 2:
        m = M.create_rectangle(None, [[-1, -1], [1, 1]], [5, 4], device="cpu")
        c, cl = O.create_rectangle_mesh([-1, -1], [1, 1], [5, 4])
    else:
        m = M.create_box(None, [[-1, -1, -1], [1, 1, 1]], [3, 4, 2], device="cpu")
        c, cl = O.create_box_mesh([-1, -1, -1], [1, 1, 1], [3, 4, 2])
    assert np.abs(m.coords.numpy() - c).max() < 1e-15
    assert (m.cells.numpy() == cl).all()
    assert m.geometry.x.shape == (c.shape[0], 3)


@pytest.mark.parametrize("dim,N,deg", [(2, 7, 1), (2, 6, 2), (3, 4, 1), (3, 3, 2)])
def test_function_space_layout(dim, N, deg):
    m = (M.create_rectangle(None, [[-1, -1], [1, 1]], [N, N], device="cpu") if dim == 2
         else M.create_box(None, [[-1, -1, -1], [1, 1, 1]], [N, N, N], device="cpu"))
    V = fem.FunctionSpace(m, deg, window=128)
    P = V.pattern
    n = V.num_dofs
    # the numbering is a permutation and cell_dofs is consistent with the dof coordinates
    assert sorted(V._rank_initial.tolist()) == list(range(n))
    x = V.x.numpy()
    cd = V.cell_dofs.numpy()
    kc = V.cells_in_kernel_order()
    vx = m.coords.numpy()[kc]
    assert np.abs(x[cd[:, : dim + 1]] - vx).max() < 1e-15
    if deg == 2:
        for e, (a, b) in enumerate(fem.local_edges(dim)):
            assert np.abs(x[cd[:, dim + 1 + e]] - 0.5 * (vx[:, a] + vx[:, b])).max() < 1e-15
    # the pattern equals the oracle's on the same numbering
    F = O.Forms(m.coords.numpy(), kc, deg, 1, vd=cd, qd=kc, nv_dofs=n,
                nq_dofs=m.num_vertices)
    Mo = F.mass_v()
    assert P.nnz == Mo.nnz
    vals = P.values_from_csr(Mo)
    assert abs(P.to_csr(vals) - Mo).max() == 0.0
    # rows are sorted by decreasing length inside each window; slice widths cover their rows
    rl = P.row_len.numpy()
    for w0 in range(0, n, 128):
        seg = rl[w0:w0 + 128]
        assert (np.diff(seg) <= 0).all()
    for s in range(P.n_slices):
        assert P.widths[s] >= rl[s * 64:(s + 1) * 64].max() and P.widths[s] % fem.KV == 0
    # padding slots point at the row itself (value 0 keeps SpMV exact)
    rows, k = P.slot_rows_k()
    pad = (rows < n) & (k >= np.concatenate([rl, np.zeros(P.n_slices * 64 - n, int)])[rows])
    assert (P.cols.numpy()[pad] == rows[pad]).all()
    # adjacency: every (dof, cell) pair once, positions point at the right columns
    A = V.adj
    cell = A.adj_cell.numpy()
    assert (cell >= 0).sum() == cd.size
    ap = A.adj_ptr.numpy()
    sp_ = P.slice_ptr.numpy()
    cols = P.cols.numpy()
    pos = A.adj_pos.numpy()
    loc = A.adj_loc.numpy()
    pidx = np.nonzero(cell >= 0)[0]
    s = np.searchsorted(ap, pidx, side="right") - 1
    lane = (pidx - ap[s]) % 64
    r = s * 64 + lane
    assert (cd[cell[pidx], loc[pidx]] == r).all()
    for j in range(V.nd):
        kk = pos[pidx, j].astype(np.int64)
        off = sp_[s] + (kk // 2) * 128 + lane * 2 + kk % 2
        assert (cols[off] == cd[cell[pidx], j]).all()


def test_locate_dofs_and_meshtags():
    m = M.create_rectangle(None, [[-1, -1], [1, 1]], [6, 6], device="cpu")
    V = fem.FunctionSpace(m, 2, window=64)
    dim = m.topology.dim - 1
    m.topology.create_connectivity(dim, dim + 1)
    facets = M.exterior_facet_indices(m.topology)
    assert facets.shape[0] == 4 * 6
    tags = M.meshtags(m, dim, np.sort(facets), np.full(facets.shape, 3, dtype=np.int32))
    assert (tags.find(3) == np.sort(facets)).all() and tags.dim == dim
    topo = fem.locate_dofs_topological(V, dim, tags.find(3))
    geo = fem.locate_dofs_geometrical(V, lambda x: np.isclose(np.abs(x[0]), 1) | np.isclose(np.abs(x[1]), 1))
    assert (np.sort(topo) == np.sort(geo)).all() and topo.shape[0] == 4 * 12
    left = M.locate_entities_boundary(m, dim, lambda x: np.isclose(x[0], -1))
    assert left.shape[0] == 6
    with pytest.raises(RuntimeError):
        M.meshtags(m, dim, facets[::-1].copy(), np.zeros(facets.shape, dtype=np.int32))
    # 3-D: boundary faces of a box
    b = M.create_box(None, [[0, 0, 0], [1, 1, 1]], [2, 2, 2], device="cpu")
    assert M.exterior_facet_indices(b.topology).shape[0] == 6 * 2 * 2 * 2


def test_dirichlet_bc_host_side():
    """reference test/test_bcs.py restated for the parts that do not need the device: the dofs and
    values a DirichletBC will impose (geometrical / topological, float / Constant / callable)."""
    from oasisx_amd import DirichletBC, LocatorMethod

    m = M.create_unit_square(None, 5, 5, device="cpu")
    for P in (1, 2, 3, 4):  # test_bcs.py:19 parametrises P = 1..4
        V = fem.functionspace(m, ("Lagrange", P))
        assert V.num_dofs == (5 * P + 1) ** 2
        X = V.tabulate_dof_coordinates()
        assert len(np.unique(np.round(X[:, :2] * 1e9).astype(np.int64), axis=0)) == V.num_dofs
        clock = {"t": 0.1}
        f = lambda x: x[0] + 2 * x[1] ** 2 + clock["t"]  # noqa: E731
        bc = DirichletBC(f, LocatorMethod.GEOMETRICAL, lambda x: np.isclose(x[0], 0))
        bc.create_bc(V)
        exp_dofs = np.nonzero(np.isclose(X[:, 0], 0))[0]
        assert (np.sort(bc._dofs) == exp_dofs).all()
        for t in (0.1, 0.2, 0.3):
            clock["t"] = t
            bc.update_bc()
            assert np.allclose(bc.values_host(), f(X[bc._dofs].T))
        c = fem.Constant(m, 3.0)
        bcc = DirichletBC(c, LocatorMethod.GEOMETRICAL, lambda x: np.isclose(x[1], 1))
        bcc.create_bc(V)
        assert np.allclose(bcc.values_host(), 3.0)
        c.value = 5.0  # tracked by reference (test_bcs.py:100-125)
        assert np.allclose(bcc.values_host(), 5.0)
        dim = m.topology.dim - 1
        facets = M.locate_entities_boundary(m, dim, lambda x: np.isclose(x[0], 1))
        tags = M.meshtags(m, dim, facets, np.full(facets.shape, 2, dtype=np.int32))
        bct = DirichletBC(1.5, LocatorMethod.TOPOLOGICAL, (tags, 2))
        bct.create_bc(V)
        assert (np.sort(bct._dofs) == np.nonzero(np.isclose(X[:, 0], 1))[0]).all()
        assert bct._bc._cpp_object.dof_indices()[1] == bct._dofs.shape[0]


def test_field_storage_and_function_views():
    m = M.create_unit_square(None, 3, 3, device="cpu")
    V = fem.FunctionSpace(m, 2, window=64)
    W = fem.VectorFunctionSpace(V, 2)
    S = fem.FieldStorage(V.num_dofs, 2, "cpu")
    u0, u1 = fem.Function(V, "a", S, 0), fem.Function(V, "b", S, 1)
    u0.interpolate(lambda x: x[0])
    u1.x.array[:] = 7.0
    w = fem.Function(W, "w", S, None)
    assert w.x.array.shape[0] == 2 * V.num_dofs
    assert np.allclose(w.x.array[1::2], 7.0) and np.allclose(w.x.array[0::2], V.tabulate_dof_coordinates()[:, 0])
    Vs, idx = W.sub(1).collapse()
    assert Vs is V and (idx == np.arange(V.num_dofs) * 2 + 1).all()


def test_lattice_detection_and_curve_order():
    """fem.mesh_is_lattice: box meshes are lattices (tiled lexicographic order), a Delaunay triangulation of
    jittered points is not (Z-order curve); the Z-order key keeps near points near in the numbering."""
    import torch

    from oasisx_amd import fem
    from oasisx_amd import mesh as M

    box = M.create_box(None, [[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]], [6, 7, 5], device="cpu")
    rect = M.create_rectangle(None, [[0.0, 0.0], [1.0, 2.0]], [9, 12], device="cpu")
    dl = M.create_delaunay_box(None, [[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]], 6, seed=1, device="cpu")
    assert fem.mesh_is_lattice(box) and fem.mesh_is_lattice(rect) and not fem.mesh_is_lattice(dl)
    x = dl.coords
    lo = x.min(dim=0).values
    span = x.max(dim=0).values - lo
    key = fem.locality_key(x, lo, span, 1, 9, curve=True)
    assert key.unique().numel() == x.shape[0]  # 18 bits per coordinate: every jittered point has its own key
    order = torch.argsort(key)
    # consecutive points of the curve are close on average (much closer than random pairs)
    d_curve = (x[order][1:] - x[order][:-1]).norm(dim=1).mean()
    perm = torch.randperm(x.shape[0], generator=torch.Generator().manual_seed(0))
    d_rand = (x[perm][1:] - x[perm][:-1]).norm(dim=1).mean()
    assert float(d_curve) < 0.5 * float(d_rand)


def test_small_width_bins_are_merged_forward():
    """fem.merge_small_bins: ascending widths, every slice kept, a merged bin is at least as wide as every
    slice in it, and only the last bin may stay below the threshold."""
    from oasisx_amd import fem

    widths = np.array([10, 14, 16, 18, 20, 22, 24, 26, 28, 32, 34, 36, 120], dtype=np.int32)
    counts = np.array([1, 25, 590, 29, 1060, 12, 1160, 3, 880, 530, 1, 26, 2])
    ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    bw, bp = fem.merge_small_bins(widths, ptr)
    n = int(ptr[-1])
    thresh = max(64, n // 64)
    assert bp[0] == 0 and bp[-1] == n and (np.diff(bw) > 0).all() and len(bw) < len(widths)
    for b in range(len(bw)):
        inside = widths[(ptr[1:] > bp[b]) & (ptr[:-1] < bp[b + 1])]
        assert inside.max() == bw[b]  # LDS sized for the widest slice of the merged bin
        assert bp[b + 1] - bp[b] >= thresh or b == len(bw) - 1


def test_high_order_lagrange_space_3d():
    """fem.HighOrderLagrangeSpace on tetrahedra: P3 / P4 dof counts (vertices + (p-1) per edge + (p-1)(p-2)/2 per face
    + (p-1)(p-2)(p-3)/6 per cell), distinct dof coordinates, and the dofs a DirichletBC finds on a face of the cube --
    geometrically and through the face's facets -- are the (p N + 1)^2 lattice points of that face."""
    from oasisx_amd import DirichletBC, LocatorMethod

    N = 2
    m = M.create_unit_cube(None, N, N, N, device="cpu")
    nv = m.num_vertices
    ne, nf, nc = (m._entities(k)[0].shape[0] for k in (1, 2, 3))
    for p in (3, 4):
        V = fem.functionspace(m, ("Lagrange", p))
        assert V.num_dofs == nv + (p - 1) * ne + (p - 1) * (p - 2) // 2 * nf + (p - 1) * (p - 2) * (p - 3) // 6 * nc
        assert V.num_dofs == (p * N + 1) ** 3  # the box mesh's dofs are the points of the refined lattice
        X = V.tabulate_dof_coordinates()
        assert len(np.unique(np.round(X * 1e9).astype(np.int64), axis=0)) == V.num_dofs
        on_face = lambda x: np.isclose(x[2], 1.0)  # noqa: E731
        bc = DirichletBC(lambda x: x[0] + 2 * x[1], LocatorMethod.GEOMETRICAL, on_face)
        bc.create_bc(V)
        assert bc._dofs.shape[0] == (p * N + 1) ** 2
        facets = M.locate_entities_boundary(m, 2, on_face)
        tags = M.meshtags(m, 2, facets, np.full(facets.shape, 7, dtype=np.int32))
        bct = DirichletBC(1.0, LocatorMethod.TOPOLOGICAL, (tags, 7))
        bct.create_bc(V)
        assert (np.sort(bct._dofs) == np.sort(bc._dofs)).all()
        assert np.allclose(bc.values_host(), X[bc._dofs, 0] + 2 * X[bc._dofs, 1])


def test_p3_basis_matches_the_oracle_and_is_nodal():
    """fem.lagrange_basis / lagrange_basis_derivs (host plumbing of load vectors, PressureBC and error functionals) against
    the oracle's tabulation, for every built element; P3 (gll_warped) is nodal at its own nodes."""
    for d, deg in ((2, 1), (2, 2), (2, 3), (3, 1), (3, 2)):
        bary, _ = O.simplex_quadrature(d, 4)
        phi, dphi = O.tabulate(d, deg, bary)
        assert np.abs(fem.lagrange_basis(d, deg, bary) - phi).max() < 1e-13
        assert np.abs(fem.lagrange_basis_derivs(d, deg, bary) - dphi).max() < 1e-12
    nodes = O.p3_nodes_2d()
    assert np.abs(fem.lagrange_basis(2, 3, nodes) - np.eye(10)).max() < 1e-13
    assert abs(fem.GLL3[0] - O.GLL3[0]) < 1e-16 and abs(fem.GLL3[0] + fem.GLL3[1] - 1.0) < 1e-16


@pytest.mark.parametrize("dim", [2, 3])
def test_uniform_refinement_is_conforming(dim):
    """mesh.refine_uniform: 2^dim children per cell, the same volume, every interior facet shared by exactly two cells,
    the old vertices first; twice on a Delaunay mesh (what bench.py's unstructured leg is built from)."""
    import itertools

    X, T = M.refine_uniform(*_delaunay_arrays(dim))
    X0, T0 = _delaunay_arrays(dim)
    assert T.shape[0] == T0.shape[0] * 2 ** dim and np.array_equal(X[: X0.shape[0]], X0)
    X, T = M.refine_uniform(X, T)

    def vol(P, C):
        J = np.stack([P[C[:, a]] - P[C[:, 0]] for a in range(1, dim + 1)], axis=2)
        return np.abs(np.linalg.det(J)).sum()
    assert abs(vol(X, T) - vol(X0, T0)) < 1e-12 * vol(X0, T0) and np.abs(np.linalg.det(
        np.stack([X[T[:, a]] - X[T[:, 0]] for a in range(1, dim + 1)], axis=2))).min() > 0
    f = np.concatenate([np.sort(T[:, list(c)], axis=1) for c in itertools.combinations(range(dim + 1), dim)])
    _, cnt = np.unique(f, axis=0, return_counts=True)
    assert set(np.unique(cnt).tolist()) <= {1, 2}
    f0 = np.concatenate([np.sort(T0[:, list(c)], axis=1) for c in itertools.combinations(range(dim + 1), dim)])
    _, cnt0 = np.unique(f0, axis=0, return_counts=True)
    assert (cnt == 1).sum() == (cnt0 == 1).sum() * 2 ** (2 * (dim - 1))  # every boundary facet became 4^... children


def _delaunay_arrays(dim):
    m = M.create_delaunay_box(None, [[-1.0] * dim, [1.0] * dim], 4, device="cpu")
    return m.coords.numpy().copy(), m.cells.numpy().astype(np.int64)


def test_scaling_model_reproduces_the_committed_prediction():
    """tools/scaling_model.py (the step-time model of the predicted 1/2/4/8-GPU curves) applied to the per-rank costs and the
    iteration profile stored in profiles/r05_predicted_scaling.json gives the ms/step stored beside them -- the model the
    files were made with is the one in the tree -- and tools/predict_workloads.py runs on the CPU."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    try:
        import scaling_model as sm
    finally:
        sys.path.pop(0)
    d = json.loads(open(os.path.join(root, "profiles", "r05_predicted_scaling.json")).readline())
    prof = d["iteration_profile_per_step"]
    for P, e in d["P"].items():
        ms = max(sum(sm.predict(m, prof, int(P)).values()) for m in e["ranks"].values())
        assert abs(ms - e["ms_per_step"]) < 1e-9 * e["ms_per_step"], (P, ms, e["ms_per_step"])
    assert d["label"].startswith("PREDICTION")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "predict_workloads.py")], capture_output=True, text=True,
                         check=True).stdout
    w = json.loads(out)
    assert w["label"].startswith("PREDICTION") and len(w["workloads"]) == 2
    for leg in w["workloads"].values():
        assert abs(leg["model_error_vs_measured_P1"]) < 0.05 and set(leg["P"]) == {"1", "2", "4", "8"}
