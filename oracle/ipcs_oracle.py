"""
TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy/scipy) of the per-time-step
IPCS hot path of oasisx's ``FractionalStep_AB_CN``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Nothing under ``oasisx_amd/`` imports it: the
product path is the HIP library and fails loudly without it.

PARITY STATUS: the arithmetic of this path lives in un-vendored third-party
dependencies of the reference (``fenics-dolfinx>=0.10`` with Basix/UFL/FFCx,
PETSc/petsc4py -- reference ``pyproject.toml:13``) that are absent from this
container (ordinary ModuleNotFoundError, no permission denial).  The
reference's own tests hold no numeric golden vectors (they compare against
DOLFINx itself at run time), so this oracle is pinned by

  1. the analytic 2-D Taylor-Green solution the demo uses
     (reference ``demo/taylor_green.py:36-53,176-191``),
  2. the restated ``test_tentative`` identity: split-operator A / rhs1 equal a
     monolithic assembly of the tentative-velocity form
     (reference ``test/test_tentative_velocity.py:43-84,213-235``),
  3. exactness identities (sum M = |Omega|, K.1 = 0, C.1 = 0, closed-form
     reference-element matrices),
  4. scipy's ``splu`` / ``cg`` / ``bicgstab`` on the same matrices.

``pressure_assemble``, ``pressure_solve``, ``velocity_update`` and ``solve``
are untested in the reference itself: for those rows parity is pinned by
(1), (3), (4) only -- "parity unpinned" by reference fixtures.

Every function cites the reference file:line it restates (paths relative to
/root/reference).  All arithmetic is float64.

What is restated from the third-party semantics (public documentation):
  * Lagrange P1/P2 on affine simplices (``gll_warped`` == equispaced for
    degree <= 2, reference fracstep.py:170,181);
  * every form here is polynomial on affine cells and FFCx integrates it with
    a rule of its total degree, i.e. exactly -> any exact rule reproduces the
    same matrices to round-off.  This file uses collapsed Gauss-Jacobi rules
    (independent of the 7/14-point rules the HIP kernels use);
  * ``assemble_*`` adds into its target, ``set_bc`` overwrites with g,
    ``zeroRowsLocal(rows, 1.0)`` keeps columns, constant-nullspace ``remove``
    subtracts the arithmetic mean.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from scipy.special import roots_jacobi

# ----------------------------------------------------------------------------
# Reference simplex: quadrature and Lagrange tabulation
# ----------------------------------------------------------------------------


def simplex_quadrature(d: int, n: int):
    """Collapsed Gauss-Jacobi rule on the reference d-simplex, exact to degree
    2n-1.  Returns barycentric points (Q, d+1) and weights (Q,) with
    sum(w) = 1/d!  (so  int_cell f = |detJ| * sum_q w_q f(x_q))."""
    if d == 2:
        s1, w1 = roots_jacobi(n, 1.0, 0.0)
        s0, w0 = roots_jacobi(n, 0.0, 0.0)
        x = (1 + s1) / 2
        wx = w1 / 4
        t = (1 + s0) / 2
        wt = w0 / 2
        X, T = np.meshgrid(x, t, indexing="ij")
        W = np.outer(wx, wt)
        px = X.ravel()
        py = ((1 - X) * T).ravel()
        bary = np.stack([1 - px - py, px, py], axis=1)
        return bary, W.ravel()
    if d == 3:
        s2, w2 = roots_jacobi(n, 2.0, 0.0)
        s1, w1 = roots_jacobi(n, 1.0, 0.0)
        s0, w0 = roots_jacobi(n, 0.0, 0.0)
        x = (1 + s2) / 2
        wx = w2 / 8
        t = (1 + s1) / 2
        wt = w1 / 4
        s = (1 + s0) / 2
        ws = w0 / 2
        X, T, S = np.meshgrid(x, t, s, indexing="ij")
        W = wx[:, None, None] * wt[None, :, None] * ws[None, None, :]
        px = X.ravel()
        py = ((1 - X) * T).ravel()
        pz = ((1 - X) * (1 - T) * S).ravel()
        bary = np.stack([1 - px - py - pz, px, py, pz], axis=1)
        return bary, W.ravel()
    raise ValueError(d)


def local_edges(d: int):
    """Local edge -> (vertex a, vertex b) table used for the P2 edge dofs."""
    if d == 2:
        return [(1, 2), (0, 2), (0, 1)]
    return [(2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1)]


def local_faces(d: int):
    """Local face f of a tetrahedron = the three vertices other than f (the P3 face dofs, in this order)."""
    if d != 3:
        raise ValueError("faces carry dofs on tetrahedra only")
    return [(1, 2, 3), (0, 2, 3), (0, 1, 3), (0, 1, 2)]


def num_cell_dofs(d: int, degree: int) -> int:
    if degree == 3:
        return 10 if d == 2 else 20  # vertices + 2 per edge + (the cell's own dof | one per face)
    return d + 1 if degree == 1 else (d + 1) + len(local_edges(d))


# Lagrange degree 3 on triangles, ``lagrange_variant=gll_warped`` (reference fracstep.py:170,181): the two interior
# nodes of an edge sit at the Gauss-Lobatto-Legendre points of the edge, (1 -+ 1/sqrt 5)/2, instead of 1/3 and 2/3;
# the interior node stays at the centroid (Basix warps the equispaced lattice by the 1-D GLL displacement along every
# barycentric direction: documented in its "variants" demo; for degree <= 2 the variant coincides with equispaced).
GLL3 = (0.5 - 0.5 / math.sqrt(5.0), 0.5 + 0.5 / math.sqrt(5.0))


def p3_nodes(d: int) -> np.ndarray:
    """Barycentric coordinates (nd, d + 1) of the P3 nodes: vertices, then per local edge (a, b) the node nearer a and
    the node nearer b, then -- triangles -- the centroid, or -- tetrahedra -- the centroids of the four faces."""
    nv = d + 1
    nodes = [np.eye(nv)[a] for a in range(nv)]
    for a, b in local_edges(d):
        for t in GLL3:
            v = np.zeros(nv)
            v[a], v[b] = 1.0 - t, t
            nodes.append(v)
    if d == 2:
        nodes.append(np.full(3, 1.0 / 3.0))
    else:
        for f in local_faces(3):
            v = np.zeros(4)
            v[list(f)] = 1.0 / 3.0
            nodes.append(v)
    return np.array(nodes)


def p3_nodes_2d() -> np.ndarray:
    return p3_nodes(2)


def _p3_exponents(d: int):
    if d == 2:
        return [(i, j) for i in range(4) for j in range(4 - i)]
    return [(i, j, k) for i in range(4) for j in range(4 - i) for k in range(4 - i - j)]


def _p3_monomials(*xs):
    """The monomials of degree <= 3 in d = len(xs) variables (10 / 20) and their partial derivatives at the points:
    (m, dm_0, ..., dm_{d-1})."""
    d = len(xs)
    ex = _p3_exponents(d)

    def mono(e, skip=None):
        out = np.ones_like(xs[0])
        for v in range(d):
            pw = e[v] - (1 if v == skip else 0)
            out = out * xs[v] ** max(pw, 0)
        return out
    m = np.stack([mono(e) for e in ex], axis=1)
    ders = []
    for v in range(d):
        ders.append(np.stack([e[v] * mono(e, skip=v) if e[v] > 0 else np.zeros_like(xs[0]) for e in ex], axis=1))
    return (m, *ders)


_P3_COEF = {}


def _p3_coefficients(d: int = 2):
    if d not in _P3_COEF:
        n = p3_nodes(d)
        V = _p3_monomials(*[n[:, a] for a in range(1, d + 1)])[0]  # reference coordinates = (lambda_1, ..., lambda_d)
        _P3_COEF[d] = np.linalg.inv(V)  # column i: monomial coefficients of phi_i
    return _P3_COEF[d]


def tabulate(d: int, degree: int, bary: np.ndarray):
    """Lagrange basis on the reference simplex at barycentric points.
    Returns phi (Q, nd) and dphi (Q, nd, d+1) = d phi / d lambda_b (lambdas
    treated as independent; grad phi = sum_b dphi_b grad lambda_b)."""
    Q = bary.shape[0]
    nv = d + 1
    if degree == 1:
        phi = bary.copy()
        dphi = np.zeros((Q, nv, nv))
        for a in range(nv):
            dphi[:, a, a] = 1.0
        return phi, dphi
    if degree == 2:
        edges = local_edges(d)
        nd = nv + len(edges)
        phi = np.zeros((Q, nd))
        dphi = np.zeros((Q, nd, nv))
        for a in range(nv):
            phi[:, a] = bary[:, a] * (2 * bary[:, a] - 1)
            dphi[:, a, a] = 4 * bary[:, a] - 1
        for e, (a, b) in enumerate(edges):
            phi[:, nv + e] = 4 * bary[:, a] * bary[:, b]
            dphi[:, nv + e, a] = 4 * bary[:, b]
            dphi[:, nv + e, b] = 4 * bary[:, a]
        return phi, dphi
    if degree == 3:
        # phi as a polynomial of (lambda_1, ..., lambda_d) alone (lambda_0 = 1 - sum eliminated): its derivative with
        # respect to lambda_0 is then 0 and grad phi = sum_{b >= 1} d phi / d lambda_b grad lambda_b
        Cf = _p3_coefficients(d)
        tabs = _p3_monomials(*[bary[:, a] for a in range(1, d + 1)])
        phi = tabs[0] @ Cf
        dphi = np.zeros((Q, phi.shape[1], nv))
        for b in range(1, nv):
            dphi[:, :, b] = tabs[b] @ Cf
        return phi, dphi
    raise ValueError("Lagrange degree 1, 2 and 3 are restated")


# ----------------------------------------------------------------------------
# Meshes (DOLFINx generators, restated from their documented layout)
# ----------------------------------------------------------------------------


def create_rectangle_mesh(p0, p1, n):
    """dolfinx.mesh.create_rectangle(..., CellType.triangle), default
    DiagonalType.right: each quad (v0 v1 / v2 v3) -> [v0,v1,v3],[v0,v2,v3].
    Used by reference demo/taylor_green.py:126-131 and the tests' unit square."""
    nx, ny = n
    xs = np.linspace(p0[0], p1[0], nx + 1)
    ys = np.linspace(p0[1], p1[1], ny + 1)
    X, Y = np.meshgrid(xs, ys, indexing="xy")  # row = iy
    coords = np.stack([X.ravel(), Y.ravel()], axis=1)
    ix, iy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    v0 = (iy * (nx + 1) + ix).ravel()
    v1 = v0 + 1
    v2 = v0 + (nx + 1)
    v3 = v2 + 1
    cells = np.empty((2 * nx * ny, 3), dtype=np.int64)
    cells[0::2] = np.stack([v0, v1, v3], axis=1)
    cells[1::2] = np.stack([v0, v2, v3], axis=1)
    return coords, cells


def create_box_mesh(p0, p1, n):
    """dolfinx.mesh.create_box(..., CellType.tetrahedron): every hexahedron is
    cut into 6 tetrahedra that share the main diagonal v0-v7."""
    nx, ny, nz = n
    xs = np.linspace(p0[0], p1[0], nx + 1)
    ys = np.linspace(p0[1], p1[1], ny + 1)
    zs = np.linspace(p0[2], p1[2], nz + 1)
    Z, Y, X = np.meshgrid(zs, ys, xs, indexing="ij")
    coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    iz, iy, ix = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    sx, sy = 1, nx + 1
    sz = (nx + 1) * (ny + 1)
    v0 = (iz * sz + iy * sy + ix).ravel()
    v1 = v0 + sx
    v2 = v0 + sy
    v3 = v0 + sx + sy
    v4 = v0 + sz
    v5 = v4 + sx
    v6 = v4 + sy
    v7 = v4 + sx + sy
    tets = [
        (v0, v1, v3, v7),
        (v0, v1, v7, v5),
        (v0, v5, v7, v4),
        (v0, v3, v2, v7),
        (v0, v6, v4, v7),
        (v0, v2, v6, v7),
    ]
    cells = np.empty((6 * nx * ny * nz, 4), dtype=np.int64)
    for k, t in enumerate(tets):
        cells[k::6] = np.stack(t, axis=1)
    return coords, cells


def cell_geometry(coords: np.ndarray, cells: np.ndarray):
    """Affine geometry: G[c, a, :] = grad lambda_a, adet[c] = |det J|."""
    d = coords.shape[1]
    x0 = coords[cells[:, 0]]
    J = np.stack([coords[cells[:, a]] - x0 for a in range(1, d + 1)], axis=2)  # (nc, d, d) columns
    det = np.linalg.det(J)
    Jinv = np.linalg.inv(J)  # rows = grad lambda_1..d
    G = np.empty((cells.shape[0], d + 1, d))
    G[:, 1:, :] = Jinv
    G[:, 0, :] = -Jinv.sum(axis=1)
    return G, np.abs(det)


def build_dofmap(cells: np.ndarray, nverts: int, degree: int):
    """Lagrange dofmap: vertex dofs = vertex ids, P2 edge dofs = nverts + edge id
    (edges numbered by first appearance of the sorted vertex pair)."""
    d = cells.shape[1] - 1
    if degree == 1:
        return cells.copy(), nverts, None
    edges = local_edges(d)
    pairs = np.stack(
        [np.sort(cells[:, [a, b]], axis=1) for (a, b) in edges], axis=1
    )  # (nc, ne, 2)
    key = pairs[:, :, 0] * np.int64(nverts) + pairs[:, :, 1]
    uniq, inv = np.unique(key.ravel(), return_inverse=True)
    edge_ids = inv.reshape(key.shape)
    edge_verts = np.stack([uniq // nverts, uniq % nverts], axis=1)
    if degree == 3:
        # two dofs per edge, numbered from its lower global vertex to its higher one: nverts + 2 e + {0, 1}; the
        # cell lists, per local edge (a, b), the node nearer a first; then one interior dof per cell (triangles) or one
        # dof per face, faces numbered by ascending sorted vertex triple (tetrahedra)
        ne, nc = uniq.shape[0], cells.shape[0]
        ed = np.empty((nc, len(edges), 2), dtype=np.int64)
        for k, (a, b) in enumerate(edges):
            flip = (cells[:, a] > cells[:, b]).astype(np.int64)  # the node nearer a is the edge's SECOND dof
            ed[:, k, 0] = nverts + 2 * edge_ids[:, k] + flip
            ed[:, k, 1] = nverts + 2 * edge_ids[:, k] + 1 - flip
        if d == 2:
            interior = nverts + 2 * ne + np.arange(nc, dtype=np.int64)
            cell_dofs = np.concatenate([cells, ed.reshape(nc, 6), interior[:, None]], axis=1)
            return cell_dofs, nverts + 2 * ne + nc, edge_verts
        _, face_ids, nf = _face_ids(cells, nverts)
        cell_dofs = np.concatenate([cells, ed.reshape(nc, 12), nverts + 2 * ne + face_ids], axis=1)
        return cell_dofs, nverts + 2 * ne + nf, edge_verts
    cell_dofs = np.concatenate([cells, nverts + edge_ids], axis=1)
    return cell_dofs, nverts + uniq.shape[0], edge_verts


def _face_ids(cells, nverts):
    """Faces of a tetrahedral mesh numbered by ascending sorted vertex triple: (face vertices (nf, 3), ids (nc, 4), nf)."""
    tri = np.stack([np.sort(cells[:, list(f)], axis=1) for f in local_faces(3)], axis=1)  # (nc, 4, 3)
    nv64 = np.int64(nverts)
    if float(nverts) ** 3 >= 2.0 ** 62:
        raise ValueError("face keys overflow int64")
    key = (tri[:, :, 0] * nv64 + tri[:, :, 1]) * nv64 + tri[:, :, 2]
    uniq, inv = np.unique(key.ravel(), return_inverse=True)
    fv = np.stack([uniq // (nv64 * nv64), (uniq // nv64) % nv64, uniq % nv64], axis=1)
    return fv, inv.reshape(key.shape), int(uniq.shape[0])


def dof_coordinates(coords, degree, edge_verts, cells=None):
    if degree == 1:
        return coords.copy()
    if degree == 3:
        lo, hi = coords[edge_verts[:, 0]], coords[edge_verts[:, 1]]
        on_edges = np.stack([(1.0 - t) * lo + t * hi for t in GLL3], axis=1).reshape(-1, coords.shape[1])
        if coords.shape[1] == 3:
            fv, _, _ = _face_ids(cells, coords.shape[0])
            return np.concatenate([coords, on_edges, coords[fv].mean(axis=1)], axis=0)
        return np.concatenate([coords, on_edges, coords[cells].mean(axis=1)], axis=0)
    mid = 0.5 * (coords[edge_verts[:, 0]] + coords[edge_verts[:, 1]])
    return np.concatenate([coords, mid], axis=0)


# ----------------------------------------------------------------------------
# Element kernels (one numpy einsum per form family), CSR assembly
# ----------------------------------------------------------------------------


class Forms:
    """The form families of reference fracstep.py:277-358 on one mesh.

    ``vd``/``qd`` are the cell->dof tables of the scalar velocity-component
    space Vi and of the pressure space Q (any numbering; callers may pass the
    product's numbering so fields compare index by index)."""

    def __init__(self, coords, cells, u_deg, p_deg, vd=None, qd=None, nv_dofs=None, nq_dofs=None):
        self.coords = np.asarray(coords, dtype=np.float64)
        self.cells = np.asarray(cells, dtype=np.int64)
        self.d = self.coords.shape[1]
        self.u_deg, self.p_deg = u_deg, p_deg
        nverts = self.coords.shape[0]
        if vd is None:
            vd, nv_dofs, ev = build_dofmap(self.cells, nverts, u_deg)
            self.x_v = dof_coordinates(self.coords, u_deg, ev, self.cells)
        if qd is None:
            qd, nq_dofs, eq = build_dofmap(self.cells, nverts, p_deg)
            self.x_q = dof_coordinates(self.coords, p_deg, eq, self.cells)
        self.vd = np.asarray(vd, dtype=np.int64)
        self.qd = np.asarray(qd, dtype=np.int64)
        self.nv, self.nq = int(nv_dofs), int(nq_dofs)
        self.G, self.adet = cell_geometry(self.coords, self.cells)
        # rule exact to degree 2*4-1 = 7 >= 5 (P2 convection: 2+1+2); degree 9 >= 8 for P3 (3+2+3)
        self.bary, self.w = simplex_quadrature(self.d, 4 if max(u_deg, p_deg) <= 2 else 5)
        self.phi_v, self.dphi_v = tabulate(self.d, u_deg, self.bary)
        self.phi_q, self.dphi_q = tabulate(self.d, p_deg, self.bary)
        # physical gradients at quadrature points: (nc, Q, nd, d)
        self.grad_v = np.einsum("qib,cbk->cqik", self.dphi_v, self.G, optimize=True)
        self.grad_q = np.einsum("qib,cbk->cqik", self.dphi_q, self.G, optimize=True)

    # -- assembly helpers --------------------------------------------------
    def _csr(self, Ae, rd, cd, shape):
        nr, ncol = rd.shape[1], cd.shape[1]
        rows = np.repeat(rd, ncol, axis=1).ravel()
        cols = np.tile(cd, (1, nr)).ravel()
        A = sp.coo_matrix((Ae.ravel(), (rows, cols)), shape=shape).tocsr()
        A.sum_duplicates()
        A.sort_indices()
        return A

    def _vec(self, be, rd, n):
        out = np.zeros(n)
        np.add.at(out, rd.ravel(), be.ravel())
        return out

    # -- bilinear forms ----------------------------------------------------
    def mass_v(self):
        """u*v*dx on Vi (reference fracstep.py:292,373)."""
        Mref = np.einsum("q,qi,qj->ij", self.w, self.phi_v, self.phi_v)
        Ae = self.adet[:, None, None] * Mref[None]
        return self._csr(Ae, self.vd, self.vd, (self.nv, self.nv))

    def mass_q(self):
        Mref = np.einsum("q,qi,qj->ij", self.w, self.phi_q, self.phi_q)
        Ae = self.adet[:, None, None] * Mref[None]
        return self._csr(Ae, self.qd, self.qd, (self.nq, self.nq))

    def stiffness_v(self):
        """inner(grad u, grad v)*dx on Vi (reference fracstep.py:297-299,375)."""
        Ae = np.einsum("q,cqik,cqjk->cij", self.w, self.grad_v, self.grad_v, optimize=True)
        Ae *= self.adet[:, None, None]
        return self._csr(Ae, self.vd, self.vd, (self.nv, self.nv))

    def stiffness_q(self):
        """inner(grad p, grad q)*dx on Q (reference fracstep.py:321-324,379)."""
        Ae = np.einsum("q,cqik,cqjk->cij", self.w, self.grad_q, self.grad_q, optimize=True)
        Ae *= self.adet[:, None, None]
        return self._csr(Ae, self.qd, self.qd, (self.nq, self.nq))

    def convection(self, uab):
        """inner(dot(uab, nabla_grad(u)), v)*dx (reference fracstep.py:355-358):
        C_ij = int (uab . grad phi_j) phi_i.  ``uab`` is (nv, d)."""
        ucell = uab[self.vd]  # (nc, nd, d)
        uq = np.einsum("qk,ckd->cqd", self.phi_v, ucell, optimize=True)
        t = np.einsum("cqd,cqjd->cqj", uq, self.grad_v, optimize=True)
        Ae = np.einsum("q,qi,cqj->cij", self.w, self.phi_v, t, optimize=True)
        Ae *= self.adet[:, None, None]
        return self._csr(Ae, self.vd, self.vd, (self.nv, self.nv))

    def p_vdxi_mat(self, i):
        """p * v.dx(i) * dx, rows Vi, cols Q (reference fracstep.py:311-315)."""
        Ae = np.einsum("q,cqr,qs->crs", self.w, self.grad_v[:, :, :, i], self.phi_q, optimize=True)
        Ae *= self.adet[:, None, None]
        return self._csr(Ae, self.vd, self.qd, (self.nv, self.nq))

    def grad_p_mat(self, i):
        """p.dx(i) * v * dx, rows Vi, cols Q (reference fracstep.py:348-352)."""
        Ae = np.einsum("q,qr,cqs->crs", self.w, self.phi_v, self.grad_q[:, :, :, i], optimize=True)
        Ae *= self.adet[:, None, None]
        return self._csr(Ae, self.vd, self.qd, (self.nv, self.nq))

    def divu_mat(self, i):
        """u.dx(i) * q * dx, rows Q, cols Vi (reference fracstep.py:332-336)."""
        Ae = np.einsum("q,qr,cqs->crs", self.w, self.phi_q, self.grad_v[:, :, :, i], optimize=True)
        Ae *= self.adet[:, None, None]
        return self._csr(Ae, self.qd, self.vd, (self.nq, self.nv))

    # -- linear forms (the low-memory, matrix-free variants) -----------------
    def body_force_vec(self, f):
        """f * v * dx with constant f (reference fracstep.py:284-289,387-390)."""
        be = f * self.adet[:, None] * np.einsum("q,qi->i", self.w, self.phi_v)[None]
        return self._vec(be, self.vd, self.nv)

    def load_vec(self, f, space="v", nq=5):
        """f * v * dx for a spatial expression ``f(x)``, x: (3, npts) -> (npts,) -- what a UFL expression of the
        coordinates is as a body force (reference fracstep.py:284-289: ``force * v * dx`` for any ``force``) or as
        a projector's source (function.py:75) -- with a degree-(2 nq - 1) collapsed Gauss-Jacobi rule."""
        bary, w = simplex_quadrature(self.d, nq)
        deg = self.u_deg if space == "v" else self.p_deg
        dofs, n = (self.vd, self.nv) if space == "v" else (self.qd, self.nq)
        phi, _ = tabulate(self.d, deg, bary)
        xq = np.einsum("qa,cak->cqk", bary, self.coords[self.cells], optimize=True)
        X = np.zeros((3, xq.shape[0] * xq.shape[1]))
        X[: self.d] = xq.reshape(-1, self.d).T
        fq = np.asarray(f(X), dtype=np.float64).reshape(xq.shape[0], xq.shape[1])
        be = self.adet[:, None] * np.einsum("q,cq,qi->ci", w, fq, phi, optimize=True)
        return self._vec(be, dofs, n)

    def p_vdxi_vec(self, ps, i):
        """ps * v.dx(i) * dx (reference fracstep.py:306-309,487-497)."""
        pq = np.einsum("qs,cs->cq", self.phi_q, ps[self.qd], optimize=True)
        be = np.einsum("q,cq,cqr->cr", self.w, pq, self.grad_v[:, :, :, i], optimize=True)
        be *= self.adet[:, None]
        return self._vec(be, self.vd, self.nv)

    def divu_vec(self, u):
        """div(u) * q * dx, u is (nv, d) (reference fracstep.py:328-330,538)."""
        ucell = u[self.vd]
        div = np.einsum("cqkd,ckd->cq", self.grad_v, ucell, optimize=True)
        be = np.einsum("q,cq,qr->cr", self.w, div, self.phi_q, optimize=True)
        be *= self.adet[:, None]
        return self._vec(be, self.qd, self.nq)

    def grad_p_vec(self, dp, i):
        """dp.dx(i) * v * dx (reference fracstep.py:343-346,618)."""
        g = np.einsum("cqs,cs->cq", self.grad_q[:, :, :, i], dp[self.qd], optimize=True)
        be = np.einsum("q,cq,qr->cr", self.w, g, self.phi_v, optimize=True)
        be *= self.adet[:, None]
        return self._vec(be, self.vd, self.nv)

    def exterior_facets(self):
        """(cell, opposite local vertex) of every exterior facet."""
        d = self.d
        nvl = d + 1
        keys, owner = [], []
        for a in range(nvl):
            fv = np.sort(np.delete(self.cells, a, axis=1), axis=1)
            k = np.zeros(fv.shape[0], dtype=np.int64)
            for j in range(d):
                k = k * np.int64(self.coords.shape[0]) + fv[:, j]
            keys.append(k)
            owner.append(np.stack([np.arange(self.cells.shape[0]), np.full(self.cells.shape[0], a)], axis=1))
        keys = np.concatenate(keys)
        owner = np.concatenate(owner)
        uniq, idx, cnt = np.unique(keys, return_index=True, return_counts=True)
        ext = idx[cnt == 1]
        return owner[ext, 0], owner[ext, 1]

    def facet_midpoints(self, fc, fa):
        x = self.coords[self.cells[fc]]  # (nf, d+1, d)
        mask = np.ones((fc.shape[0], self.d + 1), dtype=bool)
        mask[np.arange(fc.shape[0]), fa] = False
        return x[mask].reshape(fc.shape[0], self.d, self.d).mean(axis=1)

    def pressure_surface_vec(self, fc, fa, h, i):
        """int_Gamma h n_i dv/dx_i ds over the facets (cell fc, opposite vertex fa); h is a vector
        of Q-dof values (reference bcs.py:226-242, fracstep.py:461-465)."""
        d = self.d
        bf, wf = simplex_quadrature(d - 1, 3) if d > 2 else (None, None)
        if d == 2:
            from numpy.polynomial.legendre import leggauss
            s, w = leggauss(3)
            bf = np.stack([(1 - s) / 2, (1 + s) / 2], axis=1)
            wf = w / 2
        out = np.zeros(self.nv)
        for f in range(fc.shape[0]):
            c, a = fc[f], fa[f]
            others = [b for b in range(d + 1) if b != a]
            bary = np.zeros((bf.shape[0], d + 1))
            bary[:, others] = bf
            phi_q, _ = tabulate(d, self.p_deg, bary)
            _, dphi = tabulate(d, self.u_deg, bary)
            G = self.G[c]
            gnorm = np.linalg.norm(G[a])
            n = -G[a] / gnorm
            area = self.adet[c] * gnorm  # x (d-1)! folded into the rule's weights below
            hq = phi_q @ h[self.qd[c]]
            gi = np.einsum("qrb,b->qr", dphi, G[:, i])
            fac = math.factorial(d - 1)
            out_c = np.einsum("q,q,qr->r", wf * fac, hq, gi) * n[i] * area / fac
            np.add.at(out, self.vd[c], out_c)
        return out

    def volume(self):
        """assemble_scalar(1*dx) (reference fracstep.py:581-584)."""
        return float(self.adet.sum() / math.factorial(self.d))

    def integral_q(self, f):
        """assemble_scalar(f*dx) for f in Q (reference fracstep.py:585-590)."""
        fq = np.einsum("qs,cs->cq", self.phi_q, f[self.qd], optimize=True)
        return float(np.einsum("q,cq,c->", self.w, fq, self.adet, optimize=True))

    def l2_error_sq(self, uh, exact, space="v", nq=6):
        """int (uh - exact)^2 dx with a degree-(2nq-1) rule; ``exact`` maps
        x:(3, npts) -> (npts,) (demo/taylor_green.py:193-207)."""
        bary, w = simplex_quadrature(self.d, nq)
        deg = self.u_deg if space == "v" else self.p_deg
        dofs = self.vd if space == "v" else self.qd
        phi, _ = tabulate(self.d, deg, bary)
        xq = np.einsum("qa,cak->cqk", bary, self.coords[self.cells], optimize=True)
        X = np.zeros((3, xq.shape[0] * xq.shape[1]))
        X[: self.d] = xq.reshape(-1, self.d).T
        ex = np.asarray(exact(X)).reshape(xq.shape[0], xq.shape[1])
        uq = np.einsum("qi,ci->cq", phi, uh[dofs], optimize=True)
        return float(np.einsum("q,cq,c->", w, (uq - ex) ** 2, self.adet, optimize=True))


# ----------------------------------------------------------------------------
# Krylov solvers with PETSc's conventions (what KSPSolver.solve delegates to,
# reference ksp.py:71-78): zero initial guess unless told otherwise, left
# preconditioning, convergence on the preconditioned residual norm
#     ||B r|| <= max(rtol * ||B b||, atol),
# KSPConvergedReason integers (2 rtol, 3 atol, -3 max-its, -5 breakdown, -9 nan).
# ----------------------------------------------------------------------------

CONVERGED_RTOL, CONVERGED_ATOL, CONVERGED_ITS = 2, 3, 4
DIVERGED_ITS, DIVERGED_DTOL, DIVERGED_BREAKDOWN, DIVERGED_NANORINF = -3, -4, -5, -9


def _reason(rn, bn, rtol, atol, divtol=1e4):
    """KSPConvergedDefault (PETSc src/ksp/ksp/interface/iterativ.c; what ``KSP.solve`` of reference ksp.py:76 tests
    with the options of ksp.py:38-53): NaN -> -9; |r| <= max(rtol |b|, atol) -> 2 / 3; |r| >= divtol |b| -> -4
    (``-ksp_divtol``, default 1e4).  The divergence test needs |b| > 0 (PETSc's rnorm0 = 0 with a nonzero guess
    flags ANY residual: a quirk neither side reproduces)."""
    if not np.isfinite(rn):
        return DIVERGED_NANORINF
    if rn <= atol:
        return CONVERGED_ATOL
    if rn <= rtol * bn:
        return CONVERGED_RTOL
    if bn > 0.0 and rn >= divtol * bn:
        return DIVERGED_DTOL
    return 0


def jacobi_cg(A, b, x0=None, rtol=1e-5, atol=1e-50, max_it=10000, dinv=None, divtol=1e4):
    """Jacobi-preconditioned conjugate gradients (PETSc KSPCG + PCJACOBI)."""
    n = b.shape[0]
    dinv = 1.0 / A.diagonal() if dinv is None else dinv
    x = np.zeros(n) if x0 is None else x0.copy()
    r = b - A @ x if x0 is not None else b.copy()
    z = dinv * r
    bn = np.linalg.norm(dinv * b)
    rn = np.linalg.norm(z)
    reason = _reason(rn, bn, rtol, atol, divtol)
    it = 0
    if reason:
        return x, reason, it, rn
    p = z.copy()
    rz = r @ z
    while it < max_it:
        q = A @ p
        pq = p @ q
        if pq == 0.0:
            return x, DIVERGED_BREAKDOWN, it, rn
        alpha = rz / pq
        x += alpha * p
        r -= alpha * q
        z = dinv * r
        rn = np.linalg.norm(z)
        it += 1
        reason = _reason(rn, bn, rtol, atol, divtol)
        if reason:
            return x, reason, it, rn
        rz_new = r @ z
        beta = rz_new / rz
        rz = rz_new
        p = z + beta * p
    return x, DIVERGED_ITS, it, rn


def jacobi_bicgstab(A, b, x0=None, rtol=1e-5, atol=1e-50, max_it=10000, dinv=None, divtol=1e4):
    """Left-Jacobi-preconditioned BiCGStab (PETSc KSPBCGS + PCJACOBI): the
    recurrence runs on the preconditioned system B A x = B b."""
    n = b.shape[0]
    dinv = 1.0 / A.diagonal() if dinv is None else dinv
    x = np.zeros(n) if x0 is None else x0.copy()
    r = dinv * (b - A @ x) if x0 is not None else dinv * b
    bn = np.linalg.norm(dinv * b)
    rn = np.linalg.norm(r)
    it = 0
    reason = _reason(rn, bn, rtol, atol, divtol)
    if reason:
        return x, reason, it, rn
    rhat = r.copy()
    rho = alpha = omega = 1.0
    v = np.zeros(n)
    p = np.zeros(n)
    while it < max_it:
        rho_new = rhat @ r
        if rho_new == 0.0:
            return x, DIVERGED_BREAKDOWN, it, rn
        beta = (rho_new / rho) * (alpha / omega)
        rho = rho_new
        p = r + beta * (p - omega * v)
        v = dinv * (A @ p)
        rv = rhat @ v
        if rv == 0.0:
            return x, DIVERGED_BREAKDOWN, it, rn
        alpha = rho / rv
        s = r - alpha * v
        t = dinv * (A @ s)
        tt = t @ t
        omega = (t @ s) / tt if tt != 0.0 else 0.0
        x += alpha * p + omega * s
        r = s - omega * t
        rn = np.linalg.norm(r)
        it += 1
        reason = _reason(rn, bn, rtol, atol, divtol)
        if reason:
            return x, reason, it, rn
        if omega == 0.0:
            return x, DIVERGED_BREAKDOWN, it, rn
    return x, DIVERGED_ITS, it, rn


class OracleKSP:
    """CPU stand-in for what reference ksp.py:14-91 wraps: a PETSc KSP
    configured from a string-keyed options dict."""

    def __init__(self, options=None):
        o = dict(options or {})
        self.ksp_type = str(o.get("ksp_type", "preonly"))
        self.pc_type = str(o.get("pc_type", "lu"))
        self.rtol = float(o.get("ksp_rtol", 1e-5))
        self.atol = float(o.get("ksp_atol", 1e-50))
        self.max_it = int(o.get("ksp_max_it", 10000))
        self.divtol = float(o.get("ksp_divtol", 1e4))
        self.error_if_not_converged = o.get("ksp_error_if_not_converged", False) not in (False, 0, "0", "false")
        self.nonzero_guess = bool(o.get("ksp_initial_guess_nonzero", False))
        self.A = None
        self._lu = None
        self.its = 0

    def set_operator(self, A):
        self.A = A.tocsr()
        self._lu = None

    def solve(self, b, x):
        """x <- A^-1 b in place; returns the converged reason."""
        if self.ksp_type == "preonly":
            if self._lu is None:
                self._lu = spla.splu(self.A.tocsc())
            x[:] = self._lu.solve(b)
            self.its = 1
            return CONVERGED_ITS
        x0 = x.copy() if self.nonzero_guess else None
        dinv = 1.0 / self.A.diagonal() if self.pc_type == "jacobi" else np.ones(b.shape[0])
        fn = jacobi_cg if self.ksp_type == "cg" else jacobi_bicgstab
        sol, reason, its, _ = fn(self.A, b, x0, self.rtol, self.atol, self.max_it, dinv, self.divtol)
        x[:] = sol
        self.its = its
        if self.error_if_not_converged and reason <= 0:  # PETSc: error 91 out of KSPSolve
            raise RuntimeError(f"KSPSolve has not converged (reason {reason} after {its} iterations)")
        return reason


# ----------------------------------------------------------------------------
# The fractional-step solver
# ----------------------------------------------------------------------------


class DirichletData:
    """(dofs, value) pair; ``value`` is a float or a callable x:(3,npts)->(npts,)
    (reference bcs.py:103-139: create_bc / update_bc / apply = set_bc)."""

    def __init__(self, dofs, value):
        self.dofs = np.asarray(dofs, dtype=np.int64)
        self.value = value
        self.g = None

    def update(self, xdofs):
        if callable(self.value):
            X = np.zeros((3, xdofs.shape[0]))
            X[: xdofs.shape[1]] = xdofs.T
            self.g = np.asarray(self.value(X), dtype=np.float64)
        else:
            self.g = np.full(xdofs.shape[0], float(self.value))

    def apply(self, b):
        b[self.dofs] = self.g[self.dofs]


class PressureData:
    """Natural pressure condition on a set of exterior facets (reference bcs.py:142-268):
    value h (float or callable on x) enters b_first_i as int h n_i dv/dx_i ds, and the pressure
    correction gets a homogeneous Dirichlet condition on the facets' Q dofs."""

    def __init__(self, facet_cells, facet_opp, value):
        self.fc, self.fa, self.value = np.asarray(facet_cells), np.asarray(facet_opp), value
        self.dofs = None
        self.h = None

    def create(self, F, x_q):
        d = F.d
        dofs = []
        for c, a in zip(self.fc, self.fa):
            loc = [b for b in range(d + 1) if b != a]
            if F.p_deg == 2:  # the facet's edge dofs too: local edges whose two vertices lie on the facet
                loc += [d + 1 + e for e, (va, vb) in enumerate(local_edges(d)) if va != a and vb != a]
            dofs.extend(F.qd[c][loc])
        self.dofs = np.unique(np.asarray(dofs, dtype=np.int64))
        self.update(x_q)

    def update(self, x_q):
        if callable(self.value):
            X = np.zeros((3, x_q.shape[0]))
            X[: x_q.shape[1]] = x_q.T
            self.h = np.asarray(self.value(X), dtype=np.float64)
        else:
            self.h = np.full(x_q.shape[0], float(self.value))


class OracleFractionalStep:
    """Restatement of reference fracstep.py:149-705 (``FractionalStep_AB_CN``)
    for Dirichlet-only velocity BCs and no pressure BC (all BASELINE configs).

    Velocity fields are stored as (nv, d) arrays; column i is the reference's
    ``_u[i].x.array``."""

    def __init__(self, forms: Forms, x_v, x_q, bcs_u, solver_options=None, body_force=None,
                 low_memory=True, bcs_p=None, rotational=False):
        self.F = forms
        d = forms.d
        self.d = d
        self.x_v, self.x_q = x_v, x_q
        nv, nq = forms.nv, forms.nq
        self.u = np.zeros((nv, d))
        self.u1 = np.zeros((nv, d))
        self.u2 = np.zeros((nv, d))
        self.uab = np.zeros((nv, d))
        self.rhs1 = np.zeros((nv, d))
        self.b_first = np.zeros((nv, d))
        self.p = np.zeros(nq)
        self.ps = np.zeros(nq)
        self.dp = np.zeros(nq)
        self.b2 = np.zeros(nq)
        self.bcs_u = bcs_u  # list (per component) of lists of DirichletData
        for bcl in bcs_u:
            for bc in bcl:
                bc.update(x_v)  # create_bc interpolates once (bcs.py:121-126)
        so = solver_options or {}
        self.solver_u = OracleKSP(so.get("tentative"))
        self.solver_p = OracleKSP(so.get("pressure"))
        self.solver_c = OracleKSP(so.get("scalar"))
        self.rotational = rotational
        if rotational:  # Projector(p + dp - xi nu div u, Q) with the "scalar" options (fracstep.py:237-247)
            self.Mq = forms.mass_q()
            self.solver_proj = OracleKSP(so.get("scalar"))
            self.solver_proj.set_operator(self.Mq)
            self.xi = 0.5
        self.low_memory = low_memory
        f = (0.0,) * d if body_force is None else body_force
        # _preassemble (fracstep.py:360-409)
        self.M = forms.mass_v()
        self.K = forms.stiffness_v()
        self.Ap = forms.stiffness_q()
        self.bcs_p = list(bcs_p or [])
        for bp in self.bcs_p:
            bp.create(forms, x_q)
        if self.bcs_p:  # assemble_matrix(Ap, bcs): rows and columns -> identity (fracstep.py:379)
            pd = np.unique(np.concatenate([bp.dofs for bp in self.bcs_p]))
            keep = np.ones(nq)
            keep[pd] = 0.0
            D = sp.diags(keep)
            self.Ap = (D @ self.Ap @ D + sp.diags(1.0 - keep)).tocsr()
            self.p_bc_dofs = pd
        # a constant or a spatial expression per component (fracstep.py:284-289,387-390)
        self.b0 = np.stack([forms.load_vec(f[i]) if callable(f[i]) else forms.body_force_vec(float(f[i]))
                            for i in range(d)], axis=1)
        if not low_memory:
            self.P = [forms.p_vdxi_mat(i) for i in range(d)]
            self.Gm = [forms.grad_p_mat(i) for i in range(d)]
            self.D = [forms.divu_mat(i) for i in range(d)]
        self.vol = forms.volume()
        self.wq = np.asarray(forms.mass_q().sum(axis=0)).ravel()  # int psi_i dx
        self.solver_p.set_operator(self.Ap)
        self.solver_c.set_operator(self.M)
        self.A = None
        self.its = {}

    # fracstep.py:411-472
    def assemble_first(self, dt, nu):
        self.uab[:] = 1.5 * self.u1 - 0.5 * self.u2
        C = self.F.convection(self.uab)
        A = -0.5 * C + (1.0 / dt) * self.M + (-0.5 * nu) * self.K
        for bp in self.bcs_p:  # fracstep.py:445-446
            bp.update(self.x_q)
        for i in range(self.d):
            self.b_first[:, i] = A @ self.u1[:, i] + self.b0[:, i]
            for bp in self.bcs_p:  # fracstep.py:461-465
                self.b_first[:, i] += self.F.pressure_surface_vec(bp.fc, bp.fa, bp.h, i)
        A = -A + (2.0 / dt) * self.M
        A = A.tolil()
        for bc in self.bcs_u[0]:  # bcs_u[0] ONLY (fracstep.py:470-472)
            for r in np.unique(bc.dofs):
                A.rows[r] = list(A.rows[r])
                A.data[r] = [1.0 if c == r else 0.0 for c in A.rows[r]]
        self.A = A.tocsr()
        self.solver_u.set_operator(self.A)

    # fracstep.py:474-506
    def velocity_tentative_assemble(self):
        for i in range(self.d):
            if self.low_memory:
                pv = self.F.p_vdxi_vec(self.ps, i)
            else:
                pv = self.P[i] @ self.ps
            self.rhs1[:, i] = self.b_first[:, i] + pv

    # fracstep.py:508-525
    def velocity_tentative_solve(self):
        diff = 0.0
        errors = np.zeros(self.d, dtype=np.int32)
        its = []
        for i in range(self.d):
            for bc in self.bcs_u[i]:
                bc.apply(self.rhs1[:, i])
            old = self.u[:, i].copy()
            x = self.u[:, i].copy()
            errors[i] = self.solver_u.solve(self.rhs1[:, i], x)
            self.u[:, i] = x
            its.append(self.solver_u.its)
            diff += np.linalg.norm(old - x)
        self.its["tentative"] = its
        return diff, errors

    # fracstep.py:527-551
    def pressure_assemble(self, dt):
        if self.low_memory:
            self.b2[:] = self.F.divu_vec(self.u)
        else:
            self.b2[:] = sum(self.D[i] @ self.u[:, i] for i in range(self.d))
        self.b2 *= -1.0 / dt
        if self.bcs_p:  # set_bc(b2, bcs_p): homogeneous (fracstep.py:549-550)
            self.b2[self.p_bc_dofs] = 0.0

    # fracstep.py:553-605
    def _update_ps(self, nu):
        if self.rotational:  # fracstep.py:593-602
            if nu is None:
                raise RuntimeWarning("Kinematic viscosity not set for rotational pressure correction")
            rhs = self.Mq @ (self.p + self.dp) - self.xi * nu * self.F.divu_vec(self.u)
            x = np.zeros_like(self.ps)
            assert self.solver_proj.solve(rhs, x) > 0
            self.ps[:] = x
        else:
            self.ps[:] = self.p + self.dp  # fracstep.py:604

    def pressure_solve(self, nu=None):
        if self.bcs_p:  # non-singular: plain solve, no mean handling
            x = self.dp.copy()
            reason = self.solver_p.solve(self.b2, x)
            self.dp[:] = x
            self.its["pressure"] = self.solver_p.its
            self._update_ps(nu)
            return reason
        self.b2 -= self.b2.mean()  # nullspace.remove (fracstep.py:573-574)
        if self.solver_p.ksp_type == "preonly":
            # MUMPS with null-pivot detection (fracstep.py:564-571) returns *a*
            # solution of the singular system; the mass-weighted shift below makes
            # the result unique, so solve the bordered system for that one directly.
            n = self.F.nq
            w = self.wq
            Kb = sp.bmat([[self.Ap, sp.csr_matrix(w[:, None])],
                          [sp.csr_matrix(w[None, :]), None]], format="csc")
            sol = spla.splu(Kb).solve(np.concatenate([self.b2, [0.0]]))
            self.dp[:] = sol[:n]
            reason = CONVERGED_ITS
            self.its["pressure"] = 1
        else:
            x = self.dp.copy()
            reason = self.solver_p.solve(self.b2, x)
            self.dp[:] = x
            self.its["pressure"] = self.solver_p.its
        phi_avg = self.F.integral_q(self.dp) / self.vol  # fracstep.py:579-591
        self.dp -= phi_avg
        self._update_ps(nu)
        return reason

    # fracstep.py:607-658 (un-BC'd M, no BC re-imposition)
    def velocity_update(self, dt):
        errors = np.zeros(self.d, dtype=np.int32)
        its = []
        for i in range(self.d):
            b3 = self.M @ self.u[:, i]
            if self.low_memory:
                g = self.F.grad_p_vec(self.dp, i)
            else:
                g = self.Gm[i] @ self.dp
            b3 -= dt * g
            x = self.u[:, i].copy()
            errors[i] = self.solver_c.solve(b3, x)
            self.u[:, i] = x
            its.append(self.solver_c.its)
        self.its["update"] = its
        return errors

    # fracstep.py:660-696
    def solve(self, dt, nu, max_error=1e-12, max_iter=10):
        inner_it = 0
        diff = 1e8
        self.ps[:] = self.p
        for bcl in self.bcs_u:
            for bc in bcl:
                bc.update(self.x_v)
        self.assemble_first(dt, nu)
        while inner_it < max_iter and diff > max_error:
            inner_it += 1
            self.velocity_tentative_assemble()
            diff, errors = self.velocity_tentative_solve()
            assert (errors > 0).all()
            self.pressure_assemble(dt)
            error_p = self.pressure_solve(nu)
            assert int(error_p) > 0
        self.velocity_update(dt)
        self.u2[:] = self.u1
        self.u1[:] = self.u
        self.p[:] = self.ps
        return diff


# ----------------------------------------------------------------------------
# Analytic Taylor-Green fields (demo/taylor_green.py:36-53,176-191); the 3-D
# benchmark uses the z-extruded field (w = 0) on [-1,1]^3.
# ----------------------------------------------------------------------------


def tg_u(x, t, nu):
    return -np.cos(np.pi * x[0]) * np.sin(np.pi * x[1]) * np.exp(-2.0 * nu * np.pi ** 2 * t)


def tg_v(x, t, nu):
    return np.cos(np.pi * x[1]) * np.sin(np.pi * x[0]) * np.exp(-2.0 * nu * np.pi ** 2 * t)


def tg_w(x, t, nu):
    return np.zeros_like(x[0])


def tg_p(x, t, nu):
    return -0.25 * (np.cos(2 * np.pi * x[0]) + np.cos(2 * np.pi * x[1])) * np.exp(
        -4.0 * nu * np.pi ** 2 * t)


def boundary_dofs(xdofs, p0, p1, tol=1e-10):
    """Dofs on the surface of the axis-aligned box [p0, p1]."""
    on = np.zeros(xdofs.shape[0], dtype=bool)
    for k in range(xdofs.shape[1]):
        on |= np.abs(xdofs[:, k] - p0[k]) < tol
        on |= np.abs(xdofs[:, k] - p1[k]) < tol
    return np.nonzero(on)[0]


def taylor_green_problem(N, dim=2, u_deg=2, p_deg=1, nu=0.01, dt=0.005, t0=0.0,
                         solver_options=None, low_memory=True, mesh=None, vd=None, qd=None,
                         x_v=None, x_q=None, rotational=False, body_force=None):
    """Set up the demo's problem (demo/taylor_green.py:104-182): exact Dirichlet
    velocity on every exterior facet, no pressure BC, u2(t0-dt), u1(t0), p(t0-dt/2).
    Returns (solver, clock) where clock['t'] is the time the BC callables read."""
    if mesh is None:
        if dim == 2:
            coords, cells = create_rectangle_mesh([-1, -1], [1, 1], [N, N])
        else:
            coords, cells = create_box_mesh([-1, -1, -1], [1, 1, 1], [N, N, N])
    else:
        coords, cells = mesh
    if vd is None:
        F = Forms(coords, cells, u_deg, p_deg)
        x_v, x_q = F.x_v, F.x_q
    else:
        F = Forms(coords, cells, u_deg, p_deg, vd=vd, qd=qd, nv_dofs=x_v.shape[0],
                  nq_dofs=x_q.shape[0])
    d = coords.shape[1]
    clock = {"t": t0}
    fns = [tg_u, tg_v, tg_w][:d]
    lo, hi = coords.min(axis=0), coords.max(axis=0)
    bd = boundary_dofs(x_v, lo, hi)
    bcs_u = [[DirichletData(bd, (lambda x, f=f: f(x, clock["t"], nu)))] for f in fns]
    S = OracleFractionalStep(F, x_v, x_q, bcs_u, solver_options=solver_options,
                             low_memory=low_memory, rotational=rotational, body_force=body_force)
    X = np.zeros((3, x_v.shape[0]))
    X[:d] = x_v.T
    Xq = np.zeros((3, x_q.shape[0]))
    Xq[:d] = x_q.T
    for i, f in enumerate(fns):
        S.u2[:, i] = f(X, t0 - dt, nu)
        S.u1[:, i] = f(X, t0, nu)
    S.p[:] = tg_p(Xq, t0 - dt / 2.0, nu)
    return S, clock
