"""Boundary conditions with the reference's surface (reference src/oasisx/bcs.py:1-268).

``DirichletBC`` keeps the dof list and the boundary values on the device; ``apply`` is the
``set_bc`` kernel (b[dofs] = g, reference bcs.py:135-139).  A callable value is evaluated at
the coordinates of the constrained dofs only -- the reference interpolates the whole field
(bcs.py:121-133) but ``set_bc`` reads nothing except those entries.
"""
from __future__ import annotations

from enum import Enum
import numpy as np
import torch

from . import _lib
from .fem import Constant, FunctionSpace, Vector, locate_dofs_geometrical, locate_dofs_topological

__all__ = ["DirichletBC", "PressureBC", "LocatorMethod"]


class LocatorMethod(Enum):
    """Search methods for Dirichlet BCs (reference bcs.py:22-33)."""

    GEOMETRICAL = 1
    TOPOLOGICAL = 2


LocatorMethod.TOPOLOGICAL.__doc__ = "Topogical search for dofs"
LocatorMethod.GEOMETRICAL.__doc__ = "Geometrical search for dofs"


class _CppBC:
    def __init__(self, dofs):
        self._dofs = dofs

    def dof_indices(self):
        return self._dofs, int(self._dofs.shape[0])


class _BCHandle:
    """What ``bc._bc`` exposes to callers (``_cpp_object.dof_indices()``)."""

    def __init__(self, dofs):
        self._cpp_object = _CppBC(dofs)


class DirichletBC:
    """Dirichlet condition on a velocity-component space (reference bcs.py:36-139).

    Args:
        value: a float, a ``Constant`` (tracked by reference) or a callable
            ``x:(3, npts) -> (npts,)``.
        method: ``LocatorMethod.GEOMETRICAL`` (marker = callable on dof coordinates) or
            ``LocatorMethod.TOPOLOGICAL`` (marker = ``(meshtags, value)``).
    """

    def __init__(self, value, method: LocatorMethod, marker):
        if method == LocatorMethod.GEOMETRICAL:
            self._method = method
            self._locator = marker
        elif method == LocatorMethod.TOPOLOGICAL:
            self._method = method
            self._entities = marker[0].find(marker[1])
            self._e_dim = marker[0].dim
        else:
            raise ValueError(f"unknown locator method {method!r}")
        self._value = value
        self._V: FunctionSpace | None = None

    def set_dofs(self, dofs):
        self._dofs = np.asarray(dofs, dtype=np.int32)

    def _locate_dofs(self, V: FunctionSpace):
        if self._method == LocatorMethod.GEOMETRICAL:
            self._dofs = locate_dofs_geometrical(V, self._locator)
        else:
            V.mesh.topology.create_connectivity(self._e_dim, V.mesh.topology.dim)
            self._dofs = locate_dofs_topological(V, self._e_dim, self._entities)

    def create_bc(self, V: FunctionSpace):
        if not hasattr(self, "_dofs"):
            self._locate_dofs(V)
        self._V = V
        dev = V.mesh.device
        self._dofs = np.asarray(self._dofs, dtype=np.int32)
        self._dofs_dev = torch.from_numpy(self._dofs).to(dev)
        # matrix rows exist for owned dofs only (mesh-partitioned runs keep ghosts in the vectors)
        self._rows_dev = torch.from_numpy(self._dofs[self._dofs < V.n_owned]).to(dev)
        self._xbc = np.ascontiguousarray(V.tabulate_dof_coordinates()[self._dofs].T)  # (3, nbc)
        self._g_dev = torch.zeros(self._dofs.shape[0], dtype=torch.float64, device=dev)
        self._g_stamp = None
        self._bc = _BCHandle(self._dofs)
        if callable(self._value):
            self._u = True  # marks "value is a function": update_bc re-evaluates it
            self._evaluate()
        else:
            self._refresh_constant()

    def _evaluate(self):
        # A callable marked ``supports_torch = True`` is handed the boundary coordinates as a
        # (3, nbc) torch tensor ON THE DEVICE and must return a device tensor: the Dirichlet values
        # then never touch the host (the reference's numpy callables keep working unmarked).
        if getattr(self._value, "supports_torch", False):
            if not hasattr(self, "_xbc_dev"):
                self._xbc_dev = torch.from_numpy(self._xbc).to(self._g_dev.device)
            g = self._value(self._xbc_dev)
            if not torch.is_tensor(g):
                raise TypeError("a supports_torch callable must return a torch tensor")
            self._g_dev.copy_(g.reshape(-1).to(torch.float64))
            return
        g = np.asarray(self._value(self._xbc), dtype=np.float64).reshape(-1)
        if g.shape[0] != self._dofs.shape[0]:
            g = np.broadcast_to(g, self._dofs.shape).copy()
        self._g_dev.copy_(torch.from_numpy(np.ascontiguousarray(g)))

    def _refresh_constant(self):
        v = float(self._value.value) if isinstance(self._value, Constant) else float(self._value)
        if v != self._g_stamp:
            self._g_dev.fill_(v)
            self._g_stamp = v

    def update_bc(self):
        """Re-evaluate a callable value (reference bcs.py:128-133)."""
        if hasattr(self, "_u"):
            self._evaluate()

    def values_host(self) -> np.ndarray:
        if not hasattr(self, "_u"):
            self._refresh_constant()
        return self._g_dev.cpu().numpy()

    def apply(self, x: Vector):
        """b[dofs] = g (reference bcs.py:135-139)."""
        if not hasattr(self, "_u"):
            self._refresh_constant()
        s = x._s
        comp = 0 if x._c is None else x._c
        lib = _lib.load()
        _lib.check(lib.ox_set_bc(s.ptr(), _lib.ptr(self._dofs_dev), _lib.ptr(self._g_dev),
                                 int(self._dofs.shape[0]), s.nc, comp, _lib.current_stream()),
                   "ox_set_bc")


class _SurfaceForm:
    """What ``PressureBC.rhs(i)`` returns: the assembled-on-demand outlet term of component i."""

    def __init__(self, bc, i):
        self.bc, self.i, self.rank = bc, i, 1


class PressureBC:
    """Natural pressure condition on a set of tagged exterior facets (reference bcs.py:142-268).

    The value h (float, ``Constant`` or callable interpolated into Q) enters the tentative
    velocity RHS as ``int_Gamma h n_i dv/dx_i ds`` (bcs.py:226-242, fracstep.py:461-465) and the
    pressure correction gets a homogeneous Dirichlet condition on the facets' pressure dofs
    (bcs.py:245-253).

    Device form: the term is linear in the nodal values of h, so ``create_bcs`` builds, once,
    one small SELL-64 operator per component on the rows (velocity dofs) of the cells behind the
    facets; per step the term is d tiny SpMVs plus a scatter-add -- no atomics."""

    def __init__(self, value, marker):
        self._subdomain_data, self._subdomain_id = marker
        self._value = value

    def create_bcs(self, V, Q):
        import itertools

        from .fem import build_sell
        from .la import SellMatrix

        mesh = V.mesh
        assert mesh.topology is self._subdomain_data.topology
        dev = mesh.device
        d = mesh.gdim
        tags = self._subdomain_data
        if isinstance(self._subdomain_id, tuple):
            facets = tags.indices[np.isin(tags.values, np.asarray(self._subdomain_id, dtype=np.int32))]
        else:
            facets = tags.find(np.int32(self._subdomain_id))
        self._facets = np.asarray(facets, dtype=np.int64)
        # facet -> (cell, opposite local vertex)
        _, cf = mesh._entities(d - 1)
        combos = list(itertools.combinations(range(d + 1), d))
        opp_of_combo = np.array([[a for a in range(d + 1) if a not in c][0] for c in combos])
        mask = np.isin(cf, self._facets)
        fcell, fslot = np.nonzero(mask)
        fopp = opp_of_combo[fslot]
        # local cell index of each facet's cell (mesh-partitioned runs keep a subset of the cells)
        lc = V.local_cells.cpu().numpy()  # kernel-side cell order (and, partitioned, a subset)
        srt = np.argsort(lc)
        pos = np.searchsorted(lc[srt], fcell)
        ok = (pos < lc.shape[0]) & (lc[srt][np.minimum(pos, lc.shape[0] - 1)] == fcell)
        fcell_g, fopp, floc = fcell[ok], fopp[ok], srt[pos[ok]]
        nf = fcell_g.shape[0]
        self._V, self._Q = V, Q
        self._n_facets = nf
        # geometry of those cells: all d+1 barycentric gradients, |detJ|
        xc = mesh.coords[mesh.cells[torch.from_numpy(fcell_g).to(dev)]]  # (nf, d+1, d)
        J = (xc[:, 1:, :] - xc[:, :1, :]).transpose(1, 2)
        adet = torch.linalg.det(J).abs()
        Ginv = torch.linalg.inv(J)
        G = torch.cat([-Ginv.sum(dim=1, keepdim=True), Ginv], dim=1)  # (nf, d+1, d)
        # facet quadrature (degree >= 3) in the barycentric coordinates of the cell
        from .fem import _simplex_rule

        if d == 2:
            from numpy.polynomial.legendre import leggauss

            sq, wq = leggauss(3)
            bf, wf = np.stack([(1 - sq) / 2, (1 + sq) / 2], axis=1), wq / 2
        else:
            bf, wf = _simplex_rule(2, 3)
        # weights sum to 1/(d-1)!;  int_F g n_i ds = -|detJ| (grad lambda_opp)_i sum_q w_q g(x_q)
        nq_ = bf.shape[0]
        nvl = d + 1
        # pressure basis and velocity-basis derivatives at the facet points (any built pair: P1-P1, P2-P1, P3-P2);
        # the points of a facet depend on its opposite vertex only: one tabulation per local facet
        from .fem import lagrange_basis, lagrange_basis_derivs

        psi = torch.zeros((nf, nq_, Q.nd), dtype=torch.float64, device=dev)
        dphi = torch.zeros((nf, nq_, V.nd, nvl), dtype=torch.float64, device=dev)
        for f in range(nvl):
            sel = torch.from_numpy(fopp == f).to(dev)
            if not bool(sel.any()):
                continue
            others = [b_ for b_ in range(nvl) if b_ != f]
            tmp = np.zeros((nq_, nvl))
            tmp[:, others] = bf
            psi[sel] = torch.from_numpy(lagrange_basis(d, Q.degree, tmp)).to(dev)
            dphi[sel] = torch.from_numpy(lagrange_basis_derivs(d, V.degree, tmp)).to(dev)
        grad = torch.einsum("fqrb,fbk->fqrk", dphi, G)  # (nf, q, nd_v, d)
        ia = torch.from_numpy(fopp).to(dev)
        Ga = G[torch.arange(nf, device=dev), ia]  # (nf, d) = grad lambda_opp;  n |F| = -|detJ| Ga
        w = torch.from_numpy(wf).to(dev)
        # S_i[r, c] = sum_q w_q psi_c dphi_r/dx_i * (-|detJ| Ga_i)
        vals = torch.einsum("q,fqc,fqrk->fkrc", w, psi, grad) * (-(adet.unsqueeze(1) * Ga)).unsqueeze(2).unsqueeze(3)
        tl = torch.from_numpy(floc).to(dev)
        rows = V.cell_dofs[tl].to(torch.int64)  # (nf, nd_v)
        cols = Q.cell_dofs[tl].to(torch.int64)  # (nf, nd_q)
        R = rows.unsqueeze(2).expand(-1, -1, cols.shape[1]).reshape(-1)
        Cc = cols.unsqueeze(1).expand(-1, rows.shape[1], -1).reshape(-1)
        keep = R < V.n_owned
        R, Cc = R[keep], Cc[keep]
        nq_loc = Q.n_local
        key = R * nq_loc + Cc
        ukey, inv = torch.unique(key, return_inverse=True)
        urow = torch.div(ukey, nq_loc, rounding_mode="floor")
        ucol = (ukey - urow * nq_loc).to(torch.int32)
        self._rows_b, rinv = torch.unique(urow, return_inverse=True)  # touched velocity rows
        nrb = int(self._rows_b.shape[0])
        row_len = torch.bincount(rinv, minlength=nrb)
        row_ptr = torch.zeros(nrb + 1, dtype=torch.int64, device=dev)
        row_ptr[1:] = torch.cumsum(row_len, 0)
        self._rows_b32 = self._rows_b.to(torch.int32)
        self._S = []
        if nrb > 0:
            pat = build_sell(nrb, nq_loc, rinv * nq_loc + ucol.to(torch.int64), row_len, row_ptr)
            k = torch.arange(ukey.shape[0], device=dev) - row_ptr[rinv]
            off = pat.slice_ptr[rinv // 64] + (k // 2) * 128 + (rinv % 64) * 2 + (k % 2)
            for i in range(d):
                Sm = SellMatrix(pat, name=f"S{i}")
                csr_vals = torch.zeros(ukey.shape[0], dtype=torch.float64, device=dev)
                csr_vals.index_add_(0, inv, vals[:, i].reshape(-1)[keep])
                Sm.vals[off] = csr_vals
                self._S.append(Sm)
        self._y = torch.zeros(max(nrb, 1), dtype=torch.float64, device=dev)
        # homogeneous Dirichlet condition of the pressure correction (bcs.py:245-253)
        mesh.topology.create_connectivity(d - 1, d)
        dofs = locate_dofs_topological(Q, d - 1, self._facets)
        self._dofs = np.asarray(dofs, dtype=np.int32)
        self._dofs_dev = torch.from_numpy(self._dofs).to(dev)
        self._zero_dev = torch.zeros(self._dofs.shape[0], dtype=torch.float64, device=dev)
        self._bc = _BCHandle(self._dofs)
        # nodal values of h on the facets' pressure dofs
        self._h = torch.zeros(nq_loc, dtype=torch.float64, device=dev)
        self._xq = np.ascontiguousarray(Q.tabulate_dof_coordinates()[self._dofs].T)
        self._rhs = [_SurfaceForm(self, i) for i in range(d)]
        if callable(self._value):
            self._u = True
        self.update_bc(force=True)

    def update_bc(self, force: bool = False):
        """Re-evaluate h (reference bcs.py:255-260 re-interpolates a callable value)."""
        if callable(self._value):
            if hasattr(self, "_u") or force:
                g = np.asarray(self._value(self._xq), dtype=np.float64).reshape(-1)
                self._h[self._dofs_dev.to(torch.int64)] = torch.from_numpy(np.ascontiguousarray(g)).to(self._h.device)
        else:
            v = float(self._value.value) if isinstance(self._value, Constant) else float(self._value)
            self._h[self._dofs_dev.to(torch.int64)] = v

    def add_surface_terms(self, B):
        """B[:, i] += int_Gamma h n_i dv/dx_i ds for every component (fracstep.py:461-465)."""
        if not callable(self._value):
            self.update_bc()  # Constants are tracked by reference
        lib = _lib.load()
        st = _lib.current_stream()
        nrb = int(self._rows_b.shape[0])
        for i, Sm in enumerate(self._S):
            _lib.check(lib.ox_spmv(Sm.ref(), _lib.ptr(self._h), _lib.ptr(self._y), 1, None, st), "ox_spmv")
            _lib.check(lib.ox_scatter_add(B.ptr(), _lib.ptr(self._rows_b32), _lib.ptr(self._y), nrb, B.nc, i, 1.0, st),
                       "ox_scatter_add")

    def surface_vector_host(self, i: int) -> np.ndarray:
        """The assembled term of component i as a full vector (tests; reference test_bcs.py:166-217)."""
        from .fem import FieldStorage

        B = FieldStorage(self._V.n_local, len(self._S) or 1, self._V.mesh.device)
        self.add_surface_terms(B)
        return B.host()[:, i].copy()

    def apply_homogeneous(self, x: Vector):
        """set_bc(b2, [bc]) with value 0 (fracstep.py:549-550)."""
        s = x._s
        _lib.check(_lib.load().ox_set_bc(s.ptr(), _lib.ptr(self._dofs_dev), _lib.ptr(self._zero_dev),
                                        int(self._dofs.shape[0]), s.nc, 0 if x._c is None else x._c,
                                        _lib.current_stream()), "ox_set_bc")

    @property
    def bc(self):
        return self._bc

    def rhs(self, i: int):
        assert i < len(self._rhs)
        return self._rhs[i]
