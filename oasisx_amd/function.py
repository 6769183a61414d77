"""``Projector``: mass-matrix (L2) projection into a scalar Lagrange space with its own Krylov
solver (reference src/oasisx/function.py:13-143).

The reference projects an arbitrary UFL expression; without UFL the right-hand side
``int function * v dx`` comes from the object passed as ``function``:

* a :class:`oasisx_amd.fem.Function` on the same space -> ``b = M f`` (one SpMV),
* any object with ``assemble_rhs_into(storage)`` that fills the device block with the assembled
  linear form -- ``FractionalStep_AB_CN`` passes the rotational pressure update
  ``p + dp - xi nu div(u)`` this way (reference fracstep.py:237-247), or
* a callable ``f(x)`` with ``x`` of shape (3, npts) -> (npts,) (what ``ufl`` expressions of the
  coordinates are in the reference's ``test_projector.py``): ``int f v dx`` by a Gauss-Jacobi rule exact
  for polynomials of degree ``2 * quadrature_points - 1`` times the test function.
"""
from __future__ import annotations

import ctypes as C

from . import _lib
import numpy as np
import torch

from .fem import SLICE, DGSpace, FieldStorage, Function, FunctionSpace, _simplex_rule, cell_geometry, local_edges
from .ksp import KSPSolver
from .la import SellMatrix

__all__ = ["Projector", "LumpedProject", "grad"]


class Grad:
    """``ufl.grad(u)`` of a Lagrange field: the one expression of a field the reference's own projector test
    projects (test_projector.py:33)."""

    def __init__(self, u: Function):
        if not isinstance(u, Function) or not isinstance(u.function_space, FunctionSpace):
            raise TypeError("grad: a Function on a Lagrange space")
        self.u = u


def grad(u: Function) -> Grad:
    return Grad(u)


class Projector:
    def __init__(self, function, space, bcs=None, petsc_options=None, jit_options=None,
                 form_compiler_options=None, metadata=None):
        if bcs:
            raise NotImplementedError("Projector: Dirichlet conditions on the projection are not supported")
        lib = _lib.load()
        self._function = function
        self._space = space
        self._metadata = metadata or {}
        mesh = space.mesh
        self._dg = isinstance(space, DGSpace)
        if self._dg:
            # Discontinuous P1 target (test_projector.py:26-35): the mass matrix is block diagonal, one
            # (gdim+1) x (gdim+1) block per cell and component; its inverse is applied cell by cell in closed
            # form by ox_dg1_mass -- the direct solve the reference asks PETSc for (preonly + lu), whatever
            # Krylov options are passed.
            self._geom = space.geom
            self._cells = _lib.ox_cells(mesh.gdim, 0, int(self._geom.shape[0]), self._geom.data_ptr())
            self._A = None
            dev = mesh.device
            self._B = FieldStorage(space.num_dofs, space.dim, dev)
            self._X = FieldStorage(space.num_dofs, space.dim, dev)
            self._b = Function(space, "b", self._B, None if space.dim > 1 else 0)
            self._x = Function(space, "x", self._X, None if space.dim > 1 else 0)
            self._ksp = KSPSolver(mesh.comm, petsc_options, prefix="oasis_projector")
            if isinstance(function, Grad):
                Vu = function.u.function_space
                if Vu.mesh is not mesh or space.dim != mesh.gdim:
                    raise ValueError("Projector(grad(u), W): W must be the gdim-vector DG1 space on u's mesh")
                if not torch.equal(Vu.local_cells, space.local_cells):
                    raise ValueError("Projector(grad(u), W): the two spaces do not share the kernel cell order")
            return
        self._geom = cell_geometry(mesh, space.local_cells)
        cells = _lib.ox_cells(mesh.gdim, 0, int(self._geom.shape[0]), self._geom.data_ptr())
        # the mass matrix, assembled once (function.py:62-71)
        self._A = SellMatrix(space.pattern, symmetric=True, name="projector_mass")
        adj = space.adj.struct()
        nb, bptr, bsl, bw = space.pattern.bins_args()
        _lib.check(lib.ox_assemble_matrix(0, space.degree, C.byref(cells), _lib.ptr(space.cell_dofs), C.byref(adj),
                                          _lib.ptr(space.adj.adj_pos), space.adj.pw, self._A.ref(), nb, bptr, bsl, bw,
                                          _lib.current_stream()), "ox_assemble_matrix")
        self._A.version += 1
        dev = mesh.device
        self._B = FieldStorage(space.n_local, 1, dev)
        self._X = FieldStorage(space.n_local, 1, dev)
        self._b = Function(space, "b", self._B, 0)
        self._x = Function(space, "x", self._X, 0)
        self._ksp = KSPSolver(mesh.comm, petsc_options, prefix="oasis_projector")
        self._ksp.setOperators(self._A)

    def assemble_rhs(self):
        """Update the RHS by re-assembling (function.py:110-119)."""
        f = self._function
        if self._dg:
            lib, st = _lib.load(), _lib.current_stream()
            if isinstance(f, Grad):  # assemble_vector(inner(grad(u), v) * dx)
                Vu = f.u.function_space
                _lib.check(lib.ox_dg1_grad_rhs(Vu.degree, C.byref(self._cells), _lib.ptr(Vu.cell_dofs),
                                               C.c_void_p(f.u._storage.dev().data_ptr() if f.u._storage.nc == 1 else
                                                          f.u._storage.dev()[:, f.u._comp].contiguous().data_ptr()),
                                               self._B.ptr(), st), "ox_dg1_grad_rhs")
            elif isinstance(f, Function) and f.function_space is self._space:  # b = M f
                _lib.check(lib.ox_dg1_mass(0, C.byref(self._cells), self._space.dim, f._storage.ptr(), self._B.ptr(), st),
                           "ox_dg1_mass")
            else:
                raise TypeError("Projector into a DG space: `function` must be grad(u) of a Lagrange field or a "
                                "Function on the space")
            return
        if isinstance(f, Function):
            self._A.mult(f._storage.dev(), self._B.dev(), 1)
        elif hasattr(f, "assemble_rhs_into"):
            f.assemble_rhs_into(self._B)
        elif callable(f):
            self._B.dev()[: self._space.n_local, 0] = load_vector(self._space, f, self._geom,
                                                                 (metadata_points(self._metadata, self._space.degree)))
        else:
            raise TypeError("Projector: `function` must be a Function on the space, a callable f(x) or provide "
                            "assemble_rhs_into()")

    def solve(self, assemble_rhs: bool = True):
        """Compute the projection; returns the KSP converged reason (function.py:121-135)."""
        if assemble_rhs:
            self.assemble_rhs()
        if self._dg:  # block-diagonal mass matrix: the cell-wise inverse IS the solve
            _lib.check(_lib.load().ox_dg1_mass(1, C.byref(self._cells), self._space.dim, self._B.ptr(), self._X.ptr(),
                                               _lib.current_stream()), "ox_dg1_mass")
            direct = str(self._ksp._options.get("ksp_type", "")).lower() == "preonly"
            return _lib.CONVERGED_ITS if direct else _lib.CONVERGED_RTOL
        return self._ksp.solve_block(self._B, self._X)[0]

    @property
    def x(self):
        return self._x


def metadata_points(metadata: dict, degree: int) -> int:
    """Points per direction of the Gauss-Jacobi rule: ``metadata["quadrature_degree"]`` (the reference's
    form-compiler metadata) if given, else exact for a degree ``degree + 3`` integrand times the test function."""
    q = metadata.get("quadrature_degree")
    if q is None:
        q = 2 * degree + 3
    return int(q) // 2 + 1


def load_vector(V: FunctionSpace, f, geom: torch.Tensor, n_points: int, chunk: int = 1 << 20) -> torch.Tensor:
    """b_i = int f phi_i dx over the local cells, for a callable ``f(x)``, x: (3, npts).  Element vectors
    cell by cell, then every dof SUMS its (cell, local index) contributions in the fixed order of the
    space's dof -> cell adjacency (no atomics: bit-reproducible).  A set-up / diagnostics functional in
    torch, not part of the time-step path.  A callable marked ``supports_torch`` is evaluated on the
    device, any other on the host."""
    mesh = V.mesh
    d, dev = mesh.gdim, mesh.device
    bary_np, w_np = _simplex_rule(d, n_points)
    nv = d + 1
    if V.degree == 1:
        phi_np = bary_np
    else:
        cols = [bary_np[:, a] * (2 * bary_np[:, a] - 1) for a in range(nv)]
        cols += [4 * bary_np[:, a] * bary_np[:, b] for a, b in local_edges(d)]
        phi_np = np.stack(cols, axis=1)
    bary = torch.from_numpy(bary_np).to(dev)
    wphi = torch.from_numpy(w_np[:, None] * phi_np).to(dev)  # (NQ, nd)
    cells = mesh.cells[V.local_cells]
    nc, nd = int(cells.shape[0]), V.nd
    adet = geom[:, d * d]
    bloc = torch.empty((nc, nd), dtype=torch.float64, device=dev)
    on_dev = getattr(f, "supports_torch", False)
    for c0 in range(0, nc, chunk):
        xc = mesh.coords[cells[c0:c0 + chunk]]  # (m, d+1, d)
        xq = torch.einsum("qa,mak->mqk", bary, xc)  # (m, NQ, d)
        X = torch.zeros((3, xq.shape[0] * xq.shape[1]), dtype=torch.float64, device=dev)
        X[:d] = xq.reshape(-1, d).T
        fq = f(X) if on_dev else torch.from_numpy(np.asarray(f(X.cpu().numpy()), dtype=np.float64)).to(dev)
        fq = fq.reshape(xq.shape[0], xq.shape[1])
        bloc[c0:c0 + chunk] = adet[c0:c0 + chunk, None] * (fq @ wphi)
    # gather: pair (t, lane) of slice s sits at adj_ptr[s] + t*64 + lane
    adj = V.adj
    n = V.n_local
    out = torch.zeros(n, dtype=torch.float64, device=dev)
    bflat = bloc.reshape(-1)
    T_all = ((adj.adj_ptr[1:] - adj.adj_ptr[:-1]) // SLICE)
    rows_per = max(SLICE, (chunk // max(int(T_all.max().item()), 1)) // SLICE * SLICE)
    for r0 in range(0, n, rows_per):
        r = torch.arange(r0, min(r0 + rows_per, n), device=dev)
        sl, lane = r // SLICE, r % SLICE
        T = T_all[sl]
        tmax = int(T.max().item())
        t = torch.arange(tmax, device=dev)
        idx = adj.adj_ptr[sl][:, None] + t[None, :] * SLICE + lane[:, None]
        ok = t[None, :] < T[:, None]
        idx = torch.where(ok, idx, torch.zeros_like(idx))
        cell = adj.adj_cell[idx].to(torch.int64)
        ok = ok & (cell >= 0)
        val = bflat[torch.where(ok, cell * nd + adj.adj_loc[idx].to(torch.int64), torch.zeros_like(cell))]
        val = torch.where(ok, val, torch.zeros_like(val))
        acc = torch.zeros(r.shape[0], dtype=torch.float64, device=dev)
        for k in range(tmax):  # fixed order of the adjacency: the same sum on every run
            acc = acc + val[:, k]
        out[r] = acc
    return out


class LumpedProject:
    """Projector using a lumped mass matrix (raises in the reference too, function.py:146-153)."""

    def __init__(self):
        raise NotImplementedError
