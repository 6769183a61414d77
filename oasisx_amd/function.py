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
        lib = _lib.load()
        self._bcs = list(bcs or [])
        self._function = function
        self._space = space
        self._metadata = metadata or {}
        mesh = space.mesh
        self._dg = isinstance(space, DGSpace)
        if self._bcs and self._dg:
            raise NotImplementedError("Projector: Dirichlet conditions on a discontinuous space")
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
        if self._bcs:
            # assemble_matrix(lhs, bcs=bcs) (function.py:69-70): rows AND columns of the constrained dofs -> identity;
            # the untouched matrix is kept for the lifting of the right-hand side (function.py:114-115)
            for bc in self._bcs:
                if getattr(bc, "_V", None) is None:
                    bc.create_bc(space)
                elif bc._V is not space:
                    raise ValueError("Projector: a Dirichlet condition created on another space")
            self._A0 = SellMatrix(space.pattern, symmetric=True, name="projector_mass_unconstrained")
            self._A0.vals.copy_(self._A.vals)
            self._A0.version += 1
            is_bc = torch.zeros(space.n_local, dtype=torch.uint8, device=dev)
            for bc in self._bcs:
                is_bc[bc._dofs_dev.to(torch.int64)] = 1
            _lib.check(lib.ox_zero_rows_cols(self._A.ref(), _lib.ptr(is_bc), 1.0, _lib.current_stream()),
                       "ox_zero_rows_cols")
            self._A.version += 1
            self._G = FieldStorage(space.n_local, 1, dev)  # the Dirichlet values as a field, zero elsewhere
            self._L = FieldStorage(space.n_local, 1, dev)  # its image under the unconstrained matrix
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
                # one column of an interleaved block is gathered into a buffer this object OWNS until the next
                # assembly (a temporary would be released before the kernel runs)
                us = f.u._storage
                self._ucol = us.rdev() if us.nc == 1 else us.rdev()[:, f.u._comp].contiguous()
                _lib.check(lib.ox_dg1_grad_rhs(Vu.degree, C.byref(self._cells), _lib.ptr(Vu.cell_dofs),
                                               C.c_void_p(self._ucol.data_ptr()), self._B.ptr(), st), "ox_dg1_grad_rhs")
            elif isinstance(f, Function) and f.function_space is self._space:  # b = M f
                _lib.check(lib.ox_dg1_mass(0, C.byref(self._cells), self._space.dim, f._storage.ptr(), self._B.ptr(), st),
                           "ox_dg1_mass")
            else:
                raise TypeError("Projector into a DG space: `function` must be grad(u) of a Lagrange field or a "
                                "Function on the space")
            return
        A0 = self._A0 if self._bcs else self._A
        if isinstance(f, Function):
            src = f._storage.rdev() if f._storage.nc == 1 else f._storage.rdev()[:, f._comp].contiguous().unsqueeze(1)
            A0.mult(src, self._B.dev(), 1)
        elif hasattr(f, "assemble_rhs_into"):
            f.assemble_rhs_into(self._B)
        elif callable(f):
            load_vector(self._space, f, self._geom, metadata_points(self._metadata, self._space.degree),
                        out=self._B.dev()[:, 0])
        else:
            raise TypeError("Projector: `function` must be a Function on the space, a callable f(x) or provide "
                            "assemble_rhs_into()")
        if self._bcs:
            # apply_lifting(b, [lhs], bcs=[bcs]): b -= A0 g with g the Dirichlet values (zero elsewhere), then
            # set_bc(b, bcs): b[dofs] = g                                            (function.py:114-118)
            lib, st = _lib.load(), _lib.current_stream()
            n = self._space.n_local
            self._G.dev().zero_()
            for bc in self._bcs:
                bc.apply(self._G_vec())
            self._A0.mult(self._G.dev(), self._L.dev(), 1)
            _lib.check(lib.ox_axpby(n, 1.0, self._B.ptr(), -1.0, self._L.ptr(), self._B.ptr(), st), "ox_axpby")
            for bc in self._bcs:
                bc.apply(self._b.x)

    def _G_vec(self):
        from .fem import Vector

        return Vector(self._G, 0)

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


def load_vector(V: FunctionSpace, f, geom: torch.Tensor, n_points: int, chunk: int = 1 << 20, out=None) -> torch.Tensor:
    """b_i = int f phi_i dx over the local cells, for a callable ``f(x)``, x: (3, npts) -> (npts,): the
    ``force * v * dx`` / ``inner(function, v) * dx`` of a spatial expression (reference fracstep.py:284-289,
    function.py:75).  ``f`` is tabulated at the quadrature points of every cell (collapsed Gauss-Jacobi rule with
    ``n_points`` per direction; a callable marked ``supports_torch`` on the device, any other on the host); the
    sums are the library's (``ox_assemble_load_vector``: one lane per row over the row's cells in adjacency order,
    no atomics -- bit-reproducible)."""
    mesh = V.mesh
    d, dev = mesh.gdim, mesh.device
    bary_np, w_np = _simplex_rule(d, n_points)
    from .fem import lagrange_basis

    phi_np = lagrange_basis(d, V.degree, bary_np)
    bary = torch.from_numpy(bary_np).to(dev)
    wphi = torch.from_numpy(np.ascontiguousarray(w_np[:, None] * phi_np)).to(dev)  # (NQ, nd)
    cells = mesh.cells[V.local_cells]
    nc, nq = int(cells.shape[0]), int(bary_np.shape[0])
    fq = torch.empty((nc, nq), dtype=torch.float64, device=dev)
    on_dev = getattr(f, "supports_torch", False)
    for c0 in range(0, nc, chunk):
        xc = mesh.coords[cells[c0:c0 + chunk]]  # (m, d+1, d)
        xq = torch.einsum("qa,mak->mqk", bary, xc)  # (m, NQ, d)
        X = torch.zeros((3, xq.shape[0] * xq.shape[1]), dtype=torch.float64, device=dev)
        X[:d] = xq.reshape(-1, d).T
        v = f(X) if on_dev else torch.from_numpy(np.array(np.broadcast_to(
            np.asarray(f(X.cpu().numpy()), dtype=np.float64), (X.shape[1],)))).to(dev)
        fq[c0:c0 + chunk] = v.reshape(xq.shape[0], xq.shape[1])
    if out is None:
        out = torch.zeros(V.n_local, dtype=torch.float64, device=dev)
    cstruct = _lib.ox_cells(d, 0, int(geom.shape[0]), geom.data_ptr())
    adj = V.adj.struct()
    _lib.check(_lib.load().ox_assemble_load_vector(C.byref(cstruct), C.byref(adj), V.n_owned, V.nd, nq, _lib.ptr(wphi),
                                                   _lib.ptr(fq), _lib.ptr(out), _lib.current_stream()),
               "ox_assemble_load_vector")
    return out


class LumpedProject:
    """Projector using a lumped mass matrix (raises in the reference too, function.py:146-153)."""

    def __init__(self):
        raise NotImplementedError
