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
  for polynomials of degree ``2 * quadrature_points - 1`` times the test function, or
* an :class:`Expression` ``Expression(fn, *operands)``: a POINTWISE expression of the coordinates, of fields and of
  gradients of fields -- ``fn(x, *values)`` is called with the operands tabulated at the quadrature points (what a UFL
  expression built from ``SpatialCoordinate``, ``Function`` and ``grad(Function)`` evaluates to there), scalar or
  vector valued.  The target may be a scalar or blocked Lagrange space (one mass matrix, ``dim`` right-hand sides in
  one block solve) or a DG1 space.
"""
from __future__ import annotations

import ctypes as C

from . import _lib
import numpy as np
import torch

from .fem import (DGSpace, FieldStorage, Function, FunctionSpace, VectorFunctionSpace, _simplex_rule, cell_geometry,
                  lagrange_basis, lagrange_basis_derivs)
from .ksp import KSPSolver
from .la import SellMatrix

__all__ = ["Projector", "LumpedProject", "grad", "Expression"]


class Grad:
    """``ufl.grad(u)`` of a Lagrange field: the one expression of a field the reference's own projector test
    projects (test_projector.py:33)."""

    def __init__(self, u: Function):
        if not isinstance(u, Function) or not isinstance(u.function_space, (FunctionSpace, VectorFunctionSpace)):
            raise TypeError("grad: a Function on a Lagrange space")
        self.u = u


def grad(u: Function) -> Grad:
    return Grad(u)


def _tabulate_operand(op, mesh, local_cells, geom, bary_np, c0: int, c1: int) -> torch.Tensor:
    """Values of a field (or of its gradient) at the points ``bary_np`` of the cells [c0, c1) of the kernel cell
    order: (m, nq) for a scalar field, (dim, m, nq) for a blocked one; the gradient inserts an axis of length gdim
    in front of the cells: (gdim, m, nq) / (dim, gdim, m, nq)."""
    u = op.u if isinstance(op, Grad) else op
    V = u.function_space
    Vs = V.scalar if isinstance(V, VectorFunctionSpace) else V
    if Vs.mesh is not mesh or not torch.equal(Vs.local_cells, local_cells):
        raise ValueError("Expression: every operand must live on the target's mesh (same kernel cell order)")
    d, dev = mesh.gdim, mesh.device
    vals = u._storage.rdev()
    cols = vals if u._comp is None else vals[:, u._comp:u._comp + 1]
    if getattr(Vs, "is_dg", False):
        cd = torch.arange(c0 * (d + 1), c1 * (d + 1), device=dev).reshape(c1 - c0, d + 1)
        degree = 1
    else:
        cd = Vs.cell_dofs[c0:c1].to(torch.int64)
        degree = Vs.degree
    uc = cols[cd]  # (m, nd, k)
    if isinstance(op, Grad):
        dphi = torch.from_numpy(lagrange_basis_derivs(d, degree, bary_np)).to(dev)  # (nq, nd, d + 1)
        g = geom[c0:c1, : d * d].reshape(-1, d, d)  # grad(lambda_1..d)
        gl = torch.cat([-g.sum(dim=1, keepdim=True), g], dim=1)  # (m, d + 1, d)
        out = torch.einsum("qab,mbj,mak->kjmq", dphi, gl, uc)
    else:
        phi = torch.from_numpy(np.ascontiguousarray(lagrange_basis(d, degree, bary_np))).to(dev)  # (nq, nd)
        out = torch.einsum("qa,mak->kmq", phi, uc)
    return out[0] if (u._comp is not None or u._storage.nc == 1) else out


class Expression:
    """A pointwise expression of the coordinates and of discrete fields: what a UFL expression of
    ``SpatialCoordinate``, ``Function`` and ``grad(Function)`` is to the reference's ``Projector`` (function.py:75:
    ``inner(function, v) * dx``) and body force (fracstep.py:284-289), without UFL.

    ``fn(x, *values)`` gets ``x`` (3, npts) and one value per operand -- a scalar field (npts,), a blocked field
    (dim, npts), ``grad`` of a scalar field (gdim, npts), ``grad`` of a blocked field (dim, gdim, npts) -- as torch
    tensors on the mesh's device, and returns (npts,) (or a number) for a scalar target, a sequence / tensor of
    ``dim`` such rows for a vector target.  It is evaluated at the quadrature points, never differentiated: derivatives
    enter through ``grad(u)`` operands."""

    def __init__(self, fn, *operands):
        if not callable(fn):
            raise TypeError("Expression: fn(x, *values) must be callable")
        for op in operands:
            if not isinstance(op, (Function, Grad)):
                raise TypeError("Expression: operands are Functions or grad(Function)")
        self.fn, self.operands = fn, operands

    def at(self, mesh, local_cells, geom, bary_np, c0: int, c1: int, dim: int = 0) -> torch.Tensor:
        """(m, nq) -- or (dim, m, nq) for ``dim`` > 0 -- values at the points ``bary_np`` of the cells [c0, c1)."""
        d, dev = mesh.gdim, mesh.device
        m, nq = c1 - c0, int(bary_np.shape[0])
        bary = torch.from_numpy(bary_np).to(dev)
        xq = torch.einsum("qa,mak->mqk", bary, mesh.coords[mesh.cells[local_cells[c0:c1]]])
        X = torch.zeros((3, m * nq), dtype=torch.float64, device=dev)
        X[:d] = xq.reshape(-1, d).T
        vals = []
        for op in self.operands:
            v = _tabulate_operand(op, mesh, local_cells, geom, bary_np, c0, c1)
            vals.append(v.reshape(*v.shape[:-2], m * nq))
        out = self.fn(X, *vals)
        rows = list(out) if (dim and not torch.is_tensor(out)) else None
        if rows is not None:
            out = torch.stack([torch.as_tensor(r, dtype=torch.float64, device=dev).expand(m * nq) for r in rows])
        out = torch.as_tensor(out, dtype=torch.float64, device=dev)
        want = (dim, m * nq) if dim else (m * nq,)
        try:
            out = out.expand(*want)
        except RuntimeError:
            raise ValueError(f"Expression: fn returned shape {tuple(out.shape)}, expected {want}") from None
        return out.reshape(*want[:-1], m, nq)



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
        # a blocked target: ONE mass matrix on the scalar space, ``dim`` right-hand sides solved as one block
        self._vector = isinstance(space, VectorFunctionSpace)
        self._dim = space.dim if self._vector else 1
        if self._vector and self._bcs:
            raise NotImplementedError("Projector: Dirichlet conditions on a blocked space (project the components)")
        space = space.scalar if self._vector else space
        self._scalar = space
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
        self._B = FieldStorage(space.n_local, self._dim, dev)
        self._X = FieldStorage(space.n_local, self._dim, dev)
        self._b = Function(self._space, "b", self._B, None if self._vector else 0)
        self._x = Function(self._space, "x", self._X, None if self._vector else 0)
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
            elif isinstance(f, Expression):  # assemble_vector(inner(f, v) * dx), cell by cell
                sp, d = self._space, self._space.mesh.gdim
                bary_np, w_np = _simplex_rule(d, metadata_points(self._metadata, 1))
                wphi = torch.from_numpy(w_np[:, None] * bary_np).to(sp.mesh.device)  # the P1 basis IS the barycentric point
                ncell, B = int(self._geom.shape[0]), self._B.dev()
                for c0 in range(0, ncell, 1 << 18):
                    c1 = min(ncell, c0 + (1 << 18))
                    fq = f.at(sp.mesh, sp.local_cells, self._geom, bary_np, c0, c1, sp.dim if sp.dim > 1 else 0)
                    fq = fq.reshape(sp.dim, c1 - c0, -1)
                    b = torch.einsum("qa,kmq,m->mak", wphi, fq, self._geom[c0:c1, d * d])
                    B[c0 * (d + 1): c1 * (d + 1)] = b.reshape(-1, sp.dim)
            else:
                raise TypeError("Projector into a DG space: `function` must be grad(u) of a Lagrange field, a "
                                "Function on the space or an Expression")
            return
        A0 = self._A0 if self._bcs else self._A
        V = self._scalar
        if isinstance(f, Function):
            if (f._comp is None and f._storage.nc > 1) != self._vector or (self._vector and f._storage.nc != self._dim):
                raise ValueError("Projector: a Function on the target space (blocked for a blocked target)")
            if self._vector or f._storage.nc == 1:
                src = f._storage.rdev()
            else:
                src = f._storage.rdev()[:, f._comp].contiguous().unsqueeze(1)
            A0.mult(src, self._B.dev(), self._dim)
        elif hasattr(f, "assemble_rhs_into"):
            f.assemble_rhs_into(self._B)
        elif isinstance(f, Expression) or callable(f):
            npts = metadata_points(self._metadata, V.degree)
            if self._vector:
                load_vector(V, f, self._geom, npts, out=self._B.dev(), dim=self._dim)
            else:
                load_vector(V, f, self._geom, npts, out=self._B.dev()[:, 0])
        else:
            raise TypeError("Projector: `function` must be a Function on the space, an Expression, a callable f(x) or "
                            "provide assemble_rhs_into()")
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


def load_vector(V: FunctionSpace, f, geom: torch.Tensor, n_points: int, chunk: int = 1 << 20, out=None,
                dim: int = 0) -> torch.Tensor:
    """b_i = int f phi_i dx over the local cells, for a callable ``f(x)``, x: (3, npts) -> (npts,): the
    ``force * v * dx`` / ``inner(function, v) * dx`` of a spatial expression (reference fracstep.py:284-289,
    function.py:75).  ``f`` is tabulated at the quadrature points of every cell (collapsed Gauss-Jacobi rule with
    ``n_points`` per direction; a callable marked ``supports_torch`` on the device, any other on the host); the
    sums are the library's (``ox_assemble_load_vector``: one lane per row over the row's cells in adjacency order,
    no atomics -- bit-reproducible).  ``f`` may be an :class:`Expression` (fields and their gradients enter the
    integrand); with ``dim`` > 0 the integrand has ``dim`` components and the result is (n_local, dim)."""
    mesh = V.mesh
    d, dev = mesh.gdim, mesh.device
    bary_np, w_np = _simplex_rule(d, n_points)
    from .fem import lagrange_basis

    phi_np = lagrange_basis(d, V.degree, bary_np)
    bary = torch.from_numpy(bary_np).to(dev)
    wphi = torch.from_numpy(np.ascontiguousarray(w_np[:, None] * phi_np)).to(dev)  # (NQ, nd)
    cells = mesh.cells[V.local_cells]
    nc, nq = int(cells.shape[0]), int(bary_np.shape[0])
    k = max(int(dim), 1)
    fq = torch.empty((k, nc, nq), dtype=torch.float64, device=dev)
    on_dev = getattr(f, "supports_torch", False)
    for c0 in range(0, nc, chunk):
        if isinstance(f, Expression):
            c1 = min(nc, c0 + chunk)
            fq[:, c0:c1] = f.at(mesh, V.local_cells, geom, bary_np, c0, c1, dim).reshape(k, c1 - c0, nq)
            continue
        xc = mesh.coords[cells[c0:c0 + chunk]]  # (m, d+1, d)
        xq = torch.einsum("qa,mak->mqk", bary, xc)  # (m, NQ, d)
        X = torch.zeros((3, xq.shape[0] * xq.shape[1]), dtype=torch.float64, device=dev)
        X[:d] = xq.reshape(-1, d).T
        want = (k, X.shape[1]) if dim else (X.shape[1],)
        v = f(X).expand(*want) if on_dev else torch.from_numpy(np.array(np.broadcast_to(
            np.asarray(f(X.cpu().numpy()), dtype=np.float64), want))).to(dev)
        fq[:, c0:c0 + chunk] = v.reshape(k, xq.shape[0], xq.shape[1])
    if out is None:
        out = torch.zeros((V.n_local, k) if dim else V.n_local, dtype=torch.float64, device=dev)
    cstruct = _lib.ox_cells(d, 0, int(geom.shape[0]), geom.data_ptr())
    adj = V.adj.struct()
    for comp in range(k):
        col = torch.zeros(V.n_local, dtype=torch.float64, device=dev) if dim else out
        _lib.check(_lib.load().ox_assemble_load_vector(C.byref(cstruct), C.byref(adj), V.n_owned, V.nd, nq, _lib.ptr(wphi),
                                                       C.c_void_p(fq[comp].data_ptr()), _lib.ptr(col),
                                                       _lib.current_stream()), "ox_assemble_load_vector")
        if dim:
            out[: V.n_local, comp] = col
    return out


class LumpedProject:
    """Projector using a lumped mass matrix (raises in the reference too, function.py:146-153)."""

    def __init__(self):
        raise NotImplementedError
