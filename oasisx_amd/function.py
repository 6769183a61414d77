"""``Projector``: mass-matrix (L2) projection into a scalar Lagrange space with its own Krylov
solver (reference src/oasisx/function.py:13-143).

The reference projects an arbitrary UFL expression; without UFL the right-hand side
``int function * v dx`` comes from the object passed as ``function``:

* a :class:`oasisx_amd.fem.Function` on the same space -> ``b = M f`` (one SpMV), or
* any object with ``assemble_rhs_into(storage)`` that fills the device block with the assembled
  linear form -- ``FractionalStep_AB_CN`` passes the rotational pressure update
  ``p + dp - xi nu div(u)`` this way (reference fracstep.py:237-247).
"""
from __future__ import annotations

import ctypes as C

from . import _lib
from .fem import FieldStorage, Function, FunctionSpace, cell_geometry
from .ksp import KSPSolver
from .la import SellMatrix

__all__ = ["Projector", "LumpedProject"]


class Projector:
    def __init__(self, function, space: FunctionSpace, bcs=None, petsc_options=None, jit_options=None,
                 form_compiler_options=None, metadata=None):
        if bcs:
            raise NotImplementedError("Projector: Dirichlet conditions on the projection are not supported")
        lib = _lib.load()
        self._function = function
        self._space = space
        mesh = space.mesh
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
        if isinstance(f, Function):
            self._A.mult(f._storage.dev(), self._B.dev(), 1)
        elif hasattr(f, "assemble_rhs_into"):
            f.assemble_rhs_into(self._B)
        else:
            raise TypeError("Projector: `function` must be a Function on the space or provide assemble_rhs_into()")

    def solve(self, assemble_rhs: bool = True):
        """Compute the projection; returns the KSP converged reason (function.py:121-135)."""
        if assemble_rhs:
            self.assemble_rhs()
        return self._ksp.solve_block(self._B, self._X)[0]

    @property
    def x(self):
        return self._x


class LumpedProject:
    """Projector using a lumped mass matrix (raises in the reference too, function.py:146-153)."""

    def __init__(self):
        raise NotImplementedError
