"""Boundary conditions with the reference's surface (reference src/oasisx/bcs.py:1-268).

``DirichletBC`` keeps the dof list and the boundary values on the device; ``apply`` is the
``set_bc`` kernel (b[dofs] = g, reference bcs.py:135-139).  A callable value is evaluated at
the coordinates of the constrained dofs only -- the reference interpolates the whole field
(bcs.py:121-133) but ``set_bc`` reads nothing except those entries.
"""
from __future__ import annotations

from enum import Enum
from typing import Callable

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


class PressureBC:
    """Natural pressure condition on outlet facets (reference bcs.py:142-268).

    Not on the hot path of any benchmark configuration (all are enclosed, Dirichlet-only
    flows); constructing one is allowed, using it in ``FractionalStep_AB_CN`` raises
    ``NotImplementedError`` until the facet kernel lands."""

    def __init__(self, value, marker):
        self._subdomain_data, self._subdomain_id = marker
        self._value = value

    def create_bcs(self, V, Q):
        raise NotImplementedError("PressureBC (outlet facet term) is not implemented on the HIP path yet")

    def update_bc(self):
        raise NotImplementedError

    @property
    def bc(self):
        raise NotImplementedError

    def rhs(self, i: int):
        raise NotImplementedError
