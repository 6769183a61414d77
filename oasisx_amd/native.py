"""ctypes views of the library's set-up objects (``ox_mesh`` / ``ox_space`` / ``ox_rect``,
include/oasisx_hip.h): on the GPU the mesh -> dof numbering -> SELL-64 pattern -> adjacency pipeline
runs inside ``liboasisx_hip.so`` (csrc/ox_setup.hip); the arrays it owns are wrapped as torch tensors
WITHOUT copying so the rest of the Python host (boundary conditions, tests, diagnostics) reads them
as before.  The torch implementation in fem.py remains for CPU-only hosts (host-logic tests) and for
mesh-partitioned spaces."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib


class _DevArray:
    """Borrowed device memory exposed through __cuda_array_interface__; keeps its owner alive."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(int(s) for s in shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}
        self._owner = owner


_TYPES = {torch.float64: "<f8", torch.int64: "<i8", torch.int32: "<i4", torch.int16: "<i2", torch.uint8: "|u1"}


def dev_tensor(ptr, shape, dtype, owner, device) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    if n == 0 or not ptr:
        return torch.zeros(tuple(shape), dtype=dtype, device=device)
    return torch.as_tensor(_DevArray(ptr, shape, _TYPES[dtype], owner), device=device)


def host_array(ptr, n, ctype, dtype) -> np.ndarray:
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(int(n),)).astype(dtype, copy=True)


class _Handle:
    """Owner of a library object.  It is destroyed in the process that created it only: a forked child
    (multiprocessing) inherits a copy of this wrapper but has no GPU context, and its garbage collector
    must not call into HIP."""

    def __init__(self, ptr, destroy):
        self.ptr, self._destroy, self._pid = ptr, destroy, os.getpid()

    def __del__(self):
        try:
            if self.ptr and os.getpid() == self._pid:
                self._destroy(self.ptr)
        except Exception:
            pass
        self.ptr = None


class NativeMesh:
    """``ox_mesh`` of a :class:`oasisx_amd.mesh.Mesh` (kernel cell order, geometry); cached on the mesh."""

    def __init__(self, mesh, tile_bits: int = -1):
        lib = _lib.load()
        out = C.c_void_p()
        cells32 = mesh.cells.to(torch.int32).contiguous()
        _lib.check(lib.ox_mesh_create(_lib.ptr(mesh.coords), mesh.num_vertices, _lib.ptr(cells32), mesh.num_cells,
                                      mesh.gdim, 1, int(tile_bits), C.byref(out)), "ox_mesh_create")
        self.handle = _Handle(out, lib.ox_mesh_destroy)
        v = _lib.ox_mesh_info()
        _lib.check(lib.ox_mesh_view(out, C.byref(v)), "ox_mesh_view")
        self.info = v
        dev = mesh.device
        gs = 6 if mesh.gdim == 2 else 10
        self.cell_perm = dev_tensor(v.cell_perm, (mesh.num_cells,), torch.int32, self.handle, dev)
        self.geom = dev_tensor(v.cells_struct.geom, (mesh.num_cells, gs), torch.float64, self.handle, dev)
        self.tile_bits = int(v.tile_bits)

    @staticmethod
    def of(mesh):
        nm = getattr(mesh, "_native", None)
        if nm is None:
            nm = mesh._native = NativeMesh(mesh)
        return nm


def pattern_from_info(P, owner, device, SellPattern):
    """fem.SellPattern over the library-owned arrays of an ``ox_pattern_info``."""
    ns = int(P.sell.n_slices)
    slice_ptr = dev_tensor(P.sell.slice_ptr, (ns + 1,), torch.int64, owner, device)
    cols = dev_tensor(P.sell.cols, (int(P.size),), torch.int32, owner, device)
    row_len = dev_tensor(P.row_len, (int(P.sell.n_rows),), torch.int32, owner, device)
    widths = host_array(P.widths_host, ns, C.c_int32, np.int32)
    row_blk = (dev_tensor(P.row_blk_ptr, (int(P.n_row_blocks) + 1,), torch.int32, owner, device), int(P.row_blk_entries))
    pat = SellPattern(int(P.sell.n_rows), int(P.sell.n_cols), slice_ptr, cols, row_len, widths, nnz=int(P.nnz),
                      row_blk=row_blk)
    if P.size > 0 and P.sell.cols16:
        pat.cols16 = dev_tensor(P.sell.cols16, (int(P.size),), torch.int16, owner, device)
        pat.cbase = dev_tensor(P.sell.cbase, (2 * (int(P.size) // 128),), torch.int32, owner, device)
        pat.frac16 = int(P.n_compressed) / int(P.size)
    return pat


class NativeSubMesh:
    """``ox_mesh`` of ONE RANK'S PART of a mesh (``ox_mesh_create_sub``): the rank's local cells with their
    vertices renumbered in ascending global id, ordered on the whole mesh's key lattice.  Cached on the
    partition object so that the velocity and the pressure space share cells and geometry."""

    def __init__(self, mesh, part):
        from .fem import default_tile_bits, mesh_is_lattice

        lib = _lib.load()
        dev = mesh.device
        self.cells_global = part.local_cells  # ascending global cell ids
        cg = mesh.cells[self.cells_global]
        self.verts = torch.unique(cg.reshape(-1))  # ascending global vertex ids = the part's vertex numbering
        cells32 = torch.searchsorted(self.verts, cg.reshape(-1)).reshape(cg.shape).to(torch.int32).contiguous()
        coords = mesh.coords[self.verts].contiguous()
        lo = mesh.coords.min(dim=0).values
        span = (mesh.coords.max(dim=0).values - lo).clamp_min(1e-300)
        lo3, sp3 = (C.c_double * 3)(0.0, 0.0, 0.0), (C.c_double * 3)(1.0, 1.0, 1.0)
        for k in range(mesh.gdim):
            lo3[k], sp3[k] = float(lo[k]), float(span[k])
        out = C.c_void_p()
        nc = int(cells32.shape[0])
        _lib.check(lib.ox_mesh_create_sub(_lib.ptr(coords), int(self.verts.shape[0]), _lib.ptr(cells32), nc, mesh.gdim, 1,
                                          lo3, sp3, int(mesh_is_lattice(mesh)), int(default_tile_bits(mesh)),
                                          mesh.num_cells, C.byref(out)), "ox_mesh_create_sub")
        self.handle = _Handle(out, lib.ox_mesh_destroy)
        v = _lib.ox_mesh_info()
        _lib.check(lib.ox_mesh_view(out, C.byref(v)), "ox_mesh_view")
        self.info = v
        gs = 6 if mesh.gdim == 2 else 10
        self.cell_perm = dev_tensor(v.cell_perm, (nc,), torch.int32, self.handle, dev)
        self.geom = dev_tensor(v.cells_struct.geom, (nc, gs), torch.float64, self.handle, dev)
        self.tile_bits = int(v.tile_bits)

    @staticmethod
    def of(mesh, part):
        nm = getattr(part, "_native_sub", None)
        if nm is None:
            nm = part._native_sub = NativeSubMesh(mesh, part)
        return nm


class NativeSpace:
    def __init__(self, mesh, degree: int, window: int, part=None, owner=None, n_dofs_whole: int = 0, brick: bool = False):
        lib = _lib.load()
        out = C.c_void_p()
        if part is not None:  # one rank's piece of a mesh-partitioned space
            self.nmesh = NativeSubMesh.of(mesh, part)
            _lib.check(lib.ox_space_create_part(self.nmesh.handle.ptr, int(degree), int(window), _lib.ptr(owner),
                                                int(owner.shape[0]), int(part.rank), int(n_dofs_whole), C.byref(out)),
                       "ox_space_create_part")
        else:
            self.nmesh = NativeMesh.of(mesh)
            _lib.check(lib.ox_space_create_ordered(self.nmesh.handle.ptr, int(degree), int(window), 1 if brick else 0,
                                                   C.byref(out)), "ox_space_create_ordered")
        self.handle = _Handle(out, lib.ox_space_destroy)
        self.handle._mesh = self.nmesh  # the space reads the mesh object: keep it alive
        v = _lib.ox_space_info()
        _lib.check(lib.ox_space_view(out, C.byref(v)), "ox_space_view")
        self.info = v


class NativeRect:
    def __init__(self, R: NativeSpace, Cs: NativeSpace):
        lib = _lib.load()
        out = C.c_void_p()
        _lib.check(lib.ox_rect_create(R.handle.ptr, Cs.handle.ptr, C.byref(out)), "ox_rect_create")
        self.handle = _Handle(out, lib.ox_rect_destroy)
        self.handle._spaces = (R, Cs)
        v = _lib.ox_rect_info()
        _lib.check(lib.ox_rect_view(out, C.byref(v)), "ox_rect_view")
        self.info = v
