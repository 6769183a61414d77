"""Lagrange P1/P2 function spaces, fields and the device data layout of the IPCS path.

A ``FunctionSpace`` owns, as torch tensors on the setup device:

* the cell->dof table in the FINAL dof numbering.  Dofs are first ordered by (tile_z, tile_y, z, y, x)
  (``locality_key``) for gather coalescing + L2 locality and then, inside windows of ``window`` rows,
  stably by decreasing row length: the numbering itself is the SELL-64 row order, so a
  wave's 64 rows have (nearly) equal length and equal cell counts -- no padding waste, no
  divergence, no row permutation at run time;
* the SELL-64 sparsity pattern of the square operator on the space (see include/oasisx_hip.h);
* the dof->cell adjacency in the same slice order, with per-(row, cell) position bytes that
  tell the row-centric assembly kernels where each cell dof sits inside the row.

The DOLFINx-shaped surface (``Function.x.array``, ``interpolate``, ``Constant``,
``locate_dofs_*``) is what the oasisx callers touch (reference bcs.py:103-139,
demo/taylor_green.py:142-182, fracstep.py:187-216,698-705).
"""
from __future__ import annotations

import ctypes as C
import itertools

import numpy as np
import torch

from . import _lib
from .mesh import Mesh

KV = 2
SLICE = 64
default_scalar_type = np.float64


def local_edges(gdim: int):
    return [(1, 2), (0, 2), (0, 1)] if gdim == 2 else [(2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1)]


import os as _os


def default_tile_bits(mesh: Mesh) -> int:
    """log2 of the tiles per slow direction for ``locality_key``: tiles of about 24 vertex lines
    (48 P2 lines) a side, whatever the mesh size -- 2 at 128^3 (measured on the whole step: 117.7 ms
    with 2, 119.1 with 3, 120.8 with 4), 3 at 256^3.  OX_TILE_BITS overrides (tuning hook)."""
    env = _os.environ.get("OX_TILE_BITS")
    if env is not None:
        return int(env)
    lines = max(float(mesh.num_vertices) ** (1.0 / mesh.gdim), 1.0)
    return int(min(6, max(0, round(np.log2(lines / 24.0)))))


def default_key_bits(n_points: int, gdim: int, tile_bits: int) -> int:
    """Bits per coordinate of ``locality_key``'s tiled order: 4 lattice steps per mean point spacing
    (L = ceil(n_points^(1/gdim)) "lines", bits = ceil(log2 L) + 2, in integer arithmetic so that the
    library's C++ twin agrees bit for bit): fine enough that the lines of a lattice mesh stay distinct
    and ordered exactly.  (Meshes that are not lattices do not use the tiled order: ``mesh_is_lattice``.)
    OX_KEY_BITS overrides."""
    env = _os.environ.get("OX_KEY_BITS")
    if env is not None:
        return int(env)
    L = 1
    while L ** gdim < n_points:
        L += 1 if L < 64 else max(1, L // 64)
    while L > 1 and (L - 1) ** gdim >= n_points:
        L -= 1
    b = 0
    while (1 << b) < L:
        b += 1
    return int(min(18, max(b + 2, tile_bits + 1)))


def default_brick_shift(n_points: int, gdim: int, bits: int) -> int:
    """Brick order of the degree-2 spaces on lattice meshes (``locality_key(brick_shift=)``): bricks of about 8
    points a side -- 512 rows, one window block of the LDS-window SpMV (``k_spmv_win``); shift = bits -
    floor(log2(L / 8)) with L = points per direction as ``default_key_bits`` counts them, 0 (no bricks) below 16
    points per direction.  The library's twin: ox_setup.hip ``default_brick_shift``."""
    L = 1
    while L ** gdim < n_points:
        L += 1 if L < 64 else max(1, L // 64)
    while L > 1 and (L - 1) ** gdim >= n_points:
        L -= 1
    nb = L // 8
    if nb < 2:
        return 0
    lg = 0
    while (2 << lg) <= nb:
        lg += 1
    bs = bits - lg
    return bs if 0 < bs < bits else 0


def mesh_is_lattice(mesh: Mesh) -> bool:
    """Vertices on a tensor grid?  Such a mesh has about n^(1/d) distinct values per coordinate, an
    unstructured one about n.  Lattice meshes are ordered by ``locality_key``'s tiles (whole x-lines stay
    contiguous: the gathers of a wave coalesce); anything else along a Z-order curve, so that 64
    consecutive rows are a compact cluster instead of a thin tube through the whole x extent (Delaunay
    mesh, 2.3 M P2 rows: SpMV 350 -> 262 us).  The library's C++ twin: ox_mesh_create.  OX_ORDER=tiles|curve
    overrides.  Cached on the mesh."""
    cached = getattr(mesh, "_lattice", None)
    if cached is not None:
        return cached
    env = _os.environ.get("OX_ORDER")
    if env is not None:
        mesh._lattice = env != "curve"
        return mesh._lattice
    n, d = mesh.num_vertices, mesh.gdim
    L = 1
    while L ** d < n:
        L += 1
    lo = mesh.coords.min(dim=0).values
    span = (mesh.coords.max(dim=0).values - lo).clamp_min(1e-300)
    q = torch.round((mesh.coords - lo) * (float((1 << 20) - 1) / span)).to(torch.int64)
    lattice = all(int(torch.unique(q[:, k]).numel()) <= 4 * L + 4 for k in range(d))
    mesh._lattice = lattice
    return lattice


def locality_key(x: torch.Tensor, lo: torch.Tensor, span: torch.Tensor, tile_bits: int,
                 bits: int = 18, curve: bool = False, brick_shift: int = 0) -> torch.Tensor:
    """Ordering key of points: (tile_z, tile_y, z, y, x) -- lexicographic inside tiles that span the
    whole x extent and 1/2^tile_bits of the y and z extents.

    * whole x-lines stay contiguous, so the 64 rows of a wave (and their neighbours' columns) are
      consecutive: the x gathers of a wave-instruction coalesce;
    * rows that an XCD works on at one time form one compact y-z tile, so their gather footprint
      (tile + halo) stays near the 4 MiB L2 instead of ~5 whole lattice planes (8 MB) under
      plane-by-plane lexicographic order.
    On these LATTICE meshes a full Z-order (Morton) curve was measured 15 % slower: it destroys the
    coalescing.  ``curve=True`` (meshes that are not lattices, ``mesh_is_lattice``) is that Z-order curve."""
    d = x.shape[1]
    if curve:  # Z-order curve (unstructured meshes, mesh_is_lattice): 18 bits per coordinate
        bits = min(18, 63 // d)
    q = torch.round((x - lo) / span * float((1 << bits) - 1)).to(torch.int64)
    key = torch.zeros(x.shape[0], dtype=torch.int64, device=x.device)
    if curve:
        for b in range(bits - 1, -1, -1):
            for k in range(d - 1, -1, -1):
                key = (key << 1) | ((q[:, k] >> b) & 1)
        return key
    for k in range(d - 1, 0, -1):  # tile index of the slow directions, slowest first
        key = (key << tile_bits) | (q[:, k] >> (bits - tile_bits))
    if brick_shift > 0:  # (tile, brick_z, brick_y, brick_x, z, y, x): bricks of 2^brick_shift lattice steps a side
        for k in range(d - 1, -1, -1):
            key = (key << (bits - brick_shift)) | (q[:, k] >> brick_shift)
        for k in range(d - 1, -1, -1):
            key = (key << brick_shift) | (q[:, k] & ((1 << brick_shift) - 1))
        return key
    for k in range(d - 1, -1, -1):
        key = (key << bits) | q[:, k]
    return key


def cell_geometry(mesh: Mesh, cell_ids=None, chunk: int = 1 << 23) -> torch.Tensor:
    """[n_cells][gs] rows of grad(lambda_1..d) then |detJ| (gs = 6 in 2-D, 10 in 3-D); closed-form
    2x2 / 3x3 inverses, evaluated in chunks of cells."""
    d = mesh.gdim
    ids = torch.arange(mesh.num_cells, device=mesh.device) if cell_ids is None else cell_ids
    ncl = int(ids.shape[0])
    gs = 6 if d == 2 else 10
    geom = torch.zeros((ncl, gs), dtype=torch.float64, device=mesh.device)
    for c0 in range(0, ncl, chunk):
        x = mesh.coords[mesh.cells[ids[c0:c0 + chunk]]]  # (m, d+1, d)
        e = x[:, 1:, :] - x[:, :1, :]  # rows = edge vectors x_a - x_0; J = e^T
        g = geom[c0:c0 + chunk]
        if d == 2:
            a, b, c, dd = e[:, 0, 0], e[:, 1, 0], e[:, 0, 1], e[:, 1, 1]  # J = [[a, b], [c, dd]]
            det = a * dd - b * c
            g[:, 0], g[:, 1], g[:, 2], g[:, 3] = dd / det, -b / det, -c / det, a / det
        else:
            # rows of J^-1 = cross products of the edge vectors / det
            e1, e2, e3 = e[:, 0], e[:, 1], e[:, 2]
            c23 = torch.linalg.cross(e2, e3)
            det = (e1 * c23).sum(dim=1)
            g[:, 0:3] = c23 / det.unsqueeze(1)
            g[:, 3:6] = torch.linalg.cross(e3, e1) / det.unsqueeze(1)
            g[:, 6:9] = torch.linalg.cross(e1, e2) / det.unsqueeze(1)
        g[:, d * d] = det.abs()
    return geom


def merge_small_bins(bin_width: np.ndarray, bin_ptr: np.ndarray):
    """Width bins of the LDS-accumulating row kernels (one launch per bin, LDS sized for the bin's width):
    a bin with fewer than max(64, n_slices / 64) slices joins the next wider one.  A box mesh keeps its three
    big bins and loses the handful of two-block launches; an unstructured mesh (row lengths 10..125: ~55
    distinct widths) goes from ~55 launches per assembly to about ten.  The library's twin: finish_pattern."""
    n_slices = int(bin_ptr[-1]) if len(bin_ptr) else 0
    thresh = max(64, n_slices // 64)
    bw, bp = [], [0]
    for b in range(len(bin_width)):
        last = b == len(bin_width) - 1
        if int(bin_ptr[b + 1]) - bp[-1] >= thresh or last:
            bw.append(int(bin_width[b]))
            bp.append(int(bin_ptr[b + 1]))
    return np.asarray(bw, dtype=np.int32), np.asarray(bp, dtype=np.int64)


def row_blocks(widths: np.ndarray):
    """Row blocks of the one-launch assembly kernels (``ox_assemble_first_blocks``; the library's twin: finish_pattern):
    consecutive slices in storage order, greedily, at most ROW_BLOCK_WAVES per block and ROW_BLOCK_LDS bytes of LDS
    accumulators (width * 64 doubles per slice).  Returns (blk_ptr int32 [n_blocks + 1], entries of the largest block);
    no blocks at all when one slice alone exceeds the budget (the width bins serve such a pattern)."""
    cap = _lib.ROW_BLOCK_LDS // 8
    blk, used, cnt, big = [0], 0, 0, 0
    for s_, w in enumerate(np.asarray(widths, dtype=np.int64)):
        e = int(w) * SLICE
        if e > cap:
            return np.zeros(1, dtype=np.int32), 0
        if cnt == _lib.ROW_BLOCK_WAVES or used + e > cap:
            blk.append(s_)
            used, cnt = 0, 0
        used, cnt = used + e, cnt + 1
        big = max(big, used)
    if len(widths):
        blk.append(len(widths))
    return np.asarray(blk, dtype=np.int32), int(big)


class SellPattern:
    """SELL-64 sparsity pattern shared by every matrix on one (row space, col space)."""

    def __init__(self, n_rows, n_cols, slice_ptr, cols, row_len, widths, nnz=None, row_blk=None):
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        self.slice_ptr = slice_ptr  # int64 [n_slices+1] device
        self.cols = cols  # int32 device
        self.row_len = row_len  # int32 [n_rows] device
        self.widths = widths  # int32 [n_slices] host numpy
        self.n_slices = int(widths.shape[0])
        self.size = int(cols.shape[0])
        self.nnz = int(row_len.sum().item()) if nnz is None else int(nnz)
        self.device = cols.device
        self.dist = None  # ox_dist* (halo plan) of the column space, mesh-partitioned runs
        self.cols16 = self.cbase = None  # 16-bit column stream, built by struct() on the GPU
        self.frac16 = 0.0
        self.ib_slices, self.n_interior = None, 0  # interior / boundary split (split_interior)
        # width bins for the LDS-accumulating row kernels
        w = torch.from_numpy(widths.astype(np.int64))
        order = torch.argsort(w, stable=True)
        ws = w[order]
        uw, counts = torch.unique_consecutive(ws, return_counts=True)
        self.bin_width, self.bin_ptr = merge_small_bins(uw.numpy().astype(np.int32),
                                                        np.concatenate([[0], np.cumsum(counts.numpy())]).astype(np.int64))
        self.bin_slices = order.to(torch.int32).to(self.device)
        # row blocks of the one-launch assembly kernels: (device blk_ptr, largest block's entries); the library's own when
        # it built the pattern (native.pattern_from_info), else the twin
        if row_blk is None:
            bp, big = row_blocks(widths)
            row_blk = (torch.from_numpy(bp).to(self.device), big)
        self.row_blk_ptr, self.row_blk_entries = row_blk
        self.n_row_blocks = int(self.row_blk_ptr.shape[0]) - 1

    def blocks_args(self):
        """(n_blocks, blk_ptr, lds_entries) of ``ox_assemble_first_blocks`` / ``ox_assemble_matrix_blocks``."""
        return self.n_row_blocks, _lib.ptr(self.row_blk_ptr), int(self.row_blk_entries)

    def split_interior(self, n_owned: int):
        """Mesh-partitioned operators: list the slices with the interior ones first (no ghost column,
        i.e. no column >= n_owned, in any row), then the boundary ones -- the distributed mat-vecs
        multiply the former while the halo exchange is in flight (``ox_sell.ib_slices``)."""
        if self.n_slices == 0 or self.size == 0:
            return
        width = torch.from_numpy(self.widths.astype(np.int64)).to(self.device) * SLICE
        sl = torch.repeat_interleave(torch.arange(self.n_slices, device=self.device), width)
        ghost = torch.zeros(self.n_slices, dtype=torch.int32, device=self.device)
        ghost.index_put_((sl,), (self.cols >= n_owned).to(torch.int32), accumulate=True)
        interior = torch.nonzero(ghost == 0).reshape(-1)
        boundary = torch.nonzero(ghost > 0).reshape(-1)
        self.ib_slices = torch.cat([interior, boundary]).to(torch.int32).contiguous()
        self.n_interior = int(interior.shape[0])
        self._slice_has_ghost = ghost > 0
        self.order_window_blocks()

    def order_window_blocks(self):
        """LDS-window stream of a mesh-partitioned operator: list the window blocks with the interior ones (no slice
        with a ghost column) first -- the overlapped mat-vec multiplies those while the halo exchange is in flight
        (``ox_sell.n_wb_interior``).  The library-owned arrays are copied in the new order."""
        if getattr(self, "wcode", None) is None or getattr(self, "_slice_has_ghost", None) is None \
                or getattr(self, "n_wb_interior", None) is not None:
            return
        dev = self.device
        sl = self.wb_slices.to(torch.int64)
        flag = (self._slice_has_ghost[sl.clamp_min(0)] & (sl >= 0)).any(dim=1)
        perm = torch.argsort(flag.to(torch.int8), stable=True)
        w = (self.wb_ptr[1:] - self.wb_ptr[:-1])[perm]
        new_ptr = torch.zeros_like(self.wb_ptr)
        new_ptr[1:] = torch.cumsum(w, 0)
        src = torch.repeat_interleave(self.wb_ptr[:-1][perm] - new_ptr[:-1], w) + torch.arange(int(new_ptr[-1].item()), device=dev)
        self.wlist = self.wlist[src].contiguous()
        self.wb_ptr = new_ptr
        self.wb_slices = self.wb_slices[perm].contiguous()
        self.wb_waves = self.wb_waves[perm].contiguous()
        self.n_wb_interior = int((~flag).sum().item())

    def new_values(self) -> torch.Tensor:
        return torch.zeros(self.size, dtype=torch.float64, device=self.device)

    def build_windows(self, row_pos: torch.Tensor, sort_window: int, slices_per_block: int = 8):
        """LDS-window stream of the pattern (``ox_sell.wb_*`` / ``wlist`` / ``wcode``; kernel: k_spmv_win).

        ``row_pos[r]``: position of row r in the pure locality order (before the length sort inside windows of
        ``sort_window`` rows).  Inside every sort window the slices are ordered by the mean position of their rows and
        cut into blocks of ``slices_per_block``: on a box mesh in brick order a block is then the rows of one 8^3
        brick, whatever their lengths; on any mesh, rows that are neighbours along the curve.  Per block: the sorted
        distinct columns of its entries (the window), a 16-bit index into it for every entry, and the wave that
        multiplies each of its slices (longest-processing-time schedule over 4 waves)."""
        dev = self.device
        ns = self.n_slices
        if ns == 0 or self.size == 0:
            return False
        spb = int(slices_per_block)
        assert 1 <= spb <= 8
        width = torch.from_numpy(self.widths.astype(np.int64)).to(dev)  # entries per row of each slice
        # mean locality position of each slice's rows
        pos = torch.zeros(ns * SLICE, dtype=torch.float64, device=dev)
        pos[: self.n_rows] = row_pos[: self.n_rows].to(torch.float64)
        if self.n_rows < ns * SLICE:
            pos[self.n_rows:] = pos[self.n_rows - 1]
        cent = pos.reshape(ns, SLICE).mean(dim=1)
        spw = max(1, sort_window // SLICE)
        sl = torch.arange(ns, device=dev)
        wkey = (sl // spw).to(torch.float64) * float(2 * ns * SLICE + 2) + cent  # (sort window, centroid)
        order = torch.argsort(wkey, stable=True)
        # blocks never straddle a sort window
        win_of = (order // spw)
        first = torch.ones(ns, dtype=torch.bool, device=dev)
        first[1:] = win_of[1:] != win_of[:-1]
        idx_in_win = torch.arange(ns, device=dev) - torch.cummax(torch.where(first, torch.arange(ns, device=dev), torch.zeros_like(sl)), 0).values
        newblk = first | (idx_in_win % spb == 0)
        blk_of_ordered = torch.cumsum(newblk.to(torch.int64), 0) - 1
        nb = int(blk_of_ordered[-1].item()) + 1
        slot_in_blk = torch.arange(ns, device=dev) - torch.cummax(torch.where(newblk, torch.arange(ns, device=dev), torch.zeros_like(sl)), 0).values
        wb_slices = torch.full((nb, 8), -1, dtype=torch.int32, device=dev)
        wb_slices[blk_of_ordered, slot_in_blk] = order.to(torch.int32)
        blk_of_slice = torch.empty(ns, dtype=torch.int64, device=dev)
        blk_of_slice[order] = blk_of_ordered
        # longest-processing-time schedule of each block's slices over 4 waves
        wmat = torch.zeros((nb, 8), dtype=torch.int64, device=dev)
        ok = wb_slices >= 0
        wmat[ok] = width[wb_slices[ok].to(torch.int64)]
        by_w = torch.argsort(-wmat, dim=1, stable=True)
        load = torch.zeros((nb, 4), dtype=torch.int64, device=dev)
        waves = torch.zeros((nb, 8), dtype=torch.int64, device=dev)
        rows_b = torch.arange(nb, device=dev)
        for j in range(8):
            col = by_w[:, j]
            w = torch.argmin(load, dim=1)
            waves[rows_b, col] = w
            load[rows_b, w] += wmat[rows_b, col]
        sched = torch.zeros(nb, dtype=torch.int64, device=dev)
        for j in range(8):
            sched |= waves[:, j] << (2 * j)
        # windows: distinct (block, column) keys, chunked over blocks to bound the sort's memory
        slot_per_slice = width * SLICE
        wb_ptr = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
        wcode = torch.zeros(self.size, dtype=torch.int16, device=dev)
        lists = []
        ncols = self.n_cols
        # slices of a block are scattered: go through the slot array once per chunk of SLICES (in storage order)
        chunk_slices = max(1, int((1 << 27) // max(int(slot_per_slice.max().item()), 1)))
        sp = self.slice_ptr
        uk_all = []
        for s0 in range(0, ns, chunk_slices):
            s1 = min(ns, s0 + chunk_slices)
            a, b = int(sp[s0].item()), int(sp[s1].item())
            bs = torch.repeat_interleave(blk_of_slice[s0:s1], slot_per_slice[s0:s1])
            uk_all.append(torch.unique(bs * ncols + self.cols[a:b].to(torch.int64)))
        ukey = torch.unique(torch.cat(uk_all)) if len(uk_all) > 1 else uk_all[0]
        del uk_all
        ub = torch.div(ukey, ncols, rounding_mode="floor")
        wb_ptr[1:] = torch.cumsum(torch.bincount(ub, minlength=nb), 0)
        wlist = (ukey - ub * ncols).to(torch.int32)
        w_all = wb_ptr[1:] - wb_ptr[:-1]
        w_max = int(w_all.max().item())
        big = w_all > 65535  # a 16-bit index cannot address such a window: its codes stay 0, the kernel reads `cols`
        for s0 in range(0, ns, chunk_slices):
            s1 = min(ns, s0 + chunk_slices)
            a, b = int(sp[s0].item()), int(sp[s1].item())
            bs = torch.repeat_interleave(blk_of_slice[s0:s1], slot_per_slice[s0:s1])
            code = torch.searchsorted(ukey, bs * ncols + self.cols[a:b].to(torch.int64)) - wb_ptr[bs]
            code = torch.where(big[bs], torch.zeros_like(code), code)
            wcode[a:b] = (code & 0xffff).to(torch.int32).to(torch.int16)  # two's complement: the kernel reads uint16
        # tile layout of the code stream (ox_sell.wt_ptr): 2 pairs = 4 entries of a lane per 8-byte load
        npair = width // KV
        wt_ptr = torch.zeros(ns + 1, dtype=torch.int64, device=dev)
        wt_ptr[1:] = torch.cumsum((npair + 1) // 2, 0)
        n_tiles = int(wt_ptr[-1].item())
        plain = _lib.ox_sell(self.n_rows, self.n_cols, self.n_slices, 0, self.slice_ptr.data_ptr(), self.cols.data_ptr())
        tiled = torch.empty(n_tiles * 256, dtype=torch.int16, device=dev)
        _lib.check(_lib.load().ox_window_retile(C.byref(plain), _lib.ptr(wt_ptr), _lib.ptr(wcode), 2, _lib.ptr(tiled),
                                                _lib.current_stream()), "ox_window_retile")
        del wcode
        self.wb_slices, self.wb_waves = wb_slices.contiguous(), sched.to(torch.int32).to(torch.int16).contiguous()
        self.wb_ptr, self.wlist, self.wcode, self.wt_ptr = wb_ptr, wlist.contiguous(), tiled, wt_ptr
        self.n_wblocks, self.w_max = nb, w_max
        wq = torch.quantile(w_all.to(torch.float64), torch.tensor([0.5, 0.9, 0.99], dtype=torch.float64, device=dev))
        self.w_stats = {"blocks": nb, "w_mean": float(w_all.to(torch.float64).mean().item()), "w_max": w_max,
                        "w_p50_p90_p99": [int(v) for v in wq.tolist()],
                        "share_over_2176": float((w_all > 2176).to(torch.float64).mean().item()),
                        "list_bytes": int(wlist.numel()) * 4, "over_16bit": int(big.sum().item())}
        return True

    def struct(self, vals: torch.Tensor, compress: bool = True) -> _lib.ox_sell:
        """ox_sell of a value array on this pattern.  On the GPU the pattern's 16-bit column stream
        is built on first use (``ox_sell_compress_cols``) and shared by all its matrices."""
        assert vals.shape[0] % self.size == 0 and vals.dtype == torch.float64
        c16 = cb = None
        if compress and self.device.type == "cuda" and self.size > 0:
            if self.cols16 is None:
                self.cols16 = torch.empty(self.size, dtype=torch.int16, device=self.device)
                self.cbase = torch.empty(2 * (self.size // (SLICE * KV)), dtype=torch.int32, device=self.device)
                plain = _lib.ox_sell(self.n_rows, self.n_cols, self.n_slices, 0, self.slice_ptr.data_ptr(),
                                     self.cols.data_ptr(), None, None, None, None, None)
                n16 = C.c_int64(0)
                _lib.check(_lib.load().ox_sell_compress_cols(C.byref(plain), _lib.ptr(self.cols16),
                                                             _lib.ptr(self.cbase), C.byref(n16),
                                                             _lib.current_stream()), "ox_sell_compress_cols")
                self.frac16 = n16.value / self.size  # share of the stored entries read as 16 bit
            c16, cb = self.cols16.data_ptr(), self.cbase.data_ptr()
        S = _lib.ox_sell(self.n_rows, self.n_cols, self.n_slices, 0, self.slice_ptr.data_ptr(),
                         self.cols.data_ptr(), vals.data_ptr(), c16, cb, None, None)
        if self.ib_slices is not None:
            S.ib_slices, S.n_interior = self.ib_slices.data_ptr(), self.n_interior
        if getattr(self, "wcode", None) is not None:
            S.wb_slices, S.wb_waves, S.wb_ptr = self.wb_slices.data_ptr(), self.wb_waves.data_ptr(), self.wb_ptr.data_ptr()
            S.wlist, S.wcode, S.n_wblocks, S.w_max = self.wlist.data_ptr(), self.wcode.data_ptr(), self.n_wblocks, self.w_max
            S.wt_ptr = self.wt_ptr.data_ptr()
            S.n_wb_interior = int(getattr(self, "n_wb_interior", None) or 0)
        return S

    def bins_args(self):
        return (int(self.bin_width.shape[0]), self.bin_ptr.ctypes.data_as(C.POINTER(C.c_int64)),
                _lib.ptr(self.bin_slices), self.bin_width.ctypes.data_as(C.POINTER(C.c_int32)))

    # ---- host-side conversions (tests / diagnostics, small sizes) ----------------------
    def slot_rows_k(self):
        """For every storage slot: (row, k) -- host numpy."""
        sp = self.slice_ptr.cpu().numpy()
        o = np.arange(self.size, dtype=np.int64)
        s = np.searchsorted(sp, o, side="right") - 1
        loc = o - sp[s]
        k = (loc // (SLICE * KV)) * KV + loc % KV
        lane = (loc % (SLICE * KV)) // KV
        return s * SLICE + lane, k

    def to_csr(self, vals: torch.Tensor):
        import scipy.sparse as sp

        rows, k = self.slot_rows_k()
        rl = np.zeros(self.n_slices * SLICE, dtype=np.int64)
        rl[: self.n_rows] = self.row_len.cpu().numpy()
        ok = k < rl[rows]
        A = sp.coo_matrix((vals.cpu().numpy()[ok], (rows[ok], self.cols.cpu().numpy()[ok])),
                          shape=(self.n_rows, self.n_cols)).tocsr()
        A.sort_indices()
        return A

    def values_from_csr(self, A) -> torch.Tensor:
        """SELL value array holding the entries of scipy CSR ``A`` (same pattern or a subset)."""
        rows, k = self.slot_rows_k()
        cols = self.cols.cpu().numpy()
        rl = np.zeros(self.n_slices * SLICE, dtype=np.int64)
        rl[: self.n_rows] = self.row_len.cpu().numpy()
        ok = k < rl[rows]
        out = np.zeros(self.size)
        A = A.tocsr()
        out[ok] = np.asarray(A[rows[ok], cols[ok]]).ravel()
        return torch.from_numpy(out).to(self.device)


class AdjTable:
    """dof -> cell adjacency in slice order (+ in-row positions of the cell dofs)."""

    def __init__(self, n_slices, nd, adj_ptr, adj_cell, adj_loc, adj_pos, pw):
        self.n_slices, self.nd = n_slices, nd
        self.adj_ptr, self.adj_cell, self.adj_loc = adj_ptr, adj_cell, adj_loc
        self.adj_pos, self.pw = adj_pos, pw

    def struct(self) -> _lib.ox_adj:
        return _lib.ox_adj(self.n_slices, self.nd, self.adj_ptr.data_ptr(), self.adj_cell.data_ptr(),
                           self.adj_loc.data_ptr())


class _Element:
    def __init__(self, degree, gdim):
        self.degree = degree
        self.family = "Lagrange"
        self.gdim = gdim


class _DofMap:
    def __init__(self, V):
        self._V = V
        self.index_map = type("IndexMap", (), {"size_local": V.n_owned, "num_ghosts": V.n_local - V.n_owned,
                                               "size_global": V.num_dofs_global})()
        self.index_map_bs = 1

    def cell_dofs(self, c):
        return self._V.cell_dofs[c].cpu().numpy()


class FunctionSpace:
    """Scalar Lagrange space of degree 1 or 2 on a simplicial mesh.

    With ``part`` (a :class:`oasisx_amd.parallel.MeshPartition`) the space is the rank-local
    piece of a mesh-partitioned space: rows = the dofs this rank owns (SELL order), columns =
    owned dofs followed by the ghost dofs grouped by owner rank, cells = every cell that touches
    an owned dof (own cells + one ghost layer), so that every owned row is assembled locally."""

    def __init__(self, mesh: Mesh, degree: int, window: int = 4096, part=None,
                 block_pairs: int = 1 << 24, block_nnz: int = 1 << 27, brick: bool | None = None):
        if degree not in (1, 2, 3):
            raise ValueError("oasisx_amd supports Lagrange degree 1, 2 and (on triangles) 3")
        if degree == 3 and part is not None:
            if mesh.device.type != "cuda" or _os.environ.get("OX_SETUP", "native") == "torch":
                raise NotImplementedError("Lagrange degree 3 on a mesh partition: built by the library (GPU hosts)")

        # brick order of the numbering (lattice meshes, one GPU): what the LDS-window SpMV needs, a loss for the
        # lane = row kernels -- only together with ``build_windows`` (FractionalStep_AB_CN options["spmv_windows"]);
        # OX_BRICK=1 forces it for tuning runs
        if brick is None:
            brick = _os.environ.get("OX_BRICK", "0") == "1"
        self.brick = bool(brick) and part is None
        self.mesh = mesh
        self.degree = degree
        self.element = _Element(degree, mesh.gdim)
        self.num_sub_spaces = 0
        self.part = part
        self.native = None
        self.window = int(window)
        dev = mesh.device
        if dev.type == "cuda" and _os.environ.get("OX_SETUP", "native") != "torch":
            # the whole set-up runs inside liboasisx_hip.so (csrc/ox_setup.hip, behind the C ABI of
            # include/oasisx_hip.h); this class only wraps the arrays the library owns.  A mesh-partitioned
            # space hands the library this rank's cells and the owner of each of their dofs
            # (ox_mesh_create_sub / ox_space_create_part); the halo plan is then read off the partition.
            self._init_native(window, part)
            return
        if degree == 3:
            raise NotImplementedError("Lagrange degree 3 spaces are built by the library (GPU hosts); the torch twin "
                                      "of the set-up covers degree 1 and 2")
        d = mesh.gdim
        nverts = mesh.num_vertices
        rank = 0 if part is None else part.rank
        # ---- 1. global initial dof ids: vertices, then edges; restricted to the local cells ----
        lo = mesh.coords.min(dim=0).values
        span = (mesh.coords.max(dim=0).values - lo).clamp_min(1e-300)
        cell_ids = torch.arange(mesh.num_cells, device=dev) if part is None else part.local_cells
        tb = default_tile_bits(mesh)
        curve = not mesh_is_lattice(mesh)
        ckey = locality_key(mesh.coords[mesh.cells[cell_ids]].mean(dim=1), lo, span, tb,
                            default_key_bits(mesh.num_cells, mesh.gdim, tb), curve)
        # kernel-side cell order: tiled order of the centroids (index i of every per-cell array = this list's i)
        self.local_cells = cell_ids[torch.argsort(ckey, stable=True)]
        del ckey
        cells = mesh.cells[self.local_cells]
        nc = int(cells.shape[0])
        if degree == 1:
            cd0g = cells
            n_glob = nverts
            self._edge_keys = None
        else:
            if part is None:
                ea = torch.tensor([e[0] for e in local_edges(d)], device=dev)
                eb = torch.tensor([e[1] for e in local_edges(d)], device=dev)
                a, b = cells[:, ea], cells[:, eb]
                key = torch.minimum(a, b) * nverts + torch.maximum(a, b)
                uniq, inv = torch.unique(key.reshape(-1), return_inverse=True)
                cell_edges = inv.reshape(nc, -1)
                del a, b, key, inv
            else:  # the edges of this rank's window (parallel.MeshPartition): ids are positions among ITS keys
                uniq = part.edge_keys
                cell_edges = part.cell_edges_of(self.local_cells)
            self._edge_keys = uniq
            cd0g = torch.cat([cells, nverts + cell_edges], dim=1)
            n_glob = nverts + int(uniq.shape[0])
        nd = int(cd0g.shape[1])
        self.nd = nd
        self.num_dofs_global = n_glob if part is None else part.num_dofs_global(degree)
        if part is None:
            gl = None
            cdL = cd0g
            nL = n_glob
            owned = None
        else:
            gl = torch.unique(cd0g.reshape(-1))  # sorted global ids of the local dofs
            cdL = torch.searchsorted(gl, cd0g.reshape(-1)).reshape(nc, nd)
            nL = int(gl.shape[0])
            ownerL = part.owner0(degree)[gl]
            owned = ownerL == rank
        self._gl = gl
        # dof coordinates in the local initial numbering
        ids = torch.arange(n_glob, device=dev) if gl is None else gl
        isv = ids < nverts
        xL = torch.empty((nL, d), dtype=torch.float64, device=dev)
        xL[isv] = mesh.coords[ids[isv]]
        if degree == 2:
            ek = self._edge_keys[ids[~isv] - nverts]
            xL[~isv] = 0.5 * (mesh.coords[torch.div(ek, nverts, rounding_mode="floor")] + mesh.coords[ek % nverts])
        del isv
        # ---- 2. owned dofs in tiled spatial order, then ghosts by (owner, global id) --------------
        kb = default_key_bits(self.num_dofs_global, mesh.gdim, tb)
        bs = default_brick_shift(self.num_dofs_global, mesh.gdim, kb) if (self.brick and not curve) else 0
        self._key = (lo, span, tb, kb, curve, bs)
        skey = locality_key(xL, lo, span, tb, kb, curve, bs)
        if owned is None:
            perm1 = torch.argsort(skey, stable=True)
            n_owned = nL
            ghost_owner = None
        else:
            io = torch.nonzero(owned).reshape(-1)
            io = io[torch.argsort(skey[io], stable=True)]
            ig = torch.nonzero(~owned).reshape(-1)  # ascending global id already
            ig = ig[torch.argsort(ownerL[ig], stable=True)]
            perm1 = torch.cat([io, ig])
            n_owned = int(io.shape[0])
            ghost_owner = ownerL[ig]
        rank1 = torch.empty_like(perm1)
        rank1[perm1] = torch.arange(nL, device=dev)
        del skey
        self.n_owned, self.n_local = n_owned, nL
        self.num_dofs = nL  # DOLFINx convention: local arrays hold owned dofs, then ghosts
        cd1 = rank1[cdL]
        # ---- 3. pattern of the owned rows, built in ROW BLOCKS -------------------------------------
        # (dof, cell) pairs grouped by dof give, per block of rows, exactly the cells that touch
        # them; the unique (row, col) keys of a block are final because row blocks are disjoint.
        # No global sort/unique over all keys is needed (torch's CUB calls refuse > 2^31 elements;
        # 256^3 P2 has 3.9 G keys) and the peak memory is one block.
        dof = cd1.reshape(-1)
        order = torch.argsort(dof, stable=True)
        dof_s = dof[order]
        keep = dof_s < n_owned
        order, dof_s = order[keep], dof_s[keep]
        cell_s = torch.div(order, nd, rounding_mode="floor")
        del dof, order, keep
        cnt1 = torch.bincount(dof_s, minlength=n_owned)[:n_owned]
        start1 = torch.zeros(n_owned + 1, dtype=torch.int64, device=dev)
        start1[1:] = torch.cumsum(cnt1, 0)
        start1_h = start1.cpu().numpy()
        key_blocks, len_blocks = [], []
        r0 = 0
        while r0 < n_owned:
            r1 = int(np.searchsorted(start1_h, start1_h[r0] + block_pairs, side="right")) - 1
            r1 = min(n_owned, max(r1, r0 + 1))
            a, b = int(start1_h[r0]), int(start1_h[r1])
            kk = torch.unique((dof_s[a:b].unsqueeze(1) * nL + cd1[cell_s[a:b]]).reshape(-1))
            key_blocks.append(kk)
            len_blocks.append(torch.bincount(torch.div(kk, nL, rounding_mode="floor") - r0, minlength=r1 - r0))
            r0 = r1
        keys = torch.cat(key_blocks) if key_blocks else torch.zeros(0, dtype=torch.int64, device=dev)
        len1 = torch.cat(len_blocks) if len_blocks else torch.zeros(0, dtype=torch.int64, device=dev)
        del key_blocks, len_blocks, dof_s, cell_s, cnt1, start1
        # ---- 4. window sort of the owned rows by decreasing length (stable) --------------------
        lmax = int(len1.max().item()) if n_owned else 0
        wkey = (torch.arange(n_owned, device=dev) // window) * (lmax + 1) + (lmax - len1)
        perm2 = torch.argsort(wkey, stable=True)
        rank2 = torch.arange(nL, device=dev)
        rank2[perm2] = torch.arange(n_owned, device=dev)
        del wkey
        rank_f = rank2[rank1]  # local initial dof -> final local dof
        self._rank_initial = rank_f
        self.cell_dofs = rank_f[cdL].to(torch.int32).contiguous()
        xf = torch.empty_like(xL)
        xf[rank_f] = xL
        self.x = xf  # (n_local, gdim) dof coordinates
        self._x3 = None
        self.dofmap = _DofMap(self)
        # ---- 5. final pattern: relabel + sort, in window-aligned row blocks (rows only move inside
        #         their window, so a block's entries stay inside the block's slice of the array) --
        rp1 = torch.zeros(n_owned + 1, dtype=torch.int64, device=dev)
        rp1[1:] = torch.cumsum(len1, 0)
        rp1_h = rp1.cpu().numpy()
        r0 = 0
        while r0 < n_owned:
            r1 = int(np.searchsorted(rp1_h, rp1_h[r0] + block_nnz, side="right")) - 1
            r1 = max(r1, r0 + 1)
            r1 = min(n_owned, ((r1 + window - 1) // window) * window)
            a, b = int(rp1_h[r0]), int(rp1_h[r1])
            seg = keys[a:b]
            row = torch.div(seg, nL, rounding_mode="floor")
            keys[a:b] = torch.sort(rank2[row] * nL + rank2[seg - row * nL]).values
            r0 = r1
        del rp1
        row_len = len1[perm2]  # final row r is the old row perm2[r]
        row_ptr = torch.zeros(n_owned + 1, dtype=torch.int64, device=dev)
        row_ptr[1:] = torch.cumsum(row_len, 0)
        self.pattern = build_sell(n_owned, nL, keys, row_len, row_ptr)
        self._build_adjacency(keys, row_ptr)
        del keys
        # ---- 6. halo plan ---------------------------------------------------------------------------
        self.halo = None
        self.dist = None
        if part is not None:
            self.halo = self._build_halo(part, ghost_owner, cd0g)

    def build_windows(self, split: int = 0) -> bool:
        """LDS-window stream of the space's square pattern (M, K, A share it): see ``SellPattern.build_windows``.
        The rows' positions in the pure locality order are recomputed from the dof coordinates with the key the
        numbering was made with.  ``split`` > 0 (library-built spaces): blocks whose window holds more entries are cut
        in two -- for patterns multiplied with three right-hand sides (``ox_space_windows_split``)."""
        mesh = self.mesh
        if mesh.device.type != "cuda" or self.pattern.size == 0 or getattr(self.pattern, "wcode", None) is not None:
            return getattr(self.pattern, "wcode", None) is not None
        if self.native is not None:  # the library builds and owns the stream (ox_space_windows, csrc/ox_setup.hip)
            from . import native as N

            P, own, dev = self.pattern, self.native.handle, mesh.device
            w = _lib.ox_window_info()
            _lib.check(_lib.load().ox_space_windows_split(own.ptr, int(split), C.byref(w)), "ox_space_windows")
            nb = int(w.n_wblocks)
            if nb == 0:
                return False
            P.wb_slices = N.dev_tensor(w.wb_slices, (nb, 8), torch.int32, own, dev)
            P.wb_waves = N.dev_tensor(w.wb_waves, (nb,), torch.int16, own, dev)
            P.wb_ptr = N.dev_tensor(w.wb_ptr, (nb + 1,), torch.int64, own, dev)
            P.wlist = N.dev_tensor(w.wlist, (int(w.n_list),), torch.int32, own, dev)
            P.wt_ptr = N.dev_tensor(w.wt_ptr, (P.n_slices + 1,), torch.int64, own, dev)
            P.wcode = N.dev_tensor(w.wcode, (int(w.n_tiles) * 256,), torch.int16, own, dev)
            P.n_wblocks, P.w_max = nb, int(w.w_max)
            w_all = (P.wb_ptr[1:] - P.wb_ptr[:-1]).to(torch.float64)
            wq = torch.quantile(w_all, torch.tensor([0.5, 0.9, 0.99], dtype=torch.float64, device=dev))
            P.w_stats = {"blocks": nb, "w_mean": float(w_all.mean().item()), "w_max": P.w_max,
                         "w_p50_p90_p99": [int(v) for v in wq.tolist()],
                         "share_over_2176": float((w_all > 2176).to(torch.float64).mean().item()),
                         "list_bytes": int(w.n_list) * 4, "over_16bit": int(w.n_over_16bit)}
            P.order_window_blocks()  # (a mesh-partitioned operator whose interior / boundary split is known already)
            return True
        lo = mesh.coords.min(dim=0).values
        span = (mesh.coords.max(dim=0).values - lo).clamp_min(1e-300)
        tb = self.native.nmesh.tile_bits if self.native is not None else default_tile_bits(mesh)
        curve = not mesh_is_lattice(mesh)
        kb = default_key_bits(self.num_dofs_global, mesh.gdim, tb)
        bs = default_brick_shift(self.num_dofs_global, mesh.gdim, kb) if (self.brick and not curve) else 0
        key = locality_key(self.x[: self.n_owned], lo, span, tb, kb, curve, bs)
        order = torch.argsort(key, stable=True)
        row_pos = torch.empty_like(order)
        row_pos[order] = torch.arange(order.shape[0], device=order.device)
        ok = self.pattern.build_windows(row_pos, self.window)
        self.pattern.order_window_blocks()
        return ok

    def _init_native(self, window: int, part=None):
        from . import native as N

        mesh, dev = self.mesh, self.mesh.device
        d = mesh.gdim
        if part is None:
            ns = N.NativeSpace(mesh, self.degree, window, brick=self.brick)
            nc = mesh.num_cells
        else:
            # initial dof ids of this rank's part: its vertices (ascending global id), then its edges (ascending key)
            sub = N.NativeSubMesh.of(mesh, part)
            nverts = mesh.num_vertices
            if self.degree == 2:
                uniq_e = torch.unique(part.cell_edges_of(sub.cells_global).reshape(-1))  # window edge indices, ascending
                gl = torch.cat([sub.verts, nverts + uniq_e])
            elif self.degree == 3:
                # the library's initial ids of the part: its vertices, two dofs per edge (ascending key), then one per face
                # (ascending key: tetrahedra) or per cell (KERNEL cell order: triangles); here in the window numbering of
                # parallel.MeshPartition.owner0(3)
                uniq_e = torch.unique(part.cell_edges_of(sub.cells_global).reshape(-1))
                ne_w = int(part.edge_keys.shape[0])
                e2 = (nverts + 2 * uniq_e).repeat_interleave(2) + torch.arange(2, device=dev).repeat(int(uniq_e.shape[0]))
                if d == 3:
                    if part.face_keys is None:
                        raise ValueError("a degree-3 space on a partitioned tetrahedral mesh needs MeshPartition(faces=True)")
                    last = nverts + 2 * ne_w + torch.unique(part.cell_faces_of(sub.cells_global).reshape(-1))
                else:
                    if not part.cells_own_dofs:
                        raise ValueError("a degree-3 space on a partitioned triangular mesh needs MeshPartition(faces=True)")
                    last = nverts + 2 * ne_w + part._win_pos(sub.cells_global)  # (the part's cells: ascending global id)
                gl = torch.cat([sub.verts, e2, last])
            else:
                gl = sub.verts
            owner = part.owner0(self.degree)[gl].to(torch.int32).contiguous()
            self.num_dofs_global = part.num_dofs_global(self.degree)
            ns = N.NativeSpace(mesh, self.degree, window, part=part, owner=owner, n_dofs_whole=self.num_dofs_global)
            nc = int(sub.cells_global.shape[0])
        self.native = ns
        v, own = ns.info, ns.handle
        n = int(v.n_dofs)
        self.nd = int(v.nd)
        self._rank_initial = N.dev_tensor(v.rank_initial, (n,), torch.int32, own, dev)
        if part is None:
            self.local_cells = ns.nmesh.cell_perm.to(torch.int64)
            self.num_dofs_global = self.n_owned = self.n_local = self.num_dofs = n
            self._gl = None
            self._edge_keys = (N.dev_tensor(v.edge_keys, (int(v.n_edges),), torch.int64, own, dev)
                               if self.degree >= 2 else None)
            self._face_keys = (N.dev_tensor(v.face_keys, (int(v.n_faces),), torch.int64, own, dev)
                               if (self.degree == 3 and d == 3) else None)
        else:
            self.local_cells = sub.cells_global[ns.nmesh.cell_perm.to(torch.int64)]  # global cell ids, kernel order
            self.n_owned, self.n_local = int(v.pattern.sell.n_rows), n
            self.num_dofs = n  # DOLFINx convention: local arrays hold owned dofs, then ghosts
            self._gl = gl
            self._gl_order = None
            if gl.numel() > 1 and not bool((gl[1:] > gl[:-1]).all()):  # (degree 3 on triangles: cells in kernel order)
                self._gl_order = torch.argsort(gl)
                self._gl = gl[self._gl_order]
            self._edge_keys = part.edge_keys if self.degree >= 2 else None
            self._face_keys = part.face_keys if (self.degree == 3 and d == 3) else None
        self.cell_dofs = N.dev_tensor(v.cell_dofs, (nc, self.nd), torch.int32, own, dev)
        self.x = N.dev_tensor(v.x, (n, d), torch.float64, own, dev)
        self._x3 = None
        self.dofmap = _DofMap(self)
        self.pattern = N.pattern_from_info(v.pattern, own, dev, SellPattern)
        npairs, pw = int(v.n_pairs), int(v.pw)
        nsl = int(v.adj.n_slices)
        self.adj = AdjTable(nsl, self.nd, N.dev_tensor(v.adj.adj_ptr, (nsl + 1,), torch.int64, own, dev),
                            N.dev_tensor(v.adj.adj_cell, (npairs,), torch.int32, own, dev),
                            N.dev_tensor(v.adj.adj_loc, (npairs,), torch.uint8, own, dev),
                            N.dev_tensor(v.adj_pos, (npairs, pw), torch.uint8, own, dev), pw)
        start = N.dev_tensor(v.pair_start, (n + 1,), torch.int64, own, dev)
        self.adj_count = start[1:] - start[:-1]
        self.halo = None
        self.dist = None
        if part is not None:
            owner_f = torch.empty(n, dtype=torch.int64, device=dev)
            owner_f[self._rank_initial.to(torch.int64)] = owner.to(torch.int64)
            self.halo = self._build_halo(part, owner_f[self.n_owned:], None)

    # ---------------------------------------------------------------------------------
    def global_to_local(self, gids: torch.Tensor) -> torch.Tensor:
        """Final local dof of global initial dof ids (-1 where the dof is not local)."""
        if self._gl is None:
            return self._rank_initial[gids]
        pos = torch.searchsorted(self._gl, gids).clamp_max(self._gl.shape[0] - 1)
        ok = self._gl[pos] == gids
        if getattr(self, "_gl_order", None) is not None:  # (_gl sorted; the library's initial id is the ORIGINAL position)
            pos = self._gl_order[pos]
        return torch.where(ok, self._rank_initial[pos], torch.full_like(pos, -1))

    def _build_halo(self, part, ghost_owner, cd0g):
        """Who sends what: ghosts are ordered by (owner, initial id) on the receiver, and the owner lists
        exactly those dofs in the same order (vertices by id, then edges by key) -- each side computes its list
        from its own window of the mesh (parallel.MeshPartition), no communication at set-up."""
        dev = self.mesh.device
        nverts = self.mesh.num_vertices
        owner0 = part.owner0(self.degree)
        peers, send_lists, recv_counts = [], [], []
        cand = set(part.peers()) | set(int(q) for q in torch.unique(ghost_owner).tolist())
        for qr in sorted(cand):
            if qr == part.rank:
                continue
            nrecv = int((ghost_owner == qr).sum().item())
            # the cells both ranks keep: qr's ghosts that this rank owns are dofs of exactly these cells
            cm = part.cells_shared_with(qr)
            cq = self.mesh.cells[cm]
            if self.degree == 1:
                ids = cq
            elif self.degree == 2:
                ids = torch.cat([cq, nverts + part.cell_edges_of(cm)], dim=1)
            else:  # degree 3, window numbering of owner0(3): two dofs per edge, then faces (tetrahedra) / cells (triangles)
                ce = part.cell_edges_of(cm)
                ne_w = int(part.edge_keys.shape[0])
                last = (part.cell_faces_of(cm) if self.mesh.gdim == 3 else part._win_pos(cm).unsqueeze(1))
                ids = torch.cat([cq, nverts + 2 * ce, nverts + 2 * ce + 1, nverts + 2 * ne_w + last], dim=1)
            ids = torch.unique(ids.reshape(-1))
            ids = ids[owner0[ids] == part.rank]  # ascending id (vertices, then edges by key) = the receiver's order
            if nrecv == 0 and ids.shape[0] == 0:
                continue
            loc = self.global_to_local(ids)
            if loc.numel():
                assert int(loc.min().item()) >= 0 and int(loc.max().item()) < self.n_owned
            peers.append(qr)
            send_lists.append(loc.to(torch.int32))
            recv_counts.append(nrecv)
        send_off = np.zeros(len(peers) + 1, dtype=np.int64)
        recv_off = np.zeros(len(peers) + 1, dtype=np.int64)
        for i in range(len(peers)):
            send_off[i + 1] = send_off[i] + send_lists[i].shape[0]
            recv_off[i + 1] = recv_off[i] + recv_counts[i]
        assert recv_off[-1] == self.n_local - self.n_owned
        send_idx = (torch.cat(send_lists) if send_lists else torch.zeros(0, dtype=torch.int32, device=dev)).contiguous()
        return {"peers": np.asarray(peers, dtype=np.int32), "send_off": send_off, "recv_off": recv_off,
                "send_idx": send_idx}

    def attach_comm(self, comm):
        """Create the device halo plan (ox_dist) of this space and give it a transport according to
        ``comm.transport``: the direct xGMI windows (self-tested, collectively agreed), the RCCL
        communicator, or the host-staged rehearsal transport (see oasisx_amd.parallel)."""
        if self.halo is None or comm is None:
            return
        want = getattr(comm, "transport", "auto")
        on_gpu = self.mesh.device.type == "cuda"
        if not on_gpu:
            return  # CPU: partition logic only, no device plan
        if comm.handle is None and getattr(comm, "enable_p2p", None) is None:
            return  # a bare rank/size object (tests of rank-local data): nothing to exchange with
        lib = _lib.load()
        h = self.halo

        def host_transport():
            self.dist = comm.make_transport(self)
            self._apply_plan_options()
            self.pattern.dist = self.dist
            self.pattern.split_interior(self.n_owned)
            if not getattr(comm, "self_loop", False):  # (parallel.SelfLoopComm: a plan folded onto one rank, timing only)
                self.check_halo()
            comm.active[self.degree] = "self-loop" if getattr(comm, "self_loop", False) else "host"

        if want == "host" or (want == "rccl" and comm.handle is None):
            if getattr(comm, "make_transport", None) is not None:
                host_transport()
            return
        out = C.c_void_p()
        _lib.check(lib.ox_dist_create(comm.handle, comm.rank, comm.size, int(h["peers"].shape[0]),
                                      h["peers"].ctypes.data_as(C.POINTER(C.c_int32)),
                                      h["send_off"].ctypes.data_as(C.POINTER(C.c_int64)), _lib.ptr(h["send_idx"]),
                                      h["recv_off"].ctypes.data_as(C.POINTER(C.c_int64)), self.n_owned,
                                      self.n_local - self.n_owned, C.byref(out)), "ox_dist_create")
        if want in ("auto", "p2p") and comm.enable_p2p(self, out):
            self.dist = out
            self._apply_plan_options(comm_is_p2p=True)
            self.pattern.dist = out
            self.pattern.split_interior(self.n_owned)
            comm.active[self.degree] = "p2p"
            return
        if want == "p2p":
            raise RuntimeError("OX_TRANSPORT=p2p: the xGMI transport could not be enabled: "
                               + getattr(comm, "p2p_error", "a peer rank failed"))
        if comm.handle is None:  # no RCCL communicator (gloo job): rehearsal transport
            lib.ox_dist_destroy(out)
            self.dist = None
            host_transport()
            return
        self.dist = out
        self._apply_plan_options()
        self.pattern.dist = out
        self.pattern.split_interior(self.n_owned)
        self.check_halo()
        comm.active[self.degree] = "rccl"

    def _apply_plan_options(self, comm_is_p2p: bool = False):
        """Schedule knobs of the halo plan from the environment, read by the host layer and set through the ABI
        (``OX_HALO_OVERLAP=0|1``: exchange-then-multiply / overlapped mat-vecs; unset: the transport's default)."""
        v = _os.environ.get("OX_HALO_OVERLAP")
        if v is not None and self.dist is not None:
            _lib.check(_lib.load().ox_dist_set_overlap(self.dist, 1 if v not in ("0", "false", "") else 0), "ox_dist_set_overlap")
        # OX_P2P_RELEASE=fast: the round-5 release protocol of the xGMI windows (one fence per launch, relaxed flags);
        # unset / "conservative": per-wave system fences and release-scope flags -- the library's default until the
        # windows have crossed a link (include/oasisx_hip.h, ox_dist_set_p2p_release)
        v = _os.environ.get("OX_P2P_RELEASE")
        if v is not None and self.dist is not None and comm_is_p2p:
            _lib.check(_lib.load().ox_dist_set_p2p_release(self.dist, 0 if v.lower() == "fast" else 1), "ox_dist_set_p2p_release")

    def check_halo(self):
        """Self-test of the attached halo plan + transport: exchange the dof coordinates and require
        every ghost entry to receive exactly its own coordinate (run once per space at set-up; a
        wrong plan or a broken transport fails here, loudly, not as wrong physics later)."""
        if self.dist is None:
            return
        lib = _lib.load()
        d = self.mesh.gdim
        X = self.x.clone().contiguous()
        X[self.n_owned:] = float("nan")
        _lib.check(lib.ox_halo_forward(self.dist, _lib.ptr(X), d, _lib.current_stream()), "ox_halo_forward")
        if self.mesh.device.type == "cuda":
            torch.cuda.synchronize()
        if not torch.equal(X, self.x):
            bad = int((X != self.x).any(dim=1).sum().item())
            raise RuntimeError(f"halo exchange self-test failed: {bad} of {self.n_local - self.n_owned} ghost "
                               f"dofs of the P{self.degree} space received a wrong value")

    # ---------------------------------------------------------------------------------
    def _build_adjacency(self, keys_sorted, row_ptr, chunk_pairs: int = 1 << 24):
        dev = self.mesh.device
        nd, n_owned, nL = self.nd, self.n_owned, self.n_local
        n_slices = (n_owned + SLICE - 1) // SLICE
        dof = self.cell_dofs.reshape(-1).to(torch.int64)
        order = torch.argsort(dof, stable=True)  # pairs grouped by dof, cells ascending
        dof_s = dof[order]
        keep = dof_s < n_owned  # rows exist for owned dofs only
        order, dof_s = order[keep], dof_s[keep]
        cell_s = torch.div(order, nd, rounding_mode="floor")
        loc_s = order - cell_s * nd
        cnt = torch.bincount(dof_s, minlength=n_owned)
        start = torch.zeros(n_owned + 1, dtype=torch.int64, device=dev)
        start[1:] = torch.cumsum(cnt, 0)
        t = torch.arange(dof_s.shape[0], device=dev) - start[dof_s]
        cpad = torch.zeros(n_slices * SLICE, dtype=torch.int64, device=dev)
        cpad[:n_owned] = cnt
        T = cpad.reshape(n_slices, SLICE).max(dim=1).values
        adj_ptr = torch.zeros(n_slices + 1, dtype=torch.int64, device=dev)
        adj_ptr[1:] = torch.cumsum(T * SLICE, 0)
        npairs = int(adj_ptr[-1].item())
        off = adj_ptr[dof_s // SLICE] + t * SLICE + (dof_s % SLICE)
        adj_cell = torch.full((npairs,), -1, dtype=torch.int32, device=dev)
        adj_cell[off] = cell_s.to(torch.int32)
        adj_loc = torch.zeros(npairs, dtype=torch.uint8, device=dev)
        adj_loc[off] = loc_s.to(torch.uint8)
        pw = 4 if nd <= 4 else (8 if nd <= 8 else 16)
        if int(self.pattern.widths.max()) > 255:
            raise ValueError("row longer than 255 entries: position bytes overflow")
        adj_pos = torch.zeros((npairs, pw), dtype=torch.uint8, device=dev)
        P = dof_s.shape[0]
        for p0 in range(0, P, chunk_pairs):
            sl = slice(p0, min(P, p0 + chunk_pairs))
            r = dof_s[sl]
            cd = self.cell_dofs[cell_s[sl]].to(torch.int64)  # (m, nd)
            g = torch.searchsorted(keys_sorted, (r.unsqueeze(1) * nL + cd).reshape(-1))
            k = g.reshape(-1, nd) - row_ptr[r].unsqueeze(1)
            adj_pos[off[sl], :nd] = k.to(torch.uint8)
        self.adj = AdjTable(n_slices, nd, adj_ptr, adj_cell, adj_loc, adj_pos, pw)
        self.adj_count = cnt

    def cells_in_kernel_order(self) -> np.ndarray:
        """Vertex ids of the cells in the order every per-cell device array uses (host copy)."""
        return self.mesh.cells[self.local_cells].cpu().numpy()

    def kernel_cell_index(self, cell_ids) -> np.ndarray:
        """Position of mesh cell ids in the kernel-side cell list (-1 if the cell is not local)."""
        lc = self.local_cells.cpu().numpy()
        srt = np.argsort(lc)
        cell_ids = np.asarray(cell_ids)
        pos = np.searchsorted(lc[srt], cell_ids)
        pos = np.minimum(pos, lc.shape[0] - 1)
        return np.where(lc[srt][pos] == cell_ids, srt[pos], -1)

    # ---- DOLFINx-shaped helpers -------------------------------------------------------
    def tabulate_dof_coordinates(self) -> np.ndarray:
        if self._x3 is None:
            x = np.zeros((self.num_dofs, 3))
            x[:, : self.mesh.gdim] = self.x.cpu().numpy()
            self._x3 = x
        return self._x3

    def entity_dofs(self, dim: int, entities) -> np.ndarray:
        """Local dofs on the closure of mesh entities (vertices + edges for P2)."""
        mesh = self.mesh
        dev = mesh.device
        if len(entities) == 0:  # (test_bcs.py's dim = 2 case: no cell lies in the marked line)
            return np.zeros(0, dtype=np.int32)
        ev, _ = mesh._entities(dim)
        verts = ev[np.asarray(entities, dtype=np.int64)].reshape(len(entities), -1)
        gids = [torch.from_numpy(verts.ravel().astype(np.int64)).to(dev)]
        if self.degree >= 2 and verts.shape[1] >= 2:
            ek = self._edge_keys
            nv = mesh.num_vertices
            for a, b in itertools.combinations(range(verts.shape[1]), 2):
                lo = torch.from_numpy(np.minimum(verts[:, a], verts[:, b]).astype(np.int64)).to(dev)
                hi = torch.from_numpy(np.maximum(verts[:, a], verts[:, b]).astype(np.int64)).to(dev)
                key = lo * nv + hi
                pos = torch.searchsorted(ek, key).clamp_max(max(int(ek.shape[0]) - 1, 0))
                hit = ek[pos] == key  # (a partitioned space knows the edges of its window only)
                if self.degree == 2:
                    gids.append((nv + pos)[hit])
                else:  # degree 3: two dofs per edge
                    gids.append((nv + 2 * pos)[hit])
                    gids.append((nv + 2 * pos + 1)[hit])
        if self.degree == 3 and mesh.gdim == 3:
            # tetrahedra: one dof per face, initial id = nv + 2 n_edges + face id (faces by ascending sorted vertex triple)
            if verts.shape[1] >= 3:
                fk, nv = self._face_keys, mesh.num_vertices
                for combo in itertools.combinations(range(verts.shape[1]), 3):
                    tri = np.sort(verts[:, list(combo)], axis=1).astype(np.int64)
                    key = torch.from_numpy((tri[:, 0] * nv + tri[:, 1]) * nv + tri[:, 2]).to(dev)
                    pos = torch.searchsorted(fk, key).clamp_max(max(int(fk.shape[0]) - 1, 0))
                    hit = fk[pos] == key
                    gids.append((nv + 2 * int(self._edge_keys.shape[0]) + pos)[hit])
        elif self.degree == 3 and dim == mesh.gdim:
            # the cells' own dofs: initial id = nv + 2 n_edges + the cell's index as the library was handed it (its global
            # id on one GPU, its position in the rank's window -- ascending global id -- on a partition)
            ce = torch.from_numpy(np.asarray(entities, dtype=np.int64)).to(dev)
            if self.part is not None:
                ce = self.part._win_pos(ce)
            gids.append(mesh.num_vertices + 2 * int(self._edge_keys.shape[0]) + ce)
        loc = self.global_to_local(torch.cat(gids))
        loc = loc[loc >= 0]
        return np.unique(loc.cpu().numpy()).astype(np.int32)


def build_sell(n_rows, n_cols, keys, row_len, row_ptr, block: int = 1 << 27) -> SellPattern:
    """SELL-64 layout from the sorted (row * n_cols + col) keys of a pattern (device tensors).
    Built in blocks of slices / rows so that no temporary exceeds ``block`` elements."""
    dev = keys.device
    n_slices = (n_rows + SLICE - 1) // SLICE
    lpad = torch.zeros(n_slices * SLICE, dtype=torch.int64, device=dev)
    lpad[:n_rows] = row_len
    width = lpad.reshape(n_slices, SLICE).max(dim=1).values
    width = ((width + KV - 1) // KV) * KV
    width = torch.clamp(width, min=KV)
    slice_ptr = torch.zeros(n_slices + 1, dtype=torch.int64, device=dev)
    slice_ptr[1:] = torch.cumsum(width * SLICE, 0)
    sp_h = slice_ptr.cpu().numpy()
    size = int(sp_h[-1])
    cols = torch.empty(size, dtype=torch.int32, device=dev)
    # padding slots repeat the row's own index (value 0); rows past n_rows point at column 0
    nlim = min(n_rows, n_cols)
    s0 = 0
    while s0 < n_slices:
        s1 = int(np.searchsorted(sp_h, sp_h[s0] + block, side="right")) - 1
        s1 = min(n_slices, max(s1, s0 + 1))
        a, b = int(sp_h[s0]), int(sp_h[s1])
        o = torch.arange(a, b, device=dev)
        sl = torch.searchsorted(slice_ptr[s0:s1 + 1], o, right=True) - 1 + s0
        own = sl * SLICE + ((o - slice_ptr[sl]) % (SLICE * KV)) // KV
        cols[a:b] = torch.where(own < nlim, own, torch.zeros_like(own)).to(torch.int32)
        s0 = s1
    rp_h = row_ptr.cpu().numpy()
    r0 = 0
    while r0 < n_rows:
        r1 = int(np.searchsorted(rp_h, rp_h[r0] + block, side="right")) - 1
        r1 = min(n_rows, max(r1, r0 + 1))
        a, b = int(rp_h[r0]), int(rp_h[r1])
        seg = keys[a:b]
        rowf = torch.div(seg, n_cols, rounding_mode="floor")
        k = torch.arange(a, b, device=dev) - row_ptr[rowf]
        off = slice_ptr[rowf // SLICE] + (k // KV) * (SLICE * KV) + (rowf % SLICE) * KV + (k % KV)
        cols[off] = (seg - rowf * n_cols).to(torch.int32)
        r0 = r1
    return SellPattern(n_rows, n_cols, slice_ptr, cols, row_len.to(torch.int32),
                       width.cpu().numpy().astype(np.int32))


def build_rect_pattern(R: "FunctionSpace", Cs: "FunctionSpace", block_pairs: int = 1 << 24):
    """SELL-64 pattern of a rectangular operator (rows = owned dofs of R, columns = local dofs of
    Cs) on the same mesh / partition, plus the position bytes [pairs of R.adj][pw] that give, per
    (row, cell) pair, the in-row index of every Cs dof of the cell (reference fracstep.py:315,336,352:
    ``create_matrix`` of the mixed forms)."""
    dev = R.mesh.device
    assert R.mesh is Cs.mesh and R.local_cells.shape == Cs.local_cells.shape
    if getattr(R, "native", None) is not None and getattr(Cs, "native", None) is not None:
        from . import native as N

        nr = N.NativeRect(R.native, Cs.native)
        pattern = N.pattern_from_info(nr.info.pattern, nr.handle, dev, SellPattern)
        pw = int(nr.info.pw)
        pos = N.dev_tensor(nr.info.pos, (int(R.native.info.n_pairs), pw), torch.uint8, nr.handle, dev)
        pattern.dist = Cs.dist
        return pattern, pos, pw
    n_rows, n_cols, nd_r, nd_c = R.n_owned, Cs.n_local, R.nd, Cs.nd
    dof = R.cell_dofs.reshape(-1).to(torch.int64)
    order = torch.argsort(dof, stable=True)
    dof_s = dof[order]
    keep = dof_s < n_rows
    order, dof_s = order[keep], dof_s[keep]
    cell_s = torch.div(order, nd_r, rounding_mode="floor")
    del dof, order, keep
    cnt = torch.bincount(dof_s, minlength=n_rows)[:n_rows]
    start = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
    start[1:] = torch.cumsum(cnt, 0)
    start_h = start.cpu().numpy()
    cdofs = Cs.cell_dofs.to(torch.int64)
    key_blocks, len_blocks = [], []
    r0 = 0
    while r0 < n_rows:
        r1 = int(np.searchsorted(start_h, start_h[r0] + block_pairs, side="right")) - 1
        r1 = min(n_rows, max(r1, r0 + 1))
        a, b = int(start_h[r0]), int(start_h[r1])
        kk = torch.unique((dof_s[a:b].unsqueeze(1) * n_cols + cdofs[cell_s[a:b]]).reshape(-1))
        key_blocks.append(kk)
        len_blocks.append(torch.bincount(torch.div(kk, n_cols, rounding_mode="floor") - r0, minlength=r1 - r0))
        r0 = r1
    keys = torch.cat(key_blocks)
    row_len = torch.cat(len_blocks)
    del key_blocks, len_blocks
    row_ptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
    row_ptr[1:] = torch.cumsum(row_len, 0)
    pattern = build_sell(n_rows, n_cols, keys, row_len, row_ptr)
    if int(pattern.widths.max()) > 255:
        raise ValueError("row longer than 255 entries: position bytes overflow")
    # positions, in the (padded) order of R's adjacency table
    adj = R.adj
    pw = 4 if nd_c <= 4 else (8 if nd_c <= 8 else 16)
    npairs = int(adj.adj_cell.shape[0])
    pos = torch.zeros((npairs, pw), dtype=torch.uint8, device=dev)
    ap_h = adj.adj_ptr.cpu().numpy()
    s0 = 0
    while s0 < adj.n_slices:
        s1 = int(np.searchsorted(ap_h, ap_h[s0] + block_pairs, side="right")) - 1
        s1 = min(adj.n_slices, max(s1, s0 + 1))
        a, b = int(ap_h[s0]), int(ap_h[s1])
        pidx = torch.arange(a, b, device=dev)
        cell = adj.adj_cell[a:b].to(torch.int64)
        ok = cell >= 0
        sl = torch.searchsorted(adj.adj_ptr[s0:s1 + 1], pidx, right=True) - 1 + s0
        row = sl * SLICE + (pidx - adj.adj_ptr[sl]) % SLICE
        pidx, cell, row = pidx[ok], cell[ok], row[ok]
        g = torch.searchsorted(keys, (row.unsqueeze(1) * n_cols + cdofs[cell]).reshape(-1))
        pos[pidx, :nd_c] = (g.reshape(-1, nd_c) - row_ptr[row].unsqueeze(1)).to(torch.uint8)
        s0 = s1
    pattern.dist = Cs.dist
    return pattern, pos, pw


def functionspace(mesh: Mesh, element, **kwargs):
    """``dolfinx.fem.functionspace(mesh, ("Lagrange", k))`` (also with a shape tuple)."""
    if isinstance(element, FunctionSpace):
        return element
    family, degree = element[0], int(element[1])
    if str(family).lower() in ("dg", "discontinuous lagrange", "dp"):
        return DGSpace(mesh, degree, shape=element[2] if len(element) > 2 else None)
    if str(family).lower() not in ("lagrange", "p", "cg"):
        raise ValueError(f"unsupported element family {family!r}")
    if degree > 2:  # boundary conditions and interpolation only: no operators are assembled on these
        if len(element) > 2 and element[2]:
            raise NotImplementedError("vector-valued Lagrange spaces of degree > 2")
        return HighOrderLagrangeSpace(mesh, degree)
    V = FunctionSpace(mesh, degree, **kwargs)
    if len(element) > 2 and element[2]:
        return VectorFunctionSpace(V, int(element[2][0]))
    return V


class DGSpace:
    """Discontinuous Lagrange space of degree 1, scalar or vector valued -- what
    ``basix.ufl.element("DG", cell, 1, shape=(k,))`` gives the reference's ``test_projector.py:26-27``.
    Dof (cell e, vertex a) at index e * (gdim+1) + a, ``dim`` components interleaved; cells in the kernel cell
    order of the mesh (the order every continuous space on it uses: ``local_cells``)."""

    is_dg = True

    def __init__(self, mesh: Mesh, degree: int = 1, shape=None):
        if degree != 1:
            raise NotImplementedError("DGSpace: degree 1")
        self.mesh, self.degree = mesh, 1
        self.dim = int(shape[0]) if shape else 1
        self.num_sub_spaces = self.dim if shape else 0
        dev = mesh.device
        if dev.type == "cuda" and _os.environ.get("OX_SETUP", "native") != "torch":
            from . import native as N

            nm = N.NativeMesh.of(mesh)
            self.local_cells, self.geom, self._owner = nm.cell_perm.to(torch.int64), nm.geom, nm
        else:
            lo = mesh.coords.min(dim=0).values
            span = (mesh.coords.max(dim=0).values - lo).clamp_min(1e-300)
            tb = default_tile_bits(mesh)
            ckey = locality_key(mesh.coords[mesh.cells].mean(dim=1), lo, span, tb,
                                default_key_bits(mesh.num_cells, mesh.gdim, tb), not mesh_is_lattice(mesh))
            self.local_cells = torch.argsort(ckey, stable=True)
            self.geom = cell_geometry(mesh, self.local_cells)
        self.nd = mesh.gdim + 1
        self.num_dofs = self.n_local = self.n_owned = int(self.local_cells.shape[0]) * self.nd
        self.element = type("Element", (), {"degree": 1, "family": "DG"})()

    def tabulate_dof_coordinates(self) -> np.ndarray:
        """(num_dofs, 3): the vertices of every cell, cell by cell."""
        x = self.mesh.coords[self.mesh.cells[self.local_cells]].reshape(-1, self.mesh.gdim).cpu().numpy()
        out = np.zeros((x.shape[0], 3))
        out[:, : self.mesh.gdim] = x
        return out


class HighOrderLagrangeSpace:
    """Scalar Lagrange space of degree >= 3 on a simplicial mesh, for what the reference's ``DirichletBC`` needs of
    it (test/test_bcs.py:19-160 runs P1-P4): a dof numbering, dof coordinates, the dofs on the closure of mesh
    entities, nodal interpolation and vectors.  No operator of the time step is assembled on it (the solver's
    spaces are P1 / P2; ``FractionalStep_AB_CN`` says so).

    Dofs are numbered entity by entity -- vertices, then the interior points of the edges, of the triangles and of
    the tetrahedra -- with the points of an entity ordered by their barycentric multi-index relative to the entity's
    vertices in ascending global id (orientation-independent).  Points are EQUISPACED; the reference asks Basix for
    the ``gll_warped`` variant (fracstep.py:170,181), which moves the interior points from degree 3 on -- values
    imposed by a ``DirichletBC`` are the boundary function at the space's own points either way."""

    def __init__(self, mesh: Mesh, degree: int):
        if degree < 3:
            raise ValueError("HighOrderLagrangeSpace: degree >= 3 (use FunctionSpace for 1 and 2)")
        self.mesh, self.degree = mesh, int(degree)
        self.num_sub_spaces = 0
        self.element = _Element(degree, mesh.gdim)
        p, d = self.degree, mesh.gdim
        xv = mesh.coords.cpu().numpy()
        coords, self._first, self._multi = [], {}, {}
        n = 0
        for k in range(d + 1):
            ev, _ = mesh._entities(k)  # (ne, k+1) sorted vertex ids
            mi = [m for m in itertools.product(range(1, p), repeat=k + 1) if sum(m) == p] if k > 0 else [(p,)]
            self._first[k], self._multi[k] = n, len(mi)
            if not mi:
                continue
            w = np.asarray(mi, dtype=np.float64) / p  # (npts, k+1) barycentric weights
            pts = np.einsum("pa,eak->epk", w, xv[ev])  # (ne, npts, d)
            coords.append(pts.reshape(-1, d))
            n += ev.shape[0] * len(mi)
        self.num_dofs = self.n_local = self.n_owned = self.num_dofs_global = n
        x = np.concatenate(coords, axis=0)
        self.x = torch.from_numpy(x).to(mesh.device)
        self._x3 = np.zeros((n, 3))
        self._x3[:, :d] = x
        self.dofmap = None

    def tabulate_dof_coordinates(self) -> np.ndarray:
        return self._x3

    def entity_dofs(self, dim: int, entities) -> np.ndarray:
        """Dofs on the closure of mesh entities: the entity's own interior points and those of all its
        sub-entities (what ``locate_dofs_topological`` returns)."""
        mesh = self.mesh
        if len(entities) == 0:
            return np.zeros(0, dtype=np.int32)
        ev, _ = mesh._entities(dim)
        verts = ev[np.asarray(entities, dtype=np.int64)].reshape(len(entities), -1)
        out = []
        nv = mesh.num_vertices
        for k in range(dim + 1):
            if self._multi[k] == 0:
                continue
            evk, _ = mesh._entities(k)
            key_all = np.zeros(evk.shape[0], dtype=np.int64)
            for c in range(k + 1):
                key_all = key_all * np.int64(nv) + evk[:, c]
            order = np.argsort(key_all)
            for combo in itertools.combinations(range(verts.shape[1]), k + 1):
                sub = np.sort(verts[:, list(combo)], axis=1)
                key = np.zeros(sub.shape[0], dtype=np.int64)
                for c in range(k + 1):
                    key = key * np.int64(nv) + sub[:, c]
                idx = order[np.searchsorted(key_all[order], key)]
                base = self._first[k] + idx * self._multi[k]
                out.append((base[:, None] + np.arange(self._multi[k])[None, :]).ravel())
        return np.unique(np.concatenate(out)).astype(np.int32) if out else np.zeros(0, dtype=np.int32)


class VectorFunctionSpace:
    """Blocked space (bs = dim) over a scalar space: dof (i, c) at index i*dim + c."""

    def __init__(self, Vi: FunctionSpace, dim: int):
        self.scalar = Vi
        self.mesh = Vi.mesh
        self.dim = dim
        self.num_sub_spaces = dim
        self.num_dofs = Vi.num_dofs
        self.element = Vi.element

    def sub(self, i):
        return _SubSpace(self, i)

    def tabulate_dof_coordinates(self):
        return self.scalar.tabulate_dof_coordinates()


class _SubSpace:
    def __init__(self, V, i):
        self._V, self._i = V, i

    def collapse(self):
        n, dim = self._V.num_dofs, self._V.dim
        return self._V.scalar, (np.arange(n, dtype=np.int32) * dim + self._i)


# ---- fields ----------------------------------------------------------------------------------


class FieldStorage:
    """(n_alloc, nc) float64 block on the device with a lazily synchronised host mirror.

    The device copy is authoritative.  ``host()`` checks the block out: it is copied to the
    host and the returned numpy array may be read and written; the next device use copies
    it back.  (On a CPU-only machine both views share memory.)"""

    def __init__(self, n: int, nc: int, device, n_alloc: int | None = None):
        self.n, self.nc = int(n), int(nc)
        self.n_alloc = int(n if n_alloc is None else n_alloc)
        self._dev = torch.zeros((self.n_alloc, self.nc), dtype=torch.float64, device=device)
        self._shared = self._dev.device.type == "cpu"
        self._host = self._dev.numpy()[: self.n] if self._shared else None
        self._out = False
        # bumped whenever the block MAY have been written: every hand-out of a writable view -- a host check-out,
        # ``dev()``, ``ptr()`` (DirichletBC.apply, KSPSolver.solve_block and any ``S._U.dev()[...] = ...`` go through
        # them) -- counts as a write.  FractionalStep_AB_CN keeps "u still equals u1 bit for bit" on it and reads the
        # two blocks through ``rdev()`` / ``rptr()`` (read-only by contract) while that matters.  Monitoring code between
        # steps (error norms, writers, callbacks) should read through ``rdev()`` / ``rptr()`` / ``rhost()`` too: a
        # ``dev()`` for a pure read is safe but silently costs the tentative solve its free first mat-vec.
        self.generation = 0

    def mark_written(self):
        self.generation += 1

    def host(self) -> np.ndarray:
        if self._shared:
            self.generation += 1  # the caller may write through the shared view
            return self._host
        if not self._out:
            if self._host is None:
                self._host = np.empty((self.n, self.nc))
            self._host[...] = self._dev[: self.n].cpu().numpy()
            self._out = True
        return self._host

    def _sync(self) -> torch.Tensor:
        if self._out:
            self._dev[: self.n].copy_(torch.from_numpy(self._host))
            self._out = False
            self.generation += 1
        return self._dev

    def dev(self) -> torch.Tensor:
        """The device block, writable: counts as a write (see ``generation``)."""
        self._sync()
        self.generation += 1
        return self._dev

    def ptr(self):
        """Device pointer for a kernel that may write the block: counts as a write."""
        return C.c_void_p(self.dev().data_ptr())

    def rdev(self) -> torch.Tensor:
        """The device block for READING only (the caller promises not to write through it)."""
        return self._sync()

    def rptr(self):
        """Device pointer for a kernel that only reads the block."""
        return C.c_void_p(self._sync().data_ptr())

    def rhost(self) -> np.ndarray:
        """A host COPY of the block for reading (writers, monitors, error norms): unlike ``host()`` nothing is checked
        out, so the block does not count as written (``generation`` stays: the solver's ``A u1`` shortcut survives)."""
        if self._shared:
            return self._host.copy()
        return self._sync()[: self.n].cpu().numpy()


class Vector:
    """``dolfinx.la.Vector`` / PETSc Vec stand-in over one column (or all) of a FieldStorage."""

    def __init__(self, storage: FieldStorage, comp: int | None):
        self._s, self._c = storage, comp

    @property
    def array(self) -> np.ndarray:
        h = self._s.host()
        return h.reshape(-1) if self._c is None else h[:, self._c]

    @property
    def petsc_vec(self):
        return self

    def scatter_forward(self):
        pass

    def scatter_reverse(self, mode=None):
        pass


class Constant:
    """``dolfinx.fem.Constant``: a mutable value shared by reference."""

    def __init__(self, mesh, value):
        self.mesh = mesh
        self._v = np.asarray(value, dtype=np.float64).copy()

    @property
    def value(self):
        return self._v

    @value.setter
    def value(self, v):
        self._v[...] = np.asarray(v, dtype=np.float64)

    def __float__(self):
        return float(self._v)


class Function:
    """Field on a scalar space (one column of a shared block) or on a blocked space."""

    def __init__(self, V, name: str = "f", storage: FieldStorage | None = None, comp: int | None = None):
        self.function_space = V
        self.name = name
        if storage is None:
            if isinstance(V, VectorFunctionSpace) or (getattr(V, "is_dg", False) and V.dim > 1):
                storage, comp = FieldStorage(V.num_dofs, V.dim, V.mesh.device), None
            else:
                storage, comp = FieldStorage(V.num_dofs, 1, V.mesh.device), 0
        self._storage, self._comp = storage, comp
        self.x = Vector(storage, comp)

    def interpolate(self, f):
        """Nodal interpolation of ``f(x)``, x of shape (3, npts) (reference bcs.py:125,133)."""
        if isinstance(f, Function):
            self.x.array[:] = f.x.array
            return
        V = self.function_space
        if getattr(f, "supports_torch", False) and self._comp is not None:
            # a callable marked ``supports_torch`` (as for DirichletBC values, bcs.py) gets the dof
            # coordinates as a (3, n) DEVICE tensor and returns a device tensor: no host round trip
            n = V.n_local
            Xd = torch.zeros((3, n), dtype=torch.float64, device=V.mesh.device)
            Xd[: V.mesh.gdim] = V.x[:n].T
            vals_d = f(Xd)
            if not torch.is_tensor(vals_d):
                raise TypeError("a supports_torch callable must return a torch tensor")
            self._storage.dev()[:n, self._comp] = vals_d.reshape(-1).to(torch.float64)
            self._storage.mark_written()
            return
        X = V.tabulate_dof_coordinates().T
        vals = np.asarray(f(X), dtype=np.float64)
        if self._comp is None:
            dim = self._storage.nc
            self._storage.host()[:, :] = vals.reshape(dim, -1).T
        else:
            self.x.array[:] = vals.reshape(-1)


_marker_scope: dict | None = None  # see shared_marker_evaluations()


class shared_marker_evaluations:
    """Scope inside which ``locate_dofs_geometrical`` evaluates one marker object once per space.
    ``FractionalStep_AB_CN.__init__`` opens it around the creation of its boundary conditions: the gdim
    velocity components usually share their marker, and at 128^3 a numpy marker over 17 M dof coordinates
    costs 0.4 s each time.  Outside such a scope every call evaluates the marker, as DOLFINx does
    (reference bcs.py:98-103) -- a marker that reads mutable state is never served from a stale cache."""

    def __enter__(self):
        global _marker_scope
        self._outer, _marker_scope = _marker_scope, {}
        return self

    def __exit__(self, *exc):
        global _marker_scope
        _marker_scope = self._outer
        return False


def locate_dofs_geometrical(V, marker) -> np.ndarray:
    """Dofs whose coordinates the marker selects (reference bcs.py:98-103)."""
    cache, key = _marker_scope, None
    if cache is not None:
        try:
            key = (id(V), marker)
            hit = cache.get(key)
        except TypeError:  # unhashable callable
            hit, cache = None, None
        if hit is not None:
            return hit.copy()
    X = V.tabulate_dof_coordinates().T
    dofs = np.nonzero(np.asarray(marker(X), dtype=bool))[0].astype(np.int32)
    if cache is not None:
        cache[key] = dofs
    return dofs.copy()


def locate_dofs_topological(V, entity_dim: int, entities) -> np.ndarray:
    return V.entity_dofs(entity_dim, np.asarray(entities))


# ---- error functionals of the demo harness (reference demo/taylor_green.py:193-207) -----------


def _simplex_rule(d: int, n: int):
    """Collapsed Gauss-Jacobi rule, exact to degree 2n-1; barycentric points, weights sum 1/d!."""
    from scipy.special import roots_jacobi

    pts = []
    for a in range(d - 1, -1, -1):
        s, w = roots_jacobi(n, float(a), 0.0)
        pts.append(((1 + s) / 2, w / 2 ** (a + 1)))
    grids = np.meshgrid(*[p[0] for p in pts], indexing="ij")
    W = np.ones_like(grids[0])
    for k, p in enumerate(pts):
        shape = [1] * d
        shape[k] = n
        W = W * p[1].reshape(shape)
    coords, rem = [], np.ones_like(grids[0])
    for g in grids:
        coords.append(rem * g)
        rem = rem * (1 - g)
    X = np.stack([c.ravel() for c in coords], axis=1)
    bary = np.concatenate([1 - X.sum(axis=1, keepdims=True), X], axis=1)
    return bary, W.ravel()


GLL3 = (0.5 - 0.5 / np.sqrt(5.0), 0.5 + 0.5 / np.sqrt(5.0))  # interior edge nodes of the gll_warped P3 element


def lagrange_basis(d: int, degree: int, bary: np.ndarray) -> np.ndarray:
    """phi (npts, nd) of the Lagrange element the solver's spaces use, at barycentric points: P1, P2, and on triangles
    the ``gll_warped`` P3 element (reference fracstep.py:170,181; the same node layout as csrc/fe_tables_h.h: vertices,
    per local edge the node nearer its first vertex then the one nearer its second, the centroid)."""
    nv = d + 1
    if degree == 1:
        return bary
    if degree == 2:
        cols = [bary[:, a] * (2 * bary[:, a] - 1) for a in range(nv)]
        cols += [4 * bary[:, a] * bary[:, b] for a, b in local_edges(d)]
        return np.stack(cols, axis=1)
    if degree == 3:
        return _p3_mono(d, bary)[0] @ _p3_coef(d)
    raise NotImplementedError(f"Lagrange degree {degree} on a {d}-simplex")


_P3_COEFS = {}


def _p3_mono(d: int, bary: np.ndarray):
    """The monomials of degree <= 3 in (lambda_1, ..., lambda_d) at barycentric points and their partial derivatives:
    (m, dm_1, ..., dm_d), each (npts, 10 | 20)."""
    xs = [bary[:, a] for a in range(1, d + 1)]
    ex = ([(i, j) for i in range(4) for j in range(4 - i)] if d == 2 else
          [(i, j, k) for i in range(4) for j in range(4 - i) for k in range(4 - i - j)])

    def mono(e, skip=None):
        out = np.ones_like(xs[0])
        for v in range(d):
            out = out * xs[v] ** max(e[v] - (1 if v == skip else 0), 0)
        return out
    m = np.stack([mono(e) for e in ex], axis=1)
    ders = [np.stack([e[v] * mono(e, v) if e[v] > 0 else np.zeros_like(xs[0]) for e in ex], axis=1) for v in range(d)]
    return (m, *ders)


def p3_nodes(d: int) -> np.ndarray:
    """Barycentric coordinates of the ``gll_warped`` P3 nodes in the order of csrc/fe_tables_h.h / fe_tables_h3.h:
    vertices, per local edge the node nearer its first vertex then the one nearer its second, then the centroid
    (triangles) or the centroids of the four faces (tetrahedra: face f = the vertices other than f)."""
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
        for f in range(4):
            v = np.full(4, 1.0 / 3.0)
            v[f] = 0.0
            nodes.append(v)
    return np.array(nodes)


def _p3_coef(d: int) -> np.ndarray:
    if d not in _P3_COEFS:
        _P3_COEFS[d] = np.linalg.inv(_p3_mono(d, p3_nodes(d))[0])  # column i: monomial coefficients of phi_i
    return _P3_COEFS[d]


def lagrange_basis_derivs(d: int, degree: int, bary: np.ndarray) -> np.ndarray:
    """dphi (npts, nd, d+1) = d(phi_i)/d(lambda_b) with the barycentric coordinates treated as independent variables
    (grad phi_i = sum_b dphi[:, i, b] grad lambda_b).  The degree-3 element is written as a polynomial of (lambda_1,
    lambda_2) alone: its column b = 0 is zero (the convention of csrc/fe_tables_h.h)."""
    nv = d + 1
    n = bary.shape[0]
    if degree == 1:
        return np.broadcast_to(np.eye(nv), (n, nv, nv)).copy()
    if degree == 2:
        edges = local_edges(d)
        out = np.zeros((n, nv + len(edges), nv))
        for a in range(nv):
            out[:, a, a] = 4 * bary[:, a] - 1
        for e, (a, b) in enumerate(edges):
            out[:, nv + e, a] = 4 * bary[:, b]
            out[:, nv + e, b] = 4 * bary[:, a]
        return out
    if degree == 3:
        tabs, Cf = _p3_mono(d, bary), _p3_coef(d)
        out = np.zeros((n, Cf.shape[0], nv))
        for b in range(1, nv):
            out[:, :, b] = tabs[b] @ Cf
        return out
    raise NotImplementedError(f"Lagrange degree {degree} on a {d}-simplex")


def assemble_l2_error_sq(u: Function, exact, degree_raise: int = 3) -> float:
    """int (u_h - exact)^2 dx over the local cells (``assemble_scalar`` of the demo's error form).
    ``exact`` maps x:(3, npts) -> (npts,).  Host numpy; a harness functional, not the hot path."""
    V = u.function_space
    mesh = V.mesh
    d = mesh.gdim
    bary, w = _simplex_rule(d, V.degree + degree_raise)
    phi = lagrange_basis(d, V.degree, bary)
    lc = V.local_cells
    cd = V.cell_dofs
    if V.part is not None:  # integrate over the cells this rank owns (the ghost layer belongs to others)
        mine = V.part.cell_rank[lc] == V.part.rank
        lc, cd = lc[mine], cd[mine]
    cells = mesh.cells[lc].cpu().numpy()
    xc = mesh.coords.cpu().numpy()[cells]
    xq = np.einsum("qa,cak->cqk", bary, xc)
    X = np.zeros((3, xq.shape[0] * xq.shape[1]))
    X[:d] = xq.reshape(-1, d).T
    ex = np.asarray(exact(X)).reshape(xq.shape[0], xq.shape[1])
    uh = u.x.array[cd.cpu().numpy()] @ phi.T
    J = np.moveaxis(xc[:, 1:, :] - xc[:, :1, :], 1, 2)
    adet = np.abs(np.linalg.det(J))
    return float(np.einsum("q,cq,c->", w, (uh - ex) ** 2, adet))
