"""Device matrices of the IPCS path: a SELL-64 pattern plus a float64 value array."""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from .fem import SellPattern


def _env_int(name):
    try:
        return int(os.environ[name])
    except (KeyError, ValueError):
        return None


# tuning defaults of the mat-vec, read by the HOST layer once and written into every matrix's own ox_sell (the library
# keeps no process-wide switch): storage levels (include/oasisx_hip.h: ox_sell.levels) and the LDS-window budget
ENV_SPMV_LEVELS = _env_int("OX_SPMV_VARIANT")
ENV_WIN_CAP = _env_int("OX_WIN_CAP")


def _apply_schedule(struct, levels, w_cap):
    struct.levels = 0 if levels is None else (32 | (int(levels) & 31))
    struct.w_cap = 0 if not w_cap else int(w_cap)


class SellMatrix:
    """What stands in for a PETSc ``Mat`` on this path (reference fracstep.py:293-300,324)."""

    def __init__(self, pattern: SellPattern, symmetric: bool = False, name: str = "A"):
        self.pattern = pattern
        self.vals = pattern.new_values()
        self.symmetric = symmetric
        self.name = name
        self.version = 0  # bumped whenever the values change (Jacobi setup is cached on it)
        self.vcode = self.vdict = self._vc_version = None  # value dictionary, see freeze()
        self.ps_ptr = self.ps_code = self.ps_base = None  # pair-slot stream, see freeze()
        self.ps_wide = 0
        self.wvcode = None  # value codes in the tile layout of the pattern's LDS-window stream, see freeze()
        self._struct = pattern.struct(self.vals)
        self.levels, self.w_cap = ENV_SPMV_LEVELS, ENV_WIN_CAP
        _apply_schedule(self._struct, self.levels, self.w_cap)

    def set_levels(self, mask: int | None, w_cap: int | None = None):
        """Storage levels the mat-vecs of THIS matrix may use (bit 0 nontemporal stream, 1 16-bit columns, 2 value codes,
        3 pair slots, 4 LDS windows; None = all it carries) and, optionally, the LDS-window budget in entries.  A/B hook
        of tests and tools: results never depend on it."""
        self.levels = mask
        if w_cap is not None:
            self.w_cap = w_cap or None
        _apply_schedule(self._struct, self.levels, self.w_cap)

    @property
    def struct(self) -> _lib.ox_sell:
        return self._struct

    def ref(self):
        if self._vc_version is not None and self._vc_version != self.version:
            self._drop_codes()  # the values changed after freeze(): the codes no longer describe them
        return C.byref(self._struct)

    def _drop_codes(self):
        self.vcode = self.vdict = self._vc_version = None
        self._struct.vcode = self._struct.vdict = None
        self._struct.n_dict = 0
        self.ps_ptr = self.ps_code = self.ps_base = None
        self._struct.ps_ptr = self._struct.ps_code = self._struct.ps_base = None
        self.wvcode = None
        self._struct.wvcode = None

    def freeze(self, block: int = 1 << 27, pairs: str = "auto") -> bool:
        """Value dictionary for a matrix whose values will not change any more (M, Ap: assembled
        once, reference fracstep.py:373-380): if the stored values take at most 256 distinct bit
        patterns -- mass and stiffness matrices on meshes of congruent cells do: 49 / 14 on the box
        meshes -- the SpMV streams one code byte per entry instead of the 8-byte value and looks the
        value up in an LDS copy of the dictionary.  Same values, same order of sums: bit-identical
        results.  Returns whether a dictionary was built (False: nothing changes).
        ``pairs``: "auto" builds the pair-slot stream where it stores at most 0.72 slots per entry slot (P1
        patterns: 0.64; P2: 0.87, where it is slower), "always" / "never" force it."""
        P = self.pattern
        if P.device.type != "cuda" or P.cols16 is None or P.size == 0:
            return False
        code = torch.empty(P.size, dtype=torch.uint8, device=P.device)
        vdict = torch.zeros(256, dtype=torch.float64, device=P.device)
        nd = C.c_int(0)
        _lib.check(_lib.load().ox_value_dictionary(_lib.ptr(self.vals), P.size, 1, _lib.ptr(code), _lib.ptr(vdict),
                                                   C.byref(nd), _lib.current_stream()), "ox_value_dictionary")
        if nd.value == 0:
            return False
        self.vcode, self.vdict = code, vdict[: nd.value]
        self._vc_version = self.version
        self._struct.vcode, self._struct.vdict = self.vcode.data_ptr(), self.vdict.data_ptr()
        self._struct.n_dict = int(nd.value)
        if pairs != "never":
            self._build_pair_stream(force=pairs == "always")
        if getattr(P, "wt_ptr", None) is not None and self.ps_code is None:
            # the pattern carries an LDS-window stream: the value codes once more in its tile layout (4 codes of a
            # lane per 4-byte load)
            self.wvcode = torch.empty(int(P.wt_ptr[-1].item()) * 256, dtype=torch.uint8, device=P.device)
            _lib.check(_lib.load().ox_window_retile(C.byref(self._struct), _lib.ptr(P.wt_ptr), _lib.ptr(self.vcode), 1,
                                                    _lib.ptr(self.wvcode), _lib.current_stream()), "ox_window_retile")
            self._struct.wvcode = self.wvcode.data_ptr()
        return True

    def _build_pair_stream(self, force: bool = False):
        """Pair-slot stream of the frozen matrix (``ox_sell.ps_*``): entries in adjacent columns share
        one 16-byte gather.  The SpMV on dictionary matrices is bound by the number of vector-memory
        instructions, not by bytes; bit-identical results."""
        P, lib = self.pattern, _lib.load()
        ps_ptr = torch.empty(P.n_slices + 1, dtype=torch.int64, device=P.device)
        n = C.c_int64(0)
        _lib.check(lib.ox_pair_stream_size(C.byref(self._struct), _lib.ptr(P.row_len), _lib.ptr(ps_ptr), C.byref(n),
                                           _lib.current_stream()), "ox_pair_stream_size")
        if n.value == 0 or (not force and n.value > 0.72 * P.size):
            return
        code = torch.empty(n.value, dtype=torch.int32, device=P.device)
        base = torch.empty(2 * (n.value // 256), dtype=torch.int32, device=P.device)
        wide = C.c_int64(0)
        _lib.check(lib.ox_pair_stream_fill(C.byref(self._struct), _lib.ptr(P.row_len), _lib.ptr(ps_ptr), _lib.ptr(code),
                                           _lib.ptr(base), C.byref(wide), _lib.current_stream()), "ox_pair_stream_fill")
        self.ps_ptr, self.ps_code, self.ps_base, self.ps_wide = ps_ptr, code, base, int(wide.value)
        self._struct.ps_ptr, self._struct.ps_code, self._struct.ps_base = ps_ptr.data_ptr(), code.data_ptr(), base.data_ptr()

    def getSize(self):
        return (self.pattern.n_rows, self.pattern.n_cols)

    def to_scipy(self):
        """Host copy as scipy CSR (tests / diagnostics)."""
        return self.pattern.to_csr(self.vals)

    def mult(self, x: torch.Tensor, y: torch.Tensor, ncomp: int = 1):
        """y = A x on interleaved (n, ncomp) device blocks (PETSc Mat.mult); in mesh-partitioned
        runs the ghost block of x is refreshed first (halo exchange)."""
        lib = _lib.load()
        _lib.check(lib.ox_spmv(self.ref(), _lib.ptr(x), _lib.ptr(y), ncomp, self.pattern.dist,
                               _lib.current_stream()), "ox_spmv")

    def zero_rows(self, rows_dev: torch.Tensor, diag: float = 1.0, au=None, u1=None, ncomp: int = 1):
        """Mat.zeroRowsLocal(rows, diag): keeps the columns (reference fracstep.py:471-472).  ``au``/``u1``
        (device pointers): also set ``au[rows] = diag * u1[rows]``, the identity rows of a product A u1 that was
        formed before the rows were zeroed (FractionalStep_AB_CN.assemble_first) -- same launch."""
        lib = _lib.load()
        _lib.check(lib.ox_zero_rows_au(self.ref(), _lib.ptr(rows_dev), int(rows_dev.shape[0]), float(diag), au, u1,
                                       int(ncomp), _lib.current_stream()), "ox_zero_rows")
        self.version += 1


class MultiSellMatrix:
    """Rectangular operator with ``gdim`` values per entry on one SELL-64 pattern: the reference's
    pre-assembled ``p*v.dx(i)*dx`` / ``p.dx(i)*v*dx`` / ``u.dx(i)*q*dx`` matrices for all i
    (fracstep.py:311-315,332-336,348-352) stored together so one pass serves every component."""

    def __init__(self, pattern: SellPattern, gdim: int, name: str = "R"):
        self.pattern, self.gdim, self.name = pattern, gdim, name
        self.vals = torch.zeros(pattern.size * gdim, dtype=torch.float64, device=pattern.device)
        self._struct = pattern.struct(self.vals, compress=False)
        self.vcode = self.vdict = None
        self.levels = ENV_SPMV_LEVELS
        _apply_schedule(self._struct, self.levels, None)

    def set_levels(self, mask: int | None):
        """As ``SellMatrix.set_levels`` (bit 2 off: the f64 value streams instead of the packed value codes)."""
        self.levels = mask
        _apply_schedule(self._struct, self.levels, None)

    def ref(self):
        return C.byref(self._struct)

    def freeze(self, block: int = 1 << 26) -> bool:
        """Value dictionary for the (constant) operator, as ``SellMatrix.freeze``: if all
        ``gdim`` value arrays together take at most 256 distinct bit patterns (they do on meshes
        of congruent cells), every entry keeps one packed uint32 of ``gdim`` code bytes; with the
        16-bit column stream the mat-vec then reads 6 B per entry instead of 4 + 8*gdim.
        Bit-identical results.  Returns whether the dictionary was built."""
        P = self.pattern
        if P.device.type != "cuda" or P.size == 0:
            return False
        full = P.struct(self.vals)  # builds / fetches the pattern's 16-bit column stream
        if P.cols16 is None:
            return False
        code = torch.zeros(P.size, dtype=torch.int32, device=P.device)
        vdict = torch.zeros(256, dtype=torch.float64, device=P.device)
        nd = C.c_int(0)
        _lib.check(_lib.load().ox_value_dictionary(_lib.ptr(self.vals), P.size, self.gdim, _lib.ptr(code), _lib.ptr(vdict),
                                                   C.byref(nd), _lib.current_stream()), "ox_value_dictionary")
        if nd.value == 0:
            return False
        self.vcode, self.vdict = code, vdict[: nd.value]
        full.vcode, full.vdict, full.n_dict = self.vcode.data_ptr(), self.vdict.data_ptr(), int(nd.value)
        self._plain = self._struct
        self._struct = full
        _apply_schedule(self._struct, self.levels, None)
        return True

    def unfreeze(self):
        """Back to the f64 value stream (bench.py's dictionary-off leg); same results."""
        if self.vcode is not None:
            self._struct = self._plain
            _apply_schedule(self._struct, self.levels, None)
            self.vcode = self.vdict = None

    def mult(self, v2s: bool, x, base, scale: float, y):
        """y = base + scale * (A applied to x); x, base, y are device pointers (c_void_p)."""
        lib = _lib.load()
        _lib.check(lib.ox_spmv_multi(int(v2s), self.gdim, self.ref(), x, base, float(scale), y,
                                     self.pattern.dist, _lib.current_stream()), "ox_spmv_multi")

    def to_scipy(self, d: int):
        """Component ``d`` as scipy CSR (tests)."""
        return self.pattern.to_csr(self.vals.reshape(-1, self.gdim)[:, d].contiguous())
