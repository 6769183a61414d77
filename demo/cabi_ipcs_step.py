#!/usr/bin/env python3
"""The IPCS time step driven through the C ABI alone: numpy arrays + ctypes + include/oasisx_hip.h.

This is the binding INTEGRATION.md section 2 describes, written out: nothing of the ``oasisx_amd``
Python package is imported (no torch, no fem.py) -- the ctypes declarations come from
``oasisx_amd/_lib.py`` loaded as a stand-alone file (it only mirrors the header), device memory from
``ox_malloc``.  Every hot call site of the reference's ``FractionalStep_AB_CN.solve``
(src/oasisx/fracstep.py:660-696) is one call below; the set-up (mesh -> spaces -> patterns ->
M, K, Ap) is ``ox_mesh_create`` / ``ox_space_create`` / ``ox_assemble_matrix``.

    python demo/cabi_ipcs_step.py --dim 2 -N 16 --steps 3 --out /tmp/step.npz
"""
from __future__ import annotations

import argparse
import ctypes as C
import importlib.util
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_binding():
    spec = importlib.util.spec_from_file_location("ox_binding", os.path.join(ROOT, "oasisx_amd", "_lib.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, mod.load()


def box_mesh(dim, N, lo=-1.0, hi=1.0):
    """Any conforming simplicial mesh will do; this one is the N^dim box cut into 2 triangles per
    square (diagonal alternating) or 5 tetrahedra per cube (parity-alternating so faces match) --
    deliberately NOT the generator of the product or of the oracle."""
    ax = np.linspace(lo, hi, N + 1)
    if dim == 2:
        X, Y = np.meshgrid(ax, ax, indexing="ij")
        coords = np.stack([X.ravel(), Y.ravel()], axis=1)
        vid = lambda i, j: i * (N + 1) + j  # noqa: E731
        cells = []
        for i in range(N):
            for j in range(N):
                a, b, c, d = vid(i, j), vid(i + 1, j), vid(i, j + 1), vid(i + 1, j + 1)
                cells += [[a, b, d], [a, d, c]] if (i + j) % 2 == 0 else [[a, b, c], [b, d, c]]
        return coords, np.asarray(cells, dtype=np.int32)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    vid = lambda i, j, k: (i * (N + 1) + j) * (N + 1) + k  # noqa: E731
    cells = []
    for i in range(N):
        for j in range(N):
            for k in range(N):
                v = [vid(i + a, j + b, k + c) for a in (0, 1) for b in (0, 1) for c in (0, 1)]  # index 4a+2b+c
                if (i + j + k) % 2 == 0:
                    cells += [[v[0], v[4], v[2], v[1]], [v[6], v[2], v[4], v[7]], [v[5], v[4], v[1], v[7]],
                              [v[3], v[1], v[2], v[7]], [v[4], v[2], v[1], v[7]]]
                else:
                    cells += [[v[1], v[0], v[3], v[5]], [v[2], v[0], v[6], v[3]], [v[4], v[0], v[5], v[6]],
                              [v[7], v[3], v[6], v[5]], [v[0], v[3], v[5], v[6]]]
    return coords, np.asarray(cells, dtype=np.int32)


class Device:
    """ox_malloc / ox_memcpy around numpy arrays."""

    def __init__(self, L, lib):
        self.L, self.lib, self.live = L, lib, []

    def zeros(self, n, dtype=np.float64):
        p = C.c_void_p()
        nbytes = int(n) * np.dtype(dtype).itemsize
        self.L.check(self.lib.ox_malloc(max(nbytes, 8), C.byref(p)), "ox_malloc")
        self.L.check(self.lib.ox_memset(p, 0, max(nbytes, 8), None), "ox_memset")
        self.live.append(p)
        return p

    def upload(self, a):
        a = np.ascontiguousarray(a)
        p = self.zeros(a.size, a.dtype)
        if a.size:
            self.L.check(self.lib.ox_memcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1, None), "ox_memcpy")
        return p

    def set(self, p, a):
        a = np.ascontiguousarray(a)
        self.L.check(self.lib.ox_memcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1, None), "ox_memcpy")

    def download(self, p, shape, dtype=np.float64):
        out = np.zeros(shape, dtype=dtype)
        if out.size:
            self.L.check(self.lib.ox_memcpy(out.ctypes.data_as(C.c_void_p), p, out.nbytes, 0, None), "ox_memcpy")
        return out

    def free_all(self):
        for p in self.live:
            self.lib.ox_free(p)
        self.live = []


class IPCS:
    """FractionalStep_AB_CN for Dirichlet velocity data on the whole boundary, no pressure condition,
    low_memory_version=True, Jacobi-BiCGStab / Jacobi-CG -- every operation a C-ABI call."""

    def __init__(self, coords, cells, u_deg, rtol=1e-11, compress=False, cg_merged=False, windows=False):
        self.L, self.lib = L, lib = load_binding()
        self.windows = windows  # brick order of the velocity numbering + the LDS-window stream of its pattern
        self.cg_pressure = L.KSP_CG_MERGED if cg_merged else L.KSP_CG  # OX_KSP_CG_MERGED: one synchronisation point per iteration
        self.dev = dev = Device(L, lib)
        self.gdim = d = coords.shape[1]
        self.rtol = rtol
        ck = L.check
        mesh = C.c_void_p()
        ck(lib.ox_mesh_create(np.ascontiguousarray(coords, dtype=np.float64).ctypes.data_as(C.c_void_p), coords.shape[0],
                              np.ascontiguousarray(cells, dtype=np.int32).ctypes.data_as(C.c_void_p), cells.shape[0], d, 0, -1,
                              C.byref(mesh)), "ox_mesh_create")
        self.mesh = mesh
        self.mv = L.ox_mesh_info()
        ck(lib.ox_mesh_view(mesh, C.byref(self.mv)), "ox_mesh_view")
        self.V, self.vv = self._space(u_deg, brick=windows)
        self.Q, self.qv = (self.V, self.vv) if u_deg == 1 else self._space(1)
        self.window_info = None
        if windows:  # ox_space_windows: built once per space, the arrays belong to it
            self.window_info = w = L.ox_window_info()
            ck(lib.ox_space_windows(self.V, C.byref(w)), "ox_space_windows")
        self.n_u, self.n_q = int(self.vv.n_dofs), int(self.qv.n_dofs)
        self.x_v = dev.download(self.vv.x, (self.n_u, d))
        self.x_q = dev.download(self.qv.x, (self.n_q, d))
        # matrices: one value array each on the space's pattern (M, K, A share it: fracstep.py:293-294)
        self.M, self.K, self.A = (self._matrix(self.vv.pattern) for _ in range(3))
        self.Ap = self._matrix(self.qv.pattern)
        self._assemble(0, self.vv, self.M)
        self._assemble(1, self.vv, self.K)
        self._assemble(1, self.qv, self.Ap)
        # optional storage levels of the constant matrices (bit-identical results): 1-byte value codes where
        # the matrix has <= 256 distinct values, and on top of them the pair-slot stream
        self.compressed = {}
        if compress:
            for name, A, pat in (("M", self.M, self.vv.pattern), ("K", self.K, self.vv.pattern), ("Ap", self.Ap, self.qv.pattern)):
                self.compressed[name] = self._compress(A, pat, pairs=(name == "Ap"))
        self.wq = dev.zeros(self.n_q)
        ck(lib.ox_assemble_weights(1, C.byref(self.mv.cells_struct), C.byref(self.qv.adj), self.n_q, self.wq, None),
           "ox_assemble_weights")
        self.vol = float(dev.download(self.wq, (self.n_q,)).sum())
        nvec = self.n_u * d
        (self.U, self.U1, self.U2, self.UAB, self.RHS1, self.B0, self.BFIRST, self.B3) = (dev.zeros(nvec) for _ in range(8))
        self.PS, self.P, self.DP, self.B2 = (dev.zeros(self.n_q) for _ in range(4))
        lo, hi = coords.min(axis=0), coords.max(axis=0)
        on = np.zeros(self.n_u, dtype=bool)
        for k in range(d):
            on |= np.isclose(self.x_v[:, k], lo[k]) | np.isclose(self.x_v[:, k], hi[k])
        self.bc_dofs = np.nonzero(on)[0].astype(np.int32)
        self.bc_dofs_dev = dev.upload(self.bc_dofs)
        self.g_dev = dev.zeros(self.bc_dofs.shape[0])
        self.dinvA, self.dinvM, self.dinvP = dev.zeros(self.n_u), dev.zeros(self.n_u), dev.zeros(self.n_q)
        ck(lib.ox_jacobi_setup(C.byref(self.M), self.dinvM, None), "ox_jacobi_setup")
        ck(lib.ox_jacobi_setup(C.byref(self.Ap), self.dinvP, None), "ox_jacobi_setup")
        # (sized on the operators themselves: a window stream may launch more blocks than the lane = row grid)
        wb = max(lib.ox_ksp_work_bytes_for(C.byref(self.A), d, L.KSP_BCGS), lib.ox_ksp_work_bytes_for(C.byref(self.M), d, L.KSP_CG),
                 lib.ox_ksp_work_bytes_for(C.byref(self.Ap), 1, L.KSP_CG))
        self.work_bytes = int(wb)
        self.work = dev.zeros(wb, np.uint8)
        self.its = {}

    def _space(self, degree, brick=False):
        s = C.c_void_p()
        self.L.check(self.lib.ox_space_create_ordered(self.mesh, degree, 0, 1 if brick else 0, C.byref(s)),
                     "ox_space_create_ordered")
        v = self.L.ox_space_info()
        self.L.check(self.lib.ox_space_view(s, C.byref(v)), "ox_space_view")
        return s, v

    def _matrix(self, pat):
        A = self.L.ox_sell()
        C.memmove(C.byref(A), C.byref(pat.sell), C.sizeof(A))
        A.vals = self.dev.zeros(int(pat.size)).value
        w = getattr(self, "window_info", None)
        if w is not None and pat is self.vv.pattern and w.n_wblocks > 0:  # the velocity matrices run on the window stream
            for f in ("wb_slices", "wb_waves", "wb_ptr", "wlist", "wt_ptr", "wcode", "n_wblocks", "w_max"):
                setattr(A, f, getattr(w, f))
        return A

    def _compress(self, A, pat, pairs):
        """ox_value_dictionary, then ox_pair_stream_size / _fill; returns what was built."""
        dev, lib, ck = self.dev, self.lib, self.L.check
        n = int(pat.size)
        codes, vdict, nd = dev.zeros(n, np.uint8), dev.zeros(256), C.c_int(0)
        ck(lib.ox_value_dictionary(A.vals, n, 1, codes, vdict, C.byref(nd), None), "ox_value_dictionary")
        if nd.value == 0:
            return {"n_dict": 0, "pair_codes": 0}
        A.vcode, A.vdict, A.n_dict = codes.value, vdict.value, nd.value
        built = {"n_dict": int(nd.value), "pair_codes": 0}
        if A.n_wblocks > 0:  # a dictionary matrix on the window stream: its value codes in the tile layout
            tiled = dev.zeros(int(self.window_info.n_tiles) * 256, np.uint8)
            ck(lib.ox_window_retile(C.byref(A), A.wt_ptr, codes, 1, tiled, None), "ox_window_retile")
            A.wvcode = tiled.value
            built["tiled_codes"] = int(self.window_info.n_tiles) * 256
        if pairs:
            ps_ptr, ncodes = dev.zeros(int(A.n_slices) + 1, np.int64), C.c_int64(0)
            ck(lib.ox_pair_stream_size(C.byref(A), pat.row_len, ps_ptr, C.byref(ncodes), None), "ox_pair_stream_size")
            if ncodes.value > 0:
                code, base, wide = dev.zeros(ncodes.value, np.uint32), dev.zeros(2 * (ncodes.value // 256), np.int32), C.c_int64(0)
                ck(lib.ox_pair_stream_fill(C.byref(A), pat.row_len, ps_ptr, code, base, C.byref(wide), None), "ox_pair_stream_fill")
                A.ps_ptr, A.ps_code, A.ps_base = ps_ptr.value, code.value, base.value
                built.update(pair_codes=int(ncodes.value), wide_slices=int(wide.value))
        return built

    def _assemble(self, kind, sv, A):
        p = sv.pattern
        self.L.check(self.lib.ox_assemble_matrix(kind, sv.degree, C.byref(self.mv.cells_struct), sv.cell_dofs, C.byref(sv.adj),
                                                 sv.adj_pos, sv.pw, C.byref(A), p.n_bins,
                                                 C.cast(p.bin_ptr_host, C.POINTER(C.c_int64)), p.bin_slices,
                                                 C.cast(p.bin_width_host, C.POINTER(C.c_int32)), None), "ox_assemble_matrix")

    def set_field(self, dev_ptr, values):  # values: (n, ncomp) or (n,)
        self.dev.set(dev_ptr, np.ascontiguousarray(values, dtype=np.float64))

    def _solve(self, kind, A, dinv, b, x, ncomp, name):
        res = self.L.ox_ksp_result()
        self.L.check(self.lib.ox_ksp_solve(kind, C.byref(A), dinv, b, x, ncomp, self.rtol, 1e-30, 10000, 0, 8, 0, self.work,
                                           self.work_bytes, C.byref(res), None, None), "ox_ksp_solve")
        reasons = [int(res.reason[c]) for c in range(ncomp)]
        assert all(r > 0 for r in reasons), (name, reasons)
        self.its[name] = [int(res.its[c]) for c in range(ncomp)]

    def step(self, dt, nu, g):
        """One time step (reference fracstep.py:660-696); g: (gdim, n_bc) Dirichlet values."""
        lib, ck, d = self.lib, self.L.check, self.gdim
        vv, qv, cs = self.vv, self.qv, C.byref(self.mv.cells_struct)
        n, nq, nvec = self.n_u, self.n_q, self.n_u * d
        pv = vv.pattern
        ck(lib.ox_axpby(nq, 1.0, self.P, 0.0, None, self.PS, None), "ps = p")
        ck(lib.ox_axpby(nvec, 1.5, self.U1, -0.5, self.U2, self.UAB, None), "u_ab")  # :432-434
        ck(lib.ox_assemble_first(vv.degree, cs, vv.cell_dofs, C.byref(vv.adj), vv.adj_pos, vv.pw, C.byref(self.A),
                                 C.byref(self.M), C.byref(self.K), self.UAB, self.U1, self.B0, self.BFIRST, dt, nu,
                                 pv.n_bins, C.cast(pv.bin_ptr_host, C.POINTER(C.c_int64)), pv.bin_slices,
                                 C.cast(pv.bin_width_host, C.POINTER(C.c_int32)), None), "ox_assemble_first")  # :435-469
        ck(lib.ox_zero_rows(C.byref(self.A), self.bc_dofs_dev, self.bc_dofs.shape[0], 1.0, None), "ox_zero_rows")  # :470-472
        ck(lib.ox_assemble_grad_vector(0, vv.degree, 1, cs, qv.cell_dofs, C.byref(vv.adj), n, self.PS, self.BFIRST, 1.0,
                                       self.RHS1, None), "rhs1")  # :487-506
        for c in range(d):  # bc.apply(rhs1[i]) (:517-518)
            self.dev.set(self.g_dev, g[c])
            ck(lib.ox_set_bc(self.RHS1, self.bc_dofs_dev, self.g_dev, self.bc_dofs.shape[0], d, c, None), "ox_set_bc")
        ck(lib.ox_jacobi_setup(C.byref(self.A), self.dinvA, None), "ox_jacobi_setup")
        self._solve(self.L.KSP_BCGS, self.A, self.dinvA, self.RHS1, self.U, d, "tentative")  # :521
        ck(lib.ox_assemble_div_vector(1, vv.degree, cs, vv.cell_dofs, C.byref(qv.adj), nq, self.U, -1.0 / dt, self.B2, None),
           "b2")  # :538-546
        ck(lib.ox_remove_mean(nq, nq, self.B2, None, float(nq), None, None), "nullspace.remove")  # :573-574
        self._solve(self.cg_pressure, self.Ap, self.dinvP, self.B2, self.DP, 1, "pressure")  # :578
        ck(lib.ox_remove_mean(nq, nq, self.DP, self.wq, self.vol, None, None), "mean shift")  # :579-591
        ck(lib.ox_axpby(nq, 1.0, self.P, 1.0, self.DP, self.PS, None), "ps = p + dp")  # :604
        ck(lib.ox_spmv(C.byref(self.M), self.U, self.B3, d, None, None), "M u")  # :615
        ck(lib.ox_assemble_grad_vector(1, vv.degree, 1, cs, qv.cell_dofs, C.byref(vv.adj), n, self.DP, self.B3, -dt, self.B3,
                                       None), "b3")  # :618-622
        self._solve(self.L.KSP_CG, self.M, self.dinvM, self.B3, self.U, d, "update")  # :634
        ck(lib.ox_axpby(nvec, 1.0, self.U1, 0.0, None, self.U2, None), "u2 = u1")  # :689-693
        ck(lib.ox_axpby(nvec, 1.0, self.U, 0.0, None, self.U1, None), "u1 = u")
        ck(lib.ox_axpby(nq, 1.0, self.PS, 0.0, None, self.P, None), "p = ps")
        ck(lib.ox_synchronize(None), "ox_synchronize")

    def close(self):
        self.dev.free_all()
        if self.Q.value != self.V.value:
            self.lib.ox_space_destroy(self.Q)
        self.lib.ox_space_destroy(self.V)
        self.lib.ox_mesh_destroy(self.mesh)


def tg(dim, nu):
    def u(x, t):
        return -np.cos(np.pi * x[0]) * np.sin(np.pi * x[1]) * math.exp(-2 * nu * np.pi ** 2 * t)

    def v(x, t):
        return np.sin(np.pi * x[0]) * np.cos(np.pi * x[1]) * math.exp(-2 * nu * np.pi ** 2 * t)

    def w(x, t):
        return np.zeros_like(x[0])

    def p(x, t):
        return -0.25 * (np.cos(2 * np.pi * x[0]) + np.cos(2 * np.pi * x[1])) * math.exp(-4 * nu * np.pi ** 2 * t)
    return [u, v, w][:dim], p


def run(dim=2, N=8, u_deg=2, steps=2, nu=0.01, dt=0.005, rtol=1e-11, compress=False, cg_merged=False, windows=False):
    coords, cells = box_mesh(dim, N)
    S = IPCS(coords, cells, u_deg, rtol, compress, cg_merged, windows)
    fns, pf = tg(dim, nu)
    X, Xq = S.x_v.T, S.x_q.T
    S.set_field(S.U2, np.stack([f(X, -dt) for f in fns], axis=1))
    S.set_field(S.U1, np.stack([f(X, 0.0) for f in fns], axis=1))
    S.set_field(S.P, pf(Xq, -dt / 2))
    Xb = S.x_v[S.bc_dofs].T
    t = 0.0
    for _ in range(steps):
        t += dt
        S.step(dt, nu, np.stack([f(Xb, t) for f in fns]))
    out = {"coords": coords, "cells": cells, "x_v": S.x_v, "x_q": S.x_q, "u": S.dev.download(S.U1, (S.n_u, dim)),
           "p": S.dev.download(S.P, (S.n_q,)), "its_pressure": np.asarray(S.its["pressure"]), "t": t,
           "imported_package": np.asarray("oasisx_amd" in sys.modules), "imported_torch": np.asarray("torch" in sys.modules),
           "compressed": np.asarray(repr(S.compressed)),
           "window_blocks": np.asarray(0 if S.window_info is None else int(S.window_info.n_wblocks)),
           "window_max": np.asarray(0 if S.window_info is None else int(S.window_info.w_max))}
    S.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=2)
    ap.add_argument("-N", type=int, default=8)
    ap.add_argument("--udeg", type=int, default=2)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--out", default=None)
    ap.add_argument("--compress", action="store_true", help="value dictionaries and the pair-slot stream for M, K, Ap")
    ap.add_argument("--cg-merged", action="store_true", help="OX_KSP_CG_MERGED for the pressure solve")
    ap.add_argument("--windows", action="store_true",
                    help="brick order of the velocity numbering (ox_space_create_ordered) and the LDS-window stream of its "
                         "pattern (ox_space_windows, ox_window_retile): M, K, A multiply through k_spmv_win")
    a = ap.parse_args()
    r = run(a.dim, a.N, a.udeg, a.steps, compress=a.compress, cg_merged=a.cg_merged, windows=a.windows)
    ex = [f(r["x_v"].T, r["t"]) for f in tg(a.dim, 0.01)[0]]
    print("C-ABI step: n_u", r["x_v"].shape[0], "n_p", r["x_q"].shape[0], "max |u - u_exact| =",
          float(max(np.abs(r["u"][:, i] - ex[i]).max() for i in range(a.dim))), "pressure iterations", r["its_pressure"])
    if a.out:
        np.savez(a.out, **r)
