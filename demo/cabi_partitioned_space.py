#!/usr/bin/env python3
"""One rank's piece of a mesh-partitioned space through the C ABI alone: numpy arrays + ctypes +
include/oasisx_hip.h (``ox_mesh_create_sub``, ``ox_space_create_part``) -- what DOLFINx's ``functionspace`` gives
the reference on a distributed mesh (src/oasisx/fracstep.py:186-216).  Nothing of the ``oasisx_amd`` package is
imported (no torch, no fem.py, no parallel.py): the partition below -- cells cut in two by their centroids, a dof owned
by the lowest rank among its cells, a rank keeping every cell that touches a dof it owns -- is written in numpy.

For every rank the script builds the part's space and mass matrix and checks them against the whole mesh's
(``ox_mesh_create`` / ``ox_space_create``) through the dof coordinates: the owned rows hold the same entries, the
owned dofs of all ranks tile the space, the ghosts come after the owned dofs ordered by (owner, initial id).

    python demo/cabi_partitioned_space.py --dim 3 -N 5 --degree 2 --parts 3
"""
from __future__ import annotations

import argparse
import ctypes as C
import importlib.util
import itertools
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_binding():
    spec = importlib.util.spec_from_file_location("ox_binding", os.path.join(ROOT, "oasisx_amd", "_lib.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, mod.load()


def box_mesh(dim, N):
    ax = np.linspace(-1.0, 1.0, N + 1)
    if dim == 2:
        X, Y = np.meshgrid(ax, ax, indexing="ij")
        coords = np.stack([X.ravel(), Y.ravel()], axis=1)
        vid = lambda i, j: i * (N + 1) + j  # noqa: E731
        cells = []
        for i in range(N):
            for j in range(N):
                a, b, c, d = vid(i, j), vid(i + 1, j), vid(i, j + 1), vid(i + 1, j + 1)
                cells += [[a, b, d], [a, d, c]]
        return coords, np.asarray(cells, dtype=np.int64)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    vid = lambda i, j, k: (i * (N + 1) + j) * (N + 1) + k  # noqa: E731
    cells = []
    for i in range(N):
        for j in range(N):
            for k in range(N):
                v = [vid(i + a, j + b, k + c) for a in (0, 1) for b in (0, 1) for c in (0, 1)]
                for p in itertools.permutations(range(3)):  # Kuhn: 6 tetrahedra around the main diagonal
                    path, cur = [0], 0
                    for ax_ in p:
                        cur += (4, 2, 1)[ax_]
                        path.append(cur)
                    cells.append([v[q] for q in path])
    return coords, np.asarray(cells, dtype=np.int64)


class Dev:
    def __init__(self, L, lib):
        self.L, self.lib, self.live = L, lib, []

    def zeros(self, n, dtype=np.float64):
        p = C.c_void_p()
        nb = max(int(n) * np.dtype(dtype).itemsize, 8)
        self.L.check(self.lib.ox_malloc(nb, C.byref(p)), "ox_malloc")
        self.L.check(self.lib.ox_memset(p, 0, nb, None), "ox_memset")
        self.live.append(p)
        return p

    def upload(self, a):
        a = np.ascontiguousarray(a)
        p = self.zeros(max(a.size, 1), a.dtype)
        if a.size:
            self.L.check(self.lib.ox_memcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1, None), "ox_memcpy")
        return p

    def download(self, p, shape, dtype=np.float64):
        out = np.zeros(shape, dtype=dtype)
        if out.size:
            self.L.check(self.lib.ox_memcpy(out.ctypes.data_as(C.c_void_p), p, out.nbytes, 0, None), "ox_memcpy")
        return out


def mass_rows(L, lib, dev, mesh_view, space_view, degree):
    """Assemble the mass matrix on the space's pattern; return (x, n_rows, {row: {col: value}})."""
    pat = space_view.pattern
    A = L.ox_sell()
    C.memmove(C.byref(A), C.byref(pat.sell), C.sizeof(A))
    A.vals = dev.zeros(int(pat.size)).value
    L.check(lib.ox_assemble_matrix(0, degree, C.byref(mesh_view.cells_struct), space_view.cell_dofs, C.byref(space_view.adj),
                                   space_view.adj_pos, space_view.pw, C.byref(A), pat.n_bins,
                                   C.cast(pat.bin_ptr_host, C.POINTER(C.c_int64)), pat.bin_slices,
                                   C.cast(pat.bin_width_host, C.POINTER(C.c_int32)), None), "ox_assemble_matrix")
    L.check(lib.ox_synchronize(None), "ox_synchronize")
    n_rows, ns, size = int(pat.sell.n_rows), int(pat.sell.n_slices), int(pat.size)
    sp = dev.download(pat.sell.slice_ptr, (ns + 1,), np.int64)
    cols = dev.download(pat.sell.cols, (size,), np.int32)
    vals = dev.download(A.vals, (size,))
    rl = dev.download(pat.row_len, (n_rows,), np.int32)
    rows = {}
    for r in range(n_rows):
        s, l = r >> 6, r & 63
        k = np.arange(rl[r])
        e = sp[s] + (k // 2) * 128 + l * 2 + (k % 2)
        rows[r] = dict(zip(cols[e].tolist(), vals[e].tolist()))
    d = int(space_view.gdim)
    x = dev.download(space_view.x, (int(space_view.n_dofs), d))
    return x, n_rows, rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("-N", type=int, default=5)
    ap.add_argument("--degree", type=int, default=2)
    ap.add_argument("--parts", type=int, default=2)
    ap.add_argument("--split", default="slabs", choices=["slabs", "octants"],
                    help="slabs along the last axis (equal counts), or -- 8 parts in 3-D -- the 2 x 2 x 2 octants of the box "
                         "(what recursive coordinate bisection gives an 8-GPU job: peers that share a face, one edge line, one vertex)")
    args = ap.parse_args()
    L, lib = load_binding()
    dev = Dev(L, lib)
    ck = L.check
    d, deg, P = args.dim, args.degree, args.parts
    coords, cells = box_mesh(d, args.N)
    nv, nc = coords.shape[0], cells.shape[0]

    def key(x):
        q = np.round((x + 1.0) * 4096).astype(np.int64)
        k = q[:, 0]
        for j in range(1, x.shape[1]):
            k = k * (1 << 20) + q[:, j]
        return k

    # ---- the whole mesh ------------------------------------------------------------------------------------
    mesh = C.c_void_p()
    c32 = np.ascontiguousarray(cells, dtype=np.int32)
    ck(lib.ox_mesh_create(coords.ctypes.data_as(C.c_void_p), nv, c32.ctypes.data_as(C.c_void_p), nc, d, 0, -1, C.byref(mesh)),
       "ox_mesh_create")
    mv = L.ox_mesh_info()
    ck(lib.ox_mesh_view(mesh, C.byref(mv)), "ox_mesh_view")
    V = C.c_void_p()
    ck(lib.ox_space_create(mesh, deg, 0, C.byref(V)), "ox_space_create")
    vv = L.ox_space_info()
    ck(lib.ox_space_view(V, C.byref(vv)), "ox_space_view")
    xg, ng, rows_g = mass_rows(L, lib, dev, mv, vv, deg)
    kg = key(xg)
    og = np.argsort(kg)

    # ---- the partition, in numpy -----------------------------------------------------------------------------
    cen = coords[cells].mean(axis=1)
    order = np.lexsort((cen[:, 1], cen[:, 0], cen[:, d - 1]))  # slabs along the last axis, equal counts
    cell_rank = np.empty(nc, dtype=np.int64)
    cell_rank[order] = np.arange(nc) * P // nc
    if args.split == "octants":
        if d != 3 or P != 8:
            raise SystemExit("--split octants: 3-D, 8 parts")
        cell_rank = (4 * (cen[:, 2] > 0) + 2 * (cen[:, 1] > 0) + (cen[:, 0] > 0)).astype(np.int64)
    vown = np.full(nv, P, dtype=np.int64)
    np.minimum.at(vown, cells.ravel(), np.repeat(cell_rank, d + 1))
    pairs = list(itertools.combinations(range(d + 1), 2))
    ea, eb = np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs])
    a, b = cells[:, ea], cells[:, eb]
    ekey = np.minimum(a, b) * nv + np.maximum(a, b)
    ukeys, inv = np.unique(ekey.ravel(), return_inverse=True)
    cell_edges = inv.reshape(nc, -1)
    eown = np.full(ukeys.shape[0], P, dtype=np.int64)
    np.minimum.at(eown, cell_edges.ravel(), np.repeat(cell_rank, len(pairs)))

    lo = coords.min(axis=0)
    span = coords.max(axis=0) - lo
    lo3, sp3 = (C.c_double * 3)(0, 0, 0), (C.c_double * 3)(1, 1, 1)
    for k in range(d):
        lo3[k], sp3[k] = float(lo[k]), float(span[k])
    report = {"ranks": [], "n_global": int(ng)}
    owned_global = []
    for r in range(P):
        touches = (vown[cells] == r).any(axis=1)
        if deg == 2:
            touches |= (eown[cell_edges] == r).any(axis=1)
        lc = np.nonzero(touches)[0]
        verts = np.unique(cells[lc])
        cl = np.searchsorted(verts, cells[lc]).astype(np.int32)
        owner = vown[verts]
        if deg == 2:
            le = np.unique(cell_edges[lc])  # ascending global key = ascending local (min, max) pair
            owner = np.concatenate([owner, eown[le]])
        owner32 = np.ascontiguousarray(owner, dtype=np.int32)
        sub = C.c_void_p()
        xs = np.ascontiguousarray(coords[verts])
        ck(lib.ox_mesh_create_sub(xs.ctypes.data_as(C.c_void_p), verts.shape[0], cl.ctypes.data_as(C.c_void_p), lc.shape[0], d, 0,
                                  lo3, sp3, int(mv.lattice), int(mv.tile_bits), nc, C.byref(sub)), "ox_mesh_create_sub")
        sv = L.ox_mesh_info()
        ck(lib.ox_mesh_view(sub, C.byref(sv)), "ox_mesh_view")
        W = C.c_void_p()
        ck(lib.ox_space_create_part(sub, deg, 0, dev.upload(owner32), owner32.shape[0], r, ng, C.byref(W)),
           "ox_space_create_part")
        wv = L.ox_space_info()
        ck(lib.ox_space_view(W, C.byref(wv)), "ox_space_view")
        xp, n_own, rows_p = mass_rows(L, lib, dev, sv, wv, deg)
        n_loc = xp.shape[0]
        assert n_own == int((owner == r).sum()) and n_loc == owner.shape[0]
        g_of = og[np.searchsorted(kg[og], key(xp))]  # global dof of every local dof, through the coordinates
        assert (kg[g_of] == key(xp)).all()
        # ghosts: behind the owned dofs, ordered by (owner, initial id) -- initial ids ascend with (vertex id | edge key)
        rank_initial = dev.download(wv.rank_initial, (n_loc,), np.int32)
        init_of = np.empty(n_loc, dtype=np.int64)
        init_of[rank_initial] = np.arange(n_loc)
        gh = init_of[n_own:]
        assert (owner[init_of[:n_own]] == r).all() and (owner[gh] != r).all()
        assert (np.diff(owner[gh] * n_loc + gh) > 0).all(), "ghost order"
        worst = 0.0
        for row in range(n_own):
            ref = rows_g[int(g_of[row])]
            got = {int(g_of[c]): v for c, v in rows_p[row].items()}
            assert set(got) == set(ref), (r, row)
            worst = max(worst, max(abs(got[c] - ref[c]) / max(abs(ref[c]), 1e-300) for c in ref))
        assert worst < 1e-12, worst
        owned_global.append(g_of[:n_own])
        report["ranks"].append({"rank": r, "cells": int(lc.shape[0]), "owned": int(n_own), "ghosts": int(n_loc - n_own),
                                "max_rel_diff_of_owned_mass_rows": worst})
        lib.ox_space_destroy(W)
        lib.ox_mesh_destroy(sub)
    allo = np.concatenate(owned_global)
    assert np.array_equal(np.sort(allo), np.arange(ng)), "the owned dofs of the ranks do not tile the space"
    report["imported_package"] = any(m == "oasisx_amd" or m.startswith("oasisx_amd.") for m in sys.modules)
    report["imported_torch"] = "torch" in sys.modules
    print(json.dumps(report))


if __name__ == "__main__":
    main()
