#!/usr/bin/env python3
"""IPCS time steps on a MESH-PARTITIONED problem through the C ABI alone: one process per rank, numpy arrays +
ctypes + include/oasisx_hip.h.  Nothing of the ``oasisx_amd`` package is imported (no torch, no fem.py, no
parallel.py, no fracstep.py): the partition, the halo plans and the transport are written here in numpy and
``multiprocessing`` pipes -- what a binding in another language would do with MPI.

Per rank (reference src/oasisx/fracstep.py on a distributed DOLFINx mesh, :186-216, :411-696):
``ox_mesh_create_sub`` -> ``ox_space_create_part`` (P2 velocity, P1 pressure) -> ``ox_dist_create_custom`` (the
halo plan of each space on a caller-supplied transport: the two callbacks below stand where ncclSend/ncclRecv and
ncclAllReduce are in an RCCL job) -> the operators on the owned rows -> every phase of the time step with the plans
handed to the solvers, mat-vecs and reductions.  The ranks may share one GPU (a rehearsal) or have one each.

The parent process runs the same steps on the whole mesh with one rank (demo/cabi_ipcs_step.py) and compares the
fields through the dof coordinates.

    python demo/cabi_partitioned_step.py --dim 3 -N 6 --parts 2 --steps 2
"""
from __future__ import annotations

import argparse
import ctypes as C
import importlib.util
import itertools
import json
import math
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_single():
    spec = importlib.util.spec_from_file_location("cabi_ipcs_step", os.path.join(ROOT, "demo", "cabi_ipcs_step.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ------------------------------------------------------------------------------------------------------------
# the partition, in numpy: cells cut into slabs by their centroids, a dof owned by the lowest rank among its
# cells, a rank keeps every cell that touches a dof it owns (one ghost layer)
# ------------------------------------------------------------------------------------------------------------
class Partition:
    def __init__(self, coords, cells, P):
        cells = np.asarray(cells, dtype=np.int64)
        self.coords, self.cells, self.P = coords, cells, P
        nv, nc, d = coords.shape[0], cells.shape[0], coords.shape[1]
        self.nv, self.nc, self.d = nv, nc, d
        cen = coords[cells].mean(axis=1)
        order = np.lexsort((cen[:, 1], cen[:, 0], cen[:, d - 1]))
        self.cell_rank = np.empty(nc, dtype=np.int64)
        self.cell_rank[order] = np.arange(nc) * P // nc
        self.vown = np.full(nv, P, dtype=np.int64)
        np.minimum.at(self.vown, cells.ravel(), np.repeat(self.cell_rank, d + 1))
        pairs = list(itertools.combinations(range(d + 1), 2))
        ea, eb = np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs])
        a, b = cells[:, ea], cells[:, eb]
        ukeys, inv = np.unique((np.minimum(a, b) * nv + np.maximum(a, b)).ravel(), return_inverse=True)
        self.cell_edges = inv.reshape(nc, -1)
        self.ne = ukeys.shape[0]
        self.eown = np.full(self.ne, P, dtype=np.int64)
        np.minimum.at(self.eown, self.cell_edges.ravel(), np.repeat(self.cell_rank, len(pairs)))

    def local_cells(self, r):
        """One cell set for BOTH spaces (they share the sub-mesh): every cell with a vertex or an edge of rank r."""
        t = (self.vown[self.cells] == r).any(axis=1) | (self.eown[self.cell_edges] == r).any(axis=1)
        return np.nonzero(t)[0]

    def local_dofs(self, r, degree):
        """(global ids, owners) of rank r's local dofs in the library's INITIAL order: the sub-mesh's vertices in
        ascending global id, then (degree 2) its edges in ascending global edge id."""
        lc = self.local_cells(r)
        verts = np.unique(self.cells[lc])
        gid, own = verts, self.vown[verts]
        if degree == 2:
            le = np.unique(self.cell_edges[lc])
            gid, own = np.concatenate([gid, self.nv + le]), np.concatenate([own, self.eown[le]])
        return gid, own

    def n_global(self, degree):
        return self.nv + (self.ne if degree == 2 else 0)

    def halo_plan(self, r, degree, local_of_gid):
        """peers, send_off, send_idx (local dofs, per peer in THE PEER'S ghost order), recv_off of rank r.
        A rank's ghost block is ordered by (owner, global id): ox_space_create_part's numbering."""
        ghosts = {}
        for q in range(self.P):
            gid, own = self.local_dofs(q, degree)
            g = own != q
            o = np.lexsort((gid[g], own[g]))
            ghosts[q] = (gid[g][o], own[g][o])
        mine_g, mine_o = ghosts[r]
        peers = sorted(set(mine_o.tolist()) | {q for q in range(self.P) if q != r and (ghosts[q][1] == r).any()})
        send_off, recv_off, send_idx = [0], [0], []
        for q in peers:
            want = ghosts[q][0][ghosts[q][1] == r]  # ascending global id = q's order of my block
            send_idx.append(local_of_gid(want))
            send_off.append(send_off[-1] + want.shape[0])
            recv_off.append(recv_off[-1] + int((mine_o == q).sum()))
        idx = np.concatenate(send_idx) if send_idx else np.zeros(0, dtype=np.int64)
        return (np.asarray(peers, dtype=np.int32), np.asarray(send_off, dtype=np.int64), idx.astype(np.int32),
                np.asarray(recv_off, dtype=np.int64))


# ------------------------------------------------------------------------------------------------------------
# the transport: pipes between the rank processes; pairwise exchanges in ascending peer order (no cycle can wait)
# ------------------------------------------------------------------------------------------------------------
class Transport:
    def __init__(self, rank, nranks, pipes):
        self.rank, self.nranks, self.pipes = rank, nranks, pipes  # pipes[q]: duplex connection to rank q

    def exchange(self, q, payload: bytes) -> bytes:
        c = self.pipes[q]
        if self.rank < q:
            c.send_bytes(payload)
            return c.recv_bytes()
        got = c.recv_bytes()
        c.send_bytes(payload)
        return got

    def allreduce(self, v: np.ndarray) -> np.ndarray:
        """Sum in rank order on every rank: the same bits everywhere."""
        parts = {self.rank: v}
        for q in range(self.nranks):
            if q != self.rank:
                parts[q] = np.frombuffer(self.exchange(q, v.tobytes()), dtype=np.float64)
        out = np.zeros_like(v)
        for q in range(self.nranks):
            out = out + parts[q]
        return out


class Plan:
    """ox_dist of one space on the pipe transport."""

    def __init__(self, L, lib, dev, tr, peers, send_off, send_idx, recv_off, n_owned, n_ghost):
        self.L, self.lib, self.tr = L, lib, tr
        self.peers, self.send_off, self.recv_off = peers, send_off, recv_off
        HALO = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int)
        ARED = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)
        self.calls = {"halo": 0, "allreduce": 0}

        def halo(_user, send_dev, ghost_dev, ncomp):
            try:
                self.calls["halo"] += 1
                ns, ng = int(send_off[-1]) * ncomp, int(recv_off[-1]) * ncomp
                send = np.zeros(max(ns, 1))
                if ns:
                    lib.ox_memcpy(send.ctypes.data_as(C.c_void_p), C.c_void_p(send_dev), ns * 8, 0, None)
                ghost = np.zeros(max(ng, 1))
                for k, q in enumerate(peers.tolist()):
                    got = np.frombuffer(tr.exchange(q, send[send_off[k] * ncomp: send_off[k + 1] * ncomp].tobytes()), dtype=np.float64)
                    assert got.shape[0] == (recv_off[k + 1] - recv_off[k]) * ncomp
                    ghost[recv_off[k] * ncomp: recv_off[k + 1] * ncomp] = got
                if ng:
                    lib.ox_memcpy(C.c_void_p(ghost_dev), ghost.ctypes.data_as(C.c_void_p), ng * 8, 1, None)
                return 0
            except Exception as exc:  # never let an exception cross the C ABI
                print(f"[rank {tr.rank}] halo callback: {exc!r}", file=sys.stderr)
                return -1

        def allreduce(_user, buf_dev, n):
            try:
                self.calls["allreduce"] += 1
                v = np.zeros(n)
                lib.ox_memcpy(v.ctypes.data_as(C.c_void_p), C.c_void_p(buf_dev), n * 8, 0, None)
                v = np.ascontiguousarray(tr.allreduce(v))
                lib.ox_memcpy(C.c_void_p(buf_dev), v.ctypes.data_as(C.c_void_p), n * 8, 1, None)
                return 0
            except Exception as exc:
                print(f"[rank {tr.rank}] allreduce callback: {exc!r}", file=sys.stderr)
                return -1

        self._cbs = (HALO(halo), ARED(allreduce))  # kept alive with the plan
        self.send_idx_dev = dev.upload(send_idx)
        self.handle = C.c_void_p()
        L.check(lib.ox_dist_create_custom(tr.rank, tr.nranks, int(peers.shape[0]), peers.ctypes.data_as(C.POINTER(C.c_int32)),
                                          send_off.ctypes.data_as(C.POINTER(C.c_int64)), self.send_idx_dev,
                                          recv_off.ctypes.data_as(C.POINTER(C.c_int64)), int(n_owned), int(n_ghost),
                                          C.cast(self._cbs[0], C.c_void_p), C.cast(self._cbs[1], C.c_void_p), None,
                                          C.byref(self.handle)), "ox_dist_create_custom")


# ------------------------------------------------------------------------------------------------------------
# one rank of the partitioned solver
# ------------------------------------------------------------------------------------------------------------
class RankIPCS:
    """FractionalStep_AB_CN on one rank's part: Dirichlet velocity data on the whole boundary, no pressure
    condition, low_memory_version=True, Jacobi-BiCGStab / Jacobi-CG -- every operation a C-ABI call."""

    def __init__(self, single, part: Partition, rank, tr: Transport, rtol, frame):
        self.L, self.lib = L, lib = single.load_binding()
        self.dev = dev = single.Device(L, lib)
        self.part, self.rank, self.tr, self.rtol = part, rank, tr, rtol
        ck, d = L.check, part.d
        self.gdim = d
        lc = part.local_cells(rank)
        verts = np.unique(part.cells[lc])
        cl = np.ascontiguousarray(np.searchsorted(verts, part.cells[lc]), dtype=np.int32)
        xs = np.ascontiguousarray(part.coords[verts])
        lo3, sp3 = (C.c_double * 3)(0, 0, 0), (C.c_double * 3)(1, 1, 1)
        lo, span = part.coords.min(axis=0), part.coords.max(axis=0) - part.coords.min(axis=0)
        for k in range(d):
            lo3[k], sp3[k] = float(lo[k]), float(span[k])
        self.mesh = C.c_void_p()
        ck(lib.ox_mesh_create_sub(xs.ctypes.data_as(C.c_void_p), verts.shape[0], cl.ctypes.data_as(C.c_void_p), lc.shape[0], d, 0,
                                  lo3, sp3, frame["lattice"], frame["tile_bits"], part.nc, C.byref(self.mesh)), "ox_mesh_create_sub")
        self.mv = L.ox_mesh_info()
        ck(lib.ox_mesh_view(self.mesh, C.byref(self.mv)), "ox_mesh_view")
        self.V, self.vv, self.gid_v, self.du = self._space(2)
        self.Q, self.qv, self.gid_q, self.dq = self._space(1)
        self.n_u, self.n_q = int(self.vv.n_dofs), int(self.qv.n_dofs)  # local: owned + ghost
        self.no_u, self.no_q = int(self.vv.pattern.sell.n_rows), int(self.qv.pattern.sell.n_rows)
        self.x_v = dev.download(self.vv.x, (self.n_u, d))
        self.x_q = dev.download(self.qv.x, (self.n_q, d))
        self.M, self.K, self.A = (self._matrix(self.vv.pattern) for _ in range(3))
        self.Ap = self._matrix(self.qv.pattern)
        self._assemble(0, self.vv, self.M)
        self._assemble(1, self.vv, self.K)
        self._assemble(1, self.qv, self.Ap)
        self.wq = dev.zeros(self.n_q)
        ck(lib.ox_assemble_weights(1, C.byref(self.mv.cells_struct), C.byref(self.qv.adj), self.no_q, self.wq, None),
           "ox_assemble_weights")
        vol = dev.upload(np.array([dev.download(self.wq, (self.no_q,)).sum()]))
        ck(lib.ox_allreduce_sum(self.dq.handle, vol, 1, None), "ox_allreduce_sum")  # assemble_scalar(1*dx) + allreduce (:581-584)
        self.vol = float(dev.download(vol, (1,))[0])
        nvec = self.n_u * d
        (self.U, self.U1, self.U2, self.UAB, self.RHS1, self.B0, self.BFIRST, self.B3) = (dev.zeros(nvec) for _ in range(8))
        self.PS, self.P, self.DP, self.B2 = (dev.zeros(self.n_q) for _ in range(4))
        glo, ghi = part.coords.min(axis=0), part.coords.max(axis=0)
        on = np.zeros(self.no_u, dtype=bool)  # Dirichlet rows: the OWNED dofs on the boundary of the whole domain
        for k in range(d):
            on |= np.isclose(self.x_v[: self.no_u, k], glo[k]) | np.isclose(self.x_v[: self.no_u, k], ghi[k])
        self.bc_dofs = np.nonzero(on)[0].astype(np.int32)
        self.bc_dofs_dev = dev.upload(self.bc_dofs)
        self.g_dev = dev.zeros(max(self.bc_dofs.shape[0], 1))
        self.dinvA, self.dinvM, self.dinvP = dev.zeros(self.n_u), dev.zeros(self.n_u), dev.zeros(self.n_q)
        ck(lib.ox_jacobi_setup(C.byref(self.M), self.dinvM, None), "ox_jacobi_setup")
        ck(lib.ox_jacobi_setup(C.byref(self.Ap), self.dinvP, None), "ox_jacobi_setup")
        wb = max(lib.ox_ksp_work_bytes_for(C.byref(self.A), d, L.KSP_BCGS), lib.ox_ksp_work_bytes_for(C.byref(self.M), d, L.KSP_CG),
                 lib.ox_ksp_work_bytes_for(C.byref(self.Ap), 1, L.KSP_CG))
        self.work_bytes = int(wb)
        self.work = dev.zeros(wb, np.uint8)
        self.its = {}

    def _space(self, degree):
        L, lib, dev, part, r = self.L, self.lib, self.dev, self.part, self.rank
        gid, own = part.local_dofs(r, degree)
        owner32 = np.ascontiguousarray(own, dtype=np.int32)
        s = C.c_void_p()
        L.check(lib.ox_space_create_part(self.mesh, degree, 0, dev.upload(owner32), owner32.shape[0], r, part.n_global(degree),
                                         C.byref(s)), "ox_space_create_part")
        v = L.ox_space_info()
        L.check(lib.ox_space_view(s, C.byref(v)), "ox_space_view")
        n_loc, n_own = int(v.n_dofs), int(v.pattern.sell.n_rows)
        assert n_loc == gid.shape[0] and n_own == int((own == r).sum())
        rank_initial = dev.download(v.rank_initial, (n_loc,), np.int32)  # initial id -> local dof
        gid_of_local = np.empty(n_loc, dtype=np.int64)
        gid_of_local[rank_initial] = gid
        order = np.argsort(gid)
        sorted_gid, local_sorted = gid[order], rank_initial[order]

        def local_of_gid(g):
            pos = np.searchsorted(sorted_gid, g)
            assert (sorted_gid[pos] == g).all()
            return local_sorted[pos]

        peers, send_off, send_idx, recv_off = part.halo_plan(r, degree, local_of_gid)
        assert (send_idx < n_own).all() and int(recv_off[-1]) == n_loc - n_own
        plan = Plan(L, lib, dev, self.tr, peers, send_off, send_idx, recv_off, n_own, n_loc - n_own)
        return s, v, gid_of_local, plan

    def _matrix(self, pat):
        A = self.L.ox_sell()
        C.memmove(C.byref(A), C.byref(pat.sell), C.sizeof(A))
        A.vals = self.dev.zeros(int(pat.size)).value
        return A

    def _assemble(self, kind, sv, A):
        p = sv.pattern
        self.L.check(self.lib.ox_assemble_matrix(kind, sv.degree, C.byref(self.mv.cells_struct), sv.cell_dofs, C.byref(sv.adj),
                                                 sv.adj_pos, sv.pw, C.byref(A), p.n_bins,
                                                 C.cast(p.bin_ptr_host, C.POINTER(C.c_int64)), p.bin_slices,
                                                 C.cast(p.bin_width_host, C.POINTER(C.c_int32)), None), "ox_assemble_matrix")

    def set_field(self, dev_ptr, values):
        self.dev.set(dev_ptr, np.ascontiguousarray(values, dtype=np.float64))

    def _solve(self, kind, A, dinv, b, x, ncomp, plan, name):
        res = self.L.ox_ksp_result()
        self.L.check(self.lib.ox_ksp_solve(kind, C.byref(A), dinv, b, x, ncomp, self.rtol, 1e-30, 10000, 0, 4, 0, self.work,
                                           self.work_bytes, C.byref(res), plan.handle, None), "ox_ksp_solve")
        self.L.check(self.lib.ox_halo_forward(plan.handle, x, ncomp, None), "x.scatter_forward")  # reference ksp.py:77
        reasons = [int(res.reason[c]) for c in range(ncomp)]
        assert all(r > 0 for r in reasons), (name, reasons)
        self.its[name] = [int(res.its[c]) for c in range(ncomp)]

    def step(self, dt, nu, g):
        """One time step (reference fracstep.py:660-696); g: (gdim, n_bc) Dirichlet values of the owned rows."""
        lib, ck, d = self.lib, self.L.check, self.gdim
        vv, qv, cs = self.vv, self.qv, C.byref(self.mv.cells_struct)
        n, nq, nvec, no, nqo = self.n_u, self.n_q, self.n_u * d, self.no_u, self.no_q
        pv = vv.pattern
        ck(lib.ox_axpby(nq, 1.0, self.P, 0.0, None, self.PS, None), "ps = p")
        ck(lib.ox_axpby(nvec, 1.5, self.U1, -0.5, self.U2, self.UAB, None), "u_ab")  # :432-434 (ghosts included)
        ck(lib.ox_assemble_first(vv.degree, cs, vv.cell_dofs, C.byref(vv.adj), vv.adj_pos, vv.pw, C.byref(self.A),
                                 C.byref(self.M), C.byref(self.K), self.UAB, self.U1, self.B0, self.BFIRST, dt, nu,
                                 pv.n_bins, C.cast(pv.bin_ptr_host, C.POINTER(C.c_int64)), pv.bin_slices,
                                 C.cast(pv.bin_width_host, C.POINTER(C.c_int32)), None), "ox_assemble_first")  # :435-469
        ck(lib.ox_zero_rows(C.byref(self.A), self.bc_dofs_dev, self.bc_dofs.shape[0], 1.0, None), "ox_zero_rows")  # :470-472
        ck(lib.ox_assemble_grad_vector(0, vv.degree, 1, cs, qv.cell_dofs, C.byref(vv.adj), no, self.PS, self.BFIRST, 1.0,
                                       self.RHS1, None), "rhs1")  # :487-506
        for c in range(d):  # bc.apply(rhs1[i]) (:517-518)
            if self.bc_dofs.shape[0]:
                self.dev.set(self.g_dev, g[c])
            ck(lib.ox_set_bc(self.RHS1, self.bc_dofs_dev, self.g_dev, self.bc_dofs.shape[0], d, c, None), "ox_set_bc")
        ck(lib.ox_jacobi_setup(C.byref(self.A), self.dinvA, None), "ox_jacobi_setup")
        self._solve(self.L.KSP_BCGS, self.A, self.dinvA, self.RHS1, self.U, d, self.du, "tentative")  # :521
        ck(lib.ox_assemble_div_vector(1, vv.degree, cs, vv.cell_dofs, C.byref(qv.adj), nqo, self.U, -1.0 / dt, self.B2, None),
           "b2")  # :538-546
        ck(lib.ox_remove_mean(nqo, nqo, self.B2, None, float(self.part.n_global(1)), self.dq.handle, None),
           "nullspace.remove")  # :573-574
        self._solve(self.L.KSP_CG, self.Ap, self.dinvP, self.B2, self.DP, 1, self.dq, "pressure")  # :578
        ck(lib.ox_remove_mean(nqo, nq, self.DP, self.wq, self.vol, self.dq.handle, None), "mean shift")  # :579-591
        ck(lib.ox_axpby(nq, 1.0, self.P, 1.0, self.DP, self.PS, None), "ps = p + dp")  # :604
        ck(lib.ox_spmv(C.byref(self.M), self.U, self.B3, d, self.du.handle, None), "M u")  # :615
        ck(lib.ox_assemble_grad_vector(1, vv.degree, 1, cs, qv.cell_dofs, C.byref(vv.adj), no, self.DP, self.B3, -dt, self.B3,
                                       None), "b3")  # :618-622
        self._solve(self.L.KSP_CG, self.M, self.dinvM, self.B3, self.U, d, self.du, "update")  # :634
        ck(lib.ox_axpby(nvec, 1.0, self.U1, 0.0, None, self.U2, None), "u2 = u1")  # :689-693
        ck(lib.ox_axpby(nvec, 1.0, self.U, 0.0, None, self.U1, None), "u1 = u")
        ck(lib.ox_axpby(nq, 1.0, self.PS, 0.0, None, self.P, None), "p = ps")
        ck(lib.ox_synchronize(None), "ox_synchronize")
        for plan in (self.du, self.dq):
            ck(lib.ox_dist_status(plan.handle), "ox_dist_status")


def rank_main(rank, nranks, pipes, args, frame, out_q):
    """Body of one rank process."""
    try:
        single = load_single()
        coords, cells = single.box_mesh(args["dim"], args["N"])
        part = Partition(coords, cells, nranks)
        tr = Transport(rank, nranks, pipes)
        S = RankIPCS(single, part, rank, tr, args["rtol"], frame)
        d, nu, dt = args["dim"], args["nu"], args["dt"]
        ue, pe = single.tg(d, nu)
        xv, xq = np.zeros((3, S.n_u)), np.zeros((3, S.n_q))
        xv[:d], xq[:d] = S.x_v.T, S.x_q.T
        field = lambda t: np.stack([f(xv, t) for f in ue], axis=1)  # noqa: E731
        S.set_field(S.U2, field(-dt))
        S.set_field(S.U1, field(0.0))
        S.set_field(S.U, field(0.0))
        S.set_field(S.P, pe(xq, -dt / 2))
        xb = np.zeros((3, S.bc_dofs.shape[0]))
        xb[:d] = S.x_v[S.bc_dofs].T
        t = 0.0
        its = []
        for _ in range(args["steps"]):
            t += dt
            S.step(dt, nu, [f(xb, t) for f in ue])
            its.append({k: list(v) for k, v in S.its.items()})
        U = S.dev.download(S.U, (S.n_u, d))
        Pv = S.dev.download(S.P, (S.n_q,))
        # ghosts agree with their owners' values after the last scatter_forward: shipped for the parent to check
        out_q.put({"rank": rank, "ok": True, "x_v": S.x_v, "x_q": S.x_q, "U": U, "P": Pv, "no_u": S.no_u, "no_q": S.no_q,
                   "its": its, "halo_calls": S.du.calls["halo"] + S.dq.calls["halo"],
                   "allreduce_calls": S.du.calls["allreduce"] + S.dq.calls["allreduce"],
                   "imported_package": any(m == "oasisx_amd" or m.startswith("oasisx_amd.") for m in sys.modules),
                   "imported_torch": "torch" in sys.modules})
    except BaseException as exc:  # noqa: BLE001 -- reported to the parent, which fails the run
        import traceback

        out_q.put({"rank": rank, "ok": False, "error": f"{exc!r}\n{traceback.format_exc()}"})


def run(dim=3, N=6, parts=2, steps=2, nu=0.01, dt=0.005, rtol=1e-11, timeout=600.0):
    single = load_single()
    # ---- one rank on the whole mesh (the parent; also gives the numbering frame of the whole mesh)
    coords, cells = single.box_mesh(dim, N)
    ref = single.IPCS(coords, cells, 2, rtol=rtol)
    frame = {"lattice": int(ref.mv.lattice), "tile_bits": int(ref.mv.tile_bits)}
    ue, pe = single.tg(dim, nu)
    xv, xq = np.zeros((3, ref.n_u)), np.zeros((3, ref.n_q))
    xv[:dim], xq[:dim] = ref.x_v.T, ref.x_q.T
    field = lambda t: np.stack([f(xv, t) for f in ue], axis=1)  # noqa: E731
    ref.set_field(ref.U2, field(-dt))
    ref.set_field(ref.U1, field(0.0))
    ref.set_field(ref.U, field(0.0))
    ref.set_field(ref.P, pe(xq, -dt / 2))
    xb = np.zeros((3, ref.bc_dofs.shape[0]))
    xb[:dim] = ref.x_v[ref.bc_dofs].T
    t, ref_its = 0.0, []
    for _ in range(steps):
        t += dt
        ref.step(dt, nu, [f(xb, t) for f in ue])
        ref_its.append({k: list(v) for k, v in ref.its.items()})
    U_ref = ref.dev.download(ref.U, (ref.n_u, dim))
    P_ref = ref.dev.download(ref.P, (ref.n_q,))
    x_v_ref, x_q_ref = ref.x_v.copy(), ref.x_q.copy()
    ref.close()

    # ---- the ranks
    ctx = mp.get_context("spawn")  # a forked child must not inherit the parent's GPU context
    conns = {r: {} for r in range(parts)}
    for a in range(parts):
        for b in range(a + 1, parts):
            ca, cb = ctx.Pipe(duplex=True)
            conns[a][b], conns[b][a] = ca, cb
    out_q = ctx.Queue()
    args = {"dim": dim, "N": N, "steps": steps, "nu": nu, "dt": dt, "rtol": rtol}
    procs = [ctx.Process(target=rank_main, args=(r, parts, conns[r], args, frame, out_q)) for r in range(parts)]
    for p in procs:
        p.start()
    results = {}
    try:
        for _ in range(parts):
            res = out_q.get(timeout=timeout)
            if not res["ok"]:
                raise RuntimeError(f"rank {res['rank']} failed:\n{res['error']}")
            results[res["rank"]] = res
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()

    def key(x):
        q = np.round((x + 1.0) * 4096).astype(np.int64)
        k = q[:, 0]
        for j in range(1, x.shape[1]):
            k = k * (1 << 20) + q[:, j]
        return k

    kv, kq = key(x_v_ref), key(x_q_ref)
    ov, oq = np.argsort(kv), np.argsort(kq)
    report = {"dim": dim, "N": N, "parts": parts, "steps": steps, "ranks": [], "reference_iterations": ref_its}
    seen_v, seen_q = np.zeros(kv.shape[0], dtype=int), np.zeros(kq.shape[0], dtype=int)
    worst_u = worst_p = worst_gu = worst_gp = 0.0
    su, sp = np.abs(U_ref).max(), np.abs(P_ref).max()
    for r in range(parts):
        res = results[r]
        gv = ov[np.searchsorted(kv[ov], key(res["x_v"]))]
        gq = oq[np.searchsorted(kq[oq], key(res["x_q"]))]
        assert (kv[gv] == key(res["x_v"])).all() and (kq[gq] == key(res["x_q"])).all()
        no_u, no_q = res["no_u"], res["no_q"]
        seen_v[gv[:no_u]] += 1
        seen_q[gq[:no_q]] += 1
        worst_u = max(worst_u, float(np.abs(res["U"][:no_u] - U_ref[gv[:no_u]]).max() / su))
        worst_p = max(worst_p, float(np.abs(res["P"][:no_q] - P_ref[gq[:no_q]]).max() / sp))
        if res["U"].shape[0] > no_u:
            worst_gu = max(worst_gu, float(np.abs(res["U"][no_u:] - U_ref[gv[no_u:]]).max() / su))
        if res["P"].shape[0] > no_q:
            worst_gp = max(worst_gp, float(np.abs(res["P"][no_q:] - P_ref[gq[no_q:]]).max() / sp))
        report["ranks"].append({"rank": r, "owned_u": int(no_u), "ghost_u": int(res["U"].shape[0] - no_u), "owned_p": int(no_q),
                                "ghost_p": int(res["P"].shape[0] - no_q), "iterations": res["its"],
                                "halo_exchanges": res["halo_calls"], "all_reduces": res["allreduce_calls"],
                                "imported_package": res["imported_package"], "imported_torch": res["imported_torch"]})
    assert (seen_v == 1).all() and (seen_q == 1).all(), "the owned dofs of the ranks do not tile the spaces"
    report.update(max_rel_diff_u=worst_u, max_rel_diff_p=worst_p, max_rel_diff_ghost_u=worst_gu, max_rel_diff_ghost_p=worst_gp)
    return report


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("-N", type=int, default=6)
    ap.add_argument("--parts", type=int, default=2)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--rtol", type=float, default=1e-11)
    a = ap.parse_args()
    print(json.dumps(run(a.dim, a.N, a.parts, a.steps, rtol=a.rtol)))


if __name__ == "__main__":
    main()
