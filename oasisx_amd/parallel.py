"""Mesh partitioning and the RCCL communicator for runs on several GPUs of one node
(one process per GPU, launched by ``torch.distributed.run``).

The reference gets its parallelism from DOLFINx/PETSc over MPI (row-partitioned matrices,
owned + ghost vectors, ``scatter_forward`` -- reference fracstep.py:390,453,497,551,632,655;
ksp.py:77).  Here:

* every rank generates the (global) mesh and computes the SAME partition metadata (cell ->
  rank, vertex/edge -> owner rank); nothing is communicated at set-up;
* cells are dealt to ``nparts`` parts by recursive coordinate bisection of their centroids (equal
  counts, compact parts, any ``nparts``; ``OX_PARTITION=slabs`` restores the r01 z-slabs); a dof
  belongs to the lowest rank among the cells that contain it; a rank keeps every cell that touches one of
  its dofs (its own cells + one ghost layer), so every owned matrix row / vector entry is
  assembled locally and no matrix entries are ever communicated (PETSc's ``Mat.assemble``
  stash exchange disappears);
* per SpMV the owner sends its interface values straight into the neighbour's ghost block
  (``ox_halo_forward``); per Krylov synchronisation point one small all-reduce merges all dot
  products.  Two device transports sit behind those call sites: direct xGMI stores into
  IPC-mapped windows (``p2p``, two small kernels per exchange) and RCCL (grouped
  ncclSend/ncclRecv, ncclAllReduce).  Default: ``rccl`` when the job has an RCCL communicator (the
  conservative choice until the windows have run between two real GPUs); ``OX_TRANSPORT=auto`` enables
  p2p, self-tests it collectively and keeps RCCL when any rank fails; ``p2p`` / ``rccl`` / ``host``
  force one.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .fem import local_edges
from .mesh import Mesh


class Comm:
    """Rank/size plus (on GPUs) the RCCL communicator handle used by liboasisx_hip.so and the
    transport policy (see the module docstring)."""

    collective = True  # allreduce() really spans the job's ranks (stand-in communicators of tests do not say so)

    def __init__(self, rank=0, size=1, handle=None, transport=None):
        import os

        self.rank, self.size, self.handle = rank, size, handle
        # default: RCCL where the job has an RCCL communicator (the library transport, until the xGMI
        # windows have moved bytes between two real GPUs); a gloo job (rehearsals on one GPU) has none
        # and falls through "auto": the windows, self-tested, else the host-staged transport
        default = "rccl" if handle is not None else "auto"
        self.transport = (transport or os.environ.get("OX_TRANSPORT", default)).lower()
        if self.transport not in ("auto", "p2p", "rccl", "host"):
            raise ValueError(f"OX_TRANSPORT={self.transport!r}: expected auto, p2p, rccl or host")
        self.active = {}  # space degree -> transport that ended up serving its plan

    def _all_ok(self, ok: bool) -> bool:
        """Collective AND over the ranks."""
        import torch.distributed as dist

        t = torch.tensor([1 if ok else 0], dtype=torch.int32,
                         device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def enable_p2p(self, V, plan) -> bool:
        """Switch the halo plan ``plan`` of space ``V`` to the direct xGMI transport: create this
        rank's window, swap IPC handles and ghost offsets over torch.distributed, map the peers'
        windows, then self-test halo exchange and all-reduce.  Collective; returns the ranks'
        common verdict (False leaves the plan on its previous transport)."""
        import os

        import numpy as np
        import torch.distributed as dist

        lib = _lib.load()
        h = V.halo
        ng = V.n_local - V.n_owned
        win, handle = C.c_void_p(), C.create_string_buffer(64)
        ok = True
        try:
            nbytes = lib.ox_p2p_window_bytes(self.size, ng)
            _lib.check(lib.ox_p2p_window_create(nbytes, C.byref(win), handle), "ox_p2p_window_create")
        except _lib.OasisxHipError as e:
            ok, self.p2p_error = False, str(e)
        info = {"ok": ok, "handle": handle.raw, "n_ghost": ng,
                "recv_off": {int(q): int(h["recv_off"][i]) for i, q in enumerate(h["peers"])}}
        infos = [None] * self.size
        dist.all_gather_object(infos, info)
        wins = (C.c_void_p * self.size)()
        opened = []
        if all(i["ok"] for i in infos):
            for r in range(self.size):
                if r == self.rank:
                    wins[r] = win.value
                    continue
                w = C.c_void_p()
                try:
                    _lib.check(lib.ox_p2p_window_open(infos[r]["handle"], C.byref(w)), "ox_p2p_window_open")
                    wins[r] = w.value
                    opened.append(w)
                except _lib.OasisxHipError as e:
                    ok, self.p2p_error = False, str(e)
                    break
        else:
            ok = False
        if ok:
            try:
                peers = [int(q) for q in h["peers"]]
                off = np.asarray([infos[q]["recv_off"].get(self.rank, 0) for q in peers] or [0], dtype=np.int64)
                png = np.asarray([infos[q]["n_ghost"] for q in peers] or [0], dtype=np.int64)
                _lib.check(lib.ox_dist_enable_p2p(plan, win, wins, off.ctypes.data_as(C.POINTER(C.c_int64)),
                                                  png.ctypes.data_as(C.POINTER(C.c_int64)),
                                                  float(os.environ.get("OX_P2P_SELFTEST_TIMEOUT_S", "15"))),
                           "ox_dist_enable_p2p")  # the ranks are aligned here: a short bound suffices
            except _lib.OasisxHipError as e:
                ok, self.p2p_error = False, str(e)
        if not self._all_ok(ok):
            if ok:
                lib.ox_dist_disable_p2p(plan)  # frees the window and the mappings
            else:
                for w in opened:
                    lib.ox_p2p_window_close(w)
                if win.value:
                    lib.ox_p2p_window_free(win)
            return False
        # self-test (the plan now owns window and mappings)
        try:
            V.dist = plan
            V.check_halo()
            for k in range(3):  # both parities of the mailboxes
                buf = torch.ones(9, dtype=torch.float64, device=V.mesh.device)
                buf[0] = self.rank + 1.0 + k
                _lib.check(lib.ox_allreduce_sum(plan, _lib.ptr(buf), 9, _lib.current_stream()), "ox_allreduce_sum")
                torch.cuda.synchronize()
                _lib.check(lib.ox_dist_status(plan), "ox_dist_status")
                want = torch.full((9,), float(self.size), dtype=torch.float64)
                want[0] = self.size * (self.size + 1) / 2.0 + k * self.size
                if not torch.equal(buf.cpu(), want):
                    raise RuntimeError(f"p2p all-reduce self-test: got {buf.tolist()}")
            # stress: back-to-back exchanges with a payload that changes every round, checked on the
            # device -- a flag that overtakes its data, or a buffer reused too early, shows up here
            # and not as wrong physics later
            rounds = int(os.environ.get("OX_P2P_STRESS_ROUNDS", "200"))
            d = V.mesh.gdim
            bad = torch.zeros((), dtype=torch.int64, device=V.mesh.device)
            X = V.x.clone().contiguous()
            red = torch.empty(3, dtype=torch.float64, device=V.mesh.device)
            for k in range(rounds):
                f = 1.0 + 0.5 * k
                X[: V.n_owned] = V.x[: V.n_owned] * f
                X[V.n_owned:] = float("nan")
                _lib.check(lib.ox_halo_forward(plan, _lib.ptr(X), d, _lib.current_stream()), "ox_halo_forward")
                bad += (X[V.n_owned:] != V.x[V.n_owned:] * f).sum()
                red[0], red[1], red[2] = self.rank + f, 1.0, -f
                _lib.check(lib.ox_allreduce_sum(plan, _lib.ptr(red), 3, _lib.current_stream()), "ox_allreduce_sum")
                bad += (red[0] != self.size * (self.size - 1) / 2.0 + self.size * f).to(torch.int64)
                bad += (red[1] != float(self.size)).to(torch.int64) + (red[2] != -f * self.size).to(torch.int64)
            torch.cuda.synchronize()
            _lib.check(lib.ox_dist_status(plan), "ox_dist_status")
            if int(bad.item()) != 0:
                raise RuntimeError(f"p2p stress self-test: {int(bad.item())} wrong values in {rounds} rounds")
        except (RuntimeError, _lib.OasisxHipError) as e:
            ok, self.p2p_error = False, str(e)
        if not self._all_ok(ok):
            lib.ox_dist_disable_p2p(plan)
            return False
        # production bound: ranks may reach an exchange far apart (set-up work, host-side I/O)
        _lib.check(lib.ox_dist_p2p_timeout(plan, float(os.environ.get("OX_P2P_TIMEOUT_S", "120"))),
                   "ox_dist_p2p_timeout")
        return True

    def time_transports(self, V, reps: int = 200) -> dict:
        """Exchange micro-benchmark of the halo plan of ``V`` on every transport this job can bring up
        (collective; bench.py's N > 1 line): microseconds per halo exchange (one and gdim components) and
        per 9-value all-reduce, HIP-event timed, back to back.  The active plan is timed as it is; the
        other device transport (RCCL <-> xGMI windows) on a temporary plan."""
        import torch

        lib = _lib.load()
        if V.halo is None or V.dist is None:
            return {}
        dev = V.mesh.device
        d = V.mesh.gdim
        st = _lib.current_stream()

        def timed(plan):
            X = V.x.clone().contiguous()
            x1 = X[:, 0].contiguous()
            buf = torch.ones(9, dtype=torch.float64, device=dev)
            out = {}
            for name, call in (("halo_1comp_us", lambda: lib.ox_halo_forward(plan, _lib.ptr(x1), 1, st)),
                               (f"halo_{d}comp_us", lambda: lib.ox_halo_forward(plan, _lib.ptr(X), d, st)),
                               ("allreduce_9_us", lambda: lib.ox_allreduce_sum(plan, _lib.ptr(buf), 9, st))):
                for _ in range(10):
                    _lib.check(call(), name)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    buf.fill_(1.0)
                    _lib.check(call(), name)
                e1.record()
                torch.cuda.synchronize()
                out[name] = 1e3 * e0.elapsed_time(e1) / reps
            _lib.check(lib.ox_dist_status(plan), "ox_dist_status")
            return out

        active = self.active.get(V.degree, "?")
        res = {active: timed(V.dist)}
        h = V.halo
        other = "p2p" if active == "rccl" else ("rccl" if active == "p2p" and self.handle is not None else None)
        if other is not None:
            tmp = C.c_void_p()
            _lib.check(lib.ox_dist_create(self.handle, self.rank, self.size, int(h["peers"].shape[0]),
                                          h["peers"].ctypes.data_as(C.POINTER(C.c_int32)),
                                          h["send_off"].ctypes.data_as(C.POINTER(C.c_int64)), _lib.ptr(h["send_idx"]),
                                          h["recv_off"].ctypes.data_as(C.POINTER(C.c_int64)), V.n_owned,
                                          V.n_local - V.n_owned, C.byref(tmp)), "ox_dist_create")
            keep = V.dist
            try:
                if other == "rccl":
                    res["rccl"] = timed(tmp)
                elif self.enable_p2p(V, tmp):  # self-tested, collectively agreed
                    res["p2p"] = timed(tmp)
                else:
                    res["p2p"] = {"error": getattr(self, "p2p_error", "a peer rank failed")}
            finally:
                V.dist = keep
                lib.ox_dist_destroy(tmp)
        return res

    def info(self) -> dict:
        """What the library's RCCL communicator itself reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice);
        a job without one (gloo rehearsals, one rank) says so."""
        if self.handle is None:
            return {"rccl": False, "nranks": self.size, "rank": self.rank}
        n, r, d = C.c_int(0), C.c_int(0), C.c_int(0)
        _lib.check(_lib.load().ox_comm_info(self.handle, C.byref(n), C.byref(r), C.byref(d)), "ox_comm_info")
        return {"rccl": True, "nranks": int(n.value), "rank": int(r.value), "device": int(d.value)}

    def allreduce(self, v, op=None):
        import torch.distributed as dist

        if self.size == 1:
            return v
        t = torch.tensor([float(v)], dtype=torch.float64,
                         device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
        return float(t.item())

    def Barrier(self):
        import torch.distributed as dist

        if self.size > 1:
            dist.barrier()

    def make_transport(self, V):
        """Host-staged rehearsal transport over the job's (gloo) process group: the halo plan of
        space ``V`` as an ox_dist whose exchange points call back into Python.  For testing the
        partitioned path where RCCL cannot run (several ranks on ONE GPU); never the fast path."""
        import numpy as np
        import torch.distributed as dist

        lib = _lib.load()
        h = V.halo
        npeer = int(h["peers"].shape[0])
        ns, ng = int(h["send_off"][-1]), V.n_local - V.n_owned
        HALO = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int)
        ARED = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)

        def halo_cb(user, send_dev, ghost_dev, nc):
            try:
                sb = np.empty(max(ns * nc, 1))
                gb = np.empty(max(ng * nc, 1))
                if ns:
                    _lib.check(lib.ox_memcpy(sb.ctypes.data, send_dev, ns * nc * 8, 0, None), "ox_memcpy")
                reqs, recvs = [], []
                for i, q in enumerate(h["peers"]):
                    s0, s1 = int(h["send_off"][i]) * nc, int(h["send_off"][i + 1]) * nc
                    r0, r1 = int(h["recv_off"][i]) * nc, int(h["recv_off"][i + 1]) * nc
                    if s1 > s0:
                        reqs.append(dist.isend(torch.from_numpy(sb[s0:s1].copy()), int(q)))
                    if r1 > r0:
                        t = torch.empty(r1 - r0, dtype=torch.float64)
                        reqs.append(dist.irecv(t, int(q)))
                        recvs.append((t, r0, r1))
                for r in reqs:
                    r.wait()
                for t, r0, r1 in recvs:
                    gb[r0:r1] = t.numpy()
                if ng:
                    _lib.check(lib.ox_memcpy(ghost_dev, gb.ctypes.data, ng * nc * 8, 1, None), "ox_memcpy")
                return 0
            except Exception:  # noqa: BLE001 -- never unwind through the C frame
                import traceback

                traceback.print_exc()
                return 1

        def ared_cb(user, buf_dev, n):
            try:
                b = np.empty(n)
                _lib.check(lib.ox_memcpy(b.ctypes.data, buf_dev, n * 8, 0, None), "ox_memcpy")
                t = torch.from_numpy(b)
                dist.all_reduce(t)
                _lib.check(lib.ox_memcpy(buf_dev, b.ctypes.data, n * 8, 1, None), "ox_memcpy")
                return 0
            except Exception:  # noqa: BLE001
                import traceback

                traceback.print_exc()
                return 1

        cbs = (HALO(halo_cb), ARED(ared_cb))
        self._keep = getattr(self, "_keep", []) + [cbs]  # keep the thunks alive
        out = C.c_void_p()
        _lib.check(lib.ox_dist_create_custom(
            self.rank, self.size, npeer, h["peers"].ctypes.data_as(C.POINTER(C.c_int32)),
            h["send_off"].ctypes.data_as(C.POINTER(C.c_int64)), _lib.ptr(h["send_idx"]),
            h["recv_off"].ctypes.data_as(C.POINTER(C.c_int64)), V.n_owned, ng,
            C.cast(cbs[0], C.c_void_p), C.cast(cbs[1], C.c_void_p), None, C.byref(out)), "ox_dist_create_custom")
        return out


class SelfLoopComm(Comm):
    """ONE rank of a ``size``-rank job, alone on the device (tools/predict_scaling.py): partition, spaces, halo plans,
    operators and Krylov methods are those of rank ``rank`` in the real job -- interior / boundary slice lists, pack
    kernels, grouped ncclSend / ncclRecv, ncclAllReduce at every synchronisation point, the partitioned defaults of
    KSPSolver --, but the plans exchange with THIS rank itself through a one-rank RCCL communicator: every ghost entry
    receives some owned value, every all-reduce sums one contribution.  Kernel and call-site costs of the rank are real,
    its results are not those of the job (``collective`` is False: nothing here spans ranks).  Never a product path."""

    collective = False
    self_loop = True

    def __init__(self, rank: int, size: int, device_transport: str = "rccl"):
        """``device_transport``: "rccl" (the one-rank communicator) or "p2p" -- the plans push into and pull from THIS
        rank's own xGMI window: the push / pull kernels and the single-kernel all-reduce of the window transport, with
        no link under them."""
        if device_transport not in ("rccl", "p2p"):
            raise ValueError("SelfLoopComm: device_transport is 'rccl' or 'p2p'")
        self.device_transport = device_transport
        lib = _lib.load()
        buf = C.create_string_buffer(128)
        _lib.check(lib.ox_comm_unique_id(buf), "ox_comm_unique_id")
        handle = C.c_void_p()
        _lib.check(lib.ox_comm_create(buf.raw, 0, 1, C.byref(handle)), "ox_comm_create")
        super().__init__(rank, size, handle, transport="host")  # ("host": attach_comm asks make_transport for the plan)

    def make_transport(self, V):
        """The rank's halo plan folded onto itself: one peer (this rank), as many values sent as the real plan
        RECEIVES (the ghost block is filled completely, the pack kernel gathers as many entries as the real exchange
        moves in), taken from the real send lists in turn."""
        import numpy as np

        lib = _lib.load()
        h = V.halo
        ng = V.n_local - V.n_owned
        src = h["send_idx"]
        if ng and src.numel() == 0:
            src = torch.zeros(1, dtype=torch.int32, device=V.mesh.device)
        send_idx = src.repeat((ng + max(src.numel(), 1) - 1) // max(src.numel(), 1))[:ng].contiguous() if ng else src[:0]
        self._keep = getattr(self, "_keep", []) + [send_idx]
        peers = np.zeros(1 if ng else 0, dtype=np.int32)
        off = np.asarray([0, ng] if ng else [0], dtype=np.int64)
        out = C.c_void_p()
        _lib.check(lib.ox_dist_create(self.handle, 0, 1, int(peers.shape[0]), peers.ctypes.data_as(C.POINTER(C.c_int32)),
                                      off.ctypes.data_as(C.POINTER(C.c_int64)), _lib.ptr(send_idx),
                                      off.ctypes.data_as(C.POINTER(C.c_int64)), V.n_owned, ng, C.byref(out)), "ox_dist_create")
        if self.device_transport == "p2p":  # the plan's only peer is this rank: its own window is the remote one
            win, handle = C.c_void_p(), C.create_string_buffer(64)
            _lib.check(lib.ox_p2p_window_create(lib.ox_p2p_window_bytes(1, ng), C.byref(win), handle), "ox_p2p_window_create")
            wins = (C.c_void_p * 1)(win.value)
            zero, png = np.zeros(1, dtype=np.int64), np.asarray([ng], dtype=np.int64)
            _lib.check(lib.ox_dist_enable_p2p(out, win, wins, zero.ctypes.data_as(C.POINTER(C.c_int64)),
                                              png.ctypes.data_as(C.POINTER(C.c_int64)), 15.0), "ox_dist_enable_p2p")
            import os

            if os.environ.get("OX_P2P_RELEASE", "").lower() == "fast":  # (the library's default is the conservative form)
                _lib.check(lib.ox_dist_set_p2p_release(out, 0), "ox_dist_set_p2p_release")
        return out

    def allreduce(self, v, op=None):
        return v

    def _all_ok(self, ok: bool) -> bool:
        return ok

    def Barrier(self):
        return None


def init_comm() -> Comm:
    """Communicator of the current ``torch.distributed`` job.  With the nccl (= RCCL) backend the
    library's own ncclComm is created from a unique id broadcast over torch.distributed."""
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return Comm(0, 1, None)
    rank, size = dist.get_rank(), dist.get_world_size()
    if dist.get_backend() != "nccl":
        # gloo: partition logic on the CPU; on a GPU the plans use the xGMI windows (several ranks
        # may share one device there) or, with OX_TRANSPORT=host, the host-staged rehearsal transport
        return Comm(rank, size, None)
    lib = _lib.load()
    buf = C.create_string_buffer(128)
    if rank == 0:
        _lib.check(lib.ox_comm_unique_id(buf), "ox_comm_unique_id")
    t = torch.tensor(list(buf.raw), dtype=torch.uint8, device="cuda")
    dist.broadcast(t, src=0)
    raw = bytes(t.cpu().tolist())
    handle = C.c_void_p()
    _lib.check(lib.ox_comm_create(raw, rank, size, C.byref(handle)), "ox_comm_create")
    return Comm(rank, size, handle)


def slab_partition(cen: torch.Tensor, coords: torch.Tensor, nparts: int) -> torch.Tensor:
    """Cells cut into ``nparts`` slabs of equal counts in (z, y, x) order of the centroids."""
    nc, d = cen.shape
    lo = coords.min(dim=0).values
    span = (coords.max(dim=0).values - lo).clamp_min(1e-300)
    q = torch.round((cen - lo) / span * float(1 << 20)).to(torch.int64)
    key = q[:, d - 1]
    for k in range(d - 2, -1, -1):
        key = key * (1 << 21) + q[:, k]
    order = torch.argsort(key, stable=True)
    cell_rank = torch.empty(nc, dtype=torch.int64, device=cen.device)
    cell_rank[order] = torch.div(torch.arange(nc, device=cen.device) * nparts, nc, rounding_mode="floor")
    return cell_rank


def recursive_coordinate_bisection(cen: torch.Tensor, nparts: int) -> torch.Tensor:
    """Geometric graph-free partition of points (cell centroids) into ``nparts`` parts of equal
    counts: split the current set across its longest extent (ties: the highest axis, so box meshes
    are first cut in z as the r01 slabs were) at the weighted median -- parts proportional to the
    number of ranks on each side, so any ``nparts`` works -- and recurse.  The stand-in for the graph
    partitioner DOLFINx calls for the reference ("any graph partition computed on
    host"): compact parts (2 x 2 x 2 blocks on a cube at 8 ranks, a third of the slabs' interface
    area), no adjacency needed, O(n log n) device sorts.  Deterministic (stable sorts), so every
    rank derives the same map from the replicated mesh."""
    n = cen.shape[0]
    out = torch.zeros(n, dtype=torch.int64, device=cen.device)
    todo = [(torch.arange(n, device=cen.device), 0, int(nparts))]  # (cell ids, first rank, ranks)
    while todo:
        ids, r0, k = todo.pop()
        if k == 1 or ids.numel() == 0:
            out[ids] = r0
            continue
        x = cen[ids]
        ext = x.max(dim=0).values - x.min(dim=0).values
        axis = int(max(range(cen.shape[1]), key=lambda a: (float(ext[a]) * (1.0 + 1e-12 * a), a)))
        order = torch.argsort(x[:, axis], stable=True)
        k_lo = k // 2
        cut = (ids.numel() * k_lo) // k
        todo.append((ids[order[:cut]], r0, k_lo))
        todo.append((ids[order[cut:]], r0 + k_lo, k - k_lo))
    return out


class MeshPartition:
    """Partition metadata of a (global) mesh for one rank.

    Global (cheap, one pass over the cells): the cell -> rank map (recursive coordinate bisection: every rank
    derives the same map by deterministic stable sorts) and the vertex owners (lowest rank among the cells of a
    vertex).  Everything else is computed on this rank's WINDOW only -- its own cells plus two rings of
    vertex-neighbours: the edges (P2 dofs), their owners, the rank's local cells (those that touch an owned
    dof: own cells + one ghost layer) and, for the halo plan, the cells it shares with each peer.  Two rings
    are what exact owners need: a dof of a local cell lives in cells that share a vertex with that cell, and a
    local cell shares a vertex with an own cell.  (Rounds 1-2 enumerated the edges of the WHOLE mesh on every
    rank: 46 GiB and 2.9 s per rank at 256^3 x 8, not shrinking with the number of ranks.)

    Dof ids used between ranks ("global initial ids"): a vertex is its vertex id, an edge is ``num_vertices`` +
    its position among this window's sorted edge keys (min_vertex * num_vertices + max_vertex).  Positions
    differ from rank to rank, their ORDER does not -- and the halo plan only needs both sides to list the
    shared dofs in the same order (vertices by id, then edges by key)."""

    def __init__(self, mesh: Mesh, rank: int, nparts: int, comm=None, faces: bool = False):
        """``faces``: the job carries a degree-3 space on tetrahedra (one dof per face, owned by the lowest rank among the
        face's cells): the faces of the window are enumerated too, and a cell that touches nothing but an owned FACE dof
        is a local cell as well -- for EVERY space of the job (the velocity and pressure spaces share one cell set)."""
        self.mesh, self.rank, self.nparts = mesh, int(rank), int(nparts)
        self.with_faces = bool(faces) and mesh.gdim == 3
        self.cells_own_dofs = bool(faces) and mesh.gdim == 2  # (degree 3 on triangles: one dof inside every cell)
        self.comm = comm if comm is not None else getattr(mesh, "comm", None)
        dev = mesh.device
        d = mesh.gdim
        cells = mesh.cells
        nc, nverts = mesh.num_cells, mesh.num_vertices
        # ---- cells -> parts: recursive coordinate bisection of the centroids (OX_PARTITION=slabs: the
        #      r01 z-slabs) -- deterministic: every rank computes the same map ----------------------------
        import os

        cen = mesh.coords[cells].mean(dim=1)
        if os.environ.get("OX_PARTITION", "rcb").lower() == "slabs":
            cell_rank = slab_partition(cen, mesh.coords, nparts)
        else:
            cell_rank = recursive_coordinate_bisection(cen, nparts)
        del cen
        self.cell_rank = cell_rank
        # ---- vertex owners: lowest rank among the cells that contain the vertex (global) -----------------
        big = torch.full((nverts,), nparts, dtype=torch.int64, device=dev)
        self.vown = big.scatter_reduce(0, cells.reshape(-1), cell_rank.repeat_interleave(d + 1), reduce="amin")
        del big
        # ---- this rank's window: own cells + two rings of vertex-neighbours ------------------------------
        vm = torch.zeros(nverts, dtype=torch.bool, device=dev)
        vm[cells[cell_rank == self.rank].reshape(-1)] = True
        ring1 = vm[cells].any(dim=1)
        vm.zero_()
        vm[cells[ring1].reshape(-1)] = True
        self.win_cells = torch.nonzero(vm[cells].any(dim=1)).reshape(-1)  # ascending global cell ids
        del vm, ring1
        wc = cells[self.win_cells]
        wrank = cell_rank[self.win_cells]
        # ---- edges of the window, their owners ------------------------------------------------------
        ea = torch.tensor([e[0] for e in local_edges(d)], device=dev)
        eb = torch.tensor([e[1] for e in local_edges(d)], device=dev)
        a, b = wc[:, ea], wc[:, eb]
        ekey = torch.minimum(a, b) * nverts + torch.maximum(a, b)
        self.edge_keys, inv = torch.unique(ekey.reshape(-1), return_inverse=True)
        self._win_cell_edges = inv.reshape(wc.shape[0], -1)
        del a, b, ekey, inv
        ne = self._win_cell_edges.shape[1]
        bige = torch.full((int(self.edge_keys.shape[0]),), nparts, dtype=torch.int64, device=dev)
        # exact for every edge of a cell within one ring of an own cell (all its cells are in the window);
        # an edge of the outer ring only can come out too HIGH, never as this rank, and is not a local dof
        self.eown = bige.scatter_reduce(0, self._win_cell_edges.reshape(-1), wrank.repeat_interleave(ne), reduce="amin")
        del bige
        self._own2 = None
        self._own3 = None
        # ---- faces of the window (degree 3 on tetrahedra): keys of the sorted vertex triples, owners ------------
        self.face_keys = self.fown = self._win_cell_faces = None
        if self.with_faces:
            if float(nverts) ** 3 >= 2.0 ** 62:
                raise ValueError("MeshPartition: face keys overflow int64")
            tri = torch.stack([torch.sort(wc[:, [a for a in range(4) if a != f]], dim=1).values for f in range(4)], dim=1)
            fkey = (tri[:, :, 0] * nverts + tri[:, :, 1]) * nverts + tri[:, :, 2]
            self.face_keys, finv = torch.unique(fkey.reshape(-1), return_inverse=True)
            self._win_cell_faces = finv.reshape(wc.shape[0], 4)
            bigf = torch.full((int(self.face_keys.shape[0]),), nparts, dtype=torch.int64, device=dev)
            self.fown = bigf.scatter_reduce(0, self._win_cell_faces.reshape(-1), wrank.repeat_interleave(4), reduce="amin")
            del tri, fkey, finv, bigf
        # ---- local cells: those that touch a vertex or an edge (or, with_faces, a face) this rank owns -------------
        self._win_local = (self.vown[wc] == self.rank).any(dim=1) | (self.eown[self._win_cell_edges] == self.rank).any(dim=1)
        if self.with_faces:
            self._win_local |= (self.fown[self._win_cell_faces] == self.rank).any(dim=1)
        if self.cells_own_dofs:  # (an own cell all of whose vertices and edges belong to lower ranks still owns its cell dof)
            self._win_local |= wrank == self.rank
        self.local_cells = self.win_cells[self._win_local]
        self._n_edges_global = None

    # -- window look-ups ---------------------------------------------------------------------------------
    def _win_pos(self, cell_ids: torch.Tensor) -> torch.Tensor:
        pos = torch.searchsorted(self.win_cells, cell_ids)
        # a cell outside the window would silently read a neighbouring cell's edges (wrong owners, a wrong halo plan):
        # an error, not a debug assertion; one device round trip per call, set-up only
        if not bool((self.win_cells[pos.clamp_max(self.win_cells.shape[0] - 1)] == cell_ids).all()):
            raise ValueError("MeshPartition: a cell outside this rank's window was looked up")
        return pos

    def cell_edges_of(self, cell_ids: torch.Tensor) -> torch.Tensor:
        """Window edge indices (into ``edge_keys``) of the edges of the given (global) cells of the window."""
        return self._win_cell_edges[self._win_pos(cell_ids)]

    def cell_faces_of(self, cell_ids: torch.Tensor) -> torch.Tensor:
        """Window face indices (into ``face_keys``) of the four faces (face f = the vertices other than f) of the given
        (global) cells of the window."""
        return self._win_cell_faces[self._win_pos(cell_ids)]

    def owner0(self, degree: int) -> torch.Tensor:
        """Owner rank of every initial dof id of this rank's window: vertices, then window edges (degree 2), or -- degree
        3 -- two dofs per window edge, then one per window face (tetrahedra) or per window cell (triangles: the cell's
        own dof belongs to the cell's rank)."""
        if degree == 1:
            return self.vown
        if degree == 2:
            if self._own2 is None:
                self._own2 = torch.cat([self.vown, self.eown])
            return self._own2
        if self._own3 is None:
            last = self.fown if self.mesh.gdim == 3 else self.cell_rank[self.win_cells]
            if last is None:
                raise ValueError("MeshPartition: a degree-3 space on tetrahedra needs faces=True")
            self._own3 = torch.cat([self.vown, self.eown.repeat_interleave(2), last])
        return self._own3

    def cells_shared_with(self, q: int) -> torch.Tensor:
        """This rank's local cells (global ids) that also touch a dof owned by rank ``q``: exactly the cells both
        ranks keep, i.e. where ``q``'s ghosts owned by this rank live."""
        wc = self.mesh.cells[self.win_cells]
        hit = (self.vown[wc] == q).any(dim=1) | (self.eown[self._win_cell_edges] == q).any(dim=1)
        if self.with_faces:
            hit |= (self.fown[self._win_cell_faces] == q).any(dim=1)
        if self.cells_own_dofs:  # (degree 3 on triangles: the cell's own dof belongs to the cell's rank)
            hit |= self.cell_rank[self.win_cells] == q
        return self.win_cells[self._win_local & hit]

    def peers(self):
        """Ranks this rank shares a cell with."""
        wc = self.mesh.cells[self.local_cells]
        own = [self.vown[wc].reshape(-1), self.eown[self.cell_edges_of(self.local_cells)].reshape(-1)]
        if self.with_faces:
            own.append(self.fown[self.cell_faces_of(self.local_cells)].reshape(-1))
        if self.cells_own_dofs:
            own.append(self.cell_rank[self.local_cells])
        own = torch.cat(own)
        return [int(q) for q in torch.unique(own).tolist() if int(q) != self.rank and int(q) < self.nparts]

    def n_edges_global(self) -> int:
        """Edges of the whole mesh: every edge has exactly one owner, so the owned counts add up (one scalar
        all-reduce); without a communicator (single-process tools and tests) they are counted directly."""
        if self._n_edges_global is None:
            mine = int((self.eown == self.rank).sum().item())
            comm = self.comm
            if comm is not None and getattr(comm, "size", 1) == self.nparts and getattr(comm, "collective", False):
                self._n_edges_global = int(round(comm.allreduce(float(mine))))
            else:
                mesh, d = self.mesh, self.mesh.gdim
                nverts = mesh.num_vertices
                ea = torch.tensor([e[0] for e in local_edges(d)], device=mesh.device)
                eb = torch.tensor([e[1] for e in local_edges(d)], device=mesh.device)
                total, step = 0, 1 << 24
                seen = None
                for c0 in range(0, mesh.num_cells, step):  # chunked: a global unique in one piece is what this class avoids
                    cc = mesh.cells[c0:c0 + step]
                    a, b = cc[:, ea], cc[:, eb]
                    k = torch.unique((torch.minimum(a, b) * nverts + torch.maximum(a, b)).reshape(-1))
                    seen = k if seen is None else torch.unique(torch.cat([seen, k]))
                total = 0 if seen is None else int(seen.shape[0])
                self._n_edges_global = total
        return self._n_edges_global

    def n_faces_global(self) -> int:
        """Faces of the whole tetrahedral mesh: 4 per cell, interior faces counted twice (exterior ones once)."""
        if getattr(self, "_n_faces_global", None) is None:
            self._n_faces_global = (4 * self.mesh.num_cells + int(self.mesh.exterior_facets().shape[0])) // 2
        return self._n_faces_global

    def num_dofs_global(self, degree: int) -> int:
        if degree == 3:
            last = self.n_faces_global() if self.mesh.gdim == 3 else self.mesh.num_cells
            return self.mesh.num_vertices + 2 * self.n_edges_global() + last
        return self.mesh.num_vertices + (self.n_edges_global() if degree == 2 else 0)
