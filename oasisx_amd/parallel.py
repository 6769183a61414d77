"""Mesh partitioning and the RCCL communicator for runs on several GPUs of one node
(one process per GPU, launched by ``torch.distributed.run``).

The reference gets its parallelism from DOLFINx/PETSc over MPI (row-partitioned matrices,
owned + ghost vectors, ``scatter_forward`` -- reference fracstep.py:390,453,497,551,632,655;
ksp.py:77).  Here:

* every rank generates the (global) mesh and computes the SAME partition metadata (cell ->
  rank, vertex/edge -> owner rank); nothing is communicated at set-up;
* cells are cut into ``nparts`` slabs by centroid (z-major sort, equal counts); a dof belongs to
  the lowest rank among the cells that contain it; a rank keeps every cell that touches one of
  its dofs (its own cells + one ghost layer), so every owned matrix row / vector entry is
  assembled locally and no matrix entries are ever communicated (PETSc's ``Mat.assemble``
  stash exchange disappears);
* per SpMV the owner sends its interface values straight into the neighbour's ghost block
  (grouped ncclSend/ncclRecv, ``ox_halo_forward``); per Krylov synchronisation point one small
  ncclAllReduce merges all dot products.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .fem import local_edges
from .mesh import Mesh


class Comm:
    """Rank/size plus (on GPUs) the RCCL communicator handle used by liboasisx_hip.so."""

    def __init__(self, rank=0, size=1, handle=None):
        self.rank, self.size, self.handle = rank, size, handle

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


def init_comm() -> Comm:
    """Communicator of the current ``torch.distributed`` job.  With the nccl (= RCCL) backend the
    library's own ncclComm is created from a unique id broadcast over torch.distributed."""
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return Comm(0, 1, None)
    rank, size = dist.get_rank(), dist.get_world_size()
    if dist.get_backend() != "nccl":
        return Comm(rank, size, None)  # CPU rehearsal (gloo): partition logic only
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


class MeshPartition:
    """Replicated partition metadata of a (global) mesh for one rank."""

    def __init__(self, mesh: Mesh, rank: int, nparts: int):
        self.mesh, self.rank, self.nparts = mesh, int(rank), int(nparts)
        dev = mesh.device
        d = mesh.gdim
        cells = mesh.cells
        nc, nverts = mesh.num_cells, mesh.num_vertices
        # ---- cells -> slabs by centroid (z, y, x) order, equal counts --------------------------
        cen = mesh.coords[cells].mean(dim=1)
        lo = mesh.coords.min(dim=0).values
        span = (mesh.coords.max(dim=0).values - lo).clamp_min(1e-300)
        q = torch.round((cen - lo) / span * float(1 << 20)).to(torch.int64)
        key = q[:, d - 1]
        for k in range(d - 2, -1, -1):
            key = key * (1 << 21) + q[:, k]
        order = torch.argsort(key, stable=True)
        cell_rank = torch.empty(nc, dtype=torch.int64, device=dev)
        cell_rank[order] = torch.div(torch.arange(nc, device=dev) * nparts, nc, rounding_mode="floor")
        self.cell_rank = cell_rank
        # ---- edges (global ids) -------------------------------------------------------------
        ea = torch.tensor([e[0] for e in local_edges(d)], device=dev)
        eb = torch.tensor([e[1] for e in local_edges(d)], device=dev)
        a, b = cells[:, ea], cells[:, eb]
        ekey = torch.minimum(a, b) * nverts + torch.maximum(a, b)
        self.edge_keys, inv = torch.unique(ekey.reshape(-1), return_inverse=True)
        self.cell_edges = inv.reshape(nc, -1)
        del a, b, ekey, inv
        # ---- owners: lowest rank among the cells that contain the entity ------------------------
        big = torch.full((nverts,), nparts, dtype=torch.int64, device=dev)
        self.vown = big.scatter_reduce(0, cells.reshape(-1), cell_rank.repeat_interleave(d + 1), reduce="amin")
        ne = cells.shape[1] * (cells.shape[1] - 1) // 2
        bige = torch.full((int(self.edge_keys.shape[0]),), nparts, dtype=torch.int64, device=dev)
        self.eown = bige.scatter_reduce(0, self.cell_edges.reshape(-1), cell_rank.repeat_interleave(ne),
                                        reduce="amin")
        self._own2 = None
        self.local_cells = torch.nonzero(self.cell_mask(self.rank)).reshape(-1)

    def owner0(self, degree: int) -> torch.Tensor:
        """Owner rank of every global initial dof (vertices, then edges for degree 2)."""
        if degree == 1:
            return self.vown
        if self._own2 is None:
            self._own2 = torch.cat([self.vown, self.eown])
        return self._own2

    def cell_mask(self, q: int) -> torch.Tensor:
        """Cells rank ``q`` keeps: those touching a vertex or an edge it owns."""
        return (self.vown[self.mesh.cells] == q).any(dim=1) | (self.eown[self.cell_edges] == q).any(dim=1)
