"""Shared set-up for the parity tests: the same Taylor-Green problem on the HIP path and on the
CPU oracle, with the oracle fed the product's mesh arrays and dof numbering so that fields and
matrices compare index by index."""
from __future__ import annotations

import numpy as np

from oracle import ipcs_oracle as O

KRYLOV = {
    "tentative": {"ksp_type": "bcgs", "pc_type": "jacobi", "ksp_rtol": 1e-11, "ksp_atol": 1e-30},
    "pressure": {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-11, "ksp_atol": 1e-30},
    "scalar": {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-11, "ksp_atol": 1e-30},
}
LU = {k: {"ksp_type": "preonly", "pc_type": "lu"} for k in ("tentative", "pressure", "scalar")}


def tg_mesh(dim, N, device=None):
    from oasisx_amd import mesh as M

    if dim == 2:
        return M.create_rectangle(None, [[-1.0, -1.0], [1.0, 1.0]], [N, N], device=device)
    return M.create_box(None, [[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]], [N, N, N], device=device)


def on_boundary(x):
    on = np.isclose(np.abs(x[0]), 1.0) | np.isclose(np.abs(x[1]), 1.0)
    return on


def on_boundary3(x):
    return on_boundary(x) | np.isclose(np.abs(x[2]), 1.0)


def make_hip_problem(dim, N, u_deg=2, nu=0.01, dt=0.005, t0=0.0, solver_options=None, window=256,
                     body_force=None, rotational=False, low_memory=True):
    """FractionalStep_AB_CN on the HIP path with the demo's set-up
    (reference demo/taylor_green.py:104-182)."""
    import oasisx_amd as ox

    mesh = tg_mesh(dim, N)
    clock = {"t": t0}
    marker = on_boundary if dim == 2 else on_boundary3
    fns = [O.tg_u, O.tg_v, O.tg_w][:dim]
    bcs_u = [[ox.DirichletBC(lambda x, f=f: f(x, clock["t"], nu), ox.LocatorMethod.GEOMETRICAL, marker)]
             for f in fns]
    S = ox.FractionalStep_AB_CN(mesh, ("Lagrange", u_deg), ("Lagrange", 1), bcs_u=bcs_u, bcs_p=[],
                                solver_options=solver_options or KRYLOV, body_force=body_force,
                                options={"sell_window": window, "low_memory_version": low_memory},
                                rotational=rotational)
    for i, f in enumerate(fns):
        S._u2[i].interpolate(lambda x, f=f: f(x, t0 - dt, nu))
        S._u1[i].interpolate(lambda x, f=f: f(x, t0, nu))
    S._p.interpolate(lambda x: O.tg_p(x, t0 - dt / 2.0, nu))
    return S, clock, mesh


def make_oracle_twin(S, mesh, dim, u_deg=2, nu=0.01, dt=0.005, t0=0.0, solver_options=None, rotational=False):
    """The CPU oracle on the product's mesh arrays and dof numbering."""
    Vi, Q = S._Vi[0][0], S._Q
    return O.taylor_green_problem(
        0, dim, u_deg=u_deg, p_deg=1, nu=nu, dt=dt, t0=t0, solver_options=solver_options or KRYLOV,
        mesh=(mesh.coords.cpu().numpy(), Vi.cells_in_kernel_order()),
        vd=Vi.cell_dofs.cpu().numpy(), qd=Q.cell_dofs.cpu().numpy(),
        x_v=Vi.x.cpu().numpy(), x_q=Q.x.cpu().numpy(), rotational=rotational)


def run_tg_pair(dim=2, N=8, u_deg=2, steps=2, nu=0.01, dt=0.005, hip_options=None, oracle_options=None,
                rotational=False, low_memory=True):
    S, clock, mesh = make_hip_problem(dim, N, u_deg, nu, dt, solver_options=hip_options, rotational=rotational,
                                      low_memory=low_memory)
    R, rclock = make_oracle_twin(S, mesh, dim, u_deg, nu, dt, solver_options=oracle_options, rotational=rotational)
    t = 0.0
    for _ in range(steps):
        t += dt
        clock["t"] = t
        rclock["t"] = t
        S.solve(dt, nu, max_iter=1)
        R.solve(dt, nu, max_iter=1)
    u = S.u.x.array.reshape(-1, dim)
    p = S._p.x.array
    return {"du": float(np.abs(u - R.u1).max()), "dp": float(np.abs(p - R.p).max()),
            "umax": float(np.abs(R.u1).max()), "its_hip": S.iteration_counts(), "its_oracle": dict(R.its),
            "S": S, "R": R}


def delaunay_box_mesh(n, dim=3, seed=0, jitter=0.35):
    """Arrays (points (nv, dim), cells (nc, dim+1) int64) of oasisx_amd.mesh.create_delaunay_box on [-1, 1]^dim:
    a genuinely unstructured simplicial mesh (Delaunay triangulation of a jittered lattice)."""
    from oasisx_amd import mesh as M

    m = M.create_delaunay_box(None, [[-1.0] * dim, [1.0] * dim], n, seed=seed, jitter=jitter, device="cpu")
    return m.coords.cpu().numpy().copy(), m.cells.cpu().numpy().astype(np.int64)


# ---- the ranks of a partitioned job as THREADS of one process (test infrastructure) --------------------------------
class ThreadWorld:
    """Meeting point of ``size`` rank threads: FIFO mailboxes per ordered pair of ranks (halo pieces) and a double
    barrier around a shared table (all-reduces).  Every wait is bounded: a rank that died breaks the barrier for all."""

    def __init__(self, size: int, timeout_s: float = 120.0):
        import queue
        import threading

        self.size, self.timeout = size, timeout_s
        self.barrier = threading.Barrier(size)
        self.mail = {(r, q): queue.Queue() for r in range(size) for q in range(size) if r != q}
        self.table = [None] * size
        self.exchanges = [0] * size
        self.allreduces = [0] * size

    def allreduce(self, rank, value, combine):
        self.table[rank] = value
        self.barrier.wait(self.timeout)
        out = combine(list(self.table))
        self.barrier.wait(self.timeout)
        return out


def make_thread_comm(world: ThreadWorld, rank: int):
    """A communicator of oasisx_amd.parallel for rank ``rank`` of ``world``: the halo plans get the library's callback
    transport (``ox_dist_create_custom``), whose exchange points meet the other rank THREADS through ``world`` --
    the same plans, pack kernels, call sites and Krylov loops as a job over RCCL, staged through the host."""
    import ctypes as C

    from oasisx_amd import _lib
    from oasisx_amd.parallel import Comm

    class ThreadComm(Comm):
        collective = True

        def __init__(self):
            super().__init__(rank, world.size, None, transport="host")
            self.world = world

        def _all_ok(self, ok):
            return bool(world.allreduce(rank, bool(ok), all))

        def allreduce(self, v, op=None):
            return float(world.allreduce(rank, float(v), max if op == "max" else sum))

        def Barrier(self):
            world.barrier.wait(world.timeout)

        def make_transport(self, V):
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
                    for i, q in enumerate(h["peers"]):
                        s0, s1 = int(h["send_off"][i]) * nc, int(h["send_off"][i + 1]) * nc
                        if s1 > s0:
                            world.mail[(rank, int(q))].put(sb[s0:s1].copy())
                    for i, q in enumerate(h["peers"]):
                        r0, r1 = int(h["recv_off"][i]) * nc, int(h["recv_off"][i + 1]) * nc
                        if r1 > r0:
                            piece = world.mail[(int(q), rank)].get(timeout=world.timeout)
                            assert piece.shape[0] == r1 - r0, (rank, int(q), piece.shape, r1 - r0)
                            gb[r0:r1] = piece
                    if ng:
                        _lib.check(lib.ox_memcpy(ghost_dev, gb.ctypes.data, ng * nc * 8, 1, None), "ox_memcpy")
                    world.exchanges[rank] += 1
                    return 0
                except Exception:  # noqa: BLE001 -- never unwind through the C frame
                    import traceback

                    traceback.print_exc()
                    world.barrier.abort()
                    return 1

            def ared_cb(user, buf_dev, n):
                try:
                    b = np.empty(n)
                    _lib.check(lib.ox_memcpy(b.ctypes.data, buf_dev, n * 8, 0, None), "ox_memcpy")
                    tot = world.allreduce(rank, b, lambda parts: np.sum(np.stack(parts), axis=0))  # (rank order: the same sum everywhere)
                    _lib.check(lib.ox_memcpy(buf_dev, tot.ctypes.data, n * 8, 1, None), "ox_memcpy")
                    world.allreduces[rank] += 1
                    return 0
                except Exception:  # noqa: BLE001
                    import traceback

                    traceback.print_exc()
                    world.barrier.abort()
                    return 1

            cbs = (HALO(halo_cb), ARED(ared_cb))
            self._keep = getattr(self, "_keep", []) + [cbs]
            out = C.c_void_p()
            _lib.check(lib.ox_dist_create_custom(
                rank, world.size, npeer, h["peers"].ctypes.data_as(C.POINTER(C.c_int32)),
                h["send_off"].ctypes.data_as(C.POINTER(C.c_int64)), _lib.ptr(h["send_idx"]),
                h["recv_off"].ctypes.data_as(C.POINTER(C.c_int64)), V.n_owned, ng,
                C.cast(cbs[0], C.c_void_p), C.cast(cbs[1], C.c_void_p), None, C.byref(out)), "ox_dist_create_custom")
            return out

    return ThreadComm()


def run_rank_threads(size: int, target, timeout_s: float = 600.0):
    """``target(comm)`` on ``size`` threads, one communicator each; returns their results in rank order.  A rank that
    raises breaks the barrier for the others; the first exception is re-raised here."""
    import threading

    world = ThreadWorld(size)
    results, errors = [None] * size, [None] * size

    def body(r):
        try:
            import torch

            torch.cuda.set_device(0)
            results[r] = target(make_thread_comm(world, r))
        except BaseException as e:  # noqa: BLE001
            errors[r] = e
            world.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout_s)
    alive = [r for r, t in enumerate(threads) if t.is_alive()]
    first = next((e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)), None)
    if first is None:
        first = next((e for e in errors if e is not None), None)
    if first is not None:
        raise first
    assert not alive, f"rank threads {alive} still running after {timeout_s} s"
    return results, world
