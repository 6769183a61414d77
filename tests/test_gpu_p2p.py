"""Direct xGMI transport, single-process checks: a peer that never answers makes the bounded waits
time out, the error is sticky (later exchanges do not wait again) and the host call reports it.
(The exchanges themselves are covered by tests/test_gpu_dist_rehearsal.py with real peers.)"""
import ctypes as C
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_absent_peer_times_out_and_is_reported(hip):
    from oasisx_amd import _lib

    lib = _lib.load()
    nr, ng, n_owned = 2, 4, 8
    nbytes = lib.ox_p2p_window_bytes(nr, ng)
    assert nbytes >= 2 * nr * 128 + 2 * ng * 3 * 8
    w0, w1, h = C.c_void_p(), C.c_void_p(), C.create_string_buffer(64)
    _lib.check(lib.ox_p2p_window_create(nbytes, C.byref(w0), h), "window_create")
    _lib.check(lib.ox_p2p_window_create(nbytes, C.byref(w1), h), "window_create")  # stands in for rank 1: silent
    peers = np.asarray([1], dtype=np.int32)
    send_off = np.asarray([0, 2], dtype=np.int64)
    recv_off = np.asarray([0, ng], dtype=np.int64)
    send_idx = torch.tensor([0, 1], dtype=torch.int32, device="cuda")
    plan = C.c_void_p()
    _lib.check(lib.ox_dist_create(None, 0, nr, 1, peers.ctypes.data_as(C.POINTER(C.c_int32)),
                                  send_off.ctypes.data_as(C.POINTER(C.c_int64)), _lib.ptr(send_idx),
                                  recv_off.ctypes.data_as(C.POINTER(C.c_int64)), n_owned, ng, C.byref(plan)),
               "ox_dist_create")
    x = torch.arange(n_owned + ng, dtype=torch.float64, device="cuda")
    # no transport yet: loud failure, not a silent no-op
    assert lib.ox_halo_forward(plan, _lib.ptr(x), 1, _lib.current_stream()) != 0
    assert b"no transport" in lib.ox_last_error()
    wins = (C.c_void_p * nr)(w0.value, w1.value)
    off = np.asarray([0], dtype=np.int64)
    png = np.asarray([ng], dtype=np.int64)
    _lib.check(lib.ox_dist_enable_p2p(plan, w0, wins, off.ctypes.data_as(C.POINTER(C.c_int64)),
                                      png.ctypes.data_as(C.POINTER(C.c_int64)), 0.25), "ox_dist_enable_p2p")
    _lib.check(lib.ox_dist_status(plan), "ox_dist_status")
    t0 = time.perf_counter()
    _lib.check(lib.ox_halo_forward(plan, _lib.ptr(x), 1, _lib.current_stream()), "ox_halo_forward")
    torch.cuda.synchronize()
    waited = time.perf_counter() - t0
    assert 0.2 < waited < 5.0, waited
    assert lib.ox_dist_status(plan) != 0 and b"timed out" in lib.ox_last_error()
    # the values this rank pushed did arrive in the peer's window (staging of parity 1 = first exchange)
    # and the error is sticky: the next exchanges return without waiting
    buf = torch.ones(3, dtype=torch.float64, device="cuda")
    t0 = time.perf_counter()
    _lib.check(lib.ox_allreduce_sum(plan, _lib.ptr(buf), 3, _lib.current_stream()), "ox_allreduce_sum")
    _lib.check(lib.ox_halo_forward(plan, _lib.ptr(x), 1, _lib.current_stream()), "ox_halo_forward")
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 0.2
    assert lib.ox_dist_status(plan) != 0
    lib.ox_dist_destroy(plan)  # owns w0; w1 was never an IPC mapping
    lib.ox_p2p_window_free(w1)


def _window_allreduce_ranks(nr, n, fast=False):
    """The in-block all-reduce of the window transport with ``nr`` ranks living in ONE process: every rank has its own
    window, its plan maps all of them, the ``nr`` kernels run concurrently on ``nr`` streams and wait for each other's
    flags.  Sixteen lanes per rank move the values; the sums are formed in rank order: the same bits on every rank, equal to
    the host's left-to-right sum.  Twice, for both parities of the slots."""
    from oasisx_amd import _lib

    lib = _lib.load()
    ng = 0
    nbytes = lib.ox_p2p_window_bytes(nr, ng)
    wins_v, plans = [], []
    h = C.create_string_buffer(64)
    for r in range(nr):
        w = C.c_void_p()
        _lib.check(lib.ox_p2p_window_create(nbytes, C.byref(w), h), "window_create")
        wins_v.append(w)
    wins = (C.c_void_p * nr)(*[w.value for w in wins_v])
    empty32, off1 = np.zeros(0, dtype=np.int32), np.zeros(1, dtype=np.int64)
    idx = torch.zeros(1, dtype=torch.int32, device="cuda")
    for r in range(nr):
        plan = C.c_void_p()
        _lib.check(lib.ox_dist_create(None, r, nr, 0, empty32.ctypes.data_as(C.POINTER(C.c_int32)),
                                      off1.ctypes.data_as(C.POINTER(C.c_int64)), _lib.ptr(idx),
                                      off1.ctypes.data_as(C.POINTER(C.c_int64)), 8, ng, C.byref(plan)), "ox_dist_create")
        _lib.check(lib.ox_dist_enable_p2p(plan, wins_v[r], wins, off1.ctypes.data_as(C.POINTER(C.c_int64)),
                                          off1.ctypes.data_as(C.POINTER(C.c_int64)), 5.0), "ox_dist_enable_p2p")
        if fast:  # (the library's default is the conservative release protocol)
            _lib.check(lib.ox_dist_set_p2p_release(plan, 0), "ox_dist_set_p2p_release")
        plans.append(plan)
    streams = [torch.cuda.Stream() for _ in range(nr)]
    rng = np.random.default_rng(7)
    for rnd in range(2):
        vals = rng.standard_normal((nr, n)) * 10.0 ** rng.integers(-6, 6, size=(nr, n))  # (no sum order gives the same bits)
        bufs = [torch.from_numpy(vals[r].copy()).cuda() for r in range(nr)]
        torch.cuda.synchronize()
        for r in range(nr):
            _lib.check(lib.ox_allreduce_sum(plans[r], _lib.ptr(bufs[r]), n, C.c_void_p(streams[r].cuda_stream)), "ox_allreduce_sum")
        torch.cuda.synchronize()
        want = np.zeros(n)
        for r in range(nr):
            want = want + vals[r]
        for r in range(nr):
            _lib.check(lib.ox_dist_status(plans[r]), "ox_dist_status")
            assert np.array_equal(bufs[r].cpu().numpy(), want), (rnd, r)
    for r in range(nr):
        lib.ox_dist_destroy(plans[r])  # (a plan owns its own window; the others were never IPC mappings)


def _in_a_child_with_eight_queues(call):
    """``nr`` mutually waiting kernels need ``nr`` hardware queues: a process has four by default (one of them the null
    stream's) and two pool streams may share one -- the kernels would then serialise and run into their bounded waits.
    Every in-process multi-rank case therefore runs in a child process with GPU_MAX_HW_QUEUES=8 (set before HIP starts)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", f"import tests.test_gpu_p2p as t; {call}; print('ranks ok')"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ranks ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.parametrize("fast", [False, True])
@pytest.mark.parametrize("nr,n", [(3, 15), (3, 5), (2, 1)])
def test_window_allreduce_of_several_ranks_in_one_process(hip, nr, n, fast):
    """Two and three ranks, 1 to 15 values, both release protocols (conservative = the default; fast = round 5's).  ONE
    configuration per child process: the first streams a process takes from the pool land on distinct hardware queues;
    later ones need not (a second configuration in the same process was seen to share a queue and time out)."""
    _in_a_child_with_eight_queues(f"t._window_allreduce_ranks({nr}, {n}, {fast})")


@pytest.mark.parametrize("fast", [False, True])
def test_window_allreduce_of_six_ranks_two_waves_of_lanes(hip, fast):
    """Six ranks -- 96 lanes, two waves, what an 8-GPU job runs -- need six kernels in flight."""
    _in_a_child_with_eight_queues(f"t._window_allreduce_ranks(6, 15, {fast})")


@pytest.mark.parametrize("transport", ["rccl", "p2p"])
def test_self_loop_plans_step_without_errors(hip, transport):
    """``parallel.SelfLoopComm`` (tools/predict_scaling.py: one rank of a P-rank job alone on the device, its plans folded
    onto itself): with both device transports the rank's spaces build, a whole time step runs through the partitioned
    defaults of the Krylov solvers, and no bounded wait of the window transport runs out.  (The numbers are not the job's:
    every ghost receives some owned value.  What is checked is that the code path the predictions time is alive.)"""
    import oasisx_amd as ox
    from oasisx_amd import _lib
    from oasisx_amd import mesh as M
    from oasisx_amd.parallel import SelfLoopComm

    comm = SelfLoopComm(1, 3, transport)
    mesh = M.create_unit_cube(comm, 6, 6, 6)
    on = lambda x: np.isclose(x[0], 0.0) | np.isclose(x[0], 1.0)  # noqa: E731
    bcs = [[ox.DirichletBC(0.0, ox.LocatorMethod.GEOMETRICAL, on)] for _ in range(3)]
    ksp = {"pc_type": "jacobi", "ksp_rtol": 1e-6, "ksp_atol": 1e-12, "ksp_max_it": 50}
    S = ox.FractionalStep_AB_CN(mesh, ("Lagrange", 2), ("Lagrange", 1), bcs_u=bcs, bcs_p=[], body_force=(0.0, 0.0, -1.0),
                                solver_options={"tentative": dict(ksp, ksp_type="bcgs"),
                                                "pressure": dict(ksp, ksp_type="cg", ksp_error_if_not_converged=False),
                                                "scalar": dict(ksp, ksp_type="cg")})
    Vi, Q = S._Vi[0][0], S._Q
    assert Vi.dist is not None and Vi.n_local > Vi.n_owned and comm.active == {2: "self-loop", 1: "self-loop"}
    lib = _lib.load()
    for _ in range(2):
        S.assemble_first(0.01, 0.01)
        S.velocity_tentative_assemble()
        S.velocity_tentative_solve()
        S.pressure_assemble(0.01)
        S.pressure_solve()
        S.velocity_update(0.01)
    torch.cuda.synchronize()
    for V in (Vi, Q):
        _lib.check(lib.ox_dist_status(V.dist), "ox_dist_status")
    t = comm.time_transports(Q, reps=5)
    assert set(t) == {"self-loop"} and all(v > 0 for v in t["self-loop"].values())
    assert np.isfinite(S._U1.rhost()).all()
