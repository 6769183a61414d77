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
