"""PREDICTED strong-scaling curve of the IPCS step at 1 / 2 / 4 / 8 GPUs from ONE-GPU measurements.
No multi-GPU node has been available in rounds 1-5: this is a prediction, not a measurement, and says so in its output.

For every P in --P and every rank r of the P-rank job this builds rank r ALONE on the GPU (parallel.SelfLoopComm: the
rank's partition, spaces, halo plans, operators and the partitioned Krylov defaults are those of the real job; the plans
exchange with the rank itself through a one-rank RCCL communicator) and times, with HIP events,

  * the phases that do not iterate: assemble_first, the two right-hand-side assemblies, the update's right-hand side;
  * one Krylov iteration of each solve -- three-column BiCGStab on A, three-column CG on M, their one-column (narrowed)
    forms, one-column CG on Ap -- as (T(k2) - T(k1)) / (k2 - k1) of solves forced to k1 < k2 iterations, and the fixed
    cost of a solve (set-up kernels, host read-backs, finish) as T(k1) - k1 * t_iter.  The self-loop exchanges and
    one-rank all-reduces are INSIDE these times (pack / unpack kernels, RCCL launch costs);
  * what the link adds is MODELLED from the rank's real halo plan: max over peers of bytes / link bandwidth + a latency
    per exchange, and a latency per all-reduce (``ASSUMED`` below: no measurement exists; bench.py --gpus N reports the
    measured exchange times under config.transport_exchange_us for the first hardware run to replace them).

The iteration profile (three-column iterations, narrowed iterations, pressure iterations per step) is that of the REAL
one-GPU run of the same workload (20 timed steps after 5, as bench.py): the partitioned solves are the same global Krylov
methods.  Predicted step time of P ranks = max over ranks of the sum of its phases; the P = 1 prediction against the
measured P = 1 step is the model's own error.

    python tools/predict_scaling.py --N 128 --P 1 2 4 8 --out profiles/r05_predicted_scaling.json
"""
import argparse
import gc
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np
import torch

import oasisx_amd as ox
from oasisx_amd import _lib
from oasisx_amd import mesh as M
from oasisx_amd.fem import FieldStorage
from oasisx_amd.ksp import KSPSolver
from oasisx_amd.parallel import SelfLoopComm

from scaling_model import ASSUMED, ASSUMED_P2P, predict as _predict  # noqa: E402  (tools/scaling_model.py)

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=128)
ap.add_argument("--P", type=int, nargs="+", default=[1, 2, 4, 8])
ap.add_argument("--ranks", default="all", help="all | ends (first and last rank of each P only: large meshes)")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--matrix-free", action="store_true")
ap.add_argument("--out", default=None)
ap.add_argument("--transport", default="rccl", choices=["rccl", "p2p"],
                help="device transport of the self-loop plans: the one-rank RCCL communicator, or the xGMI-window kernels")
args = ap.parse_args()
if args.transport == "p2p":
    ASSUMED = ASSUMED_P2P


def predict(m, profile, P):
    return _predict(m, profile, P, ASSUMED)


N, pi = args.N, math.pi
nu, dt = 0.01, 0.005 * 32.0 / N
box = ([-1.0] * 3, [1.0] * 3)
clock = {"t": 0.0}


def xp(x):
    return torch if torch.is_tensor(x) else np


fns = [lambda x, t: -xp(x).cos(pi * x[0]) * xp(x).sin(pi * x[1]) * math.exp(-2.0 * nu * pi ** 2 * t),
       lambda x, t: xp(x).cos(pi * x[1]) * xp(x).sin(pi * x[0]) * math.exp(-2.0 * nu * pi ** 2 * t),
       lambda x, t: xp(x).zeros_like(x[0])]


def pres(x, t):
    return -0.25 * (np.cos(2 * pi * x[0]) + np.cos(2 * pi * x[1])) * math.exp(-4.0 * nu * pi ** 2 * t)


def on_boundary(x):
    on = np.zeros(x.shape[1], dtype=bool)
    for k in range(3):
        on |= np.isclose(x[k], -1.0) | np.isclose(x[k], 1.0)
    return on


def at(f, t=None):
    def g(x):
        return f(x, clock["t"] if t is None else t)
    g.supports_torch = True
    return g


KSP = {"pc_type": "jacobi", "ksp_rtol": 1e-8, "ksp_atol": 1e-14, "ksp_max_it": 10000, "ksp_initial_guess_nonzero": True}


def build(comm):
    mesh = M.create_box(comm, list(box), [N, N, N])
    bcs = [[ox.DirichletBC(at(f), ox.LocatorMethod.GEOMETRICAL, on_boundary)] for f in fns]
    so = {"tentative": dict(KSP, ksp_type="bcgs"), "pressure": dict(KSP, ksp_type="cg", ksp_error_if_not_converged=False),
          "scalar": dict(KSP, ksp_type="cg")}
    S = ox.FractionalStep_AB_CN(mesh, ("Lagrange", 2), ("Lagrange", 1), bcs_u=bcs, bcs_p=[], solver_options=so,
                                options={"low_memory_version": args.matrix_free})
    clock["t"] = 0.0
    for i, f in enumerate(fns):
        S._u2[i].interpolate(at(f, -dt))
        S._u1[i].interpolate(at(f, 0.0))
        S._u[i].interpolate(at(f, 0.0))
    S._p.interpolate(lambda x: pres(x, -dt / 2.0))
    return mesh, S


def ev_time(fn, reps=5, warm=1):
    """Median of `reps` individually event-timed calls (ms): one slow call (a page fault, a clock ramp) does not decide."""
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def krylov_costs(comm, A, nc, kind, k1, k2, n_rows_local):
    """(ms per iteration, fixed ms per solve) of `kind` on A with nc columns, partitioned defaults of KSPSolver."""
    g = torch.Generator(device="cuda").manual_seed(nc)
    B, X = FieldStorage(n_rows_local, nc, "cuda"), FieldStorage(n_rows_local, nc, "cuda")
    B.dev().copy_(torch.randn(n_rows_local, nc, dtype=torch.float64, device="cuda", generator=g))
    out = {}
    for k in (k1, k2):
        ksp = KSPSolver(comm, {"ksp_type": kind, "pc_type": "jacobi", "ksp_rtol": 0.0, "ksp_atol": 0.0, "ksp_max_it": k})
        ksp.setOperators(A)

        def run():
            X.dev().zero_()
            ksp.solve_block(B, X)
        run()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()  # wall time: a solve contains host read-backs
            run()
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        its = max(int(ksp.last_result.its[c]) for c in range(nc))
        out[k] = (sorted(ts)[2], its, [int(ksp.last_result.reason[c]) for c in range(nc)])
    (ta, ia, ra), (tb, ib, rb) = out[k1], out[k2]
    it_ms = (tb - ta) / max(ib - ia, 1)
    return {"iter_ms": it_ms, "fixed_ms": max(ta - ia * it_ms, 0.0), "its": [ia, ib], "reasons": [ra, rb],
            "method": int(ksp._method()[0]) if not (nc == 1 and kind == "cg" and ksp._cg_merged()) else int(_lib.KSP_CG_MERGED),
            "check_every": int(ksp._interval_for(nc, ksp._method()[0]))}


def measure_rank(comm):
    t0 = time.perf_counter()
    mesh, S = build(comm)
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    Vi, Q = S._Vi[0][0], S._Q
    gd = 3
    m = {"setup_s": setup_s, "velocity_rows": int(Vi.n_owned), "velocity_ghosts": int(Vi.n_local - Vi.n_owned),
         "pressure_rows": int(Q.n_owned), "pressure_ghosts": int(Q.n_local - Q.n_owned), "cells": int(Vi.local_cells.shape[0])}
    if Vi.halo is not None:
        hv, hq = Vi.halo, Q.halo
        m["peers"] = [int(p_) for p_ in hv["peers"]]
        # the REAL plan's traffic: per peer, values sent (the link carries gd or 1 doubles each)
        m["send_max_u"] = int(np.diff(hv["send_off"]).max()) if len(hv["peers"]) else 0
        m["recv_max_u"] = int(np.diff(hv["recv_off"]).max()) if len(hv["peers"]) else 0
        m["send_max_p"] = int(np.diff(hq["send_off"]).max()) if len(hq["peers"]) else 0
        m["recv_max_p"] = int(np.diff(hq["recv_off"]).max()) if len(hq["peers"]) else 0
    clock["t"] = dt
    for bcu in S._bcs_u:
        for bc in bcu:
            bc.update_bc()
    m["assemble_first_ms"] = ev_time(lambda: S.assemble_first(dt, nu))
    m["tentative_assemble_ms"] = ev_time(S.velocity_tentative_assemble)
    m["pressure_assemble_ms"] = ev_time(lambda: S.pressure_assemble(dt))
    # the update's right-hand side: M u* and the gradient of dp (velocity_update without its solve)
    lib, st = S._lib, _lib.current_stream()

    def update_rhs():
        S._M.mult(S._U.dev(), S._WRK.dev(), gd)
        if not S._low_memory:
            S._grad_p_Mat.mult(False, S._DP.ptr(), S._WRK.ptr(), -float(dt), S._B3.ptr())
    m["update_rhs_ms"] = ev_time(update_rhs)
    # the per-step vector work of solve() itself: ps <- p, the three shifts, diff (6 axpby-sized passes over u, 2 over p)
    n = S._n_u * gd

    def shifts():
        for a_, b_ in ((S._U1, S._U2), (S._U, S._U1), (S._U, S._WRK), (S._WRK, S._U2)):
            _lib.check(lib.ox_axpby(n, 1.0, a_.rptr(), 0.0, None, b_.ptr(), st), "ox_axpby")
        _lib.check(lib.ox_axpby(S._n_q, 1.0, S._PS.ptr(), 0.0, None, S._P.ptr(), st), "ox_axpby")
    S._U2.dev()
    keep = S._U2.dev().clone()
    m["step_vector_work_ms"] = ev_time(shifts)
    S._U2.dev().copy_(keep)
    del keep
    c = comm
    m["bcgs3"] = krylov_costs(c, S._A, 3, "bcgs", 3, 9, Vi.n_local)
    m["bcgs1"] = krylov_costs(c, S._A, 1, "bcgs", 4, 12, Vi.n_local)
    m["cgM3"] = krylov_costs(c, S._M, 3, "cg", 3, 9, Vi.n_local)
    m["cgM1"] = krylov_costs(c, S._M, 1, "cg", 6, 18, Vi.n_local)
    m["cgP1"] = krylov_costs(c, S._Ap, 1, "cg", 32, 96, Q.n_local)
    if c is not None:
        m["transport_self_loop_us"] = {"velocity": c.time_transports(Vi, reps=100), "pressure": c.time_transports(Q, reps=100)}
    return mesh, S, m


def real_run():
    """The one-GPU run as bench.py times it: iteration profile and phase times."""
    mesh, S, m = measure_rank(None)
    clock["t"] = 0.0
    for i, f in enumerate(fns):
        S._u2[i].interpolate(at(f, -dt))
        S._u1[i].interpolate(at(f, 0.0))
    S._p.interpolate(lambda x: pres(x, -dt / 2.0))
    prof = []

    def step():
        clock["t"] += dt
        S.solve(dt, nu, max_iter=1)
        it = S.iteration_counts()
        t_, u_ = sorted(it["tentative"][:3]), sorted(it["update"][:3])
        prof.append({"tent3": t_[1], "tent1": t_[2] - t_[1], "upd3": u_[1], "upd1": u_[2] - u_[1], "pressure": it["pressure"][0]})
    for _ in range(args.warmup):
        step()
    prof.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    profile = {k: float(np.mean([p_[k] for p_ in prof])) for k in prof[0]}
    return m, profile, 1e3 * el / args.steps


def free(*objs):
    del objs
    gc.collect()
    torch.cuda.empty_cache()


out = {"label": "PREDICTION from one-GPU measurements -- NOT a measurement of a multi-GPU run", "assumed": ASSUMED,
       "transport": {"rccl": "RCCL plans (the default of a job with an RCCL communicator), exchange-then-multiply",
                     "p2p": "xGMI-window plans (OX_TRANSPORT=p2p): push / pull kernels, one-kernel all-reduce, overlapped mat-vecs"}[args.transport],
       "workload": f"3D Taylor-Green {N}^3 x 6 tets P2-P1 (BASELINE configs[2] / [4]), nu={nu}, dt={dt:g}, bcgs+jacobi / cg+jacobi "
                   f"rtol 1e-8, initial_guess_nonzero, low_memory_version={args.matrix_free}",
       "device": torch.cuda.get_device_name(0), "P": {}}
m1, profile, measured_ms = real_run()
out["iteration_profile_per_step"] = profile
out["measured_one_gpu_ms_per_step"] = measured_ms
ph1 = predict(m1, profile, 1)
out["P"]["1"] = {"ranks": {"0": m1}, "phases_ms": ph1, "ms_per_step": sum(ph1.values()), "steps_per_s": 1e3 / sum(ph1.values()),
                 "model_error_vs_measured": sum(ph1.values()) / measured_ms - 1.0}
print(f"P=1: measured {measured_ms:.2f} ms/step, model {sum(ph1.values()):.2f} ms ({100 * out['P']['1']['model_error_vs_measured']:+.1f} %); "
      f"profile {profile}", flush=True)
gc.collect()
torch.cuda.empty_cache()
for P in [p_ for p_ in args.P if p_ > 1]:
    ranks = list(range(P)) if args.ranks == "all" else sorted({0, P - 1})
    per, phs = {}, {}
    for r in ranks:
        comm = SelfLoopComm(r, P, args.transport)
        mesh, S, m = measure_rank(comm)
        del mesh, S
        gc.collect()
        torch.cuda.empty_cache()
        per[str(r)] = m
        phs[str(r)] = predict(m, profile, P)
        print(f"P={P} rank {r}: rows u {m['velocity_rows']} (+{m['velocity_ghosts']} ghosts) p {m['pressure_rows']}, peers {m.get('peers')}, "
              f"step {sum(phs[str(r)].values()):.2f} ms  {', '.join(f'{k} {v:.2f}' for k, v in phs[str(r)].items())}", flush=True)
    worst = max(phs, key=lambda r_: sum(phs[r_].values()))
    phase_max = {k: max(phs[r_][k] for r_ in phs) for k in phs[worst]}
    ms = sum(phs[worst].values())
    out["P"][str(P)] = {"ranks": per, "phases_ms_by_rank": phs, "slowest_rank": int(worst), "phases_ms": phs[worst],
                        "phase_max_over_ranks_ms": phase_max, "ms_per_step": ms, "steps_per_s": 1e3 / ms,
                        "speedup_vs_model_P1": sum(ph1.values()) / ms, "parallel_efficiency": sum(ph1.values()) / ms / P,
                        "efficiency_by_phase": {k: ph1[k] / (P * phs[worst][k]) if phs[worst][k] > 0 else None for k in ph1},
                        "ranks_measured": ranks}
    print(f"P={P}: predicted {1e3 / ms:.2f} steps/s ({ms:.2f} ms/step on rank {worst}), speed-up {sum(ph1.values()) / ms:.2f}, "
          f"efficiency {sum(ph1.values()) / ms / P:.2f}", flush=True)
line = json.dumps(out)
if args.out:
    open(args.out, "w").write(line + "\n")
print(line)
