#!/usr/bin/env python3
"""bench.py -- time-steps/sec and pressure-CG SpMV GB/s of the IPCS hot path on MI355X.

Workload (BASELINE.json configs[2]): 3-D Taylor-Green on [-1,1]^3, N^3 x 6 tetrahedra
(N = 128 by default), P2-P1 Taylor-Hood, z-extruded analytic Taylor-Green field with exact
Dirichlet data on every face, nu = 0.01, dt = 0.005*32/N, max_iter = 1, BiCGStab+Jacobi
tentative velocity, CG+Jacobi pressure and velocity update, rtol 1e-8 / atol 1e-14.
A "step" is one FractionalStep_AB_CN.solve().  Synthetic data, float64 throughout.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL and the xGMI windows share device
# memory between the ranks); exported by the launcher normally -- make sure before HIP starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def spmv_bytes(nnz, n_rows, n_cols):
    """Algorithmic bytes of one CSR SpMV, f64 values + int32 columns."""
    return 12 * nnz + 4 * (n_rows + 1) + 8 * n_cols + 8 * n_rows


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("-N", type=int, default=128, help="cubes per direction")
    ap.add_argument("--udeg", type=int, default=2)
    ap.add_argument("--rtol", type=float, default=1e-8)
    ap.add_argument("--zero-guess", action="store_true",
                    help="PETSc default: zero the solution before every Krylov solve "
                         "(default here: -ksp_initial_guess_nonzero, the previous field is the guess)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-cpu-one-core", action="store_true",
                    help="skip the 1-core repetition of the cpu_baseline step (about a minute at 128^3)")
    ap.add_argument("--matrix-free", action="store_true",
                    help="low_memory_version=True: matrix-free vector kernels for the p*, div(u) and grad(phi) "
                         "terms instead of the pre-assembled rectangular operators (reference "
                         "fracstep.py:392-404; the demo's default is the pre-assembled form)")
    ap.add_argument("--window", type=int, default=None,
                    help="rows per length-sorting window of the SELL-64 numbering (tuning; default: the library's)")
    ap.add_argument("--profile-setup", action="store_true",
                    help="cProfile the set-up phase and print rank 0's top entries to stderr")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1: nccl = RCCL over xGMI (the product path); "
                         "gloo = rehearsal of the same partitioned path through the library's host-staged "
                         "callback transport (several ranks may then share one GPU)")
    ap.add_argument("--verbose", action="store_true")
    return ap.parse_args()


def host_cores():
    """Cores this process may actually use (affinity mask, capped by the cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


def main():
    args = parse()
    os.environ.setdefault("OMP_NUM_THREADS", str(host_cores()))  # cpu_baseline leg (oracle/ipcs_cpu.c)
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist

        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    import oasisx_amd as ox
    from oasisx_amd import _lib
    from oasisx_amd import mesh as M

    lib = _lib.load()
    N, nu = args.N, 0.01
    dt = 0.005 * 32.0 / N
    clock = {"t": 0.0}

    def log(*a):
        if args.verbose and rank == 0:
            print("[bench]", *a, file=sys.stderr, flush=True)

    import math

    # array-API style: numpy for the initial conditions, torch device tensors for the per-step
    # Dirichlet values (DirichletBC's opt-in ``supports_torch`` path keeps them off the host)
    def xp(x):
        return torch if torch.is_tensor(x) else np

    def tg_u(x, t):
        return -xp(x).cos(np.pi * x[0]) * xp(x).sin(np.pi * x[1]) * math.exp(-2.0 * nu * np.pi ** 2 * t)

    def tg_v(x, t):
        return xp(x).cos(np.pi * x[1]) * xp(x).sin(np.pi * x[0]) * math.exp(-2.0 * nu * np.pi ** 2 * t)

    def tg_w(x, t):
        return xp(x).zeros_like(x[0])

    def tg_p(x, t):
        return -0.25 * (np.cos(2 * np.pi * x[0]) + np.cos(2 * np.pi * x[1])) * math.exp(-4.0 * nu * np.pi ** 2 * t)

    def on_boundary(x):
        return (np.isclose(np.abs(x[0]), 1.0) | np.isclose(np.abs(x[1]), 1.0) | np.isclose(np.abs(x[2]), 1.0))

    prof = None
    if args.profile_setup:
        import cProfile

        prof = cProfile.Profile()
        prof.enable()
    t_setup = time.perf_counter()
    comm = None
    if world > 1:
        from oasisx_amd.parallel import init_comm

        comm = init_comm()  # RCCL communicator of the library, bootstrapped over torch.distributed
    mesh = M.create_box(comm, [[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]], [N, N, N])
    fns = [tg_u, tg_v, tg_w]
    def bc_value(f):
        def g(x):
            return f(x, clock["t"])
        g.supports_torch = True
        return g

    bcs_u = [[ox.DirichletBC(bc_value(f), ox.LocatorMethod.GEOMETRICAL, on_boundary)] for f in fns]
    ksp = {"pc_type": "jacobi", "ksp_rtol": args.rtol, "ksp_atol": 1e-14, "ksp_max_it": 10000,
           "ksp_initial_guess_nonzero": not args.zero_guess}
    solver_options = {"tentative": dict(ksp, ksp_type="bcgs"), "pressure": dict(ksp, ksp_type="cg"),
                      "scalar": dict(ksp, ksp_type="cg")}
    S = ox.FractionalStep_AB_CN(mesh, ("Lagrange", args.udeg), ("Lagrange", 1), bcs_u=bcs_u, bcs_p=[],
                                solver_options=solver_options,
                                options=dict({"low_memory_version": args.matrix_free},
                                             **({"sell_window": args.window} if args.window else {})))
    for i, f in enumerate(fns):
        S._u2[i].interpolate(lambda x, f=f: f(x, -dt))
        S._u1[i].interpolate(lambda x, f=f: f(x, 0.0))
    S._p.interpolate(lambda x: tg_p(x, -dt / 2.0))
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    if prof is not None:
        import pstats

        prof.disable()
        if rank == 0:
            pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(45)
    log(f"setup {t_setup:.1f} s; n_u={S._n_u} n_p={S._n_q} nnz_u={S._M.pattern.nnz} nnz_p={S._Ap.pattern.nnz}; "
        f"mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")

    # per-phase device time of the timed steps: the phase methods of THIS instance are wrapped with a
    # pair of events on the launching stream (bench-only instrumentation; 12 events per step)
    phase_events = {}

    def timed_phase(name):
        fn = getattr(S, name)

        def wrapped(*a, **k):
            if not phase_events.get("_on"):
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            phase_events.setdefault(name, []).append((e0, e1))
            return r

        setattr(S, name, wrapped)

    for name in ("assemble_first", "velocity_tentative_assemble", "velocity_tentative_solve", "pressure_assemble",
                 "pressure_solve", "velocity_update"):
        timed_phase(name)

    def step():
        clock["t"] += dt
        return S.solve(dt, nu, max_iter=1)

    for _ in range(args.warmup):
        step()
        log("warmup step", S.iteration_counts())

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    # HIP events on the launching stream around every 8th launch of each kernel tag (an event pair costs
    # ~12 us of stream bubbles; every launch would slow the 150 us pressure iteration by 8 %)
    _lib.check(lib.ox_profile_begin(200000, 8), "ox_profile_begin")
    phase_events["_on"] = True
    t0 = time.perf_counter()
    its = []
    for _ in range(args.steps):
        step()
        its.append(S.iteration_counts())
    barrier()
    elapsed = time.perf_counter() - t0
    phase_events["_on"] = False
    _lib.check(lib.ox_profile_end(), "ox_profile_end")
    phase_ms = {k: sum(a.elapsed_time(b) for a, b in v) / args.steps for k, v in phase_events.items() if k != "_on"}
    # accuracy at the end of the timed steps: nodal error against the analytic field (rank-local dofs)
    Xd = S._Vi[0][0].x[: S._n_u].T
    Ud = S._U.dev()[: S._n_u]
    err_u = max(float((Ud[:, i] - f(Xd, clock["t"])).abs().max()) for i, f in enumerate(fns))
    if world > 1:
        import torch.distributed as dist

        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    def prof(tag, key=-1):
        cnt, ms = C.c_longlong(0), C.c_double(0.0)
        _lib.check(lib.ox_profile_get(tag, key, C.byref(cnt), C.byref(ms)), "ox_profile_get")
        return int(cnt.value), float(ms.value)

    gd = mesh.gdim
    Pp, Pu = S._Ap.pattern, S._M.pattern
    b_p = spmv_bytes(Pp.nnz, Pp.n_rows, Pp.n_cols)
    # velocity SpMM on gd interleaved vectors: matrix read once, gd x/y vectors
    b_u = 12 * Pu.nnz + 4 * (Pu.n_rows + 1) + gd * 8 * (Pu.n_cols + Pu.n_rows)
    kernels = {}
    ku, kp = Pu.n_rows, Pp.n_rows  # SpMV records are keyed by the matrix's row count
    b_u1 = spmv_bytes(Pu.nnz, Pu.n_rows, Pu.n_cols)  # narrowed (1-column) solves on the velocity matrix
    for name, tag, key, nbytes in (
            ("pressure_cg_spmv", 11, kp, b_p),
            ("velocity_bcgs_spmv_v", 10 * gd + 2, ku, b_u), ("velocity_bcgs_spmv_t", 10 * gd + 3, ku, b_u),
            ("velocity_bcgs_spmv_v_narrowed", 12, ku, b_u1), ("velocity_bcgs_spmv_t_narrowed", 13, ku, b_u1),
            ("mass_cg_spmv", 10 * gd + 1, ku, b_u), ("mass_cg_spmv_narrowed", 11, ku, b_u1),
            ("mass_spmv", 10 * gd + 0, ku, b_u), ("assemble_first", 100, -1, None),
            ("grad_vector_p", 110, -1, None), ("grad_vector_dp", 111, -1, None), ("div_vector", 120, -1, None),
            ("rect_spmv_p_and_gradp", 140, -1, None), ("rect_spmv_div", 141, -1, None),
            # mesh-partitioned runs: exchanges on rank 0 (both include the wait for the peers)
            ("halo_exchange_1comp", 150, 1, None), ("halo_exchange_3comp", 150, 3, None),
            ("krylov_sync_point", 151, -1, None)):
        if key == kp and ku == kp and name != "pressure_cg_spmv":
            continue  # P1-P1: both matrices have the same size; keep the pressure entry only
        cnt, ms = prof(tag, key)
        if cnt:
            k = {"launches": cnt, "avg_us": 1e3 * ms / cnt, "total_ms": ms}
            if nbytes:
                k["algorithmic_bytes"] = nbytes
                k["gbs"] = nbytes / (1e6 * ms / cnt)
            kernels[name] = k
    cg = kernels.get("pressure_cg_spmv")

    def pmc_traffic():
        """HBM bytes per launch of the pressure SpMV from the latest committed PMC pass
        (profiles/*_pmc_hbm.csv: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this
        command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)."""
        import csv
        import glob

        best = None
        grid = 256 * ((((Pp.n_slices + 3) // 4) + 7) // 8 * 8)  # launch grid of the pressure SpMV
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.csv"))):
            for r in csv.DictReader(open(path)):
                if r["kernel"].replace(" ", "").startswith("voidk_spmv<1,1") and int(r.get("grid_size", grid)) == grid:
                    best = (float(r["hbm_total_MB_per_launch"]) * 1e6, os.path.basename(path))
        return best

    def stored_bytes(A, nc=1):
        """Bytes one SpMV launch actually streams from this SELL-64 storage: f64 values (or 1-byte
        value codes where the matrix has a dictionary), 16-bit column codes where the pattern carries
        them (int32 elsewhere), the per-pair bases, slice offsets and the x / y vectors -- padding
        included."""
        P = A.pattern
        cols = P.size * (2 * P.frac16 + 4 * (1.0 - P.frac16))
        bases = 8 * (P.size // 128) if P.frac16 > 0 else 0
        vals = (1 if A.vcode is not None else 8) * P.size
        return int(vals + cols + bases + 8 * (P.n_slices + 1) + nc * 8 * (P.n_cols + P.n_rows))

    roofline = None
    if cg:
        sb = stored_bytes(S._Ap)
        roofline = {"kernel": "k_spmv<1,OX_EPI_DOT> (pressure-Poisson CG SpMV, SELL-64, f64)", "bound": "hbm",
                    "achieved": cg["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": cg["gbs"] / HBM_PEAK_GBS, "traffic": None, "traffic_source": None,
                    "algorithmic_bytes_per_launch": b_p, "avg_launch_us": cg["avg_us"],
                    "launches": cg["launches"],
                    # "achieved" prices the launch at the metric's CSR figure (12 B per nonzero,
                    # BASELINE.md).  The kernel streams fewer bytes -- 16-bit column codes and, where the
                    # matrix has <= 256 distinct values (Laplacian / mass on box meshes), 1-byte value
                    # codes, both lossless -- so frac can exceed 1; the bytes really moved are here:
                    "stored_bytes_per_launch": sb, "stored_gbs": sb / (1e3 * cg["avg_us"]),
                    "stored_frac": sb / (1e3 * cg["avg_us"]) / HBM_PEAK_GBS,
                    "cols16_fraction": Pp.frac16,
                    "value_dictionary_entries": int(S._Ap._struct.n_dict),
                    "note": ("achieved/frac use the metric's CSR byte count (BASELINE.md); frac > 1 means the "
                             "kernel moves fewer bytes than CSR (lossless column/value codes) -- see stored_* "
                             "and traffic for the bytes actually moved") if sb < b_p else None}

    if roofline and N == 128 and args.udeg == 2:
        tr = pmc_traffic()
        if tr:
            roofline["traffic"], roofline["traffic_source"] = tr[0], "profiles/" + tr[1]
    transport_check = None
    if world > 1:  # the transports once more after the timed steps: every ghost dof must still get its owner's value
        S._Vi[0][0].check_halo()
        S._Q.check_halo()
        transport_check = "halo self-test passed after the timed steps"
    nnz_glob = [Pu.nnz, Pp.nnz]
    if world > 1:
        import torch.distributed as dist

        tn = torch.tensor(nnz_glob, dtype=torch.int64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tn)
        nnz_glob = [int(v) for v in tn.tolist()]
    if rank == 0:
        mean_its = {k: float(np.mean([np.max(i[k]) if len(i[k]) else 0 for i in its])) for k in its[0]}
        out = {
            "metric": "time-steps/sec, 3D Taylor-Green %d^3 P2-P1 (with pressure-CG SpMV GB/s in roofline)" % N
            if args.udeg == 2 else "time-steps/sec, 3D Taylor-Green %d^3 P%d-P1" % (N, args.udeg),
            "value": args.steps / elapsed, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"3D Taylor-Green {N}^3x6 tets P{args.udeg}-P1, nu={nu}, dt={dt:g}, "
                                   f"bcgs+jacobi / cg+jacobi rtol={args.rtol:g} atol=1e-14 "
                                   f"initial_guess_nonzero={not args.zero_guess}, max_iter=1, "
                                   f"low_memory_version={args.matrix_free}",
                       "cells": mesh.num_cells, "n_u_per_component": S._Vi[0][0].num_dofs_global,
                       "n_p": S._Q.num_dofs_global,
                       "nnz_velocity": nnz_glob[0], "nnz_pressure": nnz_glob[1], "parallelism": f"mesh-partition x{world}",
                       "transport": (None if world == 1 else
                                     "/".join(sorted(set(comm.active.values()))) or "none")
                       + ("" if args.backend == "nccl" or world == 1 else " (rehearsal: torch.distributed over gloo)")
                       if world > 1 else None,
                       "transport_check": transport_check},
            "cg_spmv_gbs": roofline["achieved"] if roofline else None,  # the metric's second figure
            "roofline": roofline,
            "krylov_iterations_per_step": mean_its,
            "phase_ms_per_step": phase_ms,  # device time between events around each phase method (rank 0)
            # whole Jacobi-CG iteration of the pressure solve in the metric's bytes (SURVEY.md 8d:
            # SpMV + 16 vector passes), over the measured time per iteration
            "pressure_cg_iteration": ({"us": 1e3 * phase_ms["pressure_solve"] / max(mean_its["pressure"], 1.0),
                                       "algorithmic_gbs": (b_p + 128 * Pp.n_rows) * max(mean_its["pressure"], 1.0)
                                       / (1e6 * phase_ms["pressure_solve"])}
                                      if phase_ms.get("pressure_solve") else None),
            "accuracy": {"max_nodal_error_u_vs_analytic": err_u, "t_end": clock["t"]},
            "kernels": kernels,
            "setup_s": t_setup,
            "hbm_gib": torch.cuda.max_memory_allocated() / 2 ** 30,
            "cpu_baseline": None,  # timed on rank 0 at N = 1 only (below)
        }
        if not args.no_cpu and world == 1:
            try:
                from oracle.cpu_baseline import run_cpu_baseline

                ksp_cpu = {"rtol": args.rtol, "atol": 1e-14, "max_it": 10000, "guess": not args.zero_guess}
                out["cpu_baseline"] = run_cpu_baseline(
                    S, clock, dt, nu, ksp_cpu, lambda X, t: np.stack([f(X, t) for f in fns]), gpu_step=step,
                    mesh_def=([-1.0, -1.0, -1.0], [1.0, 1.0, 1.0], [N, N, N]), threads_1=not args.no_cpu_one_core)
                out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            except Exception as e:  # the baseline is a reported figure, never the product path
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
