#!/usr/bin/env python3
"""bench.py -- time-steps/sec and pressure-CG SpMV GB/s of the IPCS hot path on MI355X.

Default workload (BASELINE.json configs[2]): 3-D Taylor-Green on [-1,1]^3, N^3 x 6 tetrahedra
(N = 128), P2-P1 Taylor-Hood, z-extruded analytic Taylor-Green field with exact Dirichlet data on
every face, nu = 0.01, dt = 0.005*32/N, max_iter = 1, BiCGStab+Jacobi tentative velocity, CG+Jacobi
pressure and velocity update, rtol 1e-8 / atol 1e-14.  A "step" is one FractionalStep_AB_CN.solve().
Synthetic data, float64 throughout.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

Other workloads: ``--workload beltrami`` (Ethier-Steinman, w != 0: SURVEY.md 8d's stronger field),
``--workload cavity`` (BASELINE.json configs[3]: unit cube, lid u = (1,0,0) on z = 1, nu = 1e-3,
dt = 1/N, from rest).

Prints ONE compact JSON line (<= 8000 bytes) on rank 0's stdout -- the contract's keys, `roofline`, `cpu_baseline`,
`headline_petsc_default` -- as soon as the headline, the counter passes and the CPU baseline exist, and NOTHING after it.
The full record (kernel tables, Krylov series, variant legs, predictions) goes to ``gpurun_out/bench_full.json``
(``--full-out``).  The heavy legs (Beltrami / cavity / Delaunay workloads with their host cross-checks, the one-core CPU
repetition) run only with ``--extras``, after the line is out, and land in the side file.
  roofline      the pressure-Poisson CG SpMV.  `traffic` = HBM bytes per launch from two rocprofv3 --pmc passes
                (FETCH_SIZE x 2, WRITE_SIZE) of this same command, run as child processes before this process touches
                the GPU; `avg_launch_us` = the same kernel's duration in a third child pass (--kernel-trace);
                `achieved` = traffic / that time, `frac` = achieved / 8 TB/s.  `frac_algorithmic` prices the same
                launch at SURVEY.md 8d's CSR bytes (12 B per nonzero); `hip_event` = the stored bytes over the HIP-event
                time measured inside the timed region; `past_cache` = the same kernel on the 256^3 pressure matrix
                (0.9 GB stored: beyond the 256 MB Infinity Cache), with its own counters from the same child passes;
                `f64_values` = the same launch on the same matrix with the value dictionary off (plain f64 value stream,
                HBM-resident), counters from a second set of child passes (--no-dictionary)
  headline_petsc_default  the same timed steps with PETSc's default zero initial guess, and with the value
                dictionaries off (what a mesh without bit-identical cells gets)
  cpu_baseline  one step of the same workload on the host cores (oracle/ipcs_cpu.c), all cores (one core too with
                --extras), set up from the mesh definition alone
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL and the xGMI windows share device
# memory between the ranks); exported by the launcher normally -- make sure before HIP starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def spmv_bytes(nnz, n_rows, n_cols):
    """Bytes of one CSR SpMV, f64 values + int32 columns (SURVEY.md 8d / BASELINE.md)."""
    return 12 * nnz + 4 * (n_rows + 1) + 8 * n_cols + 8 * n_rows


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)  # the driver's command line: --steps 20 --warmup 5
    ap.add_argument("--warmup", type=int, default=5)  # (the first steps after the start need more Krylov iterations)
    ap.add_argument("-N", type=int, default=128, help="cubes per direction")
    ap.add_argument("--udeg", type=int, default=2)
    ap.add_argument("--pdeg", type=int, default=None, help="pressure degree (default 1; 2 with --udeg 3: Taylor-Hood P3-P2)")
    ap.add_argument("--rtol", type=float, default=1e-8)
    ap.add_argument("--workload", default="tg", choices=["tg", "beltrami", "cavity"])
    ap.add_argument("--zero-guess", action="store_true",
                    help="PETSc default: zero the solution before every Krylov solve "
                         "(default here: -ksp_initial_guess_nonzero, the previous field is the guess)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-cpu-one-core", action="store_true",
                    help="(kept for old command lines; the 1-core repetition of the cpu_baseline step -- about a minute at "
                         "128^3 -- runs only with --extras, and then unless this flag is given)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child passes (roofline.traffic = null)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the cheap variant legs (steady state, zero guess, dictionaries off) and the 256^3 past-cache SpMV")
    ap.add_argument("--extras", action="store_true",
                    help="AFTER the JSON line is out: the Beltrami / cavity / refined-Delaunay workload legs with their host "
                         "cross-checks and the one-core cpu_baseline repetition; results go to the --full-out side file only")
    ap.add_argument("--full-out", default=None,
                    help="path of the full record (default gpurun_out/bench_full.json under the repository root)")
    ap.add_argument("--probe-transports", action="store_true",
                    help="N > 1: AFTER the JSON line is out, time both device transports' halo exchanges (brings up the "
                         "xGMI-window transport on a temporary plan); results go to stderr and the side file")
    ap.add_argument("--pmc-leg", default=None, help=argparse.SUPPRESS)  # child of an --extras leg's counter passes
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cg-merged", default=None, choices=["true", "false"],
                    help="force the merged-reduction CG (OX_KSP_CG_MERGED: one synchronisation point, three kernels per "
                         "iteration) for the one-column pressure solve")
    ap.add_argument("--cg-fold-blocks", type=int, default=None,
                    help="blocks of the folded one-column CG update kernels (solver option ksp_cg_fold_blocks: 0 = the five-kernel "
                         "iteration, default one block per compute unit)")
    ap.add_argument("--cg-single-reduction", default=None, choices=["true", "false"],
                    help="force -ksp_cg_single_reduction for the CG solves (default: true on partitioned operators only)")
    ap.add_argument("--bcgs-merged", default=None, choices=["true", "false"],
                    help="force the merged-reduction BiCGStab (default: true on partitioned operators only)")
    ap.add_argument("--matrix-free", action="store_true",
                    help="low_memory_version=True: matrix-free vector kernels for the p*, div(u) and grad(phi) "
                         "terms instead of the pre-assembled rectangular operators (reference "
                         "fracstep.py:392-404; the demo's default is the pre-assembled form)")
    ap.add_argument("--no-dictionary", action="store_true", help="options['value_dictionary'] = False for the main run")
    ap.add_argument("--spmv-windows", default=None, choices=["true", "false"],
                    help="force the LDS-window stream of the velocity matrices on / off (default: on for P2 on one GPU)")
    ap.add_argument("--window", type=int, default=None,
                    help="rows per length-sorting window of the SELL-64 numbering (tuning; default: the library's)")
    ap.add_argument("--profile-setup", action="store_true",
                    help="cProfile the set-up phase and print rank 0's top entries to stderr")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1: nccl = RCCL over xGMI (the product path); "
                         "gloo = rehearsal of the same partitioned path through the library's host-staged "
                         "callback transport (several ranks may then share one GPU)")
    ap.add_argument("--mesh", default="box", choices=["box", "delaunay"],
                    help="box: the N^3 x 6 tetrahedra of the BASELINE configurations; delaunay: Delaunay triangulation of "
                         "a jittered (N+1)^3 lattice (unstructured: no value dictionaries, Z-order numbering); implies --no-extras")
    ap.add_argument("--refine", type=int, default=0,
                    help="--mesh delaunay: levels of uniform refinement of the triangulation (Qhull takes minutes beyond "
                         "~3e5 points; -N 32 --refine 2 = 14.0 M tetrahedra, 18.9 M P2 dofs per component in seconds)")
    ap.add_argument("--profile-every", type=int, default=32,
                    help="HIP-event pair around every n-th launch of a kernel tag inside the timed region")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    if a.pdeg is None:
        a.pdeg = 2 if a.udeg >= 3 else 1
    return a


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


# ---- HBM traffic and kernel durations: rocprofv3 child passes of this command ---------------------
def rocprof_passes(child_argv, log, label, budget_s=240):
    """Run ``bench.py --pmc-child <child_argv>`` (1 warm-up + 1 timed step, then -- box meshes -- the 256^3 pressure
    SpMV) three times: under ``rocprofv3 --pmc FETCH_SIZE``, ``--pmc WRITE_SIZE`` (separate passes, as
    MI355X_MICROARCH.md's HBM section prescribes) and ``--kernel-trace``.  Returns the per-(kernel, grid) averages of all
    three and the child's own record (which names the pressure SpMV's kernel and grid).
    Must run BEFORE this process touches the GPU (the children are ordinary child processes of a process that has
    no HIP state; the interpreter after ``--`` is the real binary, so nothing re-execs behind the profiler's preloaded
    library).  FETCH_SIZE counts 128-B requests as 64 B on gfx950: a streamed read is 2 x FETCH_SIZE (same section)."""
    import csv
    import glob

    exe = shutil.which("rocprofv3")
    if exe is None:
        return {"error": "rocprofv3 not found"}
    tmp = tempfile.mkdtemp(prefix="ox_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    py = os.path.realpath(sys.executable)
    table, child = {}, None
    t0 = time.perf_counter()

    def key(name, grid):
        return name.split("(")[0].replace(" ", ""), int(grid)

    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE", "KERNEL_TRACE"):
            out = os.path.join(tmp, counter)
            mode = ["--kernel-trace"] if counter == "KERNEL_TRACE" else ["--pmc", counter]
            cmd = [exe, *mode, "--output-format", "csv", "-d", out, "--", py, os.path.join(ROOT, "bench.py"), "--pmc-child",
                   *child_argv]
            # bounded: a pass that hangs must not cost the run its line (the driver's limit is 600 s for everything).  The
            # pass runs in its own process group, so a time-out takes rocprofv3 AND the python under it down (by pid group,
            # never by pattern)
            import signal

            # (``budget_s`` covers the three passes together: the first may pay a cold start of the interpreter and its
            # libraries, the others are then quick)
            pass_timeout = max(10.0, budget_s - (time.perf_counter() - t0))
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
            try:
                so, se = proc.communicate(timeout=pass_timeout)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.communicate()
                return {"error": f"rocprofv3 {counter} pass of {label} did not finish within its share ({pass_timeout:.0f} s) of the {budget_s} s budget (killed)"}
            if proc.returncode != 0:
                return {"error": f"rocprofv3 {counter} pass of {label} failed (rc {proc.returncode}): "
                                 + se.decode(errors="replace")[-300:]}
            for line in so.decode(errors="replace").splitlines():
                if line.startswith("{") and '"pmc_child"' in line:
                    child = json.loads(line)
            if child is None:
                return {"error": f"rocprofv3 {counter} pass of {label}: the child printed no record"}
            acc = {}
            if counter == "KERNEL_TRACE":
                for f in glob.glob(os.path.join(out, "**", "*_kernel_trace.csv"), recursive=True):
                    for r in csv.DictReader(open(f)):
                        acc.setdefault(key(r["Kernel_Name"], r["Grid_Size_X"]), []).append(
                            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
                for k, v in acc.items():
                    table.setdefault(k, {}).update({"dispatches": len(v), "avg_us": sum(v) / len(v)})
                continue
            for f in glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == counter:
                        acc.setdefault(key(r["Kernel_Name"], r["Grid_Size"]), []).append(float(r["Counter_Value"]))
            if not acc:
                return {"error": f"rocprofv3 --pmc {counter} pass of {label}: no counter rows"}
            for k, v in acc.items():
                table.setdefault(k, {}).update({counter + "_KiB": sum(v) / len(v), "pmc_dispatches": len(v)})
    except Exception as e:  # the counters are a reported figure: never fail the bench for them
        return {"error": repr(e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for v in table.values():
        if "FETCH_SIZE_KiB" in v and "WRITE_SIZE_KiB" in v:
            v["read_bytes"] = 2.0 * v["FETCH_SIZE_KiB"] * 1024.0
            v["write_bytes"] = v["WRITE_SIZE_KiB"] * 1024.0
            v["traffic"] = v["read_bytes"] + v["write_bytes"]
    log(f"rocprofv3 passes of {label}: {time.perf_counter() - t0:.1f} s")
    return {"kernels": table, "child": child, "seconds": time.perf_counter() - t0,
            "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --kernel-trace child passes of this command "
                      "(FETCH_SIZE x 2: gfx950 counts 128-B requests as 64 B)"}


def child_argv_of(args):
    """The flags that make a --pmc-child run the same workload as this command."""
    av = ["-N", str(args.N), "--udeg", str(args.udeg), "--pdeg", str(args.pdeg), "--workload", args.workload,
          "--steps", "1", "--warmup", "1", "--rtol", str(args.rtol)]
    for flag, on in (("--zero-guess", args.zero_guess), ("--matrix-free", args.matrix_free),
                     ("--no-dictionary", args.no_dictionary), ("--no-extras", args.no_extras)):
        if on:
            av.append(flag)
    if args.window:
        av += ["--window", str(args.window)]
    if args.spmv_windows:
        av += ["--spmv-windows", args.spmv_windows]
    if args.mesh != "box":
        av += ["--mesh", args.mesh, "--refine", str(args.refine)]
    return av


def pick_kernel(passes, prefix, grid, epi=None):
    """The (kernel, grid) entry of a rocprof_passes() table whose demangled name starts with ``prefix``.  With ``epi``
    (the pressure CG's epilogue id) and no exact match -- a matrix on the LDS-window stream launches ``k_spmv_win`` on a
    grid the host does not know --: the MOST DISPATCHED one-column mat-vec with that epilogue (the pressure solve's: 200-1000
    launches per step against a handful of narrowed mass products)."""
    import re

    if not passes or "error" in passes:
        return None
    for (name, g), v in passes["kernels"].items():
        if g == grid and name.startswith(prefix) and "traffic" in v and "avg_us" in v:
            return dict(v, kernel=name, grid=g)
    if epi is None:
        return None
    pat = re.compile(r"^voidk_spmv(_ps|_win)?<1,%d[,>]" % int(epi))
    best = None
    for (name, g), v in passes["kernels"].items():
        if pat.match(name) and "traffic" in v and "avg_us" in v and (best is None or v["dispatches"] > best["dispatches"]):
            best = dict(v, kernel=name, grid=g)
    return best


def rocprof_kernel_table(passes, n=14):
    """The n kernels of a rocprof_passes() table with the most time, with counter-derived fractions of the HBM peak."""
    if not passes or "error" in passes:
        return passes
    rows = []
    for (name, g), v in passes["kernels"].items():
        if "avg_us" in v and "traffic" in v and ("k_spmv" in name or "k_assemble_rows" in name or "k_cg" in name or "k_bcgs" in name):
            rows.append({"kernel": name.replace("void", "", 1), "grid": g, "dispatches": v["dispatches"], "avg_us": v["avg_us"],
                         "traffic_MB": v["traffic"] / 1e6, "gbs": v["traffic"] / (1e3 * v["avg_us"]),
                         "frac": v["traffic"] / (1e3 * v["avg_us"]) / HBM_PEAK_GBS})
    rows.sort(key=lambda r: -r["avg_us"] * r["dispatches"])
    return rows[:n]


# ---- workloads ----------------------------------------------------------------------------------
def make_workload(name, N, np, torch):
    """Analytic fields as array-API callables (numpy on the host, torch tensors on the device for the
    per-step Dirichlet values).  ``N``: cells per direction the time step is sized for (dt ~ 1/N)."""
    pi = math.pi

    def xp(x):
        return torch if torch.is_tensor(x) else np

    if name == "tg":
        nu, dt = 0.01, 0.005 * 32.0 / N
        box = ([-1.0, -1.0, -1.0], [1.0, 1.0, 1.0])
        fns = [lambda x, t: -xp(x).cos(pi * x[0]) * xp(x).sin(pi * x[1]) * math.exp(-2.0 * nu * pi ** 2 * t),
               lambda x, t: xp(x).cos(pi * x[1]) * xp(x).sin(pi * x[0]) * math.exp(-2.0 * nu * pi ** 2 * t),
               lambda x, t: xp(x).zeros_like(x[0])]

        def pres(x, t):
            return -0.25 * (np.cos(2 * pi * x[0]) + np.cos(2 * pi * x[1])) * math.exp(-4.0 * nu * pi ** 2 * t)
        desc = "3D Taylor-Green (z-extruded analytic field, exact Dirichlet data)"
    elif name == "beltrami":
        nu, dt = 0.01, 0.005 * 32.0 / N
        box = ([-1.0, -1.0, -1.0], [1.0, 1.0, 1.0])
        a, d = pi / 4.0, pi / 2.0

        def comp(i, j, k):  # -a [e^{a x_i} sin(a x_j + d x_k) + e^{a x_k} cos(a x_i + d x_j)] e^{-nu d^2 t}
            return lambda x, t: -a * (xp(x).exp(a * x[i]) * xp(x).sin(a * x[j] + d * x[k])
                                      + xp(x).exp(a * x[k]) * xp(x).cos(a * x[i] + d * x[j])) * math.exp(-nu * d * d * t)
        fns = [comp(0, 1, 2), comp(1, 2, 0), comp(2, 0, 1)]

        def pres(x, t):
            X, Y, Z = x[0], x[1], x[2]
            s = (np.exp(2 * a * X) + np.exp(2 * a * Y) + np.exp(2 * a * Z)
                 + 2 * np.sin(a * X + d * Y) * np.cos(a * Z + d * X) * np.exp(a * (Y + Z))
                 + 2 * np.sin(a * Y + d * Z) * np.cos(a * X + d * Y) * np.exp(a * (Z + X))
                 + 2 * np.sin(a * Z + d * X) * np.cos(a * Y + d * Z) * np.exp(a * (X + Y)))
            return -0.5 * a * a * s * math.exp(-2.0 * nu * d * d * t)
        desc = "3D Ethier-Steinman Beltrami flow (a = pi/4, d = pi/2, exact Dirichlet data)"
    else:
        nu, dt = 1e-3, 1.0 / N
        box = ([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])

        def lid(x, t):
            on = x[2] > 1.0 - 1e-12
            return on.to(torch.float64) if torch.is_tensor(x) else on.astype(np.float64)
        fns = [lid, lambda x, t: xp(x).zeros_like(x[0]), lambda x, t: xp(x).zeros_like(x[0])]
        pres = None
        desc = "3D lid-driven cavity, unit cube, lid u = (1,0,0) on z = 1, Re = 1000, from rest"
    return {"nu": nu, "dt": dt, "box": box, "fns": fns, "p": pres, "desc": desc, "analytic": name != "cavity"}


# ---- the line the driver reads -------------------------------------------------------------------
COMPACT_LIMIT = 8000  # bytes: the whole line fits the driver's stdout tail
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data")


def _round(x, sig=6):
    if isinstance(x, float):
        return x if x == 0.0 or not math.isfinite(x) else float(f"{x:.{sig}g}")
    if isinstance(x, dict):
        return {k: _round(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_round(v, sig) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_line(full, limit=COMPACT_LIMIT):
    """The ONE stdout line: the contract's keys, `roofline`, `cpu_baseline`, `headline_petsc_default` and a few small
    tables, from the full record -- a pure function of it (tests/test_bench_contract.py feeds it the committed full
    records).  Optional blocks are dropped, last first, until the line is within ``limit`` bytes."""
    out = {k: full[k] for k in CONTRACT_KEYS}
    cfg = full["config"]
    out["config"] = _pick(cfg, ("workload", "cells", "n_u_per_component", "n_p", "nnz_velocity", "nnz_pressure", "parallelism",
                                "transport", "transport_check", "rccl_nranks", "launched_by", "krylov_sync_points_per_iteration"))
    if cfg.get("ranks"):  # N > 1: every rank's share, without the communicator reports
        out["config"]["ranks"] = [_pick(r, ("rank", "device", "cells", "velocity_rows", "velocity_ghosts", "pressure_rows",
                                            "pressure_ghosts", "peers")) for r in cfg["ranks"]]
    else:
        out["config"]["ranks"] = None
    r = full.get("roofline")
    if r:
        rr = _pick(r, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "basis", "avg_launch_us",
                       "algorithmic_bytes_per_launch", "achieved_algorithmic", "frac_algorithmic", "frac_algorithmic_note",
                       "stored_bytes_per_launch", "traffic_over_stored", "hip_event", "counters_error"))
        if isinstance(r.get("f64_values"), dict):
            rr["f64_values"] = _pick(r["f64_values"], ("what", "achieved", "frac", "traffic", "avg_launch_us", "stored_bytes_per_launch",
                                                       "traffic_over_stored", "hip_event", "counters_error"))
        if isinstance(r.get("past_cache"), dict):
            rr["past_cache"] = _pick(r["past_cache"], ("workload", "achieved", "frac", "traffic", "basis", "avg_launch_us",
                                                       "stored_bytes_per_launch", "algorithmic_bytes_per_launch",
                                                       "frac_algorithmic", "hip_event", "error"))
        out["roofline"] = rr
    else:
        out["roofline"] = None
    c = full.get("cpu_baseline")
    if isinstance(c, dict):
        cc = _pick(c, ("value", "unit", "cores", "kind", "sample", "cpu_model", "seconds", "setup_seconds", "gpu_over_cpu",
                       "gpu_vs_cpu_rel_l2_u", "gpu_vs_cpu_rel_l2_p", "gpu_vs_cpu_max_abs_u", "krylov_iterations", "error"))
        if isinstance(c.get("one_core"), dict):
            cc["one_core"] = _pick(c["one_core"], ("value", "seconds"))
        if isinstance(cc.get("sample"), str) and len(cc["sample"]) > 300:
            cc["sample"] = cc["sample"][:297] + "..."
        out["cpu_baseline"] = cc
    else:
        out["cpu_baseline"] = None
    optional = []  # (key, value): dropped from the END of this list first when the line is too long
    if full.get("headline_petsc_default"):
        optional.append(("headline_petsc_default", full["headline_petsc_default"]))
    for k in ("cg_spmv_gbs", "krylov_iterations_per_step", "phase_ms_per_step", "phase_ms_per_step_max_over_ranks"):
        if full.get(k) is not None:
            optional.append((k, full[k]))
    if full.get("pressure_cg_iteration"):
        optional.append(("pressure_cg_iteration", _pick(full["pressure_cg_iteration"], (
            "us", "bytes_moved", "stored_gbs", "frac_stored", "kernels_per_iteration", "recurrences"))))
    if full.get("accuracy"):
        optional.append(("accuracy", full["accuracy"]))
    if full.get("steady_state"):
        optional.append(("steady_state", _pick(full["steady_state"], ("value", "ms_per_step", "steps", "first_timed_step",
                                                                      "krylov_iterations_per_step"))))
    for k in ("setup_s", "hbm_gib", "full_record"):
        if full.get(k) is not None:
            optional.append((k, full[k]))
    if full.get("kernels"):  # velocity-side mat-vecs and the assembly, four numbers each
        optional.append(("kernels", {k: _pick(v, ("launches", "avg_us", "stored_gbs", "frac_stored"))
                                     for k, v in full["kernels"].items()}))
    if full.get("kernels_rocprofv3") and isinstance(full["kernels_rocprofv3"], list):
        optional.append(("kernels_rocprofv3", [_pick(k, ("kernel", "avg_us", "traffic_MB", "frac"))
                                               for k in full["kernels_rocprofv3"][:8]]))
    if full.get("krylov_iterations_series"):
        optional.append(("krylov_iterations_series", full["krylov_iterations_series"]))

    def render(n):
        d = dict(out)
        for k, v in optional[:n]:
            d[k] = v
        d = {k: (v if k in ("value", "ms_per_step") else _round(v)) for k, v in d.items()}
        return json.dumps(d, separators=(",", ":"))

    n = len(optional)
    line = render(n)
    while len(line.encode()) > limit and n > 0:
        n -= 1
        line = render(n)
    if len(line.encode()) > limit:  # (cannot happen with the key sets above; a bound is a bound)
        raise RuntimeError(f"bench.py: the contract line is {len(line.encode())} bytes (> {limit})")
    return line


def write_full(path, full):
    """The full record as one JSON line in the side file (rewritten whole: a later leg replaces the earlier copy)."""
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path + ".tmp", "w") as f:
            f.write(json.dumps(full) + "\n")
        os.replace(path + ".tmp", path)
        return path
    except OSError as e:
        print(f"[bench] could not write {path}: {e!r}", file=sys.stderr, flush=True)
        return None


def predicted_line(args, world):
    """What profiles/r05_predicted_scaling.json predicted for this workload, mesh size and rank count (None: no prediction
    on file for it).  A PREDICTION made from one-GPU measurements (tools/predict_scaling.py), never a measurement."""
    if args.workload != "tg" or args.mesh != "box" or args.udeg != 2:
        return None
    try:
        path = os.path.join(ROOT, "profiles", "r05_predicted_scaling.json")
        for line in open(path).read().splitlines():
            d = json.loads(line)
            if f"{args.N}^3" in d.get("workload", "") and str(world) in d.get("P", {}):
                e = d["P"][str(world)]
                return {"label": d["label"], "steps_per_s": e["steps_per_s"], "ms_per_step": e["ms_per_step"],
                        "phases_ms": e["phases_ms"], "parallel_efficiency": e.get("parallel_efficiency"),
                        "assumed": d["assumed"], "source": "profiles/r05_predicted_scaling.json"}
    except Exception:
        return None
    return None


def self_launch(args):
    """``python bench.py --gpus N`` without a launcher (no WORLD_SIZE in the environment): this process -- which has
    not touched the GPU -- starts the driver's own N > 1 command line as a CHILD process (one rank per GPU under
    ``torch.distributed.run``), relays its output and returns its exit code."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print("[bench] --gpus %d without a launcher: starting %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=dict(os.environ, OX_BENCH_SELF_LAUNCHED="1")).returncode


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # a launch line whose rank count is not the one asked for must not print a figure
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks "
                         f"(use `python bench.py --gpus N`, or torch.distributed.run --nproc-per-node N ... --gpus N)")

    def log(*a):
        if args.verbose and rank == 0:
            print("[bench]", *a, file=sys.stderr, flush=True)

    # the rocprofv3 child passes come first: this process has not touched the GPU yet
    passes = leg_passes = None
    DELAUNAY_LEG = (32, 2)  # --extras: the unstructured leg's mesh (jittered 33^3 lattice, refined twice: 14.0 M tets)
    extras_legs = (world == 1 and args.extras and not args.pmc_child and args.workload == "tg" and args.mesh == "box")
    passes_f64 = None  # the same command with the value dictionaries off: the pressure SpMV on plain f64 values
    if world == 1 and not args.no_pmc and not args.pmc_child:
        passes = rocprof_passes(child_argv_of(args), log, "the headline workload")
        # (the second set only where the first one was quick: the line must be out long before the driver's limit)
        if (not args.no_extras and not args.no_dictionary and args.mesh == "box" and "error" not in passes
                and passes["seconds"] < 90.0):
            passes_f64 = rocprof_passes([a for a in child_argv_of(args) if a != "--no-extras"] + ["--no-dictionary", "--no-extras"],
                                        log, "the headline workload without value dictionaries", budget_s=120)
        if extras_legs and args.udeg == 2 and args.N >= 96:
            leg_passes = rocprof_passes(["-N", str(DELAUNAY_LEG[0]), "--mesh", "delaunay", "--refine", str(DELAUNAY_LEG[1]),
                                         "--workload", "beltrami", "--udeg", str(args.udeg), "--pdeg", str(args.pdeg),
                                         "--steps", "1", "--warmup", "1", "--rtol", str(args.rtol), "--no-extras"],
                                        log, "the unstructured leg", budget_s=600)
    full_path = args.full_out or os.path.join(ROOT, "gpurun_out", "bench_full.json")

    os.environ.setdefault("OMP_NUM_THREADS", str(host_cores()))  # cpu_baseline leg (oracle/ipcs_cpu.c)
    import numpy as np
    import torch

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
    N = args.N
    if args.mesh != "box":
        args.no_extras = True  # the 256^3 box line and the dictionary-off leg describe the box workload
    W = make_workload(args.workload, N << args.refine, np, torch)  # dt follows the refined mesh's spacing
    nu, dt, fns = W["nu"], W["dt"], W["fns"]
    clock = {"t": 0.0}
    p0, p1 = W["box"]

    def on_boundary(x):
        on = np.zeros(x.shape[1], dtype=bool)
        for k in range(3):
            on |= np.isclose(x[k], p0[k]) | np.isclose(x[k], p1[k])
        return on

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

    def bc_value(f):
        def g(x):
            return f(x, clock["t"])
        g.supports_torch = True
        return g

    def build(n, udeg, opts, zero_guess):
        if args.mesh == "delaunay":  # Delaunay triangulation of a jittered (n+1)^3 lattice: "what an unstructured mesh gets"
            mesh = M.create_delaunay_box(comm, [p0, p1], n, refine=args.refine)
        else:
            mesh = M.create_box(comm, [p0, p1], [n, n, n])
        bcs_u = [[ox.DirichletBC(bc_value(f), ox.LocatorMethod.GEOMETRICAL, on_boundary)] for f in fns]
        ksp = {"pc_type": "jacobi", "ksp_rtol": args.rtol, "ksp_atol": 1e-14, "ksp_max_it": 10000,
               "ksp_initial_guess_nonzero": not zero_guess}
        tent = dict(ksp, ksp_type="bcgs")
        if args.workload == "cavity":
            # a start from rest leaves the first BiCGStab residual on the identity rows only: rho = 0
            # after one iteration, where PETSc's KSPBCGS reports DIVERGED_BREAKDOWN; the documented
            # extension re-seeds the shadow residual instead (oasisx_amd/ksp.py)
            tent["ksp_bcgs_restarts"] = 5
        so = {"tentative": tent, "pressure": dict(ksp, ksp_type="cg"), "scalar": dict(ksp, ksp_type="cg")}
        if args.cg_single_reduction is not None:
            for k in ("pressure", "scalar"):
                so[k]["ksp_cg_single_reduction"] = args.cg_single_reduction == "true"
        if args.bcgs_merged is not None:
            so["tentative"]["ksp_bcgs_merged_reduction"] = args.bcgs_merged == "true"
        if args.cg_merged is not None:
            so["pressure"]["ksp_cg_merged_reduction"] = args.cg_merged == "true"
        if args.cg_fold_blocks is not None:
            for k in ("pressure", "scalar", "tentative"):
                so[k]["ksp_cg_fold_blocks"] = int(args.cg_fold_blocks)
        S_ = ox.FractionalStep_AB_CN(mesh, ("Lagrange", udeg), ("Lagrange", args.pdeg), bcs_u=bcs_u, bcs_p=[],
                                     solver_options=so, options=opts)
        return mesh, S_

    options = dict({"low_memory_version": args.matrix_free, "value_dictionary": not args.no_dictionary},
                   **({"sell_window": args.window} if args.window else {}),
                   **({"spmv_windows": args.spmv_windows == "true"} if args.spmv_windows else {}),
                   **({"spmv_windows_pressure": os.environ["OX_WIN_P"] == "1"} if os.environ.get("OX_WIN_P") else {}))
    mesh, S = build(N, args.udeg, options, args.zero_guess)

    def workload_leg(wname, steps=5, warmup=3, delaunay=None):
        """Another workload with the main run's mesh size, options and Krylov settings: its own solver, set up,
        warmed up and timed here (barrier + synchronize on both sides), then released.  ``delaunay=(n, refine)``: on
        an UNSTRUCTURED mesh of the same size instead -- the Delaunay triangulation of a jittered (n+1)^3 lattice,
        refined uniformly ``refine`` times (32, 2: 14.0 M tetrahedra, 18.9 M P2 dofs per component; no value
        dictionaries, Z-order numbering, the velocity mat-vecs on the LDS-window stream)."""
        W2 = make_workload(wname, N if delaunay is None else delaunay[0] << delaunay[1], np, torch)
        q0, q1 = W2["box"]
        clk = {"t": 0.0}

        def on_bnd(x):
            on = np.zeros(x.shape[1], dtype=bool)
            for k in range(3):
                on |= np.isclose(x[k], q0[k]) | np.isclose(x[k], q1[k])
            return on

        def bcv(f):
            def g(x):
                return f(x, clk["t"])
            g.supports_torch = True
            return g

        t0 = time.perf_counter()
        if delaunay is None:
            m2 = M.create_box(comm, [q0, q1], [N, N, N])
        else:
            m2 = M.create_delaunay_box(comm, [q0, q1], delaunay[0], refine=delaunay[1])
        t_mesh = time.perf_counter() - t0
        ksp = {"pc_type": "jacobi", "ksp_rtol": args.rtol, "ksp_atol": 1e-14, "ksp_max_it": 10000,
               "ksp_initial_guess_nonzero": not args.zero_guess}
        tent = dict(ksp, ksp_type="bcgs")
        if wname == "cavity":
            tent["ksp_bcgs_restarts"] = 5  # a start from rest: see build()
        S2 = ox.FractionalStep_AB_CN(
            m2, ("Lagrange", args.udeg), ("Lagrange", args.pdeg),
            bcs_u=[[ox.DirichletBC(bcv(f), ox.LocatorMethod.GEOMETRICAL, on_bnd)] for f in W2["fns"]], bcs_p=[],
            solver_options={"tentative": tent, "pressure": dict(ksp, ksp_type="cg"), "scalar": dict(ksp, ksp_type="cg")},
            options=options)
        if W2["analytic"]:
            for i, f in enumerate(W2["fns"]):
                S2._u2[i].interpolate(at(f, -W2["dt"]))
                S2._u1[i].interpolate(at(f, 0.0))
            S2._p.interpolate(lambda x: W2["p"](x, -W2["dt"] / 2.0))
        torch.cuda.synchronize()
        t_set = time.perf_counter() - t0
        its2 = []

        def one():
            clk["t"] += W2["dt"]
            S2.solve(W2["dt"], W2["nu"], max_iter=1)

        for _ in range(warmup):
            one()
        torch.cuda.synchronize()
        _lib.check(lib.ox_profile_begin(200000, args.profile_every), "ox_profile_begin")
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
            its2.append(S2.iteration_counts())
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        _lib.check(lib.ox_profile_end(), "ox_profile_end")
        leg_kernels = kernel_table(S2, bool(S2._solver_p._cg_merged()))
        mesh_txt = (f"{N}^3x6 tets" if delaunay is None else
                    f"Delaunay mesh of a jittered {delaunay[0] + 1}^3 lattice refined uniformly {delaunay[1]}x "
                    f"({m2.num_cells} tets, {S2._Vi[0][0].num_dofs_global} P2 dofs per component)")
        res = {"value": steps / el, "unit": "steps/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "warmup": warmup,
               "workload": f"{W2['desc']}, {mesh_txt} P{args.udeg}-P{args.pdeg}, nu={W2['nu']}, dt={W2['dt']:g}",
               "krylov_iterations_per_step": mean_iterations(its2), "krylov_iterations_series": iteration_series(its2),
               "setup_s": t_set, "mesh_generation_s": t_mesh,
               # stored bytes over the HIP-event time; the counter-derived fractions of the same workload are in
               # kernels_rocprofv3 (unstructured leg: child passes run before this process touched the GPU)
               "kernels": {k: {a: b for a, b in v.items() if a in ("launches", "avg_us", "bytes_moved", "stored_gbs", "frac_stored")}
                           for k, v in leg_kernels.items()}}
        if delaunay is not None and leg_passes is not None:
            res["kernels_rocprofv3"] = rocprof_kernel_table(leg_passes)
        if delaunay is not None:
            P2_ = S2._M.pattern
            res["storage"] = {"nnz_velocity": P2_.nnz, "slots": P2_.size, "padding": P2_.size / P2_.nnz - 1.0,
                              "cols16_fraction": P2_.frac16, "value_dictionary": S2._M.vcode is not None,
                              "spmv_windows": getattr(P2_, "w_stats", None)}
        if W2["analytic"]:
            Xd2 = S2._Vi[0][0].x[: S2._n_u].T
            res["max_nodal_error_u_vs_analytic"] = max(
                float((S2._U.rdev()[: S2._n_u][:, i] - f(Xd2, clk["t"])).abs().max()) for i, f in enumerate(W2["fns"]))
        if not args.no_cpu:
            # the next step of THIS workload on the host (oracle/ipcs_cpu.c on its own mesh, numbering and operators,
            # fed the device's state through the dof coordinates) against the device's: a timed figure with its check
            try:
                from oracle.cpu_baseline import run_cpu_baseline

                chk = run_cpu_baseline(
                    S2, clk, W2["dt"], W2["nu"], {"rtol": args.rtol, "atol": 1e-14, "max_it": 10000, "guess": not args.zero_guess},
                    lambda X, t: np.stack([np.asarray(f(X, t), dtype=np.float64) for f in W2["fns"]]), gpu_step=one,
                    mesh_def=((q0, q1, [N, N, N]) if delaunay is None else
                              {"coords": m2.coords.cpu().numpy(), "cells": m2.cells.cpu().numpy(), "lo": q0, "hi": q1}),
                    threads_1=False, scipy_check=False, reuse_setup=True)
                res["cpu_cross_check"] = {k: chk[k] for k in (
                    "gpu_vs_cpu_rel_l2_u", "gpu_vs_cpu_rel_l2_p", "gpu_vs_cpu_max_abs_u", "gpu_vs_cpu_shared", "value", "cores",
                    "seconds", "setup_seconds", "setup_reused", "krylov_iterations", "gpu_krylov_iterations") if k in chk}
                res["cpu_cross_check"]["step"] = warmup + steps + 1
            except Exception as e:
                res["cpu_cross_check"] = {"error": repr(e)}
        del S2, m2
        torch.cuda.empty_cache()
        return res

    def at(f, t):  # the analytic fields are array-API callables: evaluated on the device
        def g(x):
            return f(x, t)
        g.supports_torch = True
        return g

    def set_initial_state():
        clock["t"] = 0.0
        if W["analytic"]:
            for i, f in enumerate(fns):
                S._u2[i].interpolate(at(f, -dt))
                S._u1[i].interpolate(at(f, 0.0))
            S._p.interpolate(lambda x: W["p"](x, -dt / 2.0))

    set_initial_state()
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

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    def timed_run(steps, warmup, main_run):
        """W untimed + K timed steps, barrier + synchronize on both sides; returns seconds, iterations."""
        for _ in range(warmup):
            step()
            log("warmup step", S.iteration_counts())
        barrier()
        # HIP events on the launching stream around every --profile-every-th launch of each kernel tag (an
        # event pair costs ~12 us of stream bubbles; every launch would slow the 75 us pressure iteration by 16 %)
        _lib.check(lib.ox_profile_begin(200000, args.profile_every), "ox_profile_begin")
        phase_events["_on"] = main_run
        t0 = time.perf_counter()
        its = []
        for _ in range(steps):
            step()
            its.append(S.iteration_counts())
        barrier()
        el = time.perf_counter() - t0
        phase_events["_on"] = False
        _lib.check(lib.ox_profile_end(), "ox_profile_end")
        if world > 1:
            import torch.distributed as dist

            tmax = torch.tensor([el], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        return el, its

    def last_solves():
        out = {}
        for name, ksp, nc in (("tentative", S._solver_u, mesh.gdim), ("pressure", S._solver_p, 1), ("update", S._solver_c, mesh.gdim)):
            r = getattr(ksp, "last_result", None)
            if r is not None:
                out[name] = {"iterations": [int(r.its[c]) for c in range(nc)], "bnorm": [float(r.bnorm[c]) for c in range(nc)],
                             "rnorm": [float(r.rnorm[c]) for c in range(nc)], "reason": [int(r.reason[c]) for c in range(nc)]}
        return out

    def mean_iterations(its):
        return {k: float(np.mean([np.max(i[k]) if len(i[k]) else 0 for i in its])) for k in its[0]}

    def iteration_series(its):
        """Per timed step: the lockstep's iteration count of each solve (the slowest column)."""
        return {k: [int(np.max(i[k])) if len(i[k]) else 0 for i in its] for k in its[0]}

    elapsed, its = timed_run(args.steps, args.warmup, True)
    phase_ms = {k: sum(a.elapsed_time(b) for a, b in v) / args.steps for k, v in phase_events.items() if k != "_on"}
    # accuracy at the end of the timed steps: nodal error against the analytic field (rank-local dofs)
    Xd = S._Vi[0][0].x[: S._n_u].T
    Ud = S._U.rdev()[: S._n_u]
    err_u = max(float((Ud[:, i] - f(Xd, clock["t"])).abs().max()) for i, f in enumerate(fns)) if W["analytic"] else None
    umax = float(Ud.abs().max())

    def prof_get(tag, key=-1):
        cnt, ms = C.c_longlong(0), C.c_double(0.0)
        _lib.check(lib.ox_profile_get(tag, key, C.byref(cnt), C.byref(ms)), "ox_profile_get")
        return int(cnt.value), float(ms.value)

    def stored_bytes(A, nc=1):
        """Bytes one SpMV launch streams from this SELL-64 storage: f64 values (or 1-byte value codes
        where the matrix has a dictionary), 16-bit column codes where the pattern carries them (int32
        elsewhere), the per-pair bases, slice offsets and the x / y vectors -- padding included.  A matrix
        with a pair-slot stream is read from that instead (4 B per slot of two adjacent columns)."""
        P = A.pattern
        if getattr(A, "ps_code", None) is None and getattr(P, "wcode", None) is not None and A._struct.n_wblocks > 0:
            # LDS-window stream: values (f64, or 1-byte codes in the tile layout), 2-byte window indices (tile layout),
            # the window lists, x[window] per block, y
            tiles = P.wcode.numel()
            vals = tiles if getattr(A, "wvcode", None) is not None else 8 * P.size
            return int(vals + 2 * tiles + 4 * P.wlist.numel() + 8 * (P.n_slices + 1) + 8 * (P.n_wblocks + 1)
                       + nc * 8 * (P.wlist.numel() + P.n_rows))
        if getattr(A, "ps_code", None) is not None:  # pair-slot stream: 4 B per slot, per-group bases, slice offsets
            n = A.ps_code.numel()
            return int(4 * n + 8 * (n // 256) + 8 * (P.n_slices + 1) + nc * 8 * (P.n_cols + P.n_rows))
        cols = P.size * (2 * P.frac16 + 4 * (1.0 - P.frac16))
        bases = 8 * (P.size // 128) if P.frac16 > 0 else 0
        vals = (1 if A.vcode is not None else 8) * P.size
        return int(vals + cols + bases + 8 * (P.n_slices + 1) + nc * 8 * (P.n_cols + P.n_rows))

    def pressure_iteration_line(ms, n_its):
        """Whole Jacobi-CG iteration of the pressure solve ("report SpMV GB/s separately and whole-iteration
        GB/s"): the SpMV's stored bytes + the vector kernels' passes (r, q | r: update 1; r, p, x | x, p: update 2; the
        diagonal as 8-byte values or, with its dictionary, 1-byte codes) over the measured time per iteration."""
        us = 1e3 * ms / max(n_its, 1.0)
        nq = S._Q.n_owned
        dcode = getattr(S._solver_p, "_dcode", None) is not None
        if cg_merged:  # mat-vec epilogue reads the diagonal; one update kernel: x, r, p, q | x, r, p, and the diagonal
            vec = 8 * nq * 7 + 2 * nq * (1 if dcode else 8)
        else:
            vec = 8 * nq * 8 + 2 * nq * (1 if dcode else 8)
        spmv = stored_bytes(S._Ap)
        return {"us": us, "bytes_moved": int(spmv + vec), "spmv_bytes": int(spmv), "vector_bytes": int(vec),
                "stored_gbs": (spmv + vec) / (1e3 * us), "frac_stored": (spmv + vec) / (1e3 * us) / HBM_PEAK_GBS,
                "kernels_per_iteration": int(S._solver_p._cg_kernels_per_iteration()),
                "recurrences": "merged-reduction CG (OX_KSP_CG_MERGED)" if cg_merged else
                ("standard CG, both synchronisation points folded into the update kernels (k_cg_update1f / 2f)"
                 if cg_folded else "standard CG")}

    S_main = S
    cg_merged = cg_merged_main = bool(S._solver_p._cg_merged())  # the one-column pressure solve runs OX_KSP_CG_MERGED
    cg_folded = bool(S._solver_p._cg_folded())
    # (computed now: the variant legs below drop the dictionaries of this solver)
    piter_line = (pressure_iteration_line(phase_ms["pressure_solve"], mean_iterations(its)["pressure"])
                  if phase_ms.get("pressure_solve") else None)

    gd = mesh.gdim
    Pp, Pu = S._Ap.pattern, S._M.pattern
    b_p = spmv_bytes(Pp.nnz, Pp.n_rows, Pp.n_cols)
    b_u = 12 * Pu.nnz + 4 * (Pu.n_rows + 1) + gd * 8 * (Pu.n_cols + Pu.n_rows)  # matrix once, gd x/y vectors
    b_u1 = spmv_bytes(Pu.nnz, Pu.n_rows, Pu.n_cols)  # narrowed (1-column) solves on the velocity matrix
    ku, kp = Pu.n_rows, Pp.n_rows  # SpMV records are keyed by the matrix's row count

    def kernel_table(S=None, cg_merged=None):
        S = S_main if S is None else S
        cg_merged = cg_merged_main if cg_merged is None else cg_merged
        Pp, Pu = S._Ap.pattern, S._M.pattern
        b_p = spmv_bytes(Pp.nnz, Pp.n_rows, Pp.n_cols)
        b_u = 12 * Pu.nnz + 4 * (Pu.n_rows + 1) + gd * 8 * (Pu.n_cols + Pu.n_rows)  # matrix once, gd x/y vectors
        b_u1 = spmv_bytes(Pu.nnz, Pu.n_rows, Pu.n_cols)  # narrowed (1-column) solves on the velocity matrix
        ku, kp = Pu.n_rows, Pp.n_rows  # SpMV records are keyed by the matrix's row count
        kernels = {}
        sA3, sA1 = stored_bytes(S._A, gd), stored_bytes(S._A, 1)
        sM3, sM1 = stored_bytes(S._M, gd), stored_bytes(S._M, 1)
        for name, tag, key, csr, moved in (
                ("pressure_cg_spmv", 15 if cg_merged else 11, kp, b_p, stored_bytes(S._Ap)),
                ("velocity_bcgs_spmv_v", 10 * gd + 2, ku, b_u, sA3), ("velocity_bcgs_spmv_t", 10 * gd + 3, ku, b_u, sA3),
                ("velocity_bcgs_spmv_v_narrowed", 12, ku, b_u1, sA1), ("velocity_bcgs_spmv_t_narrowed", 13, ku, b_u1, sA1),
                ("mass_cg_spmv", 10 * gd + 1, ku, b_u, sM3), ("mass_cg_spmv_narrowed", 11, ku, b_u1, sM1),
                ("mass_spmv", 10 * gd + 0, ku, b_u, sM3), ("assemble_first", 100, -1, None, None),
                ("grad_vector_p", 110, -1, None, None), ("grad_vector_dp", 111, -1, None, None),
                ("div_vector", 120, -1, None, None),
                ("rect_spmv_p_and_gradp", 140, -1, None, None), ("rect_spmv_div", 141, -1, None, None),
                # mesh-partitioned runs: exchanges on rank 0 (both include the wait for the peers)
                ("halo_exchange_1comp", 150, 1, None, None), ("halo_exchange_3comp", 150, 3, None, None),
                ("krylov_sync_point", 151, -1, None, None)):
            if key == kp and ku == kp and name != "pressure_cg_spmv":
                continue  # P1-P1: both matrices have the same size; keep the pressure entry only
            cnt, ms = prof_get(tag, key)
            if cnt:
                k = {"launches": cnt, "avg_us": 1e3 * ms / cnt, "total_ms": ms}
                if csr:
                    us = 1e3 * ms / cnt
                    # stored bytes (what the kernel's streams hold) over the HIP-event time: the counter-derived
                    # fractions are in kernels_rocprofv3 / roofline
                    k.update({"bytes_moved": moved, "stored_gbs": moved / (1e3 * us),
                              "frac_stored": moved / (1e3 * us) / HBM_PEAK_GBS,
                              "csr_bytes": csr, "csr_equivalent_gbs": csr / (1e3 * us)})
                kernels[name] = k
        return kernels

    kernels = kernel_table()
    cg = kernels.get("pressure_cg_spmv")

    def spmv_kernel_id(A, epi):
        """Demangled-name prefix (spaces removed) and grid size of the one-column SpMV launch of ``A`` (how the
        rocprofv3 tables of the child passes are searched)."""
        P = A.pattern
        var = 7 if A.vcode is not None else (3 if P.frac16 > 0 else 1)
        grid = 256 * ((((P.n_slices + 3) // 4) + 7) // 8 * 8)
        return (f"voidk_spmv_ps<1,{epi}>" if A.ps_code is not None else f"voidk_spmv<1,{epi},{var}>"), grid

    def roofline_of(stored, csr, ev_us, ev_launches, counters):
        """The fractions of one kernel: counter-derived where the child passes found it (HBM traffic / rocprofv3 kernel
        time), else the stored bytes over the HIP-event time -- `basis` says which."""
        r = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "stored_bytes_per_launch": stored,
             "algorithmic_bytes_per_launch": csr,
             "hip_event": {"avg_launch_us": ev_us, "launches": ev_launches, "achieved_stored": stored / (1e3 * ev_us),
                           "frac_stored": stored / (1e3 * ev_us) / HBM_PEAK_GBS}}
        if counters is not None:
            us = counters["avg_us"]
            r.update({"achieved": counters["traffic"] / (1e3 * us), "traffic": counters["traffic"], "avg_launch_us": us,
                      "traffic_detail": _pick(counters, ("kernel", "grid", "dispatches", "pmc_dispatches", "read_bytes", "write_bytes")),
                      "traffic_over_stored": counters["traffic"] / stored,
                      "basis": "HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, separate child passes of "
                               "this command) / the kernel's rocprofv3 --kernel-trace duration (third child pass)"})
        else:
            us = ev_us
            r.update({"achieved": stored / (1e3 * us), "traffic": None, "avg_launch_us": us,
                      "basis": "NO COUNTERS in this run: bytes the kernel's storage holds (matrix streams + x + y) / HIP-event time"})
        r["frac"] = r["achieved"] / HBM_PEAK_GBS
        r["achieved_algorithmic"] = csr / (1e3 * us)
        r["frac_algorithmic"] = csr / (1e3 * us) / HBM_PEAK_GBS
        return r

    roofline = None
    if cg:
        epi_name = "OX_EPI_CG_M2" if cg_merged else "OX_EPI_DOT"
        kid = spmv_kernel_id(S._Ap, 5 if cg_merged else 1)
        on_windows = (S._Ap.ps_code is None and getattr(Pp, "wcode", None) is not None and S._Ap._struct.n_wblocks > 0)
        roofline = {"kernel": (f"k_spmv_ps<1,{epi_name}> (pressure-Poisson CG SpMV, SELL-64 pair-slot stream, f64)"
                               if S._Ap.ps_code is not None else
                               f"k_spmv_win<1,{epi_name},*> (pressure-Poisson CG SpMV, SELL-64 LDS-window stream, f64)" if on_windows else
                               f"k_spmv<1,{epi_name},*> (pressure-Poisson CG SpMV, SELL-64, f64)")}
        roofline.update(roofline_of(cg["bytes_moved"], b_p, cg["avg_us"], cg["launches"],
                                    pick_kernel(passes, *kid, epi=5 if cg_merged else 1)))
        if passes is not None and "error" in passes:
            roofline["counters_error"] = passes["error"]
        roofline.update({
            "frac_algorithmic_note": "SURVEY.md 8d's CSR bytes (12 B per nonzero) over the same time; above 1 where the matrix is "
                                     "stored smaller than CSR (16-bit column codes, 1-byte value codes: lossless, f64 arithmetic) "
                                     "and fits the 256 MB Infinity Cache between CG iterations -- not a fraction of a roofline; "
                                     "past_cache is the HBM-resident size",
            "cols16_fraction": Pp.frac16, "value_dictionary_entries": int(S._Ap._struct.n_dict),
            "pair_slots_per_row": (S._Ap.ps_code.numel() / 64 / max(Pp.n_slices, 1)) if S._Ap.ps_code is not None else None,
            "address_unit_note": "on compressed storage this kernel is bound by dependent vector-memory rounds per slice, not "
                                 "by bytes: tools/ubench/dispatch_rate.hip, DESIGN.md sections 3 and 9"})

    # ---- variant legs on the same solver (reported beside the headline, never instead of it) -------
    variants = {}
    steady = None
    if world == 1 and not args.no_extras and not args.pmc_child:
        sv = min(args.steps, 5)

        def leg(label, note, warm=1):
            el, it = timed_run(sv, warm, False)
            kt = kernel_table()
            variants[label] = {"value": sv / el, "unit": "steps/s", "ms_per_step": 1e3 * el / sv, "steps": sv,
                               "krylov_iterations_per_step": mean_iterations(it),
                               "krylov_iterations_series": iteration_series(it),
                               "pressure_cg_spmv": kt.get("pressure_cg_spmv"), "note": note}
            return variants[label]

        # the headline's settings past the start-up transient: steps 41.. of the same run (the driver's command line
        # times steps 6-25, where the previous field is still a better and better initial guess step after step)
        done = args.warmup + args.steps
        first = max(done, 40)
        steady = dict(leg("steady_state", "the headline's own solver and settings, timed after step %d" % first,
                          warm=first - done), first_timed_step=first + 1)
        del variants["steady_state"]

        guess_now = not args.zero_guess
        for sol in (S._solver_u, S._solver_p, S._solver_c):
            sol.updateOptions({"ksp_initial_guess_nonzero": not guess_now})
        leg("initial_guess_nonzero=%s" % (not guess_now),
            "PETSc's default is a zero initial guess; the headline uses -ksp_initial_guess_nonzero" if guess_now else
            "the previous field as the initial guess (-ksp_initial_guess_nonzero)")
        for sol in (S._solver_u, S._solver_p, S._solver_c):
            sol.updateOptions({"ksp_initial_guess_nonzero": guess_now})
        if not args.no_dictionary and S._Ap.vcode is not None:
            for A in (S._M, S._K, S._Ap):
                A._drop_codes()
            if not args.matrix_free:
                for A in (S._p_vdxi_Mat, S._grad_p_Mat, S._divu_Mat):
                    A.unfreeze()
            nd = leg("value_dictionary=False",
                     "f64 value streams everywhere: what a mesh without bit-identical cells (any unstructured mesh) gets; "
                     "same arithmetic, same results")
            ev = nd.get("pressure_cg_spmv")
            if roofline is not None and ev and ev.get("bytes_moved"):
                # the SAME mat-vec on the SAME matrix with plain f64 values (16-bit column codes only): HBM-resident
                # (0.47 GB > the 256 MB Infinity Cache), i.e. ON the HBM roofline -- counters from a second set of child
                # passes (--no-dictionary).  The shipped value dictionary makes it ~2 x faster and takes it off that roofline.
                kid_f64 = spmv_kernel_id(S._Ap, 5 if cg_merged else 1)  # (the codes are dropped: k_spmv<1, epi, 3>)
                f64 = roofline_of(ev["bytes_moved"], b_p, ev["avg_us"], ev["launches"], pick_kernel(passes_f64, *kid_f64))
                f64["what"] = ("the same launch with the value dictionary OFF (f64 value stream, 16-bit column codes): what a "
                               "mesh without bit-identical cells gets, and the form that is bound by HBM bytes")
                if passes_f64 is not None and "error" in passes_f64:
                    f64["counters_error"] = passes_f64["error"]
                roofline["f64_values"] = f64

    # ---- the metric's kernel past the Infinity Cache: 256^3 pressure matrix -----------------------
    # (also in the --pmc-child runs, so that the counter passes hold this launch too)
    past_id = None
    if world == 1 and not args.no_extras and args.mesh == "box" and (roofline is not None or args.pmc_child):
        try:
            t0 = time.perf_counter()
            reps, warm = (20, 5) if args.pmc_child else (200, 20)
            m2 = M.create_box(None, [p0, p1], [256, 256, 256])
            bc2 = [[ox.DirichletBC(0.0, ox.LocatorMethod.GEOMETRICAL, on_boundary)] for _ in range(3)]
            S2 = ox.FractionalStep_AB_CN(m2, ("Lagrange", 1), ("Lagrange", 1), bcs_u=bc2, bcs_p=[],
                                         options={"low_memory_version": True, "value_dictionary": not args.no_dictionary})
            A2 = S2._Ap
            P2 = A2.pattern
            x2 = (torch.sin(torch.arange(P2.n_cols, device="cuda", dtype=torch.float64) * 1e-3) + 1).reshape(-1, 1)
            y2 = torch.zeros_like(x2)
            for _ in range(warm):
                A2.mult(x2, y2, 1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                A2.mult(x2, y2, 1)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            past_id = spmv_kernel_id(A2, 0)
            if roofline is not None:
                pc = {"workload": "pressure Laplacian of the 256^3 box mesh (P1, 16 974 593 rows, SURVEY.md 8 C5), "
                                  "x_j = sin(j*1e-3)+1, 200 launches after 20 warm-up (SURVEY.md 8d), plain y = A x",
                      "nnz": P2.nnz}
                pc.update(roofline_of(stored_bytes(A2), spmv_bytes(P2.nnz, P2.n_rows, P2.n_cols), us, reps,
                                      pick_kernel(passes, *past_id)))
                pc.update({"value_dictionary_entries": int(A2._struct.n_dict), "seconds": time.perf_counter() - t0})
                roofline["past_cache"] = pc
            del S2, A2, x2, y2, m2
            torch.cuda.empty_cache()
        except Exception as e:
            if roofline is not None:
                roofline["past_cache"] = {"error": repr(e)}

    if args.pmc_child:  # what the parent needs to find this run's pressure SpMV in the rocprofv3 tables
        prefix, grid = spmv_kernel_id(S._Ap, 5 if cg_merged else 1)
        print(json.dumps({"pmc_child": True, "kernel_prefix": prefix, "grid_size": grid,
                          "past_cache": list(past_id) if past_id else None}), flush=True)
        return

    transport_check = None
    if world > 1:  # the transport once more after the timed steps: every ghost dof must still get its owner's value
        S._Vi[0][0].check_halo()
        S._Q.check_halo()
        transport_check = "halo self-test passed after the timed steps"
    nnz_glob = [Pu.nnz, Pp.nnz]
    per_rank = None
    phase_ms_max = None
    if world > 1:
        import torch.distributed as dist

        # per phase the SLOWEST rank's device time: what the step waits for, and what tools/predict_scaling.py predicts
        names = sorted(phase_ms)
        tp = torch.tensor([phase_ms[k] for k in names], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tp, op=dist.ReduceOp.MAX)
        phase_ms_max = {k: float(v) for k, v in zip(names, tp.tolist())}

        tn = torch.tensor(nnz_glob, dtype=torch.int64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tn)
        nnz_glob = [int(v) for v in tn.tolist()]
        # every rank's share: owned rows and ghosts of both spaces, its peers, what ITS communicator reports
        Vu_, Q_ = S._Vi[0][0], S._Q
        mine = {"rank": rank, "device": torch.cuda.current_device(), "cells": int(Vu_.local_cells.shape[0]),
                "velocity_rows": int(Vu_.n_owned), "velocity_ghosts": int(Vu_.n_local - Vu_.n_owned),
                "pressure_rows": int(Q_.n_owned), "pressure_ghosts": int(Q_.n_local - Q_.n_owned),
                "peers": [int(p_) for p_ in (Vu_.halo["peers"] if Vu_.halo is not None else [])], "comm": comm.info()}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    out = None
    if rank == 0:
        mean_its = mean_iterations(its)
        short = {"tg": "Taylor-Green", "beltrami": "Beltrami (Ethier-Steinman)", "cavity": "lid-driven cavity"}[args.workload]
        out = {
            "metric": f"time-steps/sec, 3D {short} {N}^3 P{args.udeg}-P{args.pdeg} (with pressure-CG SpMV GB/s in roofline)",
            "value": args.steps / elapsed, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{W['desc']}, "
                                   + (f"{N}^3x6 tets" if args.mesh == "box" else
                                      f"Delaunay mesh of a jittered {N + 1}^3 lattice"
                                      + (f", uniformly refined {args.refine}x" if args.refine else "") + f" ({mesh.num_cells} tets)")
                                   + f" P{args.udeg}-P{args.pdeg}, nu={nu}, dt={dt:g}, "
                                   f"bcgs+jacobi / cg+jacobi rtol={args.rtol:g} atol=1e-14 "
                                   f"initial_guess_nonzero={not args.zero_guess}, max_iter=1, "
                                   f"low_memory_version={args.matrix_free}, value_dictionary={not args.no_dictionary}",
                       "cells": mesh.num_cells, "n_u_per_component": S._Vi[0][0].num_dofs_global,
                       "n_p": S._Q.num_dofs_global,
                       "nnz_velocity": nnz_glob[0], "nnz_pressure": nnz_glob[1], "parallelism": f"mesh-partition x{world}",
                       "transport": (None if world == 1 else
                                     "/".join(sorted(set(comm.active.values()))) or "none")
                       + ("" if args.backend == "nccl" or world == 1 else " (rehearsal: torch.distributed over gloo)")
                       if world > 1 else None,
                       "transport_check": transport_check, "transport_exchange_us": None,
                       # N > 1: the rank count the library's RCCL communicator reports (ncclCommCount; None in a gloo
                       # rehearsal, which has none), and every rank's rows / ghosts / peers
                       "rccl_nranks": (None if world == 1 or not per_rank[0]["comm"]["rccl"] else
                                       sorted({r_["comm"]["nranks"] for r_ in per_rank})),
                       "ranks": per_rank,
                       "launched_by": "bench.py --gpus N (self-launched torch.distributed.run child)"
                       if os.environ.get("OX_BENCH_SELF_LAUNCHED") else ("torch.distributed.run" if world > 1 else "python"),
                       # all-reduces per Krylov iteration on a partitioned operator: single-reduction CG
                       # (Chronopoulos-Gear) and merged-reduction BiCGStab are the defaults there (oasisx_amd/ksp.py)
                       "krylov_sync_points_per_iteration": (
                           None if world == 1 else
                           {"pressure cg": 1 if (cg_merged or S._solver_p._method()[0] == _lib.KSP_CG_SINGLE) else 2,
                            "velocity bcgs": 2 if S._solver_u._method()[0] == _lib.KSP_BCGS_MERGED else 3,
                            "update cg": 1 if S._solver_c._method()[0] == _lib.KSP_CG_SINGLE else 2})},
            "cg_spmv_gbs": roofline["achieved"] if roofline else None,  # the roofline's figure (see roofline.basis)
            "roofline": roofline,
            "krylov_iterations_per_step": mean_its,
            "krylov_iterations_series": iteration_series(its),  # per timed step (the warm start's transient shows here)
            "steady_state": steady,  # 5 steps after step 40 with the same solver and settings
            # the last step's solves, column by column: |D^-1 b|, final |D^-1 r| (what the convergence test compares with
            # rtol |D^-1 b| and atol = 1e-14) and iterations -- on the z-extruded field the w column's right-hand side is
            # what the other solves' tolerance leaves of the z-invariance (|D^-1 b_w| ~ 3e-8 |D^-1 b_u|); it is solved to
            # ITS OWN rtol like any column, as PETSc's per-component solves would
            "krylov_last_solve": last_solves(),
            "phase_ms_per_step": phase_ms,  # device time between events around each phase method (rank 0)
            "phase_ms_per_step_max_over_ranks": phase_ms_max,  # N > 1: the slowest rank per phase
            # the PREDICTED line for this rank count (tools/predict_scaling.py: one-GPU measurements of every rank's local
            # work + a modelled link; committed BEFORE any multi-GPU run) -- laid beside the measurement above
            "prediction": predicted_line(args, world),
            # whole Jacobi-CG iteration of the pressure solve (SpMV + vector kernels + scalar kernels)
            "pressure_cg_iteration": piter_line,
            "accuracy": {"max_nodal_error_u_vs_analytic": err_u, "max_abs_u": umax, "t_end": clock["t"]},
            # HIP-event table (stored bytes / event time) and the counter-derived table of the child passes
            "kernels": kernels,
            "kernels_rocprofv3": rocprof_kernel_table(passes),
            "variants": variants,
            "setup_s": t_setup,
            "hbm_gib": torch.cuda.max_memory_allocated() / 2 ** 30,
            "cpu_baseline": None,  # timed on rank 0 at N = 1 only (below)
            "full_record": os.path.relpath(full_path, ROOT),
        }
        if variants:  # PETSc's own defaults in one place: zero initial guess, no value dictionaries
            zg = variants.get("initial_guess_nonzero=False")
            out["headline_petsc_default"] = {
                "what": "the same workload as PETSc's defaults leave it: zero initial guess for every Krylov solve "
                        "(the headline uses -ksp_initial_guess_nonzero), and -- what any mesh without bit-identical "
                        "cells gets -- f64 value streams instead of the 1-byte value dictionaries",
                "zero_initial_guess_steps_per_s": zg["value"] if zg else None,
                "no_value_dictionary_steps_per_s": (variants.get("value_dictionary=False") or {}).get("value"),
                "headline_steps_per_s": out["value"]}
        ksp_cpu = {"rtol": args.rtol, "atol": 1e-14, "max_it": 10000, "guess": not args.zero_guess}

        def host_fields(X, t):
            return np.stack([np.asarray(f(X, t), dtype=np.float64) for f in fns])

        mesh_def = ((p0, p1, [N, N, N]) if args.mesh == "box" else
                    {"coords": mesh.coords.cpu().numpy(), "cells": mesh.cells.cpu().numpy(), "lo": p0, "hi": p1})
        if not args.no_cpu and world == 1:
            try:
                from oracle.cpu_baseline import run_cpu_baseline

                out["cpu_baseline"] = run_cpu_baseline(S, clock, dt, nu, ksp_cpu, host_fields, gpu_step=step, mesh_def=mesh_def,
                                                       threads_1=False, reuse_setup=True)
                out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            except Exception as e:  # the baseline is a reported figure, never the product path
                out["cpu_baseline"] = {"error": repr(e)}
        # ---- THE line: the full record to the side file first, then the one stdout line, then nothing on stdout ----
        write_full(full_path, out)
        sys.stderr.flush()
        print(compact_line(out), flush=True)

    # ================= everything below runs AFTER the line is out; results go to stderr and the side file =================
    if world > 1 and args.probe_transports:
        # both device transports' exchange times (collective; brings up the xGMI-window transport on a temporary plan)
        try:
            transport_us = {"velocity_space": comm.time_transports(S._Vi[0][0]), "pressure_space": comm.time_transports(S._Q)}
        except Exception as e:
            transport_us = {"error": repr(e)}
        if rank == 0:
            out["config"]["transport_exchange_us"] = transport_us
            print("[bench] transport_exchange_us " + json.dumps(transport_us), file=sys.stderr, flush=True)
            write_full(full_path, out)
    if rank == 0 and world == 1 and args.extras:
        def note(msg):
            print("[bench] extras: " + msg, file=sys.stderr, flush=True)

        if not args.no_cpu and not args.no_cpu_one_core and isinstance(out["cpu_baseline"], dict) and "value" in out["cpu_baseline"]:
            try:  # the same host step on ONE core (about a minute at 128^3)
                from oracle.cpu_baseline import run_cpu_baseline

                one = run_cpu_baseline(S, clock, dt, nu, ksp_cpu, host_fields, gpu_step=step, mesh_def=mesh_def,
                                       threads_1=True, scipy_check=False, reuse_setup=True)
                out["cpu_baseline"]["one_core"] = one.get("one_core")
            except Exception as e:
                out["cpu_baseline"]["one_core"] = {"error": repr(e)}
            write_full(full_path, out)
            note("one-core cpu_baseline done")
        # the other BASELINE workloads on the same mesh size beside the headline
        # (Beltrami: w != 0, no round-off right-hand side; cavity: BASELINE.json configs[3]'s 1-GPU line)
        if extras_legs:
            import gc

            del S  # (the phase wrappers hold it in a cycle: collect before the next 49 GiB solver is built)
            S_main = None
            gc.collect()
            torch.cuda.empty_cache()
            for wname in ("beltrami", "cavity"):
                try:
                    out["variants"]["workload=" + wname] = workload_leg(wname)
                except Exception as e:
                    out["variants"]["workload=" + wname] = {"error": repr(e)}
                write_full(full_path, out)
                note("workload=" + wname + " done")
            # what an UNSTRUCTURED mesh of the metric's size gets (north_star: "the unstructured mesh"): same fields,
            # same Krylov settings, a refined Delaunay mesh of 18.9 M P2 dofs per component
            if args.udeg == 2 and N >= 96:
                try:
                    # (Beltrami: all three components live -- on this mesh the z-extruded field's w column, a round-off
                    # right-hand side, would add ~180 narrowed BiCGStab iterations per step to both sides of the check)
                    out["variants"]["mesh=delaunay"] = workload_leg("beltrami", delaunay=DELAUNAY_LEG)
                except Exception as e:
                    out["variants"]["mesh=delaunay"] = {"error": repr(e)}
                write_full(full_path, out)
                note("mesh=delaunay done")
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
