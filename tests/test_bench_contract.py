"""bench.py's output contract: ONE compact JSON line (<= 8000 bytes) with the driver's keys plus `roofline` and
`cpu_baseline`, the full record in a side file.  CPU: the newest committed full record through ``compact_line``;
GPU: a live small run of the default flags."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}
LIMIT = 8000


def _check(d, live_cpu=True, counters=None):
    assert KEYS <= set(d), KEYS - set(d)
    assert d["unit"] == "steps/s" and d["higher_is_better"] is True and d["dtype"] == "f64"
    assert d["data"] == "synthetic" and d["vs_baseline"] is None and d["scaling"] in ("strong", "weak")
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5 and "traffic" in r and r["basis"]
    assert 0.0 < r["frac"] < 1.0  # a fraction of the HBM peak: bytes really moved, not CSR-priced
    if counters is None:
        counters = r["traffic"] is not None
    if counters:  # counter-derived: HBM traffic of the rocprofv3 --pmc child passes / the kernel-trace duration
        assert r["traffic"] > 0 and "rocprofv3" in r["basis"]
        assert abs(r["achieved"] - r["traffic"] / (1e3 * r["avg_launch_us"])) < 1e-4 * r["achieved"]
        assert abs(r["traffic_over_stored"] - r["traffic"] / r["stored_bytes_per_launch"]) < 1e-4
    else:
        assert r["traffic"] is None and "NO COUNTERS" in r["basis"]
        assert abs(r["achieved"] - r["stored_bytes_per_launch"] / (1e3 * r["avg_launch_us"])) < 1e-4 * r["achieved"]
    assert abs(r["frac_algorithmic"] - r["algorithmic_bytes_per_launch"] / (1e3 * r["avg_launch_us"]) / r["peak"]) < 1e-4 * r["frac_algorithmic"]
    h = r["hip_event"]  # the stored bytes over the HIP-event time measured inside the timed region
    assert abs(h["frac_stored"] - r["stored_bytes_per_launch"] / (1e3 * h["avg_launch_us"]) / r["peak"]) < 1e-4 and h["launches"] > 0
    c = d["cpu_baseline"]
    if live_cpu:
        assert {"value", "unit", "cores", "kind", "sample"} <= set(c), c
        assert c["kind"] == "port" and c["unit"] == "steps/s" and c["cores"] >= 1
        assert c["gpu_vs_cpu_rel_l2_u"] < 1e-8 and c["gpu_vs_cpu_rel_l2_p"] < 1e-6


def _newest_full_record():
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_full.json")))
    assert paths, "no committed full bench record (profiles/rNN_bench_full.json)"
    return paths[-1]


def test_the_compact_line_of_the_newest_committed_record_meets_the_contract():
    """What the driver reads: ``compact_line`` of the newest committed full record is within 8000 bytes, carries the
    contract's keys, `roofline` with counter-derived fractions and `cpu_baseline`; the committed line beside it is that line."""
    import bench

    path = _newest_full_record()
    full = json.loads(open(path).read().strip().splitlines()[-1])
    line = bench.compact_line(full)
    assert len(line.encode()) <= LIMIT, len(line.encode())
    d = json.loads(line)
    _check(d, counters=True)
    assert d["n_gpus"] == 1 and "128^3" in d["config"]["workload"] and d["steps"] == 20 and d["warmup"] == 5
    r = d["roofline"]
    assert 0.9 < r["traffic_over_stored"] < 1.2  # no wasted re-reads: the counters agree with the stored bytes
    assert r["frac_algorithmic"] > r["frac"]  # the CSR-priced figure (above 1 at 128^3: see frac_algorithmic_note)
    pc = r["past_cache"]  # the same kernel on the 256^3 matrix, its own counters from the same child passes
    assert 0.0 < pc["frac"] < 1.0 and pc["stored_bytes_per_launch"] > 256 * 2 ** 20 and pc["traffic"] > 256 * 2 ** 20
    assert "rocprofv3" in pc["basis"]
    f64 = r["f64_values"]  # the same launch on plain f64 values (second set of child passes): HBM-resident, on the roofline
    assert f64["traffic"] > 256 * 2 ** 20 and 0.9 < f64["traffic_over_stored"] < 1.1 and f64["frac"] >= 0.60 > r["frac"]
    assert f64["avg_launch_us"] > 1.5 * r["avg_launch_us"]  # ... and the dictionary form is that much faster
    hp = d["headline_petsc_default"]
    assert hp["zero_initial_guess_steps_per_s"] < hp["headline_steps_per_s"] and hp["no_value_dictionary_steps_per_s"] > 0
    assert d["cpu_baseline"]["cpu_model"] and d["cpu_baseline"]["gpu_over_cpu"] > 1.0
    # the line committed beside the record (the stdout of the same run) is this function's output
    side = path.replace("_bench_full.json", "_bench_line.json")
    committed = open(side).read().strip()
    assert len(committed.encode()) <= LIMIT and len(committed.splitlines()) == 1
    dc = json.loads(committed)
    assert KEYS <= set(dc) and dc["value"] == d["value"] and dc["roofline"]["frac"] == d["roofline"]["frac"]
    # the full record keeps what the line leaves out
    assert {"variants", "kernels", "kernels_rocprofv3", "krylov_last_solve", "prediction"} <= set(full)


@pytest.mark.parametrize("rec", ["r04_bench_default.json", "r05_bench_default.json"])
def test_earlier_rounds_records_shrink_to_the_limit(rec):
    """r05's 20 KB line was the one the driver could not parse: the compact form of it (and of r04's) fits."""
    import bench

    raw = open(os.path.join(ROOT, "profiles", rec)).read().strip().splitlines()[-1]
    assert len(raw) > 2 * LIMIT
    line = bench.compact_line(json.loads(raw))
    d = json.loads(line)
    assert len(line.encode()) <= LIMIT and KEYS <= set(d) and d["cpu_baseline"]["value"] > 0 and d["roofline"]["frac"] > 0


def test_the_compact_line_drops_optional_blocks_before_it_breaks_the_limit():
    """Eight ranks with long peer lists and a long Krylov series: the contract's keys, `roofline` and `cpu_baseline`
    stay, the optional tables go."""
    import bench

    full = json.loads(open(_newest_full_record()).read().strip().splitlines()[-1])
    full["n_gpus"] = 8
    full["config"]["ranks"] = [{"rank": r, "device": r, "cells": 1572864, "velocity_rows": 2121824, "velocity_ghosts": 100000,
                                "pressure_rows": 268336, "pressure_ghosts": 20000, "peers": [q for q in range(8) if q != r],
                                "comm": {"rccl": True, "nranks": 8, "blob": "x" * 2000}} for r in range(8)]
    full["krylov_iterations_series"] = {k: list(range(400)) for k in ("tentative", "pressure", "update")}
    line = bench.compact_line(full)
    d = json.loads(line)
    assert len(line.encode()) <= LIMIT and KEYS <= set(d) and "headline_petsc_default" in d
    assert "krylov_iterations_series" not in d and len(d["config"]["ranks"]) == 8 and "comm" not in d["config"]["ranks"][0]


@pytest.mark.gpu
def test_bench_runs_and_prints_one_json_line(tmp_path):
    """The driver's own command but for the mesh size: default flags (counter child passes, variant legs, past-cache
    SpMV, cpu_baseline), ONE stdout line within the limit, the full record in the side file."""
    side = str(tmp_path / "full.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "-N", "16", "--steps", "2", "--warmup", "1",
                          "--full-out", side], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    assert len(lines[0].encode()) <= LIMIT
    d = json.loads(lines[0])
    _check(d)
    assert d["steps"] == 2 and d["warmup"] == 1 and d["n_gpus"] == 1
    assert d["roofline"]["traffic"] is not None, d["roofline"].get("counters_error")  # the child passes ran and found the kernel
    assert d["roofline"]["past_cache"]["traffic"] is not None and d["headline_petsc_default"]["zero_initial_guess_steps_per_s"] > 0
    assert d["roofline"]["f64_values"]["traffic"] is not None, d["roofline"]["f64_values"].get("counters_error")
    full = json.loads(open(side).read())
    assert full["value"] == d["value"] and len(full["variants"]) >= 2 and full["kernels"] and full["kernels_rocprofv3"]


def test_bench_refuses_a_rank_count_other_than_the_one_asked_for():
    """`--gpus N` is the rank count: a launcher that started another number of ranks must not get a figure
    (decided before anything touches the GPU, so this runs anywhere)."""
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert out.returncode != 0 and "--gpus 2" in out.stderr and "WORLD_SIZE=3" in out.stderr, (out.returncode, out.stderr[-500:])
    assert not out.stdout.strip()
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert out.returncode != 0 and not out.stdout.strip()


@pytest.mark.gpu
def test_bench_gpus_2_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent (which never touches the GPU) runs the driver's
    N > 1 command line as a child process and relays its one JSON line.  Rehearsed over gloo on the one GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OX_P2P_TIMEOUT_S"] = "60"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "-N", "16", "--backend", "gloo"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    _check(d, live_cpu=False)
    assert d["n_gpus"] == 2 and d["config"]["launched_by"].startswith("bench.py --gpus N")
    ranks = d["config"]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and d["config"]["rccl_nranks"] is None  # gloo rehearsal: no RCCL communicator
    assert sum(r["velocity_rows"] for r in ranks) == 33 ** 3 and sum(r["pressure_rows"] for r in ranks) == 17 ** 3
    assert all(r["velocity_ghosts"] > 0 and r["peers"] == [1 - r["rank"]] for r in ranks)
    assert len(d["krylov_iterations_series"]["pressure"]) == 2


@pytest.mark.gpu
def test_bench_under_the_driver_launch_line_with_two_ranks(tmp_path):
    """The driver's N > 1 command (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`), rehearsed with 2 ranks on the
    one GPU of the box: torch.distributed over gloo (RCCL refuses two ranks on one device), the
    data path over the xGMI windows.  One JSON line from rank 0, whole-job figures."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, OX_P2P_TIMEOUT_S="60")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "-N", "16", "--backend", "gloo",
           "--probe-transports", "--full-out", str(tmp_path / "full.json")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    _check(d, live_cpu=False)
    assert len(lines[0].encode()) <= LIMIT
    # the transport probe ran AFTER the line: nothing of it on stdout, its times on stderr and in the side file
    assert "transport_exchange_us" in out.stderr and "transport_exchange_us" not in out.stdout
    full = json.loads(open(tmp_path / "full.json").read())
    assert full["value"] == d["value"] and full["config"]["transport_exchange_us"]["velocity_space"]
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong"
    assert d["config"]["parallelism"] == "mesh-partition x2" and d["config"]["transport"].startswith("p2p")
    assert d["cpu_baseline"] is None or "value" in d["cpu_baseline"]
    assert d["config"]["n_u_per_component"] == 33 ** 3 and d["config"]["n_p"] == 17 ** 3  # global sizes


@pytest.mark.gpu
def test_the_launcher_line_with_one_rank_is_the_plain_line():
    """The driver's SCALE series starts at N = 1 under ``torch.distributed.run --nproc-per-node 1 bench.py --gpus 1``; its
    BENCH line is plain ``python bench.py``.  The two must describe the same run: same config (but for ``launched_by``),
    same Krylov iteration series, same kernels per iteration -- a one-rank job takes no partitioned code path."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    tail = ["-N", "16", "--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu", "--no-pmc"]
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *tail], capture_output=True, text=True,
                           timeout=600, cwd=ROOT)
    assert plain.returncode == 0, plain.stderr[-2000:]
    # (the launched run WITH the counter child passes, as the driver's SCALE series runs it at N = 1: the rocprofv3
    # children inherit the launcher's WORLD_SIZE = 1 / RANK / MASTER_* environment)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", *[t for t in tail if t != "--no-pmc"]]
    launched = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert launched.returncode == 0, launched.stderr[-3000:]
    a, b = (json.loads([ln for ln in o.stdout.strip().splitlines() if ln.strip().startswith("{")][-1]) for o in (plain, launched))
    for d in (a, b):
        assert len(json.dumps(d)) <= LIMIT
        assert d["n_gpus"] == 1 and d["config"]["parallelism"] == "mesh-partition x1" and d["config"]["transport"] is None
        assert d["config"]["ranks"] is None and d.get("phase_ms_per_step_max_over_ranks") is None
    ca, cb = dict(a["config"]), dict(b["config"])
    assert ca.pop("launched_by") == "python" and cb.pop("launched_by") == "python"  # (one rank: no partitioned path either way)
    assert ca == cb
    assert a["krylov_iterations_series"] == b["krylov_iterations_series"]
    assert a["pressure_cg_iteration"]["kernels_per_iteration"] == b["pressure_cg_iteration"]["kernels_per_iteration"]
    assert a["roofline"]["kernel"] == b["roofline"]["kernel"] and a["roofline"]["stored_bytes_per_launch"] == b["roofline"]["stored_bytes_per_launch"]
    assert a["roofline"]["traffic"] is None and "NO COUNTERS" in a["roofline"]["basis"]
    assert b["roofline"]["traffic"] is not None and "rocprofv3" in b["roofline"]["basis"], b["roofline"].get("counters_error")
