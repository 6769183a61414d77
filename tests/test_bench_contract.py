"""bench.py's output contract: ONE JSON line with the driver's keys plus `roofline` and
`cpu_baseline`.  CPU: the committed line of the last profiled run; GPU: a live small run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def _check(d, live_cpu=True):
    assert KEYS <= set(d), KEYS - set(d)
    assert d["unit"] == "steps/s" and d["higher_is_better"] is True and d["dtype"] == "f64"
    assert d["data"] == "synthetic" and d["vs_baseline"] is None and d["scaling"] in ("strong", "weak")
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r
    assert 0.0 < r["frac"] < 1.0  # a fraction of the HBM peak: bytes really moved, not CSR-priced
    assert abs(r["achieved"] - r["bytes_moved_per_launch"] / (1e3 * r["avg_launch_us"])) < 1e-6 * r["achieved"]
    assert r["csr_equivalent_gbs"] >= r["achieved"]
    c = d["cpu_baseline"]
    if live_cpu:
        assert {"value", "unit", "cores", "kind", "sample"} <= set(c), c
        assert c["kind"] == "port" and c["unit"] == "steps/s" and c["cores"] >= 1
        assert c["gpu_vs_cpu_rel_l2_u"] < 1e-8 and c["gpu_vs_cpu_rel_l2_p"] < 1e-6


def test_committed_bench_line_meets_the_contract():
    path = os.path.join(ROOT, "profiles", "r02_bench_default.json")
    d = json.loads(open(path).read().strip().splitlines()[-1])
    _check(d)
    assert d["n_gpus"] == 1 and "128^3" in d["config"]["workload"]
    r = d["roofline"]
    assert r["traffic"] is not None and r["traffic_detail"]["dispatches"] > 0  # live PMC child passes of the command
    assert 0.5 < r["traffic"] / r["bytes_moved_per_launch"] < 2.0  # counters agree with the stored bytes
    assert 0.0 < r["past_cache"]["frac"] < 1.0 and r["past_cache"]["bytes_moved_per_launch"] > 256 * 2 ** 20
    assert any(k.startswith("value_dictionary") for k in d["variants"]) and any("guess" in k for k in d["variants"])
    assert d["cpu_baseline"]["one_core"]["value"] < d["cpu_baseline"]["value"] and d["cpu_baseline"]["cpu_model"]


@pytest.mark.gpu
def test_bench_runs_and_prints_one_json_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "-N", "16", "--steps", "2", "--warmup", "1", "--no-extras"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    _check(d)
    assert d["steps"] == 2 and d["warmup"] == 1 and d["n_gpus"] == 1


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
def test_bench_under_the_driver_launch_line_with_two_ranks():
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
           "--gpus", "2", "--steps", "2", "--warmup", "1", "-N", "16", "--backend", "gloo"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    _check(d, live_cpu=False)
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
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", *tail]
    launched = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert launched.returncode == 0, launched.stderr[-3000:]
    a, b = (json.loads([ln for ln in o.stdout.strip().splitlines() if ln.strip().startswith("{")][-1]) for o in (plain, launched))
    for d in (a, b):
        assert d["n_gpus"] == 1 and d["config"]["parallelism"] == "mesh-partition x1" and d["config"]["transport"] is None
        assert d["config"]["ranks"] is None and d["phase_ms_per_step_max_over_ranks"] is None
    ca, cb = dict(a["config"]), dict(b["config"])
    assert ca.pop("launched_by") == "python" and cb.pop("launched_by") == "python"  # (one rank: no partitioned path either way)
    assert ca == cb
    assert a["krylov_iterations_series"] == b["krylov_iterations_series"]
    assert a["pressure_cg_iteration"]["kernels_per_iteration"] == b["pressure_cg_iteration"]["kernels_per_iteration"]
    assert a["roofline"]["kernel"] == b["roofline"]["kernel"] and a["roofline"]["bytes_moved_per_launch"] == b["roofline"]["bytes_moved_per_launch"]
