"""Predicted 1 / 2 / 4 / 8-GPU curves of OTHER workloads on the 128^3 mesh from the per-rank costs that
tools/predict_scaling.py measured (profiles/r05_predicted_scaling.json): the matrices M, K, Ap and the pattern of A are the
Taylor-Green run's, so a rank's cost per Krylov iteration, per assembly and per exchange is the same; what changes is the
iteration profile, taken from the measured one-GPU legs of bench.py (profiles/r05_bench_default.json).  A PREDICTION, like
its source; runs on the CPU.
    python tools/predict_workloads.py > profiles/r05_predicted_scaling_workloads.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from scaling_model import ASSUMED, predict  # noqa: E402

src = json.loads(open(os.path.join(ROOT, "profiles", "r05_predicted_scaling.json")).readline())
bench = json.loads(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")).read().strip().splitlines()[-1])


def profile_of(per_column_tent, per_column_upd, pressure):
    """lock-step iterations with three columns, then the narrowed continuation of the last live one (csrc/ox_ksp.hip)"""
    t, u = sorted(per_column_tent), sorted(per_column_upd)
    return {"tent3": float(t[1]), "tent1": float(t[2] - t[1]), "upd3": float(u[1]), "upd1": float(u[2] - u[1]),
            "pressure": float(pressure)}


legs = {}
for key, name in (("workload=cavity", "lid-driven cavity Re = 1000 from rest (BASELINE configs[3])"),
                  ("workload=beltrami", "Beltrami (Ethier-Steinman)")):
    leg = bench["variants"][key]
    cc = leg["cpu_cross_check"]["gpu_krylov_iterations"]  # per-column counts of one step; the means scale them
    mean = leg["krylov_iterations_per_step"]
    t, u = cc["tentative"][:3], cc["update"][:3]
    st, su = mean["tentative"] / max(max(t), 1), mean["update"] / max(max(u), 1)
    legs[name] = {"profile": profile_of([v * st for v in t], [v * su for v in u], mean["pressure"]),
                  "measured_one_gpu_steps_per_s": leg["value"]}

out = {"label": "PREDICTION from one-GPU measurements -- NOT a measurement of a multi-GPU run", "assumed": ASSUMED,
       "source": "per-rank costs: profiles/r05_predicted_scaling.json (128^3); iteration profiles: profiles/r05_bench_default.json",
       "workloads": {}}
for name, leg in legs.items():
    curve = {}
    for P, e in src["P"].items():
        ranks = e["ranks"]
        ms = max(sum(predict(m, leg["profile"], int(P)).values()) for m in ranks.values())
        curve[P] = {"ms_per_step": ms, "steps_per_s": 1e3 / ms}
    for P in curve:
        curve[P]["speedup_vs_model_P1"] = curve["1"]["ms_per_step"] / curve[P]["ms_per_step"]
    leg["model_error_vs_measured_P1"] = (1e3 / leg["measured_one_gpu_steps_per_s"]) and curve["1"]["ms_per_step"] / (1e3 / leg["measured_one_gpu_steps_per_s"]) - 1.0
    leg["P"] = curve
    out["workloads"][name] = leg
print(json.dumps(out))
