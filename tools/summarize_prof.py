"""Condense a rocprofv3 run (tools/prof_rNN.sh output under gpurun_out/) into the tracked
summaries under profiles/: kernel stats of the bench command and per-kernel PMC averages.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B, so a
streamed read is 2 x FETCH_SIZE (MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import sys

src, tag = sys.argv[1], sys.argv[2]
stats = glob.glob(f"{src}/trace/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(stats)))
with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        if float(r["Percentage"]) >= 0.05:
            w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
pm = {}
for name, pat in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    files = glob.glob(f"{src}/{pat}/*/*_counter_collection.csv")
    if not files:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == name and r["Kernel_Name"].startswith("void k_"):
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        pm.setdefault(k, {})[name] = (len(v), sum(v) / len(v))
with open(f"profiles/{tag}_pmc_hbm.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "dispatches", "FETCH_SIZE_KiB_avg", "WRITE_SIZE_KiB_avg",
                "hbm_read_MB_corrected(2x)", "hbm_write_MB", "hbm_total_MB_per_launch"])
    for k, d in sorted(pm.items()):
        fs = d.get("FETCH_SIZE", (0, 0.0))
        ws = d.get("WRITE_SIZE", (0, 0.0))
        rd, wr = 2 * fs[1] * 1024 / 1e6, ws[1] * 1024 / 1e6
        w.writerow([k, fs[0] or ws[0], f"{fs[1]:.1f}", f"{ws[1]:.1f}", f"{rd:.1f}", f"{wr:.1f}", f"{rd + wr:.1f}"])
print("wrote profiles/%s_*" % tag)
