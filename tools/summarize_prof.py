"""Condense a rocprofv3 run (tools/prof_rNN.sh output under gpurun_out/) into the tracked
summaries under profiles/:
  <tag>_kernel_stats.csv  rocprofv3's own --stats table of the bench command (>= 0.05 % rows)
  <tag>_ox_kernels.csv    our kernels, split by (kernel, grid size) from the kernel trace -- the same
                          kernel runs on the pressure and on the velocity matrix
  <tag>_pmc_hbm.csv       per (kernel, grid size) averages of the FETCH_SIZE / WRITE_SIZE passes
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

trace = glob.glob(f"{src}/trace/*/*_kernel_trace.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    n = r["Kernel_Name"]
    if n.startswith("void k_"):
        acc[(n.split("(")[0], int(r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(f"profiles/{tag}_ox_kernels.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid_size", "calls", "avg_us", "min_us", "max_us", "total_ms"])
    for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, g, len(v), f"{sum(v) / len(v) / 1e3:.2f}", f"{min(v) / 1e3:.2f}", f"{max(v) / 1e3:.2f}",
                    f"{sum(v) / 1e6:.2f}"])

pm = {}
for name, pat in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    files = glob.glob(f"{src}/{pat}/*/*_counter_collection.csv")
    if not files:
        continue
    a2 = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == name and r["Kernel_Name"].startswith("void k_"):
            a2[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    for k, v in a2.items():
        pm.setdefault(k, {})[name] = (len(v), sum(v) / len(v))
with open(f"profiles/{tag}_pmc_hbm.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid_size", "dispatches", "FETCH_SIZE_KiB_avg", "WRITE_SIZE_KiB_avg",
                "hbm_read_MB_corrected(2x)", "hbm_write_MB", "hbm_total_MB_per_launch"])
    for (k, g), d in sorted(pm.items()):
        fs = d.get("FETCH_SIZE", (0, 0.0))
        ws = d.get("WRITE_SIZE", (0, 0.0))
        rd, wr = 2 * fs[1] * 1024 / 1e6, ws[1] * 1024 / 1e6
        w.writerow([k, g, fs[0] or ws[0], f"{fs[1]:.1f}", f"{ws[1]:.1f}", f"{rd:.1f}", f"{wr:.1f}", f"{rd + wr:.1f}"])
print("wrote profiles/%s_*" % tag)
