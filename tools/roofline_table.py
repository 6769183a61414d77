"""Counter-derived fractions of the HBM peak, per (kernel, grid), from the tracked summaries of one profiled bench
command: profiles/<tag>_ox_kernels.csv (rocprofv3 --kernel-trace durations) and profiles/<tag>_pmc_hbm.csv (FETCH_SIZE x 2
+ WRITE_SIZE of the separate --pmc passes): traffic per launch / average duration / 8 TB/s.
    python tools/roofline_table.py r06_bench128 r06_delaunay > profiles/r06_roofline_table.txt"""
import csv
import sys

PEAK = 8000.0  # GB/s
for tag in sys.argv[1:]:
    k = {(r["kernel"], r["grid_size"]): r for r in csv.DictReader(open(f"profiles/{tag}_ox_kernels.csv"))}
    p = {(r["kernel"], r["grid_size"]): r for r in csv.DictReader(open(f"profiles/{tag}_pmc_hbm.csv"))}
    print(f"{tag}: profiles/{tag}_pmc_hbm.csv / profiles/{tag}_ox_kernels.csv")
    print(f"  {'kernel':58s} {'grid':>9s} {'calls':>6s} {'avg us':>9s} {'MB/launch':>10s} {'GB/s':>7s} {'of 8 TB/s':>9s} {'total ms':>9s}")
    rows = []
    for key, r in k.items():
        if key in p and float(r["total_ms"]) > 0.3:
            us, mb = float(r["avg_us"]), float(p[key]["hbm_total_MB_per_launch"])
            rows.append((float(r["total_ms"]), key, r["calls"], us, mb))
    for tot, (name, grid), calls, us, mb in sorted(rows, reverse=True)[:24]:
        gbs = mb / us * 1e3
        print(f"  {name.replace('void ', ''):58s} {grid:>9s} {calls:>6s} {us:9.2f} {mb:10.1f} {gbs:7.0f} {gbs / PEAK:9.3f} {tot:9.2f}")
    print()
