"""Per-kernel averages of a rocprofv3 --pmc pass: python tools/pmc_summary.py <dir> [name-filter]"""
import collections, csv, glob, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "k_"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{src}/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if flt in k:
            acc[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), d in sorted(acc.items()):
    print(k, "grid", g)
    for c, v in sorted(d.items()):
        print(f"    {c:28s} n={len(v):4d} avg={sum(v)/len(v):16.1f}")
