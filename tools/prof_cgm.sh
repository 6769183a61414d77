# rocprofv3 kernel trace of a short bench run with the merged-reduction CG on / off (A/B of the pressure iteration's kernels)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
for M in true false; do
  OUT=$R/gpurun_out/prof_cgm_$M
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- "$PY" bench.py --steps 3 --warmup 2 --no-cpu --no-pmc --no-extras --cg-merged $M > $OUT/bench.json 2> $OUT/err.log
  python3 - <<P
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print("merged=$M")
for r in rows:
    n=r["Name"]
    if any(k in n for k in ("k_spmv_ps","k_cg_update","k_cgm","k_ksp_scalar","k_prereduce")): print("  ", n[:60], r["Calls"], "avg_us %.2f" % (float(r["AverageNs"])/1e3))
P
done
