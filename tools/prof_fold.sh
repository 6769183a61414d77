# rocprofv3 kernel trace of tools/cg_iter_bench.py with the folded CG update kernels on (OX_CG_FOLD_BLOCKS=512) / off (0)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
for B in ${FOLDS:-512 0}; do
  OUT=$R/gpurun_out/prof_fold_$B
  mkdir -p $OUT
  export OX_CG_FOLD_BLOCKS=$B
  ROUNDS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- "$PY" tools/cg_iter_bench.py ${N:-128} 300 > $OUT/out.log 2> $OUT/err.log
  python3 - <<P
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print("OX_CG_FOLD_BLOCKS=$B")
for r in rows:
    n=r["Name"]
    if any(k in n for k in ("k_spmv_ps","k_cg_update","k_cgm","k_ksp_scalar","k_prereduce")): print("  ", n[:70], r["Calls"], "avg_us %.2f" % (float(r["AverageNs"])/1e3))
P
done
