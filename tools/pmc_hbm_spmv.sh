# HBM traffic (FETCH_SIZE x 2, WRITE_SIZE; separate passes) of the SpMV micro-benchmark, e.g. the 256^3
# pressure matrix that no longer fits the Infinity Cache.   N=256 W=p bash tools/pmc_hbm_spmv.sh out_tag
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmc_hbm_spmv}
mkdir -p $OUT
cd $R
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')  # the real interpreter: no exec hop behind rocprofv3
export REPS=5 ROUNDS=1 VARIANTS=${VARIANTS:-15} REAL=${REAL:-stiff}
N=${N:-256}
W=${W:-p}
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- "$PY" tools/spmv_bench.py $N $W > $OUT/$C.log 2>&1 || echo "$C failed" >> $OUT/progress.log
done
python3 tools/pmc_summary.py $OUT k_spmv > $OUT/summary.txt 2>&1
