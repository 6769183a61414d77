# Pressure-CG SpMV at 128^3 (and 256^3): packed pair-slot stream against the strided one (ox_sell.ps_stride = 3: code
# addresses computable, codes requested with ps_ptr) and persistent launches with the next slice's codes prefetched
# (ox_sell.ps_grid).  VARIANTS=7,15: entry stream (reference bits) / pair-slot stream.
set -e
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ps_stride_sweep.txt
: > $OUT
for N in 128 256; do
for CFG in "0 0" "3 0" "3 1536" "3 2048" "3 3072" "3 4096" "3 6144"; do
  set -- $CFG
  echo "== N=$N STRIDE=$1 GRID=$2" >> $OUT
  REAL=stiff STRIDE=$1 GRID=$2 VARIANTS=7,15 ROUNDS=5 REPS=200 python tools/spmv_bench.py $N p 2>&1 | grep -E "bit-identical|variant=15|pair-slot|dictionary" >> $OUT
done
done
cat $OUT
