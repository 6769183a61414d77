# TA / TCP / TD / LDS counter passes of the SpMV micro-benchmark (tools/spmv_bench.py): which unit is busy?
# Few counters per pass (the TA and TD blocks take two), every pass under its own timeout.
# output under gpurun_out/$1; env: N, W, VARIANTS, REAL / PALETTE
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmc_ta}
mkdir -p $OUT
cd $R
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')  # the real interpreter: no exec hop behind rocprofv3
export REPS=3 ROUNDS=1 VARIANTS=${VARIANTS:-7,15} REAL=${REAL:-stiff}
N=${N:-128}
W=${W:-p}
i=0
while read -r CTRS; do
  i=$((i+1))
  echo "pass $i: $CTRS" >> $OUT/progress.log
  timeout -k 10 150 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/p$i -- "$PY" tools/spmv_bench.py $N $W > $OUT/p$i.log 2>&1 || echo "pass $i failed" >> $OUT/progress.log
done <<'LIST'
GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TD_TD_BUSY_sum TD_TC_STALL_sum
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_LEVEL_WAVES
LIST
python3 tools/pmc_summary.py $OUT k_spmv > $OUT/summary.txt 2>&1
