# PMC passes over tools/win_counters.py (k_spmv_win): which unit bounds the LDS-window SpMV?  Output: gpurun_out/$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmc_win}
mkdir -p $OUT
cd $R
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')  # the real interpreter: no exec hop behind rocprofv3
export REPS=2
i=0
while read -r CTRS; do
  i=$((i+1))
  echo "pass $i: $CTRS" >> $OUT/progress.log
  timeout -k 10 150 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/p$i -- "$PY" tools/win_counters.py ${N:-128} > $OUT/p$i.log 2>&1 || echo "pass $i failed" >> $OUT/progress.log
done <<'LIST'
GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr
TD_TD_BUSY_sum TD_TC_STALL_sum
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_LEVEL_WAVES SQ_ACTIVE_INST_LDS
FETCH_SIZE
WRITE_SIZE
LIST
python3 tools/pmc_summary.py $OUT k_spmv_win > $OUT/summary.txt 2>&1
