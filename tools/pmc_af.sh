# PMC passes over tools/af_bench.py (k_assemble_rows<3,2,2,...>: the fused assemble_first): which unit bounds it?
# Output: gpurun_out/$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmc_af}
mkdir -p $OUT
cd $R
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')  # the real interpreter: no exec hop behind rocprofv3
i=0
while read -r CTRS; do
  i=$((i+1))
  echo "pass $i: $CTRS" >> $OUT/progress.log
  timeout -k 10 200 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/p$i -- "$PY" tools/af_bench.py ${N:-128} > $OUT/p$i.log 2>&1 || echo "pass $i failed" >> $OUT/progress.log
done <<'LIST'
GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr
TD_TD_BUSY_sum TD_TC_STALL_sum
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU
LIST
python3 tools/pmc_summary.py $OUT k_assemble_rows > $OUT/summary.txt 2>&1
