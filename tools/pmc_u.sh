# TA / TCP / TD / SQ counter passes of the VELOCITY-side SpMV kernels (tools/spmv_counters.py): which unit is busy?
# Few counters per pass, every pass under its own timeout, the real interpreter directly after `--`
# (the profiler's preloaded library has initialised the GPU before the program starts: no exec hop allowed).
# output under gpurun_out/$1; env: N
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmc_u}
mkdir -p $OUT
cd $R
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
N=${N:-128}
i=0
while read -r CTRS; do
  i=$((i+1))
  echo "pass $i: $CTRS" >> $OUT/progress.log
  timeout -k 10 240 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/p$i -- "$PY" tools/spmv_counters.py $N > $OUT/p$i.log 2>&1 || echo "pass $i failed" >> $OUT/progress.log
done <<'LIST'
GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TD_TD_BUSY_sum TD_TC_STALL_sum
SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
LIST
python3 tools/pmc_summary.py $OUT k_spmv > $OUT/summary.txt 2>&1
