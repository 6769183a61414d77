# Round 5: HBM traffic (FETCH_SIZE, WRITE_SIZE: their own passes) and TA / TD / SQ counters of assemble_first in its two
# launch forms (width bins, row blocks) on the bench meshes.  usage: tools/pmc_af_r05.sh <outdir-name> "<af_bench args>"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmc_af_r05}
ARGS=${2:-128}
mkdir -p $OUT
cd $R
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')  # the real interpreter: no exec hop behind rocprofv3
i=0
while read -r CTRS; do
  i=$((i+1))
  echo "pass $i: $CTRS" >> $OUT/progress.log
  timeout -k 10 300 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/p$i -- "$PY" tools/af_bench.py $ARGS > $OUT/p$i.log 2>&1 || echo "pass $i failed" >> $OUT/progress.log
done <<'LIST'
FETCH_SIZE
WRITE_SIZE
GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr
TD_TD_BUSY_sum TD_TC_STALL_sum
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
LIST
python3 tools/pmc_summary.py $OUT k_assemble_rows > $OUT/summary.txt 2>&1
