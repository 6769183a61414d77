# SQ counter passes of the SpMV micro-benchmark (tools/spmv_bench.py); output under gpurun_out/$1
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmc_spmv}
mkdir -p $OUT
cd $R
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')  # the real interpreter: no exec hop behind rocprofv3
export REPS=3 ROUNDS=1 PALETTE=${PALETTE:-14} VARIANTS=${VARIANTS:-7}
N=${N:-128}
W=${W:-p}
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/p1 -- "$PY" tools/spmv_bench.py $N $W > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD --output-format csv -d $OUT/p2 -- "$PY" tools/spmv_bench.py $N $W > $OUT/p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p3 -- "$PY" tools/spmv_bench.py $N $W > $OUT/p3.log 2>&1
python3 tools/pmc_summary.py $OUT k_spmv > $OUT/summary.txt 2>&1
