# rocprofv3 passes of the bench command (kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own PMC
# passes, as the guide prescribes) for the headline workload AND for the unstructured leg (refined Delaunay mesh,
# Beltrami field: the LDS-window kernels k_spmv_win).  Output under gpurun_out/; condensed into profiles/ by
# tools/summarize_prof.py.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-prof_r06}
cd $R
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')  # the real interpreter: no exec hop behind rocprofv3
for CFG in "box|-N 128" "delaunay|--mesh delaunay -N 32 --refine 2 --workload beltrami"; do
  NAME=${CFG%%|*}
  ARGS=${CFG#*|}
  OUT=$R/gpurun_out/${TAG}_$NAME
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- "$PY" bench.py $ARGS --steps 6 --warmup 2 --no-cpu --no-pmc --no-extras > $OUT/bench_trace.json 2> $OUT/trace.err
  echo $NAME trace-done
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- "$PY" bench.py $ARGS --steps 1 --warmup 1 --no-cpu --no-pmc --no-extras > $OUT/bench_fetch.json 2> $OUT/fetch.err
  echo $NAME fetch-done
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- "$PY" bench.py $ARGS --steps 1 --warmup 1 --no-cpu --no-pmc --no-extras > $OUT/bench_write.json 2> $OUT/write.err
  echo $NAME write-done
done
