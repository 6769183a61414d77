# A/B of the merged-reduction CG inside one gpurun call: bench lines with --cg-merged false / true
#   bash tools/ab_cg_merged.sh "<bench args>"      e.g. "-N 64 --udeg 1"
set -e
ARGS=${1:-"-N 64 --udeg 1"}
TAG=$(echo "$ARGS" | tr -c 'A-Za-z0-9' '_')
for M in false true false true; do
  timeout -k 10 400 python bench.py --steps 10 --warmup 4 --no-cpu --no-pmc --no-extras $ARGS --cg-merged $M > gpurun_out/ab_${TAG}_$M.json 2> gpurun_out/ab_err.log
  python - <<P
import json
d=json.loads(open("gpurun_out/ab_${TAG}_$M.json").read().strip().splitlines()[-1])
pi=d.get("pressure_cg_iteration") or {}
print("$ARGS merged=$M steps/s %.3f  pressure it %.1f us  its %s  pressure_solve %.2f ms" % (d["value"], pi.get("us", 0), d["krylov_iterations_per_step"], d["phase_ms_per_step"].get("pressure_solve", 0)))
P
done
