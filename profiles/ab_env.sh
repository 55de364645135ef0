# A/B of environment settings on one box: [BENCH_ARGS="--no-graph"] bash profiles/ab_env.sh "VAR=1" "VAR=2 OTHER=x" ...   (first run: no settings)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
i=0
for cfg in "" "$@"; do
  echo "== [$cfg] $BENCH_ARGS"
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary $BENCH_ARGS > gpurun_out/ab/b$i.json 2> gpurun_out/ab/b$i.err || tail -5 gpurun_out/ab/b$i.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab/b$i.json").read().strip().splitlines()[-1])
print("   ms_per_step", d["ms_per_step"], "value", d["value"])
PY
  i=$((i+1))
done
