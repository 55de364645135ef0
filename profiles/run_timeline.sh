# kernel trace of bench.py + concurrency picture of one replayed step.  usage: bash profiles/run_timeline.sh <tag> [env VAR=..]
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary $BENCH_ARGS > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
cd $GRAFT_REPO_ROOT
T=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
python profiles/overlap_timeline.py $T --list > $OUT/timeline.txt
python profiles/trace_summary.py $T 9 > $OUT/kernel_trace_summary.txt
rm -rf $OUT/prof
head -6 $OUT/timeline.txt
