# bench.py under rocprofv3 --kernel-trace (csv) + per-kernel summary.  usage: bash profiles/run_profile.sh <tag> [bench args]
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
cd $GRAFT_REPO_ROOT
T=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
python profiles/trace_summary.py $T 9 > $OUT/kernel_trace_summary.txt
S=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
cp $S $OUT/kernel_stats.csv
rm -rf $OUT/prof
head -5 $OUT/kernel_trace_summary.txt
