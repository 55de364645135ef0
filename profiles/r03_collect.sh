# Round-3 measurement set, one GPU box: bench (full), kernel trace, PMC FETCH / WRITE passes of the SAME build, fp32 trace,
# data-parallel rehearsal (gloo, two ranks on the one GPU), micro-benchmarks.   bash profiles/r03_collect.sh
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r03
mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== bench (full)"; python bench.py --steps 20 --warmup 5 > $O/r03_bench_bf16.json 2> $O/bench.err; cut -c1-200 $O/r03_bench_bf16.json
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary"
echo "== kernel trace bf16"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -o t -- $B > $O/r03_bench_bf16_under_rocprof.json 2> $O/prof_bf16.err
echo "== pmc fetch"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o t -- $B > /dev/null 2> $O/pmc_fetch.err
echo "== pmc write"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o t -- $B > /dev/null 2> $O/pmc_write.err
echo "== kernel trace fp32"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fp32 -o t -- $B --dtype fp32 > $O/r03_bench_fp32_under_rocprof.json 2> $O/prof_fp32.err
cd $GRAFT_REPO_ROOT
python profiles/trace_summary.py $(find $O/prof_bf16 -name '*kernel_trace.csv' | head -1) 9 > $O/r03_bf16_kernel_trace_summary.txt
python profiles/overlap_timeline.py $(find $O/prof_bf16 -name '*kernel_trace.csv' | head -1) > $O/r03_bf16_overlap_timeline.txt || true
cp $(find $O/prof_bf16 -name '*kernel_stats.csv' | head -1) $O/r03_bf16_kernel_stats.csv
python profiles/trace_summary.py $(find $O/prof_fp32 -name '*kernel_trace.csv' | head -1) 9 > $O/r03_fp32_kernel_trace_summary.txt
cp $(find $O/prof_fp32 -name '*kernel_stats.csv' | head -1) $O/r03_fp32_kernel_stats.csv
python profiles/pmc_summary.py $O/pmc_fetch $O/pmc_write --traffic-json $O/r03_pmc_traffic.json > $O/r03_pmc_bf16_fetch_write.txt
rm -rf $O/prof_bf16 $O/prof_fp32 $O/pmc_fetch $O/pmc_write
head -12 $O/r03_bf16_kernel_trace_summary.txt
echo "== side-stream picture"
python profiles/side_stamps.py > $O/r03_side_stamps.txt 2>&1 || true
tail -3 $O/r03_side_stamps.txt
bash profiles/ab_env.sh "COMA_WGRAD_SIDE=0" "COMA_NO_DUO=1" "COMA_PREP_AHEAD=0" > $O/r03_ab_side_duo.txt 2>&1 || true
cat $O/r03_ab_side_duo.txt
echo "== graph-overlapped exchange: when are the gradient buckets sendable; what a stream outside the graph sees"
GPU_MAX_HW_QUEUES=8 python profiles/dp_watch_timeline.py 128 --marks 2>&1 | grep -v amdgpu.ids > $O/r03_dp_watch_timeline.txt || true
tail -7 $O/r03_dp_watch_timeline.txt
(echo "# default hardware queues"; python profiles/external_event_probe.py 2>&1 | grep -v amdgpu.ids; echo "# GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 python profiles/external_event_probe.py 2>&1 | grep -v amdgpu.ids) > $O/r03_external_event_probe.txt || true
echo "== microbenchmarks"
python profiles/microbench_norm.py > $O/r03_microbench_norm.txt 2>&1 || true
bash profiles/mb_tconv.sh > $O/r03_microbench_tconv.txt 2>&1 || true
OTHER= bash profiles/mb_halo2.sh > $O/r03_microbench_thick.txt 2>&1 || true
python profiles/microbench_wprep.py > $O/r03_microbench_wprep.txt 2>&1 || true
echo "== data-parallel rehearsal (gloo, 2 ranks on one GPU, 32^3)"
for mode in torch capi capi-sharded; do
  COMA_BENCH_ONE_DEVICE=1 COMA_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 2 --warmup 1 --size 32 --dp $mode > $O/r03_bench_gpus2_gloo_$mode.json 2> $O/dp_$mode.err || echo "dp $mode failed"
  tail -1 $O/r03_bench_gpus2_gloo_$mode.json | cut -c1-160
done
ls $O
