mkdir -p gpurun_out/r2m
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2m/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/r2m/bench_under_rocprof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python tools/step_timeline.py $(find gpurun_out/r2m/tr -name "*kernel_trace.csv" | head -1) > gpurun_out/r2m/cfg2_step_timeline.txt
cp $(find gpurun_out/r2m/tr -name "*kernel_stats.csv" | head -1) gpurun_out/r2m/bench_cfg2_kernel_stats.csv
rm -rf gpurun_out/r2m/tr
grep "^# " gpurun_out/r2m/cfg2_step_timeline.txt | head -40
