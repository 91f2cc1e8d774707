#!/bin/bash
# rocprofv3 kernel trace of the replayed cfg2 step -> per-step timeline + kernel stats under gpurun_out/round3/
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/round3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path > $OUT/bench_cfg2_under_rocprofv3.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python tools/step_timeline.py $(find $OUT/tr -name "*kernel_trace.csv" | head -1) > $OUT/cfg2_step_timeline.txt
cp $(find $OUT/tr -name "*kernel_stats.csv" | head -1) $OUT/bench_cfg2_kernel_stats.csv
rm -rf $OUT/tr
tail -40 $OUT/cfg2_step_timeline.txt
