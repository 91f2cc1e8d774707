#!/bin/bash
# Round 5, call 2: the segmented executor (trainer._SegmentedStep) against the one-graph executor, same box.
mkdir -p gpurun_out/r5
LOG=gpurun_out/r5
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "segmented_executor or optimizer_schedule or graph_step or rollout_gradients or prepacked or early_leaf or falls_back" > $LOG/exec_tests.log 2>&1
tail -5 $LOG/exec_tests.log
run() { echo "[$1 $2] $(env $1 python bench.py --config $2 $4 --steps $3 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path 2>$LOG/last_err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4),'ms/step', 'forecast', round(d['forecast_steps_per_s'],1), 'final', d['final_loss'])
except Exception as e: print('ERR', e)
")"; tail -3 $LOG/last_err.log | grep -v amdgpu.ids; }
run "NLAM_EXEC=forks" cfg2 300
for k in 1 2 3 5; do run "NLAM_EXEC=segments NLAM_SEG_FORKS=$k" cfg2 300; done
run "NLAM_EXEC=segments NLAM_SEG_FORKS=3 NLAM_CHAIN_PRIO=0" cfg2 300
run "NLAM_EXEC=forks" cfg2 300
run "NLAM_EXEC=forks" cfg3 12; run "NLAM_EXEC=segments NLAM_SEG_FORKS=3" cfg3 12; run "NLAM_EXEC=segments NLAM_SEG_FORKS=8" cfg3 12
run "NLAM_EXEC=forks" cfg4 60; run "NLAM_EXEC=segments NLAM_SEG_FORKS=3" cfg4 60
run "NLAM_EXEC=forks" cfg4p 60; run "NLAM_EXEC=segments NLAM_SEG_FORKS=3" cfg4p 60
# timeline of the replayed cfg2 step, segmented executor
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5
cd /tmp && export TMPDIR=/tmp
NLAM_EXEC=segments rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path > $OUT/bench_cfg2_segments_under_rocprofv3.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python tools/step_timeline.py $(find $OUT/tr -name "*kernel_trace.csv" | head -1) > $OUT/cfg2_step_timeline_segments.txt
rm -rf $OUT/tr
tail -25 $OUT/cfg2_step_timeline_segments.txt
# which assertion fails under bf16x2 at bench size, and by how much
NLAM_MATMUL=bf16x2 timeout 300 python -m pytest tests/test_full_size_parity.py -q -m gpu -k "cfg2_training_step" 2>&1 | grep -E "^E  |assert" | head -20
