#!/bin/bash
mkdir -p gpurun_out/r5
LOG=gpurun_out/r5
run() { echo "[$1 $2 $4] $(env $1 python bench.py --config $2 $4 --steps $3 --warmup 3 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg 2>$LOG/last_err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4),'ms/step', 'forecast', round(d['forecast_steps_per_s'],1), 'final', d['final_loss'], 'regions', len(d['timed_regions_ms']), d['launch_mode'][:30])
except Exception as e: print('ERR', e)
")"; }
run "NLAM_X=1" cfg3 12
for n in 2 3 6; do run "NLAM_WGRAD_STREAMS=$n" cfg3 12; done
run "GPU_MAX_HW_QUEUES=8" cfg3 12
run "GPU_MAX_HW_QUEUES=8 NLAM_WGRAD_STREAMS=6" cfg3 12
run "NLAM_X=1" cfg5 3 "--precision bf16"
run "NLAM_WGRAD_STREAMS=3" cfg5 3 "--precision bf16"
run "GPU_MAX_HW_QUEUES=8" cfg5 3 "--precision bf16"
run "GPU_MAX_HW_QUEUES=8" cfg2 300
run "GPU_MAX_HW_QUEUES=8 NLAM_EXEC=segments NLAM_SEG_FORKS=3" cfg2 300
run "GPU_MAX_HW_QUEUES=8 NLAM_EXEC=segments NLAM_SEG_FORKS=5" cfg2 300
