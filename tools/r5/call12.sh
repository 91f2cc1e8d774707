#!/bin/bash
mkdir -p gpurun_out/r5
LOG=gpurun_out/r5
run() { echo "[$1 $2 $4] $(env $1 python bench.py --config $2 $4 --steps $3 --warmup 3 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg 2>$LOG/last_err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4),'ms/step', 'forecast', round(d['forecast_steps_per_s'],1), 'final', d['final_loss'], 'regions', len(d['timed_regions_ms']))
except Exception as e: print('ERR', e)
")"; }
for rep in 1 2; do run "NLAM_X=1" cfg2 300; run "NLAM_WGRAD_MIN_PARTS=64" cfg2 300; done
run "NLAM_X=1" cfg4 60; run "NLAM_WGRAD_MIN_PARTS=64" cfg4 60
run "NLAM_X=1" cfg4p 60; run "NLAM_WGRAD_MIN_PARTS=64" cfg4p 60
run "NLAM_X=1" cfg3 12 "--precision bf16"; run "NLAM_WGRAD_MIN_PARTS=64" cfg3 12 "--precision bf16"
run "NLAM_X=1" cfg3 12; run "NLAM_WGRAD_MIN_PARTS=64" cfg3 12
