#!/bin/bash
mkdir -p gpurun_out/r5
LOG=gpurun_out/r5
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "node_linear" > $LOG/call8_tests.log 2>&1
tail -3 $LOG/call8_tests.log
run() { echo "[$1 $2 $4] $(env $1 python bench.py --config $2 $4 --steps $3 --warmup 3 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg 2>$LOG/last_err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4),'ms/step', 'forecast', round(d['forecast_steps_per_s'],1), 'final', d['final_loss'], 'regions', len(d['timed_regions_ms']), d['launch_mode'][:40])
except Exception as e: print('ERR', e)
")"; }
for k in 4 8 12 20; do run "NLAM_SEG_FORKS=$k" cfg3 12; done
for k in 8 12 24; do run "NLAM_SEG_FORKS=$k" cfg5 3 "--precision bf16"; done
for k in 8 12 20; do run "NLAM_SEG_FORKS=$k" cfg4 60; done
run "NLAM_EXEC=forks" cfg3 12 "--precision bf16"; run "NLAM_SEG_FORKS=12" cfg3 12 "--precision bf16"
run "NLAM_X=1" cfg2 300; run "NLAM_X=1" cfg4p 60
