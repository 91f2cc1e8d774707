#!/bin/bash
# Round 6, call 9: gradient hand-over across the AR steps of a rollout (NLAM_F_ACC_DSRC0): parity + A/B
mkdir -p gpurun_out/r6
LOG=gpurun_out/r6
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_full_size_parity.py -x -q -m gpu -k "handover or rollout or cfg3 or cfg5 or wgrad or d512" 2>&1 | tail -5
run() { echo "[$1 $2 $4] $(env $1 python bench.py --config $2 $4 --steps $3 --warmup 2 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also 2>$LOG/last_err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4),'ms/step', 'forecast', round(d['forecast_steps_per_s'],1), 'final', d['final_loss'])
except Exception as e: print('ERR', e)
")"; }
for v in 0 1 0 1; do run "NLAM_ROLLOUT_ACC=$v" cfg5 4 "--precision bf16"; done 2>&1 | tee $LOG/ab_rollout_acc.log
for v in 0 1 0 1; do run "NLAM_ROLLOUT_ACC=$v" cfg3 8; done 2>&1 | tee -a $LOG/ab_rollout_acc.log
tail -3 $LOG/last_err.log
