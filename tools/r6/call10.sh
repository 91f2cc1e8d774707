#!/bin/bash
# Round 6, call 10: is the side-stream weight-gradient work a win at all at the wide configurations?  (one stream vs forks vs segments)
mkdir -p gpurun_out/r6
LOG=gpurun_out/r6
run() { echo "[$1 $2 $4] $(env $1 python bench.py --config $2 $4 --steps $3 --warmup 2 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also 2>$LOG/last_err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4),'ms/step', 'forecast', round(d['forecast_steps_per_s'],1), 'final', d['final_loss'])
except Exception as e: print('ERR', e)
")"; }
for v in "NLAM_OVERLAP_WGRAD=1" "NLAM_OVERLAP_WGRAD=0" "NLAM_OVERLAP_WGRAD=1 NLAM_WGRAD_STREAMS=2" "NLAM_OVERLAP_WGRAD=1 NLAM_WGRAD_STREAMS=1"; do run "$v" cfg5 4 "--precision bf16"; done 2>&1 | tee $LOG/ab_overlap.log
for v in "NLAM_OVERLAP_WGRAD=1" "NLAM_OVERLAP_WGRAD=0" "NLAM_OVERLAP_WGRAD=1 NLAM_WGRAD_STREAMS=2" "NLAM_OVERLAP_WGRAD=1 NLAM_WGRAD_STREAMS=1"; do run "$v" cfg3 8; done 2>&1 | tee -a $LOG/ab_overlap.log
for v in "NLAM_OVERLAP_WGRAD=1" "NLAM_OVERLAP_WGRAD=0"; do run "$v" cfg2 200; done 2>&1 | tee -a $LOG/ab_overlap.log
tail -3 $LOG/last_err.log
