#!/bin/bash
# Round 5, call 13: branch-free chunk accessors (V4 instantiations) of the split-bf16 wide kernels: parity, then A/B in one process environment each
mkdir -p gpurun_out/r5
LOG=gpurun_out/r5
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_full_size_parity.py -x -q -m gpu -k "wide or d128 or d256 or d512 or golden or cfg3 or cfg5_two or autocast or bf16_storage or chunk or hilam" > $LOG/call13_tests.log 2>&1
tail -4 $LOG/call13_tests.log
run() { echo "[$1 $2 $4] $(env $1 python bench.py --config $2 $4 --steps $3 --warmup 3 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg 2>$LOG/last_err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4),'ms/step', 'forecast', round(d['forecast_steps_per_s'],1), 'final', d['final_loss'], 'regions', len(d['timed_regions_ms']))
except Exception as e: print('ERR', e)
")"; }
for v in 0 1; do run "NLAM_WBF_V4=$v" cfg3 12; done
for v in 0 1; do run "NLAM_WBF_V4=$v" cfg5 3 "--precision bf16"; done
for v in 0 1; do run "NLAM_WBF_V4=$v" cfg3 12 "--precision bf16"; done
for v in 0 1; do run "NLAM_WBF_V4=$v" cfg4 60; done
for v in 0 1; do run "NLAM_WBF_V4=$v" cfg4p 60; done
