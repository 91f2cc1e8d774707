#!/bin/bash
mkdir -p gpurun_out/r6
LOG=gpurun_out/r6
python tools/r6/wgrad_check.py 2>&1 | grep -v amdgpu.ids | grep -v "^tensor\|^   \|^per 32" | tee $LOG/wgrad_check.log
python tools/r6/wgrad_check.py 255136 512 512 2>&1 | grep -v amdgpu.ids | grep -v "^tensor\|^   \|^per 32" | tee -a $LOG/wgrad_check.log
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_full_size_parity.py -x -q -m gpu -k "d512 or cfg5 or bf16_storage or autocast or wgrad or noise" 2>&1 | tail -5
run() { echo "[$1 $2 $4] $(env $1 python bench.py --config $2 $4 --steps $3 --warmup 2 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also 2>$LOG/last_err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4),'ms/step', 'forecast', round(d['forecast_steps_per_s'],1), 'final', d['final_loss'], 'loss0', d['loss_step0'])
except Exception as e: print('ERR', e)
")"; }
for v in 0 1 0 1; do run "NLAM_WGRAD_LDMA=$v" cfg5 4 "--precision bf16"; done 2>&1 | tee $LOG/ab_wgrad_ldma_steps.log
for v in 0 1; do run "NLAM_WGRAD_LDMA=$v" cfg3 8 "--precision bf16"; done 2>&1 | tee -a $LOG/ab_wgrad_ldma_steps.log
