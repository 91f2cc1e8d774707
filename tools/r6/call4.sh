#!/bin/bash
# Round 6, call 4: wgrad_ldma_kernel (LDS-DMA + transpose reads): parity, isolated launches, cfg5 / cfg3-bf16 step A/B
mkdir -p gpurun_out/r6
LOG=gpurun_out/r6
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_full_size_parity.py -x -q -m gpu -k "d512 or cfg5 or bf16_storage or autocast or wgrad" 2>&1 | tail -8
for v in 0 3; do
  echo "== NLAM_WGRAD_LDMA=$v"
  NLAM_WGRAD_LDMA=$v NLAM_KB_AUTOCAST=1 python tools/kernel_bench.py m2m 12 512 2>&1 | grep -v amdgpu.ids | grep "wgrad"
  NLAM_WGRAD_LDMA=$v NLAM_KB_AUTOCAST=1 python tools/kernel_bench.py m2g 8 512 2>&1 | grep -v amdgpu.ids | grep "wgrad"
  NLAM_WGRAD_LDMA=$v NLAM_KB_AUTOCAST=1 python tools/kernel_bench.py m2m 12 256 2>&1 | grep -v amdgpu.ids | grep "wgrad"
done 2>&1 | tee $LOG/ab_wgrad_ldma_kernels.log
run() { echo "[$1 $2 $4] $(env $1 python bench.py --config $2 $4 --steps $3 --warmup 2 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also 2>$LOG/last_err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4),'ms/step', 'forecast', round(d['forecast_steps_per_s'],1), 'final', d['final_loss'])
except Exception as e: print('ERR', e)
")"; }
for v in 0 1 3 0 3; do run "NLAM_WGRAD_LDMA=$v" cfg5 4 "--precision bf16"; done 2>&1 | tee $LOG/ab_wgrad_ldma_steps.log
for v in 0 3; do run "NLAM_WGRAD_LDMA=$v" cfg3 8 "--precision bf16"; done 2>&1 | tee -a $LOG/ab_wgrad_ldma_steps.log
tail -3 $LOG/last_err.log
