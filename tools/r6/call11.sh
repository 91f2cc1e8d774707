#!/bin/bash
# Round 6, call 11: mlp_fwd_edge_kernel (lean, software-pipelined forward of the factorised one-term edge layers): parity + A/B
mkdir -p gpurun_out/r6
LOG=gpurun_out/r6
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_full_size_parity.py -x -q -m gpu -k "d512 or d256 or cfg5 or cfg3 or autocast or bf16_storage or noise" 2>&1 | tail -5
for h in 0 1; do
  echo "== NLAM_WBF_EDGE=$h"
  NLAM_WBF_EDGE=$h NLAM_KB_AUTOCAST=1 python tools/kernel_bench.py m2m 12 512 edge 2>&1 | grep -v amdgpu.ids | grep "mlp_fwd"
  NLAM_WBF_EDGE=$h NLAM_KB_AUTOCAST=1 python tools/kernel_bench.py m2g 8 512 edge 2>&1 | grep -v amdgpu.ids | grep "mlp_fwd"
  NLAM_WBF_EDGE=$h NLAM_KB_AUTOCAST=1 python tools/kernel_bench.py g2m 8 512 edge 2>&1 | grep -v amdgpu.ids | grep "mlp_fwd"
  NLAM_WBF_EDGE=$h NLAM_KB_AUTOCAST=1 python tools/kernel_bench.py m2m 12 256 edge 2>&1 | grep -v amdgpu.ids | grep "mlp_fwd"
done 2>&1 | tee $LOG/ab_fwd_edge_kernels.log
run() { echo "[$1 $2 $4] $(env $1 python bench.py --config $2 $4 --steps $3 --warmup 2 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also 2>$LOG/last_err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4),'ms/step', 'forecast', round(d['forecast_steps_per_s'],1), 'final', d['final_loss'])
except Exception as e: print('ERR', e)
")"; }
for h in 0 1 0 1; do run "NLAM_WBF_EDGE=$h" cfg5 4 "--precision bf16"; done 2>&1 | tee $LOG/ab_fwd_edge_steps.log
for h in 0 1; do run "NLAM_WBF_EDGE=$h" cfg3 8 "--precision bf16"; done 2>&1 | tee -a $LOG/ab_fwd_edge_steps.log
tail -3 $LOG/last_err.log
