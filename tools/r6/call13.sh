#!/bin/bash
# Round 6, call 13: mlp_bwd_edge_kernel (64-row backward of the factorised d = 512 one-term edge layer): parity + A/B
mkdir -p gpurun_out/r6
LOG=gpurun_out/r6
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_full_size_parity.py -x -q -m gpu -k "d512 or cfg5 or autocast or bf16_storage or noise or handover" 2>&1 | tail -5
for h in 0 1; do
  echo "== NLAM_WBF_EDGE=$h"
  NLAM_WBF_EDGE=$h NLAM_KB_AUTOCAST=1 python tools/kernel_bench.py m2m 12 512 edge 2>&1 | grep -v amdgpu.ids | grep "mlp_"
  NLAM_WBF_EDGE=$h NLAM_KB_AUTOCAST=1 python tools/kernel_bench.py m2g 8 512 edge 2>&1 | grep -v amdgpu.ids | grep "mlp_"
  NLAM_WBF_EDGE=$h NLAM_KB_AUTOCAST=1 python tools/kernel_bench.py g2m 8 512 edge 2>&1 | grep -v amdgpu.ids | grep "mlp_"
done 2>&1 | tee $LOG/ab_bwd_edge_kernels.log
run() { echo "[$1 $2 $4] $(env $1 python bench.py --config $2 $4 --steps $3 --warmup 2 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also 2>$LOG/last_err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4),'ms/step', 'forecast', round(d['forecast_steps_per_s'],1), 'final', d['final_loss'])
except Exception as e: print('ERR', e)
")"; }
for h in 0 1 0 1; do run "NLAM_WBF_EDGE=$h" cfg5 4 "--precision bf16"; done 2>&1 | tee $LOG/ab_bwd_edge_steps.log
tail -3 $LOG/last_err.log
