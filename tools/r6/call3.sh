#!/bin/bash
# Round 6, call 3: plan 7 A/B (tuning value accepted now), CU-masked side / chain streams under the segmented executor
mkdir -p gpurun_out/r6
LOG=gpurun_out/r6
for h in 1 5; do
  echo "== NLAM_WBF_HALF=$h"
  NLAM_WBF_HALF=$h NLAM_KB_AUTOCAST=1 python tools/kernel_bench.py m2m 12 512 edge 2>&1 | grep -v amdgpu.ids | grep "mlp_fwd\|mlp_bwd"
  NLAM_WBF_HALF=$h NLAM_KB_AUTOCAST=1 python tools/kernel_bench.py m2g 8 512 edge 2>&1 | grep -v amdgpu.ids | grep "mlp_fwd\|mlp_bwd"
done 2>&1 | tee $LOG/ab_plan7_kernels.log
NLAM_WBF_HALF=5 timeout 600 python -m pytest tests/test_hip_parity.py tests/test_full_size_parity.py -x -q -m gpu -k "d512 or cfg5 or bf16_storage or autocast" 2>&1 | tail -3
run() { echo "[$1 $2 $4] $(env $1 python bench.py --config $2 $4 --steps $3 --warmup 2 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also 2>$LOG/last_err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4),'ms/step', 'forecast', round(d['forecast_steps_per_s'],1), 'final', d['final_loss'])
except Exception as e: print('ERR', e)
")"; }
for h in 1 5 1 5; do run "NLAM_WBF_HALF=$h" cfg5 4 "--precision bf16"; done 2>&1 | tee $LOG/ab_plan7_cfg5.log
for c in 0 32 64 96; do run "NLAM_SIDE_CUS=$c" cfg3 8; done 2>&1 | tee $LOG/ab_cumask.log
for c in 0 64 96; do run "NLAM_SIDE_CUS=$c" cfg5 3 "--precision bf16"; done 2>&1 | tee -a $LOG/ab_cumask.log
for c in 0 32 64; do run "NLAM_SIDE_CUS=$c NLAM_EXEC=segments" cfg2 100; done 2>&1 | tee -a $LOG/ab_cumask.log
for c in 32 64; do run "NLAM_SIDE_CUS=$c NLAM_CHAIN_MASKED=0" cfg3 8; done 2>&1 | tee -a $LOG/ab_cumask.log
tail -3 $LOG/last_err.log
