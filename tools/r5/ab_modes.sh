#!/bin/bash
# Round 5, call 1: bf16x3 (old default) vs bf16x2 (accepted as fp32-class by VERDICT r4) on every fp32 config, same box,
# then the full-size parity suite under bf16x2 at the unchanged tolerances.
mkdir -p gpurun_out/r5
run() { echo "[$1 $2] $(env $1 python bench.py --config $2 $4 --steps $3 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4),'ms/step', d['matmul_mode'], 'forecast', round(d['forecast_steps_per_s'],1), 'loss0', d['loss_step0'], 'final', d['final_loss'])
except Exception as e: print('ERR', e)
")"; }
for rep in 1 2; do for m in bf16x3 bf16x2; do run "NLAM_MATMUL=$m" cfg2 300; done; done
for m in bf16x3 bf16x2; do run "NLAM_MATMUL=$m" cfg3 12; run "NLAM_MATMUL=$m" cfg4 60; run "NLAM_MATMUL=$m" cfg4p 60; done
NLAM_MATMUL=bf16x2 timeout 900 python -m pytest tests/test_full_size_parity.py -q -m gpu 2>&1 | tail -15
NLAM_MATMUL=bf16x2 timeout 900 python -m pytest tests/test_hip_parity.py tests/test_latent.py tests/test_graph_efm.py -q -m gpu 2>&1 | tail -15
