#!/bin/bash
# A/B of the 8-wave split-bf16 wide kernels against the 4-wave "two workgroups per CU" instantiations (NLAM_WBF_HALF bits:
# 1 = forward, 2 = backward), inside one gpurun call: isolated launches (kernel_bench) and captured steps (bench.py).
#   gpurun -- 'bash tools/ab_half.sh'
OUT=${OUT:-gpurun_out/half}; mkdir -p $OUT
kb() { # which d autocast half
  NLAM_KB_AUTOCAST=$3 NLAM_WBF_HALF=$4 python tools/kernel_bench.py $1 12 $2 edge 2>&1 | sed "s/^/[half=$4 ac=$3] /"; }
step() { # cfg prec steps half
  NLAM_WBF_HALF=$4 python bench.py --config $1 --precision $2 --steps $3 --warmup 2 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('half=$4', '$1', '$2', round(d['ms_per_step'],3), 'ms/step', round(d['forecast_steps_per_s'],1), 'forecast steps/s', 'loss', d['final_loss'])"; }
{
for half in 0 3; do kb m2m 256 0 $half; kb m2g 256 0 $half; kb m2m 256 1 $half; kb m2m 512 1 $half; kb m2g 512 1 $half; kb m2m 128 0 $half; done
} > $OUT/kernel_bench.log 2>&1
{
for rep in 1 2; do for half in 0 1 3; do step cfg3 fp32 12 $half; done; done
for rep in 1 2; do for half in 0 1; do step cfg5 bf16 4 $half; done; done
for half in 0 3; do step cfg3 bf16 12 $half; step cfg4 fp32 30 $half; done
} > $OUT/steps.log 2>&1
