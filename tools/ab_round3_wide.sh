#!/bin/bash
# Round-3 A/B of the wide configurations on one box (fresh process per arm).   gpurun -- 'bash tools/ab_round3_wide.sh'
mkdir -p gpurun_out/ab3w
run() { tag=$1; cfg=$2; steps=$3; prec=$4; shift 4; env "$@" python bench.py --config $cfg --precision $prec --steps $steps --warmup 3 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path > gpurun_out/ab3w/$tag.json 2>gpurun_out/ab3w/$tag.err; python -c "
import json; d=json.load(open('gpurun_out/ab3w/$tag.json')); print('$tag', round(d['ms_per_step'],3), 'ms/step', round(d['forecast_steps_per_s'],1), 'forecast steps/s', 'loss', d['final_loss'])" || tail -3 gpurun_out/ab3w/$tag.err; }
run cfg3_base cfg3 20 fp32 A=1
run cfg3_smallwin cfg3 20 fp32 NLAM_WGRAD_BIG_MIN_ROWS=32768 NLAM_WGRAD_MIN_PARTS=32
run cfg3_smallwin128 cfg3 20 fp32 NLAM_WGRAD_BIG_MIN_ROWS=32768
run cfg5_base cfg5 6 bf16 A=1
run cfg5_smallwin cfg5 6 bf16 NLAM_WGRAD_BIG_MIN_ROWS=32768 NLAM_WGRAD_MIN_PARTS=32
run cfg2_base cfg2 300 fp32 A=1
run cfg2_base2 cfg2 300 fp32 A=1
