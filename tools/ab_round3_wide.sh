#!/bin/bash
# Round-3 A/B of the wide configurations on one box (fresh process per arm).   gpurun -- 'bash tools/ab_round3_wide.sh'
mkdir -p gpurun_out/ab3w
run() { tag=$1; cfg=$2; steps=$3; shift 3; env "$@" python bench.py --config $cfg --steps $steps --warmup 3 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path > gpurun_out/ab3w/$tag.json 2>gpurun_out/ab3w/$tag.err; python -c "
import json; d=json.load(open('gpurun_out/ab3w/$tag.json')); print('$tag', round(d['ms_per_step'],3), 'ms/step', round(d['forecast_steps_per_s'],1), 'forecast steps/s', 'loss', d['final_loss'])" || tail -3 gpurun_out/ab3w/$tag.err; }
run cfg4_base cfg4 60 A=1
run cfg4_minparts32 cfg4 60 NLAM_WGRAD_MIN_PARTS=32
run cfg4_minparts64 cfg4 60 NLAM_WGRAD_MIN_PARTS=64
run cfg3_base cfg3 20 A=1
run cfg3_minparts32 cfg3 20 NLAM_WGRAD_MIN_PARTS=32
run cfg3_minparts64 cfg3 20 NLAM_WGRAD_MIN_PARTS=64
run cfg2_minparts64 cfg2 300 NLAM_WGRAD_MIN_PARTS=64
run cfg2_base cfg2 300 A=1
