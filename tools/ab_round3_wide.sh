#!/bin/bash
# Round-3 A/B of the wide configurations on one box (fresh process per arm).   gpurun -- 'bash tools/ab_round3_wide.sh'
mkdir -p gpurun_out/ab3w
run() { tag=$1; cfg=$2; steps=$3; prec=$4; shift 4; env "$@" python bench.py --config $cfg --precision $prec --steps $steps --warmup 3 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path > gpurun_out/ab3w/$tag.json 2>gpurun_out/ab3w/$tag.err; python -c "
import json; d=json.load(open('gpurun_out/ab3w/$tag.json')); print('$tag', round(d['ms_per_step'],3), 'ms/step', round(d['forecast_steps_per_s'],1), 'forecast steps/s', 'loss', d['final_loss'])" || tail -3 gpurun_out/ab3w/$tag.err; }
# is the factorised edge MLP still worth its node-level launches when the matrix cores run one bf16 term (autocast)?
run cfg5_base cfg5 6 bf16 A=1
run cfg5_plain cfg5 6 bf16 NLAM_FACTORISE_MIN_EDGES_WIDE=1073741824
run cfg3bf_base cfg3 12 bf16 A=1
run cfg3bf_plain cfg3 12 bf16 NLAM_FACTORISE_MIN_EDGES_WIDE=1073741824
run cfg5_base2 cfg5 6 bf16 A=1
run cfg3_base cfg3 12 fp32 A=1
run cfg4_base cfg4 40 fp32 A=1
run cfg4p_base cfg4p 40 fp32 A=1
