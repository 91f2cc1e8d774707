#!/bin/bash
# Round-3 A/B of the wide configurations on one box: weights packed once per step (NLAM_PACK_WIDE) on / off.
mkdir -p gpurun_out/ab3w
run() { tag=$1; cfg=$2; steps=$3; shift 3; env "$@" python bench.py --config $cfg --steps $steps --warmup 3 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path > gpurun_out/ab3w/$tag.json 2>gpurun_out/ab3w/$tag.err; python -c "
import json; d=json.load(open('gpurun_out/ab3w/$tag.json')); print('$tag', round(d['ms_per_step'],3), 'ms/step', round(d['forecast_steps_per_s'],1), 'forecast steps/s', 'loss', d['final_loss'])" || tail -3 gpurun_out/ab3w/$tag.err; }
run cfg4_pack cfg4 60 A=1
run cfg4_nopackwide cfg4 60 NLAM_PACK_WIDE=0
run cfg4_nopack cfg4 60 NLAM_PACK_WEIGHTS=0
run cfg4p_pack cfg4p 60 A=1
run cfg4p_nopack cfg4p 60 NLAM_PACK_WEIGHTS=0
run cfg3_pack cfg3 20 A=1
run cfg3_nopackwide cfg3 20 NLAM_PACK_WIDE=0
tag=cfg5_bf16_pack; python bench.py --config cfg5 --precision bf16 --steps 8 --warmup 2 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path > gpurun_out/ab3w/$tag.json 2>gpurun_out/ab3w/$tag.err; python -c "import json; d=json.load(open('gpurun_out/ab3w/cfg5_bf16_pack.json')); print('cfg5_bf16_pack', round(d['ms_per_step'],3), 'ms/step', round(d['forecast_steps_per_s'],1))" || tail -3 gpurun_out/ab3w/$tag.err
