#!/bin/bash
# Round-3 A/B on one box (fresh process per arm, arms interleaved twice).   gpurun -- 'bash tools/ab_round3.sh'
mkdir -p gpurun_out/ab3
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --steps 300 > gpurun_out/ab3/$tag.json 2>gpurun_out/ab3/$tag.err; python -c "
import json; d=json.load(open('gpurun_out/ab3/$tag.json')); print('$tag', round(d['ms_per_step'],4), 'ms/step', round(d['forecast_steps_per_s'],1), 'forecast steps/s', 'loss', d['final_loss'])" || tail -3 gpurun_out/ab3/$tag.err; }
for rep in 1 2; do
  run base_$rep A=1
  run pack_off_$rep NLAM_PACK_WEIGHTS=0
done
