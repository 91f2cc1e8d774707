#!/bin/bash
# Round-3 A/B on one box (fresh process per arm).   gpurun -- 'bash tools/ab_round3.sh'
mkdir -p gpurun_out/ab3
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --steps 300 > gpurun_out/ab3/$tag.json 2>gpurun_out/ab3/$tag.err; python -c "
import json; d=json.load(open('gpurun_out/ab3/$tag.json')); print('$tag', round(d['ms_per_step'],4), 'ms/step', round(d['forecast_steps_per_s'],1), 'forecast steps/s', 'loss', d['final_loss'])" || tail -3 gpurun_out/ab3/$tag.err; }
run base_1 A=1
for s in 2 3 6 8; do run streams_$s NLAM_WGRAD_STREAMS=$s; done
run base_2 A=1
for c in 4 16; do run chunks_$c NLAM_WGRAD_CHUNKS=$c; done
run leaf_unfused NLAM_FUSED_LEAF_WGRAD=0
run base_3 A=1
