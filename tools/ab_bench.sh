#!/bin/bash
# A/B of the launch-shape / scheduling knobs on the cfg2 training step (one fresh process per arm, same box):
#   gpurun -- 'bash tools/ab_bench.sh'
# Arms are environment variables read at import: NLAM_WGRAD_CHUNKS (32-row chunks a weight-gradient workgroup streams),
# NLAM_WGRAD_STREAMS (side streams of the weight-gradient work), NLAM_EARLY_LEAF (early backward of leaf MLPs),
# NLAM_FACTORISE_MIN_EDGES (factorised edge MLP at d <= 64), NLAM_MATMUL (matrix mode).
mkdir -p gpurun_out/ab
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-gpu-baseline --no-roofline --steps 200 > gpurun_out/ab/$tag.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/ab/$tag.json')); print('$tag', round(d['ms_per_step'],4), 'ms/step', round(d['forecast_steps_per_s'],1), 'forecast steps/s')"; }
run base A=1
run base_again A=1
run early_leaf NLAM_EARLY_LEAF=1
for c in 1 4 16; do run wgrad_chunks$c NLAM_WGRAD_CHUNKS=$c; done
for s in 1 2 8; do run wgrad_streams$s NLAM_WGRAD_STREAMS=$s; done
run factorised_from_65536_edges NLAM_FACTORISE_MIN_EDGES=65536
for m in bf16x2 bf16 f32; do run matmul_$m NLAM_MATMUL=$m; done
python tools/chain_only.py 2>&1 | tail -1
