mkdir -p gpurun_out/r2f
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-gpu-baseline --no-roofline --steps 200 > gpurun_out/r2f/$tag.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r2f/$tag.json')); print('$tag', round(d['ms_per_step'],4), round(d['forecast_steps_per_s'],1))"; }
run base A=1
run base2 A=1
run early NLAM_EARLY_LEAF=1
run chunks8 NLAM_WGRAD_CHUNKS=8
run chunks4 NLAM_WGRAD_CHUNKS=4
run fact NLAM_FACTORISE_MIN_EDGES=65536
run fact200k NLAM_FACTORISE_MIN_EDGES=200000
run streams2 NLAM_WGRAD_STREAMS=2
run streams8 NLAM_WGRAD_STREAMS=8
