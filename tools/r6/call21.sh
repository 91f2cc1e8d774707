#!/bin/bash
# round 6, call 21: cfg2 -- more / fewer slices of the narrow weight gradients (NLAM_WGRAD_CHUNKS 8 default, NLAM_WGRAD_MIN_PARTS 128 default)
mkdir -p gpurun_out/r6c21
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for rep in 1 2; do for e in "X=1" "NLAM_WGRAD_CHUNKS=4" "NLAM_WGRAD_CHUNKS=6" "NLAM_WGRAD_CHUNKS=12" "NLAM_WGRAD_MIN_PARTS=256" "NLAM_WGRAD_MIN_PARTS=96" "NLAM_WGRAD_STREAMS=3" "NLAM_WGRAD_STREAMS=5"; do
  env $e python bench.py --config cfg2 --steps 300 --warmup 10 $B > gpurun_out/r6c21/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r6c21/x.json").read().strip().splitlines()[-1]); print("[cfg2] $e", round(d["ms_per_step"],4))
PY
done; done
