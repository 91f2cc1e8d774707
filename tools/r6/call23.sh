#!/bin/bash
# round 6, call 23: remaining launch-shape knobs on the final kernel mix
mkdir -p gpurun_out/r6c23
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for rep in 1 2; do for e in "X=1" "NLAM_WGRAD_LDMA_VAR=1" "NLAM_WGRAD_LDMA_VAR=2" "NLAM_WGRAD_LDMA_VAR=3" "NLAM_LIN_WGS=128" "NLAM_LIN_WGS=0" "NLAM_EARLY_LEAF=1" "NLAM_WGRAD_MAX_WGS=96" "NLAM_WGRAD_MAX_WGS=160"; do
  for c in "cfg5 --precision bf16 --steps 5" "cfg3 --steps 12"; do
  env $e python bench.py --config $c --warmup 2 $B > gpurun_out/r6c23/x.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r6c23/x.json").read().strip().splitlines()[-1]); print("[$c] $e", round(d["ms_per_step"],3))
except Exception as ex: print("[$c] $e FAILED", ex)
PY
done; done; done
