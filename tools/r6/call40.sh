#!/bin/bash
# round 6, call 40: knobs once more with the streams placed by hardware queue
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for rep in 1 2; do for e in "X=1" "NLAM_WGRAD_MAX_WGS=112" "NLAM_WGRAD_MAX_WGS=144" "NLAM_WGRAD_MAX_WGS=176" "NLAM_SEG_FORKS=8" "NLAM_SEG_FORKS=16" "NLAM_SEG_FORKS=24" "NLAM_WGRAD_CHUNKS=12"; do
  for c in "cfg5 --precision bf16 --steps 5" "cfg3 --steps 12"; do
  env $e python bench.py --config $c --warmup 2 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[$c] $e", round(d["ms_per_step"],3))
PY
done; done; done
