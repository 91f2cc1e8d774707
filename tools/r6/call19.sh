#!/bin/bash
# round 6, call 19: 32-row chunks per weight-gradient workgroup (NLAM_WGRAD_CHUNKS, default 8): fewer, longer slices of the BIG edge sets too
mkdir -p gpurun_out/r6c19
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for rep in 1 2; do for ch in 8 16 32 64; do
  for c in "cfg5 --precision bf16 --steps 5" "cfg3 --steps 12" "cfg2 --steps 300"; do
  NLAM_WGRAD_CHUNKS=$ch python bench.py --config $c --warmup 2 $B > gpurun_out/r6c19/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r6c19/x.json").read().strip().splitlines()[-1]); print("[$c] chunks=$ch", round(d["ms_per_step"],3))
PY
done; done; done
