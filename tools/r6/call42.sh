#!/bin/bash
# round 6, call 42: the fp32-operand forms of wgrad_ldma_kernel again, now that launches are sized to co-run (NLAM_WGRAD_LDMA bit 1: one term, bit 2: three terms)
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for rep in 1 2; do for v in 1 3 5 7; do
  for c in "cfg5 --precision bf16 --steps 5" "cfg3 --steps 12" "cfg3 --precision bf16 --steps 12"; do
  NLAM_WGRAD_LDMA=$v python bench.py --config $c --warmup 2 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[$c] NLAM_WGRAD_LDMA=$v", round(d["ms_per_step"],3))
PY
done; done; done
