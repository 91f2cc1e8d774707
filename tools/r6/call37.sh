#!/bin/bash
# round 6, call 37: NLAM_F_WGRAD_SOLO -- weight gradients launched without side streams (the eager drop-in path) keep the full-width shape
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "wgrad or graphed or eager or autograd or interaction" > /tmp/pt.log 2>&1; tail -2 /tmp/pt.log
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-also"
for c in "cfg5 --precision bf16 --steps 4" "cfg3 --steps 12"; do
  python bench.py --config $c --warmup 2 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); l=d["lightning_shaped"]
print("[$c]", round(d["ms_per_step"],3), "lightning eager / graphed / fused", [round(l[k],2) for k in ("ms_per_step_eager_torch_adamw","ms_per_step_graphed_torch_adamw","ms_per_step_graphed_fused_adamw_torch_adamw")])
PY
done
