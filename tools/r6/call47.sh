#!/bin/bash
# round 6, call 47: graphed_training_step records the eager module's weight-gradient shape (bit-identical at d = 512 too); its step time
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "graphed_training_step" 2>&1 | tail -3
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-also"
for c in "cfg5 --precision bf16 --steps 4" "cfg3 --steps 12" "cfg2 --steps 300"; do
  python bench.py --config $c --warmup 2 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); l=d["lightning_shaped"]
print("[$c]", round(d["ms_per_step"],3), "lightning eager / graphed / fused", [round(l[k],2) for k in ("ms_per_step_eager_torch_adamw","ms_per_step_graphed_torch_adamw","ms_per_step_graphed_fused_adamw_torch_adamw")])
PY
done
