#!/bin/bash
# round 6, call 57: mlp_pack_kernel with the job in scalar registers and the eight loads of a ragged / transposed item in flight together
R=$GRAFT_REPO_ROOT
python tools/r6/pack_probe.py 2>&1 | grep -v amdgpu.ids | tail -9
python -m pytest tests -q -m gpu -x -k "prepacked or pack or graphed_training_step_equals_eager or cfg2_hip_graph" 2>&1 | tail -3
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for rep in 1 2 3; do for lib in libnlam_hip.so libnlam_sk2.so; do
  NLAM_LIB=$R/neural_lam_amd/$lib python bench.py --steps 300 --warmup 20 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[cfg2] $lib", round(d["ms_per_step"],4))
PY
done; done
for c in cfg4 cfg4p; do for lib in libnlam_hip.so libnlam_sk2.so; do
  NLAM_LIB=$R/neural_lam_amd/$lib python bench.py --config $c --steps 30 --warmup 3 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[$c] $lib", round(d["ms_per_step"],4))
PY
done; done
