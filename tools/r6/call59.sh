#!/bin/bash
# round 6, call 59: reduce_jobs_kernel on 4-wave workgroups (libnlam_hip.so) against 16 (libnlam_rw16.so = the same sources with -DNLAM_RED_WAVES=16)
R=$GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -x -k "reduce or wgrad or cfg2_hip_graph or trajectory or graphed_flat" 2>&1 | tail -3
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for rep in 1 2 3; do for lib in libnlam_hip.so libnlam_rw16.so; do
  NLAM_LIB=$R/neural_lam_amd/$lib python bench.py --steps 300 --warmup 20 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[cfg2] $lib", round(d["ms_per_step"],4), "loss", d["final_loss"])
PY
done; done
for rep in 1 2; do for lib in libnlam_hip.so libnlam_rw16.so; do
  NLAM_LIB=$R/neural_lam_amd/$lib python bench.py --config cfg3 --steps 12 --warmup 2 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[cfg3] $lib", round(d["ms_per_step"],3), "loss", d["final_loss"])
PY
  NLAM_LIB=$R/neural_lam_amd/$lib python bench.py --config cfg5 --precision bf16 --steps 4 --warmup 2 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[cfg5-bf16] $lib", round(d["ms_per_step"],3), "loss", d["final_loss"])
PY
done; done
for lib in libnlam_hip.so libnlam_rw16.so; do
  NLAM_LIB=$R/neural_lam_amd/$lib python bench.py --config cfg4 --steps 30 --warmup 3 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[cfg4] $lib", round(d["ms_per_step"],3))
PY
done
