#!/bin/bash
# round 6, call 61: cfg4 / cfg4p with the two builds of call 60 in the opposite order (is the 2 % of call 60 the build or the order?)
R=$GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for lib in libnlam_prev.so libnlam_hip.so libnlam_prev.so libnlam_hip.so libnlam_prev.so libnlam_hip.so; do
  NLAM_LIB=$R/neural_lam_amd/$lib python bench.py --config cfg4 --steps 60 --warmup 5 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[cfg4] $lib", round(d["ms_per_step"],3), [round(x/60,3) for x in d["timed_regions_ms"]])
PY
done
