#!/bin/bash
# round 6, call 52: A-fragment ring depth of the three-term (d = 256) lean edge kernels: forward 2 (default) / 4 / 8, backward 4 (default) / 8; three builds
R=$GRAFT_REPO_ROOT
for lib in libnlam_hip.so libnlam_rd44.so libnlam_rd88.so; do
  echo "== $lib"
  NLAM_LIB=$R/neural_lam_amd/$lib python tools/kernel_bench.py m2m 12 256 2>&1 | grep "mlp_fwd', 57616\|mlp_bwd', 57616"
  NLAM_LIB=$R/neural_lam_amd/$lib python tools/kernel_bench.py m2g 8 256 2>&1 | grep "mlp_fwd', 255136\|mlp_bwd', 255136"
done
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for rep in 1 2; do for lib in libnlam_hip.so libnlam_rd44.so libnlam_rd88.so; do
  NLAM_LIB=$R/neural_lam_amd/$lib python bench.py --config cfg3 --steps 12 --warmup 2 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[cfg3] $lib", round(d["ms_per_step"],3))
PY
done; done
