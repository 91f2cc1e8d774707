#!/bin/bash
# round 6, call 49: mlp_bwd_edge_kernel<1, 512, .> with the A-fragment ring NOT carried across super tiles (scratch 264 -> 172 B per lane with bf16
# storage, 188 -> 88 without): two builds, one call
R=$GRAFT_REPO_ROOT
for lib in libnlam_hip.so libnlam_ring0.so; do
  echo "== $lib"
  NLAM_LIB=$R/neural_lam_amd/$lib NLAM_KB_AUTOCAST=1 python tools/kernel_bench.py m2m 12 512 2>&1 | grep "mlp_bwd', 57616"
  NLAM_LIB=$R/neural_lam_amd/$lib NLAM_KB_AUTOCAST=1 python tools/kernel_bench.py m2g 8 512 2>&1 | grep "mlp_bwd', 255136"
done
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for rep in 1 2 3; do for lib in libnlam_hip.so libnlam_ring0.so; do
  NLAM_LIB=$R/neural_lam_amd/$lib python bench.py --config cfg5 --precision bf16 --steps 5 --warmup 2 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[cfg5] $lib", round(d["ms_per_step"],3))
PY
done; done
NLAM_LIB=$R/neural_lam_amd/libnlam_ring0.so timeout 900 python -m pytest tests/test_hip_parity.py tests/test_full_size_parity.py -x -q -m gpu -k "512 or cfg5 or wide" 2>&1 | tail -2
