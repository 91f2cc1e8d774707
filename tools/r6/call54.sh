#!/bin/bash
# round 6, call 54: nlam_linear's LDS-tiled GEMM with 64-column chunks (NLAM_LIN_SK=4, the one-term 64-row-tile launches) against
# the 32-column chunks of round 5 (libnlam_sk2.so = the same sources built with -DNLAM_LIN_SK=2): isolated launches, parity, cfg5 / cfg3-bf16 steps
R=$GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -x -k "node_linear or linear_gemm or factoris" 2>&1 | tail -3
for lib in libnlam_hip.so libnlam_sk2.so; do
  echo "== $lib"
  NLAM_LIB=$R/neural_lam_amd/$lib python tools/r5/linear_bench.py 2>&1 | grep "^6561\|^rows\|^63784 512 512 bf16 "
done
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for rep in 1 2; do for lib in libnlam_hip.so libnlam_sk2.so; do
  NLAM_LIB=$R/neural_lam_amd/$lib python bench.py --config cfg5 --precision bf16 --steps 4 --warmup 2 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[cfg5-bf16] $lib", round(d["ms_per_step"],3))
PY
  NLAM_LIB=$R/neural_lam_amd/$lib python bench.py --config cfg3 --precision bf16 --steps 12 --warmup 2 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[cfg3-bf16] $lib", round(d["ms_per_step"],3))
PY
done; done
