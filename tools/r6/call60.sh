#!/bin/bash
# round 6, call 60: wgrad_dma_kernel's first chunk: one wait for its gather indices, eight LDS-DMAs back to back (libnlam_hip.so) against
# the build before (libnlam_prev.so: every address computation of the first chunk behind its own s_waitcnt vmcnt(0))
R=$GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -x -k "wgrad or cfg2_hip_graph or graphed_flat" 2>&1 | tail -3
for lib in libnlam_hip.so libnlam_prev.so; do echo "== $lib"; NLAM_LIB=$R/neural_lam_amd/$lib python tools/kernel_bench.py m2m 12 64 2>&1 | grep "wgrad"; NLAM_LIB=$R/neural_lam_amd/$lib python tools/kernel_bench.py m2g 12 64 2>&1 | grep "wgrad"; done
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for rep in 1 2 3; do for lib in libnlam_hip.so libnlam_prev.so; do
  NLAM_LIB=$R/neural_lam_amd/$lib python bench.py --steps 300 --warmup 20 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[cfg2] $lib", round(d["ms_per_step"],4), "loss", d["final_loss"])
PY
done; done
for lib in libnlam_hip.so libnlam_prev.so libnlam_hip.so libnlam_prev.so; do
  NLAM_LIB=$R/neural_lam_amd/$lib python bench.py --config cfg4 --steps 30 --warmup 3 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[cfg4] $lib", round(d["ms_per_step"],3))
PY
done
