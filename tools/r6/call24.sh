#!/bin/bash
# round 6, call 24: MORE workgroups than CUs for the wide chain kernels (the dispatcher balances them around co-running weight gradients)
mkdir -p gpurun_out/r6c24
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for rep in 1 2; do for w in 256 384 512 1024; do
  for c in "cfg5 --precision bf16 --steps 5" "cfg3 --steps 12"; do
  NLAM_CHAIN_CUS=$w python bench.py --config $c --warmup 2 $B > gpurun_out/r6c24/x.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r6c24/x.json").read().strip().splitlines()[-1]); print("[$c] chain_cus=$w", round(d["ms_per_step"],3), d.get("final_loss"))
except Exception as ex: print("[$c] chain_cus=$w FAILED", ex)
PY
done; done; done
for w in 256 512; do echo "== chain_cus $w"; NLAM_CHAIN_CUS=$w NLAM_KB_AUTOCAST=1 python tools/kernel_bench.py m2m 12 512 2>&1 | grep "mlp_"; done
