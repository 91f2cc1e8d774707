#!/bin/bash
# round 6, call 22: persistent grid of the wide chain kernels sized for fewer CUs (NLAM_CHAIN_CUS 256 default / 240 / 224 / 192)
mkdir -p gpurun_out/r6c22
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "wide and 512" > gpurun_out/r6c22/pytest.log 2>&1; tail -1 gpurun_out/r6c22/pytest.log
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for rep in 1 2; do for w in 256 240 224 192; do
  for c in "cfg5 --precision bf16 --steps 5" "cfg3 --steps 12"; do
  NLAM_CHAIN_CUS=$w python bench.py --config $c --warmup 2 $B > gpurun_out/r6c22/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r6c22/x.json").read().strip().splitlines()[-1]); print("[$c] chain_cus=$w", round(d["ms_per_step"],3))
PY
done; done; done
