#!/bin/bash
# round 6, call 20: NLAM_WGRAD_MAX_WGS 256 (rounds 2-5) / 128 (new default) / 64 / 192
mkdir -p gpurun_out/r6c20
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_boundary.py -x -q -m gpu -k "wgrad or boundary or weight" > gpurun_out/r6c20/pytest.log 2>&1; tail -2 gpurun_out/r6c20/pytest.log
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for rep in 1 2; do for w in 256 128 64 192; do
  for c in "cfg5 --precision bf16 --steps 5" "cfg3 --steps 12" "cfg3 --precision bf16 --steps 12" "cfg4 --steps 30" "cfg4p --steps 30"; do
  NLAM_WGRAD_MAX_WGS=$w python bench.py --config $c --warmup 2 $B > gpurun_out/r6c20/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r6c20/x.json").read().strip().splitlines()[-1]); print("[$c] max_wgs=$w", round(d["ms_per_step"],3))
PY
done; done; done
