#!/bin/bash
# round 6, call 18: the wide weight gradients' floor is 64 workgroups, not 64 slices (default) against NLAM_WGRAD_MIN_PARTS=64 (round 5)
mkdir -p gpurun_out/r6c18
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_boundary.py -x -q -m gpu -k "wgrad or boundary or weight" > gpurun_out/r6c18/pytest.log 2>&1; tail -2 gpurun_out/r6c18/pytest.log
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for rep in 1 2; do for mp in "" 64; do
  for c in "cfg5 --precision bf16 --steps 5" "cfg3 --steps 12" "cfg3 --precision bf16 --steps 12" "cfg4 --steps 30"; do
  NLAM_WGRAD_MIN_PARTS=$mp python bench.py --config $c --warmup 2 $B > gpurun_out/r6c18/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r6c18/x.json").read().strip().splitlines()[-1]); print("[$c] min_parts='$mp'", round(d["ms_per_step"],3))
PY
done; done; done
