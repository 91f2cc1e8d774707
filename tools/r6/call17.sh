#!/bin/bash
# round 6, call 17: row slices of the small (6 561-row) weight gradients: NLAM_WGRAD_MIN_PARTS 64 (default at d >= 256) / 32 / 16 / 8
mkdir -p gpurun_out/r6c17
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for rep in 1 2; do for mp in 64 32 16 8; do
  NLAM_WGRAD_MIN_PARTS=$mp python bench.py --config cfg5 --precision bf16 --steps 5 --warmup 2 $B > gpurun_out/r6c17/c5_$mp_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r6c17/c5_$mp_$rep.json").read().strip().splitlines()[-1]); print("cfg5 min_parts=$mp", round(d["ms_per_step"],3))
PY
  NLAM_WGRAD_MIN_PARTS=$mp python bench.py --config cfg3 --steps 12 --warmup 2 $B > gpurun_out/r6c17/c3_$mp_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r6c17/c3_$mp_$rep.json").read().strip().splitlines()[-1]); print("cfg3 min_parts=$mp", round(d["ms_per_step"],3))
PY
done; done
for mp in 64 16; do echo "== min_parts $mp"; NLAM_WGRAD_MIN_PARTS=$mp NLAM_KB_AUTOCAST=1 python tools/kernel_bench.py m2m 12 512 2>&1 | grep wgrad; NLAM_WGRAD_MIN_PARTS=$mp python tools/kernel_bench.py m2m 12 256 2>&1 | grep wgrad; done
