#!/bin/bash
# round 6, call 16: which plan for the launches the lean kernels do not take (node MLPs at 6 561 / 63 784 rows): NLAM_WBF_HALF 0 / 1 / 3
mkdir -p gpurun_out/r6c16
for h in 1 0 3; do
  echo "== NLAM_WBF_HALF=$h"
  NLAM_WBF_HALF=$h NLAM_KB_AUTOCAST=1 python tools/kernel_bench.py m2m 12 512 2>&1 | grep -v amdgpu.ids | grep "mlp_\|linear"
  NLAM_WBF_HALF=$h NLAM_KB_AUTOCAST=1 python tools/kernel_bench.py m2g 8 512 2>&1 | grep -v amdgpu.ids | grep "mlp_"
  NLAM_WBF_HALF=$h NLAM_KB_AUTOCAST=1 python tools/kernel_bench.py m2m 12 256 2>&1 | grep -v amdgpu.ids | grep "mlp_"
done > gpurun_out/r6c16/kernels.log 2>&1
cat gpurun_out/r6c16/kernels.log
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for rep in 1 2; do for h in 1 0 3; do
  NLAM_WBF_HALF=$h python bench.py --config cfg5 --precision bf16 --steps 5 --warmup 2 $B > gpurun_out/r6c16/c5_h${h}_$rep.json 2>/dev/null
  NLAM_WBF_HALF=$h python bench.py --config cfg3 --precision bf16 --steps 12 --warmup 2 $B > gpurun_out/r6c16/c3b_h${h}_$rep.json 2>/dev/null
  python - <<PY
import json
for c in ("c5","c3b"):
    d=json.loads(open("gpurun_out/r6c16/%s_h${h}_$rep.json" % c).read().strip().splitlines()[-1]); print(c, "half=$h", round(d["ms_per_step"],3))
PY
done; done
