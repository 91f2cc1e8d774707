#!/bin/bash
# round 6, call 25: mlp_bwd_edge_kernel<3, 256, false> (the lean backward for the three-term d = 256 edge layers of cfg3): parity + A/B
mkdir -p gpurun_out/r6c25
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_full_size_parity.py -x -q -m gpu -k "256 or wide or cfg3 or mailbox or factoris" > gpurun_out/r6c25/pytest.log 2>&1; tail -3 gpurun_out/r6c25/pytest.log
for e in 3 1; do echo "== NLAM_WBF_EDGE=$e"; NLAM_WBF_EDGE=$e python tools/kernel_bench.py m2m 12 256 2>&1 | grep "mlp_bwd"; NLAM_WBF_EDGE=$e python tools/kernel_bench.py m2g 8 256 2>&1 | grep "mlp_bwd"; NLAM_WBF_EDGE=$e python tools/kernel_bench.py g2m 8 256 2>&1 | grep "mlp_bwd"; done
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for rep in 1 2 3; do for e in 3 1; do
  NLAM_WBF_EDGE=$e python bench.py --config cfg3 --steps 12 --warmup 2 $B > gpurun_out/r6c25/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r6c25/x.json").read().strip().splitlines()[-1]); print("[cfg3] NLAM_WBF_EDGE=$e", round(d["ms_per_step"],3), d.get("final_loss"))
PY
done; done
