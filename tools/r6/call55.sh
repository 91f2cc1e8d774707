#!/bin/bash
# round 6, call 55: mlp_pack_kernel as one piece per workgroup (4 KB of code instead of 31 KB run once on a cold instruction cache):
# parity of the images, the cfg2 step against the previous kernel (libnlam_sk2.so: built before the change), the head of the step's timeline
R=$GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -x -k "prepacked or pack or graphed_training_step_equals_eager or cfg2_hip_graph" 2>&1 | tail -3
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for rep in 1 2 3; do for lib in libnlam_hip.so libnlam_sk2.so; do
  NLAM_LIB=$R/neural_lam_amd/$lib python bench.py --steps 300 --warmup 20 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[cfg2] $lib", round(d["ms_per_step"],4))
PY
done; done
for c in cfg4 cfg4p; do for lib in libnlam_hip.so libnlam_sk2.so; do
  NLAM_LIB=$R/neural_lam_amd/$lib python bench.py --config $c --steps 30 --warmup 3 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[$c] $lib", round(d["ms_per_step"],4))
PY
done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o t -- python $R/bench.py --steps 20 --warmup 5 $B > /dev/null 2>&1
cd $R
python tools/step_timeline.py $(find /tmp/tr -name "*kernel_trace.csv" | head -1) > gpurun_out/cfg2_step_timeline_pack.txt
head -12 gpurun_out/cfg2_step_timeline_pack.txt | cut -c1-120
