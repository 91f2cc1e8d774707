#!/bin/bash
# round 6, call 58: segment sums with row ids and rows requested in two groups (ids first) and the last 1 .. 7 rows of a segment in flight
# together; libnlam_sk2.so = the build before (ids and rows interleaved: one s_waitcnt vmcnt(0) per row, row-by-row tail)
R=$GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -x -k "segment or split_receivers or mailbox or golden or full_size" 2>&1 | tail -3
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for rep in 1 2 3; do for lib in libnlam_hip.so libnlam_sk2.so; do
  NLAM_LIB=$R/neural_lam_amd/$lib python bench.py --steps 300 --warmup 20 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[cfg2] $lib", round(d["ms_per_step"],4), "loss", d["final_loss"])
PY
done; done
for rep in 1 2; do for lib in libnlam_hip.so libnlam_sk2.so; do
  NLAM_LIB=$R/neural_lam_amd/$lib python bench.py --config cfg3 --steps 12 --warmup 2 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[cfg3] $lib", round(d["ms_per_step"],3), "loss", d["final_loss"])
PY
  NLAM_LIB=$R/neural_lam_amd/$lib python bench.py --config cfg5 --precision bf16 --steps 4 --warmup 2 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[cfg5-bf16] $lib", round(d["ms_per_step"],3), "loss", d["final_loss"])
PY
done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o t -- python $R/bench.py --steps 20 --warmup 5 $B > /dev/null 2>&1
cd $R
python tools/step_timeline.py $(find /tmp/tr -name "*kernel_trace.csv" | head -1) > gpurun_out/cfg2_step_timeline_new.txt
grep "segment_sum\|mlp_pack\|^# One" gpurun_out/cfg2_step_timeline_new.txt | cut -c1-110
