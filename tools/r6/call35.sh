#!/bin/bash
# round 6, call 35: final stream placement: tests + the driver's command line (the `also` legs after every cfg2 leg) + standalone
mkdir -p gpurun_out/r6c35
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "stream_layout or segmented or graphed or trainer" > gpurun_out/r6c35/pytest.log 2>&1; tail -2 gpurun_out/r6c35/pytest.log
for q in "" "0"; do
NLAM_QUEUE_SIDES=$q python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6c35/driver_$q.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/r6c35/driver_$q.json").read().strip().splitlines()[-1]); a=d.get("also") or {}
print("driver line QUEUE_SIDES='$q': cfg2", round(d["ms_per_step"],4), {k: round(v["ms_per_step"],3) for k,v in a.items()}, "lightning", {k: round(v,3) for k,v in d["lightning_shaped"].items() if k.startswith("ms_per")})
PY
done
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for c in "cfg5 --precision bf16 --steps 5" "cfg3 --steps 12" "cfg3 --precision bf16 --steps 12"; do for q in "" "0"; do
  NLAM_QUEUE_SIDES=$q python bench.py --config $c --warmup 2 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[$c] standalone QUEUE_SIDES='$q'", round(d["ms_per_step"],4))
PY
done; done
