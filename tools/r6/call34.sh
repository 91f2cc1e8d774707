#!/bin/bash
# round 6, call 34: the default placement (sides on the two queues neither chain nor caller use) and a few more layouts; cfg2 / cfg4 / cfg4p unaffected?
for h in "" "train"; do for q in "" "1,1" "3,3" "3,2" "2,2"; do
  echo "[QUEUE_SIDES='$q'] $(NLAM_QUEUE_SIDES=$q python tools/r6/history_probe.py cfg3 fp32 10 $h 2>&1 | grep 'history=\|Warning' | cut -c1-60)"
done; done
for h in "" "train"; do for q in "" "3,3"; do
  echo "[QUEUE_SIDES='$q'] $(NLAM_QUEUE_SIDES=$q python tools/r6/history_probe.py cfg5 bf16 5 $h 2>&1 | grep 'history=\|Warning' | cut -c1-60)"
done; done
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for rep in 1 2; do for q in "" "0"; do for c in "cfg2 --steps 300" "cfg4 --steps 30" "cfg4p --steps 30"; do
  NLAM_QUEUE_SIDES=$q python bench.py --config $c --warmup 3 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[$c] QUEUE_SIDES='$q'", round(d["ms_per_step"],4))
PY
done; done; done
