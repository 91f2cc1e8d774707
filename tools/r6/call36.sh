#!/bin/bash
# round 6, call 36: the segmented executor at d <= 128 again, now that its chain stream has a queue of its own (finding 39 measured 1.94 vs 1.75 ms)
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
for rep in 1 2; do for e in "forks" "segments"; do for c in "cfg2 --steps 300" "cfg4 --steps 30" "cfg4p --steps 30"; do
  NLAM_EXEC=$e python bench.py --config $c --warmup 3 $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[$c] NLAM_EXEC=$e", round(d["ms_per_step"],4))
PY
done; done; done
for f in 6 24 48; do NLAM_EXEC=segments NLAM_SEG_FORKS=$f python bench.py --config cfg2 --steps 300 --warmup 3 $B > /tmp/x.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("[cfg2] segments forks/segment=$f", round(d["ms_per_step"],4))
PY
done
