#!/bin/bash
# round 6, call 41: the driver's command line three times on one box (run-to-run spread of value and of the `also` legs)
for i in 1 2 3; do
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/driver_rep_$i.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/driver_rep_$i.json").read().strip().splitlines()[-1]); a=d.get("also") or {}
print("run $i: cfg2", round(d["ms_per_step"],4), "value", round(d["value"],1), "frac", round(d["roofline"]["frac"],3), {k: round(v["ms_per_step"],3) for k,v in a.items()}, "regions", [round(x,2) for x in d["timed_regions_ms"]])
PY
done
