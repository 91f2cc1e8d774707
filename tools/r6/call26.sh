#!/bin/bash
# round 6, call 26: why is cfg5 4.5 % slower under `also` than standalone?  which earlier leg of the same process leaves something behind
mkdir -p gpurun_out/r6c26
ALL="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg"
run() { tag=$1; shift
  python bench.py --gpus 1 --steps 20 --warmup 5 "$@" > gpurun_out/r6c26/$tag.json 2> gpurun_out/r6c26/$tag.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r6c26/$tag.json").read().strip().splitlines()[-1]); a=d.get("also") or {}
print("$tag", "cfg2", round(d["ms_per_step"],4), {k: round(v["ms_per_step"],3) for k,v in a.items()})
PY
}
run none $ALL
run roofline --no-cpu-baseline --no-gpu-baseline --no-data-path --no-lightning-leg
run lightning --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path
run datapath --no-cpu-baseline --no-gpu-baseline --no-roofline --no-lightning-leg
run gpubase --no-cpu-baseline --no-roofline --no-data-path --no-lightning-leg
run cpubase --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg
