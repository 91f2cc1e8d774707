#!/bin/bash
# round 6, call 28: is the history dependence of the `also` legs (cfg3 40.9 / 42.3 / 45.3, cfg5 106 / 112 / 121 ms after different earlier legs
# of the same process) the physical placement of freshly allocated memory?  expandable segments = 2 MB physical granules mapped by VMM
mkdir -p gpurun_out/r6c28
ALL="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg"
run() { tag=$1; shift
  python bench.py --gpus 1 --steps 20 --warmup 5 "$@" > gpurun_out/r6c28/$tag.json 2> gpurun_out/r6c28/$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r6c28/$tag.json").read().strip().splitlines()[-1]); a=d.get("also") or {}
    print("$tag", "cfg2", round(d["ms_per_step"],4), {k: round(v["ms_per_step"],3) for k,v in a.items()})
except Exception as e: print("$tag FAILED", e); print(open("gpurun_out/r6c28/$tag.err").read()[-600:])
PY
}
run default_none $ALL
run default_lightning --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path
export PYTORCH_HIP_ALLOC_CONF=expandable_segments:True
export PYTORCH_CUDA_ALLOC_CONF=expandable_segments:True
run expand_none $ALL
run expand_lightning --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path
run expand_all
