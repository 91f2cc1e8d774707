#!/bin/bash
# round 6, call 53: graphed_training_step(flat=True) -- parity tests (single process, DDP wrapper) and the Lightning-shaped legs at cfg2
python -m pytest tests -q -m gpu -x -k "graphed_flat_step or graphed_training_step_equals_eager or torch_distributed_data_parallel" 2>&1 | tail -5
python bench.py --gpus 1 --steps 20 --warmup 5 --no-also > gpurun_out/flat_bench.json 2> gpurun_out/flat_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/flat_bench.json").read().strip().splitlines()[-1])
print("cfg2 ms_per_step", round(d["ms_per_step"],4))
print({k: (round(v,3) if isinstance(v,float) else v) for k,v in d["lightning_shaped"].items() if k!="what"})
PY
