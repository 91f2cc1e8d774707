#!/bin/bash
# round 6, call 14: nlam_segment_sum_add (twin edge launch as mailbox consumer) -- tests + cfg2 A/B
mkdir -p gpurun_out/r6c14
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "mailbox or twin or segment or trajectory or interaction" > gpurun_out/r6c14/pytest.log 2>&1
tail -5 gpurun_out/r6c14/pytest.log
for i in 1 2; do
for mb in 1 0; do
  NLAM_GRAD_MAILBOX=$mb NLAM_BENCH_ALSO=0 timeout 600 python bench.py --steps 300 --warmup 20 > gpurun_out/r6c14/bench_mb${mb}_$i.json 2> gpurun_out/r6c14/bench_mb${mb}_$i.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r6c14/bench_mb${mb}_$i.json").read().strip().splitlines()[-1])
print("mailbox=$mb", d["ms_per_step"], d.get("value"))
PY
done
done
