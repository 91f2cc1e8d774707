#!/bin/bash
# round 6, call 32: streams placed by OBSERVED hardware queue (ops.stream_layout): layouts x process history
mkdir -p gpurun_out/r6c32
for h in "" "train"; do for q in "0" "" "2,2" "1,1,1" "2,1,1" "2,2,1"; do
  echo -n "[QUEUE_SIDES='$q'] "; NLAM_QUEUE_SIDES=$q python tools/r6/history_probe.py cfg3 fp32 10 $h 2>&1 | grep "history=\|Error" | cut -c1-80
done; done
for h in "" "train"; do for q in "0" "" "2,2" "1,1,1"; do
  echo -n "[QUEUE_SIDES='$q'] "; NLAM_QUEUE_SIDES=$q python tools/r6/history_probe.py cfg5 bf16 5 $h 2>&1 | grep "history=\|Error" | cut -c1-80
done; done
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from neural_lam_amd import ops
L = ops.stream_layout()
print("groups:", [len(g) for g in L["groups"]], "chain", L["chain"], "sides", L["sides"])
PY
