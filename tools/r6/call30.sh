#!/bin/bash
# round 6, call 30: which earlier work of the process changes a later configuration's step time
for h in "" "nograd" "train" "nograd train" "nograd train fgraph" "fgraph" "eager" "gts" "nograd train fgraph eager gts"; do python tools/r6/history_probe.py cfg3 fp32 10 $h 2>&1 | grep "history=\|Error"; done
for h in "" "train" "eager" "gts" "nograd train fgraph eager gts"; do python tools/r6/history_probe.py cfg5 bf16 5 $h 2>&1 | grep "history=\|Error"; done
