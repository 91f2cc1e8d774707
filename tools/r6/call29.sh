#!/bin/bash
# round 6, call 29: step time against a dummy allocation taken before the model is built (placement of the step's buffers)
for pad in 0 512 4096 65536 1049088 2097152 3145728 20975616 104857600 536870912; do python tools/r6/alloc_offset_probe.py cfg3 fp32 10 $pad 2>&1 | grep "pad="; done
for pad in 1049088 20975616 536870912; do python tools/r6/alloc_offset_probe.py cfg3 fp32 10 $pad 0 2>&1 | grep "pad="; done
for pad in 0 4096 1049088 20975616 536870912; do python tools/r6/alloc_offset_probe.py cfg5 bf16 5 $pad 2>&1 | grep "pad="; done
