#!/bin/bash
# round 6, call 27: step time against the number of pool streams taken before the trainer takes its own (hardware-queue placement)
for n in 0 1 2 3 4 5 6 7; do python tools/r6/stream_offset_probe.py cfg5 bf16 5 $n 2>&1 | grep dummy; done
for n in 0 1 2 3 4 5 6 7; do python tools/r6/stream_offset_probe.py cfg3 fp32 10 $n 2>&1 | grep dummy; done
