#!/bin/bash
# round 6, call 56: what mlp_pack_kernel's 32 us at the head of a cfg2 step are made of (tools/r6/pack_probe.py)
python tools/r6/pack_probe.py 2>&1 | grep -v amdgpu.ids | tail -12
