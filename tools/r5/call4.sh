#!/bin/bash
mkdir -p gpurun_out/r5
LOG=gpurun_out/r5
timeout 300 python tools/r5/diag_exec2.py cfg4p 3 8 > $LOG/diag_exec2_cfg4p.log 2>&1; grep -v "amdgpu\|Warning\|warn" $LOG/diag_exec2_cfg4p.log | tail -45
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "graphed_training or optimizer_schedule" > $LOG/call4_tests.log 2>&1
tail -5 $LOG/call4_tests.log
