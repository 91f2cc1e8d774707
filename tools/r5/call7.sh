#!/bin/bash
mkdir -p gpurun_out/r5
LOG=gpurun_out/r5
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "node_linear or gradient_mailbox" > $LOG/call7_tests.log 2>&1
tail -4 $LOG/call7_tests.log
python tools/r5/linear_bench.py 2>/dev/null > $LOG/linear_bench.log; cat $LOG/linear_bench.log
run() { echo "[$1 $2 $4] $(env $1 python bench.py --config $2 $4 --steps $3 --warmup 3 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg 2>$LOG/last_err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4),'ms/step', 'forecast', round(d['forecast_steps_per_s'],1), 'final', d['final_loss'], 'regions', len(d['timed_regions_ms']))
except Exception as e: print('ERR', e)
")"; }
run "NLAM_EXEC=forks" cfg4 60; for k in 6 16 48; do run "NLAM_EXEC=segments NLAM_SEG_FORKS=$k" cfg4 60; done
run "NLAM_EXEC=forks" cfg4p 60; for k in 6 16; do run "NLAM_EXEC=segments NLAM_SEG_FORKS=$k" cfg4p 60; done
run "NLAM_EXEC=forks" cfg3 12; for k in 8 24; do run "NLAM_EXEC=segments NLAM_SEG_FORKS=$k" cfg3 12; done
run "NLAM_EXEC=forks" cfg5 3 "--precision bf16"; run "NLAM_EXEC=segments NLAM_SEG_FORKS=16" cfg5 3 "--precision bf16"
