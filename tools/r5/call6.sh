#!/bin/bash
mkdir -p gpurun_out/r5
LOG=gpurun_out/r5
timeout 1800 python -m pytest tests -q -m gpu -x > $LOG/pytest_gpu.log 2>&1
tail -8 $LOG/pytest_gpu.log
python tools/r5/linear_bench.py 2>/dev/null > $LOG/linear_bench.log; cat $LOG/linear_bench.log
run() { echo "[$1 $2 $4] $(env $1 python bench.py --config $2 $4 --steps $3 --warmup 3 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg 2>$LOG/last_err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4),'ms/step', 'forecast', round(d['forecast_steps_per_s'],1), 'final', d['final_loss'])
except Exception as e: print('ERR', e)
")"; }
run "NLAM_GRAD_MAILBOX=0" cfg5 3 "--precision bf16"; run "NLAM_GRAD_MAILBOX=1" cfg5 3 "--precision bf16"
run "NLAM_GRAD_MAILBOX=0" cfg3 12; run "NLAM_GRAD_MAILBOX=1" cfg3 12
python bench.py --steps 100 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg2', round(d['ms_per_step'],4), d['matmul_mode'], json.dumps({k:v for k,v in d['lightning_shaped'].items() if k!='what'}))"
