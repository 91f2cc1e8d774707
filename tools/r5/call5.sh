#!/bin/bash
# Round 5, call 5: the whole GPU suite under the new defaults, the drop-in leg with side-stream weight gradients, side-stream count.
mkdir -p gpurun_out/r5
LOG=gpurun_out/r5
timeout 1500 python -m pytest tests -q -m gpu -x > $LOG/pytest_gpu.log 2>&1
tail -8 $LOG/pytest_gpu.log
python bench.py --steps 100 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg2', round(d['ms_per_step'],4), d['matmul_mode'], json.dumps({k:v for k,v in d['lightning_shaped'].items() if k!='what'}))"
run() { echo "[$1 $2 $4] $(env $1 python bench.py --config $2 $4 --steps $3 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg 2>$LOG/last_err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4),'ms/step', 'forecast', round(d['forecast_steps_per_s'],1), 'final', d['final_loss'])
except Exception as e: print('ERR', e)
")"; }
for n in 4 3 6 8; do run "NLAM_WGRAD_STREAMS=$n" cfg2 300; done
run "NLAM_MATMUL_BWD=bf16x3" cfg2 300
run "NLAM_X=1" cfg2 300
