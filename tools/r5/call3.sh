#!/bin/bash
# Round 5, call 3: new linear GEMM, drop-in path, parity by matrix mode, executor diagnostic, A/B of XCD placement / GEMM / hybrid modes.
mkdir -p gpurun_out/r5
LOG=gpurun_out/r5
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "node_linear or graphed_training or wide_embedder or interaction_net_with_two or segmented_executor or optimizer_schedule" > $LOG/call3_tests.log 2>&1
tail -15 $LOG/call3_tests.log
timeout 300 python tools/r5/parity_modes.py cfg2 > $LOG/parity_by_matmul_mode.log 2>$LOG/parity_err.log; cat $LOG/parity_by_matmul_mode.log; tail -3 $LOG/parity_err.log | grep -v amdgpu
timeout 300 python tools/r5/diag_exec.py cfg4p 3 > $LOG/diag_exec_cfg4p.log 2>&1; grep -v amdgpu $LOG/diag_exec_cfg4p.log | tail -60
run() { echo "[$1 $2 $4] $(env $1 python bench.py --config $2 $4 --steps $3 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg 2>$LOG/last_err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4),'ms/step', 'forecast', round(d['forecast_steps_per_s'],1), 'final', d['final_loss'])
except Exception as e: print('ERR', e)
")"; tail -3 $LOG/last_err.log | grep -v amdgpu.ids; }
NOX=neural_lam_amd/libnlam_hip_noxcd.so
for rep in 1 2; do run "NLAM_LIB=$NOX" cfg2 300; run "NLAM_X=1" cfg2 300; done
run "NLAM_MATMUL=bf16x2 NLAM_MATMUL_BWD=bf16x3" cfg2 300
run "NLAM_MATMUL=bf16x3 NLAM_MATMUL_BWD=bf16x2" cfg2 300
run "NLAM_MATMUL=bf16x2" cfg2 300
run "NLAM_LIB=$NOX" cfg3 12; run "NLAM_X=1" cfg3 12; run "NLAM_LIN_GEMM=0" cfg3 12
run "NLAM_LIB=$NOX" cfg4 60; run "NLAM_X=1" cfg4 60
run "NLAM_LIB=$NOX" cfg5 3 "--precision bf16"; run "NLAM_X=1" cfg5 3 "--precision bf16"; run "NLAM_LIN_GEMM=0" cfg5 3 "--precision bf16"
python bench.py --steps 100 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg2', round(d['ms_per_step'],4), json.dumps(d['lightning_shaped'], indent=1))"
