#!/bin/bash
# Round 6, call 7: early backward of static-feature embedders (NLAM_EARLY_EMB) A/B; phase cycles of the d = 512 forward plans
mkdir -p gpurun_out/r6
LOG=gpurun_out/r6
run() { echo "[$1 $2 $4] $(env $1 python bench.py --config $2 $4 --steps $3 --warmup 3 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also 2>$LOG/last_err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4),'ms/step', 'forecast', round(d['forecast_steps_per_s'],1), 'final', d['final_loss'])
except Exception as e: print('ERR', e)
")"; }
for e in "" m2g "m2g,m2m" all "" m2g; do run "NLAM_EARLY_EMB=$e" cfg2 300; done 2>&1 | tee $LOG/ab_early_emb.log
for e in "" m2g all; do run "NLAM_EARLY_EMB=$e" cfg3 8; done 2>&1 | tee -a $LOG/ab_early_emb.log
for e in "" m2g all; do run "NLAM_EARLY_EMB=$e" cfg4 40; done 2>&1 | tee -a $LOG/ab_early_emb.log
tail -3 $LOG/last_err.log
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_full_size_parity.py -x -q -m gpu -k "cfg2 or golden or graph or trainer or segment" 2>&1 | tail -3
export NLAM_LIB=neural_lam_amd/libnlam_hip_timing.so
for h in 0 1; do
  echo "=== NLAM_WBF_HALF=$h m2m 512 bf16"; NLAM_WBF_HALF=$h timeout 300 python tools/phase_timing_wbf.py m2m 512 bf16 2>&1 | grep -v amdgpu.ids | grep -A14 "training"
done > $LOG/phase_cycles_d512_plans.txt
cat $LOG/phase_cycles_d512_plans.txt
