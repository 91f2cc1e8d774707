#!/bin/bash
# A/B of several builds of the library inside one gpurun call (boxes differ by +-3 %, and by more now and again): interleaved
# repetitions of bench.py with NLAM_LIB pointing at each.
#   gpurun -- 'CFG=cfg2 STEPS=300 PREC=fp32 REPS=3 bash tools/ab_libs.sh neural_lam_amd/libnlam_hip_prev.so neural_lam_amd/libnlam_hip.so [...]'
CFG=${CFG:-cfg2}; STEPS=${STEPS:-300}; PREC=${PREC:-fp32}; REPS=${REPS:-3}
run() { lib=$1; NLAM_LIB=$lib python bench.py --config $CFG --precision $PREC --steps $STEPS --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$(basename $1)', '$CFG', round(d['ms_per_step'],4), 'ms/step', round(d['forecast_steps_per_s'],1), 'forecast steps/s', 'loss', d['final_loss'])"; }
for rep in $(seq $REPS); do for lib in "$@"; do run $lib; done; done
