#!/bin/bash
# A/B of two builds of the library on isolated wide launches (kernel_bench) and captured steps, inside one gpurun call.
#   gpurun -- 'bash tools/ab_kernels.sh neural_lam_amd/libnlam_hip_prev.so neural_lam_amd/libnlam_hip.so'
OUT=${OUT:-gpurun_out/abk}; mkdir -p $OUT
kb() { NLAM_LIB=$1 NLAM_KB_AUTOCAST=$4 python tools/kernel_bench.py $2 12 $3 edge 2>&1 | grep -v amdgpu.ids | sed "s/^/[$(basename $1) ac=$4] /"; }
{ for lib in "$@"; do kb $lib m2m 256 0; kb $lib m2g 256 0; kb $lib m2m 256 1; kb $lib m2m 512 1; kb $lib m2g 512 1; kb $lib m2m 128 0; done; } > $OUT/kernel_bench.log 2>&1
CFG=cfg3 STEPS=12 PREC=fp32 REPS=2 bash tools/ab_libs.sh "$@" > $OUT/steps.log 2>&1
CFG=cfg5 STEPS=4 PREC=bf16 REPS=2 bash tools/ab_libs.sh "$@" >> $OUT/steps.log 2>&1
CFG=cfg3 STEPS=12 PREC=bf16 REPS=1 bash tools/ab_libs.sh "$@" >> $OUT/steps.log 2>&1
CFG=cfg4 STEPS=30 PREC=fp32 REPS=1 bash tools/ab_libs.sh "$@" >> $OUT/steps.log 2>&1
