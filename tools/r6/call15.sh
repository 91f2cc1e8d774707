#!/bin/bash
# round 6, call 15: executor knobs re-measured on the round-6 kernels (side streams, fork points per segment)
mkdir -p gpurun_out/r6c15
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
run() {  # tag, config args, env...
  tag=$1; shift; cfg=$1; shift
  env "$@" timeout 600 python bench.py $cfg $B > gpurun_out/r6c15/$tag.json 2> gpurun_out/r6c15/$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r6c15/$tag.json").read().strip().splitlines()[-1]); print("$tag", round(d["ms_per_step"],3))
except Exception as e: print("$tag", "FAILED", e)
PY
}
C5="--config cfg5 --precision bf16 --steps 5 --warmup 2"
C3="--config cfg3 --steps 12 --warmup 2"
for rep in 1 2; do
run c5_base_$rep "$C5" X=1
run c5_streams3_$rep "$C5" NLAM_WGRAD_STREAMS=3
run c5_streams2_$rep "$C5" NLAM_WGRAD_STREAMS=2
run c5_forks6_$rep "$C5" NLAM_SEG_FORKS=6
run c5_forks24_$rep "$C5" NLAM_SEG_FORKS=24
run c5_forks48_$rep "$C5" NLAM_SEG_FORKS=48
done
run c3_base "$C3" X=1
run c3_streams3 "$C3" NLAM_WGRAD_STREAMS=3
run c3_streams2 "$C3" NLAM_WGRAD_STREAMS=2
run c3_forks6 "$C3" NLAM_SEG_FORKS=6
run c3_forks24 "$C3" NLAM_SEG_FORKS=24
run c3_forks48 "$C3" NLAM_SEG_FORKS=48
run c3_base2 "$C3" X=1
