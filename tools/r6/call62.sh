#!/bin/bash
# round 6, call 62: the co-run sizing of the wide weight gradients re-measured on the final build (4-wave reduce_jobs): most workgroups per launch
# (NLAM_WGRAD_MAX_WGS, default 128) and side streams (NLAM_WGRAD_STREAMS, default 4)
B="--no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also"
run() {  # tag, config args, env...
  tag=$1; shift; cfg=$1; shift
  env "$@" timeout 600 python bench.py $cfg $B > /tmp/x.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1]); print("$tag", round(d["ms_per_step"],3))
except Exception as e: print("$tag", "FAILED", e)
PY
}
C5="--config cfg5 --precision bf16 --steps 4 --warmup 2"
C3="--config cfg3 --steps 12 --warmup 2"
for rep in 1 2; do
run c5_base_$rep "$C5" X=1
run c5_wgs96_$rep "$C5" NLAM_WGRAD_MAX_WGS=96
run c5_wgs160_$rep "$C5" NLAM_WGRAD_MAX_WGS=160
run c5_wgs192_$rep "$C5" NLAM_WGRAD_MAX_WGS=192
run c3_base_$rep "$C3" X=1
run c3_wgs96_$rep "$C3" NLAM_WGRAD_MAX_WGS=96
run c3_wgs160_$rep "$C3" NLAM_WGRAD_MAX_WGS=160
run c3_wgs192_$rep "$C3" NLAM_WGRAD_MAX_WGS=192
done
