b() { tag=$1; cfg=$2; steps=$3; shift 3; env "$@" python bench.py --config $cfg --steps $steps --warmup 3 --no-cpu-baseline --no-gpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['forecast_steps_per_s'],1))"; }
for ch in 8 16 32; do b cfg2_chunks$ch cfg2 200 NLAM_WGRAD_CHUNKS=$ch; done
for ch in 8 16 32; do b cfg4_chunks$ch cfg4 10 NLAM_WGRAD_CHUNKS=$ch NLAM_FACTORISE_MIN_EDGES_WIDE=1073741824; done
for ch in 8 16; do b cfg3_chunks$ch cfg3 6 NLAM_WGRAD_CHUNKS=$ch; done
b cfg4_fact_again cfg4 10 A=1
b cfg4_plain_again cfg4 10 NLAM_FACTORISE_MIN_EDGES_WIDE=1073741824
