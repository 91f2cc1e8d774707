for m in bf16x3 bf16x2 bf16 f32; do NLAM_MATMUL=$m python bench.py --no-cpu-baseline --no-gpu-baseline --no-roofline --steps 200 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read()); print('$m', round(d['ms_per_step'],4), round(d['forecast_steps_per_s'],1))"; done
python tools/chain_only.py 2>&1 | tail -1
NLAM_MATMUL=bf16 python tools/chain_only.py 2>&1 | tail -1
