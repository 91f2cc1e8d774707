mkdir -p gpurun_out/r2i
for c in cfg3 cfg4 cfg4p; do python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-baseline --no-roofline 2>/dev/null > gpurun_out/r2i/$c.json; python -c "
import json; d=json.load(open('gpurun_out/r2i/$c.json')); print('$c', round(d['ms_per_step'],3), round(d['forecast_steps_per_s'],1))"; done
python bench.py --config cfg5 --precision bf16 --steps 4 --warmup 1 --no-cpu-baseline --no-gpu-baseline --no-roofline 2>/dev/null > gpurun_out/r2i/cfg5.json; python -c "
import json; d=json.load(open('gpurun_out/r2i/cfg5.json')); print('cfg5', round(d['ms_per_step'],3), round(d['forecast_steps_per_s'],1))"
NLAM_EARLY_LEAF=1 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read()); print('cfg4 early-leaf', round(d['ms_per_step'],3))"
NLAM_EARLY_LEAF=1 python bench.py --config cfg3 --steps 6 --warmup 2 --no-cpu-baseline --no-gpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read()); print('cfg3 early-leaf', round(d['ms_per_step'],3))"
