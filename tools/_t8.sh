python -m pytest tests/test_hip_parity.py tests/test_full_size_parity.py tests/test_latent.py -m gpu -x -q -k "model or cfg2 or cfg4 or graph_step or rollout or latent or autocast or falls_back" 2>&1 | tail -4
for i in 1 2; do python bench.py --no-cpu-baseline --no-gpu-baseline --no-roofline --steps 200 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read()); print('cfg2', round(d['ms_per_step'],4), round(d['forecast_steps_per_s'],1), d['final_loss'])"; done
python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read()); print('cfg4', round(d['ms_per_step'],3), round(d['forecast_steps_per_s'],1))"
python bench.py --config cfg3 --steps 6 --warmup 2 --no-cpu-baseline --no-gpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read()); print('cfg3', round(d['ms_per_step'],3), round(d['forecast_steps_per_s'],1))"
