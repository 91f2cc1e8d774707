mkdir -p gpurun_out/r2h
python -m pytest tests/test_hip_parity.py tests/test_full_size_parity.py tests/test_latent.py -m gpu -x -q -k "model or cfg2 or cfg4 or graph_step or rollout or latent or standardize or wmse or boundary or clamped or falls_back" 2>&1 | tail -8 > gpurun_out/r2h/pytest.log
cat gpurun_out/r2h/pytest.log
for i in 1 2; do python bench.py --no-cpu-baseline --no-gpu-baseline --no-roofline --steps 200 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read()); print('cfg2', round(d['ms_per_step'],4), round(d['forecast_steps_per_s'],1), d['final_loss'])"; done
