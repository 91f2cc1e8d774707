timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "embedder or plain_mlp or graph_step_equals or golden or prepacked or trainer" > gpurun_out/t13.log 2>&1; tail -3 gpurun_out/t13.log
timeout 900 python -m pytest tests/test_full_size_parity.py -m gpu -x -q -s -k "cfg4 or cfg3 or cfg5_full" > gpurun_out/t14.log 2>&1; grep "cfg5 (T\|passed\|failed" gpurun_out/t14.log
step() { python bench.py --config $1 --precision $2 --steps $3 --warmup 3 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', '$2', round(d['ms_per_step'],3), 'ms/step', round(d['forecast_steps_per_s'],1), 'forecast/s', 'loss', d['final_loss'])"; }
step cfg5 bf16 4; step cfg3 fp32 12; step cfg3 bf16 12; step cfg4 fp32 30; step cfg4p fp32 30
