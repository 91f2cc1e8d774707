./tools/probes/bf16_block_probe
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -k "bf16_storage or over_bf16_rows" > gpurun_out/t8.log 2>&1; grep -n "passed\|failed\|Error\|assert " gpurun_out/t8.log | head -30
timeout 900 python -m pytest tests/test_full_size_parity.py -m gpu -x -q -s -k "cfg5" > gpurun_out/t7.log 2>&1; grep "cfg5 (T\|passed\|failed\|Error" gpurun_out/t7.log
for sb in 0 1; do NLAM_STORE_BF16=$sb NLAM_KB_AUTOCAST=1 python tools/kernel_bench.py m2m 12 512 edge 2>&1 | grep -v amdgpu | sed "s/^/[store_bf16=$sb] /"; done
for rep in 1 2; do for sb in 0 1; do NLAM_STORE_BF16=$sb python bench.py --config cfg5 --precision bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('store_bf16=$sb cfg5', round(d['ms_per_step'],3), 'ms/step', round(d['forecast_steps_per_s'],1), 'forecast steps/s', 'loss', d['final_loss'])"; done; done
for sb in 0 1; do NLAM_STORE_BF16=$sb python bench.py --config cfg3 --precision bf16 --steps 12 --warmup 2 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('store_bf16=$sb cfg3-bf16', round(d['ms_per_step'],3), 'ms/step', round(d['forecast_steps_per_s'],1), 'forecast steps/s', 'loss', d['final_loss'])"; done
