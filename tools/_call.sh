timeout 1500 python -m pytest tests/test_hip_parity.py -x -q 2>&1 | tail -3
for c in cfg2 cfg4p; do echo "[$c] $(python bench.py --config $c --steps 200 --warmup 10 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')"; done
