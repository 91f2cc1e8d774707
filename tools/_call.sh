timeout 1500 python -m pytest tests/test_hip_parity.py -x -q 2>&1 | tail -6
for c in cfg4p cfg4; do python bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path 2>&1 | tail -1 | cut -c1-330; done
