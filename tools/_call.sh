timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "chunk or layer_matches or hilam_parallel or graph_step" 2>&1 | tail -15
for g in 0 1; do NLAM_GROUP_CHUNKS=$g python bench.py --config cfg4p --steps 100 --warmup 10 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path 2>&1 | tail -1 | cut -c1-400; done
