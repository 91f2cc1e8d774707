mkdir -p gpurun_out/r2j
python -m pytest tests/test_hip_parity.py tests/test_full_size_parity.py -m gpu -x -q -k "node_linear or layer_matches or meps_layer or wide_layer or awkward or cfg4 or d128 or autocast or same_tensor or families_agree or graph_step" 2>&1 | tail -12 > gpurun_out/r2j/pytest.log
cat gpurun_out/r2j/pytest.log
for w in m2g m2m; do echo "== $w d=256 factorised"; python tools/kernel_bench.py $w 8 256 2>&1 | grep -E "linear|mlp_fwd', (255136|57616)|mlp_bwd', (255136|57616)"; done
for c in cfg3 cfg4; do for fw in 0 1073741824; do NLAM_FACTORISE_MIN_EDGES_WIDE=$fw python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline --no-gpu-baseline 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read()); print('$c factorise_min_wide=$fw', round(d['ms_per_step'],3), round(d['forecast_steps_per_s'],1), d['final_loss'], [k['launch'] for k in d['roofline']['kernels'] if 'linear' in k['launch']])"; done; done
