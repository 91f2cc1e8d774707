timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "wide or autocast or plain_mlp or golden or prepacked" > gpurun_out/t4.log 2>&1; tail -6 gpurun_out/t4.log
timeout 600 python -m pytest tests/test_full_size_parity.py -m gpu -x -q -k "layer" > gpurun_out/t5.log 2>&1; tail -4 gpurun_out/t5.log
timeout 1200 bash tools/ab_kernels.sh neural_lam_amd/libnlam_hip_prev.so neural_lam_amd/libnlam_hip.so
grep "mlp_fwd\|mlp_bwd" gpurun_out/abk/kernel_bench.log
cat gpurun_out/abk/steps.log
