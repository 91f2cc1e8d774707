export NLAM_WBF_HALF=3
timeout 600 python -m pytest tests/test_full_size_parity.py -m gpu -x -q -k "layer" > gpurun_out/half_parity.log 2>&1; tail -5 gpurun_out/half_parity.log
unset NLAM_WBF_HALF
timeout 900 bash tools/ab_half.sh
cat gpurun_out/half/kernel_bench.log | grep -v "^\[.*\] *$" | head -80
cat gpurun_out/half/steps.log
