timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "split_receivers" > gpurun_out/t2.log 2>&1; tail -15 gpurun_out/t2.log
timeout 1500 python -m pytest tests/test_full_size_parity.py -m gpu -x -q -s -k "cfg5_full or cfg2_training or cfg3" > gpurun_out/t3.log 2>&1; tail -15 gpurun_out/t3.log
