./tools/probes/bf16_block_probe
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -k "bf16_storage or over_bf16_rows" > gpurun_out/t8.log 2>&1; grep -n "passed\|failed\|Error\|assert " gpurun_out/t8.log | head -30
