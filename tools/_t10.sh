mkdir -p gpurun_out
python tools/pmc_collect.py m2g_edge fetch write wave insts -- python tools/kernel_bench.py m2g 3 64 edge > /dev/null 2>&1
python tools/pmc_collect.py m2m_edge fetch write wave insts -- python tools/kernel_bench.py m2m 3 64 edge > /dev/null 2>&1
python tools/make_pmc_traffic.py gpurun_out/pmc_m2g_edge.json:255136 gpurun_out/pmc_m2m_edge.json:57616 > gpurun_out/pmc_traffic.json
rm -rf gpurun_out/pmc_m2g_edge gpurun_out/pmc_m2m_edge
python -c "
import json; d=json.load(open('gpurun_out/pmc_traffic.json')); print(d['bytes_per_launch'])"
