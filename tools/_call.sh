mkdir -p gpurun_out/round4
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-data-path > gpurun_out/round4/bench_cfg2_driver_cmdline.json 2>/dev/null
python bench.py > gpurun_out/round4/bench_cfg2.json 2> gpurun_out/round4/bench_cfg2.err
python - <<PY
import json
d = json.load(open('gpurun_out/round4/bench_cfg2.json'))
r = d['roofline']; print(d['ms_per_step'], {k: r[k] for k in ('bound','achieved','frac','traffic','traffic_source')})
for k in r['kernels']: print(k['launch'], round(k['avg_launch_ms']*1e3,1), 'us x', k['launches'], 'frac', round(k['frac'],3), k['bound'], 'alg MB', round(k['algorithmic_bytes']/1e6,1), 'traffic', k['traffic'])
PY
