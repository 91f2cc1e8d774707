mkdir -p gpurun_out/round4
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-data-path > gpurun_out/round4/bench_cfg2_driver_cmdline.json 2>/dev/null
python bench.py > gpurun_out/round4/bench_cfg2.json 2> gpurun_out/round4/bench_cfg2.err
python - <<PY
import json
d = json.load(open('gpurun_out/round4/bench_cfg2.json'))
r = d['roofline']; print(d['ms_per_step'], {k: r[k] for k in ('bound','achieved','peak','unit','frac','traffic','traffic_source')})
PY
