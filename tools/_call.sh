mkdir -p gpurun_out/r4
python bench.py --config cfg5 --precision bf16 --steps 4 --warmup 2 --no-data-path > gpurun_out/r4/bench_cfg5_bf16.json 2> gpurun_out/r4/bench_cfg5.err
python bench.py --config cfg3 --steps 12 --warmup 2 --no-data-path > gpurun_out/r4/bench_cfg3.json 2> gpurun_out/r4/bench_cfg3.err
python bench.py --config cfg4 --steps 30 --warmup 3 --no-data-path > gpurun_out/r4/bench_cfg4.json 2> gpurun_out/r4/bench_cfg4.err
tail -2 gpurun_out/r4/*.err
python - <<'PY'
import json
for f in ["bench_cfg5_bf16","bench_cfg3","bench_cfg4"]:
    try:
        d=json.load(open("gpurun_out/r4/"+f+".json"))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, round(d["ms_per_step"],2), "ms", "cpu", d.get("cpu_baseline",{}).get("ms_per_step"), "gpu_ref", {k:v for k,v in d.get("gpu_reference_equivalent",{}).items() if "ms_per" in k or "speedup" in k})
    for k in d["roofline"]["kernels"]:
        print("   %-40s n=%3d avg %.3f ms tot %.2f share %.3f mfma %.3f hbm %.3f"%(k["launch"],k["launches"],k["avg_launch_ms"],k["total_ms"],k["share_of_step"],k["mfma_frac"],k["hbm_frac"]))
PY
