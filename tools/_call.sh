OUT=$GRAFT_REPO_ROOT/gpurun_out/round4
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python tools/pmc_collect.py m2g_edge fetch write wave insts -- python tools/kernel_bench.py m2g 3 64 edge > /dev/null 2>&1
python tools/pmc_collect.py m2m_edge fetch write wave insts -- python tools/kernel_bench.py m2m 3 64 edge > /dev/null 2>&1
python tools/pmc_collect.py cfg2_step wave insts fetch write -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path > /dev/null 2>&1
python tools/make_pmc_traffic.py gpurun_out/pmc_m2g_edge.json:255136 gpurun_out/pmc_m2m_edge.json:57616 gpurun_out/pmc_cfg2_step.json:step > $OUT/pmc_traffic.json
cp gpurun_out/pmc_m2g_edge.md gpurun_out/pmc_m2m_edge.md gpurun_out/pmc_cfg2_step.md $OUT/
mkdir -p profiles/round4 && cp $OUT/pmc_traffic.json profiles/round4/pmc_traffic.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-data-path > $OUT/bench_cfg2_driver_cmdline.json 2>/dev/null
python bench.py > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
python bench.py --config cfg4p --steps 30 --warmup 3 --no-data-path > $OUT/bench_cfg4p.json 2>/dev/null
python bench.py --config cfg3 --steps 12 --warmup 2 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path > $OUT/bench_cfg3_quick.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_cfg4p -o t -- python $GRAFT_REPO_ROOT/bench.py --config cfg4p --steps 20 --warmup 3 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/queue_timeline.py $(find $OUT/tr_cfg4p -name "*kernel_trace.csv" | head -1) $OUT/cfg4p_step_queue_timeline.txt; rm -rf $OUT/tr_cfg4p
python - <<PY
import json
for c in ('cfg2','cfg4p','cfg3_quick','cfg2_driver_cmdline'):
    d = json.load(open('$OUT/bench_%s.json' % c)); g = d.get('gpu_reference_equivalent') or {}
    print(c, round(d['ms_per_step'],3), 'x%.2f / x%.2f' % (g.get('speedup_vs_nondeterministic') or 0, g.get('speedup_vs_deterministic') or 0))
d = json.load(open('$OUT/bench_cfg2.json')); r = d['roofline']
print({k: r[k] for k in ('frac','traffic','traffic_source')})
for k in r['kernels']: print(k['launch'], round(k['avg_launch_ms']*1e3,1), round(k['frac'],3), k['traffic'])
PY
