#!/bin/bash
# Regenerates the measurement artefacts of profiles/round6/ on the GPU box:  gpurun -- 'bash tools/profile_round6.sh'
# (writes under gpurun_out/round6/, which is then copied into profiles/round6/ and committed)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/round6
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
# PMC passes (counters only + kernel trace, one pass per counter group)
python tools/pmc_collect.py m2g_edge fetch write wave insts -- python tools/kernel_bench.py m2g 3 64 edge > /dev/null 2>&1
python tools/pmc_collect.py m2m_edge fetch write wave insts -- python tools/kernel_bench.py m2m 3 64 edge > /dev/null 2>&1
cp gpurun_out/pmc_m2g_edge.md $OUT/pmc_m2g_edge.md; cp gpurun_out/pmc_m2m_edge.md $OUT/pmc_m2m_edge.md
python tools/pmc_collect.py cfg2_step wave insts fetch write -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also > /dev/null 2>&1
cp gpurun_out/pmc_cfg2_step.md $OUT/pmc_cfg2_step.md
# the wide edge stages: d = 512 one term / bf16 storage (cfg5's kernels), d = 256 three terms (cfg3's)
NLAM_KB_AUTOCAST=1 python tools/pmc_collect.py wide_d512_bf16 fetch write wave insts mem -- python tools/kernel_bench.py m2m 4 512 edge > /dev/null 2>&1
python tools/pmc_collect.py wide_d256 fetch write wave insts mem -- python tools/kernel_bench.py m2m 4 256 edge > /dev/null 2>&1
cp gpurun_out/pmc_wide_d512_bf16.md $OUT/pmc_wide_d512_bf16.md; cp gpurun_out/pmc_wide_d256.md $OUT/pmc_wide_d256.md
python tools/make_pmc_traffic.py gpurun_out/pmc_m2g_edge.json:255136 gpurun_out/pmc_m2m_edge.json:57616 gpurun_out/pmc_cfg2_step.json:step gpurun_out/pmc_wide_d512_bf16.json:57616:512 gpurun_out/pmc_wide_d256.json:57616:256 > $OUT/pmc_traffic.json
mkdir -p profiles/round6 && cp $OUT/pmc_traffic.json profiles/round6/pmc_traffic.json   # (this copy of the tree: bench.py below reads roofline.traffic from it)
# the driver's command line first (with the cfg3 / cfg5 legs under "also"), then the 300-step default
t0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_cfg2_driver_cmdline.json 2>/dev/null
echo "driver command line: $(( $(date +%s) - t0 )) s wall" > $OUT/bench_driver_cmdline_wall.txt
python bench.py --no-also > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
python bench.py --config cfg3 --steps 12 --warmup 2 --no-data-path > $OUT/bench_cfg3.json 2>/dev/null
python bench.py --config cfg4 --steps 30 --warmup 3 --no-data-path > $OUT/bench_cfg4.json 2>/dev/null
python bench.py --config cfg4p --steps 30 --warmup 3 --no-data-path > $OUT/bench_cfg4p.json 2>/dev/null
python bench.py --config cfg5 --precision bf16 --steps 4 --warmup 2 --no-data-path > $OUT/bench_cfg5_bf16.json 2>/dev/null
python bench.py --config cfg3 --precision bf16 --steps 12 --warmup 2 --no-cpu-baseline --no-data-path --no-lightning-leg > $OUT/bench_cfg3_bf16.json 2>/dev/null
for w in m2g g2m m2m; do python tools/kernel_bench.py $w 12 64 2>&1 | grep -v amdgpu.ids; done > $OUT/kernel_bench_d64.log
{ NLAM_KB_AUTOCAST=1 python tools/kernel_bench.py m2m 12 512 2>&1 | grep -v amdgpu.ids | sed 's/^/[autocast bf16] /'; NLAM_KB_AUTOCAST=1 python tools/kernel_bench.py m2g 8 512 2>&1 | grep -v amdgpu.ids | sed 's/^/[autocast bf16] /'; python tools/kernel_bench.py m2m 12 256 2>&1 | grep -v amdgpu.ids; python tools/kernel_bench.py m2g 8 256 2>&1 | grep -v amdgpu.ids; } > $OUT/kernel_bench_wide.log
python tools/chain_only.py > $OUT/chain_only.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg --no-also > $OUT/bench_cfg2_under_rocprofv3.json 2>/dev/null   # graph replays only
cd $GRAFT_REPO_ROOT
python tools/step_timeline.py $(find $OUT/tr -name "*kernel_trace.csv" | head -1) > $OUT/cfg2_step_timeline.txt
cp $(find $OUT/tr -name "*kernel_stats.csv" | head -1) $OUT/bench_cfg2_kernel_stats.csv
rm -rf $OUT/tr
for c in cfg3 cfg5; do
  prec=fp32; steps=6; [ $c = cfg5 ] && prec=bf16 && steps=3
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_$c -o t -- python $GRAFT_REPO_ROOT/bench.py --config $c --precision $prec --steps $steps --warmup 2 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path --no-lightning-leg > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/step_timeline.py $(find $OUT/tr_$c -name "*kernel_trace.csv" | head -1) > $OUT/${c}_step_timeline_full.txt
  grep "^#" $OUT/${c}_step_timeline_full.txt > $OUT/${c}_step_kernel_totals.txt
  cp $(find $OUT/tr_$c -name "*kernel_stats.csv" | head -1) $OUT/bench_${c}_kernel_stats.csv
  rm -rf $OUT/tr_$c $OUT/${c}_step_timeline_full.txt
done
tail -1 $OUT/smoke.log
python - <<PY
import json
for c in ('cfg2','cfg3','cfg4','cfg4p','cfg5_bf16','cfg3_bf16'):
    d = json.load(open('$OUT/bench_%s.json' % c)); g = d.get('gpu_reference_equivalent') or {}; l = d.get('lightning_shaped') or {}
    print(c, round(d['ms_per_step'],3), 'ms/step', round(d['forecast_steps_per_s'],1), 'forecast steps/s', 'x%.2f / x%.2f vs same-GPU reference (eager / deterministic)' % (g.get('speedup_vs_nondeterministic') or 0, g.get('speedup_vs_deterministic') or 0), 'cpu', (d.get('cpu_baseline') or {}).get('ms_per_step'),
          'drop-in eager / graphed / graphed+fused AdamW', l.get('ms_per_step_eager_torch_adamw'), l.get('ms_per_step_graphed_torch_adamw'), l.get('ms_per_step_graphed_fused_adamw_torch_adamw'))
d = json.loads(open('$OUT/bench_cfg2_driver_cmdline.json').read().strip().splitlines()[-1])
print('driver line: cfg2', round(d['ms_per_step'],4), {k: (round(v['ms_per_step'],2), round((v.get('gpu_reference_equivalent') or {}).get('speedup_vs_nondeterministic') or 0, 2), round((v.get('gpu_reference_equivalent') or {}).get('speedup_vs_deterministic') or 0, 2)) for k, v in (d.get('also') or {}).items()})
r = d['roofline']; print({k: r[k] for k in ('bound','achieved','peak','unit','frac','traffic','traffic_source','kernel')})
for c in ('cfg2', 'cfg3', 'cfg5_bf16'):
    d = json.load(open('$OUT/bench_%s.json' % c))
    for k in d['roofline']['kernels']: print(c, k['launch'], round(k['avg_launch_ms']*1e3,1), 'us x', k['launches'], 'frac', round(k['frac'],3), k['bound'], 'mfma', round(k['mfma_frac'],3), 'hbm', round(k['hbm_frac'],3), 'incl', round(k['hbm_frac_incl_saved_and_partials'],3), 'traffic', k['traffic'])
PY
timeout 1800 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
