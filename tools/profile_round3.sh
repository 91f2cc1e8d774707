#!/bin/bash
# Regenerates the measurement artefacts of profiles/round3/ on the GPU box:  gpurun -- 'bash tools/profile_round3.sh'
# (writes under gpurun_out/round3/, which is then copied into profiles/round3/ and committed)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/round3
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
python bench.py > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
for c in cfg3 cfg4 cfg4p; do python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-baseline --no-data-path > $OUT/bench_$c.json 2>/dev/null; done
python bench.py --config cfg5 --precision bf16 --steps 4 --warmup 1 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_cfg5_bf16.json 2>/dev/null
python bench.py --config cfg3 --precision bf16 --steps 8 --warmup 2 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_cfg3_bf16.json 2>/dev/null
for w in m2g g2m m2m; do python tools/kernel_bench.py $w 12 64 2>&1 | grep -v amdgpu.ids; done > $OUT/kernel_bench_d64.log
for w in m2g m2m; do for fw in 0 1073741824; do echo "== $w d=256 NLAM_FACTORISE_MIN_EDGES_WIDE=$fw"; NLAM_FACTORISE_MIN_EDGES_WIDE=$fw python tools/kernel_bench.py $w 8 256 2>&1 | grep -v amdgpu.ids; done; done > $OUT/kernel_bench_d256_factorised_vs_plain.log
python tools/chain_only.py > $OUT/chain_only.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-roofline > $OUT/bench_cfg2_under_rocprofv3.json 2>/dev/null   # graph replays only: the eager roofline pass would mix its launches into the trace
cd $GRAFT_REPO_ROOT
python tools/step_timeline.py $(find $OUT/tr -name "*kernel_trace.csv" | head -1) > $OUT/cfg2_step_timeline.txt
cp $(find $OUT/tr -name "*kernel_stats.csv" | head -1) $OUT/bench_cfg2_kernel_stats.csv
rm -rf $OUT/tr
for c in cfg3 cfg4; do
  cd /tmp
  rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$c -o t -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/step_timeline.py $(find $OUT/tr_$c -name "*kernel_trace.csv" | head -1) > $OUT/${c}_step_timeline_full.txt
  grep "^#" $OUT/${c}_step_timeline_full.txt > $OUT/${c}_step_kernel_totals.txt
  rm -rf $OUT/tr_$c $OUT/${c}_step_timeline_full.txt
done
python tools/hbm_kernels_bench.py 64 2>&1 | grep -v amdgpu.ids > $OUT/hbm_kernels_d64.log
# PMC passes (counters only + kernel trace, one pass per counter group): HBM traffic of the edge stage's kernels and of the replayed step
python tools/pmc_collect.py m2g_edge fetch write wave insts -- python tools/kernel_bench.py m2g 3 64 edge > /dev/null 2>&1
python tools/pmc_collect.py m2m_edge fetch write wave insts -- python tools/kernel_bench.py m2m 3 64 edge > /dev/null 2>&1
python tools/make_pmc_traffic.py gpurun_out/pmc_m2g_edge.json:255136 gpurun_out/pmc_m2m_edge.json:57616 > $OUT/pmc_traffic.json
cp gpurun_out/pmc_m2g_edge.md $OUT/pmc_m2g_edge.md; cp gpurun_out/pmc_m2m_edge.md $OUT/pmc_m2m_edge.md
python tools/pmc_collect.py cfg2_step wave insts fetch write -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path > /dev/null 2>&1
cp gpurun_out/pmc_cfg2_step.md $OUT/pmc_cfg2_step.md; cp gpurun_out/pmc_cfg2_step.json $OUT/pmc_cfg2_step.json
tail -2 $OUT/smoke.log
python -c "
import json
for c in ('cfg2','cfg3','cfg4','cfg4p','cfg5_bf16','cfg3_bf16'):
    d = json.load(open('$OUT/bench_%s.json' % c)); print(c, round(d['ms_per_step'],3), 'ms/step', round(d['forecast_steps_per_s'],1), 'forecast steps/s')
d = json.load(open('$OUT/bench_cfg2.json'))
print({k: d[k] for k in ('oracle_loss_step0','loss_step0','loss_step0_rel_diff_vs_oracle')})
print(d['cpu_baseline']['ms_per_step'], d['gpu_reference_equivalent'])
r = d['roofline']; print({k: r[k] for k in ('bound','achieved','peak','unit','frac','traffic','kernel')})
for k in r['kernels']: print(k['launch'], round(k['avg_launch_ms']*1e3,1), 'us x', k['launches'], 'frac', round(k['frac'],3), k['bound'], 'traffic', k['traffic'])
print(r['step'])
"
