#!/bin/bash
# Round 6, call 1: the N=1 bench line with the new `also` legs (wall time of the whole command), CU-mask stream probe, per-phase
# wave cycles of the wide split-bf16 kernels on the current build (what the restructuring has to remove), isolated wide launches.
mkdir -p gpurun_out/r6
LOG=gpurun_out/r6
python -c "import __graft_entry__ as g; g.smoke()" > $LOG/smoke.log 2>&1; tail -1 $LOG/smoke.log
t0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > $LOG/bench_driver_cmdline.json 2> $LOG/bench_driver_cmdline.err
echo "bench.py driver command line: $(( $(date +%s) - t0 )) s wall"
python - <<PY
import json
d = json.loads(open('$LOG/bench_driver_cmdline.json').read().strip().splitlines()[-1])
print('cfg2', round(d['ms_per_step'], 4), 'ms/step; roofline frac', round(d['roofline']['frac'], 3), 'bytes_min', d['roofline'].get('algorithmic_bytes_min'))
for k, v in (d.get('also') or {}).items():
    g = v.get('gpu_reference_equivalent') or {}
    print(k, round(v['ms_per_step'], 3), 'ms/step', v['executor'], 'ref', g.get('ms_per_step_nondeterministic'), g.get('ms_per_step_deterministic'), 'x', g.get('speedup_vs_nondeterministic'), g.get('speedup_vs_deterministic'), 'leg wall', round(v['leg_wall_s'], 1))
PY
timeout 300 python tools/r6/cumask_probe.py > $LOG/cumask_probe.log 2>&1; cat $LOG/cumask_probe.log | grep -v amdgpu.ids
export NLAM_LIB=neural_lam_amd/libnlam_hip_timing.so
for a in "m2m 512 bf16" "m2m 256" "m2g 512 bf16" "m2m 128"; do
  echo "=== $a"; timeout 300 python tools/phase_timing_wbf.py $a 2>&1 | grep -v amdgpu.ids
done > $LOG/phase_cycles_wide.txt
unset NLAM_LIB
grep -A14 "training\|edge bwd" $LOG/phase_cycles_wide.txt | head -150
NLAM_KB_AUTOCAST=1 python tools/kernel_bench.py m2m 12 512 2>&1 | grep -v amdgpu.ids | sed 's/^/[autocast bf16] /' > $LOG/kernel_bench_wide.log
python tools/kernel_bench.py m2m 12 256 2>&1 | grep -v amdgpu.ids >> $LOG/kernel_bench_wide.log
cat $LOG/kernel_bench_wide.log
