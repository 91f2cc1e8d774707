#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/round5
mkdir -p $OUT profiles/round5
python bench.py --config cfg5 --precision bf16 --steps 4 --warmup 2 --no-data-path > $OUT/bench_cfg5_bf16.json 2>$OUT/bench_cfg5.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_cfg2_driver_cmdline.json 2>/dev/null
python - <<PY
import json
for c in ('cfg2_driver_cmdline','cfg5_bf16'):
    d = json.load(open('$OUT/bench_%s.json' % c)); g = d.get('gpu_reference_equivalent') or {}; l = d.get('lightning_shaped') or {}
    print(c, round(d['ms_per_step'],3), 'ms/step', 'x%.2f / x%.2f' % (g.get('speedup_vs_nondeterministic') or 0, g.get('speedup_vs_deterministic') or 0), g.get('ms_per_step_nondeterministic'), g.get('ms_per_step_deterministic'), 'cpu', (d.get('cpu_baseline') or {}).get('ms_per_step'),
          'drop-in', l.get('ms_per_step_eager_torch_adamw'), l.get('ms_per_step_graphed_torch_adamw'), l.get('ms_per_step_graphed_fused_adamw_torch_adamw'), 'traffic', (d.get('roofline') or {}).get('traffic'))
PY
bash tools/r5/call10.sh
