"""Long-run sanity: N optimizer steps of cfg2 in HIP-graph mode under two matrix modes; losses must stay finite,
decrease on the fixed synthetic batch, and the two trajectories must track each other."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from neural_lam_amd import ops  # noqa: E402
from neural_lam_amd.trainer import Trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfgname = sys.argv[2] if len(sys.argv) > 2 else "cfg2"   # any bench.CONFIGS key (cfg3 / cfg4 exercise the wide kernels)
dev = torch.device("cuda:0")
traj = {}
for mode in ("bf16x3", "f32"):
    ops.set_matmul_mode(mode)
    ds, graph, raw, forecaster, step, batch = bench.build(bench.CONFIGS[cfgname], dev)
    tr = Trainer(step, lr=1e-3, use_graph=True)
    losses = []
    for i in range(steps):
        losses.append(tr.step(*batch).clone())
    losses = torch.stack(losses).cpu()
    traj[mode] = losses
    print(f"{mode}: loss[0]={float(losses[0]):.6f} loss[{steps // 2}]={float(losses[steps // 2]):.6f} loss[-1]={float(losses[-1]):.6f} "
          f"finite={bool(torch.isfinite(losses).all())}")
d = (traj["bf16x3"] - traj["f32"]).abs() / traj["f32"].abs()
print(f"max relative difference between the two loss trajectories: {float(d.max()):.3e} (at step {int(d.argmax())})")
