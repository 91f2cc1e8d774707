"""Lower bound of the cfg2 step without weight-gradient work: only the first-layer biases of the input-side MLPs require a
gradient, so the whole data-gradient chain of backward runs but no dW kernels / partial-sum reductions do.  The
difference to the full step is what the side-stream weight-gradient work costs the critical chain (contention)."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from neural_lam_amd.trainer import Trainer  # noqa: E402

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
dev = torch.device("cuda:0")
ds, graph, raw, fc, step, batch = bench.build(cfg, dev)
keep = ("grid_embedder.0.bias", "g2m_embedder.0.bias", "m2g_embedder.0.bias", "m2m_embedder.0.bias", "mesh_embedder.0.bias")
for name, p in step.named_parameters():
    p.requires_grad_(any(name.endswith(k) for k in keep))
print("trainable:", [n for n, p in step.named_parameters() if p.requires_grad])
tr = Trainer(step, lr=1e-3, use_graph=True)
for _ in range(10):
    tr.step(*batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 200
for _ in range(N):
    tr.step(*batch)
torch.cuda.synchronize()
print("chain-only ms/step:", (time.perf_counter() - t0) / N * 1e3)
