"""Early timing of the MEPS-shaped GraphLAM step (eager launches, no graph capture)."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from neural_lam_amd import graph as G  # noqa: E402
from neural_lam_amd import models as hm  # noqa: E402
from neural_lam_amd.datastore import meps_like_datastore  # noqa: E402

dev = torch.device("cuda:0")
d = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = sys.argv[2] if len(sys.argv) > 2 else "graph_lam"   # graph_lam | hi_lam | hi_lam_parallel
L = int(sys.argv[3]) if len(sys.argv) > 3 else 4
T = int(sys.argv[4]) if len(sys.argv) > 4 else 1
ds = meps_like_datastore("/tmp/nlam_qb")
ext = ds.get_xy_extent("state")
t0 = time.time()
hier = model != "graph_lam"
graph = G.normalise_graph(G.create_regular_grid_graph(ds.get_xy("state"), n_max_levels=3 if hier else None, hierarchical=hier),
                          max(ext[1] - ext[0], ext[3] - ext[2]))
print(f"graph built in {time.time() - t0:.2f}s")
torch.manual_seed(42)
fc = hm.ARForecaster(hm.MODELS[model](ds, graph=graph, hidden_dim=d, processor_layers=L), ds)
print(f"model {model} d={d} L={L} T={T} params={sum(p.numel() for p in fc.parameters())}")
step = hm.ForecasterStep(fc, ds).to(dev)
N = ds.num_grid_points
torch.manual_seed(123)
init = torch.randn(1, 2, N, 17, device=dev)
target = torch.randn(1, T, N, 17, device=dev)
forcing = torch.randn(1, T, N, 18, device=dev)
params = [p for p in step.parameters()]
opt = torch.optim.AdamW(params, lr=1e-3, betas=(0.9, 0.95))


def train_step():
    opt.zero_grad(set_to_none=True)
    _, loss = step(init, target, forcing)
    loss.backward()
    opt.step()
    return loss


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, (time.time() - t0) / n * 1e3


with torch.no_grad():
    gpu_ms, wall_ms = timeit(lambda: step(init, target, forcing))
print(f"forward (no grad): {gpu_ms:.3f} ms gpu, {wall_ms:.3f} ms wall")
gpu_ms, wall_ms = timeit(train_step)
print(f"train step (fwd+loss+bwd+torch AdamW): {gpu_ms:.3f} ms gpu, {wall_ms:.3f} ms wall")
print("loss", float(train_step()))
