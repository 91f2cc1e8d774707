"""GPU check: a HIP-graph training step (Trainer(use_graph=True)) must reproduce the eager
step bit for bit (same kernels, same order) over several optimizer steps."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from neural_lam_amd import graph as G  # noqa: E402
from neural_lam_amd import models as hm  # noqa: E402
from neural_lam_amd.datastore import SyntheticDatastore  # noqa: E402
from neural_lam_amd.trainer import Trainer  # noqa: E402

dev = torch.device("cuda:0")


def make(use_graph):
    ds = SyntheticDatastore(30, 27, 5, 2, 1, root_path="/tmp/nlam_gc", boundary="random", seed=1)
    ext = ds.get_xy_extent("state")
    graph = G.normalise_graph(G.create_regular_grid_graph(ds.get_xy("state")), max(ext[1] - ext[0], ext[3] - ext[2]))
    torch.manual_seed(1)
    fc = hm.ARForecaster(hm.GraphLAM(ds, graph=graph, hidden_dim=16, processor_layers=2), ds)
    step = hm.ForecasterStep(fc, ds).to(dev)
    return ds, Trainer(step, lr=1e-3, use_graph=use_graph)


ds, t_eager = make(False)
_, t_graph = make(True)
N = ds.num_grid_points
g = torch.Generator().manual_seed(0)
for it in range(4):
    batch = [torch.randn(1, 2, N, 5, generator=g).to(dev), torch.randn(1, 2, N, 5, generator=g).to(dev),
             torch.randn(1, 2, N, 6, generator=g).to(dev)]
    le, lg = float(t_eager.step(*batch)), float(t_graph.step(*batch))
    dp = float((t_eager.fp.flat - t_graph.fp.flat).abs().max())
    dg = float((t_eager.fp.grad - t_graph.fp.grad).abs().max())
    print(f"step {it}: loss eager {le:.7f} graph {lg:.7f}   max|dparam| {dp:.3e}   max|dgrad| {dg:.3e}")
    assert le == le and abs(le - lg) <= 1e-6 * abs(le) and dp <= 1e-6, "graph step diverges from the eager step"
print("graph step == eager step")
