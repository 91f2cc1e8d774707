import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from neural_lam_amd import gnn_layers as hl, graph as G, ops  # noqa: E402
dev = torch.device("cuda:0")
raw = G.create_regular_grid_graph(G.regular_grid_xy(238, 268))
ei = raw["m2g_edge_index"]
ns, nr, E = int(ei[0].max()) + 1, int(ei[1].max()) + 1, ei.shape[1]
torch.manual_seed(0)
net = hl.InteractionNet(ei, 64, update_edges=False).to(dev)
send = torch.randn(1, ns, 64, device=dev, requires_grad=True)
rec = torch.randn(1, nr, 64, device=dev, requires_grad=True)
edge = torch.randn(1, E, 64, device=dev, requires_grad=True)
cot = torch.randn(1, nr, 64, device=dev)
for mode in ("bf16x2", "bf16x3", "bf16"):
    ops.set_matmul_mode(mode)
    aggr, _ = net._messages_and_aggregate(send, rec, edge, False, True)
    ref = None
    nbad = 0
    for it in range(12):
        gs, gr, ge = torch.autograd.grad((aggr * cot).sum(), (send, rec, edge), retain_graph=True)
        cur = (gs.clone(), gr.clone(), ge.clone())
        if ref is None:
            ref = cur
        elif not all(torch.equal(a, b) for a, b in zip(cur, ref)):
            nbad += 1
            d = [float((a - b).abs().max()) for a, b in zip(cur, ref)]
            nz = [int((a != b).sum()) for a, b in zip(cur, ref)]
            print(f"   it {it}: max diffs {d} differing elements {nz}")
    print(f"{mode}: same forward, 12 backward runs: {nbad} differ from the first")
