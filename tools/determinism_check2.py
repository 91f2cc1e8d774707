import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from neural_lam_amd import gnn_layers as hl, graph as G, ops  # noqa: E402
dev = torch.device("cuda:0")
raw = G.create_regular_grid_graph(G.regular_grid_xy(238, 268))
which = "m2g"
ei = raw[f"{which}_edge_index"]
ns, nr, E = int(ei[0].max()) + 1, int(ei[1].max()) + 1, ei.shape[1]
torch.manual_seed(0)
net = hl.InteractionNet(ei, 64, update_edges=False).to(dev)
send = torch.randn(1, ns, 64, device=dev, requires_grad=True)
rec = torch.randn(1, nr, 64, device=dev, requires_grad=True)
edge = torch.randn(1, E, 64, device=dev, requires_grad=True)
aggr_in = torch.randn(1, nr, 64, device=dev, requires_grad=True)
cot = torch.randn(1, nr, 64, device=dev)
for mode in ("f32", "bf16", "bf16x2", "bf16x3"):
    ops.set_matmul_mode(mode)
    for stage in ("edge", "node"):
        runs = []
        for it in range(4):
            for t in (send, rec, edge, aggr_in):
                t.grad = None
            net.zero_grad(set_to_none=True)
            if stage == "edge":
                aggr, _ = net._messages_and_aggregate(send, rec, edge, False, True)
                (aggr * cot).sum().backward()
                cur = dict(aggr=aggr.detach().clone(), gs=send.grad.clone(), gr=rec.grad.clone(), ge=edge.grad.clone())
            else:
                out = net._node_update(rec, aggr_in)
                (out * cot).sum().backward()
                cur = dict(out=out.detach().clone(), gr=rec.grad.clone(), ga=aggr_in.grad.clone())
            cur.update({f"grad {k}": p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
            runs.append(cur)
        bad = sorted({k for it in range(1, 4) for k in runs[0] if not torch.equal(runs[0][k], runs[it][k])})
        print(f"{which} {mode} {stage}: {'deterministic' if not bad else 'NON-DETERMINISTIC: ' + ', '.join(bad)}")
