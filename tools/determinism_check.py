"""Run one MEPS-size layer several times per matmul mode; every output / gradient must be bit-identical."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from neural_lam_amd import gnn_layers as hl, graph as G, ops  # noqa: E402
dev = torch.device("cuda:0")
raw = G.create_regular_grid_graph(G.regular_grid_xy(238, 268))
for which in ("m2g", "m2m", "g2m"):
    ei = raw[f"{which}_edge_index"] if which != "m2m" else raw["m2m_edge_index"][0]
    ns, nr, E = int(ei[0].max()) + 1, int(ei[1].max()) + 1, ei.shape[1]
    torch.manual_seed(0)
    net = hl.InteractionNet(ei, 64, update_edges=(which == "m2m")).to(dev)
    send = torch.randn(1, ns, 64, device=dev, requires_grad=True)
    rec = torch.randn(1, nr, 64, device=dev, requires_grad=True)
    edge = torch.randn(1, E, 64, device=dev, requires_grad=True)
    for mode in ("f32", "bf16x2", "bf16x3"):
        ops.set_matmul_mode(mode)
        runs = []
        for it in range(16):
            for t in (send, rec, edge):
                t.grad = None
            net.zero_grad(set_to_none=True)
            out = net(send, rec, edge)
            outs = out if isinstance(out, tuple) else (out,)
            sum(o.square().sum() for o in outs).backward()
            cur = {f"out{k}": o.detach().clone() for k, o in enumerate(outs)}
            cur.update(grad_send=send.grad.clone(), grad_rec=rec.grad.clone(), grad_edge=edge.grad.clone())
            cur.update({f"grad {k}": p.grad.clone() for k, p in net.named_parameters()})
            runs.append(cur)
        bad = sorted({k for it in range(1, 16) for k in runs[0] if not torch.equal(runs[0][k], runs[it][k])})
        print(f"{which} {mode}: {'deterministic' if not bad else 'NON-DETERMINISTIC: ' + ', '.join(bad)}")
