"""Accuracy and speed of the matrix-path modes (f32 / bf16x3 / bf16x2 / bf16) on one MEPS-size layer:
outputs and gradients vs the f32 mode (max|a-b|/max|b|), and per-launch times."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from neural_lam_amd import gnn_layers as hl  # noqa: E402
from neural_lam_amd import graph as G  # noqa: E402
from neural_lam_amd import ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "m2m"
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
raw = G.create_regular_grid_graph(G.regular_grid_xy(238, 268))
ei = raw[f"{which}_edge_index"] if which != "m2m" else raw["m2m_edge_index"][0]
ns, nr, E = int(ei[0].max()) + 1, int(ei[1].max()) + 1, ei.shape[1]
torch.manual_seed(0)
net = hl.InteractionNet(ei, d, update_edges=(which == "m2m")).to(dev)
send = torch.randn(1, ns, d, device=dev, requires_grad=True)
rec = torch.randn(1, nr, d, device=dev, requires_grad=True)
edge = torch.randn(1, E, d, device=dev, requires_grad=True)


def rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


ref = None
print(f"{which}: E={E} d={d}")
for mode in ("f32", "bf16x3", "bf16x2", "bf16"):
    ops.set_matmul_mode(mode)
    for t in (send, rec, edge):
        t.grad = None
    net.zero_grad(set_to_none=True)
    ops.PROFILE.reset(enabled=True)
    for _ in range(6):
        out = net(send, rec, edge)
        outs = out if isinstance(out, tuple) else (out,)
        sum(o.square().sum() for o in outs).backward()
    recs = ops.PROFILE.collect()
    ops.PROFILE.reset(enabled=False)
    for t in (send, rec, edge):
        t.grad = None
    net.zero_grad(set_to_none=True)
    out = net(send, rec, edge)
    outs = out if isinstance(out, tuple) else (out,)
    sum(o.square().sum() for o in outs).backward()
    cur = [o.detach().clone() for o in outs] + [send.grad.clone(), rec.grad.clone(), edge.grad.clone()] + [
        p.grad.clone() for p in net.parameters()]
    if ref is None:
        ref = cur
    errs = [rel(a, b) for a, b in zip(cur, ref)]
    times = {k: sorted(v[2:])[len(v[2:]) // 2] * 1e3 for k, v in recs.items()}
    big = {k: v for k, v in times.items() if k[1] == E}
    print(f"  {mode:7s} max rel err vs f32: outputs {max(errs[:len(outs)]):.2e}  input grads {max(errs[len(outs):len(outs) + 3]):.2e}  "
          f"param grads {max(errs[len(outs) + 3:]):.2e}   edge-set launches (us): "
          + "  ".join(f"{k[0]}{k[2:]}={v:.0f}" for k, v in big.items()))
