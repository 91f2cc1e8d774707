"""Micro-benchmark of the dominant launches (m2g edge set, d=64): edge fwd (training
mode), edge bwd, the two wgrad launches.  Used under rocprofv3 for PMC passes."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from neural_lam_amd import gnn_layers as hl  # noqa: E402
from neural_lam_amd import graph as G  # noqa: E402
from neural_lam_amd import ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "m2g"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
import os  # noqa: E402
if os.environ.get("NLAM_KB_AUTOCAST", "0") == "1":   # the one-term (plain bf16) kernels, as under --precision bf16
    torch.autocast("cuda", dtype=torch.bfloat16).__enter__()
raw = G.create_regular_grid_graph(G.regular_grid_xy(238, 268))
ei = raw[f"{which}_edge_index"] if which != "m2m" else raw["m2m_edge_index"][0]
ns, nr, E = int(ei[0].max()) + 1, int(ei[1].max()) + 1, ei.shape[1]
d = int(sys.argv[3]) if len(sys.argv) > 3 else 64
stage = sys.argv[4] if len(sys.argv) > 4 else "layer"   # layer | edge (edge kernel + its backward only: unambiguous PMC rows)
torch.manual_seed(0)
net = hl.InteractionNet(ei, d, update_edges=(which == "m2m")).to(dev)
send = torch.randn(1, ns, d, device=dev, requires_grad=True)
rec = torch.randn(1, nr, d, device=dev, requires_grad=True)
edge = torch.randn(1, E, d, device=dev, requires_grad=True)
ops.PROFILE.reset(enabled=True)
for _ in range(reps):
    if stage == "edge":
        aggr, eo = net._messages_and_aggregate(send, rec, edge, net.update_edges, True)
        (aggr.sum() + (eo.sum() if eo is not None else 0.0)).backward()
        continue
    out = net(send, rec, edge)
    outs = out if isinstance(out, tuple) else (out,)
    sum(o.sum() for o in outs).backward()
recs = ops.PROFILE.collect()
print(f"{which}: E={E} Ns={ns} Nr={nr} d={d}")
for key, v in recs.items():
    v = sorted(v[2:])
    med = v[len(v) // 2]
    name, rows = key[0], key[1]
    if name in ("mlp_fwd", "mlp_bwd"):
        kin, hid, dout = key[2:5]
        fl = 2.0 * rows * (kin * hid + hid * dout)
    else:
        fl = 2.0 * rows * key[2] * key[3]
    print(f"  {str(key):45s} median {med * 1e3:8.1f} us   {fl / (med * 1e-3) / 1e12:6.1f} TFLOP/s")
