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
ops.set_matmul_mode("bf16x3")
aggr, _ = net._messages_and_aggregate(send, rec, edge, False, True)
csr = net._csr(dev, ns)
inv = torch.empty(E, dtype=torch.long, device=dev)
inv[csr.perm.long()] = torch.arange(E, device=dev)
ref = None
for it in range(40):
    gs, gr, ge = torch.autograd.grad((aggr * cot).sum(), (send, rec, edge), retain_graph=True)
    if ref is None:
        ref = ge.clone()
        continue
    if not torch.equal(ge, ref):
        d = (ge != ref)[0]
        rows = d.any(1).nonzero().reshape(-1)
        cols = d.any(0).nonzero().reshape(-1)
        pos = inv[rows].sort().values
        tile, inrow = pos // 32, pos % 32
        print(f"it {it}: {rows.numel()} rows, csr pos {int(pos.min())}..{int(pos.max())}, tile(s) {sorted(set(tile.tolist()))}, "
              f"rows in tile {int(inrow.min())}..{int(inrow.max())}, cols {int(cols.min())}..{int(cols.max())}, "
              f"tile % 2048 = {[t % 2048 for t in sorted(set(tile.tolist()))]}")
        r0 = rows[0]
        print("    wrong", ge[0, r0, cols[:4]].tolist(), "right", ref[0, r0, cols[:4]].tolist())
print("done")
