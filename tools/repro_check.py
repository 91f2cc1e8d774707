import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from neural_lam_amd import graph as G, models as hm
from neural_lam_amd.datastore import meps_like_datastore
dev = torch.device("cuda:0")
ds = meps_like_datastore("/tmp/nlam_test_meps")
ext = ds.get_xy_extent("state")
raw = G.create_regular_grid_graph(ds.get_xy("state"))
graph = G.normalise_graph(raw, max(ext[1] - ext[0], ext[3] - ext[2]))
torch.manual_seed(42)
fc = hm.ARForecaster(hm.GraphLAM(ds, graph=graph, hidden_dim=64, processor_layers=4), ds)
step = hm.ForecasterStep(fc, ds).to(dev)
N = ds.num_grid_points
torch.manual_seed(5)
init, target, forcing = torch.randn(1, 2, N, 17, device=dev), torch.randn(1, 1, N, 17, device=dev), torch.randn(1, 1, N, 18, device=dev)
grads = []
for it in range(4):
    step.zero_grad(set_to_none=True)
    _, loss = step(init, target, forcing)
    loss.backward()
    grads.append({k: p.grad.clone() for k, p in step.named_parameters()})
    print("loss", float(loss))
for it in range(1, 4):
    bad = [(k, float((grads[0][k] - grads[it][k]).abs().max())) for k in grads[0] if not torch.equal(grads[0][k], grads[it][k])]
    print(it, "differing params:", bad[:8], len(bad))
