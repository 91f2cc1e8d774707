"""Per-phase wave cycles of the wide split-bf16 forward (instrumented build, see tools/phase_timing.py).

  python tools/phase_timing.py build
  NLAM_LIB=neural_lam_amd/libnlam_hip_timing.so python tools/phase_timing_wbf.py m2g 256 [bf16]
(third argument "bf16": under torch.autocast, i.e. the one-term kernels)
"""
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
assert os.environ.get("NLAM_LIB"), "run with NLAM_LIB=neural_lam_amd/libnlam_hip_timing.so"
import torch  # noqa: E402

from neural_lam_amd import _lib as L  # noqa: E402
from neural_lam_amd import gnn_layers as hl  # noqa: E402
from neural_lam_amd import graph as G  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "m2g"
d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
if len(sys.argv) > 3 and sys.argv[3] == "bf16":
    _ac = torch.autocast("cuda", dtype=torch.bfloat16)
    _ac.__enter__()   # for the whole script
lib = L.load()
lib.nlam_debug_phase_cycles.argtypes = [C.POINTER(C.c_ulonglong)]
raw = G.create_regular_grid_graph(G.regular_grid_xy(238, 268))
ei = raw[f"{which}_edge_index"] if which != "m2m" else raw["m2m_edge_index"][0]
ns, nr, E = int(ei[0].max()) + 1, int(ei[1].max()) + 1, ei.shape[1]
torch.manual_seed(0)
net = hl.InteractionNet(ei, d, update_edges=(which == "m2m")).to(dev)
send = torch.randn(1, ns, d, device=dev)
rec = torch.randn(1, nr, d, device=dev)
edge = torch.randn(1, E, d, device=dev)
buf = (C.c_ulonglong * 16)()
NAMES = ["row ids + sync", "first chunk load/split/sync", "GEMM1 chunks", "publish hidden (z1 store, SiLU, split) + sync",
         "GEMM2", "LN: normalise, xhat store, affine", "msg/out stores, segment reduce, syncs", "*tail drain",
         "LN: bias2, row sums", "LN: barrier 1", "LN: centre, squares", "LN: barrier 2"]


def read(label):
    lib.nlam_debug_phase_cycles(buf)
    v = list(buf)
    tiles, waves = max(v[12], 1), max(v[13], 1)
    tot = sum(v[:12])
    print(f"{label}: {tiles} super tiles over {waves} waves")
    for k, nm in enumerate(NAMES):
        per = v[k] / (waves if nm.startswith("*") else tiles)
        print(f"   phase {k} {nm:48s} {per:10.0f} cyc   {100.0 * v[k] / tot:5.1f} %")
    print(f"   total wave-cycles per super tile {tot / tiles:.0f}")


for mode in ("inference", "training"):
    for rep in range(2):
        with torch.set_grad_enabled(mode == "training"):
            if mode == "training":
                edge.requires_grad_(True)
            aggr, eo = net._messages_and_aggregate(send, rec, edge, net.update_edges, True)
        torch.cuda.synchronize()
        lib.nlam_debug_phase_cycles(buf) if rep == 0 else read(f"edge fwd ({mode}) {which} d={d} E={E}")
with torch.no_grad():
    out = net._node_update(rec, aggr.detach())
    lib.nlam_debug_phase_cycles(buf)
    out = net._node_update(rec, aggr.detach())
    torch.cuda.synchronize()
read(f"node fwd {which} d={d} N={nr}")

BNAMES = ["tile setup (descriptors, row pointers)", "A: dmsg / xhat loads, LN sums, column sums", "A: LN finish, dz2 store, db2, publish, barrier",
          "GEMM dh = W2^T dz2", "B: barrier, z1 load, silu', dz1 store, db1, publish, barrier", "GEMMs dx_s = W1_s^T dz1",
          "C: residual terms, row stores / segment sums", "*tail drain", "A: barrier (row statistics)", "end barrier", "-", "-"]
NAMES[:] = BNAMES
edge.requires_grad_(True)
send.requires_grad_(True)
rec.requires_grad_(True)
for rep in range(2):
    aggr, eo = net._messages_and_aggregate(send, rec, edge, net.update_edges, True)
    torch.cuda.synchronize()
    lib.nlam_debug_phase_cycles(buf)
    (aggr.sum() + (eo.sum() if eo is not None else 0.0)).backward()
    torch.cuda.synchronize()
    if rep == 1:
        read(f"edge bwd {which} d={d} E={E}")
    else:
        lib.nlam_debug_phase_cycles(buf)
