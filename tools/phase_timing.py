"""Per-phase wave-cycle accounting of the fused kernels (instrumented build, -DNLAM_TIMING).

  python tools/phase_timing.py build            # here (cross-compile) -> neural_lam_amd/libnlam_hip_timing.so
  NLAM_LIB=neural_lam_amd/libnlam_hip_timing.so python tools/phase_timing.py run m2g 64   # on the GPU box

Prints, per launch kind, the average s_memtime cycles per tile a wave spends in each
phase (s_memtime ticks = shader cycles; the instrumented build drains the load queue at
the phase-1 mark so phase 1 = the descriptor -> index -> row dependent-load chain)."""
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
TLIB = ROOT / "neural_lam_amd" / "libnlam_hip_timing.so"

if sys.argv[1] == "build":
    from neural_lam_amd import _lib

    print(_lib.build(verbose=True, out=TLIB, defines=("NLAM_TIMING",)))
    sys.exit(0)

assert os.environ.get("NLAM_LIB"), "run with NLAM_LIB=neural_lam_amd/libnlam_hip_timing.so"
import torch  # noqa: E402

from neural_lam_amd import _lib as L  # noqa: E402
from neural_lam_amd import gnn_layers as hl  # noqa: E402
from neural_lam_amd import graph as G  # noqa: E402

which = sys.argv[2] if len(sys.argv) > 2 else "m2g"
d = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda:0")
lib = L.load()
lib.nlam_debug_phase_cycles.argtypes = [C.POINTER(C.c_ulonglong)]
raw = G.create_regular_grid_graph(G.regular_grid_xy(238, 268))
ei = raw[f"{which}_edge_index"] if which != "m2m" else raw["m2m_edge_index"][0]
ns, nr, E = int(ei[0].max()) + 1, int(ei[1].max()) + 1, ei.shape[1]
torch.manual_seed(0)
net = hl.InteractionNet(ei, d, update_edges=(which == "m2m")).to(dev)
send = torch.randn(1, ns, d, device=dev, requires_grad=True)
rec = torch.randn(1, nr, d, device=dev, requires_grad=True)
edge = torch.randn(1, E, d, device=dev, requires_grad=True)
buf = (C.c_ulonglong * 16)()


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"   [launch sequence took {e0.elapsed_time(e1) * 1e3:.1f} us]")
    return r


def read(label, names):
    lib.nlam_debug_phase_cycles(buf)
    v = list(buf)
    tiles, waves = max(v[12], 1), max(v[13], 1)
    print(f"{label}: {tiles} tiles over {waves} waves")
    tot = sum(v[:12])
    for k, nm in enumerate(names):
        per = v[k] / (waves if nm.startswith("*") else tiles)
        print(f"   phase {k} {nm:38s} {per:10.0f} cyc/{'wave' if nm.startswith('*') else 'tile'}   {100.0 * v[k] / tot:5.1f} %")
    print(f"   total wave-cycles per tile {tot / tiles:.0f}")


FWD = ["*prologue (first loads + weights -> LDS)", "tile top: late units, idx(n+1), epilogue idx", "GEMM1 (+ row waits)",
       "resid, rows(n+1) issue, z1 store, SiLU", "GEMM2", "LN, xhat store", "msg, segment reduce, out store", "*tail drain"]
for rep in range(3):
    out = net(send, rec, edge)  # edge kernel then node kernel
    torch.cuda.synchronize()
lib.nlam_debug_phase_cycles(buf)
# edge kernel only: call the edge stage directly
aggr, eo = timed(lambda: net._messages_and_aggregate(send, rec, edge, net.update_edges, True))
read(f"edge fwd (training mode) {which} d={d} E={E}", FWD)
with torch.no_grad():
    aggr, eo = timed(lambda: net._messages_and_aggregate(send, rec, edge, net.update_edges, True))
read(f"edge fwd (inference mode) {which} d={d}", FWD)
out = timed(lambda: net._node_update(rec, aggr.detach()))
read(f"node fwd {which} d={d} N={nr}", FWD)

BWD = ["*prologue (weights -> LDS)", "indices", "dmsg loads, LN backward, dbeta/dgamma", "dz2 rows out + db2",
       "dh GEMM, silu', dz1 rows out + db1", "dx GEMMs + data-gradient outputs", "*tail drain", "-"]
aggr, eo = net._messages_and_aggregate(send, rec, edge, net.update_edges, True)
lib.nlam_debug_phase_cycles(buf)
(aggr.sum() + (eo.sum() if eo is not None else 0.0)).backward()
read(f"edge bwd {which} d={d} E={E} (counters also include the 2 wgrad kernels: none instrumented)", BWD)
