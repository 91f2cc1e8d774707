"""Achieved HBM bandwidth of the bandwidth-bound helpers on the MEPS edge sets (north_star: "achieved HBM GB/s on
the gather/scatter"): the CSC scatter-by-sender segment sum, the CSR aggregate, and the flat AdamW step.

    python tools/hbm_kernels_bench.py [d]      # on the GPU box

Algorithmic bytes = rows read once + rows written once + the int32 index / pointer arrays; peak 8 TB/s."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from neural_lam_amd import graph as G  # noqa: E402
from neural_lam_amd import ops  # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
raw = G.create_regular_grid_graph(G.regular_grid_xy(238, 268))


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2] * 1e-3


print(f"d = {d}; peak 8000 GB/s")
for which in ("m2g", "g2m", "m2m"):
    ei = raw[f"{which}_edge_index"] if which != "m2m" else raw["m2m_edge_index"][0]
    ns, nr, E = int(ei[0].max()) + 1, int(ei[1].max()) + 1, ei.shape[1]
    csr = G.build_edge_csr(ei, ns, nr).to(dev) if hasattr(G, "build_edge_csr") else None
    if csr is None:
        from neural_lam_amd.gnn_layers import InteractionNet

        csr = InteractionNet(ei, 8).to(dev).csr
    rows = torch.randn(1, E, d, device=dev)
    # aggregate by receiver (CSR order = identity permutation of the sorted edge list)
    t = timed(lambda: ops.segment_sum(rows, E * d, csr.rowptr, csr.perm, None, csr.num_rec, d, 1))
    by = 4.0 * d * (E + nr) + 4.0 * (E + nr)
    print(f"  {which}: aggregate by receiver   E={E:7d} -> {nr:6d} rows   {t * 1e6:7.1f} us   {by / t / 1e9:7.0f} GB/s  ({by / t / 8e12:.2f} of peak)")
    # scatter by sender (CSC view: gather-sum of the rows of each sender)
    t = timed(lambda: ops.segment_sum(rows, E * d, csr.colptr, csr.cperm, None, csr.num_send, d, 1))
    by = 4.0 * d * (E + ns) + 4.0 * (E + ns)
    print(f"  {which}: scatter by sender       E={E:7d} -> {ns:6d} rows   {t * 1e6:7.1f} us   {by / t / 1e9:7.0f} GB/s  ({by / t / 8e12:.2f} of peak)")

for n in (214865, 5160209, 20544017):   # cfg2 / cfg3 / cfg5 parameter counts
    flat, grad = torch.randn(n, device=dev), torch.randn(n, device=dev)
    opt = ops.AdamWFlat(flat, grad, lr=1e-3)
    t = timed(lambda: opt.step(1.0))
    by = 4.0 * n * 7   # p, g, m, v read; p, m, v written
    print(f"  adamw: {n:9d} parameters   {t * 1e6:7.1f} us   {by / t / 1e9:7.0f} GB/s  ({by / t / 8e12:.2f} of peak)")
