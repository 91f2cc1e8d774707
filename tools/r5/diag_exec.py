"""Which parameter gradients differ between the one-graph and the segmented executor at bench size (cfg4p by default)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from neural_lam_amd.trainer import Trainer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4p"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
cfg = bench.CONFIGS[name]
out = {}
for ex in ("forks", "segments", "eager"):
    _, _, _, fc, step, batch = bench.build(cfg, dev)
    tr = Trainer(step, lr=0.0, use_graph=ex != "eager", executor=ex if ex != "eager" else None, forks_per_segment=K)
    grads = []
    for it in range(3):
        loss = tr.step(*batch)
        torch.cuda.synchronize()
        grads.append(tr.fp.grad.clone())
    names = [k for k, p in fc.named_parameters() if p.requires_grad]
    out[ex] = (float(loss), grads, names, tr)
    print(ex, "loss", float(loss), "replays equal:", [bool(torch.equal(grads[0], g)) for g in grads[1:]], flush=True)
ref = out["eager"]
for ex in ("forks", "segments"):
    loss, grads, names, tr = out[ex]
    bad = []
    for i, p in enumerate(tr.fp.params):
        o, n = tr.fp.offsets[i], p.numel()
        a, b = grads[-1][o : o + n], ref[1][-1][o : o + n]
        if not torch.equal(a, b):
            bad.append((names[i], float((a - b).abs().max()), float(b.abs().max())))
    print(ex, "vs eager:", len(bad), "parameters differ")
    for b in bad[:40]:
        print("   ", b)
if isinstance(out["segments"][3]._graph, object) and hasattr(out["segments"][3]._graph, "chain"):
    g = out["segments"][3]._graph
    print("segments:", len(g.chain), "chain graphs,", sum(len(s) for s in g.side), "side graphs,", g.nforks, "forks")
