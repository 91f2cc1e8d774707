"""Training trajectories (lr = 1e-3) of the one-graph executor, the segmented executor and the eager step at bench size: first
step at which the weights differ, and which parameters."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from neural_lam_amd.trainer import Trainer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4p"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
cfg = bench.CONFIGS[name]
runs = {}
for ex in ("eager", "forks", "segments"):
    _, _, _, fc, step, batch = bench.build(cfg, dev)
    tr = Trainer(step, lr=1e-3, use_graph=ex != "eager", executor=ex if ex != "eager" else None, forks_per_segment=K)
    names = [k for k, p in fc.named_parameters() if p.requires_grad]
    losses, flats, grads = [], [], []
    for it in range(nsteps):
        losses.append(float(tr.step(*batch)))
        torch.cuda.synchronize()
        flats.append(tr.fp.flat.clone())
        grads.append(tr.fp.grad.clone())
    runs[ex] = (losses, flats, grads, names, tr)
    print(ex, ["%.7f" % l for l in losses], flush=True)
ref = runs["eager"]
for ex in ("forks", "segments"):
    losses, flats, grads, names, tr = runs[ex]
    for it in range(nsteps):
        same_w, same_g = torch.equal(flats[it], ref[1][it]), torch.equal(grads[it], ref[2][it])
        if not (same_w and same_g):
            bad = []
            for i, p in enumerate(tr.fp.params):
                o, n = tr.fp.offsets[i], p.numel()
                dg = float((grads[it][o : o + n] - ref[2][it][o : o + n]).abs().max())
                dw = float((flats[it][o : o + n] - ref[1][it][o : o + n]).abs().max())
                if dg > 0 or dw > 0:
                    bad.append((names[i], "dgrad %.3e" % dg, "dweight %.3e" % dw))
            print(ex, "first difference at step", it, "weights equal:", same_w, "grads equal:", same_g, len(bad), "parameters")
            for b in bad[:25]:
                print("    ", b)
            break
    else:
        print(ex, "identical to eager over", nsteps, "steps")
