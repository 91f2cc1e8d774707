"""Which earlier work of the same process changes the step time of a later configuration?  (`also` legs of bench.py: cfg3 41 / 46 ms,
cfg5 107 / 121 ms depending on which legs ran before.)

    python tools/r6/history_probe.py <later cfg> <prec> <steps> <history tokens...>
history tokens, executed in order before the later configuration is built:
    nograd    one no-grad step of cfg2            train    cfg2 Trainer (one graph), 30 steps
    fgraph    cfg2 forecaster graph + 20 replays   eager    cfg2 eager autograd steps (3), torch AdamW
    gts       cfg2 trainer.graphed_training_step + 5 steps
    free      del + gc + empty_cache (always done at the end of the history as bench.py does)
"""
import gc
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import bench  # noqa: E402
from neural_lam_amd import ops  # noqa: E402
from neural_lam_amd.trainer import Trainer, graphed_training_step  # noqa: E402

name, prec, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
hist = sys.argv[4:]
dev = torch.device("cuda:0")
if hist:
    c2 = bench.CONFIGS["cfg2"]
    _, _, raw, fc, step, batch = bench.build(c2, dev)
    keep = []
    for h in hist:
        if h == "nograd":
            with torch.no_grad():
                float(step(*batch)[1])
        elif h == "train":
            tr = Trainer(step, lr=1e-3, use_graph=True)
            for _ in range(30):
                tr.step(*batch)
            torch.cuda.synchronize()
            keep.append(tr)
        elif h == "fgraph":
            with torch.no_grad():
                for _ in range(2):
                    step.forecaster(batch[0], batch[2], batch[1])
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    step.forecaster(batch[0], batch[2], batch[1])
                for _ in range(20):
                    g.replay()
                torch.cuda.synchronize()
                keep.append(g)
        elif h == "eager":
            opt = torch.optim.AdamW(step.parameters(), lr=1e-3, betas=(0.9, 0.95))
            for _ in range(3):
                opt.zero_grad(set_to_none=True)
                _, loss = step(*batch)
                loss.backward()
                opt.step()
            torch.cuda.synchronize()
            del opt, loss
        elif h == "gts":
            opt = torch.optim.AdamW(step.parameters(), lr=1e-3, betas=(0.9, 0.95))
            g = graphed_training_step(step, *batch)
            for _ in range(5):
                opt.zero_grad(set_to_none=True)
                _, loss = g(*batch)
                loss.backward()
                opt.step()
            torch.cuda.synchronize()
            del opt, loss, g
        elif h == "free":
            pass
        else:
            raise SystemExit(f"unknown history token {h}")
    del keep, step, fc, batch, raw
    if "tr" in dir():
        del tr
    if "g" in dir():
        del g
    gc.collect()
    torch.cuda.empty_cache()
state = {k: (len(v) if hasattr(v, "__len__") else v) for k, v in vars(ops).items()
         if k in ("PACKER", "GRAD_LISTENER", "DIRECT_PARAM_GRADS", "EARLY_LEAF_BACKWARD", "_MAILBOX", "_MAIL_CONSUMERS", "ROLLOUT_SHARED", "_ROLLOUT_USES", "_ROLLOUT_ACC", "MAIL_TOKEN")}
cfg = bench.CONFIGS[name]
_, _, raw, _, step, batch = bench.build(cfg, dev)
with torch.autocast("cuda", dtype=torch.bfloat16, enabled=prec == "bf16"):
    tr = Trainer(step, lr=1e-3, use_graph=True)
    for _ in range(2):
        tr.step(*batch)
    res = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step(*batch)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / steps * 1e3)
seg = tr._graph
info = ""
if hasattr(seg, "chain"):
    info = f"chain graphs {len(seg.chain)}"
print(f"{name} {prec} history={'+'.join(hist) or '-'} ms/step {sorted(res)[1]:.3f} executor {tr.executor} {info} state {state}")
