"""Does the step time depend on WHICH pool streams torch hands the trainer?  torch.cuda.Stream() deals streams out of a per-device pool
round-robin, and the runtime maps streams onto a few hardware queues: N dummy streams taken first shift every later stream by N.

    python tools/r6/stream_offset_probe.py cfg5 bf16 5 N      -> ms/step of the trainer's step with N dummy streams taken first
"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import bench  # noqa: E402
from neural_lam_amd.trainer import Trainer  # noqa: E402

name, prec, steps, ndummy = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda:0")
dummies = [torch.cuda.Stream() for _ in range(ndummy)]
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
print(f"{name} {prec} dummy_streams={ndummy} ms/step {sorted(res)[1]:.3f} (regions {[round(r, 2) for r in res]})")
