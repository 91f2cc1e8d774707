"""Does the step time depend on WHERE the caching allocator places the step's buffers?  A dummy allocation of PAD bytes taken (and kept,
or freed again with KEEP=0) before the model is built shifts every later block.

    python tools/r6/alloc_offset_probe.py cfg3 fp32 10 PAD [KEEP]
"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import bench  # noqa: E402
from neural_lam_amd.trainer import Trainer  # noqa: E402

name, prec, steps, pad = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
keep = int(sys.argv[5]) if len(sys.argv) > 5 else 1
dev = torch.device("cuda:0")
dummy = torch.empty(pad, dtype=torch.uint8, device=dev) if pad > 0 else None
if not keep:
    del dummy   # the block goes back to the allocator's cache: the next allocation of that size class re-uses it
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
print(f"{name} {prec} pad={pad} keep={keep} ms/step {sorted(res)[1]:.3f} (regions {[round(r, 2) for r in res]})")
