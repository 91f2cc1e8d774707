"""cProfile of the EAGER Lightning-shaped cfg2 step (every launch from Python): where the ~15 us of host time per launch go."""
import cProfile
import pstats
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

dev = torch.device("cuda:0")
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
_, _, _, _, step, batch = bench.build(cfg, dev)
opt = torch.optim.AdamW(step.parameters(), lr=1e-3, betas=(0.9, 0.95))


def one():
    opt.zero_grad(set_to_none=True)
    _, loss = step(*batch)
    loss.backward()
    opt.step()


for _ in range(5):
    one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    one()
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) / 50 * 1e3)
# host time only (no sync inside): forward / backward / optimizer separately
tf = tb = to = 0.0
for _ in range(50):
    a = time.perf_counter(); opt.zero_grad(set_to_none=True); _, loss = step(*batch); b = time.perf_counter(); loss.backward(); c = time.perf_counter(); opt.step(); d = time.perf_counter()
    tf += b - a; tb += c - b; to += d - c
torch.cuda.synchronize()
print("host ms per step: forward %.2f backward %.2f optimizer %.2f" % (tf / 50 * 1e3, tb / 50 * 1e3, to / 50 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    one()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
