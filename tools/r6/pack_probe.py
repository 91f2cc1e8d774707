"""What the 32 us of mlp_pack_kernel at the head of a cfg2 step are made of: the launch (a) back to back in a HIP graph (warm
caches), (b) behind a kernel that rewrites the weights (as AdamW does) and behind 1 GB of unrelated traffic (cold L2 / TLB),
(c) with the job table cut to the first n jobs."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from neural_lam_amd import _lib as L  # noqa: E402
from neural_lam_amd.trainer import Trainer  # noqa: E402
import ctypes as C  # noqa: E402

dev = torch.device("cuda:0")
_, _, _, _, step, batch = bench.build(bench.CONFIGS["cfg2"], dev)
tr = Trainer(step, lr=1e-3, use_graph=True)
for _ in range(4):
    tr.step(*batch)
torch.cuda.synchronize()
pk = tr._packer
lib = L.load()
njobs = len(pk.table_entries)
print("jobs", njobs, "table bytes", pk.table.numel())
big = torch.empty(256 << 20, device=dev)   # 1 GiB
flat = tr.fp.flat


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def pack(n=njobs):
    L.check(lib.nlam_mlp_pack(C.c_void_p(pk.table.data_ptr()), n, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "pack")


t_pack = timed(pack)
print(f"pack alone, back to back: {t_pack:.1f} us")
for n in (1, 4, 8):
    print(f"  first {n} jobs: {timed(lambda: pack(n)):.1f} us")
t_w = timed(lambda: flat.mul_(1.0))
print(f"weights rewritten (flat.mul_): {t_w:.1f} us; + pack: {timed(lambda: (flat.mul_(1.0), pack())):.1f} us")
t_b = timed(lambda: big.fill_(0.0), reps=5)
print(f"1 GiB fill: {t_b:.1f} us; + pack: {timed(lambda: (big.fill_(0.0), pack()), reps=5):.1f} us; + weights + pack: {timed(lambda: (big.fill_(0.0), flat.mul_(1.0), pack()), reps=5):.1f} us")
small = torch.empty(53760, device=dev)
print(f"small fill: {timed(lambda: small.fill_(0.0)):.1f} us; + pack {timed(lambda: (small.fill_(0.0), pack())):.1f} us")
