"""Round 6 probe: does a stream created with hipExtStreamCreateWithCUMask confine kernels to its CUs --
for an eager launch, and for a HIP graph replayed on it?  Times one bandwidth-bound kernel (a 1 GiB copy) and one latency-bound
fused launch on: the default stream, a stream masked to 1/4 of the CUs (every bit of the first 64 / an XCD-interleaved quarter),
and a graph captured on a plain stream but launched into the masked one."""
import ctypes as C
import glob
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))

tl = os.path.join(os.path.dirname(torch.__file__), "lib")
cands = sorted(glob.glob(os.path.join(tl, "libamdhip64.so*")))
hip = C.CDLL(cands[0]) if cands else C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[0] * 8)
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def timeit(fn, stream, n=20):
    with torch.cuda.stream(stream):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n):
            fn()
        e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


dev = torch.device("cuda:0")
a = torch.empty(1 << 28, device=dev)   # 1 GiB fp32
b = torch.empty_like(a)
x = torch.randn(4096, 4096, device=dev)
copy = lambda: b.copy_(a)   # noqa: E731
sil = lambda: torch.nn.functional.silu(x)   # noqa: E731
plain = torch.cuda.Stream()
masks = {
    "first 64 bits": list(range(64)),
    "every 4th bit": list(range(0, 256, 4)),
    "bits 0-191": list(range(192)),
    "all 256": list(range(256)),
    "bits 0-31": list(range(32)),
    "bits 0-127": list(range(128)),
    "bits 64-255": list(range(64, 256)),
    "bits i % 8 == 0": list(range(0, 256, 8)),
    "bits i % 8 < 2": [i for i in range(256) if i % 8 < 2],
    "bits i % 32 < 8": [i for i in range(256) if i % 32 < 8],
    "bits i % 32 < 24": [i for i in range(256) if i % 32 < 24],
}
mm = torch.randn(4096, 4096, device=dev)
copy = lambda: torch.mm(mm, mm)   # noqa: E731  (compute-bound: its time scales with the CUs it may use)
print("plain stream: copy %.1f us, silu %.1f us" % (timeit(copy, plain), timeit(sil, plain)))
for name, bits in masks.items():
    st = masked_stream(bits)
    print("masked (%s, %d CUs): copy %.1f us, silu %.1f us" % (name, len(bits), timeit(copy, st), timeit(sil, st)))
    # a graph captured on the plain stream, replayed into the masked stream
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(plain):
        with torch.cuda.graph(g, stream=plain):
            b.copy_(a)
    print("   graph captured on a plain stream, replayed on the masked one: %.1f us" % timeit(g.replay, st))
    # a graph captured ON the masked stream, replayed on it
    g2 = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g2, stream=st):
            b.copy_(a)
        print("   graph captured on the masked stream, replayed on it: %.1f us; replayed on the plain stream: %.1f us"
              % (timeit(g2.replay, st), timeit(g2.replay, plain)))
    except Exception as exc:
        print("   capture on the masked stream failed:", repr(exc)[:200])
# concurrency: a long copy on the 3/4 mask next to a latency-bound kernel on the 1/4 mask
big, small = masked_stream(list(range(64, 256))), masked_stream(list(range(64)))
ev = torch.cuda.Event()
with torch.cuda.stream(big):
    for _ in range(10):
        copy()
t_alone = timeit(sil, small, n=50)
torch.cuda.synchronize()
with torch.cuda.stream(big):
    for _ in range(40):
        copy()
t_beside = timeit(sil, small, n=50)
torch.cuda.synchronize()
with torch.cuda.stream(plain):
    for _ in range(40):
        copy()
t_beside_unmasked = timeit(sil, torch.cuda.Stream(), n=50)
torch.cuda.synchronize()
print("silu on the 1/4 mask: alone %.1f us, beside copies on the other 3/4 %.1f us; unmasked silu beside unmasked copies %.1f us" % (t_alone, t_beside, t_beside_unmasked))
