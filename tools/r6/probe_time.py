import sys, time, torch
sys.path.insert(0, '.')
from neural_lam_amd import ops
torch.zeros(1, device="cuda").sum().item()
t0 = time.perf_counter()
L = ops.stream_layout()
torch.cuda.synchronize()
print("stream_layout: %.1f ms, groups %s" % ((time.perf_counter() - t0) * 1e3, [len(g) for g in L["groups"]]))
t0 = time.perf_counter(); torch.cuda._sleep(4_000_000); torch.cuda.synchronize(); print("_sleep(4M): %.2f ms" % ((time.perf_counter() - t0) * 1e3))
