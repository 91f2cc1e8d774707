"""Per-phase wave cycles of the grouped embedder backward with fused weight gradients (NLAM_F_LEAF_WGRAD), the tail of the
cfg2 step.  Instrumented build (tools/phase_timing.py build), on the GPU box:

  NLAM_LIB=neural_lam_amd/libnlam_hip_timing.so python tools/phase_timing_lw.py
"""
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
assert os.environ.get("NLAM_LIB"), "run with NLAM_LIB=neural_lam_amd/libnlam_hip_timing.so"
import torch  # noqa: E402

from neural_lam_amd import _lib as L  # noqa: E402
from neural_lam_amd import gnn_layers as hl  # noqa: E402

dev = torch.device("cuda:0")
lib = L.load()
lib.nlam_debug_phase_cycles.argtypes = [C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * 16)()
torch.manual_seed(0)
# the four static-feature embedders of GraphLAM at MEPS size (cfg2): g2m / m2g / m2m edge features (3 columns), mesh nodes (2)
shapes = [(79236, 3), (255136, 3), (57616, 3), (6561, 2)]
mlps = [hl.make_mlp([k, 64, 64]).to(dev) for _, k in shapes]
xs = [torch.randn(r, k, device=dev) for r, k in shapes]
NAMES = ["*prologue (weights -> LDS)", "indices, input rows -> LDS", "LN finish, next tile's xhat / rstd / input row issued", "db2",
         "dW1 weighted sums, silu(z1) fragments, dW2 MFMAs", "dz2^T fragments (A operand of dW2)", "*tail drain", "z1 recompute, sigmoid, dz1, db1",
         "g / xhat rows into registers", "dbeta, dgamma column sums", "gamma * g, row sums", "dh GEMM"]
for rep in range(3):
    outs = hl.grouped_mlp_forward(list(zip(mlps, xs)))
    torch.cuda.synchronize()
    lib.nlam_debug_phase_cycles(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    torch.autograd.backward(outs, [torch.randn_like(o) for o in outs])
    e1.record()
    torch.cuda.synchronize()
lib.nlam_debug_phase_cycles(buf)
v = list(buf)
tiles, waves = max(v[12], 1), max(v[13], 1)
tot = sum(v[:12])
print(f"grouped embedder backward: {tiles} tiles over {waves} waves; backward call {e0.elapsed_time(e1) * 1e3:.1f} us (incl. the reduction launch)")
for k, nm in enumerate(NAMES):
    per = v[k] / (waves if nm.startswith("*") else tiles)
    print(f"   phase {k} {nm:45s} {per:10.0f} cyc/{'wave' if nm.startswith('*') else 'tile'}   {100.0 * v[k] / tot:5.1f} %")
print(f"   total wave-cycles per tile {tot / tiles:.0f}")
