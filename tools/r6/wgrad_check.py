"""Round 6: wgrad_ldma_kernel against the column-per-thread kernel and a torch reference, straight through the C-ABI.
   python tools/r6/wgrad_check.py [rows] [m] [n]"""
import ctypes as C
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from neural_lam_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 57616
m = int(sys.argv[2]) if len(sys.argv) > 2 else 512
n = int(sys.argv[3]) if len(sys.argv) > 3 else 512
MM1 = 1 << 8   # NLAM_F_MM one term


def run(A, S, idx, flags, tune):
    lib.nlam_set_tuning(L.TUNE_WGRAD_LDMA, tune)
    q = L.Wgrad()
    q.A, q.m, q.batch, q.rows, q.nsrc, q.flags, q.n = A.data_ptr(), m, 1, rows, 1, flags, n
    q.src[0].ptr, q.src[0].idx, q.src[0].bstride, q.src[0].width = S.data_ptr(), (idx.data_ptr() if idx is not None else None), 0, n
    nparts = lib.nlam_wgrad_nparts(C.byref(q))
    part = torch.full((nparts, m, n), float("nan"), device=dev)
    q.partials, q.nparts = part.data_ptr(), nparts
    rc = lib.nlam_wgrad(C.byref(q), None)
    torch.cuda.synchronize()
    assert rc == 0, rc
    return part.sum(0), nparts


def report(name, got, ref):
    err = (got - ref).abs()
    rel = float(err.max()) / float(ref.abs().max())
    print(f"{name}: max|err| / max|ref| = {rel:.3e}  nan {int(torch.isnan(got).sum())}")
    if rel > 1e-3:
        blk = err.reshape(m // 32, 32, n // 32, 32).amax(dim=(1, 3)) / float(ref.abs().max())
        torch.set_printoptions(precision=2, linewidth=250, sci_mode=False)
        print("per 32x32 block max rel err (rows = m blocks, cols = n blocks):")
        print(blk.cpu())
        # inside the worst block: by row / column
        bi = int(blk.argmax())
        mb, nb = bi // (n // 32), bi % (n // 32)
        e = err[32 * mb : 32 * mb + 32, 32 * nb : 32 * nb + 32] / float(ref.abs().max())
        print("worst block", mb, nb, "err by row:", e.amax(1).cpu(), "by col:", e.amax(0).cpu())


torch.manual_seed(0)
Af = torch.randn(rows, m, device=dev)
Sf = torch.randn(rows, n, device=dev)
Ab, Sb = Af.bfloat16(), Sf.bfloat16()
idx = torch.randperm(rows, device=dev, dtype=torch.int32)
F_SILU, F_A, F_S = L.F_SILU_B, L.F_A_BF16, L.F_S_BF16
cases = [
    ("dW2-like: A bf16, S bf16 + SiLU", Ab, Sb, None, MM1 | F_SILU | F_A | F_S,
     lambda: Ab.float().t() @ torch.nn.functional.silu(Sb.float()).bfloat16().float()),
    ("dW1-like: A bf16, S fp32 gathered", Ab, Sf, idx, MM1 | F_A, lambda: Ab.float().t() @ Sf[idx.long()].bfloat16().float()),
    ("fp32 / fp32 one term", Af, Sf, idx, MM1, lambda: Af.bfloat16().float().t() @ Sf[idx.long()].bfloat16().float()),
    ("fp32 / fp32 one term + SiLU", Af, Sf, None, MM1 | F_SILU, lambda: Af.bfloat16().float().t() @ torch.nn.functional.silu(Sf).bfloat16().float()),
]
MM3 = 3 << 8
cases += [
    ("fp32 / fp32 THREE terms, gathered", Af, Sf, idx, MM3, lambda: Af.double().t() @ Sf[idx.long()].double()),
    ("fp32 / fp32 THREE terms + SiLU", Af, Sf, None, MM3 | F_SILU, lambda: Af.double().t() @ torch.nn.functional.silu(Sf).double()),
]
for name, A, S, ix, flags, ref_fn in cases:
    ref = ref_fn().float()
    old, np0 = run(A, S, ix, flags, 0)
    new, np1 = run(A, S, ix, flags, 7)
    print("  bit-identical to the column-per-thread kernel:", bool(torch.equal(old, new)))
    print(f"--- {name} (rows {rows}, m {m}, n {n}; nparts {np0}/{np1})")
    report("  column-per-thread kernel", old, ref)
    report("  wgrad_ldma_kernel       ", new, ref)
    for tune, var, label in ((0, 0, "old"), (7, 0, "new")):
        lib.nlam_set_tuning(L.TUNE_WGRAD_LDMA, tune)
        lib.nlam_set_tuning(L.TUNE_WGRAD_LDMA_VAR, var)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        q = L.Wgrad()
        q.A, q.m, q.batch, q.rows, q.nsrc, q.flags, q.n = A.data_ptr(), m, 1, rows, 1, flags, n
        q.src[0].ptr, q.src[0].idx, q.src[0].bstride, q.src[0].width = S.data_ptr(), (ix.data_ptr() if ix is not None else None), 0, n
        nparts = lib.nlam_wgrad_nparts(C.byref(q))
        part = torch.empty((nparts, m, n), device=dev)
        q.partials, q.nparts = part.data_ptr(), nparts
        rc = 0
        for _ in range(3):
            rc = lib.nlam_wgrad(C.byref(q), None)
        if rc != 0:
            print(f"  {label}: launch failed rc {rc}")
            continue
        if False:
            part.fill_(float("nan"))
            lib.nlam_wgrad(C.byref(q), None)
            got = part.sum(0)
            print(f"  {label}: max|err| / max|ref| = {float((got - ref).abs().max()) / float(ref.abs().max()):.3e}")
        e0.record()
        for _ in range(20):
            lib.nlam_wgrad(C.byref(q), None)
        e1.record()
        torch.cuda.synchronize()
        print(f"  {label}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
    lib.nlam_set_tuning(L.TUNE_WGRAD_LDMA_VAR, 0)
