"""Round 6: the four one-term weight-gradient launches of wgrad_check.py, new kernel only, a few dispatches each (for rocprofv3 --pmc)."""
import ctypes as C
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from neural_lam_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
rows, m, n = 57616, 512, 512
lib.nlam_set_tuning(L.TUNE_WGRAD_LDMA, 3)
torch.manual_seed(0)
Af, Sf = torch.randn(rows, m, device=dev), torch.randn(rows, n, device=dev)
Ab, Sb = Af.bfloat16(), Sf.bfloat16()
idx = torch.randperm(rows, device=dev, dtype=torch.int32)
MM1 = 1 << 8
for A, S, ix, flags in ((Ab, Sb, None, MM1 | L.F_SILU_B | L.F_A_BF16 | L.F_S_BF16), (Ab, Sf, idx, MM1 | L.F_A_BF16)):
    q = L.Wgrad()
    q.A, q.m, q.batch, q.rows, q.nsrc, q.flags, q.n = A.data_ptr(), m, 1, rows, 1, flags, n
    q.src[0].ptr, q.src[0].idx, q.src[0].bstride, q.src[0].width = S.data_ptr(), (ix.data_ptr() if ix is not None else None), 0, n
    nparts = lib.nlam_wgrad_nparts(C.byref(q))
    part = torch.empty((nparts, m, n), device=dev)
    q.partials, q.nparts = part.data_ptr(), nparts
    for _ in range(4):
        assert lib.nlam_wgrad(C.byref(q), None) == 0
torch.cuda.synchronize()
