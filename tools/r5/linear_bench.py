"""nlam_linear: the LDS-tiled GEMM of round 5 against the strip kernel of rounds 2-4, isolated launches (20 replayed from a HIP graph)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from neural_lam_amd import _lib as L  # noqa: E402
from neural_lam_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
lib = L.load()


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


print("rows k n mode layout | strip us | gemm us | gemm: TFLOP/s algorithmic, GB/s on 4 rows (k + n) bytes, frac of 8 TB/s")
for rows in (63784, 6561):
    for k, n in ((512, 512), (256, 256), (128, 128)):
        W = torch.randn(n, 3 * k, device=dev) / k ** 0.5
        x = torch.randn(rows, k, device=dev)
        g = torch.randn(rows, n, device=dev)
        out, dx = torch.empty(rows, n, device=dev), torch.empty(rows, k, device=dev)
        for mode in ("bf16", "bf16x3"):
            mm = ops._MM_FLAGS[mode]
            for layout, fn in (("fwd", lambda: ops._linear_launch(x, W.data_ptr() + 4 * k, 3 * k, 1, k, n, out=out, mm_flags=mm)),
                               ("dgrad", lambda: ops._linear_launch(g, W.data_ptr() + 4 * k, 1, 3 * k, n, k, out=dx, mm_flags=mm))):
                res = {}
                for name, v in (("strip", 0), ("gemm", 2)):
                    assert lib.nlam_set_tuning(L.TUNE_LIN_GEMM, v) == 0
                    res[name] = timed(fn)
                assert lib.nlam_set_tuning(L.TUNE_LIN_GEMM, 1) == 0
                t = res["gemm"] * 1e-6
                fl, by = 2.0 * rows * k * n, 4.0 * rows * (k + n)
                print(f"{rows} {k} {n} {mode} {layout} | {res['strip']:.1f} | {res['gemm']:.1f} | {fl / t / 1e12:.1f} TF/s, {by / t / 1e9:.0f} GB/s, {by / t / 8e12:.2f}", flush=True)
