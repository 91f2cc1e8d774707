"""Errors of the HIP path against the CPU oracle at bench size for every (forward, backward) matrix-mode pair.

    python tools/r5/parity_modes.py cfg2 [cfg4 ...]  > profiles/round5/parity_by_matmul_mode.log

Same quantities, same norms as tests/test_full_size_parity.py::_model_parity (prediction and loss: max-norm relative, bar
1e-4; every parameter gradient: max-norm relative, bar 1e-4, and element-relative row by row -- scaled_row_rel_err, bar
1e-3), printed instead of asserted, so that a mode can be judged by its margin."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import bench  # noqa: E402
from conftest import rel_err  # noqa: E402
from test_full_size_parity import scaled_row_rel_err  # noqa: E402

from neural_lam_amd import ops  # noqa: E402
from oracle import models as om  # noqa: E402

dev = torch.device("cuda:0")
PAIRS = [("bf16x3", None), ("bf16x2", None), ("bf16x2", "bf16x3"), ("bf16x3", "bf16x2"), ("f32", None)]
for name in sys.argv[1:] or ["cfg2"]:
    cfg = bench.CONFIGS[name]
    ds, _, _, o_fc, _, batch_cpu = bench.build(cfg, torch.device("cpu"), oracle=True)
    pvs, mask = om.per_var_std_uniform(ds), om.interior_mask_bool(ds)
    torch.set_num_threads(32)
    o_pred, o_loss = om.training_loss(o_fc, om.standardize_batch(ds, *batch_cpu), pvs, mask)
    o_loss.backward()
    o_params = dict(o_fc.named_parameters())
    for fwd, bwd in PAIRS:
        ops.set_matmul_mode(fwd)
        ops.MATMUL_MODE_BWD = bwd
        _, _, _, h_fc, step, batch = bench.build(cfg, dev)
        h_pred, h_loss = step(*batch)
        h_loss.backward()
        torch.cuda.synchronize()
        worst_max, worst_row = (0.0, ""), (0.0, "")
        for k, p in h_fc.named_parameters():
            g = o_params[k].grad
            e1 = float((p.grad.cpu() - g).abs().max()) / max(float(g.abs().max()), 1e-6)
            e2 = scaled_row_rel_err(p.grad.cpu(), g)
            if e1 > worst_max[0]:
                worst_max = (e1, k)
            if e2 > worst_row[0]:
                worst_row = (e2, k)
        print(f"{name} fwd={fwd} bwd={bwd or fwd}: pred {rel_err(h_pred.cpu(), o_pred):.3e}  loss {abs(float(h_loss) - float(o_loss)) / abs(float(o_loss)):.3e}  "
              f"grad max-norm {worst_max[0]:.3e} ({worst_max[1]})  grad row-relative {worst_row[0]:.3e} ({worst_row[1]})  [bars 1e-4 / 1e-4 / 1e-4 / 1e-3]",
              flush=True)
        del h_fc, step, batch, h_pred, h_loss
