"""HIP path vs the CPU oracle at the sizes bench.py measures ("what's weak" 1-3).

The golden / seeded-oracle cases of test_hip_parity.py stop at ~2 000 edges, where a wave of the pipelined forward
never takes a second tile and a wide workgroup never a second super tile.  Here the comparison runs on the real
MEPS-shaped graph (238 x 268 grid: m2g 255 136, g2m ~79-100 k, m2m 57 616 edges):

  * one InteractionNet / PropagationNet layer, forward + backward, on each edge set at d = 64 / 128 / 256 with B = 2
    (>= 2 tiles per wave, multi-super-tile workgroups), through both wide kernel families;
  * the cfg2 training step exactly as ``bench.py::build`` makes it (seed 42 weights, seed 123 batch): one-step
    output, rollout prediction, loss and every parameter gradient;
  * the cfg4 Hi-LAM (d = 128, 3 levels) training step at full size: the mid-size launch dispatch of the wide kernels;
  * ``on_after_batch_transfer`` (models/module.py:326-367) with non-trivial statistics.

Reference lines: gnn_layers.py:110-189, models/step_predictors/graph/base.py:228-344, hi_lam.py:167-376.
Tolerance: fp32, max|a-b| / max|b| <= 1e-4, plus a per-row check on LayerNorm outputs.
"""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from neural_lam_amd import _lib

    _lib.load()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def meps_raw():
    from neural_lam_amd import graph as G

    return G.create_regular_grid_graph(G.regular_grid_xy(238, 268))


def row_rel_err(a, b):
    """max over rows of max|a-b| / max|b| of that row: pins small-magnitude rows a global max-norm would hide."""
    a, b = a.detach().reshape(-1, a.shape[-1]), b.detach().reshape(-1, b.shape[-1])
    denom = b.abs().amax(dim=-1).clamp(min=1e-3)
    return float(((a - b).abs().amax(dim=-1) / denom).max())


def scaled_row_rel_err(a, b, floor=1e-2):
    """Element-relative check of a parameter gradient: per row, max|a-b| / max(max|b| of the row, floor * max|b| of the
    tensor).  The global max-norm (conftest.rel_err) lets a row whose gradient is 100x smaller than the largest one be wrong in
    every digit; the floor keeps rows that are numerically zero from dividing by noise.  Vectors count as one row."""
    a, b = a.detach().double(), b.detach().double()
    if a.dim() < 2:
        a, b = a.reshape(1, -1), b.reshape(1, -1)
    a, b = a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1])
    row = b.abs().amax(dim=-1)
    denom = torch.maximum(row, floor * row.max()).clamp(min=1e-30)
    return float(((a - b).abs().amax(dim=-1) / denom).max())


class _Tuning:
    """Force a wide kernel family / the factorised or plain edge MLP for the duration of a block."""

    def __init__(self, family, factorise):
        self.family, self.factorise = family, factorise

    def __enter__(self):
        from neural_lam_amd import _lib as L
        from neural_lam_amd import gnn_layers as hl

        self.hl, self.lib, self.L = hl, L.load(), L
        self.old = (hl.FACTORISE_MIN_EDGES_WIDE, hl.FACTORISE_MIN_WORK_WIDE, hl.FACTORISE_MIN_WIDTH_WIDE)
        hl.FACTORISE_MIN_WORK_WIDE = 0
        hl.FACTORISE_MIN_WIDTH_WIDE = 0
        if self.factorise == "plain":
            hl.FACTORISE_MIN_EDGES_WIDE = 1 << 30
        if self.family == "wbf":
            assert self.lib.nlam_set_tuning(L.TUNE_WBF_MIN_SUPERTILES, 0) == 0
        return self

    def __exit__(self, *exc):
        hl = self.hl
        hl.FACTORISE_MIN_EDGES_WIDE, hl.FACTORISE_MIN_WORK_WIDE, hl.FACTORISE_MIN_WIDTH_WIDE = self.old
        assert self.lib.nlam_set_tuning(self.L.TUNE_WBF_MIN_SUPERTILES, 192) == 0


class _Threads:
    """torch's CPU scatter / index_add paths degrade when oversubscribed on a many-core host (bench.py's probe): the big
    oracle runs use a bounded thread count."""

    def __init__(self, n=32):
        self.n = n

    def __enter__(self):
        import os

        self.old = torch.get_num_threads()
        torch.set_num_threads(max(1, min(self.n, os.cpu_count() or 1)))

    def __exit__(self, *exc):
        torch.set_num_threads(self.old)


LAYERS = [
    # (edge set, d, class, update_edges, batch)
    ("m2g", 64, "InteractionNet", False, 2),
    ("g2m", 64, "InteractionNet", False, 2),
    ("m2m", 64, "InteractionNet", True, 2),
    ("m2g", 64, "PropagationNet", True, 2),
    ("m2g", 128, "InteractionNet", False, 2),
    ("m2m", 128, "InteractionNet", True, 2),
    ("g2m", 128, "PropagationNet", False, 2),
    ("m2g", 256, "InteractionNet", False, 2),
    ("g2m", 256, "InteractionNet", False, 2),
    ("m2m", 256, "InteractionNet", True, 2),
    # cfg5 width (BASELINE configs[4]) on the real edge sets, fp32 matrix mode (bf16x3), one sample
    ("m2g", 512, "InteractionNet", False, 1),
    ("m2m", 512, "InteractionNet", True, 1),
]


def _variants(d):
    """(kernel family, factorised?) combinations a width can take: one family at d <= 64; above, the launch-size dispatch
    ("auto"), the forced super-tile family ("wbf"), each with the factorised edge MLP, and the plain kernels."""
    if d <= 64:
        return [("auto", "factorised")]
    return [("auto", "factorised"), ("auto", "plain"), ("wbf", "factorised")]


@pytest.mark.parametrize("which,d,cls_name,update_edges,B", LAYERS)
def test_meps_layer_matches_oracle(dev, meps_raw, which, d, cls_name, update_edges, B):
    """One oracle run per layer, every kernel family / factorisation variant of the HIP path against it."""
    from neural_lam_amd import gnn_layers as hl
    from oracle import gnn_layers as og

    ei = meps_raw[f"{which}_edge_index"] if which != "m2m" else meps_raw["m2m_edge_index"][0]
    ns, nr, E = int(ei[0].max()) + 1, int(ei[1].max()) + 1, ei.shape[1]
    torch.manual_seed(7)
    ref = getattr(og, cls_name)(ei, d, update_edges=update_edges)
    send, rec, edge = torch.randn(B, ns, d), torch.randn(B, nr, d), torch.randn(B, E, d)
    srg, rrg, erg = (t.clone().requires_grad_() for t in (send, rec, edge))
    with _Threads():
        r_out = ref(srg, rrg, erg)
        r_outs = r_out if isinstance(r_out, tuple) else (r_out,)
        cots = [torch.randn_like(o) for o in r_outs]
        sum((o * c).sum() for o, c in zip(r_outs, cots)).backward()
    r_outs = [o.detach() for o in r_outs]
    ref_grads = {k: p.grad for k, p in ref.named_parameters()}

    for family, factorise in _variants(d):
        tag = f"{family}/{factorise}"
        with _Tuning(family, factorise):
            net = getattr(hl, cls_name)(ei, d, update_edges=update_edges)
            net.load_state_dict(ref.state_dict(), strict=True)
            net.to(dev)
            sg, rg, eg = (t.to(dev).requires_grad_() for t in (send, rec, edge))
            h_out = net(sg, rg, eg)
            h_outs = h_out if isinstance(h_out, tuple) else (h_out,)
            assert len(h_outs) == len(r_outs)
            for o, r in zip(h_outs, r_outs):
                assert rel_err(o.cpu(), r) < TOL, tag
                assert row_rel_err(o.cpu(), r) < 10 * TOL, tag   # per-row: rows are LayerNorm outputs + residual, O(1) each
            sum((o * c.to(dev)).sum() for o, c in zip(h_outs, cots)).backward()
            assert rel_err(sg.grad.cpu(), srg.grad) < TOL, tag
            assert rel_err(rg.grad.cpu(), rrg.grad) < TOL, tag
            assert rel_err(eg.grad.cpu(), erg.grad) < TOL, tag
            for k, p in net.named_parameters():
                assert rel_err(p.grad.cpu(), ref_grads[k]) < TOL, (tag, k)
            del net, sg, rg, eg, h_out, h_outs


def _model_parity(dev, cfg_name, check_one_step=True):
    import bench
    from oracle import models as om

    cfg = bench.CONFIGS[cfg_name]
    ds, _, _, o_fc, _, batch_cpu = bench.build(cfg, torch.device("cpu"), oracle=True)
    _, _, _, h_fc, step, batch = bench.build(cfg, dev)
    # bench.py seeds both builds with 42: the two stacks must come out with identical weights
    o_sd = o_fc.state_dict()
    for k, v in h_fc.state_dict().items():
        assert torch.equal(v.cpu(), o_sd[k]), f"seed-42 init differs between the oracle and the HIP model: {k}"
    pvs, mask = om.per_var_std_uniform(ds), om.interior_mask_bool(ds)
    o_pred, o_loss = om.training_loss(o_fc, batch_cpu, pvs, mask)
    o_loss.backward()
    if check_one_step:
        with torch.no_grad():
            o_one, _ = o_fc.predictor(batch_cpu[0][:, 1], batch_cpu[0][:, 0], batch_cpu[2][:, 0])
            h_one, _ = h_fc.predictor(batch[0][:, 1], batch[0][:, 0], batch[2][:, 0])
        assert rel_err(h_one.cpu(), o_one) < TOL
        assert row_rel_err(h_one.cpu(), o_one) < 10 * TOL
    h_pred, h_loss = step(*batch)
    assert rel_err(h_pred.cpu(), o_pred) < TOL
    assert abs(float(h_loss) - float(o_loss)) < TOL * abs(float(o_loss))
    h_loss.backward()
    o_params = dict(o_fc.named_parameters())
    for k, p in h_fc.named_parameters():
        assert p.grad is not None, k
        g = o_params[k].grad
        assert float((p.grad.cpu() - g).abs().max()) < TOL * max(float(g.abs().max()), 1e-6), k
        assert scaled_row_rel_err(p.grad.cpu(), g) < 10 * TOL, k   # element-relative, row by row
    return float(h_loss), float(o_loss)


def test_cfg2_training_step_matches_oracle_at_bench_size(dev):
    """BASELINE configs[1] exactly as bench.py runs it."""
    _model_parity(dev, "cfg2")


def test_cfg4_hilam_d128_training_step_matches_oracle_at_bench_size(dev):
    """BASELINE configs[3]: Hi-LAM, 3 levels, d = 128, full MEPS size (46 layer calls on 544 ... 255 136-edge sets)."""
    _model_parity(dev, "cfg4")


def test_cfg4p_hilam_parallel_d128_training_step_matches_oracle_at_bench_size(dev):
    """HiLAMParallel (hi_lam_parallel.py:145-218) at the benchmarked shape (bench.py cfg4p: d = 128, full MEPS size): the
    chunked edge / node MLPs (gnn_layers.py:274-324) ride the WIDE kernels on row windows here, which the d = 8 goldens
    never reach."""
    _model_parity(dev, "cfg4p")


def _oracle_step(cfg, T=None):
    """Oracle prediction, loss and parameter gradients of bench.py's workload ``cfg`` (optionally with fewer AR steps)."""
    import bench
    from oracle import models as om

    ds, _, _, o_fc, _, batch_cpu = bench.build(cfg, torch.device("cpu"), oracle=True)
    if T is not None:
        batch_cpu = (batch_cpu[0], batch_cpu[1][:, :T].contiguous(), batch_cpu[2][:, :T].contiguous())
    pvs, mask = om.per_var_std_uniform(ds), om.interior_mask_bool(ds)
    with _Threads():
        o_pred, o_loss = om.training_loss(o_fc, om.standardize_batch(ds, *batch_cpu), pvs, mask)
        o_loss.backward()
    grads = {k: p.grad.clone() for k, p in o_fc.named_parameters()}
    sd = {k: v.clone() for k, v in o_fc.state_dict().items()}
    return o_pred.detach(), float(o_loss), grads, sd


def test_cfg3_rollout_trainer_step_matches_oracle_at_bench_size(dev):
    """BASELINE configs[2] as bench.py runs it on one GPU: GraphLAM d = 256, 8 processor layers, **ar_steps 4**, full
    MEPS size -- the rollout case in which every parameter's gradient is a read-modify-write accumulation over four
    back-propagated AR steps (autoregressive.py:113-149).  The product's trainer step (direct gradient accumulation into
    the flat buffer) must give the oracle's loss and every oracle parameter gradient in all four launch modes: eager /
    HIP-graph replay x weight-gradient side streams on / off.  lr = 0 keeps the weights at their seed-42 values, so the
    four modes and the oracle see the same model."""
    import gc

    import bench
    from neural_lam_amd.trainer import Trainer

    cfg = bench.CONFIGS["cfg3"]
    o_pred, o_loss, o_grads, o_sd = _oracle_step(cfg)
    _, _, _, h_fc, step, batch = bench.build(cfg, dev)
    for k, v in h_fc.state_dict().items():
        assert torch.equal(v.cpu(), o_sd[k]), f"seed-42 init differs between the oracle and the HIP model: {k}"
    with torch.no_grad():
        h_pred, h_loss0 = step(*batch)
    assert rel_err(h_pred.cpu(), o_pred) < TOL
    assert abs(float(h_loss0) - o_loss) < TOL * abs(o_loss)
    del h_pred
    for use_graph in (False, True):
        for overlap in (True, False):
            tag = f"graph={use_graph} overlap_wgrad={overlap}"
            tr = Trainer(step, lr=0.0, use_graph=use_graph, overlap_wgrad=overlap)
            for it in range(2):   # the second step re-uses the zeroed flat buffer / replays the captured graph
                loss = tr.step(*batch)
                torch.cuda.synchronize()
                assert abs(float(loss) - o_loss) < TOL * abs(o_loss), (tag, it)
                for k, p in h_fc.named_parameters():
                    g = o_grads[k]
                    assert p.grad is not None, (tag, k)
                    assert float((p.grad.cpu() - g).abs().max()) < TOL * max(float(g.abs().max()), 1e-6), (tag, it, k)
                    if it == 0:
                        assert scaled_row_rel_err(p.grad.cpu(), g) < 10 * TOL, (tag, k)
            if use_graph:
                assert tr._graph is not None, "the HIP-graph capture fell back to eager launches"
            del tr
            gc.collect()
            torch.cuda.synchronize()


CFG5_AUTOCAST_TOL = 3e-2   # bf16 operands (what autocast does to the reference's nn.Linear), fp32 accumulate / LN / aggregation


def test_cfg5_two_step_rollout_under_bf16_autocast_at_bench_size(dev):
    """BASELINE configs[4]: GraphLAM d = 512, 8 processor layers, bf16 mixed precision (Lightning --precision bf16-mixed =
    torch.autocast, train_model.py:163-168), full MEPS size, two AR steps of the eight (the oracle needs ~40 GB of host
    memory per pair).  Against the fp32 oracle: prediction and loss within 3e-2; every parameter gradient finite and within
    bf16-operand distance of the oracle's (relative L2 error <= 1e-1, cosine >= 0.99)."""
    import bench
    from neural_lam_amd.trainer import Trainer

    cfg = dict(bench.CONFIGS["cfg5"])
    T = 2
    o_pred, o_loss, o_grads, o_sd = _oracle_step(cfg, T=T)
    _, _, _, h_fc, step, batch = bench.build(cfg, dev)
    batch = (batch[0], batch[1][:, :T].contiguous(), batch[2][:, :T].contiguous())
    for k, v in h_fc.state_dict().items():
        assert torch.equal(v.cpu(), o_sd[k]), k
    with torch.autocast("cuda", dtype=torch.bfloat16):
        with torch.no_grad():
            h_pred, h_loss0 = step(*batch)
        assert rel_err(h_pred.float().cpu(), o_pred) < CFG5_AUTOCAST_TOL
        assert abs(float(h_loss0) - o_loss) < CFG5_AUTOCAST_TOL * abs(o_loss)
        del h_pred
        tr = Trainer(step, lr=0.0, use_graph=True)
        loss = tr.step(*batch)
        torch.cuda.synchronize()
    assert abs(float(loss) - o_loss) < CFG5_AUTOCAST_TOL * abs(o_loss)
    worst_l2, worst_cos = 0.0, 1.0
    for k, p in h_fc.named_parameters():
        g, h = o_grads[k].double().reshape(-1), p.grad.cpu().double().reshape(-1)
        assert bool(torch.isfinite(h).all()), k
        l2 = float((h - g).norm() / g.norm().clamp(min=1e-30))
        cos = float((h @ g) / (h.norm() * g.norm()).clamp(min=1e-30))
        worst_l2, worst_cos = max(worst_l2, l2), min(worst_cos, cos)
        assert l2 < 1e-1 and cos > 0.99, (k, l2, cos)
    print(f"cfg5 (T={T}) bf16 autocast vs fp32 oracle: worst gradient rel-L2 {worst_l2:.3e}, worst cosine {worst_cos:.6f}")


def test_cfg5_full_rollout_under_bf16_autocast_against_the_oracle_on_the_gpu(dev):
    """BASELINE configs[4] at its FULL ar_steps 8: the oracle restatement needs ~20 GB of saved
    activations per AR step at d = 512, so it runs on the GPU here (fp32, stock PyTorch ops -- test infrastructure, as in
    bench.py's gpu_reference_equivalent), its loss and gradients are moved to the host, and the product's captured trainer
    step under torch.autocast(bfloat16) is compared with them.  The oracle's own rounding error is bounded first: one
    d = 512 mesh layer in fp64 against the same layer in fp32.
    Tolerances (bf16 operands against an fp32 reference, eight chained AR steps): prediction 3e-2, loss 1e-2, parameter
    gradients relative L2 <= 6e-2 and cosine >= 0.998; the measured values are printed (round 4, fp32 storage: prediction
    7.0e-3, loss 6.3e-4, worst gradient rel-L2 1.3e-2, worst cosine 0.99993, oracle fp32-vs-fp64 8.9e-7)."""
    import gc

    import bench
    from neural_lam_amd import graph as G
    from oracle import gnn_layers as og

    # ---- the oracle's own error: fp32 against fp64 on one mesh layer of this width
    raw = G.create_regular_grid_graph(G.regular_grid_xy(238, 268))
    ei = raw["m2m_edge_index"][0]
    torch.manual_seed(3)
    l32 = og.InteractionNet(ei, 512)
    n, E = int(ei.max()) + 1, ei.shape[1]
    x, e = torch.randn(1, n, 512), torch.randn(1, E, 512)
    l64 = og.InteractionNet(ei, 512).double()
    l64.load_state_dict({k: v.double() for k, v in l32.state_dict().items()})
    with torch.no_grad():
        r32, e32 = l32.to(dev)(x.to(dev), x.to(dev), e.to(dev))
        r64, e64 = l64.to(dev)(x.double().to(dev), x.double().to(dev), e.double().to(dev))
    oracle_err = max(rel_err(r32.double().cpu(), r64.cpu()), rel_err(e32.double().cpu(), e64.cpu()))
    assert oracle_err < 1e-5, oracle_err
    del l32, l64, r32, e32, r64, e64
    gc.collect()
    torch.cuda.empty_cache()

    # ---- oracle, full rollout, on the device: fp32 (the reference) and under torch.autocast (what --precision bf16-mixed makes of it)
    cfg = dict(bench.CONFIGS["cfg5"])
    report = _bf16_noise_against_reference_autocast(dev, cfg, "cfg5 (T=8)")
    e_pred, e_loss, worst_l2, worst_cos = report["hip"]["pred"], report["hip"]["loss"], report["hip"]["worst_l2"], report["hip"]["worst_cos"]
    print(f"oracle fp32-vs-fp64 layer error {oracle_err:.2e}")
    assert e_pred < 3e-2 and e_loss < 1e-2
    assert worst_l2 < 6e-2 and worst_cos > 0.998, (worst_l2, worst_cos)


def _gpu_oracle_step(cfg, dev, autocast):
    """Prediction, loss and parameter gradients of the oracle restatement run on the GPU through stock PyTorch ops (test
    infrastructure, as bench.py's gpu_reference_equivalent), in fp32 or inside torch.autocast(bfloat16)."""
    import gc

    import bench
    from oracle import models as om

    ds, _, _, o_fc, _, batch_cpu = bench.build(cfg, torch.device("cpu"), oracle=True)
    sd = {k: v.clone() for k, v in o_fc.state_dict().items()}
    o_fc = o_fc.to(dev)
    pvs, mask = om.per_var_std_uniform(ds).to(dev), om.interior_mask_bool(ds).to(dev)
    o_batch = tuple(b.to(dev) for b in om.standardize_batch(ds, *batch_cpu))
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        o_pred, o_loss_t = om.training_loss(o_fc, o_batch, pvs, mask)
    o_loss_t.backward()
    torch.cuda.synchronize()
    res = {"pred": o_pred.detach().float().cpu(), "loss": float(o_loss_t), "grads": {k: p.grad.detach().float().cpu() for k, p in o_fc.named_parameters()}, "sd": sd}
    del o_fc, o_batch, o_loss_t, o_pred, pvs, mask
    gc.collect()
    torch.cuda.empty_cache()
    return res


def _errors_against(ref, pred, loss, grads):
    """max-norm relative error of the prediction, relative error of the loss, per-parameter relative L2 error and cosine."""
    out = {"pred": rel_err(pred, ref["pred"]), "loss": abs(loss - ref["loss"]) / abs(ref["loss"]), "l2": {}, "cos": {}}
    for k, g in ref["grads"].items():
        g, h = g.double().reshape(-1), grads[k].double().reshape(-1)
        assert bool(torch.isfinite(h).all()), k
        out["l2"][k] = float((h - g).norm() / g.norm().clamp(min=1e-30))
        out["cos"][k] = float((h @ g) / (h.norm() * g.norm()).clamp(min=1e-30))
    out["worst_l2"], out["worst_cos"] = max(out["l2"].values()), min(out["cos"].values())
    return out


BF16_NOISE_FACTOR = 1.5    # HIP-under-autocast may be at most this much noisier than the reference-under-autocast ...
BF16_NOISE_FLOOR = 1e-3    # ... plus this absolute allowance on a relative error (tensors the reference happens to get almost exactly)


def _bf16_noise_against_reference_autocast(dev, cfg, tag):
    """the bf16 tolerances of this file are stated against the fp32 oracle; what matters to a user of
    ``--precision bf16-mixed`` (train_model.py:163-168) is whether this library is NOISIER than the reference's own
    autocast run.  Three runs of bench.py's workload on the GPU -- oracle fp32, oracle inside torch.autocast(bfloat16), the
    product's captured trainer step inside torch.autocast(bfloat16) (bf16 operands AND bf16 saved tensors) -- and per tensor
    err(HIP vs fp32) <= 1.5 x err(reference-autocast vs fp32) + 1e-3."""
    import gc

    import bench
    from neural_lam_amd.trainer import Trainer

    ref = _gpu_oracle_step(cfg, dev, autocast=False)
    amp = _gpu_oracle_step(cfg, dev, autocast=True)
    e_ref = _errors_against(ref, amp["pred"], amp["loss"], amp["grads"])
    del amp
    _, _, _, h_fc, step, batch = bench.build(cfg, dev)
    for k, v in h_fc.state_dict().items():
        assert torch.equal(v.cpu(), ref["sd"][k]), k
    with torch.autocast("cuda", dtype=torch.bfloat16):
        with torch.no_grad():
            h_pred, _ = step(*batch)
        h_pred = h_pred.float().cpu()
        tr = Trainer(step, lr=0.0, use_graph=True)
        loss = tr.step(*batch)
        torch.cuda.synchronize()
    e_hip = _errors_against(ref, h_pred, float(loss), {k: p.grad.detach().float().cpu() for k, p in h_fc.named_parameters()})
    worst_ratio, worst_k = 0.0, ""
    for k in e_hip["l2"]:
        r = e_hip["l2"][k] / max(e_ref["l2"][k], 1e-30)
        if e_hip["l2"][k] > BF16_NOISE_FLOOR and r > worst_ratio:
            worst_ratio, worst_k = r, k
    print(f"{tag} under bf16 autocast, errors against the fp32 oracle (GPU): HIP prediction {e_hip['pred']:.3e} / loss {e_hip['loss']:.3e} / worst gradient "
          f"rel-L2 {e_hip['worst_l2']:.3e} / worst cosine {e_hip['worst_cos']:.6f}; reference-under-autocast {e_ref['pred']:.3e} / {e_ref['loss']:.3e} / "
          f"{e_ref['worst_l2']:.3e} / {e_ref['worst_cos']:.6f}; worst per-tensor ratio HIP / reference {worst_ratio:.2f} ({worst_k})")
    assert e_hip["pred"] <= BF16_NOISE_FACTOR * e_ref["pred"] + BF16_NOISE_FLOOR
    assert e_hip["loss"] <= BF16_NOISE_FACTOR * e_ref["loss"] + BF16_NOISE_FLOOR
    for k in e_hip["l2"]:
        assert e_hip["l2"][k] <= BF16_NOISE_FACTOR * e_ref["l2"][k] + BF16_NOISE_FLOOR, (k, e_hip["l2"][k], e_ref["l2"][k])
    del tr, step, h_fc
    gc.collect()
    torch.cuda.empty_cache()
    return {"hip": e_hip, "ref": e_ref}


def test_hilam_d128_bf16_noise_is_not_above_the_reference_under_autocast(dev):
    """One Hi-LAM (BASELINE configs[3]: d = 128, 3 levels) training step under bf16 autocast: the fp32 one-tile wide kernels of
    the small launches run with bf16 operands, the big edge sets on the split-bf16 super-tile kernels with bf16 saved tensors."""
    import bench

    _bf16_noise_against_reference_autocast(dev, dict(bench.CONFIGS["cfg4"]), "cfg4 (Hi-LAM d = 128)")


def test_cfg1_training_step_matches_oracle_as_configured(dev):
    """BASELINE configs[0] AS CONFIGURED (missing 7): Keisler 1-level mesh on the 64 x 64 dummy grid,
    hidden_dim 16, batch 2, ar_steps 1 -- bench.CONFIGS["cfg1"], the reference's own CPU-runnable case."""
    _model_parity(dev, "cfg1")


def test_cfg2_hip_graph_trainer_step_matches_oracle_adamw(dev):
    """The product's training step (HIP-graph replay + flat fused AdamW, standardisation inside the step) against
    the oracle + torch.optim.AdamW over three optimizer steps at cfg2 size: losses and final weights."""
    import bench
    from neural_lam_amd.trainer import Trainer
    from oracle import models as om

    cfg = bench.CONFIGS["cfg2"]
    ds, _, _, o_fc, _, batch_cpu = bench.build(cfg, torch.device("cpu"), oracle=True)
    _, _, _, h_fc, step, batch = bench.build(cfg, dev)
    pvs, mask = om.per_var_std_uniform(ds), om.interior_mask_bool(ds)
    opt = torch.optim.AdamW(o_fc.parameters(), lr=1e-3, betas=(0.9, 0.95))
    tr = Trainer(step, lr=1e-3, use_graph=True)
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        _, o_loss = om.training_loss(o_fc, om.standardize_batch(ds, *batch_cpu), pvs, mask)
        o_loss.backward()
        opt.step()
        h_loss = tr.step(*batch)
        assert abs(float(h_loss) - float(o_loss)) < TOL * abs(float(o_loss))
    assert tr._graph is not None
    o_sd = o_fc.state_dict()
    for k, v in h_fc.state_dict().items():
        # three Adam steps of lr 1e-3 move a weight by <= 3e-3; compare the weights themselves
        if v.numel():   # the (empty) clamping buffers are part of the state dict too
            assert float((v.cpu() - o_sd[k]).abs().max()) < 2e-4, k


def test_cfg3_segmented_trainer_trajectory_matches_oracle_adamw(dev):
    """The wide kernels under the executor the trainer picks for them (segmented chain graphs + weight-gradient side graphs),
    with the optimizer ON: BASELINE configs[2] (GraphLAM d = 256, 8 processor layers, full MEPS size) with two of its four AR
    steps (the oracle's memory and minutes), three optimizer steps against the oracle + torch.optim.AdamW -- every step's loss and
    the final weights.  The lr = 0 test above proves one gradient; this one proves that replay, gradient zeroing, the weight
    re-pack at the start of each step and the rollout's gradient hand-over stay right while the weights move."""
    import bench
    from neural_lam_amd.trainer import Trainer
    from oracle import models as om

    cfg = bench.CONFIGS["cfg3"]
    T = 2
    ds, _, _, o_fc, _, batch_cpu = bench.build(cfg, torch.device("cpu"), oracle=True)
    _, _, _, h_fc, step, batch = bench.build(cfg, dev)
    batch_cpu = (batch_cpu[0], batch_cpu[1][:, :T].contiguous(), batch_cpu[2][:, :T].contiguous())
    batch = (batch[0], batch[1][:, :T].contiguous(), batch[2][:, :T].contiguous())
    pvs, mask = om.per_var_std_uniform(ds), om.interior_mask_bool(ds)
    opt = torch.optim.AdamW(o_fc.parameters(), lr=1e-3, betas=(0.9, 0.95))
    tr = Trainer(step, lr=1e-3, use_graph=True)
    assert tr.executor == "segments", tr.executor
    with _Threads():
        for it in range(3):
            opt.zero_grad(set_to_none=True)
            _, o_loss = om.training_loss(o_fc, om.standardize_batch(ds, *batch_cpu), pvs, mask)
            o_loss.backward()
            opt.step()
            h_loss = tr.step(*batch)
            assert abs(float(h_loss) - float(o_loss)) < TOL * abs(float(o_loss)), it
    assert tr._graph is not None
    o_sd = o_fc.state_dict()
    for k, v in h_fc.state_dict().items():
        if v.numel():
            assert float((v.cpu() - o_sd[k]).abs().max()) < 2e-4, k


def test_standardize_matches_reference_formula(dev, tmp_path):
    """ForecasterStep.standardize == ForecasterModule.on_after_batch_transfer (models/module.py:326-367): per-variable
    state statistics, forcing statistics tiled feature-major over the window (repeat_interleave, :352-358)."""
    import numpy as np

    from neural_lam_amd import graph as G
    from neural_lam_amd import models as hm
    from neural_lam_amd.datastore import SyntheticDatastore
    from oracle import models as om

    rng = np.random.default_rng(3)
    ds = SyntheticDatastore(30, 27, 5, 2, 1, root_path=tmp_path, boundary="random", seed=1,
                            state_stats={"state_mean": rng.normal(size=5) * 3, "state_std": rng.uniform(0.5, 4.0, size=5)})
    ds._forcing_stats.forcing_mean.values = rng.normal(size=2).astype(np.float32)
    ds._forcing_stats.forcing_std.values = rng.uniform(0.5, 2.0, size=2).astype(np.float32)
    ext = ds.get_xy_extent("state")
    graph = G.normalise_graph(G.create_regular_grid_graph(ds.get_xy("state")), max(ext[1] - ext[0], ext[3] - ext[2]))
    torch.manual_seed(2)
    o_fc = om.ARForecaster(om.GraphLAM(ds, graph, hidden_dim=16, processor_layers=1), ds)
    h_fc = hm.ARForecaster(hm.GraphLAM(ds, graph=graph, hidden_dim=16, processor_layers=1), ds)
    h_fc.load_state_dict(o_fc.state_dict())
    step = hm.ForecasterStep(h_fc, ds).to(dev)
    N = ds.num_grid_points
    init, target, forcing = torch.randn(2, 2, N, 5) * 3 + 1, torch.randn(2, 3, N, 5) * 3 + 1, torch.randn(2, 3, N, 6)
    o_batch = om.standardize_batch(ds, init, target, forcing)
    h_batch = step.standardize(init.to(dev), target.to(dev), forcing.to(dev))
    for h, o in zip(h_batch, o_batch):
        assert rel_err(h.cpu(), o) < 1e-6
    # the window is feature-major: forcing column f * window + w uses the statistics of feature f
    fm, fs = ds._forcing_stats.forcing_mean.values, ds._forcing_stats.forcing_std.values
    col = 1 * 3 + 2
    expect = (forcing[..., col] - float(fm[1])) / float(fs[1])
    assert torch.allclose(h_batch[2][..., col].cpu(), expect, atol=1e-5)
    # and the whole step with standardize=True equals the oracle run on the standardised batch
    _, o_loss = om.training_loss(o_fc, o_batch, om.per_var_std_uniform(ds), om.interior_mask_bool(ds))
    _, h_loss = step(init.to(dev), target.to(dev), forcing.to(dev), standardize=True)
    assert abs(float(h_loss) - float(o_loss)) < TOL * abs(float(o_loss))
