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


@pytest.fixture(params=["auto", "wbf"])
def wide_family(request):
    from neural_lam_amd import _lib as L

    lib = L.load()
    if request.param == "wbf":
        assert lib.nlam_set_tuning(L.TUNE_WBF_MIN_SUPERTILES, 0) == 0
    yield request.param
    assert lib.nlam_set_tuning(L.TUNE_WBF_MIN_SUPERTILES, 192) == 0


@pytest.fixture(params=["factorised", "plain"])
def wide_factorise(request):
    """Widths above 64 run the factorised edge MLP (node-level products + gathered addends) wherever the split-bf16
    super-tile kernels apply; "plain" switches it off so the un-factorised kernels stay covered at full size."""
    from neural_lam_amd import gnn_layers as hl

    old, old_w = hl.FACTORISE_MIN_EDGES_WIDE, hl.FACTORISE_MIN_WORK_WIDE
    hl.FACTORISE_MIN_WORK_WIDE = 0
    hl.FACTORISE_MIN_WIDTH_WIDE = 0
    if request.param == "plain":
        hl.FACTORISE_MIN_EDGES_WIDE = 1 << 30
    yield request.param
    hl.FACTORISE_MIN_EDGES_WIDE, hl.FACTORISE_MIN_WORK_WIDE = old, old_w
    hl.FACTORISE_MIN_WIDTH_WIDE = 256


LAYERS = [
    # (edge set, d, class, update_edges)
    ("m2g", 64, "InteractionNet", False),
    ("g2m", 64, "InteractionNet", False),
    ("m2m", 64, "InteractionNet", True),
    ("m2g", 64, "PropagationNet", True),
    ("m2g", 128, "InteractionNet", False),
    ("m2m", 128, "InteractionNet", True),
    ("g2m", 128, "PropagationNet", False),
    ("m2g", 256, "InteractionNet", False),
    ("g2m", 256, "InteractionNet", False),
    ("m2m", 256, "InteractionNet", True),
]


@pytest.mark.parametrize("which,d,cls_name,update_edges", LAYERS)
def test_meps_layer_matches_oracle(dev, meps_raw, which, d, cls_name, update_edges, wide_family, wide_factorise):
    if d <= 64 and (wide_family == "wbf" or wide_factorise == "plain"):
        pytest.skip("one kernel family at d <= 64")
    if wide_family == "wbf" and wide_factorise == "plain":
        pytest.skip("covered by the auto family: both take the super-tile kernels at this size")
    from neural_lam_amd import gnn_layers as hl
    from oracle import gnn_layers as og

    ei = meps_raw[f"{which}_edge_index"] if which != "m2m" else meps_raw["m2m_edge_index"][0]
    ns, nr, E = int(ei[0].max()) + 1, int(ei[1].max()) + 1, ei.shape[1]
    B = 2
    torch.manual_seed(7)
    ref = getattr(og, cls_name)(ei, d, update_edges=update_edges)
    net = getattr(hl, cls_name)(ei, d, update_edges=update_edges)
    net.load_state_dict(ref.state_dict(), strict=True)
    net.to(dev)
    send, rec, edge = torch.randn(B, ns, d), torch.randn(B, nr, d), torch.randn(B, E, d)
    srg, rrg, erg = (t.clone().requires_grad_() for t in (send, rec, edge))
    r_out = ref(srg, rrg, erg)
    r_outs = r_out if isinstance(r_out, tuple) else (r_out,)
    cots = [torch.randn_like(o) for o in r_outs]
    sum((o * c).sum() for o, c in zip(r_outs, cots)).backward()

    sg, rg, eg = (t.to(dev).requires_grad_() for t in (send, rec, edge))
    h_out = net(sg, rg, eg)
    h_outs = h_out if isinstance(h_out, tuple) else (h_out,)
    assert len(h_outs) == len(r_outs)
    for o, r in zip(h_outs, r_outs):
        assert rel_err(o.cpu(), r) < TOL
        assert row_rel_err(o.cpu(), r) < 10 * TOL   # per-row: rows are LayerNorm outputs + residual, O(1) each
    sum((o * c.to(dev)).sum() for o, c in zip(h_outs, cots)).backward()
    assert rel_err(sg.grad.cpu(), srg.grad) < TOL
    assert rel_err(rg.grad.cpu(), rrg.grad) < TOL
    assert rel_err(eg.grad.cpu(), erg.grad) < TOL
    ref_grads = dict(ref.named_parameters())
    for k, p in net.named_parameters():
        assert rel_err(p.grad.cpu(), ref_grads[k].grad) < TOL, k


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
    return float(h_loss), float(o_loss)


def test_cfg2_training_step_matches_oracle_at_bench_size(dev):
    """BASELINE configs[1] exactly as bench.py runs it."""
    _model_parity(dev, "cfg2")


def test_cfg4_hilam_d128_training_step_matches_oracle_at_bench_size(dev):
    """BASELINE configs[3]: Hi-LAM, 3 levels, d = 128, full MEPS size (46 layer calls on 544 ... 255 136-edge sets)."""
    _model_parity(dev, "cfg4")


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
