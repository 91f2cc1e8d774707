"""Properties the reference's own tests assert on this path, restated against
both the CPU oracle (runs everywhere) and the HIP product (``gpu`` marker).

  reference test                                         restated here
  tests/test_prediction_model_classes.py:38-73           test_ar_rollout_overwrites_boundary_with_truth
  tests/test_clamping.py:15-292                          test_clamped_state_stays_inside_limits
  tests/test_training.py:31-142                          test_short_training_loop_stays_finite
  tests/test_gnn_layers.py:297-319, 450-502              test_propagation_differs_from_interaction_with_shared_weights,
                                                         test_chunked_mlps_differ_from_unchunked

No reference code runs here; the assertions follow the cited lines, the inputs are synthetic."""
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

BACKENDS = [pytest.param("oracle", id="oracle-cpu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


def _impl(backend):
    if backend == "oracle":
        from oracle import gnn_layers as layers
        from oracle import models

        return layers, models, torch.device("cpu")
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from neural_lam_amd import gnn_layers as layers
    from neural_lam_amd import models

    return layers, models, torch.device("cuda:0")


def _datastore(tmp_path, **kw):
    from neural_lam_amd.datastore import SyntheticDatastore

    args = dict(nx=14, ny=12, num_state=5, num_forcing=2, num_static=1, root_path=tmp_path, boundary="random")
    args.update(kw)
    return SyntheticDatastore(**args)


def _graph(ds, **kw):
    from neural_lam_amd import graph as G

    ext = ds.get_xy_extent("state")
    raw = G.create_regular_grid_graph(ds.get_xy("state"), **kw)
    return G.normalise_graph(raw, max(ext[1] - ext[0], ext[3] - ext[2]))


def _graph_lam(models, ds, graph, **kw):
    if models.__name__.startswith("oracle"):
        return models.GraphLAM(ds, graph, **kw)
    return models.GraphLAM(ds, graph=graph, **kw)


@pytest.mark.parametrize("backend", BACKENDS)
def test_ar_rollout_overwrites_boundary_with_truth(backend, tmp_path):
    """test_prediction_model_classes.py:38-73: with a predictor that returns zeros, one interior node and truth = 5
    everywhere, the rollout is 0 on the interior node and 5 on every boundary node, for all steps."""
    _, models, dev = _impl(backend)
    ds = _datastore(tmp_path)

    class ZeroPredictor(models.StepPredictor):
        def forward(self, prev_state, prev_prev_state, forcing):
            return torch.zeros_like(prev_state), None

    predictor = ZeroPredictor(ds, output_std=False)
    forecaster = models.ARForecaster(predictor, ds).to(dev)
    forecaster.interior_mask = torch.zeros_like(forecaster.interior_mask)
    forecaster.interior_mask[0, 0] = 1
    forecaster.boundary_mask = 1 - forecaster.interior_mask
    B, N, T = 2, ds.num_grid_points, 3
    ns, nf = ds.get_num_data_vars("state"), ds.get_num_data_vars("forcing") * 3
    init = torch.ones(B, 2, N, ns, device=dev)
    forcing = torch.ones(B, T, N, nf, device=dev)
    truth = torch.full((B, T, N, ns), 5.0, device=dev)
    pred, std = forecaster(init, forcing, truth)
    assert pred.shape == (B, T, N, ns) and std is None
    assert torch.all(pred[:, :, 0, :] == 0.0)
    assert torch.all(pred[:, :, 1:, :] == 5.0)


@pytest.mark.parametrize("backend", BACKENDS)
def test_clamped_state_stays_inside_limits(backend, tmp_path):
    """test_clamping.py:15-292: zero delta is the identity (1e-6); 100 steps of a constant delta move unclamped
    variables by exactly 100 and keep the clamped ones inside their limits, in both directions and in physical
    units; a state outside the limits is pulled back inside by a zero delta."""
    _, models, dev = _impl(backend)
    stats = {
        "state_mean": [1.0, -0.5, 0.25, 2.0, 0.0],
        "state_std": [2.0, 0.5, 0.1, 10.0, 1.0],
    }
    ds = _datastore(tmp_path, state_stats=stats)
    names = ds.get_vars_names("state")
    lower = {names[0]: 0.0, names[2]: 0.0}
    upper = {names[2]: 1.0, names[3]: 100.0}
    model = _graph_lam(models, ds, _graph(ds, n_max_levels=1), hidden_dim=4, processor_layers=2,
                       output_clamping_lower=lower, output_clamping_upper=upper).to(dev)
    lu, lo, hi = model.clamp_lower_upper_idx, model.clamp_lower_idx, model.clamp_upper_idx
    assert lu.tolist() == [2] and lo.tolist() == [0] and hi.tolist() == [3]
    free = sorted(set(range(len(names))) - set(lu.tolist()) - set(lo.tolist()) - set(hi.tolist()))

    state0 = torch.zeros(1, 1, len(names), device=dev)
    state0[:, :, lu] = (model.sigmoid_lower_lims + model.sigmoid_upper_lims) / 2
    state0[:, :, lo] = model.softplus_lower_lims + 10
    state0[:, :, hi] = model.softplus_upper_lims - 10
    delta = torch.ones_like(state0)
    delta[:, :, lu] = (model.sigmoid_upper_lims - model.sigmoid_lower_lims) / 3
    delta[:, :, lo] = -5
    delta[:, :, hi] = 5
    zero = torch.zeros_like(state0)

    same = model.get_clamped_new_state(zero, state0)
    assert torch.all((same - state0).abs() < 1e-6)

    def inside(x):
        return (bool(torch.all(model.sigmoid_lower_lims <= x[:, :, lu])) and bool(torch.all(x[:, :, lu] <= model.sigmoid_upper_lims))
                and bool(torch.all(model.softplus_lower_lims <= x[:, :, lo])) and bool(torch.all(x[:, :, hi] <= model.softplus_upper_lims)))

    def physical_inside(x):
        phys = x * model.state_std + model.state_mean
        idx = {n: i for i, n in enumerate(names)}
        return all(float(phys[0, 0, idx[n]]) >= v - 1e-5 for n, v in lower.items()) and all(
            float(phys[0, 0, idx[n]]) <= v + 1e-5 for n, v in upper.items())

    for sign in (1.0, -1.0):
        x = same.clone()
        for _ in range(100):
            x = model.get_clamped_new_state(sign * delta, x)
        assert torch.all((x[:, :, free] - sign * 100).abs() < 1e-4)
        assert inside(x) and physical_inside(x)
        bad = state0 + sign * 5 * delta   # + : every clamped variable leaves its range; - : the two-sided one does
        assert not bool(torch.any((model.sigmoid_lower_lims <= bad[:, :, lu]) & (bad[:, :, lu] <= model.sigmoid_upper_lims)))
        if sign > 0:
            assert bool(torch.all(bad[:, :, hi] > model.softplus_upper_lims)) and bool(torch.all(bad[:, :, lo] < model.softplus_lower_lims))
        pulled = model.get_clamped_new_state(zero, bad)
        assert inside(pulled)


@pytest.mark.parametrize("backend", BACKENDS)
def test_short_training_loop_stays_finite(backend, tmp_path):
    """test_training.py:31-142 without Lightning: a few optimizer steps on the dummy-sized problem under
    torch.autograd.detect_anomaly; the loss and every gradient stay finite and the loss goes down."""
    _, models, dev = _impl(backend)
    ds = _datastore(tmp_path, nx=16, ny=16)
    graph = _graph(ds, n_max_levels=1)
    torch.manual_seed(3)
    predictor = _graph_lam(models, ds, graph, hidden_dim=8, processor_layers=2)
    forecaster = models.ARForecaster(predictor, ds).to(dev)
    B, T, N = 2, 2, ds.num_grid_points
    ns, nf = ds.get_num_data_vars("state"), ds.get_num_data_vars("forcing") * 3
    g = torch.Generator().manual_seed(5)
    init, target, forcing = (torch.randn(*s, generator=g).to(dev) for s in ((B, 2, N, ns), (B, T, N, ns), (B, T, N, nf)))
    if backend == "oracle":
        pvs, mask = models.per_var_std_uniform(ds), models.interior_mask_bool(ds)
        step = lambda: models.training_loss(forecaster, (init, target, forcing), pvs, mask)[1]  # noqa: E731
    else:
        fstep = models.ForecasterStep(forecaster, ds).to(dev)
        step = lambda: fstep(init, target, forcing)[1]  # noqa: E731
    opt = torch.optim.AdamW(forecaster.parameters(), lr=1e-2, betas=(0.9, 0.95))
    losses = []
    with torch.autograd.detect_anomaly():
        for _ in range(6):
            opt.zero_grad(set_to_none=True)
            loss = step()
            loss.backward()
            assert torch.isfinite(loss)
            for name, p in forecaster.named_parameters():
                assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name
            opt.step()
            losses.append(float(loss))
    assert losses[-1] < losses[0]


def _layer_inputs(dev, ns=9, nr=7, e=40, d=8, seed=1):
    g = torch.Generator().manual_seed(seed)
    ei = torch.stack([torch.randint(0, ns, (e,), generator=g), torch.randint(0, nr, (e,), generator=g)])
    ei[1, -1] = nr - 1
    ei[0, -1] = ns - 1
    send, rec, edge = (torch.randn(n, d, generator=g).to(dev) for n in (ns, nr, e))
    return ei, send, rec, edge


@pytest.mark.parametrize("backend", BACKENDS)
def test_propagation_differs_from_interaction_with_shared_weights(backend):
    """test_gnn_layers.py:297-319: the two layer types take each other's state_dict and still give different
    outputs (sender residual in the message, mean aggregation, residual on the aggregate)."""
    layers, _, dev = _impl(backend)
    ei, send, rec, edge = _layer_inputs(dev)
    torch.manual_seed(0)
    inet = layers.InteractionNet(ei, 8).to(dev)
    pnet = layers.PropagationNet(ei, 8).to(dev)
    assert issubclass(layers.PropagationNet, layers.InteractionNet) and pnet.aggr == "mean"
    pnet.load_state_dict(inet.state_dict())
    with torch.no_grad():
        a, b = inet(send, rec, edge), pnet(send, rec, edge)
    assert a[0].shape == b[0].shape and not torch.allclose(a[0], b[0])


@pytest.mark.parametrize("backend", BACKENDS)
def test_chunked_mlps_differ_from_unchunked(backend):
    """test_gnn_layers.py:450-502: edge / aggregation chunk sizes give per-chunk MLPs (``mlps.{k}`` parameters); the
    layer runs, keeps the output shapes and does not coincide with the un-chunked layer."""
    layers, _, dev = _impl(backend)
    ei, send, rec, edge = _layer_inputs(dev)
    torch.manual_seed(0)
    plain = layers.InteractionNet(ei, 8).to(dev)
    torch.manual_seed(0)
    chunked = layers.InteractionNet(ei, 8, edge_chunk_sizes=[15, 25], aggr_chunk_sizes=[3, 4]).to(dev)
    assert any(".mlps.1." in k for k in chunked.state_dict())
    with torch.no_grad():
        a, b = plain(send, rec, edge), chunked(send, rec, edge)
    assert a[0].shape == b[0].shape == (7, 8) and a[1].shape == b[1].shape == (40, 8)
    assert torch.isfinite(b[0]).all() and not torch.allclose(a[0], b[0])
