"""Graph-EFM step predictors (neural_lam_amd.graph_efm) against golden vectors produced by the reference's own
models/step_predictors/graph/graph_efm.py (tests/golden/make_golden.py::efm_case): reference state dicts load strictly;
with the prior noise the reference drew, the latent sample, the predicted mean / std, the variational encoder's
distribution and every parameter gradient match within the fp32 tolerance.
"""
import pytest
import torch

from conftest import graph_from_case, load_golden, rel_err

TOL = 1e-4
CASES = ["efm_hi_81x30", "efm_ms_30x27", "efm_hi_constprior_81x30"]


def _build(case, tmp_path):
    from neural_lam_amd import graph as G
    from neural_lam_amd import graph_efm
    from neural_lam_amd.datastore import SyntheticDatastore

    ds = SyntheticDatastore(root_path=tmp_path, **case["ds_kwargs"])
    G.save_graph(tmp_path / "graph" / "g", graph_from_case(case, "ref_graph_raw"))
    model = getattr(graph_efm, case["cls"])(ds, graph_name="g", **case["model_kwargs"])
    return ds, model


@pytest.mark.parametrize("name", CASES)
def test_reference_state_dict_loads_strictly(name, tmp_path):
    case = load_golden(name)
    _, model = _build(case, tmp_path)
    r = model.load_state_dict(case["state_dict"], strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    assert {k for k, _ in model.named_parameters()} == set(case["ref_grads"])
    assert (case["cls"] == "GraphEFM") == model.hierarchical


def test_graph_type_is_checked(tmp_path):
    """graph_efm.py:667-687, :938-958: the hierarchical model refuses a flat graph and vice versa."""
    from neural_lam_amd import graph as G
    from neural_lam_amd import graph_efm
    from neural_lam_amd.datastore import SyntheticDatastore

    hi, flat = load_golden("efm_hi_81x30"), load_golden("efm_ms_30x27")
    ds = SyntheticDatastore(root_path=tmp_path, **hi["ds_kwargs"])
    G.save_graph(tmp_path / "graph" / "hi", graph_from_case(hi, "ref_graph_raw"))
    with pytest.raises(ValueError, match="requires a flat mesh graph"):
        graph_efm.GraphEFMMultiScale(ds, graph_name="hi", hidden_dim=8)
    ds2 = SyntheticDatastore(root_path=tmp_path / "b", **flat["ds_kwargs"])
    G.save_graph(tmp_path / "b" / "graph" / "ms", graph_from_case(flat, "ref_graph_raw"))
    with pytest.raises(ValueError, match="requires a hierarchical mesh graph"):
        graph_efm.GraphEFM(ds2, graph_name="ms", hidden_dim=8)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_graph_efm_matches_reference_golden(name, tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    case = load_golden(name)
    _, model = _build(case, tmp_path)
    model.load_state_dict(case["state_dict"], strict=True)
    model.to(dev)
    prev, prev_prev, forcing, noise = (case[k].to(dev) for k in ("prev", "prev_prev", "forcing", "noise"))
    seen = {}
    model.decoder.register_forward_pre_hook(lambda mod, args: seen.__setitem__("latent", args[1].detach()))
    pred_mean, pred_std = model(prev, prev_prev, forcing, latent_noise=noise)
    assert rel_err(seen["latent"].cpu(), case["ref_latent"]) < TOL
    assert rel_err(pred_mean.cpu(), case["ref_pred_mean"]) < TOL
    assert (pred_std is None) == (case["ref_pred_std"] is None)
    loss = (pred_mean * case["cotangents"]["mean"].to(dev)).sum()
    if pred_std is not None:
        assert rel_err(pred_std.cpu(), case["ref_pred_std"]) < TOL
        loss = loss + (pred_std * case["cotangents"]["std"].to(dev)).sum()
    loss.backward()
    for k, p in model.named_parameters():
        ref = case["ref_grads"][k]
        if ref is None:   # off forward's path (the variational encoder, its grid embedder)
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        else:
            assert p.grad is not None and rel_err(p.grad.cpu(), ref) < TOL, k
    # the variational encoder on the grid embedding that includes the target state (embedd_grid_with_target, :301-343)
    with torch.no_grad():
        grid_emb, graph_emb = model.embedd_grid_and_graph(prev, prev_prev, forcing)
        enc = model.encoder(model.embedd_grid_with_target(prev, prev_prev, forcing, prev + 0.1), graph_emb=graph_emb)
    assert rel_err(enc.mean.cpu(), case["ref_enc_mean"]) < TOL and rel_err(enc.stddev.cpu(), case["ref_enc_std"]) < TOL
    # without handed-over noise the model samples like the reference (rsample): right shape, finite, different draws differ
    with torch.no_grad():
        a, _ = model(prev, prev_prev, forcing)
        b, _ = model(prev, prev_prev, forcing)
    assert a.shape == pred_mean.shape and bool(torch.isfinite(a).all()) and not torch.equal(a, b)


@pytest.mark.gpu
def test_graph_efm_rolls_out_inside_the_forecaster(tmp_path):
    """ARForecaster (autoregressive.py:63-149) drives the EFM predictor like any other step predictor; the static embeddings
    are computed once for the rollout (static_cache) and give the same result as recomputing them every step."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from neural_lam_amd import models as hm

    dev = torch.device("cuda:0")
    case = load_golden("efm_hi_constprior_81x30")   # constant N(0, 1) prior
    ds, model = _build(case, tmp_path)
    model.load_state_dict(case["state_dict"], strict=True)
    fc = hm.ARForecaster(model, ds).to(dev)
    N = ds.num_grid_points
    g = torch.Generator().manual_seed(0)
    init, target, forcing = torch.randn(1, 2, N, 5, generator=g).to(dev), torch.randn(1, 3, N, 5, generator=g).to(dev), torch.randn(1, 3, N, 6, generator=g).to(dev)
    torch.manual_seed(5)
    with torch.no_grad():
        pred, std = fc(init, forcing, target)
    assert pred.shape == (1, 3, N, 5) and std is None and bool(torch.isfinite(pred).all())
    # same noise, embeddings recomputed per step
    torch.manual_seed(5)
    with torch.no_grad():
        prev_prev, prev = init[:, 0], init[:, 1]
        for t in range(3):
            p, _ = model(prev, prev_prev, forcing[:, t])
            new = fc.boundary_mask * target[:, t] + fc.interior_mask * p
            assert rel_err(pred[:, t], new) < 1e-6
            prev_prev, prev = prev, new


def test_construction_variants_and_no_cpu_fallback(tmp_path):
    """CPU: prior kinds (learnable graph encoder / constant N(0, 1)), latent sizes, predicted-std output width; a CPU
    forward raises instead of falling back to anything."""
    from neural_lam_amd import graph as G
    from neural_lam_amd import graph_efm, latent
    from neural_lam_amd.datastore import SyntheticDatastore

    ds = SyntheticDatastore(30, 27, 5, 2, 1, root_path=tmp_path, boundary="random", seed=1)
    G.save_graph(tmp_path / "graph" / "multiscale", G.create_regular_grid_graph(ds.get_xy("state")))
    m = graph_efm.GraphEFMMultiScale(ds, hidden_dim=16, prior_m2m_layers=1, encoder_m2m_layers=1, decoder_m2m_layers=1)
    assert isinstance(m.prior_model, latent.GraphLatentEncoder) and m.latent_dim == 16 and m.latent_spatial_dim == 81
    assert m.encoder.output_dist == "diagonal" and m.prior_model.output_dist == "isotropic"
    m2 = graph_efm.GraphEFMMultiScale(ds, hidden_dim=16, learn_prior=False, latent_dim=4, output_std=True)
    assert isinstance(m2.prior_model, latent.ConstantLatentEncoder) and m2.latent_dim == 4 and m2.grid_output_dim == 10
    assert m2.decoder.param_map[-1].out_features == 10
    N = ds.num_grid_points
    with pytest.raises(RuntimeError, match="MI355X only"):
        m(torch.zeros(1, N, 5), torch.zeros(1, N, 5), torch.zeros(1, N, 6))
