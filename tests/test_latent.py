"""Graph-EFM latent encoder / decoder (neural_lam_amd.latent) against golden vectors produced by the reference's own
models/latent/{base_encoder,graph_encoder,base_decoder,graph_decoder,hi_graph_encoder,hi_graph_decoder}.py
(tests/golden/make_golden.py::latent_case, ::hi_latent_case).

These modules instantiate their layers through get_gnn_class / make_gnn_seq / make_mlp -- the drop-in surface -- so
this is the "callers that reuse the layers" row of SURVEY.md section 8(f)4: reference state dicts load strictly, and
the distribution parameters, decoder outputs and every gradient match within the fp32 tolerance.
"""
import pytest
import torch

from conftest import load_golden, rel_err

TOL = 1e-4
CASES = ["latent_flat_d64", "latent_flat_d16_prop"]


def _build(case):
    from neural_lam_amd import latent

    enc = latent.GraphLatentEncoder(case["latent_dim"], case["g2m_edge_index"], case["m2m_edge_index"], case["d"], case["m2m_layers"],
                                    hidden_layers=1, g2m_gnn_type=case["g2m_gnn_type"], output_dist=case["output_dist"])
    dec = latent.GraphLatentDecoder(case["g2m_edge_index"], case["m2m_edge_index"], case["m2g_edge_index"], case["d"], case["latent_dim"],
                                    case["num_state"], case["m2m_layers"], hidden_layers=1, g2m_gnn_type=case["g2m_gnn_type"],
                                    m2g_gnn_type=case["m2g_gnn_type"], output_std=True)
    return enc, dec


@pytest.mark.parametrize("name", CASES)
def test_reference_state_dicts_load_strictly(name):
    """CPU: same parameter names / shapes as the reference modules (no compute)."""
    case = load_golden(name)
    enc, dec = _build(case)
    r1 = enc.load_state_dict(case["enc_state_dict"], strict=True)
    r2 = dec.load_state_dict(case["dec_state_dict"], strict=True)
    assert not r1.missing_keys and not r1.unexpected_keys and not r2.missing_keys and not r2.unexpected_keys
    assert any(k.startswith("m2m_gnns.module_0.edge_mlp.0") for k in case["enc_state_dict"])   # pyg.nn.Sequential child names


def test_constant_encoder_and_unknown_distribution():
    from neural_lam_amd import latent

    enc = latent.ConstantLatentEncoder(4, 7, output_dist="diagonal")
    dist = enc(torch.zeros(2, 11, 3))
    assert dist.mean.shape == (2, 7, 4) and float(dist.mean.abs().max()) == 0.0
    assert torch.allclose(dist.stddev, torch.full((2, 7, 4), 1e-4 + float(torch.nn.functional.softplus(torch.zeros(())))))
    with pytest.raises(ValueError):
        latent.ConstantLatentEncoder(4, 7, output_dist="full")


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_latent_stack_matches_reference_golden(name):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    case = load_golden(name)
    enc, dec = _build(case)
    enc.load_state_dict(case["enc_state_dict"], strict=True)
    dec.load_state_dict(case["dec_state_dict"], strict=True)
    enc.to(dev), dec.to(dev)
    leaves = {k: v.to(dev).requires_grad_() for k, v in case["inputs"].items() if k != "eps"}
    emb = {k: leaves[k] for k in ("mesh", "g2m", "m2m", "m2g")}
    dist = enc(leaves["grid_rep"], graph_emb=emb)
    z = dist.mean + dist.stddev * case["inputs"]["eps"].to(dev)
    mean_delta, pred_std = dec(leaves["grid_rep"], z, emb)
    assert rel_err(dist.mean.cpu(), case["ref_latent_mean"]) < TOL
    assert rel_err(dist.stddev.cpu(), case["ref_latent_std"]) < TOL
    assert rel_err(mean_delta.cpu(), case["ref_mean_delta"]) < TOL
    assert rel_err(pred_std.cpu(), case["ref_pred_std"]) < TOL
    cot = {k: v.to(dev) for k, v in case["cotangents"].items()}
    loss = (dist.mean * cot["mean"]).sum() + (dist.stddev * cot["std"]).sum() + (mean_delta * cot["delta"]).sum() + (pred_std * cot["pstd"]).sum()
    loss.backward()
    for k, v in leaves.items():
        assert rel_err(v.grad.cpu(), case["ref_grad_inputs"][k]) < TOL, k
    for k, p in enc.named_parameters():
        assert rel_err(p.grad.cpu(), case["ref_grad_enc"][k]) < TOL, k
    for k, p in dec.named_parameters():
        assert rel_err(p.grad.cpu(), case["ref_grad_dec"][k]) < TOL, k


# ---- hierarchical variants (hi_graph_encoder.py, hi_graph_decoder.py) ----
HI_CASES = ["latent_hi_d32", "latent_hi_d16_nointra"]


def _build_hi(case):
    from neural_lam_amd import latent

    enc = latent.HiGraphLatentEncoder(case["latent_dim"], case["g2m_edge_index"], case["m2m_edge_index"], case["mesh_up_edge_index"],
                                      case["d"], case["intra_layers"], hidden_layers=1, g2m_gnn_type=case["g2m_gnn_type"],
                                      output_dist=case["output_dist"])
    dec = latent.HiGraphLatentDecoder(case["g2m_edge_index"], case["m2m_edge_index"], case["m2g_edge_index"], case["mesh_up_edge_index"],
                                      case["mesh_down_edge_index"], case["d"], case["latent_dim"], case["num_state"], case["intra_layers"],
                                      hidden_layers=1, g2m_gnn_type=case["g2m_gnn_type"], m2g_gnn_type=case["m2g_gnn_type"], output_std=True)
    return enc, dec


def _hi_emb(case, leaves):
    L = case["levels"]
    return {"g2m": leaves["g2m"], "m2g": leaves["m2g"], "mesh": [leaves[f"mesh_{lv}"] for lv in range(L)],
            "m2m": [leaves[f"m2m_{lv}"] for lv in range(L)], "mesh_up": [leaves[f"mesh_up_{lv}"] for lv in range(L - 1)],
            "mesh_down": [leaves[f"mesh_down_{lv}"] for lv in range(L - 1)]}


@pytest.mark.parametrize("name", HI_CASES)
def test_hierarchical_reference_state_dicts_load_strictly(name):
    case = load_golden(name)
    enc, dec = _build_hi(case)
    r1 = enc.load_state_dict(case["enc_state_dict"], strict=True)
    r2 = dec.load_state_dict(case["dec_state_dict"], strict=True)
    assert not r1.missing_keys and not r1.unexpected_keys and not r2.missing_keys and not r2.unexpected_keys
    assert (dec.intra_down_gnns is None) == (case["intra_layers"] == 0)
    if case["intra_layers"]:
        assert len(dec.intra_up_gnns) == case["levels"] and len(dec.intra_down_gnns) == case["levels"] - 1


def test_hierarchical_latent_needs_two_levels():
    from neural_lam_amd import latent

    case = load_golden("latent_hi_d16_nointra")
    with pytest.raises(ValueError, match="at least 2 mesh levels"):
        latent.HiGraphLatentEncoder(4, case["g2m_edge_index"], case["m2m_edge_index"][:1], [], 16, 1)
    with pytest.raises(ValueError, match="at least 2 mesh levels"):
        latent.HiGraphLatentDecoder(case["g2m_edge_index"], case["m2m_edge_index"][:1], case["m2g_edge_index"], [], [], 16, 4, 5, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("name", HI_CASES)
def test_hierarchical_latent_stack_matches_reference_golden(name):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    case = load_golden(name)
    enc, dec = _build_hi(case)
    enc.load_state_dict(case["enc_state_dict"], strict=True)
    dec.load_state_dict(case["dec_state_dict"], strict=True)
    enc.to(dev), dec.to(dev)
    leaves = {k: v.to(dev).requires_grad_() for k, v in case["inputs"].items() if k != "eps"}
    emb = _hi_emb(case, leaves)
    dist = enc(leaves["grid_rep"], graph_emb=emb)
    z = dist.mean + dist.stddev * case["inputs"]["eps"].to(dev)
    mean_delta, pred_std = dec(leaves["grid_rep"], z, emb)
    assert rel_err(dist.mean.cpu(), case["ref_latent_mean"]) < TOL
    assert rel_err(dist.stddev.cpu(), case["ref_latent_std"]) < TOL
    assert rel_err(mean_delta.cpu(), case["ref_mean_delta"]) < TOL
    assert rel_err(pred_std.cpu(), case["ref_pred_std"]) < TOL
    cot = {k: v.to(dev) for k, v in case["cotangents"].items()}
    loss = (dist.mean * cot["mean"]).sum() + (dist.stddev * cot["std"]).sum() + (mean_delta * cot["delta"]).sum() + (pred_std * cot["pstd"]).sum()
    loss.backward()
    for k, v in leaves.items():
        ref = case["ref_grad_inputs"][k]
        if ref is None:   # a leaf the reference computation never reaches (edge states discarded by the last intra-level layer ...)
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
        else:
            assert v.grad is not None and rel_err(v.grad.cpu(), ref) < TOL, k
    for k, p in enc.named_parameters():
        assert rel_err(p.grad.cpu(), case["ref_grad_enc"][k]) < TOL, k
    for k, p in dec.named_parameters():
        assert rel_err(p.grad.cpu(), case["ref_grad_dec"][k]) < TOL, k
