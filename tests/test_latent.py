"""Graph-EFM latent encoder / decoder (neural_lam_amd.latent) against golden vectors produced by the reference's own
models/latent/{base_encoder,graph_encoder,base_decoder,graph_decoder}.py (tests/golden/make_golden.py::latent_case).

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
