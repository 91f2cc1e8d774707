"""Generate tests/golden/*.pt by running the REFERENCE's own code.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Every tensor stored under key "ref_*" was computed by the reference's files
(gnn_layers.py, utils/networks.py, utils/graph.py, create_graph.py,
models/step_predictors/**, models/forecasters/autoregressive.py, models/latent/*.py, metrics.py)
imported unmodified through tests/golden/ref_harness.py.  The only non-reference
code in the loop is the torch_geometric stand-in of ref_harness.py (PyG 2.3.1 is
not installed and there is no network) and the duck-typed SyntheticDatastore.
"""
import sys
import tempfile
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent.parent))

import ref_harness as rh  # noqa: E402
from neural_lam_amd import graph as G  # noqa: E402
from neural_lam_amd.datastore import SyntheticDatastore  # noqa: E402

NOTE = (
    "ref_* tensors computed by /root/reference code (mllam/neural-lam) on top of a "
    "torch_geometric stand-in (tests/golden/ref_harness.py); fp32 CPU, torch "
    + torch.__version__
)


def rand_edge_index(ns, nr, e, seed, force_max=True):
    g = torch.Generator().manual_seed(seed)
    s = torch.randint(0, ns, (e,), generator=g)
    r = torch.randint(0, nr, (e,), generator=g)
    if force_max:
        r[-1] = nr - 1
        s[-1] = ns - 1
    return torch.stack([s, r])


def layer_case(ref, name, cls_name, ns, nr, e, d, batch, seed, skip_rec=None, **kw):
    ei = rand_edge_index(ns, nr, e, seed)
    if skip_rec is not None:  # leave a receiver below the max without edges
        ei[1][ei[1] == skip_rec] = (skip_rec + 1) % nr
        ei[1][-1] = nr - 1
    torch.manual_seed(seed + 1)
    net = getattr(ref.gnn_layers, cls_name)(ei, d, **kw)
    # non-trivial LayerNorm affine so the parameters are pinned too
    with torch.no_grad():
        for n_, p in net.named_parameters():
            if n_.endswith("3.weight"):
                p.add_(0.1 * torch.randn_like(p))
            if n_.endswith("3.bias"):
                p.add_(0.1 * torch.randn_like(p))
    shape = (lambda n: (n, d)) if batch is None else (lambda n: (batch, n, d))
    g = torch.Generator().manual_seed(seed + 2)
    send = torch.randn(shape(ns), generator=g, requires_grad=True)
    rec = torch.randn(shape(nr), generator=g, requires_grad=True)
    edge = torch.randn(shape(e), generator=g, requires_grad=True)
    out = net(send, rec, edge)
    outs = out if isinstance(out, tuple) else (out,)
    # fixed random cotangents -> scalar
    cots = [torch.randn(o.shape, generator=g) for o in outs]
    loss = sum((o * c).sum() for o, c in zip(outs, cots))
    loss.backward()
    case = {
        "cls": cls_name,
        "kwargs": kw,
        "d": d,
        "edge_index": ei.to(torch.int32),
        "state_dict": {k: v.detach().clone() for k, v in net.state_dict().items()},
        "send": send.detach(),
        "rec": rec.detach(),
        "edge": edge.detach(),
        "cotangents": cots,
        "ref_out": [o.detach() for o in outs],
        "ref_grad_send": send.grad,
        "ref_grad_rec": rec.grad,
        "ref_grad_edge": edge.grad,
        "ref_grad_params": {k: p.grad.clone() for k, p in net.named_parameters()},
    }
    print(f"  layer case {name}: E={e} d={d} out={[tuple(o.shape) for o in outs]}")
    return case


def make_layers(ref):
    cases = {}
    cases["inet_sum_update_d8"] = layer_case(ref, "inet_sum_update_d8", "InteractionNet", 5, 4, 10, 8, None, 0)
    cases["inet_mean_noupdate_b2_d8"] = layer_case(
        ref, "inet_mean_noupdate_b2_d8", "InteractionNet", 7, 6, 20, 8, 2, 10, update_edges=False, aggr="mean"
    )
    cases["propnet_b2_d8"] = layer_case(ref, "propnet_b2_d8", "PropagationNet", 6, 5, 17, 8, 2, 20)
    cases["propnet_noupdate_d16"] = layer_case(
        ref, "propnet_noupdate_d16", "PropagationNet", 9, 4, 30, 16, None, 25, update_edges=False
    )
    cases["inet_chunked_d8"] = layer_case(
        ref, "inet_chunked_d8", "InteractionNet", 6, 6, 12, 8, 2, 30, edge_chunk_sizes=[5, 7], aggr_chunk_sizes=[2, 4]
    )
    cases["inet_100to10_gap_d16"] = layer_case(
        ref, "inet_100to10_gap_d16", "InteractionNet", 100, 10, 200, 16, None, 40, skip_rec=3
    )
    cases["inet_sum_update_b2_d64"] = layer_case(ref, "inet_sum_update_b2_d64", "InteractionNet", 50, 40, 300, 64, 2, 50)
    cases["inet_highdeg_d32"] = layer_case(
        ref, "inet_highdeg_d32", "InteractionNet", 300, 3, 500, 32, None, 60, aggr="mean"
    )
    cases["inet_hidden12_d8"] = layer_case(ref, "inet_hidden12_d8", "InteractionNet", 5, 4, 10, 8, None, 70, hidden_dim=8)
    torch.save({"note": NOTE, "cases": cases}, HERE / "layers.pt")


def make_layers_wide(ref):
    """d in {128, 256}: the workgroup-cooperative ("wide") kernels of csrc/nlam_wide.inc."""
    cases = {}
    cases["inet_sum_update_b2_d128"] = layer_case(ref, "inet_sum_update_b2_d128", "InteractionNet", 40, 30, 200, 128, 2, 80)
    cases["propnet_d256"] = layer_case(ref, "propnet_d256", "PropagationNet", 30, 25, 150, 256, None, 90)
    cases["inet_mean_noupdate_d128"] = layer_case(
        ref, "inet_mean_noupdate_d128", "InteractionNet", 300, 5, 400, 128, None, 95, update_edges=False, aggr="mean"
    )
    torch.save({"note": NOTE, "cases": cases}, HERE / "layers_wide.pt")


def make_layers_wide_chunked(ref):
    """SplitMLPs (gnn_layers.py:274-324) at the width HiLAMParallel is benchmarked at (cfg4p, d = 128): chunked edge and
    node MLPs on row windows that start inside a 32-row tile, through the wide kernels.  Own file so that the older
    fixtures stay byte-identical."""
    cases = {}
    cases["inet_chunked_b2_d128"] = layer_case(
        ref, "inet_chunked_b2_d128", "InteractionNet", 45, 37, 700, 128, 2, 97,
        edge_chunk_sizes=[301, 250, 149], aggr_chunk_sizes=[20, 10, 7])
    cases["inet_chunked_noupdate_d128"] = layer_case(
        ref, "inet_chunked_noupdate_d128", "InteractionNet", 70, 33, 450, 128, None, 98, update_edges=False,
        edge_chunk_sizes=[64, 386], aggr_chunk_sizes=[32, 1])
    torch.save({"note": NOTE, "cases": cases}, HERE / "layers_wide_chunked.pt")


def compress_graph(raw):
    out = {}
    for k, v in raw.items():
        if isinstance(v, list):
            out[k] = [t.to(torch.int32) if t.dtype == torch.int64 else t for t in v]
        else:
            out[k] = v.to(torch.int32) if v.dtype == torch.int64 else v
    return out


def model_case(ref, name, model_name, ds_kwargs, graph_kwargs, model_kwargs, B, T, seed):
    tmp = tempfile.mkdtemp()
    ds = SyntheticDatastore(root_path=tmp, **ds_kwargs)
    gdir = Path(tmp) / "graph" / "g"
    ref.create_graph.create_graph(str(gdir), ds.get_xy("state"), **graph_kwargs)
    raw = G.read_graph_files(gdir)
    torch.manual_seed(seed)
    cls = getattr(ref, model_name)
    predictor = cls(ds, graph_name="g", **model_kwargs)
    forecaster = ref.ARForecaster(predictor, ds)
    n_state, n_forc = ds.get_num_data_vars("state"), ds.get_num_data_vars("forcing")
    N = ds.num_grid_points
    g = torch.Generator().manual_seed(seed + 1)
    init = torch.randn(B, 2, N, n_state, generator=g)
    target = torch.randn(B, T, N, n_state, generator=g)
    forcing = torch.randn(B, T, N, n_forc * 3, generator=g)
    if model_kwargs.get("output_clamping_lower") or model_kwargs.get("output_clamping_upper"):
        # states must start inside the clamp range for the inverse maps to be meaningful
        init = init.abs() * 0.1 + 0.2
        target = target.abs() * 0.1 + 0.2
    # reference per_var_std (module.py:158-176, uniform weights) and interior mask
    st = ds.get_standardization_dataarray("state")
    diff_std = torch.tensor(st.state_diff_std_standardized.values, dtype=torch.float32)
    w = torch.tensor([1.0 / n_state] * n_state, dtype=torch.float32)
    per_var_std = diff_std / torch.sqrt(w)
    interior = (1.0 - torch.tensor(ds.boundary_mask.values, dtype=torch.float32)).to(torch.bool)
    pred, loss = rh.ref_training_loss(ref, forecaster, (init, target, forcing), per_var_std, interior)
    loss.backward()
    # reference-loaded (normalised) graph tensors, straight from the module's buffers
    names = [
        "g2m_edge_index", "m2g_edge_index", "m2m_edge_index", "mesh_up_edge_index", "mesh_down_edge_index",
        "g2m_features", "m2g_features", "m2m_features", "mesh_up_features", "mesh_down_features",
        "mesh_static_features",
    ]
    loaded = {}
    for n_ in names:
        v = getattr(predictor, n_)
        loaded[n_] = v.clone() if torch.is_tensor(v) else [t.clone() for t in v]
    one_step, one_std = predictor(init[:, 1], init[:, 0], forcing[:, 0])
    case = {
        "note": NOTE,
        "model": model_name,
        "ds_kwargs": ds_kwargs,
        "graph_kwargs": graph_kwargs,
        "model_kwargs": model_kwargs,
        "ref_graph_raw": compress_graph(raw),
        "ref_graph_loaded": compress_graph(loaded),
        "ref_hierarchical": bool(predictor.hierarchical),
        "state_dict": {k: v.detach().clone() for k, v in forecaster.state_dict().items()},
        "init": init,
        "target": target,
        "forcing": forcing,
        "ref_one_step": one_step.detach(),
        "ref_one_std": None if one_std is None else one_std.detach(),
        "ref_prediction": pred.detach(),
        "ref_loss": loss.detach(),
        "ref_grads": {k: p.grad.clone() for k, p in forecaster.named_parameters()},
    }
    torch.save(case, HERE / f"{name}.pt")
    print(f"  model case {name}: loss={float(loss):.6f} params={sum(p.numel() for p in forecaster.parameters())}")


DS_SMALL = dict(nx=30, ny=27, num_state=5, num_forcing=2, num_static=1, boundary="random", seed=3,
                state_stats={"state_mean": [0.1, -0.2, 0.3, 0.0, 0.5], "state_std": [1.0, 2.0, 0.5, 1.5, 1.0],
                             "state_diff_mean_standardized": [0.01, -0.02, 0.0, 0.03, 0.0],
                             "state_diff_std_standardized": [0.5, 0.8, 1.0, 1.2, 0.9]})


def latent_case(ref, name, d, latent_dim, m2m_layers, B, seed, output_dist="diagonal", g2m_gnn_type="InteractionNet",
                m2g_gnn_type="InteractionNet"):
    """Graph-EFM latent encoder + decoder (reference files models/latent/{base,graph}_{encoder,decoder}.py,
    imported unmodified) on a small flat graph: distribution parameters, decoder outputs, all gradients."""
    import importlib

    latent = importlib.import_module("neural_lam.models.latent")
    xy = G.regular_grid_xy(30, 27)
    raw = G.create_regular_grid_graph(xy, n_max_levels=None, hierarchical=False)
    g2m, m2g, m2m = raw["g2m_edge_index"], raw["m2g_edge_index"], raw["m2m_edge_index"][0]
    n_grid, n_mesh = 30 * 27, int(raw["mesh_features"][0].shape[0])
    num_state = 5
    torch.manual_seed(seed)
    enc = latent.GraphLatentEncoder(latent_dim, g2m, m2m, d, m2m_layers, hidden_layers=1, g2m_gnn_type=g2m_gnn_type,
                                    output_dist=output_dist)
    dec = latent.GraphLatentDecoder(g2m, m2m, m2g, d, latent_dim, num_state, m2m_layers, hidden_layers=1,
                                    g2m_gnn_type=g2m_gnn_type, m2g_gnn_type=m2g_gnn_type, output_std=True)
    gen = torch.Generator().manual_seed(seed + 1)
    t = lambda *shape: torch.randn(*shape, generator=gen)  # noqa: E731
    inputs = {"grid_rep": t(B, n_grid, d), "mesh": t(B, n_mesh, d), "g2m": t(B, g2m.shape[1], d), "m2m": t(B, m2m.shape[1], d),
              "m2g": t(B, m2g.shape[1], d), "eps": t(B, n_mesh, latent_dim)}
    leaves = {k: v.clone().requires_grad_() for k, v in inputs.items() if k != "eps"}
    emb = {k: leaves[k] for k in ("mesh", "g2m", "m2m", "m2g")}
    dist = enc(leaves["grid_rep"], graph_emb=emb)
    z = dist.mean + dist.stddev * inputs["eps"]   # reparameterised sample (rsample with a stored noise)
    mean_delta, pred_std = dec(leaves["grid_rep"], z, emb)
    cot = {"mean": t(*dist.mean.shape), "std": t(*dist.stddev.shape), "delta": t(*mean_delta.shape), "pstd": t(*pred_std.shape)}
    loss = (dist.mean * cot["mean"]).sum() + (dist.stddev * cot["std"]).sum() + (mean_delta * cot["delta"]).sum() + (pred_std * cot["pstd"]).sum()
    loss.backward()
    case = {
        "note": NOTE, "d": d, "latent_dim": latent_dim, "m2m_layers": m2m_layers, "num_state": num_state, "output_dist": output_dist,
        "g2m_gnn_type": g2m_gnn_type, "m2g_gnn_type": m2g_gnn_type,
        "g2m_edge_index": g2m, "m2m_edge_index": m2m, "m2g_edge_index": m2g,
        "inputs": inputs, "cotangents": cot,
        "enc_state_dict": {k: v.clone() for k, v in enc.state_dict().items()},
        "dec_state_dict": {k: v.clone() for k, v in dec.state_dict().items()},
        "ref_latent_mean": dist.mean.detach().clone(), "ref_latent_std": dist.stddev.detach().clone(),
        "ref_mean_delta": mean_delta.detach().clone(), "ref_pred_std": pred_std.detach().clone(),
        "ref_grad_inputs": {k: v.grad.clone() for k, v in leaves.items()},
        "ref_grad_enc": {k: p.grad.clone() for k, p in enc.named_parameters()},
        "ref_grad_dec": {k: p.grad.clone() for k, p in dec.named_parameters()},
    }
    torch.save(case, HERE / f"{name}.pt")
    print(f"  latent case {name}: loss={float(loss):.6f}")


def hi_latent_case(ref, name, d, latent_dim, intra_layers, B, seed, output_dist="diagonal", g2m_gnn_type="InteractionNet",
                   m2g_gnn_type="InteractionNet"):
    """Hierarchical Graph-EFM latent encoder + decoder (reference files models/latent/hi_graph_{encoder,decoder}.py,
    imported unmodified) on a 3-level hierarchy: distribution parameters, decoder outputs, all gradients."""
    import importlib

    latent = importlib.import_module("neural_lam.models.latent")
    nx, ny = 81, 30
    raw = G.create_regular_grid_graph(G.regular_grid_xy(nx, ny), n_max_levels=3, hierarchical=True)
    g2m, m2g = raw["g2m_edge_index"], raw["m2g_edge_index"]
    m2m, up, down = list(raw["m2m_edge_index"]), list(raw["mesh_up_edge_index"]), list(raw["mesh_down_edge_index"])
    n_grid = nx * ny
    n_mesh = [int(f.shape[0]) for f in raw["mesh_features"]]
    num_state = 5
    torch.manual_seed(seed)
    enc = latent.HiGraphLatentEncoder(latent_dim, g2m, m2m, up, d, intra_layers, hidden_layers=1, g2m_gnn_type=g2m_gnn_type,
                                      output_dist=output_dist)
    dec = latent.HiGraphLatentDecoder(g2m, m2m, m2g, up, down, d, latent_dim, num_state, intra_layers, hidden_layers=1,
                                      g2m_gnn_type=g2m_gnn_type, m2g_gnn_type=m2g_gnn_type, output_std=True)
    gen = torch.Generator().manual_seed(seed + 1)
    t = lambda *shape: torch.randn(*shape, generator=gen)  # noqa: E731
    inputs = {"grid_rep": t(B, n_grid, d), "g2m": t(B, g2m.shape[1], d), "m2g": t(B, m2g.shape[1], d), "eps": t(B, n_mesh[-1], latent_dim)}
    for lv, n in enumerate(n_mesh):
        inputs[f"mesh_{lv}"] = t(B, n, d)
        inputs[f"m2m_{lv}"] = t(B, m2m[lv].shape[1], d)
    for lv in range(len(up)):
        inputs[f"mesh_up_{lv}"] = t(B, up[lv].shape[1], d)
        inputs[f"mesh_down_{lv}"] = t(B, down[lv].shape[1], d)
    leaves = {k: v.clone().requires_grad_() for k, v in inputs.items() if k != "eps"}
    L = len(n_mesh)
    emb = {"g2m": leaves["g2m"], "m2g": leaves["m2g"], "mesh": [leaves[f"mesh_{lv}"] for lv in range(L)],
           "m2m": [leaves[f"m2m_{lv}"] for lv in range(L)], "mesh_up": [leaves[f"mesh_up_{lv}"] for lv in range(L - 1)],
           "mesh_down": [leaves[f"mesh_down_{lv}"] for lv in range(L - 1)]}
    dist = enc(leaves["grid_rep"], graph_emb=emb)
    z = dist.mean + dist.stddev * inputs["eps"]
    mean_delta, pred_std = dec(leaves["grid_rep"], z, emb)
    cot = {"mean": t(*dist.mean.shape), "std": t(*dist.stddev.shape), "delta": t(*mean_delta.shape), "pstd": t(*pred_std.shape)}
    loss = (dist.mean * cot["mean"]).sum() + (dist.stddev * cot["std"]).sum() + (mean_delta * cot["delta"]).sum() + (pred_std * cot["pstd"]).sum()
    loss.backward()
    case = {
        "note": NOTE, "d": d, "latent_dim": latent_dim, "intra_layers": intra_layers, "num_state": num_state, "output_dist": output_dist,
        "g2m_gnn_type": g2m_gnn_type, "m2g_gnn_type": m2g_gnn_type, "levels": L,
        "g2m_edge_index": g2m, "m2g_edge_index": m2g, "m2m_edge_index": m2m, "mesh_up_edge_index": up, "mesh_down_edge_index": down,
        "inputs": inputs, "cotangents": cot,
        "enc_state_dict": {k: v.clone() for k, v in enc.state_dict().items()},
        "dec_state_dict": {k: v.clone() for k, v in dec.state_dict().items()},
        "ref_latent_mean": dist.mean.detach().clone(), "ref_latent_std": dist.stddev.detach().clone(),
        "ref_mean_delta": mean_delta.detach().clone(), "ref_pred_std": pred_std.detach().clone(),
        # (a leaf the computation never reaches has no gradient: stored as None and checked as such)
        "ref_grad_inputs": {k: (None if v.grad is None else v.grad.clone()) for k, v in leaves.items()},
        "ref_grad_enc": {k: p.grad.clone() for k, p in enc.named_parameters()},
        "ref_grad_dec": {k: p.grad.clone() for k, p in dec.named_parameters()},
    }
    torch.save(case, HERE / f"{name}.pt")
    print(f"  hierarchical latent case {name}: levels={L} nodes={n_mesh} loss={float(loss):.6f}")


def legacy_graph_cases(ref):
    """Legacy (pre-spec) graph directories -- one node index space, no metainfo.yaml -- loaded by the reference's own
    ``utils.load_graph`` (utils/graph.py:146-422).  The directories are made from the reference generator's output by
    re-applying the legacy offsets: mesh levels first then the grid (hierarchical case), grid first then the mesh (flat)."""
    import warnings

    cases = {}
    for name, nx, ny, kw, mesh_first in (("hi_mesh_first", 81, 30, dict(n_max_levels=3, hierarchical=True), True),
                                         ("flat_grid_first", 30, 27, dict(n_max_levels=None, hierarchical=False), False)):
        tmp = Path(tempfile.mkdtemp())
        ref.create_graph.create_graph(str(tmp), G.regular_grid_xy(nx, ny), **kw)
        raw = G.read_graph_files(tmp)
        n_mesh = [int(f.shape[0]) for f in raw["mesh_features"]]
        lvl_off = [sum(n_mesh[:l]) for l in range(len(n_mesh))]
        n_grid = nx * ny
        legacy = dict(raw)
        legacy["m2m_edge_index"] = [e + lvl_off[l] + (0 if mesh_first else n_grid) for l, e in enumerate(raw["m2m_edge_index"])]
        if mesh_first:
            legacy["g2m_edge_index"] = torch.stack((raw["g2m_edge_index"][0] + sum(n_mesh), raw["g2m_edge_index"][1]))
            legacy["m2g_edge_index"] = torch.stack((raw["m2g_edge_index"][0], raw["m2g_edge_index"][1] + sum(n_mesh)))
        else:
            legacy["g2m_edge_index"] = torch.stack((raw["g2m_edge_index"][0], raw["g2m_edge_index"][1] + n_grid))
            legacy["m2g_edge_index"] = torch.stack((raw["m2g_edge_index"][0] + n_grid, raw["m2g_edge_index"][1]))
        if "mesh_up_edge_index" in raw:
            legacy["mesh_up_edge_index"] = [torch.stack((e[0] + lvl_off[l], e[1] + lvl_off[l + 1])) for l, e in enumerate(raw["mesh_up_edge_index"])]
            legacy["mesh_down_edge_index"] = [torch.stack((e[0] + lvl_off[l + 1], e[1] + lvl_off[l])) for l, e in enumerate(raw["mesh_down_edge_index"])]
        legacy["mesh_features"] = [f / 7.5 for f in raw["mesh_features"]]   # "already normalised": any fixed scaling
        ldir = tmp / "legacy"
        ldir.mkdir()
        for k, v in legacy.items():
            if k != "spec_version":
                torch.save(v, ldir / f"{k}.pt")
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            hier, loaded = ref.utils.load_graph(str(ldir), 123.0)   # the scaling must be ignored for legacy graphs
        assert any(issubclass(x.category, RuntimeWarning) for x in w)
        loaded = {k: (list(v) if not torch.is_tensor(v) else v) for k, v in loaded.items()}
        cases[name] = {"legacy_files": compress_graph({k: v for k, v in legacy.items() if k != "spec_version"}),
                       "ref_hierarchical": hier, "ref_graph_loaded": compress_graph(loaded)}
        print(f"  legacy graph case {name}: hierarchical={hier} levels={len(n_mesh)}")
    torch.save({"note": NOTE, "cases": cases}, HERE / "legacy_graphs.pt")


def efm_case(ref, name, cls_name, ds_kwargs, graph_kwargs, model_kwargs, B, seed):
    """Graph-EFM step predictor (reference file models/step_predictors/graph/graph_efm.py, imported unmodified): one forward
    with the prior sample the reference drew (the standard-normal noise is recorded and checked against the latent the
    decoder received), its outputs, and the gradients of a fixed linear functional of them."""
    import importlib

    efm = importlib.import_module("neural_lam.models.step_predictors.graph.graph_efm")
    tmp = tempfile.mkdtemp()
    ds = SyntheticDatastore(root_path=tmp, **ds_kwargs)
    gdir = Path(tmp) / "graph" / "g"
    ref.create_graph.create_graph(str(gdir), ds.get_xy("state"), **graph_kwargs)
    raw = G.read_graph_files(gdir)
    torch.manual_seed(seed)
    model = getattr(efm, cls_name)(ds, graph_name="g", **model_kwargs)
    n_state, n_forc, N = ds.get_num_data_vars("state"), ds.get_num_data_vars("forcing"), ds.num_grid_points
    g = torch.Generator().manual_seed(seed + 1)
    prev, prev_prev = torch.randn(B, N, n_state, generator=g), torch.randn(B, N, n_state, generator=g)
    forcing = torch.randn(B, N, n_forc * 3, generator=g)
    if model_kwargs.get("output_clamping_lower") or model_kwargs.get("output_clamping_upper"):
        prev, prev_prev = prev.abs() * 0.1 + 0.2, prev_prev.abs() * 0.1 + 0.2
    seen = {}
    model.decoder.register_forward_pre_hook(lambda mod, args: seen.__setitem__("latent", args[1].detach().clone()))
    # the noise Normal.rsample will draw: same generator state, same shape (nothing else in forward touches the RNG)
    lat_shape = (B, model.latent_spatial_dim, model.latent_dim)
    torch.manual_seed(seed + 2)
    noise = torch.empty(lat_shape).normal_()
    torch.manual_seed(seed + 2)
    pred_mean, pred_std = model(prev, prev_prev, forcing)
    with torch.no_grad():
        grid_emb, graph_emb = model.embedd_grid_and_graph(prev, prev_prev, forcing)
        dist = model.prior_model(grid_emb, graph_emb=graph_emb)
        assert torch.allclose(seen["latent"], dist.mean + dist.stddev * noise, rtol=0, atol=0), "recorded noise is not what rsample drew"
        # the variational encoder is not on forward's path: its output goes into the fixture on its own
        enc_in = model.embedd_grid_with_target(prev, prev_prev, forcing, prev + 0.1)
        enc_dist = model.encoder(enc_in, graph_emb=graph_emb)
    cot = {"mean": torch.randn(pred_mean.shape, generator=g)}
    loss = (pred_mean * cot["mean"]).sum()
    if pred_std is not None:
        cot["std"] = torch.randn(pred_std.shape, generator=g)
        loss = loss + (pred_std * cot["std"]).sum()
    loss.backward()
    case = {
        "note": NOTE, "cls": cls_name, "ds_kwargs": ds_kwargs, "graph_kwargs": graph_kwargs, "model_kwargs": model_kwargs,
        "ref_graph_raw": compress_graph(raw),
        "state_dict": {k: v.detach().clone() for k, v in model.state_dict().items()},
        "prev": prev, "prev_prev": prev_prev, "forcing": forcing, "noise": noise, "cotangents": cot,
        "ref_latent": seen["latent"], "ref_pred_mean": pred_mean.detach(), "ref_pred_std": None if pred_std is None else pred_std.detach(),
        "ref_enc_mean": enc_dist.mean.clone(), "ref_enc_std": enc_dist.stddev.clone(),
        # parameters off forward's path (the variational encoder, grid_current_embedder) have no gradient: None
        "ref_grads": {k: (None if p.grad is None else p.grad.clone()) for k, p in model.named_parameters()},
    }
    torch.save(case, HERE / f"{name}.pt")
    print(f"  EFM case {name}: loss={float(loss):.6f} params={sum(p.numel() for p in model.parameters())} latent={lat_shape}")


def efm_cases(ref):
    ds_hi = dict(nx=81, ny=30, num_state=5, num_forcing=2, num_static=1, boundary="frame", boundary_width=4, seed=5)
    efm_case(ref, "efm_hi_81x30", "GraphEFM", ds_hi, dict(n_max_levels=3, hierarchical=True),
             dict(hidden_dim=16, latent_dim=8, prior_intra_level_layers=1, encoder_intra_level_layers=1, decoder_intra_level_layers=2,
                  output_std=True), B=1, seed=60)
    efm_case(ref, "efm_ms_30x27", "GraphEFMMultiScale", DS_SMALL, dict(n_max_levels=None, hierarchical=False),
             dict(hidden_dim=32, prior_m2m_layers=1, encoder_m2m_layers=1, decoder_m2m_layers=2, prior_dist="diagonal",
                  output_clamping_lower={"state_var_0": 0.0}, output_clamping_upper={"state_var_2": 1.0}), B=2, seed=61)
    efm_case(ref, "efm_hi_constprior_81x30", "GraphEFM", ds_hi, dict(n_max_levels=3, hierarchical=True),
             dict(hidden_dim=8, latent_dim=4, prior_intra_level_layers=0, encoder_intra_level_layers=0, decoder_intra_level_layers=0,
                  learn_prior=False, g2m_gnn_type="PropagationNet", m2g_gnn_type="PropagationNet"), B=1, seed=62)


def main():
    ref = rh.load_reference()
    if "--efm-only" in sys.argv:
        efm_cases(ref)
        return
    if "--legacy-only" in sys.argv:
        legacy_graph_cases(ref)
        return
    if "--latent-only" in sys.argv:
        latent_case(ref, "latent_flat_d64", 64, 16, 2, 2, 50)
        latent_case(ref, "latent_flat_d16_prop", 16, 8, 1, 1, 51, output_dist="isotropic", g2m_gnn_type="PropagationNet",
                    m2g_gnn_type="PropagationNet")
        hi_latent_case(ref, "latent_hi_d32", 32, 8, 1, 1, 52)
        hi_latent_case(ref, "latent_hi_d16_nointra", 16, 4, 0, 1, 53, output_dist="isotropic", g2m_gnn_type="PropagationNet")
        return
    if "--wide-chunked-only" in sys.argv:
        make_layers_wide_chunked(ref)
        return
    if "--wide-only" in sys.argv:
        make_layers_wide(ref)
        model_case(ref, "graphlam_30x27_d128", "GraphLAM", DS_SMALL, dict(n_max_levels=None, hierarchical=False),
                   dict(hidden_dim=128, hidden_layers=1, processor_layers=1), B=1, T=1, seed=46)
        return
    make_layers(ref)
    make_layers_wide(ref)
    make_layers_wide_chunked(ref)
    legacy_graph_cases(ref)
    efm_cases(ref)
    latent_case(ref, "latent_flat_d64", 64, 16, 2, 2, 50)
    latent_case(ref, "latent_flat_d16_prop", 16, 8, 1, 1, 51, output_dist="isotropic", g2m_gnn_type="PropagationNet",
                m2g_gnn_type="PropagationNet")
    hi_latent_case(ref, "latent_hi_d32", 32, 8, 1, 1, 52)
    hi_latent_case(ref, "latent_hi_d16_nointra", 16, 4, 0, 1, 53, output_dist="isotropic", g2m_gnn_type="PropagationNet")
    ds_small = DS_SMALL
    model_case(ref, "graphlam_30x27", "GraphLAM", ds_small, dict(n_max_levels=None, hierarchical=False),
               dict(hidden_dim=16, hidden_layers=1, processor_layers=2), B=2, T=2, seed=42)
    model_case(ref, "graphlam_30x27_variants", "GraphLAM", ds_small, dict(n_max_levels=1, hierarchical=False),
               dict(hidden_dim=8, hidden_layers=1, processor_layers=1, mesh_aggr="mean", output_std=True,
                    g2m_gnn_type="PropagationNet", m2g_gnn_type="PropagationNet",
                    output_clamping_lower={"state_var_0": 0.0, "state_var_2": 0.0},
                    output_clamping_upper={"state_var_2": 1.0, "state_var_3": 5.0}), B=1, T=3, seed=43)
    ds_hi = dict(nx=81, ny=30, num_state=5, num_forcing=2, num_static=1, boundary="frame", boundary_width=4, seed=5)
    model_case(ref, "graphlam_30x27_d128", "GraphLAM", ds_small, dict(n_max_levels=None, hierarchical=False),
               dict(hidden_dim=128, hidden_layers=1, processor_layers=1), B=1, T=1, seed=46)
    model_case(ref, "hilam_81x30", "HiLAM", ds_hi, dict(n_max_levels=3, hierarchical=True),
               dict(hidden_dim=8, hidden_layers=1, processor_layers=1), B=1, T=1, seed=44)
    model_case(ref, "hilam_parallel_81x30", "HiLAMParallel", ds_hi, dict(n_max_levels=3, hierarchical=True),
               dict(hidden_dim=8, hidden_layers=1, processor_layers=1, mesh_up_gnn_type="PropagationNet"), B=1, T=1, seed=45)
    # reference generator edge counts at MEPS size (slow pure-python path, run once)
    if "--meps" in sys.argv:
        tmp = tempfile.mkdtemp()
        xy = G.regular_grid_xy(238, 268)
        ref.create_graph.create_graph(tmp + "/ms", xy, n_max_levels=None, hierarchical=False)
        ref.create_graph.create_graph(tmp + "/hi", xy, n_max_levels=3, hierarchical=True)
        sizes = {"multiscale": G.graph_summary(G.read_graph_files(tmp + "/ms")),
                 "hierarchical": G.graph_summary(G.read_graph_files(tmp + "/hi"))}
        torch.save({"note": NOTE, "ref_meps_graph_sizes": sizes}, HERE / "meps_graph_sizes.pt")
        print(sizes)


if __name__ == "__main__":
    main()
