"""Parity tests proper: the HIP path (through the C-ABI) against
  (1) golden vectors computed by the reference's own code (tests/golden),
  (2) the CPU oracle on seeded inputs,
  (3) size-independent properties at BASELINE.json's full MEPS size,
and the hot-path assertions of the reference's tests/test_gnn_layers.py restated
against the HIP classes.

Tolerance: fp32, max|a-b| / max|b| <= 1e-4; observed ~1e-6.
"""
import pytest
import torch

from conftest import graph_from_case, load_golden, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from neural_lam_amd import _lib

    _lib.load()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def _hl():
    from neural_lam_amd import gnn_layers

    return gnn_layers


@pytest.fixture(params=["auto", "wbf", "wbf4"])
def wide_family(request):
    """Widths above 64 have two kernel families: the fp32 MFMA one (one 32-row tile per workgroup) and the
    split-bf16 one (super tiles of 64-256 rows), chosen by launch size.  "wbf" forces the second at test sizes on its 8-wave
    workgroups (one per CU), "wbf4" on the 4-wave instantiations, forward and backward (two workgroups per CU,
    NLAM_TUNE_WBF_HALF; widths without one keep the 8-wave kernel).  The fixture value is "wbf" for both."""
    from neural_lam_amd import _lib as L

    lib = L.load()
    if request.param != "auto":
        assert lib.nlam_set_tuning(L.TUNE_WBF_MIN_SUPERTILES, 0) == 0
        assert lib.nlam_set_tuning(L.TUNE_WBF_HALF, 3 if request.param == "wbf4" else 0) == 0
    yield "auto" if request.param == "auto" else "wbf"
    assert lib.nlam_set_tuning(L.TUNE_WBF_MIN_SUPERTILES, 192) == 0
    assert lib.nlam_set_tuning(L.TUNE_WBF_HALF, 1) == 0


@pytest.fixture(params=["auto", "factorised"])
def factorise(request):
    """"factorised" forces the factorised edge MLP (node-level products + gathered pre-activation addends, normally
    used from 65 536 edges up) at test sizes; it applies to InteractionNet layers with 32- / 64-wide features."""
    from neural_lam_amd import gnn_layers as hl

    old = hl.FACTORISE_MIN_EDGES
    old_w = hl.FACTORISE_MIN_WORK_WIDE
    hl.FACTORISE_MIN_WIDTH_WIDE = 0
    hl.FACTORISE_MIN_WORK_WIDE = 0   # widths above 64: factorised wherever the super-tile kernels apply (the "wbf" family at test sizes)
    if request.param == "factorised":
        hl.FACTORISE_MIN_EDGES = 0
    yield request.param
    hl.FACTORISE_MIN_EDGES = old
    hl.FACTORISE_MIN_WORK_WIDE = old_w
    hl.FACTORISE_MIN_WIDTH_WIDE = 256


LAYER_CASES = [
    "inet_sum_update_d8", "inet_mean_noupdate_b2_d8", "propnet_b2_d8", "propnet_noupdate_d16",
    "inet_chunked_d8", "inet_100to10_gap_d16", "inet_sum_update_b2_d64", "inet_highdeg_d32", "inet_hidden12_d8",
    "inet_sum_update_b2_d128", "propnet_d256", "inet_mean_noupdate_d128",
    "inet_chunked_b2_d128", "inet_chunked_noupdate_d128",
]


@pytest.mark.parametrize("name", LAYER_CASES)
def test_layer_matches_reference_golden(dev, golden_layers, name, wide_family, factorise):
    hl = _hl()
    case = golden_layers[name]
    if factorise == "factorised" and (case["d"] not in (32, 64) or wide_family == "wbf"):
        pytest.skip("the factorised edge MLP covers 32- / 64-wide layers")
    net = hl.get_gnn_class(case["cls"])(case["edge_index"].to(torch.int64), case["d"], **case["kwargs"])
    net.load_state_dict(case["state_dict"], strict=True)
    net.to(dev)
    send, rec, edge = (case[k].to(dev).requires_grad_() for k in ("send", "rec", "edge"))
    out = net(send, rec, edge)
    outs = out if isinstance(out, tuple) else (out,)
    assert len(outs) == len(case["ref_out"])
    for o, r in zip(outs, case["ref_out"]):
        assert o.shape == r.shape and rel_err(o.cpu(), r) < TOL
    sum((o * c.to(dev)).sum() for o, c in zip(outs, case["cotangents"])).backward()
    assert rel_err(send.grad.cpu(), case["ref_grad_send"]) < TOL
    assert rel_err(rec.grad.cpu(), case["ref_grad_rec"]) < TOL
    assert rel_err(edge.grad.cpu(), case["ref_grad_edge"]) < TOL
    for k, p in net.named_parameters():
        assert rel_err(p.grad.cpu(), case["ref_grad_params"][k]) < TOL, k


@pytest.mark.parametrize("name", ["inet_chunked_b2_d128", "inet_chunked_noupdate_d128"])
def test_chunks_of_the_wide_family_share_one_grid(dev, golden_layers, name):
    """The chunks of a SplitMLPs layer at d = 128 (hi_lam_parallel.py:127-143) run as members of ONE launch each way
    (nlam_mlp_fwd_group / nlam_mlp_bwd_group on the fp32 wide kernels): same results as a launch per chunk -- outputs and
    data gradients bit for bit (a tile's arithmetic does not depend on which workgroup runs it), bias / LayerNorm sums to
    rounding (their partial sums are grouped by workgroup)."""
    from neural_lam_amd import ops

    hl = _hl()
    case = golden_layers[name]

    def run(grouped):
        old, old_w = ops.GROUP_CHUNKS, ops.GROUP_WGRADS
        ops.GROUP_CHUNKS = ops.GROUP_WGRADS = grouped   # the chunks' weight gradients share launches too (nlam_wgrad_group)
        ops.PROFILE.reset(True)
        try:
            net = hl.get_gnn_class(case["cls"])(case["edge_index"].to(torch.int64), case["d"], **case["kwargs"])
            net.load_state_dict(case["state_dict"], strict=True)
            net.to(dev)
            send, rec, edge = (case[k].to(dev).requires_grad_() for k in ("send", "rec", "edge"))
            out = net(send, rec, edge)
            outs = out if isinstance(out, tuple) else (out,)
            sum((o * c.to(dev)).sum() for o, c in zip(outs, case["cotangents"])).backward()
            keys = set(k[0] for k in ops.PROFILE.collect())
        finally:
            ops.GROUP_CHUNKS, ops.GROUP_WGRADS = old, old_w
            ops.PROFILE.reset(False)
        return [o.detach() for o in outs], [send.grad, rec.grad, edge.grad], {k: p.grad for k, p in net.named_parameters()}, keys

    o1, g1, p1, k1 = run(True)
    o0, g0, p0, k0 = run(False)
    assert "mlp_fwd_group_wide" in k1 and "mlp_bwd_group_wide" in k1 and "wgrad_group" in k1
    assert "mlp_fwd_group_wide" not in k0 and "mlp_bwd_group_wide" not in k0 and "wgrad_group" not in k0
    for k in p1:   # weight matrices: the same partial sums in the same order
        if p1[k].dim() == 2:
            assert torch.equal(p1[k], p0[k]), k
    for a, b in zip(o1 + g1, o0 + g0):
        assert torch.equal(a, b)
    for k in p1:
        assert rel_err(p1[k].cpu(), p0[k].cpu()) < 1e-5, k
    for o, r in zip(o1, case["ref_out"]):
        assert rel_err(o.cpu(), r) < TOL


MODEL_CASES = ["graphlam_30x27", "graphlam_30x27_variants", "graphlam_30x27_d128", "hilam_81x30", "hilam_parallel_81x30"]


@pytest.mark.parametrize("name", MODEL_CASES)
def test_model_training_step_matches_reference_golden(dev, name, tmp_path, wide_family):
    from neural_lam_amd import models as hm
    from neural_lam_amd.datastore import SyntheticDatastore

    case = load_golden(name)
    ds = SyntheticDatastore(root_path=tmp_path, **case["ds_kwargs"])
    graph = (case["ref_hierarchical"], graph_from_case(case))
    cls = {"GraphLAM": hm.GraphLAM, "HiLAM": hm.HiLAM, "HiLAMParallel": hm.HiLAMParallel}[case["model"]]
    forecaster = hm.ARForecaster(cls(ds, graph=graph, **case["model_kwargs"]), ds)
    res = forecaster.load_state_dict(case["state_dict"], strict=True)  # reference parameter names
    assert not res.missing_keys and not res.unexpected_keys
    step = hm.ForecasterStep(forecaster, ds).to(dev)
    init, target, forcing = (case[k].to(dev) for k in ("init", "target", "forcing"))
    with torch.no_grad():
        one, one_std = forecaster.predictor(init[:, 1], init[:, 0], forcing[:, 0])
    assert rel_err(one.cpu(), case["ref_one_step"]) < TOL
    if case["ref_one_std"] is not None:
        assert rel_err(one_std.cpu(), case["ref_one_std"]) < TOL
    pred, loss = step(init, target, forcing)
    assert rel_err(pred.cpu(), case["ref_prediction"]) < TOL
    assert abs(float(loss) - float(case["ref_loss"])) < TOL * abs(float(case["ref_loss"]))
    loss.backward()
    for k, p in forecaster.named_parameters():
        assert p.grad is not None, k
        ref_g = case["ref_grads"][k]
        assert float((p.grad.cpu() - ref_g).abs().max()) < 1e-4 * max(float(ref_g.abs().max()), 1e-3), k


@pytest.mark.parametrize("name", ["graphlam_30x27", "graphlam_30x27_d128", "hilam_81x30"])
def test_model_step_under_bf16_autocast(dev, name, tmp_path, wide_family):
    """The whole training step inside torch.autocast(bfloat16) (= Lightning --precision bf16-mixed): runs, every
    parameter gets a finite gradient, and prediction / loss stay within bf16-operand distance (5e-2) of the fp32
    reference run stored in the golden file."""
    from neural_lam_amd import models as hm
    from neural_lam_amd.datastore import SyntheticDatastore

    case = load_golden(name)
    ds = SyntheticDatastore(root_path=tmp_path, **case["ds_kwargs"])
    graph = (case["ref_hierarchical"], graph_from_case(case))
    cls = {"GraphLAM": hm.GraphLAM, "HiLAM": hm.HiLAM, "HiLAMParallel": hm.HiLAMParallel}[case["model"]]
    forecaster = hm.ARForecaster(cls(ds, graph=graph, **case["model_kwargs"]), ds)
    forecaster.load_state_dict(case["state_dict"], strict=True)
    step = hm.ForecasterStep(forecaster, ds).to(dev)
    init, target, forcing = (case[k].to(dev) for k in ("init", "target", "forcing"))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pred, loss = step(init, target, forcing)
    loss.float().backward()
    assert rel_err(pred.float().cpu(), case["ref_prediction"]) < 5e-2
    assert abs(float(loss) - float(case["ref_loss"])) < 5e-2 * abs(float(case["ref_loss"]))
    for k, p in forecaster.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k


# ---------------------------------------------------------------------------
# HIP vs oracle on seeded inputs (sizes the oracle finishes in seconds)
# ---------------------------------------------------------------------------
def _rand_ei(ns, nr, e, seed):
    g = torch.Generator().manual_seed(seed)
    ei = torch.stack([torch.randint(0, ns, (e,), generator=g), torch.randint(0, nr, (e,), generator=g)])
    ei[1, -1] = nr - 1
    return ei


@pytest.mark.parametrize("cls_name", ["InteractionNet", "PropagationNet"])
@pytest.mark.parametrize("d", [4, 16, 24, 64])
@pytest.mark.parametrize("update_edges", [True, False])
def test_layer_matches_oracle(dev, cls_name, d, update_edges, factorise):
    from oracle import gnn_layers as og

    hl = _hl()
    ns, nr, e, B = 37, 29, 333, 3
    ei = _rand_ei(ns, nr, e, seed=d)
    torch.manual_seed(d)
    ref = getattr(og, cls_name)(ei, d, update_edges=update_edges)
    net = getattr(hl, cls_name)(ei, d, update_edges=update_edges)
    net.load_state_dict(ref.state_dict())
    net.to(dev)
    send, rec, edge = torch.randn(B, ns, d), torch.randn(B, nr, d), torch.randn(B, e, d)
    s1, r1, e1 = (t.clone().requires_grad_() for t in (send, rec, edge))
    s2, r2, e2 = (t.to(dev).requires_grad_() for t in (send, rec, edge))
    o1, o2 = ref(s1, r1, e1), net(s2, r2, e2)
    o1 = o1 if isinstance(o1, tuple) else (o1,)
    o2 = o2 if isinstance(o2, tuple) else (o2,)
    for a, b in zip(o2, o1):
        assert rel_err(a.cpu(), b) < TOL
    sum(o.square().sum() for o in o1).backward()
    sum(o.square().sum() for o in o2).backward()
    for a, b in ((s2, s1), (r2, r1), (e2, e1)):
        assert rel_err(a.grad.cpu(), b.grad) < TOL
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert rel_err(p.grad.cpu(), q.grad) < TOL, k


# cfg3 / cfg4 / cfg5 widths (BASELINE.json configs[2..4]) run on the workgroup-cooperative kernels
@pytest.mark.parametrize("cls_name,d,update_edges", [
    ("InteractionNet", 128, True), ("PropagationNet", 128, False), ("InteractionNet", 256, True),
    ("PropagationNet", 256, True), ("InteractionNet", 512, True), ("InteractionNet", 96, False),
    ("InteractionNet", 200, True),
])
def test_wide_layer_matches_oracle(dev, cls_name, d, update_edges, wide_family):
    from oracle import gnn_layers as og

    hl = _hl()
    ns, nr, e, B = 61, 47, 501, 2
    ei = _rand_ei(ns, nr, e, seed=d)
    torch.manual_seed(d)
    ref = getattr(og, cls_name)(ei, d, update_edges=update_edges)
    net = getattr(hl, cls_name)(ei, d, update_edges=update_edges)
    net.load_state_dict(ref.state_dict())
    net.to(dev)
    send, rec, edge = torch.randn(B, ns, d), torch.randn(B, nr, d), torch.randn(B, e, d)
    s1, r1, e1 = (t.clone().requires_grad_() for t in (send, rec, edge))
    s2, r2, e2 = (t.to(dev).requires_grad_() for t in (send, rec, edge))
    o1, o2 = ref(s1, r1, e1), net(s2, r2, e2)
    o1 = o1 if isinstance(o1, tuple) else (o1,)
    o2 = o2 if isinstance(o2, tuple) else (o2,)
    for a, b in zip(o2, o1):
        assert rel_err(a.cpu(), b) < TOL
    sum(o.square().sum() for o in o1).backward()
    sum(o.square().sum() for o in o2).backward()
    for a, b in ((s2, s1), (r2, r1), (e2, e1)):
        assert rel_err(a.grad.cpu(), b.grad) < TOL
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert rel_err(p.grad.cpu(), q.grad) < TOL, k


@pytest.mark.parametrize("cls_name,d,ns,nr,e", [
    ("InteractionNet", 128, 40, 7, 1500),     # in-degree ~214: every receiver spans several 32-row tiles (split tiles)
    ("PropagationNet", 128, 5, 300, 1400),    # out-degree ~280 senders, mean aggregation, sender residual
    ("InteractionNet", 256, 300, 300, 5),     # almost every node isolated; a handful of edges
    ("PropagationNet", 256, 33, 65, 2081),    # ragged: 65 tiles + 1 row
])
def test_wide_layers_on_awkward_graphs(dev, cls_name, d, ns, nr, e, wide_family):
    """Split receivers (atomic adds in the aggregate and in the receiver gradients), isolated nodes and ragged last
    tiles through both wide kernel families."""
    from oracle import gnn_layers as og

    hl = _hl()
    ei = _rand_ei(ns, nr, e, seed=e)
    torch.manual_seed(e)
    ref = getattr(og, cls_name)(ei, d)
    net = getattr(hl, cls_name)(ei, d)
    net.load_state_dict(ref.state_dict())
    net.to(dev)
    B = 2
    send, rec, edge = torch.randn(B, ns, d), torch.randn(B, nr, d), torch.randn(B, e, d)
    s1, r1, e1 = (t.clone().requires_grad_() for t in (send, rec, edge))
    s2, r2, e2 = (t.to(dev).requires_grad_() for t in (send, rec, edge))
    o1, o2 = ref(s1, r1, e1), net(s2, r2, e2)
    for a, b in zip(o2, o1):
        assert rel_err(a.cpu(), b) < TOL
    sum((o * o).sum() for o in o1).backward()
    sum((o * o).sum() for o in o2).backward()
    for a, b in ((s2, s1), (r2, r1), (e2, e1)):
        assert rel_err(a.grad.cpu(), b.grad) < TOL
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert rel_err(p.grad.cpu(), q.grad) < TOL, k


@pytest.mark.parametrize("kin,hid,dout,ln", [
    (3, 64, 64, True), (56, 64, 64, True), (64, 64, 17, False), (2, 16, 16, True), (7, 12, 5, False),
    (56, 256, 256, True), (256, 256, 17, False), (3, 128, 128, True), (128, 64, 64, True), (130, 96, 40, True),
    (512, 512, 34, False), (18, 512, 512, True),
    # output widths that are not whole 32-column blocks on the split-bf16 kernels (output_map: no LayerNorm): zero-padded W2,
    # dword output stores, 32-padded dz2 for the weight gradient; with a ragged input too; and the shapes that must stay generic
    (32, 32, 5, False), (64, 32, 17, False), (56, 64, 17, False), (64, 64, 31, False), (32, 64, 33, False),
])
def test_plain_mlp_matches_oracle(dev, kin, hid, dout, ln, wide_family):
    from oracle import gnn_layers as og

    hl = _hl()
    torch.manual_seed(kin)
    ref = og.make_mlp([kin, hid, dout], layer_norm=ln)
    net = hl.make_mlp([kin, hid, dout], layer_norm=ln)
    net.load_state_dict(ref.state_dict())
    net.to(dev)
    x = torch.randn(2, 1000, kin)
    x1, x2 = x.clone().requires_grad_(), x.to(dev).requires_grad_()
    y1, y2 = ref(x1), net(x2)
    assert rel_err(y2.cpu(), y1) < TOL
    y1.sin().sum().backward()
    y2.sin().sum().backward()
    assert rel_err(x2.grad.cpu(), x1.grad) < TOL
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert rel_err(p.grad.cpu(), q.grad) < TOL, k


@pytest.mark.parametrize("fused_wgrad", [True, False])
@pytest.mark.parametrize("d,shapes,ln", [
    (64, [(40003, 3), (131075, 3), (6561, 2), (33, 3)], True),   # > 1 016 waves x 32 rows: a wave walks several tiles (next-tile prefetch)
    (64, [(1, 3), (31, 1), (64, 3)], True),                      # fewer tiles than waves, a single row, a one-column member
    (32, [(5000, 3), (70001, 2)], True),
    (64, [(3000, 3), (2000, 5)], True),                          # a 5-column member: not a leaf-weight-gradient shape -> the plain grouped backward
    (64, [(50001, 3), (40000, 2)], False),                       # no LayerNorm: the prefetch has no xhat / rstd rows to ask for
])
def test_grouped_static_embedders_match_oracle(dev, monkeypatch, d, shapes, ln, fused_wgrad):
    """gnn_layers.grouped_mlp_forward: the embedders of the static features (graph/base.py:286-295) as ONE launch each way.
    With <= 3 input columns the backward accumulates the weight gradients in the kernel (NLAM_F_LEAF_WGRAD: z1 recomputed
    from the input row, the next tile's g_out / xhat / rstd / input rows requested a tile ahead); outputs and every parameter
    gradient against the oracle MLPs, with the fused weight gradients on and off, ragged last tiles included."""
    from neural_lam_amd import ops
    from oracle import gnn_layers as og

    hl = _hl()
    monkeypatch.setattr(ops, "FUSED_LEAF_WGRAD", fused_wgrad)
    torch.manual_seed(d + len(shapes))
    refs = [og.make_mlp([k, d, d], layer_norm=ln) for _, k in shapes]
    nets = []
    for r, (_, k) in zip(refs, shapes):
        n = hl.make_mlp([k, d, d], layer_norm=ln)
        n.load_state_dict(r.state_dict())
        nets.append(n.to(dev))
    xs = [torch.randn(rows, k) for rows, k in shapes]
    cots = [torch.randn(rows, d) for rows, _ in shapes]
    outs = hl.grouped_mlp_forward([(n, x.to(dev)) for n, x in zip(nets, xs)])
    for rep in range(2):   # twice: .grad accumulates (the kernel adds its partial sums to what is there)
        if rep:
            outs = hl.grouped_mlp_forward([(n, x.to(dev)) for n, x in zip(nets, xs)])
        torch.autograd.backward(outs, [c.to(dev) for c in cots])
    for r, x, c, o in zip(refs, xs, cots, outs):
        y = r(x)
        assert rel_err(o.cpu(), y) < TOL
        (2.0 * (y * c).sum()).backward()
    for i, (n, r) in enumerate(zip(nets, refs)):
        for (k, pp), (_, q) in zip(n.named_parameters(), r.named_parameters()):
            assert rel_err(pp.grad.cpu(), q.grad) < TOL, (i, k)


@pytest.mark.parametrize("mode", ["bf16x3", "bf16", "f32"])
@pytest.mark.parametrize("widths,hid,B,shared_last", [
    ([17, 17, 18, 4], 64, 2, True),    # the grid input features of the MEPS configuration (graph/base.py:275-283): 56 columns
    ([5, 5, 6, 1], 16, 3, True),       # the reference's dummy datastore: 17 columns -> not a multiple of 4: the concat fallback
    ([5, 5, 6], 32, 1, False),         # 16 columns, three pieces, hid = 32
    ([40, 24], 64, 2, False),          # a piece boundary inside the second 32-column unit
    ([64], 64, 2, False),              # a single piece
    ([60, 60, 12], 128, 2, False),     # 132 columns: wide kernels -> the fallback
])
def test_concat_folded_into_mlp_matches_oracle(dev, mode, widths, hid, B, shared_last):
    """ops.CatMLPFunction = ``make_mlp(...)(torch.cat(pieces, -1))`` (graph/base.py:275-286) with the concatenation folded into
    the kernel's first load: outputs, gradients of every piece that takes one (a batch-shared piece gets the batch sum) and all
    parameter gradients against the oracle; the fp32 matrix mode and shapes the piece path does not serve take the in-Function
    fallback (nlam_concat + the plain launch) and must agree just the same."""
    from neural_lam_amd import ops
    from oracle import gnn_layers as og

    hl = _hl()
    kin, N = sum(widths), 777
    torch.manual_seed(kin + hid)
    ref = og.make_mlp([kin, hid, hid])
    net = hl.make_mlp([kin, hid, hid])
    net.load_state_dict(ref.state_dict())
    net.to(dev)
    pieces = [torch.randn(B, N, w) for w in widths]
    if shared_last:   # expand_to_batch: the static features, one copy for the whole batch
        pieces[-1] = pieces[-1][:1].expand(B, N, widths[-1])
    p1 = [x.clone().requires_grad_() for x in pieces[:-1]] + [pieces[-1]]
    p2 = [x.to(dev).requires_grad_() for x in pieces[:-1]] + [pieces[-1][:1].to(dev).expand(B, N, widths[-1]) if shared_last else pieces[-1].to(dev)]
    old = ops.MATMUL_MODE
    ops.set_matmul_mode(mode)
    try:
        y1 = ref(torch.cat(p1, dim=-1))
        y2 = ops.CatMLPFunction.apply(*net.params(), *p2)
        tol = 3e-2 if mode == "bf16" else TOL
        assert rel_err(y2.cpu(), y1) < tol
        with torch.no_grad():   # inference: nothing is materialised, same result
            assert torch.equal(ops.CatMLPFunction.apply(*net.params(), *[t.detach() for t in p2]), y2.detach())
        y1.sin().sum().backward()
        y2.sin().sum().backward()
    finally:
        ops.set_matmul_mode(old)
    for a, b in zip(p2[:-1], p1[:-1]):
        assert rel_err(a.grad.cpu(), b.grad) < tol
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert rel_err(p.grad.cpu(), q.grad) < tol, k


@pytest.mark.parametrize("d,family,autocast", [(64, "auto", False), (128, "auto", False), (128, "wbf", False), (128, "wbf", True), (256, "wbf", False)])
def test_prepacked_weights_give_identical_steps(dev, tmp_path, d, family, autocast):
    """Weights packed ONCE per optimizer step (ops.WeightPacker): the narrow kernels fetch LDS images written by nlam_mlp_pack
    instead of splitting the fp32 matrices in every workgroup; the wide kernels read persistent ``wpack`` buffers filled by
    nlam_pack_records (NLAM_F_WPACK_READY) instead of running a pack launch in front of every forward / backward launch.
    Same terms, same layouts -> the step must be BIT-identical with the packer on and off (losses, every gradient, the weights
    after AdamW), eager and captured, over a rollout of two AR steps -- for the fp32-MFMA wide family ("auto" at this size),
    the split-bf16 one ("wbf") and bf16 autocast."""
    import contextlib

    from neural_lam_amd import _lib as L
    from neural_lam_amd import graph as G
    from neural_lam_amd import models as hm
    from neural_lam_amd import ops
    from neural_lam_amd.datastore import SyntheticDatastore
    from neural_lam_amd.trainer import Trainer

    lib = L.load()
    ds = SyntheticDatastore(30, 27, 5, 2, 1, root_path=tmp_path, boundary="random", seed=1)
    ext = ds.get_xy_extent("state")
    graph = G.normalise_graph(G.create_regular_grid_graph(ds.get_xy("state")), max(ext[1] - ext[0], ext[3] - ext[2]))
    N = ds.num_grid_points
    g = torch.Generator().manual_seed(5)
    batch = tuple(t.to(dev) for t in (torch.randn(2, 2, N, 5, generator=g), torch.randn(2, 2, N, 5, generator=g), torch.randn(2, 2, N, 6, generator=g)))

    def run(packed, use_graph):
        torch.manual_seed(11)
        fc = hm.ARForecaster(hm.GraphLAM(ds, graph=graph, hidden_dim=d, processor_layers=2), ds)
        tr = Trainer(hm.ForecasterStep(fc, ds).to(dev), lr=1e-3, use_graph=use_graph)
        tr._packer = ops.WeightPacker()
        tr._packer.enabled = packed
        amp = torch.autocast("cuda", dtype=torch.bfloat16) if autocast else contextlib.nullcontext()
        with amp:
            losses = [float(tr.step(*batch)) for _ in range(4)]
        torch.cuda.synchronize()
        return losses, tr.fp.grad.clone(), tr.fp.flat.clone(), tr._packer

    if family == "wbf":
        assert lib.nlam_set_tuning(L.TUNE_WBF_MIN_SUPERTILES, 0) == 0
    try:
        for use_graph in (False, True):
            l0, g0, w0, _ = run(False, use_graph)
            l1, g1, w1, pk = run(True, use_graph)
            if d <= 64:
                assert pk.table is not None and len(pk.table_entries) >= 10      # edge / node / grid MLPs + the embedders registered
                assert all(e.packed_step == pk.step_id for e in pk.table_entries)   # ... and were rewritten for the last step
            else:
                wide = [e for e in pk.wide.values() if e.buf is not None]
                assert len(wide) >= 10 and all(e.packed_step == pk.step_id for e in wide)
                kinds = {e.kind for e in wide}
                assert (1 in kinds) if autocast else ((3 in kinds) if family == "wbf" else kinds == {0})
            assert l0 == l1, (use_graph, l0, l1)
            assert torch.equal(g0, g1) and torch.equal(w0, w1), use_graph
    finally:
        assert lib.nlam_set_tuning(L.TUNE_WBF_MIN_SUPERTILES, 192) == 0


# ---------------------------------------------------------------------------
# reference tests/test_gnn_layers.py restated against the HIP classes
# ---------------------------------------------------------------------------
def _zero(mlp):
    with torch.no_grad():
        for p in mlp.parameters():
            p.zero_()


def test_zeroed_edge_mlp_messages_equal_sender(dev):  # test_gnn_layers.py:226-258
    hl = _hl()
    ei = _rand_ei(5, 4, 10, 0)
    pnet = hl.PropagationNet(ei, 8).to(dev)
    _zero(pnet.edge_mlp)
    send, rec, edge = torch.randn(5, 8, device=dev), torch.randn(4, 8, device=dev), torch.randn(10, 8, device=dev)
    _, msgs = pnet.propagate(pnet.edge_index, x=torch.cat((rec, send), dim=0), edge_attr=edge)
    assert torch.allclose(msgs, send[ei[0].to(dev)], atol=1e-6)


def test_zeroed_aggr_mlp_residual_targets_aggregate(dev):  # :260-295
    hl = _hl()
    ei = _rand_ei(5, 4, 10, 0)
    pnet = hl.PropagationNet(ei, 8).to(dev)
    _zero(pnet.aggr_mlp)
    send, rec, edge = torch.randn(5, 8, device=dev), torch.randn(4, 8, device=dev), torch.randn(10, 8, device=dev)
    rec_out, _ = pnet(send, rec, edge)
    aggr, _ = pnet.propagate(pnet.edge_index, x=torch.cat((rec, send), dim=0), edge_attr=edge)
    assert torch.allclose(rec_out, aggr, atol=1e-6) and not torch.allclose(rec_out, rec, atol=1e-3)


def test_edge_residual_identity(dev):  # :359-384
    hl = _hl()
    ei = _rand_ei(5, 4, 10, 0)
    inet = hl.InteractionNet(ei, 8).to(dev)
    send, rec, edge = torch.randn(5, 8, device=dev), torch.randn(4, 8, device=dev), torch.randn(10, 8, device=dev)
    _, edge_out = inet(send, rec, edge)
    _, msgs = inet.propagate(inet.edge_index, x=torch.cat((rec, send), dim=0), edge_attr=edge)
    assert torch.allclose(edge_out, edge + msgs, atol=1e-5)


def test_batch_independence(dev):  # :395-439
    hl = _hl()
    ei = _rand_ei(5, 4, 10, 0)
    inet = hl.InteractionNet(ei, 8).to(dev)
    send, rec, edge = torch.randn(3, 5, 8, device=dev), torch.randn(3, 4, 8, device=dev), torch.randn(3, 10, 8, device=dev)
    rb, eb = inet(send, rec, edge)
    assert rb.shape == (3, 4, 8) and eb.shape == (3, 10, 8)
    for b in range(3):
        r1, e1 = inet(send[b], rec[b], edge[b])
        assert torch.allclose(rb[b], r1, atol=1e-6) and torch.allclose(eb[b], e1, atol=1e-6)


def test_disconnected_receiver_identity(dev):  # :628-655
    hl = _hl()
    ei = torch.tensor([[0, 1, 2, 0], [0, 0, 2, 2]])  # receiver 1 has no in-edges
    inet = hl.InteractionNet(ei, 8).to(dev)
    send, rec, edge = torch.randn(3, 8, device=dev), torch.randn(3, 8, device=dev), torch.randn(4, 8, device=dev)
    rec_out, _ = inet(send, rec, edge)
    expected = rec[1] + inet.aggr_mlp(torch.cat((rec, torch.zeros_like(rec)), dim=-1))[1]
    assert torch.allclose(rec_out[1], expected, atol=1e-5)


def test_gradients_reach_all_inputs_and_parameters(dev):  # :513-585
    hl = _hl()
    ei = _rand_ei(5, 4, 10, 0)
    net = hl.PropagationNet(ei, 8).to(dev)
    send, rec, edge = (torch.randn(n, 8, device=dev, requires_grad=True) for n in (5, 4, 10))
    r, e = net(send, rec, edge)
    (r.sum() + e.sum()).backward()
    for t in (send, rec, edge):
        assert t.grad is not None and torch.isfinite(t.grad).all() and t.grad.abs().sum() > 0
    for k, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k


def test_topologies_and_deep_stack_stay_finite(dev):  # :596-738
    hl = _hl()
    # 1x1 graph, self loops
    for ei in (torch.tensor([[0], [0]]), torch.tensor([[0, 1, 2], [0, 1, 2]])):
        n = int(ei.max()) + 1
        net = hl.InteractionNet(ei, 8).to(dev)
        r, e = net(torch.randn(n, 8, device=dev), torch.randn(n, 8, device=dev), torch.randn(ei.shape[1], 8, device=dev))
        assert torch.isfinite(r).all() and torch.isfinite(e).all()
    # 8 stacked layers
    ei = _rand_ei(20, 20, 80, 1)
    seq = hl.make_gnn_seq(ei, 8, 1, 8).to(dev)
    m, e = seq(torch.randn(20, 8, device=dev), torch.randn(80, 8, device=dev))
    assert torch.isfinite(m).all() and torch.isfinite(e).all()
    # high in-degree (~167 per receiver) with mean aggregation stays bounded
    ei = torch.stack([torch.arange(500), torch.arange(500) % 3])
    pnet = hl.PropagationNet(ei, 8).to(dev)
    r, _ = pnet(torch.randn(500, 8, device=dev), torch.randn(3, 8, device=dev), torch.randn(500, 8, device=dev))
    assert torch.isfinite(r).all() and r.abs().max() < 1000


# ---------------------------------------------------------------------------
# full MEPS size (BASELINE.json configs[1]): size-independent properties
# ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def meps(dev):
    from neural_lam_amd import graph as G
    from neural_lam_amd import models as hm
    from neural_lam_amd.datastore import meps_like_datastore

    ds = meps_like_datastore("/tmp/nlam_test_meps")
    ext = ds.get_xy_extent("state")
    raw = G.create_regular_grid_graph(ds.get_xy("state"))
    graph = G.normalise_graph(raw, max(ext[1] - ext[0], ext[3] - ext[2]))
    torch.manual_seed(42)
    fc = hm.ARForecaster(hm.GraphLAM(ds, graph=graph, hidden_dim=64, processor_layers=4), ds)
    step = hm.ForecasterStep(fc, ds).to(dev)
    return ds, raw, step


def test_meps_aggregation_matches_index_add(dev, meps):
    """m2g / g2m segment reduction at full size == torch index_add on the messages."""
    hl = _hl()
    _, raw, _ = meps
    for name in ("m2g", "g2m"):
        ei = raw[f"{name}_edge_index"]
        ns, nr, E = int(ei[0].max()) + 1, int(ei[1].max()) + 1, ei.shape[1]
        torch.manual_seed(1)
        net = hl.InteractionNet(ei, 64, update_edges=False).to(dev)
        send, rec, edge = torch.randn(ns, 64, device=dev), torch.randn(nr, 64, device=dev), torch.randn(E, 64, device=dev)
        aggr, msgs = net.propagate(net.edge_index, x=torch.cat((rec, send), dim=0), edge_attr=edge)
        ref = torch.zeros(nr, 64, device=dev, dtype=torch.float64).index_add_(0, ei[1].to(dev), msgs.double())
        assert rel_err(aggr.double().cpu(), ref.cpu()) < 1e-5


def test_meps_step_is_deterministic_and_batch_independent(dev, meps):
    ds, _, step = meps
    N = ds.num_grid_points
    torch.manual_seed(123)
    init, target, forcing = torch.randn(2, 2, N, 17, device=dev), torch.randn(2, 1, N, 17, device=dev), torch.randn(2, 1, N, 18, device=dev)
    with torch.no_grad():
        p1, _ = step(init, target, forcing)
        p2, _ = step(init, target, forcing)
        assert torch.equal(p1, p2)  # no atomics on this graph: bit-reproducible
        for b in range(2):
            pb, _ = step(init[b : b + 1], target[b : b + 1], forcing[b : b + 1])
            assert torch.allclose(pb[0], p1[b], atol=1e-5)
        # boundary nodes carry the true state, interior nodes the prediction (test_prediction_model_classes.py:38-73)
        bm = torch.tensor(ds.boundary_mask.values, device=dev).bool()
        assert torch.equal(p1[:, 0, bm], target[:, 0, bm]) and torch.isfinite(p1).all()


def test_meps_gradients_are_reproducible(dev, meps):
    _, _, step = meps
    ds = meps[0]
    N = ds.num_grid_points
    torch.manual_seed(5)
    init, target, forcing = torch.randn(1, 2, N, 17, device=dev), torch.randn(1, 1, N, 17, device=dev), torch.randn(1, 1, N, 18, device=dev)
    grads = []
    for _ in range(2):
        step.zero_grad(set_to_none=True)
        _, loss = step(init, target, forcing)
        loss.backward()
        grads.append({k: p.grad.clone() for k, p in step.named_parameters()})
        assert torch.isfinite(loss)
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]), k  # two-stage partial sums: fixed order


def test_fused_adamw_matches_torch(dev):
    from neural_lam_amd.ops import AdamWFlat

    torch.manual_seed(0)
    p = torch.randn(10001, device=dev)
    ref_p = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-3, betas=(0.9, 0.95))
    g = torch.zeros_like(p)
    mine = AdamWFlat(p, g, lr=1e-3)
    for i in range(5):
        grad = torch.randn_like(p)
        g.copy_(grad)
        ref_p.grad = grad.clone()
        opt.step()
        mine.step()
    assert rel_err(p.cpu(), ref_p.detach().cpu()) < 1e-5


# ---------------------------------------------------------------------------
# matrix-path modes, determinism, fused loss, HIP-graph step
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("mode,tol", [("f32", 1e-4), ("bf16x3", 1e-4), ("bf16x2", 1e-4), ("bf16", 3e-2)])
def test_matmul_modes_match_oracle(dev, mode, tol):
    """f32 = fp32 MFMA; bf16xN = operands split into N bf16 terms on the bf16 matrix cores (fp32 accumulate).
    bf16x3 is the default and fp32-class; plain bf16 is the autocast-like mode with its own (stated) tolerance."""
    from neural_lam_amd import ops
    from oracle import gnn_layers as og

    hl = _hl()
    ns, nr, e, B, d = 61, 47, 1501, 2, 64
    ei = _rand_ei(ns, nr, e, seed=7)
    torch.manual_seed(7)
    ref = og.InteractionNet(ei, d)
    net = hl.InteractionNet(ei, d)
    net.load_state_dict(ref.state_dict())
    net.to(dev)
    send, rec, edge = torch.randn(B, ns, d), torch.randn(B, nr, d), torch.randn(B, e, d)
    s1, r1, e1 = (t.clone().requires_grad_() for t in (send, rec, edge))
    s2, r2, e2 = (t.to(dev).requires_grad_() for t in (send, rec, edge))
    old = ops.MATMUL_MODE
    try:
        ops.set_matmul_mode(mode)
        o2 = net(s2, r2, e2)
        sum(o.square().sum() for o in o2).backward()
    finally:
        ops.set_matmul_mode(old)
    o1 = ref(s1, r1, e1)
    sum(o.square().sum() for o in o1).backward()
    for a, b in zip(o2, o1):
        assert rel_err(a.cpu(), b) < tol
    for a, b in ((s2, s1), (r2, r1), (e2, e1)):
        assert rel_err(a.grad.cpu(), b.grad) < tol
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert rel_err(p.grad.cpu(), q.grad) < tol, k


@pytest.mark.parametrize("cls_name,d,factorised,half", [("InteractionNet", 256, True, 1), ("InteractionNet", 256, False, 0), ("PropagationNet", 256, False, 3),
                                                        ("InteractionNet", 512, True, 1), ("InteractionNet", 512, False, 0), ("InteractionNet", 384, True, 0)])
def test_bf16_storage_of_saved_tensors_under_autocast(dev, cls_name, d, factorised, half):
    """NLAM_F_STORE_BF16: inside torch.autocast(bfloat16) the one-term wide kernels keep z1, xhat, dz1 and dz2 as bf16 rows
    (what --precision bf16-mixed makes of the reference's nn.Linear outputs and their gradients, train_model.py:163-168).
    Against the same launches with fp32 storage: the forward is bit-identical (only what is SAVED changes), every gradient
    agrees to bf16 rounding of the saved tensors (max-norm 2e-2, cosine > 0.9995); against the fp32 oracle the autocast
    tolerance 3e-2 holds.  Plain and factorised edge MLP (the sender gradient is a segment sum over bf16 dz1 rows), 8-wave and
    4-wave instantiations, a batch, an edge set of several super tiles per workgroup."""
    from neural_lam_amd import _lib as L
    from neural_lam_amd import ops
    from oracle import gnn_layers as og

    hl = _hl()
    lib = L.load()
    ns, nr, e, B = 301, 257, 9001, 2
    ei = _rand_ei(ns, nr, e, seed=13)
    torch.manual_seed(13)
    ref = getattr(og, cls_name)(ei, d)
    send, rec, edge = torch.randn(B, ns, d), torch.randn(B, nr, d), torch.randn(B, e, d)
    old = (hl.FACTORISE_MIN_EDGES_WIDE, hl.FACTORISE_MIN_WORK_WIDE, hl.FACTORISE_MIN_WIDTH_WIDE, ops.STORE_BF16)
    res = {}
    try:
        assert lib.nlam_set_tuning(L.TUNE_WBF_MIN_SUPERTILES, 0) == 0 and lib.nlam_set_tuning(L.TUNE_WBF_HALF, half) == 0
        hl.FACTORISE_MIN_WORK_WIDE, hl.FACTORISE_MIN_WIDTH_WIDE = 0, 0
        hl.FACTORISE_MIN_EDGES_WIDE = 0 if factorised else 1 << 30
        for store in (False, True):
            ops.STORE_BF16 = store
            net = getattr(hl, cls_name)(ei, d)
            net.load_state_dict(ref.state_dict())
            net.to(dev)
            args = [t.to(dev).requires_grad_() for t in (send, rec, edge)]
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = net(*args)
                loss = sum(o.float().square().sum() for o in out)
            loss.backward()
            res[store] = ([o.detach() for o in out], [a.grad for a in args], {k: p.grad for k, p in net.named_parameters()})
    finally:
        hl.FACTORISE_MIN_EDGES_WIDE, hl.FACTORISE_MIN_WORK_WIDE, hl.FACTORISE_MIN_WIDTH_WIDE, ops.STORE_BF16 = old
        assert lib.nlam_set_tuning(L.TUNE_WBF_MIN_SUPERTILES, 192) == 0 and lib.nlam_set_tuning(L.TUNE_WBF_HALF, 1) == 0

    def cos(a, b):
        a, b = a.double().reshape(-1), b.double().reshape(-1)
        return float((a @ b) / (a.norm() * b.norm()).clamp(min=1e-30))

    for a, b in zip(res[True][0], res[False][0]):
        assert torch.equal(a, b)
    for tag, a, b in [("d_in", x, y) for x, y in zip(res[True][1], res[False][1])] + [(k, res[True][2][k], res[False][2][k]) for k in res[True][2]]:
        assert rel_err(a.cpu(), b.cpu()) < 2e-2 and cos(a, b) > 0.9995, (tag, rel_err(a.cpu(), b.cpu()), cos(a, b))
    s1, r1, e1 = (t.clone().requires_grad_() for t in (send, rec, edge))
    o1 = ref(s1, r1, e1)
    sum(o.square().sum() for o in o1).backward()
    for a, b in zip(res[True][0], o1):
        assert rel_err(a.cpu(), b.detach()) < 3e-2
    for a, b in zip(res[True][1], (s1.grad, r1.grad, e1.grad)):
        assert rel_err(a.cpu(), b) < 3e-2


@pytest.mark.parametrize("kin,d,autocast", [(3, 256, False), (2, 512, True), (3, 128, False), (5, 512, False)])
def test_wide_embedder_of_a_few_static_columns(dev, kin, d, autocast):
    """The embedders of the static edge / mesh features at hidden widths above 64 (graph/base.py:286-295: [len, dx, dy] -> d):
    the input is zero-padded to a multiple of 4 columns so that the launch takes the split-bf16 super-tile kernels instead of
    the fp32 one-tile kernels; outputs and every parameter gradient (the first weight's through the padded copy) match the
    oracle, with and without a trainer-style direct gradient buffer."""
    import contextlib

    from neural_lam_amd import _lib as L
    from oracle import gnn_layers as og

    hl = _hl()
    old_min, hl.PAD_EMBEDDER_MIN_WIDTH = hl.PAD_EMBEDDER_MIN_WIDTH, 64   # (the product pads from d > 256)
    torch.manual_seed(kin + d)
    ref, net = og.make_mlp([kin, d, d]), hl.make_mlp([kin, d, d])
    net.load_state_dict(ref.state_dict())
    net.to(dev)
    x = torch.randn(30011, kin)
    y1 = ref(x)
    c = torch.randn_like(y1)
    (y1 * c).sum().backward()
    tol = 3e-2 if autocast else TOL
    try:
        assert L.load().nlam_set_tuning(L.TUNE_WBF_MIN_SUPERTILES, 0) == 0
        amp = torch.autocast("cuda", dtype=torch.bfloat16) if autocast else contextlib.nullcontext()
        for _ in range(2):   # the second pass re-uses the cached padded input and the persistent padded weight
            net.zero_grad()
            with amp:
                y2 = net(x.to(dev))
            (y2.float() * c.to(dev)).sum().backward()
            assert rel_err(y2.float().cpu(), y1.detach()) < tol
            for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
                assert p.grad is not None and rel_err(p.grad.cpu(), q.grad) < tol, k
    finally:
        assert L.load().nlam_set_tuning(L.TUNE_WBF_MIN_SUPERTILES, 192) == 0
        hl.PAD_EMBEDDER_MIN_WIDTH = old_min
    assert net._xpad is not None and net._xpad[1].shape[-1] % 4 == 0


def test_segment_sum_over_bf16_rows(dev):
    """nlam_segment_sum_bf16 (sender gradient of a factorised edge MLP running with bf16 storage) against index_add in fp64."""
    import ctypes as C

    from neural_lam_amd import _lib as L

    g = torch.Generator().manual_seed(4)
    B, rows, nseg, width = 2, 5000, 300, 256
    lens = torch.randint(0, 34, (nseg,), generator=g)
    lens[7] = 0
    lens = (lens.double() * (rows / float(lens.sum()))).floor().long()
    lens[-1] += rows - int(lens.sum())
    ptr = torch.zeros(nseg + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(lens, 0)
    order = torch.randperm(rows, generator=g).to(torch.int32)
    x = torch.randn(B, rows, width, generator=g).bfloat16()
    seg_of_pos = torch.repeat_interleave(torch.arange(nseg), lens)
    ref = torch.zeros(B, nseg, width, dtype=torch.float64).index_add_(1, seg_of_pos, x[:, order.long()].double())
    xd, pd, od, out = x.to(dev), ptr.to(dev), order.to(dev), torch.empty(B, nseg, width, device=dev)   # (named: temporaries would be freed before the launch)
    rc = L.load().nlam_segment_sum_bf16(xd.data_ptr(), rows * width, pd.data_ptr(), od.data_ptr(), None, out.data_ptr(),
                                        nseg, width, B, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    assert rel_err(out.cpu().double(), ref) < 1e-5


@pytest.mark.parametrize("d", [64, 128])
def test_autocast_region_uses_bf16_operands(dev, d, wide_family):
    """Lightning ``--precision bf16-mixed`` wraps the step in torch.autocast: the fused MLPs then take plain bf16
    operands (what autocast does to the reference's nn.Linear), accept bf16 activations, keep fp32 outputs, and give
    bit-for-bit what set_matmul_mode("bf16") gives outside autocast.  Tolerance against the fp32 oracle: 3e-2."""
    from neural_lam_amd import ops
    from oracle import gnn_layers as og

    hl = _hl()
    ns, nr, e, B = 61, 47, 1501, 2
    ei = _rand_ei(ns, nr, e, seed=11)
    torch.manual_seed(11)
    ref = og.InteractionNet(ei, d)
    net = hl.InteractionNet(ei, d)
    net.load_state_dict(ref.state_dict())
    net.to(dev)
    send, rec, edge = torch.randn(B, ns, d), torch.randn(B, nr, d), torch.randn(B, e, d)
    s1, r1, e1 = (t.clone().requires_grad_() for t in (send, rec, edge))
    s2, r2, e2 = (t.to(dev).requires_grad_() for t in (send, rec, edge))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        o2 = net(s2, r2, e2)
        loss = sum(o.float().square().sum() for o in o2)
    loss.backward()   # outside the region, as Lightning does: the launch re-uses the forward's matrix mode
    assert all(o.dtype == torch.float32 for o in o2)
    o1 = ref(s1, r1, e1)
    sum(o.square().sum() for o in o1).backward()
    for a, b in zip(o2, o1):
        assert rel_err(a.detach().cpu(), b) < 3e-2
    for a, b in ((s2, s1), (r2, r1), (e2, e1)):
        assert rel_err(a.grad.cpu(), b.grad) < 3e-2
    old = ops.MATMUL_MODE
    try:
        ops.set_matmul_mode("bf16")
        with torch.no_grad():
            o3 = net(s2, r2, e2)
            # bf16 activations coming out of an autocast Linear upstream are widened, not rejected
            with torch.autocast("cuda", dtype=torch.bfloat16):
                o4 = net(s2.bfloat16(), r2.bfloat16(), e2.bfloat16())
    finally:
        ops.set_matmul_mode(old)
    for a, b in zip(o2, o3):
        assert torch.equal(a.detach(), b)
    for a, b in zip(o4, o1):
        assert a.dtype == torch.float32 and rel_err(a.cpu(), b) < 5e-2


@pytest.mark.parametrize("mode", ["f32", "bf16x3"])
def test_layer_is_bit_reproducible_at_meps_size(dev, mode):
    """Same inputs -> identical bits, 8 runs (no atomics on the MEPS graphs; the split-bf16 MFMA groups keep
    their operands in distinct registers -- see mma_split_lds in csrc/nlam_hip.hip)."""
    from neural_lam_amd import graph as G
    from neural_lam_amd import ops

    hl = _hl()
    raw = G.create_regular_grid_graph(G.regular_grid_xy(238, 268))
    ei = raw["g2m_edge_index"]
    ns, nr, E = int(ei[0].max()) + 1, int(ei[1].max()) + 1, ei.shape[1]
    torch.manual_seed(0)
    net = hl.InteractionNet(ei, 64, update_edges=False).to(dev)
    send, rec, edge = (torch.randn(1, n, 64, device=dev, requires_grad=True) for n in (ns, nr, E))
    old = ops.MATMUL_MODE
    try:
        ops.set_matmul_mode(mode)
        runs = []
        for _ in range(8):
            for t in (send, rec, edge):
                t.grad = None
            net.zero_grad(set_to_none=True)
            out = net(send, rec, edge)
            out.square().sum().backward()
            runs.append([out.detach().clone(), send.grad.clone(), rec.grad.clone(), edge.grad.clone()]
                        + [p.grad.clone() for p in net.parameters()])
    finally:
        ops.set_matmul_mode(old)
    for r in runs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(r, runs[0]))


@pytest.mark.parametrize("d,which,update_edges", [(128, "m2g", False), (256, "m2m", True)])
def test_wide_kernel_families_agree_at_meps_size(dev, d, which, update_edges):
    """Full-size check without the oracle (too slow at this size): the fp32 MFMA kernels (one tile per workgroup,
    weights streamed per tile) and the split-bf16 super-tile kernels share no device code beyond the tile schedule,
    so agreement of outputs and of every gradient within 1e-4 at the MEPS edge sets pins both; the split-bf16 run
    must also be bit-reproducible."""
    from neural_lam_amd import graph as G
    from neural_lam_amd import ops

    hl = _hl()
    raw = G.create_regular_grid_graph(G.regular_grid_xy(238, 268))
    ei = raw[f"{which}_edge_index"] if which != "m2m" else raw["m2m_edge_index"][0]
    ns, nr, E = int(ei[0].max()) + 1, int(ei[1].max()) + 1, ei.shape[1]
    torch.manual_seed(0)
    net = hl.InteractionNet(ei, d, update_edges=update_edges).to(dev)
    send, rec, edge = (torch.randn(1, n, d, device=dev, requires_grad=True) for n in (ns, nr, E))

    def run(mode):
        old = ops.MATMUL_MODE
        try:
            ops.set_matmul_mode(mode)
            for t in (send, rec, edge):
                t.grad = None
            net.zero_grad(set_to_none=True)
            out = net(send, rec, edge)
            outs = out if isinstance(out, tuple) else (out,)
            sum(o.square().sum() for o in outs).backward()
            return [o.detach().clone() for o in outs] + [send.grad.clone(), rec.grad.clone(), edge.grad.clone()] + [
                p.grad.clone() for p in net.parameters()]
        finally:
            ops.set_matmul_mode(old)

    a, b, c = run("f32"), run("bf16x3"), run("bf16x3")
    for x, y in zip(a, b):
        assert rel_err(y.cpu(), x.cpu()) < TOL
    assert all(torch.equal(x, y) for x, y in zip(b, c))


@pytest.mark.parametrize("d", [64, 128])
def test_same_tensor_as_sender_and_receiver(dev, d, wide_family):
    """Mesh <-> mesh layers get one tensor as both node sets (graph_lam.py:168-188): the sender-side gradient is then
    accumulated into the receiver-side buffer inside the library (nlam_segment_sum_acc) and must equal the oracle's
    sum of the two autograd contributions."""
    from oracle import gnn_layers as og

    hl = _hl()
    n, e, B = 57, 1311, 2
    ei = _rand_ei(n, n, e, seed=3)
    torch.manual_seed(3)
    ref = og.InteractionNet(ei, d)
    net = hl.InteractionNet(ei, d)
    net.load_state_dict(ref.state_dict())
    net.to(dev)
    x, edge = torch.randn(B, n, d), torch.randn(B, e, d)
    x1, e1 = x.clone().requires_grad_(), edge.clone().requires_grad_()
    x2, e2 = x.to(dev).requires_grad_(), edge.to(dev).requires_grad_()
    o1, o2 = ref(x1, x1, e1), net(x2, x2, e2)
    for a, b in zip(o2, o1):
        assert rel_err(a.cpu(), b) < TOL
    sum((o * o).sum() for o in o1).backward()
    sum((o * o).sum() for o in o2).backward()
    assert rel_err(x2.grad.cpu(), x1.grad) < TOL and rel_err(e2.grad.cpu(), e1.grad) < TOL


def test_fused_wmse_loss_matches_reference_formula(dev):
    """nlam_wmse_fwd/bwd == metrics.wmse + mask_and_reduce_metric + batch/step means (metrics.py:37-137, module.py:463-510)."""
    from neural_lam_amd import models as hm
    from neural_lam_amd.ops import WmseLossFunction

    torch.manual_seed(3)
    B, T, N, V = 2, 3, 1000, 17
    pred = torch.randn(B, T, N, V, device=dev, requires_grad=True)
    target = torch.randn(B, T, N, V, device=dev)
    std = torch.rand(V, device=dev) + 0.5
    mask = torch.rand(N, device=dev) > 0.3
    w = mask.float() / mask.sum()
    loss = WmseLossFunction.apply(pred, target, 1.0 / (std * std), w)
    loss.backward()
    p2 = pred.detach().clone().requires_grad_()
    ref = torch.mean(torch.mean(hm.wmse(p2, target, std, mask=mask), dim=0))
    ref.backward()
    assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref))
    assert rel_err(pred.grad.cpu(), p2.grad.cpu()) < 1e-5


@pytest.mark.parametrize("model,kw", [
    ("graph_lam", dict(hidden_dim=16, processor_layers=2)),
    ("graph_lam", dict(hidden_dim=16, processor_layers=2, output_clamping_lower={"state_var_0": 0.0},
                       output_clamping_upper={"state_var_0": 9.0, "state_var_2": 5.0})),
    ("graph_lam", dict(hidden_dim=128, processor_layers=1)),
    ("graph_lam", dict(hidden_dim=16, processor_layers=1, output_std=True, g2m_gnn_type="PropagationNet",
                       m2g_gnn_type="PropagationNet", mesh_aggr="mean")),
    ("hi_lam", dict(hidden_dim=16, processor_layers=2)),
    ("hi_lam_parallel", dict(hidden_dim=16, processor_layers=2)),
])
def test_hip_graph_step_equals_eager_step(dev, tmp_path, model, kw):
    """Trainer(use_graph=True) replays zero-grad + fwd + loss + bwd from one HIP graph: same bits as eager, for every
    model family (the chunked HiLAMParallel path and the clamped state update included: nothing in a step may touch
    the host during capture)."""
    from neural_lam_amd import graph as G
    from neural_lam_amd import models as hm
    from neural_lam_amd.datastore import SyntheticDatastore
    from neural_lam_amd.trainer import Trainer

    hier = model != "graph_lam"

    def make(use_graph):
        ds = SyntheticDatastore(81 if hier else 30, 30 if hier else 27, 5, 2, 1, root_path=tmp_path, boundary="random", seed=1)
        ext = ds.get_xy_extent("state")
        raw = G.create_regular_grid_graph(ds.get_xy("state"), n_max_levels=3 if hier else None, hierarchical=hier)
        graph = G.normalise_graph(raw, max(ext[1] - ext[0], ext[3] - ext[2]))
        torch.manual_seed(1)
        fc = hm.ARForecaster(hm.MODELS[model](ds, graph=graph, **kw), ds)
        tr = Trainer(hm.ForecasterStep(fc, ds).to(dev), lr=1e-3, use_graph=use_graph)
        return ds, tr

    ds, t_eager = make(False)
    _, t_graph = make(True)
    N = ds.num_grid_points
    g = torch.Generator().manual_seed(0)
    for _ in range(3):
        batch = [torch.randn(1, 2, N, 5, generator=g).to(dev), torch.randn(1, 2, N, 5, generator=g).to(dev),
                 torch.randn(1, 2, N, 6, generator=g).to(dev)]
        le, lg = float(t_eager.step(*batch)), float(t_graph.step(*batch))
        # every family, the chunked HiLAMParallel path included, is deterministic (fixed-order segment sums, no atomics)
        assert le == lg
        assert torch.equal(t_eager.fp.flat, t_graph.fp.flat) and torch.equal(t_eager.fp.grad, t_graph.fp.grad)
    assert t_graph._graph is not None   # really captured, not the eager fallback


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("d,L,T", [(128, 3, 4), (64, 4, 5)])
def test_rollout_gradients_identical_with_and_without_wgrad_side_streams(dev, tmp_path, d, L, T, use_graph):
    """In a rollout every MLP is back-propagated once per AR step and each time accumulates into the same ``.grad``
    (read-modify-write).  With the weight-gradient work on side streams those accumulations must stay ordered: the
    side stream is a function of the MLP, not of the call (the MLP counts here, 13 and 15, are not multiples of the 4
    streams, so a round-robin deal would put consecutive uses of one MLP on different streams).  Bitwise equality with
    the serial (no side stream) step over several steps, eager and captured."""
    from neural_lam_amd import graph as G
    from neural_lam_amd import models as hm
    from neural_lam_amd.datastore import SyntheticDatastore
    from neural_lam_amd.trainer import Trainer

    def make(overlap):
        ds = SyntheticDatastore(60, 54, 5, 2, 1, root_path=tmp_path, boundary="random", seed=1)
        ext = ds.get_xy_extent("state")
        raw = G.create_regular_grid_graph(ds.get_xy("state"))
        graph = G.normalise_graph(raw, max(ext[1] - ext[0], ext[3] - ext[2]))
        torch.manual_seed(1)
        fc = hm.ARForecaster(hm.GraphLAM(ds, graph=graph, hidden_dim=d, processor_layers=L), ds)
        return ds, Trainer(hm.ForecasterStep(fc, ds).to(dev), lr=1e-3, use_graph=use_graph, overlap_wgrad=overlap)

    ds, t_par = make(True)
    _, t_ser = make(False)
    N = ds.num_grid_points
    g = torch.Generator().manual_seed(0)
    for _ in range(3):
        batch = [torch.randn(1, 2, N, 5, generator=g).to(dev), torch.randn(1, T, N, 5, generator=g).to(dev),
                 torch.randn(1, T, N, 6, generator=g).to(dev)]
        lp, ls = float(t_par.step(*batch)), float(t_ser.step(*batch))
        assert lp == ls
        assert torch.equal(t_par.fp.grad, t_ser.fp.grad) and torch.equal(t_par.fp.flat, t_ser.fp.flat)
    assert (t_par._graph is not None) == use_graph


@pytest.mark.parametrize("use_graph", [False, True])
def test_rollout_gradient_handover_equals_autograd_accumulation(dev, tmp_path, use_graph):
    """The static edge embeddings are computed once per rollout and consumed by one edge launch per AR step
    (models/forecasters/autoregressive.py:63-149).  With ops.ROLLOUT_ACC_ON the first back-propagated step's gradient buffer is
    what autograd holds and the later steps add into it inside the backward kernel (NLAM_F_ACC_DSRC0, split-bf16 super-tile
    family) instead of reporting T gradients for autograd to add.  Same sums up to the association of one addition: the
    step with the hand-over equals the step without it to fp32 rounding, and the hand-over really happens."""
    from neural_lam_amd import _lib as L
    from neural_lam_amd import graph as G
    from neural_lam_amd import models as hm
    from neural_lam_amd import ops
    from neural_lam_amd.datastore import SyntheticDatastore
    from neural_lam_amd.trainer import Trainer

    lib = L.load()
    assert lib.nlam_set_tuning(L.TUNE_WBF_MIN_SUPERTILES, 0) == 0   # the super-tile family at test sizes
    T = 3
    try:
        def make(on):
            ops.ROLLOUT_ACC_ON = on
            ds = SyntheticDatastore(60, 54, 5, 2, 1, root_path=tmp_path, boundary="random", seed=1)
            ext = ds.get_xy_extent("state")
            raw = G.create_regular_grid_graph(ds.get_xy("state"))
            graph = G.normalise_graph(raw, max(ext[1] - ext[0], ext[3] - ext[2]))
            torch.manual_seed(1)
            fc = hm.ARForecaster(hm.GraphLAM(ds, graph=graph, hidden_dim=128, processor_layers=2), ds)
            return ds, Trainer(hm.ForecasterStep(fc, ds).to(dev), lr=1e-3, use_graph=use_graph)

        ds, t_on = make(True)
        N = ds.num_grid_points
        g = torch.Generator().manual_seed(0)
        batch = [torch.randn(1, 2, N, 5, generator=g).to(dev), torch.randn(1, T, N, 5, generator=g).to(dev),
                 torch.randn(1, T, N, 6, generator=g).to(dev)]
        ops.ROLLOUT_ACC_STATS["accumulated"] = 0
        l_on = float(t_on.step(*batch))
        assert ops.ROLLOUT_ACC_STATS["accumulated"] >= 2 * (T - 1)   # at least the g2m and m2g edge embeddings, once per earlier step
        g_on = t_on.fp.grad.clone()
        _, t_off = make(False)
        ops.ROLLOUT_ACC_STATS["accumulated"] = 0
        l_off = float(t_off.step(*batch))
        assert ops.ROLLOUT_ACC_STATS["accumulated"] == 0
        assert l_on == l_off
        err = float((g_on - t_off.fp.grad).abs().max()) / float(t_off.fp.grad.abs().max())
        assert err < 2e-5, err   # (one addition per shared tensor and AR step associates differently: fp32 rounding, amplified by the embedders' LayerNorm backward)
    finally:
        ops.ROLLOUT_ACC_ON = True
        assert lib.nlam_set_tuning(L.TUNE_WBF_MIN_SUPERTILES, 192) == 0


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("family", ["graph_lam", "hi_lam"])
def test_early_leaf_backward_gives_the_same_step(dev, tmp_path, family, use_graph):
    """``Trainer(early_leaf_backward=True)`` (NLAM_EARLY_LEAF=1, off by default) cuts the autograd graph behind the
    embedders of input data / static features and runs their backward from a post-accumulate hook (a nested
    backward, ops.early_backward_leaf).  It requires exactly one backward per forward, which is what a Trainer step
    is.  Only the order of launches changes: the step must equal the default one bit for bit, over a rollout (each
    embedder is back-propagated once per AR step and accumulates into the same .grad), eager and captured."""
    from neural_lam_amd import graph as G
    from neural_lam_amd import models as hm
    from neural_lam_amd.datastore import SyntheticDatastore
    from neural_lam_amd.trainer import Trainer

    hier = family != "graph_lam"
    ds = SyntheticDatastore(81 if hier else 40, 30 if hier else 36, 5, 2, 1, root_path=tmp_path, boundary="random", seed=1)
    ext = ds.get_xy_extent("state")
    raw = G.create_regular_grid_graph(ds.get_xy("state"), n_max_levels=3 if hier else None, hierarchical=hier)
    graph = G.normalise_graph(raw, max(ext[1] - ext[0], ext[3] - ext[2]))
    build = lambda: hm.MODELS[family](ds, graph=graph, hidden_dim=64, processor_layers=2)

    def make(early):
        torch.manual_seed(1)
        fc = hm.ARForecaster(build(), ds)
        return Trainer(hm.ForecasterStep(fc, ds).to(dev), lr=1e-3, use_graph=use_graph, early_leaf_backward=early)

    t_early, t_base = make(True), make(False)
    assert t_early.early_leaf_backward and not t_base.early_leaf_backward
    N, T = ds.num_grid_points, 3
    g = torch.Generator().manual_seed(0)
    for _ in range(3):
        batch = [torch.randn(2, 2, N, 5, generator=g).to(dev), torch.randn(2, T, N, 5, generator=g).to(dev),
                 torch.randn(2, T, N, 6, generator=g).to(dev)]
        le, lb = float(t_early.step(*batch)), float(t_base.step(*batch))
        assert le == lb
        assert torch.equal(t_early.fp.grad, t_base.fp.grad) and torch.equal(t_early.fp.flat, t_base.fp.flat)
    assert (t_early._graph is not None) == use_graph


def test_graph_step_falls_back_to_eager_for_another_batch_shape(dev, tmp_path):
    """A HIP graph is one shape: a batch of a different shape must not be copied (broadcast) into the captured
    buffers; it takes the eager step, and each returned loss is its own tensor."""
    from neural_lam_amd import graph as G
    from neural_lam_amd import models as hm
    from neural_lam_amd.datastore import SyntheticDatastore
    from neural_lam_amd.trainer import Trainer

    ds = SyntheticDatastore(30, 27, 5, 2, 1, root_path=tmp_path, boundary="random", seed=1)
    ext = ds.get_xy_extent("state")
    graph = G.normalise_graph(G.create_regular_grid_graph(ds.get_xy("state")), max(ext[1] - ext[0], ext[3] - ext[2]))

    def make(use_graph):
        torch.manual_seed(1)
        fc = hm.ARForecaster(hm.GraphLAM(ds, graph=graph, hidden_dim=16, processor_layers=1), ds)
        return Trainer(hm.ForecasterStep(fc, ds).to(dev), lr=1e-3, use_graph=use_graph)

    tg, te = make(True), make(False)
    N = ds.num_grid_points
    g = torch.Generator().manual_seed(0)
    losses = []
    for B in (2, 2, 1, 2):
        batch = [torch.randn(B, 2, N, 5, generator=g).to(dev), torch.randn(B, 2, N, 5, generator=g).to(dev),
                 torch.randn(B, 2, N, 6, generator=g).to(dev)]
        lg, le = tg.step(*batch), te.step(*batch)
        losses.append(lg)
        assert float(lg) == float(le)
    assert tg._graph is not None and tg.use_graph
    assert len({float(x) for x in losses}) == 4 and losses[0].data_ptr() != losses[1].data_ptr()
    assert torch.equal(tg.fp.flat, te.fp.flat)


@pytest.mark.parametrize("k,n,rows,batched", [(64, 64, 6561, False), (32, 64, 100, True), (64, 32, 33, False), (32, 32, 1, False),
                                              (256, 256, 6561, False), (128, 128, 70, True), (512, 512, 300, False)])
def test_node_linear_matches_torch(dev, k, n, rows, batched):
    """nlam_linear through NodeLinearFunction: x @ W1[:, col0:col0+k].T, its data gradient and the strided weight
    gradient (only the addressed column block of W1.grad is written)."""
    from neural_lam_amd.ops import NodeLinearFunction

    torch.manual_seed(3)
    kin, col0 = 3 * k, k
    W = torch.randn(n, kin, device=dev, requires_grad=True)
    shape = (2, rows, k) if batched else (rows, k)
    x = torch.randn(*shape, device=dev, requires_grad=True)
    out = NodeLinearFunction.apply(x, W, col0)
    cot = torch.randn_like(out)
    (out * cot).sum().backward()
    xr, Wr = x.detach().double().requires_grad_(), W.detach().double().requires_grad_()
    ref = xr @ Wr[:, col0 : col0 + k].T
    (ref * cot.double()).sum().backward()
    assert rel_err(out.double().cpu(), ref.cpu()) < 1e-5
    assert rel_err(x.grad.double().cpu(), xr.grad.cpu()) < 1e-5
    assert rel_err(W.grad.double().cpu(), Wr.grad.cpu()) < 1e-5
    assert float(W.grad[:, :col0].abs().max()) == 0.0 and float(W.grad[:, col0 + k :].abs().max()) == 0.0


@pytest.mark.parametrize("d,rows", [(128, 6561), (256, 45), (512, 6561), (256, 33001)])
def test_node_linear_pair_matches_torch(dev, d, rows):
    """Both node-level products of a mesh <-> mesh layer in one launch (nlam_linear with W2 / out2)."""
    from neural_lam_amd.ops import NodeLinearPairFunction

    torch.manual_seed(4)
    W = torch.randn(d, 3 * d, device=dev, requires_grad=True)
    x = torch.randn(2, rows, d, device=dev, requires_grad=True)
    pj, pi = NodeLinearPairFunction.apply(x, W, d, 2 * d)
    cj, ci = torch.randn_like(pj), torch.randn_like(pi)
    ((pj * cj).sum() + (pi * ci).sum()).backward()
    xr, Wr = x.detach().double().requires_grad_(), W.detach().double().requires_grad_()
    rj, ri = xr @ Wr[:, d : 2 * d].T, xr @ Wr[:, 2 * d :].T
    ((rj * cj.double()).sum() + (ri * ci.double()).sum()).backward()
    assert rel_err(pj.double().cpu(), rj.cpu()) < 1e-5 and rel_err(pi.double().cpu(), ri.cpu()) < 1e-5
    assert rel_err(x.grad.double().cpu(), xr.grad.cpu()) < 1e-5
    assert rel_err(W.grad.double().cpu(), Wr.grad.cpu()) < 1e-5
    assert float(W.grad[:, :d].abs().max()) == 0.0


@pytest.mark.parametrize("mode,tol", [("bf16x3", 1e-5), ("bf16x2", 2e-4), ("bf16", 2e-2)])
@pytest.mark.parametrize("k,n,rows", [(512, 512, 40003), (256, 128, 33000), (128, 512, 6561), (384, 256, 1), (512, 128, 65)])
def test_node_linear_gemm_tiles_and_modes(dev, k, n, rows, mode, tol):
    """nlam_linear on the LDS-tiled GEMM (round 5: linear_gemm_kernel, n % 128 == 0): 128-row tiles from 32 768 rows and
    64-row tiles below, ragged last row tile (rows past the end are zero-filled operands and unwritten outputs: the output
    buffer is poisoned first), K = 4 ... 16 chunks, forward layout and the transposed data-gradient layout, every split
    mode, accumulate on top of existing values, and agreement with the strip kernel of rounds 2-4 on the same problem."""
    from neural_lam_amd import _lib as L
    from neural_lam_amd import ops

    lib = L.load()
    try:
        _node_linear_gemm_body(dev, lib, L, ops, k, n, rows, mode, tol)
    finally:
        assert lib.nlam_set_tuning(L.TUNE_LIN_GEMM, 1) == 0


def _node_linear_gemm_body(dev, lib, L, ops, k, n, rows, mode, tol):
    assert lib.nlam_set_tuning(L.TUNE_LIN_GEMM, 2) == 0   # the GEMM wherever it applies (the default dispatch leaves two shapes to the strip kernel)
    torch.manual_seed(5)
    kin, col0 = 2 * k, k
    W = torch.randn(n, kin, device=dev) / k ** 0.5
    x = torch.randn(rows, k, device=dev)
    mm = ops._MM_FLAGS[mode]
    ref = x.double() @ W[:, col0:].double().T
    guard = torch.full((rows + 200, n), float("nan"), device=dev)
    out = guard[:rows]
    ops._linear_launch(x, W.data_ptr() + 4 * col0, kin, 1, k, n, out=out, mm_flags=mm)
    assert torch.isnan(guard[rows:]).all(), "rows past the end were written"
    assert rel_err(out.double().cpu(), ref.cpu()) < tol
    ops._linear_launch(x, W.data_ptr() + 4 * col0, kin, 1, k, n, out=out, accumulate=True, mm_flags=mm)
    assert rel_err(out.double().cpu(), 2 * ref.cpu()) < tol
    # the transposed product (data gradient): dx = g . W[:, col0:]
    g = torch.randn(rows, n, device=dev)
    n_t, k_t = k, n   # output width k, reduction over n -- the GEMM path needs n_t % 128 == 0
    dx = ops._linear_launch(g, W.data_ptr() + 4 * col0, 1, kin, k_t, n_t, mm_flags=mm)
    assert rel_err(dx.double().cpu(), (g.double() @ W[:, col0:].double()).cpu()) < tol
    if k % 64 == 0 and n % 64 == 0:   # the strip kernel takes these shapes too: same split products, other summation grouping
        try:
            assert lib.nlam_set_tuning(L.TUNE_LIN_GEMM, 0) == 0
            old = ops._linear_launch(x, W.data_ptr() + 4 * col0, kin, 1, k, n, mm_flags=mm)
        finally:
            assert lib.nlam_set_tuning(L.TUNE_LIN_GEMM, 2) == 0
        new = ops._linear_launch(x, W.data_ptr() + 4 * col0, kin, 1, k, n, mm_flags=mm)
        assert rel_err(new.cpu(), old.cpu()) < 1e-5


class _NoLibraryGemm:
    """Fails the test if a torch library GEMM / activation / LayerNorm runs: every depth of utils.make_mlp goes through the fused
    kernels (missing 2)."""

    NAMES = ("linear", "silu", "layer_norm")

    def __enter__(self):
        import torch.nn.functional as F

        self.F, self.saved = F, {n: getattr(F, n) for n in self.NAMES}

        def boom(*a, **k):
            raise AssertionError("a torch library op ran inside the product's MLP path")

        for n in self.NAMES:
            setattr(F, n, boom)
        self.mm = torch.Tensor.__matmul__
        torch.Tensor.__matmul__ = boom

    def __exit__(self, *exc):
        for n, f in self.saved.items():
            setattr(self.F, n, f)
        torch.Tensor.__matmul__ = self.mm


@pytest.mark.parametrize("hidden_layers", [0, 2, 3])
@pytest.mark.parametrize("width,dout,ln", [(64, 64, True), (24, 17, False), (160, 160, True)])
def test_other_mlp_depths_match_oracle(dev, hidden_layers, width, dout, ln):
    """utils.make_mlp with hidden_layers != 1 (utils/networks.py:8-40, CLI flag train_model.py:193-197): a chain of launches of the
    fused kernels -- [Linear -> SiLU] prefixes with an identity second Linear, Linear [-> LayerNorm] with NLAM_F_NO_ACT --
    narrow, ragged (the output_map shape: 17 columns, no LayerNorm) and wide widths."""
    from oracle import gnn_layers as og

    hl = _hl()
    torch.manual_seed(hidden_layers)
    bp = [24] + [width] * hidden_layers + [dout]
    ref, net = og.make_mlp(bp, layer_norm=ln), hl.make_mlp(bp, layer_norm=ln)
    assert list(ref.state_dict().keys()) == list(net.state_dict().keys())
    net.load_state_dict(ref.state_dict())
    net.to(dev)
    x = torch.randn(2, 700, 24)
    x1, x2 = x.clone().requires_grad_(), x.to(dev).requires_grad_()
    y1 = ref(x1)
    y1.sin().sum().backward()
    with _NoLibraryGemm():
        y2 = net(x2)
        y2.sin().sum().backward()
    assert rel_err(y2.cpu(), y1) < TOL
    assert rel_err(x2.grad.cpu(), x1.grad) < TOL
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert rel_err(p.grad.cpu(), q.grad) < TOL, k


@pytest.mark.parametrize("cls_name", ["InteractionNet", "PropagationNet"])
@pytest.mark.parametrize("hidden_layers,d", [(0, 32), (2, 32), (3, 64), (0, 128), (2, 128)])
def test_gnn_layers_of_other_depths_keep_the_fused_gather(dev, cls_name, hidden_layers, d):
    """InteractionNet / PropagationNet with ``hidden_layers`` 0 / 2 / 3 (gnn_layers.py:23-108 builds edge_mlp / aggr_mlp with
    utils.make_mlp of that depth): gather, concat, residuals and aggregation stay inside the fused launches -- no index_select /
    cat / library GEMM -- incl. a receiver that is cut over several tiles, with a batch, forward and backward."""
    from oracle import gnn_layers as og

    hl = _hl()
    ei = _rand_ei(23, 19, 211, seed=5)
    ei[1, :70] = 4                       # in-degree > 32: the deterministic two-pass reduction
    torch.manual_seed(5)
    ref = getattr(og, cls_name)(ei, d, hidden_layers=hidden_layers)
    net = getattr(hl, cls_name)(ei, d, hidden_layers=hidden_layers)
    net.load_state_dict(ref.state_dict())
    net.to(dev)
    send, rec, edge = torch.randn(2, 23, d), torch.randn(2, 19, d), torch.randn(2, 211, d)
    s1, r1, e1 = (t.clone().requires_grad_() for t in (send, rec, edge))
    s2, r2, e2 = (t.to(dev).requires_grad_() for t in (send, rec, edge))
    o1 = ref(s1, r1, e1)
    sum(o.square().sum() for o in o1).backward()
    real_select, real_cat = torch.Tensor.index_select, torch.cat
    try:
        def no_gather(*a, **k):
            raise AssertionError("explicit gather in a GNN layer")

        torch.Tensor.index_select = no_gather
        with _NoLibraryGemm():
            o2 = net(s2, r2, e2)
            sum(o.square().sum() for o in o2).backward()
    finally:
        torch.Tensor.index_select = real_select
    for a, b in zip(o2, o1):
        assert rel_err(a.cpu(), b) < TOL
    for a, b in ((s2, s1), (r2, r1), (e2, e1)):
        assert rel_err(a.grad.cpu(), b.grad) < TOL
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert rel_err(p.grad.cpu(), q.grad) < TOL, k


def test_interaction_net_with_two_hidden_layers(dev):
    from oracle import gnn_layers as og

    hl = _hl()
    ei = _rand_ei(23, 19, 211, seed=5)
    torch.manual_seed(5)
    ref = og.InteractionNet(ei, 32, hidden_layers=2)
    net = hl.InteractionNet(ei, 32, hidden_layers=2)
    net.load_state_dict(ref.state_dict())
    net.to(dev)
    send, rec, edge = torch.randn(2, 23, 32), torch.randn(2, 19, 32), torch.randn(2, 211, 32)
    s1, r1, e1 = (t.clone().requires_grad_() for t in (send, rec, edge))
    s2, r2, e2 = (t.to(dev).requires_grad_() for t in (send, rec, edge))
    o1, o2 = ref(s1, r1, e1), net(s2, r2, e2)
    for a, b in zip(o2, o1):
        assert rel_err(a.cpu(), b) < TOL
    sum(o.square().sum() for o in o1).backward()
    sum(o.square().sum() for o in o2).backward()
    for a, b in ((s2, s1), (r2, r1), (e2, e1)):
        assert rel_err(a.grad.cpu(), b.grad) < TOL
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert rel_err(p.grad.cpu(), q.grad) < TOL, k


@pytest.mark.gpu
@pytest.mark.parametrize("width,nseg,avg,batch", [(64, 700, 39, 2), (32, 300, 20, 1), (128, 200, 17, 2), (64, 50, 3, 1), (20, 90, 25, 1),
                                                  (256, 40, 30, 1), (64, 6561, 39, 1)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_segment_sum_long_and_short_segments_match_index_add(width, nseg, avg, batch, accumulate):
    """nlam_segment_sum / _acc directly (the CSC scatter-by-sender and CSR aggregate of the layers): long segments take the
    kernel that gives one segment to several lane groups (widths 32 / 64: four groups, 128: two), short ones, odd widths and
    width 256 the one-thread-per-column kernel; ragged segment lengths incl. empty segments, a gather order, per-segment
    scales, a batch stride, accumulation onto an existing output; deterministic."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from neural_lam_amd import ops

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(width * 1000 + nseg)
    lens = torch.randint(0, 2 * avg + 1, (nseg,), generator=g)
    lens[::7] = 0                                   # some empty segments
    rows = int(lens.sum())
    ptr = torch.zeros(nseg + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(lens, 0).to(torch.int32)
    order = torch.randperm(rows, generator=g).to(torch.int32)
    x = torch.randn(batch, rows, width, generator=g)
    scale = torch.rand(nseg, generator=g) + 0.5
    base = torch.randn(batch, nseg, width, generator=g)
    seg_of_pos = torch.repeat_interleave(torch.arange(nseg), lens)
    ref = torch.zeros(batch, nseg, width, dtype=torch.float64)
    ref.index_add_(1, seg_of_pos, x[:, order.long()].double())
    ref = ref * scale.double()[None, :, None] + (base.double() if accumulate else 0.0)
    out = base.clone().to(dev) if accumulate else None
    got = ops.segment_sum(x.to(dev), rows * width, ptr.to(dev), order.to(dev), scale.to(dev), nseg, width, batch, out=out,
                          accumulate=accumulate)
    again = ops.segment_sum(x.to(dev), rows * width, ptr.to(dev), order.to(dev), scale.to(dev), nseg, width, batch,
                            out=base.clone().to(dev) if accumulate else None, accumulate=accumulate)
    assert torch.equal(got, again)
    assert rel_err(got.cpu().double(), ref) < 1e-5
    assert float(got[:, ::7].cpu().sub(base[:, ::7] if accumulate else 0.0).abs().max()) == 0.0   # empty segments: zero contribution


def test_split_receivers_are_reduced_deterministically(dev):
    """Receivers with in-degree > 32 (none on the MEPS graphs; a finer grid / coarser mesh g2m has them) are cut over several
    tiles.  The reference trains such graphs with ``Trainer(deterministic=True)`` (train_model.py:566; PyG's scatter,
    gnn_layers.py:175-189, has a deterministic path): here the pieces reduce into virtual segments with plain stores and
    ``nlam_split_combine`` adds them up in CSR order, so the layer runs under ``torch.use_deterministic_algorithms(True)``,
    is bit-reproducible forward and backward, and matches the oracle -- for sum and mean aggregation, with a batch.
    The C-ABI's one-pass NLAM_TILE_SPLIT tiles (atomic adds; ``graph.VIRTUAL_SPLIT = False``) still refuse that mode."""
    from neural_lam_amd import graph as G
    from oracle import gnn_layers as og

    hl = _hl()
    E = 300
    ei = torch.stack([torch.arange(E) % 50, torch.zeros(E, dtype=torch.int64)])   # receiver 0: in-degree 200
    ei[1, 200:260] = 2                                                              # receiver 2: in-degree 60; 1 isolated
    ei[1, 260:] = torch.arange(40) % 5 + 3
    from neural_lam_amd import _lib as L

    for cls_name, d, wbf in (("InteractionNet", 64, False), ("PropagationNet", 16, False), ("InteractionNet", 128, False),
                             ("InteractionNet", 128, True)):
        # d = 128: the fp32 one-tile-per-workgroup wide kernels, then the split-bf16 super-tile kernels forced at this size
        assert L.load().nlam_set_tuning(L.TUNE_WBF_MIN_SUPERTILES, 0 if wbf else 192) == 0
        torch.manual_seed(1)
        ref = getattr(og, cls_name)(ei, d)
        net = getattr(hl, cls_name)(ei, d)
        net.load_state_dict(ref.state_dict())
        net.to(dev)
        assert net._host_csr[0].comb_ptr is not None and not net._host_csr[2]
        send, rec, edge = torch.randn(2, 50, d), torch.randn(2, 8, d), torch.randn(2, E, d)
        r_out = ref(*(t.clone().requires_grad_() for t in (send, rec, edge)))
        cots = [torch.randn_like(o) for o in r_out]   # (a plain sum of LayerNorm outputs has a zero gradient)
        runs = []
        try:
            torch.use_deterministic_algorithms(True)
            for _ in range(2):
                args = [t.to(dev).requires_grad_() for t in (send, rec, edge)]
                out = net(*args)
                sum((o * c.to(dev)).sum() for o, c in zip(out, cots)).backward()
                runs.append([o.detach().clone() for o in out] + [a.grad.clone() for a in args]
                            + [p.grad.clone() for p in net.parameters()])
                net.zero_grad()
        finally:
            torch.use_deterministic_algorithms(False)
        for a, b in zip(*runs):
            assert torch.equal(a, b), cls_name
        for o, r in zip(runs[0][:2], r_out):
            assert rel_err(o.cpu(), r.detach()) < TOL, cls_name
        sg, rg, eg = (t.clone().requires_grad_() for t in (send, rec, edge))
        o2 = ref(sg, rg, eg)
        sum((o * c).sum() for o, c in zip(o2, cots)).backward()
        for h, r in zip(runs[0][2:5], (sg.grad, rg.grad, eg.grad)):
            assert rel_err(h.cpu(), r) < TOL, cls_name
        for h, (k, p) in zip(runs[0][5:], ref.named_parameters()):
            assert rel_err(h.cpu(), p.grad) < TOL, (cls_name, k)
    assert L.load().nlam_set_tuning(L.TUNE_WBF_MIN_SUPERTILES, 192) == 0

    # the one-pass schedule of the C-ABI (atomic partial sums): same results within tolerance, refused in deterministic mode
    try:
        G.VIRTUAL_SPLIT = False
        torch.manual_seed(1)
        ref = og.InteractionNet(ei, 64)
        bad = hl.InteractionNet(ei, 64)
        bad.load_state_dict(ref.state_dict())
        bad.to(dev)
        assert bad._host_csr[2] and bad._host_csr[0].comb_ptr is None
        send, rec, edge = torch.randn(50, 64), torch.randn(8, 64), torch.randn(E, 64)
        o = bad(send.to(dev), rec.to(dev), edge.to(dev))
        r = ref(send, rec, edge)
        assert rel_err(o[0].cpu(), r[0].detach()) < TOL and rel_err(o[1].cpu(), r[1].detach()) < TOL
        try:
            torch.use_deterministic_algorithms(True)
            with pytest.raises(RuntimeError, match="in-degree > 32"):
                bad(send.to(dev), rec.to(dev), edge.to(dev))
            torch.use_deterministic_algorithms(True, warn_only=True)
            with pytest.warns(UserWarning, match="in-degree > 32"):
                bad(send.to(dev), rec.to(dev), edge.to(dev))
        finally:
            torch.use_deterministic_algorithms(False)
    finally:
        G.VIRTUAL_SPLIT = True


@pytest.mark.parametrize("K", [1, 3, 1000])
@pytest.mark.parametrize("model,T", [("graph_lam", 3), ("hi_lam_parallel", 1)])
def test_segmented_executor_equals_one_graph_and_eager(dev, tmp_path, model, T, K):
    """trainer._SegmentedStep (a chain of linear graphs on one stream + weight-gradient graphs on side streams, tied by
    events) against the one-graph executor (forks inside the capture) and the eager step: same bits over several optimizer
    steps, for every segment length -- one fork per segment, the default, and a single segment -- on a rollout (every MLP
    back-propagated once per AR step: the read-modify-write accumulations of a parameter must stay ordered across
    segments) and on the chunked HiLAMParallel stack."""
    from neural_lam_amd import graph as G
    from neural_lam_amd import models as hm
    from neural_lam_amd.datastore import SyntheticDatastore
    from neural_lam_amd.trainer import Trainer, _SegmentedStep

    hier = model != "graph_lam"

    def make(**kw):
        ds = SyntheticDatastore(81 if hier else 60, 30 if hier else 54, 5, 2, 1, root_path=tmp_path, boundary="random", seed=1)
        ext = ds.get_xy_extent("state")
        raw = G.create_regular_grid_graph(ds.get_xy("state"), n_max_levels=3 if hier else None, hierarchical=hier)
        graph = G.normalise_graph(raw, max(ext[1] - ext[0], ext[3] - ext[2]))
        torch.manual_seed(1)
        fc = hm.ARForecaster(hm.MODELS[model](ds, graph=graph, hidden_dim=32 if hier else 64, processor_layers=2), ds)
        return ds, Trainer(hm.ForecasterStep(fc, ds).to(dev), lr=1e-3, **kw)

    ds, t_seg = make(use_graph=True, executor="segments", forks_per_segment=K)
    _, t_one = make(use_graph=True, executor="forks")
    _, t_eager = make(use_graph=False)
    N = ds.num_grid_points
    g = torch.Generator().manual_seed(0)
    for _ in range(3):
        batch = [torch.randn(1, 2, N, 5, generator=g).to(dev), torch.randn(1, T, N, 5, generator=g).to(dev),
                 torch.randn(1, T, N, 6, generator=g).to(dev)]
        ls, lo, le = float(t_seg.step(*batch)), float(t_one.step(*batch)), float(t_eager.step(*batch))
        assert ls == lo == le
        assert torch.equal(t_seg.fp.grad, t_eager.fp.grad) and torch.equal(t_seg.fp.flat, t_eager.fp.flat)
        assert torch.equal(t_one.fp.flat, t_eager.fp.flat)
    seg = t_seg._graph
    assert isinstance(seg, _SegmentedStep) and not isinstance(t_one._graph, _SegmentedStep)
    assert seg.nforks > 0 and len(seg.chain) == len(seg.side) and seg.tail is not None
    if K == 1000:   # one segment per dead-end MLP at most: nothing else cuts the chain
        assert len(seg.chain) <= 4
    if K == 1:
        assert len(seg.chain) >= min(seg.nforks, 4)


@pytest.mark.parametrize("executor", ["forks", "segments"])
def test_optimizer_schedule_does_not_rerecord_the_step(dev, tmp_path, executor):
    """A learning-rate schedule under the captured step (advisor finding, round 4): the first change re-records the
    optimizer's own graph only, the second moves the optimizer behind the replay for good -- the chain is recorded once --
    and the weights follow torch.optim.AdamW with the same schedule on the eager trainer."""
    from neural_lam_amd import graph as G
    from neural_lam_amd import models as hm
    from neural_lam_amd.datastore import SyntheticDatastore
    from neural_lam_amd.trainer import Trainer

    def make(use_graph):
        ds = SyntheticDatastore(30, 27, 5, 2, 1, root_path=tmp_path, boundary="random", seed=1)
        ext = ds.get_xy_extent("state")
        graph = G.normalise_graph(G.create_regular_grid_graph(ds.get_xy("state")), max(ext[1] - ext[0], ext[3] - ext[2]))
        torch.manual_seed(1)
        fc = hm.ARForecaster(hm.GraphLAM(ds, graph=graph, hidden_dim=16, processor_layers=2), ds)
        return ds, Trainer(hm.ForecasterStep(fc, ds).to(dev), lr=1e-3, use_graph=use_graph, executor=executor)

    ds, tg = make(True)
    _, te = make(False)
    N = ds.num_grid_points
    g = torch.Generator().manual_seed(0)
    recorded = []
    for it, lr in enumerate([1e-3, 1e-3, 5e-4, 2.5e-4, 1.25e-4, 1e-4]):
        tg.opt.lr = te.opt.lr = lr
        batch = [torch.randn(1, 2, N, 5, generator=g).to(dev), torch.randn(1, 2, N, 5, generator=g).to(dev),
                 torch.randn(1, 2, N, 6, generator=g).to(dev)]
        assert float(tg.step(*batch)) == float(te.step(*batch))
        assert torch.equal(tg.fp.flat, te.fp.flat), (it, lr)
        recorded.append(tg._graph.chain[0] if executor == "segments" else tg._graph)
    assert tg._opt_eager and not tg._opt_in_graph
    if executor == "segments":
        assert all(r is recorded[0] for r in recorded) and tg._graph.tail is None   # the chain was recorded once
    else:   # one graph: the first change re-records it with the new values, the second once more without the optimizer -- and never again
        assert recorded[1] is recorded[0] and recorded[4] is recorded[3] and recorded[5] is recorded[3]


@pytest.mark.parametrize("model,kw,autocast", [
    ("graph_lam", dict(hidden_dim=16, processor_layers=2), False),
    ("graph_lam", dict(hidden_dim=128, processor_layers=1), False),
    ("graph_lam", dict(hidden_dim=128, processor_layers=1), True),
    ("hi_lam", dict(hidden_dim=16, processor_layers=2), False),
    # d = 512: four 256 x 256 windows per weight gradient -- the row-slice count of a launch sized to co-run (side streams in use)
    # differs from the eager module's; the captured backward must record the eager shape (ops.wgrad_shape) to stay bit-identical
    ("graph_lam", dict(hidden_dim=512, processor_layers=1), False),
    ("graph_lam", dict(hidden_dim=512, processor_layers=1), True),
])
@pytest.mark.parametrize("overlap", [True, False])
def test_graphed_training_step_equals_eager(dev, tmp_path, model, kw, autocast, overlap):
    """trainer.graphed_training_step -- the drop-in path: forward and backward each replay one HIP graph, the caller keeps
    ``loss.backward()`` and ``torch.optim.AdamW`` (models/module.py:293-304, 394-417) -- against the same module launched
    eagerly: loss, prediction, every ``.grad`` autograd delivers and the weights after three optimizer steps, bit for bit;
    a batch of another shape falls through to the eager module."""
    import contextlib

    from neural_lam_amd import graph as G
    from neural_lam_amd import models as hm
    from neural_lam_amd.datastore import SyntheticDatastore
    from neural_lam_amd.trainer import graphed_training_step

    hier = model != "graph_lam"

    def make():
        ds = SyntheticDatastore(81 if hier else 30, 30 if hier else 27, 5, 2, 1, root_path=tmp_path, boundary="random", seed=1)
        ext = ds.get_xy_extent("state")
        raw = G.create_regular_grid_graph(ds.get_xy("state"), n_max_levels=3 if hier else None, hierarchical=hier)
        graph = G.normalise_graph(raw, max(ext[1] - ext[0], ext[3] - ext[2]))
        torch.manual_seed(1)
        fc = hm.ARForecaster(hm.MODELS[model](ds, graph=graph, **kw), ds)
        step = hm.ForecasterStep(fc, ds).to(dev)
        return ds, step, torch.optim.AdamW(step.parameters(), lr=1e-3, betas=(0.9, 0.95))

    amp = (lambda: torch.autocast("cuda", dtype=torch.bfloat16)) if autocast else contextlib.nullcontext
    ds, s_e, o_e = make()
    _, s_g, o_g = make()
    N = ds.num_grid_points
    g = torch.Generator().manual_seed(0)

    def batch(T=2):
        return [torch.randn(1, 2, N, 5, generator=g).to(dev), torch.randn(1, T, N, 5, generator=g).to(dev), torch.randn(1, T, N, 6, generator=g).to(dev)]

    with amp():
        graphed = graphed_training_step(s_g, *batch(), overlap_wgrad=overlap)
    for _ in range(3):
        b = batch()
        res = []
        for fn, opt, mod in ((s_e, o_e, s_e), (graphed, o_g, s_g)):
            opt.zero_grad(set_to_none=True)
            with amp():
                pred, loss = fn(*b)
            loss.backward()
            res.append((pred.detach().clone(), float(loss), [p.grad.clone() for p in mod.parameters()]))
            opt.step()
        assert res[0][1] == res[1][1]
        assert torch.equal(res[0][0], res[1][0])
        for a, c in zip(res[0][2], res[1][2]):
            assert torch.equal(a, c)
        for a, c in zip(s_e.parameters(), s_g.parameters()):
            assert torch.equal(a, c)
    b = batch(T=1)   # another rollout length: not the captured shape
    with amp():
        _, l_e = s_e(*b)
        _, l_g = graphed(*b)
    assert float(l_e) == float(l_g)


@pytest.mark.parametrize("model,kw,autocast,fused", [
    ("graph_lam", dict(hidden_dim=64, processor_layers=2), False, False),
    ("graph_lam", dict(hidden_dim=64, processor_layers=2), False, True),
    ("graph_lam", dict(hidden_dim=128, processor_layers=1), True, False),
    ("hi_lam", dict(hidden_dim=16, processor_layers=2), False, False),
])
def test_graphed_flat_step_equals_eager(dev, tmp_path, model, kw, autocast, fused):
    """graphed_training_step(flat=True): the module's parameters as views of one flat leaf, ONE gradient handed to autograd, the
    caller's torch.optim.AdamW built over that leaf (models/module.py:293-304: one parameter group, no per-tensor exceptions, so
    the element-wise update is the same) -- against the eager module under AdamW(module.parameters()): loss, prediction, the
    gradient slice of every parameter and the weights after three optimizer steps, bit for bit; names / shapes / state_dict
    keys unchanged; a batch of another shape is refused under grad mode and falls through to the eager module without it."""
    import contextlib

    from neural_lam_amd import graph as G
    from neural_lam_amd import models as hm
    from neural_lam_amd.datastore import SyntheticDatastore
    from neural_lam_amd.trainer import FlatStepModule, graphed_training_step

    hier = model != "graph_lam"

    def make():
        ds = SyntheticDatastore(81 if hier else 30, 30 if hier else 27, 5, 2, 1, root_path=tmp_path, boundary="random", seed=1)
        ext = ds.get_xy_extent("state")
        raw = G.create_regular_grid_graph(ds.get_xy("state"), n_max_levels=3 if hier else None, hierarchical=hier)
        graph = G.normalise_graph(raw, max(ext[1] - ext[0], ext[3] - ext[2]))
        torch.manual_seed(1)
        fc = hm.ARForecaster(hm.MODELS[model](ds, graph=graph, **kw), ds)
        return ds, hm.ForecasterStep(fc, ds).to(dev)

    amp = (lambda: torch.autocast("cuda", dtype=torch.bfloat16)) if autocast else contextlib.nullcontext
    ds, s_e = make()
    _, s_g = make()
    keys = list(s_g.state_dict().keys())
    N = ds.num_grid_points
    g = torch.Generator().manual_seed(0)

    def batch(T=2):
        return [torch.randn(1, 2, N, 5, generator=g).to(dev), torch.randn(1, T, N, 5, generator=g).to(dev), torch.randn(1, T, N, 6, generator=g).to(dev)]

    with amp():
        graphed = graphed_training_step(s_g, *batch(), flat=True)
    leaf = graphed.flat_parameter
    assert leaf.is_leaf and leaf.requires_grad and list(s_g.state_dict().keys()) == keys
    assert [p.shape for p in s_g.parameters()] == [p.shape for p in s_e.parameters()]
    assert list(FlatStepModule(graphed, pick=1).parameters())[0] is leaf and len(list(FlatStepModule(graphed).parameters())) == 1
    o_e = torch.optim.AdamW(s_e.parameters(), lr=1e-3, betas=(0.9, 0.95), fused=fused or None)
    o_g = torch.optim.AdamW([leaf], lr=1e-3, betas=(0.9, 0.95), fused=fused or None)
    for _ in range(3):
        b = batch()
        o_e.zero_grad(set_to_none=True)
        with amp():
            pred_e, loss_e = s_e(*b)
        loss_e.backward()
        o_g.zero_grad(set_to_none=True)
        with amp():
            pred_g, loss_g = graphed(*b)
        loss_g.backward()
        assert float(loss_e) == float(loss_g) and torch.equal(pred_e, pred_g)
        assert all(p.grad is None for p in s_g.parameters())   # ONE AccumulateGrad: the leaf's
        for p, o in zip(s_e.parameters(), graphed.goffs):
            assert torch.equal(p.grad.reshape(-1), leaf.grad[o : o + p.numel()])
        o_e.step()
        o_g.step()
        for a, c in zip(s_e.parameters(), s_g.parameters()):
            assert torch.equal(a, c)
    b = batch(T=1)   # another rollout length: not the captured shape
    with pytest.raises(ValueError, match="captured"):
        with amp():
            graphed(*b)
    with torch.no_grad(), amp():
        _, l_e = s_e(*b)
        _, l_g = graphed(*b)
    assert float(l_e) == float(l_g)


@pytest.mark.parametrize("d,same,factorised", [(64, True, True), (64, False, True), (128, True, True), (256, False, True),
                                               (64, True, False), (32, True, False)])
def test_gradient_mailbox_equals_autograd_accumulation(dev, d, same, factorised):
    """A factorised layer hands the data gradient of its receiver table from the node MLP's backward to the node-level
    product's backward (ops.mail_scope: accumulated in place by nlam_linear, one autograd `add` launch less per layer and AR
    step) instead of returning two tensors for autograd to add.  Two stacked layers (the second one's table is a non-leaf with a
    residual consumer outside the layer): same gradients with the hand-over on and off, and against the oracle; the hand-over
    really happened (one post + one consume per layer and backward).  factorised=False: a mesh <-> mesh layer on the plain gather
    kernels (narrow widths, the cfg2 processor) -- there the EDGE launch is the consumer: its receiver-side and sender-side gradients
    of the node table meet the node MLP's in one nlam_segment_sum_add pass."""
    from neural_lam_amd import _lib as L
    from neural_lam_amd import ops
    from oracle import gnn_layers as og

    hl = _hl()
    ns, nr, e, B = (61, 61, 1500, 2) if same else (70, 45, 1400, 1)
    ei = _rand_ei(ns, nr, e, seed=11)
    torch.manual_seed(11)
    refs = [og.InteractionNet(ei, d), og.InteractionNet(ei, d)]
    nets = [hl.InteractionNet(ei, d), hl.InteractionNet(ei, d)]
    for r, n in zip(refs, nets):
        n.load_state_dict(r.state_dict())
        n.to(dev)
    send, rec, edge = torch.randn(B, ns, d), torch.randn(B, nr, d), torch.randn(B, e, d)

    def run(layers, s, r, ed):
        if same:
            x, ed1 = layers[0](r, r, ed)
            x = x + 0.5 * r                      # a consumer of the table outside the layer
            x2, ed2 = layers[1](x, x, ed1)
            return (x2 * x2).sum() + (ed2 * ed2).sum() + (x * x).sum()
        r1, ed1 = layers[0](s, r, ed)
        r2, ed2 = layers[1](s, r1 + 0.5 * r, ed1)
        return (r2 * r2).sum() + (ed2 * ed2).sum() + (r1 * r1).sum()

    s0, r0, e0 = (t.clone().requires_grad_() for t in (send, rec, edge))
    run(refs, s0, r0, e0).backward()
    old = (hl.FACTORISE_MIN_EDGES, hl.FACTORISE_MIN_WORK_WIDE, hl.FACTORISE_MIN_WIDTH_WIDE, hl.FACTORISE_MIN_EDGES_WIDE, ops.GRAD_MAILBOX_ON)
    hl.FACTORISE_MIN_EDGES = hl.FACTORISE_MIN_WORK_WIDE = hl.FACTORISE_MIN_WIDTH_WIDE = hl.FACTORISE_MIN_EDGES_WIDE = 0 if factorised else 1 << 30
    lib = L.load()
    res = {}
    try:
        assert lib.nlam_set_tuning(L.TUNE_WBF_MIN_SUPERTILES, 0) == 0
        for on in (True, False):
            ops.GRAD_MAILBOX_ON = on
            ops.MAIL_STATS["posted"] = ops.MAIL_STATS["consumed"] = 0
            for n in nets:
                n.zero_grad()
            s1, r1, e1 = (t.to(dev).requires_grad_() for t in (send, rec, edge))
            run(nets, s1, r1, e1).backward()
            res[on] = ([t.grad.clone() for t in ((r1, e1) if same else (s1, r1, e1))], [p.grad.clone() for n in nets for p in n.parameters()],
                       dict(ops.MAIL_STATS))
    finally:
        hl.FACTORISE_MIN_EDGES, hl.FACTORISE_MIN_WORK_WIDE, hl.FACTORISE_MIN_WIDTH_WIDE, hl.FACTORISE_MIN_EDGES_WIDE, ops.GRAD_MAILBOX_ON = old
        assert lib.nlam_set_tuning(L.TUNE_WBF_MIN_SUPERTILES, 192) == 0
    assert res[True][2] == {"posted": 2, "consumed": 2} and res[False][2] == {"posted": 0, "consumed": 0}, (res[True][2], res[False][2])   # one per layer
    for a, b in zip(res[True][0] + res[True][1], res[False][0] + res[False][1]):
        assert rel_err(a, b) < 1e-6
    for a, b in zip(res[True][0], ((r0, e0) if same else (s0, r0, e0))):
        assert rel_err(a.cpu(), b.grad) < TOL
    for a, (k, q) in zip(res[True][1], [kv for r in refs for kv in r.named_parameters()]):
        assert rel_err(a.cpu(), q.grad) < TOL, k


def test_stream_layout_gives_the_chain_a_hardware_queue_of_its_own(dev):
    """ops.stream_layout() (DESIGN finding 54): the streams of a process are bound to a few in-order hardware queues; the layout is found
    by observation -- a tiny launch behind a spinning kernel finishes late exactly when the two streams share a queue -- and must hand out
    a chain stream that shares its queue with no side stream and not with the caller's stream, side streams off the caller's queue,
    and the same objects on every call."""
    from neural_lam_amd import ops

    L1 = ops.stream_layout()
    assert ops.stream_layout() is L1
    if not L1["groups"]:
        pytest.skip("queue probe switched off (NLAM_QUEUE_PROBE=0)")
    tick = torch.zeros(1, device=dev)
    main = torch.cuda.default_stream()
    ngroups = len([g for g in L1["groups"] if g])
    assert sum(len(g) for g in L1["groups"]) == 13
    if ngroups >= 2:
        assert not ops._streams_share_queue(main, L1["chain"], tick)
    if ngroups >= 3:
        for s in L1["sides"]:
            assert not ops._streams_share_queue(L1["chain"], s, tick)
            assert not ops._streams_share_queue(main, s, tick)
    # the grouping is an equivalence: members of one group share, representatives of different groups do not
    reps = [g[0] for g in L1["groups"] if g]
    for i, a in enumerate(reps):
        for b in reps[i + 1 :]:
            assert not ops._streams_share_queue(a, b, tick)
    for g in L1["groups"]:
        if len(g) > 1:
            assert ops._streams_share_queue(g[0], g[-1], tick)
