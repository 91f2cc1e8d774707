"""Diagnostic run on the GPU box: every golden case through the HIP path, with the
error of every output / gradient printed (no early exit).  Not a test; the parity
tests proper live in tests/test_hip_parity.py."""
import sys
import time
import traceback
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from conftest import graph_from_case, load_golden, rel_err  # noqa: E402

from neural_lam_amd import gnn_layers as hl  # noqa: E402
from neural_lam_amd import models as hm  # noqa: E402
from neural_lam_amd.datastore import SyntheticDatastore  # noqa: E402

dev = torch.device("cuda:0")
worst = 0.0


def report(name, a, b):
    global worst
    e = rel_err(a.detach().cpu(), b)
    worst = max(worst, e)
    flag = "" if e < 1e-4 else "   <-- FAIL"
    print(f"    {name:40s} rel_err {e:.3e}{flag}")


def layer_cases():
    cases = dict(load_golden("layers")["cases"])
    cases.update(load_golden("layers_wide")["cases"])
    for name, case in cases.items():
        print(f"[layer] {name}")
        try:
            ei = case["edge_index"].to(torch.int64)
            net = hl.get_gnn_class(case["cls"])(ei, case["d"], **case["kwargs"])
            net.load_state_dict(case["state_dict"], strict=True)
            net = net.to(dev)
            send = case["send"].to(dev).requires_grad_()
            rec = case["rec"].to(dev).requires_grad_()
            edge = case["edge"].to(dev).requires_grad_()
            out = net(send, rec, edge)
            outs = out if isinstance(out, tuple) else (out,)
            for k, (o, r) in enumerate(zip(outs, case["ref_out"])):
                report(f"out[{k}]", o, r)
            sum((o * c.to(dev)).sum() for o, c in zip(outs, case["cotangents"])).backward()
            report("grad_send", send.grad, case["ref_grad_send"])
            report("grad_rec", rec.grad, case["ref_grad_rec"])
            report("grad_edge", edge.grad, case["ref_grad_edge"])
            for k, p in net.named_parameters():
                report(f"grad {k}", p.grad, case["ref_grad_params"][k])
        except Exception:
            traceback.print_exc()


def model_cases():
    cls = {"GraphLAM": hm.GraphLAM, "HiLAM": hm.HiLAM, "HiLAMParallel": hm.HiLAMParallel}
    for name in ["graphlam_30x27", "graphlam_30x27_variants", "graphlam_30x27_d128", "hilam_81x30", "hilam_parallel_81x30"]:
        print(f"[model] {name}")
        try:
            case = load_golden(name)
            ds = SyntheticDatastore(root_path="/tmp/nlam_gpu_check", **case["ds_kwargs"])
            graph = (case["ref_hierarchical"], graph_from_case(case))
            predictor = cls[case["model"]](ds, graph=graph, **case["model_kwargs"])
            forecaster = hm.ARForecaster(predictor, ds)
            forecaster.load_state_dict(case["state_dict"], strict=True)
            step = hm.ForecasterStep(forecaster, ds).to(dev)
            init, target, forcing = case["init"].to(dev), case["target"].to(dev), case["forcing"].to(dev)
            with torch.no_grad():
                one, one_std = predictor(init[:, 1], init[:, 0], forcing[:, 0])
            report("one_step", one, case["ref_one_step"])
            if case["ref_one_std"] is not None:
                report("one_std", one_std, case["ref_one_std"])
            pred, loss = step(init, target, forcing)
            report("prediction", pred, case["ref_prediction"])
            report("loss", loss.reshape(1), case["ref_loss"].reshape(1))
            loss.backward()
            for k, p in forecaster.named_parameters():
                if p.grad is None:
                    print(f"    grad {k}: MISSING   <-- FAIL")
                    continue
                g, r = p.grad.detach().cpu(), case["ref_grads"][k]
                e = float((g - r).abs().max()) / max(float(r.abs().max()), 1e-3)
                global worst
                worst = max(worst, e)
                if e > 1e-4:
                    print(f"    grad {k:60s} err {e:.3e}   <-- FAIL")
            print("    (parameter gradients checked)")
        except Exception:
            traceback.print_exc()


def oracle_cases():
    """HIP vs CPU oracle on seeded inputs: the wide (d > 64) kernels and odd MLP shapes."""
    from oracle import gnn_layers as og

    def rand_ei(ns, nr, e, seed):
        g = torch.Generator().manual_seed(seed)
        ei = torch.stack([torch.randint(0, ns, (e,), generator=g), torch.randint(0, nr, (e,), generator=g)])
        ei[1, -1] = nr - 1
        return ei

    for cls_name, d, upd in [("InteractionNet", 128, True), ("PropagationNet", 128, False), ("InteractionNet", 256, True),
                             ("PropagationNet", 256, True), ("InteractionNet", 512, True), ("InteractionNet", 96, False),
                             ("InteractionNet", 200, True)]:
        print(f"[oracle layer] {cls_name} d={d} update_edges={upd}")
        try:
            ns, nr, e, B = 61, 47, 501, 2
            ei = rand_ei(ns, nr, e, d)
            torch.manual_seed(d)
            ref = getattr(og, cls_name)(ei, d, update_edges=upd)
            net = getattr(hl, cls_name)(ei, d, update_edges=upd)
            net.load_state_dict(ref.state_dict())
            net.to(dev)
            send, rec, edge = torch.randn(B, ns, d), torch.randn(B, nr, d), torch.randn(B, e, d)
            s1, r1, e1 = (t.clone().requires_grad_() for t in (send, rec, edge))
            s2, r2, e2 = (t.to(dev).requires_grad_() for t in (send, rec, edge))
            o1, o2 = ref(s1, r1, e1), net(s2, r2, e2)
            o1 = o1 if isinstance(o1, tuple) else (o1,)
            o2 = o2 if isinstance(o2, tuple) else (o2,)
            for k, (a, b) in enumerate(zip(o2, o1)):
                report(f"out[{k}]", a, b)
            sum(o.square().sum() for o in o1).backward()
            sum(o.square().sum() for o in o2).backward()
            for nm, a, b in (("grad_send", s2, s1), ("grad_rec", r2, r1), ("grad_edge", e2, e1)):
                report(nm, a.grad, b.grad)
            for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
                report(f"grad {k}", p.grad, q.grad)
        except Exception:
            traceback.print_exc()
    for kin, hid, dout, ln in [(56, 256, 256, True), (256, 256, 17, False), (3, 128, 128, True), (128, 64, 64, True),
                               (130, 96, 40, True), (512, 512, 34, False), (18, 512, 512, True)]:
        print(f"[oracle mlp] {kin}->{hid}->{dout} ln={ln}")
        try:
            torch.manual_seed(kin)
            ref = og.make_mlp([kin, hid, dout], layer_norm=ln)
            net = hl.make_mlp([kin, hid, dout], layer_norm=ln)
            net.load_state_dict(ref.state_dict())
            net.to(dev)
            x = torch.randn(2, 1000, kin)
            x1, x2 = x.clone().requires_grad_(), x.to(dev).requires_grad_()
            y1, y2 = ref(x1), net(x2)
            report("out", y2, y1)
            y1.sin().sum().backward()
            y2.sin().sum().backward()
            report("grad_x", x2.grad, x1.grad)
            for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
                report(f"grad {k}", p.grad, q.grad)
        except Exception:
            traceback.print_exc()


if __name__ == "__main__":
    print(torch.__version__, torch.cuda.get_device_name(0))
    t0 = time.time()
    only = sys.argv[1:]
    if not only or "layers" in only:
        layer_cases()
    if not only or "oracle" in only:
        oracle_cases()
    if not only or "models" in only:
        model_cases()
    print(f"worst rel err {worst:.3e}   ({time.time() - t0:.1f}s)")
