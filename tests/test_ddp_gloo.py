"""Data-parallel path on CPU: world_size 2, gloo.  The collective logic (flat
gradient buckets, hooks, all-reduce, 1/world scaling) is device-independent; the
HIP AdamW kernel is swapped for an injected torch optimizer because there is no
GPU here."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


class _TorchAdamW:
    def __init__(self, flat_param, flat_grad, lr=1e-2):
        self.g = flat_grad
        self.p = torch.nn.Parameter(flat_param)  # shares storage with the flat buffer
        self.p.data = flat_param
        self.opt = torch.optim.AdamW([self.p], lr=lr, betas=(0.9, 0.95))

    def step(self, grad_scale):
        self.p.grad = self.g * grad_scale
        self.opt.step()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, bucket_bytes, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from neural_lam_amd.trainer import Trainer
    from oracle import gnn_layers as og

    torch.manual_seed(0)  # identical replicas
    ei = torch.stack([torch.randint(0, 6, (20,)), torch.randint(0, 5, (20,))])
    ei[1, -1] = 4

    class Step(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net = og.InteractionNet(ei, 8)
            self.unused = torch.nn.Linear(3, 3)  # never gets a gradient: finish_step must still reduce it

        def forward(self, send, rec, edge):
            r, e = self.net(send, rec, edge)
            return (r.square().mean() + e.square().mean(),)

    model = Step()
    trainer = Trainer(model, optimizer_factory=lambda p, g: _TorchAdamW(p, g), bucket_bytes=bucket_bytes)
    g = torch.Generator().manual_seed(100 + rank)  # different sample per rank
    batch = (torch.randn(6, 8, generator=g), torch.randn(5, 8, generator=g), torch.randn(20, 8, generator=g))
    losses = [float(trainer.step(*batch)) for _ in range(3)]
    torch.save({"flat": trainer.fp.flat.clone(), "grad": trainer.fp.grad.clone(), "losses": losses,
                "nbuckets": len(trainer.buckets.bounds), "batch": batch}, f"{out_dir}/rank{rank}.pt")
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes", [32 << 20, 1024])
def test_two_rank_gloo_matches_single_process_average(tmp_path, bucket_bytes):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), bucket_bytes, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    # replicas stay bit-identical after 3 steps
    assert torch.equal(r0["flat"], r1["flat"]) and torch.equal(r0["grad"], r1["grad"])
    if bucket_bytes == 1024:
        assert r0["nbuckets"] > 1

    # single process: average of the two per-rank losses reproduces the same parameters
    sys.path.insert(0, str(ROOT))
    from neural_lam_amd.trainer import Trainer
    from oracle import gnn_layers as og

    torch.manual_seed(0)
    ei = torch.stack([torch.randint(0, 6, (20,)), torch.randint(0, 5, (20,))])
    ei[1, -1] = 4

    class Both(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net = og.InteractionNet(ei, 8)
            self.unused = torch.nn.Linear(3, 3)

        def forward(self, b0, b1):
            tot = 0.0
            for send, rec, edge in (b0, b1):
                r, e = self.net(send, rec, edge)
                tot = tot + r.square().mean() + e.square().mean()
            return (tot / 2,)

    ref = Trainer(Both(), optimizer_factory=lambda p, g: _TorchAdamW(p, g))
    for _ in range(3):
        ref.step(r0["batch"], r1["batch"])
    assert torch.allclose(ref.fp.flat, r0["flat"], rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------
# GPU: the HIP-graph step under an initialised process group (two ranks sharing cuda:0 over gloo;
# RCCL itself needs one GPU per rank and is exercised by the driver's multi-GPU bench)
# ---------------------------------------------------------------------------
def _gpu_worker(rank, world, port, use_graph, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    try:
        probe = torch.ones(4, device=dev)
        dist.all_reduce(probe)
        assert float(probe[0]) == world
    except Exception as exc:  # this torch build's gloo cannot reduce device tensors
        torch.save({"unsupported": repr(exc)}, f"{out_dir}/rank{rank}.pt")
        dist.destroy_process_group()
        return
    from neural_lam_amd import gnn_layers as hl
    from neural_lam_amd.trainer import Trainer

    torch.manual_seed(0)  # identical replicas
    ei = torch.stack([torch.randint(0, 60, (900,)), torch.randint(0, 50, (900,))])
    ei[1, -1] = 49

    class Step(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net = hl.InteractionNet(ei, 64)

        def forward(self, send, rec, edge):
            r, e = self.net(send, rec, edge)
            return (r.square().mean() + e.square().mean(),)

    trainer = Trainer(Step().to(dev), lr=1e-2, use_graph=use_graph)
    g = torch.Generator().manual_seed(100 + rank)  # different sample per rank
    batch = tuple(torch.randn(1, n, 64, generator=g).to(dev) for n in (60, 50, 900))
    losses = [float(trainer.step(*batch)) for _ in range(4)]
    torch.cuda.synchronize()
    torch.save({"flat": trainer.fp.flat.cpu(), "grad": trainer.fp.grad.cpu(), "losses": losses, "graph": trainer._graph is not None,
                "batch": tuple(b.cpu() for b in batch)}, f"{out_dir}/rank{rank}.pt")
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("use_graph", [False, True])
def test_two_rank_step_on_gpu_matches_single_process_average(tmp_path, use_graph):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    world = 2
    mp.spawn(_gpu_worker, args=(world, _free_port(), use_graph, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    if "unsupported" in r0:
        pytest.skip(f"gloo cannot all-reduce device tensors here: {r0['unsupported']}")
    assert torch.equal(r0["flat"], r1["flat"]) and torch.equal(r0["grad"], r1["grad"])   # replicas stay bit-identical
    assert r0["graph"] == use_graph   # the captured step really ran (no silent fallback to eager)

    sys.path.insert(0, str(ROOT))
    from neural_lam_amd import gnn_layers as hl
    from neural_lam_amd.trainer import Trainer

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ei = torch.stack([torch.randint(0, 60, (900,)), torch.randint(0, 50, (900,))])
    ei[1, -1] = 49

    class Both(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net = hl.InteractionNet(ei, 64)

        def forward(self, s0, r0_, e0, s1, r1_, e1):
            tot = 0.0
            for send, rec, edge in ((s0, r0_, e0), (s1, r1_, e1)):
                r, e = self.net(send, rec, edge)
                tot = tot + r.square().mean() + e.square().mean()
            return (tot / 2,)

    ref = Trainer(Both().to(dev), lr=1e-2)
    both = tuple(b.to(dev) for b in (*r0["batch"], *r1["batch"]))
    for _ in range(4):
        ref.step(*both)
    assert torch.allclose(ref.fp.flat.cpu(), r0["flat"], rtol=2e-5, atol=2e-6)


def _rccl_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # "nccl" is RCCL on ROCm
    from neural_lam_amd import gnn_layers as hl
    from neural_lam_amd.trainer import Trainer

    warm = torch.ones(8, device=dev)
    dist.all_reduce(warm)   # communicator + watchdog thread are live before the capture
    torch.manual_seed(0)
    ei = torch.stack([torch.randint(0, 60, (900,)), torch.randint(0, 50, (900,))])
    ei[1, -1] = 49

    class Step(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net = hl.InteractionNet(ei, 64)

        def forward(self, send, rec, edge):
            r, e = self.net(send, rec, edge)
            return (r.square().mean() + e.square().mean(),)

    trainer = Trainer(Step().to(dev), lr=1e-2, use_graph=True)
    batch = tuple(torch.randn(1, n, 64, device=dev) for n in (60, 50, 900))
    losses = []
    for _ in range(4):
        losses.append(float(trainer.step(*batch)))
        dist.all_reduce(trainer.fp.grad)   # what the N > 1 step issues after every replay
    torch.cuda.synchronize()
    torch.save({"losses": losses, "graph": trainer._graph is not None}, f"{out_dir}/rccl.pt")
    dist.destroy_process_group()


@pytest.mark.gpu
def test_graph_capture_with_live_rccl_process_group(tmp_path):
    """One rank, backend "nccl" (= RCCL): the HIP-graph capture of the step has to coexist with an initialised
    communicator and its watchdog thread (capture_error_mode="thread_local"), and collectives issued between
    replays must keep working.  More ranks need more GPUs; the driver's multi-GPU bench covers those."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    r = torch.load(tmp_path / "rccl.pt", weights_only=False)
    assert r["graph"], "capture fell back to eager launches"
    assert all(l == l for l in r["losses"]) and r["losses"][-1] < r["losses"][0]
