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
