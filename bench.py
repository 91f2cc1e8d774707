#!/usr/bin/env python
"""bench.py -- training-step throughput of the GraphCast-LAM hot path on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` (for N > 1
launched under torch.distributed.run, one rank per GPU over RCCL).  W untimed
warm-up steps, then exactly K training steps bracketed by barrier +
synchronize; the max over ranks is the job time; rank 0 prints ONE JSON line.

A step = ``ForecasterModule.training_step``-equivalent on one pre-resident
synthetic batch per GPU: standardised inputs -> AR rollout
(ar_steps) through GraphLAM -> masked wmse -> backward -> gradient all-reduce
(N > 1) -> AdamW.  Workload at N = 1 = BASELINE.json configs[1]: GraphCast-LAM
multiscale mesh on the synthetic MEPS-shaped 238x268 grid, 17 state variables,
hidden_dim 64, 4 processor layers, batch 1 per GPU, ar_steps 1, fp32.

Launch mode: zero-grad + forward + loss + backward are captured once into a HIP
graph and replayed per step (the gradient all-reduce for N > 1 and the fused AdamW
run after each replay); ``--eager`` issues every launch from Python instead.

Extra objects on the JSON line:
  roofline      the dominant kernel = the fused edge kernel (gather + edge MLP +
                LayerNorm + aggregation) on the m2g edge set (255 136 edges):
                algorithmic FLOPs 8*E*d^2 / measured launch time (HIP events on the
                launch stream, averaged over the K timed steps of a second,
                instrumented eager pass) vs. the fp32-MFMA peak of the dtype; the
                algorithmic HBM bytes/launch and the PMC-measured HBM traffic of the
                same kernel (profiles/round1/pmc_traffic.json, written by
                tools/pmc_collect.py) are reported next to it.
  cpu_baseline  the oracle (pure-torch restatement of the reference) timed on this
                box's host cores on the same workload, bounded to a few steps.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)
PEAK_HBM_GBS = 8000.0

CONFIGS = {
    # name: (nx, ny, n_state, n_forcing, n_static, hidden, proc_layers, ar_steps, batch_per_gpu, model, graph kwargs)
    "cfg2": dict(nx=238, ny=268, ns=17, nf=6, nst=4, d=64, L=4, T=1, B=1, model="graph_lam",
                 graph=dict(n_max_levels=None, hierarchical=False), boundary="frame"),
    "cfg1": dict(nx=64, ny=64, ns=5, nf=2, nst=1, d=16, L=4, T=1, B=2, model="graph_lam",
                 graph=dict(n_max_levels=1, hierarchical=False), boundary="random"),
    # the other BASELINE.json configs, one sample per GPU (not the default bench line; numbers go to profiles/)
    "cfg3": dict(nx=238, ny=268, ns=17, nf=6, nst=4, d=256, L=8, T=4, B=1, model="graph_lam",
                 graph=dict(n_max_levels=None, hierarchical=False), boundary="frame"),
    "cfg4": dict(nx=238, ny=268, ns=17, nf=6, nst=4, d=128, L=4, T=1, B=1, model="hi_lam",
                 graph=dict(n_max_levels=3, hierarchical=True), boundary="frame"),
    "cfg4p": dict(nx=238, ny=268, ns=17, nf=6, nst=4, d=128, L=4, T=1, B=1, model="hi_lam_parallel",
                  graph=dict(n_max_levels=3, hierarchical=True), boundary="frame"),
    "cfg5": dict(nx=238, ny=268, ns=17, nf=6, nst=4, d=512, L=8, T=8, B=1, model="graph_lam",
                 graph=dict(n_max_levels=None, hierarchical=False), boundary="frame"),
}


def build(cfg, device, seed_offset=0, oracle=False):
    from neural_lam_amd import graph as G
    from neural_lam_amd.datastore import SyntheticDatastore

    ds = SyntheticDatastore(cfg["nx"], cfg["ny"], cfg["ns"], cfg["nf"], cfg["nst"], root_path="/tmp/nlam_bench",
                            boundary=cfg["boundary"], seed=0)
    ext = ds.get_xy_extent("state")
    raw = G.create_regular_grid_graph(ds.get_xy("state"), **cfg["graph"])
    graph = G.normalise_graph(raw, max(ext[1] - ext[0], ext[3] - ext[2]))
    torch.manual_seed(42)  # weights: default init under seed 42
    if oracle:
        from oracle import models as om

        predictor = om.MODELS[cfg["model"]](ds, graph, hidden_dim=cfg["d"], processor_layers=cfg["L"])
        forecaster = om.ARForecaster(predictor, ds)
        step = None
    else:
        from neural_lam_amd import models as hm

        predictor = hm.MODELS[cfg["model"]](ds, graph=graph, hidden_dim=cfg["d"], processor_layers=cfg["L"])
        forecaster = hm.ARForecaster(predictor, ds)
        step = hm.ForecasterStep(forecaster, ds).to(device)
    N, B, T = ds.num_grid_points, cfg["B"], cfg["T"]
    g = torch.Generator().manual_seed(123 + seed_offset)  # inputs ~ N(0,1), seed 123 (+rank)
    init = torch.randn(B, 2, N, cfg["ns"], generator=g)
    target = torch.randn(B, T, N, cfg["ns"], generator=g)
    forcing = torch.randn(B, T, N, cfg["nf"] * 3, generator=g)
    batch = tuple(t.to(device) for t in (init, target, forcing))
    return ds, graph, raw, forecaster, step, batch


def cpu_baseline(cfg, budget_s=25.0):
    """Oracle training step (fwd + wmse + bwd + AdamW) on the host cores."""
    from oracle import models as om

    ncpu = os.cpu_count() or 1
    ds, _, _, forecaster, _, batch = build(cfg, torch.device("cpu"), oracle=True)
    pvs, mask = om.per_var_std_uniform(ds), om.interior_mask_bool(ds)
    opt = torch.optim.AdamW(forecaster.parameters(), lr=1e-3, betas=(0.9, 0.95))

    def one():
        opt.zero_grad(set_to_none=True)
        _, loss = om.training_loss(forecaster, batch, pvs, mask)
        loss.backward()
        opt.step()

    # torch's CPU scatter/index_add paths degrade badly when oversubscribed on a
    # many-core host: probe a few thread counts (one step each) and keep the best.
    best, cores = None, 1
    for n in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(n)
        t0 = time.perf_counter()
        one()
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, cores = dt, n
        if time.perf_counter() - t0 > budget_s / 3:
            break
    torch.set_num_threads(cores)
    one()  # warm-up
    times = []
    t_start = time.perf_counter()
    while len(times) < 5 and (time.perf_counter() - t_start) < budget_s:
        t0 = time.perf_counter()
        one()
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {
        "value": cfg["B"] * cfg["T"] / med,
        "unit": "sample-steps/s",
        "ms_per_step": med * 1e3,
        "cores": cores,
        "kind": "port",
        "sample": f"{len(times)} full training steps of the same workload (median), oracle = PyG-free torch fp32 "
                  f"restatement of the reference, torch.set_num_threads({cores}) = best of a {{8,16,32,64}}-thread probe "
                  f"on a {ncpu}-core host",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="issue every launch from Python instead of replaying a HIP graph")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"],
                    help="bf16 = run the step inside torch.autocast(bfloat16), as Lightning --precision bf16-mixed does (cfg5)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback); CPU numbers come from the cpu_baseline leg only")
    # dry run of the N > 1 code on a 1-GPU box: NLAM_BENCH_DRYRUN=1 puts every rank on cuda:0 over gloo (numbers are
    # meaningless then; it only checks the multi-rank control flow)
    dryrun = os.environ.get("NLAM_BENCH_DRYRUN") == "1"
    if dryrun:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dryrun:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)  # "nccl" is RCCL on ROCm

    from neural_lam_amd import ops
    from neural_lam_amd.trainer import Trainer

    ds, graph, raw, forecaster, step, batch = build(cfg, device, seed_offset=rank)
    trainer = Trainer(step, lr=1e-3, use_graph=not args.eager)
    amp = torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.precision == "bf16")
    amp.__enter__()   # whole run (capture included) inside the autocast region; exited before the CPU baseline

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(*batch)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(*batch)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * cfg["B"] * cfg["T"] * args.steps / elapsed

    # forecast throughput (inference rollout, no grad), reported alongside; same launch mode as training
    with torch.no_grad():
        for _ in range(2):
            step.forecaster(batch[0], batch[2], batch[1])
        torch.cuda.synchronize()
        if args.eager:
            run_forecast = lambda: step.forecaster(batch[0], batch[2], batch[1])  # noqa: E731
        else:
            fgraph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(fgraph):
                step.forecaster(batch[0], batch[2], batch[1])
            run_forecast = fgraph.replay
        run_forecast()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run_forecast()
        torch.cuda.synchronize()
        fc_elapsed = time.perf_counter() - t0
    forecast_steps_per_s = world * cfg["B"] * cfg["T"] * args.steps / fc_elapsed

    roofline = None
    if not args.no_roofline:
        # second, instrumented pass: HIP events (on the launch stream = torch's current
        # stream) around every nlam_mlp_fwd launch; pick the m2g edge launch (largest E)
        ops.PROFILE.reset(enabled=True)
        trainer.use_graph = False   # per-launch HIP events need eager launches
        for _ in range(args.steps):
            trainer.step(*batch)
        torch.cuda.synchronize()
        recs = ops.PROFILE.collect()
        ops.PROFILE.reset(enabled=False)
        E = int(raw["m2g_edge_index"].shape[1])
        Nr = int(raw["m2g_edge_index"][1].max()) + 1
        d = cfg["d"]
        key = ("mlp_fwd", E * cfg["B"], 3 * d, d, d)
        if key in recs and recs[key]:
            avg_ms = sum(recs[key]) / len(recs[key])
            flops = 8.0 * E * cfg["B"] * d * d  # edge MLP: 2*E*(3d*d + d*d)
            achieved = flops / (avg_ms * 1e-3) / 1e12
            # algorithmic HBM bytes of one training-mode launch: edge rows in, z1 + xhat (+rstd) out per edge,
            # aggregate out per receiver; sender / receiver rows and the weights are L2 / MALL resident
            alg_bytes = cfg["B"] * (E * (4 * d + 2 * 4 * d + 4) + Nr * 4 * d)
            traffic = None
            tfile = ROOT / "profiles" / "round1" / "pmc_traffic.json"
            if tfile.exists():
                try:
                    traffic = json.loads(tfile.read_text()).get("m2g_edge_fwd_bytes_per_launch")
                except Exception:
                    traffic = None
            mode = "bf16" if args.precision == "bf16" else ops.MATMUL_MODE
            terms = {"f32": 0, "bf16": 1, "bf16x2": 2, "bf16x3": 3}[mode]
            if d > 64 and terms == 2:
                terms = 3   # the wide kernels instantiate one and three terms
            mfmas = terms * (terms + 1) // 2   # bf16 MFMAs executed per algorithmic product block
            if d <= 64:
                hb = (d + 31) // 32
                kname = ("mlp_fwd_bf_kernel<%d,%d,%d>" % (hb, hb, terms)) if terms and d % 32 == 0 else "mlp_fwd_kernel<%d,%d>" % (hb, hb)
            else:
                kname = "mlp_fwd_wbf_kernel<%d,..>" % terms if terms else "mlp_fwd_wide_kernel"
            roofline = {
                "bound": "mfma",
                "kernel": kname + " (m2g edge set: gather + edge MLP + LayerNorm + aggregate, training mode)",
                "matmul_mode": mode,
                # fp32-result FLOPs against the fp32 MFMA peak (the contract of the path is fp32 results); the same
                # launch on EXECUTED bf16 matrix FLOPs against the dense bf16 peak is reported next to it
                "executed_bf16_tflops": (achieved * mfmas) if terms else None,
                "frac_of_bf16_dense_peak": (achieved * mfmas / PEAK_BF16_MFMA_TFLOPS) if terms else None,
                "achieved": achieved,
                "peak": PEAK_FP32_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                "avg_launch_ms": avg_ms,
                "launches": len(recs[key]),
                "algorithmic_flops_per_launch": flops,
                "algorithmic_hbm_bytes_per_launch": alg_bytes,
                "hbm_GBps_algorithmic": alg_bytes / (avg_ms * 1e-3) / 1e9,
                "hbm_frac_of_8TBps": alg_bytes / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                "traffic": traffic,
            }

    amp.__exit__(None, None, None)
    if rank == 0:
        out = {
            "metric": "training sample-steps/s (fwd+wmse+bwd+allreduce+AdamW), GraphCast-LAM, MEPS-shaped grid",
            "value": value,
            "unit": "sample-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16",
            "matmul_mode": ops.MATMUL_MODE if args.precision == "fp32" else "bf16 (autocast)",
            "data": "synthetic",
            "launch_mode": "eager" if args.eager else "hip_graph (zero-grad + fwd + loss + bwd captured once; all-reduce + AdamW after each replay)",
            "forecast_steps_per_s": forecast_steps_per_s,
            "final_loss": float(loss),
            "config": {
                "workload": f"{args.config}: {cfg['model']}, grid {cfg['nx']}x{cfg['ny']} ({ds.num_grid_points} nodes), "
                            f"{cfg['ns']} state vars, hidden_dim {cfg['d']}, {cfg['L']} processor layers, "
                            f"ar_steps {cfg['T']}, batch {cfg['B']}/GPU, g2m/m2m/m2g edges "
                            f"{raw['g2m_edge_index'].shape[1]}/{raw['m2m_edge_index'][0].shape[1]}/{raw['m2g_edge_index'].shape[1]}",
                "global_batch": world * cfg["B"],
                "parallelism": f"dp{world}",
            },
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
