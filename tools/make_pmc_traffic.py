"""profiles/roundN/pmc_traffic.json from the per-kernel PMC averages of tools/pmc_collect.py.

    python tools/pmc_collect.py m2g_edge fetch write wave insts -- python tools/kernel_bench.py m2g 3 64 edge
    python tools/pmc_collect.py m2m_edge fetch write wave insts -- python tools/kernel_bench.py m2m 3 64 edge
    python tools/make_pmc_traffic.py gpurun_out/pmc_m2g_edge.json:255136 gpurun_out/pmc_m2m_edge.json:57616 > profiles/round3/pmc_traffic.json

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB): on gfx950 rocprofv3's FETCH_SIZE tallies 128-byte read requests
at 64 bytes (MI355X_MICROARCH.md, HBM section), WRITE_SIZE is taken as reported (uncalibrated).  Keys are the launch keys
of bench.py's roofline rows (kind:rows:k:n)."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from neural_lam_amd._lib import source_stamp  # noqa: E402

d = 64
NAMES = {   # kernel name prefix -> (kind, k, n) of the launch key at hidden_dim 64
    "mlp_fwd_bf_kernel<2, 2, 3, false, false, false": ("mlp_fwd", 3 * d, d),   # prefixes: later template arguments (CAT, RO) follow
    "mlp_fwd_bf_kernel<2, 2, 3, false, true, false": ("mlp_fwd", 3 * d, d),
    "mlp_bwd_fast_kernel<2, 2, ": ("mlp_bwd", 3 * d, d),   # any term count (round 5: the narrow backward runs two-term by default)
    "wgrad_dma_kernel<3, false>": ("wgrad", d, 3 * d),
    "wgrad_dma_kernel<1, true>": ("wgrad", d, d),
}
out = {"note": "HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE KiB (gfx950 FETCH_SIZE correction); separate --pmc passes, "
               "kernel trace only; one InteractionNet edge stage (forward in training mode, backward, both weight gradients)",
       "library_stamp": source_stamp(),   # bench.py refuses this file once the kernels' sources change
       "bytes_per_launch": {}, "counters": {}, "commands": []}
# launches that only exist inside the whole step (`path:step`: a pmc_collect pass over the replayed cfg2 step, keyed by kernel AND
# grid -- the node launches run the edge kernels on 206 workgroups): (kernel prefix, grid threads) -> launch key of bench.py
STEP = {
    ("mlp_bwd_fast_group_kernel<2, 2, ", None): "mlp_bwd_group_lw:398549:4:64",   # the four static-feature embedders
    ("mlp_bwd_fast_kernel<2, 2, ", 206 * 512): f"mlp_bwd:6561:{2 * d}:{d}",        # node MLP backward, 6 561 mesh nodes
    ("mlp_fwd_bf_kernel<2, 2, 3, false, true, false", 206 * 512): f"mlp_fwd:6561:{2 * d}:{d}",
}
# round 6: the wide edge stages (`path:rows:d` with d = 256: fp32 class, three terms; d = 512: one term, bf16 storage) -- kernel name
# prefix -> (kind, k, n); the two weight gradients of an edge MLP share one launch key (m = n = d): their mean
WIDE = {
    512: {"mlp_fwd_edge_kernel<1, 512": ("mlp_fwd", 1536, 512), "mlp_fwd_wbf_kernel<1,": ("mlp_fwd", 1536, 512),
          "mlp_bwd_edge_kernel<1, 512": ("mlp_bwd", 1536, 512), "mlp_bwd_wbf_kernel<1,": ("mlp_bwd", 1536, 512),
          "wgrad_ldma_kernel<1, true": ("wgrad", 512, 512)},
    256: {"mlp_fwd_edge_kernel<3, 256": ("mlp_fwd", 768, 256), "mlp_fwd_wbf_kernel<3,": ("mlp_fwd", 768, 256),
          "mlp_bwd_edge_kernel<3, 256": ("mlp_bwd", 768, 256), "mlp_bwd_wbf_kernel<3,": ("mlp_bwd", 768, 256), "wgrad_wbf_kernel<3,": ("wgrad", 256, 256), "wgrad_ldma_kernel<3,": ("wgrad", 256, 256)},
}
for arg in sys.argv[1:]:
    parts = arg.split(":")
    path, rows = parts[0], parts[1]
    js = json.load(open(path))
    out["commands"].append(js["command"])
    if len(parts) == 3:
        # per (kernel, grid): the node-level weight gradients of the same stage run the same kernels on a smaller grid -- the edge
        # launches are the ones on the LARGEST grid of each key
        acc = {}
        for kg, c in js.get("by_grid", {}).items():
            name, grid = kg.rsplit("|", 1)
            for prefix, (kind, k, n) in WIDE[int(parts[2])].items():
                if name.startswith(prefix) and "FETCH_SIZE" in c and "WRITE_SIZE" in c and grid.isdigit():
                    key = f"{kind}:{rows}:{k}:{n}"
                    acc.setdefault(key, []).append((int(grid), int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024), name, c))
        for key, lst in acc.items():
            gmax = max(g for g, _, _, _ in lst)
            lst = [e for e in lst if e[0] == gmax]
            out["bytes_per_launch"][key] = int(sum(b for _, b, _, _ in lst) / len(lst))
            out["counters"][key] = {"kernels": [nm for _, _, nm, _ in lst], "grid": gmax, **lst[0][3]}
        continue
    if rows == "step":
        for kg, c in js.get("by_grid", {}).items():
            name, grid = kg.rsplit("|", 1)
            for (prefix, g), key in STEP.items():
                if name.startswith(prefix) and (g is None or str(g) == grid) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                    out["bytes_per_launch"][key] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
                    out["counters"][key] = {"kernel": name, "grid": grid, **c}
        continue
    for name, c in js["kernels"].items():
        for prefix, (kind, k, n) in NAMES.items():
            if name.startswith(prefix) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                key = f"{kind}:{rows}:{k}:{n}"
                out["bytes_per_launch"][key] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
                out["counters"][key] = {"kernel": name, **{kk: vv for kk, vv in c.items() if kk != "grid"}}
print(json.dumps(out, indent=1))
