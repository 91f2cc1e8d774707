"""Run a command under rocprofv3 once per PMC pass (counters only + kernel trace, as the
pool requires) and print per-kernel averages.  On the GPU box:

    python tools/pmc_collect.py TAG -- python tools/kernel_bench.py m2g 4 64

Output: gpurun_out/pmc_TAG.md (+ raw csv under gpurun_out/pmc_TAG/).
Units: SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are quad-cycles summed over all waves;
SQ_VALU_MFMA_BUSY_CYCLES are cycles summed over SIMDs; FETCH_SIZE / WRITE_SIZE are KiB with
the gfx950 caveat of MI355X_MICROARCH.md (FETCH_SIZE reads 1/2 of a wide streaming read)."""
import csv
import os
import subprocess
import sys
from collections import defaultdict
from pathlib import Path

PASSES = {
    "wave": "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM",
    "insts": "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE",
    "mem": "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS",
    "fetch": "FETCH_SIZE",
    "write": "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum",
}


def main():
    tag = sys.argv[1]
    sep = sys.argv.index("--")
    passes = sys.argv[2:sep] or list(PASSES)
    cmd = sys.argv[sep + 1:]
    root = Path(os.environ.get("GRAFT_REPO_ROOT", ".")).resolve()
    out = root / "gpurun_out" / f"pmc_{tag}"
    out.mkdir(parents=True, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    stats = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> values
    dur = defaultdict(list)
    for name in passes:
        d = out / name
        full = ["rocprofv3", "--kernel-trace", "--pmc", *PASSES[name].split(), "-d", str(d), "-o", "p", "--output-format", "csv", "--"] + cmd
        r = subprocess.run(full, env=env, cwd=str(root), capture_output=True, text=True)
        if r.returncode != 0:
            print(f"pass {name} failed rc={r.returncode}\n{r.stderr[-2000:]}")
            continue
        for f in d.rglob("*counter_collection.csv"):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
                    key = (k, f'{row.get("Grid_Size", "")} lds={row.get("LDS_Block_Size", "")}')
                    stats[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for f in d.rglob("*kernel_trace.csv"):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
                    dur[(k, row.get("Grid_Size", row.get("Grid_Size_X", "")))].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    lines = [f"# PMC per-kernel averages: {' '.join(cmd)}", ""]
    for key in sorted(stats, key=lambda k: -sum(stats[k].get("SQ_WAVE_CYCLES", [0]))):
        if not key[0].startswith(("mlp_", "wgrad", "segment", "reduce", "pack", "adamw")):
            continue
        c = stats[key]
        n = max(len(v) for v in c.values())
        dd = [v for kk, v in dur.items() if kk[0] == key[0]]
        dmean = sum(sum(v) for v in dd) / max(1, sum(len(v) for v in dd)) if dd else float("nan")
        lines.append(f"## {key[0]}  grid={key[1]}  ({n} dispatches, mean duration over same-name kernels {dmean:.1f} us)")
        for cn in sorted(c):
            v = c[cn]
            lines.append(f"  {cn:34s} {sum(v) / len(v):16.0f}")
        lines.append("")
    text = "\n".join(lines)
    (root / "gpurun_out" / f"pmc_{tag}.md").write_text(text)
    import json

    js, by_grid = {}, {}
    for key in stats:
        if not key[0].startswith(("mlp_", "wgrad", "segment", "reduce", "pack", "adamw", "linear")):
            continue
        c = stats[key]
        dd = [v for kk, v in dur.items() if kk[0] == key[0]]
        dmean = sum(sum(v) for v in dd) / max(1, sum(len(v) for v in dd)) if dd else None
        js[key[0]] = {"grid": key[1], "mean_duration_us": dmean, **{cn: sum(v) / len(v) for cn, v in c.items()}}
        # the same kernel launched on different grids (edge / node launches of a whole step) kept apart
        by_grid[f"{key[0]}|{key[1].split()[0]}"] = {"dispatches": max(len(v) for v in c.values()), **{cn: sum(v) / len(v) for cn, v in c.items()}}
    (root / "gpurun_out" / f"pmc_{tag}.json").write_text(json.dumps({"command": " ".join(cmd), "kernels": js, "by_grid": by_grid}, indent=1))
    print(text)


if __name__ == "__main__":
    main()
