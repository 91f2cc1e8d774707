"""Per-step timeline of a replayed training step from a rocprofv3 kernel trace (csv): python tools/queue_timeline.py <t_kernel_trace.csv> [out.txt]
Steps are delimited by the fused AdamW launch; the last complete step of the modal launch count is printed: per (kernel, queue)
totals first, then every launch (start us, duration us, queue, workgroups, name)."""
import collections
import csv
import sys


def short(s):
    s = s.replace("(anonymous namespace)::", "").replace("void ", "").replace("nlam_detail::", "")
    return s.split("(")[0][:90]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"].lower()]
    gaps = collections.Counter(idx[i + 1] - idx[i] for i in range(len(idx) - 1))
    mode = max((g for g in gaps if g > 8), key=lambda g: gaps[g])
    for i in range(len(idx) - 2, 0, -1):
        if idx[i + 1] - idx[i] == mode:
            a, b = idx[i] + 1, idx[i + 1] + 1
            break
    step = rows[a:b]
    t0 = min(int(r["Start_Timestamp"]) for r in step)
    t1 = max(int(r["End_Timestamp"]) for r in step)
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    out.write(f"step span {(t1 - t0) / 1e6:.3f} ms, {len(step)} launches\n")
    agg = collections.defaultdict(lambda: [0, 0.0])
    busy = collections.defaultdict(float)
    for r in step:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        k = (short(r["Kernel_Name"]), r["Queue_Id"])
        agg[k][0] += 1
        agg[k][1] += d
        busy[r["Queue_Id"]] += d
    out.write("busy us per queue: " + ", ".join(f"q{q}: {v:.0f}" for q, v in sorted(busy.items())) + "\n")
    for (n, q), (k, t) in sorted(agg.items(), key=lambda x: -x[1][1]):
        out.write(f"{t:9.1f} us {k:4d}  q{q} {n}\n")
    out.write("\n")
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        wg = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
        out.write(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} q{r['Queue_Id']} g{wg:6d} {short(r['Kernel_Name'])}\n")


if __name__ == "__main__":
    main()
