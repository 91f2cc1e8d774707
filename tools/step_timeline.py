"""Timeline of ONE replayed training step from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- \
        python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-baseline --no-roofline
    python tools/step_timeline.py gpurun_out/trace/t_kernel_trace.csv > profiles/round2/cfg2_step_timeline.txt

The last complete step (from the batch copies that open a replay to the adamw_kernel that closes it) is printed as
start / duration / hardware queue / kernel / grid, followed by per-kernel totals of that step, the busy time of the
union of all kernels and the average concurrency."""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    cut = name.find("(")
    return (name[:cut] if cut > 0 else name)[:72]


def main():
    rows = []
    with open(sys.argv[1]) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"]),
                         int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1)))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if r[3].startswith("adamw_kernel")]
    if len(ends) < 2:
        raise SystemExit("need at least two optimizer steps in the trace")
    lo, hi = ends[-2] + 1, ends[-1]
    step = rows[lo : hi + 1]
    t0 = step[0][0]
    span = (step[-1][1] - t0) / 1e3
    queues = sorted({r[2] for r in step})
    qname = {q: f"q{k + 1}" for k, q in enumerate(queues)}
    print(f"# One replayed training step: span {span:.1f} us, {len(step)} kernels, {len(queues)} hardware queues")
    print("# start_us  dur_us  queue  kernel  grid")
    for s, e, q, n, g in step:
        print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  {qname[q]}  {n}  {g}")
    tot = defaultdict(lambda: [0, 0.0])
    for s, e, q, n, g in step:
        tot[n][0] += 1
        tot[n][1] += (e - s) / 1e3
    ssum = sum(v[1] for v in tot.values())
    # union of the busy intervals
    busy, cur_s, cur_e = 0.0, None, None
    for s, e, *_ in sorted(step):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    busy /= 1e3
    print(f"\n# per-kernel totals of this step (sum of durations {ssum:.1f} us; union busy {busy:.1f} us of the {span:.1f} us span; "
          f"average concurrency {ssum / busy:.2f})")
    for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"# {t:8.1f} us  {c:3d} x  {n}")


if __name__ == "__main__":
    main()
