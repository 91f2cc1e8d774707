"""Per hardware queue: kernels, busy time and the kernel families of the last replayed step of a rocprofv3 kernel trace."""
import csv
import sys
from collections import Counter, defaultdict

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), n.split("<")[0].split("(")[0][:28]))
rows.sort()
ends = [i for i, r in enumerate(rows) if r[3].startswith("adamw_kernel")]
step = rows[ends[-2] + 1 : ends[-1] + 1]
t0 = step[0][0]
print(f"step span {(step[-1][1] - t0) / 1e3:.1f} us, {len(step)} kernels")
byq = defaultdict(list)
for r in step:
    byq[r[2]].append(r)
for q, v in sorted(byq.items()):
    c = Counter(x[3] for x in v)
    print(f"  queue {q}: {len(v)} kernels, busy {sum(e - s for s, e, *_ in v) / 1e3:.0f} us, first {(v[0][0] - t0) / 1e3:.0f} last end {(max(x[1] for x in v) - t0) / 1e3:.0f}: " + ", ".join(f"{k} x{n}" for k, n in c.most_common(6)))
allq = Counter(r[2] for r in rows)
print("  queues over the whole trace:", dict(allq))
