#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage remarks (a stderr capture) as one line per kernel."""
import re
import subprocess
import sys


def parse(path):
    rows, cur = [], None
    for line in open(path, errors="replace"):
        m = re.search(r"remark: (.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    out = {}
    for r, n in zip(rows, names):
        n = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        out[n] = r
    return out


def main():
    new = parse(sys.argv[1])
    old = parse(sys.argv[2]) if len(sys.argv) > 2 else {}
    filt = sys.argv[3] if len(sys.argv) > 3 else ""
    for n, r in new.items():
        if filt and not re.search(filt, n):
            continue
        f = lambda d: "V%s A%s S%s spillS%s spillV%s scr%s occ%s lds%s" % (
            d.get("VGPRs", "?"), d.get("AGPRs", "?"), d.get("TotalSGPRs", "?"), d.get("SGPRs Spill", "?"), d.get("VGPRs Spill", "?"),
            d.get("ScratchSize [bytes/lane]", "?"), d.get("Occupancy [waves/SIMD]", "?"), d.get("LDS Size [bytes/block]", "?"))
        line = "%-62s %s" % (n[:62], f(r))
        if n in old and f(old[n]) != f(r):
            line += "   <- was " + f(old[n])
        print(line)


if __name__ == "__main__":
    main()
