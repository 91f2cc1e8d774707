#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table from hipcc's resource-usage remarks
(cross-compiles for gfx950; no GPU needed)."""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "neural_lam_amd" / "csrc" / "nlam_hip.hip"


def main():
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", str(SRC), "-o", "/tmp/nlam_res.o",
           "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp")
    rows, cur = [], None
    for line in out.stderr.splitlines():
        m = re.search(r"remark: (.*?) \[-Rpass", line)
        if not m:
            if "error" in line:
                print(line)
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            name = t.split(":", 1)[1].strip()
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            cur = {"name": name.replace("(anonymous namespace)::", "").split("(")[0][:58]}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    for r in rows:
        print("{:58s} VGPR {:>4s} AGPR {:>4s} SGPR {:>4s} scratch {:>5s} occ {:>2s}".format(
            r["name"], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("TotalSGPRs", "?"),
            r.get("ScratchSize [bytes/lane]", "?"), r.get("Occupancy [waves/SIMD]", "?")))
    return out.returncode


if __name__ == "__main__":
    sys.exit(main())
