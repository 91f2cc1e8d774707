NLAM_LIB=neural_lam_amd/libnlam_hip_prev.so python tools/_dbg.py /tmp/a.pt
NLAM_LIB=neural_lam_amd/libnlam_hip.so python tools/_dbg.py /tmp/b.pt
python - <<'PY'
import torch
a, b = torch.load("/tmp/a.pt"), torch.load("/tmp/b.pt")
for k in a:
    d = (a[k] - b[k]).abs().max().item(); print(k, "equal" if torch.equal(a[k], b[k]) else "max diff %.3e rel %.3e" % (d, d / a[k].abs().max().item()))
PY
