import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return torch.load(GOLDEN / f"{name}.pt", map_location="cpu", weights_only=True)


@pytest.fixture(scope="session")
def golden_layers():
    cases = dict(load_golden("layers")["cases"])
    cases.update(load_golden("layers_wide")["cases"])  # d = 128 / 256 (wide kernels)
    cases.update(load_golden("layers_wide_chunked")["cases"])  # SplitMLPs at d = 128 (the cfg4p / HiLAMParallel width)
    return cases


def to64(t):
    return t.to(torch.int64) if t.dtype == torch.int32 else t


def graph_from_case(case, which="ref_graph_loaded"):
    g = {}
    for k, v in case[which].items():
        g[k] = [to64(t) for t in v] if isinstance(v, list) else to64(v)
    return g


def rel_err(a, b):
    """max|a-b| / max|b| : the parity metric of BASELINE.md (1e-4 in fp32)."""
    a, b = a.detach(), b.detach()
    denom = float(b.abs().max())
    return float((a - b).abs().max()) / (denom if denom > 0 else 1.0)
