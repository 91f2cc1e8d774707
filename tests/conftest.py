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


@pytest.fixture(autouse=True)
def _poison_gpu_allocator(request):
    """GPU tests: hand the caching allocator NaN-filled memory before every test, so that a kernel which leaves part of a
    ``torch.empty`` output unwritten (rows of isolated receivers, padded columns, ...) produces NaNs instead of whatever a
    previous test left there -- fresh device memory is zero, which hid exactly such a bug in the wide backward for two
    rounds.  NLAM_TEST_POISON=0 switches it off."""
    import os

    if request.node.get_closest_marker("gpu") is None or not torch.cuda.is_available() or os.environ.get("NLAM_TEST_POISON", "1") != "1":
        yield
        return
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    n = min(int(free * 0.25), 6 << 30) // 4
    if n > 0:
        block = torch.full((n,), float("nan"), device="cuda", dtype=torch.float32)
        del block   # back to the caching allocator: the test's allocations are carved out of it
    yield
