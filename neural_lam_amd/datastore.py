"""Duck-typed synthetic datastore: only what the hot path reads.

The reference's datastores (``neural_lam/datastore/``) are an xarray/zarr I/O
layer that is out of scope.  The predictor stack touches
~10 attributes of a datastore; this class supplies them
from deterministic synthetic arrays so that the same object can drive the
reference's own model classes (tests/golden), the oracle and the HIP path.
"""
from __future__ import annotations

import datetime
from pathlib import Path
from types import SimpleNamespace

import numpy as np

from .graph import regular_grid_xy


class _Values:
    """Stand-in for an xarray.DataArray: the hot path only ever reads ``.values``."""

    def __init__(self, values):
        self.values = np.asarray(values)


class SyntheticDatastore:
    """Regular ``nx x ny`` grid with ``num_state/num_forcing/num_static`` variables.

    Defaults follow SURVEY.md §8(d): standardisation mean 0 / std 1, one-step
    difference mean 0 / std 1, static features ~ N(0,1) under ``seed``,
    boundary = frame of ``boundary_width`` cells (``boundary="random"`` gives the
    random 0/1 mask of tests/dummy_datastore.py:163-166).
    """

    def __init__(
        self,
        nx: int,
        ny: int,
        num_state: int = 17,
        num_forcing: int = 6,
        num_static: int = 4,
        root_path: str | Path = "/tmp/nlam_synth",
        spacing: float = 2500.0,
        boundary: str = "frame",
        boundary_width: int = 10,
        seed: int = 0,
        state_stats: dict | None = None,
    ):
        self.nx, self.ny = int(nx), int(ny)
        self._n = {"state": int(num_state), "forcing": int(num_forcing), "static": int(num_static)}
        self.root_path = Path(root_path)
        self.spacing = float(spacing)
        rng = np.random.default_rng(seed)
        self._xy = regular_grid_xy(self.nx, self.ny, self.spacing)
        n_grid = self.nx * self.ny
        self._static = (
            rng.standard_normal((n_grid, num_static)).astype(np.float32) if num_static > 0 else None
        )
        if boundary == "frame":
            m = np.ones((self.nx, self.ny), dtype=np.int64)
            w = boundary_width
            if 2 * w < min(self.nx, self.ny):
                m[w : self.nx - w, w : self.ny - w] = 0
            mask = m.reshape(-1)
        elif boundary == "random":
            mask = rng.integers(0, 2, size=n_grid)
        elif boundary == "none":
            mask = np.zeros(n_grid, dtype=np.int64)
        else:
            raise ValueError(f"unknown boundary kind {boundary!r}")
        self.boundary_mask = _Values(mask)
        ns = num_state
        stats = {
            "state_mean": np.zeros(ns, np.float32),
            "state_std": np.ones(ns, np.float32),
            "state_diff_mean_standardized": np.zeros(ns, np.float32),
            "state_diff_std_standardized": np.ones(ns, np.float32),
        }
        if state_stats:
            for k, v in state_stats.items():
                stats[k] = np.asarray(v, dtype=np.float32)
        self._state_stats = SimpleNamespace(**{k: _Values(v) for k, v in stats.items()})
        self._forcing_stats = SimpleNamespace(
            forcing_mean=_Values(np.zeros(num_forcing, np.float32)),
            forcing_std=_Values(np.ones(num_forcing, np.float32)),
        )
        self.step_length = datetime.timedelta(hours=3)

    # ---- the surface of SURVEY.md Appendix A ----
    @property
    def num_grid_points(self) -> int:
        return self.nx * self.ny

    def get_num_data_vars(self, category: str) -> int:
        return self._n[category]

    def get_vars_names(self, category: str) -> list[str]:
        return [f"{category}_var_{i}" for i in range(self._n[category])]

    def get_dataarray(self, category: str, split=None, standardize: bool = False):
        if category != "static":
            raise NotImplementedError("synthetic datastore only serves static fields")
        return None if self._static is None else _Values(self._static)

    def get_standardization_dataarray(self, category: str):
        if category == "state":
            return self._state_stats
        if category == "forcing":
            return self._forcing_stats
        raise KeyError(category)

    def get_xy(self, category: str = "state", stacked: bool = False):
        return self._xy.reshape(-1, 2) if stacked else self._xy

    def get_xy_extent(self, category: str = "state"):
        x, y = self._xy[..., 0], self._xy[..., 1]
        return [float(x.min()), float(x.max()), float(y.min()), float(y.max())]

    @property
    def graph_dir(self) -> Path:
        return self.root_path / "graph"


def meps_like_datastore(root_path, **kw) -> SyntheticDatastore:
    """The MEPS-shaped benchmark grid of BASELINE.md §3: 238 x 268, 17/6/4 variables."""
    return SyntheticDatastore(238, 268, 17, 6, 4, root_path=root_path, **kw)


def dummy_datastore(root_path, n: int = 64, **kw) -> SyntheticDatastore:
    """cfg1: tests/dummy_datastore.py-shaped square grid with 5/2/1 variables."""
    kw.setdefault("boundary", "random")
    return SyntheticDatastore(n, n, 5, 2, 1, root_path=root_path, **kw)
