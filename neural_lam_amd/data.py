"""The training data path on the GPU: samples cut out of time series that live in HBM.

Mirror of the reference's ``neural_lam/weather_dataset.py`` ``WeatherDataset`` for analysis data (one contiguous time
series per category, no ensemble axis) -- ``__len__`` (:118-197), ``__getitem__`` (:467-533): same constructor
arguments (``ar_steps``, ``num_past_forcing_steps``, ``num_future_forcing_steps``), same 4-tuple
``(init_states, target_states, forcing, target_times)``, same IndexError / negative-index behaviour -- but the
xarray slicing, the host tensors and the DataLoader collation are replaced by ONE launch (``nlam_window_batch``) that
writes a whole batch from the resident series, reading its sample indices on the device, optionally with
``ForecasterModule.on_after_batch_transfer`` (models/module.py:326-367) folded into the same pass.  A MEPS-sized year
(2 920 steps x 63 784 nodes x 17 + 6 variables, fp32) is 17 GB: it fits the 288 GB of one MI355X many times over, so
an epoch needs no host->device traffic at all (a resident permutation supplies the indices).

Forecast-type and ensemble datastores (weather_dataset.py:135-178, :233-254, :300-329) are out of scope here.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class DeviceWeatherDataset:
    """``WeatherDataset`` (weather_dataset.py:20-116) over device-resident series.

    state    (n_times, num_grid_nodes, num_state_vars)   float32
    forcing  (n_times, num_grid_nodes, num_forcing_vars) float32 or None
    times    (n_times,) int64 nanoseconds or None (then ``target_times`` are time indices)
    standardization: optional dict with ``state_mean, state_std, forcing_mean, forcing_std`` (the buffers
    ForecasterModule registers, module.py:159-215, std already clamped) for ``batch(..., standardize=True)``.
    """

    def __init__(self, state, forcing=None, times=None, ar_steps=3, num_past_forcing_steps=1, num_future_forcing_steps=1,
                 standardization=None, device="cuda"):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("DeviceWeatherDataset keeps its series in HBM and cuts samples with a HIP kernel: it needs a GPU "
                               "(there is no CPU fallback; the CPU restatement lives in oracle/data.py for the tests)")
        state = torch.as_tensor(state, dtype=torch.float32)
        if state.dim() != 3:
            raise ValueError("state must be (n_times, num_grid_nodes, num_state_vars)")
        self.state = state.to(dev).contiguous()
        self.forcing = None
        if forcing is not None:
            forcing = torch.as_tensor(forcing, dtype=torch.float32)
            if forcing.dim() != 3 or forcing.shape[1] != state.shape[1]:
                raise ValueError("forcing must be (n_times, num_grid_nodes, num_forcing_vars) on the same nodes as state")
            if forcing.shape[2] > 0:
                self.forcing = forcing.to(dev).contiguous()
        self.times = None if times is None else torch.as_tensor(times, dtype=torch.int64).to(dev).contiguous()
        if self.times is not None and self.times.shape != (state.shape[0],):
            raise ValueError("times must have one entry per state time step")
        self.ar_steps = int(ar_steps)
        self.num_past_forcing_steps = int(num_past_forcing_steps)
        self.num_future_forcing_steps = int(num_future_forcing_steps)
        self.device = dev
        self._lib = L.load()
        n_forc = -1 if self.forcing is None else self.forcing.shape[0]
        self._len = int(self._lib.nlam_window_len(self.state.shape[0], n_forc, self.ar_steps, self.num_past_forcing_steps,
                                                  self.num_future_forcing_steps))
        # one common time axis for the kernel: it indexes both series with the same time index
        self._n_times = int(self.state.shape[0] if self.forcing is None else min(self.state.shape[0], self.forcing.shape[0]))
        self.stats = None
        if standardization is not None:
            g = lambda k: torch.as_tensor(standardization[k], dtype=torch.float32).to(dev).contiguous()  # noqa: E731
            self.stats = {"state_mean": g("state_mean"), "state_std": g("state_std")}
            if self.forcing is not None:
                self.stats.update(forcing_mean=g("forcing_mean"), forcing_std=g("forcing_std"))

    # ---- the reference's surface ----
    @property
    def window(self):
        return self.num_past_forcing_steps + self.num_future_forcing_steps + 1

    @property
    def num_forcing_features(self):
        return 0 if self.forcing is None else self.forcing.shape[2] * self.window

    def __len__(self):
        return self._len

    def __getitem__(self, idx):
        """One UNSTANDARDISED sample, as the reference's dataset returns it (:479-480)."""
        n = len(self)
        idx = int(idx)
        if idx < 0:
            idx += n
        if not 0 <= idx < n:
            raise IndexError(f"index {idx} out of range for WeatherDataset of length {n}")
        init, target, forcing, times = self.batch(torch.tensor([idx], dtype=torch.int64, device=self.device), standardize=False)
        return init[0], target[0], forcing[0], times[0]

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def check_indices(self, indices):
        """IndexError (as weather_dataset.py:497-503) if any entry of a DEVICE-resident index tensor lies outside
        ``[0, len(self))`` -- ``batch`` does not validate such tensors (the kernel clamps the time index instead of
        faulting, so a bad index would silently repeat boundary time steps).  One host synchronisation: call it once per
        epoch on the permutation, or pass ``validate=True`` to ``batch`` while debugging."""
        idx = torch.as_tensor(indices).reshape(-1)
        if idx.numel():
            lo, hi = int(idx.min()), int(idx.max())
            if lo < 0 or hi >= len(self):
                raise IndexError(f"sample index out of range for WeatherDataset of length {len(self)}: [{lo}, {hi}]")
        return indices

    # ---- the batched launch ----
    def batch(self, indices, standardize=False, out=None, validate=False):
        """(init_states (B, 2, N, d), target_states (B, T, N, d), forcing (B, T, N, F * window), target_times (B, T)).

        ``indices``: a device int64 tensor is used as it is (not validated unless ``validate=True``, which costs a host
        synchronisation: the kernel clamps; see ``check_indices``); anything else is validated on the host like
        ``__getitem__``.  ``out``: optional tuple of four preallocated tensors (e.g. the
        static input buffers of a captured training step)."""
        if not (isinstance(indices, torch.Tensor) and indices.is_cuda):
            host = torch.as_tensor(indices, dtype=torch.int64).reshape(-1)
            n = len(self)
            host = torch.where(host < 0, host + n, host)
            if host.numel() and (int(host.min()) < 0 or int(host.max()) >= n):
                raise IndexError(f"sample index out of range for WeatherDataset of length {n}")
            indices = host.to(self.device)
        elif validate:
            self.check_indices(indices)
        indices = indices.to(torch.int64).contiguous()
        B, T = indices.numel(), self.ar_steps
        N, ds = self.state.shape[1], self.state.shape[2]
        fw = self.num_forcing_features
        if out is None:
            o = dict(device=self.device, dtype=torch.float32)
            out = (torch.empty((B, 2, N, ds), **o), torch.empty((B, T, N, ds), **o), torch.empty((B, T, N, fw), **o),
                   torch.empty((B, T), device=self.device, dtype=torch.int64))
        init, target, forcing, times = out
        for t_, shape in ((init, (B, 2, N, ds)), (target, (B, T, N, ds)), (forcing, (B, T, N, fw)), (times, (B, T))):
            if tuple(t_.shape) != shape or not t_.is_contiguous() or t_.device != self.state.device:
                raise ValueError(f"output buffer of shape {tuple(t_.shape)}: expected a contiguous {shape} tensor on {self.device}")
        if standardize and self.stats is None:
            raise ValueError("standardize=True needs the standardization statistics (constructor argument)")
        p = L.Window()
        p.state, p.forcing, p.sample_idx = _ptr(self.state), _ptr(self.forcing), _ptr(indices)
        p.init_states, p.target_states = _ptr(init), _ptr(target)
        p.forcing_windowed = _ptr(forcing) if fw else None
        p.times, p.target_times = _ptr(self.times), _ptr(times)   # times == None: the kernel reports the time index of every target step
        if standardize:
            p.state_mean, p.state_std = _ptr(self.stats["state_mean"]), _ptr(self.stats["state_std"])
            if fw:
                p.forcing_mean, p.forcing_std = _ptr(self.stats["forcing_mean"]), _ptr(self.stats["forcing_std"])
        p.n_times, p.nodes, p.d_state, p.batch = self._n_times, N, ds, B
        p.d_forcing = 0 if self.forcing is None else self.forcing.shape[2]
        p.ar_steps, p.num_past_forcing_steps, p.num_future_forcing_steps = T, self.num_past_forcing_steps, self.num_future_forcing_steps
        L.check(self._lib.nlam_window_batch(C.byref(p), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "nlam_window_batch")
        return init, target, forcing, times

    def epoch_permutation(self, seed=0):
        """A resident random permutation of the sample indices: ``perm[k * B : (k + 1) * B]`` feeds ``batch`` with no host copy."""
        g = torch.Generator(device="cpu").manual_seed(int(seed))
        return torch.randperm(len(self), generator=g).to(self.device)
