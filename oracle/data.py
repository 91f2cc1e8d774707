"""ORACLE -- test infrastructure only (see oracle/__init__.py).

numpy restatement of the reference's training-sample construction for ANALYSIS data (one contiguous time series, no
ensemble axis): ``neural_lam/weather_dataset.py`` ``WeatherDataset.__len__`` (:118-197), ``_slice_state_time`` (:199-262,
analysis branch :255-261), ``_slice_forcing_time`` (:264-361, analysis branch :330-359), ``_build_item_dataarrays``
(:378-465) and ``__getitem__`` (:467-533).

Parity status: xarray is not installed in this image, so the reference class itself cannot be imported here.  The
restatement is PINNED against the known-answer vectors the reference's own tests hold for this path --
``tests/test_time_slicing.py:86-160`` (sample 0 for ar_steps = 3, past = 0..3: init / target / windowed forcing values) and
``tests/test_datasets.py:259-296`` (dataset length for six window configurations) and ``:299-321`` (IndexError / negative
indices) -- in tests/test_data.py.  The stacking order of several forcing features (feature-major, window-minor:
``stack(forcing_feature_windowed=("forcing_feature", "window"))``, :443-445) is not covered by those single-feature vectors;
it is pinned indirectly by the reference's tiling of the forcing statistics (``repeat_interleave(window)``,
models/module.py:352-358), which only standardises correctly for that order.
"""
import numpy as np

INIT_STEPS = 2   # weather_dataset.py:231, :297


def window_span(ar_steps, num_past_forcing_steps, num_future_forcing_steps):
    """Time steps one sample spans (weather_dataset.py:184-188)."""
    return max(INIT_STEPS, num_past_forcing_steps) + ar_steps + num_future_forcing_steps


def dataset_len(n_state_times, n_forcing_times, ar_steps, num_past_forcing_steps, num_future_forcing_steps):
    """``len(WeatherDataset)`` for analysis data (:179-194); ``n_forcing_times`` None = no forcing."""
    window = window_span(ar_steps, num_past_forcing_steps, num_future_forcing_steps)
    n = n_state_times - window + 1
    if n_forcing_times is not None:
        n = min(n, n_forcing_times - window + 1)
    return max(0, n)


def build_item(state, forcing, times, idx, ar_steps, num_past_forcing_steps, num_future_forcing_steps):
    """One sample.  ``state`` (T, N, d_state), ``forcing`` (T, N, d_forcing) or None, ``times`` (T,) int64 or None.

    Returns (init_states (2, N, d_state), target_states (ar_steps, N, d_state),
             forcing (ar_steps, N, d_forcing * window), target_times (ar_steps,) or None)."""
    n = dataset_len(state.shape[0], None if forcing is None else forcing.shape[0], ar_steps, num_past_forcing_steps,
                    num_future_forcing_steps)
    if idx < 0:            # :497-503
        idx += n
    if not 0 <= idx < n:
        raise IndexError(f"index {idx} out of range for WeatherDataset of length {n}")
    past, fut = num_past_forcing_steps, num_future_forcing_steps
    start = idx + max(0, past - INIT_STEPS)                       # :255-258
    end = idx + max(INIT_STEPS, past) + ar_steps
    sl = state[start:end]
    init_states, target_states = sl[:2], sl[2:]                   # :436-437
    offset = idx + max(INIT_STEPS, past)                          # :333
    target_times = None if times is None else np.asarray(times)[offset : offset + ar_steps]
    if forcing is None:                                           # :447-460
        return init_states, target_states, np.empty((ar_steps, state.shape[1], 0), dtype=state.dtype), target_times
    steps = []
    for step in range(ar_steps):                                  # :334-356
        w = forcing[offset + step - past : offset + step + fut + 1]   # (window, N, d_forcing)
        # (window, N, f) -> (N, f, window) -> (N, f * window): feature-major, window-minor (:443-445)
        steps.append(np.transpose(w, (1, 2, 0)).reshape(w.shape[1], -1))
    return init_states, target_states, np.stack(steps), target_times


def standardize_item(init_states, target_states, forcing, state_mean, state_std, forcing_mean, forcing_std, window):
    """``ForecasterModule.on_after_batch_transfer`` (models/module.py:326-367) on one sample."""
    init_states = (init_states - state_mean) / state_std
    target_states = (target_states - state_mean) / state_std
    if forcing.shape[-1] > 0:
        forcing = (forcing - np.repeat(forcing_mean, window)) / np.repeat(forcing_std, window)   # repeat_interleave, :352-358
    return init_states, target_states, forcing
