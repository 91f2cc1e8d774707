"""The data path (weather_dataset.py:467-533): training samples cut from time series that are
resident in HBM (neural_lam_amd.data.DeviceWeatherDataset -> nlam_window_batch), against the numpy oracle
(oracle/data.py), which is itself pinned here against the known-answer vectors of the reference's own tests.
"""
import numpy as np
import pytest
import torch

from oracle import data as od


# ---- the oracle against the reference's known-answer vectors ----
@pytest.mark.parametrize("past", [0, 1, 2, 3])
def test_oracle_matches_reference_time_slicing_vectors(past):
    """tests/test_time_slicing.py:86-160 of the reference (state 0..9, forcing 10..19, ar_steps 3, future 0)."""
    state = np.arange(10, dtype=np.float32).reshape(10, 1, 1)
    forcing = np.arange(10, 20, dtype=np.float32).reshape(10, 1, 1)
    init, target, frc, _ = od.build_item(state, forcing, None, 0, 3, past, 0)
    exp_init, exp_target = [0, 1], [2, 3, 4]
    exp_forcing = {0: [[12], [13], [14]], 1: [[11, 12], [12, 13], [13, 14]], 2: [[10, 11, 12], [11, 12, 13], [12, 13, 14]],
                   3: [[10, 11, 12, 13], [11, 12, 13, 14], [12, 13, 14, 15]]}[past]
    if past == 3:
        exp_init, exp_target = [1, 2], [3, 4, 5]
    assert init.shape == (2, 1, 1) and init[:, 0, 0].tolist() == exp_init
    assert target.shape == (3, 1, 1) and target[:, 0, 0].tolist() == exp_target
    assert frc.shape == (3, 1, 1 + past)
    np.testing.assert_equal(frc[:, 0, :], np.array(exp_forcing, dtype=np.float32))


@pytest.mark.parametrize("past,future,ar_steps,reduction", [(0, 0, 1, 2), (2, 0, 1, 2), (0, 2, 1, 4), (4, 0, 1, 4), (0, 0, 5, 6), (3, 3, 2, 7)])
def test_oracle_matches_reference_dataset_lengths(past, future, ar_steps, reduction):
    """tests/test_datasets.py:259-296 of the reference (10 time steps); first and last sample can be built."""
    n = od.dataset_len(10, 10, ar_steps, past, future)
    assert n == 10 - reduction
    state = np.random.default_rng(0).normal(size=(10, 3, 2)).astype(np.float32)
    forcing = np.random.default_rng(1).normal(size=(10, 3, 2)).astype(np.float32)
    for idx in (0, n - 1):
        init, target, frc, _ = od.build_item(state, forcing, None, idx, ar_steps, past, future)
        assert init.shape == (2, 3, 2) and target.shape == (ar_steps, 3, 2) and frc.shape == (ar_steps, 3, 2 * (past + future + 1))


def test_oracle_index_errors_and_negative_indices():
    """tests/test_datasets.py:299-321 of the reference."""
    state = np.arange(10, dtype=np.float32).reshape(10, 1, 1)
    n = od.dataset_len(10, None, 1, 0, 0)
    with pytest.raises(IndexError):
        od.build_item(state, None, None, n, 1, 0, 0)
    with pytest.raises(IndexError):
        od.build_item(state, None, None, -n - 1, 1, 0, 0)
    a = od.build_item(state, None, None, -1, 1, 0, 0)
    b = od.build_item(state, None, None, n - 1, 1, 0, 0)
    assert all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3])) and a[2].shape == (1, 1, 0)


def test_oracle_window_order_is_the_one_the_statistics_are_tiled_for():
    """feature-major, window-minor (weather_dataset.py:443-445) is the order models/module.py:352-358 standardises."""
    rng = np.random.default_rng(2)
    T, N, F, past, fut = 12, 4, 3, 1, 2
    W = past + fut + 1
    mean, std = rng.normal(size=F).astype(np.float32), (0.5 + rng.random(F)).astype(np.float32)
    forcing = (rng.normal(size=(T, N, F)) * std + mean).astype(np.float32)
    state = rng.normal(size=(T, N, 2)).astype(np.float32)
    _, _, frc, _ = od.build_item(state, forcing, None, 1, 2, past, fut)
    assert frc.shape == (2, N, F * W)
    off = 1 + max(2, past)
    for f in range(F):
        for w in range(W):
            np.testing.assert_array_equal(frc[0, :, f * W + w], forcing[off - past + w, :, f])
    _, _, frc_std = od.standardize_item(state[:2], state[2:4], frc, np.zeros(2, np.float32), np.ones(2, np.float32), mean, std, W)
    # every window copy of feature f is standardised with feature f's statistics
    for f in range(F):
        np.testing.assert_allclose(frc_std[0, :, f * W + 1], (forcing[off - past + 1, :, f] - mean[f]) / std[f], rtol=1e-6)


# ---- the C-ABI on the CPU: length query and argument validation (decided before any launch) ----
@pytest.mark.parametrize("past,future,ar_steps,reduction", [(0, 0, 1, 2), (2, 0, 1, 2), (0, 2, 1, 4), (4, 0, 1, 4), (0, 0, 5, 6), (3, 3, 2, 7)])
def test_abi_window_len_matches_reference_lengths(past, future, ar_steps, reduction):
    from neural_lam_amd import _lib as L

    lib = L.load()
    assert lib.nlam_window_len(10, 10, ar_steps, past, future) == 10 - reduction == od.dataset_len(10, 10, ar_steps, past, future)
    assert lib.nlam_window_len(10, -1, ar_steps, past, future) == od.dataset_len(10, None, ar_steps, past, future)
    assert lib.nlam_window_len(10, 8, ar_steps, past, future) == od.dataset_len(10, 8, ar_steps, past, future)
    assert lib.nlam_window_len(1, -1, ar_steps, past, future) == 0


def test_abi_window_batch_rejects_bad_arguments_before_launching():
    import ctypes as C

    from neural_lam_amd import _lib as L

    lib = L.load()
    EINVAL = -1   # include/nlam_hip.h: NLAM_EINVAL
    assert lib.nlam_window_batch(None, None) == EINVAL
    p = L.Window()
    assert lib.nlam_window_batch(C.byref(p), None) == EINVAL          # no pointers at all
    buf = (C.c_float * 4)()
    a = C.cast(buf, C.c_void_p)
    p.state = p.sample_idx = p.init_states = p.target_states = a
    p.n_times, p.nodes, p.d_state, p.batch, p.ar_steps = 10, 1, 1, 0, 3
    assert lib.nlam_window_batch(C.byref(p), None) == 0                      # empty batch: nothing to do
    p.d_forcing = 2
    assert lib.nlam_window_batch(C.byref(p), None) == EINVAL          # forcing width without forcing pointers
    p.d_forcing, p.n_times = 0, 4
    assert lib.nlam_window_batch(C.byref(p), None) == EINVAL          # series shorter than one sample (2 + 3 steps)
    p.n_times, p.state_mean = 10, a
    assert lib.nlam_window_batch(C.byref(p), None) == EINVAL          # mean without std


def test_device_dataset_needs_a_gpu():
    from neural_lam_amd.data import DeviceWeatherDataset

    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    with pytest.raises(RuntimeError, match="needs a GPU"):
        DeviceWeatherDataset(np.zeros((10, 2, 1), np.float32), device="cpu")


# ---- the HIP path against the oracle ----
def _series(T, N, ds, df, seed):
    rng = np.random.default_rng(seed)
    state = rng.normal(size=(T, N, ds)).astype(np.float32)
    forcing = rng.normal(size=(T, N, df)).astype(np.float32) if df else None
    times = (np.datetime64("2020-01-01T00", "ns").astype(np.int64) + np.arange(T, dtype=np.int64) * 3 * 3600 * 10**9)
    return state, forcing, times


@pytest.mark.gpu
@pytest.mark.parametrize("past", [0, 1, 2, 3])
def test_hip_dataset_reproduces_reference_time_slicing_vectors(past):
    """The reference's own known-answer test (tests/test_time_slicing.py:86-160), through the HIP kernel."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from neural_lam_amd.data import DeviceWeatherDataset

    ds = DeviceWeatherDataset(np.arange(10, dtype=np.float32).reshape(10, 1, 1), np.arange(10, 20, dtype=np.float32).reshape(10, 1, 1),
                              ar_steps=3, num_past_forcing_steps=past, num_future_forcing_steps=0)
    init, target, frc, _ = [t.cpu().numpy() for t in ds[0]]
    exp_forcing = {0: [[12], [13], [14]], 1: [[11, 12], [12, 13], [13, 14]], 2: [[10, 11, 12], [11, 12, 13], [12, 13, 14]],
                   3: [[10, 11, 12, 13], [11, 12, 13, 14], [12, 13, 14, 15]]}[past]
    assert init[:, 0, 0].tolist() == ([1, 2] if past == 3 else [0, 1])
    assert target[:, 0, 0].tolist() == ([3, 4, 5] if past == 3 else [2, 3, 4])
    np.testing.assert_equal(frc[:, 0, :], np.array(exp_forcing, dtype=np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("T,N,dst,df,ar,past,fut", [(12, 37, 5, 2, 3, 1, 1), (9, 1, 1, 1, 1, 0, 0), (20, 130, 17, 5, 4, 2, 1),
                                                    (15, 64, 3, 0, 2, 1, 1), (11, 5, 2, 3, 1, 4, 2), (40, 2049, 7, 3, 8, 0, 3)])
def test_hip_dataset_matches_oracle_every_sample(T, N, dst, df, ar, past, fut):
    """Every sample of the dataset, raw: bit-exact (a gather).  Lengths, negative indices, IndexError as the reference."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from neural_lam_amd.data import DeviceWeatherDataset

    state, forcing, times = _series(T, N, dst, df, seed=T + N)
    ds = DeviceWeatherDataset(state, forcing, times, ar_steps=ar, num_past_forcing_steps=past, num_future_forcing_steps=fut)
    n = od.dataset_len(T, None if forcing is None else T, ar, past, fut)
    assert len(ds) == n and n > 0
    got = ds.batch(list(range(n)))
    for i in range(n):
        ref = od.build_item(state, forcing, times, i, ar, past, fut)
        for g, r in zip(got, ref):
            assert np.array_equal(g[i].cpu().numpy(), r), (i,)
    last = ds[-1]
    ref = od.build_item(state, forcing, times, n - 1, ar, past, fut)
    assert all(np.array_equal(g.cpu().numpy(), r) for g, r in zip(last, ref))
    assert last[2].shape == (ar, N, df * (past + fut + 1))
    with pytest.raises(IndexError):
        ds[n]
    with pytest.raises(IndexError):
        ds[-n - 1]
    with pytest.raises(IndexError):
        ds.batch([0, n])


@pytest.mark.gpu
def test_hip_dataset_fused_standardization_equals_reference_formula():
    """batch(standardize=True) == on_after_batch_transfer (models/module.py:326-367) applied to the raw batch: IEEE
    subtraction and division, forcing statistics tiled feature-major over the window -> bit-equal to numpy fp32."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from neural_lam_amd.data import DeviceWeatherDataset

    T, N, dst, df, ar, past, fut = 16, 301, 6, 4, 3, 2, 1
    state, forcing, times = _series(T, N, dst, df, seed=3)
    rng = np.random.default_rng(4)
    stats = {"state_mean": rng.normal(size=dst).astype(np.float32), "state_std": (0.3 + rng.random(dst)).astype(np.float32),
             "forcing_mean": rng.normal(size=df).astype(np.float32), "forcing_std": (0.3 + rng.random(df)).astype(np.float32)}
    ds = DeviceWeatherDataset(state, forcing, times, ar_steps=ar, num_past_forcing_steps=past, num_future_forcing_steps=fut,
                              standardization=stats)
    perm = ds.epoch_permutation(seed=1)
    assert sorted(perm.cpu().tolist()) == list(range(len(ds)))
    idx = perm[:5]                                   # device indices: no host round trip
    init, target, frc, tt = ds.batch(idx, standardize=True)
    for k, i in enumerate(idx.cpu().tolist()):
        raw = od.build_item(state, forcing, times, i, ar, past, fut)
        ref = od.standardize_item(raw[0], raw[1], raw[2], stats["state_mean"], stats["state_std"], stats["forcing_mean"],
                                  stats["forcing_std"], past + fut + 1)
        assert np.array_equal(init[k].cpu().numpy(), ref[0]) and np.array_equal(target[k].cpu().numpy(), ref[1])
        assert np.array_equal(frc[k].cpu().numpy(), ref[2])
        assert np.array_equal(tt[k].cpu().numpy(), raw[3])
    # written into caller-owned buffers (the static inputs of a captured step)
    out = tuple(torch.zeros_like(t) for t in (init, target, frc, tt))
    ds.batch(idx, standardize=True, out=out)
    assert all(torch.equal(a, b) for a, b in zip(out, (init, target, frc, tt)))
    with pytest.raises(ValueError):
        ds.batch(idx, out=(init[:, :1], target, frc, tt))


@pytest.mark.gpu
def test_hip_dataset_full_size_round_trip_properties():
    """MEPS size (63 784 nodes, 17 + 5 variables): size-independent properties instead of the slow oracle -- consecutive samples
    overlap by one step, window slots are shifted copies of the series, the kernel is deterministic."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from neural_lam_amd.data import DeviceWeatherDataset

    T, N, dst, df, ar, past, fut = 24, 63784, 17, 5, 4, 1, 1
    g = torch.Generator(device="cuda").manual_seed(0)
    state = torch.randn(T, N, dst, device="cuda", generator=g)
    forcing = torch.randn(T, N, df, device="cuda", generator=g)
    ds = DeviceWeatherDataset(state, forcing, None, ar_steps=ar, num_past_forcing_steps=past, num_future_forcing_steps=fut)
    idx = torch.arange(len(ds), device="cuda")
    init, target, frc, tt = ds.batch(idx)
    init2, target2, frc2, _ = ds.batch(idx)
    assert torch.equal(init, init2) and torch.equal(target, target2) and torch.equal(frc, frc2)
    off = max(2, past)
    assert torch.equal(init[:, 0], state[: len(ds)]) and torch.equal(init[:, 1], state[1 : len(ds) + 1])
    for t in range(ar):
        assert torch.equal(target[:, t], state[off + t : off + t + len(ds)])
        assert torch.equal(tt[:, t], idx + off + t)
        W = past + fut + 1
        fr = frc[:, t].reshape(len(ds), N, df, W)
        for w in range(W):
            assert torch.equal(fr[..., w], forcing[off + t - past + w : off + t - past + w + len(ds)])
    assert torch.equal(target[:-1, 1], target[1:, 0])   # sample i's second target is sample i + 1's first


@pytest.mark.gpu
@pytest.mark.parametrize("use_graph", [False, True])
def test_training_from_the_device_dataset_equals_training_from_handed_over_batches(tmp_path, use_graph):
    """Trainer.step_from(dataset, indices) -- samples written by nlam_window_batch, on_after_batch_transfer folded in, straight
    into the captured step's input buffers -- against the reference-shaped loop (raw batch handed over, the module
    standardises): identical losses and parameters, step after step."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from neural_lam_amd import graph as G
    from neural_lam_amd import models as hm
    from neural_lam_amd.data import DeviceWeatherDataset
    from neural_lam_amd.datastore import SyntheticDatastore
    from neural_lam_amd.trainer import Trainer

    dev = torch.device("cuda:0")
    import numpy as np

    rng = np.random.default_rng(5)   # non-trivial statistics: a step that skips on_after_batch_transfer must show
    dstore = SyntheticDatastore(30, 27, 5, 2, 1, root_path=tmp_path, boundary="random", seed=1,
                                state_stats={"state_mean": rng.normal(size=5) * 2, "state_std": rng.uniform(0.5, 3.0, size=5)})
    dstore._forcing_stats.forcing_mean.values = rng.normal(size=2).astype(np.float32)
    dstore._forcing_stats.forcing_std.values = rng.uniform(0.5, 2.0, size=2).astype(np.float32)
    ext = dstore.get_xy_extent("state")
    graph = G.normalise_graph(G.create_regular_grid_graph(dstore.get_xy("state")), max(ext[1] - ext[0], ext[3] - ext[2]))

    def make(std_in_module):
        torch.manual_seed(3)
        fc = hm.ARForecaster(hm.GraphLAM(dstore, graph=graph, hidden_dim=16, processor_layers=2), dstore)
        return Trainer(hm.ForecasterStep(fc, dstore, standardize=std_in_module).to(dev), lr=1e-3, use_graph=use_graph)

    t_ref, t_dev, t_mod = make(True), make(False), make(True)
    N, T, past, fut = dstore.num_grid_points, 2, 1, 1
    state, forcing, times = _series(14, N, 5, 2, seed=9)
    data = DeviceWeatherDataset(state, forcing, times, ar_steps=T, num_past_forcing_steps=past, num_future_forcing_steps=fut,
                                standardization=t_dev.module.standardization_stats())
    perm = data.epoch_permutation(seed=2)
    B = 2
    for k in range(3):
        idx = perm[k * B : (k + 1) * B]
        raw = data.batch(idx)                                     # what the reference's DataLoader would hand over
        l_ref = float(t_ref.step(raw[0], raw[1], raw[2]))
        l_dev = float(t_dev.step_from(data, idx))
        assert l_ref == l_dev, (k, l_ref, l_dev)
        assert torch.equal(t_ref.fp.flat, t_dev.fp.flat)
        assert torch.equal(t_dev.batch_times, raw[3])
        # a module that standardises itself + raw samples from the dataset: with a captured step the standardisation is
        # hoisted out of the graph, and step_from must still apply it on every call (not only on the capturing one)
        l_mod = float(t_mod.step_from(data, idx, standardize=False))
        assert l_mod == l_ref, (k, l_ref, l_mod)
        assert torch.equal(t_ref.fp.flat, t_mod.fp.flat)
        assert torch.equal(t_mod.batch_times, raw[3])
    assert (t_dev._graph is not None) == use_graph
    with pytest.raises(ValueError):
        t_ref.step_from(data, perm[:B])                           # would standardise twice
