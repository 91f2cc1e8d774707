"""Step predictors, autoregressive rollout and training step on the HIP layers.

Host-side mirror of the reference's model files for the hot path only
(SURVEY.md §8a); every MLP / GNN layer below is a ``libnlam_hip.so`` kernel,
the surrounding glue (feature concat, rescale, boundary mix, loss) is stock
PyTorch-ROCm elementwise work.

Reference -> here (same class / argument / attribute / parameter names):
  StepPredictor      models/step_predictors/base.py:15-396
  BaseGraphModel     models/step_predictors/graph/base.py:31-344
  GraphLAM           models/step_predictors/graph/graph_lam.py:28-188
  BaseHiGraphModel   models/step_predictors/graph/hierarchical.py:30-292
  HiLAM              models/step_predictors/graph/hi_lam.py:25-376
  HiLAMParallel      models/step_predictors/graph/hi_lam_parallel.py:24-218
  ARForecaster       models/forecasters/autoregressive.py:14-149
  wmse               metrics.py:37-137
  ForecasterStep     the training_step / loss / AdamW lines of models/module.py:293-304, 326-417, 463-510

Differences that do not change results: the graph may be handed in pre-loaded
(``graph=``) instead of being read from ``datastore.root_path``; the
input-independent embeddings (g2m / m2g / m2m / mesh embedders) are computed
once per rollout instead of once per AR step (``static_cache``); the final
processor layer does not materialise the edge output the reference discards
(graph_lam.py:185).
"""
from __future__ import annotations

import contextlib

import torch
from torch import nn

from . import graph as G
from .gnn_layers import GNNSequential, InteractionNet, get_gnn_class, make_mlp


def inverse_softplus(x, beta=1.0, threshold=20.0):
    """utils/tensor.py:7-51."""
    x_clamped = torch.clamp(x, min=torch.log(torch.tensor(1e-6 + 1)) / beta, max=threshold / beta)
    non_linear = torch.log(torch.expm1(x_clamped * beta)) / beta
    return torch.where(x * beta <= threshold, non_linear, x)


def inverse_sigmoid(x):
    """utils/tensor.py:53-81."""
    xc = torch.clamp(x, min=1e-6, max=1 - 1e-6)
    return torch.log(xc / (1 - xc))


# nlam_affine_mix for the elementwise tail of a step (A/B on one box: +3 % forecast rate, +1 % training rate)
FUSED_STATE_UPDATE = True
# the torch.cat of the grid input features folded into grid_embedder (ops.CatMLPFunction); NLAM_FOLD_CAT=0 for A/B runs
import os as _os

FOLD_INPUT_CAT = _os.environ.get("NLAM_FOLD_CAT", "1") == "1"
# static-feature embedders (keys of static_embedding_specs: "mesh", "g2m", "m2g", "m2m", ..; "all") whose backward runs as soon as
# the gradient of their output is complete instead of in the grouped launch at the very end of backward (NLAM_EARLY_EMB; default none)
EARLY_EMBEDDER_BACKWARD = frozenset(k for k in _os.environ.get("NLAM_EARLY_EMB", "").split(",") if k)

class BufferList(nn.Module):
    """utils/buffer_list.py:11: list of non-persistent buffers."""

    def __init__(self, tensors, persistent=False):
        super().__init__()
        self.n_buffers = len(tensors)
        for i, t in enumerate(tensors):
            self.register_buffer(f"b{i}", t, persistent=persistent)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(self.n_buffers))]
        if k < 0:
            k += self.n_buffers
        if not 0 <= k < self.n_buffers:
            raise IndexError(f"index {k} out of range for BufferList of length {self.n_buffers}")
        return getattr(self, f"b{k}")

    def __len__(self):
        return self.n_buffers

    def __iter__(self):
        return (self[i] for i in range(self.n_buffers))


class StepPredictor(nn.Module):
    def __init__(self, datastore, output_std=False, output_clamping_lower=None, output_clamping_upper=None):
        super().__init__()
        self._output_clamping_lower = dict(output_clamping_lower) if output_clamping_lower else {}
        self._output_clamping_upper = dict(output_clamping_upper) if output_clamping_upper else {}
        num_state_vars = datastore.get_num_data_vars(category="state")
        da_static = datastore.get_dataarray(category="static", split=None, standardize=True)
        if da_static is None:
            static = torch.empty((datastore.num_grid_points, 0), dtype=torch.float32)
        else:
            static = torch.tensor(da_static.values, dtype=torch.float32)
        self.register_buffer("grid_static_features", static, persistent=False)
        st = datastore.get_standardization_dataarray(category="state")
        self.register_buffer("state_mean", torch.tensor(st.state_mean.values, dtype=torch.float32), persistent=False)
        self.register_buffer("state_std", torch.tensor(st.state_std.values, dtype=torch.float32), persistent=False)
        self.output_std = bool(output_std)
        self.grid_output_dim = 2 * num_state_vars if self.output_std else num_state_vars
        self.num_grid_nodes = self.grid_static_features.shape[0]

    @property
    def predicts_std(self) -> bool:
        return self.output_std

    def expand_to_batch(self, x, batch_size):
        return x.unsqueeze(0).expand(batch_size, -1, -1)

    def prepare_clamping_params(self, datastore):
        names = datastore.get_vars_names(category="state")
        lower, upper = self._output_clamping_lower, self._output_clamping_upper
        unknown = (set(lower) - set(names)) | (set(upper) - set(names))
        if unknown:
            raise ValueError(f"State feature limits were provided for unknown features: {unknown}")

        def norm(x, i):
            return (x - self.state_mean[i]) / self.state_std[i]

        lu_idx, lu_lo, lu_hi, lo_idx, lo_lim, hi_idx, hi_lim = [], [], [], [], [], [], []
        for i, f in enumerate(names):
            if f in lower and f in upper:
                assert lower[f] < upper[f], f'Invalid clamping limits for feature "{f}"'
                lu_idx.append(i)
                lu_lo.append(norm(lower[f], i))
                lu_hi.append(norm(upper[f], i))
            elif f in lower:
                lo_idx.append(i)
                lo_lim.append(norm(lower[f], i))
            elif f in upper:
                hi_idx.append(i)
                hi_lim.append(norm(upper[f], i))
        self.register_buffer("sigmoid_lower_lims", torch.tensor(lu_lo))
        self.register_buffer("sigmoid_upper_lims", torch.tensor(lu_hi))
        self.register_buffer("softplus_lower_lims", torch.tensor(lo_lim))
        self.register_buffer("softplus_upper_lims", torch.tensor(hi_lim))
        self.register_buffer("clamp_lower_upper_idx", torch.tensor(lu_idx))
        self.register_buffer("clamp_lower_idx", torch.tensor(lo_idx))
        self.register_buffer("clamp_upper_idx", torch.tensor(hi_idx))

    def get_clamped_new_state(self, state_delta, prev_state):
        sp = torch.nn.functional.softplus
        new_state = prev_state + state_delta
        if self.clamp_lower_upper_idx.numel() > 0:
            idx = self.clamp_lower_upper_idx
            lo, hi = self.sigmoid_lower_lims, self.sigmoid_upper_lims
            inv = inverse_sigmoid((prev_state[:, :, idx] - lo) / (hi - lo))
            new_state[:, :, idx] = lo + (hi - lo) * torch.sigmoid(inv + state_delta[:, :, idx])
        if self.clamp_lower_idx.numel() > 0:
            idx = self.clamp_lower_idx
            lo = self.softplus_lower_lims
            inv = inverse_softplus(prev_state[:, :, idx] - lo, beta=1)
            new_state[:, :, idx] = lo + sp(inv + state_delta[:, :, idx], beta=1)
        if self.clamp_upper_idx.numel() > 0:
            idx = self.clamp_upper_idx
            hi = self.softplus_upper_lims
            inv = -inverse_softplus(hi - prev_state[:, :, idx], beta=1)
            new_state[:, :, idx] = hi - sp(-(inv + state_delta[:, :, idx]), beta=1)
        return new_state


def compute_grid_input_dim(datastore, num_past_forcing_steps, num_future_forcing_steps):
    """utils/graph.py:470-512."""
    n_state = datastore.get_num_data_vars(category="state")
    n_forcing = datastore.get_num_data_vars(category="forcing")
    da_static = datastore.get_dataarray(category="static", split=None)
    n_static = 0 if da_static is None else datastore.get_num_data_vars(category="static")
    return 2 * n_state + n_static + n_forcing * (num_past_forcing_steps + num_future_forcing_steps + 1)


class BaseGraphModel(StepPredictor):
    def __init__(
        self,
        datastore,
        graph_name: str = "multiscale",
        hidden_dim: int = 64,
        hidden_layers: int = 1,
        processor_layers: int = 4,
        mesh_aggr: str = "sum",
        num_past_forcing_steps: int = 1,
        num_future_forcing_steps: int = 1,
        output_std: bool = False,
        output_clamping_lower=None,
        output_clamping_upper=None,
        g2m_gnn_type: str = "InteractionNet",
        m2g_gnn_type: str = "InteractionNet",
        graph=None,
    ):
        super().__init__(datastore, output_std, output_clamping_lower, output_clamping_upper)
        self.g2m_gnn_type, self.m2g_gnn_type = g2m_gnn_type, m2g_gnn_type
        st = datastore.get_standardization_dataarray("state")
        self.register_buffer(
            "diff_mean", torch.tensor(st.state_diff_mean_standardized.values, dtype=torch.float32), persistent=False
        )
        self.register_buffer(
            "diff_std", torch.tensor(st.state_diff_std_standardized.values, dtype=torch.float32), persistent=False
        )
        self.hidden_dim, self.hidden_layers = hidden_dim, hidden_layers
        self.processor_layers, self.mesh_aggr = processor_layers, mesh_aggr
        if graph is None:
            ext = datastore.get_xy_extent(category="state")
            span = max(ext[1] - ext[0], ext[3] - ext[2])  # graph/base.py:113-117
            graph = G.load_graph(datastore.root_path / "graph" / graph_name, span)
        self.hierarchical, tensors = graph
        for name, value in tensors.items():  # utils/graph.py:461-466: non-persistent buffers
            if torch.is_tensor(value):
                self.register_buffer(name, value.clone(), persistent=False)
            else:
                setattr(self, name, BufferList([v.clone() for v in value], persistent=False))
        self.num_mesh_nodes, _ = self.get_num_mesh()
        self.grid_input_dim = compute_grid_input_dim(datastore, num_past_forcing_steps, num_future_forcing_steps)
        self.g2m_edges, g2m_dim = self.g2m_features.shape
        self.m2g_edges, m2g_dim = self.m2g_features.shape
        self.mlp_blueprint_end = [hidden_dim] * (hidden_layers + 1)
        self.grid_embedder = make_mlp([self.grid_input_dim] + self.mlp_blueprint_end)
        self.g2m_embedder = make_mlp([g2m_dim] + self.mlp_blueprint_end)
        self.m2g_embedder = make_mlp([m2g_dim] + self.mlp_blueprint_end)
        self.g2m_gnn = get_gnn_class(g2m_gnn_type)(
            self.g2m_edge_index, hidden_dim, hidden_layers=hidden_layers, update_edges=False
        )
        self.encoding_grid_mlp = make_mlp([hidden_dim] + self.mlp_blueprint_end)
        self.m2g_gnn = get_gnn_class(m2g_gnn_type)(
            self.m2g_edge_index, hidden_dim, hidden_layers=hidden_layers, update_edges=False
        )
        self.output_map = make_mlp([hidden_dim] * (hidden_layers + 1) + [self.grid_output_dim], layer_norm=False)
        self.prepare_clamping_params(datastore)
        self._static = None  # cache of input-independent embeddings during a rollout
        from .ops import MlpGeometry
        from . import _lib as L

        self._enc_geom = MlpGeometry(nsrc=1, flags=L.F_ADD_SRC0)  # grid_emb + encoding_grid_mlp(grid_emb)

    # ---- input-independent embeddings: once per rollout instead of once per AR step ----
    def static_embedding_items(self):
        """(key, thunk) in the order the step needs them (encoder first, decoder last)."""
        return [
            ("mesh", self.embedd_mesh_nodes),
            ("g2m", lambda: self.g2m_embedder(self.g2m_features)),
            ("m2g", lambda: self.m2g_embedder(self.m2g_features)),
        ]

    def static_embedding_specs(self):
        """(key, mlp | [mlps], features | [features]) in the order the step needs them: the embedders of the static
        features are independent of each other and of the input, so they run as grouped launches."""
        return [
            ("mesh", *self.mesh_embedding_spec()),
            ("g2m", self.g2m_embedder, self.g2m_features),
            ("m2g", self.m2g_embedder, self.m2g_features),
        ]

    def compute_static_embeddings(self) -> dict:
        from .gnn_layers import grouped_mlp_forward

        specs = self.static_embedding_specs()
        pairs, slots = [], []
        for key, mlps, feats in specs:
            if isinstance(mlps, (list, tuple, nn.ModuleList)):
                for i, (m_, f_) in enumerate(zip(mlps, feats)):
                    pairs.append((m_, f_))
                    slots.append((key, i))
            else:
                pairs.append((mlps, feats))
                slots.append((key, None))
        from . import ops

        outs = grouped_mlp_forward(pairs)
        # embedders whose output gradient is complete long before the end of backward run their backward right then, on a side
        # stream (ops.early_backward_leaf).  Measured (round 6, profiles/round6/ab_early_emb.log): the mesh -> grid edge embedder
        # (64 % of the grouped backward's rows at MEPS size; its gradient is final behind the decoder's backward) leaves the tail of
        # the cfg2 step -- and the step does not move (1.7564 vs 1.7565 ms; cfg3 / cfg4 within noise): the chip is time-shared
        # either way.  Off by default (NLAM_EARLY_EMB=m2g, or "all", switches it on)
        early = EARLY_EMBEDDER_BACKWARD
        outs = [ops.early_backward_leaf(o, force=True) if (key in early or "all" in early) and o.requires_grad else o
                for (key, _i), o in zip(slots, outs)]
        st = {}
        for (key, i), o in zip(slots, outs):
            if i is None:
                st[key] = o
            else:
                st.setdefault(key, []).append(o)
        # the embeddings live for a whole rollout: their consumers' backward passes hand the gradient over in one buffer
        ops.rollout_shared_reset(outs)
        for key, mlps, _ in specs:   # empty lists (a one-level hierarchy has no up / down embedders)
            if isinstance(mlps, (list, tuple, nn.ModuleList)) and key not in st:
                st[key] = []
        return st

    @contextlib.contextmanager
    def static_cache(self):
        """Embeddings of the static graph features, computed once for a whole rollout.  (Computing them on side
        streams next to the first grid MLPs was measured and dropped: -1 % with one side stream, -3 % with one per
        embedder -- DESIGN.md, "measured and dropped".)"""
        self._static = self.compute_static_embeddings()
        try:
            yield
        finally:
            self._static = None

    def can_return_raw_delta(self) -> bool:
        """True when the step's tail is the plain ``prev + delta * diff_std + diff_mean`` (no clamping, no predicted std):
        the forecaster may then fuse it with the boundary overwrite and the loss (ops.StepTailFunction)."""
        no_clamp = self.clamp_lower_upper_idx.numel() + self.clamp_lower_idx.numel() + self.clamp_upper_idx.numel() == 0
        return no_clamp and not self.output_std and self.diff_std.dim() == 1

    def forward(self, prev_state, prev_prev_state, forcing, raw_delta: bool = False):
        B = prev_state.shape[0]
        feats = (prev_state, prev_prev_state, forcing, self.expand_to_batch(self.grid_static_features, B))
        st = self._static if self._static is not None else self.compute_static_embeddings()
        fused_ok = FUSED_STATE_UPDATE and prev_state.is_cuda and all(t.dtype == torch.float32 and t.dim() == 3 for t in feats)
        if fused_ok and FOLD_INPUT_CAT and self.grid_embedder.fully_fused and self.grid_input_dim <= 64 and self.grid_input_dim % 4 == 0:
            from .ops import CatMLPFunction

            # graph/base.py:275-286: the torch.cat of the input features folded into grid_embedder's first load (the (B, N, 56)
            # tensor is only written as a by-product for the backward pass; the static features are read un-expanded)
            grid_emb = CatMLPFunction.apply(*self.grid_embedder.params(), *feats)
        else:
            if fused_ok:
                from .ops import ConcatFunction

                grid_features = ConcatFunction.apply(*feats)   # graph/base.py:275-283 in one launch; the static features are read un-expanded
            else:
                grid_features = torch.cat(feats, dim=-1)
            grid_emb = self.grid_embedder(grid_features)  # (B, N_grid, d)
        mesh_rep = self.g2m_gnn(
            grid_emb, self.expand_to_batch(st["mesh"], B), self.expand_to_batch(st["g2m"], B)
        )
        # graph/base.py:308.  Kept after the g2m step as in the reference: issuing it first (it is independent) measured
        # 2 % slower at cfg2 -- its 16 MB output then sits cold through the whole processor before m2g reads it.
        if self.encoding_grid_mlp.fully_fused:
            grid_rep, _ = self.encoding_grid_mlp.forward_fused(self._enc_geom, grid_emb)
        else:
            grid_rep = grid_emb + self.encoding_grid_mlp(grid_emb)
        mesh_rep = self.process_step(mesh_rep, st)
        grid_rep = self.m2g_gnn(mesh_rep, grid_rep, self.expand_to_batch(st["m2g"], B))
        net_output = self.output_map(grid_rep)
        if raw_delta:
            return net_output, None
        if self.output_std:
            pred_delta_mean, pred_std_raw = net_output.chunk(2, dim=-1)
            pred_std = torch.nn.functional.softplus(pred_std_raw)
        else:
            pred_delta_mean, pred_std = net_output, None
        no_clamp = self.clamp_lower_upper_idx.numel() + self.clamp_lower_idx.numel() + self.clamp_upper_idx.numel() == 0
        if (FUSED_STATE_UPDATE and no_clamp and pred_delta_mean.is_cuda and self.diff_std.dim() == 1
                and pred_delta_mean.dtype == torch.float32 and prev_state.dtype == torch.float32):
            from .ops import AffineMixFunction

            # prev_state + (delta * diff_std + diff_mean) in one pass (three elementwise launches in the reference)
            new_state = AffineMixFunction.apply(None, None, prev_state, None, pred_delta_mean, self.diff_std, self.diff_mean)
            return new_state, pred_std
        rescaled = pred_delta_mean * self.diff_std + self.diff_mean
        return self.get_clamped_new_state(rescaled, prev_state), pred_std


class GraphLAM(BaseGraphModel):
    def __init__(self, datastore, graph_name="multiscale", **kw):
        super().__init__(datastore, graph_name, **kw)
        assert not self.hierarchical, "GraphLAM does not use a hierarchical mesh graph"
        mesh_dim = self.mesh_static_features.shape[1]
        _, m2m_dim = self.m2m_features.shape
        self.mesh_embedder = make_mlp([mesh_dim] + self.mlp_blueprint_end)
        self.m2m_embedder = make_mlp([m2m_dim] + self.mlp_blueprint_end)
        self.processor = GNNSequential(
            [
                InteractionNet(
                    self.m2m_edge_index, self.hidden_dim, hidden_layers=self.hidden_layers, aggr=self.mesh_aggr
                )
                for _ in range(self.processor_layers)
            ]
        )

    def get_num_mesh(self):
        return self.mesh_static_features.shape[0], 0

    def embedd_mesh_nodes(self):
        return self.mesh_embedder(self.mesh_static_features)

    def mesh_embedding_spec(self):
        return self.mesh_embedder, self.mesh_static_features

    def static_embedding_items(self):
        items = super().static_embedding_items()
        return items[:2] + [("m2m", lambda: self.m2m_embedder(self.m2m_features))] + items[2:]

    def static_embedding_specs(self):
        specs = super().static_embedding_specs()
        return specs[:2] + [("m2m", self.m2m_embedder, self.m2m_features)] + specs[2:]

    def process_step(self, mesh_rep, st=None):
        B = mesh_rep.shape[0]
        m2m_emb = st["m2m"] if st is not None else self.m2m_embedder(self.m2m_features)
        mesh_rep, _ = self.processor(mesh_rep, self.expand_to_batch(m2m_emb, B), need_last_edges=False)
        return mesh_rep


class BaseHiGraphModel(BaseGraphModel):
    def __init__(
        self, datastore, graph_name="multiscale", mesh_up_gnn_type="InteractionNet",
        mesh_down_gnn_type="InteractionNet", **kw,
    ):
        super().__init__(datastore, graph_name, **kw)
        self.mesh_up_gnn_type, self.mesh_down_gnn_type = mesh_up_gnn_type, mesh_down_gnn_type
        self.num_levels = len(self.mesh_static_features)
        self.level_mesh_sizes = [m.shape[0] for m in self.mesh_static_features]
        mesh_dim = self.mesh_static_features[0].shape[1]
        same_dim = self.m2m_features[0].shape[1]
        up_dim = self.mesh_up_features[0].shape[1]
        down_dim = self.mesh_down_features[0].shape[1]
        end, L_ = self.mlp_blueprint_end, self.num_levels
        self.mesh_embedders = nn.ModuleList([make_mlp([mesh_dim] + end) for _ in range(L_)])
        self.mesh_same_embedders = nn.ModuleList([make_mlp([same_dim] + end) for _ in range(L_)])
        self.mesh_up_embedders = nn.ModuleList([make_mlp([up_dim] + end) for _ in range(L_ - 1)])
        self.mesh_down_embedders = nn.ModuleList([make_mlp([down_dim] + end) for _ in range(L_ - 1)])
        up_cls, down_cls = get_gnn_class(mesh_up_gnn_type), get_gnn_class(mesh_down_gnn_type)
        self.mesh_init_gnns = nn.ModuleList(
            [up_cls(ei, self.hidden_dim, hidden_layers=self.hidden_layers) for ei in self.mesh_up_edge_index]
        )
        self.mesh_read_gnns = nn.ModuleList(
            [
                down_cls(ei, self.hidden_dim, hidden_layers=self.hidden_layers, update_edges=False)
                for ei in self.mesh_down_edge_index
            ]
        )

    def get_num_mesh(self):
        n = sum(m.shape[0] for m in self.mesh_static_features)
        return n, n - self.mesh_static_features[0].shape[0]

    def embedd_mesh_nodes(self):
        return self.mesh_embedders[0](self.mesh_static_features[0])

    def mesh_embedding_spec(self):
        return self.mesh_embedders[0], self.mesh_static_features[0]

    def static_embedding_specs(self):
        specs = super().static_embedding_specs()
        mine = [
            ("levels", list(self.mesh_embedders)[1:], list(self.mesh_static_features[1:])),
            ("up", list(self.mesh_up_embedders), list(self.mesh_up_features)),
            ("same", list(self.mesh_same_embedders), list(self.m2m_features)),
            ("down", list(self.mesh_down_embedders), list(self.mesh_down_features)),
        ]
        return specs[:2] + mine + specs[2:]

    def static_embedding_items(self):
        items = super().static_embedding_items()
        mine = [
            ("levels", lambda: [emb(f) for emb, f in zip(list(self.mesh_embedders)[1:], self.mesh_static_features[1:])]),
            ("up", lambda: [emb(f) for emb, f in zip(self.mesh_up_embedders, self.mesh_up_features)]),
            ("same", lambda: [emb(f) for emb, f in zip(self.mesh_same_embedders, self.m2m_features)]),
            ("down", lambda: [emb(f) for emb, f in zip(self.mesh_down_embedders, self.mesh_down_features)]),
        ]
        return items[:2] + mine + items[2:]

    def process_step(self, mesh_rep, st=None):
        B = mesh_rep.shape[0]
        if st is None:
            st = self.compute_static_embeddings()
        ex = self.expand_to_batch
        levels = [mesh_rep] + [ex(e, B) for e in st["levels"]]
        same = [ex(e, B) for e in st["same"]]
        up = [ex(e, B) for e in st["up"]]
        down = [ex(e, B) for e in st["down"]]
        for l, gnn in enumerate(self.mesh_init_gnns, start=1):  # hierarchical.py:241-262
            levels[l], up[l - 1] = gnn(levels[l - 1], levels[l], up[l - 1])
        levels, _, _, down = self.hi_processor_step(levels, same, up, down)
        for l, gnn in zip(range(self.num_levels - 2, -1, -1), reversed(self.mesh_read_gnns)):  # :271-289
            levels[l] = gnn(levels[l + 1], levels[l], down[l])
        return levels[0]


class HiLAM(BaseHiGraphModel):
    def __init__(self, datastore, graph_name="multiscale", **kw):
        super().__init__(datastore, graph_name, **kw)
        P = self.processor_layers
        self.mesh_down_gnns = nn.ModuleList([self.make_down_gnns() for _ in range(P)])
        self.mesh_down_same_gnns = nn.ModuleList([self.make_same_gnns() for _ in range(P)])
        self.mesh_up_gnns = nn.ModuleList([self.make_up_gnns() for _ in range(P)])
        self.mesh_up_same_gnns = nn.ModuleList([self.make_same_gnns() for _ in range(P)])

    def make_same_gnns(self):
        return nn.ModuleList(
            [InteractionNet(ei, self.hidden_dim, hidden_layers=self.hidden_layers) for ei in self.m2m_edge_index]
        )

    def make_up_gnns(self):
        cls = get_gnn_class(self.mesh_up_gnn_type)
        return nn.ModuleList([cls(ei, self.hidden_dim, hidden_layers=self.hidden_layers) for ei in self.mesh_up_edge_index])

    def make_down_gnns(self):
        cls = get_gnn_class(self.mesh_down_gnn_type)
        return nn.ModuleList(
            [cls(ei, self.hidden_dim, hidden_layers=self.hidden_layers) for ei in self.mesh_down_edge_index]
        )

    def mesh_down_step(self, levels, same, down, down_gnns, same_gnns):  # hi_lam.py:167-236
        levels[-1], same[-1] = same_gnns[-1](levels[-1], levels[-1], same[-1])
        for l, dg, sg in zip(range(self.num_levels - 2, -1, -1), reversed(down_gnns), reversed(same_gnns[:-1])):
            new_node, down[l] = dg(levels[l + 1], levels[l], down[l])
            levels[l], same[l] = sg(new_node, new_node, same[l])
        return levels, same, down

    def mesh_up_step(self, levels, same, up, up_gnns, same_gnns):  # hi_lam.py:238-307
        levels[0], same[0] = same_gnns[0](levels[0], levels[0], same[0])
        for l, (ug, sg) in enumerate(zip(up_gnns, same_gnns[1:]), start=1):
            new_node, up[l - 1] = ug(levels[l - 1], levels[l], up[l - 1])
            levels[l], same[l] = sg(new_node, new_node, same[l])
        return levels, same, up

    def hi_processor_step(self, levels, same, up, down):  # hi_lam.py:309-376
        for dg, dsg, ug, usg in zip(
            self.mesh_down_gnns, self.mesh_down_same_gnns, self.mesh_up_gnns, self.mesh_up_same_gnns
        ):
            levels, same, down = self.mesh_down_step(levels, same, down, dg, dsg)
            levels, same, up = self.mesh_up_step(levels, same, up, ug, usg)
        return levels, same, up, down


class HiLAMParallel(BaseHiGraphModel):
    def __init__(self, datastore, graph_name="multiscale", **kw):
        super().__init__(datastore, graph_name, **kw)
        first = [0]
        for size in self.level_mesh_sizes[:-1]:
            first.append(first[-1] + size)
        off_m2m = [ei + o for ei, o in zip(self.m2m_edge_index, first)]
        off_up = [
            torch.stack((ei[0] + first[l], ei[1] + first[l + 1]), dim=0) for l, ei in enumerate(self.mesh_up_edge_index)
        ]
        off_down = [
            torch.stack((ei[0] + first[l + 1], ei[1] + first[l]), dim=0)
            for l, ei in enumerate(self.mesh_down_edge_index)
        ]
        total_list = off_m2m + off_up + off_down
        total = torch.cat(total_list, dim=1)
        self.edge_split_sections = [ei.shape[1] for ei in total_list]
        if self.processor_layers == 0:
            self.processor = lambda x, edge_attr: (x, edge_attr)
        else:
            self.processor = GNNSequential(
                [
                    InteractionNet(
                        total,
                        self.hidden_dim,
                        hidden_layers=self.hidden_layers,
                        edge_chunk_sizes=self.edge_split_sections,
                        aggr_chunk_sizes=self.level_mesh_sizes,
                    )
                    for _ in range(self.processor_layers)
                ]
            )

    def hi_processor_step(self, levels, same, up, down):  # hi_lam_parallel.py:145-218
        mesh_rep = torch.cat(levels, dim=1)
        edge_rep = torch.cat(same + up + down, dim=1)
        mesh_rep, edge_rep = self.processor(mesh_rep, edge_rep)
        levels = list(torch.split(mesh_rep, self.level_mesh_sizes, dim=1))
        sec = torch.split(edge_rep, self.edge_split_sections, dim=1)
        L_ = self.num_levels
        return levels, list(sec[:L_]), list(sec[L_ : L_ + (L_ - 1)]), list(sec[L_ + (L_ - 1) :])


MODELS = {"graph_lam": GraphLAM, "hi_lam": HiLAM, "hi_lam_parallel": HiLAMParallel}  # models/__init__.py:23-27


class ARForecaster(nn.Module):
    """models/forecasters/autoregressive.py:14-149."""

    def __init__(self, predictor, datastore):
        super().__init__()
        self.predictor = predictor
        bm = torch.tensor(datastore.boundary_mask.values, dtype=torch.float32).unsqueeze(0).unsqueeze(-1)
        self.register_buffer("boundary_mask", bm, persistent=False)
        self.register_buffer("interior_mask", 1.0 - self.boundary_mask, persistent=False)

    @property
    def predicts_std(self):
        return self.predictor.predicts_std

    def forward(self, init_states, forcing_features, boundary_states, loss_spec=None):
        """``loss_spec = (target_states, inv_var (F,), row_weight (N,), scale)``: also return the training loss
        ``scale * sum_t sum_n,f row_weight * inv_var * (pred - target)^2`` as a third value, with each step's state
        update + boundary overwrite + loss term fused into one pass (ops.StepTailFunction) when the predictor's tail is
        the plain rescale (no clamping / predicted std)."""
        prev_prev_state, prev_state = init_states[:, 0], init_states[:, 1]
        preds, stds = [], []
        cache = self.predictor.static_cache() if hasattr(self.predictor, "static_cache") else contextlib.nullcontext()
        fused_tail = (loss_spec is not None and FUSED_STATE_UPDATE and init_states.is_cuda and init_states.dtype == torch.float32
                      and boundary_states.dtype == torch.float32 and getattr(self.predictor, "can_return_raw_delta", lambda: False)())
        losses = []
        with cache:
            for i in range(forcing_features.shape[1]):
                if fused_tail:
                    from .ops import StepTailFunction

                    target, inv_var, row_weight, scale = loss_spec
                    delta, _ = self.predictor(prev_state, prev_prev_state, forcing_features[:, i], raw_delta=True)
                    new_state, loss_t = StepTailFunction.apply(
                        delta.float(), prev_state, boundary_states[:, i], target[:, i], self.predictor.diff_std, self.predictor.diff_mean,
                        self.boundary_mask.reshape(-1), inv_var, row_weight, scale)
                    preds.append(new_state)
                    losses.append(loss_t)
                    prev_prev_state, prev_state = prev_state, new_state
                    continue
                pred_state, pred_std = self.predictor(prev_state, prev_prev_state, forcing_features[:, i])
                if (FUSED_STATE_UPDATE and pred_state.is_cuda and pred_state.dtype == torch.float32
                        and boundary_states.dtype == torch.float32):
                    from .ops import AffineMixFunction

                    # autoregressive.py:128-131 in one pass
                    new_state = AffineMixFunction.apply(boundary_states[:, i], self.boundary_mask.reshape(-1), pred_state,
                                                        self.interior_mask.reshape(-1), None, None, None)
                else:
                    new_state = self.boundary_mask * boundary_states[:, i] + self.interior_mask * pred_state
                preds.append(new_state)
                if pred_std is not None:
                    stds.append(pred_std)
                prev_prev_state, prev_state = prev_state, new_state
        def _stack(xs):   # a one-step rollout needs no copy
            return xs[0].unsqueeze(1) if len(xs) == 1 else torch.stack(xs, dim=1)

        if loss_spec is not None:
            loss = None
            if fused_tail:
                loss = losses[0] if len(losses) == 1 else torch.stack(losses).sum()
            return _stack(preds), (_stack(stds) if stds else None), loss
        return _stack(preds), (_stack(stds) if stds else None)


def mask_and_reduce_metric(vals, mask, average_grid, sum_vars):
    """metrics.py:37-84.  ``mask`` may be the reference's boolean node mask or a precomputed
    int64 index of the selected nodes (same entries in the same order, but a static gather:
    no nonzero() -> no host synchronisation, so the step can live in a HIP graph)."""
    if mask is not None:
        vals = vals[..., mask, :] if mask.dtype == torch.bool else vals.index_select(-2, mask)
    if average_grid:
        vals = torch.mean(vals, dim=-2)
    if sum_vars:
        vals = torch.sum(vals, dim=-1)
    return vals


def wmse(pred, target, pred_std, mask=None, average_grid=True, sum_vars=True):
    """metrics.py:87-137."""
    entry = torch.nn.functional.mse_loss(pred, target, reduction="none") / (pred_std**2)
    return mask_and_reduce_metric(entry, mask, average_grid, sum_vars)


class ForecasterStep(nn.Module):
    """The training_step / loss lines of ``ForecasterModule`` (models/module.py)
    without Lightning: on-device standardisation (:326-367), rollout + masked
    ``wmse`` per step, mean over batch then over steps (:463-510, :412)."""

    def __init__(self, forecaster: ARForecaster, datastore, standardize: bool = False, state_feature_weights=None):
        """``standardize=True`` makes ``forward`` start with ``on_after_batch_transfer`` (the batch arrives
        un-standardised, as ``WeatherDataset`` hands it over).  ``state_feature_weights``: per-variable loss weights
        (``get_state_feature_weighting``, module.py:162-176 / loss_weighting.py); default = uniform ``1 / n``."""
        super().__init__()
        self.forecaster = forecaster
        self.standardize_inputs = bool(standardize)
        bm = torch.tensor(datastore.boundary_mask.values, dtype=torch.float32)
        self.register_buffer("interior_mask_bool", (1.0 - bm).to(torch.bool), persistent=False)
        self.register_buffer("interior_index", torch.nonzero(1.0 - bm > 0.5).reshape(-1), persistent=False)
        interior = (1.0 - bm > 0.5).to(torch.float32)
        self.register_buffer("interior_weight", interior / interior.sum().clamp(min=1.0), persistent=False)
        st = datastore.get_standardization_dataarray("state")
        eps = torch.finfo(torch.float32).eps
        if not forecaster.predicts_std:
            n = len(datastore.get_vars_names("state"))
            if state_feature_weights is None:
                w = torch.tensor([1.0 / n] * n, dtype=torch.float32)  # loss_weighting.py:60-79 (uniform)
            else:   # ManualStateFeatureWeighting: one weight per state variable, in variable order
                w = torch.as_tensor(state_feature_weights, dtype=torch.float32).reshape(-1)
                if w.numel() != n or bool((w <= 0).any()):
                    raise ValueError(f"state_feature_weights needs {n} positive entries, got {tuple(w.shape)}")
            diff_std = torch.tensor(st.state_diff_std_standardized.values, dtype=torch.float32)
            self.register_buffer("per_var_std", diff_std / torch.sqrt(w), persistent=False)
            self.register_buffer("inv_var", 1.0 / (self.per_var_std * self.per_var_std), persistent=False)   # 1 / std^2 of wmse
        else:
            self.per_var_std = None
            self.inv_var = None
        self.register_buffer("state_mean", torch.tensor(st.state_mean.values, dtype=torch.float32), persistent=False)
        self.register_buffer(
            "state_std", torch.clamp(torch.tensor(st.state_std.values, dtype=torch.float32), min=eps), persistent=False
        )
        if datastore.get_num_data_vars("forcing") > 0:
            fs = datastore.get_standardization_dataarray("forcing")
            self.register_buffer("forcing_mean", torch.tensor(fs.forcing_mean.values, dtype=torch.float32), persistent=False)
            self.register_buffer(
                "forcing_std", torch.clamp(torch.tensor(fs.forcing_std.values, dtype=torch.float32), min=eps),
                persistent=False,
            )
        else:
            self.forcing_mean = self.forcing_std = None

    def standardization_stats(self):
        """The statistics ``on_after_batch_transfer`` uses (module.py:159-215; std clamped to eps, :306-324), in the form
        ``neural_lam_amd.data.DeviceWeatherDataset(standardization=...)`` takes to fold the hook into its batch launch."""
        d = {"state_mean": self.state_mean, "state_std": self.state_std}
        if self.forcing_mean is not None:
            d.update(forcing_mean=self.forcing_mean, forcing_std=self.forcing_std)
        return d

    def standardize(self, init_states, target_states, forcing, out=None):
        """module.py:326-367: one launch for the three tensors (``nlam_standardize``).  ``out``: three preallocated tensors
        (the trainer hands over the static inputs of its captured step: no separate staging copy of the batch)."""
        from .ops import standardize as std_launch

        items = [(init_states, self.state_mean, self.state_std, 1), (target_states, self.state_mean, self.state_std, 1)]
        has_forcing = forcing.shape[-1] > 0 and self.forcing_mean is not None
        if has_forcing:
            window = forcing.shape[-1] // self.forcing_mean.shape[-1]
            items.append((forcing, self.forcing_mean, self.forcing_std, window))
        outs = std_launch(items, None if out is None else list(out[: len(items)]))
        if not has_forcing and out is not None:
            out[2].copy_(forcing)
        return outs[0], outs[1], (outs[2] if has_forcing else (forcing if out is None else out[2]))

    def forward(self, init_states, target_states, forcing, standardize: bool | None = None):
        if standardize is None:
            standardize = self.standardize_inputs
        if standardize:
            init_states, target_states, forcing = self.standardize(init_states, target_states, forcing)
        if (self.inv_var is not None and init_states.is_cuda and FUSED_STATE_UPDATE and isinstance(self.forecaster, ARForecaster)):
            # rollout with every step's state update + boundary overwrite + loss term in one pass (one more in backward)
            B, T = target_states.shape[0], target_states.shape[1]
            prediction, pred_std, loss = self.forecaster(init_states, forcing, target_states,
                                                         loss_spec=(target_states, self.inv_var, self.interior_weight, 1.0 / (B * T)))
            if loss is not None:
                return prediction, loss
        else:
            prediction, pred_std = self.forecaster(init_states, forcing, target_states)
        if pred_std is None and prediction.is_cuda:
            # fixed per-variable std: wmse + interior mask + the grid / batch / step means in one HBM-bound kernel pair
            from .ops import WmseLossFunction

            return prediction, WmseLossFunction.apply(prediction, target_states, self.inv_var, self.interior_weight)
        if pred_std is None:
            pred_std = self.per_var_std
        time_step_loss = torch.mean(wmse(prediction, target_states, pred_std, mask=self.interior_index), dim=0)
        return prediction, torch.mean(time_step_loss)
