"""Graph-EFM step predictors on the HIP layers.

Mirror of the reference's ``neural_lam/models/step_predictors/graph/graph_efm.py``: ``BaseGraphEFM`` (:25-471),
``GraphEFM`` (hierarchical mesh, :474-793) and ``GraphEFMMultiScale`` (flat mesh, :796-1037) -- same class names,
constructor arguments, attribute / parameter names (reference state dicts load with ``strict=True``) and the same
``forward(prev_state, prev_prev_state, forcing) -> (pred_mean, pred_std)`` contract: embed the grid input and the
static graph features, evaluate the prior over the latent variable on the mesh, draw one sample, decode it into a
state increment, rescale with the one-step difference statistics, add onto ``prev_state`` and clamp.

These classes are orchestration only: every MLP / GNN they own comes from ``gnn_layers`` (``make_mlp``, the latent
encoders / decoders of ``latent.py``), so the compute runs in the HIP library.  Two MI355X-side differences, both
invisible in the results:
  * the embedders of the static graph features are input-independent and run as ONE grouped launch
    (``gnn_layers.grouped_mlp_forward``), cached for a whole rollout through ``static_cache()`` (the reference recomputes
    them every AR step and notes the hoist as future work, :388-391);
  * ``forward`` takes an optional ``latent_noise`` (standard-normal, shape of the latent): the reparameterised sample is
    ``mean + std * noise`` -- what ``Normal.rsample`` computes -- so a test can hand over the noise the reference drew.

Like the reference (models/__init__.py:15) these models are not entries of ``MODELS``: there is no training objective
for them in the reference yet.
"""
from __future__ import annotations

import contextlib

import torch
from torch import nn

from . import graph as G
from .gnn_layers import grouped_mlp_forward, make_mlp
from .latent import ConstantLatentEncoder, GraphLatentDecoder, GraphLatentEncoder, HiGraphLatentDecoder, HiGraphLatentEncoder
from .models import BufferList, StepPredictor, compute_grid_input_dim


class BaseGraphEFM(StepPredictor):
    """graph_efm.py:25-471."""

    def __init__(self, datastore, graph_name, hidden_dim=64, hidden_layers=1, latent_dim=None, learn_prior=True,
                 prior_dist="isotropic", prior_layers=2, g2m_gnn_type="InteractionNet", num_past_forcing_steps=1,
                 num_future_forcing_steps=1, output_std=False, output_clamping_lower=None, output_clamping_upper=None, graph=None):
        super().__init__(datastore, output_std, output_clamping_lower, output_clamping_upper)
        st = datastore.get_standardization_dataarray("state")
        self.register_buffer("diff_mean", torch.tensor(st.state_diff_mean_standardized.values, dtype=torch.float32), persistent=False)
        self.register_buffer("diff_std", torch.tensor(st.state_diff_std_standardized.values, dtype=torch.float32), persistent=False)
        if graph is None:
            ext = datastore.get_xy_extent(category="state")
            graph = G.load_graph(datastore.root_path / "graph" / graph_name, max(ext[1] - ext[0], ext[3] - ext[2]))   # :103-112
        self.hierarchical, tensors = graph
        for name, value in tensors.items():   # utils/graph.py:461-466: non-persistent buffers / BufferLists
            if torch.is_tensor(value):
                self.register_buffer(name, value.clone(), persistent=False)
            else:
                setattr(self, name, BufferList([v.clone() for v in value], persistent=False))
        self.check_graph_type(graph_name)
        self.num_state_vars = datastore.get_num_data_vars(category="state")
        self.grid_dim = compute_grid_input_dim(datastore, num_past_forcing_steps, num_future_forcing_steps)
        grid_current_dim = self.grid_dim + self.num_state_vars
        self.mlp_blueprint_end = [hidden_dim] * (hidden_layers + 1)
        self.grid_prev_embedder = make_mlp([self.grid_dim] + self.mlp_blueprint_end)          # states up to t - 1
        self.grid_current_embedder = make_mlp([grid_current_dim] + self.mlp_blueprint_end)    # states including t
        self.g2m_embedder = make_mlp([self.g2m_features.shape[1]] + self.mlp_blueprint_end)
        self.m2g_embedder = make_mlp([self.m2g_features.shape[1]] + self.mlp_blueprint_end)
        self.prepare_clamping_params(datastore)
        self.latent_dim = latent_dim if latent_dim is not None else hidden_dim
        if learn_prior:
            self.prior_model = self.build_learnable_prior(latent_dim=self.latent_dim, hidden_dim=hidden_dim, hidden_layers=hidden_layers,
                                                          g2m_gnn_type=g2m_gnn_type, prior_dist=prior_dist, prior_layers=prior_layers)
        else:
            self.prior_model = ConstantLatentEncoder(latent_dim=self.latent_dim, num_mesh_nodes=self.latent_spatial_dim,
                                                     output_dist=prior_dist)
        self._static = None

    # ---- delegated to the subclass (:223-299) ----
    def check_graph_type(self, graph_name):
        raise NotImplementedError("check_graph_type not implemented")

    @property
    def latent_spatial_dim(self):
        raise NotImplementedError("latent_spatial_dim not implemented")

    def build_learnable_prior(self, latent_dim, hidden_dim, hidden_layers, g2m_gnn_type, prior_dist, prior_layers):
        raise NotImplementedError("build_learnable_prior not implemented")

    def mesh_embedding_pairs(self):
        """[(key, index | None, mlp, features)] of the subclass' static mesh embedders."""
        raise NotImplementedError("mesh_embedding_pairs not implemented")

    # ---- embeddings ----
    def compute_static_embeddings(self):
        """Embedded static graph features (un-batched), all embedders in one grouped launch (:393-402, embedd_mesh)."""
        items = [("g2m", None, self.g2m_embedder, self.g2m_features), ("m2g", None, self.m2g_embedder, self.m2g_features)]
        items += self.mesh_embedding_pairs()
        outs = grouped_mlp_forward([(m, f) for _, _, m, f in items])
        emb = {}
        for (key, idx, _, _), o in zip(items, outs):
            if idx is None:
                emb[key] = o
            else:
                emb.setdefault(key, []).append(o)
        return emb

    @contextlib.contextmanager
    def static_cache(self):
        """The static embeddings once per rollout (models.ARForecaster enters this around its AR loop)."""
        self._static = self.compute_static_embeddings()
        try:
            yield
        finally:
            self._static = None

    def embedd_mesh(self, batch_size):
        raise NotImplementedError("embedd_mesh not implemented")

    def _batched(self, emb, batch_size):
        return {k: ([self.expand_to_batch(x, batch_size) for x in v] if isinstance(v, list) else self.expand_to_batch(v, batch_size))
                for k, v in emb.items()}

    def embedd_grid_with_target(self, prev_state, prev_prev_state, forcing, current_state):
        """:301-343 (the encoder's input: grid features including the target state)."""
        B = prev_state.shape[0]
        feats = torch.cat((prev_prev_state, prev_state, forcing, self.expand_to_batch(self.grid_static_features, B), current_state), dim=-1)
        return self.grid_current_embedder(feats)

    def embedd_grid_and_graph(self, prev_state, prev_prev_state, forcing):
        """:364-413."""
        B = prev_state.shape[0]
        feats = torch.cat((prev_prev_state, prev_state, forcing, self.expand_to_batch(self.grid_static_features, B)), dim=-1)
        grid_emb = self.grid_prev_embedder(feats)
        emb = self._static if self._static is not None else self.compute_static_embeddings()
        graph_emb = self._batched(emb, B)
        graph_emb.setdefault("m2m", [])   # hierarchical model without intra-level layers (:785-786)
        return grid_emb, graph_emb

    def forward(self, prev_state, prev_prev_state, forcing, latent_noise=None):
        """:415-471."""
        grid_prev_emb, graph_emb = self.embedd_grid_and_graph(prev_state, prev_prev_state, forcing)
        prior_dist = self.prior_model(grid_prev_emb, graph_emb=graph_emb)
        if latent_noise is None:
            latent_samples = prior_dist.rsample()
        else:
            latent_samples = prior_dist.mean + prior_dist.stddev * latent_noise
        mean_delta, pred_std = self.decoder(grid_prev_emb, latent_samples, graph_emb)
        rescaled_mean_delta = mean_delta * self.diff_std + self.diff_mean
        return self.get_clamped_new_state(rescaled_mean_delta, prev_state), pred_std


class GraphEFM(BaseGraphEFM):
    """graph_efm.py:474-793: hierarchical mesh; HiGraphLatentEncoder prior / encoder, HiGraphLatentDecoder."""

    def __init__(self, datastore, graph_name="hierarchical", hidden_dim=64, hidden_layers=1, latent_dim=None,
                 prior_intra_level_layers=2, encoder_intra_level_layers=2, decoder_intra_level_layers=4, learn_prior=True,
                 prior_dist="isotropic", num_past_forcing_steps=1, num_future_forcing_steps=1, g2m_gnn_type="InteractionNet",
                 m2g_gnn_type="InteractionNet", output_std=False, output_clamping_lower=None, output_clamping_upper=None, graph=None):
        super().__init__(datastore, graph_name, hidden_dim=hidden_dim, hidden_layers=hidden_layers, latent_dim=latent_dim,
                         learn_prior=learn_prior, prior_dist=prior_dist, prior_layers=prior_intra_level_layers,
                         g2m_gnn_type=g2m_gnn_type, num_past_forcing_steps=num_past_forcing_steps,
                         num_future_forcing_steps=num_future_forcing_steps, output_std=output_std,
                         output_clamping_lower=output_clamping_lower, output_clamping_upper=output_clamping_upper, graph=graph)
        num_levels = len(self.mesh_static_features)
        mk = lambda dim: make_mlp([dim] + self.mlp_blueprint_end)  # noqa: E731
        self.mesh_embedders = nn.ModuleList([mk(self.mesh_static_features[0].shape[1]) for _ in range(num_levels)])
        self.mesh_up_embedders = nn.ModuleList([mk(self.mesh_up_features[0].shape[1]) for _ in range(num_levels - 1)])
        self.mesh_down_embedders = nn.ModuleList([mk(self.mesh_down_features[0].shape[1]) for _ in range(num_levels - 1)])
        # m2m edges are embedded only if some component has intra-level layers (:598-613)
        self.embedd_m2m = max(prior_intra_level_layers, encoder_intra_level_layers, decoder_intra_level_layers) > 0
        if self.embedd_m2m:
            self.m2m_embedders = nn.ModuleList([mk(self.m2m_features[0].shape[1]) for _ in range(num_levels)])
        self.encoder = HiGraphLatentEncoder(self.latent_dim, self.g2m_edge_index, list(self.m2m_edge_index), list(self.mesh_up_edge_index),
                                            hidden_dim, encoder_intra_level_layers, hidden_layers=hidden_layers,
                                            g2m_gnn_type=g2m_gnn_type, output_dist="diagonal")
        self.decoder = HiGraphLatentDecoder(self.g2m_edge_index, list(self.m2m_edge_index), self.m2g_edge_index,
                                            list(self.mesh_up_edge_index), list(self.mesh_down_edge_index), hidden_dim, self.latent_dim,
                                            self.num_state_vars, decoder_intra_level_layers, hidden_layers=hidden_layers,
                                            g2m_gnn_type=g2m_gnn_type, m2g_gnn_type=m2g_gnn_type, output_std=bool(output_std))

    def check_graph_type(self, graph_name):
        if not self.hierarchical:
            raise ValueError(f"{type(self).__name__} requires a hierarchical mesh graph, but graph '{graph_name}' is flat")

    @property
    def latent_spatial_dim(self):
        return self.mesh_static_features[-1].shape[0]

    def build_learnable_prior(self, latent_dim, hidden_dim, hidden_layers, g2m_gnn_type, prior_dist, prior_layers):
        return HiGraphLatentEncoder(latent_dim, self.g2m_edge_index, list(self.m2m_edge_index), list(self.mesh_up_edge_index), hidden_dim,
                                    prior_layers, hidden_layers=hidden_layers, g2m_gnn_type=g2m_gnn_type, output_dist=prior_dist)

    def mesh_embedding_pairs(self):
        items = [("mesh", i, m, f) for i, (m, f) in enumerate(zip(self.mesh_embedders, self.mesh_static_features))]
        items += [("mesh_up", i, m, f) for i, (m, f) in enumerate(zip(self.mesh_up_embedders, self.mesh_up_features))]
        items += [("mesh_down", i, m, f) for i, (m, f) in enumerate(zip(self.mesh_down_embedders, self.mesh_down_features))]
        if self.embedd_m2m:
            items += [("m2m", i, m, f) for i, (m, f) in enumerate(zip(self.m2m_embedders, self.m2m_features))]
        return items

    def embedd_mesh(self, batch_size):
        """:746-793 (the mesh part of the graph embedding, batched)."""
        emb = self._static if self._static is not None else self.compute_static_embeddings()
        out = self._batched({k: emb.get(k, []) for k in ("mesh", "mesh_up", "mesh_down", "m2m")}, batch_size)
        return out


class GraphEFMMultiScale(BaseGraphEFM):
    """graph_efm.py:796-1037: flat (multiscale) mesh; GraphLatentEncoder prior / encoder, GraphLatentDecoder."""

    def __init__(self, datastore, graph_name="multiscale", hidden_dim=64, hidden_layers=1, latent_dim=None, prior_m2m_layers=2,
                 encoder_m2m_layers=2, decoder_m2m_layers=4, learn_prior=True, prior_dist="isotropic", num_past_forcing_steps=1,
                 num_future_forcing_steps=1, g2m_gnn_type="InteractionNet", m2g_gnn_type="InteractionNet", output_std=False,
                 output_clamping_lower=None, output_clamping_upper=None, graph=None):
        super().__init__(datastore, graph_name, hidden_dim=hidden_dim, hidden_layers=hidden_layers, latent_dim=latent_dim,
                         learn_prior=learn_prior, prior_dist=prior_dist, prior_layers=prior_m2m_layers, g2m_gnn_type=g2m_gnn_type,
                         num_past_forcing_steps=num_past_forcing_steps, num_future_forcing_steps=num_future_forcing_steps,
                         output_std=output_std, output_clamping_lower=output_clamping_lower,
                         output_clamping_upper=output_clamping_upper, graph=graph)
        self.mesh_embedder = make_mlp([self.mesh_static_features.shape[1]] + self.mlp_blueprint_end)
        self.m2m_embedder = make_mlp([self.m2m_features.shape[1]] + self.mlp_blueprint_end)
        self.encoder = GraphLatentEncoder(self.latent_dim, self.g2m_edge_index, self.m2m_edge_index, hidden_dim, encoder_m2m_layers,
                                          hidden_layers=hidden_layers, g2m_gnn_type=g2m_gnn_type, output_dist="diagonal")
        self.decoder = GraphLatentDecoder(self.g2m_edge_index, self.m2m_edge_index, self.m2g_edge_index, hidden_dim, self.latent_dim,
                                          self.num_state_vars, decoder_m2m_layers, hidden_layers=hidden_layers,
                                          g2m_gnn_type=g2m_gnn_type, m2g_gnn_type=m2g_gnn_type, output_std=bool(output_std))

    def check_graph_type(self, graph_name):
        if self.hierarchical:
            raise ValueError(f"{type(self).__name__} requires a flat mesh graph, but graph '{graph_name}' is hierarchical")

    @property
    def latent_spatial_dim(self):
        return len(self.mesh_static_features)

    def build_learnable_prior(self, latent_dim, hidden_dim, hidden_layers, g2m_gnn_type, prior_dist, prior_layers):
        return GraphLatentEncoder(latent_dim, self.g2m_edge_index, self.m2m_edge_index, hidden_dim, prior_layers,
                                  hidden_layers=hidden_layers, g2m_gnn_type=g2m_gnn_type, output_dist=prior_dist)

    def mesh_embedding_pairs(self):
        return [("mesh", None, self.mesh_embedder, self.mesh_static_features), ("m2m", None, self.m2m_embedder, self.m2m_features)]

    def embedd_mesh(self, batch_size):
        """:1015-1037."""
        emb = self._static if self._static is not None else self.compute_static_embeddings()
        return self._batched({"mesh": emb["mesh"], "m2m": emb["m2m"]}, batch_size)
