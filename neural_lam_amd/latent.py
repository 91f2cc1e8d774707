"""Latent encoder / decoder of the Graph-EFM family on the HIP layers (flat graph).

Mirror of the reference's ``neural_lam/models/latent/`` for the modules that are compositions of the hot-path
layers -- they instantiate their GNNs through ``get_gnn_class`` / ``utils.make_gnn_seq`` / ``utils.make_mlp``
(``graph_encoder.py:56-80``, ``graph_decoder.py:66-93``, ``base_decoder.py:55-70``), which is exactly the drop-in
surface of ``neural_lam_amd.gnn_layers``.  Same class names, constructor arguments, attribute / parameter names and
``forward`` contracts, so reference state dicts load with ``load_state_dict(strict=True)``.

Reference -> here:
  BaseLatentEncoder        models/latent/base_encoder.py:8-94
  ConstantLatentEncoder    models/latent/constant_encoder.py:10-62
  GraphLatentEncoder       models/latent/graph_encoder.py:11-105
  BaseGraphLatentDecoder   models/latent/base_decoder.py:9-171
  GraphLatentDecoder       models/latent/graph_decoder.py:11-134
  HiGraphLatentEncoder     models/latent/hi_graph_encoder.py:14-175
  HiGraphLatentDecoder     models/latent/hi_graph_decoder.py:17-270

The ``GraphEFM`` step predictor that drives them (``step_predictors/graph/graph_efm.py``: ELBO terms, prior /
variational sampling) is out of scope of the hot path; it composes these classes unchanged.
"""
from __future__ import annotations

import torch
from torch import distributions as tdists
from torch import nn

from . import _lib as L
from .gnn_layers import InteractionNet, PropagationNet, get_gnn_class, make_gnn_seq, make_mlp
from .ops import MlpGeometry


class BaseLatentEncoder(nn.Module):
    """base_encoder.py:8-94: raw parameters -> ``Normal`` over the latent variable on the mesh nodes."""

    def __init__(self, latent_dim, output_dist="isotropic"):
        super().__init__()
        self.output_dist = output_dist
        if output_dist == "isotropic":
            self.output_dim = latent_dim
        elif output_dist == "diagonal":
            self.output_dim = 2 * latent_dim
            self.latent_std_eps = 1e-4
        else:
            raise ValueError(f"Unknown encoder output distribution: {output_dist}")

    def compute_dist_params(self, grid_rep, *args, **kwargs):
        raise NotImplementedError("compute_dist_params not implemented")

    def forward(self, grid_rep, **kwargs):
        params = self.compute_dist_params(grid_rep, **kwargs)
        if self.output_dist == "diagonal":
            latent_mean, latent_std_raw = params.chunk(2, dim=-1)
            latent_std = self.latent_std_eps + nn.functional.softplus(latent_std_raw)
        else:
            latent_mean = params
            latent_std = torch.ones_like(latent_mean)
        return tdists.Normal(latent_mean, latent_std)


class ConstantLatentEncoder(BaseLatentEncoder):
    """constant_encoder.py:10-62: all-zero parameters (the input only supplies batch size and device)."""

    def __init__(self, latent_dim, num_mesh_nodes, output_dist="isotropic"):
        super().__init__(latent_dim, output_dist)
        self.num_mesh_nodes = num_mesh_nodes

    def compute_dist_params(self, grid_rep, **kwargs):
        return torch.zeros(grid_rep.shape[0], self.num_mesh_nodes, self.output_dim, device=grid_rep.device)


class GraphLatentEncoder(BaseLatentEncoder):
    """graph_encoder.py:11-105: g2m GNN -> stack of on-mesh InteractionNets -> latent parameter map."""

    def __init__(self, latent_dim, g2m_edge_index, m2m_edge_index, hidden_dim, m2m_layers, hidden_layers=1,
                 g2m_gnn_type="InteractionNet", output_dist="isotropic"):
        super().__init__(latent_dim, output_dist)
        self.g2m_gnn = get_gnn_class(g2m_gnn_type)(g2m_edge_index, hidden_dim, hidden_layers=hidden_layers, update_edges=False)
        self.m2m_gnns = make_gnn_seq(m2m_edge_index, m2m_layers, hidden_layers, hidden_dim) if m2m_layers > 0 else None
        self.latent_param_map = make_mlp([hidden_dim] * (hidden_layers + 1) + [self.output_dim], layer_norm=False)

    def compute_dist_params(self, grid_rep, graph_emb, **kwargs):
        mesh_rep = self.g2m_gnn(grid_rep, graph_emb["mesh"], graph_emb["g2m"])
        if self.m2m_gnns is not None:
            mesh_rep, _ = self.m2m_gnns(mesh_rep, graph_emb["m2m"], need_last_edges=False)   # the last edge update is discarded (:103)
        return self.latent_param_map(mesh_rep)


class BaseGraphLatentDecoder(nn.Module):
    """base_decoder.py:9-171: latent sample on the mesh + grid representation -> next-state increment (and std)."""

    def __init__(self, hidden_dim, latent_dim, num_state_vars, hidden_layers=1, output_std=True):
        super().__init__()
        self.grid_update_mlp = make_mlp([hidden_dim] * (hidden_layers + 2))
        self.latent_embedder = make_mlp([latent_dim] + [hidden_dim] * (hidden_layers + 1))
        self.output_std = output_std
        output_dim = 2 * num_state_vars if output_std else num_state_vars
        self.param_map = make_mlp([hidden_dim] * (hidden_layers + 1) + [output_dim], layer_norm=False)
        self._resid_geom = MlpGeometry(nsrc=1, flags=L.F_ADD_SRC0)   # grid_rep + grid_update_mlp(grid_rep) in one launch

    def combine_with_latent(self, original_grid_rep, latent_rep, residual_grid_rep, graph_emb):
        raise NotImplementedError("combine_with_latent not implemented")

    def forward(self, grid_rep, latent_samples, graph_emb):
        latent_emb = self.latent_embedder(latent_samples)
        if self.grid_update_mlp.fully_fused and grid_rep.is_cuda:
            residual_grid_rep, _ = self.grid_update_mlp.forward_fused(self._resid_geom, grid_rep)   # base_decoder.py:160
        else:
            residual_grid_rep = grid_rep + self.grid_update_mlp(grid_rep)
        combined = self.combine_with_latent(grid_rep, latent_emb, residual_grid_rep, graph_emb)
        state_params = self.param_map(combined)
        if self.output_std:
            mean_delta, std_raw = state_params.chunk(2, dim=-1)
            return mean_delta, nn.functional.softplus(std_raw)
        return state_params, None


class GraphLatentDecoder(BaseGraphLatentDecoder):
    """graph_decoder.py:11-134: g2m (latent as the initial mesh state) -> on-mesh stack -> m2g onto the residual grid rep."""

    def __init__(self, g2m_edge_index, m2m_edge_index, m2g_edge_index, hidden_dim, latent_dim, num_state_vars, m2m_layers,
                 hidden_layers=1, g2m_gnn_type="InteractionNet", m2g_gnn_type="InteractionNet", output_std=True):
        super().__init__(hidden_dim, latent_dim, num_state_vars, hidden_layers, output_std)
        self.g2m_gnn = get_gnn_class(g2m_gnn_type)(g2m_edge_index, hidden_dim, hidden_layers=hidden_layers, update_edges=False)
        self.m2m_gnns = make_gnn_seq(m2m_edge_index, m2m_layers, hidden_layers, hidden_dim) if m2m_layers > 0 else None
        self.m2g_gnn = get_gnn_class(m2g_gnn_type)(m2g_edge_index, hidden_dim, hidden_layers=hidden_layers, update_edges=False)

    def combine_with_latent(self, original_grid_rep, latent_rep, residual_grid_rep, graph_emb):
        mesh_rep = self.g2m_gnn(original_grid_rep, latent_rep, graph_emb["g2m"])
        if self.m2m_gnns is not None:
            mesh_rep, _ = self.m2m_gnns(mesh_rep, graph_emb["m2m"], need_last_edges=False)
        return self.m2g_gnn(mesh_rep, residual_grid_rep, graph_emb["m2g"])


def _need_levels(name, flat_name, m2m_edge_index):
    if len(m2m_edge_index) < 2:   # hi_graph_encoder.py:69-74 / hi_graph_decoder.py:88-93
        raise ValueError(f"{name} requires at least 2 mesh levels (got {len(m2m_edge_index)}). Use {flat_name} for flat graphs.")


class HiGraphLatentEncoder(BaseLatentEncoder):
    """hi_graph_encoder.py:14-175: grid -> bottom mesh level (g2m), then up the hierarchy through PropagationNets with
    optional intra-level stacks; the latent parameters are read out on the top level."""

    def __init__(self, latent_dim, g2m_edge_index, m2m_edge_index, mesh_up_edge_index, hidden_dim, intra_level_layers,
                 hidden_layers=1, g2m_gnn_type="InteractionNet", output_dist="isotropic"):
        super().__init__(latent_dim, output_dist)
        _need_levels("HiGraphLatentEncoder", "GraphLatentEncoder", m2m_edge_index)
        self.g2m_gnn = get_gnn_class(g2m_gnn_type)(g2m_edge_index, hidden_dim, hidden_layers=hidden_layers, update_edges=False)
        # always PropagationNets: an upward step must push information into nodes that start from their static embedding (:83-97)
        self.mesh_up_gnns = nn.ModuleList(
            [PropagationNet(ei, hidden_dim, hidden_layers=hidden_layers, update_edges=False) for ei in mesh_up_edge_index])
        self.intra_level_gnns = (
            nn.ModuleList([make_gnn_seq(ei, intra_level_layers, hidden_layers, hidden_dim) for ei in m2m_edge_index])
            if intra_level_layers > 0 else None)
        self.latent_param_map = make_mlp([hidden_dim] * (hidden_layers + 1) + [self.output_dim], layer_norm=False)

    def compute_dist_params(self, grid_rep, graph_emb, **kwargs):
        cur = self.g2m_gnn(grid_rep, graph_emb["mesh"][0], graph_emb["g2m"])
        if self.intra_level_gnns is not None:
            cur, _ = self.intra_level_gnns[0](cur, graph_emb["m2m"][0], need_last_edges=False)   # edge outputs are discarded (:151-153)
        for level, (up_gnn, up_rep, mesh_rep) in enumerate(zip(self.mesh_up_gnns, graph_emb["mesh_up"], graph_emb["mesh"][1:]), start=1):
            cur = up_gnn(cur, mesh_rep, up_rep)
            if self.intra_level_gnns is not None:
                cur, _ = self.intra_level_gnns[level](cur, graph_emb["m2m"][level], need_last_edges=False)
        return self.latent_param_map(cur)


class HiGraphLatentDecoder(BaseGraphLatentDecoder):
    """hi_graph_decoder.py:17-270: g2m onto the bottom level, up the hierarchy (InteractionNets; the latent is the
    receiver state of the last upward step), down again (PropagationNets onto the upward pass' level states, intra-level
    stacks on the upward pass' edge states), m2g onto the residual grid representation."""

    def __init__(self, g2m_edge_index, m2m_edge_index, m2g_edge_index, mesh_up_edge_index, mesh_down_edge_index, hidden_dim,
                 latent_dim, num_state_vars, intra_level_layers, hidden_layers=1, g2m_gnn_type="InteractionNet",
                 m2g_gnn_type="InteractionNet", output_std=True):
        super().__init__(hidden_dim, latent_dim, num_state_vars, hidden_layers, output_std)
        _need_levels("HiGraphLatentDecoder", "GraphLatentDecoder", m2m_edge_index)
        self.g2m_gnn = get_gnn_class(g2m_gnn_type)(g2m_edge_index, hidden_dim, hidden_layers=hidden_layers, update_edges=False)
        self.m2g_gnn = get_gnn_class(m2g_gnn_type)(m2g_edge_index, hidden_dim, hidden_layers=hidden_layers, update_edges=False)
        self.mesh_up_gnns = nn.ModuleList(
            [InteractionNet(ei, hidden_dim, hidden_layers=hidden_layers, update_edges=False) for ei in mesh_up_edge_index])
        self.mesh_down_gnns = nn.ModuleList(
            [PropagationNet(ei, hidden_dim, hidden_layers=hidden_layers, update_edges=False) for ei in mesh_down_edge_index])
        if intra_level_layers > 0:
            self.intra_up_gnns = nn.ModuleList([make_gnn_seq(ei, intra_level_layers, hidden_layers, hidden_dim) for ei in m2m_edge_index])
            self.intra_down_gnns = nn.ModuleList(
                [make_gnn_seq(ei, intra_level_layers, hidden_layers, hidden_dim) for ei in list(m2m_edge_index)[:-1]])
        else:
            self.intra_up_gnns = self.intra_down_gnns = None

    def combine_with_latent(self, original_grid_rep, latent_rep, residual_grid_rep, graph_emb):
        cur = self.g2m_gnn(original_grid_rep, graph_emb["mesh"][0], graph_emb["g2m"])
        mesh_level_reps, m2m_level_reps = [], []
        receivers = list(graph_emb["mesh"][1:-1]) + [latent_rep]
        for level, (up_gnn, up_rep, mesh_rep) in enumerate(zip(self.mesh_up_gnns, graph_emb["mesh_up"], receivers)):
            new_mesh, new_m2m = cur, None
            if self.intra_up_gnns is not None:   # these edge states feed the downward pass (:246-248): keep the last layer's
                new_mesh, new_m2m = self.intra_up_gnns[level](new_mesh, graph_emb["m2m"][level])
            mesh_level_reps.append(new_mesh)
            m2m_level_reps.append(new_m2m)
            cur = up_gnn(new_mesh, mesh_rep, up_rep)
        if self.intra_up_gnns is not None:
            cur, _ = self.intra_up_gnns[-1](cur, graph_emb["m2m"][-1], need_last_edges=False)
        for level in reversed(range(len(self.mesh_down_gnns))):
            cur = self.mesh_down_gnns[level](cur, mesh_level_reps[level], graph_emb["mesh_down"][level])
            if self.intra_down_gnns is not None:
                cur, _ = self.intra_down_gnns[level](cur, m2m_level_reps[level], need_last_edges=False)
        return self.m2g_gnn(cur, residual_grid_rep, graph_emb["m2g"])
