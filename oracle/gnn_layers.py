"""ORACLE (test infrastructure, never shipped, never measured as the product).

CPU restatement, in plain PyTorch fp32 and without torch_geometric, of the
reference's message-passing layers.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this module.

Follows, function by function:
  * ``make_mlp``        -> neural_lam/utils/networks.py:8-40
  * ``InteractionNet``  -> neural_lam/gnn_layers.py:14-189
  * ``PropagationNet``  -> neural_lam/gnn_layers.py:192-249
  * ``SplitMLPs``       -> neural_lam/gnn_layers.py:274-324
  * ``GNN_TYPES`` / ``get_gnn_class`` -> neural_lam/gnn_layers.py:252-271
  * ``Sequential``      -> the behaviour of ``pyg.nn.Sequential`` as used at
                           graph_lam.py:117-126 and utils/networks.py:93-106
and the PyG-2.3.1 ``MessagePassing.propagate`` / scatter semantics the reference
relies on (third-party dependency ``torch-geometric==2.3.1``, pyproject.toml:33,
absent from /root/reference; published algorithm restated in SURVEY.md App. B):
``x_j = x[edge_index[0]]``, ``x_i = x[edge_index[1]]`` on dim -2; sum = zeros
``index_add_``; mean = sum / clamp(count, 1).

Pinning: tests/golden/*.pt were produced by the reference's own
``gnn_layers.py`` / model files executed from /root/reference on top of a
minimal PyG stand-in (tests/golden/ref_harness.py, make_golden.py);
tests/test_oracle_golden.py checks this oracle against them.  Parameter names
and shapes are identical to the reference's, so state_dicts interchange.
"""
from __future__ import annotations

import torch
from torch import nn


def make_mlp(blueprint, layer_norm: bool = True) -> nn.Sequential:
    """[Linear -> SiLU] * hidden_layers -> Linear [-> LayerNorm] (utils/networks.py:8-40)."""
    hidden_layers = len(blueprint) - 2
    assert hidden_layers >= 0, "Invalid MLP blueprint"
    layers = []
    for layer_i, (d1, d2) in enumerate(zip(blueprint[:-1], blueprint[1:])):
        layers.append(nn.Linear(d1, d2))
        if layer_i != hidden_layers:
            layers.append(nn.SiLU())
    if layer_norm:
        layers.append(nn.LayerNorm(blueprint[-1]))
    return nn.Sequential(*layers)


class SplitMLPs(nn.Module):
    """Chunks of dim -2 through separate MLPs (gnn_layers.py:274-324)."""

    def __init__(self, mlps, chunk_sizes):
        super().__init__()
        assert len(mlps) == len(chunk_sizes), "Number of MLPs must match the number of chunks"
        self.mlps = nn.ModuleList(mlps)
        self.chunk_sizes = chunk_sizes

    def forward(self, x):
        chunks = torch.split(x, self.chunk_sizes, dim=-2)
        return torch.cat([mlp(c) for mlp, c in zip(self.mlps, chunks)], dim=-2)


class InteractionNet(nn.Module):
    """gnn_layers.py:14-189 with PyG's propagate/aggregate written out."""

    def __init__(
        self,
        edge_index,
        input_dim,
        update_edges=True,
        hidden_layers=1,
        hidden_dim=None,
        edge_chunk_sizes=None,
        aggr_chunk_sizes=None,
        aggr="sum",
    ):
        if aggr not in ("sum", "mean"):
            raise ValueError(f"Unknown aggregation method: {aggr}")  # gnn_layers.py:65-66
        super().__init__()
        self.aggr = aggr
        if hidden_dim is None:
            hidden_dim = input_dim
        self.num_rec = edge_index[1].max() + 1  # 0-dim tensor, gnn_layers.py:73
        # receivers -> [0, num_rec), senders -> [num_rec, num_rec + num_send)  (:82-84)
        edge_index = torch.stack((edge_index[0] + self.num_rec, edge_index[1]), dim=0)
        self.register_buffer("edge_index", edge_index, persistent=False)
        edge_recipe = [3 * input_dim] + [hidden_dim] * (hidden_layers + 1)
        aggr_recipe = [2 * input_dim] + [hidden_dim] * (hidden_layers + 1)
        if edge_chunk_sizes is None:
            self.edge_mlp = make_mlp(edge_recipe)
        else:
            self.edge_mlp = SplitMLPs([make_mlp(edge_recipe) for _ in edge_chunk_sizes], edge_chunk_sizes)
        if aggr_chunk_sizes is None:
            self.aggr_mlp = make_mlp(aggr_recipe)
        else:
            self.aggr_mlp = SplitMLPs([make_mlp(aggr_recipe) for _ in aggr_chunk_sizes], aggr_chunk_sizes)
        self.update_edges = update_edges

    # -- PyG MessagePassing.propagate (flow source_to_target, node_dim -2) --
    def propagate(self, edge_index, x, edge_attr):
        x_j = x.index_select(-2, edge_index[0])
        x_i = x.index_select(-2, edge_index[1])
        msgs = self.message(x_j, x_i, edge_attr)
        return self.aggregate(msgs, edge_index[1], None, x.shape[-2])

    def message(self, x_j, x_i, edge_attr):
        return self.edge_mlp(torch.cat((edge_attr, x_j, x_i), dim=-1))  # gnn_layers.py:172

    def aggregate(self, inputs, index, ptr, dim_size):
        # aggregate only onto the receivers (gnn_layers.py:188): PyG scatter
        num_rec = int(self.num_rec)
        size = list(inputs.shape)
        size[-2] = num_rec
        aggr = inputs.new_zeros(size).index_add_(-2, index, inputs)
        if self.aggr == "mean":
            count = inputs.new_zeros(num_rec).index_add_(0, index, inputs.new_ones(index.shape[0]))
            aggr = aggr / count.clamp(min=1).view(-1, 1)
        return aggr, inputs

    def node_residual_target(self, rec_rep, edge_rep_aggr):
        return rec_rep

    def forward(self, send_rep, rec_rep, edge_rep):
        node_reps = torch.cat((rec_rep, send_rep), dim=-2)  # :144
        edge_rep_aggr, edge_diff = self.propagate(self.edge_index, x=node_reps, edge_attr=edge_rep)
        rec_diff = self.aggr_mlp(torch.cat((rec_rep, edge_rep_aggr), dim=-1))  # :148
        rec_rep = self.node_residual_target(rec_rep, edge_rep_aggr) + rec_diff  # :151
        if self.update_edges:
            return rec_rep, edge_rep + edge_diff  # :153-155
        return rec_rep


class PropagationNet(InteractionNet):
    """gnn_layers.py:192-249: forced mean aggregation, sender residual in the
    message, node residual onto the aggregate."""

    def __init__(
        self,
        edge_index,
        input_dim,
        update_edges=True,
        hidden_layers=1,
        hidden_dim=None,
        edge_chunk_sizes=None,
        aggr_chunk_sizes=None,
        aggr="sum",
    ):
        super().__init__(
            edge_index,
            input_dim,
            update_edges=update_edges,
            hidden_layers=hidden_layers,
            hidden_dim=hidden_dim,
            edge_chunk_sizes=edge_chunk_sizes,
            aggr_chunk_sizes=aggr_chunk_sizes,
            aggr="mean",
        )

    def node_residual_target(self, rec_rep, edge_rep_aggr):
        return edge_rep_aggr

    def message(self, x_j, x_i, edge_attr):
        return x_j + self.edge_mlp(torch.cat((edge_attr, x_j, x_i), dim=-1))


GNN_TYPES = {"InteractionNet": InteractionNet, "PropagationNet": PropagationNet}


def get_gnn_class(gnn_type: str):
    if gnn_type not in GNN_TYPES:
        raise ValueError(f"Unknown GNN type '{gnn_type}'. Available types: {list(GNN_TYPES.keys())}")
    return GNN_TYPES[gnn_type]


class Sequential(nn.Module):
    """N same-edge-set layers ``(mesh, mesh, edge) -> (mesh, edge)`` applied in
    order; children named ``module_{i}`` as PyG names them."""

    def __init__(self, layers):
        super().__init__()
        self._n = len(layers)
        for i, layer in enumerate(layers):
            self.add_module(f"module_{i}", layer)

    def forward(self, mesh_rep, edge_rep):
        for i in range(self._n):
            mesh_rep, edge_rep = getattr(self, f"module_{i}")(mesh_rep, mesh_rep, edge_rep)
        return mesh_rep, edge_rep


def make_gnn_seq(edge_index, num_gnn_layers, hidden_layers, hidden_dim, gnn_type="InteractionNet"):
    """utils/networks.py:43-106."""
    if num_gnn_layers < 1:
        raise ValueError(
            f"make_gnn_seq requires num_gnn_layers >= 1 (got {num_gnn_layers}); skip the stage for a no-op."
        )
    cls = get_gnn_class(gnn_type)
    return Sequential([cls(edge_index, hidden_dim, hidden_layers=hidden_layers) for _ in range(num_gnn_layers)])
