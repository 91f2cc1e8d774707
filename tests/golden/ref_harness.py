"""Import the *real* reference hot-path modules from /root/reference.

TEST INFRASTRUCTURE ONLY (used by tests/golden/make_golden.py in the build
container; /root/reference does not exist on the GPU box, and nothing here is
imported by the product).

The reference package cannot be imported as-is: `neural_lam/__init__.py` pulls in
xarray / pytorch_lightning / cartopy, and `gnn_layers.py` subclasses
`torch_geometric.nn.MessagePassing` (PyG 2.3.1, pyproject.toml:33) which is
not installed and cannot be installed (no network).  This harness

  * registers *namespace stand-ins* for the `neural_lam` package levels so that
    the reference's own files `gnn_layers.py`, `utils/networks.py`,
    `utils/graph.py`, `utils/tensor.py`, `utils/buffer_list.py`, `metrics.py`,
    `create_graph.py`, `models/step_predictors/**`, `models/forecasters/**`
    are executed unmodified from where they lie under /root/reference, and
  * provides a minimal stand-in for the four PyG symbols those files touch
    (`nn.MessagePassing`, `nn.Sequential`, `utils.convert.from_networkx`,
    `data.Data`), restating the PyG-2.3.1 behaviour listed in SURVEY.md
    Appendix B.  That stand-in is the only part of the golden vectors that is
    not the reference's own code, and the fixtures say so.
"""
from __future__ import annotations

import importlib
import re
import sys
import types
from pathlib import Path

import numpy as np
import torch
from torch import nn

REF_ROOT = Path("/root/reference")


# --------------------------------------------------------------------------
# torch_geometric stand-in (PyG 2.3.1 semantics)
# --------------------------------------------------------------------------
class _MessagePassing(nn.Module):
    """flow=source_to_target, node_dim=-2; x_j = x[edge_index[0]], x_i = x[edge_index[1]]."""

    def __init__(self, aggr="sum"):
        super().__init__()
        self.aggr = aggr
        self.node_dim = -2

    def propagate(self, edge_index, x, edge_attr):
        j, i = edge_index[0], edge_index[1]
        x_j = x.index_select(self.node_dim, j)
        x_i = x.index_select(self.node_dim, i)
        msgs = self.message(x_j=x_j, x_i=x_i, edge_attr=edge_attr)
        out = self.aggregate(msgs, i, None, x.shape[self.node_dim])
        return self.update(out)

    def message(self, x_j, x_i, edge_attr):  # pragma: no cover - overridden
        return x_j

    def update(self, out):
        return out

    def aggregate(self, inputs, index, ptr, dim_size):
        # scatter(src, index, dim=-2, dim_size, reduce)
        dim_size = int(dim_size)
        size = list(inputs.shape)
        size[-2] = dim_size
        idx = index.view([1] * (inputs.dim() - 2) + [-1, 1]).expand_as(inputs)
        out = inputs.new_zeros(size).scatter_add_(-2, idx, inputs)
        if self.aggr == "sum":
            return out
        if self.aggr == "mean":
            count = inputs.new_zeros(dim_size).scatter_add_(
                0, index, inputs.new_ones(index.shape[0])
            )
            return out / count.clamp(min=1).view(-1, 1)
        raise ValueError(self.aggr)


class _Sequential(nn.Module):
    """pyg.nn.Sequential("a, b", [(module, "a, a, b -> a, b"), ...]).

    Children are registered as `module_{i}` like PyG does, so state_dict keys
    match reference checkpoints.
    """

    def __init__(self, input_args, modules):
        super().__init__()
        self._in = [s.strip() for s in input_args.split(",")]
        self._descs = []
        for i, (mod, desc) in enumerate(modules):
            lhs, rhs = desc.split("->")
            self.add_module(f"module_{i}", mod)
            self._descs.append(
                ([s.strip() for s in lhs.split(",")], [s.strip() for s in rhs.split(",")])
            )

    def forward(self, *args):
        env = dict(zip(self._in, args))
        out = None
        for i, (lhs, rhs) in enumerate(self._descs):
            out = getattr(self, f"module_{i}")(*[env[k] for k in lhs])
            if not isinstance(out, tuple):
                out = (out,)
            env.update(zip(rhs, out))
        return out if len(out) > 1 else out[0]


class _Data:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def clone(self):
        return _Data(
            **{k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.__dict__.items()}
        )


def _from_networkx(G):
    """Node order = G.nodes() order; edges = G.edges() order (both directions for
    undirected graphs); node/edge attributes stacked into tensors."""
    import networkx as nx

    G = G.to_directed() if not nx.is_directed(G) else G
    mapping = dict(zip(G.nodes(), range(G.number_of_nodes())))
    edges = list(G.edges(data=True))
    ei = torch.empty((2, len(edges)), dtype=torch.long)
    for k, (u, v, _) in enumerate(edges):
        ei[0, k] = mapping[u]
        ei[1, k] = mapping[v]
    data = {"edge_index": ei, "num_nodes": G.number_of_nodes()}
    nodes = list(G.nodes(data=True))
    if nodes:
        for key in nodes[0][1].keys():
            data[key] = torch.from_numpy(np.stack([np.asarray(d[key]) for _, d in nodes]))
    if edges:
        for key in edges[0][2].keys():
            data[key] = torch.from_numpy(np.stack([np.asarray(d[key]) for _, _, d in edges]))
    return _Data(**data)


def _install_pyg():
    pyg = types.ModuleType("torch_geometric")
    pyg.__version__ = "2.3.1-standin"
    pyg.nn = types.ModuleType("torch_geometric.nn")
    pyg.nn.MessagePassing = _MessagePassing
    pyg.nn.Sequential = _Sequential
    pyg.data = types.ModuleType("torch_geometric.data")
    pyg.data.Data = _Data
    pyg.utils = types.ModuleType("torch_geometric.utils")
    pyg.utils.convert = types.ModuleType("torch_geometric.utils.convert")
    pyg.utils.convert.from_networkx = _from_networkx
    pyg.utils.is_undirected = lambda ei: False
    for name, mod in [
        ("torch_geometric", pyg),
        ("torch_geometric.nn", pyg.nn),
        ("torch_geometric.data", pyg.data),
        ("torch_geometric.utils", pyg.utils),
        ("torch_geometric.utils.convert", pyg.utils.convert),
    ]:
        sys.modules[name] = mod


# --------------------------------------------------------------------------
# neural_lam namespace stand-ins (skip the heavyweight __init__ files)
# --------------------------------------------------------------------------
def _namespace(name: str, path: Path) -> types.ModuleType:
    mod = types.ModuleType(name)
    mod.__path__ = [str(path)]
    mod.__package__ = name
    sys.modules[name] = mod
    return mod


_LOADED = None


def load_reference():
    """Returns a namespace with the reference's own hot-path modules."""
    global _LOADED
    if _LOADED is not None:
        return _LOADED
    if not REF_ROOT.exists():
        raise RuntimeError("/root/reference is not present (only exists in the build container)")
    _install_pyg()

    loguru = types.ModuleType("loguru")
    loguru.logger = types.SimpleNamespace(
        info=lambda *a, **k: None, warning=lambda *a, **k: None, debug=lambda *a, **k: None
    )
    sys.modules.setdefault("loguru", loguru)

    pkg = REF_ROOT / "neural_lam"
    _namespace("neural_lam", pkg)

    ds = _namespace("neural_lam.datastore", pkg / "datastore")

    class BaseDatastore:  # typing only on the hot path
        pass

    class BaseRegularGridDatastore(BaseDatastore):
        pass

    ds.BaseDatastore = BaseDatastore
    ds_base = types.ModuleType("neural_lam.datastore.base")
    ds_base.BaseDatastore = BaseDatastore
    ds_base.BaseRegularGridDatastore = BaseRegularGridDatastore
    sys.modules["neural_lam.datastore.base"] = ds_base

    cfg = types.ModuleType("neural_lam.config")
    cfg.load_config_and_datastore = None
    sys.modules["neural_lam.config"] = cfg

    utils = _namespace("neural_lam.utils", pkg / "utils")
    sys.modules["neural_lam"].utils = utils
    for sub in ("buffer_list", "tensor", "networks", "graph"):
        m = importlib.import_module(f"neural_lam.utils.{sub}")
        for k, v in vars(m).items():
            if not k.startswith("_") and callable(v) and getattr(v, "__module__", "") == m.__name__:
                setattr(utils, k, v)
    utils.log_on_rank_zero = lambda *a, **k: None

    gnn_layers = importlib.import_module("neural_lam.gnn_layers")
    metrics = importlib.import_module("neural_lam.metrics")
    create_graph = importlib.import_module("neural_lam.create_graph")

    _namespace("neural_lam.models", pkg / "models")
    sp_base = importlib.import_module("neural_lam.models.step_predictors.base")
    graph_pkg = importlib.import_module("neural_lam.models.step_predictors.graph")
    forecasters = importlib.import_module("neural_lam.models.forecasters")

    _LOADED = types.SimpleNamespace(
        utils=utils,
        gnn_layers=gnn_layers,
        metrics=metrics,
        create_graph=create_graph,
        StepPredictor=sp_base.StepPredictor,
        GraphLAM=graph_pkg.GraphLAM,
        HiLAM=graph_pkg.HiLAM,
        HiLAMParallel=graph_pkg.HiLAMParallel,
        ARForecaster=forecasters.ARForecaster,
    )
    return _LOADED


def ref_training_loss(ref, forecaster, batch, per_var_std, interior_mask_bool):
    """ForecasterModule.training_step restated on top of the reference's own
    ARForecaster + metrics.wmse (module.py itself needs Lightning):
    models/module.py:388-391, 496-504, 412."""
    init_states, target_states, forcing = batch
    prediction, pred_std = forecaster(init_states, forcing, target_states)
    if pred_std is None:
        pred_std = per_var_std
    time_step_loss = torch.mean(
        ref.metrics.wmse(prediction, target_states, pred_std, mask=interior_mask_bool), dim=0
    )
    return prediction, torch.mean(time_step_loss)
