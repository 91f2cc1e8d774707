"""MI355X drop-in for ``neural_lam/gnn_layers.py`` and ``neural_lam/utils/networks.py``.

Same class names, constructor arguments, public attributes, parameter names and
``forward`` contract as the reference, so reference checkpoints
load with ``load_state_dict`` and the reference's model code can instantiate
these classes unchanged.  Underneath, each layer is two kernel launches of
``libnlam_hip.so`` (edge kernel: gather + edge MLP + LayerNorm + segment
aggregation [+ edge update]; node kernel: aggr MLP + LayerNorm + residual).

Reference -> here:
  utils.make_mlp (utils/networks.py:8-40)            -> make_mlp / FusedMLP
  InteractionNet (gnn_layers.py:14-189)              -> InteractionNet
  PropagationNet (gnn_layers.py:192-249)             -> PropagationNet
  SplitMLPs (gnn_layers.py:274-324)                  -> SplitMLPs
  GNN_TYPES / get_gnn_class (gnn_layers.py:252-271)  -> same names
  pyg.nn.Sequential of layers (graph_lam.py:117-126) -> GNNSequential
"""
from __future__ import annotations

import torch
from torch import nn

from . import _lib as L
from .graph import EdgeCSR, build_edge_csr, build_tile_schedule
import os

from . import ops
from .ops import (ChunkedGeometry, ChunkedMLPFunction, FusedMLPFunction, MlpGeometry, NodeLinearFunction, NodeLinearPairFunction,
                  as_batched, segment_sum)

# Edge sets with at least this many edges (per batch item) run the FACTORISED edge MLP:
#   W1 [e | x_j | x_i] = W1_e e + (W1_j x)[sender] + (W1_i x)[receiver]
# -- the two node-level products are computed once per node (nlam_linear) and gathered as pre-activation addends, so the
# edge kernels run one third of the first GEMM (half of all their matrix work, a third of the weight staging and of the
# dW1 weight gradient), and the sender-side data gradient is a segment sum of dz1 instead of a (E, d) round trip.
# Below the threshold the two extra node-level launches cost more than they save (the mesh-level layers of cfg2 are
# latency-bound single-tile-per-wave launches).
FACTORISE_MIN_EDGES = int(os.environ.get("NLAM_FACTORISE_MIN_EDGES", str(1 << 30)))
# Widths above 64 (cfg3 / cfg4 / cfg5) are bound by the matrix cores, where dropping two thirds of the first GEMM pays
# (m2g at d = 256: forward 854 -> 581 us, backward 1029 -> 694 us, dW1 643 -> 239 us; cfg3 step 54.6 -> 49.1 ms) as soon
# as the edge set is large enough to amortise the node-level launches: edges x width >= FACTORISE_MIN_WORK_WIDE
# (the 57 616-edge mesh layers at d = 256 gain, the Hi-LAM level layers at d = 128 lose: cfg4 12.5 -> 13.4 ms when
# everything is factorised), on the edge sets the split-bf16 super-tile kernels take (nlam_pre_add_supported).
FACTORISE_MIN_EDGES_WIDE = int(os.environ.get("NLAM_FACTORISE_MIN_EDGES_WIDE", "0"))
FACTORISE_MIN_WORK_WIDE = int(os.environ.get("NLAM_FACTORISE_MIN_WORK_WIDE", "14000000"))
# d = 128 (cfg4, Hi-LAM): the one qualifying edge set (m2g) is faster in isolation (forward 328 -> 245 us, backward 401 ->
# 297 us) but the captured step measured slower with it (12.3 -> 13.4 ms, reproducibly; equal under the profiler), so the
# factorised path starts at d = 256
FACTORISE_MIN_WIDTH_WIDE = int(os.environ.get("NLAM_FACTORISE_MIN_WIDTH_WIDE", "256"))


PAD_EMBEDDER_MIN_WIDTH = int(os.environ.get("NLAM_PAD_EMBEDDER_MIN_WIDTH", "256"))


_DEVICE_LAYOUTS: dict = {}   # (id of the host EdgeCSR, device) -> device EdgeCSR (+ tiles): shared by every layer built on that edge set


def clear_layout_caches():
    """Drop the module-level edge-layout caches (host layouts by content hash in ``graph``, their device copies here).  Layers
    that exist keep their own references; a process that builds many distinct graphs (sweeps, test suites) calls this between
    them so that index tensors of graphs no layer uses any more are released."""
    from . import graph as _graph

    _DEVICE_LAYOUTS.clear()
    getattr(_graph, "_LAYOUT_CACHE", {}).clear()


_CONSTS: dict = {}   # (kind, n, device) -> identity matrix / zero vector shared by every MLP that needs one (never trained)


def _const(kind: str, n: int, device):
    key = (kind, n, str(device))
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.eye(n, device=device) if kind == "eye" else torch.zeros(n, device=device)
    return t


class FusedMLP(nn.Sequential):
    """``[Linear -> SiLU] * hidden_layers -> Linear [-> LayerNorm]`` with the reference's child names
    (``0``, ``2``, ..: Linear; last: LayerNorm; utils/networks.py:8-40).

    ``hidden_layers == 1`` (every BASELINE config, the reference's default, train_model.py:193-197) is ONE launch of the
    fused kernel ``Linear -> SiLU -> Linear [-> LayerNorm]`` and takes the gather / concat / residual / aggregation geometry of
    the GNN layers.  Every other depth is a chain of launches of the SAME kernels (no library GEMM, no eager op):

    * a leading ``Linear -> SiLU`` is the fused kernel with ``W2 = I, b2 = 0`` and no LayerNorm (an exact product in the
      fp32-class matrix modes: ``1.0`` is one bf16 term) -- the first one with the caller's gather / concat geometry;
    * the trailing ``Linear -> SiLU -> Linear [-> LayerNorm]`` is the fused kernel as it is; when the caller wants residual
      terms it reads the residual sources as extra inputs whose columns of the first weight matrix are zero;
    * ``hidden_layers == 0`` (``Linear [-> LayerNorm]``) is ONE launch with ``NLAM_F_NO_ACT`` and ``W2 = I``, full geometry.

    The identity / zero operands are constants shared per width and device; they get no gradient.  GPU only."""

    def __init__(self, blueprint, layer_norm: bool = True):
        hidden_layers = len(blueprint) - 2
        assert hidden_layers >= 0, "Invalid MLP blueprint"
        layers = []
        for i, (d1, d2) in enumerate(zip(blueprint[:-1], blueprint[1:])):
            layers.append(nn.Linear(d1, d2))
            if i != hidden_layers:
                layers.append(nn.SiLU())
        if layer_norm:
            layers.append(nn.LayerNorm(blueprint[-1]))
        super().__init__(*layers)
        self.has_layer_norm = layer_norm
        self.hidden_layers = hidden_layers
        self._geom = MlpGeometry(nsrc=1)
        self._geom_noact = MlpGeometry(nsrc=1, flags=L.F_NO_ACT)
        self._geom_padded = MlpGeometry(nsrc=1, no_pack=True)

    @property
    def fully_fused(self) -> bool:
        return self.hidden_layers == 1

    def _linears(self):
        return [m for m in self if isinstance(m, nn.Linear)]

    def _ln(self):
        ln = self[len(self) - 1] if self.has_layer_norm else None
        return (ln.weight, ln.bias) if ln is not None else (None, None)

    def params(self):
        """(W1, b1, W2, b2, ln_w, ln_b) of the last ``Linear -> SiLU -> Linear [-> LN]`` block."""
        lin = self._linears()
        return (lin[-2].weight, lin[-2].bias, lin[-1].weight, lin[-1].bias, *self._ln())

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("neural_lam_amd layers run on MI355X only (no CPU / eager fallback; see oracle/ for a CPU reference)")
        leaf_input = not x.requires_grad
        lin0 = self[0]
        if (self.hidden_layers == 1 and leaf_input and x.shape[-1] % 4 != 0 and x.dtype == torch.float32
                and max(lin0.out_features, self[2].out_features) > PAD_EMBEDDER_MIN_WIDTH):
            # A WIDE embedder of a few static feature columns (edge features [len, dx, dy], mesh coordinates: 2 - 3 columns,
            # graph/base.py:286-295): the split-bf16 super-tile kernels read their inputs as 16-byte pieces, so a width that is
            # not a multiple of 4 fell to the fp32 one-tile kernels (at d = 512 the <8,2> instantiation: 2.0 ms per launch at
            # 255 136 rows, 8.1 ms = 5 % of the cfg5 step for the four embedders).  The input gets zero columns up to the next
            # multiple of 4 (cached: the features are static) and the first weight matrix matching zero columns.  From d > 256
            # only: the padded first weight is not a trainer-owned gradient view, so this MLP's weight-gradient launches stay on the
            # backward stream -- at d = 128 / 256, where the fp32 kernels are not pathological, that cost more than the kernels
            # gained (cfg4 11.57 -> 11.93 ms, cfg3 48.5 -> 49.0; cfg5 152.8 -> 146.4 ms, forecast 170 -> 201 steps/s).
            out, _ = FusedMLPFunction.apply(self._geom_padded, _PadWeight.apply(lin0.weight, self._padded_weight(lin0.weight)), lin0.bias,
                                            self[2].weight, self[2].bias, *self._ln(), self._padded_input(x))
            return ops.early_backward_leaf(out)
        out, _ = self.forward_fused(self._geom, x)
        # an MLP of input data / static features: nothing upstream needs its data gradient (ops.early_backward_leaf)
        return ops.early_backward_leaf(out) if leaf_input else out

    _xpad = None
    _wpad = None

    def _padded_input(self, x):
        """Zero-padded copy of ``x``, cached for as long as the caller keeps handing in THE SAME tensor object unmodified (the
        registered static-feature buffers: graph/base.py:286-295).  The key is object identity through a weak reference plus
        the version counter -- never the address: a fresh activation that the caching allocator places where an earlier one
        lived is another tensor and is padded again (advisor finding, round 4: 16 stale hits out of 20 tensors by address)."""
        import weakref

        c = self._xpad
        if c is None or c[0]() is not x or c[2] != x._version:
            c = self._xpad = (weakref.ref(x), torch.nn.functional.pad(x.detach(), (0, -x.shape[-1] % 4)).contiguous(), x._version)
        return c[1]

    def _padded_weight(self, W):
        """Persistent (hid, padded width) buffer: its address is what the weight packer keys on."""
        kp = W.shape[1] + (-W.shape[1] % 4)
        if self._wpad is None or self._wpad.device != W.device or tuple(self._wpad.shape) != (W.shape[0], kp):
            self._wpad = torch.zeros((W.shape[0], kp), device=W.device, dtype=W.dtype)
        return self._wpad

    def forward_fused(self, geom: MlpGeometry, *srcs):
        """Run with a caller-supplied geometry (gathered / concatenated sources, residuals, aggregation) -> (out, aggr)."""
        lin = self._linears()
        n = len(lin)
        dev = srcs[0].device
        if n == 2:
            return FusedMLPFunction.apply(geom, *self.params(), *srcs)
        if n == 1:   # Linear [-> LayerNorm]: no activation, identity second Linear
            dout = lin[0].out_features
            g = self._geom_noact if geom is self._geom else _with_flags(geom, geom.flags | L.F_NO_ACT)
            return FusedMLPFunction.apply(g, lin[0].weight, lin[0].bias, _const("eye", dout, dev), _const("zero", dout, dev),
                                          *self._ln(), *srcs)
        # ---- hidden_layers >= 2: [Linear -> SiLU] prefixes, then the fused pair ----
        first = _prefix_geometry(geom)
        h = lin[0].out_features
        a, _ = FusedMLPFunction.apply(first, lin[0].weight, lin[0].bias, _const("eye", h, dev), _const("zero", h, dev), None, None, *srcs)
        for k in range(1, n - 2):
            h = lin[k].out_features
            a, _ = FusedMLPFunction.apply(self._geom, lin[k].weight, lin[k].bias, _const("eye", h, dev), _const("zero", h, dev),
                                          None, None, a)
        last, res_srcs = _suffix_geometry(geom, srcs)
        W1 = lin[n - 2].weight
        if res_srcs:   # the residual sources ride along as inputs with zero weight columns
            W1 = torch.cat([W1.new_zeros(W1.shape[0], sum(t.shape[-1] for t in res_srcs)), W1], dim=1)
        return FusedMLPFunction.apply(last, W1, lin[n - 2].bias, lin[n - 1].weight, lin[n - 1].bias, *self._ln(), *res_srcs, a)


class _PadWeight(torch.autograd.Function):
    """``W`` (hid, k) copied into the leading columns of a persistent zero-padded buffer (hid, k') -> that buffer; the gradient
    is the leading (hid, k) block of the buffer's gradient.  The copy is a (hid x k) elementwise launch per step."""

    @staticmethod
    def forward(ctx, W, buf):
        buf[:, : W.shape[1]].copy_(W.detach())
        ctx.k = W.shape[1]
        return buf.view_as(buf)   # a fresh view: the buffer itself stays a plain tensor without history

    @staticmethod
    def backward(ctx, g):
        return (g[:, : ctx.k].contiguous() if g is not None else None), None


def _with_flags(geom: MlpGeometry, flags: int) -> MlpGeometry:
    import dataclasses

    key = ("flags", flags)
    cache = geom.__dict__.setdefault("_derived", {})
    if key not in cache:
        cache[key] = dataclasses.replace(geom, flags=flags)
        cache[key].__dict__.pop("_derived", None)
    return cache[key]


def _prefix_geometry(geom: MlpGeometry) -> MlpGeometry:
    """Geometry of the FIRST launch of a deeper MLP: the caller's sources (gather indices, tiles, gradient modes), output rows
    in tile-row order, no residual, no aggregation."""
    import dataclasses

    cache = geom.__dict__.setdefault("_derived", {})
    if "prefix" not in cache:
        g = dataclasses.replace(geom, flags=0, out_idx=None, out_rows=None, want_out=True, aggregate=False)
        g.__dict__.pop("_derived", None)
        cache["prefix"] = g
    return cache["prefix"]


def _suffix_geometry(geom: MlpGeometry, srcs):
    """Geometry of the LAST launch of a deeper MLP, whose input is the previous launch's output (tile-row order, identity
    index): the caller's output index / aggregation / residual flags.  NLAM_F_ADD_SRC0 / _SRC1 name sources 0 / 1 of a launch, so
    those sources stay in place (gathered as the caller gathers them, zero weight columns) and the real input comes last."""
    import dataclasses

    keep = 2 if geom.flags & L.F_ADD_SRC1 else (1 if geom.flags & L.F_ADD_SRC0 else 0)
    cache = geom.__dict__.setdefault("_derived", {})
    if "suffix" not in cache:
        idx = [geom.src_idx[k] if k < len(geom.src_idx) else None for k in range(keep)] + [None]
        dm = [geom.dmode[k] for k in range(keep)] + [1]
        # residual sources ride along with a per-call zero-padded first weight (torch.cat in forward_fused): a temporary the
        # trainer's weight packer must never register -- it keys on, and re-reads at the start of every later step, the address
        g = dataclasses.replace(geom, nsrc=keep + 1, src_idx=idx + [None] * (3 - len(idx)), dmode=dm + [1] * (3 - len(dm)),
                                no_pack=bool(geom.no_pack or keep > 0))
        g.__dict__.pop("_derived", None)
        cache["suffix"] = g
    return cache["suffix"], tuple(srcs[:keep])


def grouped_mlp_forward(pairs):
    """``[mlp(x) for mlp, x in pairs]`` for independent MLPs of data inputs (static feature embedders), as grouped launches of
    up to 8 members (ops.GroupedMLPFunction) where the members are one-kernel MLPs of the same shape on GPU tensors."""
    def ok(m, x):
        # (wide members have no grouped kernel: they run one by one, through FusedMLP.forward and its input padding)
        return (isinstance(m, FusedMLP) and m.fully_fused and x.is_cuda and x.dim() == 2 and not x.requires_grad
                and x.dtype == torch.float32 and max(m[0].out_features, m[2].out_features) <= PAD_EMBEDDER_MIN_WIDTH)

    outs = [None] * len(pairs)
    classes = {}
    for i, (m, x) in enumerate(pairs):
        if ok(m, x):
            classes.setdefault((m[0].out_features, m[2].out_features, m.has_layer_norm), []).append(i)
        else:
            outs[i] = m(x)
    for idxs in classes.values():
        for c0 in range(0, len(idxs), 8):
            chunk = idxs[c0 : c0 + 8]
            if len(chunk) == 1:
                outs[chunk[0]] = pairs[chunk[0]][0](pairs[chunk[0]][1])
                continue
            flat = [q for i in chunk for q in pairs[i][0].params()]
            res = ops.GroupedMLPFunction.apply(len(chunk), *flat, *[pairs[i][1] for i in chunk])
            for i, r in zip(chunk, res):
                outs[i] = r
    return outs


def make_mlp(blueprint, layer_norm: bool = True) -> FusedMLP:
    """utils/networks.py:8-40."""
    return FusedMLP(blueprint, layer_norm=layer_norm)


class SplitMLPs(nn.Module):
    """gnn_layers.py:274-324: chunks of dim -2 through separate MLPs."""

    def __init__(self, mlps, chunk_sizes):
        super().__init__()
        assert len(mlps) == len(chunk_sizes), "Number of MLPs must match the number of chunks"
        self.mlps = nn.ModuleList(mlps)
        self.chunk_sizes = chunk_sizes

    @property
    def fully_fused(self) -> bool:
        """Every chunk is a one-kernel MLP of the same shape: the chunked native path (ops.ChunkedMLPFunction) applies."""
        first = self.mlps[0]
        return all(isinstance(m, FusedMLP) and m.fully_fused and m.has_layer_norm == first.has_layer_norm
                   and m[0].weight.shape == first[0].weight.shape and m[2].weight.shape == first[2].weight.shape for m in self.mlps)

    def flat_params(self):
        return [q for m in self.mlps for q in m.params()]

    def forward(self, x):
        if x.is_cuda and self.fully_fused and sum(self.chunk_sizes) == x.shape[-2]:
            # one launch per chunk on its row window of x, written into one output buffer: no split / cat copies
            key = ("plain", str(x.device))
            if getattr(self, "_geom", None) is None or self._geom[0] != key:
                bounds, r = [], 0
                for n in self.chunk_sizes:
                    bounds.append((r, r + n))
                    r += n
                self._geom = (key, ChunkedGeometry(nsrc=1, chunks=bounds, rows=r, src_mode=["slice"], src_idx=[None]))
            out, _ = ChunkedMLPFunction.apply(self._geom[1], len(self.mlps), *self.flat_params(), x)
            return out
        chunks = torch.split(x, self.chunk_sizes, dim=-2)
        return torch.cat([mlp(c.contiguous()) for mlp, c in zip(self.mlps, chunks)], dim=-2)


class _SegmentAggregate(torch.autograd.Function):
    """messages in original edge order -> sum/mean per receiver (CSR segment sum)."""

    @staticmethod
    def forward(ctx, msgs, csr: EdgeCSR, mean: bool):
        t, B, bstride, lead = as_batched(msgs)
        d = msgs.shape[-1]
        out = segment_sum(t, bstride, csr.rowptr, csr.perm, csr.inv_deg if mean else None, csr.num_rec, d, B)
        ctx.csr, ctx.mean, ctx.shape = csr, mean, msgs.shape
        return out.reshape(*lead, csr.num_rec, d)

    @staticmethod
    def backward(ctx, g):
        csr = ctx.csr
        g = g.reshape(-1, csr.num_rec, g.shape[-1])
        if ctx.mean:
            g = g * csr.inv_deg.view(1, -1, 1)
        gm = torch.empty((g.shape[0], csr.num_edges, g.shape[-1]), device=g.device, dtype=g.dtype)
        gm[:, csr.perm.long()] = g[:, csr.rec.long()]
        return gm.reshape(ctx.shape), None, None


class InteractionNet(nn.Module):
    """Interaction network layer (Battaglia et al. 2016) as in gnn_layers.py:14-189."""

    def __init__(
        self,
        edge_index: torch.Tensor,
        input_dim: int,
        update_edges: bool = True,
        hidden_layers: int = 1,
        hidden_dim: int | None = None,
        edge_chunk_sizes: list[int] | None = None,
        aggr_chunk_sizes: list[int] | None = None,
        aggr: str = "sum",
    ) -> None:
        if aggr not in ("sum", "mean"):
            raise ValueError(f"Unknown aggregation method: {aggr}")
        super().__init__()
        self.aggr = aggr
        if hidden_dim is None:
            hidden_dim = input_dim
        self.num_rec = edge_index[1].max() + 1  # 0-dim tensor like the reference (gnn_layers.py:73)
        self._edge_index_local = edge_index.detach().cpu().to(torch.int64)
        self.register_buffer(
            "edge_index", torch.stack((edge_index[0] + self.num_rec, edge_index[1]), dim=0), persistent=False
        )
        edge_recipe = [3 * input_dim] + [hidden_dim] * (hidden_layers + 1)
        aggr_recipe = [2 * input_dim] + [hidden_dim] * (hidden_layers + 1)
        self.hidden_layers = hidden_layers
        if edge_chunk_sizes is None:
            self.edge_mlp = make_mlp(edge_recipe)
        else:
            self.edge_mlp = SplitMLPs([make_mlp(edge_recipe) for _ in edge_chunk_sizes], edge_chunk_sizes)
        if aggr_chunk_sizes is None:
            self.aggr_mlp = make_mlp(aggr_recipe)
        else:
            self.aggr_mlp = SplitMLPs([make_mlp(aggr_recipe) for _ in aggr_chunk_sizes], aggr_chunk_sizes)
        self.update_edges = update_edges
        self._csr_cache: dict = {}
        self._geom_cache: dict = {}
        # receiver-sorted CSR / sender-sorted CSC views and the wave-tile schedule: built here, once, on the host (the
        # first forward only copies them to the device; a sender tensor with more rows than the largest sender index
        # + 1 -- legal, gnn_layers.py:73 only fixes num_rec -- gets its own CSC built on demand)
        self._num_send_default = int(self._edge_index_local[0].max()) + 1
        self._host_csr = self._build_host_csr(self._num_send_default)

    # reference hook points kept for API parity (gnn_layers.py:159-166, 231-239)
    propagates_sender = False  # message() adds x_j               (PropagationNet)
    residual_on_aggregate = False  # node residual targets the aggregate (PropagationNet)

    def node_residual_target(self, rec_rep, edge_rep_aggr):
        return edge_rep_aggr if self.residual_on_aggregate else rec_rep

    # ---- device-side graph structure, built lazily per (device, num_send) ----
    def _csr(self, device, num_send: int) -> EdgeCSR:
        key = (str(device), num_send)
        if key not in self._csr_cache:
            if num_send <= int(self._edge_index_local[0].max()):
                raise RuntimeError("send_rep has fewer rows than the largest sender index in edge_index")
            host, tiles, has_split = self._host_csr if num_send == self._num_send_default else self._build_host_csr(num_send)
            dkey = (id(host), str(device))   # layers on the same edge set (content-cached host layout) share ONE device copy
            csr = _DEVICE_LAYOUTS.get(dkey)
            if csr is None or csr._host is not host:
                csr = host.to(device)
                csr.tiles = tiles.to(device)
                csr.has_split = has_split
                csr._host = host   # keeps the host object alive: its id() is the key
                _DEVICE_LAYOUTS[dkey] = csr
            self._csr_cache[key] = csr
        return self._csr_cache[key]

    def _build_host_csr(self, num_send: int):
        # content-cached: the processor layers share one edge set, and load_graph may have brought the layout along
        from .graph import edge_layout

        return edge_layout(self._edge_index_local, num_send=num_send, num_rec=int(self.num_rec))

    def _edge_geom(self, csr: EdgeCSR, want_out: bool, add_edge: bool, key, pre: bool = False) -> MlpGeometry:
        gkey = (key, want_out, add_edge, pre)
        if gkey not in self._geom_cache:
            flags = L.F_PRE_ADD if pre else 0
            if add_edge:
                flags |= L.F_ADD_SRC0
            if self.propagates_sender:
                flags |= L.F_ADD_SRC1
            if self.aggr == "mean":
                flags |= L.F_MEAN
            self._geom_cache[gkey] = MlpGeometry(
                nsrc=3,
                flags=flags,
                src_idx=[csr.perm, csr.send, csr.rec],
                rows=csr.num_edges,
                tiles=csr.tiles,
                out_idx=csr.perm,
                out_rows=csr.num_edges,
                want_out=want_out,
                aggregate=True,
                # receivers cut over several tiles reduce piecewise into virtual segments (extended row pointers / scales),
                # summed up afterwards by nlam_split_combine: deterministic, the kernels see ordinary tiles
                rowptr=csr.rowptr_ext if csr.comb_ptr is not None else csr.rowptr,
                inv_deg=csr.inv_deg_ext if csr.comb_ptr is not None else csr.inv_deg,
                seg_of_row=csr.rec,
                nseg_total=csr.num_rec,
                nseg_ext=csr.nseg_ext,
                comb=(csr.comb_ptr, csr.comb_src, csr.comb_dst) if csr.comb_ptr is not None else None,
                has_split=csr.has_split,
                dmode=[1, 2, 3],
                colptr=csr.colptr,
                cperm=csr.cperm,
                num_send=csr.num_send,
            )
        return self._geom_cache[gkey]

    def _node_geom(self) -> MlpGeometry:
        if "node" not in self._geom_cache:
            flags = L.F_ADD_SRC1 if self.residual_on_aggregate else L.F_ADD_SRC0
            self._geom_cache["node"] = MlpGeometry(nsrc=2, flags=flags)
        return self._geom_cache["node"]

    def _check_inputs(self, send_rep, rec_rep, edge_rep):
        n_rec = int(self.num_rec)
        if rec_rep.shape[-2] != n_rec:
            raise RuntimeError(f"rec_rep has {rec_rep.shape[-2]} rows, layer was built for num_rec={n_rec}")
        if edge_rep.shape[-2] != self._edge_index_local.shape[1]:
            raise RuntimeError("edge_rep rows do not match the number of edges")

    def _messages_and_aggregate(self, send_rep, rec_rep, edge_rep, want_out: bool, add_edge: bool, rec_alias: bool = False):
        """-> (aggr, edge_out | None); edge_out = msg (+ edge_rep if add_edge), original edge order."""
        self._check_inputs(send_rep, rec_rep, edge_rep)
        csr = self._csr(send_rep.device, send_rep.shape[-2])
        if isinstance(self.edge_mlp, SplitMLPs) and self.edge_mlp.fully_fused:
            return self._messages_chunked(csr, send_rep, rec_rep, edge_rep, want_out, add_edge)
        if isinstance(self.edge_mlp, SplitMLPs):   # chunked AND another depth than the default: per-chunk MLP chains on an explicit gather
            return self._messages_generic(csr, send_rep, rec_rep, edge_rep, want_out, add_edge)
        key = (str(send_rep.device), send_rep.shape[-2])
        if csr.has_split and torch.are_deterministic_algorithms_enabled():
            # only with graph.VIRTUAL_SPLIT switched off (the one-pass NLAM_TILE_SPLIT tiles of the C-ABI): partial sums of a
            # receiver's pieces then meet through atomic adds, the one summation order of the fused path that is not fixed
            # (train_model.py:566 trains with deterministic=True).  The default schedule reduces such receivers in two
            # fixed-order passes and never gets here.
            msg = (f"this edge set has receivers with in-degree > 32 (max {csr.max_in_degree}) on NLAM_TILE_SPLIT tiles: their aggregation "
                   "and receiver gradients use atomic adds and are not bit-reproducible")
            if torch.is_deterministic_algorithms_warn_only_enabled():
                import warnings

                warnings.warn(msg)
            else:
                raise RuntimeError(msg + "; torch.use_deterministic_algorithms(True) is set")
        if self._factorise(csr, send_rep, rec_rep, edge_rep):
            W1 = self.edge_mlp[0].weight
            d = edge_rep.shape[-1]
            # mail: the product of the RECEIVER table accumulates its data gradient onto the node MLP's (ops.mail_scope; the
            # caller -- forward() -- has made rec_rep a private alias, so no other gradient meets these two)
            mail = rec_alias
            if send_rep is rec_rep and d > 64:   # mesh <-> mesh: both products of the one node table in a single launch
                p_send, p_rec = NodeLinearPairFunction.apply(send_rep, W1, d, 2 * d, mail)
            else:
                # (a table that is sender AND receiver: whichever product's backward runs first takes the posted buffer, the other
                # reports its gradient to autograd as usual -- correct in either order)
                p_send = NodeLinearFunction.apply(send_rep, W1, d, mail and send_rep is rec_rep)   # (W1_j x) per sender node
                p_rec = NodeLinearFunction.apply(rec_rep, W1, 2 * d, mail)                          # (W1_i x) per receiver node
            geom = self._edge_geom(csr, want_out, add_edge, key, pre=True)
            edge_out, aggr = self.edge_mlp.forward_fused(geom, edge_rep, p_send, p_rec)
            return aggr, edge_out
        geom = self._edge_geom(csr, want_out, add_edge, key)
        edge_out, aggr = self.edge_mlp.forward_fused(geom, edge_rep, send_rep, rec_rep)
        return aggr, edge_out

    def _factorise(self, csr, send_rep, rec_rep, edge_rep) -> bool:
        """Factorised edge MLP (see FACTORISE_MIN_EDGES): InteractionNet messages (no ``x_j +`` term), split-bf16 matrix
        modes, widths that the narrow kernels take as whole 32-column units."""
        if not self.edge_mlp.fully_fused:
            return False
        d, hid = edge_rep.shape[-1], self.edge_mlp[0].out_features
        dout = self.edge_mlp[2].out_features
        wide = max(d, hid, dout) > 64
        if self.propagates_sender or csr.num_edges < (FACTORISE_MIN_EDGES_WIDE if wide else FACTORISE_MIN_EDGES):
            return False
        if wide and (csr.num_edges * max(d, hid) < FACTORISE_MIN_WORK_WIDE or max(d, hid) < FACTORISE_MIN_WIDTH_WIDE):
            return False
        mm = ops._mm_flags()
        if (mm >> 8) & 3 == 0 or send_rep.shape[-1] != d or rec_rep.shape[-1] != d:
            return False
        B = 1
        for n in max((t.shape[:-2] for t in (send_rep, rec_rep, edge_rep)), key=len):
            B *= n
        key = ("pre_ok", d, hid, dout, mm, B)
        if key not in self._geom_cache:   # ask the library whether this launch has a factorised kernel (shape-only query)
            import ctypes as C

            q = L.MlpFwd()
            q.nsrc, q.batch, q.rows, q.ntiles = 3, B, csr.num_edges, int(csr.tiles.shape[0])
            for k, w in enumerate((d, hid, hid)):
                q.src[k].width = w
            q.hid, q.dout, q.flags = hid, dout, L.F_PRE_ADD | mm
            self._geom_cache[key] = bool(L.load().nlam_pre_add_supported(C.byref(q)))
        return self._geom_cache[key]

    def _chunk_bounds(self, sizes):
        bounds, r = [], 0
        for n in sizes:
            bounds.append((r, r + n))
            r += n
        return bounds, r

    def _messages_chunked(self, csr, send_rep, rec_rep, edge_rep, want_out, add_edge):
        """Chunked edge MLPs (HiLAMParallel, hi_lam_parallel.py:127-143) natively: per chunk one fused gather + MLP
        launch on its window of the edge rows, one CSR segment sum for the aggregation; deterministic backward
        (ops.ChunkedMLPFunction).  Rows stay in the original edge order."""
        gkey = ("chunked_edge", str(send_rep.device), send_rep.shape[-2])
        if gkey not in self._geom_cache:
            dev = send_rep.device
            ei = self._edge_index_local
            bounds, total = self._chunk_bounds(self.edge_mlp.chunk_sizes)
            assert total == ei.shape[1], "edge_chunk_sizes must add up to the number of edges"
            send_i, rec_i = ei[0].to(torch.int32).contiguous().to(dev), ei[1].to(torch.int32).contiguous().to(dev)
            order_send = torch.argsort(ei[0], stable=True).to(torch.int32).contiguous().to(dev)   # CSC position -> edge id
            flags = (L.F_ADD_SRC1 if self.propagates_sender else 0) | (L.F_MEAN if self.aggr == "mean" else 0)
            self._geom_cache[gkey] = ChunkedGeometry(
                nsrc=3, chunks=bounds, rows=total, src_mode=["slice", "gather", "gather"], src_idx=[None, send_i, rec_i],
                flags_fwd=flags, flags_bwd=flags,   # the edge residual (edge_rep + messages) is added outside: the aggregate needs the bare messages
                rowptr=csr.rowptr, perm=csr.perm, inv_deg=csr.inv_deg, seg_of_row=rec_i, num_rec=csr.num_rec, mean=self.aggr == "mean",
                scatter=[None, (csr.colptr, order_send, csr.num_send), (csr.rowptr, csr.perm, csr.num_rec)],
            )
        geom = self._geom_cache[gkey]
        lead = max((t.shape[:-2] for t in (send_rep, rec_rep, edge_rep)), key=len)
        edge_b = edge_rep.expand(*lead, -1, -1) if edge_rep.shape[:-2] != lead else edge_rep
        msgs, aggr = ChunkedMLPFunction.apply(geom, len(self.edge_mlp.mlps), *self.edge_mlp.flat_params(), edge_b, send_rep, rec_rep)
        edge_out = None
        if want_out:
            edge_out = edge_rep + msgs if add_edge else msgs
        return aggr, edge_out

    def _messages_generic(self, csr, send_rep, rec_rep, edge_rep, want_out, add_edge):
        # chunked edge MLPs (HiLAMParallel) of another depth than hidden_layers = 1: every chunk's MLP is a chain of fused
        # launches (FusedMLP) over an explicitly gathered concat; aggregation by the CSR segment-sum kernel.  The gather /
        # concat / split here are data movement through torch; every GEMM, activation and LayerNorm runs in the library.
        # device copy of the local edge index, made once per device (a host-to-device copy per call would also be
        # illegal inside a HIP-graph capture; the trainer's warm-up steps populate this cache before capturing)
        dkey = ("ei", str(send_rep.device))
        if dkey not in self._geom_cache:
            self._geom_cache[dkey] = self._edge_index_local.to(send_rep.device)
        ei = self._geom_cache[dkey]
        x_j = send_rep.index_select(-2, ei[0])
        x_i = rec_rep.index_select(-2, ei[1])
        msgs = self.edge_mlp(torch.cat((edge_rep.expand(*x_j.shape[:-2], -1, -1), x_j, x_i), dim=-1))
        if self.propagates_sender:
            msgs = x_j + msgs
        aggr = _SegmentAggregate.apply(msgs, csr, self.aggr == "mean")
        edge_out = None
        if want_out:
            edge_out = edge_rep + msgs if add_edge else msgs
        return aggr, edge_out

    def _node_update(self, rec_rep, aggr):
        if isinstance(self.aggr_mlp, SplitMLPs) and self.aggr_mlp.fully_fused and rec_rep.is_cuda:
            # chunked node MLPs: per chunk one fused launch [rec | aggr] -> MLP -> LayerNorm -> + residual on its window of the nodes
            gkey = ("chunked_node", str(rec_rep.device))
            if gkey not in self._geom_cache:
                bounds, total = self._chunk_bounds(self.aggr_mlp.chunk_sizes)
                flags = L.F_ADD_SRC1 if self.residual_on_aggregate else L.F_ADD_SRC0
                self._geom_cache[gkey] = ChunkedGeometry(nsrc=2, chunks=bounds, rows=total, src_mode=["slice", "slice"], src_idx=[None, None],
                                                         flags_fwd=flags, flags_bwd=flags)
            geom = self._geom_cache[gkey]
            if geom.rows == rec_rep.shape[-2]:
                lead = max((rec_rep.shape[:-2], aggr.shape[:-2]), key=len)
                rec_b = rec_rep.expand(*lead, -1, -1) if rec_rep.shape[:-2] != lead else rec_rep
                out, _ = ChunkedMLPFunction.apply(geom, len(self.aggr_mlp.mlps), *self.aggr_mlp.flat_params(), rec_b, aggr)
                return out
        if isinstance(self.aggr_mlp, SplitMLPs):
            rec_diff = self.aggr_mlp(torch.cat((rec_rep, aggr), dim=-1))
            return self.node_residual_target(rec_rep, aggr) + rec_diff
        if self.aggr_mlp.fully_fused:   # ONE launch: its dense gradient of rec_rep may be posted for the node-level product's backward
            with ops.mail_post():
                out, _ = self.aggr_mlp.forward_fused(self._node_geom(), rec_rep, aggr)
            return out
        out, _ = self.aggr_mlp.forward_fused(self._node_geom(), rec_rep, aggr)
        return out

    def forward(self, send_rep, rec_rep, edge_rep, need_edges: bool | None = None):
        """``(send (..,N_s,d), rec (..,N_r,d), edge (..,E,d)) -> rec | (rec, edge)``
        (gnn_layers.py:110-157).  ``need_edges=False`` lets a caller that discards
        the edge output (graph_lam.py:185) skip writing it."""
        if need_edges is None:
            need_edges = self.update_edges
        with ops.mail_scope() as tok:
            # fp32 tables only: the fused functions widen other dtypes INSIDE the Function, autograd then casts the fp32 gradient they
            # return into a new tensor and a posted buffer would be orphaned (the product's data gradient silently lost)
            alias = tok is not None and rec_rep.is_cuda and rec_rep.requires_grad and rec_rep.dtype == torch.float32
            if alias:
                # a private alias of the receiver table for THIS layer's two consumers of it (node-level product, node MLP):
                # their gradients meet in the alias' own autograd node, where the second can be accumulated onto the first
                same = send_rep is rec_rep
                rec_rep = rec_rep.view_as(rec_rep)
                if same:
                    send_rep = rec_rep
                ops.MAIL_ALIASED = tok
            aggr, edge_out = self._messages_and_aggregate(send_rep, rec_rep, edge_rep, need_edges, True, rec_alias=alias)
            rec_out = self._node_update(rec_rep, aggr)
        if self.update_edges:
            return rec_out, edge_out
        return rec_out

    def propagate(self, edge_index, x, edge_attr):
        """Compatibility shim for code that drives PyG's ``propagate`` directly
        (tests/test_gnn_layers.py:249, 290, 380): ``x = cat(rec, send)`` ->
        ``(aggregate, messages)``."""
        n_rec = int(self.num_rec)
        rec_rep, send_rep = x[..., :n_rec, :], x[..., n_rec:, :]
        aggr, msgs = self._messages_and_aggregate(send_rep, rec_rep, edge_attr, True, False)
        return aggr, msgs


class PropagationNet(InteractionNet):
    """gnn_layers.py:192-249: mean aggregation, ``x_j + edge_mlp(..)`` messages,
    node residual onto the aggregate."""

    propagates_sender = True
    residual_on_aggregate = True

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


GNN_TYPES = {"InteractionNet": InteractionNet, "PropagationNet": PropagationNet}


def get_gnn_class(gnn_type: str):
    if gnn_type not in GNN_TYPES:
        raise ValueError(f"Unknown GNN type '{gnn_type}'. Available types: {list(GNN_TYPES.keys())}")
    return GNN_TYPES[gnn_type]


class GNNSequential(nn.Module):
    """Stack of same-edge-set layers ``(mesh, mesh, edge) -> (mesh, edge)``; child
    names ``module_{i}`` follow ``pyg.nn.Sequential`` so checkpoint keys match."""

    def __init__(self, layers):
        super().__init__()
        self._n = len(layers)
        for i, layer in enumerate(layers):
            self.add_module(f"module_{i}", layer)

    def forward(self, mesh_rep, edge_rep, need_last_edges: bool = True):
        for i in range(self._n):
            layer = getattr(self, f"module_{i}")
            last = i == self._n - 1
            if last and not need_last_edges:
                mesh_rep, edge_rep = layer(mesh_rep, mesh_rep, edge_rep, need_edges=False)
            else:
                mesh_rep, edge_rep = layer(mesh_rep, mesh_rep, edge_rep)
        return mesh_rep, edge_rep


def make_gnn_seq(edge_index, num_gnn_layers, hidden_layers, hidden_dim, gnn_type="InteractionNet"):
    """utils/networks.py:43-106."""
    if num_gnn_layers < 1:
        raise ValueError(
            f"make_gnn_seq requires num_gnn_layers >= 1 (got {num_gnn_layers}); skip the stage for a no-op."
        )
    cls = get_gnn_class(gnn_type)
    return GNNSequential([cls(edge_index, hidden_dim, hidden_layers=hidden_layers) for _ in range(num_gnn_layers)])
