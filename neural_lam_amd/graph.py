"""Graph assets for the message-passing hot path: generation, on-disk format, CSR.

Host-side logic (numpy/scipy/torch CPU); nothing here touches the GPU.

* ``create_regular_grid_graph`` restates the *algorithm* of the reference's
  offline generator ``neural_lam/create_graph.py:356-862`` for regular grids in
  vectorised numpy (the reference walks networkx graphs edge by edge and queries
  three KD-trees; here the neighbour searches are closed-form index windows on
  the lattices, ``knn_lattice`` / ``ball_lattice``); the edge *sets* and features
  are identical, the edge *order* is sender-major with ascending receivers (the
  reference's order is networkx insertion order; the model is invariant to a
  consistent permutation of edges).
* ``save_graph`` / ``load_graph`` speak the reference's graph storage spec
  v0.1.0 (``docs/graph_storage_spec.md``; loader semantics of
  ``neural_lam/utils/graph.py:146-422``): int64 ``[2,E]`` edge indices that are
  zero-based per node set, float32 ``[E,3] = [len, vdiff_x, vdiff_y]`` features,
  lists per level, ``metainfo.yaml``.
* ``EdgeCSR`` is the MI355X-side layout: receiver-sorted (CSR) edge order for
  coalesced segment reduction, plus the sender-sorted (CSC) view the backward
  pass needs, all int32.
"""
from __future__ import annotations

import math
import os
import warnings
from dataclasses import dataclass
from pathlib import Path

import numpy as np
import torch
import yaml
from scipy.spatial import KDTree  # the reference's class: decides what the closed-form lattice searches cannot (ties, irregular coordinates)

GRAPH_SPEC_VERSION = "0.1.0"  # create_graph.py:24
METAINFO_FILENAME = "metainfo.yaml"  # create_graph.py:23
LEGACY_GRAPH_SPEC_VERSION = "legacy"   # utils/graph.py:18
DM_SCALE = 0.67  # create_graph.py:698
MESH_CHILDREN = 3  # create_graph.py:436 (nx)


# --------------------------------------------------------------------------
# generation
# --------------------------------------------------------------------------
def _mesh_level_positions(xy: np.ndarray, n: int) -> np.ndarray:
    """Cell-centre node positions of an n x n mesh level (create_graph.py:296-309)."""
    xm, xM = np.amin(xy[:, :, 0][:, 0]), np.amax(xy[:, :, 0][:, 0])
    ym, yM = np.amin(xy[:, :, 1][0, :]), np.amax(xy[:, :, 1][0, :])
    dx = (xM - xm) / n
    dy = (yM - ym) / n
    lx = np.linspace(xm + dx / 2, xM - dx / 2, n)
    ly = np.linspace(ym + dy / 2, yM - dy / 2, n)
    gx, gy = np.meshgrid(lx, ly, indexing="ij")
    return np.stack([gx, gy], axis=-1)  # (n, n, 2)


_NEIGH8 = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]


def _level_edges(pos: np.ndarray):
    """8-neighbour bidirectional edges of one n x n level (create_graph.py:306-329).

    Returns (send_ij, rec_ij, length, vdiff) with vdiff = pos[sender]-pos[receiver].
    """
    n = pos.shape[0]
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    send, rec = [], []
    for di, dj in _NEIGH8:
        ri, rj = ii + di, jj + dj
        ok = (ri >= 0) & (ri < n) & (rj >= 0) & (rj < n)
        send.append(np.stack([ii[ok], jj[ok]], axis=-1))
        rec.append(np.stack([ri[ok], rj[ok]], axis=-1))
    send = np.concatenate(send)
    rec = np.concatenate(rec)
    vdiff = pos[send[:, 0], send[:, 1]] - pos[rec[:, 0], rec[:, 1]]
    length = np.sqrt(np.sum(vdiff**2, axis=1))
    return send, rec, length, vdiff


# --------------------------------------------------------------------------
# neighbour searches on regular lattices (no KD-tree)
#
# The reference builds three KD-trees per graph (create_graph.py:488, 731, 777).  On a regular grid / mesh lattice the
# candidates of a query point are known in closed form -- the lattice cells around it -- so the searches below evaluate a
# small index window per query, vectorised, with the same distance arithmetic as the tree ((dx*dx + dy*dy) in float64).
# Whatever the window cannot decide EXACTLY like the tree is handed to the tree for that query only: a k-th / (k+1)-th
# distance tie (the tree's traversal order breaks those), a point within rounding distance of the ball radius, a window
# that might be too small, coordinates that are not a lattice.  The edge sets therefore equal the reference's bit for bit;
# tests/test_graph.py compares against scipy.spatial.KDTree on anisotropic, offset and tie-rich lattices.
# --------------------------------------------------------------------------
def _lattice_axes(pts: np.ndarray, shape, rtol=1e-9):
    """(x0, dx, y0, dy) if ``pts`` ((n0 * n1, 2), index i * n1 + j) is a regular lattice pts[i, j] = (x0 + i dx, y0 + j dy)
    to within ``rtol`` of its extent (mesh positions carry linspace rounding, coarser-level overrides differ in the last
    bits), else None."""
    n0, n1 = shape
    if n0 < 2 or n1 < 2 or pts.shape[0] != n0 * n1:
        return None
    P = pts.reshape(n0, n1, 2)
    x0, y0 = P[0, 0]
    dx, dy = (P[-1, 0, 0] - x0) / (n0 - 1), (P[0, -1, 1] - y0) / (n1 - 1)
    if dx <= 0 or dy <= 0:
        return None
    ii, jj = np.meshgrid(np.arange(n0), np.arange(n1), indexing="ij")
    tol = rtol * max(abs(dx) * n0, abs(dy) * n1)
    if np.abs(P[..., 0] - (x0 + ii * dx)).max() > tol or np.abs(P[..., 1] - (y0 + jj * dy)).max() > tol:
        return None
    return float(x0), float(dx), float(y0), float(dy)


def _window_candidates(query, axes, shape, half):
    """Lattice indices (flat, -1 where outside) of the (2 half + 2)^2 window around each query point."""
    x0, dx, y0, dy = axes
    n0, n1 = shape
    a0 = np.floor((query[:, 0] - x0) / dx).astype(np.int64)
    b0 = np.floor((query[:, 1] - y0) / dy).astype(np.int64)
    off = np.arange(-half, half + 2)
    a = a0[:, None, None] + off[None, :, None]
    b = b0[:, None, None] + off[None, None, :]
    ok = (a >= 0) & (a < n0) & (b >= 0) & (b < n1)
    flat = np.where(ok, a * n1 + b, -1).reshape(query.shape[0], -1)
    # distance from the query to the nearest point NOT covered by the window (outside lattice bounds nothing is missing)
    lo_a, hi_a = a0 - half, a0 + half + 1
    lo_b, hi_b = b0 - half, b0 + half + 1
    big = np.inf
    gap = np.minimum.reduce([
        np.where(lo_a > 0, query[:, 0] - (x0 + (lo_a - 1) * dx), big), np.where(hi_a < n0 - 1, (x0 + (hi_a + 1) * dx) - query[:, 0], big),
        np.where(lo_b > 0, query[:, 1] - (y0 + (lo_b - 1) * dy), big), np.where(hi_b < n1 - 1, (y0 + (hi_b + 1) * dy) - query[:, 1], big),
    ])
    return flat, gap


def knn_lattice(points: np.ndarray, shape, query: np.ndarray, k: int, half: int = 1):
    """Indices (n_query, k) of the k nearest ``points`` (a regular lattice of ``shape``) of every query point: the set
    scipy's ``KDTree(points).query(query, k)`` returns.  Falls back to the tree for queries it cannot decide identically."""
    axes = _lattice_axes(points, shape)
    if axes is None or points.shape[0] < k:
        return KDTree(points).query(query, k)[1].reshape(query.shape[0], k)
    flat, gap = _window_candidates(query, axes, shape, half)
    cand = points[np.maximum(flat, 0)]
    d0, d1 = cand[..., 0] - query[:, None, 0], cand[..., 1] - query[:, None, 1]
    dist2 = np.where(flat >= 0, d0 * d0 + d1 * d1, np.inf)
    kk = min(k + 1, dist2.shape[1])
    part = np.argpartition(dist2, kk - 1, axis=1)[:, :kk]                       # the k + 1 smallest, unordered
    order = np.take_along_axis(part, np.argsort(np.take_along_axis(dist2, part, axis=1), axis=1, kind="stable"), axis=1)
    dsort = np.take_along_axis(dist2, order, axis=1)
    out = np.take_along_axis(flat, order[:, :k], axis=1)
    # undecidable here: fewer than k candidates, a tie between the k-th and the (k+1)-th, or a k-th neighbour farther away than
    # the nearest lattice line outside the window (relative slack: the tree's distances carry their own rounding)
    kth = dsort[:, k - 1]
    nxt = dsort[:, k] if dsort.shape[1] > k else np.full_like(kth, np.inf)
    with np.errstate(invalid="ignore"):   # inf - inf where a window holds fewer than k + 1 lattice points
        bad = ~np.isfinite(kth) | ~(nxt - kth > 1e-9 * np.maximum(kth, 1e-300)) | (np.sqrt(kth) * (1 + 1e-9) >= gap)
    if bad.any():
        out[bad] = KDTree(points).query(query[bad], k)[1].reshape(int(bad.sum()), k)
    return out


def ball_lattice(points: np.ndarray, shape, query: np.ndarray, radius: float):
    """(query index, point index) pairs with |point - query| <= radius for lattice ``points``: the pairs scipy's
    ``KDTree(points).query_ball_point(query, radius)`` lists.  Queries with a point within rounding distance of the sphere
    are decided by the tree."""
    axes = _lattice_axes(points, shape)
    if axes is None:
        neigh = KDTree(points).query_ball_point(query, radius)
        return np.repeat(np.arange(query.shape[0]), [len(x) for x in neigh]), np.concatenate([np.asarray(x, dtype=np.int64) for x in neigh])
    half = int(math.ceil(radius / min(axes[1], axes[3]))) + 1
    flat, _ = _window_candidates(query, axes, shape, half)
    cand = points[np.maximum(flat, 0)]
    d0, d1 = cand[..., 0] - query[:, None, 0], cand[..., 1] - query[:, None, 1]
    dist = np.sqrt(np.where(flat >= 0, d0 * d0 + d1 * d1, np.inf))
    inside = dist <= radius
    edge_case = (np.abs(dist - radius) <= 1e-9 * radius).any(axis=1)
    qi, ci = np.nonzero(inside & ~edge_case[:, None])
    q_all, p_all = [qi], [flat[qi, ci]]
    if edge_case.any():
        rows = np.nonzero(edge_case)[0]
        neigh = KDTree(points).query_ball_point(query[rows], radius)
        q_all.append(np.repeat(rows, [len(x) for x in neigh]))
        p_all.append(np.concatenate([np.asarray(x, dtype=np.int64) for x in neigh]) if len(rows) else np.zeros(0, np.int64))
    return np.concatenate(q_all), np.concatenate(p_all).astype(np.int64)


def _sender_major(send_idx, rec_idx, length, vdiff):
    order = np.lexsort((rec_idx, send_idx))
    ei = torch.from_numpy(np.stack([send_idx[order], rec_idx[order]]).astype(np.int64))
    feat = torch.from_numpy(
        np.concatenate([length[order, None], vdiff[order]], axis=1).astype(np.float32)
    )
    return ei, feat


def mesh_level_sizes(nx_grid: int, ny_grid: int, n_max_levels: int | None = None):
    """Side lengths of the mesh levels (create_graph.py:436-453)."""
    nlev = int(np.log(max(nx_grid, ny_grid)) / np.log(MESH_CHILDREN))
    nleaf = MESH_CHILDREN**nlev
    mesh_levels = nlev - 1
    if n_max_levels:
        mesh_levels = min(mesh_levels, n_max_levels)
    return [int(nleaf / (MESH_CHILDREN**lev)) for lev in range(1, mesh_levels + 1)]


def create_regular_grid_graph(
    xy: np.ndarray, n_max_levels: int | None = None, hierarchical: bool = False
) -> dict:
    """Build all graph components for grid coordinates ``xy`` of shape (Nx, Ny, 2).

    Returns the *raw* (un-normalised, on-disk) tensors keyed like the files the
    reference writes (create_graph.py:366-410).
    """
    xy = np.asarray(xy, dtype=np.float64)
    Nx, Ny = xy.shape[:2]
    sizes = mesh_level_sizes(Nx, Ny, n_max_levels)
    if not sizes:
        raise ValueError(f"grid {Nx}x{Ny} too small for a mesh (needs >= 9 points on a side)")
    pos_levels = [_mesh_level_positions(xy, n) for n in sizes]
    out: dict = {}

    if hierarchical:
        m2m_ei, m2m_feat, mesh_pos = [], [], []
        for pos in pos_levels:
            n = pos.shape[0]
            s, r, length, vd = _level_edges(pos)
            ei, feat = _sender_major(s[:, 0] * n + s[:, 1], r[:, 0] * n + r[:, 1], length, vd)
            m2m_ei.append(ei)
            m2m_feat.append(feat)
            mesh_pos.append(torch.from_numpy(pos.reshape(-1, 2).astype(np.float32)))
        up_ei, up_feat, down_ei, down_feat = [], [], [], []
        for lower, upper in zip(pos_levels[:-1], pos_levels[1:]):
            lo = lower.reshape(-1, 2)
            up = upper.reshape(-1, 2)
            # each lower node -> its single nearest upper node (create_graph.py:488-509)
            nearest = knn_lattice(up, upper.shape[:2], lo, 1)[:, 0]
            vd = lo - up[nearest]
            length = np.sqrt(np.sum(vd**2, axis=1))
            ei, feat = _sender_major(np.arange(lo.shape[0]), nearest, length, vd)
            up_ei.append(ei)
            up_feat.append(feat)
            # down = reversed up with negated vdiff (create_graph.py:573-577)
            down_ei.append(torch.stack((ei[1], ei[0]), dim=0))
            dfeat = feat.clone()
            dfeat[:, 1:] = -dfeat[:, 1:]
            down_feat.append(dfeat)
        out.update(
            mesh_up_edge_index=up_ei,
            mesh_up_features=up_feat,
            mesh_down_edge_index=down_ei,
            mesh_down_features=down_feat,
        )
        bottom_pos = pos_levels[0].reshape(-1, 2)
    else:
        # multiscale: coarser levels live on the fine nodes [1::3, 1::3]
        # (create_graph.py:646-659); node id = i * n0 + j after sorting (:667-669)
        n0 = sizes[0]
        s_all, r_all, len_all, vd_all = [], [], [], []
        node_pos = pos_levels[0].copy()
        for lev, pos in enumerate(pos_levels):
            stride = MESH_CHILDREN**lev
            off = (stride - 1) // 2
            s, r, length, vd = _level_edges(pos)
            s_all.append((off + stride * s[:, 0]) * n0 + off + stride * s[:, 1])
            r_all.append((off + stride * r[:, 0]) * n0 + off + stride * r[:, 1])
            len_all.append(length)
            vd_all.append(vd)
            if lev > 0:
                # networkx.compose lets the coarser level's "pos" win (:659)
                n = pos.shape[0]
                idx = off + stride * np.arange(n)
                node_pos[np.ix_(idx, idx)] = pos
        ei, feat = _sender_major(
            np.concatenate(s_all), np.concatenate(r_all), np.concatenate(len_all), np.concatenate(vd_all)
        )
        m2m_ei, m2m_feat = [ei], [feat]
        mesh_pos = [torch.from_numpy(node_pos.reshape(-1, 2).astype(np.float32))]
        bottom_pos = node_pos.reshape(-1, 2)

    out.update(m2m_edge_index=m2m_ei, m2m_features=m2m_feat, mesh_features=mesh_pos)

    # ---- grid2mesh: grid nodes within 0.67*dm of each bottom mesh node (:698-758)
    n0 = sizes[0]
    p00 = bottom_pos[0]  # node (0, 0, 0)
    p10 = bottom_pos[n0]  # node (0, 1, 0)
    dm = np.sqrt(np.sum((p10 - p00) ** 2))
    grid_pos = xy.reshape(-1, 2)  # flat index i * Ny + j
    rec, send = ball_lattice(grid_pos, (Nx, Ny), bottom_pos, dm * DM_SCALE)
    vd = grid_pos[send] - bottom_pos[rec]
    out["g2m_edge_index"], out["g2m_features"] = _sender_major(
        send, rec, np.sqrt(np.sum(vd**2, axis=1)), vd
    )

    # ---- mesh2grid: 4 nearest bottom mesh nodes of each grid node (:780-793)
    nn4 = knn_lattice(bottom_pos, (n0, n0), grid_pos, 4)
    rec = np.repeat(np.arange(grid_pos.shape[0]), 4)
    send = nn4.reshape(-1)
    vd = bottom_pos[send] - grid_pos[rec]
    out["m2g_edge_index"], out["m2g_features"] = _sender_major(
        send, rec, np.sqrt(np.sum(vd**2, axis=1)), vd
    )
    return out


def regular_grid_xy(nx_grid: int, ny_grid: int, spacing: float = 2500.0) -> np.ndarray:
    """xy[i, j] = (spacing*i, spacing*j): the synthetic MEPS-shaped grid."""
    ii, jj = np.meshgrid(np.arange(nx_grid), np.arange(ny_grid), indexing="ij")
    return np.stack([ii * spacing, jj * spacing], axis=-1).astype(np.float64)


# --------------------------------------------------------------------------
# on-disk format
# --------------------------------------------------------------------------
_LIST_KEYS = ("m2m", "mesh_up", "mesh_down")


def save_graph(graph_dir: str | os.PathLike, raw: dict, with_layouts: bool = False) -> None:
    """Write the reference's file set (create_graph.py:132-166, 688-690, 857-862); ``with_layouts`` adds ``edge_layouts.pt``
    (the MI355X-side CSR / CSC / tile layouts of every edge set, see ``write_edge_layouts``)."""
    graph_dir = Path(graph_dir)
    graph_dir.mkdir(parents=True, exist_ok=True)
    for name in ("g2m", "m2g"):
        torch.save(raw[f"{name}_edge_index"], graph_dir / f"{name}_edge_index.pt")
        torch.save(raw[f"{name}_features"], graph_dir / f"{name}_features.pt")
    for name in _LIST_KEYS:
        if f"{name}_edge_index" in raw:
            torch.save(list(raw[f"{name}_edge_index"]), graph_dir / f"{name}_edge_index.pt")
            torch.save(list(raw[f"{name}_features"]), graph_dir / f"{name}_features.pt")
    torch.save(list(raw["mesh_features"]), graph_dir / "mesh_features.pt")
    with open(graph_dir / METAINFO_FILENAME, "w", encoding="utf-8") as fp:
        yaml.dump({"spec_version": GRAPH_SPEC_VERSION}, fp)
    if with_layouts:
        write_edge_layouts(graph_dir)


def read_graph_files(graph_dir: str | os.PathLike) -> dict:
    graph_dir = Path(graph_dir)

    def ld(fn):
        return torch.load(graph_dir / fn, map_location="cpu", weights_only=True)

    raw = {
        "mesh_features": ld("mesh_features.pt"),
        "m2m_edge_index": ld("m2m_edge_index.pt"),
        "m2m_features": ld("m2m_features.pt"),
        "g2m_edge_index": ld("g2m_edge_index.pt"),
        "g2m_features": ld("g2m_features.pt"),
        "m2g_edge_index": ld("m2g_edge_index.pt"),
        "m2g_features": ld("m2g_features.pt"),
    }
    if (graph_dir / "mesh_up_edge_index.pt").exists():
        for name in ("mesh_up", "mesh_down"):
            raw[f"{name}_edge_index"] = ld(f"{name}_edge_index.pt")
            raw[f"{name}_features"] = ld(f"{name}_features.pt")
    meta = graph_dir / METAINFO_FILENAME
    if not meta.exists():
        # utils/graph.py:239-252: no metainfo = the legacy pre-spec format
        warnings.warn(
            "Graph metainfo file is missing; assuming this graph uses the legacy pre-spec format. Mesh node feature "
            "normalization will be skipped because legacy mesh node features are assumed to already be normalized. Edge "
            "indices will be zero-offset on load to convert legacy offset node labels to the per-node-set zero-based index "
            "spaces required by the current graph spec.",
            RuntimeWarning, stacklevel=3,
        )
        raw["spec_version"] = LEGACY_GRAPH_SPEC_VERSION
        return raw
    try:
        parsed = yaml.safe_load(meta.read_text(encoding="utf-8"))
    except yaml.YAMLError as exc:
        raise ValueError(f"Failed to parse {METAINFO_FILENAME}: {exc}") from exc
    spec = None if parsed is None else parsed.get("spec_version")
    if spec is None:
        raise ValueError(f"{METAINFO_FILENAME} is missing 'spec_version' entry")
    if spec != GRAPH_SPEC_VERSION:
        raise ValueError(f"Unsupported graph spec version {spec!r} in {METAINFO_FILENAME}")
    return raw


def zero_index_legacy(raw: dict) -> dict:
    """Legacy (pre-spec) graphs label all nodes in ONE index space (mesh levels and grid offset against each other);
    the current spec wants every node set zero-based.  utils/graph.py:20-143 (``zero_index_edge_index``,
    ``zero_index_m2g``, ``zero_index_g2m``) and :305-323, :362-370 of ``load_graph``."""

    def zero(ei):   # both rows start at 0
        return ei - ei.min(dim=1, keepdim=True)[0]

    out = dict(raw)
    out["m2m_edge_index"] = [zero(e) for e in raw["m2m_edge_index"]]
    g2m, m2g = raw["g2m_edge_index"], raw["m2g_edge_index"]
    mins = m2g.min(dim=1, keepdim=True)[0]
    mesh_first = bool(mins[0] < mins[1])
    if mesh_first:   # grid labels were offset by the TOTAL mesh node count (all levels)
        n_mesh = sum(int(f.shape[0]) for f in raw["mesh_features"])
        out["g2m_edge_index"] = torch.stack((g2m[0] - n_mesh, g2m[1]), dim=0)
        out["m2g_edge_index"] = torch.stack((m2g[0], m2g[1] - n_mesh), dim=0)
    else:            # grid first: mesh labels were offset by the number of grid (g2m) / interior (m2g) nodes
        out["g2m_edge_index"] = torch.stack((g2m[0], g2m[1] - (g2m[0].max() + 1)), dim=0)
        out["m2g_edge_index"] = torch.stack((m2g[0] - (m2g[1].max() + 1), m2g[1]), dim=0)
    assert int(out["m2g_edge_index"].min()) >= 0, "Negative node index in m2g"
    assert int(out["g2m_edge_index"].min()) >= 0, "Negative node index in g2m"
    for name in ("mesh_up", "mesh_down"):
        if f"{name}_edge_index" in raw:
            out[f"{name}_edge_index"] = [zero(e) for e in raw[f"{name}_edge_index"]]
    return out


def normalise_graph(raw: dict, mesh_node_features_scaling: float):
    """Load-time normalisation of utils/graph.py:291-303 (mesh coords / max grid
    span) and :343-350 (edge features / longest m2m edge); flat graphs unwrap
    level 0 (:399-408).  Returns (hierarchical, dict of tensors / lists)."""
    legacy = raw.get("spec_version") == LEGACY_GRAPH_SPEC_VERSION
    if legacy:   # :286-323: legacy mesh features are already normalised; node labels become zero-based per node set
        raw = zero_index_legacy(raw)
    mesh = [m.clone().to(torch.float32) for m in raw["mesh_features"]]
    if not legacy:
        if mesh_node_features_scaling == 0:
            warnings.warn("Mesh node feature scaling is zero; falling back to 1.0 so mesh node coordinates are left unchanged "
                          "after graph loading.", RuntimeWarning, stacklevel=2)
            mesh_node_features_scaling = 1.0
        for m in mesh:
            m[:, :2] /= mesh_node_features_scaling
    m2m_ei = [e.clone() for e in raw["m2m_edge_index"]]
    hierarchical = len(m2m_ei) > 1
    longest = max(torch.max(f[:, 0]) for f in raw["m2m_features"])
    out = {
        "g2m_edge_index": raw["g2m_edge_index"].clone(),
        "m2g_edge_index": raw["m2g_edge_index"].clone(),
        "g2m_features": raw["g2m_features"] / longest,
        "m2g_features": raw["m2g_features"] / longest,
    }
    # BufferList.__itruediv__ multiplies by the reciprocal (utils/buffer_list.py),
    # while g2m/m2g use a true division (utils/graph.py:343-350): keep both.
    recip = 1.0 / longest
    m2m_feat = [f * recip for f in raw["m2m_features"]]
    assert len(m2m_feat) == len(m2m_ei) == len(mesh), "Inconsistent number of levels in mesh"
    if hierarchical:
        out.update(
            m2m_edge_index=m2m_ei,
            m2m_features=m2m_feat,
            mesh_static_features=mesh,
            mesh_up_edge_index=[e.clone() for e in raw["mesh_up_edge_index"]],
            mesh_down_edge_index=[e.clone() for e in raw["mesh_down_edge_index"]],
            mesh_up_features=[f * recip for f in raw["mesh_up_features"]],
            mesh_down_features=[f * recip for f in raw["mesh_down_features"]],
        )
    else:
        out.update(
            m2m_edge_index=m2m_ei[0],
            m2m_features=m2m_feat[0],
            mesh_static_features=mesh[0],
            mesh_up_edge_index=[],
            mesh_down_edge_index=[],
            mesh_up_features=[],
            mesh_down_features=[],
        )
    return hierarchical, out


def load_graph(graph_dir, mesh_node_features_scaling: float):
    """Mirror of ``utils.load_graph`` (utils/graph.py:146-422): spec-0.1.0 graphs and, with a RuntimeWarning, the legacy
    pre-spec format (no metainfo file: offset node labels, pre-normalised mesh coordinates).  If the directory also holds
    ``edge_layouts.pt`` (``save_graph(..., with_layouts=True)`` / ``write_edge_layouts``), the int32 CSR / CSC views and tile
    schedules of its edge sets are loaded with it, so the layers built on this graph skip the host-side sorting."""
    out = normalise_graph(read_graph_files(graph_dir), mesh_node_features_scaling)
    preload_edge_layouts(graph_dir)
    return out


# --------------------------------------------------------------------------
# MI355X-side edge layout
# --------------------------------------------------------------------------
@dataclass
class EdgeCSR:
    """Receiver-sorted edge list + sender-sorted view, int32, on one device.

    position p in [0, E) is the CSR (receiver-major, stable) position of original
    edge ``perm[p]``.  ``rowptr[r]:rowptr[r+1]`` are receiver r's positions.
    ``cperm`` lists CSR positions grouped by sender (``colptr`` delimits them),
    which turns the backward scatter-by-sender into another segment reduction.
    ``tiles`` partitions the receivers into runs of whole receivers whose edges
    fit one workgroup tile ("tile schedule").
    """

    num_send: int
    num_rec: int
    num_edges: int
    perm: torch.Tensor  # (E,) original edge id at CSR position p
    send: torch.Tensor  # (E,) sender of CSR position p
    rec: torch.Tensor  # (E,) receiver of CSR position p
    rowptr: torch.Tensor  # (N_r + 1,)
    colptr: torch.Tensor  # (N_s + 1,)
    cperm: torch.Tensor  # (E,) CSR positions sorted by sender (stable)
    inv_deg: torch.Tensor  # (N_r,) 1 / max(in_degree, 1)  float32
    max_in_degree: int
    # Receivers with more in-edges than a tile holds (build_tile_schedule, "virtual" split): the pieces of such a receiver are
    # tiles of their own that reduce into VIRTUAL segments behind the real ones, and ``nlam_split_combine`` adds the pieces
    # up in a fixed order -- no atomics anywhere.  None / 0 when the edge set has no such receiver.
    rowptr_ext: torch.Tensor | None = None    # (nseg_ext + 1,): rowptr, then the row ranges of the virtual segments
    inv_deg_ext: torch.Tensor | None = None   # (nseg_ext,): inv_deg, then the scale of each virtual segment's real receiver
    nseg_ext: int = 0                         # rows of an aggregation buffer: num_rec + 1 + virtual segments
    comb_ptr: torch.Tensor | None = None      # (n_split + 1,): pieces of split receiver s = comb_src[comb_ptr[s]:comb_ptr[s+1]]
    comb_src: torch.Tensor | None = None      # virtual segment ids, in CSR order
    comb_dst: torch.Tensor | None = None      # (n_split,): the real receiver

    def to(self, device):
        kw = {}
        for k, v in self.__dict__.items():
            kw[k] = v.to(device) if torch.is_tensor(v) else v
        return EdgeCSR(**kw)


def build_edge_csr(edge_index: torch.Tensor, num_send: int | None = None, num_rec: int | None = None) -> EdgeCSR:
    """edge_index: int64 (2, E), row 0 senders, row 1 receivers, zero-based per node set.

    ``num_rec`` defaults to ``edge_index[1].max() + 1`` exactly like
    ``InteractionNet.__init__`` (gnn_layers.py:73).
    """
    assert edge_index.dim() == 2 and edge_index.shape[0] == 2
    ei = edge_index.detach().cpu().to(torch.int64)
    E = ei.shape[1]
    if E == 0:
        raise ValueError("edge_index must contain at least one edge")
    if num_rec is None:
        num_rec = int(ei[1].max()) + 1
    if num_send is None:
        num_send = int(ei[0].max()) + 1
    if E >= 2**31 or num_send >= 2**31 or num_rec >= 2**31:
        raise ValueError("int32 index space exceeded")
    perm = torch.argsort(ei[1], stable=True)
    send = ei[0][perm]
    rec = ei[1][perm]
    deg = torch.bincount(rec, minlength=num_rec)
    rowptr = torch.zeros(num_rec + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(deg, 0)
    cperm = torch.argsort(send, stable=True)
    sdeg = torch.bincount(send, minlength=num_send)
    colptr = torch.zeros(num_send + 1, dtype=torch.int64)
    colptr[1:] = torch.cumsum(sdeg, 0)
    i32 = lambda t: t.to(torch.int32).contiguous()
    return EdgeCSR(
        num_send=num_send,
        num_rec=num_rec,
        num_edges=E,
        perm=i32(perm),
        send=i32(send),
        rec=i32(rec),
        rowptr=i32(rowptr),
        colptr=i32(colptr),
        cperm=i32(cperm),
        inv_deg=(1.0 / deg.clamp(min=1).to(torch.float32)).contiguous(),
        max_in_degree=int(deg.max()),
    )


def graph_summary(raw: dict) -> dict:
    """Sizes used in DESIGN.md / bench config strings."""
    s = {
        "mesh_nodes": [int(m.shape[0]) for m in raw["mesh_features"]],
        "m2m_edges": [int(e.shape[1]) for e in raw["m2m_edge_index"]],
        "g2m_edges": int(raw["g2m_edge_index"].shape[1]),
        "m2g_edges": int(raw["m2g_edge_index"].shape[1]),
    }
    if "mesh_up_edge_index" in raw:
        s["up_edges"] = [int(e.shape[1]) for e in raw["mesh_up_edge_index"]]
    return s


# --------------------------------------------------------------------------
# wave-tile schedule (consumed by csrc/nlam_hip.hip through nlam_tile_t)
# --------------------------------------------------------------------------
TILE_ROWS = 32  # columns of v_mfma_f32_32x32x2_f32 = rows (edges) one wave owns
TILE_SPLIT = 1 << 30


# Receivers with more than TILE_ROWS in-edges: "virtual" = deterministic two-pass reduction (default: the reference trains with
# deterministic=True, train_model.py:566, on whatever graph create_graph.py emits); False = the one-pass NLAM_TILE_SPLIT tiles of
# the C-ABI, whose partial sums meet through atomic adds (kept for ABI users and covered by a test).
VIRTUAL_SPLIT = True


def build_tile_schedule(rowptr: torch.Tensor, tile_rows: int = TILE_ROWS, virtual_split: bool | None = None):
    """Partition the receivers into tiles of whole receivers with <= ``tile_rows``
    edges (and <= ``tile_rows`` receivers).  A receiver with more in-edges than a
    tile holds is cut into pieces of <= ``tile_rows`` edges:

    * ``virtual_split`` (default): every piece is an ordinary one-receiver tile whose segment id is a VIRTUAL receiver
      behind the real ones -- ids ``num_rec + 1 ..``; ``split["rowptr_ext"]`` continues ``rowptr`` with the pieces' row
      ranges (one terminating entry per split receiver, a gap segment no tile owns) -- so the kernels reduce it with plain
      stores like any tile, and ``nlam_split_combine`` then sums a receiver's pieces in CSR order into its real row.
      Deterministic; no kernel knows about it.
    * otherwise the pieces are flagged ``TILE_SPLIT`` and the kernels add their partial sums atomically into a zeroed output.

    Returns (int32 tensor (ntiles, 4) = [row0, nrows, seg0, nseg|flag], has_split, split) -- ``has_split`` is True only for
    flagged tiles; ``split`` is None or a dict(rowptr_ext, seg_real, nseg_ext, comb_ptr, comb_src, comb_dst) of int32 tensors.
    """
    if virtual_split is None:
        virtual_split = VIRTUAL_SPLIT
    rp = rowptr.detach().cpu().numpy().astype(np.int64)
    nrec = rp.shape[0] - 1
    tiles = []
    has_split = False
    cur_r0, cur_e0, cur_ne, cur_nr = 0, 0, 0, 0
    ext_ptr, ext_real, comb_ptr, comb_src, comb_dst = [], [], [0], [], []

    def flush():
        nonlocal cur_ne, cur_nr
        if cur_nr > 0:
            tiles.append((cur_e0, cur_ne, cur_r0, cur_nr))
        cur_ne, cur_nr = 0, 0

    for r in range(nrec):
        deg = int(rp[r + 1] - rp[r])
        if deg > tile_rows:
            flush()
            if virtual_split:
                for e in range(int(rp[r]), int(rp[r + 1]), tile_rows):
                    v = nrec + 1 + len(ext_ptr)          # rowptr_ext[v] = e, rowptr_ext[v + 1] = start of the next piece / the end
                    tiles.append((e, min(tile_rows, int(rp[r + 1]) - e), v, 1))
                    ext_ptr.append(e)
                    ext_real.append(r)
                    comb_src.append(v)
                ext_ptr.append(int(rp[r + 1]))           # terminates the last piece; as a segment it is a gap nothing writes or reads
                ext_real.append(r)
                comb_ptr.append(len(comb_src))
                comb_dst.append(r)
            else:
                has_split = True
                for e in range(int(rp[r]), int(rp[r + 1]), tile_rows):
                    tiles.append((e, min(tile_rows, int(rp[r + 1]) - e), r, 1 | TILE_SPLIT))
            cur_r0, cur_e0 = r + 1, int(rp[r + 1])
            continue
        if cur_nr == 0:
            cur_r0, cur_e0 = r, int(rp[r])
        elif cur_ne + deg > tile_rows or cur_nr == tile_rows:
            flush()
            cur_r0, cur_e0 = r, int(rp[r])
        cur_ne += deg
        cur_nr += 1
    flush()
    t = torch.tensor(tiles, dtype=torch.int64).reshape(-1, 4).to(torch.int32)
    split = None
    if comb_dst:
        i32 = lambda v: torch.tensor(v, dtype=torch.int32)
        nseg_ext = nrec + 1 + len(ext_ptr)
        # nseg_ext + 1 entries: the real row pointers (nrec + 1), the virtual ones, one terminator for the last gap segment
        rowptr_ext = torch.cat([rowptr.detach().cpu().to(torch.int32), i32(ext_ptr), i32([int(rp[nrec])])])
        split = dict(rowptr_ext=rowptr_ext.contiguous(), seg_real=i32(ext_real), nseg_ext=nseg_ext,
                     comb_ptr=i32(comb_ptr), comb_src=i32(comb_src), comb_dst=i32(comb_dst))
    return t.contiguous(), has_split, split


def attach_split(csr: EdgeCSR, split) -> EdgeCSR:
    """Store a virtual-split plan (build_tile_schedule) on the layout: extended row pointers / scales and the combine lists."""
    if split is None:
        return csr
    csr.rowptr_ext = split["rowptr_ext"]
    real = split["seg_real"].long()
    csr.inv_deg_ext = torch.cat([csr.inv_deg, torch.ones(1, dtype=torch.float32), csr.inv_deg[real]]).contiguous()
    csr.nseg_ext = int(split["nseg_ext"])
    csr.comb_ptr, csr.comb_src, csr.comb_dst = split["comb_ptr"], split["comb_src"], split["comb_dst"]
    return csr


# --------------------------------------------------------------------------
# edge layouts: built once per distinct edge set, optionally stored beside the graph files
#
# Every InteractionNet needs the CSR / CSC views and the tile schedule of its edge set.  A model builds many layers on the
# same edge index (the 4-8 processor layers on m2m, Hi-LAM's per-level stacks), and a training run re-reads the same graph
# directory every time: the layout is therefore cached by CONTENT (sha1 of the int64 edge index + the node counts) and can be
# written next to the reference's files (``edge_layouts.pt``: a file the reference's loader does not look for, so the
# directory stays a valid spec-v0.1.0 graph) -- ``load_graph`` then hands the layers ready-made int32 CSR / CSC / tiles.
# --------------------------------------------------------------------------
EDGE_LAYOUT_FILENAME = "edge_layouts.pt"
_LAYOUT_CACHE: dict = {}
_LAYOUT_FIELDS = ("perm", "send", "rec", "rowptr", "colptr", "cperm", "inv_deg")
_SPLIT_FIELDS = ("rowptr_ext", "inv_deg_ext", "comb_ptr", "comb_src", "comb_dst")   # present when a receiver exceeds a tile


def edge_layout_key(edge_index: torch.Tensor, num_send: int, num_rec: int) -> str:
    import hashlib

    ei = edge_index.detach().cpu().to(torch.int64).contiguous()
    h = hashlib.sha1(ei.numpy().tobytes())
    h.update(f"{tuple(ei.shape)}|{int(num_send)}|{int(num_rec)}|{TILE_ROWS}|split={'virtual' if VIRTUAL_SPLIT else 'atomic'}".encode())
    return h.hexdigest()


def edge_layout(edge_index: torch.Tensor, num_send: int | None = None, num_rec: int | None = None):
    """(EdgeCSR on the host, tiles (ntiles, 4) int32, has_split) of an edge set: from the content cache (filled by earlier
    layers on the same edges or by ``load_graph`` from ``edge_layouts.pt``) or built now."""
    ei = edge_index.detach().cpu().to(torch.int64)
    if num_rec is None:
        num_rec = int(ei[1].max()) + 1
    if num_send is None:
        num_send = int(ei[0].max()) + 1
    key = edge_layout_key(ei, num_send, num_rec)
    hit = _LAYOUT_CACHE.get(key)
    if hit is None:
        csr = build_edge_csr(ei, num_send=num_send, num_rec=num_rec)
        tiles, has_split, split = build_tile_schedule(csr.rowptr)
        attach_split(csr, split)
        hit = _LAYOUT_CACHE[key] = (csr, tiles, has_split)
    return hit


def _layout_to_record(csr: EdgeCSR, tiles: torch.Tensor, has_split: bool) -> dict:
    rec = {f: getattr(csr, f) for f in _LAYOUT_FIELDS}
    rec.update(tiles=tiles, has_split=bool(has_split), num_send=csr.num_send, num_rec=csr.num_rec, num_edges=csr.num_edges,
               max_in_degree=csr.max_in_degree)
    if csr.rowptr_ext is not None:
        rec.update({f: getattr(csr, f) for f in _SPLIT_FIELDS}, nseg_ext=int(csr.nseg_ext))
    return rec


def _record_to_layout(rec: dict):
    csr = EdgeCSR(num_send=int(rec["num_send"]), num_rec=int(rec["num_rec"]), num_edges=int(rec["num_edges"]),
                  max_in_degree=int(rec["max_in_degree"]), **{f: rec[f] for f in _LAYOUT_FIELDS})
    if "rowptr_ext" in rec:
        for f in _SPLIT_FIELDS:
            setattr(csr, f, rec[f])
        csr.nseg_ext = int(rec["nseg_ext"])
    return csr, rec["tiles"], bool(rec["has_split"])


def _graph_edge_sets(tensors: dict):
    """(edge index, num_send, num_rec) of every edge set of a loaded graph, with the node counts the model's layers use
    (InteractionNet: num_rec = max receiver + 1, senders = rows of the sender tensor = max sender + 1 for these graphs)."""
    out = []
    for name in ("g2m_edge_index", "m2g_edge_index", "m2m_edge_index", "mesh_up_edge_index", "mesh_down_edge_index"):
        v = tensors.get(name)
        if v is None:
            continue
        for ei in (v if isinstance(v, (list, tuple)) else [v]):
            if torch.is_tensor(ei) and ei.numel() > 0:
                out.append((ei, int(ei[0].max()) + 1, int(ei[1].max()) + 1))
    return out


def write_edge_layouts(graph_dir, mesh_node_features_scaling: float = 1.0) -> Path:
    """Build the layout of every edge set of the graph stored in ``graph_dir`` and write them to ``edge_layouts.pt``."""
    graph_dir = Path(graph_dir)
    _, tensors = normalise_graph(read_graph_files(graph_dir), mesh_node_features_scaling)
    records = {}
    for ei, ns, nr in _graph_edge_sets(tensors):
        key = edge_layout_key(ei, ns, nr)
        if key not in records:
            records[key] = _layout_to_record(*edge_layout(ei, ns, nr))
    path = graph_dir / EDGE_LAYOUT_FILENAME
    torch.save({"tile_rows": TILE_ROWS, "layouts": records}, path)
    return path


def preload_edge_layouts(graph_dir) -> int:
    """Put the layouts stored beside a graph into the content cache (they are looked up by the hash of the edge index a layer
    is built with, so a stale or foreign file is simply never hit).  Returns the number of layouts read."""
    path = Path(graph_dir) / EDGE_LAYOUT_FILENAME
    if not path.exists():
        return 0
    blob = torch.load(path, map_location="cpu", weights_only=True)
    if int(blob.get("tile_rows", -1)) != TILE_ROWS:
        return 0
    n = 0
    for key, rec in blob["layouts"].items():
        if key not in _LAYOUT_CACHE:
            _LAYOUT_CACHE[key] = _record_to_layout(rec)
            n += 1
    return n
