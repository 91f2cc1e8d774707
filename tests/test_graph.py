"""Graph generator / loader / CSR host logic against the reference's own
create_graph + load_graph outputs stored in the golden fixtures.  CPU only."""
import numpy as np
import pytest
import torch

from conftest import graph_from_case, load_golden
from neural_lam_amd import graph as G
from neural_lam_amd.datastore import SyntheticDatastore


def _edge_dict(ei, feat):
    return {(int(s), int(r)): feat[k] for k, (s, r) in enumerate(ei.t().tolist())}


def _cmp_edges(ei_a, f_a, ei_b, f_b):
    da, db = _edge_dict(ei_a, f_a), _edge_dict(ei_b, f_b)
    assert da.keys() == db.keys()
    fa = torch.stack([da[k] for k in sorted(da)])
    fb = torch.stack([db[k] for k in sorted(da)])
    assert torch.allclose(fa, fb, rtol=1e-6, atol=1e-6 * float(fb.abs().max()))


@pytest.mark.parametrize("name", ["graphlam_30x27", "graphlam_30x27_variants", "hilam_81x30"])
def test_generator_matches_reference_create_graph(name, tmp_path):
    case = load_golden(name)
    ds = SyntheticDatastore(root_path=tmp_path, **case["ds_kwargs"])
    mine = G.create_regular_grid_graph(ds.get_xy("state"), **case["graph_kwargs"])
    ref = graph_from_case(case, "ref_graph_raw")
    names = ["g2m", "m2g"]
    for n in names:
        _cmp_edges(mine[f"{n}_edge_index"], mine[f"{n}_features"], ref[f"{n}_edge_index"], ref[f"{n}_features"])
    list_names = ["m2m"] + (["mesh_up", "mesh_down"] if case["graph_kwargs"]["hierarchical"] else [])
    for n in list_names:
        assert len(mine[f"{n}_edge_index"]) == len(ref[f"{n}_edge_index"])
        for l in range(len(ref[f"{n}_edge_index"])):
            _cmp_edges(mine[f"{n}_edge_index"][l], mine[f"{n}_features"][l], ref[f"{n}_edge_index"][l], ref[f"{n}_features"][l])
    for a, b in zip(mine["mesh_features"], ref["mesh_features"]):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-3)


@pytest.mark.parametrize("name", ["graphlam_30x27", "hilam_81x30"])
def test_save_load_roundtrip_matches_reference_load_graph(name, tmp_path):
    """Reference-written raw tensors -> our save_graph/load_graph == the tensors the
    reference's own utils.load_graph registered on its model."""
    case = load_golden(name)
    ds = SyntheticDatastore(root_path=tmp_path, **case["ds_kwargs"])
    raw = graph_from_case(case, "ref_graph_raw")
    G.save_graph(tmp_path / "graph" / "g", raw)
    ext = ds.get_xy_extent("state")
    span = max(ext[1] - ext[0], ext[3] - ext[2])  # graph/base.py:113-117
    hier, loaded = G.load_graph(tmp_path / "graph" / "g", span)
    assert hier == case["ref_hierarchical"]
    ref = graph_from_case(case, "ref_graph_loaded")
    for k, v in ref.items():
        mine = loaded[k]
        if isinstance(v, list):
            assert len(mine) == len(v), k
            for a, b in zip(mine, v):
                assert torch.equal(a, b), k  # bit-exact: same fp32 ops as the reference loader
        else:
            assert torch.equal(mine, v), k


def test_meps_sizes_match_reference_generator():
    """The reference's generator on the 238x268 grid (recorded once by make_golden.py --meps)."""
    ref = load_golden("meps_graph_sizes")["ref_meps_graph_sizes"]
    xy = G.regular_grid_xy(238, 268)
    assert G.graph_summary(G.create_regular_grid_graph(xy)) == ref["multiscale"]
    assert G.graph_summary(G.create_regular_grid_graph(xy, n_max_levels=3, hierarchical=True)) == ref["hierarchical"]


@pytest.mark.parametrize("name", ["hi_mesh_first", "flat_grid_first"])
def test_legacy_graph_loads_like_the_reference_loader(name, tmp_path):
    """Pre-spec graph directories (one node index space, no metainfo.yaml): zero-based per node set, mesh coordinates
    left alone, RuntimeWarning -- against what the reference's own utils.load_graph returned for the same files
    (tests/golden/make_golden.py::legacy_graph_cases; utils/graph.py:20-143, :239-252, :286-323, :362-370)."""
    from conftest import to64

    case = load_golden("legacy_graphs")["cases"][name]
    for k, v in case["legacy_files"].items():
        torch.save([to64(x) for x in v] if isinstance(v, list) else to64(v), tmp_path / f"{k}.pt")
    with pytest.warns(RuntimeWarning, match="legacy pre-spec format"):
        hier, loaded = G.load_graph(tmp_path, 123.0)   # the scaling is ignored for legacy graphs
    assert hier == case["ref_hierarchical"]
    for k, v in case["ref_graph_loaded"].items():
        mine = loaded[k]
        if isinstance(v, list):
            assert len(mine) == len(v), k
            for a, b in zip(mine, v):
                assert torch.equal(a, to64(b)), k
        else:
            assert torch.equal(mine, to64(v)), k
    for ei in [loaded["g2m_edge_index"], loaded["m2g_edge_index"]]:
        assert int(ei.min()) >= 0   # (not necessarily 0: a node set need not be fully connected, utils/graph.py:46)


def test_metainfo_without_spec_version_raises(tmp_path):
    raw = G.create_regular_grid_graph(G.regular_grid_xy(27, 27))
    G.save_graph(tmp_path, raw)
    (tmp_path / G.METAINFO_FILENAME).write_text("something_else: 1\n")
    with pytest.raises(ValueError, match="spec_version"):
        G.load_graph(tmp_path, 1.0)


def test_unsupported_spec_raises(tmp_path):
    raw = G.create_regular_grid_graph(G.regular_grid_xy(27, 27))
    G.save_graph(tmp_path, raw)
    (tmp_path / G.METAINFO_FILENAME).write_text("spec_version: '9.9'\n")
    with pytest.raises(ValueError):
        G.load_graph(tmp_path, 1.0)


def test_edge_csr_structure():
    g = torch.Generator().manual_seed(0)
    ns, nr, E = 13, 7, 60
    ei = torch.stack([torch.randint(0, ns, (E,), generator=g), torch.randint(0, nr - 1, (E,), generator=g)])
    ei[1][ei[1] == 2] = 3  # receiver 2 has no edges
    csr = G.build_edge_csr(ei, num_send=ns)
    assert csr.num_rec == int(ei[1].max()) + 1  # gnn_layers.py:73 semantics
    perm = csr.perm.long()
    assert torch.equal(csr.send.long(), ei[0][perm]) and torch.equal(csr.rec.long(), ei[1][perm])
    assert torch.all(csr.rec[1:] >= csr.rec[:-1])
    for r in range(csr.num_rec):
        seg = csr.rec[csr.rowptr[r] : csr.rowptr[r + 1]]
        assert torch.all(seg == r)
    assert csr.rowptr[3] - csr.rowptr[2] == 0
    # CSC view: positions grouped by sender
    cs = csr.send.long()[csr.cperm.long()]
    assert torch.all(cs[1:] >= cs[:-1])
    for s in range(ns):
        seg = cs[csr.colptr[s] : csr.colptr[s + 1]]
        assert torch.all(seg == s)
    deg = torch.bincount(ei[1], minlength=csr.num_rec).clamp(min=1).float()
    assert torch.allclose(csr.inv_deg, 1.0 / deg)


@pytest.mark.parametrize("n0,n1,dx,dy", [(81, 81, 7315.0, 8241.0), (9, 9, 1.0, 1.0), (30, 27, 2.0, 0.7), (5, 40, 3.0, 1.0), (2, 2, 1.0, 1.0)])
def test_lattice_neighbour_searches_equal_the_kd_tree(n0, n1, dx, dy):
    """graph.knn_lattice / graph.ball_lattice (closed-form index windows, no KD-tree) return exactly the neighbour SETS of
    scipy.spatial.KDTree -- the class create_graph.py:488, 731, 777 queries -- on anisotropic and offset lattices, for
    queries outside the lattice and for queries that sit exactly between lattice points (ties go to the tree)."""
    import numpy as np
    from scipy.spatial import KDTree

    from neural_lam_amd import graph as G

    def lattice(m0, m1, ddx, ddy):
        ii, jj = np.meshgrid(np.arange(m0), np.arange(m1), indexing="ij")
        return np.stack([0.3 + ii * ddx, -1.0 + jj * ddy], -1).reshape(-1, 2)

    rng = np.random.default_rng(n0 * 100 + n1)
    pts = lattice(n0, n1, dx, dy)
    q = np.concatenate([rng.uniform([-2 * dx, -2 * dy], [(n0 + 1) * dx, (n1 + 1) * dy], size=(3000, 2)),
                        lattice(2 * n0, 2 * n1, dx / 2, dy / 2)[:3000]])   # half-cell points: exact distance ties
    tree = KDTree(pts)
    for k in (1, 4):
        if k > pts.shape[0]:
            continue
        a, b = G.knn_lattice(pts, (n0, n1), q, k), tree.query(q, k)[1].reshape(len(q), k)
        assert (np.sort(a, 1) == np.sort(b, 1)).all()
    for r in (0.67 * dx, 1.5 * max(dx, dy), 0.5 * dx):   # 0.5 dx: lattice points exactly on the sphere
        qi, pi = G.ball_lattice(pts, (n0, n1), q, r)
        ref = tree.query_ball_point(q, r)
        assert set(zip(qi.tolist(), pi.tolist())) == {(i, j) for i, x in enumerate(ref) for j in x}
    # a non-lattice point set takes the tree itself
    jit = pts + rng.normal(size=pts.shape) * 0.2 * min(dx, dy)
    assert G._lattice_axes(jit, (n0, n1)) is None
    if jit.shape[0] >= 4:
        assert (np.sort(G.knn_lattice(jit, (n0, n1), q[:200], 4), 1) == np.sort(KDTree(jit).query(q[:200], 4)[1], 1)).all()


def test_edge_layouts_travel_with_the_graph_directory(tmp_path):
    """``save_graph(..., with_layouts=True)`` writes edge_layouts.pt beside the reference's files (the reference's loader,
    utils/graph.py:146-422, reads its own file names only, so the directory stays a valid spec-v0.1.0 graph); ``load_graph``
    preloads the int32 CSR / CSC views + tile schedules, and layers built on the loaded edge sets take them from the content
    cache instead of sorting on the host -- bit-identical to a fresh build."""
    from neural_lam_amd import gnn_layers as hl

    raw = G.create_regular_grid_graph(G.regular_grid_xy(40, 36), n_max_levels=3, hierarchical=True)
    G.save_graph(tmp_path, raw, with_layouts=True)
    assert (tmp_path / G.EDGE_LAYOUT_FILENAME).exists() and (tmp_path / G.METAINFO_FILENAME).exists()
    G._LAYOUT_CACHE.clear()
    _, tensors = G.load_graph(tmp_path, 100.0)
    sets = G._graph_edge_sets(tensors)
    assert len(G._LAYOUT_CACHE) == len({G.edge_layout_key(*s) for s in sets}) > 0
    before = dict(G._LAYOUT_CACHE)
    for ei, ns, nr in sets:
        layer = hl.InteractionNet(ei, 8)
        csr, tiles, has_split = layer._host_csr
        assert G._LAYOUT_CACHE == before                      # nothing was rebuilt
        ref = G.build_edge_csr(ei, num_send=ns, num_rec=nr)
        ref_tiles, ref_split, _ = G.build_tile_schedule(ref.rowptr)
        assert all(torch.equal(getattr(csr, f), getattr(ref, f)) for f in G._LAYOUT_FIELDS)
        assert torch.equal(tiles, ref_tiles) and has_split == ref_split and csr.max_in_degree == ref.max_in_degree
    # layers on the same edge set share the host layout object (and with it one device copy)
    a, b = hl.InteractionNet(sets[0][0], 8), hl.InteractionNet(sets[0][0].clone(), 16)
    assert a._host_csr[0] is b._host_csr[0]
    # a different edge set is not served by a stale entry
    other = sets[0][0].clone()
    other[1, 0] = (other[1, 0] + 1) % (int(other[1].max()) + 1)
    assert G.edge_layout_key(other, sets[0][1], sets[0][2]) not in before
