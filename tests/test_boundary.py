"""C-ABI / drop-in boundary checks that need no GPU: the library loads and exports
every symbol include/nlam_hip.h declares, ctypes mirrors the C struct layout, the
product never routes through the oracle or a CPU fallback, and the host classes
keep the reference's constructor conventions (tests/test_gnn_layers.py:134-181)."""
import ctypes as C
import re
import subprocess
import sys
from pathlib import Path

import pytest
import torch

from conftest import ROOT
from neural_lam_amd import _lib as L
from neural_lam_amd import gnn_layers as hl
from neural_lam_amd import graph as G
from oracle import gnn_layers as og


def _edge_index(ns=5, nr=4, e=10, seed=0):
    torch.manual_seed(seed)
    ei = torch.stack([torch.randint(0, ns, (e,)), torch.randint(0, nr, (e,))])
    ei[1, -1] = nr - 1
    return ei


def test_library_exports_every_declared_symbol():
    header = (ROOT / "include" / "nlam_hip.h").read_text()
    declared = set(re.findall(r"^int(?:32|64)_t\s+(nlam_\w+)\s*\(", header, flags=re.M))
    assert declared and declared == set(L.EXPORTS)
    lib = L.load()  # raises if the .so is missing
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.nlam_abi_version() == L.ABI_VERSION == 8
    assert lib.nlam_max_width() >= 64
    assert lib.nlam_num_blocks(1) == 1 and lib.nlam_num_blocks(10**6) == 256
    # tuning knob: known key accepted (and restored), unknown key / negative value rejected
    assert lib.nlam_set_tuning(L.TUNE_WBF_MIN_SUPERTILES, 192) == 0
    assert lib.nlam_set_tuning(L.TUNE_WBF_MIN_SUPERTILES, -1) == -1 and lib.nlam_set_tuning(99, 1) == -1
    assert lib.nlam_set_tuning(L.TUNE_WBF_HALF, 8) == -1 and lib.nlam_set_tuning(L.TUNE_WBF_HALF, 1) == 0
    assert lib.nlam_set_tuning(L.TUNE_WGRAD_LDMA, 8) == -1 and lib.nlam_set_tuning(L.TUNE_WGRAD_LDMA, 3) == 0   # (3 = the default: the setting is process-wide)
    assert lib.nlam_set_tuning(L.TUNE_WGRAD_LDMA_VAR, 4) == -1 and lib.nlam_set_tuning(L.TUNE_WGRAD_LDMA_VAR, 0) == 0
    assert lib.nlam_set_tuning(L.TUNE_WBF_EDGE, 4) == -1 and lib.nlam_set_tuning(L.TUNE_WBF_EDGE, 1) == 0
    assert lib.nlam_set_tuning(L.TUNE_WGRAD_MAX_WGS, 3) == -1 and lib.nlam_set_tuning(L.TUNE_WGRAD_MAX_WGS, 128) == 0
    assert lib.nlam_set_tuning(L.TUNE_CHAIN_CUS, 8) == -1 and lib.nlam_set_tuning(L.TUNE_CHAIN_CUS, 256) == 0


def test_ctypes_structs_match_c_layout(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "nlam_hip.h"\n'
        'int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(nlam_src_t), sizeof(nlam_mlp_fwd_t),'
        " sizeof(nlam_mlp_bwd_t), sizeof(nlam_wgrad_t), offsetof(nlam_mlp_fwd_t, rstd),"
        " offsetof(nlam_mlp_bwd_t, vec_partials), offsetof(nlam_wgrad_t, partials), sizeof(nlam_window_t),"
        " offsetof(nlam_window_t, n_times), offsetof(nlam_window_t, ar_steps), sizeof(nlam_pack_job_t), offsetof(nlam_pack_job_t, flags)); return 0;}\n"
    )
    exe = tmp_path / "sz"
    subprocess.run(["gcc", f"-I{ROOT / 'include'}", str(src), "-o", str(exe)], check=True)
    sizes = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert sizes == [
        C.sizeof(L.Src), C.sizeof(L.MlpFwd), C.sizeof(L.MlpBwd), C.sizeof(L.Wgrad),
        L.MlpFwd.rstd.offset, L.MlpBwd.vec_partials.offset, L.Wgrad.partials.offset,
        C.sizeof(L.Window), L.Window.n_times.offset, L.Window.ar_steps.offset,
        C.sizeof(L.PackJob), L.PackJob.flags.offset,
    ]


def test_product_never_imports_oracle():
    for py in (ROOT / "neural_lam_amd").rglob("*.py"):
        text = py.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), py
    # and importing the product does not pull the oracle in
    code = "import sys; sys.path.insert(0, %r); import neural_lam_amd.models, neural_lam_amd.trainer; " \
           "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules)" % str(ROOT)
    subprocess.run([sys.executable, "-c", code], check=True)


def test_cpu_tensors_fail_loudly():
    net = hl.InteractionNet(_edge_index(), 8)
    with pytest.raises(RuntimeError, match="MI355X only"):
        net(torch.randn(5, 8), torch.randn(4, 8), torch.randn(10, 8))
    with pytest.raises(RuntimeError, match="MI355X only"):
        hl.make_mlp([3, 8, 8])(torch.randn(7, 3))


def test_constructor_conventions_match_reference():
    ei = _edge_index()
    assert issubclass(hl.PropagationNet, hl.InteractionNet)
    pnet = hl.PropagationNet(ei, 8, aggr="sum")
    assert pnet.aggr == "mean"  # forced mean aggregation (gnn_layers.py:217-227)
    inet = hl.InteractionNet(ei, 8)
    assert inet.aggr == "sum" and inet.update_edges
    assert inet.edge_mlp[0].in_features == 24 and inet.aggr_mlp[0].in_features == 16
    assert int(inet.num_rec) == int(ei[1].max()) + 1
    assert torch.equal(inet.edge_index[0], ei[0] + inet.num_rec) and torch.equal(inet.edge_index[1], ei[1])
    assert "edge_index" not in inet.state_dict()  # non-persistent buffer (gnn_layers.py:86)
    with pytest.raises(ValueError):
        hl.InteractionNet(ei, 8, aggr="max")
    with pytest.raises(ValueError):
        hl.get_gnn_class("Nope")
    assert hl.get_gnn_class("PropagationNet") is hl.PropagationNet


@pytest.mark.parametrize("kw", [{}, {"edge_chunk_sizes": [4, 6], "aggr_chunk_sizes": [1, 3]}, {"update_edges": False}])
def test_state_dict_keys_and_shapes_equal_oracle(kw):
    ei = _edge_index()
    a, b = hl.InteractionNet(ei, 8, **kw).state_dict(), og.InteractionNet(ei, 8, **kw).state_dict()
    assert list(a.keys()) == list(b.keys())
    assert all(a[k].shape == b[k].shape for k in a)


def test_sequential_child_names_follow_pyg():
    seq = hl.make_gnn_seq(_edge_index(5, 5), 3, 1, 8)
    keys = list(seq.state_dict().keys())
    assert keys[0] == "module_0.edge_mlp.0.weight" and any(k.startswith("module_2.aggr_mlp.3") for k in keys)
    with pytest.raises(ValueError):
        hl.make_gnn_seq(_edge_index(5, 5), 0, 1, 8)


def test_other_depths_keep_reference_structure():
    """hidden_layers != 1 (utils/networks.py:8-40): same children / parameter names as the reference's Sequential."""
    from oracle import gnn_layers as og

    for h in (0, 2, 3):
        bp = [6] + [8] * (h + 1)
        mine, ref = hl.make_mlp(bp), og.make_mlp(bp)
        assert list(mine.state_dict().keys()) == list(ref.state_dict().keys())
        assert [type(m).__name__ for m in mine] == [type(m).__name__ for m in ref]
        with pytest.raises(RuntimeError):   # no CPU / eager fallback at any depth
            mine(torch.zeros(3, 6))



def _skewed_edges():
    g = torch.Generator().manual_seed(3)
    ns, nr, E = 40, 25, 900  # some receivers exceed 32 in-edges -> cut over several tiles
    ei = torch.stack([torch.randint(0, ns, (E,), generator=g), (torch.rand(E, generator=g) ** 3 * nr).long()])
    ei[1, -1] = nr - 1
    return ei, ns, nr, E


def test_tile_schedule_covers_every_edge_and_receiver_once():
    """The C-ABI's one-pass schedule: pieces of a receiver with more than 32 in-edges are NLAM_TILE_SPLIT tiles (atomic adds)."""
    ei, ns, nr, E = _skewed_edges()
    csr = G.build_edge_csr(ei, num_send=ns)
    tiles, has_split, split = G.build_tile_schedule(csr.rowptr, virtual_split=False)
    assert has_split == (csr.max_in_degree > 32) and split is None
    cov_e = torch.zeros(E, dtype=torch.int32)
    cov_r = torch.zeros(csr.num_rec, dtype=torch.int32)
    for row0, nrows, seg0, nseg in tiles.tolist():
        split = bool(nseg & G.TILE_SPLIT)
        nseg &= ~G.TILE_SPLIT
        assert 0 <= nrows <= 32 and 1 <= nseg <= 32
        cov_e[row0 : row0 + nrows] += 1
        if split:
            assert nseg == 1 and int(csr.rowptr[seg0]) <= row0 and row0 + nrows <= int(csr.rowptr[seg0 + 1])
            cov_r[seg0] = 1
        else:
            assert int(csr.rowptr[seg0]) == row0 and int(csr.rowptr[seg0 + nseg]) == row0 + nrows
            cov_r[seg0 : seg0 + nseg] += 1
    assert torch.all(cov_e == 1) and torch.all(cov_r == 1)


def test_virtual_split_schedule_is_a_plain_schedule_on_extended_row_pointers():
    """The default (deterministic) schedule: no flagged tile; a piece of a long receiver is an ordinary one-receiver tile on
    a VIRTUAL segment whose row range the extended row pointers give, and the combine lists map the pieces back to the
    receiver in CSR order.  Emulating kernel + nlam_split_combine on the host reproduces index_add."""
    ei, ns, nr, E = _skewed_edges()
    csr = G.build_edge_csr(ei, num_send=ns)
    tiles, has_split, split = G.build_tile_schedule(csr.rowptr)
    assert not has_split and split is not None and csr.max_in_degree > 32
    G.attach_split(csr, split)
    rp = csr.rowptr_ext.tolist()
    assert len(rp) == csr.nseg_ext + 1 and rp[: nr + 1] == csr.rowptr.tolist() and csr.inv_deg_ext.shape[0] == csr.nseg_ext
    cov_e = torch.zeros(E, dtype=torch.int32)
    written = set()
    vals = torch.randn(E, 3, dtype=torch.float64)             # per CSR position
    buf = torch.full((csr.nseg_ext, 3), float("nan"), dtype=torch.float64)
    for row0, nrows, seg0, nseg in tiles.tolist():
        assert not (nseg & G.TILE_SPLIT) and 0 <= nrows <= 32 and 1 <= nseg <= 32
        assert rp[seg0] == row0 and rp[seg0 + nseg] == row0 + nrows   # what the kernels read: rowptr[seg0 .. seg0 + nseg]
        cov_e[row0 : row0 + nrows] += 1
        for sg in range(seg0, seg0 + nseg):
            assert sg not in written
            written.add(sg)
            buf[sg] = vals[rp[sg] : rp[sg + 1]].sum(0) * float(csr.inv_deg_ext[sg])
    assert torch.all(cov_e == 1)
    cp, cs, cd = csr.comb_ptr.tolist(), csr.comb_src.tolist(), csr.comb_dst.tolist()
    long_recs = [r for r in range(nr) if int(csr.rowptr[r + 1] - csr.rowptr[r]) > 32]
    assert cd == long_recs and all(r not in written for r in cd)
    for s, r in enumerate(cd):
        pieces = cs[cp[s] : cp[s + 1]]
        assert all(v > nr and v in written for v in pieces) and len(pieces) == -(-int(csr.rowptr[r + 1] - csr.rowptr[r]) // 32)
        buf[r] = buf[pieces].sum(0)
    ref = torch.zeros(nr, 3, dtype=torch.float64).index_add_(0, csr.rec.long(), vals) * csr.inv_deg.double()[:, None]
    assert torch.allclose(buf[:nr], ref, atol=1e-12)


# ---------------------------------------------------------------------------
# host-side planning of the C-ABI (no GPU needed: these entry points only compute sizes)
# ---------------------------------------------------------------------------
def _fwd_desc(widths, hid, dout, rows, mm_bits, batch=1):
    p = L.MlpFwd()
    p.nsrc, p.batch, p.rows, p.ntiles = len(widths), batch, rows, (rows + 31) // 32
    for k, w in enumerate(widths):
        p.src[k].width = w
    p.hid, p.dout, p.flags = hid, dout, mm_bits << 8
    return p


def test_packed_weight_scratch_follows_the_kernel_family():
    """nlam_mlp_fwd_wpack_floats: none for d <= 64 (weights live in LDS); fp32 A-operand quads for the one-tile wide
    kernels (small launches or NLAM_MATMUL=f32); NS bf16 terms in K=16 groups for the split super-tile kernels."""
    import ctypes as C

    lib = L.load()
    f = lambda p: lib.nlam_mlp_fwd_wpack_floats(C.byref(p))  # noqa: E731
    assert f(_fwd_desc([64, 64, 64], 64, 64, 255136, 3)) == 0
    d, E = 256, 255136
    hbt = obt = d // 32
    fp32 = (hbt * (3 * d // 32) + obt * hbt) * 1024
    assert f(_fwd_desc([d, d, d], d, d, E, 0)) == fp32                       # matrix mode f32
    assert f(_fwd_desc([d, d, d], d, d, 32 * 40, 3)) == fp32                 # 40 tiles: too few super tiles
    for ns in (1, 3):
        groups = hbt * (3 * d // 16) + obt * 2 * hbt
        assert f(_fwd_desc([d, d, d], d, d, E, ns)) == groups * ns * 256
    assert f(_fwd_desc([d, d, d], d, d, E, 2)) == (hbt * (3 * d // 16) + obt * 2 * hbt) * 3 * 256   # two terms run as three
    # source widths that are not a multiple of 4 stay on the fp32 kernels
    assert f(_fwd_desc([18, ], 128, 128, E, 3)) == (4 * 1 + 4 * 4) * 1024
    # the tuning knob moves the boundary (and is restored)
    assert lib.nlam_set_tuning(L.TUNE_WBF_MIN_SUPERTILES, 0) == 0
    try:
        assert f(_fwd_desc([d, d, d], d, d, 32 * 40, 3)) != fp32
    finally:
        assert lib.nlam_set_tuning(L.TUNE_WBF_MIN_SUPERTILES, 192) == 0


def test_launch_shape_queries():
    import ctypes as C

    lib = L.load()
    assert lib.nlam_num_blocks(0) == 1 and lib.nlam_num_blocks(255) == 255 and lib.nlam_num_blocks(256) == 256
    q = L.Wgrad()
    q.m, q.n, q.batch, q.rows, q.nsrc = 64, 192, 1, 255136, 3
    for k in range(3):
        q.src[k].width = 64
    assert lib.nlam_wgrad_nparts(C.byref(q)) == 512          # narrow: one partial per 32-row chunk, capped
    q.rows = 96
    assert lib.nlam_wgrad_nparts(C.byref(q)) == 3
    q.m, q.n, q.rows, q.flags = 256, 768, 255136, 3 << 8
    for k in range(3):
        q.src[k].width = 256
    assert lib.nlam_wgrad_nparts(C.byref(q)) == 128 // 3      # split-bf16: three 256 x 256 windows, at most 128 workgroups (NLAM_TUNE_WGRAD_MAX_WGS)
    q.flags = (3 << 8) | L.F_WGRAD_SOLO
    assert lib.nlam_wgrad_nparts(C.byref(q)) == 256 // 3      # ... and one per CU when the launch has the chip to itself (NLAM_F_WGRAD_SOLO)
    # small wide problems: the slice floor is 64 WORKGROUPS (row slices x 256 x 256 windows), 64 slices for a solo launch
    q.m, q.n, q.rows, q.nsrc, q.flags = 512, 512, 6561, 1, 1 << 8
    q.src[0].width = 512
    assert lib.nlam_wgrad_nparts(C.byref(q)) == 26            # ceil(206 chunks / 8) >= 64 / 4 windows
    q.flags = (1 << 8) | L.F_WGRAD_SOLO
    assert lib.nlam_wgrad_nparts(C.byref(q)) == 64
    q.m, q.n, q.rows, q.nsrc, q.flags = 256, 768, 255136, 3, 0
    for k in range(3):
        q.src[k].width = 256
    assert lib.nlam_wgrad_nparts(C.byref(q)) == 1024 // 12    # fp32: twelve 128 x 128 windows


def test_backward_planning_queries():
    """nlam_mlp_bwd_wpack_floats / nlam_mlp_bwd_blocks: transposed-weight scratch per source with a data gradient,
    and the number of partial-sum rows the caller has to provide (workgroups x row groups)."""
    import ctypes as C

    lib = L.load()

    def desc(d, rows, mm_bits, dmode=(1, 2, 3)):
        p = L.MlpBwd()
        p.nsrc, p.batch, p.rows, p.ntiles = 3, 1, rows, (rows + 31) // 32
        for k in range(3):
            p.src[k].width = d
            p.dmode[k] = dmode[k]
        p.hid, p.dout, p.flags = d, d, mm_bits << 8
        return p

    wp = lambda p: lib.nlam_mlp_bwd_wpack_floats(C.byref(p))  # noqa: E731
    nb = lambda p: lib.nlam_mlp_bwd_blocks(C.byref(p))  # noqa: E731
    assert wp(desc(64, 255136, 3)) == 0 and nb(desc(64, 255136, 3)) == 256
    d, hbt = 256, 8
    assert wp(desc(d, 255136, 0)) == (hbt * hbt + 3 * hbt * hbt) * 1024                       # fp32 quads, 3 sources
    assert wp(desc(d, 255136, 3)) == (hbt * 2 * hbt + 3 * hbt * 2 * hbt) * 3 * 256            # split terms, K = 16 groups
    assert wp(desc(d, 255136, 3, dmode=(1, 0, 0))) == (hbt * 2 * hbt + hbt * 2 * hbt) * 3 * 256
    assert nb(desc(d, 255136, 3)) == 256                      # 64-row super tiles, one row group
    assert nb(desc(128, 255136, 3)) == 512                    # d <= 128: two row groups per workgroup
    assert nb(desc(d, 32 * 10, 3)) == 10                      # small launch: fp32 one-tile kernels, one workgroup per tile


def test_kernel_family_and_grouped_backward_planning():
    """nlam_mlp_*_family (0 narrow / 1 fp32 wide / 2 split-bf16 super tiles) and nlam_mlp_bwd_group_blocks: the chunks of a
    SplitMLPs layer at d = 128 (hi_lam_parallel.py:127-143) are fp32-wide members of one grid, dealt workgroups in proportion
    to their tiles; a member of another family or shape is refused before anything is launched."""
    import ctypes as C

    lib = L.load()
    assert lib.nlam_mlp_fwd_family(None) == -1 and lib.nlam_mlp_bwd_family(None) == -1
    fam = lambda p: lib.nlam_mlp_fwd_family(C.byref(p))  # noqa: E731
    assert fam(_fwd_desc([64, 64, 64], 64, 64, 255136, 3)) == 0
    assert fam(_fwd_desc([128, 128, 128], 128, 128, 32 * 206, 3)) == 1      # 206 tiles: below the super-tile threshold
    assert fam(_fwd_desc([128, 128, 128], 128, 128, 255136, 3)) == 2
    assert fam(_fwd_desc([128, 128, 128], 128, 128, 255136, 0)) == 1        # matrix mode f32

    one = C.c_float(0.0)
    ptr = C.cast(C.pointer(one), C.c_void_p)   # any non-null pointer: nothing below reaches a kernel

    def bwd(d, rows, mm_bits=3):
        p = L.MlpBwd()
        p.nsrc, p.batch, p.rows, p.ntiles = 3, 1, rows, (rows + 31) // 32
        for k in range(3):
            p.src[k].width = d
            p.dmode[k] = 0
        p.hid, p.dout, p.flags = d, d, mm_bits << 8
        p.W1 = p.W2 = p.z1 = p.wpack = ptr
        p.wpack_floats = lib.nlam_mlp_bwd_wpack_floats(C.byref(p))
        return p

    tiles = [206, 23, 3, 173, 17]
    arr = (L.MlpBwd * 5)(*[bwd(128, 32 * t) for t in tiles])
    assert all(lib.nlam_mlp_bwd_family(C.byref(arr[k])) == 1 for k in range(5))
    blocks = (C.c_int32 * 5)()
    assert lib.nlam_mlp_bwd_group_blocks(arr, 5, blocks) == 0
    assert list(blocks) == tiles                                # fewer tiles than resident workgroups: one workgroup per tile
    big = [1500, 500, 3]
    arr = (L.MlpBwd * 3)(*[bwd(128, 32 * t, mm_bits=0) for t in big])
    assert lib.nlam_mlp_bwd_group_blocks(arr, 3, blocks) == 0
    got = list(blocks)[:3]
    single = lib.nlam_mlp_bwd_blocks(C.byref(arr[0]))           # what one launch keeps resident
    assert sum(got) <= single and got[2] == 1 and abs(got[0] - 3 * got[1]) <= 3
    # a member of another shape / family: refused
    arr = (L.MlpBwd * 2)(bwd(128, 32 * 20), bwd(96, 32 * 20))
    assert lib.nlam_mlp_bwd_group_blocks(arr, 2, blocks) == -2
    arr = (L.MlpBwd * 2)(bwd(128, 32 * 20), bwd(128, 255136))
    assert lib.nlam_mlp_bwd_group_blocks(arr, 2, blocks) == -2
    assert lib.nlam_mlp_bwd_group_blocks(None, 2, blocks) == -1 and lib.nlam_mlp_bwd_group_blocks(arr, 9, blocks) == -1
    assert lib.nlam_mlp_bwd_group(arr, 2, None) == -2           # the launch runs the same checks
    # grouped weight gradients: one shape, split-bf16 wide family, nparts as nlam_wgrad_nparts says
    def wg(rows, m=128, mm_bits=3):
        q = L.Wgrad()
        q.A = q.partials = ptr
        q.m, q.n, q.batch, q.rows, q.nsrc, q.flags = m, 384, 1, rows, 3, mm_bits << 8
        for k in range(3):
            q.src[k].width = 128
            q.src[k].ptr = ptr
        q.nparts = lib.nlam_wgrad_nparts(C.byref(q))
        return q

    assert lib.nlam_wgrad_group(None, 2, None) == -1
    qs = (L.Wgrad * 2)(wg(32 * 20), wg(32 * 200, m=96))
    assert lib.nlam_wgrad_group(qs, 2, None) == -2              # another shape
    qs = (L.Wgrad * 2)(wg(32 * 20), wg(32 * 200, mm_bits=0))
    assert lib.nlam_wgrad_group(qs, 2, None) == -2              # a member of the fp32 family
    qs = (L.Wgrad * 2)(wg(32 * 20), wg(32 * 200))
    qs[1].nparts += 1
    assert lib.nlam_wgrad_group(qs, 2, None) == -1              # partials sized for another launch shape


def test_argument_errors_are_reported_before_any_launch():
    """Error convention of the C-ABI (include/nlam_hip.h): NLAM_EINVAL (-1) for inconsistent arguments, NLAM_EUNSUP (-2)
    for widths this build does not instantiate, 0 for an empty problem -- all decided on the host, so this runs
    without a GPU."""
    import ctypes as C

    lib = L.load()
    EINVAL, EUNSUP = -1, -2
    one = C.c_float(0.0)
    ptr = C.cast(C.pointer(one), C.c_void_p)   # any non-null pointer: nothing below reaches a kernel
    assert lib.nlam_mlp_fwd(None, None) == EINVAL
    p = _fwd_desc([64, 64, 64], 64, 64, 320, 3)
    assert lib.nlam_mlp_fwd(C.byref(p), None) == EINVAL          # no weights
    p.W1 = p.W2 = ptr
    p.nsrc = 0
    assert lib.nlam_mlp_fwd(C.byref(p), None) == EINVAL
    p.nsrc = 3
    p.flags |= L.F_ADD_SRC0
    p.src[0].width = 32
    assert lib.nlam_mlp_fwd(C.byref(p), None) == EINVAL          # residual source narrower than the output
    p.src[0].width = 64
    p.flags = (3 << 8) | L.F_MEAN
    assert lib.nlam_mlp_fwd(C.byref(p), None) == EINVAL          # mean aggregation without the degree table
    p.flags = 3 << 8
    p.rows = 0
    assert lib.nlam_mlp_fwd(C.byref(p), None) == 0               # empty problem: nothing to do
    w = _fwd_desc([256, 256, 256], 256, 256, 255136, 3)
    w.W1 = w.W2 = ptr
    assert lib.nlam_mlp_fwd(C.byref(w), None) == EINVAL          # wide shapes need the packed-weight scratch
    big = _fwd_desc([1024], 1024, 1024, 320, 0)
    big.W1 = big.W2 = ptr
    assert lib.nlam_mlp_fwd(C.byref(big), None) == EUNSUP        # beyond nlam_max_width()
    assert lib.nlam_wgrad(None, None) == EINVAL
    assert lib.nlam_affine_mix(None, None, None, None, None, None, None, None, 1, 1, 1, None) == EINVAL
    assert lib.nlam_segment_sum(None, 0, None, None, None, None, 1, 1, 1, None) == EINVAL
    assert lib.nlam_segment_sum_acc(None, 0, None, None, None, None, 1, 1, 1, None) == EINVAL
    assert lib.nlam_segment_sum_add(None, 0, None, None, None, None, None, 1, 1, 1, None) == EINVAL   # (no `extra`: that is nlam_segment_sum_acc)


def test_pack_and_layout_queries_are_host_logic():
    """The shape queries behind the round-3 entry points run without a GPU: image sizes of nlam_mlp_pack, the dz2 row stride of a
    ragged output width, the pack records of a wide launch."""
    lib = L.load()
    job = L.PackJob()
    w1 = torch.zeros(64, 192)
    w2 = torch.zeros(64, 64)
    job.W1, job.W2, job.hid, job.dout, job.nsrc = w1.data_ptr(), w2.data_ptr(), 64, 64, 3
    for k in range(3):
        job.width[k] = 64
    job.flags = 3 << 8   # bf16x3
    # forward image [W1s | W2s]: 3 terms x (64 x 192 + 64 x 64) bf16 = 6 B per weight; backward image the same weights transposed
    assert lib.nlam_mlp_pack_floats(C.byref(job), 0) == 3 * (64 * 192 + 64 * 64) // 2
    assert lib.nlam_mlp_pack_floats(C.byref(job), 1) == 3 * (64 * 64 + 3 * 64 * 64) // 2
    job.flags = 0            # fp32 MFMA mode: no images
    assert lib.nlam_mlp_pack_floats(C.byref(job), 0) == 0
    job.flags, job.hid = 3 << 8, 128   # wide: the images are a narrow-kernel format
    assert lib.nlam_mlp_pack_floats(C.byref(job), 0) == 0
    job.hid, job.nsrc, job.dout = 64, 1, 17   # output_map: one output block, W2 zero-padded to 32 rows
    job.width[0] = 64
    assert lib.nlam_mlp_pack_floats(C.byref(job), 0) == 3 * (64 * 64 + 32 * 64) // 2
    job.flags |= L.F_PRE_ADD   # factorised: only source 0 has columns in W1
    job.nsrc, job.dout = 3, 64
    assert lib.nlam_mlp_pack_floats(C.byref(job), 1) == 3 * (64 * 64 + 64 * 64) // 2

    p = L.MlpBwd()
    g = torch.zeros(4)
    p.nsrc, p.batch, p.rows, p.ntiles, p.hid, p.dout, p.flags = 1, 1, 1000, 32, 64, 17, 3 << 8
    p.src[0].width, p.dmode[0], p.g_out = 64, 1, g.data_ptr()
    assert lib.nlam_mlp_bwd_dz2_ld(C.byref(p)) == 32          # split-bf16 fast kernel: dz2 padded to the output block
    p.flags = 0
    assert lib.nlam_mlp_bwd_dz2_ld(C.byref(p)) == 0           # fp32 matrix mode: the generic kernel, rows of dout floats
    p.flags, p.dout = 3 << 8, 64
    assert lib.nlam_mlp_bwd_dz2_ld(C.byref(p)) == 0
    p.dout, p.ln_w = 17, g.data_ptr()                         # with a LayerNorm the ragged width stays on the generic kernel
    assert lib.nlam_mlp_bwd_dz2_ld(C.byref(p)) == 0

    f = L.MlpFwd()
    w1w, w2w = torch.zeros(256, 768), torch.zeros(256, 256)
    f.nsrc, f.batch, f.rows, f.ntiles, f.hid, f.dout, f.flags = 3, 1, 57616, 1801, 256, 256, 3 << 8
    for k in range(3):
        f.src[k].width = 256
    f.W1, f.W2 = w1w.data_ptr(), w2w.data_ptr()
    nwp = lib.nlam_mlp_fwd_wpack_floats(C.byref(f))
    assert nwp > 0
    buf = torch.zeros(8)
    f.wpack, f.wpack_floats = buf.data_ptr(), nwp     # only the address goes into the records
    recs, kind = (L.PackRec * 4)(), C.c_int32(-1)
    assert lib.nlam_mlp_fwd_pack_records(C.byref(f), recs, 4, C.byref(kind)) == 4 and kind.value == 3   # three W1 sources + W2, three bf16 terms
    assert lib.nlam_mlp_fwd_pack_records(C.byref(f), recs, 2, C.byref(kind)) == -1                      # capacity
    f.hid = f.dout = 64
    for k in range(3):
        f.src[k].width = 64
    assert lib.nlam_mlp_fwd_pack_records(C.byref(f), recs, 4, C.byref(kind)) == 0                       # a narrow launch has no wide scratch
    assert lib.nlam_set_tuning(L.TUNE_WGRAD_MIN_PARTS, 0) == -1 and lib.nlam_set_tuning(L.TUNE_WGRAD_MIN_PARTS, 128) == 0
    assert lib.nlam_set_tuning(L.TUNE_WGRAD_BIG_MIN_ROWS, -1) == -1 and lib.nlam_set_tuning(L.TUNE_WGRAD_BIG_MIN_ROWS, 0) == 0


def test_padded_input_cache_is_keyed_by_tensor_identity():
    """Advisor finding (round 4, high): FusedMLP._padded_input cached the zero-padded input under (address, shape, version); a
    fresh activation that the caching allocator placed where an earlier one lived hit the cache with other contents (16 stale
    returns out of 20 tensors in a CPU check).  The key is now the tensor OBJECT (weak reference) plus its version counter."""
    mlp = hl.make_mlp([3, 8, 8])
    stale = 0
    for k in range(20):
        x = torch.full((5, 3), float(k))   # same size every time: the allocator hands out the same block again and again
        p = mlp._padded_input(x)
        stale += int(not torch.equal(p[:, :3], x))
        assert p.shape == (5, 4) and float(p[:, 3].abs().max()) == 0.0
        del x, p
    assert stale == 0
    x = torch.randn(7, 3)
    p1 = mlp._padded_input(x)
    assert mlp._padded_input(x) is p1            # the same, unmodified tensor: served from the cache (the static feature buffers)
    x.add_(1.0)                                   # modified in place: the version counter moves, the copy is refreshed
    p2 = mlp._padded_input(x)
    assert p2 is not p1 and torch.equal(p2[:, :3], x)
    y = x.clone()
    assert mlp._padded_input(y) is not p2         # equal contents, another object: padded again


def test_suffix_geometry_with_residual_sources_never_registers_a_weight_image():
    """Advisor finding (round 4, medium): with hidden_layers >= 2 the last launch of a GNN-layer MLP gets a per-call
    ``torch.cat([zeros, W1])`` as its first weight; the trainer's weight packer keys on, and re-reads at the start of every later
    step, the ADDRESS of what it registers -- so that launch's geometry must carry no_pack."""
    from neural_lam_amd.ops import MlpGeometry

    geom = MlpGeometry(nsrc=3, flags=L.F_ADD_SRC0)
    srcs = (torch.zeros(4, 8), torch.zeros(4, 8), torch.zeros(4, 8))
    last, res = hl._suffix_geometry(geom, srcs)
    assert len(res) == 1 and last.no_pack and last.nsrc == 2
    plain, res0 = hl._suffix_geometry(MlpGeometry(nsrc=1), (torch.zeros(4, 8),))
    assert res0 == () and not plain.no_pack


def test_trainer_picks_its_executor_and_plans_bucket_launches_on_the_host():
    """trainer.Trainer(executor="auto") and Trainer._plan_buckets are host logic: the segmented executor for modules with a fused
    width above 64 and no chunked stage, the one-graph executor otherwise; a gradient bucket is launched behind the LAST chain
    segment whose side work reported one of its parameters, never before a bucket of lower index."""
    from neural_lam_amd.trainer import Trainer

    class Narrow(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(192, 64)

    class Wide(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(64, 256)

    class SplitMLPs(torch.nn.Module):   # (the name is what marks a chunked stage)
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(64, 256)

    class Chunked(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.s = SplitMLPs()

    assert Trainer._pick_executor(Narrow()) == "forks"
    assert Trainer._pick_executor(Wide()) == "segments"
    assert Trainer._pick_executor(Chunked()) == "forks"

    net = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 8), torch.nn.Linear(8, 8))
    tr = Trainer(net, optimizer_factory=lambda p, g: type("O", (), {"step": lambda self, s=1.0: None})(), bucket_bytes=1)
    assert len(tr.buckets.bounds) == 6                      # one parameter per bucket, in backward order: 2.bias, 2.weight, 1.bias ...
    P = tr.fp.params                                        # registration order: 0.weight, 0.bias, 1.weight, 1.bias, 2.weight, 2.bias

    class Seg:
        chain = [0, 1, 2, 3]
        # segment 0 completes the last layer's parameters, segment 1 layer 1's weight, segment 2 layer 1's bias AND again 2.weight
        # (a rollout reports a parameter once per AR step: the last report counts); layer 0 is reported by nobody (last segment)
        seg_params = [{id(P[5]), id(P[4])}, {id(P[2])}, {id(P[3]), id(P[4])}, set()]

    plan = tr._plan_buckets(Seg())
    # buckets: 0 = 2.bias (ready at 0), 1 = 2.weight (ready at 2), 2 = 1.bias (ready at 2), 3 = 1.weight (ready at 1, held back by
    # bucket 1), 4 = 0.bias, 5 = 0.weight (ready with the last segment)
    assert plan == [[0], [], [1, 2, 3], [4, 5]]

    class SegChain(Seg):
        # 2.bias is reported by segment 0's side work AND written by a launch of the chain itself (a later segment may add to it):
        # complete with the last segment only, and every bucket behind it is held back with it (advisor finding, round 5)
        chain_params = {id(P[5])}

    assert tr._plan_buckets(SegChain()) == [[], [], [], [0, 1, 2, 3, 4, 5]]
