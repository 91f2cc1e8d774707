"""ctypes binding of libnlam_hip.so (the C-ABI declared in include/nlam_hip.h).

The shared library is built in-tree by ``__graft_entry__.build()`` /
``python -m neural_lam_amd.build``.  There is no CPU fallback: if the library is
missing, or a tensor is not on a HIP device, the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

HERE = Path(__file__).resolve().parent
# NLAM_LIB lets the profiling tools load an instrumented build (tools/phase_timing.py); the product default is the in-tree library
LIB_PATH = Path(os.environ["NLAM_LIB"]) if os.environ.get("NLAM_LIB") else HERE / "libnlam_hip.so"

NLAM_MAX_SRC = 3
NLAM_MAX_CAT = 6
NLAM_MAX_REDUCE_JOBS = 40
NLAM_MAX_GROUP = 8
F_ADD_SRC0, F_ADD_SRC1, F_MEAN, F_SILU_B, F_PRE_ADD, F_LEAF_WGRAD, F_WPACK_READY, F_NO_ACT = 1, 2, 4, 8, 16, 32, 64, 128
F_STORE_BF16, F_A_BF16, F_S_BF16 = 1 << 10, 1 << 10, 1 << 11
F_ACC_DSRC0 = 1 << 12
F_WGRAD_SOLO = 1 << 13
TUNE_WBF_MIN_SUPERTILES = 1   # nlam_set_tuning keys (include/nlam_hip.h)
TUNE_WGRAD_CHUNKS = 2
TUNE_LIN_WGS = 3
TUNE_WGRAD_MIN_PARTS = 4
TUNE_WGRAD_BIG_MIN_ROWS = 5
TUNE_WBF_HALF = 6
TUNE_LIN_GEMM = 7
TUNE_WBF_V4 = 8
TUNE_WGRAD_LDMA = 9
TUNE_WGRAD_LDMA_VAR = 10
TUNE_WBF_EDGE = 11
TUNE_WGRAD_MAX_WGS = 12
TUNE_CHAIN_CUS = 13
TILE_SPLIT = 1 << 30

EXPORTS = [
    "nlam_abi_version",
    "nlam_grid_waves",
    "nlam_num_blocks",
    "nlam_max_width",
    "nlam_set_tuning",
    "nlam_affine_mix",
    "nlam_mlp_fwd_wpack_floats",
    "nlam_mlp_bwd_wpack_floats",
    "nlam_mlp_bwd_blocks",
    "nlam_mlp_bwd_dz2_ld",
    "nlam_wgrad_nparts",
    "nlam_mlp_fwd",
    "nlam_mlp_bwd",
    "nlam_mlp_pack_floats",
    "nlam_mlp_pack",
    "nlam_mlp_fwd_pack_records",
    "nlam_mlp_bwd_pack_records",
    "nlam_pack_records",
    "nlam_wgrad",
    "nlam_segment_sum",
    "nlam_segment_sum_acc",
    "nlam_segment_sum_add",
    "nlam_split_combine",
    "nlam_segment_sum_bf16",
    "nlam_store_bf16_supported",
    "nlam_reduce_partials",
    "nlam_reduce_jobs",
    "nlam_wmse_fwd",
    "nlam_wmse_bwd",
    "nlam_adamw_step",
    "nlam_adamw_step_resident",
    "nlam_standardize",
    "nlam_mlp_group_blocks",
    "nlam_mlp_fwd_group",
    "nlam_mlp_bwd_group",
    "nlam_mlp_fwd_family",
    "nlam_mlp_bwd_family",
    "nlam_mlp_bwd_group_blocks",
    "nlam_wgrad_group",
    "nlam_linear",
    "nlam_pre_add_supported",
    "nlam_step_tail_fwd",
    "nlam_step_tail_bwd",
    "nlam_concat",
    "nlam_window_len",
    "nlam_window_batch",
]


class Src(C.Structure):
    _fields_ = [
        ("ptr", C.c_void_p),
        ("idx", C.c_void_p),
        ("bstride", C.c_int64),
        ("width", C.c_int32),
        ("_pad", C.c_int32),
    ]


class MlpFwd(C.Structure):
    _fields_ = [
        ("src", Src * NLAM_MAX_SRC),
        ("nsrc", C.c_int32),
        ("batch", C.c_int32),
        ("rows", C.c_int32),
        ("ntiles", C.c_int32),
        ("tiles", C.c_void_p),
        ("W1", C.c_void_p),
        ("b1", C.c_void_p),
        ("W2", C.c_void_p),
        ("b2", C.c_void_p),
        ("ln_w", C.c_void_p),
        ("ln_b", C.c_void_p),
        ("eps", C.c_float),
        ("hid", C.c_int32),
        ("dout", C.c_int32),
        ("flags", C.c_uint32),
        ("out", C.c_void_p),
        ("out_idx", C.c_void_p),
        ("out_bstride", C.c_int64),
        ("aggr", C.c_void_p),
        ("rowptr", C.c_void_p),
        ("inv_deg", C.c_void_p),
        ("nseg_total", C.c_int32),
        ("ldw1", C.c_int32),
        ("z1", C.c_void_p),
        ("xhat", C.c_void_p),
        ("rstd", C.c_void_p),
        ("wpack", C.c_void_p),
        ("wpack_floats", C.c_int64),
        ("ncat", C.c_int32),
        ("_pad3", C.c_int32),
        ("cat_ptr", C.c_void_p * NLAM_MAX_CAT),
        ("cat_bstride", C.c_int64 * NLAM_MAX_CAT),
        ("cat_width", C.c_int32 * NLAM_MAX_CAT),
        ("cat_out", C.c_void_p),
    ]


class MlpBwd(C.Structure):
    _fields_ = [
        ("src", Src * NLAM_MAX_SRC),
        ("nsrc", C.c_int32),
        ("batch", C.c_int32),
        ("rows", C.c_int32),
        ("ntiles", C.c_int32),
        ("tiles", C.c_void_p),
        ("W1", C.c_void_p),
        ("W2", C.c_void_p),
        ("ln_w", C.c_void_p),
        ("hid", C.c_int32),
        ("dout", C.c_int32),
        ("flags", C.c_uint32),
        ("nseg_total", C.c_int32),
        ("g_out", C.c_void_p),
        ("out_idx", C.c_void_p),
        ("out_bstride", C.c_int64),
        ("g_aggr", C.c_void_p),
        ("seg_of_row", C.c_void_p),
        ("rowptr", C.c_void_p),
        ("inv_deg", C.c_void_p),
        ("z1", C.c_void_p),
        ("xhat", C.c_void_p),
        ("rstd", C.c_void_p),
        ("dz1", C.c_void_p),
        ("dz2", C.c_void_p),
        ("dsrc", C.c_void_p * NLAM_MAX_SRC),
        ("dsrc_bstride", C.c_int64 * NLAM_MAX_SRC),
        ("dmode", C.c_int32 * NLAM_MAX_SRC),
        ("ldw1", C.c_int32),
        ("vec_partials", C.c_void_p),
        ("vec_partials_rows", C.c_int32),
        ("vec_stride", C.c_int32),
        ("wpack", C.c_void_p),
        ("wpack_floats", C.c_int64),
        ("b1", C.c_void_p),
        ("dz2_ld", C.c_int32),
        ("_pad2", C.c_int32),
    ]


class Wgrad(C.Structure):
    _fields_ = [
        ("A", C.c_void_p),
        ("m", C.c_int32),
        ("batch", C.c_int32),
        ("rows", C.c_int32),
        ("nsrc", C.c_int32),
        ("src", Src * NLAM_MAX_SRC),
        ("flags", C.c_uint32),
        ("n", C.c_int32),
        ("partials", C.c_void_p),
        ("nparts", C.c_int32),
        ("_pad", C.c_int32),
    ]


class ReduceJob(C.Structure):
    _fields_ = [
        ("partials", C.c_void_p),
        ("out", C.c_void_p),
        ("stride", C.c_int64),
        ("nparts", C.c_int32),
        ("n", C.c_int32),
        ("accumulate", C.c_int32),
        ("ncols", C.c_int32),
        ("ld", C.c_int32),
        ("_pad", C.c_int32),
    ]


class ReduceJobs(C.Structure):
    _fields_ = [("job", ReduceJob * NLAM_MAX_REDUCE_JOBS), ("njobs", C.c_int32), ("_pad", C.c_int32)]


class Linear(C.Structure):
    _fields_ = [
        ("x", C.c_void_p),
        ("W", C.c_void_p),
        ("out", C.c_void_p),
        ("rows", C.c_int64),
        ("ldn", C.c_int64),
        ("ldk", C.c_int64),
        ("k", C.c_int32),
        ("n", C.c_int32),
        ("accumulate", C.c_int32),
        ("flags", C.c_uint32),
        ("W2", C.c_void_p),
        ("out2", C.c_void_p),
    ]


class Cat(C.Structure):
    _fields_ = [
        ("ptr", C.c_void_p * 6),
        ("bstride", C.c_int64 * 6),
        ("width", C.c_int32 * 6),
        ("nsrc", C.c_int32),
        ("batch", C.c_int32),
        ("nodes", C.c_int32),
        ("_pad", C.c_int32),
        ("out", C.c_void_p),
    ]


class Window(C.Structure):
    _fields_ = [
        ("state", C.c_void_p),
        ("forcing", C.c_void_p),
        ("times", C.c_void_p),
        ("sample_idx", C.c_void_p),
        ("init_states", C.c_void_p),
        ("target_states", C.c_void_p),
        ("forcing_windowed", C.c_void_p),
        ("target_times", C.c_void_p),
        ("state_mean", C.c_void_p),
        ("state_std", C.c_void_p),
        ("forcing_mean", C.c_void_p),
        ("forcing_std", C.c_void_p),
        ("n_times", C.c_int64),
        ("nodes", C.c_int32),
        ("d_state", C.c_int32),
        ("d_forcing", C.c_int32),
        ("batch", C.c_int32),
        ("ar_steps", C.c_int32),
        ("num_past_forcing_steps", C.c_int32),
        ("num_future_forcing_steps", C.c_int32),
        ("_pad", C.c_int32),
    ]


class StdJob(C.Structure):
    _fields_ = [
        ("x", C.c_void_p),
        ("out", C.c_void_p),
        ("mean", C.c_void_p),
        ("std", C.c_void_p),
        ("rows", C.c_int64),
        ("width", C.c_int32),
        ("rep", C.c_int32),
    ]


class StdJobs(C.Structure):
    _fields_ = [("job", StdJob * 4), ("njobs", C.c_int32), ("_pad", C.c_int32)]


class PackJob(C.Structure):
    _fields_ = [
        ("W1", C.c_void_p),
        ("W2", C.c_void_p),
        ("fwd_image", C.c_void_p),
        ("bwd_image", C.c_void_p),
        ("hid", C.c_int32),
        ("dout", C.c_int32),
        ("nsrc", C.c_int32),
        ("ldw1", C.c_int32),
        ("width", C.c_int32 * NLAM_MAX_SRC),
        ("flags", C.c_uint32),
    ]


class PackRec(C.Structure):
    _fields_ = [("bytes", C.c_ubyte * 64)]


ABI_VERSION = 8
_lib = None


def lib_path() -> Path:
    return LIB_PATH


def load():
    """dlopen the in-tree library once; fail loudly if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()').  "
            "neural_lam_amd has no CPU / eager fallback."
        )
    import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first so both share one HIP runtime)

    lib = C.CDLL(str(LIB_PATH))
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise RuntimeError(f"{LIB_PATH} does not export {name}")
    i32, i64, vp, f32 = C.c_int32, C.c_int64, C.c_void_p, C.c_float
    lib.nlam_abi_version.restype = i32
    lib.nlam_grid_waves.restype = i32
    lib.nlam_max_width.restype = i32
    lib.nlam_affine_mix.restype = i32
    lib.nlam_affine_mix.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp]
    lib.nlam_set_tuning.restype = i32
    lib.nlam_set_tuning.argtypes = [i32, i32]
    lib.nlam_num_blocks.argtypes = [i64]
    lib.nlam_num_blocks.restype = i32
    lib.nlam_mlp_fwd_wpack_floats.argtypes = [C.POINTER(MlpFwd)]
    lib.nlam_mlp_fwd_wpack_floats.restype = i64
    lib.nlam_mlp_bwd_wpack_floats.argtypes = [C.POINTER(MlpBwd)]
    lib.nlam_mlp_bwd_wpack_floats.restype = i64
    lib.nlam_mlp_bwd_dz2_ld.argtypes = [C.POINTER(MlpBwd)]
    lib.nlam_mlp_bwd_dz2_ld.restype = i32
    lib.nlam_mlp_bwd_blocks.argtypes = [C.POINTER(MlpBwd)]
    lib.nlam_mlp_bwd_blocks.restype = i32
    lib.nlam_wgrad_nparts.argtypes = [C.POINTER(Wgrad)]
    lib.nlam_wgrad_nparts.restype = i32
    lib.nlam_mlp_pack_floats.argtypes = [C.POINTER(PackJob), i32]
    lib.nlam_mlp_pack_floats.restype = i64
    lib.nlam_mlp_pack.argtypes = [vp, i32, vp]
    lib.nlam_mlp_pack.restype = i32
    lib.nlam_mlp_fwd_pack_records.argtypes = [C.POINTER(MlpFwd), C.POINTER(PackRec), i32, C.POINTER(C.c_int32)]
    lib.nlam_mlp_fwd_pack_records.restype = i32
    lib.nlam_mlp_bwd_pack_records.argtypes = [C.POINTER(MlpBwd), C.POINTER(PackRec), i32, C.POINTER(C.c_int32)]
    lib.nlam_mlp_bwd_pack_records.restype = i32
    lib.nlam_pack_records.argtypes = [vp, i32, i32, vp]
    lib.nlam_pack_records.restype = i32
    lib.nlam_mlp_fwd.argtypes = [C.POINTER(MlpFwd), vp]
    lib.nlam_mlp_fwd.restype = i32
    lib.nlam_mlp_bwd.argtypes = [C.POINTER(MlpBwd), vp]
    lib.nlam_mlp_bwd.restype = i32
    lib.nlam_wgrad.argtypes = [C.POINTER(Wgrad), vp]
    lib.nlam_wgrad.restype = i32
    lib.nlam_segment_sum.argtypes = [vp, i64, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.nlam_segment_sum.restype = i32
    lib.nlam_segment_sum_acc.argtypes = [vp, i64, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.nlam_segment_sum_acc.restype = i32
    lib.nlam_segment_sum_add.argtypes = [vp, i64, vp, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.nlam_segment_sum_add.restype = i32
    lib.nlam_segment_sum_bf16.argtypes = [vp, i64, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.nlam_segment_sum_bf16.restype = i32
    lib.nlam_store_bf16_supported.argtypes = [C.POINTER(MlpFwd)]
    lib.nlam_store_bf16_supported.restype = i32
    lib.nlam_split_combine.argtypes = [vp, i64, vp, vp, vp, i32, i32, i32, vp]
    lib.nlam_split_combine.restype = i32
    lib.nlam_reduce_partials.argtypes = [vp, i32, i64, i32, vp, i32, vp]
    lib.nlam_reduce_partials.restype = i32
    lib.nlam_reduce_jobs.argtypes = [C.POINTER(ReduceJobs), vp]
    lib.nlam_reduce_jobs.restype = i32
    lib.nlam_wmse_fwd.argtypes = [vp, vp, vp, vp, i64, i32, i32, f32, vp, i32, vp]
    lib.nlam_wmse_fwd.restype = i32
    lib.nlam_wmse_bwd.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, f32, vp, vp]
    lib.nlam_wmse_bwd.restype = i32
    lib.nlam_adamw_step.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, f32, vp]
    lib.nlam_adamw_step.restype = i32
    lib.nlam_adamw_step_resident.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp, vp, f32, vp]
    lib.nlam_adamw_step_resident.restype = i32
    lib.nlam_step_tail_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, i32, i64, i32, i32, vp]
    lib.nlam_step_tail_fwd.restype = i32
    lib.nlam_step_tail_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, i64, i32, i32, vp]
    lib.nlam_step_tail_bwd.restype = i32
    lib.nlam_concat.argtypes = [C.POINTER(Cat), vp]
    lib.nlam_concat.restype = i32
    lib.nlam_window_len.argtypes = [i64, i64, i32, i32, i32]
    lib.nlam_window_len.restype = i64
    lib.nlam_window_batch.argtypes = [C.POINTER(Window), vp]
    lib.nlam_window_batch.restype = i32
    lib.nlam_mlp_group_blocks.argtypes = [C.POINTER(C.c_int64), i32, C.POINTER(C.c_int32)]
    lib.nlam_mlp_group_blocks.restype = i32
    lib.nlam_mlp_fwd_group.argtypes = [C.POINTER(MlpFwd), i32, vp]
    lib.nlam_mlp_fwd_group.restype = i32
    lib.nlam_mlp_bwd_group.argtypes = [C.POINTER(MlpBwd), i32, vp]
    lib.nlam_mlp_bwd_group.restype = i32
    lib.nlam_mlp_fwd_family.argtypes = [C.POINTER(MlpFwd)]
    lib.nlam_mlp_fwd_family.restype = i32
    lib.nlam_mlp_bwd_family.argtypes = [C.POINTER(MlpBwd)]
    lib.nlam_mlp_bwd_family.restype = i32
    lib.nlam_mlp_bwd_group_blocks.argtypes = [C.POINTER(MlpBwd), i32, C.POINTER(i32)]
    lib.nlam_mlp_bwd_group_blocks.restype = i32
    lib.nlam_wgrad_group.argtypes = [C.POINTER(Wgrad), i32, vp]
    lib.nlam_wgrad_group.restype = i32
    lib.nlam_pre_add_supported.argtypes = [C.POINTER(MlpFwd)]
    lib.nlam_pre_add_supported.restype = i32
    lib.nlam_linear.argtypes = [C.POINTER(Linear), vp]
    lib.nlam_linear.restype = i32
    lib.nlam_standardize.argtypes = [C.POINTER(StdJobs), vp]
    lib.nlam_standardize.restype = i32
    if lib.nlam_abi_version() != ABI_VERSION:
        raise RuntimeError("libnlam_hip.so ABI version mismatch")
    if os.environ.get("NLAM_LIN_WGS"):
        lib.nlam_set_tuning(TUNE_LIN_WGS, int(os.environ["NLAM_LIN_WGS"]))
    if os.environ.get("NLAM_WGRAD_BIG_MIN_ROWS"):
        lib.nlam_set_tuning(TUNE_WGRAD_BIG_MIN_ROWS, int(os.environ["NLAM_WGRAD_BIG_MIN_ROWS"]))
    if os.environ.get("NLAM_WGRAD_MIN_PARTS"):
        lib.nlam_set_tuning(TUNE_WGRAD_MIN_PARTS, int(os.environ["NLAM_WGRAD_MIN_PARTS"]))
    if os.environ.get("NLAM_WBF_HALF"):
        check(lib.nlam_set_tuning(TUNE_WBF_HALF, int(os.environ["NLAM_WBF_HALF"])), "nlam_set_tuning(NLAM_WBF_HALF)")
    if os.environ.get("NLAM_WBF_V4"):
        lib.nlam_set_tuning(TUNE_WBF_V4, int(os.environ["NLAM_WBF_V4"]))
    if os.environ.get("NLAM_LIN_GEMM"):
        lib.nlam_set_tuning(TUNE_LIN_GEMM, int(os.environ["NLAM_LIN_GEMM"]))
    if os.environ.get("NLAM_WGRAD_LDMA"):
        check(lib.nlam_set_tuning(TUNE_WGRAD_LDMA, int(os.environ["NLAM_WGRAD_LDMA"])), "nlam_set_tuning(NLAM_WGRAD_LDMA)")
    if os.environ.get("NLAM_WBF_EDGE"):
        check(lib.nlam_set_tuning(TUNE_WBF_EDGE, int(os.environ["NLAM_WBF_EDGE"])), "nlam_set_tuning(NLAM_WBF_EDGE)")
    if os.environ.get("NLAM_WGRAD_LDMA_VAR"):
        check(lib.nlam_set_tuning(TUNE_WGRAD_LDMA_VAR, int(os.environ["NLAM_WGRAD_LDMA_VAR"])), "nlam_set_tuning(NLAM_WGRAD_LDMA_VAR)")
    if os.environ.get("NLAM_CHAIN_CUS"):
        check(lib.nlam_set_tuning(TUNE_CHAIN_CUS, int(os.environ["NLAM_CHAIN_CUS"])), "nlam_set_tuning(NLAM_CHAIN_CUS)")
    if os.environ.get("NLAM_WGRAD_MAX_WGS"):
        check(lib.nlam_set_tuning(TUNE_WGRAD_MAX_WGS, int(os.environ["NLAM_WGRAD_MAX_WGS"])), "nlam_set_tuning(NLAM_WGRAD_MAX_WGS)")
    if os.environ.get("NLAM_WGRAD_CHUNKS"):
        lib.nlam_set_tuning(TUNE_WGRAD_CHUNKS, int(os.environ["NLAM_WGRAD_CHUNKS"]))
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        kind = {-1: "NLAM_EINVAL (bad arguments)", -2: "NLAM_EUNSUP (width not instantiated in this build)"}.get(
            rc, f"hipError_t {rc}"
        )
        raise RuntimeError(f"{what} failed: {kind}")


SLICES = (1, 2, 3, 4, 5)   # -DNLAM_TU=k translation-unit slices of csrc/nlam_hip.hip (see the comment at its top)


def source_stamp() -> str:
    """Hash of the library's sources (csrc/*.hip, csrc/*.inc, include/nlam_hip.h): measurements that belong to one build
    of the kernels (profiles/roundN/pmc_traffic.json) carry it, and bench.py refuses them when it no longer matches."""
    import hashlib

    h = hashlib.sha256()
    csrc = HERE / "csrc"
    for f in sorted([*csrc.glob("*.hip"), *csrc.glob("*.inc"), HERE.parent / "include" / "nlam_hip.h"]):
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def build(verbose: bool = False, out: Path | None = None, defines=(), single_tu: bool | None = None) -> Path:
    """Compile csrc/nlam_hip.hip for gfx950 into the in-tree shared library.

    Default: the five -DNLAM_TU=k slices are compiled in parallel (objects under csrc/_obj/, re-used when neither the
    sources nor the flags changed) and linked.  ``single_tu=True`` (and any build with extra ``defines``, e.g. the
    NLAM_TIMING instrumentation) is the one-command build: one hipcc invocation, everything in one translation unit."""
    import hashlib
    import subprocess
    from concurrent.futures import ThreadPoolExecutor

    csrc = HERE / "csrc"
    src = csrc / "nlam_hip.hip"
    out = Path(out) if out is not None else HERE / "libnlam_hip.so"
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *[f"-D{d}" for d in defines]]
    if single_tu is None:
        single_tu = bool(defines) or os.environ.get("NLAM_SINGLE_TU") == "1"
    if single_tu:
        cmd = [*base, "-shared", str(src), "-o", str(out)]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        return out

    objdir = csrc / "_obj"
    objdir.mkdir(exist_ok=True)
    h = hashlib.sha256(" ".join(base).encode())
    for f in sorted([*csrc.glob("*.hip"), *csrc.glob("*.inc"), HERE.parent / "include" / "nlam_hip.h"]):
        h.update(f.read_bytes())
    stamp = h.hexdigest()[:16]

    def compile_slice(k):
        obj = objdir / f"nlam_tu{k}_{stamp}.o"
        if not obj.exists():
            for old in objdir.glob(f"nlam_tu{k}_*.o"):
                old.unlink()
            cmd = [*base, f"-DNLAM_TU={k}", "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(len(SLICES)) as ex:
        objs = list(ex.map(compile_slice, SLICES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(out)]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out
