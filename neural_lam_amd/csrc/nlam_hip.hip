// nlam_hip.hip -- gfx950 (MI355X, CDNA4) kernels behind include/nlam_hip.h.
//
// Design (DESIGN.md has the derivation and the measurements behind each choice):
//  * d <= 64: one *wave* owns a tile of <= 32 rows (edges in receiver-sorted CSR order, or
//    nodes).  The tile is the N (column) side of the MFMA; the weight matrix is the A
//    operand (rows = output features), staged once per persistent workgroup into LDS.  In
//    this transposed form the accumulator layout of GEMM1 *is* the B-operand layout of
//    GEMM2 (the K order is a permutation applied to the weights when they are staged), so
//    Linear -> SiLU -> Linear -> LayerNorm runs register to register.
//  * Matrix path: fp32 MFMA (v_mfma_f32_32x32x2_f32, exact fmaf chains; it shares the fp32
//    vector lanes with VALU work) or, by default, the bf16 matrix cores with every fp32
//    operand split into 3 bf16 terms (6 MFMAs per product block, fp32 accumulate: fp32-class
//    accuracy, co-issues with VALU).  Kernel families:
//      mlp_fwd_bf_kernel      forward, split-bf16, software-pipelined across tiles
//      mlp_fwd_kernel         forward, fp32 MFMA: FAST shapes and the fully generic path
//      mlp_bwd_fast_kernel    backward, FAST shapes, fp32 or split-bf16
//      mlp_bwd_kernel         backward, generic shapes
//      wgrad_dma_kernel / wgrad_kernel / wgrad_smalln_kernel   weight gradients (LDS-DMA row streaming /
//                             odd widths / 2-3 column inputs), reduce_jobs_kernel their deterministic 2nd stage
//      nlam_wide.inc          64 < d <= 512: workgroup-cooperative forward / backward / weight gradients
//  * Tiles consist of whole receivers (host-built schedule), so the segment reduction (sum /
//    mean aggregation) happens inside the wave through an LDS block with plain stores: no
//    atomics, deterministic.  Receivers with in-degree > 32 are split over several tiles and
//    only those use atomics.
//  * Every row store goes through a per-wave [32][36] LDS block and leaves as whole 128-B lines.
//
// "chunk" below always means: 4 consecutive features [8t + 4hi, 8t + 4hi + 4) of
// one row, held by lane (j = lane & 31, hi = lane >> 5) as one float4.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/nlam_hip.h"
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

// ---------------------------------------------------------------------------
// Translation-unit slices.  Every kernel here is a template that is instantiated only where a launcher names it, so
// the same source can be compiled several times with -DNLAM_TU=k, each time emitting one family's launchers (and with
// them that family's kernels): the four objects build in parallel and link into one library (neural_lam_amd/_lib.py).
// NLAM_TU undefined or 0 = everything in one translation unit (the one-command build of INTEGRATION.md, and the
// -DNLAM_TIMING build, whose counters live in one device variable).
//   1  C-ABI entry points, argument checks, HBM-bound helper kernels, narrow (d <= 64) forward
//   2  narrow backward + weight gradients
//   3  fp32 MFMA kernels for 64 < d <= 512 (nlam_wide.inc)
//   4  split-bf16 kernels for 64 < d <= 512 (nlam_wbf.inc)
//   5  node-level linear kernels of the factorised edge MLP (nlam_linear)
// ---------------------------------------------------------------------------
#ifndef NLAM_TU
#define NLAM_TU 0
#endif
#ifndef NLAM_RES_PRE
#define NLAM_RES_PRE 3   // narrow forward with a residual stash: input units prefetched a tile ahead (4 spilled in the tile loop;
                         // A/B at cfg2: 3 vs 4 vs 2 = same training step, +2.5 % forecast rate for 3)
#endif
#ifndef NLAM_BWD_G_BATCH
#define NLAM_BWD_G_BATCH 2   // narrow backward, requests of the g_out / aggregated-gradient / xhat rows: 2 = the whole tile back to back,
                             // 1 = one 32-column block at a time, 0 = one 16-byte chunk at a time (unconditional either way)
#endif
#ifndef NLAM_LW_PREFETCH_X
#define NLAM_LW_PREFETCH_X 1   // grouped embedder backward: the next tile's xhat rows requested a tile ahead (1) or this tile's at the tile
                               // top (0: 438 instead of 461 registers -- a wgrad_dma<1> wave then fits next to it, reduce_jobs still
                               // does not -- and one request batch exposed per tile; cfg2 1.767-1.784 against 1.777-1.783 ms: noise)
#endif
#ifndef NLAM_BWD_Z_EARLY
#define NLAM_BWD_Z_EARLY 1   // z1 rows requested ahead of the dz2 stores (1) or one block at a time where they are used (0)
#endif
#ifndef NLAM_BWD_R_EARLY
#define NLAM_BWD_R_EARLY 0   // residual (d src0 += g_out) rows requested ahead of the dh GEMM (1), right ahead of the dx GEMM of source 0 (2),
                             // ahead of the dh GEMM as that GEMM's initial accumulators (3), or chunk by chunk behind the dx GEMM (0).
                             // A/B at cfg2, five builds inside one gpurun call: G_BATCH 0 / 1 / 2 and Z_EARLY 0 / 1 are within noise of each
                             // other and 2 % ahead of the branchy loads; R_EARLY = 1 makes the isolated kernel 10 % faster and the captured
                             // step 6 % SLOWER (1.80 -> 1.91 ms): 225 instead of 196 VGPRs.  A chain workgroup is 2 waves per SIMD; at
                             // 196 (allocated 200) it leaves 112 registers per SIMD, enough for a wave of wgrad_dma<1> / <2> (56 / 93),
                             // reduce_jobs (78) or segment_sum (54) of a side stream to share the CU; at 225 it leaves 48: nothing fits.
                             // 2: 225 VGPRs as well; 3: 213 VGPRs, 1.742-1.747 against 1.726-1.740 ms for 0 on one box.
#endif
#ifndef NLAM_LIN_SK
#define NLAM_LIN_SK 4   // K = 16 steps per chunk of the one-term 64-row-tile nlam_linear GEMM (2: the 32-column chunks of round 5; A/B builds)
#endif
#define NLAM_IN_TU(k) (NLAM_TU == 0 || NLAM_TU == (k))

namespace nlam_detail {   // launchers: external linkage, each defined in exactly one slice; arguments are already validated
int32_t fwd_narrow(const nlam_mlp_fwd_t* p, hipStream_t stream);   // slice 1
int32_t bwd_narrow(const nlam_mlp_bwd_t* p, hipStream_t stream);   // slice 2
int32_t wgrad_narrow(const nlam_wgrad_t* p, hipStream_t stream);   // slice 2
int32_t fwd_wide(const nlam_mlp_fwd_t* p, hipStream_t stream);     // slice 3
int32_t bwd_wide(const nlam_mlp_bwd_t* p, hipStream_t stream);     // slice 3
int32_t fwd_wide_group(const nlam_mlp_fwd_t* ps, int n, hipStream_t stream);   // slice 3
int32_t bwd_wide_group(const nlam_mlp_bwd_t* ps, int n, hipStream_t stream);   // slice 3
int32_t fwd_check(const nlam_mlp_fwd_t* p);                        // argument checks of nlam_mlp_fwd (slice 1)
int32_t bwd_check(const nlam_mlp_bwd_t* p);                        // argument checks of nlam_mlp_bwd (slice 1)
int32_t wgrad_wide(const nlam_wgrad_t* p, hipStream_t stream);     // slice 3
int32_t fwd_wbf(const nlam_mlp_fwd_t* p, hipStream_t stream);      // slice 4
int32_t bwd_wbf(const nlam_mlp_bwd_t* p, hipStream_t stream);      // slice 4
int32_t wgrad_wbf(const nlam_wgrad_t* p, hipStream_t stream);      // slice 4
int32_t wgrad_wbf_group(const nlam_wgrad_t* ps, int n, hipStream_t stream);   // slice 4
extern int wbf_min_supertiles;                                     // nlam_set_tuning (defined in slice 1)
extern int wbf_half;                                               // nlam_set_tuning (defined in slice 1)
extern int wbf_v4;                                                 // branch-free chunk accessors in the split-bf16 wide kernels (NLAM_TUNE_WBF_V4)
extern int wgrad_chunks_per_wg;                                    // nlam_set_tuning (defined in slice 1)
extern int wgrad_ldma;                                             // one-term weight gradients on wgrad_ldma_kernel (NLAM_TUNE_WGRAD_LDMA)
extern int chain_cus;                                              // persistent workgroups (x occupancy) of a wide fused-MLP launch (NLAM_TUNE_CHAIN_CUS)
extern int wgrad_max_wgs;                                          // workgroups of a big split-bf16 weight gradient (NLAM_TUNE_WGRAD_MAX_WGS)
extern int wbf_edge;                                               // mlp_fwd_edge_kernel for the factorised one-term edge layers (NLAM_TUNE_WBF_EDGE)
extern int wgrad_ldma_var;                                         // its (rows per stage, ring depth) variant (NLAM_TUNE_WGRAD_LDMA_VAR)
extern int wgrad_min_parts;                                        // nlam_set_tuning (defined in slice 1)
extern int wgrad_min_parts_wide;                                   // the same for weight matrices of more than 128 rows
extern int wgrad_big_min_rows;                                     // nlam_set_tuning (defined in slice 1)
extern int lin_resident_wgs;                                       // workgroups of a resident-weight nlam_linear launch (slice 1)
extern int lin_gemm;                                               // nlam_linear on the LDS-tiled GEMM where it applies (NLAM_TUNE_LIN_GEMM)
extern long lin_gemm_big_rows;                                     // rows from which it uses 128-row tiles
}  // namespace nlam_detail

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// ---------------------------------------------------------------------------
// optional per-phase cycle accounting (tools/phase_timing.py builds a second
// library with -DNLAM_TIMING; the shipped library has none of this)
// ---------------------------------------------------------------------------
#ifdef NLAM_TIMING
__device__ unsigned long long g_phase_cycles[16];
#define NLAM_T_DECL unsigned long long t_prev_ = __builtin_amdgcn_s_memtime(); \
    unsigned int t_a0_ = 0, t_a1_ = 0, t_a2_ = 0, t_a3_ = 0, t_a4_ = 0, t_a5_ = 0, t_a6_ = 0, t_a7_ = 0, t_a8_ = 0, t_a9_ = 0, t_a10_ = 0, t_a11_ = 0;
#define NLAM_T_DRAIN asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#define NLAM_T_MARK(k) { const unsigned long long t_now_ = __builtin_amdgcn_s_memtime(); t_a##k##_ += (unsigned int)(t_now_ - t_prev_); t_prev_ = t_now_; }
#define NLAM_T_FLUSH(ntiles_) if ((threadIdx.x & 63) == 0) { \
    atomicAdd(&g_phase_cycles[0], (unsigned long long)t_a0_); atomicAdd(&g_phase_cycles[1], (unsigned long long)t_a1_); \
    atomicAdd(&g_phase_cycles[2], (unsigned long long)t_a2_); atomicAdd(&g_phase_cycles[3], (unsigned long long)t_a3_); \
    atomicAdd(&g_phase_cycles[4], (unsigned long long)t_a4_); atomicAdd(&g_phase_cycles[5], (unsigned long long)t_a5_); \
    atomicAdd(&g_phase_cycles[6], (unsigned long long)t_a6_); atomicAdd(&g_phase_cycles[7], (unsigned long long)t_a7_); \
    atomicAdd(&g_phase_cycles[8], (unsigned long long)t_a8_); atomicAdd(&g_phase_cycles[9], (unsigned long long)t_a9_); \
    atomicAdd(&g_phase_cycles[10], (unsigned long long)t_a10_); atomicAdd(&g_phase_cycles[11], (unsigned long long)t_a11_); \
    atomicAdd(&g_phase_cycles[12], (unsigned long long)(ntiles_)); atomicAdd(&g_phase_cycles[13], 1ULL); }
#else
#define NLAM_T_DECL
#define NLAM_T_DRAIN
#define NLAM_T_MARK(k)
#define NLAM_T_FLUSH(ntiles_)
#endif

namespace {

// ---------------------------------------------------------------------------
// roctx ranges around every launching entry point (the reference has no tracing of its own; rocprofv3
// --marker-trace shows these next to the kernels).  The marker library is looked up at run time -- the product does not link
// against the profiler -- and only when NLAM_ROCTX=1: otherwise a range is one predictable branch.
// ---------------------------------------------------------------------------
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char* on = getenv("NLAM_ROCTX");
        if (on == nullptr || on[0] != '1') return;
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (h == nullptr) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (h == nullptr) return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (push == nullptr || pop == nullptr) push = nullptr, pop = nullptr;
    }
};
inline const Roctx& roctx() {
    static const Roctx r;
    return r;
}
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char* name) : on(roctx().push != nullptr) {
        if (on) roctx().push(name);
    }
    ~RoctxRange() {
        if (on) roctx().pop();
    }
};
#define NLAM_RANGE(name) RoctxRange nlam_range_(name)

constexpr int kWavesPerBlock = 8;                  // backward kernels: 512 threads, two waves per SIMD
#ifndef NLAM_FWD_WAVES
#define NLAM_FWD_WAVES 8
#endif
constexpr int kFwdWaves = NLAM_FWD_WAVES;          // forward kernels: most waves per workgroup (small launches use fewer, see fwd_launch_shape)
constexpr int kFwdThreads = kFwdWaves * 64;
constexpr int kStgStride = 36;                     // per-wave [32][36] staging block (32 columns + 4 pad)
constexpr int kBlockThreads = kWavesPerBlock * 64;
constexpr int kNumCUs = 256;
constexpr int kMaxGridBlocks = kNumCUs;            // one persistent workgroup per CU
constexpr int kMaxWidth = 64;                      // widest hidden/output width of the narrow (weights-in-LDS) kernels

__host__ __device__ __forceinline__ int round_up(int x, int m) { return (x + m - 1) / m * m; }

// XCD-aware tile placement (round 5).  Workgroup b of a grid runs on XCD b % 8 (observed on gfx950, not promised: used for
// speed only -- MI355X_MICROARCH.md "Workgroup dispatch") and every XCD has its own 4 MiB L2.  The tile schedules below deal
// tile t to "slot" t % g; with slot == blockIdx.x consecutive tiles -- consecutive receivers of the CSR order, whose senders are
// their neighbours on the mesh -- went to eight different L2s and every XCD fetched the whole node table (TCC_HIT 55 %, 1.36x the
// algorithmic HBM bytes on the m2m edge backward).  xcd_slot gives the workgroups of XCD x the contiguous slot
// range [x g / 8, (x + 1) g / 8): an XCD then works on an eighth of the receivers (per wave round).
// MEASURED AND SWITCHED OFF (round 5, A/B of two builds inside one gpurun call, profiles/round5/ab_xcd_gemm_modes.log): cfg2
// 1.749 / 1.750 ms without against 1.751 / 1.760 with, cfg3 47.6 against 48.5, cfg4 11.25 against 11.30, cfg5 143.6 against 144.5 --
// no gain anywhere, a little loss at the wide widths (an eighth of the receivers per XCD is also an eighth of the senders'
// rows per L2: the node tables of these graphs, 1.7 - 13 MB, mostly fit the 4 MiB L2s or the Infinity Cache either way, and the
// kernels are latency-bound, not L2-fill-bound).  -DNLAM_XCD_REMAP=1 rebuilds with it.
#ifndef NLAM_XCD_REMAP
#define NLAM_XCD_REMAP 0
#endif
__device__ __forceinline__ int xcd_slot(int b, int g) {
#if NLAM_XCD_REMAP
    const int x = b & 7, base = g >> 3, rem = g & 7;   // XCD y owns base + (y < rem) workgroups: b = y, y + 8, ...
    return x * base + (x < rem ? x : rem) + (b >> 3);
#else
    return b;
#endif
}

// Order the wave's own LDS traffic (cross-lane exchange through LDS inside one
// wave; other waves of the block are at unrelated points, so no s_barrier).
__device__ __forceinline__ void wave_lds_sync() {
    // LDS instructions of one wave are issued and serviced in order, so a wave-scope
    // fence restricted to the LDS address space is all the ordering needed; unlike a
    // "memory" clobber it lets the compiler keep global loads in flight across it.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// ---------------------------------------------------------------------------
// packed A-operand staging
//   dst[(((mb * T + t) * 2 + hi) * 32 + i) * 4 + c] = A[mb*32 + i][8*t + 4*hi + c]
//   with A[m][k] = W[m * ldm + k * ldk] for m < M, k < Kw (zero elsewhere); the
//   Kw columns are placed at chunk offset t0 of a matrix that has T chunks total.
// ---------------------------------------------------------------------------
__device__ void stage_packed(float* dst, int T, int t0, const float* W, long ldm, long ldk, int M, int MB, int Kw) {
    const int nt = (Kw + 7) >> 3;
    if (ldk == 1 && (Kw & 3) == 0 && (ldm & 3) == 0 && ((uintptr_t)W & 15) == 0) {
        // row-major weights: consecutive threads walk along a row (coalesced 16-B reads);
        // float4 q of row m holds k = 4q .. 4q+3 = chunk (t = q >> 1, hi = q & 1), c = 0..3
        const int q_per_row = nt * 2;
        const int total = MB * 32 * q_per_row;
        for (int s = threadIdx.x; s < total; s += blockDim.x) {
            const int q = s % q_per_row;
            const int m = s / q_per_row;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < M && 4 * q < Kw) v = *reinterpret_cast<const f32x4*>(W + (long)m * ldm + 4 * q);
            const int mb = m >> 5, i = m & 31, t = q >> 1, hi = q & 1;
            *reinterpret_cast<f32x4*>(&dst[((((size_t)mb * T + (t0 + t)) * 2 + hi) * 32 + i) * 4]) = v;
        }
        return;
    }
    const int total = MB * nt * 64;
    for (int s = threadIdx.x; s < total; s += blockDim.x) {
        const int i = s & 31;
        const int hi = (s >> 5) & 1;
        const int rest = s >> 6;
        const int t = rest % nt;
        const int mb = rest / nt;
        const int m = mb * 32 + i;
        const int kb = 8 * t + 4 * hi;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < M) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = kb + c;
                if (k < Kw) v[c] = W[(long)m * ldm + (long)k * ldk];
            }
        }
        *reinterpret_cast<f32x4*>(&dst[((((size_t)mb * T + (t0 + t)) * 2 + hi) * 32 + i) * 4]) = v;
    }
}

__device__ void stage_vec(float* dst, const float* v, int n, int np, float fill) {
    for (int s = threadIdx.x; s < np; s += blockDim.x) dst[s] = (v != nullptr && s < n) ? v[s] : fill;
}

// four MFMAs (K = 8 features) against every 32-row block of the packed A matrix
template <int MB>
__device__ __forceinline__ void mma_chunk(f32x16 (&acc)[MB], const float* Ap, int T, int t, f32x4 x, int lane) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(&Ap[(((size_t)mb * T + t) * 64 + lane) * 4]);
        acc[mb] = MFMA32(a[0], x[0], acc[mb]);
        acc[mb] = MFMA32(a[1], x[1], acc[mb]);
        acc[mb] = MFMA32(a[2], x[2], acc[mb]);
        acc[mb] = MFMA32(a[3], x[3], acc[mb]);
    }
}
// note: (((mb*T + t)*2 + hi)*32 + i) == ((mb*T + t)*64 + lane) because lane = hi*32 + i.

// chunk t of one row (row pointer already resolved); zero outside [0, w)
// (c0 is made opaque: `c0 < w` is invariant across the tiles of a launch, and hoisted out of the tile loops these compares
// become hundreds of scalar masks spilled to VGPR lanes -- a v_readlane pair plus hazard slots at every use instead of one v_cmp)
__device__ __forceinline__ f32x4 load_chunk(const float* row, int w, int t, int hi, bool valid) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    int c0 = 8 * t + 4 * hi;
    asm volatile("" : "+v"(c0));
    if (valid && c0 < w) {
        if ((w & 3) == 0) {
            v = *reinterpret_cast<const f32x4*>(row + c0);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c0 + c < w) v[c] = row[c0 + c];
        }
    }
    return v;
}

__device__ __forceinline__ void store_chunk(float* row, int w, int t, int hi, bool valid, f32x4 v) {
    int c0 = 8 * t + 4 * hi;
    asm volatile("" : "+v"(c0));
    if (valid && c0 < w) {
        if ((w & 3) == 0) {
            *reinterpret_cast<f32x4*>(row + c0) = v;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c0 + c < w) row[c0 + c] = v[c];
        }
    }
}

// 16 zero bytes: what an absent tensor / a lane outside its row reads, so that a load can stay unconditional (finding 26)
__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// V4 = every width of the launch is a multiple of 4 (compile time, chosen by the launcher): a chunk is one 16-byte access and
// the load is UNCONDITIONAL -- a lane outside its row, or whose tensor is absent, reads the zero buffer through a pointer select.
// The generic load_chunk is a lane-predicated branch around the access with the element-wise fallback compiled into every call
// site: an exec-mask save / restore per chunk, a join at which the wait-counter pass drains the load queue (finding 26), and
// hundreds of hoisted scalar masks (finding 29).  Round 5, the split-bf16 wide kernels.
template <bool V4>
__device__ __forceinline__ f32x4 load_chunk_v(const float* row, int w, int t, int hi, bool valid) {
    if constexpr (!V4) {
        return load_chunk(row, w, t, hi, valid);
    } else {
        int c0 = 8 * t + 4 * hi;
        asm volatile("" : "+v"(c0));
        const float* q = (valid && c0 < w) ? row + c0 : g_zero16;
        return *reinterpret_cast<const f32x4*>(q);
    }
}

template <bool V4>
__device__ __forceinline__ void store_chunk_v(float* row, int w, int t, int hi, bool valid, f32x4 v) {
    if constexpr (!V4) {
        store_chunk(row, w, t, hi, valid, v);
    } else {
        int c0 = 8 * t + 4 * hi;
        asm volatile("" : "+v"(c0));
        if (valid && c0 < w) *reinterpret_cast<f32x4*>(row + c0) = v;
    }
}

__device__ __forceinline__ f32x4 acc_chunk(const f32x16& a, int tt) {
    f32x4 v;
    v[0] = a[4 * tt + 0];
    v[1] = a[4 * tt + 1];
    v[2] = a[4 * tt + 2];
    v[3] = a[4 * tt + 3];
    return v;
}

// v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division sequence
__device__ __forceinline__ float silu_f(float z) { return z * __builtin_amdgcn_rcpf(1.f + __expf(-z)); }
__device__ __forceinline__ float silu_grad_f(float z) {
    const float s = __builtin_amdgcn_rcpf(1.f + __expf(-z));
    return s * (1.f + z * (1.f - s));
}

// sum over the two half-wave partners (lane, lane ^ 32): full-row reductions
__device__ __forceinline__ float row_allreduce(float v) { return v + __shfl_xor(v, 32, 64); }

struct TileInfo {
    int row0, nrows, seg0, nseg;
    bool split;
};

__device__ __forceinline__ TileInfo get_tile(const nlam_tile_t* tiles, int ti, int rows) {
    TileInfo t;
    if (tiles != nullptr) {
        const nlam_tile_t d = tiles[ti];
        t.row0 = d.row0;
        t.nrows = d.nrows;
        t.seg0 = d.seg0;
        t.split = (d.nseg & NLAM_TILE_SPLIT) != 0;
        t.nseg = d.nseg & ~NLAM_TILE_SPLIT;
    } else {
        t.row0 = ti * 32;
        t.nrows = min(32, rows - t.row0);
        t.seg0 = t.row0;
        t.nseg = t.nrows;
        t.split = false;
    }
    return t;
}

// Segment-sum the wave's staged tile (stg[row][col], stride S) over the tile's
// receivers and write (or atomically add, for split receivers) the result.
__device__ __forceinline__ void tile_segment_reduce(const float* stg, int S, const TileInfo& tl, const int32_t* rowptr,
                                                    const float* inv_deg, float* out_b /* (nseg_total, w) */, int w,
                                                    int wb /* columns staged */, int lane) {
    // one coalesced fetch of the tile's <= 33 row pointers (and scales); broadcast per segment below
    int my_ptr = 0;
    float my_scale = 1.f;
    if (!tl.split && lane <= tl.nseg) my_ptr = rowptr[tl.seg0 + lane] - tl.row0;
    if (inv_deg != nullptr && lane < tl.nseg) my_scale = inv_deg[tl.seg0 + lane];
    for (int sg = 0; sg < tl.nseg; ++sg) {
        const int r = tl.seg0 + sg;
        int lo = __shfl(my_ptr, sg, 64), hi_ = __shfl(my_ptr, sg + 1, 64);
        if (tl.split) {
            lo = 0;
            hi_ = tl.nrows;
        }
        const float scale = __shfl(my_scale, sg, 64);
        for (int c = lane; c < wb; c += 64) {
            float s = 0.f;
            for (int q = lo; q < hi_; ++q) s += stg[q * S + c];
            s *= scale;
            float* dst = out_b + (size_t)r * w + c;
            if (tl.split)
                atomicAdd(dst, s);
            else
                *dst = s;
        }
    }
}

// A 32-column block of the wave's tile, staged as stg[row][36], -> global rows with
// whole-line coalescing: lane -> (row, float4 column), 8 rows x 128 B per store instruction
// (the accumulator layout itself would store 32-B pieces of 32 different rows).
// wb = real columns of the block (multiple of 4), dst(r) = pointer to the block's first
// column in the destination row of tile-row r.
template <typename FD>
__device__ __forceinline__ void block_rows_out(const float* stg, int nrows, int wb, int lane, FD&& dst) {
    const int vpr = wb >> 2;
    const int total = nrows * vpr;
    // wave-uniform trip count: dst() may shuffle across lanes, so every lane evaluates it
    for (int base = 0; base < total; base += 64) {
        const int item = base + lane;
        const int r = min((vpr == 8) ? (item >> 3) : item / vpr, 31);
        const int c4 = item - r * vpr;
        float* d = dst(r);
        if (item < total) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&stg[r * kStgStride + 4 * c4]);
            *reinterpret_cast<f32x4*>(d + 4 * c4) = v;
        }
    }
}

// Segment-sum of a staged 32-column block over the tile's receivers.  Lane l holds
// raw_ptr = rowptr[seg0 + l] (l <= nseg) and the scale of segment l in registers, so there
// is no dependent global load here.  lane -> (segment of this pass, float4 column).
__device__ __forceinline__ void block_segment_reduce(const float* stg, const TileInfo& tl, int raw_ptr, float my_scale,
                                                     float* out_b /* + column offset */, int w, int wb, int lane) {
    const int vpr = wb >> 2;
    const int spp = 64 / vpr;                    // segments per pass
    const int sl = (vpr == 8) ? (lane >> 3) : lane / vpr;
    const int c4 = lane - sl * vpr;
    for (int s0 = 0; s0 < tl.nseg; s0 += spp) {
        const int sg = s0 + sl;
        int lo = __shfl(raw_ptr, sg & 63, 64) - tl.row0, hi_ = __shfl(raw_ptr, (sg + 1) & 63, 64) - tl.row0;
        const float scale = __shfl(my_scale, sg & 63, 64);
        if (tl.split) {
            lo = 0;
            hi_ = tl.nrows;
        }
        if (sl < spp && sg < tl.nseg) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            for (int q = lo; q < hi_; ++q) s += *reinterpret_cast<const f32x4*>(&stg[q * kStgStride + 4 * c4]);
            s *= scale;
            float* dst = out_b + (size_t)(tl.seg0 + sg) * w + 4 * c4;
            if (tl.split) {
#pragma unroll
                for (int c = 0; c < 4; ++c) atomicAdd(dst + c, s[c]);
            } else {
                *reinterpret_cast<f32x4*>(dst) = s;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// forward:  [gather | concat] -> Linear -> SiLU -> Linear -> [LayerNorm]
//           -> [+src1] -> {segment-reduce -> aggr} -> [+src0] -> out
// HB = padded hidden width / 32, OB = padded output width / 32
//
// Latency hiding is by occupancy: 16 waves per workgroup = 4 per SIMD (<= 128 VGPRs), one
// tile per wave at a time.  A tile's life has ~16k cycles of MFMA issue and several
// dependent memory waits (descriptor -> gather index -> rows; gfx950 retires loads and
// stores through ONE in-order counter, so every load behind a store also waits for that
// store's L2 acknowledgement, ~5-20k cycles here); with two waves per SIMD (round-1 design)
// the MFMA pipe idled 60 % of the time, measured with tools/phase_timing.py.
// Inputs stream through two half-source register buffers (16 VGPRs each); outputs leave
// through a per-wave [32][36] LDS block so every store instruction writes whole 128-B lines.
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// split-bf16 matrix path ("NS" terms per fp32 value)
//
// v_mfma_f32_32x32x2_f32 runs on the SIMD's fp32 vector lanes: it issues at the VALU rate
// and does NOT overlap with VALU work (tools/probes/mfma_probe.hip: MFMA time + VALU time add
// up at any occupancy).  The bf16 matrix cores are 16x faster and do co-issue with VALU.
// So an fp32 operand x is split into NS bf16 terms x = x_0 + x_1 (+ x_2), each the
// round-to-nearest bf16 of the remainder, and a product block becomes the MFMAs
// sum_{p+q < NS} A_p B_q accumulated in fp32:
//   NS = 1   plain bf16 operands (autocast-like),         1 MFMA / block-step
//   NS = 2   ~2^-16 relative product error,               3 MFMAs
//   NS = 3   ~2^-24 (fp32 class; drops only 2^-24 terms), 6 MFMAs = 192 cycles per K = 16
//            vs 8 fp32 MFMAs = 512 cycles.
// K order inside a 16-wide step is "slot" order: lane (j, hi) supplies slots q = 0..7 of
// k = 8*hi + q; GEMM2 takes its B operand straight from GEMM1's accumulators by permuting
// the K order of W2 when it is staged (slot q of half h of block hb <-> feature
// 32*hb + 16*h + (q & 3) + 8*(q >> 2) + 4*hi).
// ---------------------------------------------------------------------------
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, (a)), __builtin_bit_cast(s16x8, (b)), (c), 0, 0, 0)

template <int NS>
struct BfFrag {
    u32x4 t[NS];   // term p: 8 bf16 = slots 0..7
};

// two fp32 values -> NS packed bf16 pairs (v_cvt_pk_bf16_f32, shift/and, v_pk_add_f32 per extra term)
template <int NS>
__device__ __forceinline__ void split_pair(float a, float b, unsigned (&out)[NS]) {
    f32x2 v = {a, b};
#pragma unroll
    for (int p = 0; p < NS; ++p) {
        const unsigned bits = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
        out[p] = bits;
        if (p + 1 < NS) {
            v[0] -= __builtin_bit_cast(float, bits << 16);
            v[1] -= __builtin_bit_cast(float, bits & 0xffff0000u);
        }
    }
}

template <int NS>
__device__ __forceinline__ BfFrag<NS> split8(const float (&x)[8]) {
    BfFrag<NS> f;
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
        unsigned o[NS];
        split_pair<NS>(x[2 * pr], x[2 * pr + 1], o);
#pragma unroll
        for (int p = 0; p < NS; ++p) f.t[p][pr] = o[p];
    }
    return f;
}

// acc[blk] += sum_{p+q < NS} A_p(blk) B_q for NB blocks sharing one B fragment.
//  * block index innermost: consecutive MFMAs go to different accumulators.  A chain of
//    dependent MFMAs on ONE accumulator is issued ahead of its execution and fetches its A/B
//    registers late; VALU code that re-uses those registers for the next fragment then
//    corrupted rows 16..31 of a tile now and again (run-to-run differences of bf16-level size,
//    found with tools/determinism_check*.py).  With a lone accumulator every MFMA is followed
//    by 32 wait states instead.
//  * all A terms of a step are fetched into distinct registers before its first MFMA.
template <int NS, int NB>
__device__ __forceinline__ void mma_split_lds(f32x16 (&acc)[NB], const u32x4* Wp, int MB, int S, int st, int lane,
                                              const BfFrag<NS>& B) {
    // every A term of the step in its own registers before the first MFMA: nothing an in-flight MFMA reads is
    // rewritten until the whole group has been issued.  NB is exact (no run-time predicate on the MFMAs: a
    // conditional accumulator update makes the compiler copy the 16-register accumulators around).
    u32x4 a[NS][NB];
#pragma unroll
    for (int pa = 0; pa < NS; ++pa)
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) a[pa][blk] = Wp[(((size_t)pa * MB + blk) * S + st) * 64 + lane];
#pragma unroll
    for (int ord = NS - 1; ord >= 0; --ord)
#pragma unroll
        for (int pa = 0; pa <= ord; ++pa)
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                acc[blk] = MFMA_BF16(a[pa][blk], B.t[ord - pa], acc[blk]);
                if (NB == 1) {
                    asm volatile("s_nop 15");
                    asm volatile("s_nop 15");
                }
            }
}

// Stage an (M x K) weight slice as split-bf16 A fragments:
//   dst[((term * MB + mb) * S + s0 + s) * 64 + lane] (16 B) = slots q = 0..7 of row mb*32 + (lane & 31),
//   k = perm2 ? 32*(s>>1) + 16*(s&1) + (q&3) + 8*(q>>2) + 4*hi : 16*s + 8*hi + q ;  A[m][k] = W[m*ldm + k]
// (tid, nthr): the caller's thread index / thread count -- a workgroup staging into its LDS passes (threadIdx.x, blockDim.x),
// the pack kernel (nlam_mlp_pack: the same images written ONCE per optimizer step into global memory) its grid-wide ones.
template <int NS>
__device__ void stage_split_impl(u32x4* dst, int S, int s0, const float* W, long ldm, int M, int MB, int K, bool perm2, long ldk,
                                 int Kpad, int tid, int nthr) {
    const int nst = (Kpad > 0 ? Kpad : K) >> 4;   // steps written (columns k >= K are zero)
    const int total = MB * nst * 64;
    for (int idx = tid; idx < total; idx += nthr) {
        const int lane = idx & 63;
        const int rest = idx >> 6;
        const int st = rest % nst, mb = rest / nst;
        const int i = lane & 31, hi = lane >> 5;
        const int m = mb * 32 + i;
        float x[8];
        // slots 0..3 and 4..7 are each 4 consecutive k: with unit k stride and 16-B aligned rows that is two dwordx4 loads
        const int kA = perm2 ? 32 * (st >> 1) + 16 * (st & 1) + 4 * hi : 16 * st + 8 * hi;
        const int kB = perm2 ? kA + 8 : kA + 4;
        const float* rowp = W + (long)m * ldm;
        if (ldk == 1 && m < M && kB + 4 <= K && ((ldm & 3) == 0) && ((reinterpret_cast<uintptr_t>(W) & 15) == 0)) {
            const f32x4 va = *reinterpret_cast<const f32x4*>(rowp + kA);
            const f32x4 vb = *reinterpret_cast<const f32x4*>(rowp + kB);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                x[c] = va[c];
                x[4 + c] = vb[c];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = (q < 4 ? kA : kB) + (q & 3);
                x[q] = (m < M && k < K) ? W[(long)m * ldm + (long)k * ldk] : 0.f;
            }
        }
        const BfFrag<NS> f = split8<NS>(x);
#pragma unroll
        for (int p = 0; p < NS; ++p) dst[(((size_t)p * MB + mb) * S + s0 + st) * 64 + lane] = f.t[p];
    }
}

template <int NS>
__device__ __forceinline__ void stage_split(u32x4* dst, int S, int s0, const float* W, long ldm, int M, int MB, int K, bool perm2,
                                            long ldk = 1, int Kpad = 0) {
#ifdef NLAM_EXPERIMENT_NOSTAGE   // tools/ab experiments only: how much of a launch is weight staging (results are garbage)
    return;
#endif
    stage_split_impl<NS>(dst, S, s0, W, ldm, M, MB, K, perm2, ldk, Kpad, (int)threadIdx.x, (int)blockDim.x);
}

// Pre-packed weights (nlam_mlp_pack): the split-bf16 A fragments of an MLP exactly as the narrow kernels lay them out in LDS,
// written once per optimizer step; a workgroup then fetches its weights by LDS-DMA (global_load_lds_dwordx4: 1 KiB per
// wave-instruction, no staging registers, no conversion work) instead of loading fp32 rows and splitting them itself.
// `bytes` is a multiple of 1024 (every image piece is: 2048 * NS * blocks * steps bytes); dst / src are 16-byte aligned; the
// caller's next __syncthreads() (which drains the wave's LDS-DMA: vmcnt(0)) publishes the data.
__device__ __forceinline__ void lds_dma_copy(float* dst_lds, const float* src, size_t bytes) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = (int)(blockDim.x >> 6);
    const int npieces = (int)(bytes >> 10);
    for (int c = wave; c < npieces; c += nw)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)c * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(dst_lds + (size_t)c * 256), 16, 0, 0);
}

// FAST = every source width is a multiple of 8, hid == 32*HB and dout == 32*OB: no
// per-element bounds, no scalar tails, no masked loads (rows past the tile's end read a
// clamped row and are discarded) -- the shape of every InteractionNet / PropagationNet
// layer and of the d -> d -> d grid MLPs.  The generic instantiation keeps all checks.
// NS = 0: fp32 MFMA (exact fmaf chains); NS > 0: split-bf16 matrix path (FAST shapes whose
// source widths are multiples of 32), see above.
template <int HB, int OB, bool FAST, int NS>
__global__ __launch_bounds__(kFwdThreads) void mlp_fwd_kernel(const nlam_mlp_fwd_t p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int DPH = HB * 32, OP = OB * 32;
    constexpr int NSW = NS > 0 ? NS : 1;
    static_assert(NS == 0 || FAST, "the split-bf16 path covers FAST shapes only");

    int T1 = 0;
    for (int s = 0; s < p.nsrc; ++s) T1 += (p.src[s].width + 7) >> 3;
    constexpr int T2 = DPH / 8;
    const int S1 = T1 >> 1;                           // K = 16 steps of GEMM1 (split path)
    constexpr int S2 = DPH / 16;

    // weights: fp32 packed (NS = 0: DPH*8*T1 + OP*DPH floats) or NS split-bf16 copies (half the bytes each)
    float* W1p = smem;
    float* W2p = W1p + (NS > 0 ? (size_t)NS * DPH * 8 * T1 / 2 : (size_t)DPH * 8 * T1);
    float* b1l = W2p + (NS > 0 ? (size_t)NS * OP * DPH / 2 : (size_t)OP * DPH);   // DPH
    u32x4* W1s = reinterpret_cast<u32x4*>(W1p);       // [NS][HB][S1][64] x 16 B
    u32x4* W2s = reinterpret_cast<u32x4*>(W2p);       // [NS][OB][S2][64] x 16 B
    float* b2l = b1l + DPH;                           // OP
    float* gml = b2l + OP;                            // OP
    float* btl = gml + OP;                            // OP
    float* stg_all = btl + OP;                        // kFwdWaves x 32 x kStgStride
    NLAM_T_DECL
    int kin = 0;
    for (int s = 0; s < p.nsrc; ++s) kin += p.src[s].width;
    if constexpr (NS > 0) {
        int s0 = 0, off = 0;
        for (int s = 0; s < p.nsrc; ++s) {
            const int w = p.src[s].width;
            stage_split<NSW>(W1s, S1, s0, p.W1 + off, kin, p.hid, HB, w, false);
            off += w;
            s0 += w >> 4;
        }
        stage_split<NSW>(W2s, S2, 0, p.W2, p.hid, p.dout, OB, p.hid, true);
    } else {
        int t0 = 0, off = 0;
        for (int s = 0; s < p.nsrc; ++s) {
            const int w = p.src[s].width;
            stage_packed(W1p, T1, t0, p.W1 + off, kin, 1, p.hid, HB, w);
            off += w;
            t0 += (w + 7) >> 3;
        }
        stage_packed(W2p, T2, 0, p.W2, p.hid, 1, p.dout, OB, p.hid);
    }
    if constexpr (!FAST) {
        // columns >= hid of W2p beyond (hid+7)/8 chunks must be zero too
        const int nt_used = (p.hid + 7) >> 3;
        for (int s = threadIdx.x; s < OB * (T2 - nt_used) * 64; s += blockDim.x) {
            const int l = s & 63;
            const int rest = s >> 6;
            const int t = nt_used + rest % (T2 - nt_used);
            const int mb = rest / (T2 - nt_used);
            *reinterpret_cast<f32x4*>(&W2p[(((size_t)mb * T2 + t) * 64 + l) * 4]) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    stage_vec(b1l, p.b1, p.hid, DPH, 0.f);
    stage_vec(b2l, p.b2, p.dout, OP, 0.f);
    stage_vec(gml, p.ln_w, p.dout, OP, 1.f);
    stage_vec(btl, p.ln_b, p.dout, OP, 0.f);
    __syncthreads();
    NLAM_T_MARK(0)

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 31, hi = lane >> 5;
    float* stg = stg_all + (size_t)wave * 32 * kStgStride;
    const bool has_ln = p.ln_w != nullptr;
    const float inv_dout = 1.f / (float)p.dout;
    const bool hid_vec = FAST || (p.hid & 3) == 0, out_vec = FAST || (p.dout & 3) == 0;
#ifdef NLAM_TIMING
    int t_ntiles_ = 0;
#endif

    const long total_tiles = (long)p.ntiles * p.batch;
    const int nwaves = blockDim.x >> 6;   // 1..kFwdWaves, chosen per launch so that small problems still cover the chip
    // tile -> (workgroup, wave) with the workgroup index fastest: small launches put one tile on each CU while all
    // 8 waves of every workgroup still share the weight staging
    for (long gt = (long)wave * gridDim.x + blockIdx.x; gt < total_tiles; gt += (long)gridDim.x * nwaves) {
#ifdef NLAM_TIMING
        ++t_ntiles_;
#endif
        const int b = (int)(gt / p.ntiles);
        const TileInfo tl = get_tile(p.tiles, (int)(gt % p.ntiles), p.rows);
        const bool valid = j < tl.nrows;
        const int prow = tl.row0 + j;
        // FAST: lanes past the tile's end read a real (clamped) row; their results are never stored
        const int prow_c = FAST ? min(tl.row0 + max(min(j, tl.nrows - 1), 0), p.rows - 1) : prow;
        const bool ldv = FAST ? true : valid;

        // ---- every index the tile needs, requested together ----
        const float* srow[NLAM_MAX_SRC] = {nullptr, nullptr, nullptr};
        int swidth[NLAM_MAX_SRC] = {0, 0, 0};
#pragma unroll
        for (int s = 0; s < NLAM_MAX_SRC; ++s) {
            if (s < p.nsrc) {
                const nlam_src_t S = p.src[s];
                long ridx = prow_c;
                if (ldv && S.idx != nullptr) ridx = S.idx[prow_c];
                srow[s] = S.ptr + (long)b * S.bstride + (ldv ? ridx : 0) * (long)S.width;
                swidth[s] = S.width;
            }
        }
        int oidx = prow;
        if (p.out != nullptr && valid && p.out_idx != nullptr) oidx = p.out_idx[prow];
        int raw_ptr = 0;                                       // lane l: rowptr[seg0 + l]
        float my_scale = 1.f;
        if (p.aggr != nullptr) {
            if (!tl.split && lane <= tl.nseg) raw_ptr = p.rowptr[tl.seg0 + lane];
            if ((p.flags & NLAM_F_MEAN) && lane < tl.nseg) my_scale = p.inv_deg[tl.seg0 + lane];
        }
        auto ldc = [&](const float* row, int w, int t) -> f32x4 {
            if constexpr (FAST)
                return *reinterpret_cast<const f32x4*>(row + 8 * t + 4 * hi);
            else
                return load_chunk(row, w, t, hi, valid);
        };
        NLAM_T_MARK(1)

        // ---- GEMM1: units of half a source (4 chunks = 16 MFMAs per block) through two register buffers ----
        f32x16 acc1[HB];
#pragma unroll
        for (int hb = 0; hb < HB; ++hb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[hb][r] = 0.f;
        if constexpr (NS > 0) {
            // units of 32 columns = two K = 16 steps; lane (j, hi) reads 8 consecutive floats per step
            f32x4 xbuf[2][4];
            auto load_unit = [&](const float* row, int c32, f32x4(&x)[4]) {
                const float* r = row + 32 * c32 + 8 * hi;
                x[0] = *reinterpret_cast<const f32x4*>(r);
                x[1] = *reinterpret_cast<const f32x4*>(r + 4);
                x[2] = *reinterpret_cast<const f32x4*>(r + 16);
                x[3] = *reinterpret_cast<const f32x4*>(r + 20);
            };
            // flat list of units: (source, 32-column block)
            int us = 0, uc = 0;           // next unit to load
            auto advance = [&]() {
                ++uc;
                if (us < p.nsrc && 32 * uc >= (us == 0 ? swidth[0] : (us == 1 ? swidth[1] : swidth[2]))) {
                    ++us;
                    uc = 0;
                }
            };
            auto load_next = [&](f32x4(&x)[4]) {
                const float* base = us == 0 ? srow[0] : (us == 1 ? srow[1] : srow[2]);
                load_unit(base, uc, x);
                advance();
            };
            auto consume = [&](const f32x4(&x)[4], int u) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float xs8[8] = {x[2 * h][0], x[2 * h][1], x[2 * h][2], x[2 * h][3],
                                          x[2 * h + 1][0], x[2 * h + 1][1], x[2 * h + 1][2], x[2 * h + 1][3]};
                    const BfFrag<NSW> B = split8<NSW>(xs8);
                    mma_split_lds<NSW, HB>(acc1, W1s, HB, S1, 2 * u + h, lane, B);
                }
            };
            int nunits = 0;
            for (int s = 0; s < p.nsrc; ++s) nunits += swidth[s] >> 5;
            load_next(xbuf[0]);
            if (nunits > 1) load_next(xbuf[1]);
            for (int u = 0; u < nunits; u += 2) {   // static buffer indices: the pair (u, u + 1)
                consume(xbuf[0], u);
                if (u + 2 < nunits) load_next(xbuf[0]);
                if (u + 1 < nunits) {
                    consume(xbuf[1], u + 1);
                    if (u + 3 < nunits) load_next(xbuf[1]);
                }
            }
        } else {
            f32x4 xbuf[2][4];
            auto load_unit = [&](int u, f32x4(&x)[4]) {
                const int s = u >> 1, h = u & 1;
                const int nt = (swidth[s] + 7) >> 3;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (FAST && 4 * h + q >= nt) continue;   // lane-uniform: widths are multiples of 8
                    x[q] = ldc(srow[s], swidth[s], 4 * h + q);
                }
            };
            load_unit(0, xbuf[0]);
            load_unit(1, xbuf[1]);
            int tg = 0;
#pragma unroll
            for (int u = 0; u < 2 * NLAM_MAX_SRC; ++u) {
                const int s = u >> 1, h = u & 1;
                if (s < p.nsrc) {
                    const int nt = (swidth[s] + 7) >> 3;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (4 * h + q < nt) mma_chunk<HB>(acc1, W1p, T1, tg + 4 * h + q, xbuf[u & 1][q], lane);
                    if (h == 1) tg += nt;
                    if (((u + 2) >> 1) < p.nsrc) load_unit(u + 2, xbuf[u & 1]);
                }
            }
        }
        NLAM_T_MARK(2)

        // ---- bias, save pre-activation, SiLU -> B operand of GEMM2 ----
        {
            float* z1row = (p.z1 != nullptr && !hid_vec) ? p.z1 + ((size_t)b * p.rows + prow) * p.hid : nullptr;
            float* zbase = p.z1 != nullptr ? p.z1 + ((size_t)b * p.rows + tl.row0) * p.hid : nullptr;
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int t = hb * 4 + tt;
                    const f32x4 bias = *reinterpret_cast<const f32x4*>(&b1l[8 * t + 4 * hi]);
                    f32x4 z = acc_chunk(acc1[hb], tt) + bias;
                    if (p.z1 != nullptr) {
                        if (hid_vec)
                            *reinterpret_cast<f32x4*>(&stg[j * kStgStride + 8 * tt + 4 * hi]) = z;
                        else
                            store_chunk(z1row, p.hid, t, hi, valid, z);
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc1[hb][4 * tt + c] = (p.flags & NLAM_F_NO_ACT) ? z[c] : silu_f(z[c]);
                }
                const int wb = FAST ? 32 : min(32, p.hid - 32 * hb);
                if (p.z1 != nullptr && hid_vec && wb > 0) {
                    wave_lds_sync();
                    block_rows_out(stg, tl.nrows, wb, lane, [&](int r) { return zbase + (size_t)r * p.hid + 32 * hb; });
                    wave_lds_sync();
                }
            }
        }
        NLAM_T_MARK(3)

        // ---- GEMM2 straight from the accumulators ----
        f32x16 acc2[OB];
#pragma unroll
        for (int ob = 0; ob < OB; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[ob][r] = 0.f;
        if constexpr (NS > 0) {
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float xs8[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) xs8[q] = acc1[hb][8 * h + q];
                    const BfFrag<NSW> B = split8<NSW>(xs8);
                    mma_split_lds<NSW, OB>(acc2, W2s, OB, S2, 2 * hb + h, lane, B);
                }
        } else {
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) mma_chunk<OB>(acc2, W2p, T2, hb * 4 + tt, acc_chunk(acc1[hb], tt), lane);
        }
        NLAM_T_MARK(4)

        // ---- bias 2 + LayerNorm over the real dout features ----
        float sum = 0.f;
#pragma unroll
        for (int ob = 0; ob < OB; ++ob)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int c0 = 8 * (ob * 4 + tt) + 4 * hi;
                const f32x4 bias = *reinterpret_cast<const f32x4*>(&b2l[c0]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float v = acc2[ob][4 * tt + c] + bias[c];
                    acc2[ob][4 * tt + c] = v;
                    sum += (FAST || c0 + c < p.dout) ? v : 0.f;
                }
            }
        if (has_ln) {
            const float mean = row_allreduce(sum) * inv_dout;
            float sq = 0.f;
#pragma unroll
            for (int ob = 0; ob < OB; ++ob)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int c0 = 8 * (ob * 4 + tt) + 4 * hi;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float dlt = acc2[ob][4 * tt + c] - mean;
                        acc2[ob][4 * tt + c] = dlt;
                        sq += (FAST || c0 + c < p.dout) ? dlt * dlt : 0.f;
                    }
                }
            const float rstd = rsqrtf(row_allreduce(sq) * inv_dout + p.eps);
            if (p.rstd != nullptr && valid && hi == 0) p.rstd[(size_t)b * p.rows + prow] = rstd;
            float* xrow = (p.xhat != nullptr && !out_vec) ? p.xhat + ((size_t)b * p.rows + prow) * p.dout : nullptr;
            float* xbase = p.xhat != nullptr ? p.xhat + ((size_t)b * p.rows + tl.row0) * p.dout : nullptr;
#pragma unroll
            for (int ob = 0; ob < OB; ++ob) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int t = ob * 4 + tt;
                    const int c0 = 8 * t + 4 * hi;
                    f32x4 xh;
#pragma unroll
                    for (int c = 0; c < 4; ++c) xh[c] = acc2[ob][4 * tt + c] * rstd;
                    if (p.xhat != nullptr) {
                        if (out_vec)
                            *reinterpret_cast<f32x4*>(&stg[j * kStgStride + 8 * tt + 4 * hi]) = xh;
                        else
                            store_chunk(xrow, p.dout, t, hi, valid, xh);
                    }
                    const f32x4 g = *reinterpret_cast<const f32x4*>(&gml[c0]);
                    const f32x4 be = *reinterpret_cast<const f32x4*>(&btl[c0]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc2[ob][4 * tt + c] = xh[c] * g[c] + be[c];
                }
                const int wb = FAST ? 32 : min(32, p.dout - 32 * ob);
                if (p.xhat != nullptr && out_vec && wb > 0) {
                    wave_lds_sync();
                    block_rows_out(stg, tl.nrows, wb, lane, [&](int r) { return xbase + (size_t)r * p.dout + 32 * ob; });
                    wave_lds_sync();
                }
            }
        }
        NLAM_T_MARK(5)

        // ---- msg = mlp [+ src1]; aggregate; out = msg [+ src0] ----
        {
            float* orow = (p.out != nullptr && !out_vec) ? p.out + (long)b * p.out_bstride + (valid ? (long)oidx : 0L) * (long)p.dout : nullptr;
            float* obase = p.out != nullptr ? p.out + (long)b * p.out_bstride : nullptr;
            float* abase = p.aggr != nullptr ? p.aggr + (size_t)b * p.nseg_total * p.dout : nullptr;
#pragma unroll
            for (int ob = 0; ob < OB; ++ob) {
                const int wb = FAST ? 32 : min(32, p.dout - 32 * ob);
                f32x4 m[4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    m[tt] = acc_chunk(acc2[ob], tt);
                    if (p.flags & NLAM_F_ADD_SRC1) m[tt] += ldc(srow[1], p.dout, ob * 4 + tt);
                    if (!valid) m[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                if (p.aggr != nullptr && wb > 0) {
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) *reinterpret_cast<f32x4*>(&stg[j * kStgStride + 8 * tt + 4 * hi]) = m[tt];
                    wave_lds_sync();
                    if (out_vec) {
                        block_segment_reduce(stg, tl, raw_ptr, my_scale, abase + 32 * ob, p.dout, wb, lane);
                    } else {
                        if constexpr (!FAST)
                            tile_segment_reduce(stg, kStgStride, tl, p.rowptr, (p.flags & NLAM_F_MEAN) ? p.inv_deg : nullptr,
                                                abase + 32 * ob, p.dout, wb, lane);
                    }
                    wave_lds_sync();
                }
                if (p.out != nullptr && wb > 0) {
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        if (p.flags & NLAM_F_ADD_SRC0) m[tt] += ldc(srow[0], p.dout, ob * 4 + tt);
                        if (out_vec)
                            *reinterpret_cast<f32x4*>(&stg[j * kStgStride + 8 * tt + 4 * hi]) = m[tt];
                        else
                            store_chunk(orow, p.dout, ob * 4 + tt, hi, valid, m[tt]);
                    }
                    if (out_vec) {
                        wave_lds_sync();
                        block_rows_out(stg, tl.nrows, wb, lane,
                                       [&](int r) { return obase + (long)__shfl(oidx, r, 64) * p.dout + 32 * ob; });
                        wave_lds_sync();
                    }
                }
            }
        }
        NLAM_T_MARK(6)
    }
    NLAM_T_DRAIN
    NLAM_T_MARK(7)
    NLAM_T_FLUSH(t_ntiles_)
}

// ---------------------------------------------------------------------------
// forward, split-bf16 matrix path, FAST shapes, software-pipelined across tiles.
// One wave = one tile at a time (two waves per SIMD).  With the GEMMs on the bf16 cores a
// tile holds only ~6k cycles of MFMA and ~8k of VALU, so memory latency decides: every load
// of tile n+1 is issued while tile n still computes, and -- because gfx950 retires loads and
// stores through ONE in-order counter -- always AHEAD of tile n's stores:
//   top of tile n   : rows of tile n are already in registers (96 VGPRs: 3 sources x 64 columns);
//                     request desc(n+2)?  no: desc(n+1) arrived during tile n-1; request idx(n+1),
//                     the epilogue indices of tile n (rowptr, out rows) and desc(n+2)
//   GEMM1(n)        : consumes the row registers
//   right after     : rows of tile n+1 -> the same registers (idx(n+1) has landed)
//   SiLU, GEMM2, LayerNorm, then ALL stores of tile n (z1, xhat, aggr / out)
// ---------------------------------------------------------------------------
// RAG = some source is narrower than / not a multiple of a 32-column unit (element-wise, zero-filled loads)
// RES = the launch adds a residual row (NLAM_F_ADD_SRC0 with an output, or NLAM_F_ADD_SRC1): only then are the 32
// VGPRs of the stashed residual chunks allocated (the kernel sits at the 256-VGPR limit; spills are VMEM ops and
// would queue behind the tile's stores)
// PRE (NLAM_F_PRE_ADD, the factorised edge MLP): only source 0 goes through GEMM1; sources 1.. are gathered
// pre-activation addends of width hid, fetched in accumulator ("chunk") layout a tile ahead and added to GEMM1's result.
// The kernel body takes its workgroup index and grid size as arguments: the plain launch passes blockIdx.x / gridDim.x,
// the GROUPED launch (nlam_mlp_fwd_group: several independent same-shape MLPs in one grid, each workgroup bound to one
// member -- the embedders of the static graph features) passes the member-local ones.
// CAT (with RAG): the single source is a concatenation of pieces (p.ncat > 0); its own instantiation, because the piece
// descriptors are ~35 more scalar registers the other ragged launches (embedders, output_map) should not carry
template <int HB, int OB, int NS, bool RAG, bool RES, bool PRE = false, bool CAT = false>
__device__ __forceinline__ void mlp_fwd_bf_body(const nlam_mlp_fwd_t& p, const int wg_id, const int wg_count) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int DPH = HB * 32, OP = OB * 32;
    constexpr int S2 = DPH / 16;
    NLAM_T_DECL
#ifdef NLAM_TIMING
    int t_ntiles_ = 0;
#endif
    constexpr int MAXU = (RAG || PRE) ? 2 : 2 * NLAM_MAX_SRC;  // 32-column units per tile row (widths <= 64; RAG / PRE: a single GEMM source)

    // every source is padded to whole 32-column units (zero weights / zero-filled loads past its width), so
    // narrow or odd inputs (the 2- / 3-feature embedder inputs, kin = 56) run on the same path
    const int ngemm = PRE ? 1 : p.nsrc;   // sources that feed GEMM1
    int kin = 0, nunits = 0;
    for (int s = 0; s < ngemm; ++s) {
        kin += p.src[s].width;
        nunits += (p.src[s].width + 31) >> 5;
    }
    const int ldw1 = (PRE && p.ldw1 > 0) ? p.ldw1 : kin;   // floats between rows of W1
    const int S1 = 2 * nunits;

    u32x4* W1s = reinterpret_cast<u32x4*>(smem);                       // [NS][HB][S1][64] x 16 B
    u32x4* W2s = W1s + (size_t)NS * HB * S1 * 64;                      // [NS][OB][S2][64] x 16 B
    float* b1l = reinterpret_cast<float*>(W2s + (size_t)NS * OB * S2 * 64);   // DPH
    float* b2l = b1l + DPH;                                            // OP
    float* gml = b2l + OP;                                             // OP
    float* btl = gml + OP;                                             // OP
    float* stg_all = btl + OP;                                         // kFwdWaves x 32 x kStgStride
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 31, hi = lane >> 5;
    float* stg = stg_all + (size_t)wave * 32 * kStgStride;
    const bool has_ln = p.ln_w != nullptr;
    const float inv_dout = 1.f / (float)p.dout;

    // unit u -> (source, first column) ; fixed per launch
    int usrc[MAXU], ucol[MAXU];
    {
        int u = 0;
#pragma unroll
        for (int k2 = 0; k2 < MAXU; ++k2) {
            usrc[k2] = 0;
            ucol[k2] = 0;
        }
        for (int s = 0; s < ngemm; ++s)
            for (int c = 0; c < ((p.src[s].width + 31) >> 5); ++c) {
#pragma unroll
                for (int k = 0; k < MAXU; ++k)
                    if (k == u) {
                        usrc[k] = s;
                        ucol[k] = 32 * c;
                    }
                ++u;
            }
    }

    const int nwaves = blockDim.x >> 6;
    const long total_tiles = (long)p.ntiles * p.batch;
    const long stride = (long)wg_count * nwaves;
    long gt = (long)wave * wg_count + wg_id;   // workgroup index fastest (see mlp_fwd_kernel)

    auto tile_of = [&](long g, int& b_out) -> TileInfo {
        b_out = (int)(g / p.ntiles);
        return get_tile(p.tiles, (int)(g % p.ntiles), p.rows);
    };
    auto clamp_row = [&](const TileInfo& t) { return min(t.row0 + max(min(j, t.nrows - 1), 0), p.rows - 1); };
    // gathered rows of a tile -> per-source row pointers
    auto row_ptrs = [&](const TileInfo& t, int b, const int (&ridx)[NLAM_MAX_SRC], const float* (&rp)[NLAM_MAX_SRC]) {
#pragma unroll
        for (int s = 0; s < NLAM_MAX_SRC; ++s)
            rp[s] = s < p.nsrc ? p.src[s].ptr + (long)b * p.src[s].bstride + (long)ridx[s] * p.src[s].width : nullptr;
        if constexpr (CAT)   // concatenated pieces: no single row pointer -- pack (batch, row) for load_unit
            rp[0] = reinterpret_cast<const float*>(((uintptr_t)(unsigned)b << 32) | (uintptr_t)(unsigned)ridx[0]);
    };
    auto load_idx = [&](const TileInfo& t, int (&ridx)[NLAM_MAX_SRC]) {
        const int pr = clamp_row(t);
#pragma unroll
        for (int s = 0; s < NLAM_MAX_SRC; ++s) {
            ridx[s] = pr;
            if (s < p.nsrc && p.src[s].idx != nullptr) ridx[s] = p.src[s].idx[pr];
        }
    };
    constexpr int kPre = (RAG || PRE) ? 2 : (RES ? NLAM_RES_PRE : 4);   // units prefetched a tile ahead (the rest stream in at the top of their own tile); one fewer
                                                    // when 32 registers hold the residual rows: four spilled inside the tile loop
    constexpr int kTop = MAXU - kPre > 0 ? MAXU - kPre : 1;
    f32x4 xp[kPre][4];        // loop-carried: rows of the next tile
    // RAG with p.ncat > 0: the single source is the row-wise concatenation of up to NLAM_MAX_CAT pieces (the torch.cat of the
    // grid input features, graph/base.py:275-283, folded into this load): rp[0] then carries (row index, batch) instead of
    // a row pointer -- see row_ptrs -- and every element picks its piece by column
    auto load_unit = [&](const float* const (&rp)[NLAM_MAX_SRC], int u, f32x4(&xu)[4]) {
        if constexpr (CAT) {
            // at most kCatPieces pieces (the launcher checks); an unused trailing piece has width 0, i.e. starts at the total width
            // and is never selected for a column below it
            constexpr int kCatPieces = 4;
            const long crow = (long)(reinterpret_cast<uintptr_t>(rp[0]) & 0xffffffffu);
            const long cb = (long)(reinterpret_cast<uintptr_t>(rp[0]) >> 32);
            const int w0 = p.cat_width[0], w1 = p.cat_width[1], w2 = p.cat_width[2], w3 = p.cat_width[3];
            const int s1 = w0, s2 = w0 + w1, s3 = w0 + w1 + w2;
            // element pointers: q_k[col] is column `col` of the concatenated row
            const float* q0 = p.cat_ptr[0] + cb * p.cat_bstride[0] + crow * w0;
            const float* q1 = p.cat_ptr[1] + cb * p.cat_bstride[1] + crow * w1 - s1;
            const float* q2 = p.cat_ptr[2] + cb * p.cat_bstride[2] + crow * w2 - s2;
            const float* q3 = p.cat_ptr[3] + cb * p.cat_bstride[3] + crow * w3 - s3;
            const int w = p.src[0].width, c0 = ucol[u] + 8 * hi;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int cq = c0 + (q & 1) * 4 + (q >> 1) * 16;
                // opaque to the optimiser: the column-range compares below are tile-invariant, and hoisted out of the tile loop they
                // become ~340 scalar masks spilled to VGPR lanes (v_readlane pairs at every use, six VGPRs, and with them two
                // VGPR spills = a scratch segment, which slows every dispatch of the kernel); recomputed they are one v_cmp each
                asm volatile("" : "+v"(cq));
                xu[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (ucol[u] + (q & 1) * 4 + (q >> 1) * 16 >= w) continue;   // wave-uniform: the whole quad lies past the row's end for both lane halves
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int col = min(cq + c, w - 1);   // clamped: the load is unconditional (a predicated load is a branch + a full wait per element)
                    const float* src = col >= s3 ? q3 : (col >= s2 ? q2 : (col >= s1 ? q1 : q0));
                    const float v = src[col];
                    xu[q][c] = cq + c < w ? v : 0.f;
                }
            }
            (void)kCatPieces;
            return;
        }
        const float* row = usrc[u] == 0 ? rp[0] : (usrc[u] == 1 ? rp[1] : rp[2]);
        const int w = usrc[u] == 0 ? p.src[0].width : (usrc[u] == 1 ? p.src[1].width : p.src[2].width);
        const int c0 = ucol[u] + 8 * hi;
        if (!RAG || (w & 31) == 0) {   // whole units, 16-B aligned rows
            xu[0] = *reinterpret_cast<const f32x4*>(row + c0);
            xu[1] = *reinterpret_cast<const f32x4*>(row + c0 + 4);
            xu[2] = *reinterpret_cast<const f32x4*>(row + c0 + 16);
            xu[3] = *reinterpret_cast<const f32x4*>(row + c0 + 20);
        } else {               // ragged source: element-wise, zero past its width (clamped column: unconditional loads)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int cq = c0 + (q & 1) * 4 + (q >> 1) * 16;
                asm volatile("" : "+v"(cq));   // see the concatenated-pieces path: keeps the column compares inside the tile loop
                xu[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (ucol[u] + (q & 1) * 4 + (q >> 1) * 16 >= w) continue;   // wave-uniform: a quad past the row's end for both lane halves costs no loads
                                                  // (the 2- / 3-column embedder inputs: 4 loads per lane instead of 16)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float v = row[min(cq + c, w - 1)];
                    xu[q][c] = cq + c < w ? v : 0.f;
                }
            }
        }
    };
    f32x4 pa[PRE ? 2 : 1][PRE ? HB * 4 : 1];   // loop-carried: the next tile's addend rows, chunk layout (PRE)
    auto load_pre = [&](const float* const (&rp)[NLAM_MAX_SRC]) {
#pragma unroll
        for (int u = 0; u < kPre; ++u)
            if (u < nunits) load_unit(rp, u, xp[u]);
        if constexpr (PRE) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int t = 0; t < HB * 4; ++t)
                    pa[k][t] = 1 + k < p.nsrc ? *reinterpret_cast<const f32x4*>(rp[1 + k] + 8 * t + 4 * hi) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // ---- pipeline prologue: tile 0 fully resolved and loaded, tile 1's descriptor requested ----
    TileInfo tl = {0, 0, 0, 0, false}, tln = {0, 0, 0, 0, false};
    int b = 0, bn = 0;
    int ridx[NLAM_MAX_SRC] = {0, 0, 0}, ridx_n[NLAM_MAX_SRC] = {0, 0, 0};
    const float* srow[NLAM_MAX_SRC] = {nullptr, nullptr, nullptr};
    if (gt < total_tiles) {
        tl = tile_of(gt, b);
        load_idx(tl, ridx);
        row_ptrs(tl, b, ridx, srow);
        load_pre(srow);
        if (gt + stride < total_tiles) tln = tile_of(gt + stride, bn);
    }

    // ---- weights -> LDS while the first tile's descriptor / index / row loads are in flight ----
    if (p.wpack != nullptr) {
        // pre-packed image (nlam_mlp_pack, once per optimizer step): [W1s | W2s] in exactly this LDS layout
        lds_dma_copy(smem, p.wpack, ((size_t)NS * HB * S1 * 64 + (size_t)NS * OB * S2 * 64) * 16);
    } else {
        int s0 = 0, off = 0;
        for (int s = 0; s < ngemm; ++s) {
            const int w = p.src[s].width;
            stage_split<NS>(W1s, S1, s0, p.W1 + off, ldw1, p.hid, HB, w, false, 1, ((w + 31) >> 5) * 32);
            off += w;
            s0 += 2 * ((w + 31) >> 5);
        }
        stage_split<NS>(W2s, S2, 0, p.W2, p.hid, p.dout, OB, p.hid, true);
    }
    stage_vec(b1l, p.b1, p.hid, DPH, 0.f);
    stage_vec(b2l, p.b2, p.dout, OP, 0.f);
    stage_vec(gml, p.ln_w, p.dout, OP, 1.f);
    stage_vec(btl, p.ln_b, p.dout, OP, 0.f);
    __syncthreads();
    NLAM_T_MARK(0)

    for (; gt < total_tiles; gt += stride) {
#ifdef NLAM_TIMING
        ++t_ntiles_;
#endif
        const bool valid = j < tl.nrows;
        const int prow = tl.row0 + j;
        const bool has_next = gt + stride < total_tiles;

        // ---- the last units of this tile (behind the previous tile's stores, but needed last in GEMM1) ----
        f32x4 xt[kTop][4];
#pragma unroll
        for (int u = kPre; u < MAXU; ++u) {
#pragma unroll
            for (int q = 0; q < 4; ++q) xt[u - kPre][q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (u < nunits) load_unit(srow, u, xt[u - kPre]);
        }
        // ---- requests that must be ahead of this tile's stores: idx(n+1), epilogue indices of tile n, desc(n+2) ----
        if (has_next) load_idx(tln, ridx_n);
        int oidx = ridx[0];
        if (p.out != nullptr && p.out_idx != p.src[0].idx) oidx = p.out_idx != nullptr ? p.out_idx[clamp_row(tl)] : clamp_row(tl);
        int raw_ptr = 0;
        float my_scale = 1.f;
        if (p.aggr != nullptr) {
            if (!tl.split && lane <= tl.nseg) raw_ptr = p.rowptr[tl.seg0 + lane];
            if ((p.flags & NLAM_F_MEAN) && lane < tl.nseg) my_scale = p.inv_deg[tl.seg0 + lane];
        }
        TileInfo tlnn = {0, 0, 0, 0, false};
        int bnn = 0;
        if (gt + 2 * stride < total_tiles) tlnn = tile_of(gt + 2 * stride, bnn);

        NLAM_T_MARK(1)
        // ---- GEMM1 over the resident rows ----
        f32x16 acc1[HB];
#pragma unroll
        for (int hb = 0; hb < HB; ++hb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[hb][r] = 0.f;
        if constexpr (PRE) {
            // (W1_j x)[sender] + (W1_i x)[receiver]: the addend rows prefetched a tile ago ARE the initial accumulators
            // (chunk layout = accumulator layout), so they cost no registers of their own during GEMM1
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc1[hb][4 * tt + c] = pa[0][hb * 4 + tt][c] + pa[1][hb * 4 + tt][c];
        }
        f32x4 resid[RES ? 2 : 1][4];   // the 64 columns of source 0 (edge / node residual) or source 1 (PropagationNet)
#pragma unroll
        for (int u = 0; u < MAXU; ++u) {
            if (u < nunits) {
                const f32x4(&x)[4] = u < kPre ? xp[u < kPre ? u : 0] : xt[u >= kPre ? (u - kPre < kTop ? u - kPre : 0) : 0];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float xs8[8] = {x[2 * h][0], x[2 * h][1], x[2 * h][2], x[2 * h][3],
                                          x[2 * h + 1][0], x[2 * h + 1][1], x[2 * h + 1][2], x[2 * h + 1][3]};
                    const BfFrag<NS> B = split8<NS>(xs8);
                    mma_split_lds<NS, HB>(acc1, W1s, HB, S1, 2 * u + h, lane, B);
                }
            }
        }
        NLAM_T_MARK(2)
        if (CAT && p.cat_out != nullptr) {
            // the concatenated input rows, for the weight gradient / backward: unit by unit through the staging block, whole
            // 16-byte pieces per lane (the row width is a multiple of 4: checked by the launcher)
            float* cbase = p.cat_out + ((size_t)b * p.rows + tl.row0) * p.src[0].width;
#pragma unroll
            for (int u = 0; u < MAXU; ++u) {
                if (u < nunits) {
                    const f32x4(&x)[4] = xp[u < kPre ? u : 0];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<f32x4*>(&stg[j * kStgStride + 8 * hi + (q & 1) * 4 + (q >> 1) * 16]) = x[q];
                    wave_lds_sync();
                    const int wb = min(32, p.src[0].width - 32 * u);
                    block_rows_out(stg, tl.nrows, wb, lane, [&](int r) { return cbase + (size_t)r * p.src[0].width + 32 * u; });
                    wave_lds_sync();
                }
            }
        }
        // residual rows (C-layout chunks 8t + 4hi differ from the unit layout 8hi + ..: re-read below, ahead of the stores)
        const bool add0 = RES && (p.flags & NLAM_F_ADD_SRC0) != 0 && p.out != nullptr;
        const bool add1 = RES && (p.flags & NLAM_F_ADD_SRC1) != 0;
        if (RES && (add0 || add1)) {
            const float* rr = add1 ? srow[1] : srow[0];
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
                    if (ob < OB) resid[RES ? ob : 0][tt] = *reinterpret_cast<const f32x4*>(rr + 8 * (ob * 4 + tt) + 4 * hi);
        }
        // ---- rows of tile n+1 into the same registers (still ahead of every store of tile n) ----
        const float* srow_n[NLAM_MAX_SRC] = {nullptr, nullptr, nullptr};
        if (has_next) {
            row_ptrs(tln, bn, ridx_n, srow_n);
            load_pre(srow_n);
        }

        // ---- bias, save pre-activation, SiLU -> B operand of GEMM2 ----
        {
            float* zbase = p.z1 != nullptr ? p.z1 + ((size_t)b * p.rows + tl.row0) * p.hid : nullptr;
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const f32x4 bias = *reinterpret_cast<const f32x4*>(&b1l[8 * (hb * 4 + tt) + 4 * hi]);
                    const f32x4 z = acc_chunk(acc1[hb], tt) + bias;
                    if (p.z1 != nullptr) *reinterpret_cast<f32x4*>(&stg[j * kStgStride + 8 * tt + 4 * hi]) = z;
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc1[hb][4 * tt + c] = silu_f(z[c]);
                }
                if (p.z1 != nullptr) {
                    wave_lds_sync();
                    block_rows_out(stg, tl.nrows, 32, lane, [&](int r) { return zbase + (size_t)r * p.hid + 32 * hb; });
                    wave_lds_sync();
                }
            }
        }

        NLAM_T_MARK(3)
        // ---- GEMM2 straight from the accumulators ----
        f32x16 acc2[OB];
#pragma unroll
        for (int ob = 0; ob < OB; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[ob][r] = 0.f;
#pragma unroll
        for (int hb = 0; hb < HB; ++hb)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float xs8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) xs8[q] = acc1[hb][8 * h + q];
                const BfFrag<NS> B = split8<NS>(xs8);
                mma_split_lds<NS, OB>(acc2, W2s, OB, S2, 2 * hb + h, lane, B);
            }

        NLAM_T_MARK(4)
        // ---- bias 2 + LayerNorm ----
        float sum = 0.f;
#pragma unroll
        for (int ob = 0; ob < OB; ++ob)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const f32x4 bias = *reinterpret_cast<const f32x4*>(&b2l[8 * (ob * 4 + tt) + 4 * hi]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float v = acc2[ob][4 * tt + c] + bias[c];
                    acc2[ob][4 * tt + c] = v;
                    sum += v;
                }
            }
        if (has_ln) {
            const float mean = row_allreduce(sum) * inv_dout;
            float sq = 0.f;
#pragma unroll
            for (int ob = 0; ob < OB; ++ob)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float dlt = acc2[ob][r] - mean;
                    acc2[ob][r] = dlt;
                    sq += dlt * dlt;
                }
            const float rstd = rsqrtf(row_allreduce(sq) * inv_dout + p.eps);
            if (p.rstd != nullptr && valid && hi == 0) p.rstd[(size_t)b * p.rows + prow] = rstd;
            float* xbase = p.xhat != nullptr ? p.xhat + ((size_t)b * p.rows + tl.row0) * p.dout : nullptr;
#pragma unroll
            for (int ob = 0; ob < OB; ++ob) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int c0 = 8 * (ob * 4 + tt) + 4 * hi;
                    f32x4 xh;
#pragma unroll
                    for (int c = 0; c < 4; ++c) xh[c] = acc2[ob][4 * tt + c] * rstd;
                    if (p.xhat != nullptr) *reinterpret_cast<f32x4*>(&stg[j * kStgStride + 8 * tt + 4 * hi]) = xh;
                    const f32x4 g = *reinterpret_cast<const f32x4*>(&gml[c0]);
                    const f32x4 be = *reinterpret_cast<const f32x4*>(&btl[c0]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc2[ob][4 * tt + c] = xh[c] * g[c] + be[c];
                }
                if (p.xhat != nullptr) {
                    wave_lds_sync();
                    block_rows_out(stg, tl.nrows, 32, lane, [&](int r) { return xbase + (size_t)r * p.dout + 32 * ob; });
                    wave_lds_sync();
                }
            }
        }

        NLAM_T_MARK(5)
        // ---- msg = mlp [+ src1]; aggregate; out = msg [+ src0] ----
        {
            float* obase = p.out != nullptr ? p.out + (long)b * p.out_bstride : nullptr;
            float* abase = p.aggr != nullptr ? p.aggr + (size_t)b * p.nseg_total * p.dout : nullptr;
#pragma unroll
            for (int ob = 0; ob < OB; ++ob) {
                f32x4 m[4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    m[tt] = acc_chunk(acc2[ob], tt);
                    if (add1) m[tt] += resid[RES ? ob : 0][tt];
                    if (!valid) m[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                if (p.aggr != nullptr) {
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) *reinterpret_cast<f32x4*>(&stg[j * kStgStride + 8 * tt + 4 * hi]) = m[tt];
                    wave_lds_sync();
                    block_segment_reduce(stg, tl, raw_ptr, my_scale, abase + 32 * ob, p.dout, 32, lane);
                    wave_lds_sync();
                }
                if (p.out != nullptr) {
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        if (add0) {
                            // both residuals at once (PropagationNet with edge update) re-reads source 0 behind the stores
                            if (add1)
                                m[tt] += *reinterpret_cast<const f32x4*>(srow[0] + 8 * (ob * 4 + tt) + 4 * hi);
                            else
                                m[tt] += resid[RES ? ob : 0][tt];
                        }
                        *reinterpret_cast<f32x4*>(&stg[j * kStgStride + 8 * tt + 4 * hi]) = m[tt];
                    }
                    wave_lds_sync();
                    if (RAG && (p.dout & 31) != 0) {
                        // output width not a whole 32-column block (output_map, graph/base.py:322: 17 state variables; no
                        // LayerNorm): rows of p.dout floats are not 16-byte aligned, so the block leaves as dwords -- lane ->
                        // (row, column) in row-major order of the tile, which is contiguous in memory for identity row order
                        const int wb = min(32, p.dout - 32 * ob);
                        const int total = tl.nrows * wb;
                        for (int base = 0; base < total; base += 64) {
                            const int item = base + lane;
                            const int r = min(item / wb, 31), c = item - r * wb;
                            float* d = obase + (long)__shfl(oidx, r, 64) * p.dout + 32 * ob;
                            if (item < total) d[c] = stg[r * kStgStride + c];
                        }
                    } else {
                        block_rows_out(stg, tl.nrows, 32, lane,
                                       [&](int r) { return obase + (long)__shfl(oidx, r, 64) * p.dout + 32 * ob; });
                    }
                    wave_lds_sync();
                }
            }
        }

        NLAM_T_MARK(6)
        // ---- rotate ----
        tl = tln;
        b = bn;
        tln = tlnn;
        bn = bnn;
#pragma unroll
        for (int s = 0; s < NLAM_MAX_SRC; ++s) {
            ridx[s] = ridx_n[s];
            srow[s] = srow_n[s];
        }
    }
    NLAM_T_DRAIN
    NLAM_T_MARK(7)
    NLAM_T_FLUSH(t_ntiles_)
}

template <int HB, int OB, int NS, bool RAG, bool RES, bool PRE = false, bool CAT = false>
__global__ __launch_bounds__(kFwdThreads) void mlp_fwd_bf_kernel(const nlam_mlp_fwd_t p) {
    mlp_fwd_bf_body<HB, OB, NS, RAG, RES, PRE, CAT>(p, xcd_slot((int)blockIdx.x, (int)gridDim.x), (int)gridDim.x);
}

struct fwd_group_t {
    nlam_mlp_fwd_t g[NLAM_MAX_GROUP];
    int n;
    int first[NLAM_MAX_GROUP + 1];   // first workgroup of each member; first[n] = grid size
};

template <int HB, int OB, int NS>
__global__ __launch_bounds__(kFwdThreads) void mlp_fwd_bf_group_kernel(const fwd_group_t G) {
    int gi = 0;
#pragma unroll
    for (int k = 1; k < NLAM_MAX_GROUP; ++k)
        if (k < G.n && (int)blockIdx.x >= G.first[k]) gi = k;
    mlp_fwd_bf_body<HB, OB, NS, true, false, false>(G.g[gi], (int)blockIdx.x - G.first[gi], G.first[gi + 1] - G.first[gi]);
}

// ---------------------------------------------------------------------------
// backward (data gradients + bias / LayerNorm-affine partial sums)
// ---------------------------------------------------------------------------
// column sums of a staged 32 x w tile, accumulated into one register per lane
// (two for w > 64 are not needed: kMaxWidth = 64)
__device__ __forceinline__ float tile_colsum(const float* stg, int S, int w, int lane) {
    float s = 0.f;
    if (lane < w) {
#pragma unroll 8
        for (int q = 0; q < 32; ++q) s += stg[q * S + lane];
    }
    return s;
}

template <int HB, int OB>
__global__ __launch_bounds__(kBlockThreads) void mlp_bwd_kernel(const nlam_mlp_bwd_t p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int DPH = HB * 32, OP = OB * 32;
    constexpr int WMAX = (DPH > OP ? DPH : OP) > 64 ? (DPH > OP ? DPH : OP) : 64;
    constexpr int STG = WMAX + 4;
    constexpr int T2 = OP / 8;    // K chunks of dh = W2^T dz2   (K = dout)
    constexpr int T1 = DPH / 8;   // K chunks of dx = W1^T dz1   (K = hid)

    int kin = 0;
    for (int s = 0; s < p.nsrc; ++s) kin += p.src[s].width;

    // W2^T packed: M = hid (HB blocks), K = dout
    float* W2t = smem;                              // DPH x OP
    float* W1t = W2t + (size_t)DPH * OP;            // per source: round32(w) x DPH, only for sources with dmode != 0
    size_t w1t_floats = 0;
    int w1t_off[NLAM_MAX_SRC];
    for (int s = 0; s < p.nsrc; ++s) {
        w1t_off[s] = (int)w1t_floats;
        if (p.dmode[s] != 0) w1t_floats += (size_t)round_up(p.src[s].width, 32) * DPH;
    }
    float* gml = W1t + w1t_floats;                  // OP
    float* stg_all = gml + OP;                      // kWavesPerBlock x 32 x STG

    // A[m][k] = W2[k][m]  ->  ldm = 1, ldk = hid
    stage_packed(W2t, T2, 0, p.W2, 1, p.hid, p.hid, HB, p.dout);
    {
        const int nt_used = (p.dout + 7) >> 3;
        if (nt_used < T2)
            for (int s = threadIdx.x; s < HB * (T2 - nt_used) * 64; s += blockDim.x) {
                const int l = s & 63;
                const int rest = s >> 6;
                const int t = nt_used + rest % (T2 - nt_used);
                const int mb = rest / (T2 - nt_used);
                *reinterpret_cast<f32x4*>(&W2t[(((size_t)mb * T2 + t) * 64 + l) * 4]) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    }
    {
        int off = 0;
        for (int s = 0; s < p.nsrc; ++s) {
            const int w = p.src[s].width;
            if (p.dmode[s] != 0) {
                // A[m][k] = W1[k][off + m]  ->  ldm = 1, ldk = kin ; M = w, K = hid
                float* dst = W1t + w1t_off[s];
                const int MBs = round_up(w, 32) / 32;
                stage_packed(dst, T1, 0, p.W1 + off, 1, kin, w, MBs, p.hid);
                const int nt_used = (p.hid + 7) >> 3;
                if (nt_used < T1)
                    for (int q = threadIdx.x; q < MBs * (T1 - nt_used) * 64; q += blockDim.x) {
                        const int l = q & 63;
                        const int rest = q >> 6;
                        const int t = nt_used + rest % (T1 - nt_used);
                        const int mb = rest / (T1 - nt_used);
                        *reinterpret_cast<f32x4*>(&dst[(((size_t)mb * T1 + t) * 64 + l) * 4]) = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
            }
            off += w;
        }
    }
    stage_vec(gml, p.ln_w, p.dout, OP, 1.f);
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 31, hi = lane >> 5;
    float* stg = stg_all + (size_t)wave * 32 * STG;
    const bool has_ln = p.ln_w != nullptr;
    const float inv_dout = 1.f / (float)p.dout;

    // per-lane column accumulators (lane = column): db1, db2, dgamma, dbeta
    float acc_db1 = 0.f, acc_db2 = 0.f, acc_dg = 0.f, acc_dbt = 0.f;

    const long total_tiles = (long)p.ntiles * p.batch;
    for (long gt = (long)wave * gridDim.x + blockIdx.x; gt < total_tiles; gt += (long)gridDim.x * kWavesPerBlock) {
        const int b = (int)(gt / p.ntiles);
        const int ti = (int)(gt % p.ntiles);
        const TileInfo tl = get_tile(p.tiles, ti, p.rows);
        const bool valid = j < tl.nrows;
        const int prow = tl.row0 + j;
        const size_t srow_id = (size_t)b * p.rows + prow;

        // ---- upstream gradient wrt msg (C layout chunks) ----
        const float* grow = nullptr;
        if (p.g_out != nullptr) {
            long oidx = prow;
            if (valid && p.out_idx != nullptr) oidx = p.out_idx[prow];
            grow = p.g_out + (long)b * p.out_bstride + (valid ? oidx : 0) * (long)p.dout;
        }
        const float* garow = nullptr;
        float gscale = 1.f;
        if (p.g_aggr != nullptr) {
            const int sg = valid ? p.seg_of_row[prow] : 0;
            garow = p.g_aggr + ((size_t)b * p.nseg_total + sg) * p.dout;
            if (p.flags & NLAM_F_MEAN) gscale = p.inv_deg[sg];
        }
        const float* xrow = has_ln ? p.xhat + srow_id * p.dout : nullptr;
        const float rstd = (has_ln && valid) ? p.rstd[srow_id] : 0.f;

        f32x16 dz2[OB];
        // pass 1: dmsg, LN-affine partials, LN reductions
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int ob = 0; ob < OB; ++ob)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int t = ob * 4 + tt;
                f32x4 g = {0.f, 0.f, 0.f, 0.f};
                if (grow != nullptr) g += load_chunk(grow, p.dout, t, hi, valid);
                if (garow != nullptr) g += load_chunk(garow, p.dout, t, hi, valid) * gscale;
#pragma unroll
                for (int c = 0; c < 4; ++c) dz2[ob][4 * tt + c] = g[c];
                // stage dmsg for the dbeta column sum
                *reinterpret_cast<f32x4*>(&stg[j * STG + 8 * t + 4 * hi]) = g;
            }
        if (has_ln) {
            wave_lds_sync();
            acc_dbt += tile_colsum(stg, STG, p.dout, lane);
            wave_lds_sync();
#pragma unroll
            for (int ob = 0; ob < OB; ++ob)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int t = ob * 4 + tt;
                    const f32x4 xh = load_chunk(xrow, p.dout, t, hi, valid);
                    const f32x4 gm = *reinterpret_cast<const f32x4*>(&gml[8 * t + 4 * hi]);
                    f32x4 gx;  // dmsg * xhat -> dgamma
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float g = dz2[ob][4 * tt + c];
                        gx[c] = g * xh[c];
                        const float gy = g * gm[c];
                        dz2[ob][4 * tt + c] = gy;
                        m1 += gy;          // padded columns: g == 0
                        m2 += gy * xh[c];
                    }
                    *reinterpret_cast<f32x4*>(&stg[j * STG + 8 * t + 4 * hi]) = gx;
                }
            wave_lds_sync();
            acc_dg += tile_colsum(stg, STG, p.dout, lane);
            wave_lds_sync();
            m1 = row_allreduce(m1) * inv_dout;
            m2 = row_allreduce(m2) * inv_dout;
#pragma unroll
            for (int ob = 0; ob < OB; ++ob)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int t = ob * 4 + tt;
                    const int c0 = 8 * t + 4 * hi;
                    const f32x4 xh = load_chunk(xrow, p.dout, t, hi, valid);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float v = rstd * (dz2[ob][4 * tt + c] - m1 - xh[c] * m2);
                        dz2[ob][4 * tt + c] = (valid && c0 + c < p.dout) ? v : 0.f;
                    }
                }
        }
        // ---- dz2 out (for wgrad) + db2 partial ----
        {
            float* drow = p.dz2 != nullptr ? p.dz2 + srow_id * p.dout : nullptr;
#pragma unroll
            for (int ob = 0; ob < OB; ++ob)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int t = ob * 4 + tt;
                    const f32x4 v = acc_chunk(dz2[ob], tt);
                    if (drow != nullptr) store_chunk(drow, p.dout, t, hi, valid, v);
                    *reinterpret_cast<f32x4*>(&stg[j * STG + 8 * t + 4 * hi]) = v;
                }
            wave_lds_sync();
            acc_db2 += tile_colsum(stg, STG, p.dout, lane);
            wave_lds_sync();
        }

        // ---- dh = W2^T dz2 ; dz1 = dh * silu'(z1) ----
        f32x16 dz1[HB];
#pragma unroll
        for (int hb = 0; hb < HB; ++hb)
#pragma unroll
            for (int r = 0; r < 16; ++r) dz1[hb][r] = 0.f;
#pragma unroll
        for (int ob = 0; ob < OB; ++ob)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) mma_chunk<HB>(dz1, W2t, T2, ob * 4 + tt, acc_chunk(dz2[ob], tt), lane);
        {
            const float* zrow = p.z1 + srow_id * p.hid;
            float* drow = p.dz1 != nullptr ? p.dz1 + srow_id * p.hid : nullptr;
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int t = hb * 4 + tt;
                    const f32x4 z = load_chunk(zrow, p.hid, t, hi, valid);
                    f32x4 v;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        v[c] = valid ? dz1[hb][4 * tt + c] * ((p.flags & NLAM_F_NO_ACT) ? 1.f : silu_grad_f(z[c])) : 0.f;
                        dz1[hb][4 * tt + c] = v[c];
                    }
                    if (drow != nullptr) store_chunk(drow, p.hid, t, hi, valid, v);
                    *reinterpret_cast<f32x4*>(&stg[j * STG + 8 * t + 4 * hi]) = v;
                }
            wave_lds_sync();
            acc_db1 += tile_colsum(stg, STG, p.hid, lane);
            wave_lds_sync();
        }

        // ---- dx_s = W1_s^T dz1 per source ----
        for (int s = 0; s < p.nsrc; ++s) {
            const int mode = p.dmode[s];
            if (mode == 0) continue;
            const nlam_src_t S = p.src[s];
            const int w = S.width;
            const float* A = W1t + w1t_off[s];
            const int MBs = (w + 31) >> 5;  // 1 or 2 (kMaxWidth = 64)
            f32x16 dx[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) dx[mb][r] = 0.f;
            if (MBs == 1) {
                f32x16 d1[1];
                d1[0] = dx[0];
#pragma unroll
                for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) mma_chunk<1>(d1, A, T1, hb * 4 + tt, acc_chunk(dz1[hb], tt), lane);
                dx[0] = d1[0];
            } else {
#pragma unroll
                for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) mma_chunk<2>(dx, A, T1, hb * 4 + tt, acc_chunk(dz1[hb], tt), lane);
            }
            // residual paths
            const bool add_gout = (s == 0) && (p.flags & NLAM_F_ADD_SRC0) && grow != nullptr;
            const bool add_gmsg = (s == 1) && (p.flags & NLAM_F_ADD_SRC1);
            float* drow = nullptr;
            if (mode == 1) {
                long ridx = prow;
                if (valid && S.idx != nullptr) ridx = S.idx[prow];
                drow = p.dsrc[s] + (long)b * p.dsrc_bstride[s] + (valid ? ridx : 0) * (long)w;
            } else if (mode == 2) {
                drow = p.dsrc[s] + srow_id * w;
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                if (mb < MBs) {
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        const int t = mb * 4 + tt;
                        f32x4 v = acc_chunk(dx[mb], tt);
                        if (add_gout) v += load_chunk(grow, p.dout, t, hi, valid);
                        if (add_gmsg) {
                            f32x4 g = {0.f, 0.f, 0.f, 0.f};
                            if (grow != nullptr) g += load_chunk(grow, p.dout, t, hi, valid);
                            if (garow != nullptr) g += load_chunk(garow, p.dout, t, hi, valid) * gscale;
                            v += g;
                        }
                        if (mode == 3) {
                            if (!valid) v = f32x4{0.f, 0.f, 0.f, 0.f};
                            *reinterpret_cast<f32x4*>(&stg[j * STG + 8 * t + 4 * hi]) = v;
                        } else {
                            store_chunk(drow, w, t, hi, valid, v);
                        }
                    }
                }
            }
            if (mode == 3) {
                wave_lds_sync();
                tile_segment_reduce(stg, STG, tl, p.rowptr, nullptr, p.dsrc[s] + (long)b * p.dsrc_bstride[s], w, w, lane);
                wave_lds_sync();
            }
        }
    }

    // ---- combine the 8 waves' vector partials through LDS; one row per workgroup ----
    if (p.vec_partials != nullptr) {
        __syncthreads();  // every wave is done with its staging area
        float* red = stg_all;  // kWavesPerBlock x 4 x 64 floats (fits: staging is 8 x 32 x 68)
        red[(wave * 4 + 0) * 64 + lane] = acc_db1;
        red[(wave * 4 + 1) * 64 + lane] = acc_db2;
        red[(wave * 4 + 2) * 64 + lane] = acc_dg;
        red[(wave * 4 + 3) * 64 + lane] = acc_dbt;
        __syncthreads();
        if (wave < 4) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < kWavesPerBlock; ++w) s += red[(w * 4 + wave) * 64 + lane];
            p.vec_partials[((size_t)blockIdx.x * 4 + wave) * p.vec_stride + lane] = s;
        }
    }
}

// ---------------------------------------------------------------------------
// backward, FAST shapes (hid == 32*HB, dout == 32*OB, every source with a data gradient
// 32 or 64 wide): no bounds / scalar tails, rows leave through the per-wave [32][36] LDS
// block as whole 128-B lines, column sums (bias / LayerNorm-affine gradients) are taken
// from the same staged blocks, and the three GEMM groups (dh = W2^T dz2, dx_s = W1_s^T dz1)
// run on the fp32 MFMA (NS = 0) or the split-bf16 matrix cores (NS = 1..3) with their B
// operands taken straight from the accumulator registers (slot-permuted K order, see the
// split-bf16 notes above).
// ---------------------------------------------------------------------------
// 16 zero bytes: the source of a load whose tensor is absent (a null row pointer becomes this address with stride 0, so the
// load stays unconditional) and of LDS-DMA lanes past the end of a row

// column sums of the wave's staged 32-column block: lane -> (column, half of the rows)
__device__ __forceinline__ float block_colsum_half(const float* stg, int lane) {
    const int c = lane & 31, r0 = (lane >> 5) * 16;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += stg[(r0 + q) * kStgStride + c];
    return s;
}

// LW (NLAM_F_LEAF_WGRAD): a leaf MLP of <= 4 input columns (the embedders of the static features) accumulates its own
// weight gradients: dW2 = dz2^T silu(z1) as a transposed MFMA product of the blocks the kernel stages anyway (rows are
// the K dimension: lane (i, hi) reads rows 8 hi .. 8 hi + 7 of column i of the row-major staged block), dW1[:, k] as
// column sums of dz1 weighted with input column k.  dz1 / dz2 never leave the chip; p.dz2 receives the workgroup's
// (dout x hid) partial, vec_partials has 7 rows per workgroup (db1, db2, dgamma, dbeta, dW1[:, 0..2]).
// RO: the output width is not a whole 32-column block (output_map, graph/base.py:322: dout = 17, no LayerNorm): g_out rows are
// read element-wise and zero-filled past dout, dz2 is written with a row stride of OB * 32 floats (p.dz2_ld; zero columns past
// dout), so that its weight gradient runs on the LDS-DMA kernel with m = OB * 32.
template <int HB, int OB, int NS, bool LW = false, int NWV = kWavesPerBlock, bool RO = false>
__device__ __forceinline__ void mlp_bwd_fast_body(const nlam_mlp_bwd_t& p, const int wg_id, const int wg_count) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int DPH = HB * 32, OP = OB * 32;
    constexpr int NSW = NS > 0 ? NS : 1;
    constexpr int T2 = OP / 8, T1 = DPH / 8;     // fp32 K chunks of dh (K = dout) and dx (K = hid)
    constexpr int S2 = OP / 16, S1 = DPH / 16;   // split-bf16 K steps
    NLAM_T_DECL
#ifdef NLAM_TIMING
    int t_ntiles_ = 0;
#endif

    // NLAM_F_PRE_ADD (factorised edge MLP): only source 0 has columns in W1; sources 1.. are pre-activation addends whose
    // gradient is dz1 itself (segment-summed over the tile's receivers for dmode 3, left to the caller otherwise)
    const bool pre = (p.flags & NLAM_F_PRE_ADD) != 0;
    int kin = 0;
    for (int s = 0; s < (pre ? 1 : p.nsrc); ++s) kin += p.src[s].width;
    if (pre && p.ldw1 > 0) kin = p.ldw1;   // floats between rows of W1
    auto gemm_src = [&](int s) { return s < p.nsrc && p.dmode[s] != 0 && (!pre || s == 0); };

    // ---- weights: W2^T (hid x dout) then W1_s^T (w_s x hid) per source with a data gradient ----
    const size_t w2_floats = NS > 0 ? (size_t)NS * DPH * OP / 2 : (size_t)DPH * OP;
    float* W2t = smem;
    float* W1t = W2t + w2_floats;
    size_t w1_floats = 0;
    int w1_off[NLAM_MAX_SRC] = {0, 0, 0};
#pragma unroll
    for (int s = 0; s < NLAM_MAX_SRC; ++s) {
        w1_off[s] = (int)w1_floats;
        if (gemm_src(s)) w1_floats += NS > 0 ? (size_t)NS * p.src[s].width * DPH / 2 : (size_t)p.src[s].width * DPH;
    }
    float* gml = W1t + w1_floats;                         // OP
    float* stg_all = gml + OP;                            // NWV x 32 x kStgStride
    float* colacc_all = stg_all + (size_t)NWV * 32 * kStgStride;   // NWV x 8 x 64 column accumulators
    float* xs_all = colacc_all + (size_t)NWV * 8 * 64;             // LW: NWV x [32][4] input rows of the tile
    float* w1l = xs_all + (size_t)NWV * 32 * 4;                    // LW with z1 == NULL: W1 rows padded to 4 columns [DPH][4], then b1 [DPH]
    if (NS > 0 && p.wpack != nullptr) {
        // pre-packed image (nlam_mlp_pack): [W2^T | W1_0^T | W1_1^T | W1_2^T], a W1 piece for every source whose width is a
        // multiple of 32 (source 0 only for the factorised MLP); pieces of sources without a data gradient are skipped
        lds_dma_copy(W2t, p.wpack, w2_floats * sizeof(float));
        size_t ioff = w2_floats;
        for (int s = 0; s < p.nsrc; ++s) {
            const int w = p.src[s].width;
            const size_t piece = ((w & 31) == 0 && (!pre || s == 0)) ? (size_t)NS * w * DPH / 2 : 0;
            if (gemm_src(s)) lds_dma_copy(W1t + w1_off[s], p.wpack + ioff, piece * sizeof(float));
            ioff += piece;
        }
    } else if constexpr (NS > 0) {
        // A[m = hidden][k = out (slot-permuted)] = W2[k][m]
        stage_split<NSW>(reinterpret_cast<u32x4*>(W2t), S2, 0, p.W2, 1, p.hid, HB, p.dout, true, p.hid, OP);
        int off = 0;
        for (int s = 0; s < p.nsrc; ++s) {
            const int w = p.src[s].width;
            if (gemm_src(s))   // A[m = source column][k = hidden (slot-permuted)] = W1[k][off + m]
                stage_split<NSW>(reinterpret_cast<u32x4*>(W1t + w1_off[s]), S1, 0, p.W1 + off, 1, w, w >> 5, p.hid, true, kin);
            off += w;
        }
    } else {
        stage_packed(W2t, T2, 0, p.W2, 1, p.hid, p.hid, HB, p.dout);
        int off = 0;
        for (int s = 0; s < p.nsrc; ++s) {
            const int w = p.src[s].width;
            if (gemm_src(s)) stage_packed(W1t + w1_off[s], T1, 0, p.W1 + off, 1, kin, w, w >> 5, p.hid);
            off += w;
        }
    }
    stage_vec(gml, p.ln_w, p.dout, OP, 1.f);
    // LW always recomputes z1 = W1 x + b1 from the tile's input row (the grouped forward does not save it); a run-time choice
    // between that and a load would put a full wait for every outstanding request -- the next tile's rows -- behind the branch
    constexpr bool recompute_z = LW;
    if constexpr (LW) {
        const int kw = p.src[0].width;
        for (int e = threadIdx.x; e < DPH * 4; e += blockDim.x) {
            const int f_ = e >> 2, k_ = e & 3;
            w1l[e] = (f_ < p.hid && k_ < kw) ? p.W1[(long)f_ * kw + k_] : 0.f;
        }
        for (int e = threadIdx.x; e < DPH; e += blockDim.x) w1l[DPH * 4 + e] = e < p.hid ? p.b1[e] : 0.f;
    }
    __syncthreads();
    NLAM_T_MARK(0)

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 31, hi = lane >> 5;
    float* stg = stg_all + (size_t)wave * 32 * kStgStride;
    const bool has_ln = p.ln_w != nullptr;
    const float inv_dout = 1.f / (float)p.dout;
    const bool add_gout = (p.flags & NLAM_F_ADD_SRC0) != 0 && p.g_out != nullptr && p.dmode[0] != 0;
    const bool add_gmsg = (p.flags & NLAM_F_ADD_SRC1) != 0 && p.nsrc > 1 && p.dmode[1] != 0;

    // per-lane column accumulators, lane -> (column lane & 31, row half lane >> 5) of each 32-column block; they
    // live in LDS (8 per lane: db1, db2, dgamma, dbeta x 2 blocks) because the kernel sits at the 256-VGPR limit
    float* colacc = colacc_all + (size_t)wave * 8 * 64 + lane;
    enum { kDb1 = 0, kDb2 = 2, kDg = 4, kDbt = 6 };
    // In registers (kColRegs) they cost far more than eight: <2,2,3> goes from 196 to 249 VGPRs (measured, round 3), and the
    // register count of a chain kernel decides which side-stream workgroups can share its CU (NLAM_BWD_R_EARLY); the
    // fused-weight-gradient variant (452 registers) spills 12 VGPRs with them.  They stay in LDS.
    constexpr bool kColRegs = false;
    if constexpr (!kColRegs) {
#pragma unroll
        for (int k = 0; k < 8; ++k) colacc[k * 64] = 0.f;
    }
    float creg[kColRegs ? 8 : 1];
#pragma unroll
    for (int k = 0; k < (kColRegs ? 8 : 1); ++k) creg[k] = 0.f;
    auto cadd = [&](int k, float v) {   // k is a compile-time constant at every (unrolled) call site
        if constexpr (kColRegs) creg[k] += v;
        else colacc[k * 64] += v;
    };
    auto cget = [&](int k) -> float {
        if constexpr (kColRegs) return creg[k];
        else return colacc[k * 64];
    };
    float* xs = xs_all + (size_t)wave * 32 * 4;
    f32x16 dw2[LW ? OB : 1][LW ? HB : 1];   // LW: this wave's share of dW2 (block (ob, hb): rows = output features, lane & 31 = hidden feature)
    float dw1c[LW ? HB : 1][3];             // LW: this lane's share of dW1[32 hb + (lane & 31)][k]
    if constexpr (LW) {
#pragma unroll
        for (int ob = 0; ob < OB; ++ob)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int r = 0; r < 16; ++r) dw2[ob][hb][r] = 0.f;
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) dw1c[hb][0] = dw1c[hb][1] = dw1c[hb][2] = 0.f;
    }

    const long total_tiles = (long)p.ntiles * p.batch;
    // LW (one wave per SIMD: nothing else hides a load, and 512 registers per lane to itself): every global row of a tile --
    // g_out, xhat, rstd, the input row -- is requested a whole tile ahead.  The launcher admits plain 32-row tiles in identity
    // row order only (nlam_mlp_bwd_group), so the next tile's rows are known without loads; the requests are unconditional
    // (a wave's last tile asks for the last tile again): with a conditional request the compiler has to merge the wait
    // counters of both paths and waits for the fresh requests at the next use of ANY loaded value.
    f32x4 gpre[LW ? OB : 1][4];
    f32x4 xpre[LW ? OB : 1][4];
    float rpre = 0.f;
    float xipre[4] = {0.f, 0.f, 0.f, 0.f};
    auto lw_row = [&](long gn, int& bn, long& rown) {
        gn = gn < total_tiles ? gn : total_tiles - 1;
        bn = (int)(gn / p.ntiles);
        rown = min((long)(gn % p.ntiles) * 32 + j, (long)p.rows - 1);
    };
    auto lw_request_g = [&](long gn) {
        if constexpr (!LW) return;   // (no captures, no closure object, in the other instantiations)
        int bn;
        long rown;
        lw_row(gn, bn, rown);
        const float* gnrow = p.g_out + (long)bn * p.out_bstride + rown * p.dout;
#pragma unroll
        for (int o2 = 0; o2 < (LW ? OB : 1); ++o2)
#pragma unroll
            for (int t2 = 0; t2 < 4; ++t2) gpre[o2][t2] = *reinterpret_cast<const f32x4*>(gnrow + 8 * (o2 * 4 + t2) + 4 * hi);
    };
    auto lw_request_x = [&](long gn) {
        if constexpr (!LW) return;
        int bn;
        long rown;
        lw_row(gn, bn, rown);
        const size_t srn = (size_t)bn * p.rows + rown;
        {   // without a LayerNorm the same requests read g_out again (values unused): unconditional, see above
            const float* xn = has_ln ? p.xhat + srn * p.dout : p.g_out + (long)bn * p.out_bstride + rown * p.dout;
            if constexpr (NLAM_LW_PREFETCH_X == 1) {
#pragma unroll
                for (int o2 = 0; o2 < (LW ? OB : 1); ++o2)
#pragma unroll
                    for (int t2 = 0; t2 < 4; ++t2) xpre[o2][t2] = *reinterpret_cast<const f32x4*>(xn + 8 * (o2 * 4 + t2) + 4 * hi);
            }
            rpre = *(has_ln ? p.rstd + srn : xn);
        }
        const int kw = p.src[0].width;
        const float* xr = p.src[0].ptr + (long)bn * p.src[0].bstride + rown * kw;
#pragma unroll
        for (int k = 0; k < 4; ++k) xipre[k] = xr[k < kw ? k : 0];   // <= 4 columns: clamp, select below
#pragma unroll
        for (int k = 0; k < 4; ++k) xipre[k] = k < kw ? xipre[k] : 0.f;
    };
    if constexpr (LW) {
        lw_request_g((long)wave * wg_count + wg_id);
        lw_request_x((long)wave * wg_count + wg_id);
    }
    for (long gt = (long)wave * wg_count + wg_id; gt < total_tiles; gt += (long)wg_count * NWV) {
        const int b = (int)(gt / p.ntiles);
        const TileInfo tl = get_tile(LW ? nullptr : p.tiles, (int)(gt % p.ntiles), p.rows);
        const bool valid = j < tl.nrows;
        const int prow = tl.row0 + j;
        const int prow_c = min(tl.row0 + max(min(j, tl.nrows - 1), 0), p.rows - 1);   // clamped: loads never fault
        const size_t srow_c = (size_t)b * p.rows + prow_c;
        const size_t tile_row0 = (size_t)b * p.rows + tl.row0;

#ifdef NLAM_TIMING
        ++t_ntiles_;
#endif
        // ---- indices ----
        // (LW: the launcher admits plain row-ordered leaf MLPs only -- no output index, no aggregated gradient, no gathered source,
        // no receivers; none of these conditional loads may exist in its tile loop, see the prefetch above)
        int oidx = prow_c;
        int sg = 0;
        float gscale = 1.f;
        int sidx[NLAM_MAX_SRC] = {prow_c, prow_c, prow_c};
        int raw_ptr = 0;
        if constexpr (!LW) {
            if (p.g_out != nullptr && p.out_idx != nullptr) oidx = p.out_idx[prow_c];
            if (p.g_aggr != nullptr) {
                sg = p.seg_of_row[prow_c];
                if (p.flags & NLAM_F_MEAN) gscale = p.inv_deg[sg];
            }
#pragma unroll
            for (int s = 0; s < NLAM_MAX_SRC; ++s)
                if (s < p.nsrc && p.dmode[s] == 1 && p.src[s].idx != nullptr) sidx[s] = p.src[s].idx[prow_c];
            if (p.rowptr != nullptr && !tl.split && lane <= tl.nseg) raw_ptr = p.rowptr[tl.seg0 + lane];
        }
        constexpr bool from_pre = LW;   // this tile's g_out / xhat rows, rstd and input row were requested a tile ago
        float rstd = 0.f;
        // LW: take this tile's rows out of the prefetch registers and request the next tile's right away: they have the whole
        // tile to arrive, and no other global load (hence no wait on the load counter) follows before the next tile's top
        f32x4 gcur[LW ? OB : 1][4], xcur[LW ? OB : 1][4];
        float xicur[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (LW) {
            rstd = rpre;
#pragma unroll
            for (int o2 = 0; o2 < OB; ++o2)
#pragma unroll
                for (int t2 = 0; t2 < 4; ++t2) {
                    gcur[o2][t2] = gpre[o2][t2];
                    if constexpr (NLAM_LW_PREFETCH_X == 1) {
                        xcur[o2][t2] = xpre[o2][t2];
                    } else {   // requested first: the wait for it does not cover the next tile's requests below
                        const float* xc = has_ln ? p.xhat + srow_c * p.dout : p.g_out;
                        xcur[o2][t2] = *reinterpret_cast<const f32x4*>(xc + (has_ln ? 8 * (o2 * 4 + t2) + 4 * hi : 0));
                    }
                }
#pragma unroll
            for (int k = 0; k < 4; ++k) xicur[k] = xipre[k];
            lw_request_g(gt + (long)wg_count * NWV);
            lw_request_x(gt + (long)wg_count * NWV);
        } else {
            rstd = has_ln ? p.rstd[srow_c] : 0.f;
        }
        const float* grow = p.g_out != nullptr ? p.g_out + (long)b * p.out_bstride + (long)oidx * p.dout : nullptr;
        const float* garow = p.g_aggr != nullptr ? p.g_aggr + ((size_t)b * p.nseg_total + sg) * p.dout : nullptr;
        const float* xrow = has_ln ? p.xhat + srow_c * p.dout : nullptr;
        const float* zrow = p.z1 != nullptr ? p.z1 + srow_c * p.hid : nullptr;
        if constexpr (LW) {   // the tile's input rows (<= 4 columns), zero past the width and past the tile's rows
            if (hi == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) xs[j * 4 + k] = valid ? xicur[k] : 0.f;
            }
            wave_lds_sync();
        }

        // Every row the tile needs -- g_out, the aggregated gradient, xhat, z1 -- is requested here, unconditionally and back to
        // back (an absent tensor reads one zero chunk: pointer select + stride 0).  As `if (ptr) g += load` per chunk, each chunk
        // was a branch with its own full wait (the compiler merges the wait counters of both paths): 8 + 8 serial round trips per
        // tile, the z1 ones behind the dz2 stores on top (a load behind a store waits for the store's acknowledgement).
        f32x4 xhs[LW ? 1 : OB][4];
        f32x4 gq[LW ? 1 : OB][4], aq[LW ? 1 : OB][4], zq[LW ? 1 : HB][4];
        if constexpr (!LW) {
            const float* gp = grow != nullptr ? grow : g_zero16;
            const float* ap = garow != nullptr ? garow : g_zero16;
            const float* xp = has_ln ? xrow : g_zero16;
            const int gm = grow != nullptr ? 1 : 0, am = garow != nullptr ? 1 : 0, xm = has_ln ? 1 : 0;
            if constexpr (NLAM_BWD_G_BATCH == 2) {
#pragma unroll
                for (int ob = 0; ob < OB; ++ob)
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        const int c0 = 8 * (ob * 4 + tt) + 4 * hi;
                        if constexpr (!RO) gq[ob][tt] = *reinterpret_cast<const f32x4*>(gp + gm * c0);
                        aq[ob][tt] = *reinterpret_cast<const f32x4*>(ap + am * c0);
                        xhs[ob][tt] = *reinterpret_cast<const f32x4*>(xp + xm * c0);
                    }
                __builtin_amdgcn_sched_barrier(0);   // the scheduler otherwise pulls the first chunks' arithmetic (and their waits) in between the requests
            }
        }
        auto request_block = [&](int ob) {   // NLAM_BWD_G_BATCH == 1
            const float* gp = grow != nullptr ? grow : g_zero16;
            const float* ap = garow != nullptr ? garow : g_zero16;
            const float* xp = has_ln ? xrow : g_zero16;
            const int gm = grow != nullptr ? 1 : 0, am = garow != nullptr ? 1 : 0, xm = has_ln ? 1 : 0;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int c0 = 8 * (ob * 4 + tt) + 4 * hi;
                if constexpr (!RO) gq[LW ? 0 : ob][tt] = *reinterpret_cast<const f32x4*>(gp + gm * c0);
                aq[LW ? 0 : ob][tt] = *reinterpret_cast<const f32x4*>(ap + am * c0);
                xhs[LW ? 0 : ob][tt] = *reinterpret_cast<const f32x4*>(xp + xm * c0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto request_chunk = [&](int ob, int tt) {   // NLAM_BWD_G_BATCH == 0
            const float* gp = grow != nullptr ? grow : g_zero16;
            const float* ap = garow != nullptr ? garow : g_zero16;
            const float* xp = has_ln ? xrow : g_zero16;
            const int gm = grow != nullptr ? 1 : 0, am = garow != nullptr ? 1 : 0, xm = has_ln ? 1 : 0;
            const int c0 = 8 * (ob * 4 + tt) + 4 * hi;
            if constexpr (!RO) gq[LW ? 0 : ob][tt] = *reinterpret_cast<const f32x4*>(gp + gm * c0);
            aq[LW ? 0 : ob][tt] = *reinterpret_cast<const f32x4*>(ap + am * c0);
            xhs[LW ? 0 : ob][tt] = *reinterpret_cast<const f32x4*>(xp + xm * c0);
        };

        NLAM_T_MARK(1)
        // ---- dmsg (C-layout chunks), LayerNorm backward ----
        f32x16 dz2[OB];
        {
            float m1 = 0.f, m2 = 0.f;
            // xhat stays in registers between its two uses (the big edge sets are HBM-bound: one read); the fused-weight-gradient
            // variant carries 64 accumulator registers through the tile loop and takes it from its prefetch registers
#pragma unroll
            for (int ob = 0; ob < OB; ++ob) {
                f32x4(&xh)[4] = xhs[LW ? 0 : ob];
                if constexpr (!LW && NLAM_BWD_G_BATCH == 1) request_block(ob);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int c0 = 8 * (ob * 4 + tt) + 4 * hi;
                    if constexpr (!LW && NLAM_BWD_G_BATCH == 0) request_chunk(ob, tt);
                    f32x4 g = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (from_pre) g = gcur[LW ? ob : 0][tt];
                    else if (RO) {
                        if (grow != nullptr) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) g[c] = c0 + c < p.dout ? grow[c0 + c] : 0.f;
                        }
                    } else {
                        g = gq[LW ? 0 : ob][tt];
                    }
                    if constexpr (!LW) g += aq[LW ? 0 : ob][tt] * gscale;
                    if (!valid) g = f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (from_pre) {
                        if (has_ln) xh[tt] = xcur[LW ? ob : 0][tt];
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) dz2[ob][4 * tt + c] = g[c];
                }
                if constexpr (LW) { NLAM_T_MARK(8) }
                if (has_ln) {
                    // dbeta = colsum(dmsg), dgamma = colsum(dmsg * xhat), through the staged block
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) *reinterpret_cast<f32x4*>(&stg[j * kStgStride + 8 * tt + 4 * hi]) = acc_chunk(dz2[ob], tt);
                    wave_lds_sync();
                    cadd(kDbt + ob, block_colsum_half(stg, lane));
                    wave_lds_sync();
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
                        *reinterpret_cast<f32x4*>(&stg[j * kStgStride + 8 * tt + 4 * hi]) = acc_chunk(dz2[ob], tt) * xh[tt];
                    wave_lds_sync();
                    cadd(kDg + ob, block_colsum_half(stg, lane));
                    wave_lds_sync();
                    if constexpr (LW) { NLAM_T_MARK(9) }
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        const f32x4 gm = *reinterpret_cast<const f32x4*>(&gml[8 * (ob * 4 + tt) + 4 * hi]);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float gy = dz2[ob][4 * tt + c] * gm[c];
                            dz2[ob][4 * tt + c] = gy;
                            m1 += gy;
                            m2 += gy * xh[tt][c];
                        }
                    }
                    if constexpr (LW) { NLAM_T_MARK(10) }
                }
            }
            if (has_ln) {
                m1 = row_allreduce(m1) * inv_dout;
                m2 = row_allreduce(m2) * inv_dout;
#pragma unroll
                for (int ob = 0; ob < OB; ++ob)
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        const f32x4 xh = LW ? xcur[LW ? ob : 0][tt] : xhs[LW ? 0 : ob][tt];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float v = rstd * (dz2[ob][4 * tt + c] - m1 - xh[c] * m2);
                            dz2[ob][4 * tt + c] = valid ? v : 0.f;
                        }
                    }
            }
        }
        NLAM_T_MARK(2)
        if constexpr (!LW && NLAM_BWD_Z_EARLY == 1) {   // z1 rows: requested ahead of the dz2 stores, needed behind the dh GEMM
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) zq[hb][tt] = *reinterpret_cast<const f32x4*>(zrow + 8 * (hb * 4 + tt) + 4 * hi);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- dz2 rows out (for wgrad) + db2 ----  (LW: db2 falls out of the transposed dz2 fragments of dW2 below)
#pragma unroll
        for (int ob = 0; ob < (LW ? 0 : OB); ++ob) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) *reinterpret_cast<f32x4*>(&stg[j * kStgStride + 8 * tt + 4 * hi]) = acc_chunk(dz2[ob], tt);
            wave_lds_sync();
            cadd(kDb2 + ob, block_colsum_half(stg, lane));
            if (!LW && p.dz2 != nullptr) {
                const size_t ldz2 = RO ? (size_t)OP : (size_t)p.dout;
                float* dbase = p.dz2 + tile_row0 * ldz2 + 32 * ob;
                block_rows_out(stg, tl.nrows, 32, lane, [&](int r) { return dbase + (size_t)r * ldz2; });
            }
            wave_lds_sync();
        }

        NLAM_T_MARK(3)
        // out = msg + src0 (edge update / node residual): d src0 += g_out.  NLAM_BWD_R_EARLY = 1 requests its rows again here, ahead
        // of the dz1 stores; the shipped 0 re-reads them chunk by chunk behind the dx GEMM of source 0 (see the macro)
        f32x4 rq[LW ? 1 : OB][4];
        if constexpr (!LW && (NLAM_BWD_R_EARLY == 1 || NLAM_BWD_R_EARLY == 3)) {
            const float* rp = add_gout ? grow : g_zero16;
            const int rm = add_gout ? 1 : 0;
#pragma unroll
            for (int mb = 0; mb < OB; ++mb)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) rq[mb][tt] = *reinterpret_cast<const f32x4*>(rp + rm * (8 * (mb * 4 + tt) + 4 * hi));
        }
        // ---- dh = W2^T dz2 ; dz1 = dh * silu'(z1) ----
        f32x16 dz1[HB];
#pragma unroll
        for (int hb = 0; hb < HB; ++hb)
#pragma unroll
            for (int r = 0; r < 16; ++r) dz1[hb][r] = 0.f;
        if constexpr (NS > 0) {
#pragma unroll
            for (int ob = 0; ob < OB; ++ob)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float xs8[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) xs8[q] = dz2[ob][8 * h + q];
                    const BfFrag<NSW> Bf = split8<NSW>(xs8);
                    mma_split_lds<NSW, HB>(dz1, reinterpret_cast<const u32x4*>(W2t), HB, S2, 2 * ob + h, lane, Bf);
                }
        } else {
#pragma unroll
            for (int ob = 0; ob < OB; ++ob)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) mma_chunk<HB>(dz1, W2t, T2, ob * 4 + tt, acc_chunk(dz2[ob], tt), lane);
        }
        // LW: dz2^T fragments (A operand of dW2: feature lane & 31, rows 16 s + 8 hi .. + 7), made AFTER the dh GEMM so that
        // they are not live across it (the blocks are re-staged from the registers the GEMM has just read)
        BfFrag<NSW> a2[LW ? OB : 1][2];
        if constexpr (LW) { NLAM_T_MARK(11) }
        if constexpr (LW) {
#pragma unroll
            for (int ob = 0; ob < OB; ++ob) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) *reinterpret_cast<f32x4*>(&stg[j * kStgStride + 8 * tt + 4 * hi]) = acc_chunk(dz2[ob], tt);
                wave_lds_sync();
                float cs = 0.f;   // db2: this lane reads column j, rows 8 hi .. + 7 and 16 + 8 hi .. + 7 -- half of the column's sum
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    float a8[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) a8[q] = stg[(16 * s2 + 8 * hi + q) * kStgStride + j];
#pragma unroll
                    for (int q = 0; q < 8; ++q) cs += a8[q];
                    a2[ob][s2] = split8<NSW>(a8);
                    __builtin_amdgcn_sched_barrier(0);   // one fragment at a time (the scheduler otherwise interleaves the conversions: VGPRs)
                }
                cadd(kDb2 + ob, cs);
                wave_lds_sync();
            }
            NLAM_T_MARK(5)
        }
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) {
            f32x4 hh[LW ? 4 : 1];   // LW: silu(z1) of this block, the B operand of dW2
            if constexpr (!LW && NLAM_BWD_Z_EARLY == 0) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) zq[hb][tt] = *reinterpret_cast<const f32x4*>(zrow + 8 * (hb * 4 + tt) + 4 * hi);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                f32x4 z;
                if constexpr (recompute_z) {   // z1 = W1 x + b1 from the tile's input row (three FMAs per element) instead of a saved copy
                    const int f0 = 8 * (hb * 4 + tt) + 4 * hi;
                    const f32x4 xr = *reinterpret_cast<const f32x4*>(&xs[j * 4]);
                    z = *reinterpret_cast<const f32x4*>(&w1l[DPH * 4 + f0]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const f32x4 wr = *reinterpret_cast<const f32x4*>(&w1l[(f0 + c) * 4]);
                        z[c] += wr[0] * xr[0] + wr[1] * xr[1] + wr[2] * xr[2] + wr[3] * xr[3];
                    }
                } else {
                    z = zq[LW ? 0 : hb][tt];
                }
                f32x4 v;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if constexpr (LW) {   // one sigmoid for silu and its derivative
                        const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-z[c]));
                        v[c] = valid ? dz1[hb][4 * tt + c] * (sg * (1.f + z[c] * (1.f - sg))) : 0.f;
                        hh[tt][c] = valid ? z[c] * sg : 0.f;
                    } else {
                        v[c] = valid ? dz1[hb][4 * tt + c] * silu_grad_f(z[c]) : 0.f;
                    }
                    dz1[hb][4 * tt + c] = v[c];
                }
                *reinterpret_cast<f32x4*>(&stg[j * kStgStride + 8 * tt + 4 * hi]) = v;
            }
            wave_lds_sync();
            cadd(kDb1 + hb, block_colsum_half(stg, lane));
            if constexpr (LW) { NLAM_T_MARK(7) }
            if constexpr (LW) {
                __builtin_amdgcn_sched_barrier(0);
                // dW1[:, k] += sum_rows dz1[row][:] * x[row][k]: the column sums of db1 with the input column as weight
                const int c_ = lane & 31, r0_ = (lane >> 5) * 16;
                float w0 = 0.f, w1 = 0.f, w2 = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float dv = stg[(r0_ + q) * kStgStride + c_];
                    const f32x4 xr = *reinterpret_cast<const f32x4*>(&xs[(r0_ + q) * 4]);
                    w0 += dv * xr[0];
                    w1 += dv * xr[1];
                    w2 += dv * xr[2];
                }
                dw1c[hb][0] += w0;
                dw1c[hb][1] += w1;
                dw1c[hb][2] += w2;
                wave_lds_sync();
                // dW2[ob][hb] += dz2_ob^T . silu(z1)_hb  (K = the tile's 32 rows = two K steps)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) *reinterpret_cast<f32x4*>(&stg[j * kStgStride + 8 * tt + 4 * hi]) = hh[tt];
                wave_lds_sync();
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    float b8[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) b8[q] = stg[(16 * s2 + 8 * hi + q) * kStgStride + j];
                    const BfFrag<NSW> bh = split8<NSW>(b8);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ord = NSW - 1; ord >= 0; --ord)
#pragma unroll
                        for (int pa = 0; pa <= ord; ++pa)
#pragma unroll
                            for (int ob = 0; ob < OB; ++ob) {   // accumulators alternate (OB = 2) or are fenced (see mma_split_lds)
                                dw2[ob][hb] = MFMA_BF16(a2[ob][s2].t[pa], bh.t[ord - pa], dw2[ob][hb]);
                                if (OB == 1) {
                                    asm volatile("s_nop 15");
                                    asm volatile("s_nop 15");
                                }
                            }
                }
            }
            if (!LW && p.dz1 != nullptr) {
                float* dbase = p.dz1 + tile_row0 * p.hid + 32 * hb;
                block_rows_out(stg, tl.nrows, 32, lane, [&](int r) { return dbase + (size_t)r * p.hid; });
            }
            if (pre) {   // gradient of a receiver-gathered addend = dz1 summed over each receiver's rows (all inside this tile)
#pragma unroll
                for (int s = 1; s < NLAM_MAX_SRC; ++s)
                    if (s < p.nsrc && p.dmode[s] == 3)
                        block_segment_reduce(stg, tl, raw_ptr, 1.f, p.dsrc[s] + (long)b * p.dsrc_bstride[s] + 32 * hb, p.hid, 32, lane);
            }
            wave_lds_sync();
        }

        NLAM_T_MARK(4)
        if constexpr (!LW) {   // a leaf MLP has no data gradients: the whole stage is compiled out of the fused-weight-gradient variant
        // ---- dx_s = W1_s^T dz1 per source ----
        // split-bf16: the B fragments of dz1 are the same for every source -- convert once
        BfFrag<NSW> bz[NS > 0 ? HB * 2 : 1];
        if constexpr (NS > 0) {
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float xs8[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) xs8[q] = dz1[hb][8 * h + q];
                    bz[2 * hb + h] = split8<NSW>(xs8);
                    __builtin_amdgcn_sched_barrier(0);   // one fragment at a time (the scheduler otherwise interleaves all conversions: VGPRs)
                }
        }
#pragma unroll
        for (int s = 0; s < NLAM_MAX_SRC; ++s) {
            if (!gemm_src(s)) continue;
            const int mode = p.dmode[s];
            const int w = p.src[s].width;
            const int MBs = w >> 5;   // 1 or 2
            f32x16 dx[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) dx[mb][r] = 0.f;
            if constexpr (NLAM_BWD_R_EARLY == 3) {   // the residual rows ARE the initial accumulators of source 0 (chunk layout = accumulator layout)
                if (s == 0) {
#pragma unroll
                    for (int mb = 0; mb < OB; ++mb)
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                            for (int c = 0; c < 4; ++c) dx[mb][4 * tt + c] = rq[LW ? 0 : mb][tt][c];
                }
            }
            if constexpr (NLAM_BWD_R_EARLY == 2) {   // residual rows of source 0: requested right ahead of its dx GEMM
                if (s == 0) {
                    const float* rp = add_gout ? grow : g_zero16;
                    const int rm = add_gout ? 1 : 0;
#pragma unroll
                    for (int mb = 0; mb < OB; ++mb)
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) rq[mb][tt] = *reinterpret_cast<const f32x4*>(rp + rm * (8 * (mb * 4 + tt) + 4 * hi));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (NS > 0) {
                const u32x4* A1 = reinterpret_cast<const u32x4*>(W1t + w1_off[s]);
                if (MBs == 2) {
#pragma unroll
                    for (int st = 0; st < HB * 2; ++st) mma_split_lds<NSW, 2>(dx, A1, 2, S1, st, lane, bz[st]);
                } else {
                    f32x16 d1[1];
                    d1[0] = dx[0];
#pragma unroll
                    for (int st = 0; st < HB * 2; ++st) mma_split_lds<NSW, 1>(d1, A1, 1, S1, st, lane, bz[st]);
                    dx[0] = d1[0];
                }
            } else {
                const float* A1 = W1t + w1_off[s];
                if (MBs == 1) {
                    f32x16 d1[1];
                    d1[0] = dx[0];
#pragma unroll
                    for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) mma_chunk<1>(d1, A1, T1, hb * 4 + tt, acc_chunk(dz1[hb], tt), lane);
                    dx[0] = d1[0];
                } else {
#pragma unroll
                    for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) mma_chunk<2>(dx, A1, T1, hb * 4 + tt, acc_chunk(dz1[hb], tt), lane);
                }
            }
            float* dbase = p.dsrc[s] + (long)b * p.dsrc_bstride[s];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                if (mb >= MBs) continue;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    f32x4 v = acc_chunk(dx[mb], tt);
                    if constexpr (NLAM_BWD_R_EARLY == 1 || NLAM_BWD_R_EARLY == 2) {
                        if (s == 0 && mb < OB) v += rq[LW ? 0 : (mb < OB ? mb : 0)][tt];   // zeros unless out = msg + src0
                    } else if constexpr (NLAM_BWD_R_EARLY == 0) {
                        if (s == 0 && add_gout) v += *reinterpret_cast<const f32x4*>(grow + 8 * (mb * 4 + tt) + 4 * hi);
                    }
                    if (s == 1 && add_gmsg) {   // msg = mlp + src1: d src1 += dmsg (re-read: rare PropagationNet path)
                        f32x4 g = {0.f, 0.f, 0.f, 0.f};
                        const int c0 = 8 * (mb * 4 + tt) + 4 * hi;
                        if (grow != nullptr) g += *reinterpret_cast<const f32x4*>(grow + c0);
                        if (garow != nullptr) g += *reinterpret_cast<const f32x4*>(garow + c0) * gscale;
                        v += g;
                    }
                    if (!valid) v = f32x4{0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4*>(&stg[j * kStgStride + 8 * tt + 4 * hi]) = v;
                }
                wave_lds_sync();
                if (mode == 1) {
                    block_rows_out(stg, tl.nrows, 32, lane, [&](int r) { return dbase + (long)__shfl(sidx[s], r, 64) * w + 32 * mb; });
                } else if (mode == 2) {
                    float* tb = p.dsrc[s] + (tile_row0 * w) + 32 * mb;
                    block_rows_out(stg, tl.nrows, 32, lane, [&](int r) { return tb + (size_t)r * w; });
                } else {
                    block_segment_reduce(stg, tl, raw_ptr, 1.f, dbase + 32 * mb, w, 32, lane);
                }
                wave_lds_sync();
            }
        }
        }
        NLAM_T_MARK(5)
    }

    NLAM_T_DRAIN
    NLAM_T_MARK(6)
    NLAM_T_FLUSH(t_ntiles_)
    if constexpr (LW) {
        // ---- the workgroup's dW2 partial: the eight waves' 32 x 32 blocks summed through the staging blocks ----
        float* P2 = p.dz2 + (size_t)wg_id * p.dout * p.hid;
#pragma unroll
        for (int ob = 0; ob < OB; ++ob)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                __syncthreads();
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)   // stg[n = hidden feature][m = output feature]
                    *reinterpret_cast<f32x4*>(&stg[j * kStgStride + 8 * tt + 4 * hi]) = acc_chunk(dw2[ob][hb], tt);
                __syncthreads();
                for (int e = threadIdx.x; e < 1024; e += (NWV * 64)) {
                    const int n = e & 31, m = e >> 5;
                    float sum = 0.f;
#pragma unroll
                    for (int w = 0; w < NWV; ++w) sum += stg_all[(size_t)w * 32 * kStgStride + n * kStgStride + m];
                    if (32 * ob + m < p.dout && 32 * hb + n < p.hid) P2[(size_t)(32 * ob + m) * p.hid + 32 * hb + n] = sum;
                }
            }
    }
    // ---- combine the waves' column partials through LDS; one row per workgroup ----
    if (p.vec_partials != nullptr) {
        __syncthreads();
        float* red = stg_all;   // NWV x 4 x 64 floats (fits: 8 x 32 x 36 staging)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float v1 = k < HB ? cget(kDb1 + k) : 0.f;
            const float v2 = k < OB ? cget(kDb2 + k) : 0.f;
            const float v3 = k < OB ? cget(kDg + k) : 0.f;
            const float v4 = k < OB ? cget(kDbt + k) : 0.f;
            // lane (c, half): add the two row halves, keep in lanes < 32 -> column 32 * k + c
            const float s1 = v1 + __shfl_xor(v1, 32, 64), s2 = v2 + __shfl_xor(v2, 32, 64);
            const float s3 = v3 + __shfl_xor(v3, 32, 64), s4 = v4 + __shfl_xor(v4, 32, 64);
            if (lane < 32) {
                red[(wave * 4 + 0) * 64 + 32 * k + lane] = s1;
                red[(wave * 4 + 1) * 64 + 32 * k + lane] = s2;
                red[(wave * 4 + 2) * 64 + 32 * k + lane] = s3;
                red[(wave * 4 + 3) * 64 + 32 * k + lane] = s4;
            }
        }
        __syncthreads();
        constexpr int kVecRows = LW ? 7 : 4;
        if (wave < 4) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) s += red[(w * 4 + wave) * 64 + lane];
            p.vec_partials[((size_t)wg_id * kVecRows + wave) * p.vec_stride + lane] = s;
        }
        if constexpr (LW) {   // dW1[:, 0..2]: the same reduction with the weighted column sums
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 2; ++k) {
#pragma unroll
                for (int c3 = 0; c3 < 3; ++c3) {
                    const float v = k < HB ? dw1c[k < HB ? k : 0][c3] : 0.f;
                    const float sv = v + __shfl_xor(v, 32, 64);
                    if (lane < 32) red[(wave * 4 + c3) * 64 + 32 * k + lane] = sv;
                }
            }
            __syncthreads();
            if (wave < 3) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < NWV; ++w) s += red[(w * 4 + wave) * 64 + lane];
                p.vec_partials[((size_t)wg_id * kVecRows + 4 + wave) * p.vec_stride + lane] = s;
            }
        }
    }
}

template <int HB, int OB, int NS, bool RO = false>
__global__ __launch_bounds__(kBlockThreads) void mlp_bwd_fast_kernel(const nlam_mlp_bwd_t p) {
    mlp_bwd_fast_body<HB, OB, NS, false, kWavesPerBlock, RO>(p, xcd_slot((int)blockIdx.x, (int)gridDim.x), (int)gridDim.x);
}

struct bwd_group_t {
    nlam_mlp_bwd_t g[NLAM_MAX_GROUP];
    int n;
    int first[NLAM_MAX_GROUP + 1];
};

// The fused-weight-gradient variant carries 64 + accumulator registers through the tile loop: with two waves per SIMD (256
// registers each) it spilled 428 bytes per lane and ran 217 us against 129 us for the plain grouped kernel; its per-tile work
// is VALU-bound (LayerNorm backward + operand splits), so it runs ONE wave per SIMD (4-wave workgroups, up to 512 registers).
constexpr int kLeafWaves = 4;
template <int HB, int OB, int NS, bool LW>
__global__ __launch_bounds__(LW ? kLeafWaves * 64 : kBlockThreads) void mlp_bwd_fast_group_kernel(const bwd_group_t G) {
    int gi = 0;
#pragma unroll
    for (int k = 1; k < NLAM_MAX_GROUP; ++k)
        if (k < G.n && (int)blockIdx.x >= G.first[k]) gi = k;
    mlp_bwd_fast_body<HB, OB, NS, LW, LW ? kLeafWaves : kWavesPerBlock>(G.g[gi], (int)blockIdx.x - G.first[gi], G.first[gi + 1] - G.first[gi]);
}

// ---------------------------------------------------------------------------
// weight gradients:  C[m][n] = sum_rows A[row][m] * B[row][n]
//   A = dz (contiguous), B = gathered concat of sources (optionally SiLU'd)
// One workgroup (4 waves) = one partial; each wave owns up to NBW 32x32 blocks.
// ---------------------------------------------------------------------------
constexpr int kWgradThreads = 256;
constexpr int kWgradRows = 32;
constexpr int kWgTile = kWgradRows * 64;  // floats in one [32 rows][64 cols] LDS tile


// Fast path: m <= 64, every source width <= 64, all multiples of 4.
// LDS holds, per buffer, one [32][64] tile for A and one per source, filled by
// LDS-DMA (global_load_lds_dwordx4: no staging registers, 1 KiB = 4 rows per
// wave-instruction, per-lane gathered source address, lane-linear destination).
// Two buffers: chunk k+1 streams in while chunk k feeds the MFMAs; one barrier
// per chunk.  MFMA operands are k-major b32 reads of the row-major tiles
// (consecutive lanes = consecutive columns: conflict-free, the two half-waves read
// adjacent rows).
template <int NBW, bool SILU>
__global__ __launch_bounds__(kWgradThreads) void wgrad_dma_kernel(const nlam_wgrad_t p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ntile = 1 + p.nsrc;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, hi = lane >> 5;
    const int MB = (p.m + 31) >> 5;

    // this wave's output blocks: q -> (mb, source tile, 32-column block inside the source)
    int blk_mb[NBW], blk_tile[NBW], blk_nb[NBW], blk_col0[NBW], blk_w[NBW];
    {
        int nb_total = 0;
        for (int s = 0; s < p.nsrc; ++s) nb_total += (p.src[s].width + 31) >> 5;
#pragma unroll
        for (int q = 0; q < NBW; ++q) {
            const int blk = wave + 4 * q;
            blk_mb[q] = -1;
            blk_tile[q] = 1;
            blk_nb[q] = 0;
            blk_col0[q] = 0;
            blk_w[q] = 0;
            if (blk < MB * nb_total) {
                blk_mb[q] = blk / nb_total;
                int nbg = blk % nb_total, off = 0;
                for (int s = 0; s < p.nsrc; ++s) {
                    const int nbs = (p.src[s].width + 31) >> 5;
                    if (nbg < nbs) {
                        blk_tile[q] = 1 + s;
                        blk_nb[q] = nbg;
                        blk_col0[q] = off + nbg * 32;
                        blk_w[q] = off + p.src[s].width;
                        break;
                    }
                    nbg -= nbs;
                    off += p.src[s].width;
                }
            }
        }
    }

    f32x16 acc[NBW];
#pragma unroll
    for (int q = 0; q < NBW; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    const int chunks_per_batch = (p.rows + kWgradRows - 1) / kWgradRows;
    const long total_chunks = (long)chunks_per_batch * p.batch;
    const int lrow = lane >> 4, lcol = (lane & 15) << 2;  // lane -> (row within the 4-row group, first column)

    // gather indices of a chunk (one per fetched row and source); loaded one chunk ahead of the DMA that uses
    // them, so the index -> row dependent-load chain never sits in front of the MFMAs
    auto load_idx = [&](long ch, int(&ix)[2][NLAM_MAX_SRC]) {
        const int r0 = (int)(ch % chunks_per_batch) * kWgradRows;
        const int nr = min(kWgradRows, p.rows - r0);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int r = wave * 8 + g * 4 + lrow;
            const int prow = r0 + (r < nr ? r : 0);
#pragma unroll
            for (int s = 0; s < NLAM_MAX_SRC; ++s) {
                ix[g][s] = prow;
                if (s < p.nsrc && r < nr && p.src[s].idx != nullptr) ix[g][s] = p.src[s].idx[prow];
            }
        }
    };
    auto issue = [&](long ch, int buf, const int(&ix)[2][NLAM_MAX_SRC]) {
        const int b = (int)(ch / chunks_per_batch);
        const int r0 = (int)(ch % chunks_per_batch) * kWgradRows;
        const int nr = min(kWgradRows, p.rows - r0);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int r = wave * 8 + g * 4 + lrow;   // tile row this lane fetches
            const bool rv = r < nr;
            const long prow = r0 + (rv ? r : 0);
            // tile 0: A
            {
                const float* src = (rv && lcol < p.m) ? p.A + ((size_t)b * p.rows + prow) * p.m + lcol : g_zero16;
                float* dst = smem + ((size_t)(buf * ntile) * kWgTile) + (wave * 8 + g * 4) * 64;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < NLAM_MAX_SRC; ++s) {
                if (s >= p.nsrc) continue;
                const nlam_src_t S = p.src[s];
                const float* src = g_zero16;
                if (rv && lcol < S.width) src = S.ptr + (long)b * S.bstride + (long)ix[g][s] * S.width + lcol;
                float* dst = smem + ((size_t)(buf * ntile + 1 + s) * kWgTile) + (wave * 8 + g * 4) * 64;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        }
    };

    long ch = xcd_slot((int)blockIdx.x, (int)gridDim.x);   // chunk c -> slot c % g: the rows an XCD gathers stay in an eighth of the tables
    int buf = 0;
    int ix[2][NLAM_MAX_SRC];
    if (ch < total_chunks) {
        load_idx(ch, ix);
        // ONE wait for the first chunk's gather indices, and the registers re-defined by the (empty) asm: with the index loads still
        // pending in the compiler's books, every address computation in issue() got an s_waitcnt vmcnt(0) of its own -- behind an
        // LDS-DMA it cannot count past, so each of the chunk's eight DMAs waited for the one before (seen in the ISA, round 6:
        // eight dependent round trips at the head of a launch of ~14 chunks per workgroup).  The loop's chunks never had the problem.
        static_assert(NLAM_MAX_SRC == 3, "the asm below lists ix[2][3]");
        asm volatile("" : "+v"(ix[0][0]), "+v"(ix[0][1]), "+v"(ix[0][2]), "+v"(ix[1][0]), "+v"(ix[1][1]), "+v"(ix[1][2]));
        issue(ch, 0, ix);
    }
    if (ch + gridDim.x < total_chunks) load_idx(ch + gridDim.x, ix);
    for (; ch < total_chunks; ch += gridDim.x) {
        __syncthreads();  // drains this wave's LDS-DMA (vmcnt(0)); everyone is done reading buf ^ 1
        if (ch + gridDim.x < total_chunks) issue(ch + gridDim.x, buf ^ 1, ix);
        if (ch + 2 * (long)gridDim.x < total_chunks) load_idx(ch + 2 * (long)gridDim.x, ix);
        const float* At = smem + (size_t)(buf * ntile) * kWgTile;
#pragma unroll
        for (int q = 0; q < NBW; ++q) {
            if (blk_mb[q] >= 0) {
                const float* Bt = smem + (size_t)(buf * ntile + blk_tile[q]) * kWgTile;
                const int ac = blk_mb[q] * 32 + i, bc = blk_nb[q] * 32 + i;
                // all operands of the chunk first (one lgkmcnt wait), then 16 back-to-back MFMAs;
                // SiLU is a compile-time variant: as a run-time flag it was evaluated (and discarded) for every MFMA
                float a[kWgradRows / 2], bv[kWgradRows / 2];
#pragma unroll
                for (int ks = 0; ks < kWgradRows / 2; ++ks) {
                    a[ks] = At[(2 * ks + hi) * 64 + ac];
                    bv[ks] = Bt[(2 * ks + hi) * 64 + bc];
                }
                if constexpr (SILU) {
#pragma unroll
                    for (int ks = 0; ks < kWgradRows / 2; ++ks) bv[ks] = silu_f(bv[ks]);
                }
#pragma unroll
                for (int ks = 0; ks < kWgradRows / 2; ++ks) acc[q] = MFMA32(a[ks], bv[ks], acc[q]);
            }
        }
        buf ^= 1;
    }
    // ---- write this workgroup's partial (m x n) ----
    float* P = p.partials + (size_t)blockIdx.x * p.m * p.n;
#pragma unroll
    for (int q = 0; q < NBW; ++q) {
        if (blk_mb[q] >= 0) {
            const int n = blk_col0[q] + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = blk_mb[q] * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m < p.m && n < blk_w[q]) P[(size_t)m * p.n + n] = acc[q][r];
            }
        }
    }
}

// Generic widths: element-wise staging (small / odd shapes only).
template <int NBW>
__global__ __launch_bounds__(kWgradThreads) void wgrad_kernel(const nlam_wgrad_t p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int MP = round_up(p.m, 32), NP = round_up(p.n, 32);
    const int SA = MP + 4, SB = NP + 4;
    float* As = smem;                    // kWgradRows x SA
    float* Bs = As + kWgradRows * SA;    // kWgradRows x SB
    const int NB = NP / 32;
    const int nblocks = (MP / 32) * NB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, hi = lane >> 5;

    f32x16 acc[NBW];
#pragma unroll
    for (int q = 0; q < NBW; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    for (int s = threadIdx.x; s < kWgradRows * (SA + SB); s += blockDim.x) smem[s] = 0.f;  // padding stays zero
    __syncthreads();

    const int chunks_per_batch = (p.rows + kWgradRows - 1) / kWgradRows;
    const long total_chunks = (long)chunks_per_batch * p.batch;
    for (long ch = blockIdx.x; ch < total_chunks; ch += gridDim.x) {
        const int b = (int)(ch / chunks_per_batch);
        const int r0 = (int)(ch % chunks_per_batch) * kWgradRows;
        const int nr = min(kWgradRows, p.rows - r0);
        for (int s = threadIdx.x; s < kWgradRows * p.m; s += blockDim.x) {
            const int r = s / p.m, c = s % p.m;
            As[r * SA + c] = (r < nr) ? p.A[((size_t)b * p.rows + r0 + r) * p.m + c] : 0.f;
        }
        int off = 0;
        for (int q = 0; q < p.nsrc; ++q) {
            const nlam_src_t S = p.src[q];
            const int w = S.width;
            for (int s = threadIdx.x; s < kWgradRows * w; s += blockDim.x) {
                const int r = s / w, c = s % w;
                float v = 0.f;
                if (r < nr) {
                    const long prow = r0 + r;
                    const long ridx = S.idx != nullptr ? S.idx[prow] : prow;
                    v = S.ptr[(long)b * S.bstride + ridx * w + c];
                    if (p.flags & NLAM_F_SILU_B) v = silu_f(v);
                }
                Bs[r * SB + off + c] = v;
            }
            off += w;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NBW; ++q) {
            const int blk = wave + 4 * q + 4 * NBW * blockIdx.y;
            if (blk < nblocks) {
                const int mb = blk / NB, nb = blk % NB;
#pragma unroll
                for (int ks = 0; ks < kWgradRows / 2; ++ks) {
                    const float a = As[(2 * ks + hi) * SA + mb * 32 + i];
                    const float bv = Bs[(2 * ks + hi) * SB + nb * 32 + i];
                    acc[q] = MFMA32(a, bv, acc[q]);
                }
            }
        }
        __syncthreads();
    }
    float* P = p.partials + (size_t)blockIdx.x * p.m * p.n;
#pragma unroll
    for (int q = 0; q < NBW; ++q) {
        const int blk = wave + 4 * q + 4 * NBW * blockIdx.y;
        if (blk < nblocks) {
            const int mb = blk / NB, nb = blk % NB;
            const int n = nb * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m < p.m && n < p.n) P[(size_t)m * p.n + n] = acc[q][r];
            }
        }
    }
}

// Weight gradient against a very narrow B (n <= 8 columns, one source: the 2- / 3-feature inputs of
// the mesh / edge embedders): no MFMA tile would be more than 1/4 full, so this is a streaming
// reduction: lane = column m of A, every wave walks rows, B's few values are broadcast loads.
// HBM-bound (reads A once).  partials layout as the MFMA kernels: (nparts, m, n).
constexpr int kSmallN = 8;
// m % 4 == 0: lane -> (row of the pass, 4 columns of A): one coalesced 16-B load per lane covers 64 / (m/4) rows
__global__ __launch_bounds__(256) void wgrad_smalln_kernel(const nlam_wgrad_t p) {
    __shared__ float red[256 * 4 * kSmallN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const nlam_src_t S = p.src[0];
    const int n = S.width;
    const long total_rows = (long)p.batch * p.rows;
    const long per = (total_rows + gridDim.x - 1) / gridDim.x;
    const long r_begin = (long)blockIdx.x * per, r_end = min(total_rows, r_begin + per);
    const bool flat_b = p.batch == 1 || (S.idx == nullptr && S.bstride == (long)p.rows * n);
    for (int m0 = 0; m0 < p.m; m0 += 256) {
        const int cols = min(256, p.m - m0);
        const int vpr = cols >> 2;                 // float4 per row in this column pass (<= 64)
        const int rpp = 64 / vpr;                  // rows per wave-instruction
        const int rsub = lane / vpr, c4 = lane - rsub * vpr;
        const bool lane_ok = rsub < rpp;
        float acc[4][kSmallN];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int k = 0; k < kSmallN; ++k) acc[c][k] = 0.f;
        constexpr int kRB = 4;                     // passes in flight per wave
        for (long g0 = r_begin + (long)wave * rpp * kRB; g0 < r_end; g0 += 4L * rpp * kRB) {
            f32x4 a[kRB];
            float xv[kRB][kSmallN];
#pragma unroll
            for (int u = 0; u < kRB; ++u) {
                const long gr = g0 + (long)u * rpp + rsub;
                const bool ok = lane_ok && gr < r_end;
                const long grc = ok ? gr : r_begin;
                a[u] = ok ? *reinterpret_cast<const f32x4*>(p.A + grc * p.m + m0 + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
                const float* xr;
                if (flat_b && S.idx == nullptr) {
                    xr = S.ptr + grc * n;
                } else {
                    const int b = (int)(grc / p.rows);
                    const int row = (int)(grc - (long)b * p.rows);
                    const long ridx = S.idx != nullptr ? S.idx[row] : row;
                    xr = S.ptr + (long)b * S.bstride + ridx * n;
                }
#pragma unroll
                for (int k = 0; k < kSmallN; ++k) xv[u][k] = k < n ? xr[k] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < kRB; ++u)
#pragma unroll
                for (int k = 0; k < kSmallN; ++k)
                    if (k < n) {
                        float x = xv[u][k];
                        if (p.flags & NLAM_F_SILU_B) x = silu_f(x);
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[c][k] += a[u][c] * x;
                    }
        }
        // combine the 4 waves x rpp row groups: thread -> (column, k)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int k = 0; k < kSmallN; ++k) red[(tid * 4 + c) * kSmallN + k] = lane_ok ? acc[c][k] : 0.f;
        __syncthreads();
        for (int o = tid; o < cols * n; o += 256) {
            const int col = o / n, k = o - col * n;
            const int cc4 = col >> 2, c = col & 3;
            float sum = 0.f;
            for (int w = 0; w < 4; ++w)
                for (int rs = 0; rs < rpp; ++rs) sum += red[((w * 64 + rs * vpr + cc4) * 4 + c) * kSmallN + k];
            p.partials[(size_t)blockIdx.x * p.m * p.n + (size_t)(m0 + col) * p.n + k] = sum;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// small HBM-bound kernels
// ---------------------------------------------------------------------------
// buf[b, dst[s], :] = sum over the pieces src[ptr[s] .. ptr[s+1]) of buf[b, piece, :], pieces in list order.  The second pass of
// the deterministic reduction of receivers that are cut over several tiles (graph.build_tile_schedule, "virtual" split): their
// pieces reduce into virtual rows behind the real ones with plain stores, this sums them up -- a handful of rows per layer.
__global__ void split_combine_kernel(float* buf, long bstride, const int32_t* ptr, const int32_t* src, const int32_t* dst, int width) {
    const int s = blockIdx.x;
    float* base = buf + (long)blockIdx.y * bstride;
    const int q0 = ptr[s], q1 = ptr[s + 1];
    for (int c = threadIdx.x; c < width; c += blockDim.x) {
        float acc = 0.f;
        for (int q = q0; q < q1; ++q) acc += base[(long)src[q] * width + c];
        base[(long)dst[s] * width + c] = acc;
    }
}

__global__ void segment_sum_kernel(const float* in, long in_bstride, const int32_t* ptr, const int32_t* order,
                                   const float* scale, float* out, int nseg, int width, int batch, int accumulate, const float* extra) {
    const int w4 = (width + 3) >> 2;
    const long total = (long)batch * nseg * w4;
    for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(gid % w4);
        const long rest = gid / w4;
        const int sgm = (int)(rest % nseg);
        const int b = (int)(rest / nseg);
        const int lo = ptr[sgm], hi_ = ptr[sgm + 1];
        const float* base = in + (long)b * in_bstride;
        float* o = out + ((size_t)b * nseg + sgm) * width + 4 * c4;
        const float sc = scale != nullptr ? scale[sgm] : 1.f;
        if ((width & 3) == 0) {
            // 8 gathered rows in flight per thread (segments of the sender view are ~40 edges long: a plain
            // dependent loop is latency-bound); fixed summation order
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            // the addends that do not depend on the sum are requested first (behind the loop each was a round trip of its own)
            f32x4 ex = {0.f, 0.f, 0.f, 0.f}, prev = {0.f, 0.f, 0.f, 0.f};
            if (extra != nullptr) ex = *reinterpret_cast<const f32x4*>(extra + ((size_t)b * nseg + sgm) * width + 4 * c4);
            if (accumulate) prev = *reinterpret_cast<const f32x4*>(o);   // every output row has exactly one writer
            int q = lo;
            for (; q + 8 <= hi_; q += 8) {
                // all eight row ids first, then all eight rows: with id and row loads interleaved the wait for id u + 1 (vmcnt
                // counts in order) also waited for row u -- sixteen dependent round trips per batch (seen in the ISA, round 6)
                long rw[8];
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) rw[u] = order != nullptr ? order[q + u] : q + u;
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(base + rw[u] * width + 4 * c4);
                acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            }
            if (q < hi_) {
                // the last 1 .. 7 rows: all in flight together (clamped index), added one by one in row order -- the sums of the
                // row-by-row loop this replaces (mesh segments are ~9 rows long: that loop was 1 .. 7 dependent round trips)
                long rw[8];
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 7; ++u) {
                    const int qq = min(q + u, hi_ - 1);
                    rw[u] = order != nullptr ? order[qq] : qq;
                }
#pragma unroll
                for (int u = 0; u < 7; ++u) v[u] = *reinterpret_cast<const f32x4*>(base + rw[u] * width + 4 * c4);
                asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]));
#pragma unroll
                for (int u = 0; u < 7; ++u)
                    if (q + u < hi_) acc += v[u];
            }
            acc = acc * sc;
            if (extra != nullptr) acc += ex;
            if (accumulate) acc += prev;
            *reinterpret_cast<f32x4*>(o) = acc;
        } else {
            for (int c = 0; c < 4 && 4 * c4 + c < width; ++c) {
                float acc = 0.f;
                for (int q = lo; q < hi_; ++q) {
                    const long row = order != nullptr ? order[q] : q;
                    acc += base[row * width + 4 * c4 + c];
                }
                float v = acc * sc;
                if (extra != nullptr) v += extra[((size_t)b * nseg + sgm) * width + 4 * c4 + c];
                o[c] = accumulate ? o[c] + v : v;
            }
        }
    }
}

// Long segments (the sender view of the mesh -> grid edges: 6 561 segments of ~39 rows): with one thread per
// (segment, float4 column) the launch is 105 k threads -- six waves per CU walking ~40 dependent-ish rows each -- and ran
// 46 us for 65 MB.  Here S groups of width / 4 lanes share a segment: group g sums rows lo + g, lo + g + S, ... (each
// group still reads whole 256-byte rows), the groups are combined with cross-lane shuffles in a fixed order.
template <int S>
__global__ __launch_bounds__(256) void segment_sum_split_kernel(const float* in, long in_bstride, const int32_t* ptr, const int32_t* order,
                                                                const float* scale, float* out, int nseg, int width, int batch,
                                                                int accumulate, const float* extra) {
    const int w4 = width >> 2;            // lanes per group; S * w4 divides 64
    const long total = (long)batch * nseg * S * w4;
    const long nthr = (long)gridDim.x * blockDim.x;
    for (long gid0 = (long)blockIdx.x * blockDim.x; gid0 < total; gid0 += nthr) {   // whole waves stay together for the shuffles
        const long gid = gid0 + threadIdx.x;
        const bool live = gid < total;
        const long g = live ? gid : total - 1;
        const int c4 = (int)(g % w4);
        const int part = (int)((g / w4) % S);
        const long rest = g / ((long)w4 * S);
        const int sgm = (int)(rest % nseg);
        const int b = (int)(rest / nseg);
        const int lo = ptr[sgm], hi_ = ptr[sgm + 1];
        const float* base = in + (long)b * in_bstride + 4 * c4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int q = lo + part;
        for (; q + 7 * S < hi_; q += 8 * S) {
            long rw[8];   // ids first, then rows (see segment_sum_kernel)
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) rw[u] = order != nullptr ? order[q + u * S] : q + u * S;
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(base + rw[u] * width);
            acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        if (q < hi_) {   // the group's last 1 .. 7 rows: in flight together, added in row order (see segment_sum_kernel)
            long rw[8];
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                const int qq = q + u * S < hi_ ? q + u * S : q;
                rw[u] = order != nullptr ? order[qq] : qq;
            }
#pragma unroll
            for (int u = 0; u < 7; ++u) v[u] = *reinterpret_cast<const f32x4*>(base + rw[u] * width);
            asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]));
#pragma unroll
            for (int u = 0; u < 7; ++u)
                if (q + u * S < hi_) acc += v[u];
        }
#pragma unroll
        for (int k = S / 2; k >= 1; k >>= 1) {   // fixed combination order: (0 + 2) + (1 + 3) for S = 4
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] += __shfl_xor(acc[c], k * w4, 64);
        }
        if (live && part == 0) {
            const float sc = scale != nullptr ? scale[sgm] : 1.f;
            float* o = out + ((size_t)b * nseg + sgm) * width + 4 * c4;
            acc = acc * sc;
            if (extra != nullptr) acc += *reinterpret_cast<const f32x4*>(extra + ((size_t)b * nseg + sgm) * width + 4 * c4);
            if (accumulate) acc += *reinterpret_cast<const f32x4*>(o);
            *reinterpret_cast<f32x4*>(o) = acc;
        }
    }
}

// out[idx] (+)= sum_p partials[p][idx]: lane -> idx, the block's 4 waves split the parts,
// fixed summation order (deterministic)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* partials, int nparts, long stride, int n,
                                                              float* out, int accumulate) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = blockIdx.x * 64; base < n; base += gridDim.x * 64) {
        const int idx = base + lane;
        float s = 0.f;
        if (idx < n) {
            const int per = (nparts + 3) / 4;
            const int q0 = wave * per, q1 = min(nparts, q0 + per);
#pragma unroll 8
            for (int q = q0; q < q1; ++q) s += partials[(size_t)q * stride + idx];
        }
        red[wave][lane] = s;
        __syncthreads();
        if (wave == 0 && idx < n) {
            const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
            out[idx] = accumulate ? out[idx] + t : t;
        }
        __syncthreads();
    }
}

// several partial-sum reductions in one launch (blockIdx.y = job): the weight and vector
// gradients of one fused-MLP backward, optionally accumulated straight into the flat gradient buffer.
// The partials of one step add up to hundreds of MB at cfg2 (up to 512 x 48 KB per edge MLP) and every launch is a
// latency chain of dependent loads: a workgroup is 16 waves, wave w sums its slice of the parts with up to eight 16-byte
// loads per lane in flight, the slices are combined through LDS in wave order (fixed summation order: deterministic).
#ifndef NLAM_RED_WAVES
#define NLAM_RED_WAVES 4   // waves per reduce_jobs workgroup (16 up to round 6: A/B builds)
#endif
// (round 6: 4 waves instead of 16.  A 1 024-thread workgroup needs a CU with sixteen free wave slots -- beside chain kernels that
// fill every CU it waits for one to drain, and with 32 parts each of its waves had two dependent loads; a 4-wave workgroup fits
// into any free slot, has its eight parts in flight at once, and all 2 048 workgroups of a 512 x 512 pair are resident together.)
constexpr int kRedWaves = NLAM_RED_WAVES;
__global__ __launch_bounds__(kRedWaves * 64) void reduce_jobs_kernel(const nlam_reduce_jobs_t jobs) {
    __shared__ f32x4 red[kRedWaves][64];
    if ((int)blockIdx.y >= jobs.njobs) return;
    const nlam_reduce_job_t jb = jobs.job[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = (jb.nparts + kRedWaves - 1) / kRedWaves;
    const int q0 = min(jb.nparts, wave * per), q1 = min(jb.nparts, q0 + per);
    const bool vec = ((jb.n | (int)(jb.stride & 3) | jb.ncols | jb.ld) & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(jb.partials) | reinterpret_cast<uintptr_t>(jb.out)) & 15) == 0;
    if (vec) {
        for (int base = blockIdx.x * 256; base < jb.n; base += gridDim.x * 256) {
            const int idx = base + 4 * lane;
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, prev = {0.f, 0.f, 0.f, 0.f};
            float* o = jb.out + (jb.ncols > 0 ? (size_t)(idx / jb.ncols) * jb.ld + idx % jb.ncols : (size_t)idx);   // ncols % 4 == 0 here
            if (idx < jb.n) {
                // the value accumulated onto is requested with the parts (behind the reduction it was a round trip of its own)
                if (wave == 0 && jb.accumulate) prev = *reinterpret_cast<const f32x4*>(o);
                const float* pp = jb.partials + idx;
                int q = q0;
                for (; q + 8 <= q1; q += 8) {
                    f32x4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(pp + (size_t)(q + u) * jb.stride);
                    s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                }
                if (q < q1) {   // the last 1 .. 7 parts: in flight together, added in part order
                    f32x4 v[8];
#pragma unroll
                    for (int u = 0; u < 7; ++u) v[u] = *reinterpret_cast<const f32x4*>(pp + (size_t)min(q + u, q1 - 1) * jb.stride);
                    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]));
#pragma unroll
                    for (int u = 0; u < 7; ++u)
                        if (q + u < q1) s += v[u];
                }
            }
            red[wave][lane] = s;
            __syncthreads();
            if (wave == 0 && idx < jb.n) {
                f32x4 t = red[0][lane];
#pragma unroll
                for (int w = 1; w < kRedWaves; ++w) t += red[w][lane];
                if (jb.accumulate) t += prev;
                *reinterpret_cast<f32x4*>(o) = t;
            }
            __syncthreads();
        }
        return;
    }
    float* redf = reinterpret_cast<float*>(&red[0][0]);
    for (int base = blockIdx.x * 64; base < jb.n; base += gridDim.x * 64) {
        const int idx = base + lane;
        float s = 0.f;
        if (idx < jb.n) {
#pragma unroll 8
            for (int q = q0; q < q1; ++q) s += jb.partials[(size_t)q * jb.stride + idx];
        }
        redf[wave * 64 + lane] = s;
        __syncthreads();
        if (wave == 0 && idx < jb.n) {
            float t = redf[lane];
#pragma unroll
            for (int w = 1; w < kRedWaves; ++w) t += redf[w * 64 + lane];
            float* o = jb.out + (jb.ncols > 0 ? (size_t)(idx / jb.ncols) * jb.ld + idx % jb.ncols : (size_t)idx);
            *o = jb.accumulate ? *o + t : t;
        }
        __syncthreads();
    }
}

// element e of a (.., nodes, width) tensor -> (row e / width, node row % nodes); 32-bit divisions whenever the tensor has
// fewer than 2^31 elements (a 64-bit division is ~100 instructions: per element it kept these passes at ~1 TB/s)
__device__ __forceinline__ void row_and_node(long e, int width, int nodes, bool small, long& r, int& n) {
    if (small) {
        const unsigned r32 = (unsigned)e / (unsigned)width;
        r = r32;
        n = (int)(r32 % (unsigned)nodes);
    } else {
        r = e / width;
        n = (int)(r % nodes);
    }
}

// masked, weighted MSE (metrics.wmse + mask_and_reduce_metric + the batch / time means of
// training_step) as one HBM-bound pass: partial[block] = sum rw[row % nodes] * inv_var[v] * (pred - target)^2
__global__ __launch_bounds__(256) void wmse_fwd_kernel(const float* pred, const float* target, const float* inv_var,
                                                       const float* row_weight, long total, int nodes, int nvars, float scale,
                                                       float* partials) {
    __shared__ float red[4];
    float s = 0.f;
    const bool small = total < (1L << 31);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        long row;
        int n;
        row_and_node(e, nvars, nodes, small, row, n);
        const int v = (int)(e - row * nvars);
        const float w = row_weight[n];
        if (w != 0.f) {
            const float d = pred[e] - target[e];
            s += w * inv_var[v] * d * d;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
}

// out = a[node] * x + c[node] * (y + z * s[var] + m[var]); every term optional (see nlam_affine_mix)
__global__ void affine_mix_kernel(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ y,
                                  const float* __restrict__ c, const float* __restrict__ z, const float* __restrict__ s,
                                  const float* __restrict__ m, float* __restrict__ out, long total, int nodes, int width) {
    const bool small = total < (1L << 31);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        long r;
        int n;
        row_and_node(e, width, nodes, small, r, n);
        const int f = (int)(e - r * width);
        float inner = 0.f;
        if (y != nullptr) inner += y[e];
        if (z != nullptr) inner += z[e] * s[f];
        if (m != nullptr) inner += m[f];
        if (c != nullptr) inner *= c[n];
        if (x != nullptr) inner += a[n] * x[e];
        out[e] = inner;
    }
}

// on_after_batch_transfer: (x - mean[c / rep]) / std[c / rep] for up to four tensors in one launch (blockIdx.y = job)
__global__ void standardize_kernel(const nlam_std_jobs_t jobs) {
    if ((int)blockIdx.y >= jobs.njobs) return;
    const nlam_std_job_t jb = jobs.job[blockIdx.y];
    const long total = jb.rows * jb.width;
    const bool small = total < (1L << 31);   // 32-bit modulo (see step_tail_fwd_kernel)
    if ((total & 3) == 0 && small && ((reinterpret_cast<uintptr_t>(jb.x) | reinterpret_cast<uintptr_t>(jb.out)) & 15) == 0) {
        // four consecutive elements per thread (16-byte accesses), one modulo per four, the column counter advances
        const unsigned nq = (unsigned)(total >> 2), nthr = gridDim.x * blockDim.x;
        for (unsigned q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += nthr) {
            const unsigned e = 4 * q;
            int c = (int)(e % (unsigned)jb.width);
            const f32x4 x = *reinterpret_cast<const f32x4*>(jb.x + e);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int f = c / jb.rep;
                o[k] = __fdiv_rn(__fsub_rn(x[k], jb.mean[f]), jb.std[f]);
                if (++c == jb.width) c = 0;
            }
            *reinterpret_cast<f32x4*>(jb.out + e) = o;
        }
        return;
    }
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = small ? (int)((unsigned)e % (unsigned)jb.width) : (int)(e % jb.width);
        const int f = c / jb.rep;
        jb.out[e] = __fdiv_rn(__fsub_rn(jb.x[e], jb.mean[f]), jb.std[f]);
    }
}

// one AR step's elementwise tail (nlam_step_tail_fwd / _bwd): state update, boundary overwrite and the masked weighted
// MSE partial sums in one pass over the (rows, width) state; the backward in one more
__global__ __launch_bounds__(256) void step_tail_fwd_kernel(const float* __restrict__ delta, const float* __restrict__ prev,
                                                            const float* __restrict__ truth, const float* __restrict__ target,
                                                            const float* __restrict__ dstd, const float* __restrict__ dmean,
                                                            const float* __restrict__ bmask, const float* __restrict__ inv_var,
                                                            const float* __restrict__ row_weight, float scale, float* __restrict__ pred,
                                                            float* __restrict__ partials, long total, int nodes, int width) {
    __shared__ float red[4];
    float s = 0.f;
    const bool small = total < (1L << 31);
    // four consecutive elements per thread as 16-byte accesses: one (row, node) division per four elements, the variable /
    // node counters advance incrementally (the scalar loop below is the general path)
    const bool vec = (total & 3) == 0 && small &&
                     ((reinterpret_cast<uintptr_t>(delta) | reinterpret_cast<uintptr_t>(prev) | reinterpret_cast<uintptr_t>(truth) |
                       reinterpret_cast<uintptr_t>(target) | reinterpret_cast<uintptr_t>(pred)) & 15) == 0;
    if (vec) {
        const unsigned nq = (unsigned)(total >> 2), nthr = gridDim.x * blockDim.x;
        for (unsigned q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += nthr) {
            const unsigned e = 4 * q;
            unsigned r = e / (unsigned)width;
            int f = (int)(e - r * (unsigned)width);
            int n = (int)(r % (unsigned)nodes);
            const f32x4 dl = *reinterpret_cast<const f32x4*>(delta + e), pr = *reinterpret_cast<const f32x4*>(prev + e);
            const f32x4 tr = *reinterpret_cast<const f32x4*>(truth + e), tg = *reinterpret_cast<const f32x4*>(target + e);
            f32x4 pv4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float nw = pr[c] + (dstd != nullptr ? dl[c] * dstd[f] : dl[c]);
                if (dmean != nullptr) nw += dmean[f];
                const float bm = bmask[n];
                const float pv = bm * tr[c] + (1.f - bm) * nw;
                pv4[c] = pv;
                const float w = row_weight[n];
                if (w != 0.f) {
                    const float d = pv - tg[c];
                    s += w * inv_var[f] * d * d;
                }
                if (++f == width) {
                    f = 0;
                    if (++n == nodes) n = 0;
                }
            }
            *reinterpret_cast<f32x4*>(pred + e) = pv4;
        }
    }
    for (long e0 = vec ? total : (long)blockIdx.x * blockDim.x; e0 < total; e0 += (long)gridDim.x * blockDim.x) {
        const long e = e0 + threadIdx.x;
        if (e < total) {
            long r;
            int n;
            row_and_node(e, width, nodes, small, r, n);
            const int f = (int)(e - r * width);
            float nw = prev[e] + (dstd != nullptr ? delta[e] * dstd[f] : delta[e]);
            if (dmean != nullptr) nw += dmean[f];
            const float bm = bmask[n];
            const float pv = bm * truth[e] + (1.f - bm) * nw;
            pred[e] = pv;
            const float w = row_weight[n];
            if (w != 0.f) {
                const float d = pv - target[e];
                s += w * inv_var[f] * d * d;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
}

__global__ void step_tail_bwd_kernel(const float* __restrict__ g_pred, const float* __restrict__ gloss, const float* __restrict__ pred,
                                     const float* __restrict__ target, const float* __restrict__ dstd, const float* __restrict__ bmask,
                                     const float* __restrict__ inv_var, const float* __restrict__ row_weight, float scale,
                                     float* __restrict__ d_delta, float* __restrict__ d_prev, long total, int nodes, int width) {
    const float g2 = 2.f * scale * gloss[0];
    const bool small = total < (1L << 31);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        long r;
        int n;
        row_and_node(e, width, nodes, small, r, n);
        const int f = (int)(e - r * width);
        const float w = row_weight[n];
        float G = g_pred != nullptr ? g_pred[e] : 0.f;
        if (w != 0.f) G += g2 * w * inv_var[f] * (pred[e] - target[e]);
        G *= 1.f - bmask[n];
        if (d_prev != nullptr) d_prev[e] = G;
        if (d_delta != nullptr) d_delta[e] = dstd != nullptr ? G * dstd[f] : G;
    }
}

// row-wise concatenation (nlam_concat): a workgroup owns 64 consecutive rows; every source's 64 x w_k block is one
// contiguous span in memory (read coalesced into the LDS row image), and so is the 64 x wtot output block
constexpr int kCatRows = 64;
// nlam_window_batch: blockIdx.z = sample of the batch, blockIdx.y = row: 0 .. 1 + ar_steps -> one (nodes x d_state) block
// of the state series (a contiguous copy), then ar_steps rows of windowed forcing (a (window x d_forcing) ->
// (d_forcing x window) transpose per node: consecutive lanes write consecutive floats, their reads walk `window`
// contiguous streams).  Four consecutive floats per thread and 32-bit index arithmetic: one division per four
// elements, the feature / window counters advance incrementally (64-bit divisions per element made the first version
// run at a tenth of the HBM rate).
__global__ __launch_bounds__(256) void window_batch_kernel(const nlam_window_t p) {
    const int b = blockIdx.z;
    const long i = p.sample_idx[b];
    const int past = p.num_past_forcing_steps, fut = p.num_future_forcing_steps;
    const long last = p.n_times - 1;
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x, nthr = gridDim.x * blockDim.x;
    const int nstate_rows = 2 + p.ar_steps;
    if ((int)blockIdx.y < nstate_rows) {
        const int r = blockIdx.y;
        const unsigned per = (unsigned)p.nodes * (unsigned)p.d_state;
        const long t = min(max(i + max(0, past - 2) + r, 0L), last);
        const float* src = p.state + t * (long)per;
        float* dst = r < 2 ? p.init_states + ((long)b * 2 + r) * per : p.target_states + ((long)b * p.ar_steps + (r - 2)) * per;
        const bool stdz = p.state_mean != nullptr;
        const unsigned d = p.d_state;
        const bool vec = (per & 3u) == 0 && ((reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(dst)) & 15) == 0;
        for (unsigned q = tid; q < (per + 3) / 4; q += nthr) {
            const unsigned e0 = 4 * q;
            float v[4];
            if (vec) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(src + e0);
                v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = e0 + k < per ? src[e0 + k] : 0.f;
            }
            if (stdz) {
                unsigned f = e0 % d;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[k] = __fdiv_rn(__fsub_rn(v[k], p.state_mean[f]), p.state_std[f]);
                    f = f + 1 == d ? 0 : f + 1;
                }
            }
            if (vec) {
                *reinterpret_cast<f32x4*>(dst + e0) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (e0 + k < per) dst[e0 + k] = v[k];
            }
        }
        if (r == 2 && p.target_times != nullptr && blockIdx.x == 0 && (int)threadIdx.x < p.ar_steps) {
            // without time stamps the (clamped, like the data) time INDEX of every target step is reported
            const long tt = min(max(i + max(2, past) + (long)threadIdx.x, 0L), last);
            p.target_times[(long)b * p.ar_steps + threadIdx.x] = p.times != nullptr ? p.times[tt] : tt;
        }
    } else {
        if (p.d_forcing == 0) return;
        const int step = (int)blockIdx.y - nstate_rows;
        const unsigned W = past + fut + 1, df = p.d_forcing;
        const unsigned fw = df * W;
        const unsigned per_out = (unsigned)p.nodes * fw;
        const long per_in = (long)p.nodes * df;
        const long t0 = i + max(2, past) + step - past;   // time of window slot 0
        float* dst = p.forcing_windowed + ((long)b * p.ar_steps + step) * per_out;
        const bool stdz = p.forcing_mean != nullptr;
        const bool vec = (per_out & 3u) == 0 && (reinterpret_cast<size_t>(dst) & 15) == 0;
        for (unsigned q = tid; q < (per_out + 3) / 4; q += nthr) {
            const unsigned e0 = 4 * q;
            unsigned n = e0 / fw;
            unsigned jj = e0 - n * fw;
            unsigned fi = jj / W, w = jj - fi * W;
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float x = 0.f;
                if (e0 + k < per_out) {
                    const long t = min(max(t0 + (long)w, 0L), last);
                    x = p.forcing[t * per_in + (long)n * df + fi];
                    if (stdz) x = __fdiv_rn(__fsub_rn(x, p.forcing_mean[fi]), p.forcing_std[fi]);
                }
                v[k] = x;
                if (++w == W) {
                    w = 0;
                    if (++fi == df) {
                        fi = 0;
                        ++n;
                    }
                }
            }
            if (vec) {
                *reinterpret_cast<f32x4*>(dst + e0) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (e0 + k < per_out) dst[e0 + k] = v[k];
            }
        }
    }
}

__global__ __launch_bounds__(256) void concat_kernel(const nlam_cat_t p, int wtot) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const long ntile = ((long)p.nodes + kCatRows - 1) / kCatRows;
    for (long tb = blockIdx.x; tb < ntile * p.batch; tb += gridDim.x) {
        const int b = (int)(tb / ntile);
        const long n0 = (tb % ntile) * kCatRows;
        const int nr = (int)min((long)kCatRows, p.nodes - n0);
        int off = 0;
#pragma unroll
        for (int k = 0; k < NLAM_MAX_CAT; ++k) {
            if (k < p.nsrc) {
                const int w = p.width[k];
                const float* src = p.ptr[k] + (long)b * p.bstride[k] + n0 * w;
                for (int e = threadIdx.x; e < nr * w; e += blockDim.x) {
                    const int r = e / w;
                    smem[r * wtot + off + (e - r * w)] = src[e];
                }
                off += w;
            }
        }
        __syncthreads();
        float* dst = p.out + ((long)b * p.nodes + n0) * wtot;
        for (int e = threadIdx.x; e < nr * wtot; e += blockDim.x) dst[e] = smem[e];
        __syncthreads();
    }
}

__global__ void wmse_bwd_kernel(const float* pred, const float* target, const float* inv_var, const float* row_weight,
                                const float* gscalar, long total, int nodes, int nvars, float scale, float* dpred) {
    const float g = 2.f * scale * gscalar[0];
    const bool small = total < (1L << 31);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        long row;
        int n;
        row_and_node(e, nvars, nodes, small, row, n);
        const int v = (int)(e - row * nvars);
        const float w = row_weight[n];
        dpred[e] = w != 0.f ? g * w * inv_var[v] * (pred[e] - target[e]) : 0.f;
    }
}

// resident step counter (nlam_adamw_step_resident): one thread advances it and leaves the two bias corrections for the update
// kernel behind it -- nothing about the step count is a launch argument, so the pair can live inside a captured HIP graph
__global__ void adamw_prep_kernel(int32_t* step_count, float* bias_corr, float b1, float b2) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int t = *step_count + 1;
        *step_count = t;
        bias_corr[0] = 1.f - powf(b1, (float)t);
        bias_corr[1] = sqrtf(1.f - powf(b2, (float)t));
    }
}

__global__ void adamw_kernel(float* param, const float* grad, float* m, float* v, long n, float lr, float b1, float b2,
                             float eps, float wd, float bc1, float bc2_sqrt, float gscale, const float* bias_corr) {
    if (bias_corr != nullptr) {
        bc1 = bias_corr[0];
        bc2_sqrt = bias_corr[1];
    }
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        const float g = grad[idx] * gscale;
        float pv = param[idx];
        pv *= (1.f - lr * wd);                       // torch.optim.AdamW: decoupled decay first
        const float mi = b1 * m[idx] + (1.f - b1) * g;
        const float vi = b2 * v[idx] + (1.f - b2) * g * g;
        m[idx] = mi;
        v[idx] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        param[idx] = pv - (lr / bc1) * (mi / denom);
    }
}

// ---------------------------------------------------------------------------
// node-level products of the factorised edge MLP (nlam_linear):
//   out[r][h] (+)= sum_c x[r][c] * W[h * ldn + c * ldk],  k = 32 * KB, n = 32 * MB
// One wave = one 32-row tile; W staged once per persistent workgroup as split-bf16 A fragments (the GEMM1 machinery of
// mlp_fwd_bf_kernel: rows are the N side of the MFMA, output features the M side).
// ---------------------------------------------------------------------------
template <int KB, int MB, int NS>
__global__ __launch_bounds__(kFwdThreads) void linear_bf_kernel(const nlam_linear_t p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int S = 2 * KB;
    u32x4* Ws = reinterpret_cast<u32x4*>(smem);                              // [NS][MB][S][64] x 16 B
    float* stg_all = reinterpret_cast<float*>(Ws + (size_t)NS * MB * S * 64);   // kFwdWaves x 32 x kStgStride
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, hi = lane >> 5;
    float* stg = stg_all + (size_t)wave * 32 * kStgStride;
    const long ntiles = (p.rows + 31) / 32;
    const int nwaves = blockDim.x >> 6;
    long gt = (long)wave * gridDim.x + blockIdx.x;
    const long stride = (long)gridDim.x * nwaves;

    auto load_rows = [&](long t, f32x4(&xu)[KB][4]) {
        const long r = min(t * 32 + j, p.rows - 1);   // clamped: rows past the end are discarded at the store
        const float* row = p.x + r * p.k + 8 * hi;
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            xu[u][0] = *reinterpret_cast<const f32x4*>(row + 32 * u);
            xu[u][1] = *reinterpret_cast<const f32x4*>(row + 32 * u + 4);
            xu[u][2] = *reinterpret_cast<const f32x4*>(row + 32 * u + 16);
            xu[u][3] = *reinterpret_cast<const f32x4*>(row + 32 * u + 20);
        }
    };
    f32x4 xu[KB][4];
    if (gt < ntiles) load_rows(gt, xu);
    stage_split<NS>(Ws, S, 0, p.W, p.ldn, p.n, MB, p.k, false, p.ldk);
    __syncthreads();

    for (; gt < ntiles; gt += stride) {
        f32x16 acc[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
#pragma unroll
        for (int u = 0; u < KB; ++u)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float xs8[8] = {xu[u][2 * h][0], xu[u][2 * h][1], xu[u][2 * h][2], xu[u][2 * h][3],
                                      xu[u][2 * h + 1][0], xu[u][2 * h + 1][1], xu[u][2 * h + 1][2], xu[u][2 * h + 1][3]};
                const BfFrag<NS> B = split8<NS>(xs8);
                mma_split_lds<NS, MB>(acc, Ws, MB, S, 2 * u + h, lane, B);
            }
        const long t0 = gt;
        if (gt + stride < ntiles) load_rows(gt + stride, xu);   // next tile's rows ahead of this tile's stores
        const int nrows = (int)min((long)32, p.rows - t0 * 32);
        float* obase = p.out + t0 * 32 * p.n;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) *reinterpret_cast<f32x4*>(&stg[j * kStgStride + 8 * tt + 4 * hi]) = acc_chunk(acc[mb], tt);
            wave_lds_sync();
            for (int base = 0; base < nrows * 8; base += 64) {   // lane -> (row, float4 column): whole 128-B lines per row
                const int item = base + lane;
                if (item < nrows * 8) {
                    const int r = item >> 3, c4 = item & 7;
                    f32x4 v = *reinterpret_cast<const f32x4*>(&stg[r * kStgStride + 4 * c4]);
                    float* d = obase + (size_t)r * p.n + 32 * mb + 4 * c4;
                    if (p.accumulate) v += *reinterpret_cast<const f32x4*>(d);
                    *reinterpret_cast<f32x4*>(d) = v;
                }
            }
            wave_lds_sync();
        }
    }
}

// Widths above 64 (k, n multiples of 64, up to 512): W no longer fits LDS as split fragments, so the product is walked
// in 64 x 64 weight blocks.  blockIdx.y = pair of 32-column output blocks (with a second weight matrix / output, W2 and
// out2, the pairs n/64 .. 2n/64 - 1 belong to it: both node-level products of a layer whose senders are its receivers
// in one launch); the workgroup streams that pair's K dimension in 64-column chunks, the next chunk's weight block
// staged (split) into the other LDS buffer while this one is multiplied.  One wave = one 32-row tile per round.
constexpr int kLinResidentChunks = 4;   // 64-column K chunks of one output pair that stay in LDS (K <= 256)
constexpr unsigned kLinResidentFlag = 1u << 30;   // launcher -> kernel, not part of the C-ABI
template <int NS>
__global__ __launch_bounds__(kFwdThreads) void linear_bfw_kernel(const nlam_linear_t p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int S = 4;                                           // K = 16 steps per 64-column chunk
    constexpr size_t kWb = (size_t)NS * 2 * S * 64;                // u32x4 per weight buffer: [NS][2 blocks][S][64]
    const bool resident = (p.flags & kLinResidentFlag) != 0;   // set by the launcher (K <= 64 * kLinResidentChunks)
    u32x4* Ws = reinterpret_cast<u32x4*>(smem);                    // two buffers, or all K / 64 of them (resident)
    float* stg_all = reinterpret_cast<float*>(Ws + (size_t)(resident ? p.k / 64 : 2) * kWb);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, hi = lane >> 5;
    float* stg = stg_all + (size_t)wave * 32 * kStgStride;
    const long ntiles = (p.rows + 31) / 32;
    const int nwaves = blockDim.x >> 6;
    const int NP = p.n / 64, KC = p.k / 64;
    int np = blockIdx.y;
    const float* Wm = p.W;
    float* outm = p.out;
    if (np >= NP) {
        np -= NP;
        Wm = p.W2;
        outm = p.out2;
    }
    const float* Wp = Wm + (long)(64 * np) * p.ldn;
    const long rounds = (ntiles + (long)gridDim.x * nwaves - 1) / ((long)gridDim.x * nwaves);
    // K <= 256: all of this output pair's weight blocks fit LDS (4 x 24 KB as three bf16 terms) -- staged ONCE per
    // workgroup, no barrier inside the K loop and none between rounds.  (Streaming them two at a time cost a staging
    // latency + barrier per 64 columns of K and re-staged the whole strip every round: a 6 561-row product of 1.7 GFLOP
    // took 50 us with 5 % of it on the matrix cores.)
    if (resident) {
        for (int kc = 0; kc < KC; ++kc) stage_split<NS>(Ws + (size_t)kc * kWb, S, 0, Wp + (long)(64 * kc) * p.ldk, p.ldn, 64, 2, 64, false, p.ldk);
        __syncthreads();
    }
    for (long rd = 0; rd < rounds; ++rd) {
        const long t = (rd * nwaves + wave) * gridDim.x + blockIdx.x;
        const bool live = t < ntiles;
        const long r = live ? min(t * 32 + j, p.rows - 1) : 0;
        const float* xrow = p.x + r * p.k + 8 * hi;
        const int nrows = live ? (int)min((long)32, p.rows - t * 32) : 0;
        float* obase = outm + t * 32 * p.n + 64 * np;
        f32x4 xu[2][4];
        auto load_x = [&](int kc) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float* xc = xrow + 64 * kc + 32 * u;
                xu[u][0] = *reinterpret_cast<const f32x4*>(xc);
                xu[u][1] = *reinterpret_cast<const f32x4*>(xc + 4);
                xu[u][2] = *reinterpret_cast<const f32x4*>(xc + 16);
                xu[u][3] = *reinterpret_cast<const f32x4*>(xc + 20);
            }
        };
        load_x(0);
        if (!resident) {
            stage_split<NS>(Ws, S, 0, Wp, p.ldn, 64, 2, 64, false, p.ldk);
            __syncthreads();
        }
        f32x16 acc[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mb][q] = 0.f;
        for (int kc = 0; kc < KC; ++kc) {
            if (!resident && kc + 1 < KC)
                stage_split<NS>(Ws + (size_t)((kc + 1) & 1) * kWb, S, 0, Wp + (long)(64 * (kc + 1)) * p.ldk, p.ldn, 64, 2, 64, false, p.ldk);
            const u32x4* Wb = Ws + (size_t)(resident ? kc : (kc & 1)) * kWb;
            BfFrag<NS> B[4];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float xs8[8] = {xu[u][2 * h][0], xu[u][2 * h][1], xu[u][2 * h][2], xu[u][2 * h][3],
                                          xu[u][2 * h + 1][0], xu[u][2 * h + 1][1], xu[u][2 * h + 1][2], xu[u][2 * h + 1][3]};
                    B[2 * u + h] = split8<NS>(xs8);
                }
            if (kc + 1 < KC) load_x(kc + 1);   // the next chunk's rows are in flight during this chunk's MFMAs
#pragma unroll
            for (int st = 0; st < 4; ++st) mma_split_lds<NS, 2>(acc, Wb, 2, S, st, lane, B[st]);
            if (!resident) __syncthreads();   // the next block is staged; everyone is done with this one
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) *reinterpret_cast<f32x4*>(&stg[j * kStgStride + 8 * tt + 4 * hi]) = acc_chunk(acc[mb], tt);
            wave_lds_sync();
            for (int base = 0; base < nrows * 8; base += 64) {
                const int item = base + lane;
                if (item < nrows * 8) {
                    const int rr = item >> 3, c4 = item & 7;
                    f32x4 v = *reinterpret_cast<const f32x4*>(&stg[rr * kStgStride + 4 * c4]);
                    float* d = obase + (size_t)rr * p.n + 32 * mb + 4 * c4;
                    if (p.accumulate) v += *reinterpret_cast<const f32x4*>(d);
                    *reinterpret_cast<f32x4*>(d) = v;
                }
            }
            wave_lds_sync();
        }
    }
}

// ---------------------------------------------------------------------------
// nlam_mlp_pack: the LDS weight images of the narrow split-bf16 kernels, written ONCE per optimizer step.
// Every chain workgroup used to load the fp32 weights and split them into bf16 terms itself (~95 us per cfg2 step in total,
// on the critical path of ~30 latency-bound launches); with an image it issues a handful of LDS-DMA instructions instead.
//   forward image : [W1s | W2s]                         (mlp_fwd_bf_body's layout: sources padded to 32-column units)
//   backward image: [W2^T | W1_0^T | W1_1^T | W1_2^T]   (mlp_bwd_fast_body's layout; a W1 piece per source of width % 32 == 0)
// grid = (blocks, jobs); the job table lives in device memory (built once: parameter and image addresses are stable).
// ---------------------------------------------------------------------------
struct PackShape {
    int ns, HB, OB, ngemm, S1, ok;
    size_t fwd_floats, bwd_floats;
};

__host__ __device__ inline PackShape pack_shape(const nlam_pack_job_t& j) {
    PackShape r = {0, 0, 0, 0, 0, 0, 0, 0};
    r.ns = (int)((j.flags & NLAM_F_MM_MASK) >> NLAM_F_MM_SHIFT);
    if (r.ns < 1 || r.ns > 3 || j.nsrc < 1 || j.nsrc > NLAM_MAX_SRC || j.hid < 1 || j.dout < 1) return r;
    if (j.hid > kMaxWidth || j.dout > kMaxWidth || (j.hid & 31) != 0) return r;
    r.HB = j.hid >> 5;
    r.OB = (j.dout + 31) >> 5;
    const bool pre = (j.flags & NLAM_F_PRE_ADD) != 0;
    r.ngemm = pre ? 1 : j.nsrc;
    // (fixed trip counts and static indices: a dynamically indexed copy of the job makes the device compiler keep the struct
    // in LDS and read it back element by element -- mlp_pack_kernel, round 6)
    int nunits = 0;
#pragma unroll
    for (int s = 0; s < NLAM_MAX_SRC; ++s) {
        if (s >= j.nsrc) continue;
        if (j.width[s] < 1 || j.width[s] > kMaxWidth) return r;
        if (s < r.ngemm) nunits += (j.width[s] + 31) >> 5;
    }
    r.S1 = 2 * nunits;
    const int DPH = r.HB * 32, OP = r.OB * 32;
    r.fwd_floats = ((size_t)r.ns * r.HB * r.S1 * 64 + (size_t)r.ns * r.OB * (DPH / 16) * 64) * 4;
    r.bwd_floats = (size_t)r.ns * DPH * OP / 2;
#pragma unroll
    for (int s = 0; s < NLAM_MAX_SRC; ++s)
        if (s < r.ngemm && (j.width[s] & 31) == 0) r.bwd_floats += (size_t)r.ns * j.width[s] * DPH / 2;
    r.ok = 1;
    return r;
}

// One lane item of stage_split_impl in two halves, so that a thread can have the loads of ALL pieces of a job in flight before
// the first store: on gfx950 a load issued behind a store waits for that store's acknowledgement, and the piece-by-piece
// version (load, convert, store, next piece) was ten such round trips deep -- 15 us at the head of every cfg2 step.
struct PackItem {
    float x[8];
    int slot;      // destination index (in u32x4 units, term 0), -1 = this thread has no item in the piece
    int tstride;   // u32x4 units between bf16 terms
};

__device__ __forceinline__ PackItem pack_item_load(int S, int s0, const float* W, long ldm, int M, int MB, int K, bool perm2, long ldk,
                                                   int Kpad, int idx) {
    PackItem it;
    const int nst = (Kpad > 0 ? Kpad : K) >> 4;
    it.slot = -1;
    it.tstride = MB * S * 64;
#pragma unroll
    for (int q = 0; q < 8; ++q) it.x[q] = 0.f;
    if (idx >= MB * nst * 64) return it;
    const int lane = idx & 63;
    const int rest = idx >> 6;
    const int st = rest % nst, mb = rest / nst;
    const int i = lane & 31, hi = lane >> 5;
    const int m = mb * 32 + i;
    const int kA = perm2 ? 32 * (st >> 1) + 16 * (st & 1) + 4 * hi : 16 * st + 8 * hi;
    const int kB = perm2 ? kA + 8 : kA + 4;
    it.slot = (mb * S + s0 + st) * 64 + lane;
    const float* rowp = W + (long)m * ldm;
    if (ldk == 1 && m < M && kB + 4 <= K && ((ldm & 3) == 0) && ((reinterpret_cast<uintptr_t>(W) & 15) == 0)) {
        const f32x4 va = *reinterpret_cast<const f32x4*>(rowp + kA);
        const f32x4 vb = *reinterpret_cast<const f32x4*>(rowp + kB);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            it.x[c] = va[c];
            it.x[4 + c] = vb[c];
        }
    } else {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = min((q < 4 ? kA : kB) + (q & 3), K - 1);   // clamped: unconditional loads
            v[q] = W[(long)min(m, M - 1) * ldm + (long)k * ldk];
        }
        // all eight in flight before the first select: left alone the compiler sinks each load into a branch of its own select
        // and waits for it there -- eight dependent round trips per item (seen in the ISA; ~10 us of a 13 us one-job launch)
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
        for (int q = 0; q < 8; ++q) it.x[q] = (m < M && (q < 4 ? kA : kB) + (q & 3) < K) ? v[q] : 0.f;
    }
    return it;
}

template <int NS>
__device__ __forceinline__ void pack_item_store(u32x4* dst, const PackItem& it) {
    if (it.slot < 0) return;
    const BfFrag<NS> f = split8<NS>(it.x);
#pragma unroll
    for (int p = 0; p < NS; ++p) dst[(size_t)p * it.tstride + it.slot] = f.t[p];
}

// ---------------------------------------------------------------------------
// nlam_linear, widths above 64: an LDS-tiled GEMM (round 5; replaces linear_bfw_kernel wherever n % 128 == 0).
//   out[r][h] (+)= sum_c x[r][c] A[h][c],   A[h][c] = W[h ldn + c ldk]
// linear_bfw_kernel gave one wave a 32-row x 64-feature strip with its rows in registers and ONLY the weights in LDS: every
// MFMA fetched its own 1-KiB A fragment (LDS 128 B/clk/CU at one term: half the array's rate for 8 % of the matrix peak,
// 163 us for 63 784 x 512 x 512), x was re-read once per 64 output columns, and a launch of 6 561 rows was a latency chain
// of 8 K chunks behind one staging barrier each.  Here:
//  * a workgroup of 2 x 2 waves owns a tile of BN = 128 features x BM = 64 WN rows; wave (wm, wn) owns 2 feature blocks x WN
//    row blocks of 32 x 32 (the weights stay the A operand and the rows the N side, as everywhere in this file);
//  * BOTH operands go through LDS as split-bf16 fragments in fragment order (pack_item_load / pack_item_store: the routine
//    that writes the weight images), so each A fragment feeds WN MFMAs per product and each B fragment two, and every fetch is
//    a linear ds_read_b128;
//  * K runs in chunks of 32 columns through two buffers: the next chunk's fp32 pieces are requested into registers AHEAD of
//    this chunk's MFMAs and split + stored behind them; one barrier per chunk;
//  * x is read once per 128 output columns, and the workgroups that share a row tile are dealt to ONE XCD (ids 8 apart), so
//    the second to fourth of them find it in that XCD's L2;
//  * WN = 2 (128 x 128 tiles) for the grid-level products (63 784 rows: HBM-bound on fp32 in / out), WN = 1 (64 rows) for the
//    mesh-level ones (6 561 rows: 412 workgroups instead of 208 strips).
// Rounding: the same split products in the same K order per 32-column chunk as the kernels it replaces; the fp32 sums are
// grouped per chunk (two K steps), so results agree with linear_bfw_kernel to fp32 rounding, not bit for bit.
// ---------------------------------------------------------------------------
constexpr int kLinGemmThreads = 256;
template <int NS, int WN, int SK = 2>
__host__ __device__ constexpr size_t lin_gemm_lds_bytes() {
    return ((size_t)2 * NS * 4 * SK * 64 + (size_t)2 * NS * (2 * WN) * SK * 64) * 16 + (size_t)4 * 32 * kStgStride * sizeof(float);
}

// TA: the weight operand is read transposed (ldn == 1: the data-gradient product, A[h][c] = W[h + c ldk]: eight dword loads per
// fragment slot, each coalesced across the 32 features of a block) instead of along K (ldk == 1: two 16-byte loads).
// SK: K = 16 steps per chunk.  2 = 32-column chunks (rounds 5-6).  4 = 64-column chunks (round 6, the one-term mesh-level products:
// a 6 561-row launch is 412 workgroups whose K loop is a chain of global-memory round trips, one chunk in flight, ~30 ns of MFMA per
// chunk and wave -- 21 us for k = 512 = 16 chunks; twice the columns per round trip halves the chain and the barriers).
template <int NS, int WN, bool TA, int SK = 2>
__global__ __launch_bounds__(kLinGemmThreads) void linear_gemm_kernel(const nlam_linear_t p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int WM = 2;                       // feature blocks per wave
    constexpr int S = SK;                       // K = 16 steps per chunk of 16 S columns
    constexpr int SPR = 2 * S;                  // fragment slots (8 K values) per tile row and chunk
    constexpr int RP = kLinGemmThreads / SPR;   // tile rows per row-major staging pass
    constexpr int QA = S, QB = WN * S / 2;      // staging passes per chunk: 128 feature rows / RP (= the transposed staging's count too), 64 WN rows / RP
    static_assert(SK == 2 || SK == 4, "32- or 64-column chunks");
    constexpr int MBA = 2 * WM, MBB = 2 * WN;   // 32-row fragment blocks of the workgroup's A (features) / B (rows) tile
    constexpr int BN = 32 * MBA, BM = 32 * MBB;
    constexpr size_t kAv = (size_t)NS * MBA * S * 64, kBv = (size_t)NS * MBB * S * 64;   // u32x4 per buffer
    u32x4* As = reinterpret_cast<u32x4*>(smem);                    // [2][NS][MBA][S][64]
    u32x4* Bs = As + 2 * kAv;                                      // [2][NS][MBB][S][64]
    float* stg_all = reinterpret_cast<float*>(Bs + 2 * kBv);       // 4 x 32 x kStgStride
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int j = lane & 31, hi = lane >> 5;
    float* stg = stg_all + (size_t)wave * 32 * kStgStride;

    // workgroup id -> (row tile, feature tile): row tile = 8 * (id / (8 nftt)) + id % 8, so that the nftt workgroups of a row
    // tile have ids 8 apart = the same XCD, and are dispatched within one wave of the grid
    const int nft = p.n / BN;
    const int nftt = nft * (p.W2 != nullptr ? 2 : 1);
    const long id = blockIdx.x;
    const long rt = 8 * (id / (8 * (long)nftt)) + (id & 7);
    int ft = (int)((id >> 3) % nftt);
    const long r0 = rt * BM;
    if (r0 >= p.rows) return;   // the padding of the last group of eight row tiles
    const float* Wm = p.W;
    float* outm = p.out;
    if (ft >= nft) {
        ft -= nft;
        Wm = p.W2;
        outm = p.out2;
    }
    const int mrows = (int)min((long)BM, p.rows - r0);
    const float* Wp = Wm + (long)(ft * BN) * p.ldn;
    const float* xp = p.x + r0 * p.k;
    const int KC = p.k / (16 * S);

    // staging.  A fragment slot = 8 consecutive K values (32 bytes) of one tile row; a tile row's 32-column chunk = 4 slots =
    // one 128-byte line.  Rows read along K (x, and the weights in the forward layout): thread -> (row = tid / 4 + 64 q, slot
    // s = tid % 4), so four neighbouring lanes fetch one whole line and a wave instruction 16 lines (the first version gave a
    // lane a row and a half wave a slot: 32 bytes of each of 32 lines per instruction, a quarter of every line: 137 us for the
    // 63 784 x 512 x 512 one-term product, 1.9 TB/s).  Transposed weights (TA): thread -> (feature = lane % 32, slot from wave
    // and half wave): eight dword loads per slot, each coalesced across the 32 features of a block.
    // LDS: slot (row j, K step st, half hi) of a 32-row block lives at fragment (st, hi) position (j + 2 (2 st + hi)) % 32 --
    // the rotation spreads the four slots of a line, which one wave instruction writes, over all 32 banks (unrotated they are
    // 512 bytes apart: a 4-way conflict), and the MFMA loop's reads stay one contiguous (rotated) 16-slot window per lane group.
    // No branch anywhere: rows past the end of x read the last row and are zeroed by a select.
    const int ss = tid & (SPR - 1), srow = tid / SPR;        // row-major staging: slot (st = ss / 2, hi = ss % 2), row srow + RP q
    const int t_st = wave % S, t_mb = wave / S;              // transposed staging: K step, block t_mb + (4 / S) q, slot half hi, feature j
    auto frag_index = [&](int mb_, int st_, int hi_, int j_, int MB_) {   // u32x4 index inside one term of one buffer
        return (mb_ * S + st_) * 64 + hi_ * 32 + ((j_ + 2 * (2 * st_ + hi_)) & 31);
    };
    const float* a_src[QA];
    const float* b_src[QB];
    int a_dst[QA], b_dst[QB];
    bool b_live[QB];
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        if constexpr (TA) {
            const int m = 32 * (t_mb + (4 / S) * q) + j;
            a_src[q] = Wp + m + (long)(16 * t_st + 8 * hi) * p.ldk;
            a_dst[q] = frag_index(t_mb + (4 / S) * q, t_st, hi, j, MBA);
        } else {
            const int m = srow + RP * q;
            a_src[q] = Wp + (long)m * p.ldn + 8 * ss;
            a_dst[q] = frag_index(m >> 5, ss >> 1, ss & 1, m & 31, MBA);
        }
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const int r = srow + RP * q;
        b_live[q] = r < mrows;
        b_src[q] = xp + (long)min(r, mrows - 1) * p.k + 8 * ss;
        b_dst[q] = frag_index(r >> 5, ss >> 1, ss & 1, r & 31, MBB);
    }
    float ia[QA][8], ib[QB][8];
    auto request = [&](int kc) {
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            if constexpr (TA) {
                const float* src = a_src[q] + (long)(16 * S * kc) * p.ldk;
#pragma unroll
                for (int e = 0; e < 8; ++e) ia[q][e] = src[(long)e * p.ldk];
            } else {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(a_src[q] + 16 * S * kc);
                const f32x4 up = *reinterpret_cast<const f32x4*>(a_src[q] + 16 * S * kc + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) ia[q][e] = lo[e], ia[q][4 + e] = up[e];
            }
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(b_src[q] + 16 * S * kc);
            const f32x4 up = *reinterpret_cast<const f32x4*>(b_src[q] + 16 * S * kc + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) ib[q][e] = lo[e], ib[q][4 + e] = up[e];
        }
    };
    auto publish = [&](int buf) {
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            const BfFrag<NS> f = split8<NS>(ia[q]);
#pragma unroll
            for (int t = 0; t < NS; ++t) As[(size_t)buf * kAv + (size_t)t * MBA * S * 64 + a_dst[q]] = f.t[t];
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = b_live[q] ? ib[q][e] : 0.f;
            const BfFrag<NS> f = split8<NS>(x);
#pragma unroll
            for (int t = 0; t < NS; ++t) Bs[(size_t)buf * kBv + (size_t)t * MBB * S * 64 + b_dst[q]] = f.t[t];
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int mb = 0; mb < WM; ++mb)
#pragma unroll
        for (int nb = 0; nb < WN; ++nb)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mb][nb][q] = 0.f;

    request(0);
    publish(0);
    __syncthreads();
    for (int kc = 0; kc < KC; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < KC) request(kc + 1);   // in flight during this chunk's MFMAs
        const u32x4* Ab = As + (size_t)buf * kAv;
        const u32x4* Bb = Bs + (size_t)buf * kBv;
#pragma unroll
        for (int st = 0; st < S; ++st) {
            // every fragment of the step in its own registers before the first MFMA (DESIGN finding 3), accumulators alternate
            u32x4 a[NS][WM], b[NS][WN];
#pragma unroll
            for (int t = 0; t < NS; ++t) {
#pragma unroll
                for (int mb = 0; mb < WM; ++mb) a[t][mb] = Ab[(size_t)t * MBA * S * 64 + frag_index(wm * WM + mb, st, hi, j, MBA)];
#pragma unroll
                for (int nb = 0; nb < WN; ++nb) b[t][nb] = Bb[(size_t)t * MBB * S * 64 + frag_index(wn * WN + nb, st, hi, j, MBB)];
            }
#pragma unroll
            for (int ord = NS - 1; ord >= 0; --ord)
#pragma unroll
                for (int pa = 0; pa <= ord; ++pa)
#pragma unroll
                    for (int nb = 0; nb < WN; ++nb)
#pragma unroll
                        for (int mb = 0; mb < WM; ++mb) acc[mb][nb] = MFMA_BF16(a[pa][mb], b[ord - pa][nb], acc[mb][nb]);
        }
        if (kc + 1 < KC) publish(buf ^ 1);   // the other buffer: everyone left it at the previous barrier
        __syncthreads();
    }

    // ---- epilogue: accumulator blocks -> whole 128-B row pieces through the wave's staging block ----
#pragma unroll
    for (int nb = 0; nb < WN; ++nb) {
        const long rb = r0 + (long)(wn * WN + nb) * 32;
        const int nrows = (int)max((long)0, min((long)32, p.rows - rb));
#pragma unroll
        for (int mb = 0; mb < WM; ++mb) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) *reinterpret_cast<f32x4*>(&stg[j * kStgStride + 8 * tt + 4 * hi]) = acc_chunk(acc[mb][nb], tt);
            wave_lds_sync();
            float* obase = outm + rb * p.n + (long)ft * BN + (wm * WM + mb) * 32;
            for (int base = 0; base < nrows * 8; base += 64) {
                const int item = base + lane;
                if (item < nrows * 8) {
                    const int rr = item >> 3, c4 = item & 7;
                    f32x4 v = *reinterpret_cast<const f32x4*>(&stg[rr * kStgStride + 4 * c4]);
                    float* d = obase + (size_t)rr * p.n + 4 * c4;
                    if (p.accumulate) v += *reinterpret_cast<const f32x4*>(d);
                    *reinterpret_cast<f32x4*>(d) = v;
                }
            }
            wave_lds_sync();
        }
    }
}

// One PIECE of a job per workgroup column (blockIdx.z: the W1 piece of each source and W2 in the forward layout, W2^T and the W1^T
// piece of each source in the backward layout), one lane item per thread, ONE call site of pack_item_load.  Rounds 3-6 gave a
// thread lane item `tid` of EVERY piece of its job, all loads before the first store: 31 KB of straight-line code of which a wave
// executes ~10 KB exactly once -- the launch runs once per optimizer step on a cold instruction cache and took 32 us at the head
// of every cfg2 step (profiles/round6/cfg2_step_timeline.txt) for ~3 us of memory round trips.  A thread still has its load in
// flight before its store (finding 22), and eight times as many waves share the latency.
constexpr int kPackPieces = 2 * NLAM_MAX_SRC + 2;

__device__ __forceinline__ void pack_piece(const nlam_pack_job_t& j, const PackShape& sh, int piece, int idx) {
    const int DPH = sh.HB * 32, OP = sh.OB * 32;
    int wd[NLAM_MAX_SRC];   // widths of the sources that have a GEMM piece (0 past them); static indices only (see pack_shape)
    int kin = 0;
#pragma unroll
    for (int s = 0; s < NLAM_MAX_SRC; ++s) {
        wd[s] = s < sh.ngemm ? j.width[s] : 0;
        kin += wd[s];
    }
    const int ldw1 = j.ldw1 > 0 ? j.ldw1 : kin;
    const size_t NSz = (size_t)sh.ns;
    // arguments of the one pack_item_load / pack_item_store below
    int S, s0 = 0, M, MB, K, Kpad = 0;
    long ldm, ldk = 1;
    bool perm2;
    const float* W;
    u32x4* dst;
    if (piece <= NLAM_MAX_SRC) {   // forward image [W1s | W2s]
        if (j.fwd_image == nullptr) return;
        u32x4* W1s = reinterpret_cast<u32x4*>(j.fwd_image);
        if (piece < NLAM_MAX_SRC) {
            if (piece >= sh.ngemm) return;
            int off = 0, w = 0;
#pragma unroll
            for (int s = 0; s < NLAM_MAX_SRC; ++s) {
                if (s < piece) {
                    off += wd[s];
                    s0 += 2 * ((wd[s] + 31) >> 5);
                }
                if (s == piece) w = wd[s];
            }
            S = sh.S1, W = j.W1 + off, ldm = ldw1, M = j.hid, MB = sh.HB, K = w, perm2 = false, Kpad = ((w + 31) >> 5) * 32;
            dst = W1s;
        } else {
            S = DPH / 16, W = j.W2, ldm = j.hid, M = j.dout, MB = sh.OB, K = j.hid, perm2 = true;
            dst = W1s + NSz * sh.HB * sh.S1 * 64;
        }
    } else {   // backward image [W2^T | W1_s^T ..]: A[m = hidden][k = out (slot-permuted)] = W2[k][m];  A[m = source column][k = hidden] = W1[k][off + m]
        if (j.bwd_image == nullptr) return;
        const int sb = piece - NLAM_MAX_SRC - 2;
        if (sb < 0) {
            S = OP / 16, W = j.W2, ldm = 1, M = j.hid, MB = sh.HB, K = j.dout, perm2 = true, ldk = j.hid, Kpad = OP;
            dst = reinterpret_cast<u32x4*>(j.bwd_image);
        } else {
            if (sb >= sh.ngemm) return;
            size_t ioff = NSz * DPH * OP / 2;
            int off = 0, w = 0;
#pragma unroll
            for (int s = 0; s < NLAM_MAX_SRC; ++s) {
                if (s < sb) {
                    if ((wd[s] & 31) == 0) ioff += NSz * wd[s] * DPH / 2;
                    off += wd[s];
                }
                if (s == sb) w = wd[s];
            }
            if ((w & 31) != 0) return;
            S = DPH / 16, W = j.W1 + off, ldm = 1, M = w, MB = w >> 5, K = j.hid, perm2 = true, ldk = ldw1;
            dst = reinterpret_cast<u32x4*>(j.bwd_image + ioff);
        }
    }
    const PackItem it = pack_item_load(S, s0, W, ldm, M, MB, K, perm2, ldk, Kpad, idx);
    if (sh.ns == 3) pack_item_store<3>(dst, it);
    else if (sh.ns == 2) pack_item_store<2>(dst, it);
    else pack_item_store<1>(dst, it);
}

__global__ __launch_bounds__(256) void mlp_pack_kernel(const nlam_pack_job_t* jobs) {
    const nlam_pack_job_t j = jobs[blockIdx.y];
    const PackShape sh = pack_shape(j);
    if (!sh.ok) return;
    // a piece has at most 2 blocks x 4 steps x 64 lanes = 512 items (widths <= 64): nlam_mlp_pack launches 512 threads per piece
    const int tid = (int)(blockIdx.x * blockDim.x + threadIdx.x), nthr = (int)(gridDim.x * blockDim.x);
    for (int idx = tid; idx < 512; idx += nthr) pack_piece(j, sh, (int)blockIdx.z, idx);
}

#include "nlam_wide.inc"
#include "nlam_wbf.inc"

// ---------------------------------------------------------------------------
// host side helpers
// ---------------------------------------------------------------------------
size_t fwd_lds_bytes(const nlam_mlp_fwd_t* p, int HB, int OB, int NS = 0) {
    const int DPH = HB * 32, OP = OB * 32;
    const int ngemm = (p->flags & NLAM_F_PRE_ADD) ? 1 : p->nsrc;   // sources whose W1 columns are staged
    size_t T1 = 0;
    for (int s = 0; s < ngemm; ++s) T1 += (p->src[s].width + 7) / 8;
    size_t wf = (size_t)DPH * 8 * T1 + (size_t)OP * DPH;   // weights, in floats
    if (NS > 0) {                                            // NS bf16 copies, sources padded to 32-column units
        size_t k1 = 0;
        for (int s = 0; s < ngemm; ++s) k1 += (size_t)((p->src[s].width + 31) / 32) * 32;
        wf = ((size_t)DPH * k1 + (size_t)OP * DPH) * NS / 2;
    }
    size_t f = wf + DPH + 3 * OP;
    f += (size_t)kFwdWaves * 32 * kStgStride;
    return f * sizeof(float);
}

size_t bwd_lds_bytes(const nlam_mlp_bwd_t* p, int HB, int OB) {
    const int DPH = HB * 32, OP = OB * 32;
    const int wmax = (DPH > OP ? DPH : OP) > 64 ? (DPH > OP ? DPH : OP) : 64;
    size_t f = (size_t)DPH * OP + OP + (size_t)kWavesPerBlock * 32 * (wmax + 4);
    for (int s = 0; s < p->nsrc; ++s)
        if (p->dmode[s] != 0) f += (size_t)((p->src[s].width + 31) / 32 * 32) * DPH;
    return f * sizeof(float);
}

size_t bwd_fast_lds_bytes(const nlam_mlp_bwd_t* p, int HB, int OB, int NS) {
    const int DPH = HB * 32, OP = OB * 32;
    size_t wf = (size_t)DPH * OP;
    const bool pre = (p->flags & NLAM_F_PRE_ADD) != 0;
    for (int s = 0; s < p->nsrc; ++s)
        if (p->dmode[s] != 0 && (!pre || s == 0)) wf += (size_t)p->src[s].width * DPH;
    if (NS > 0) wf = wf * NS / 2;
    const size_t nw = (p->flags & NLAM_F_LEAF_WGRAD) ? kLeafWaves : kWavesPerBlock;
    const size_t xs = (p->flags & NLAM_F_LEAF_WGRAD) ? nw * 32 * 4 + (size_t)DPH * 5 : 0;   // the tile's input rows + W1 / b1 (fused leaf weight gradients)
    return (wf + OP + nw * 32 * kStgStride + nw * 8 * 64 + xs) * sizeof(float);
}

constexpr size_t kMaxLds = 160 * 1024;

// Raise a kernel's dynamic-LDS limit once (and only upward): the attribute call is not a
// stream operation, so launches stay capturable in a HIP graph without repeating it.
struct LdsGrant {
    const void* fn;
    size_t bytes;
};
LdsGrant g_lds_grants[128];
int g_lds_ngrants = 0;

template <typename K>
int set_lds(K kernel, size_t bytes) {
    if (bytes > kMaxLds) return NLAM_EUNSUP;
    if (bytes <= 48 * 1024) return 0;
    const void* fn = reinterpret_cast<const void*>(kernel);
    int slot = -1;
    for (int k = 0; k < g_lds_ngrants; ++k)
        if (g_lds_grants[k].fn == fn) slot = k;
    if (slot >= 0 && g_lds_grants[slot].bytes >= bytes) return 0;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    if (slot < 0 && g_lds_ngrants < 128) slot = g_lds_ngrants++;
    if (slot >= 0) g_lds_grants[slot] = {fn, bytes};
    return 0;
}

// backward kernels: one persistent 8-wave workgroup per CU; tiles are dealt to workgroups first, then to waves
int grid_blocks(long total_tiles) {
    long need = total_tiles < 1 ? 1 : total_tiles;
    return (int)(need < kMaxGridBlocks ? need : kMaxGridBlocks);
}

// ---- wide-path dispatch helpers ----
struct WideCfg {
    int nwv, fb;   // waves per workgroup, 32-feature blocks per wave; 0 = not covered
};

WideCfg wide_cfg(int maxw) {
    if (maxw <= 128) return {4, 1};
    if (maxw <= 256) return {8, 1};
    if (maxw <= kMaxWide) return {8, 2};
    return {0, 0};
}

bool fwd_is_wide(const nlam_mlp_fwd_t* p) {
    if (p->hid > kMaxWidth || p->dout > kMaxWidth) return true;
    for (int s = 0; s < p->nsrc; ++s)
        if (p->src[s].width > kMaxWidth) return true;
    return false;
}

bool bwd_is_wide(const nlam_mlp_bwd_t* p) {
    if (p->hid > kMaxWidth || p->dout > kMaxWidth) return true;
    for (int s = 0; s < p->nsrc; ++s)
        if (p->src[s].width > kMaxWidth) return true;
    return false;
}

// narrow backward with an output width that is not a whole 32-column block on the split-bf16 fast kernel (RO instantiation):
// one output block, no LayerNorm / aggregation / residual, a split-bf16 matrix mode, FAST hidden width and source widths
bool bwd_ragged_out(const nlam_mlp_bwd_t* p) {
    if (bwd_is_wide(p) || p->dout % 32 == 0 || p->dout > 32 || p->hid % 32 != 0) return false;
    if (p->ln_w != nullptr || p->g_aggr != nullptr || p->g_out == nullptr || p->out_idx != nullptr) return false;
    if ((p->flags & (NLAM_F_ADD_SRC0 | NLAM_F_ADD_SRC1 | NLAM_F_PRE_ADD | NLAM_F_LEAF_WGRAD)) != 0) return false;
    if ((p->flags & NLAM_F_MM_MASK) == 0) return false;
    for (int s = 0; s < p->nsrc; ++s)
        if (p->dmode[s] != 0 && p->src[s].width != 32 && p->src[s].width != 64) return false;
    return true;
}

int bwd_wide_maxw(const nlam_mlp_bwd_t* p) {
    int w = p->hid > p->dout ? p->hid : p->dout;
    for (int s = 0; s < p->nsrc; ++s)
        if (p->dmode[s] != 0 && p->src[s].width > w) w = p->src[s].width;
    return w;
}

int fwd_nq1(const nlam_mlp_fwd_t* p) {
    int q = 0;
    for (int s = 0; s < p->nsrc; ++s) q += (p->src[s].width + 31) / 32;
    return q;
}

size_t fwd_wide_lds(const nlam_mlp_fwd_t* p, int nwv) {
    const int HBT = (p->hid + 31) / 32, OBT = (p->dout + 31) / 32;
    const int WS = (HBT > OBT ? HBT : OBT) * 32 + 4;
    return ((size_t)2 * 32 * kXS + (size_t)32 * WS + (size_t)2 * nwv * 32 + NLAM_MAX_SRC * 32) * sizeof(float);
}

size_t bwd_wide_lds(const nlam_mlp_bwd_t* p, int nwv) {
    int maxb = ((p->hid > p->dout ? p->hid : p->dout) + 31) / 32;
    for (int s = 0; s < p->nsrc; ++s)
        if (p->dmode[s] == 3 && (p->src[s].width + 31) / 32 > maxb) maxb = (p->src[s].width + 31) / 32;
    const int WS = maxb * 32 + 4;
    return ((size_t)2 * 32 * WS + (size_t)2 * nwv * 32) * sizeof(float);
}

int wide_grid(long total_tiles, size_t lds, int nwv) {
    int occ = (int)(kMaxLds / lds);
    const int wave_cap = 16 / nwv;          // at most 4 waves per SIMD
    if (occ > wave_cap) occ = wave_cap;
    if (occ < 1) occ = 1;
    long g = (long)nlam_detail::chain_cus * occ;
    if (total_tiles < g) g = total_tiles;
    return (int)(g < 1 ? 1 : g);
}

void launch_pack(const pack_jobs_t& jobs, hipStream_t stream) {
    long most = 0;
    for (int k = 0; k < jobs.njobs; ++k) {
        const long ng = round_up((jobs.job[k].Kw + 7) / 8, 4);
        const long tot = (long)jobs.job[k].MB * ng * 64;
        if (tot > most) most = tot;
    }
    long blocks = (most + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pack_a_kernel, dim3((int)blocks, jobs.njobs), dim3(256), 0, stream, jobs);
}

bool wgrad_is_narrow_dma(const nlam_wgrad_t* p) {
    bool dma = (p->m % 4 == 0) && p->m <= 64;
    for (int s = 0; s < p->nsrc; ++s) dma = dma && (p->src[s].width % 4 == 0) && p->src[s].width <= 64;
    return dma;
}

bool wgrad_is_wide(const nlam_wgrad_t* p) {
    if (wgrad_is_narrow_dma(p)) return false;
    bool ok = (p->m % 4 == 0);
    for (int s = 0; s < p->nsrc; ++s) ok = ok && (p->src[s].width % 4 == 0);
    return ok;
}

int wgrad_windows_of(const nlam_wgrad_t* p, int winm, int winn) {
    int nw = 0;
    for (int s = 0; s < p->nsrc; ++s) nw += (p->src[s].width + winn - 1) / winn;
    return nw * ((p->m + winm - 1) / winm);
}

int wgrad_windows(const nlam_wgrad_t* p) { return wgrad_windows_of(p, kWWin, kWWin); }

// split-bf16 weight gradient with more than 128 output rows: 256 x 256 windows (every operand row read once per 256 output
// columns: the big edge sets are HBM-bound) -- unless the problem has few rows (nlam_set_tuning NLAM_TUNE_WGRAD_BIG_MIN_ROWS):
// then the operands are L2-resident anyway and 128 x 128 windows give the same launch width with a quarter of the row slices,
// i.e. a quarter of the partial-sum traffic (128 x 512 KB written and read back for a 20 MB node-level problem at d = 256)
bool wgrad_wbf_big(const nlam_wgrad_t* p) {
    return p->m > 128 && (long)p->rows * p->batch >= (long)nlam_detail::wgrad_big_min_rows;
}

// workgroups of a grouped launch: kMaxGridBlocks dealt in proportion to the members' tiles, at least one each and never
// more than a member has tiles
void group_blocks(const long* tiles, int n, int* blocks) {
    long tot = 0;
    for (int k = 0; k < n; ++k) tot += tiles[k] < 1 ? 1 : tiles[k];
    for (int k = 0; k < n; ++k) {
        const long t = tiles[k] < 1 ? 1 : tiles[k];
        long b = tot <= kMaxGridBlocks ? t : (t * kMaxGridBlocks) / tot;
        if (b < 1) b = 1;
        if (b > t) b = t;
        blocks[k] = (int)b;
    }
}

// members of one grouped wide launch run one kernel instantiation (hid, dout, the widest gradient decide it); the kernel reads
// source counts and widths per member, so those may differ (the static-feature embedders: 2 .. 17 input columns)
bool same_fwd_shape(const nlam_mlp_fwd_t& a, const nlam_mlp_fwd_t& b) {
    if (a.hid != b.hid || a.dout != b.dout || a.ncat != b.ncat) return false;
    return (a.flags & ~NLAM_F_WPACK_READY) == (b.flags & ~NLAM_F_WPACK_READY) && (a.ln_w == nullptr) == (b.ln_w == nullptr);
}
bool same_bwd_shape(const nlam_mlp_bwd_t& a, const nlam_mlp_bwd_t& b) {
    if (a.hid != b.hid || a.dout != b.dout) return false;
    const WideCfg ca = wide_cfg(bwd_wide_maxw(&a)), cb = wide_cfg(bwd_wide_maxw(&b));
    if (ca.nwv != cb.nwv || ca.fb != cb.fb) return false;
    return (a.flags & ~NLAM_F_WPACK_READY) == (b.flags & ~NLAM_F_WPACK_READY) && (a.ln_w == nullptr) == (b.ln_w == nullptr);
}

// the same for the fp32 wide kernels: `cap` = the workgroups one launch keeps resident (wide_grid)
void wide_group_blocks(const long* tiles, int n, long cap, int* blocks) {
    long tot = 0;
    for (int k = 0; k < n; ++k) tot += tiles[k] < 1 ? 1 : tiles[k];
    for (int k = 0; k < n; ++k) {
        const long t = tiles[k] < 1 ? 1 : tiles[k];
        long b = tot <= cap ? t : (t * cap) / tot;
        if (b < 1) b = 1;
        if (b > t) b = t;
        blocks[k] = (int)b;
    }
}

// ---------------------------------------------------------------------------
// pack jobs of the wide launches (weights -> MFMA A-fragment order in `wpack`): built here so that the launchers (slices 3 / 4)
// and the pre-pack API (nlam_mlp_*_pack_records, slice 1) describe one launch's scratch identically.
// ---------------------------------------------------------------------------
long build_fwd_wbf_jobs(const nlam_mlp_fwd_t* p, int wns, packbf_jobs_t& jobs) {
    const WbfPlan pl = fwd_wbf_choose(p, wns);
    const int HBT = (p->hid + 31) / 32, OBT = (p->dout + 31) / 32;
    const int TK1 = fwd_wbf_tk1(p, pl.kg);
    const bool pre = (p->flags & NLAM_F_PRE_ADD) != 0;
    const int ngemm = pre ? 1 : p->nsrc;
    int kin = 0;
    for (int s = 0; s < ngemm; ++s) kin += p->src[s].width;
    if (pre && p->ldw1 > 0) kin = p->ldw1;   // floats between rows of W1
    u32x4* A1 = reinterpret_cast<u32x4*>(p->wpack);
    jobs.njobs = 0;
    int off = 0, g0 = 0;
    long most = 0;
    for (int s = 0; s < ngemm; ++s) {
        const int w = p->src[s].width;
        const int ng = ((w + 16 * pl.kg - 1) / (16 * pl.kg)) * pl.kg;
        jobs.job[jobs.njobs++] = {p->W1 + off, (long)kin, 1L, p->hid, HBT, w, TK1, g0, ng, 0, A1};
        off += w;
        g0 += ng;
        if ((long)HBT * ng * 64 > most) most = (long)HBT * ng * 64;
    }
    jobs.job[jobs.njobs++] = {p->W2, (long)p->hid, 1L, p->dout, OBT, p->hid, 2 * HBT, 0, 2 * HBT, 1,
                              A1 + (size_t)HBT * TK1 * wns * 64};
    if ((long)OBT * 2 * HBT * 64 > most) most = (long)OBT * 2 * HBT * 64;
    return most;
}

long build_bwd_wbf_jobs(const nlam_mlp_bwd_t* p, int wns, packbf_jobs_t& jobs) {
    const int HBT = (p->hid + 31) / 32, OBT = (p->dout + 31) / 32;
    const bool pre = (p->flags & NLAM_F_PRE_ADD) != 0;
    int kin = 0;
    for (int s = 0; s < (pre ? 1 : p->nsrc); ++s) kin += p->src[s].width;
    if (pre && p->ldw1 > 0) kin = p->ldw1;
    const int TKA = wbf_bwd_tk(OBT, wns), TKB = wbf_bwd_tk(HBT, wns);   // padded to whole A-ring revolutions: zero groups
    u32x4* base = reinterpret_cast<u32x4*>(p->wpack);
    jobs.njobs = 0;
    long most = (long)HBT * TKA * 64;
    // A[m = hidden][k = out] = W2[k][m]; K in slot order (the B fragments come straight from accumulator layout)
    jobs.job[jobs.njobs++] = {p->W2, 1L, (long)p->hid, p->hid, HBT, p->dout, TKA, 0, TKA, 1, base};
    size_t woff = (size_t)HBT * TKA * wns * 64;
    int off = 0;
    for (int s = 0; s < p->nsrc; ++s) {
        const int w = p->src[s].width;
        if (p->dmode[s] != 0 && (s == 0 || !pre)) {
            const int SB = (w + 31) / 32;
            // A[m = source column][k = hidden] = W1[k][off + m]
            jobs.job[jobs.njobs++] = {p->W1 + off, 1L, (long)kin, w, SB, p->hid, TKB, 0, TKB, 1, base + woff};
            woff += (size_t)SB * TKB * wns * 64;
            if ((long)SB * TKB * 64 > most) most = (long)SB * TKB * 64;
        }
        off += w;
    }
    return most;
}

void build_fwd_wide_jobs(const nlam_mlp_fwd_t* p, pack_jobs_t& jobs) {
    const int HBT = (p->hid + 31) / 32, OBT = (p->dout + 31) / 32, NQ1 = fwd_nq1(p);
    int kin = 0;
    for (int s = 0; s < p->nsrc; ++s) kin += p->src[s].width;
    jobs.njobs = 0;
    int off = 0, q0 = 0;
    for (int s = 0; s < p->nsrc; ++s) {
        const int w = p->src[s].width;
        jobs.job[jobs.njobs++] = {p->W1 + off, (long)kin, 1L, p->hid, HBT, w, NQ1 * 4, q0 * 4, p->wpack};
        off += w;
        q0 += (w + 31) / 32;
    }
    jobs.job[jobs.njobs++] = {p->W2, (long)p->hid, 1L, p->dout, OBT, p->hid, HBT * 4, 0, p->wpack + (size_t)HBT * NQ1 * 1024};
}

void build_bwd_wide_jobs(const nlam_mlp_bwd_t* p, pack_jobs_t& jobs) {
    const int HBT = (p->hid + 31) / 32, OBT = (p->dout + 31) / 32;
    int kin = 0;
    for (int s = 0; s < p->nsrc; ++s) kin += p->src[s].width;
    jobs.njobs = 0;
    // A[m = hidden][k = out] = W2[k][m]
    jobs.job[jobs.njobs++] = {p->W2, 1L, (long)p->hid, p->hid, HBT, p->dout, OBT * 4, 0, p->wpack};
    size_t woff = (size_t)HBT * OBT * 1024;
    int off = 0;
    for (int s = 0; s < p->nsrc; ++s) {
        const int w = p->src[s].width;
        if (p->dmode[s] != 0) {
            const int SB = (w + 31) / 32;
            // A[m = source column][k = hidden] = W1[k][off + m]
            jobs.job[jobs.njobs++] = {p->W1 + off, 1L, (long)kin, w, SB, p->hid, HBT * 4, 0, p->wpack + woff};
            woff += (size_t)SB * HBT * 1024;
        }
        off += w;
    }
}

// the same pack kernels driven by a job TABLE in device memory (one 64-byte record per job): every wide MLP of a model packed
// by ONE launch per step (nlam_pack_records) instead of a pack launch in front of each of its forward / backward launches
static_assert(sizeof(pack_job_t) <= 64 && sizeof(packbf_job_t) <= 64, "pack records are 64 bytes");

__global__ void pack_a_table_kernel(const nlam_pack_rec_t* recs) {
    const pack_job_t jb = *reinterpret_cast<const pack_job_t*>(recs + blockIdx.y);
    const int ng = round_up((jb.Kw + 7) >> 3, 4);
    const long total = (long)jb.MB * ng * 64;
    for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(s & 63);
        const long rest = s >> 6;
        const int t = (int)(rest % ng);
        const int mb = (int)(rest / ng);
        const int m = mb * 32 + (lane & 31);
        const int kb = 8 * t + 4 * (lane >> 5);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < jb.M) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (kb + c < jb.Kw) v[c] = jb.W[(long)m * jb.ldm + (long)(kb + c) * jb.ldk];
        }
        *reinterpret_cast<f32x4*>(&jb.dst[(((size_t)mb * jb.T + jb.t0 + t) * 64 + lane) * 4]) = v;
    }
}

template <int NS>
__global__ void pack_bf_table_kernel(const nlam_pack_rec_t* recs) {
    const packbf_job_t jb = *reinterpret_cast<const packbf_job_t*>(recs + blockIdx.y);
    const long total = (long)jb.MB * jb.ng * 64;
    for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(s & 63);
        const long rest = s >> 6;
        const int g = (int)(rest % jb.ng);
        const int mb = (int)(rest / jb.ng);
        const int m = mb * 32 + (lane & 31);
        const int hi = lane >> 5;
        float x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = jb.perm2 ? 32 * (g >> 1) + 16 * (g & 1) + (q & 3) + 8 * (q >> 2) + 4 * hi : 16 * g + 8 * hi + q;
            x[q] = (m < jb.M && k < jb.Kw) ? jb.W[(long)m * jb.ldm + (long)k * jb.ldk] : 0.f;
        }
        const BfFrag<NS> f = split8<NS>(x);
#pragma unroll
        for (int pa = 0; pa < NS; ++pa) jb.dst[(((size_t)mb * jb.TK + jb.g0 + g) * NS + pa) * 64 + lane] = f.t[pa];
    }
}

// mlp_bwd_edge_kernel takes the launch: the data gradients of the factorised InteractionNet edge layer of width 512 in the one-term
// mode or of width 256 in the three-term mode (the backward of what mlp_fwd_edge_kernel<1, 512, .> / <3, 256, false> compute)
bool bwd_edge_ok(const nlam_mlp_bwd_t* p) {
    if (nlam_detail::wbf_edge == 0 || !bwd_is_wide(p)) return false;
    const int wns = bwd_wbf_ns(p);
    const int d = wns == 1 ? 512 : 256;
    if (wns != 1 && !(wns == 3 && (nlam_detail::wbf_edge & 2) == 0 && !(p->flags & NLAM_F_STORE_BF16))) return false;
    if (!(p->flags & NLAM_F_PRE_ADD) || (p->flags & (NLAM_F_ADD_SRC1 | NLAM_F_NO_ACT))) return false;
    if (p->nsrc != NLAM_MAX_SRC || p->hid != d || p->dout != d) return false;
    for (int s = 0; s < NLAM_MAX_SRC; ++s)
        if (p->src[s].width != d) return false;
    if (p->ln_w == nullptr || p->g_aggr == nullptr || p->seg_of_row == nullptr || p->rowptr == nullptr || p->dz2_ld != 0) return false;
    if (p->z1 == nullptr || p->xhat == nullptr || p->rstd == nullptr) return false;
    if ((p->dmode[0] != 0 && p->dmode[0] != 1) || p->dmode[1] != 0 || (p->dmode[2] != 0 && p->dmode[2] != 3)) return false;
    if ((p->flags & NLAM_F_MEAN) && p->inv_deg == nullptr) return false;
    return (long)p->ntiles * p->batch >= 2 * 64;
}

}  // namespace

// ---------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------
#if NLAM_IN_TU(1)
int nlam_detail::wbf_min_supertiles = 192;
int nlam_detail::wbf_half = 1;              // bit 0: forward, bit 1: backward on 4-wave workgroups, two per CU (NLAM_TUNE_WBF_HALF)
int nlam_detail::wbf_v4 = 1;                // split-bf16 wide kernels: the V4 instantiations where every width is a multiple of 4 (NLAM_TUNE_WBF_V4; 0 for A/B)
int nlam_detail::lin_resident_wgs = 256;    // NLAM_LIN_WGS (experiments): one per CU
int nlam_detail::lin_gemm = 1;              // nlam_linear: LDS-tiled GEMM for n % 128 == 0 where it wins (0: the strip kernel of rounds 2-4; 2: wherever it applies)
long nlam_detail::lin_gemm_big_rows = 32768;   // 128-row tiles from here (63 784 grid nodes), 64-row tiles below (6 561 mesh nodes)
int nlam_detail::wgrad_big_min_rows = 0;    // rows from which a wide weight gradient uses 256 x 256 windows (0 = always, the round-2 behaviour)
int nlam_detail::wgrad_min_parts = 128;     // row slices a weight gradient of more than that many 32-row chunks is cut into at least
int nlam_detail::wgrad_min_parts_wide = -1; // ... when the weight matrix has more than 128 rows: -1 = as many as give 64 WORKGROUPS (nlam_set_tuning sets both to a slice count)
int nlam_detail::wbf_edge = 1;
int nlam_detail::chain_cus = kNumCUs;
int nlam_detail::wgrad_max_wgs = 128;   // round 6: 256 (one per CU) until then -- see nlam_wgrad_nparts
int nlam_detail::wgrad_ldma_var = 0;
int nlam_detail::wgrad_ldma = 3;           // bit 0: bf16-operand launches, bit 1: fp32-operand one-term launches with 256 x 256 windows (NLAM_TUNE_WGRAD_LDMA)
int nlam_detail::wgrad_chunks_per_wg = 8;   // A/B at cfg2 (tools/ab_bench.sh): 2.13 -> 2.06 ms per step against one chunk per workgroup
#endif

#if NLAM_IN_TU(1)
extern "C" {

#ifdef NLAM_TIMING
/* debug build only: read and clear the per-phase cycle counters */
int32_t nlam_debug_phase_cycles(unsigned long long* out16) {
    hipDeviceSynchronize();
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_cycles), 16 * sizeof(unsigned long long));
    unsigned long long zero[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), zero, sizeof(zero));
    return (int32_t)e;
}
#endif

int32_t nlam_abi_version(void) { return NLAM_ABI_VERSION; }
int32_t nlam_grid_waves(void) { return kMaxGridBlocks * kWavesPerBlock; }
int32_t nlam_num_blocks(int64_t total_tiles) { return grid_blocks((long)total_tiles); }
int32_t nlam_max_width(void) { return kMaxWide; }

int32_t nlam_set_tuning(int32_t key, int32_t value) {
    if (key == NLAM_TUNE_WBF_MIN_SUPERTILES) {
        if (value < 0) return NLAM_EINVAL;
        nlam_detail::wbf_min_supertiles = value;
        return 0;
    }
    if (key == NLAM_TUNE_WBF_HALF) {
        if (value < 0 || value > 7) return NLAM_EINVAL;
        nlam_detail::wbf_half = value;
        return 0;
    }
    if (key == NLAM_TUNE_LIN_WGS) {
        if (value < 0) return NLAM_EINVAL;
        nlam_detail::lin_resident_wgs = value;
        return 0;
    }
    if (key == NLAM_TUNE_WGRAD_BIG_MIN_ROWS) {
        if (value < 0) return NLAM_EINVAL;
        nlam_detail::wgrad_big_min_rows = value;
        return 0;
    }
    if (key == NLAM_TUNE_WGRAD_MIN_PARTS) {
        if (value < 1) return NLAM_EINVAL;
        nlam_detail::wgrad_min_parts = value;
        nlam_detail::wgrad_min_parts_wide = value;
        return 0;
    }
    if (key == NLAM_TUNE_WGRAD_CHUNKS) {
        if (value < 1) return NLAM_EINVAL;
        nlam_detail::wgrad_chunks_per_wg = value;
        return 0;
    }
    if (key == NLAM_TUNE_CHAIN_CUS) {
        if (value < 16 || value > 4 * kNumCUs) return NLAM_EINVAL;
        nlam_detail::chain_cus = value;
        return 0;
    }
    if (key == NLAM_TUNE_WGRAD_MAX_WGS) {
        if (value < 4 || value > 1024) return NLAM_EINVAL;
        nlam_detail::wgrad_max_wgs = value;
        return 0;
    }
    if (key == NLAM_TUNE_WBF_EDGE) {
        if (value < 0 || value > 3) return NLAM_EINVAL;
        nlam_detail::wbf_edge = value;
        return 0;
    }
    if (key == NLAM_TUNE_WGRAD_LDMA_VAR) {
        if (value < 0 || value > 3) return NLAM_EINVAL;
        nlam_detail::wgrad_ldma_var = value;
        return 0;
    }
    if (key == NLAM_TUNE_WGRAD_LDMA) {
        if (value < 0 || value > 7) return NLAM_EINVAL;
        nlam_detail::wgrad_ldma = value;
        return 0;
    }
    if (key == NLAM_TUNE_WBF_V4) {
        if (value < 0 || value > 1) return NLAM_EINVAL;
        nlam_detail::wbf_v4 = value;
        return 0;
    }
    if (key == NLAM_TUNE_LIN_GEMM) {
        if (value < 0) return NLAM_EINVAL;
        nlam_detail::lin_gemm = value >= 2 ? (value == 2 ? 2 : 1) : value;   // 0 off, 1 where it wins, 2 wherever it applies
        if (value > 2) nlam_detail::lin_gemm_big_rows = value;   // values above 2: the row count from which 128-row tiles are used
        return 0;
    }
    return NLAM_EINVAL;
}

int64_t nlam_mlp_pack_floats(const nlam_pack_job_t* job, int32_t which) {
    if (job == nullptr || job->W1 == nullptr || job->W2 == nullptr) return 0;
    const PackShape sh = pack_shape(*job);
    if (!sh.ok) return 0;
    return (int64_t)(which == 0 ? sh.fwd_floats : sh.bwd_floats);
}

int32_t nlam_mlp_pack(const nlam_pack_job_t* jobs_device, int32_t njobs, void* hip_stream) {
    NLAM_RANGE("nlam_mlp_pack");
    if (jobs_device == nullptr || njobs < 0 || njobs > 65535) return NLAM_EINVAL;
    if (njobs == 0) return 0;
    // every piece of a job (<= 8: three W1 sources + W2, forward and backward) has <= 512 lane items: thread t of the job's two
    // blocks takes item t of each piece, all loads in flight before the first store
    // (eight one-wave workgroups per job rather than two of four waves: the ~200 KB of images a job writes then drain through
    // eight CUs' store paths instead of two)
    hipLaunchKernelGGL(mlp_pack_kernel, dim3(8, njobs, kPackPieces), dim3(64), 0, (hipStream_t)hip_stream, jobs_device);
    return (int32_t)hipGetLastError();
}

int32_t nlam_mlp_fwd_pack_records(const nlam_mlp_fwd_t* p, nlam_pack_rec_t* out, int32_t cap, int32_t* kind) {
    if (p == nullptr || out == nullptr || kind == nullptr || p->W1 == nullptr || p->W2 == nullptr) return NLAM_EINVAL;
    if (!fwd_is_wide(p)) return 0;
    if (p->wpack == nullptr || p->wpack_floats < nlam_mlp_fwd_wpack_floats(p)) return NLAM_EINVAL;
    int n = 0;
    const int wns = fwd_wbf_ns(p);
    if (wns > 0) {
        packbf_jobs_t jobs;
        build_fwd_wbf_jobs(p, wns, jobs);
        if (jobs.njobs > cap) return NLAM_EINVAL;
        for (; n < jobs.njobs; ++n) {
            memset(out + n, 0, sizeof(nlam_pack_rec_t));
            memcpy(out + n, &jobs.job[n], sizeof(packbf_job_t));
        }
        *kind = wns;
    } else {
        if ((p->flags & NLAM_F_PRE_ADD) || wide_cfg(p->hid > p->dout ? p->hid : p->dout).nwv == 0) return NLAM_EUNSUP;
        pack_jobs_t jobs;
        build_fwd_wide_jobs(p, jobs);
        if (jobs.njobs > cap) return NLAM_EINVAL;
        for (; n < jobs.njobs; ++n) {
            memset(out + n, 0, sizeof(nlam_pack_rec_t));
            memcpy(out + n, &jobs.job[n], sizeof(pack_job_t));
        }
        *kind = 0;
    }
    return n;
}

int32_t nlam_mlp_bwd_pack_records(const nlam_mlp_bwd_t* p, nlam_pack_rec_t* out, int32_t cap, int32_t* kind) {
    if (p == nullptr || out == nullptr || kind == nullptr || p->W1 == nullptr || p->W2 == nullptr) return NLAM_EINVAL;
    if (!bwd_is_wide(p)) return 0;
    if (p->wpack == nullptr || p->wpack_floats < nlam_mlp_bwd_wpack_floats(p)) return NLAM_EINVAL;
    int n = 0;
    const int wns = bwd_wbf_ns(p);
    if (wns > 0) {
        packbf_jobs_t jobs;
        build_bwd_wbf_jobs(p, wns, jobs);
        if (jobs.njobs > cap) return NLAM_EINVAL;
        for (; n < jobs.njobs; ++n) {
            memset(out + n, 0, sizeof(nlam_pack_rec_t));
            memcpy(out + n, &jobs.job[n], sizeof(packbf_job_t));
        }
        *kind = wns;
    } else {
        if ((p->flags & NLAM_F_PRE_ADD) || wide_cfg(bwd_wide_maxw(p)).nwv == 0) return NLAM_EUNSUP;
        pack_jobs_t jobs;
        build_bwd_wide_jobs(p, jobs);
        if (jobs.njobs > cap) return NLAM_EINVAL;
        for (; n < jobs.njobs; ++n) {
            memset(out + n, 0, sizeof(nlam_pack_rec_t));
            memcpy(out + n, &jobs.job[n], sizeof(pack_job_t));
        }
        *kind = 0;
    }
    return n;
}

int32_t nlam_pack_records(const nlam_pack_rec_t* recs_device, int32_t n, int32_t kind, void* hip_stream) {
    NLAM_RANGE("nlam_pack_records");
    if (recs_device == nullptr || n < 0 || n > 65535 || (kind != 0 && kind != 1 && kind != 3)) return NLAM_EINVAL;
    if (n == 0) return 0;
    hipStream_t stream = (hipStream_t)hip_stream;
    // the largest job (a 512 x 1536 first layer in A-fragment order) has 16 x 96 x 64 lane items: 16 blocks of 256 threads walk it
    const dim3 grid(16, n), block(256);
    if (kind == 0) hipLaunchKernelGGL(pack_a_table_kernel, grid, block, 0, stream, recs_device);
    else if (kind == 1) hipLaunchKernelGGL(pack_bf_table_kernel<1>, grid, block, 0, stream, recs_device);
    else hipLaunchKernelGGL(pack_bf_table_kernel<3>, grid, block, 0, stream, recs_device);
    return (int32_t)hipGetLastError();
}

int64_t nlam_mlp_fwd_wpack_floats(const nlam_mlp_fwd_t* p) {
    if (p == nullptr || !fwd_is_wide(p)) return 0;
    const int ns = fwd_wbf_ns(p);
    if (ns > 0) return fwd_wbf_wpack_floats(p, ns);
    const int64_t HBT = (p->hid + 31) / 32, OBT = (p->dout + 31) / 32;
    return (HBT * fwd_nq1(p) + OBT * HBT) * 1024;
}

int64_t nlam_mlp_bwd_wpack_floats(const nlam_mlp_bwd_t* p) {
    if (p == nullptr || !bwd_is_wide(p)) return 0;
    if (bwd_wbf_ns(p) > 0) return bwd_wbf_wpack_floats(p, bwd_wbf_ns(p));
    const int64_t HBT = (p->hid + 31) / 32, OBT = (p->dout + 31) / 32;
    int64_t f = HBT * OBT * 1024;
    for (int s = 0; s < p->nsrc; ++s)
        if (p->dmode[s] != 0) f += (int64_t)((p->src[s].width + 31) / 32) * HBT * 1024;
    return f;
}

int32_t nlam_store_bf16_supported(const nlam_mlp_fwd_t* p) {
    if (p == nullptr) return 0;
    return store_bf16_ok(p) ? 1 : 0;
}

int32_t nlam_mlp_bwd_dz2_ld(const nlam_mlp_bwd_t* p) {
    if (p == nullptr) return 0;
    if (bwd_is_wide(p))   // split-bf16 wide kernel with a ragged output width: padded so that the W2 gradient has m % 4 == 0
        return (p->dout % 4 != 0 && bwd_wbf_ns(p) > 0) ? ((p->dout + 31) / 32) * 32 : 0;
    return bwd_ragged_out(p) ? ((p->dout + 31) / 32) * 32 : 0;
}

int32_t nlam_mlp_bwd_blocks(const nlam_mlp_bwd_t* p) {
    if (p == nullptr) return 0;
    const long total = (long)p->ntiles * p->batch;
    if (!bwd_is_wide(p)) return grid_blocks(total);
    if (bwd_edge_ok(p)) {
        const long ns2 = (long)((p->ntiles + 1) / 2) * p->batch;
        return (int32_t)(ns2 < nlam_detail::chain_cus ? ns2 : nlam_detail::chain_cus);
    }
    if (bwd_wbf_ns(p) > 0) {   // partial-sum rows: one per (workgroup, row group)
        const WbfBwdPlan pl = wbf_bwd_choose(bwd_wide_maxw(p));
        const long nsuper = (long)((p->ntiles + pl.nrt - 1) / pl.nrt) * p->batch;
        const long cap = pl.nw == 4 ? 2 * nlam_detail::chain_cus : nlam_detail::chain_cus;   // one 8-wave workgroup per CU, or two of 4 waves
        return (int32_t)((nsuper < cap ? (nsuper < 1 ? 1 : nsuper) : cap) * pl.rg);
    }
    const WideCfg cfg = wide_cfg(bwd_wide_maxw(p));
    if (cfg.nwv == 0) return 0;
    return wide_grid(total, bwd_wide_lds(p, cfg.nwv), cfg.nwv);
}

// which kernel family a launch described by `p` runs on: 0 = narrow, 1 = fp32 wide, 2 = split-bf16 wide super-tiles.
// Grouped launches take members of one family (0: nlam_mlp_*_group's narrow kernels, 1: the wide group kernels).
int32_t nlam_mlp_fwd_family(const nlam_mlp_fwd_t* p) {
    if (p == nullptr) return NLAM_EINVAL;
    if (!fwd_is_wide(p)) return 0;
    return fwd_wbf_ns(p) > 0 ? 2 : 1;
}

int32_t nlam_mlp_bwd_family(const nlam_mlp_bwd_t* p) {
    if (p == nullptr) return NLAM_EINVAL;
    if (!bwd_is_wide(p)) return 0;
    return bwd_wbf_ns(p) > 0 ? 2 : 1;
}

// workgroups (= rows of `vec_partials` written) per member of a grouped backward launch
int32_t nlam_mlp_bwd_group_blocks(const nlam_mlp_bwd_t* ps, int32_t n, int32_t* blocks) {
    if (ps == nullptr || blocks == nullptr || n < 1 || n > NLAM_MAX_GROUP) return NLAM_EINVAL;
    long tiles[NLAM_MAX_GROUP];
    for (int k = 0; k < n; ++k) tiles[k] = (long)ps[k].ntiles * ps[k].batch;
    if (!bwd_is_wide(&ps[0])) {
        group_blocks(tiles, n, blocks);
        return 0;
    }
    size_t lds = 0;
    const WideCfg cfg = wide_cfg(bwd_wide_maxw(&ps[0]));
    for (int k = 0; k < n; ++k) {
        const nlam_mlp_bwd_t& p = ps[k];
        const int32_t bad = nlam_detail::bwd_check(&p);
        if (bad != 0) return bad;
        if (p.rows < 1) return NLAM_EINVAL;
        if (nlam_mlp_bwd_family(&p) != 1 || !same_bwd_shape(p, ps[0])) return NLAM_EUNSUP;
        const size_t l = bwd_wide_lds(&p, cfg.nwv);
        if (l > lds) lds = l;
    }
    wide_group_blocks(tiles, n, wide_grid(1L << 40, lds, cfg.nwv), blocks);
    return 0;
}

int32_t nlam_wgrad_nparts(const nlam_wgrad_t* p) {
    if (p == nullptr) return 0;
    const long total_chunks = (long)p->batch * ((p->rows + kWgradRows - 1) / kWgradRows);
    // small problems: one 32-row chunk per workgroup (latency-bound otherwise); larger ones give every workgroup ~8 chunks
    // to stream through its double buffer -- each workgroup writes an (m x n) partial that the reduction reads back, and
    // with 512 partials per weight matrix those were ~0.7 GB of traffic per cfg2 step (reduce_jobs: 290 us of 3.7 ms of kernel time)
    const long cpw = nlam_detail::wgrad_chunks_per_wg < 1 ? 1 : nlam_detail::wgrad_chunks_per_wg;   // nlam_set_tuning
    long np = total_chunks <= 128 ? total_chunks : (total_chunks + cpw - 1) / cpw;
    // nlam_set_tuning; defaults 128, and 64 for weight matrices of more than 128 rows (d >= 256): re-measured under the segmented
    // executor (profiles/round5/ab_wgrad_knobs_segmented.log: cfg3 44.1 -> 43.3 / 43.2 -> 42.5 ms, under autocast 29.4 -> 28.7;
    // cfg5 unchanged; at d <= 128 the value 64 LOSES 1-2 %: cfg2 1.732 -> 1.745, cfg4 10.67 -> 10.86)
    // round 6 (profiles/round6/ab_wgrad_min_parts.log): what the wide floor buys is launch width, and a 512 x 512 gradient has four
    // 256 x 256 windows per slice -- 64 slices of a 6 561-row problem are 64 MB of partial sums for 27 MB of operands.  The floor is
    // 64 workgroups, not 64 slices: cfg5 112.2 -> 109.7 ms (its isolated launch 25.6 -> 34.8 us: the step gains what the partial
    // traffic cost its neighbours), cfg3 (one window: 64 slices as before) unchanged
    int minp_cfg = p->m > 128 ? nlam_detail::wgrad_min_parts_wide : nlam_detail::wgrad_min_parts;
    const bool solo = (p->flags & NLAM_F_WGRAD_SOLO) != 0;   // nothing runs beside it: slices for isolated speed
    if (minp_cfg < 0) {
        const int win = wgrad_is_wide(p) && !solo ? wgrad_windows_of(p, 256, 256) : 1;
        minp_cfg = (64 + win - 1) / (win < 1 ? 1 : win);
    }
    const long minp = minp_cfg < 1 ? 1 : minp_cfg;
    if (np < minp && total_chunks > minp) np = minp;
    long cap = 512;
    if (p->nsrc == 1 && p->src[0].width <= kSmallN && p->m % 4 == 0) np = (total_chunks + 3) / 4;   // streaming kernel: >= 128 rows per workgroup
    if (wgrad_is_wide(p)) {
        cap = 1024 / wgrad_windows(p);
        // 8-wave workgroups on 256 x 256 windows: at most NLAM_TUNE_WGRAD_MAX_WGS of them.  One per CU (256) until round 6; measured
        // then (profiles/round6/ab_wgrad_chunks.log, ab_wgrad_max_wgs.log): a 512 x 512 gradient over 57 616 rows in 29 slices x 4
        // windows instead of 64 x 4 -- half the partial sums, half the CUs taken from the chain -- cfg5 109.9 -> 108.1 ms; a
        // 256 x 256 one (one window) is indifferent between 113 and 226 slices and loses 3-5 % below 60
        if (wgrad_wbf_ns(p) > 0 && wgrad_wbf_big(p)) cap = (solo ? kNumCUs : nlam_detail::wgrad_max_wgs) / wgrad_windows_of(p, 256, 256);
        if (cap < 4) cap = 4;
    }
    if (np > cap) np = cap;
    return (int32_t)(np < 1 ? 1 : np);
}

#define NLAM_LAUNCH_FWD1(HB_, OB_, FAST_, NS_)                                                                        \
    do {                                                                                                               \
        const size_t lds = fwd_lds_bytes(p, HB_, OB_, NS_);                                                            \
        int rc = set_lds(mlp_fwd_kernel<HB_, OB_, FAST_, NS_>, lds);                                                   \
        if (rc != 0) return rc;                                                                                        \
        hipLaunchKernelGGL((mlp_fwd_kernel<HB_, OB_, FAST_, NS_>), dim3(blocks), dim3(nwaves * 64), lds, stream, *p);  \
    } while (0)
#define NLAM_LAUNCH_FWDBF1(HB_, OB_, NS_, RAG_, RES_)                                                                          \
    do {                                                                                                                      \
        const size_t lds = fwd_lds_bytes(p, HB_, OB_, NS_);                                                                   \
        int rc = set_lds(mlp_fwd_bf_kernel<HB_, OB_, NS_, RAG_, RES_>, lds);                                                  \
        if (rc != 0) return rc;                                                                                               \
        hipLaunchKernelGGL((mlp_fwd_bf_kernel<HB_, OB_, NS_, RAG_, RES_>), dim3(blocks), dim3(nwaves * 64), lds, stream, *p); \
    } while (0)
#define NLAM_LAUNCH_FWDBF(HB_, OB_, NS_)                               \
    do {                                                               \
        if (ragged) NLAM_LAUNCH_FWDBF1(HB_, OB_, NS_, true, false);    \
        else if (resid) NLAM_LAUNCH_FWDBF1(HB_, OB_, NS_, false, true); \
        else NLAM_LAUNCH_FWDBF1(HB_, OB_, NS_, false, false);          \
    } while (0)
#define NLAM_LAUNCH_FWDPRE1(HB_, OB_, NS_, RES_)                                                                                \
    do {                                                                                                                       \
        const size_t lds = fwd_lds_bytes(p, HB_, OB_, NS_);                                                                    \
        int rc = set_lds(mlp_fwd_bf_kernel<HB_, OB_, NS_, false, RES_, true>, lds);                                            \
        if (rc != 0) return rc;                                                                                                \
        hipLaunchKernelGGL((mlp_fwd_bf_kernel<HB_, OB_, NS_, false, RES_, true>), dim3(blocks), dim3(nwaves * 64), lds, stream, *p); \
    } while (0)
#define NLAM_LAUNCH_FWDPRE(HB_, OB_)                                   \
    do {                                                               \
        if (ns == 3) { if (resid) NLAM_LAUNCH_FWDPRE1(HB_, OB_, 3, true); else NLAM_LAUNCH_FWDPRE1(HB_, OB_, 3, false); } \
        else if (ns == 2) { if (resid) NLAM_LAUNCH_FWDPRE1(HB_, OB_, 2, true); else NLAM_LAUNCH_FWDPRE1(HB_, OB_, 2, false); } \
        else { if (resid) NLAM_LAUNCH_FWDPRE1(HB_, OB_, 1, true); else NLAM_LAUNCH_FWDPRE1(HB_, OB_, 1, false); } \
    } while (0)
#define NLAM_LAUNCH_FWD(HB_, OB_)                            \
    do {                                                     \
        if (ns == 3) NLAM_LAUNCH_FWDBF(HB_, OB_, 3);         \
        else if (ns == 2) NLAM_LAUNCH_FWDBF(HB_, OB_, 2);    \
        else if (ns == 1) NLAM_LAUNCH_FWDBF(HB_, OB_, 1);    \
        else if (fast) NLAM_LAUNCH_FWD1(HB_, OB_, true, 0);  \
        else NLAM_LAUNCH_FWD1(HB_, OB_, false, 0);           \
    } while (0)

int32_t nlam_mlp_fwd(const nlam_mlp_fwd_t* p, void* hip_stream) {
    NLAM_RANGE("nlam_mlp_fwd");
    const int32_t bad = nlam_detail::fwd_check(p);
    if (bad != 0) return bad;
    if (p->rows == 0) return 0;
    hipStream_t stream = (hipStream_t)hip_stream;
    if (fwd_is_wide(p)) {
        if (fwd_wbf_ns(p) > 0) return nlam_detail::fwd_wbf(p, stream);   // split-bf16 matrix path (nlam_wbf.inc)
        return nlam_detail::fwd_wide(p, stream);
    }
    return nlam_detail::fwd_narrow(p, stream);
}

}  // extern "C"
#endif

#if NLAM_IN_TU(4)
int32_t nlam_detail::fwd_wbf(const nlam_mlp_fwd_t* p, hipStream_t stream) {
    const int wns = fwd_wbf_ns(p);
    {
            const WbfPlan pl = fwd_wbf_choose(p, wns);
            if ((p->flags & NLAM_F_WPACK_READY) == 0) {   // else: `wpack` was filled for this step already (nlam_pack_records)
                packbf_jobs_t jobs;
                const long most = build_fwd_wbf_jobs(p, wns, jobs);
                long pblocks = (most + 255) / 256;
                if (pblocks > 1024) pblocks = 1024;
                if (wns == 1) hipLaunchKernelGGL(pack_bf_kernel<1>, dim3((int)pblocks, jobs.njobs), dim3(256), 0, stream, jobs);
                else hipLaunchKernelGGL(pack_bf_kernel<3>, dim3((int)pblocks, jobs.njobs), dim3(256), 0, stream, jobs);
            }
            const size_t lds = fwd_wbf_lds(p, wns, pl);
            const long nsuper = (long)((p->ntiles + pl.nrt - 1) / pl.nrt) * p->batch;
            // one 8-wave workgroup per CU, or two of 4 waves (ONE 4-wave workgroup per CU, leaving half of every CU to the
            // weight-gradient kernels of the side streams, measured 20 % slower in the captured cfg3 step: 59.8 vs 49.4 ms)
            const long wcap = pl.nw == 4 ? 2 * nlam_detail::chain_cus : nlam_detail::chain_cus;
            const int wblocks = (int)(nsuper < wcap ? (nsuper < 1 ? 1 : nsuper) : wcap);
            bool v4 = nlam_detail::wbf_v4 != 0 && p->hid % 4 == 0 && p->dout % 4 == 0;   // every chunk a whole 16-byte piece: the branch-free chunk accessors
            for (int s_ = 0; s_ < p->nsrc; ++s_) v4 = v4 && p->src[s_].width % 4 == 0;
#define NLAM_LAUNCH_FWD_WBF1(NS_, NW_, FG_, FB_, RT_, RTP_, SBF_, V4_)                                                         \
    do {                                                                                                                      \
        int rc = set_lds(mlp_fwd_wbf_kernel<NS_, NW_, FG_, FB_, RT_, RTP_, SBF_, V4_>, lds);                                  \
        if (rc != 0) return rc;                                                                                               \
        hipLaunchKernelGGL((mlp_fwd_wbf_kernel<NS_, NW_, FG_, FB_, RT_, RTP_, SBF_, V4_>), dim3(wblocks), dim3(NW_ * 64), lds, stream, *p); \
    } while (0)
#define NLAM_LAUNCH_FWD_WBF(NS_, NW_, FG_, FB_, RT_, RTP_)                              \
    do {                                                                                \
        if (v4) NLAM_LAUNCH_FWD_WBF1(NS_, NW_, FG_, FB_, RT_, RTP_, false, true);       \
        else NLAM_LAUNCH_FWD_WBF1(NS_, NW_, FG_, FB_, RT_, RTP_, false, false);         \
    } while (0)
#define NLAM_LAUNCH_FWD_WBF_S(NW_, FG_, FB_, RT_, RTP_)                                 \
    do {                                                                                \
        if (v4) NLAM_LAUNCH_FWD_WBF1(1, NW_, FG_, FB_, RT_, RTP_, true, true);          \
        else NLAM_LAUNCH_FWD_WBF1(1, NW_, FG_, FB_, RT_, RTP_, true, false);            \
    } while (0)
            // the factorised InteractionNet edge layer of one width -- d = 512 in the one-term mode (cfg5), d = 256 in the fp32-class
            // mode (cfg3): its own software-pipelined kernel (round 6)
            if (nlam_detail::wbf_edge != 0 && v4 && (p->flags & NLAM_F_PRE_ADD) && !(p->flags & (NLAM_F_ADD_SRC1 | NLAM_F_NO_ACT)) &&
                p->nsrc == NLAM_MAX_SRC && p->hid == p->dout && ((wns == 1 && p->hid == 512) || (wns == 3 && p->hid == 256 && !(p->flags & NLAM_F_STORE_BF16))) &&
                p->src[0].width == p->hid && p->src[1].width == p->hid && p->src[2].width == p->hid && p->ln_w != nullptr && p->ln_b != nullptr &&
                p->b1 != nullptr && p->b2 != nullptr && p->aggr != nullptr && p->rowptr != nullptr && p->ncat == 0 &&
                (long)p->ntiles * p->batch >= 2 * 64) {
                if ((p->flags & NLAM_F_STORE_BF16) && !store_bf16_ok(p)) return NLAM_EUNSUP;
                const long ns2 = (long)((p->ntiles + 1) / 2) * p->batch;
                const int eblocks = (int)(ns2 < nlam_detail::chain_cus ? ns2 : nlam_detail::chain_cus);
#define NLAM_LAUNCH_FWD_EDGE(NS_, D_, SBF_)                                                                               \
    do {                                                                                                                  \
        const size_t elds = fwd_edge_lds<NS_, D_>();                                                                       \
        int rc = set_lds(mlp_fwd_edge_kernel<NS_, D_, SBF_>, elds);                                                        \
        if (rc != 0) return rc;                                                                                           \
        hipLaunchKernelGGL((mlp_fwd_edge_kernel<NS_, D_, SBF_>), dim3(eblocks), dim3(512), elds, stream, *p);              \
    } while (0)
                // (d = 256 in the one-term mode -- cfg3 under autocast, not a BASELINE configuration -- measured no faster than the
                // template's 4-wave shape, 103.3 vs 102.5 us on the m2m edges: not instantiated)
                if (wns == 3) NLAM_LAUNCH_FWD_EDGE(3, 256, false);
                else if (p->flags & NLAM_F_STORE_BF16) NLAM_LAUNCH_FWD_EDGE(1, 512, true);
                else NLAM_LAUNCH_FWD_EDGE(1, 512, false);
                return (int32_t)hipGetLastError();
            }
            if (p->flags & NLAM_F_STORE_BF16) {   // z1 / xhat as bf16 rows: one term, whole blocks, the shapes of store_bf16_ok()
                if (!store_bf16_ok(p)) return NLAM_EUNSUP;
                if (pl.cfg == 2) NLAM_LAUNCH_FWD_WBF_S(8, 8, 1, 4, 2);
                else if (pl.cfg == 5) NLAM_LAUNCH_FWD_WBF_S(4, 4, 2, 2, 1);
                else if (pl.cfg == 6) NLAM_LAUNCH_FWD_WBF_S(4, 4, 4, 1, 1);
                else if (pl.cfg == 7) NLAM_LAUNCH_FWD_WBF_S(4, 4, 4, 2, 2);
                else if (pl.cfg == 3) NLAM_LAUNCH_FWD_WBF_S(8, 8, 2, 2, 1);
                else return NLAM_EUNSUP;
                return (int32_t)hipGetLastError();
            }
            if (wns == 1) {
                if (pl.cfg == 1) NLAM_LAUNCH_FWD_WBF(1, 8, 4, 1, 4, 2);
                else if (pl.cfg == 2) NLAM_LAUNCH_FWD_WBF(1, 8, 8, 1, 4, 2);
                else if (pl.cfg == 4) NLAM_LAUNCH_FWD_WBF(1, 8, 8, 1, 2, 2);
                else if (pl.cfg == 5) NLAM_LAUNCH_FWD_WBF(1, 4, 4, 2, 2, 1);
                else if (pl.cfg == 6) NLAM_LAUNCH_FWD_WBF(1, 4, 4, 4, 1, 1);
                else if (pl.cfg == 7) NLAM_LAUNCH_FWD_WBF(1, 4, 4, 4, 2, 2);
                else NLAM_LAUNCH_FWD_WBF(1, 8, 8, 2, 2, 1);
            } else {
                if (pl.cfg == 1) NLAM_LAUNCH_FWD_WBF(3, 8, 4, 1, 4, 2);
                else if (pl.cfg == 2) NLAM_LAUNCH_FWD_WBF(3, 8, 8, 1, 4, 2);
                else if (pl.cfg == 4) NLAM_LAUNCH_FWD_WBF(3, 8, 8, 1, 2, 2);
                else if (pl.cfg == 5) NLAM_LAUNCH_FWD_WBF(3, 4, 4, 2, 2, 1);
                else NLAM_LAUNCH_FWD_WBF(3, 8, 8, 2, 2, 1);
            }
            return (int32_t)hipGetLastError();
    }
}
#endif

#if NLAM_IN_TU(3)
int32_t nlam_detail::fwd_wide(const nlam_mlp_fwd_t* p, hipStream_t stream) {
    const WideCfg cfg = wide_cfg(p->hid > p->dout ? p->hid : p->dout);
    {
        if ((p->flags & NLAM_F_WPACK_READY) == 0) {
            pack_jobs_t jobs;
            build_fwd_wide_jobs(p, jobs);
            launch_pack(jobs, stream);
        }
        const size_t lds = fwd_wide_lds(p, cfg.nwv);
        const int wblocks = wide_grid((long)p->ntiles * p->batch, lds, cfg.nwv);
#define NLAM_LAUNCH_FWD_WIDE(NWV_, FB_)                                                                       \
    do {                                                                                                      \
        int rc = set_lds(mlp_fwd_wide_kernel<NWV_, FB_>, lds);                                                \
        if (rc != 0) return rc;                                                                               \
        hipLaunchKernelGGL((mlp_fwd_wide_kernel<NWV_, FB_>), dim3(wblocks), dim3(NWV_ * 64), lds, stream, *p); \
    } while (0)
        if (cfg.nwv == 4) NLAM_LAUNCH_FWD_WIDE(4, 1);
        else if (cfg.fb == 1) NLAM_LAUNCH_FWD_WIDE(8, 1);
        else NLAM_LAUNCH_FWD_WIDE(8, 2);
        return (int32_t)hipGetLastError();
    }
}
#endif

#if NLAM_IN_TU(1)
int32_t nlam_detail::fwd_narrow(const nlam_mlp_fwd_t* p, hipStream_t stream) {
    // launch shape: one persistent workgroup of kFwdWaves waves per CU (fewer workgroups than CUs only when there
    // are fewer tiles than CUs); tiles are dealt to workgroups first, then to waves
    const long ttiles = (long)p->ntiles * p->batch;
    const int nwaves = kFwdWaves;
    const int blocks = (int)(ttiles < 1 ? 1 : (ttiles < kMaxGridBlocks ? ttiles : kMaxGridBlocks));
    const int HB = (p->hid + 31) / 32, OB = (p->dout + 31) / 32;
    // an output width that is not a whole 32-column block runs on the split-bf16 kernels when nothing downstream of GEMM2
    // needs whole blocks: no LayerNorm, no aggregation, no residual (output_map, graph/base.py:322); W2 / b2 are zero-padded
    // when staged and the ragged-input instantiation stores the block as dwords
    const bool ragged_out = (p->dout % 32 != 0) && p->ln_w == nullptr && p->aggr == nullptr && p->nsrc == 1 && p->out != nullptr &&
                            (p->flags & (NLAM_F_ADD_SRC0 | NLAM_F_ADD_SRC1 | NLAM_F_PRE_ADD)) == 0 && p->xhat == nullptr;
    const bool fast_out = (p->hid % 32 == 0) && (p->dout % 32 == 0 || ragged_out);
    bool fast = fast_out && !ragged_out, w64 = true, ragged = ragged_out;
    for (int s = 0; s < p->nsrc; ++s) {
        fast = fast && (p->src[s].width % 8 == 0);
        w64 = w64 && p->src[s].width <= 64;
        ragged = ragged || (p->src[s].width % 32 != 0);
    }
    if (ragged && p->nsrc != 1) w64 = false;   // ragged inputs are covered for single-source MLPs (embedders, grid MLPs)
    if (p->ncat > 0) ragged = true;            // concatenated pieces are read by the ragged-input instantiation
    const bool resid = ((p->flags & NLAM_F_ADD_SRC0) != 0 && p->out != nullptr) || (p->flags & NLAM_F_ADD_SRC1) != 0;
    if (ragged && resid) w64 = false;
    if ((p->flags & (NLAM_F_ADD_SRC0 | NLAM_F_ADD_SRC1)) != 0) {   // residual rows are read as whole 16-B chunks
        const int rs = (p->flags & NLAM_F_ADD_SRC1) ? 1 : 0;
        w64 = w64 && (p->src[rs].width % 32 == 0) && ((p->flags & NLAM_F_ADD_SRC0) == 0 || p->src[0].width % 32 == 0);
    }
    // matrix path: NLAM_F_MM_* asks for the split-bf16 cores; shapes they do not cover run the fp32 MFMA
    int ns = 0;
    if (fast_out && w64) ns = (int)((p->flags & NLAM_F_MM_MASK) >> NLAM_F_MM_SHIFT);
    if (ns > 3) return NLAM_EINVAL;
    if (p->ncat > 0) {   // concatenated pieces: the CAT instantiation of the ragged-input split-bf16 kernel, square MLPs
        if (ns == 0 || HB != OB || HB > 2) return NLAM_EUNSUP;
        nlam_mlp_fwd_t pc = *p;   // unused piece slots: width 0 (they start at the total width and are never selected)
        for (int k = p->ncat; k < NLAM_MAX_CAT; ++k) {
            pc.cat_ptr[k] = p->cat_ptr[0];
            pc.cat_bstride[k] = 0;
            pc.cat_width[k] = 0;
        }
#define NLAM_LAUNCH_FWDCAT1(HB_, NS_)                                                                                                    \
    do {                                                                                                                                \
        const size_t lds = fwd_lds_bytes(p, HB_, HB_, NS_);                                                                             \
        int rc = set_lds(mlp_fwd_bf_kernel<HB_, HB_, NS_, true, false, false, true>, lds);                                              \
        if (rc != 0) return rc;                                                                                                         \
        hipLaunchKernelGGL((mlp_fwd_bf_kernel<HB_, HB_, NS_, true, false, false, true>), dim3(blocks), dim3(nwaves * 64), lds, stream, pc); \
    } while (0)
#define NLAM_LAUNCH_FWDCAT(HB_)                       \
    do {                                              \
        if (ns == 3) NLAM_LAUNCH_FWDCAT1(HB_, 3);      \
        else if (ns == 2) NLAM_LAUNCH_FWDCAT1(HB_, 2); \
        else NLAM_LAUNCH_FWDCAT1(HB_, 1);              \
    } while (0)
        if (HB == 1) NLAM_LAUNCH_FWDCAT(1);
        else NLAM_LAUNCH_FWDCAT(2);
        return (int32_t)hipGetLastError();
    }
    if (p->flags & NLAM_F_PRE_ADD) {   // factorised edge MLP: split-bf16 modes, whole 32-column units, addends of width hid
        bool ok = ns > 0 && !ragged && p->nsrc >= 2 && (p->flags & NLAM_F_ADD_SRC1) == 0 && HB == OB;
        for (int s = 1; s < p->nsrc; ++s) ok = ok && p->src[s].width == p->hid;
        if (!ok) return NLAM_EUNSUP;
        if (HB == 1) NLAM_LAUNCH_FWDPRE(1, 1);
        else if (HB == 2) NLAM_LAUNCH_FWDPRE(2, 2);
        else return NLAM_EUNSUP;
        return (int32_t)hipGetLastError();
    }
    if (HB == 1 && OB == 1) NLAM_LAUNCH_FWD(1, 1);
    else if (HB == 2 && OB == 1) NLAM_LAUNCH_FWD(2, 1);
    else if (HB == 1 && OB == 2) NLAM_LAUNCH_FWD(1, 2);
    else if (HB == 2 && OB == 2) NLAM_LAUNCH_FWD(2, 2);
    else return NLAM_EUNSUP;
    return (int32_t)hipGetLastError();
}
#endif

#if NLAM_IN_TU(1)
// the argument checks of nlam_mlp_fwd / nlam_mlp_bwd, shared with the grouped entry points
int32_t nlam_detail::fwd_check(const nlam_mlp_fwd_t* p) {
    if (p == nullptr || p->nsrc < 1 || p->nsrc > NLAM_MAX_SRC || p->W1 == nullptr || p->W2 == nullptr) return NLAM_EINVAL;
    if (p->batch < 1 || p->rows < 0 || p->hid < 1 || p->dout < 1) return NLAM_EINVAL;
    if (p->aggr != nullptr && (p->rowptr == nullptr || p->tiles == nullptr)) return NLAM_EINVAL;
    if ((p->flags & NLAM_F_MEAN) && p->inv_deg == nullptr) return NLAM_EINVAL;
    if ((p->flags & NLAM_F_ADD_SRC0) && p->src[0].width != p->dout) return NLAM_EINVAL;
    if ((p->flags & NLAM_F_ADD_SRC1) && (p->nsrc < 2 || p->src[1].width != p->dout)) return NLAM_EINVAL;
    if ((p->flags & NLAM_F_NO_ACT) && ((p->flags & (NLAM_F_MM_MASK | NLAM_F_PRE_ADD)) != 0 || p->ncat != 0)) return NLAM_EUNSUP;
    if ((p->flags & NLAM_F_STORE_BF16) && !store_bf16_ok(p)) return NLAM_EUNSUP;
    if (p->ncat != 0) {   // src[0] = concatenation of pieces: one un-gathered source of <= 64 columns, whole float4s per row
        if (p->ncat < 0 || p->ncat > NLAM_MAX_CAT || p->nsrc != 1 || p->src[0].idx != nullptr) return NLAM_EINVAL;
        int wsum = 0;
        for (int k = 0; k < p->ncat; ++k) {
            if (p->cat_ptr[k] == nullptr || p->cat_width[k] < 1) return NLAM_EINVAL;
            wsum += p->cat_width[k];
        }
        if (wsum != p->src[0].width) return NLAM_EINVAL;
        if (fwd_is_wide(p) || wsum % 4 != 0 || (p->flags & (NLAM_F_ADD_SRC0 | NLAM_F_ADD_SRC1 | NLAM_F_PRE_ADD)) != 0 || p->tiles != nullptr)
            return NLAM_EUNSUP;
        if (p->ncat > 4) return NLAM_EUNSUP;   // the kernel resolves a column among four pieces
        if (p->rows > 0 && p->batch > 0 && ((long)p->rows >= (1L << 31) || p->batch >= (1 << 30))) return NLAM_EUNSUP;
    }
    if (p->rows == 0) return 0;
    if (fwd_is_wide(p)) {
        if ((p->flags & NLAM_F_PRE_ADD) && fwd_wbf_ns(p) == 0) return NLAM_EUNSUP;   // the fp32 wide kernels have no factorised variant
        const WideCfg cfg = wide_cfg(p->hid > p->dout ? p->hid : p->dout);
        if (cfg.nwv == 0) return NLAM_EUNSUP;
        const int64_t need = nlam_mlp_fwd_wpack_floats(p);
        if (p->wpack == nullptr || p->wpack_floats < need) return NLAM_EINVAL;
    }
    return 0;
}

int32_t nlam_detail::bwd_check(const nlam_mlp_bwd_t* p) {
    if (p == nullptr || p->nsrc < 1 || p->nsrc > NLAM_MAX_SRC || p->W1 == nullptr || p->W2 == nullptr) return NLAM_EINVAL;
    if (p->z1 == nullptr) return NLAM_EINVAL;
    if (p->ln_w != nullptr && (p->xhat == nullptr || p->rstd == nullptr)) return NLAM_EINVAL;
    if (p->g_aggr != nullptr && p->seg_of_row == nullptr) return NLAM_EINVAL;
    if ((p->flags & NLAM_F_MEAN) && p->g_aggr != nullptr && p->inv_deg == nullptr) return NLAM_EINVAL;
    for (int s = 0; s < p->nsrc; ++s) {
        if (p->dmode[s] != 0 && p->dsrc[s] == nullptr) return NLAM_EINVAL;
        if (p->dmode[s] == 3 && (p->rowptr == nullptr || p->tiles == nullptr)) return NLAM_EINVAL;
    }
    if (p->vec_partials != nullptr) {
        const int wmax = p->hid > p->dout ? p->hid : p->dout;
        if (p->vec_partials_rows < nlam_mlp_bwd_blocks(p) || p->vec_stride < wmax || p->vec_stride < 64 || p->vec_stride % 64 != 0)
            return NLAM_EINVAL;
    }
    if ((p->flags & NLAM_F_NO_ACT) && (p->flags & (NLAM_F_MM_MASK | NLAM_F_PRE_ADD | NLAM_F_LEAF_WGRAD)) != 0) return NLAM_EUNSUP;
    if ((p->flags & NLAM_F_STORE_BF16) && !(bwd_is_wide(p) && bwd_wbf_ns(p) == 1)) return NLAM_EUNSUP;
    if (p->rows == 0) return 0;
    if (bwd_is_wide(p) && (p->flags & NLAM_F_PRE_ADD) && bwd_wbf_ns(p) == 0) return NLAM_EUNSUP;
    if (bwd_is_wide(p)) {
        const WideCfg cfg = wide_cfg(bwd_wide_maxw(p));
        if (cfg.nwv == 0) return NLAM_EUNSUP;
        const int64_t need = nlam_mlp_bwd_wpack_floats(p);
        if (p->wpack == nullptr || p->wpack_floats < need) return NLAM_EINVAL;
        if (p->dz2_ld != nlam_mlp_bwd_dz2_ld(p)) return NLAM_EINVAL;   // the caller sized dz2 with nlam_mlp_bwd_dz2_ld
    }
    return 0;
}
#endif

#if NLAM_IN_TU(1)
extern "C" {


int32_t nlam_mlp_bwd(const nlam_mlp_bwd_t* p, void* hip_stream) {
    NLAM_RANGE("nlam_mlp_bwd");
    const int32_t bad = nlam_detail::bwd_check(p);
    if (bad != 0) return bad;
    if (p->rows == 0) return 0;
    hipStream_t stream = (hipStream_t)hip_stream;
    if (bwd_is_wide(p)) {
        if (bwd_wbf_ns(p) > 0) return nlam_detail::bwd_wbf(p, stream);   // split-bf16 matrix path (nlam_wbf.inc)
        if (p->flags & NLAM_F_ACC_DSRC0) return NLAM_EUNSUP;
        return nlam_detail::bwd_wide(p, stream);
    }
    if (p->flags & NLAM_F_ACC_DSRC0) return NLAM_EUNSUP;
    return nlam_detail::bwd_narrow(p, stream);
}

}  // extern "C"
#endif

#if NLAM_IN_TU(4)
int32_t nlam_detail::bwd_wbf(const nlam_mlp_bwd_t* p, hipStream_t stream) {
    const int wns = bwd_wbf_ns(p);
    {
            const WbfBwdPlan pl = wbf_bwd_choose(bwd_wide_maxw(p));
            if ((p->flags & NLAM_F_WPACK_READY) == 0) {
                packbf_jobs_t jobs;
                const long most = build_bwd_wbf_jobs(p, wns, jobs);
                long pblocks = (most + 255) / 256;
                if (pblocks > 1024) pblocks = 1024;
                if (wns == 1) hipLaunchKernelGGL(pack_bf_kernel<1>, dim3((int)pblocks, jobs.njobs), dim3(256), 0, stream, jobs);
                else hipLaunchKernelGGL(pack_bf_kernel<3>, dim3((int)pblocks, jobs.njobs), dim3(256), 0, stream, jobs);
            }
            if (bwd_edge_ok(p)) {
                const int eblocks = nlam_mlp_bwd_blocks(p);
                if (wns == 3) {
                    const size_t elds = bwd_edge_lds<3, 256>();
                    int rc = set_lds(mlp_bwd_edge_kernel<3, 256, false>, elds);
                    if (rc != 0) return rc;
                    hipLaunchKernelGGL((mlp_bwd_edge_kernel<3, 256, false>), dim3(eblocks), dim3(512), elds, stream, *p);
                } else if (p->flags & NLAM_F_STORE_BF16) {
                    const size_t elds = bwd_edge_lds<1, 512>();
                    int rc = set_lds(mlp_bwd_edge_kernel<1, 512, true>, elds);
                    if (rc != 0) return rc;
                    hipLaunchKernelGGL((mlp_bwd_edge_kernel<1, 512, true>), dim3(eblocks), dim3(512), elds, stream, *p);
                } else {
                    const size_t elds = bwd_edge_lds<1, 512>();
                    int rc = set_lds(mlp_bwd_edge_kernel<1, 512, false>, elds);
                    if (rc != 0) return rc;
                    hipLaunchKernelGGL((mlp_bwd_edge_kernel<1, 512, false>), dim3(eblocks), dim3(512), elds, stream, *p);
                }
                return (int32_t)hipGetLastError();
            }
            const size_t lds = bwd_wbf_lds(p, wns, pl);
            const int wblocks = nlam_mlp_bwd_blocks(p) / pl.rg;
            bool v4 = nlam_detail::wbf_v4 != 0 && p->hid % 4 == 0 && p->dout % 4 == 0 && p->dz2_ld == 0;
            for (int s_ = 0; s_ < p->nsrc; ++s_) v4 = v4 && p->src[s_].width % 4 == 0;
#define NLAM_LAUNCH_BWD_WBF1(NS_, NW_, FG_, FB_, RT_, SBF_, V4_)                                                               \
    do {                                                                                                                      \
        int rc = set_lds(mlp_bwd_wbf_kernel<NS_, NW_, FG_, FB_, RT_, SBF_, V4_>, lds);                                        \
        if (rc != 0) return rc;                                                                                               \
        hipLaunchKernelGGL((mlp_bwd_wbf_kernel<NS_, NW_, FG_, FB_, RT_, SBF_, V4_>), dim3(wblocks), dim3(NW_ * 64), lds, stream, *p); \
    } while (0)
#define NLAM_LAUNCH_BWD_WBF(NS_, NW_, FG_, FB_, RT_)                               \
    do {                                                                           \
        if (v4) NLAM_LAUNCH_BWD_WBF1(NS_, NW_, FG_, FB_, RT_, false, true);        \
        else NLAM_LAUNCH_BWD_WBF1(NS_, NW_, FG_, FB_, RT_, false, false);          \
    } while (0)
#define NLAM_LAUNCH_BWD_WBF_S(NW_, FG_, FB_, RT_)                                  \
    do {                                                                           \
        if (v4) NLAM_LAUNCH_BWD_WBF1(1, NW_, FG_, FB_, RT_, true, true);           \
        else NLAM_LAUNCH_BWD_WBF1(1, NW_, FG_, FB_, RT_, true, false);             \
    } while (0)
            if (p->flags & NLAM_F_STORE_BF16) {
                if (wns != 1 || p->hid % 32 != 0 || p->dout % 32 != 0 || p->dz2_ld != 0 || pl.cfg == 1) return NLAM_EUNSUP;
                if (pl.cfg == 2) NLAM_LAUNCH_BWD_WBF_S(8, 8, 1, 2);
                else if (pl.cfg == 5) NLAM_LAUNCH_BWD_WBF_S(4, 4, 2, 1);
                else NLAM_LAUNCH_BWD_WBF_S(8, 8, 2, 1);
                return (int32_t)hipGetLastError();
            }
            if (wns == 1) {
                if (pl.cfg == 1) NLAM_LAUNCH_BWD_WBF(1, 8, 4, 1, 2);
                else if (pl.cfg == 2) NLAM_LAUNCH_BWD_WBF(1, 8, 8, 1, 2);
                else if (pl.cfg == 5) NLAM_LAUNCH_BWD_WBF(1, 4, 4, 2, 1);
                else NLAM_LAUNCH_BWD_WBF(1, 8, 8, 2, 1);
            } else {
                if (pl.cfg == 1) NLAM_LAUNCH_BWD_WBF(3, 8, 4, 1, 2);
                else if (pl.cfg == 2) NLAM_LAUNCH_BWD_WBF(3, 8, 8, 1, 2);
                else if (pl.cfg == 5) NLAM_LAUNCH_BWD_WBF(3, 4, 4, 2, 1);
                else NLAM_LAUNCH_BWD_WBF(3, 8, 8, 2, 1);
            }
            return (int32_t)hipGetLastError();
    }
}
#endif

#if NLAM_IN_TU(3)
int32_t nlam_detail::bwd_wide(const nlam_mlp_bwd_t* p, hipStream_t stream) {
    const WideCfg cfg = wide_cfg(bwd_wide_maxw(p));
    {
        if ((p->flags & NLAM_F_WPACK_READY) == 0) {
            pack_jobs_t jobs;
            build_bwd_wide_jobs(p, jobs);
            launch_pack(jobs, stream);
        }
        const size_t lds = bwd_wide_lds(p, cfg.nwv);
        const int wblocks = nlam_mlp_bwd_blocks(p);
#define NLAM_LAUNCH_BWD_WIDE(NWV_, FB_)                                                                       \
    do {                                                                                                      \
        int rc = set_lds(mlp_bwd_wide_kernel<NWV_, FB_>, lds);                                                \
        if (rc != 0) return rc;                                                                               \
        hipLaunchKernelGGL((mlp_bwd_wide_kernel<NWV_, FB_>), dim3(wblocks), dim3(NWV_ * 64), lds, stream, *p); \
    } while (0)
        if (cfg.nwv == 4) NLAM_LAUNCH_BWD_WIDE(4, 1);
        else if (cfg.fb == 1) NLAM_LAUNCH_BWD_WIDE(8, 1);
        else NLAM_LAUNCH_BWD_WIDE(8, 2);
        return (int32_t)hipGetLastError();
    }
}
#endif

#if NLAM_IN_TU(3)
// Grouped launches of the fp32 wide kernels (members checked by nlam_mlp_fwd_group / nlam_mlp_bwd_group: one shape, every one
// of them this family).  Workgroups are dealt over the members in proportion to their tiles (wide_group_blocks).
int32_t nlam_detail::fwd_wide_group(const nlam_mlp_fwd_t* ps, int n, hipStream_t stream) {
    const WideCfg cfg = wide_cfg(ps[0].hid > ps[0].dout ? ps[0].hid : ps[0].dout);
    fwd_wide_group_t G;
    long tiles[NLAM_MAX_GROUP];
    size_t lds = 0;
    for (int k = 0; k < n; ++k) {
        if ((ps[k].flags & NLAM_F_WPACK_READY) == 0) {
            pack_jobs_t jobs;
            build_fwd_wide_jobs(&ps[k], jobs);
            launch_pack(jobs, stream);
        }
        G.g[k] = ps[k];
        tiles[k] = (long)ps[k].ntiles * ps[k].batch;
        const size_t l = fwd_wide_lds(&ps[k], cfg.nwv);
        if (l > lds) lds = l;
    }
    int blocks[NLAM_MAX_GROUP];
    wide_group_blocks(tiles, n, wide_grid(1L << 40, lds, cfg.nwv), blocks);
    G.n = n;
    G.first[0] = 0;
    for (int k = 0; k < n; ++k) G.first[k + 1] = G.first[k] + blocks[k];
    for (int k = n + 1; k <= NLAM_MAX_GROUP; ++k) G.first[k] = G.first[n];
#define NLAM_LAUNCH_FWD_WIDE_G(NWV_, FB_)                                                                              \
    do {                                                                                                               \
        int rc = set_lds(mlp_fwd_wide_group_kernel<NWV_, FB_>, lds);                                                   \
        if (rc != 0) return rc;                                                                                        \
        hipLaunchKernelGGL((mlp_fwd_wide_group_kernel<NWV_, FB_>), dim3(G.first[n]), dim3(NWV_ * 64), lds, stream, G); \
    } while (0)
    if (cfg.nwv == 4) NLAM_LAUNCH_FWD_WIDE_G(4, 1);
    else if (cfg.fb == 1) NLAM_LAUNCH_FWD_WIDE_G(8, 1);
    else NLAM_LAUNCH_FWD_WIDE_G(8, 2);
    return (int32_t)hipGetLastError();
}

int32_t nlam_detail::bwd_wide_group(const nlam_mlp_bwd_t* ps, int n, hipStream_t stream) {
    const WideCfg cfg = wide_cfg(bwd_wide_maxw(&ps[0]));
    bwd_wide_group_t G;
    int blocks[NLAM_MAX_GROUP];
    const int32_t rc0 = nlam_mlp_bwd_group_blocks(ps, n, blocks);
    if (rc0 != 0) return rc0;
    size_t lds = 0;
    for (int k = 0; k < n; ++k) {
        if ((ps[k].flags & NLAM_F_WPACK_READY) == 0) {
            pack_jobs_t jobs;
            build_bwd_wide_jobs(&ps[k], jobs);
            launch_pack(jobs, stream);
        }
        G.g[k] = ps[k];
        const size_t l = bwd_wide_lds(&ps[k], cfg.nwv);
        if (l > lds) lds = l;
    }
    G.n = n;
    G.first[0] = 0;
    for (int k = 0; k < n; ++k) G.first[k + 1] = G.first[k] + blocks[k];
    for (int k = n + 1; k <= NLAM_MAX_GROUP; ++k) G.first[k] = G.first[n];
#define NLAM_LAUNCH_BWD_WIDE_G(NWV_, FB_)                                                                              \
    do {                                                                                                               \
        int rc = set_lds(mlp_bwd_wide_group_kernel<NWV_, FB_>, lds);                                                   \
        if (rc != 0) return rc;                                                                                        \
        hipLaunchKernelGGL((mlp_bwd_wide_group_kernel<NWV_, FB_>), dim3(G.first[n]), dim3(NWV_ * 64), lds, stream, G); \
    } while (0)
    if (cfg.nwv == 4) NLAM_LAUNCH_BWD_WIDE_G(4, 1);
    else if (cfg.fb == 1) NLAM_LAUNCH_BWD_WIDE_G(8, 1);
    else NLAM_LAUNCH_BWD_WIDE_G(8, 2);
    return (int32_t)hipGetLastError();
}
#endif

#if NLAM_IN_TU(1)
extern "C" {

int32_t nlam_mlp_group_blocks(const int64_t* tiles, int32_t n, int32_t* blocks) {
    if (tiles == nullptr || blocks == nullptr || n < 1 || n > NLAM_MAX_GROUP) return NLAM_EINVAL;
    long t[NLAM_MAX_GROUP];
    for (int k = 0; k < n; ++k) t[k] = (long)tiles[k];
    group_blocks(t, n, blocks);
    return 0;
}

int32_t nlam_mlp_fwd_group(const nlam_mlp_fwd_t* ps, int32_t n, void* hip_stream) {
    NLAM_RANGE("nlam_mlp_fwd_group");
    if (ps == nullptr || n < 1 || n > NLAM_MAX_GROUP) return NLAM_EINVAL;
    if (fwd_is_wide(&ps[0])) {   // members of the fp32 wide family (the chunks of a SplitMLPs layer at d = 128): any source / flag set, one shape
        for (int k = 0; k < n; ++k) {
            const nlam_mlp_fwd_t& p = ps[k];
            const int32_t bad = nlam_detail::fwd_check(&p);
            if (bad != 0) return bad;
            if (p.rows < 1) return NLAM_EINVAL;
            if (nlam_mlp_fwd_family(&p) != 1 || !same_fwd_shape(p, ps[0])) return NLAM_EUNSUP;
        }
        return nlam_detail::fwd_wide_group(ps, n, (hipStream_t)hip_stream);
    }
    const int HB = (ps[0].hid + 31) / 32, OB = (ps[0].dout + 31) / 32;
    const int ns = (int)((ps[0].flags & NLAM_F_MM_MASK) >> NLAM_F_MM_SHIFT);
    if (ns == 0 || HB != OB || HB > 2) return NLAM_EUNSUP;
    fwd_group_t G;
    long tiles[NLAM_MAX_GROUP];
    size_t lds = 0;
    for (int k = 0; k < n; ++k) {
        const nlam_mlp_fwd_t& p = ps[k];
        if (p.W1 == nullptr || p.W2 == nullptr || p.batch < 1 || p.rows < 1 || p.out == nullptr) return NLAM_EINVAL;
        // same kernel instantiation for every member: one source of at most 64 columns, whole output blocks, no residual /
        // aggregation / scatter, the same matrix mode
        if (p.nsrc != 1 || fwd_is_wide(&p) || p.hid != ps[0].hid || p.dout != ps[0].dout || p.hid % 32 != 0 || p.dout % 32 != 0) return NLAM_EUNSUP;
        if ((p.flags & ~NLAM_F_MM_MASK) != 0 || (p.flags & NLAM_F_MM_MASK) != (ps[0].flags & NLAM_F_MM_MASK)) return NLAM_EUNSUP;
        if (p.aggr != nullptr || p.out_idx != nullptr || p.src[0].idx != nullptr || p.tiles != nullptr) return NLAM_EUNSUP;
        if ((p.ln_w == nullptr) != (ps[0].ln_w == nullptr)) return NLAM_EUNSUP;
        G.g[k] = p;
        tiles[k] = (long)p.ntiles * p.batch;
        const size_t l = fwd_lds_bytes(&p, HB, OB, ns);
        if (l > lds) lds = l;
    }
    int blocks[NLAM_MAX_GROUP];
    group_blocks(tiles, n, blocks);
    G.n = n;
    G.first[0] = 0;
    for (int k = 0; k < n; ++k) G.first[k + 1] = G.first[k] + blocks[k];
    for (int k = n + 1; k <= NLAM_MAX_GROUP; ++k) G.first[k] = G.first[n];
    hipStream_t stream = (hipStream_t)hip_stream;
#define NLAM_LAUNCH_FWDG(HB_, NS_)                                                                                         \
    do {                                                                                                                   \
        int rc = set_lds(mlp_fwd_bf_group_kernel<HB_, HB_, NS_>, lds);                                                     \
        if (rc != 0) return rc;                                                                                            \
        hipLaunchKernelGGL((mlp_fwd_bf_group_kernel<HB_, HB_, NS_>), dim3(G.first[n]), dim3(kFwdThreads), lds, stream, G); \
    } while (0)
    if (HB == 1) {
        if (ns == 3) NLAM_LAUNCH_FWDG(1, 3);
        else if (ns == 2) NLAM_LAUNCH_FWDG(1, 2);
        else NLAM_LAUNCH_FWDG(1, 1);
    } else {
        if (ns == 3) NLAM_LAUNCH_FWDG(2, 3);
        else if (ns == 2) NLAM_LAUNCH_FWDG(2, 2);
        else NLAM_LAUNCH_FWDG(2, 1);
    }
    return (int32_t)hipGetLastError();
}

}  // extern "C"
#endif

#if NLAM_IN_TU(2)
extern "C" int32_t nlam_mlp_bwd_group(const nlam_mlp_bwd_t* ps, int32_t n, void* hip_stream) {
    NLAM_RANGE("nlam_mlp_bwd_group");
    if (ps == nullptr || n < 1 || n > NLAM_MAX_GROUP) return NLAM_EINVAL;
    if (bwd_is_wide(&ps[0])) {   // fp32 wide family: see nlam_mlp_fwd_group
        if (ps[0].flags & NLAM_F_LEAF_WGRAD) return NLAM_EUNSUP;   // weight gradients inside the kernel: narrow members only
        int blocks[NLAM_MAX_GROUP];
        const int32_t rcb = nlam_mlp_bwd_group_blocks(ps, n, blocks);   // checks the members
        if (rcb != 0) return rcb;
        for (int k = 0; k < n; ++k)
            if (ps[k].vec_partials != nullptr && ps[k].vec_partials_rows < blocks[k]) return NLAM_EINVAL;
        return nlam_detail::bwd_wide_group(ps, n, (hipStream_t)hip_stream);
    }
    const int HB = (ps[0].hid + 31) / 32, OB = (ps[0].dout + 31) / 32;
    const int ns = (int)((ps[0].flags & NLAM_F_MM_MASK) >> NLAM_F_MM_SHIFT);
    if (ns == 0 || HB != OB || HB > 2) return NLAM_EUNSUP;
    bwd_group_t G;
    long tiles[NLAM_MAX_GROUP];
    size_t lds = 0;
    int blocks[NLAM_MAX_GROUP];
    for (int k = 0; k < n; ++k) {
        const nlam_mlp_bwd_t& p = ps[k];
        if (p.W1 == nullptr || p.W2 == nullptr || p.batch < 1 || p.rows < 1 || p.g_out == nullptr) return NLAM_EINVAL;
        if ((p.flags & NLAM_F_LEAF_WGRAD) ? p.b1 == nullptr : p.z1 == nullptr) return NLAM_EINVAL;   // fused weight gradients: z1 is recomputed
        if (p.nsrc != 1 || bwd_is_wide(&p) || p.hid != ps[0].hid || p.dout != ps[0].dout || p.hid % 32 != 0 || p.dout % 32 != 0) return NLAM_EUNSUP;
        if ((p.flags & ~(NLAM_F_MM_MASK | NLAM_F_LEAF_WGRAD)) != 0 || (p.flags & (NLAM_F_MM_MASK | NLAM_F_LEAF_WGRAD)) != (ps[0].flags & (NLAM_F_MM_MASK | NLAM_F_LEAF_WGRAD)))
            return NLAM_EUNSUP;
        if ((p.flags & NLAM_F_LEAF_WGRAD) && (p.src[0].width > 4 || p.src[0].ptr == nullptr || p.src[0].idx != nullptr || p.dz2 == nullptr || p.vec_partials == nullptr))
            return NLAM_EUNSUP;   // fused weight gradients: <= 4 input columns, p.dz2 = (blocks, dout, hid) partials, 7 vector rows
        if (p.g_aggr != nullptr || p.out_idx != nullptr || p.tiles != nullptr || p.dmode[0] != 0) return NLAM_EUNSUP;   // leaf MLPs: no data gradient
        if ((p.ln_w == nullptr) != (ps[0].ln_w == nullptr)) return NLAM_EUNSUP;
        G.g[k] = p;
        tiles[k] = (long)p.ntiles * p.batch;
        const size_t l = bwd_fast_lds_bytes(&p, HB, OB, ns);
        if (l > lds) lds = l;
    }
    group_blocks(tiles, n, blocks);
    G.n = n;
    G.first[0] = 0;
    for (int k = 0; k < n; ++k) {
        if (ps[k].vec_partials != nullptr && ps[k].vec_partials_rows < blocks[k]) return NLAM_EINVAL;
        G.first[k + 1] = G.first[k] + blocks[k];
    }
    for (int k = n + 1; k <= NLAM_MAX_GROUP; ++k) G.first[k] = G.first[n];
    hipStream_t stream = (hipStream_t)hip_stream;
#define NLAM_LAUNCH_BWDG1(HB_, NS_, LW_)                                                                                            \
    do {                                                                                                                            \
        int rc = set_lds(mlp_bwd_fast_group_kernel<HB_, HB_, NS_, LW_>, lds);                                                       \
        if (rc != 0) return rc;                                                                                                     \
        hipLaunchKernelGGL((mlp_bwd_fast_group_kernel<HB_, HB_, NS_, LW_>), dim3(G.first[n]), dim3(LW_ ? kLeafWaves * 64 : kBlockThreads), lds, stream, G); \
    } while (0)
#define NLAM_LAUNCH_BWDG(HB_, NS_)                            \
    do {                                                      \
        if (ps[0].flags & NLAM_F_LEAF_WGRAD) NLAM_LAUNCH_BWDG1(HB_, NS_, true); \
        else NLAM_LAUNCH_BWDG1(HB_, NS_, false);              \
    } while (0)
    if (HB == 1) {
        if (ns == 3) NLAM_LAUNCH_BWDG(1, 3);
        else if (ns == 2) NLAM_LAUNCH_BWDG(1, 2);
        else NLAM_LAUNCH_BWDG(1, 1);
    } else {
        if (ns == 3) NLAM_LAUNCH_BWDG(2, 3);
        else if (ns == 2) NLAM_LAUNCH_BWDG(2, 2);
        else NLAM_LAUNCH_BWDG(2, 1);
    }
    return (int32_t)hipGetLastError();
}
#endif

#if NLAM_IN_TU(2)
#define NLAM_LAUNCH_BWD(HB_, OB_)                                                                      \
    do {                                                                                               \
        const size_t lds = bwd_lds_bytes(p, HB_, OB_);                                                 \
        int rc = set_lds(mlp_bwd_kernel<HB_, OB_>, lds);                                               \
        if (rc != 0) return rc;                                                                        \
        hipLaunchKernelGGL((mlp_bwd_kernel<HB_, OB_>), dim3(blocks), dim3(kBlockThreads), lds, stream, *p); \
    } while (0)
int32_t nlam_detail::bwd_narrow(const nlam_mlp_bwd_t* p, hipStream_t stream) {
    const int blocks = grid_blocks((long)p->ntiles * p->batch);
    const int HB = (p->hid + 31) / 32, OB = (p->dout + 31) / 32;
    const bool ro = bwd_ragged_out(p);   // output_map: dout not a whole block, no LayerNorm -> split-bf16 fast kernel, dz2 padded to OB * 32 columns
    if ((ro ? OB * 32 : 0) != p->dz2_ld) return NLAM_EINVAL;   // the caller sized dz2 with nlam_mlp_bwd_dz2_ld
    bool fast = (p->hid % 32 == 0) && (p->dout % 32 == 0 || ro);
    for (int s = 0; s < p->nsrc; ++s)
        if (p->dmode[s] != 0) fast = fast && (p->src[s].width == 32 || p->src[s].width == 64);
    if ((p->flags & NLAM_F_ADD_SRC0) && p->dmode[0] != 0 && p->src[0].width != p->dout) fast = false;
    if (p->flags & NLAM_F_NO_ACT) fast = false;   // the generic kernel carries the activation switch
    if (p->flags & NLAM_F_PRE_ADD) {   // factorised edge MLP: FAST shapes and split-bf16 modes only
        bool ok = fast && ((p->flags & NLAM_F_MM_MASK) != 0) && (p->flags & NLAM_F_ADD_SRC1) == 0 && p->nsrc >= 2;
        ok = ok && (p->src[0].width == 32 || p->src[0].width == 64);
        for (int s = 1; s < p->nsrc; ++s) ok = ok && p->src[s].width == p->hid && (p->dmode[s] == 0 || p->dmode[s] == 2 || p->dmode[s] == 3);
        if (!ok) return NLAM_EUNSUP;
    }
    if (fast) {
        const int ns = (int)((p->flags & NLAM_F_MM_MASK) >> NLAM_F_MM_SHIFT);
#define NLAM_LAUNCH_BWDF1(HB_, OB_, NS_)                                                                            \
    do {                                                                                                            \
        const size_t lds = bwd_fast_lds_bytes(p, HB_, OB_, NS_);                                                    \
        int rc = set_lds(mlp_bwd_fast_kernel<HB_, OB_, NS_>, lds);                                                  \
        if (rc != 0) return rc;                                                                                     \
        hipLaunchKernelGGL((mlp_bwd_fast_kernel<HB_, OB_, NS_>), dim3(blocks), dim3(kBlockThreads), lds, stream, *p); \
    } while (0)
#define NLAM_LAUNCH_BWDF(HB_, OB_)                     \
    do {                                               \
        if (ns == 3) NLAM_LAUNCH_BWDF1(HB_, OB_, 3);      \
        else if (ns == 2) NLAM_LAUNCH_BWDF1(HB_, OB_, 2); \
        else if (ns == 1) NLAM_LAUNCH_BWDF1(HB_, OB_, 1); \
        else NLAM_LAUNCH_BWDF1(HB_, OB_, 0);              \
    } while (0)
#define NLAM_LAUNCH_BWDRO1(HB_, OB_, NS_)                                                                                  \
    do {                                                                                                                  \
        const size_t lds = bwd_fast_lds_bytes(p, HB_, OB_, NS_);                                                          \
        int rc = set_lds(mlp_bwd_fast_kernel<HB_, OB_, NS_, true>, lds);                                                  \
        if (rc != 0) return rc;                                                                                           \
        hipLaunchKernelGGL((mlp_bwd_fast_kernel<HB_, OB_, NS_, true>), dim3(blocks), dim3(kBlockThreads), lds, stream, *p); \
    } while (0)
#define NLAM_LAUNCH_BWDRO(HB_, OB_)                     \
    do {                                                \
        if (ns == 3) NLAM_LAUNCH_BWDRO1(HB_, OB_, 3);      \
        else if (ns == 2) NLAM_LAUNCH_BWDRO1(HB_, OB_, 2); \
        else NLAM_LAUNCH_BWDRO1(HB_, OB_, 1);              \
    } while (0)
        if (ro) {   // OB == 1 by construction (dout < 32 ... the only shape the reference has: output_map)
            if (HB == 1) NLAM_LAUNCH_BWDRO(1, 1);
            else NLAM_LAUNCH_BWDRO(2, 1);
            return (int32_t)hipGetLastError();
        }
        if (HB == 1 && OB == 1) NLAM_LAUNCH_BWDF(1, 1);
        else if (HB == 2 && OB == 1) NLAM_LAUNCH_BWDF(2, 1);
        else if (HB == 1 && OB == 2) NLAM_LAUNCH_BWDF(1, 2);
        else NLAM_LAUNCH_BWDF(2, 2);
        return (int32_t)hipGetLastError();
    }
    if (HB == 1 && OB == 1) NLAM_LAUNCH_BWD(1, 1);
    else if (HB == 2 && OB == 1) NLAM_LAUNCH_BWD(2, 1);
    else if (HB == 1 && OB == 2) NLAM_LAUNCH_BWD(1, 2);
    else if (HB == 2 && OB == 2) NLAM_LAUNCH_BWD(2, 2);
    else return NLAM_EUNSUP;
    return (int32_t)hipGetLastError();
}
#endif

#if NLAM_IN_TU(1)
extern "C" {

int32_t nlam_wgrad(const nlam_wgrad_t* p, void* hip_stream) {
    NLAM_RANGE("nlam_wgrad");
    if (p == nullptr || p->A == nullptr || p->partials == nullptr || p->nsrc < 1 || p->nsrc > NLAM_MAX_SRC) return NLAM_EINVAL;
    int n = 0;
    for (int s = 0; s < p->nsrc; ++s) n += p->src[s].width;
    if (n != p->n || p->m < 1 || p->nparts < 1) return NLAM_EINVAL;
    hipStream_t stream = (hipStream_t)hip_stream;
    if (!(p->nsrc == 1 && p->src[0].width <= kSmallN && p->m % 4 == 0) && wgrad_is_wide(p)) {
        if (wgrad_wbf_ns(p) > 0) return nlam_detail::wgrad_wbf(p, stream);   // split-bf16 matrix path (nlam_wbf.inc)
        if (p->flags & (NLAM_F_A_BF16 | NLAM_F_S_BF16)) return NLAM_EUNSUP;
        return nlam_detail::wgrad_wide(p, stream);
    }
    return nlam_detail::wgrad_narrow(p, stream);
}

// n <= NLAM_MAX_GROUP weight gradients of ONE shape (m, sources and their widths, flags) in one grid: member k's row slices are
// workgroups first[k] .. first[k + 1) of gridDim.x and land in its own `partials`.  Members of the split-bf16 family with fp32
// operands (NLAM_EUNSUP otherwise: launch them one by one).
int32_t nlam_wgrad_group(const nlam_wgrad_t* ps, int32_t n, void* hip_stream) {
    NLAM_RANGE("nlam_wgrad_group");
    if (ps == nullptr || n < 1 || n > NLAM_MAX_GROUP) return NLAM_EINVAL;
    for (int k = 0; k < n; ++k) {
        const nlam_wgrad_t* p = &ps[k];
        if (p->A == nullptr || p->partials == nullptr || p->nsrc < 1 || p->nsrc > NLAM_MAX_SRC) return NLAM_EINVAL;
        int w = 0;
        for (int s = 0; s < p->nsrc; ++s) w += p->src[s].width;
        if (w != p->n || p->m < 1 || p->nparts < 1 || p->nparts != nlam_wgrad_nparts(p)) return NLAM_EINVAL;
        if ((p->nsrc == 1 && p->src[0].width <= kSmallN && p->m % 4 == 0) || !wgrad_is_wide(p) || wgrad_wbf_ns(p) == 0) return NLAM_EUNSUP;
        if (p->flags & (NLAM_F_A_BF16 | NLAM_F_S_BF16)) return NLAM_EUNSUP;
        if (p->m != ps[0].m || p->n != ps[0].n || p->nsrc != ps[0].nsrc || p->flags != ps[0].flags || wgrad_wbf_big(p) != wgrad_wbf_big(&ps[0]))
            return NLAM_EUNSUP;
        for (int s = 0; s < p->nsrc; ++s)
            if (p->src[s].width != ps[0].src[s].width) return NLAM_EUNSUP;
    }
    return nlam_detail::wgrad_wbf_group(ps, n, (hipStream_t)hip_stream);
}

}  // extern "C"
#endif

#if NLAM_IN_TU(4)
int32_t nlam_detail::wgrad_wbf(const nlam_wgrad_t* p, hipStream_t stream) {
        const int wns = wgrad_wbf_ns(p);
        // 256 x 256 windows (8 waves) when the output has more than 128 rows, 128 x 128 windows (4 waves) otherwise
        const bool big = wgrad_wbf_big(p);
        const int winm = big ? 256 : 128, winn = big ? 256 : 128;
        const size_t lds = (size_t)2 * ((winm + winn) / 32) * wns * 1024;
        const dim3 grid(p->nparts, wgrad_windows_of(p, winm, winn));
#define NLAM_LAUNCH_WG_WBF(NS_, S_, WM_, WN_, NBW_)                                                                       \
    do {                                                                                                                  \
        int rc = set_lds(wgrad_wbf_kernel<NS_, S_, 1, WM_, WN_, NBW_>, lds);                                              \
        if (rc != 0) return rc;                                                                                           \
        hipLaunchKernelGGL((wgrad_wbf_kernel<NS_, S_, 1, WM_, WN_, NBW_>), grid, dim3(WM_ * WN_ * 64), lds, stream, *p);  \
    } while (0)
#define NLAM_LAUNCH_WG_WBF2(NS_, S_)                 \
    do {                                             \
        if (big) NLAM_LAUNCH_WG_WBF(NS_, S_, 4, 2, 4); \
        else NLAM_LAUNCH_WG_WBF(NS_, S_, 2, 2, 2);     \
    } while (0)
        const bool silu = (p->flags & NLAM_F_SILU_B) != 0;
        if (p->flags & (NLAM_F_A_BF16 | NLAM_F_S_BF16)) {   // operands stored as bf16 (layers running with NLAM_F_STORE_BF16)
            // dW1 = dz1^T [fp32 sources]: A bf16; dW2 = dz2^T silu(z1): A and the single un-gathered source bf16
            const bool sb = (p->flags & NLAM_F_S_BF16) != 0;
            if (wns != 1 || (p->flags & NLAM_F_A_BF16) == 0 || (sb && (p->nsrc != 1 || p->src[0].idx != nullptr || !silu)) || (!sb && silu))
                return NLAM_EUNSUP;
#define NLAM_LAUNCH_WG_WBF_B(S_, WM_, WN_, NBW_, SB_)                                                                             \
    do {                                                                                                                          \
        int rc = set_lds(wgrad_wbf_kernel<1, S_, 1, WM_, WN_, NBW_, true, SB_>, lds);                                              \
        if (rc != 0) return rc;                                                                                                   \
        hipLaunchKernelGGL((wgrad_wbf_kernel<1, S_, 1, WM_, WN_, NBW_, true, SB_>), grid, dim3(WM_ * WN_ * 64), lds, stream, *p);  \
    } while (0)
#define NLAM_LAUNCH_WG_LDMA(ABF_, SBF_, SILU_, R_, NB_)                                                                   \
    do {                                                                                                                  \
        const size_t lds2 = WgLdma<ABF_, SBF_, SILU_, R_, NB_>::LDS;                                                       \
        int rc = set_lds(wgrad_ldma_kernel<1, ABF_, SBF_, SILU_, R_, NB_>, lds2);                                          \
        if (rc != 0) return rc;                                                                                           \
        hipLaunchKernelGGL((wgrad_ldma_kernel<1, ABF_, SBF_, SILU_, R_, NB_>), grid, dim3(512), lds2, stream, *p);         \
    } while (0)
            const int var = nlam_detail::wgrad_ldma_var;   // (rows per stage, ring depth) variants for A/B runs (NLAM_TUNE_WGRAD_LDMA_VAR)
            if (sb) {
                if (big && (nlam_detail::wgrad_ldma & 1)) {
                    if (var == 1) NLAM_LAUNCH_WG_LDMA(true, true, true, 32, 5);
                    else if (var == 2) NLAM_LAUNCH_WG_LDMA(true, true, true, 16, 6);
                    else if (var == 3) NLAM_LAUNCH_WG_LDMA(true, true, true, 16, 8);
                    else NLAM_LAUNCH_WG_LDMA(true, true, true, 32, 4);
                } else if (big) NLAM_LAUNCH_WG_WBF_B(true, 4, 2, 4, true);
                else NLAM_LAUNCH_WG_WBF_B(true, 2, 2, 2, true);
            } else {
                if (big && (nlam_detail::wgrad_ldma & 1)) {
                    if (var == 1) NLAM_LAUNCH_WG_LDMA(true, false, false, 16, 4);
                    else if (var == 2) NLAM_LAUNCH_WG_LDMA(true, false, false, 16, 5);
                    else if (var == 3) NLAM_LAUNCH_WG_LDMA(true, false, false, 16, 6);
                    else NLAM_LAUNCH_WG_LDMA(true, false, false, 32, 3);
                } else if (big) NLAM_LAUNCH_WG_WBF_B(false, 4, 2, 4, false);
                else NLAM_LAUNCH_WG_WBF_B(false, 2, 2, 2, false);
            }
            return (int32_t)hipGetLastError();
        }
        if (wns == 1 && big && (nlam_detail::wgrad_ldma & 2)) {   // fp32 operands, one term (autocast launches without bf16 storage)
            const int var = nlam_detail::wgrad_ldma_var;
            if (silu) {
                if (var >= 1) NLAM_LAUNCH_WG_LDMA(false, false, true, 16, 5);
                else NLAM_LAUNCH_WG_LDMA(false, false, true, 16, 4);
            } else {
                if (var == 1) NLAM_LAUNCH_WG_LDMA(false, false, false, 16, 4);
                else if (var >= 2) NLAM_LAUNCH_WG_LDMA(false, false, false, 16, 5);
                else NLAM_LAUNCH_WG_LDMA(false, false, false, 16, 3);
            }
            return (int32_t)hipGetLastError();
        }
        if (wns == 3 && big && (nlam_detail::wgrad_ldma & 4)) {   // fp32 class: three terms on the LDS-DMA kernel
#define NLAM_LAUNCH_WG_LDMA3(SILU_, NB_)                                                                                  \
    do {                                                                                                                  \
        const size_t lds2 = WgLdma<false, false, SILU_, 16, NB_>::LDS;                                                     \
        int rc = set_lds(wgrad_ldma_kernel<3, false, false, SILU_, 16, NB_>, lds2);                                        \
        if (rc != 0) return rc;                                                                                           \
        hipLaunchKernelGGL((wgrad_ldma_kernel<3, false, false, SILU_, 16, NB_>), grid, dim3(512), lds2, stream, *p);       \
    } while (0)
            if (silu) NLAM_LAUNCH_WG_LDMA3(true, 4);
            else NLAM_LAUNCH_WG_LDMA3(false, 4);
            return (int32_t)hipGetLastError();
        }
        if (wns == 1 && silu) NLAM_LAUNCH_WG_WBF2(1, true);
        else if (wns == 1) NLAM_LAUNCH_WG_WBF2(1, false);
        else if (silu) NLAM_LAUNCH_WG_WBF2(3, true);
        else NLAM_LAUNCH_WG_WBF2(3, false);
        return (int32_t)hipGetLastError();
    }
#endif

#if NLAM_IN_TU(4)
// members checked by nlam_wgrad_group: split-bf16 family, fp32 operands, one shape (m, sources, flags, window size)
int32_t nlam_detail::wgrad_wbf_group(const nlam_wgrad_t* ps, int n, hipStream_t stream) {
    const nlam_wgrad_t* p = &ps[0];
    const int wns = wgrad_wbf_ns(p);
    const bool big = wgrad_wbf_big(p);
    const int winm = big ? 256 : 128, winn = big ? 256 : 128;
    const size_t lds = (size_t)2 * ((winm + winn) / 32) * wns * 1024;
    wgrad_group_t G;
    G.n = n;
    G.first[0] = 0;
    for (int k = 0; k < n; ++k) {
        G.g[k] = ps[k];
        G.first[k + 1] = G.first[k] + ps[k].nparts;
    }
    for (int k = n + 1; k <= NLAM_MAX_GROUP; ++k) G.first[k] = G.first[n];
    const dim3 grid(G.first[n], wgrad_windows_of(p, winm, winn));
#define NLAM_LAUNCH_WG_WBF_G(NS_, S_, WM_, WN_, NBW_)                                                                       \
    do {                                                                                                                    \
        int rc = set_lds(wgrad_wbf_group_kernel<NS_, S_, WM_, WN_, NBW_>, lds);                                             \
        if (rc != 0) return rc;                                                                                             \
        hipLaunchKernelGGL((wgrad_wbf_group_kernel<NS_, S_, WM_, WN_, NBW_>), grid, dim3(WM_ * WN_ * 64), lds, stream, G);  \
    } while (0)
#define NLAM_LAUNCH_WG_WBF_G2(NS_, S_)                   \
    do {                                                 \
        if (big) NLAM_LAUNCH_WG_WBF_G(NS_, S_, 4, 2, 4); \
        else NLAM_LAUNCH_WG_WBF_G(NS_, S_, 2, 2, 2);     \
    } while (0)
    const bool silu = (p->flags & NLAM_F_SILU_B) != 0;
    if (wns == 1 && silu) NLAM_LAUNCH_WG_WBF_G2(1, true);
    else if (wns == 1) NLAM_LAUNCH_WG_WBF_G2(1, false);
    else if (silu) NLAM_LAUNCH_WG_WBF_G2(3, true);
    else NLAM_LAUNCH_WG_WBF_G2(3, false);
    return (int32_t)hipGetLastError();
}
#endif

#if NLAM_IN_TU(3)
int32_t nlam_detail::wgrad_wide(const nlam_wgrad_t* p, hipStream_t stream) {
        const size_t lds = (size_t)4 * kWWTile * sizeof(float);
        if (p->flags & NLAM_F_SILU_B) {
            int rc = set_lds(wgrad_wide_kernel<true>, lds);
            if (rc != 0) return rc;
            hipLaunchKernelGGL(wgrad_wide_kernel<true>, dim3(p->nparts, wgrad_windows(p)), dim3(kWgradThreads), lds, stream, *p);
        } else {
            int rc = set_lds(wgrad_wide_kernel<false>, lds);
            if (rc != 0) return rc;
            hipLaunchKernelGGL(wgrad_wide_kernel<false>, dim3(p->nparts, wgrad_windows(p)), dim3(kWgradThreads), lds, stream, *p);
        }
        return (int32_t)hipGetLastError();
    }
#endif

#if NLAM_IN_TU(2)
int32_t nlam_detail::wgrad_narrow(const nlam_wgrad_t* p, hipStream_t stream) {
    if (p->nsrc == 1 && p->src[0].width <= kSmallN && p->m % 4 == 0) {
        hipLaunchKernelGGL(wgrad_smalln_kernel, dim3(p->nparts), dim3(256), 0, stream, *p);
        return (int32_t)hipGetLastError();
    }
    const bool dma = wgrad_is_narrow_dma(p);
    int nb_total = 0;
    for (int s = 0; s < p->nsrc; ++s) nb_total += (p->src[s].width + 31) / 32;
    if (dma) {
        const int nblocks = ((p->m + 31) / 32) * nb_total;
        const int nbw = (nblocks + 3) / 4;
        const size_t lds = (size_t)2 * (1 + p->nsrc) * kWgTile * sizeof(float);
#define NLAM_LAUNCH_WGD1(N_, S_)                                                                              \
    do {                                                                                                      \
        int rc = set_lds(wgrad_dma_kernel<N_, S_>, lds);                                                      \
        if (rc != 0) return rc;                                                                               \
        hipLaunchKernelGGL((wgrad_dma_kernel<N_, S_>), dim3(p->nparts), dim3(kWgradThreads), lds, stream, *p); \
    } while (0)
#define NLAM_LAUNCH_WGD(N_)                                        \
    do {                                                           \
        if (p->flags & NLAM_F_SILU_B) NLAM_LAUNCH_WGD1(N_, true);  \
        else NLAM_LAUNCH_WGD1(N_, false);                          \
    } while (0)
        if (nbw <= 1) NLAM_LAUNCH_WGD(1);
        else if (nbw <= 2) NLAM_LAUNCH_WGD(2);
        else if (nbw <= 3) NLAM_LAUNCH_WGD(3);
        else return NLAM_EUNSUP;
        return (int32_t)hipGetLastError();
    }
    const int MP = (p->m + 31) / 32 * 32, NP = (p->n + 31) / 32 * 32;
    const int nblocks = (MP / 32) * (NP / 32);
    const size_t lds = (size_t)kWgradRows * (MP + 4 + NP + 4) * sizeof(float);
    const int nbw = (nblocks + 3) / 4;
    const int ywin = (nblocks + 11) / 12;   // blockIdx.y windows of 12 blocks when one workgroup cannot hold them all
#define NLAM_LAUNCH_WG(N_, Y_)                                                                                \
    do {                                                                                                      \
        int rc = set_lds(wgrad_kernel<N_>, lds);                                                              \
        if (rc != 0) return rc;                                                                               \
        hipLaunchKernelGGL((wgrad_kernel<N_>), dim3(p->nparts, Y_), dim3(kWgradThreads), lds, stream, *p);    \
    } while (0)
    if (nbw <= 1) NLAM_LAUNCH_WG(1, 1);
    else if (nbw <= 2) NLAM_LAUNCH_WG(2, 1);
    else if (nbw <= 3) NLAM_LAUNCH_WG(3, 1);
    else NLAM_LAUNCH_WG(3, ywin);
    return (int32_t)hipGetLastError();
}
#endif

#if NLAM_IN_TU(5)
extern "C" {

int64_t nlam_window_len(int64_t n_state_times, int64_t n_forcing_times, int32_t ar_steps, int32_t past, int32_t future) {
    const int64_t window = (int64_t)(past > 2 ? past : 2) + ar_steps + future;
    int64_t n = n_state_times - window + 1;
    if (n_forcing_times >= 0 && n_forcing_times - window + 1 < n) n = n_forcing_times - window + 1;
    return n > 0 ? n : 0;
}

int32_t nlam_window_batch(const nlam_window_t* p, void* hip_stream) {
    NLAM_RANGE("nlam_window_batch");
    if (p == nullptr || p->state == nullptr || p->sample_idx == nullptr || p->init_states == nullptr || p->target_states == nullptr)
        return NLAM_EINVAL;
    if (p->batch < 0 || p->nodes < 0 || p->d_state < 1 || p->d_forcing < 0 || p->ar_steps < 1 || p->ar_steps > 256 ||
        p->num_past_forcing_steps < 0 || p->num_future_forcing_steps < 0 || p->n_times < 1)
        return NLAM_EINVAL;
    if (p->d_forcing > 0 && (p->forcing == nullptr || p->forcing_windowed == nullptr)) return NLAM_EINVAL;
    if ((p->state_mean == nullptr) != (p->state_std == nullptr) || (p->forcing_mean == nullptr) != (p->forcing_std == nullptr)) return NLAM_EINVAL;
    if (p->state_mean != nullptr && p->d_forcing > 0 && p->forcing_mean == nullptr) return NLAM_EINVAL;   // all or nothing, as the reference's hook
    if (nlam_window_len(p->n_times, p->d_forcing > 0 ? p->n_times : -1, p->ar_steps, p->num_past_forcing_steps, p->num_future_forcing_steps) < 1)
        return NLAM_EINVAL;   // the series is shorter than one sample
    if (p->batch == 0 || p->nodes == 0) return 0;
    const long window = p->num_past_forcing_steps + p->num_future_forcing_steps + 1;
    const long e_state = (long)p->nodes * p->d_state;                       // one row of the launch (blockIdx.y)
    const long e_forc = (long)p->nodes * p->d_forcing * window;
    if (e_state >= (1L << 31) || e_forc >= (1L << 31) || p->batch > 65535) return NLAM_EUNSUP;   // 32-bit offsets inside a row
    long blocks = ((e_state > e_forc ? e_state : e_forc) + 2047) / 2048;    // four elements per thread, two rounds
    if (blocks < 1) blocks = 1;
    if (blocks > 512) blocks = 512;
    const int rows = 2 + p->ar_steps + (p->d_forcing > 0 ? p->ar_steps : 0);
    hipLaunchKernelGGL(window_batch_kernel, dim3((int)blocks, rows, p->batch), dim3(256), 0, (hipStream_t)hip_stream, *p);
    return (int32_t)hipGetLastError();
}

int32_t nlam_pre_add_supported(const nlam_mlp_fwd_t* p) {
    /* would nlam_mlp_fwd (and the matching nlam_mlp_bwd) run this NLAM_F_PRE_ADD problem on a factorised kernel?
       Only shapes, flags, ntiles and batch are read. */
    if (p == nullptr || (p->flags & NLAM_F_PRE_ADD) == 0 || p->nsrc < 2 || (p->flags & NLAM_F_ADD_SRC1)) return 0;
    const int ns = (int)((p->flags & NLAM_F_MM_MASK) >> NLAM_F_MM_SHIFT);
    if (ns == 0 || p->hid % 32 != 0 || p->dout % 32 != 0 || p->src[0].width % 32 != 0) return 0;
    for (int s = 1; s < p->nsrc; ++s)
        if (p->src[s].width != p->hid) return 0;
    if (!fwd_is_wide(p)) return p->hid == p->dout && (p->src[0].width == 32 || p->src[0].width == 64) ? 1 : 0;
    if (fwd_wbf_ns(p) == 0) return 0;
    nlam_mlp_bwd_t q = {};
    q.nsrc = p->nsrc;
    for (int s = 0; s < p->nsrc; ++s) {
        q.src[s] = p->src[s];
        q.dmode[s] = s == 0 ? 1 : (s == 1 ? 0 : 3);
    }
    q.batch = p->batch, q.rows = p->rows, q.ntiles = p->ntiles, q.hid = p->hid, q.dout = p->dout, q.flags = p->flags;
    if (bwd_wbf_ns(&q) == 0) return 0;
    const int k = p->src[0].width;   // node-level products: k = input width, n = hid (nlam_linear)
    return (k % 64 == 0 && p->hid % 64 == 0) ? 1 : 0;
}

// operand layout the LDS-tiled GEMM can read: 1 = along K (ldk == 1, 16-byte aligned rows: the forward product), 2 = transposed
// (ldn == 1: the data-gradient product), 0 = neither (the strip kernel reads any strides)
static int lin_gemm_layout(const nlam_linear_t* p) {
    auto ok = [&](const float* W) {
        if (p->ldk == 1) return (p->ldn % 4 == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0) ? 1 : 0;
        return p->ldn == 1 ? 2 : 0;
    };
    const int a = ok(p->W);
    if (p->W2 != nullptr && ok(p->W2) != a) return 0;
    return a;
}

int32_t nlam_linear(const nlam_linear_t* p, void* hip_stream) {
    NLAM_RANGE("nlam_linear");
    if (p == nullptr || p->x == nullptr || p->W == nullptr || p->out == nullptr || p->rows < 0 || p->k < 1 || p->n < 1) return NLAM_EINVAL;
    if (p->rows == 0) return 0;
    const int ns = (int)((p->flags & NLAM_F_MM_MASK) >> NLAM_F_MM_SHIFT);
    if (ns == 0 || p->k % 32 != 0 || p->n % 32 != 0) return NLAM_EUNSUP;
    if (((reinterpret_cast<uintptr_t>(p->x) | reinterpret_cast<uintptr_t>(p->out)) & 15) != 0) return NLAM_EINVAL;
    if ((p->W2 == nullptr) != (p->out2 == nullptr)) return NLAM_EINVAL;
    if (p->W2 != nullptr && (p->k <= 64 && p->n <= 64)) return NLAM_EUNSUP;   // the dual form exists for the wide kernel only
    const int KB = p->k / 32, MB = p->n / 32;
    const long ntiles = (p->rows + 31) / 32;
    const int blocks = (int)(ntiles < kMaxGridBlocks ? ntiles : kMaxGridBlocks);
    const size_t lds = (size_t)ns * MB * 2 * KB * 64 * 16 + (size_t)kFwdWaves * 32 * kStgStride * sizeof(float);
    hipStream_t stream = (hipStream_t)hip_stream;
#define NLAM_LAUNCH_LIN1(KB_, MB_, NS_)                                                                          \
    do {                                                                                                        \
        int rc = set_lds(linear_bf_kernel<KB_, MB_, NS_>, lds);                                                 \
        if (rc != 0) return rc;                                                                                 \
        hipLaunchKernelGGL((linear_bf_kernel<KB_, MB_, NS_>), dim3(blocks), dim3(kFwdThreads), lds, stream, *p); \
    } while (0)
#define NLAM_LAUNCH_LIN(KB_, MB_)                         \
    do {                                                  \
        if (ns == 3) NLAM_LAUNCH_LIN1(KB_, MB_, 3);       \
        else if (ns == 2) NLAM_LAUNCH_LIN1(KB_, MB_, 2);  \
        else NLAM_LAUNCH_LIN1(KB_, MB_, 1);               \
    } while (0)
    if (KB == 1 && MB == 1) NLAM_LAUNCH_LIN(1, 1);
    else if (KB == 2 && MB == 1) NLAM_LAUNCH_LIN(2, 1);
    else if (KB == 1 && MB == 2) NLAM_LAUNCH_LIN(1, 2);
    else if (KB == 2 && MB == 2) NLAM_LAUNCH_LIN(2, 2);
    else if (nlam_detail::lin_gemm != 0 && p->n % 128 == 0 && p->k <= kMaxWide && p->n <= kMaxWide && lin_gemm_layout(p) != 0 &&
             // where it wins (profiles/round5/linear_bench.log, isolated launches): one term everywhere (1.3-2.1x); two / three terms
             // -- 114 KB of LDS, one workgroup per CU -- on the mesh-level products up to K = 256 (1.4-1.5x) and on the grid-level
             // ones from K = 512 (1.1x); the strip kernel keeps three-term 63 784 x 256 (0.9x) and 6 561 x 512 (0.9x).
             // nlam_set_tuning(NLAM_TUNE_LIN_GEMM, 2) forces it everywhere it applies (tests).
             (ns == 1 || nlam_detail::lin_gemm == 2 ||
              (p->rows >= nlam_detail::lin_gemm_big_rows ? p->k >= 512 : p->k <= 256))) {
        // LDS-tiled GEMM (linear_gemm_kernel): 128 x 128 tiles from 32 768 rows, 64-row tiles below
        const bool ta = lin_gemm_layout(p) == 2;
        const bool big = p->rows >= nlam_detail::lin_gemm_big_rows;
        const long bm = big ? 128 : 64;
        const long nrt = (p->rows + bm - 1) / bm;
        const long nftt = (long)(p->n / 128) * (p->W2 != nullptr ? 2 : 1);
        const long nwg = (nrt + 7) / 8 * 8 * nftt;
        if (nwg > 0x7fffffffL) return NLAM_EUNSUP;
#define NLAM_LAUNCH_LING1(NS_, WN_, TA_, SK_)                                                                        \
    do {                                                                                                             \
        const size_t glds = lin_gemm_lds_bytes<NS_, WN_, SK_>();                                                     \
        int rc = set_lds(linear_gemm_kernel<NS_, WN_, TA_, SK_>, glds);                                              \
        if (rc != 0) return rc;                                                                                      \
        hipLaunchKernelGGL((linear_gemm_kernel<NS_, WN_, TA_, SK_>), dim3((unsigned)nwg), dim3(kLinGemmThreads), glds, stream, *p); \
    } while (0)
#define NLAM_LAUNCH_LING(NS_, WN_)                       \
    do {                                                 \
        if (ta) NLAM_LAUNCH_LING1(NS_, WN_, true, 2);    \
        else NLAM_LAUNCH_LING1(NS_, WN_, false, 2);      \
    } while (0)
        if (big) {
            if (ns == 3) NLAM_LAUNCH_LING(3, 2);
            else if (ns == 2) NLAM_LAUNCH_LING(2, 2);
            else NLAM_LAUNCH_LING(1, 2);
        } else {
            if (ns == 3) NLAM_LAUNCH_LING(3, 1);
            else if (ns == 2) NLAM_LAUNCH_LING(2, 1);
            else if (p->k % 64 == 0 && NLAM_LIN_SK == 4) {   // one term, 64-row tiles: 64-column chunks (66 KB of LDS, two workgroups per CU)
                if (ta) NLAM_LAUNCH_LING1(1, 1, true, 4);
                else NLAM_LAUNCH_LING1(1, 1, false, 4);
            } else NLAM_LAUNCH_LING(1, 1);
        }
    } else if (p->k % 64 == 0 && p->n % 64 == 0 && p->k <= kMaxWide && p->n <= kMaxWide) {
        // 64 x 64 weight blocks streamed through two LDS buffers (linear_bfw_kernel)
        const int kc_all = p->k / 64;
        const bool resident = kc_all <= kLinResidentChunks && nlam_detail::lin_resident_wgs > 0;   // the output pair's whole weight strip stays in LDS
        nlam_linear_t q = *p;
        q.flags = (q.flags & ~kLinResidentFlag) | (resident ? kLinResidentFlag : 0u);
        const size_t wlds = (size_t)(resident ? kc_all : 2) * ns * 2 * 4 * 64 * 16 + (size_t)kFwdWaves * 32 * kStgStride * sizeof(float);
        const int wby = (p->n / 64) * (p->W2 != nullptr ? 2 : 1);
        long wbx = (ntiles + kFwdWaves - 1) / kFwdWaves;   // eight tiles (one per wave) share each staged weight block
        // resident weights: one workgroup per CU (135 KB of LDS), each walking several rounds of tiles over its one staging
        const long cap = resident ? nlam_detail::lin_resident_wgs : 4 * kNumCUs;
        if (wbx * wby > cap) wbx = (cap + wby - 1) / wby;
        if (wbx < 1) wbx = 1;
#define NLAM_LAUNCH_LINW(NS_)                                                                              \
    do {                                                                                                   \
        int rc = set_lds(linear_bfw_kernel<NS_>, wlds);                                                    \
        if (rc != 0) return rc;                                                                            \
        hipLaunchKernelGGL((linear_bfw_kernel<NS_>), dim3(wbx, wby), dim3(kFwdThreads), wlds, stream, q);  \
    } while (0)
        if (ns == 3) NLAM_LAUNCH_LINW(3);
        else if (ns == 2) NLAM_LAUNCH_LINW(2);
        else NLAM_LAUNCH_LINW(1);
    } else return NLAM_EUNSUP;
    return (int32_t)hipGetLastError();
}

}  // extern "C"
#endif

#if NLAM_IN_TU(1)
extern "C" {

// the same segment sum over rows stored as bf16 (the dz1 rows of a layer running with NLAM_F_STORE_BF16: gradient of a
// sender-gathered addend of the factorised edge MLP); width % 8 == 0: thread -> (batch, segment, 8 columns), 16-byte loads, 8 rows in
// flight, fp32 accumulation in a fixed order
__global__ void segment_sum_bf16_kernel(const unsigned short* in, long in_bstride, const int32_t* ptr, const int32_t* order,
                                        const float* scale, float* out, int nseg, int width, int batch) {
    const int w8 = width >> 3;
    const long total = (long)batch * nseg * w8;
    for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(gid % w8);
        const long rest = gid / w8;
        const int sgm = (int)(rest % nseg);
        const int b = (int)(rest / nseg);
        const int lo = ptr[sgm], hi_ = ptr[sgm + 1];
        const unsigned short* base = in + (long)b * in_bstride + 8 * c8;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        auto add = [&](const u32x4 v) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc[2 * k] += __builtin_bit_cast(float, v[k] << 16);
                acc[2 * k + 1] += __builtin_bit_cast(float, v[k] & 0xffff0000u);
            }
        };
        int q = lo;
        for (; q + 8 <= hi_; q += 8) {
            long rw[8];   // ids first, then rows (see segment_sum_kernel)
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) rw[u] = order != nullptr ? order[q + u] : q + u;
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const u32x4*>(base + rw[u] * width);
#pragma unroll
            for (int u = 0; u < 8; ++u) add(v[u]);
        }
        if (q < hi_) {   // the last 1 .. 7 rows: in flight together, added in row order (see segment_sum_kernel)
            long rw[8];
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                const int qq = min(q + u, hi_ - 1);
                rw[u] = order != nullptr ? order[qq] : qq;
            }
#pragma unroll
            for (int u = 0; u < 7; ++u) v[u] = *reinterpret_cast<const u32x4*>(base + rw[u] * width);
            asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]));
#pragma unroll
            for (int u = 0; u < 7; ++u)
                if (q + u < hi_) add(v[u]);
        }
        const float sc = scale != nullptr ? scale[sgm] : 1.f;
        float* o = out + ((size_t)b * nseg + sgm) * width + 8 * c8;
        *reinterpret_cast<f32x4*>(o) = f32x4{acc[0] * sc, acc[1] * sc, acc[2] * sc, acc[3] * sc};
        *reinterpret_cast<f32x4*>(o + 4) = f32x4{acc[4] * sc, acc[5] * sc, acc[6] * sc, acc[7] * sc};
    }
}

static int32_t segment_sum_launch(const float* in, int64_t in_bstride, const int32_t* ptr, const int32_t* order, const float* scale,
                                  float* out, int32_t nseg, int32_t width, int32_t batch, int accumulate, void* hip_stream,
                                  const float* extra = nullptr) {
    if (in == nullptr || ptr == nullptr || out == nullptr || nseg < 0 || width < 1 || batch < 1) return NLAM_EINVAL;
    if (nseg == 0) return 0;
    // segments of 16+ rows on average (known when the input has its own batch stride): several lane groups per segment
    const int w4 = width / 4;
    const long rows = in_bstride > 0 ? in_bstride / width : 0;
    if ((width & 3) == 0 && w4 <= 32 && 64 % w4 == 0 && rows >= 16L * nseg) {
        const int S = w4 <= 16 ? 4 : 2;
        const long tot = (long)batch * nseg * S * w4;
        long blk = (tot + 255) / 256;
        if (blk > 256 * 16) blk = 256 * 16;
        if (S == 4)
            hipLaunchKernelGGL(segment_sum_split_kernel<4>, dim3((int)blk), dim3(256), 0, (hipStream_t)hip_stream, in, (long)in_bstride, ptr,
                               order, scale, out, nseg, width, batch, accumulate, extra);
        else
            hipLaunchKernelGGL(segment_sum_split_kernel<2>, dim3((int)blk), dim3(256), 0, (hipStream_t)hip_stream, in, (long)in_bstride, ptr,
                               order, scale, out, nseg, width, batch, accumulate, extra);
        return (int32_t)hipGetLastError();
    }
    const long total = (long)batch * nseg * ((width + 3) / 4);
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(segment_sum_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)hip_stream, in, (long)in_bstride,
                       ptr, order, scale, out, nseg, width, batch, accumulate, extra);
    return (int32_t)hipGetLastError();
}

int32_t nlam_segment_sum(const float* in, int64_t in_bstride, const int32_t* ptr, const int32_t* order, const float* scale,
                         float* out, int32_t nseg, int32_t width, int32_t batch, void* hip_stream) {
    NLAM_RANGE("nlam_segment_sum");
    return segment_sum_launch(in, in_bstride, ptr, order, scale, out, nseg, width, batch, 0, hip_stream);
}

int32_t nlam_segment_sum_bf16(const uint16_t* in, int64_t in_bstride, const int32_t* ptr, const int32_t* order, const float* scale,
                              float* out, int32_t nseg, int32_t width, int32_t batch, void* hip_stream) {
    NLAM_RANGE("nlam_segment_sum_bf16");
    if (in == nullptr || ptr == nullptr || out == nullptr || nseg < 0 || width < 8 || batch < 1) return NLAM_EINVAL;
    if (width % 8 != 0) return NLAM_EUNSUP;
    if (nseg == 0) return 0;
    const long total = (long)batch * nseg * (width / 8);
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(segment_sum_bf16_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)hip_stream, in, (long)in_bstride, ptr,
                       order, scale, out, nseg, width, batch);
    return (int32_t)hipGetLastError();
}

int32_t nlam_segment_sum_acc(const float* in, int64_t in_bstride, const int32_t* ptr, const int32_t* order, const float* scale,
                             float* out, int32_t nseg, int32_t width, int32_t batch, void* hip_stream) {
    NLAM_RANGE("nlam_segment_sum_acc");
    return segment_sum_launch(in, in_bstride, ptr, order, scale, out, nseg, width, batch, 1, hip_stream);
}

int32_t nlam_segment_sum_add(const float* in, int64_t in_bstride, const int32_t* ptr, const int32_t* order, const float* scale,
                             const float* extra, float* out, int32_t nseg, int32_t width, int32_t batch, void* hip_stream) {
    NLAM_RANGE("nlam_segment_sum_add");
    if (extra == nullptr) return NLAM_EINVAL;
    return segment_sum_launch(in, in_bstride, ptr, order, scale, out, nseg, width, batch, 1, hip_stream, extra);
}

int32_t nlam_split_combine(float* buf, int64_t bstride, const int32_t* ptr, const int32_t* src, const int32_t* dst, int32_t n,
                           int32_t width, int32_t batch, void* hip_stream) {
    NLAM_RANGE("nlam_split_combine");
    if (buf == nullptr || ptr == nullptr || src == nullptr || dst == nullptr || n < 0 || width < 1 || batch < 1) return NLAM_EINVAL;
    if (n == 0) return 0;
    if (batch > 65535) return NLAM_EUNSUP;
    const int threads = width >= 256 ? 256 : (width >= 128 ? 128 : 64);
    hipLaunchKernelGGL(split_combine_kernel, dim3(n, batch), dim3(threads), 0, (hipStream_t)hip_stream, buf, (long)bstride, ptr, src,
                       dst, width);
    return (int32_t)hipGetLastError();
}

int32_t nlam_reduce_partials(const float* partials, int32_t nparts, int64_t stride, int32_t n, float* out,
                             int32_t accumulate, void* hip_stream) {
    NLAM_RANGE("nlam_reduce_partials");
    if (partials == nullptr || out == nullptr || nparts < 1 || n < 1) return NLAM_EINVAL;
    int blocks = (n + 63) / 64;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)hip_stream, partials, nparts,
                       (long)stride, n, out, accumulate);
    return (int32_t)hipGetLastError();
}

int32_t nlam_reduce_jobs(const nlam_reduce_jobs_t* jobs, void* hip_stream) {
    NLAM_RANGE("nlam_reduce_jobs");
    if (jobs == nullptr || jobs->njobs < 1 || jobs->njobs > NLAM_MAX_REDUCE_JOBS) return NLAM_EINVAL;
    int nmax = 0;
    for (int k = 0; k < jobs->njobs; ++k) {
        const nlam_reduce_job_t& j = jobs->job[k];
        if (j.partials == nullptr || j.out == nullptr || j.nparts < 1 || j.n < 1) return NLAM_EINVAL;
        if (j.ncols < 0 || (j.ncols > 0 && (j.ld < j.ncols || j.n % j.ncols != 0))) return NLAM_EINVAL;
        if (j.n > nmax) nmax = j.n;
    }
    int blocks = (nmax + 255) / 256;   // 256 elements per block and pass on the 16-B path (the scalar path strides by gridDim.x * 64)
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(reduce_jobs_kernel, dim3(blocks, jobs->njobs), dim3(kRedWaves * 64), 0, (hipStream_t)hip_stream, *jobs);
    return (int32_t)hipGetLastError();
}

int32_t nlam_wmse_fwd(const float* pred, const float* target, const float* inv_var, const float* row_weight, int64_t rows,
                      int32_t nodes, int32_t nvars, float scale, float* partials, int32_t nparts, void* hip_stream) {
    NLAM_RANGE("nlam_wmse_fwd");
    if (pred == nullptr || target == nullptr || inv_var == nullptr || row_weight == nullptr || partials == nullptr) return NLAM_EINVAL;
    if (rows < 1 || nodes < 1 || nvars < 1 || nparts < 1) return NLAM_EINVAL;
    hipLaunchKernelGGL(wmse_fwd_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)hip_stream, pred, target, inv_var, row_weight,
                       (long)rows * nvars, nodes, nvars, scale, partials);
    return (int32_t)hipGetLastError();
}

int32_t nlam_wmse_bwd(const float* pred, const float* target, const float* inv_var, const float* row_weight, const float* gscalar,
                      int64_t rows, int32_t nodes, int32_t nvars, float scale, float* dpred, void* hip_stream) {
    NLAM_RANGE("nlam_wmse_bwd");
    if (pred == nullptr || target == nullptr || inv_var == nullptr || row_weight == nullptr || gscalar == nullptr || dpred == nullptr)
        return NLAM_EINVAL;
    if (rows < 1 || nodes < 1 || nvars < 1) return NLAM_EINVAL;
    const long total = (long)rows * nvars;
    long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wmse_bwd_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)hip_stream, pred, target, inv_var, row_weight,
                       gscalar, total, nodes, nvars, scale, dpred);
    return (int32_t)hipGetLastError();
}

int32_t nlam_affine_mix(const float* x, const float* a, const float* y, const float* c, const float* z, const float* s,
                        const float* m, float* out, int64_t rows, int32_t nodes, int32_t width, void* hip_stream) {
    NLAM_RANGE("nlam_affine_mix");
    if (out == nullptr || rows < 1 || nodes < 1 || width < 1) return NLAM_EINVAL;
    if ((x == nullptr) != (a == nullptr) || (z == nullptr) != (s == nullptr)) return NLAM_EINVAL;
    if (x == nullptr && y == nullptr && z == nullptr && m == nullptr) return NLAM_EINVAL;
    const long total = (long)rows * width;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(affine_mix_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)hip_stream, x, a, y, c, z, s, m, out, total,
                       nodes, width);
    return (int32_t)hipGetLastError();
}

int32_t nlam_step_tail_fwd(const float* delta, const float* prev, const float* truth, const float* target, const float* dstd,
                           const float* dmean, const float* bmask, const float* inv_var, const float* row_weight, float scale,
                           float* pred, float* partials, int32_t nparts, int64_t rows, int32_t nodes, int32_t width,
                           void* hip_stream) {
    NLAM_RANGE("nlam_step_tail_fwd");
    if (delta == nullptr || prev == nullptr || truth == nullptr || target == nullptr || bmask == nullptr || inv_var == nullptr ||
        row_weight == nullptr || pred == nullptr || partials == nullptr)
        return NLAM_EINVAL;
    if (rows < 1 || nodes < 1 || width < 1 || nparts < 1) return NLAM_EINVAL;
    hipLaunchKernelGGL(step_tail_fwd_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)hip_stream, delta, prev, truth, target, dstd,
                       dmean, bmask, inv_var, row_weight, scale, pred, partials, (long)rows * width, nodes, width);
    return (int32_t)hipGetLastError();
}

int32_t nlam_step_tail_bwd(const float* g_pred, const float* gloss, const float* pred, const float* target, const float* dstd,
                           const float* bmask, const float* inv_var, const float* row_weight, float scale, float* d_delta,
                           float* d_prev, int64_t rows, int32_t nodes, int32_t width, void* hip_stream) {
    NLAM_RANGE("nlam_step_tail_bwd");
    if (gloss == nullptr || pred == nullptr || target == nullptr || bmask == nullptr || inv_var == nullptr || row_weight == nullptr)
        return NLAM_EINVAL;
    if ((d_delta == nullptr && d_prev == nullptr) || rows < 1 || nodes < 1 || width < 1) return NLAM_EINVAL;
    const long total = (long)rows * width;
    long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(step_tail_bwd_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)hip_stream, g_pred, gloss, pred, target,
                       dstd, bmask, inv_var, row_weight, scale, d_delta, d_prev, total, nodes, width);
    return (int32_t)hipGetLastError();
}

int32_t nlam_concat(const nlam_cat_t* p, void* hip_stream) {
    NLAM_RANGE("nlam_concat");
    if (p == nullptr || p->out == nullptr || p->nsrc < 1 || p->nsrc > NLAM_MAX_CAT || p->batch < 1 || p->nodes < 0) return NLAM_EINVAL;
    int wtot = 0;
    for (int k = 0; k < p->nsrc; ++k) {
        if (p->ptr[k] == nullptr || p->width[k] < 0 || p->bstride[k] < 0) return NLAM_EINVAL;
        wtot += p->width[k];
    }
    if (p->nodes == 0 || wtot == 0) return 0;
    if ((size_t)kCatRows * wtot * sizeof(float) > 48 * 1024) return NLAM_EUNSUP;   // rows wider than 192 floats: not a grid feature concat
    long blocks = (long)p->batch * (((long)p->nodes + kCatRows - 1) / kCatRows);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(concat_kernel, dim3((int)blocks), dim3(256), (size_t)kCatRows * wtot * sizeof(float), (hipStream_t)hip_stream, *p, wtot);
    return (int32_t)hipGetLastError();
}

int32_t nlam_standardize(const nlam_std_jobs_t* jobs, void* hip_stream) {
    NLAM_RANGE("nlam_standardize");
    if (jobs == nullptr || jobs->njobs < 1 || jobs->njobs > NLAM_MAX_STD_JOBS) return NLAM_EINVAL;
    long most = 0;
    for (int k = 0; k < jobs->njobs; ++k) {
        const nlam_std_job_t& j = jobs->job[k];
        if (j.x == nullptr || j.out == nullptr || j.mean == nullptr || j.std == nullptr) return NLAM_EINVAL;
        if (j.rows < 0 || j.width < 1 || j.rep < 1 || j.width % j.rep != 0) return NLAM_EINVAL;
        if (j.rows * j.width > most) most = j.rows * j.width;
    }
    if (most == 0) return 0;
    long blocks = (most + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(standardize_kernel, dim3((int)blocks, jobs->njobs), dim3(256), 0, (hipStream_t)hip_stream, *jobs);
    return (int32_t)hipGetLastError();
}

int32_t nlam_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                        float beta1, float beta2, float eps, float weight_decay, int32_t step_count, float grad_scale,
                        void* hip_stream) {
    NLAM_RANGE("nlam_adamw_step");
    if (param == nullptr || grad == nullptr || exp_avg == nullptr || exp_avg_sq == nullptr || n < 0 || step_count < 1)
        return NLAM_EINVAL;
    if (n == 0) return 0;
    const float bc1 = 1.f - powf(beta1, (float)step_count);
    const float bc2 = 1.f - powf(beta2, (float)step_count);
    long blocks = (n + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(adamw_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)hip_stream, param, grad, exp_avg,
                       exp_avg_sq, (long)n, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), grad_scale, (const float*)nullptr);
    return (int32_t)hipGetLastError();
}

int32_t nlam_adamw_step_resident(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, int32_t* step_count_dev,
                                 float* bias_corr_dev, float grad_scale, void* hip_stream) {
    NLAM_RANGE("nlam_adamw_step_resident");
    if (param == nullptr || grad == nullptr || exp_avg == nullptr || exp_avg_sq == nullptr || n < 0 || step_count_dev == nullptr ||
        bias_corr_dev == nullptr)
        return NLAM_EINVAL;
    hipStream_t stream = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(adamw_prep_kernel, dim3(1), dim3(64), 0, stream, step_count_dev, bias_corr_dev, beta1, beta2);
    if (n == 0) return (int32_t)hipGetLastError();
    long blocks = (n + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(adamw_kernel, dim3((int)blocks), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq, (long)n, lr, beta1,
                       beta2, eps, weight_decay, 1.f, 1.f, grad_scale, (const float*)bias_corr_dev);
    return (int32_t)hipGetLastError();
}

}  // extern "C"
#endif
