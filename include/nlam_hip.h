/* nlam_hip.h -- C-ABI of the MI355X (gfx950) GNN message-passing hot path.
 *
 * The reference (mllam/neural-lam) has no FFI seam for this path: its "operator
 * interface" is the Python class API of neural_lam/gnn_layers.py and
 * neural_lam/utils/networks.py executed through ATen/PyG ops (SURVEY.md §2.4,
 * §8b).  This header is the seam a maintainer binds instead (ctypes stub in
 * INTEGRATION.md).  Every entry point takes plain device pointers + sizes and a
 * HIP stream; nothing allocates, nothing synchronises, no torch types.
 *
 * All float tensors are fp32, row-major, innermost dimension contiguous.
 * All index tensors are int32 on the device.  Return value: 0 on success,
 * a negative NLAM_E* code on bad arguments, a positive hipError_t on launch
 * failure.
 *
 * Replaces, per entry point:
 *   nlam_mlp_fwd / nlam_mlp_bwd
 *       utils.make_mlp blocks  Linear -> SiLU -> Linear [-> LayerNorm]
 *       (neural_lam/utils/networks.py:8-40) together with the torch.cat of their
 *       inputs and the residual add around them:
 *         aggr_mlp(cat(rec_rep, aggr)) + residual   gnn_layers.py:148-151
 *         grid_emb + encoding_grid_mlp(grid_emb)    models/step_predictors/graph/base.py:308
 *         grid_embedder / *_embedder / output_map    graph/base.py:286-295, 322
 *   nlam_mlp_fwd / nlam_mlp_bwd with three gathered sources, a tile schedule and `aggr` set
 *       InteractionNet / PropagationNet message + aggregate (+ edge update):
 *         PyG propagate: x_j/x_i index_select, message() = edge_mlp(cat(edge, x_j, x_i))
 *         [+ x_j], aggregate() = scatter sum/mean onto num_rec receivers,
 *         edge_rep + edge_diff                       gnn_layers.py:144-155, 168-189, 241-249
 *   nlam_wgrad
 *       autograd's weight gradients of the nn.Linear layers inside those MLPs.
 *   nlam_segment_sum, nlam_segment_sum_acc
 *       autograd of index_select (= index_add by sender), as a CSC segment sum; _acc adds onto
 *       an existing gradient buffer (what autograd's accumulation would do with one more launch).
 *   nlam_split_combine
 *       the deterministic form of PyG's scatter for receivers that exceed one tile
 *       (gnn_layers.py:175-189 under Trainer(deterministic=True), train_model.py:566).
 *   nlam_affine_mix
 *       the elementwise tail of an AR step: prev_state + delta * diff_std + diff_mean
 *       (step_predictors/graph/base.py:331-343) and the boundary overwrite
 *       (forecasters/autoregressive.py:128-131), forward and backward.
 *   nlam_wmse_fwd / nlam_wmse_bwd
 *       metrics.wmse + mask_and_reduce_metric (metrics.py:37-137) with the batch / step means of
 *       models/module.py:463-510.
 *   nlam_reduce_partials
 *       deterministic second stage of the per-workgroup partial sums.
 *   nlam_adamw_step
 *       torch.optim.AdamW(lr, betas=(0.9, 0.95)) of models/module.py:293-304 on
 *       one flat fp32 parameter buffer.
 */
#ifndef NLAM_HIP_H
#define NLAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NLAM_ABI_VERSION 8
#define NLAM_MAX_SRC 3
#define NLAM_MAX_CAT 6

#define NLAM_EINVAL (-1)   /* inconsistent sizes / null pointers        */
#define NLAM_EUNSUP (-2)   /* width outside what this build instantiates */

/* flags */
#define NLAM_F_ADD_SRC0   1u   /* out  = msg + src[0]   (edge update / node residual)          */
#define NLAM_F_ADD_SRC1   2u   /* msg  = mlp + src[1]   (PropagationNet sender / aggr residual) */
#define NLAM_F_MEAN       4u   /* aggregate = mean over in-edges (sum otherwise)               */
#define NLAM_F_SILU_B     8u   /* wgrad: apply SiLU to the B operand while loading             */
/* Factorised first Linear of an edge MLP (gnn_layers.py:168-172, edge_mlp(cat(edge_attr, x_j, x_i))):
 *   W1 [e | x_j | x_i] = W1_e e + (W1_j x)[sender] + (W1_i x)[receiver],
 * the two node-level products being computed once per NODE (nlam_linear) instead of once per EDGE.  With this flag
 * only src[0] goes through the first GEMM (columns 0 .. src[0].width of W1, whose rows are `ldw1` floats apart);
 * src[1..] are gathered pre-activation ADDENDS of width hid:  z1 = W1_e src0 + b1 + sum_k src[k][idx_k].
 * Backward: dmode[0] as usual; dmode[k >= 1] = 3 segment-sums dz1 over the tile's receivers into dsrc[k]
 * ((nseg_total, hid): the gradient of a receiver-gathered addend), 0 / 2 leave it to the caller (a CSC
 * nlam_segment_sum over the dz1 rows gives the gradient of a sender-gathered addend).  Split-bf16 matrix modes,
 * widths that are multiples of 32 and <= 64 (NLAM_EUNSUP otherwise); not combined with NLAM_F_ADD_SRC1. */
#define NLAM_F_PRE_ADD   16u
/* nlam_mlp_bwd_group only: a leaf MLP (no data gradients) of <= 4 input columns accumulates its weight gradients in the
 * backward kernel itself -- dz1 / dz2 are never written: `dz2` receives the (workgroups, dout, hid) partial sums of dW2,
 * `dz1` is unused, `vec_partials` has SEVEN rows per workgroup: db1, db2, dgamma, dbeta, dW1[:, 0], dW1[:, 1], dW1[:, 2].
 * z1 may be NULL (then `b1` is required): the forward of such an MLP need not save its pre-activation. */
#define NLAM_F_LEAF_WGRAD 32u
/* wide launches (a width above 64) of nlam_mlp_fwd / nlam_mlp_bwd: `wpack` already holds the packed weights of exactly this
 * launch -- written by nlam_pack_records from the records nlam_mlp_*_pack_records gave for it, since the weights last
 * changed -- so the pack launch in front of the kernel is skipped. */
#define NLAM_F_WPACK_READY 64u
/* NLAM_F_NO_ACT: no activation between the two Linears (z1 goes into the second GEMM as it is, silu' = 1 in backward).  With
 * W2 = identity, b2 = 0 the launch is `Linear [-> LayerNorm]` with the full gather / concat / residual / aggregation geometry:
 * how utils.make_mlp's hidden_layers = 0 case (utils/networks.py:8-40: blueprint [in, out]) runs on the fused kernels.  fp32
 * matrix path only (NLAM_F_MM_* bits must be 0: NLAM_EUNSUP otherwise); not for grouped / factorised / concatenated launches. */
#define NLAM_F_NO_ACT 128u
/* NLAM_F_STORE_BF16 (nlam_mlp_fwd / nlam_mlp_bwd): z1, xhat (forward: written, backward: read) and dz1, dz2 (backward: written)
 * are rows of bfloat16 instead of float -- the pointers of the structs are then reinterpreted, (batch, rows, width) contiguous.
 * What Lightning's --precision bf16-mixed (train_model.py:163-168) makes of the reference's nn.Linear outputs and their
 * gradients; halves the bytes a layer saves for and hands to its own backward.  Only where nlam_store_bf16_supported() says so
 * (one-term split-bf16 wide kernels, hid and dout whole 32-column blocks); NLAM_EUNSUP otherwise.
 * nlam_wgrad: NLAM_F_A_BF16 = `A` (dz1 / dz2) is bf16, NLAM_F_S_BF16 = src[0] (z1; nsrc must be 1, no gather index) is bf16. */
#define NLAM_F_STORE_BF16 (1u << 10)
#define NLAM_F_A_BF16     (1u << 10)
/* NLAM_F_ACC_DSRC0 (nlam_mlp_bwd, round 6): the data gradient of source 0 (dmode 1: rows scattered through the unique gather
 * index) is ADDED to what dsrc[0] holds instead of overwriting it -- a tensor every autoregressive step of a rollout consumes (the
 * static edge embeddings, models/forecasters/autoregressive.py:63-149) then collects its gradient in ONE buffer, in the order the
 * steps are back-propagated, instead of T buffers and T - 1 add launches over 0.1-0.5 GB each.  Split-bf16 super-tile family
 * only (nlam_mlp_bwd_family == 2): NLAM_EUNSUP elsewhere. */
#define NLAM_F_ACC_DSRC0  (1u << 12)
#define NLAM_F_S_BF16     (1u << 11)
/* NLAM_F_WGRAD_SOLO (nlam_wgrad / nlam_wgrad_nparts, round 6): the launch has the chip to itself -- every launch of the step on one
 * stream, the reference's drop-in path launched from Python -- so its row slices are chosen for ISOLATED speed: up to one
 * workgroup per CU and at least 64 slices (the rounds 2-5 shape).  Without it a big split-bf16 gradient takes at most
 * NLAM_TUNE_WGRAD_MAX_WGS workgroups: it is sized to run beside the data-gradient chain (DESIGN.md finding 53). */
#define NLAM_F_WGRAD_SOLO (1u << 13)
/* matrix path of the GEMMs (bits 8-9): 0 = v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chains);
 * n = 1..3: operands split into n bf16 terms on the bf16 matrix cores, fp32 accumulate
 * (1 = plain bf16 operands, 2 = ~2^-16 product error, 3 = fp32-class ~2^-24).  Shapes the
 * split kernels do not cover (widths not multiples of 32) run the fp32 path. */
#define NLAM_F_MM_SHIFT   8
#define NLAM_F_MM_MASK    (3u << NLAM_F_MM_SHIFT)
#define NLAM_F_MM_BF16X1  (1u << NLAM_F_MM_SHIFT)
#define NLAM_F_MM_BF16X2  (2u << NLAM_F_MM_SHIFT)
#define NLAM_F_MM_BF16X3  (3u << NLAM_F_MM_SHIFT)

typedef struct {
    const float* ptr;     /* (batch|1, n_rows_of_source, width) */
    const int32_t* idx;   /* per tile-row gather index into the source rows, or NULL = row id */
    int64_t bstride;      /* elements between batches; 0 = shared by all batches */
    int32_t width;        /* columns taken from this source */
    int32_t _pad;
} nlam_src_t;

/* One wave-tile of <= 32 consecutive rows (CSR positions) made of whole receivers. */
typedef struct {
    int32_t row0;    /* first row (CSR position)                   */
    int32_t nrows;   /* rows in tile, 0..32                        */
    int32_t seg0;    /* first receiver covered                     */
    int32_t nseg;    /* receivers covered (bit 30 set: partial sum of a split receiver -> atomic add) */
} nlam_tile_t;
#define NLAM_TILE_SPLIT (1 << 30)

typedef struct {
    /* ---- inputs: concatenated (and gathered) sources ---- */
    nlam_src_t src[NLAM_MAX_SRC];
    int32_t nsrc;
    int32_t batch;
    int32_t rows;          /* rows per batch (edges or nodes) */
    int32_t ntiles;        /* tiles per batch; tiles==NULL -> ceil(rows/32) dense tiles */
    const nlam_tile_t* tiles;
    /* ---- MLP parameters (torch.nn.Linear layout: weight (out, in)) ---- */
    const float* W1; const float* b1;   /* (hid, kin), (hid) ; kin = sum of src widths */
    const float* W2; const float* b2;   /* (dout, hid), (dout) */
    const float* ln_w; const float* ln_b; /* (dout) or NULL = no LayerNorm */
    float eps;
    int32_t hid;
    int32_t dout;
    uint32_t flags;
    /* ---- outputs ---- */
    float* out;            /* (batch, out_rows, dout) row output or NULL */
    const int32_t* out_idx;/* per tile-row scatter index (unique) or NULL = row id */
    int64_t out_bstride;
    float* aggr;           /* (batch, nseg_total, dout) segment-reduced msg or NULL */
    const int32_t* rowptr; /* (nseg_total + 1) CSR row pointers (needed with aggr)  */
    const float* inv_deg;  /* (nseg_total) 1/max(deg,1) (needed with NLAM_F_MEAN)   */
    int32_t nseg_total;
    int32_t ldw1;          /* NLAM_F_PRE_ADD: floats between rows of W1 (0 = sum of the source widths) */
    /* ---- saved for backward (all nullable; tile-row order) ---- */
    float* z1;             /* (batch, rows, hid)  pre-activation  */
    float* xhat;           /* (batch, rows, dout) normalised, pre-affine (LN only) */
    float* rstd;           /* (batch, rows) */
    /* ---- scratch for the wide kernels (hid or dout or a source wider than 64) ---- */
    float* wpack;          /* >= nlam_mlp_fwd_wpack_floats(p) floats, or NULL when that is 0 */
    int64_t wpack_floats;  /* capacity of wpack */
    /* ---- optional: src[0] as a row-wise CONCATENATION of ncat pieces ----
     * The torch.cat of the grid input features in front of grid_embedder (step_predictors/graph/base.py:275-286:
     * prev_state | prev_prev_state | forcing | static features) folded into the MLP's first load instead of a launch of its
     * own that materialises (batch, rows, 56).  With ncat > 0: nsrc == 1, src[0].ptr is ignored, src[0].idx must be NULL,
     * src[0].width = sum of cat_width (<= 64); piece k holds rows of cat_width[k] floats, cat_bstride[k] floats apart between
     * batch items (0 = shared: expand_to_batch).  cat_out (optional, (batch, rows, src[0].width)): the kernel also writes the
     * concatenated rows -- what the weight gradient and the backward pass read as the MLP's input.  Served by the
     * split-bf16 narrow kernels (NLAM_EUNSUP otherwise: concatenate with nlam_concat and launch without pieces). */
    int32_t ncat;
    int32_t _pad3;
    const float* cat_ptr[NLAM_MAX_CAT];
    int64_t cat_bstride[NLAM_MAX_CAT];
    int32_t cat_width[NLAM_MAX_CAT];
    float* cat_out;
} nlam_mlp_fwd_t;

typedef struct {
    /* forward geometry, exactly as in the forward call */
    nlam_src_t src[NLAM_MAX_SRC];
    int32_t nsrc;
    int32_t batch;
    int32_t rows;
    int32_t ntiles;
    const nlam_tile_t* tiles;
    const float* W1; const float* W2;
    const float* ln_w;     /* NULL = no LayerNorm */
    int32_t hid;
    int32_t dout;
    uint32_t flags;
    int32_t nseg_total;
    /* upstream gradients */
    const float* g_out;    /* grad of `out` (batch, out_rows, dout) or NULL */
    const int32_t* out_idx;
    int64_t out_bstride;
    const float* g_aggr;   /* grad of `aggr` (batch, nseg_total, dout) or NULL */
    const int32_t* seg_of_row; /* (rows) receiver of each tile-row (needed with g_aggr) */
    const int32_t* rowptr;     /* (nseg_total + 1) CSR row pointers (needed with dmode 3) */
    const float* inv_deg;
    /* saved */
    const float* z1; const float* xhat; const float* rstd;
    /* outputs */
    float* dz1;            /* (batch, rows, hid)   for nlam_wgrad */
    float* dz2;            /* (batch, rows, dout)  for nlam_wgrad */
    float* dsrc[NLAM_MAX_SRC];     /* gradient wrt each source, see dmode */
    int64_t dsrc_bstride[NLAM_MAX_SRC];
    int32_t dmode[NLAM_MAX_SRC];   /* 0 none | 1 rows scattered through src[k].idx (unique)
                                      | 2 tile-row order (rows, width) for a later segment sum
                                      | 3 segment-summed over the tile's receivers -> (nseg_total, width) */
    int32_t ldw1;          /* NLAM_F_PRE_ADD: floats between rows of W1 (0 = sum of the source widths) */
    float* vec_partials;   /* (nblocks, 4, vec_stride): per-workgroup db1, db2, dgamma, dbeta partial sums */
    int32_t vec_partials_rows; /* capacity in rows; must be >= nlam_mlp_bwd_blocks(p) */
    int32_t vec_stride;    /* row stride of vec_partials: multiple of 64, >= max(hid, dout) */
    float* wpack;          /* >= nlam_mlp_bwd_wpack_floats(p) floats, or NULL when that is 0 */
    int64_t wpack_floats;
    const float* b1;       /* NLAM_F_LEAF_WGRAD only (required there): first-layer bias; that kernel recomputes the pre-activation
                              z1 = W1 x + b1 from the (<= 4-column) input row and never reads `z1` (which may be NULL) */
    int32_t dz2_ld;        /* floats between rows of dz2: MUST equal nlam_mlp_bwd_dz2_ld(p) -- 0 = dout, except for an output width
                              that is not a multiple of 32 on the split-bf16 kernel (output_map, dout = 17): then 32-padded, the
                              columns past dout are written as zeros, and nlam_wgrad takes dz2 with m = that stride */
    int32_t _pad2;
} nlam_mlp_bwd_t;

typedef struct {
    const float* A;        /* (batch*rows, m) contiguous: dz1 or dz2 */
    int32_t m;
    int32_t batch;
    int32_t rows;
    int32_t nsrc;
    nlam_src_t src[NLAM_MAX_SRC]; /* B operand = concat of gathered sources (tile-row order idx) */
    uint32_t flags;        /* NLAM_F_SILU_B */
    int32_t n;             /* sum of widths */
    float* partials;       /* (nparts, m, n) */
    int32_t nparts;        /* number of row-slices = workgroups launched */
    int32_t _pad;
} nlam_wgrad_t;

/* number of persistent waves the fwd/bwd kernels launch (for sizing vec_partials) */
int32_t nlam_grid_waves(void);
int32_t nlam_abi_version(void);
/* workgroups nlam_mlp_fwd / nlam_mlp_bwd launch for `total_tiles` = ntiles * batch */
int32_t nlam_num_blocks(int64_t total_tiles);
/* widest hidden/output width the fused kernels of this build instantiate */
int32_t nlam_max_width(void);

/* Launch-shape tuning (process-wide; tests use it to force a kernel family at small sizes).
 *   NLAM_TUNE_WBF_MIN_SUPERTILES: wide (> 64) fused-MLP launches with fewer super tiles than `value`
 *   run on the fp32 MFMA kernels even when a split-bf16 matrix mode is requested (default 192:
 *   a 128-row super tile per workgroup needs that many to occupy 256 CUs).
 * Returns 0, or NLAM_EINVAL for an unknown key / negative value. */
#define NLAM_TUNE_WBF_MIN_SUPERTILES 1
/*   NLAM_TUNE_WGRAD_CHUNKS: 32-row chunks a weight-gradient workgroup streams through before it writes its (m x n)
 *   partial, for problems of more than 128 chunks (default 8; 1 = one partial per chunk up to the cap of 512 workgroups;
 *   larger values trade launch width for less partial-sum traffic). */
#define NLAM_TUNE_WGRAD_CHUNKS 2
/*   NLAM_TUNE_LIN_WGS: workgroups of an nlam_linear launch with k <= 256 on the wide kernel, whose weight strip stays in
 *   LDS for the whole launch (default 256 = one per CU; 0 = stream the strip through two LDS buffers as for k > 256). */
#define NLAM_TUNE_LIN_WGS 3
/*   NLAM_TUNE_WGRAD_MIN_PARTS: least number of row slices (= partial sums per element) a weight gradient over more than that
 *   many 32-row chunks is cut into (default 128; for weight matrices of more than 128 rows as many as give 64 workgroups:
 *   64 / number of 256 x 256 windows -- launch width for the mid-size problems; smaller = less partial-sum traffic; setting it
 *   sets both to the given slice count). */
#define NLAM_TUNE_WGRAD_MIN_PARTS 4
/*   NLAM_TUNE_WGRAD_BIG_MIN_ROWS: rows (x batch) from which a split-bf16 weight gradient with more than 128 output rows uses
 *   256 x 256 windows; below it 128 x 128 windows (same launch width, a quarter of the row slices and partial sums). */
#define NLAM_TUNE_WGRAD_BIG_MIN_ROWS 5
/*   NLAM_TUNE_WBF_HALF: split-bf16 wide kernels on 4-wave workgroups of half the rows, TWO co-resident per CU, so that one
 *   workgroup's load / store phases overlap the other's matrix phases (bit 0: forward, bit 1: backward; shapes without a
 *   half-size instantiation keep the 8-wave kernels).  Bit 2 (round 6): the one-term forward above d = 256 on 4-wave workgroups
 *   of 64 rows (two MFMAs per weight fragment fetched from L2 instead of one). */
#define NLAM_TUNE_WBF_HALF 6
/*   NLAM_TUNE_LIN_GEMM: nlam_linear with n % 128 == 0 on the LDS-tiled GEMM: 1 (default) = where it beats the strip kernel of
 *   rounds 2-4 (one term: everywhere; two / three terms: 6 561-row class up to K = 256, 63 784-row class from K = 512), 0 = never,
 *   2 = wherever it applies (tests); a value above 2 sets the row count from which its tiles are 128 rows high (default 32 768). */
#define NLAM_TUNE_LIN_GEMM 7
/*   NLAM_TUNE_WBF_V4: the split-bf16 wide kernels' instantiations with branch-free 16-byte chunk accesses for launches whose
 *   widths are all multiples of 4 (default 1; 0 = the generic lane-predicated accessors everywhere, for A/B runs). */
#define NLAM_TUNE_WBF_V4 8
/*   NLAM_TUNE_WGRAD_LDMA (round 6): one-term weight gradients with 256 x 256 windows on wgrad_ldma_kernel (both operands streamed
 *   by LDS-DMA into a 3-stage ring, bf16 operands read with the LDS transpose read): bit 0 = launches with bf16 operands
 *   (NLAM_F_A_BF16), bit 1 = fp32-operand one-term launches, bit 2 = fp32-class (three-term) launches; default 3.  With fp32 operands
 *   the kernel alone is no faster than the column-per-thread kernel of rounds 1-5 (profiles/round6/wgrad_check.log: one term 112 vs
 *   98 us without / 84 vs 102 us with the SiLU; three terms 67.9 vs 64.8 us at 57 616 x 256 x 256, bit-identical results); inside the
 *   step the one-term form gains 1.2 % at cfg5 (profiles/round6/ab_wgrad_ldma_fp32.log), the three-term form nothing.  0 = the
 *   column-per-thread kernel everywhere. */
#define NLAM_TUNE_WGRAD_LDMA 9
/*   NLAM_TUNE_WGRAD_LDMA_VAR: (rows per stage, ring depth) variant of wgrad_ldma_kernel, 0 = default (A/B runs). */
#define NLAM_TUNE_WGRAD_LDMA_VAR 10
/*   NLAM_TUNE_WBF_EDGE (round 6): the factorised InteractionNet edge layers of width 512 in the one-term matrix mode and of width
 *   256 in the three-term mode on mlp_fwd_edge_kernel / mlp_bwd_edge_kernel (shapes as template constants, software-pipelined across
 *   super tiles); default 1; 0 = the template kernels (mlp_fwd_wbf_kernel / mlp_bwd_wbf_kernel); 3 = as 1 without the three-term
 *   backward. */
#define NLAM_TUNE_WBF_EDGE 11
/*   NLAM_TUNE_WGRAD_MAX_WGS (round 6): most workgroups (row slices x 256 x 256 windows) of a split-bf16 weight gradient with more
 *   than 128 output rows (default 128: half the CUs -- the launch runs beside the data-gradient chain; 256 = one per CU, rounds 2-5;
 *   4 .. 1024). */
#define NLAM_TUNE_WGRAD_MAX_WGS 12
/*   NLAM_TUNE_CHAIN_CUS (round 6): CUs a wide (> 64) fused-MLP launch sizes its persistent grid for (default 256 = all of them;
 *   16 .. 1024: values above 256 oversubscribe the CUs; set before the first launch of a step is recorded -- nlam_mlp_bwd_blocks
 *   follows it.  Measured (profiles/round6/ab_chain_cus.log): 192 .. 240 lose 1.5-5 %, 384 .. 1024 lose 0.5-5 %). */
#define NLAM_TUNE_CHAIN_CUS 13
int32_t nlam_set_tuning(int32_t key, int32_t value);
/* Scratch the wide kernels need for the packed (MFMA A-operand order) weights of this
 * call; 0 when the call runs on the narrow (weights-in-LDS) kernels. */
int64_t nlam_mlp_fwd_wpack_floats(const nlam_mlp_fwd_t* p);
int64_t nlam_mlp_bwd_wpack_floats(const nlam_mlp_bwd_t* p);
/* workgroups nlam_mlp_bwd launches for this call (rows of vec_partials it writes) */
int32_t nlam_mlp_bwd_blocks(const nlam_mlp_bwd_t* p);
/* 1 if this forward launch (every field but z1 / xhat / wpack set) and the backward / weight-gradient launches that belong to it
 * can run with NLAM_F_STORE_BF16 (z1, xhat, dz1, dz2 as bfloat16 rows) */
int32_t nlam_store_bf16_supported(const nlam_mlp_fwd_t* p);
/* row stride the call's dz2 buffer must have (set p->dz2_ld to it); 0 = dout (every shape but a ragged output width) */
int32_t nlam_mlp_bwd_dz2_ld(const nlam_mlp_bwd_t* p);
/* number of row slices (p->nparts) that fills the chip for this weight-gradient shape */
int32_t nlam_wgrad_nparts(const nlam_wgrad_t* p);

int32_t nlam_mlp_fwd(const nlam_mlp_fwd_t* p, void* hip_stream);
int32_t nlam_mlp_bwd(const nlam_mlp_bwd_t* p, void* hip_stream);

/* Pre-packed weights for the narrow (hid, dout <= 64) split-bf16 kernels.  The reference's nn.Linear weights
 * (utils/networks.py:8-40) change once per optimizer step (models/module.py:293-304) but every fused-MLP workgroup of
 * every launch used to re-read them as fp32 and split them into bf16 terms itself.  nlam_mlp_pack writes, for a table of
 * MLPs, the images the kernels lay out in LDS; nlam_mlp_fwd / nlam_mlp_bwd (and the grouped launches) then take such an
 * image in `wpack` (wpack_floats = its size) and fetch it by LDS-DMA.  An image is valid for the (source widths, hid,
 * dout, NLAM_F_PRE_ADD, matrix mode, ldw1) it was packed for and until the weights change; passing NULL keeps the
 * self-staging path.  Launches the images do not serve (fp32 matrix mode, generic shapes) ignore `wpack`; the wide
 * kernels keep their own meaning of `wpack` (scratch they pack per launch), so pass an image to narrow launches only.
 * nlam_mlp_pack_floats: size of the forward (which = 0) / backward (which = 1) image, 0 when the shape is not served. */
typedef struct {
    const float* W1;       /* (hid, ldw1 or sum of widths) */
    const float* W2;       /* (dout, hid) */
    float* fwd_image;      /* >= nlam_mlp_pack_floats(job, 0) floats, 16-byte aligned, or NULL */
    float* bwd_image;      /* >= nlam_mlp_pack_floats(job, 1) floats, 16-byte aligned, or NULL */
    int32_t hid, dout, nsrc;
    int32_t ldw1;          /* floats between rows of W1; 0 = sum of the GEMM sources' widths */
    int32_t width[NLAM_MAX_SRC];
    uint32_t flags;        /* NLAM_F_PRE_ADD | NLAM_F_MM_BF16X1..3 */
} nlam_pack_job_t;
int64_t nlam_mlp_pack_floats(const nlam_pack_job_t* job, int32_t which);
/* The same idea for the WIDE kernels, whose `wpack` scratch is packed per launch by default (a pack kernel in front of every
 * wide forward / backward launch: 184 + 147 launches per Hi-LAM d = 128 step).  nlam_mlp_fwd_pack_records /
 * nlam_mlp_bwd_pack_records describe, as opaque 64-byte records, the pack jobs that fill p->wpack for exactly the launch `p`
 * (same shapes, rows, tiles, batch, flags, dmode: they select the kernel family and with it the layout); they return the
 * number of records (<= 4; 0 for a narrow launch) and the record kind (0: fp32 A-fragment order, 1 / 3: one / three bf16
 * terms).  nlam_pack_records runs a table of records of ONE kind (device memory) in one launch.  A launch whose wpack was
 * filled this way passes NLAM_F_WPACK_READY. */
typedef struct {
    unsigned char bytes[64];
} nlam_pack_rec_t;
int32_t nlam_mlp_fwd_pack_records(const nlam_mlp_fwd_t* p, nlam_pack_rec_t* out, int32_t cap, int32_t* kind);
int32_t nlam_mlp_bwd_pack_records(const nlam_mlp_bwd_t* p, nlam_pack_rec_t* out, int32_t cap, int32_t* kind);
int32_t nlam_pack_records(const nlam_pack_rec_t* recs_device, int32_t n, int32_t kind, void* hip_stream);
/* `jobs_device`: the job table in DEVICE memory (addresses are stable, so it is uploaded once); one launch packs all */
int32_t nlam_mlp_pack(const nlam_pack_job_t* jobs_device, int32_t njobs, void* hip_stream);

/* GROUPED launches: n <= NLAM_MAX_GROUP independent fused MLPs of the same kernel shape in ONE grid (every workgroup is
 * bound to one member, the 256 workgroups are dealt in proportion to the members' tile counts).  For the embedders of
 * the static grid / mesh / edge features (utils.make_mlp blocks called back to back at graph/base.py:286-295,
 * hierarchical.py:195-231): as separate launches each is a latency-bound chain link.  Covered: single-source members
 * without residual / aggregation whose widths share the 32-column block counts, split-bf16 matrix modes, hid and dout
 * <= 64 (NLAM_EUNSUP otherwise: launch the members one by one).  nlam_mlp_group_blocks gives the workgroups each
 * member gets: member k of the backward writes blocks[k] rows of its vec_partials. */
#define NLAM_MAX_GROUP 8
int32_t nlam_mlp_group_blocks(const int64_t* tiles, int32_t n, int32_t* blocks);
int32_t nlam_mlp_fwd_group(const nlam_mlp_fwd_t* ps, int32_t n, void* hip_stream);
int32_t nlam_mlp_bwd_group(const nlam_mlp_bwd_t* ps, int32_t n, void* hip_stream);
/* Round 4: the grouped entry points also take members of the fp32 WIDE family (a width in 65 .. 128 columns below the
 * split-bf16 super-tile threshold, or matrix mode 0): the chunks of a `SplitMLPs` layer (gnn_layers.py:274-324 called from
 * hi_lam_parallel.py:127-143 -- 3 .. 206 tiles each at the bench size) and the static-feature embedders at d = 128; any
 * source / residual / aggregation set the single launch takes, one kernel shape (hid, dout, flags, LayerNorm or not; source
 * counts and widths may differ) for all members, each with its own `wpack` scratch.  nlam_mlp_*_family says which
 * kernel family a launch runs on (0 narrow, 1 fp32 wide, 2 split-bf16 wide: members of family 2 are launched one by one);
 * nlam_mlp_bwd_group_blocks gives the workgroups (= vec_partials rows written) per member of a grouped backward of either
 * family, checking the members as the launch would. */
int32_t nlam_mlp_fwd_family(const nlam_mlp_fwd_t* p);
int32_t nlam_mlp_bwd_family(const nlam_mlp_bwd_t* p);
int32_t nlam_mlp_bwd_group_blocks(const nlam_mlp_bwd_t* ps, int32_t n, int32_t* blocks);
int32_t nlam_wgrad(const nlam_wgrad_t* p, void* hip_stream);
/* n <= NLAM_MAX_GROUP weight gradients of one shape (m, source widths, flags) in ONE grid, each member with its own rows and
 * `partials` (nparts = nlam_wgrad_nparts of that member): the chunks of a `SplitMLPs` layer (gnn_layers.py:311-324).  Members of
 * the split-bf16 wide family with fp32 operands; NLAM_EUNSUP otherwise (launch them one by one). */
int32_t nlam_wgrad_group(const nlam_wgrad_t* ps, int32_t n, void* hip_stream);

/* out[b, s, :] = scale(s) * sum_{q in [ptr[s], ptr[s+1])} in[b, order[q], :]   (order NULL = q) */
int32_t nlam_segment_sum(const float* in, int64_t in_bstride, const int32_t* ptr, const int32_t* order,
                         const float* scale, float* out, int32_t nseg, int32_t width, int32_t batch,
                         void* hip_stream);
/* same, added onto what `out` already holds (out += ...): lets the sender-side gradient of a layer whose senders
 * and receivers are the same nodes (mesh <-> mesh) land in the receiver-side gradient buffer instead of a second
 * tensor that autograd would add with one more launch */
int32_t nlam_segment_sum_acc(const float* in, int64_t in_bstride, const int32_t* ptr, const int32_t* order,
                         const float* scale, float* out, int32_t nseg, int32_t width, int32_t batch,
                         void* hip_stream);
/* out[b, s] += extra[b, s] + scale[s] * sum_{q in ptr[s]:ptr[s+1]} in[b, order[q]]   (round 6: a layer's sender-side gradient, its
 * receiver-side gradient `extra` and the node MLP's gradient already in `out` meet in ONE pass -- the torch add launch behind
 * every mesh <-> mesh layer's backward (gnn_layers.py:110-157: send_rep is rec_rep) is gone). */
int32_t nlam_segment_sum_add(const float* in, int64_t in_bstride, const int32_t* ptr, const int32_t* order, const float* scale,
                             const float* extra, float* out, int32_t nseg, int32_t width, int32_t batch, void* hip_stream);

/* nlam_segment_sum over input rows stored as bfloat16 (width % 8 == 0; out is float): the dz1 rows of a backward launch that ran with
 * NLAM_F_STORE_BF16 */
int32_t nlam_segment_sum_bf16(const uint16_t* in, int64_t in_bstride, const int32_t* ptr, const int32_t* order,
                              const float* scale, float* out, int32_t nseg, int32_t width, int32_t batch, void* hip_stream);

/* buf[b, dst[s], :] = sum_{q in [ptr[s], ptr[s+1])} buf[b, src[q], :], q ascending, for s < n; rows of `width` floats, batch
 * stride `bstride` floats.  Second pass of the DETERMINISTIC reduction of receivers with more in-edges than a 32-row tile
 * (reference: PyG's scatter under Trainer(deterministic=True), gnn_layers.py:175-189, train_model.py:566): the tile schedule
 * gives every piece of such a receiver a virtual segment behind the real ones (rowptr / inv_deg are extended accordingly, the
 * kernels see ordinary one-receiver tiles and use plain stores), and this sums the pieces into the receiver's row.  The
 * one-pass alternative -- NLAM_TILE_SPLIT tiles, atomic adds into a zeroed buffer -- stays available to ABI users. */
int32_t nlam_split_combine(float* buf, int64_t bstride, const int32_t* ptr, const int32_t* src, const int32_t* dst, int32_t n,
                           int32_t width, int32_t batch, void* hip_stream);

/* out[i] (+)= sum_p partials[p * stride + i], i < n ; accumulate != 0 adds into out */
int32_t nlam_reduce_partials(const float* partials, int32_t nparts, int64_t stride, int32_t n, float* out,
                             int32_t accumulate, void* hip_stream);

/* Up to NLAM_MAX_REDUCE_JOBS reductions of the nlam_reduce_partials kind in ONE launch: the dW1, dW2,
 * db1, db2, dgamma, dbeta of one fused-MLP backward (autograd's AccumulateGrad `grad += new`, folded in
 * when accumulate != 0 and out points into the gradient buffer). */
#define NLAM_MAX_REDUCE_JOBS 40   /* e.g. the 4 x 9 reductions of the four static-feature embedders of one grouped backward */
typedef struct {
    const float* partials;
    float* out;
    int64_t stride;        /* elements between partials */
    int32_t nparts;
    int32_t n;
    int32_t accumulate;
    int32_t ncols;         /* > 0: element i of the reduction goes to out[(i / ncols) * ld + i % ncols] (a column block */
    int32_t ld;            /*      of a wider row-major matrix: one source's slice of W1.grad); 0: out[i]                */
    int32_t _pad;
} nlam_reduce_job_t;
typedef struct {
    nlam_reduce_job_t job[NLAM_MAX_REDUCE_JOBS];
    int32_t njobs;
    int32_t _pad;
} nlam_reduce_jobs_t;
int32_t nlam_reduce_jobs(const nlam_reduce_jobs_t* jobs, void* hip_stream);

/* Node-level product of the factorised edge MLP (NLAM_F_PRE_ADD) and its data gradient:
 *   out[r][h] (+)= sum_c x[r][c] * W[h * ldn + c * ldk],   r < rows, h < n, c < k
 * (forward: W = W1 + column offset, ldn = row stride of W1, ldk = 1; backward: the transposed product with
 * ldn = 1, ldk = row stride of W1).  x (rows, k) and out (rows, n) are contiguous; k, n in {32, 64, 128};
 * matrix mode from flags (NLAM_F_MM_BF16X1..3; the fp32-MFMA mode is not instantiated: NLAM_EUNSUP). */
typedef struct {
    const float* x;
    const float* W;
    float* out;
    int64_t rows;
    int64_t ldn;
    int64_t ldk;
    int32_t k;
    int32_t n;
    int32_t accumulate;    /* != 0: out += */
    uint32_t flags;
    const float* W2;       /* optional second product over the same x (widths above 64 only): out2 = x . A2^T, same shapes and */
    float* out2;           /* strides (the sender- and the receiver-side product of a layer whose senders are its receivers)   */
} nlam_linear_t;
int32_t nlam_linear(const nlam_linear_t* p, void* hip_stream);
/* 1 when nlam_mlp_fwd / nlam_mlp_bwd / nlam_linear have a factorised (NLAM_F_PRE_ADD) kernel for this problem, else 0
 * (only nsrc, the source widths, hid, dout, flags, ntiles and batch of *p are read) */
int32_t nlam_pre_add_supported(const nlam_mlp_fwd_t* p);

/* Training loss of ForecasterModule.training_step (models/module.py:463-510) with metrics.wmse /
 * mask_and_reduce_metric (metrics.py:37-137) for a per-variable std:
 *   loss = scale * sum_{row, v} row_weight[row % nodes] * inv_var[v] * (pred - target)^2,
 * row_weight = interior mask / #interior nodes, scale = 1 / (batch * ar_steps).  The forward writes one
 * partial per block (nparts blocks; finish with nlam_reduce_partials, fixed order); the backward takes the
 * upstream scalar gradient from device memory. */
int32_t nlam_wmse_fwd(const float* pred, const float* target, const float* inv_var, const float* row_weight, int64_t rows,
                      int32_t nodes, int32_t nvars, float scale, float* partials, int32_t nparts, void* hip_stream);
int32_t nlam_wmse_bwd(const float* pred, const float* target, const float* inv_var, const float* row_weight,
                      const float* gscalar, int64_t rows, int32_t nodes, int32_t nvars, float scale, float* dpred,
                      void* hip_stream);

/* State update of one autoregressive step, all optional terms in one pass over (rows, width) fp32 rows:
 *   out[r][f] = a[r % nodes] * x[r][f]  +  c[r % nodes] * ( y[r][f] + z[r][f] * s[f] + m[f] )
 * NULL x (with a), y, z (with s), m drop their term; NULL c means 1.  Replaces the elementwise chains
 *   prev_state + (delta * diff_std + diff_mean)                step_predictors/graph/base.py:331-343, base.py:335-396 (no clamp)
 *   boundary_mask * true_state + interior_mask * pred_state    forecasters/autoregressive.py:128-131
 * and, with the roles of the operands permuted, their backward (g * interior_mask, g * diff_std). */
int32_t nlam_affine_mix(const float* x, const float* a, const float* y, const float* c, const float* z, const float* s,
                        const float* m, float* out, int64_t rows, int32_t nodes, int32_t width, void* hip_stream);

/* ForecasterModule.on_after_batch_transfer (models/module.py:326-367): up to NLAM_MAX_STD_JOBS tensors
 * standardised in ONE launch,  out[r][c] = (x[r][c] - mean[c / rep]) / std[c / rep].  `rep` = 1 for the states;
 * for the forcing it is the window size: the per-feature statistics are tiled feature-major over the window
 * (repeat_interleave, module.py:352-358).  IEEE subtraction and division: bit-equal to the reference's formula. */
#define NLAM_MAX_STD_JOBS 4
typedef struct {
    const float* x;
    float* out;
    const float* mean;     /* (width / rep) */
    const float* std;      /* (width / rep), already clamped to >= eps by the caller (module.py:306-324) */
    int64_t rows;
    int32_t width;
    int32_t rep;
} nlam_std_job_t;
typedef struct {
    nlam_std_job_t job[NLAM_MAX_STD_JOBS];
    int32_t njobs;
    int32_t _pad;
} nlam_std_jobs_t;
int32_t nlam_standardize(const nlam_std_jobs_t* jobs, void* hip_stream);

/* The elementwise tail of ONE autoregressive step in one pass (forward) and one pass (backward):
 *   new   = prev + delta * dstd[f] + dmean[f]          step_predictors/graph/base.py:331-343 (no clamping configured)
 *   pred  = bmask[n] * truth + (1 - bmask[n]) * new     forecasters/autoregressive.py:128-131
 *   loss += scale * row_weight[n] * inv_var[f] * (pred - target)^2    metrics.py:37-137 + module.py:463-510
 * rows = batch * nodes rows of `width` state variables; dstd / dmean may be NULL (1 / 0).  The forward writes one loss
 * partial per block (`nparts` blocks; finish with nlam_reduce_partials).  The backward takes the gradient of `pred` that
 * later AR steps produced (g_pred, NULL for the last step) and the scalar gradient of the loss from device memory:
 *   G = g_pred + 2 * scale * gloss * row_weight[n] * inv_var[f] * (pred - target)
 *   d_delta = (1 - bmask[n]) * dstd[f] * G ,   d_prev = (1 - bmask[n]) * G   (either output may be NULL). */
int32_t nlam_step_tail_fwd(const float* delta, const float* prev, const float* truth, const float* target, const float* dstd,
                           const float* dmean, const float* bmask, const float* inv_var, const float* row_weight, float scale,
                           float* pred, float* partials, int32_t nparts, int64_t rows, int32_t nodes, int32_t width,
                           void* hip_stream);
int32_t nlam_step_tail_bwd(const float* g_pred, const float* gloss, const float* pred, const float* target, const float* dstd,
                           const float* bmask, const float* inv_var, const float* row_weight, float scale, float* d_delta,
                           float* d_prev, int64_t rows, int32_t nodes, int32_t width, void* hip_stream);

/* Row-wise concatenation of up to NLAM_MAX_CAT sources into out (rows, sum of widths): the torch.cat of the grid input
 * features (prev_state, prev_prev_state, forcing, static features; step_predictors/graph/base.py:275-283).  A source
 * with bstride 0 is shared by all batch items (expand_to_batch, step_predictors/base.py:122-139). */
typedef struct {
    const float* ptr[NLAM_MAX_CAT];
    int64_t bstride[NLAM_MAX_CAT];   /* floats between batch items of the source; 0 = shared */
    int32_t width[NLAM_MAX_CAT];
    int32_t nsrc;
    int32_t batch;
    int32_t nodes;                   /* rows per batch item */
    int32_t _pad;
    float* out;                      /* (batch, nodes, sum of widths) */
} nlam_cat_t;
int32_t nlam_concat(const nlam_cat_t* p, void* hip_stream);

/* The data path: WeatherDataset.__getitem__ (weather_dataset.py:467-533) for a BATCH of sample indices, cut straight out
 * of analysis time series that are resident in HBM -- _slice_state_time (:199-262, analysis branch :255-261),
 * _slice_forcing_time (:264-361, analysis branch :330-359), the feature-major / window-minor stacking of the forcing
 * window (:443-445) -- optionally followed, in the same pass, by ForecasterModule.on_after_batch_transfer
 * (models/module.py:326-367: (x - mean) / std, forcing statistics tiled over the window, IEEE sub + div).
 *   sample i = sample_idx[b]:  state rows  [i + max(0, past - 2), i + max(2, past) + ar_steps)  -> 2 init + ar_steps target
 *                              forcing of step t, window slot w:  time  i + max(2, past) + t - past + w
 *                              forcing_windowed[b][t][n][f * window + w],  window = past + future + 1
 * Sample indices are read on the device (a resident permutation drives an epoch without host traffic); they must lie
 * in [0, nlam_window_len): the kernel clamps the time index instead of faulting, the Python wrapper validates host
 * indices (IndexError as weather_dataset.py:497-503).  HBM-bound: every output float is read once and written once. */
typedef struct {
    const float* state;          /* (n_times, nodes, d_state) */
    const float* forcing;        /* (n_times, nodes, d_forcing) or NULL when d_forcing == 0 */
    const int64_t* times;        /* (n_times) time stamps (ns) or NULL */
    const int64_t* sample_idx;   /* device, (batch) */
    float* init_states;          /* (batch, 2, nodes, d_state) */
    float* target_states;        /* (batch, ar_steps, nodes, d_state) */
    float* forcing_windowed;     /* (batch, ar_steps, nodes, d_forcing * window) or NULL when d_forcing == 0 */
    int64_t* target_times;       /* (batch, ar_steps) or NULL; with times == NULL it receives the (clamped) time index */
    const float* state_mean;     /* (d_state)   all four NULL: raw values, as the reference's dataset returns them */
    const float* state_std;      /* (d_state) */
    const float* forcing_mean;   /* (d_forcing) */
    const float* forcing_std;    /* (d_forcing) */
    int64_t n_times;
    int32_t nodes, d_state, d_forcing, batch;
    int32_t ar_steps, num_past_forcing_steps, num_future_forcing_steps, _pad;
} nlam_window_t;
/* len(WeatherDataset) for analysis data (weather_dataset.py:179-194); n_forcing_times < 0: no forcing */
int64_t nlam_window_len(int64_t n_state_times, int64_t n_forcing_times, int32_t ar_steps, int32_t num_past_forcing_steps,
                        int32_t num_future_forcing_steps);
int32_t nlam_window_batch(const nlam_window_t* p, void* hip_stream);

/* decoupled-weight-decay Adam on flat buffers; step_count is the 1-based step */
int32_t nlam_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                        float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step_count,
                        float grad_scale, void* hip_stream);
/* the same update with the step count RESIDENT on the device: a one-thread launch advances *step_count_dev (int32, starts at
 * 0) and leaves the bias corrections 1 - beta1^t, sqrt(1 - beta2^t) in bias_corr_dev[0..1] for the update launch behind it.
 * No launch argument depends on the step, so the optimizer can be part of a captured HIP graph that is replayed per step. */
int32_t nlam_adamw_step_resident(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, int32_t* step_count_dev,
                                 float* bias_corr_dev, float grad_scale, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* NLAM_HIP_H */
