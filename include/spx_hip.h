/*
 * spx_hip.h — C ABI of libspx_hip.so: the MI355X (gfx950) implementation of
 * ScaleProtoSeg's prototype-distance hot path.
 *
 * The reference has no FFI: its boundary for this path is the Python surface of
 * the PPNetMultiScale nn.Module.  The entry points below are the
 * device-side operators that surface needs; the Python package
 * `scaleprotoseg_amd` binds them with ctypes and mirrors the reference's
 * module/function names above them.  Each entry point cites the reference
 * code it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless named host_*; the caller owns and
 *    allocates every buffer (torch.empty); nothing is retained after return;
 *  - `stream` is a hipStream_t passed as void* (0 = default stream); calls only
 *    enqueue work, they never synchronise;
 *  - return 0 on success, non-zero on error; spx_last_error() gives the message
 *    of the last failing call of the calling thread;
 *  - tensors are dense, row-major, in the reference's layouts:
 *      features X   [B, C, H*W]      bf16 (x_dtype 0) or fp32 (x_dtype 1), C = S*Cs; the reference feeds a Sigmoid's
 *                                    output (values in (0, 1)).  The prototype gradient's product runs in fp16: features are
 *                                    taken exactly for 2^-14 <= |x| < 65520 and saturate at +-65504 beyond (never inf)
 *      bank         [P, Cs]          fp32   (prototype_vectors.view(P, Cs))
 *      distances    [B, P, H*W]      fp32
 *      activations  [B*H*W, P]       fp32   (NHWC pixel order)
 *      logits       [B*H*W, K]       fp32
 *      last layer W [K, P]           fp32
 */
#ifndef SPX_HIP_H
#define SPX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPX_MAX_PANELS 64   /* (scale, <=192-prototype block) work units per pixel tile */
#define SPX_ABI_VERSION 15

/* How the prototype bank is cut into MFMA panels.  Filled by spx_make_plan(). */
typedef struct spx_plan {
    int32_t num_prototypes;        /* P */
    int32_t num_classes;           /* K */
    int32_t num_scales;            /* S */
    int32_t channels_per_scale;    /* Cs (multiple of 16) */
    int32_t kc;                    /* channels staged per LDS step (32; a 16-channel tail is zero-filled) */
    int32_t npb;                   /* 32-prototype blocks per panel (2, 4 or 6) */
    int32_t ncb;                   /* 32-class blocks of the head (1, 2 or 5) */
    int32_t npanels;
    int32_t panel_ch0[SPX_MAX_PANELS];  /* first feature channel of the panel's scale */
    int32_t panel_p0[SPX_MAX_PANELS];   /* first prototype row of the panel */
    int32_t panel_np[SPX_MAX_PANELS];   /* prototypes in the panel (<= 32*npb) */
} spx_plan;

int spx_version(void);
const char* spx_last_error(void);

/* Build the panel plan from the module's scale table
 * (scale_num_prototypes, segmentation/model/model_multiscale.py:146-149, re-packed by
 * prune_prototypes :408-423).  host_scale_lo/hi: S entries each, HOST memory. */
int spx_make_plan(int32_t P, int32_t K, int32_t S, int32_t Cs,
                  const int32_t* host_scale_lo, const int32_t* host_scale_hi, spx_plan* out);

/* Sizes (bytes) of the packed operand buffers for a plan. */
size_t spx_packed_bank_bytes(const spx_plan* plan);   /* bf16 MFMA A-fragments of the bank          */
size_t spx_packed_bankT_bytes(const spx_plan* plan);  /* bf16 A-fragments of -2 bank^T (backward dX) */
size_t spx_packed_p2_bytes(const spx_plan* plan);     /* fp32 |p|^2 per padded prototype            */
size_t spx_packed_head_bytes(const spx_plan* plan);   /* split-bf16 (hi,lo) fragments of W          */
size_t spx_packed_headT_bytes(const spx_plan* plan);  /* split-bf16 fragments of W^T (backward)     */

/* Re-pack the fp32 bank into bf16 MFMA fragment order and compute |p|^2
 * (replaces the per-forward p**2 / sum of model_multiscale.py:270-274).
 * packed_bankT may be NULL when no backward is needed. */
int spx_pack_bank(const spx_plan* plan, const float* bank, void* packed_bank, void* packed_bankT,
                  float* packed_p2, void* stream);

/* Re-pack a dense [K, P] head matrix (last_layer.weight, model_multiscale.py:225; or the dense
 * form of the grouping projections) into split-bf16 fragments.  packed_headT may be NULL. */
int spx_pack_head(const spx_plan* plan, const float* W, void* packed_head, void* packed_headT, void* stream);

/* Everything one forward (+ backward) needs in ONE launch: spx_pack_bank + spx_pack_head (+ spx_pack_group_tail, and then the
 * head^T in the unit order of spx_pack_headT_units) - the per-step form for training, where every parameter changes between
 * steps and four small launches cost more than the packing itself.  W / Wg and their outputs may be NULL (no head / no
 * grouping tail); packed_bankT, packed_headT, packed_tailT NULL = no backward.  Same buffers and sizes as the single calls. */
int spx_pack_all(const spx_plan* plan, const float* bank, const float* W, const float* Wg, int32_t K2, void* packed_bank,
                 void* packed_bankT, float* packed_p2, void* packed_head, void* packed_headT, void* packed_tail,
                 void* packed_tailT, void* stream);

/* Fused forward: distances -> similarity -> linear head.
 * Replaces _scale_l2_convolution + _l2_convolution (model_multiscale.py:255-317),
 * distance_2_similarity (:324-330), the permute/contiguous/reshape (:369-370) and
 * run_last_layer (:243-244).  Any of distances/activations/logits may be NULL (not written).
 * act_fn: 0 = "log" (log((d+1)/(d+eps))), 1 = "linear" (-d). */
int spx_dist_fwd(const spx_plan* plan, const void* x, int32_t x_dtype, int32_t B, int32_t HW,
                 const void* packed_bank, const float* packed_p2, const void* packed_head,
                 float* distances, float* activations, float* logits,
                 float epsilon, int32_t act_fn, void* stream);

/* Backward, pixel side: recomputes the distance tile, forms
 *   G = (dDist + (dAct + dLogits.W) * act'(d)) * [d > 0]
 * and writes dX = 2 (rowsum_s(G) x - G.P) in X's dtype, plus two scratch buffers for spx_bank_bwd, both opaque to the caller:
 *   g_out  spx_bwd_scratch_bytes(): G as fp16 MFMA fragments with one power-of-two scale per 128-pixel tile;
 *   a_out  spx_bwd_head_scratch_bytes(): what d_W = dLogits^T . A needs - for heads of at most 32 rows the product itself,
 *          formed in this kernel as one fp32 partial per (panel, tile, 32-prototype block); for wider heads the activations
 *          as block-scaled int16 MFMA fragments.
 * Replaces autograd through model_multiscale.py:255-281,324-330,243-244.  d_dist / d_act / d_logits may be NULL (treated as
 * 0); dx may be NULL (X frozen); g_out / a_out may be NULL when the bank / head are frozen.
 *   dx_acc spx_bwd_dx_scratch_bytes() (0 for most plans -> NULL): a scale of more than 192 prototypes is walked as several
 *          panels, each adding its share of dX; with bf16 features that sum runs in this fp32 scratch and dX is rounded
 *          once (NULL: the partial sums pass through the bf16 dX, one rounding per panel). */
int spx_dist_bwd(const spx_plan* plan, const void* x, int32_t x_dtype, int32_t B, int32_t HW,
                 const void* packed_bank, const void* packed_bankT, const float* packed_p2,
                 const void* packed_headT,
                 const float* d_dist, const float* d_act, const float* d_logits,
                 void* dx, void* dx_acc, void* g_out, void* a_out,
                 float epsilon, int32_t act_fn, void* stream);

/* Class-gathered variants: instead of the P-wide distance map, every pixel keeps only the
 * distances to the prototypes of ITS OWN class - the only entries KLDLoss reads for that pixel
 * (segmentation/model/loss.py:89-107; caller segmentation/model/module_multiscale.py:239-242) - so the fp32 map
 * and its gradient never cross HBM.
 *   labels     int32 [B, HW]: class 0..K-1 of the pixel; any other value = no class (its slots read 0)
 *   proto_key  uint32 [npanels * 32*npb], in the plan's padded row order: (class << 16) | slot, slot < J =
 *              rank of the prototype among its class's prototypes (ascending index); 0xFFFFFFFF = none / padding
 *   class_distances fp32 [B, J, HW] (slot planes): entry (slot, px) = distance of pixel px to prototype `slot` of class
 *              labels[px]; entries that no prototype maps to (every slot of a pixel without a class, the slots past its class's
 *              prototype count) are written as 0 by the kernel for classes below 1024 - the planes may arrive uninitialised;
 *              tables with larger class ids: the caller zero-fills.
 * Everything else as in spx_dist_fwd / spx_dist_bwd. */
int spx_dist_fwd_cls(const spx_plan* plan, const void* x, int32_t x_dtype, int32_t B, int32_t HW,
                     const void* packed_bank, const float* packed_p2, const void* packed_head,
                     const int32_t* labels, const uint32_t* proto_key, int32_t J,
                     float* class_distances, float* activations, float* logits,
                     float epsilon, int32_t act_fn, void* stream);
int spx_dist_bwd_cls(const spx_plan* plan, const void* x, int32_t x_dtype, int32_t B, int32_t HW,
                     const void* packed_bank, const void* packed_bankT, const float* packed_p2,
                     const void* packed_headT,
                     const int32_t* labels, const uint32_t* proto_key, int32_t J,
                     const float* d_class_distances, const float* d_act, const float* d_logits,
                     void* dx, void* dx_acc, void* g_out, void* a_out,
                     float epsilon, int32_t act_fn, void* stream);

/* Scale-parallel forward for pixel grids that do not fill the chip (the reference trains on crops: 10 x 65 x 65 latent
 * pixels = 331 tiles for 512+ workgroup slots, each tile walking every scale's panels one after the other).  The scales
 * are independent work (segmentation/model/model_multiscale.py:283-317 is a Python loop over them), so each runs as its
 * own workgroup; only the logits - a sum over ALL prototypes - need the per-scale partials summed, in scale order, by a
 * second small kernel.  spx_fwd_split_groups(): how many groups the library uses for this problem (1 = the single walk);
 * spx_dist_fwd_ws = spx_dist_fwd / spx_dist_fwd_cls (labels_cls / proto_key NULL = the P-wide map) with a workspace of
 * spx_fwd_split_workspace_bytes() for those partials (NULL or 0 bytes: the single walk).  Launches without a logits
 * output (and the pixel-side backward, which has no cross-scale term at all) split on their own. */
int32_t spx_fwd_split_groups(const spx_plan* plan, int32_t B, int32_t HW);
size_t spx_fwd_split_workspace_bytes(const spx_plan* plan, int32_t B, int32_t HW);
int spx_dist_fwd_ws(const spx_plan* plan, const void* x, int32_t x_dtype, int32_t B, int32_t HW,
                    const void* packed_bank, const float* packed_p2, const void* packed_head,
                    const int32_t* labels_cls, const uint32_t* proto_key, int32_t J, float* class_distances,
                    float* distances, float* activations, float* logits, void* split_workspace,
                    float epsilon, int32_t act_fn, void* stream);

/* Pixel-wise cross entropy on the path's logits (segmentation/model/loss.py:9-48, caller
 * segmentation/model/module_multiscale.py:239): CE = mean over the non-ignored pixels of
 * logsumexp_k(logits[px]) - logits[px, label[px]].
 *   labels    int32 [B*HW]: class 0..K-1 of the pixel (the reference's target - 1); any other value = ignored
 *   lse       fp32  [B*HW]: per-pixel logsumexp (forward output, backward input)
 *   pred      int32 [B*HW]: argmax class, lowest index on ties (optional forward output: `correct`, loss.py:43-46)
 *   partials  fp32  [2 * spx_ce_partials(...)]: (sum of the per-pixel losses, number of non-ignored pixels) pairs, one
 *             per wave; the caller sums them (fixed shape, fixed order: deterministic) and divides
 *   logits    the forward's logits (backward input)
 *   coef      DEVICE scalar: dLoss/dCE / number of non-ignored pixels
 *   d_logits_out fp32 [B*HW, K]: coef * (softmax - onehot) (0 on ignored pixels), formed by the pixel-side backward for
 *             spx_bank_bwd's d_logits operand
 * spx_dist_fwd_ce = spx_dist_fwd / spx_dist_fwd_cls (labels_cls / proto_key NULL = the P-wide map) with the CE
 * statistics computed on the logits tile while it is still in registers (8 B/px of extra output instead of separate
 * log_softmax / nll passes over the [B*HW, K] tensor); spx_dist_bwd_ce = spx_dist_bwd / spx_dist_bwd_cls taking
 * (logits, lse, labels, coef) in place of d_logits.  spx_ce_fwd / spx_ce_bwd are the same arithmetic as stand-alone
 * kernels over any [M, K] logits (heads the fused kernels do not carry: more than 160 classes, the grouping tail). */
typedef struct spx_ce {
    const int32_t* labels;
    float* lse;
    int32_t* pred;
    float* partials;
    const float* logits;
    const float* coef;
    float* d_logits_out;
} spx_ce;
size_t spx_ce_partials(int32_t B, int32_t HW);       /* (sum, count) pairs the fused forward writes */
size_t spx_ce_partials_flat(int64_t M);              /* ... and spx_ce_fwd */
int spx_dist_fwd_ce(const spx_plan* plan, const void* x, int32_t x_dtype, int32_t B, int32_t HW,
                    const void* packed_bank, const float* packed_p2, const void* packed_head,
                    const int32_t* labels_cls, const uint32_t* proto_key, int32_t J, float* class_distances,
                    float* distances, float* activations, float* logits, const spx_ce* ce,
                    float epsilon, int32_t act_fn, void* stream);
int spx_dist_bwd_ce(const spx_plan* plan, const void* x, int32_t x_dtype, int32_t B, int32_t HW,
                    const void* packed_bank, const void* packed_bankT, const float* packed_p2,
                    const void* packed_headT,
                    const int32_t* labels_cls, const uint32_t* proto_key, int32_t J,
                    const float* d_dist, const float* d_class_distances, const float* d_act, const spx_ce* ce,
                    void* dx, void* dx_acc, void* g_out, void* a_out,
                    float epsilon, int32_t act_fn, void* stream);
int spx_ce_fwd(const float* logits, const int32_t* labels, int64_t M, int32_t K, float* lse, int32_t* pred,
               float* partials, void* stream);
int spx_ce_bwd(const float* logits, const float* lse, const int32_t* labels, const float* coef, int64_t M, int32_t K,
               float* d_logits, void* stream);
/* mean loss -> loss[0], (count, loss sum) -> count_sum[0..1] from the (sum, count) partial pairs the cross-entropy kernels leave
 * (spx_ce_fwd / the fused epilogues / the grouping tail): one launch, fixed summation order; 0 / 0 = nan as torch's mean over
 * no pixel (loss.py:36-40).  spx_shift_labels: out = labels - 1 as int32 (loss.py:32) from int64 (is_int64 = 1) or int32 labels. */
int spx_ce_finish(const float* partials, int64_t n_pairs, float* loss, float* count_sum, void* stream);
int spx_shift_labels(const void* labels, int32_t is_int64, int64_t n, int32_t* out, void* stream);

/* out [n1, n2] = a^T . b for tall-skinny fp32 operands a [M, n1], b [M, n2] with n1 * n2 <= 8192: the d W_g = d_logits^T . g
 * product of the grouping tail (segmentation/model/model_multiscale_group.py:305-308 through autograd) and its relatives.
 * Deterministic (per-workgroup partials summed in workgroup order).  workspace: spx_pixel_outer_workspace_bytes(). */
size_t spx_pixel_outer_workspace_bytes(int64_t M, int32_t n1, int32_t n2);
int spx_pixel_outer(const float* a, const float* b, int64_t M, int32_t n1, int32_t n2, float* out, void* workspace, void* stream);

/* Heads wider than the distance kernels carry (more than 160 rows: scaleproto_coco.gin's 182 classes, the 450 / 546 grouping
 * units of group_scaleproto_ade.gin / _coco.gin; a grouping tail over more than 32 classes) and the head behind a user-supplied
 * similarity: the three products of the reference's nn.Linear (segmentation/model/model_multiscale.py:243-244,
 * model_multiscale_group.py:283-308, and their autograd) as ONE fp32 MFMA kernel family on row-major device tensors,
 *     C[i][j] = sum_k A(i, k) * B(j, k),   A(i, k) = A[i * a_row_stride + k * a_k_stride]  (B alike),
 * for i < M, j < N, k < K; per operand one of the two strides must be 1.  y = a . w^T: (a, P, 1), (w, P, 1); d_a = g . w:
 * (g, N, 1), (w, 1, P); d_w = g^T . a: (g, 1, N), (a, 1, P).  flags: 1 = A elements enter as exp(A) and 2 = B elements as exp(B)
 * (the grouping tail's exp(units)), 4 = C[i][j] is multiplied by exp(E[i * lde + j]) (its backward d_units).  fp32 operands and
 * accumulation (v_mfma_f32_32x32x2_f32); long contractions over few output tiles are split into workspace slabs summed in a fixed
 * order (deterministic).  workspace: spx_rows_gemm_workspace_bytes() (may be 0 -> NULL accepted). */
size_t spx_rows_gemm_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t flags);
int spx_rows_gemm(const float* A, int64_t a_row_stride, int64_t a_k_stride, const float* B, int64_t b_row_stride,
                  int64_t b_k_stride, float* C, int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t flags, const float* E,
                  int64_t lde, void* workspace, void* stream);

/* Grouping head with the tail as its own kernel: the unit product runs in the distance kernel, then a small fp32 kernel
 * forms g = exp(units), logits = W_g . g and - when `ce` is given - the cross-entropy statistics of the logits it has just
 * formed (partials: spx_ce_partials_flat(B*HW) pairs).  Wg: the RAW last_layer_group.weight [K2, U] fp32.
 * workspace: spx_group_tail_workspace_bytes().  spx_dist_bwd_group_ce = spx_dist_bwd_group with (logits, lse, labels,
 * coef) in place of d_logits; d_logits_out then carries the formed [B*HW, K2] gradient (for d_W_g = d_logits^T . g). */
size_t spx_group_tail_workspace_bytes(const spx_plan* plan, int32_t B, int32_t HW);
int spx_dist_fwd_group_ws(const spx_plan* plan, const void* x, int32_t x_dtype, int32_t B, int32_t HW,
                          const void* packed_bank, const float* packed_p2, const void* packed_head,
                          const float* Wg, int32_t K2, float* distances, float* activations,
                          float* group_activations, float* logits, const spx_ce* ce, void* workspace,
                          float epsilon, int32_t act_fn, void* stream);
int spx_dist_bwd_group_ce(const spx_plan* plan, const void* x, int32_t x_dtype, int32_t B, int32_t HW,
                          const void* packed_bank, const void* packed_bankT, const float* packed_p2,
                          const void* packed_headT_units, const void* packed_tailT, int32_t K2,
                          const float* group_activations, const float* d_dist, const float* d_act,
                          const spx_ce* ce, const float* d_group_activations, float* d_units, void* dx, void* dx_acc, void* g_out,
                          void* a_out, float epsilon, int32_t act_fn, void* stream);

/* Grouping head with its tail fused (segmentation/model/model_multiscale_group.py:283-308, run_last_layer):
 *   units = act . Wd^T   (Wd = the dense [U = G*K', P] form of the per-class group_projection matrices, packed with
 *                         spx_pack_head for a plan whose num_classes = U)
 *   g = exp(units);  logits = g . W_g^T   (W_g = last_layer_group.weight [K2, U], K2 <= 32)
 * spx_pack_group_tail re-packs W_g (packed_tail: forward, packed_tailT: backward, spx_packed_tail_bytes() each);
 * spx_pack_headT_units is spx_pack_head's transposed output with the unit index in accumulator order (the
 * backward builds its dUnits operand in registers): spx_packed_headT_bytes().
 * Forward: logits [B*HW, K2]; group_activations [B*HW, U] = g (optional output, required by the backward).
 * Backward: d_logits [B*HW, K2] in, optionally d_group_activations [B*HW, U] (a gradient on g itself: KLDLossGroup on
 * compute_group's list, segmentation/model/module_multiscale_group_train.py:242-262; NULL = none); d_units [B*HW, U] =
 * (d_logits . W_g + d_group_activations) * g out (the d_logits operand of spx_bank_bwd, which then yields d_Wd);
 * d_W_g = d_logits^T . g is a [K2, U] product left to the caller.
 * spx_exp / spx_exp_bwd: y = exp(x) and dx = g * y over n fp32 elements - the same g for heads the fused kernels do not
 * carry, and for compute_group() on activations that did not come out of the fused forward. */
/* Dense [U, P] head matrix of the grouping head from the per-class projection weights (group_projection[j].weight [g_j, n_j],
 * segmentation/model/model_multiscale_group.py:249-269): out[u][p] = W_j[row_local[u]][col_local[p]] where row_block[u] ==
 * col_block[p] == j, 0 elsewhere (one launch; every element written).  block_ptrs / block_cols: HOST arrays of nblocks (<= 192)
 * DEVICE pointers / column counts n_j; row_block, row_local [U] and col_block, col_local [P]: device int32 tables (block -1 =
 * belongs to no class present).  spx_group_dense_bwd: the adjoint - d_flat[i] = d_out[flat_row[i]][flat_col[i]] for the n weight
 * elements in block order (the caller hands out views of d_flat as the weights' gradients). */
int spx_group_dense(const float* const* block_ptrs, const int32_t* block_cols, int32_t nblocks, const int32_t* row_block,
                    const int32_t* row_local, const int32_t* col_block, const int32_t* col_local, int32_t U, int32_t P, float* out,
                    void* stream);
int spx_group_dense_bwd(const float* d_out, const int32_t* flat_row, const int32_t* flat_col, int64_t n, int32_t P, float* d_flat,
                        void* stream);
size_t spx_packed_tail_bytes(const spx_plan* plan);
int spx_pack_group_tail(const spx_plan* plan, const float* Wg, int32_t K2, void* packed_tail, void* packed_tailT,
                        void* stream);
int spx_pack_headT_units(const spx_plan* plan, const float* W, void* packed_headT_units, void* stream);
int spx_dist_fwd_group(const spx_plan* plan, const void* x, int32_t x_dtype, int32_t B, int32_t HW,
                       const void* packed_bank, const float* packed_p2, const void* packed_head,
                       const void* packed_tail, int32_t K2,
                       float* distances, float* activations, float* group_activations, float* logits,
                       float epsilon, int32_t act_fn, void* stream);
int spx_dist_bwd_group(const spx_plan* plan, const void* x, int32_t x_dtype, int32_t B, int32_t HW,
                       const void* packed_bank, const void* packed_bankT, const float* packed_p2,
                       const void* packed_headT_units, const void* packed_tailT, int32_t K2,
                       const float* group_activations, const float* d_dist, const float* d_act,
                       const float* d_logits, const float* d_group_activations, float* d_units, void* dx, void* dx_acc, void* g_out,
                       void* a_out, float epsilon, int32_t act_fn, void* stream);
int spx_exp(const float* x, float* y, int64_t n, void* stream);
int spx_exp_bwd(const float* g, const float* y, float* dx, int64_t n, void* stream);

/* Bytes of the g_out and of the a_out scratch of spx_dist_bwd. */
size_t spx_bwd_scratch_bytes(const spx_plan* plan, int32_t B, int32_t HW);
size_t spx_bwd_head_scratch_bytes(const spx_plan* plan, int32_t B, int32_t HW);
size_t spx_bwd_dx_scratch_bytes(const spx_plan* plan, int32_t x_dtype, int32_t B, int32_t HW);

/* Backward, parameter side: d_bank [P, Cs] = 2 (p colsum(G) - G^T X) as a pixel-split MFMA reduction with per-workgroup
 * fp32 partial slabs, and d_W [K, P] = dLogits^T A from spx_dist_bwd's a_out (the tile partials of a head of at most 32 rows
 * are summed; for a wider head the product is formed here from the activation fragments and d_logits, which may be NULL
 * otherwise).  Every sum runs in a fixed order (no float atomics: replicas stay bit-identical).
 * workspace: spx_bank_bwd_workspace_bytes().  d_bank / d_W may be NULL. */
size_t spx_bank_bwd_workspace_bytes(const spx_plan* plan, int32_t B, int32_t HW);
int spx_bank_bwd(const spx_plan* plan, const void* x, int32_t x_dtype, int32_t B, int32_t HW,
                 const float* bank, const void* g_in, const void* a_in, const float* d_logits,
                 float* d_bank, float* d_W, void* workspace, void* stream);

/* Class-masked per-prototype argmin over the latent grid (prototype push).
 * Replaces the one_hot / matmul / masked add / two min() reductions of
 * segmentation/push_multiscale_optimization.py:74-91.  labels: int32 [B, HW], already resized with
 * resize_label (dataset.py:22-30); label == void_class (or outside 0..K) matches no prototype;
 * void_class < 0 means labels are 0..K-1 with no void.  class_identity: fp32 [P, K].
 * Outputs: indices int64 [B, P] (flat i*W+j, lowest index on ties), values fp32 [B, P].
 * scratch: uint64 [B*P]. */
int spx_push_argmin(const float* distances, const int32_t* labels, const float* class_identity,
                    int32_t B, int32_t P, int32_t K, int32_t HW, int32_t void_class, float max_dist,
                    int64_t* indices, float* values, uint64_t* scratch, void* stream);

/* The same reduction FUSED into the distance kernel (the P-wide distance map is never written): features in, per
 * (image, prototype) minimum + flat index out.  labels / void_class / K: exactly as spx_push_argmin takes them (raw int32
 * [B, HW] labels; label == void_class or outside the K classes matches no prototype; void_class < 0: labels are 0..K-1);
 * proto_key: the (class << 16 | slot) table of spx_dist_fwd_cls (class of every padded prototype row) in place of the
 * fp32 class_identity - i.e. a one-hot prototype_class_identity, which is what the reference builds
 * (model_multiscale.py:89-97).
 * Arithmetic and tie rule as spx_push_argmin (distances as spx_dist_fwd computes them; d + max_dist * (1 - mask) rounded
 * as the reference rounds it; lowest flat index on ties; absent class -> (0, max_dist)).  Integer atomic minima:
 * run-to-run identical.  scratch: uint64 [B*P]. */
int spx_dist_push_min(const spx_plan* plan, const void* x, int32_t x_dtype, int32_t B, int32_t HW, const void* packed_bank,
                      const float* packed_p2, const int32_t* labels, int32_t void_class, int32_t K, const uint32_t* proto_key,
                      float max_dist, int64_t* indices, float* values, uint64_t* scratch, void* stream);

/* Lexicographic (value, image) argmin over images per prototype: tot_dist.argmin(dim=0),
 * push_multiscale_optimization.py:135-137.  values fp32 [N, P] -> best int64 [P]. */
int spx_argmin_images(const float* values, int32_t N, int32_t P, int64_t* best, void* stream);

/* KLD loss over class-gathered distances (segmentation/model/loss.py:51-146).  vals fp32 [B, J, HW]
 * (the slot planes of spx_dist_fwd_cls), labels int32 [B, HW].  A segment = (image, class).  Four streaming passes,
 * each reading vals once; all segment reductions use integer atomics (run-to-run identical results):
 *   spx_kld_segment_max    smax_keys uint32 [B, K, J] (caller zero-fills): ordered key of max_px vals over the segment
 *                          (key k -> float: k & 0x80000000 ? k ^ 0x80000000 : ~k); counts uint32 [B, K] (zero-filled,
 *                          may be NULL): pixels per segment (loss.py:113-127 skips segments of fewer than two);
 *                          range_keys uint32 [2] (zero-filled, may be NULL): ordered keys of max(v) and max(-v) over
 *                          every value that enters a segment (the value range the fixed-point scale below is sized for)
 *   spx_kld_segment_sumexp ssum_fx uint64 [B, K, J] (zero-filled): sum_px exp(vals - smax) * 2^40, smax from the keys
 *   spx_kld_segment_lse    lse fp32 [n = B*K*J]: smax + log(ssum_fx / 2^40), 0 for a segment without pixels; with `scale`
 *                          (ONE double in device memory; needs range_keys and HW) also the fixed-point scale of the
 *                          pair sums: the power of two 2^floor(log2(2^61 / (HW * (max - min + 32)))) - no host sync
 *   spx_kld_pair_sums      a_fx int64 [B, K, J, J] (zero-filled): sum_px p_j * (l_k - l_j) * scale = -KL(j || k) of the
 *                          segment, i.e. the Gram matrix sum_px p_j l_k minus its row's diagonal entry (diagonal 0);
 *                          scale: ONE double in DEVICE memory, so the caller can derive it from the data without a
 *                          host sync; l = vals - lse (the log_softmax over the segment's pixels, loss.py:110), p = exp(l).
 *   spx_kld_backward       grad fp32 [B, J, HW] = dLoss/dvals given A = a_fx / scale and Cf = dLoss/dA [B, K, J, J]
 *                          (diagonal entries of Cf are not read: A's diagonal is identically 0)
 * W (the three reduction passes): row length of the pixel grid (must divide HW) lets a wave take the pixels as compact
 * 16 x 4 blocks down a 16-pixel column strip, which cross fewer class boundaries than a row segment (partial results are
 * published per class run); 0 = unknown
 * (linear walk).  W changes only the rounding of fp32 partial sums.
 *   spx_kld_gram_loss      the [B, K, J, J]-sized algebra between the passes (loss.py:113-142), one workgroup per segment
 *                          + a one-workgroup finish: A = a_fx / scale [nseg = B*K, J, J]; kld_jk = (A_jj + A_kk - A_jk -
 *                          A_kj) / 2; an entry is valid when pair_ok[class][j][k] (uint8 [K, J, J]: j < k, both slots exist,
 *                          same scale) and the segment has >= 2 pixels; loss[0] = mean over the valid entries of exp(-kld)
 *                          (0 if none), loss[1] = 1 / max(number of valid entries, 1); Cf = dLoss/dA WITHOUT that factor
 *                          (spx_kld_backward applies it: cf_scale); partials: double [2 * nseg] scratch.  Fixed-order sums.
 *   spx_kld_backward       cf_scale: ONE float in device memory that multiplies Cf (NULL = 1): the caller's
 *                          dLoss_total/dLoss times loss[1], formed on the device
 * J <= 16 and K*J*J*8 bytes must fit the LDS table (~60 KiB). */
int spx_kld_segment_max(const float* vals, const int32_t* labels, int32_t B, int32_t J, int32_t HW, int32_t W, int32_t K,
                        uint32_t* smax_keys, uint32_t* counts, uint32_t* range_keys, void* stream);
int spx_kld_segment_sumexp(const float* vals, const int32_t* labels, int32_t B, int32_t J, int32_t HW, int32_t W, int32_t K,
                           const uint32_t* smax_keys, uint64_t* ssum_fx, void* stream);
int spx_kld_segment_lse(const uint32_t* smax_keys, const uint64_t* ssum_fx, int32_t n, float* lse, const uint32_t* range_keys,
                        int32_t HW, double* scale, void* stream);
int spx_kld_gram_loss(const int64_t* a_fx, const double* scale, const uint32_t* counts, const uint8_t* pair_ok, int32_t nseg, int32_t K,
                      int32_t J, float* A, float* Cf, double* partials, float* loss, void* stream);
int spx_kld_pair_sums(const float* vals, const int32_t* labels, int32_t B, int32_t J, int32_t HW, int32_t W, int32_t K,
                      const float* lse, const double* scale, int64_t* a_fx, void* stream);
int spx_kld_backward(const float* vals, const int32_t* labels, int32_t B, int32_t J, int32_t HW, int32_t K,
                     const float* lse, const float* A, const float* Cf, const float* cf_scale, float* grad, void* stream);

/* Evaluation maps: F.interpolate(src, size=(H, W), mode="bilinear", align_corners=False) followed
 * by argmin (take_max = 0) or argmax (take_max = 1) over the channel dimension, without materialising the
 * upsampled [N, C, H, W] tensor (segmentation/eval_valid_multiscale.py:229-234, :375-383: nearest prototype per
 * full-resolution pixel from the distance map, predicted class from the logits).  src fp32 [N, C, h, w];
 * indices int64 [N, H, W] (lowest channel on ties); values fp32 [N, H, W] (the extremum; may be NULL). */
int spx_upsample_argext(const float* src, int32_t N, int32_t C, int32_t h, int32_t w, int32_t H, int32_t W,
                        int32_t take_max, int64_t* indices, float* values, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPX_HIP_H */
