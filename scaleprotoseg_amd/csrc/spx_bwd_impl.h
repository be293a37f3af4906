// Backward of the prototype-distance path (replaces autograd through
// segmentation/model/model_multiscale.py:255-281, :324-330, :243-244).
//
// With d_raw = |x|^2 - 2 x.p + |p|^2, d = relu(d_raw), a = act(d), logits = a.W^T:
//   gA = dAct + dLogits.W                                 [pixel x proto]
//   G  = (dDist + gA * act'(d)) * [d_raw > 0]
//   dX[m, c in s] = 2 (x[m,c] * rowsum_s(G)[m] - (G.P)[m,c])
//   dP[p, c]      = 2 (p[p,c] * colsum(G)[p]   - (G^T.X)[p,c])
//   dW[k, p]      = (dLogits^T . a)[k,p]
//
// Kernel 1 (pixel side, spx_bwd_kernel): same tiling and main loop as the forward (the x.p tile is
// recomputed with identical arithmetic, so the relu mask is the forward's bit for bit — cheaper than
// re-reading the fp32 distance map: 2*P*C flop/px on the matrix pipe vs 4*P bytes/px of HBM).
//   phase 1: G and a in accumulator layout (lane = pixel).  G is packed to fp16 MFMA B-fragments, dumped verbatim
//            ("blobs": 1 KiB per 32 px x 16 prototypes, one 16-B store per lane) for kernel 2, and kept in registers for
//            phase 2.  The activations feed d_W = dLogits^T . a: for heads of one class block (K <= 32) right here - the
//            "d_W stage": a block's activations are turned through LDS so that the PIXEL becomes the MFMA k, and the
//            workgroup leaves one fp32 [K][32] partial per prototype block and tile (summed in a fixed order by
//            spx_dw_reduce_kernel / kernel 3); for wider heads as a 16-bit blob for kernel 2 (spx_common.h).
//   phase 2: dX^T[ch x px] = P^T . G with the G fragments as B operand (accumulator -> operand, no LDS) and
//            the P^T fragments streamed through LDS; the result is transposed through LDS so that X is read
//            and dX written in whole 256-B pixel rows.
// Kernel 2 (parameter side, spx_bank_bwd_kernel): pixel-split MFMA reduction G^T.X (and a^T.dLogits for the wide heads);
// the blobs are laid out [pixel][prototype] in LDS and read back with ds_read_b64_tr_b16 (pixel becomes the MFMA k).
// Per-workgroup fp32 partial slabs, summed in a fixed order by kernel 3 (no float atomics).
#pragma once
#include "spx_args.h"
#include "spx_mainloop.h"
#include <type_traits>

#define SPX_BWD_WAVES 2                   // two workgroups per CU (256 registers per wave)
#define SPX_BWD_XRING(xf32) 2             // X chunks in flight ahead of the MFMAs

#define SPX_T_ROW 528                     // fp32 transpose tile row: 128 px * 4 B + 16 B pad
#define SPX_T_BYTES (32 * SPX_T_ROW)

// LDS carve of the pixel kernel (two workgroups per CU need <= 80 KiB each):
//   region 0: main loop: 2 stages | phase 1: the d_W stage's images (+ the dAct turn scratch) | phase 2: 2 P^T stages
//             (NPB * 2 KiB each) + transpose tile 0
//   region 1: head^T fragments (phase 1 only) | phase 2: transpose tile 1
//   |p|^2 of the panel, rowsum(G) of the tile
// d_W stage (one class block): [dL image 16 KiB][activation image(s) 16 KiB each][dAct scratch], see the kernel.
#define SPX_DW_IMG 16384                  // one image: 4 waves x 2 cells x (hi, lo) x 1 KiB
#define SPX_DACT_SC 4352                  // per wave: 32 x 33 floats + pad
template <int NCB, bool DACT>
__host__ __device__ constexpr int spx_bwd_dw_nabuf() { return DACT ? 1 : 2; }
template <int NCB, bool DACT>
__host__ __device__ constexpr int spx_bwd_phase1_bytes() {
    return (NCB == 1 ? (1 + spx_bwd_dw_nabuf<NCB, DACT>()) * SPX_DW_IMG : 0) + (DACT ? 4 * SPX_DACT_SC : 0);
}
template <int NPB>
__host__ __device__ constexpr int spx_bwd_bt_bytes() { return NPB * 2 * 1024; }
template <int NPB, int NCB>
__host__ __device__ constexpr int spx_bwd_head_lds_bytes() { return NCB * NPB <= 6 ? NPB * NCB * 2 * 2048 : 0; }   // <= 24 KiB
template <int NPB, int NCB, bool DACT>
__host__ __device__ constexpr int spx_bwd_region0_bytes() {
    constexpr int a = 2 * spx_stage_bytes(NPB);
    constexpr int b = 2 * spx_bwd_bt_bytes<NPB>() + SPX_T_BYTES;
    constexpr int c = spx_bwd_phase1_bytes<NCB, DACT>();
    return (a > b ? a : b) > c ? (a > b ? a : b) : c;
}
template <int NPB, int NCB>
__host__ __device__ constexpr int spx_bwd_region1_bytes() {
    constexpr int h = spx_bwd_head_lds_bytes<NPB, NCB>();
    return h > SPX_T_BYTES ? h : SPX_T_BYTES;
}
template <int NPB, int NCB, bool DACT>
__host__ __device__ constexpr int spx_bwd_lds_bytes() {
    return spx_bwd_region0_bytes<NPB, NCB, DACT>() + spx_bwd_region1_bytes<NPB, NCB>() + 3 * NPB * 32 * 4 + SPX_TILE_PX * 4 + 32;   // + |p|^2, class keys, slot plane offsets, rowsum(G)
}
static_assert(spx_bwd_lds_bytes<6, 1, false>() <= 80 * 1024 && spx_bwd_lds_bytes<6, 1, true>() <= 80 * 1024,
              "pixel kernel must fit two workgroups per CU");
// bf16 elements of one G (or a) scratch: [panel][tile][wave][pb][s2] fragments of 512 elements

// ------------------------------------------------------------------------------------------------
// kernel 1: pixel side
// ------------------------------------------------------------------------------------------------
// GATHER: the distance gradient arrives class-gathered ([B, HW, J], spx_dist_bwd_cls) instead of P-wide.
// DACT: a gradient arrives on the [pixel][P] activations (kept out of the default instance).
// ACC: bf16 features and a scale that spans several panels: its partial dX is summed in the fp32 scratch a.dx_acc (its own
// instances: as a run-time branch the code cost the default instance 0.02-0.035 ms - it sits at the register cliff).
template <int NPB, int NCB, bool XF32, int VM, bool GATHER, bool DACT, bool ACC = false>
__global__ __launch_bounds__(256, SPX_BWD_WAVES) void spx_bwd_kernel(const SpxBwdArgs a) {
    constexpr bool VEC = VM != 0, RAG = VM == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const spx_plan& pl = a.plan;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int tiles_per_img = (a.HW + SPX_TILE_PX - 1) / SPX_TILE_PX;
    const int b = blockIdx.x / a.tiles_launch;
    const int tile_i = a.tile_first + (int)(((long long)(blockIdx.x % a.tiles_launch) * a.tile_mul) % a.tiles_launch);
    const int px0 = tile_i * SPX_TILE_PX;
    const size_t ntiles = (size_t)a.B * tiles_per_img;
    const int Cs = pl.channels_per_scale;
    const int C = pl.num_scales * Cs;
    const int P = pl.num_prototypes, K = pl.num_classes;
    const int nchb = (Cs + 31) / 32;
    constexpr int ncstep = NCB * 2;
    const uint32_t HW = (uint32_t)a.HW;
    constexpr int ESZ = XF32 ? 4 : 2;
    constexpr int XR = SPX_BWD_XRING(XF32);
    using Pipe = SpxPipeline<NPB, XF32, VM, XR>;

    const char* x_img = (const char*)a.x + (size_t)b * C * a.HW * ESZ;
    const SpxTileCtx tc = SpxXStager<XF32, VM>::make_ctx(x_img, a.HW, px0, tid);

    constexpr int stage = spx_stage_bytes(NPB);
    constexpr int chunk_bytes = NPB * 2 * 1024;
    constexpr int head_lds = spx_bwd_head_lds_bytes<NPB, NCB>();
    const int nchunks = (Cs + SPX_KC - 1) / SPX_KC;
    char* const hlds = smem + spx_bwd_region0_bytes<NPB, NCB, DACT>();
    float* const p2s = (float*)(hlds + spx_bwd_region1_bytes<NPB, NCB>());
    uint32_t* const keys = (uint32_t*)(p2s + NPB * 32);    // GATHER: (class << 16) | slot per padded prototype row
    uint32_t* const koff = keys + NPB * 32;                // GATHER: byte offset of the row's slot plane (slot * HW * 4)
    float* const rss = p2s + 3 * NPB * 32;
    float* const gmaxs = rss + SPX_TILE_PX;                // [4] wave maxima of |G| (the tile's fp16 scale)
    // d_W stage (one class block, see the epilogue): dL image, activation image(s), then the dAct turn scratch
    constexpr bool DW = NCB == 1;
    constexpr int NABUF = spx_bwd_dw_nabuf<NCB, DACT>();
    char* const l_img = smem;
    char* const a_img = smem + SPX_DW_IMG;
    char* const dact_sc = smem + (DW ? (1 + NABUF) * SPX_DW_IMG : 0);
    const bool want_dw = DW && a.a_out != nullptr;          // workgroup-uniform

    const int px = px0 + 32 * wave + r;
    const bool px_ok = px < a.HW;
    const uint32_t voff_d = px_ok ? ((uint32_t)(4 * h) * HW + (uint32_t)px) * 4u : SPX_OOB;            // [row][px] fp32
    const uint32_t voff_a = px_ok ? ((uint32_t)px * (uint32_t)P + (uint32_t)(4 * h)) * 4u : SPX_OOB;   // [px][row] fp32
    const spx_rsrc htr = make_rsrc(a.packed_headT);
    const spx_rsrc btr = make_rsrc_pred(a.packed_bankT);
    const spx_rsrc p2r = make_rsrc(a.p2);
    const spx_rsrc htp = make_rsrc_pred(a.packed_headT);
    const spx_rsrc p2p = make_rsrc_pred(a.p2);
    // GATHER: this pixel's class (0xFFFE = none; padding rows carry class 0xFFFF) and its [px][slot] row offset
    uint32_t lab16 = 0xFFFEu, voff_c = SPX_OOB;
    const spx_rsrc keyr = make_rsrc_pred(GATHER ? a.proto_key : nullptr);
    const spx_rsrc cdr = make_rsrc_pred((GATHER && a.d_cls_dist) ? a.d_cls_dist + (size_t)b * a.J * a.HW : nullptr);
    if (GATHER && a.d_cls_dist) {
        const spx_rsrc labr = make_rsrc_pred(a.labels + (size_t)b * a.HW);
        const uint32_t l = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(labr, px_ok ? (uint32_t)px * 4u : SPX_OOB, 0, 0);
        lab16 = (px_ok && l < 0xFFFEu) ? l : 0xFFFEu;
        voff_c = px_ok ? (uint32_t)px * 4u : SPX_OOB;          // [slot][px] planes
    }
    const bool have_dd = GATHER ? a.d_cls_dist != nullptr : a.d_dist != nullptr;

    const bool have_dl = a.d_logits != nullptr || a.ce_labels != nullptr;     // a gradient reaches the logits
    // scale-parallel launch (grid.y = scale group): this workgroup's panels; group 0 alone writes the per-pixel by-products
    const int q_begin = a.ngroups > 1 ? a.group_first[blockIdx.y] : 0;
    const int q_end = a.ngroups > 1 ? a.group_first[blockIdx.y + 1] : pl.npanels;
    const bool g0 = blockIdx.y == 0;
    const bool act_is_log = a.act_fn == 0;
    const float act_c1 = act_is_log ? -(1.0f - a.eps) : -1.0f;
    // dLogits of this lane's pixel as split-bf16 B fragments: element j of k-step c <-> class 16c + 8h + j
    bf16x8 dlhi[NCB * 2], dllo[NCB * 2];
    // [pixel][n] fp32 tensors (dLogits, group activations, dUnits): a wave's 32 pixels x n values are ONE contiguous
    // block in memory; it moves as coalesced dword accesses through the wave's LDS scratch (the stages are idle until
    // the first panel) instead of n strided accesses per lane (0.9 ms per 2 Mpx for n = 57, measured).
    constexpr int BSC_BYTES = 8192;                       // per wave: 32 px x (<= 64) values
    constexpr bool BLK = NCB <= 2;
    float* const bsc = (float*)(smem + wave * BSC_BYTES);
    const int pxw0 = px0 + 32 * wave;
    const int npx_w = a.HW - pxw0 < 32 ? (a.HW - pxw0 > 0 ? a.HW - pxw0 : 0) : 32;
    auto block_fetch = [&](const float* gimg, int n) {    // bsc <- [32][n] of image b (zeros past the image)
        const spx_rsrc rs = make_rsrc_pred(gimg + (size_t)pxw0 * n);
        const int nvalid = npx_w * n;
        // every load is issued before the first LDS write (a rolled load -> write loop would expose one memory round
        // trip per iteration); n <= 32 NCB, so NCB * 16 passes of 64 lanes cover the block
        float v[NCB * 16];
#pragma unroll
        for (int it = 0; it < NCB * 16; ++it) {
            const int i = lane + 64 * it;
            v[it] = buf_load_f32(rs, i < nvalid ? (uint32_t)i * 4u : SPX_OOB, 0);
        }
#pragma unroll
        for (int it = 0; it < NCB * 16; ++it) {
            const int i = lane + 64 * it;
            if (i < 32 * n) bsc[i] = v[it];
        }
    };
    auto block_flush = [&](float* gimg, int n) {
        const spx_rsrc rs = make_rsrc_pred(gimg + (size_t)pxw0 * n);
        const int nvalid = npx_w * n;
#pragma unroll 1
        for (int i = lane; i < 32 * n; i += 64) buf_store_f32(bsc[i], rs, i < nvalid ? (uint32_t)i * 4u : SPX_OOB, 0);
    };
    if (a.packed_tailT) {
        // grouping-head tail: d_logits is [px][K2]; dUnits[u, px] = (sum_k W_g[k, u] dLogits[px, k]) * g[px, u] is
        // formed here as accumulator tiles (rows = units), written out for the parameter kernel, and becomes the
        // B operand of the head^T product below in ACCUMULATOR row order (packed_headT is the _units variant).
        const int K2 = a.K2;
        // d_logits [px][K2]: given, or - fused cross entropy - formed from the forward's logits (see the plain branch below)
        const bool ce = a.ce_labels != nullptr;
        const float* const lsrc = ce ? a.ce_logits : a.d_logits;
        const spx_rsrc lr = make_rsrc_pred(lsrc + (size_t)b * a.HW * K2);
        const uint32_t voff_l = px_ok ? ((uint32_t)px * (uint32_t)K2 + (uint32_t)(8 * h)) * 4u : SPX_OOB;
        float ce_lse = 0.0f, ce_c = 0.0f;
        int ce_lab = -1;
        if (ce) {
            const uint32_t vo1 = px_ok ? (uint32_t)px * 4u : SPX_OOB;
            ce_lab = (int)__builtin_amdgcn_raw_buffer_load_b32(make_rsrc_pred(a.ce_labels + (size_t)b * a.HW), vo1, 0, 0);
            ce_lse = buf_load_f32(make_rsrc_pred(a.ce_lse + (size_t)b * a.HW), vo1, 0);
            ce_c = (px_ok && (uint32_t)ce_lab < (uint32_t)K2) ? *a.ce_coef : 0.0f;
        }
        const spx_rsrc dor = make_rsrc_pred((ce && a.ce_dlogits_out) ? a.ce_dlogits_out + (size_t)b * a.HW * K2 : nullptr);
        bf16x8 l2hi[2], l2lo[2];
        if (BLK) block_fetch(lsrc + (size_t)b * a.HW * K2, K2);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int cls = c * 16 + 8 * h + j;
                float v;
                if (BLK) v = cls < K2 ? bsc[r * K2 + cls] : 0.0f;
                else v = buf_load_f32(lr, cls < K2 ? voff_l : SPX_OOB, (uint32_t)((c * 16 + j) * 4));
                if (ce) {
                    v = cls < K2 ? ce_c * (ce_exp(v - ce_lse) - (cls == ce_lab ? 1.0f : 0.0f)) : 0.0f;
                    if (BLK) {
                        if (cls < K2) bsc[r * K2 + cls] = v;
                    } else {
                        buf_store_f32(v, dor, (cls < K2 && g0) ? voff_l : SPX_OOB, (uint32_t)((c * 16 + j) * 4));
                    }
                }
                __bf16 hi, lo;
                split_bf16(v, hi, lo);
                l2hi[c][j] = hi;
                l2lo[c][j] = lo;
            }
        }
        if (BLK && ce && a.ce_dlogits_out && g0) block_flush(a.ce_dlogits_out + (size_t)b * a.HW * K2, K2);
        const spx_rsrc ttr = make_rsrc(a.packed_tailT);
        const spx_rsrc gir = make_rsrc_pred(a.gact + (size_t)b * a.HW * K);
        const spx_rsrc dur = make_rsrc_pred(a.d_units + (size_t)b * a.HW * K);
        // a gradient that arrives on the group activations themselves (KLDLossGroup on compute_group's list,
        // module_multiscale_group_train.py:242-262): dUnits = (W_g^T . dLogits + dG) * g.  Plain per-lane loads: the step that
        // carries this term runs on crops, and the coalescing scratch is taken by g / dUnits.
        const spx_rsrc dgr = make_rsrc_pred(a.d_gact ? a.d_gact + (size_t)b * a.HW * K : nullptr);
        const uint32_t voff_u = px_ok ? ((uint32_t)px * (uint32_t)K + (uint32_t)(4 * h)) * 4u : SPX_OOB;     // [px][unit]
        if (BLK) block_fetch(a.gact + (size_t)b * a.HW * K, K);       // g of the wave's pixels; dUnits overwrite it in place
#pragma unroll
        for (int ub = 0; ub < NCB; ++ub) {
            f32x16 dg;
#pragma unroll
            for (int i = 0; i < 16; ++i) dg[i] = 0.0f;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const uint32_t so = (uint32_t)((ub * 2 + c) * 2048);
                const bf16x8 whi = __builtin_bit_cast(bf16x8, buf_load_b128(ttr, (uint32_t)lane * 16u, so));
                const bf16x8 wlo = __builtin_bit_cast(bf16x8, buf_load_b128(ttr, (uint32_t)lane * 16u, so + 1024u));
                dg = mfma_bf16(whi, l2hi[c], dg);
                dg = mfma_bf16(wlo, l2hi[c], dg);
                dg = mfma_bf16(whi, l2lo[c], dg);
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int u = ub * 32 + acc_row(reg, h);
                const uint32_t so = (uint32_t)((ub * 32 + (reg & 3) + 8 * (reg >> 2)) * 4);
                if (a.d_gact) dg[reg] += buf_load_f32(dgr, u < K ? voff_u : SPX_OOB, so);     // (wave-uniform branch)
                if (BLK) {
                    const float gval = u < K ? bsc[r * K + u] : 0.0f;     // zeros past the image: dropped loads
                    dg[reg] *= gval;
                    if (u < K) bsc[r * K + u] = dg[reg];
                } else {
                    const float gval = buf_load_f32(gir, u < K ? voff_u : SPX_OOB, so);
                    dg[reg] *= gval;                               // dropped loads return 0: padded units / pixels
                    buf_store_f32(dg[reg], dur, (u < K && g0) ? voff_u : SPX_OOB, so);
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    __bf16 hi, lo;
                    split_bf16(act_c1 * dg[8 * s2 + j], hi, lo);
                    dlhi[ub * 2 + s2][j] = hi;
                    dllo[ub * 2 + s2][j] = lo;
                }
            }
        }
        if (BLK && g0) block_flush(a.d_units + (size_t)b * a.HW * K, K);
    } else {
        // d_logits of the wave's pixels: given ([px][K] fp32), or - fused cross entropy - formed here from the forward's
        // logits: coef * (softmax - onehot) on the non-ignored pixels (loss.py:9-48 through autograd), written out once
        // for the parameter kernel
        const bool ce = a.ce_labels != nullptr;
        const float* const lsrc = ce ? a.ce_logits : a.d_logits;
        const spx_rsrc lr = make_rsrc_pred(lsrc ? lsrc + (size_t)b * a.HW * K : nullptr);
        const uint32_t voff_l = (lsrc && px_ok) ? ((uint32_t)px * (uint32_t)K + (uint32_t)(8 * h)) * 4u : SPX_OOB;
        float ce_lse = 0.0f, ce_c = 0.0f;
        int ce_lab = -1;
        if (ce) {
            const uint32_t vo1 = px_ok ? (uint32_t)px * 4u : SPX_OOB;
            ce_lab = (int)__builtin_amdgcn_raw_buffer_load_b32(make_rsrc_pred(a.ce_labels + (size_t)b * a.HW), vo1, 0, 0);
            ce_lse = buf_load_f32(make_rsrc_pred(a.ce_lse + (size_t)b * a.HW), vo1, 0);
            ce_c = (px_ok && (uint32_t)ce_lab < (uint32_t)K) ? *a.ce_coef : 0.0f;
        }
        const spx_rsrc dor = make_rsrc_pred((ce && a.ce_dlogits_out) ? a.ce_dlogits_out + (size_t)b * a.HW * K : nullptr);
        if (BLK && lsrc) block_fetch(lsrc + (size_t)b * a.HW * K, K);
#pragma unroll
        for (int c = 0; c < NCB * 2; ++c) {
            u32x4 hw, lw;
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
                f32x2 v2;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int j = 2 * j2 + e;
                    const int cls = c * 16 + 8 * h + j;
                    float v;
                    if (BLK) v = (lsrc && cls < K) ? bsc[r * K + cls] : 0.0f;
                    else v = buf_load_f32(lr, cls < K ? voff_l : SPX_OOB, (uint32_t)((c * 16 + j) * 4));
                    if (ce) {
                        v = cls < K ? ce_c * (ce_exp(v - ce_lse) - (cls == ce_lab ? 1.0f : 0.0f)) : 0.0f;
                        if (BLK) {
                            if (cls < K) bsc[r * K + cls] = v;
                        } else {
                            buf_store_f32(v, dor, (cls < K && g0) ? voff_l : SPX_OOB, (uint32_t)((c * 16 + j) * 4));
                        }
                    }
                    v2[e] = v;
                }
                // pre-scaled by the constant factor of act'(d) (log: -(1-eps) / ((d+1)(d+eps)); linear: -1), so the
                // element loop multiplies by 1/((d+1)(d+eps)) only
                uint32_t hi, lo;
                split_bf16x2(v2 * act_c1, hi, lo);
                hw[j2] = hi;
                lw[j2] = lo;
            }
            dlhi[c] = __builtin_bit_cast(bf16x8, hw);
            dllo[c] = __builtin_bit_cast(bf16x8, lw);
        }
        if (BLK && ce && a.ce_dlogits_out && g0) block_flush(a.ce_dlogits_out + (size_t)b * a.HW * K, K);
    }
    if (BLK) __syncthreads();      // the scratch sits in the main-loop stages: every wave is done with it before they fill

    Pipe pipe;
    f32x16 acc[NPB];
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[pb][i] = 0.0f;
    float x2part = 0.0f;

    constexpr int HPASS = head_lds / 4096;
    u32x4 hreg[HPASS > 0 ? HPASS : 1];
    float p2reg = 0.0f;
    uint32_t keyreg = 0xFFFFFFFFu;
    auto consts_issue = [&](int panel) {
#pragma unroll
        for (int i = 0; i < HPASS; ++i)
            hreg[i] = buf_load_b128(htp, have_dl ? (uint32_t)(i * 4096 + tid * 16) : SPX_OOB, (uint32_t)(panel * head_lds));
        p2reg = buf_load_f32(p2p, tid < NPB * 32 ? (uint32_t)tid * 4u : SPX_OOB, (uint32_t)(panel * NPB * 32 * 4));
        if (GATHER) keyreg = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(keyr, tid < NPB * 32 ? (uint32_t)tid * 4u : SPX_OOB, (uint32_t)(panel * NPB * 32 * 4), 0);
    };
    auto consts_commit = [&]() {
#pragma unroll
        for (int i = 0; i < HPASS; ++i) *(u32x4*)(hlds + i * 4096 + tid * 16) = hreg[i];
        if (tid < NPB * 32) p2s[tid] = p2reg;
        if (GATHER && tid < NPB * 32) {
            keys[tid] = keyreg;
            koff[tid] = (keyreg & 0xFFFFu) * HW * 4u;
        }
    };

#ifdef SPX_DIAG_STAMPS
    unsigned long long dg_t2 = 0;
#endif

    // ---------------- panel epilogue ----------------
    auto epilogue = [&](int panel) {
        const float x2 = x2part + __shfl_xor(x2part, 32);
        const int p0 = pl.panel_p0[panel], np = pl.panel_np[panel];
        const int ch0 = pl.panel_ch0[panel];
        const int nv = (np + 31) >> 5;                   // prototype blocks holding >= 1 real prototype
        const size_t tile_g = (size_t)b * tiles_per_img + tile_i;
        const size_t blob0 = (((size_t)panel * ntiles + tile_g) * 4) * NPB * 2 * 1024;   // bytes
        const spx_rsrc gr = make_rsrc(a.g_out ? (const char*)a.g_out + blob0 : nullptr);
        const size_t blob_total = (size_t)pl.npanels * ntiles * 4 * NPB * 2 * 1024;
        // wide heads: the activation blob, its block exponents (one word per (lane, block), [panel][tile][wave][block][lane])
        // behind the blobs, the head scale behind those
        const bool want_ab = !DW && a.a_out != nullptr;
        const spx_rsrc ar = make_rsrc(want_ab ? (const char*)a.a_out + blob0 : nullptr);
        const spx_rsrc asr = make_rsrc(want_ab ? (const char*)a.a_out + blob_total + blob0 / 8 : nullptr);
        if (a.a_out && blockIdx.x == 0 && blockIdx.y == 0 && panel == 0 && tid == 0) {
            // the head scale (spx_common.h): the d_W stage's partials carry a / ln 2 times c1 dLogits, the blob a / ln 2
            if (DW) *(float*)((char*)a.a_out + spx_dw_partial_bytes(pl.npanels, ntiles, NPB, K)) = 0.69314718056f / act_c1;
            else *(float*)((char*)a.a_out + spx_ablob_scale_offset(blob_total)) = 0.69314718056f;
        }
        // ---- d_W stage, set-up (one class block): the wave's dLogits fragments (split bf16, pre-scaled by c1) as a
        // [pixel][class] image, cell-major: cell = 16 classes x 32 pixels of ONE wave x (hi | lo) = 1 KiB with 32-B pixel rows,
        // so that ds_read_b64_tr_b16 hands a lane 8 PIXELS of its class (the pixel becomes the MFMA k).  Rows are permuted
        // (bit 2 flipped where bit 3 is set): the two row groups a transposed read touches per 32-lane half, 8 rows apart,
        // then fall on disjoint banks.  The stages are idle from here to phase 2; rebuilt per panel (they are not in between).
        const int rperm = r ^ (((r >> 3) & 1) << 2);
        if (want_dw) {
            char* const lw = l_img + wave * 4096 + rperm * 32 + h * 16;
#pragma unroll
            for (int c = 0; c < (DW ? 2 : 0); ++c) {
                *(bf16x8*)(lw + c * 2048) = dlhi[c];
                *(bf16x8*)(lw + c * 2048 + 1024) = dllo[c];
            }
        }

        // ---- phase 1: G, a — a ROLLED loop over the panel's 32-prototype blocks.  The block being processed is
        // always acc[0]; acc is rotated after each block and the block's G enters a register queue, so the body is
        // compiled once with a fixed register footprint (an unrolled version spilled hundreds of registers).  Exactly NPB
        // iterations, so the queue ends aligned.  G (fp32, 16 registers per block) is stored in the accumulator slot that
        // the rotation frees, so after NPB iterations acc[i] holds block i's G and no second register array is needed.
        float rs = 0.0f, gmax = 0.0f;
        float ddA[16], ddB[16];     // dDist of the current / next block (double-buffered: the loop is unrolled by 2)
        auto load_ddist = [&](int pb, float (&dst)[16]) {
            if (GATHER) {
                // gathered gradient: a lane reads [px][slot] for the rows of its pixel's class, 0 elsewhere; one
                // ballot skips the 16 load instructions of a block no lane of the wave has a class in
                uint32_t vo[16];
                bool any = false;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const u32x4 kv = *(const u32x4*)(keys + pb * 32 + 8 * g4 + 4 * h);
                    const u32x4 ko = *(const u32x4*)(koff + pb * 32 + 8 * g4 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool m = (kv[e] >> 16) == lab16;
                        any |= m;
                        vo[4 * g4 + e] = m ? voff_c + ko[e] : SPX_OOB;
                    }
                }
                if (__builtin_amdgcn_ballot_w64(any) != 0) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) dst[reg] = buf_load_f32(cdr, vo[reg], 0);
                } else {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) dst[reg] = 0.0f;
                }
                return;
            }
            const spx_rsrc ddr = make_rsrc_pred(a.d_dist + ((size_t)b * P + p0 + pb * 32) * a.HW);
            if (pb * 32 + 32 <= np) {     // wave-uniform: whole block real, no row predication
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int rb = (reg & 3) + 8 * (reg >> 2);
                    dst[reg] = buf_load_f32(ddr, voff_d, (uint32_t)rb * HW * 4u);
                }
            } else {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int rb = (reg & 3) + 8 * (reg >> 2);
                    dst[reg] = buf_load_f32(ddr, (pb * 32 + rb + 4 * h < np) ? voff_d : SPX_OOB, (uint32_t)rb * HW * 4u);
                }
            }
        };
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) ddA[reg] = ddB[reg] = 0.0f;
        if (have_dd) load_ddist(0, ddA);
        const bool tile_full = px0 + SPX_TILE_PX <= a.HW;      // wave-uniform: every pixel of the tile is real

        // one prototype block, read from accumulator slot SLOT (static); ddc = its dDist, ddnext = prefetch target;
        // gout = its G (fp32, accumulator layout)
        auto block = [&](int pb, auto slot_c, float (&ddc)[16], float (&ddnext)[16], f32x16& gout) {
            constexpr int SLOT = decltype(slot_c)::value;
            u32x4 anew[2];     // wide heads: the activation blob, int16 codes of a / ln 2 packed in pairs (spx_common.h)
            // G leaves the block as fp32 (it is packed to fp16 with ONE power-of-two scale per tile once the tile's largest |G|
            // is known, see below); the activation blob's block exponent travels in one word per (lane, block) behind the blob:
            // bits 8-15 = exponent of the activation scale + 128 (bits 0-7: 128)
            int ex_a = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) gout[j] = 0.0f;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 4; ++j) anew[s2][j] = 0u;
            if (pb < nv) {
                if (have_dd && pb + 1 < nv) load_ddist(pb + 1, ddnext);
                const bool full = pb * 32 + 32 <= np;
                // (dAct + dLogits.W) * c1, c1 = the constant factor of act'(d).  The chain starts from a literal zero C operand
                // (no 16 v_mov per block); with the head image in LDS it runs unconditionally - without a logits gradient both
                // the image and the dLogits fragments are zeros
                f32x16 ga;
#pragma unroll
                for (int i = 0; i < 16; ++i) ga[i] = 0.0f;
                if (head_lds != 0 || have_dl) {
#pragma unroll
                    for (int c = 0; c < ncstep; ++c) {
                        bf16x8 whi, wlo;
                        if (head_lds) {
                            const char* wf = hlds + (pb * ncstep + c) * 2048 + lane * 16;
                            whi = *(const bf16x8*)wf;
                            wlo = *(const bf16x8*)(wf + 1024);
                        } else {
                            const uint32_t so = (uint32_t)(((panel * NPB + pb) * ncstep + c) * 2048);
                            whi = __builtin_bit_cast(bf16x8, buf_load_b128(htr, (uint32_t)lane * 16u, so));
                            wlo = __builtin_bit_cast(bf16x8, buf_load_b128(htr, (uint32_t)lane * 16u, so + 1024u));
                        }
                        ga = mfma_bf16(whi, dlhi[c], ga);
                        ga = mfma_bf16(wlo, dlhi[c], ga);
                        ga = mfma_bf16(whi, dllo[c], ga);
                    }
                }
                if (DACT) {   // gradient arriving on the activations ([pixel][P] rows)
                    // loaded with a prototype column per lane (two 128-B row pieces per instruction, all 16 loads in
                    // flight), turned through the wave's LDS scratch into the accumulator layout (pixel per lane)
                    const spx_rsrc dar = make_rsrc_pred(a.d_act + (size_t)b * a.HW * P + p0 + pb * 32);
                    float* const sc = (float*)(dact_sc + wave * SPX_DACT_SC);     // 32 x 33 floats + pad; the stages are idle in phase 1
                    const int col = lane & 31;
                    const bool col_ok = pb * 32 + col < np;
                    float dv[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int pxs = px0 + 32 * wave + 2 * q + (lane >> 5);
                        dv[q] = buf_load_f32(dar, (col_ok && pxs < a.HW) ? ((uint32_t)pxs * (uint32_t)P + (uint32_t)col) * 4u : SPX_OOB, 0);
                    }
#pragma unroll
                    for (int q = 0; q < 16; ++q) sc[(2 * q + (lane >> 5)) * 33 + col] = dv[q];
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        ga[reg] = __builtin_fmaf(act_c1, sc[r * 33 + (reg & 3) + 8 * (reg >> 2) + 4 * h], ga[reg]);
                }
                // straight-line element math on register PAIRS (packed fp32 / packed converts), no per-element control
                // flow: d, 1/((d+1)(d+eps)), a / ln 2, then G.  Pair i = registers 2i, 2i+1 = two consecutive prototype rows.
                f32x2 dr[8], av[8], gv[8];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 p2v = *(const f32x4*)(p2s + pb * 32 + 8 * g4 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dr[2 * g4 + (e >> 1)][e & 1] = __builtin_fmaf(-2.0f, acc[SLOT][4 * g4 + e], p2v[e]) + x2;
                }
                auto pair_of = [](const auto& v, int i) {
                    f32x2 p;
                    p[0] = v[2 * i];
                    p[1] = v[2 * i + 1];
                    return p;
                };
                if (act_is_log) {
                    // stage by stage over four pairs at a time: the packed ops of one pair depend on each other back to back,
                    // four independent pairs fill the issue slots between them
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        f32x2 t1[4], m[4], rpv[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            f32x2 d;
                            d[0] = relu_f32(dr[4 * g + i][0]);
                            d[1] = relu_f32(dr[4 * g + i][1]);
                            t1[i] = d + 1.0f;
                            m[i] = t1[i] * (d + a.eps);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            f32x2 rp;
                            rp[0] = __builtin_amdgcn_rcpf(m[i][0]);                  // 1 / ((d+1)(d+eps))
                            rp[1] = __builtin_amdgcn_rcpf(m[i][1]);
                            rpv[i] = rp;
                            t1[i] = t1[i] * t1[i];
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const f32x2 q = t1[i] * rpv[i];
                            av[4 * g + i][0] = __builtin_amdgcn_logf(q[0]);          // log2((d+1)/(d+eps)): the blob is a / ln 2
                            av[4 * g + i][1] = __builtin_amdgcn_logf(q[1]);
                            gv[4 * g + i] = __builtin_elementwise_fma(pair_of(ga, 4 * g + i), rpv[i], pair_of(ddc, 4 * g + i));
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
#pragma unroll
                        for (int e = 0; e < 2; ++e)
                            av[i][e] = -1.44269504089f * relu_f32(dr[i][e]);
                        gv[i] = pair_of(ga, i) + pair_of(ddc, i);        // act' = -1 (folded into ga)
                    }
                }
                if (full && tile_full) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        gv[i][0] = dr[i][0] > 0.0f ? gv[i][0] : 0.0f;
                        gv[i][1] = dr[i][1] > 0.0f ? gv[i][1] : 0.0f;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const bool valid = px_ok && (pb * 32 + acc_row(2 * i + e, h) < np);
                            gv[i][e] = (valid && dr[i][e] > 0.0f) ? gv[i][e] : 0.0f;
                            av[i][e] = valid ? av[i][e] : 0.0f;
                        }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) gmax = fmaxf(gmax, __builtin_fabsf(gv[i >> 1][i & 1]));
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    gout[2 * i] = gv[i][0];
                    gout[2 * i + 1] = gv[i][1];
                }
                if (DW) {
                    // d_W stage, operand: this block's a / ln 2 as an exact-to-2^-17 bf16 hi + lo pair into the activation image
                    // (the dL image's cell layout; cell = 16 prototypes: registers 8 cell .. 8 cell + 7 of the lane, i.e. position
                    // 8 h + j of the cell's 32-B row <-> prototype 8 (j >> 2) + 4 h + (j & 3) of the cell)
                    if (want_dw) {
                        char* const aw = a_img + (NABUF == 2 ? SLOT * SPX_DW_IMG : 0) + wave * 4096 + rperm * 32 + h * 16;
#pragma unroll
                        for (int cell = 0; cell < 2; ++cell) {
                            u32x4 hw, lw;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                uint32_t hi, lo;
                                split_bf16x2(av[4 * cell + i], hi, lo);
                                hw[i] = hi;
                                lw[i] = lo;
                            }
                            *(u32x4*)(aw + cell * 2048) = hw;
                            *(u32x4*)(aw + cell * 2048 + 1024) = lw;
                        }
                    }
                } else {
                    // activation blob: amax = m * 2^ea, m in [0.5, 1): codes round(a * 2^-ea * 32767), |code| <= 32767
                    float amax = 0.0f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) amax = fmaxf(amax, __builtin_fabsf(av[i >> 1][i & 1]));
                    int ea = __builtin_amdgcn_frexp_expf(amax);
                    ea = ea < -100 ? -100 : (ea > 100 ? 100 : ea);
                    const float ascale_dn = __builtin_amdgcn_ldexpf(1.0f, -ea);
                    ex_a = ea;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const f32x2 an = av[i] * ascale_dn;                       // in [-1, 1]
                        anew[i >> 2][i & 3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pknorm_i16(an[0], an[1]));
                    }
                }
            }
            if (!DW) {
                // activation fragment dump for kernel 2 (wholly padded blocks are written as zeros: kernel 2 reads them)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const uint32_t so = (uint32_t)(((wave * NPB + pb) * 2 + s2) * 1024);
                    const uint32_t vo = spx_blob_slot(r, h, s2) * 16u;
                    if (want_ab) buf_store_b128(anew[s2], ar, vo, so);
                }
                const float exw = __uint_as_float(128u | ((uint32_t)(ex_a + 128) << 8));
                if (want_ab) buf_store_f32(exw, asr, (uint32_t)lane * 4u, (uint32_t)((wave * NPB + pb) * 256));
            }
        };
        // ---- d_W stage, product: after the barrier every wave takes one 16 x 16 tile of the block's d_W^T [32 prototypes x 32
        // classes] (wave w: prototype cell w & 1, class cell w >> 1) over ALL 128 pixels of the tile: 4 k-steps of 32 pixels
        // (= the four waves' images) x 3 split-bf16 MFMAs (hi.hi, lo.hi, hi.lo), v_mfma_f32_16x16x32_bf16.  No cross-wave sum,
        // no atomics: the tile's partial leaves as ONE 16-B store per lane, [class][32 prototypes] fp32 rows per block.
        // BUF: which activation image (two images: block pb + 1 is written while slower waves still read block pb; with the
        // dAct scratch in the way there is one image and a second barrier behind the reads).
        auto dw_stage = [&](int pb, auto buf_c) {
            constexpr int BUF = decltype(buf_c)::value;
            __syncthreads();
            const int lane = spx_opaque((int)threadIdx.x) & 63;     // (see spx_opaque)
            const int g = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
            const int pg = wave & 1, cg = wave >> 1;
            const bool live = 16 * cg < K;                          // wave-uniform: the class cell holds a real class
            // (two accumulators: the hi.hi chain and the two correction terms run side by side, summed once)
            f32x4 dw, dw2;
#pragma unroll
            for (int i = 0; i < 4; ++i) dw[i] = dw2[i] = 0.0f;
            if (live) {
                // transposed-read rows of this lane: pixels 8 g + qq and 8 g + qq + 4 of the k-step, under the row permutation
                const int row0 = 8 * g + qq + 4 * (g & 1), row1 = 8 * g + qq + 4 * (1 - (g & 1));
                const char* const ab = a_img + (NABUF == 2 ? BUF * SPX_DW_IMG : 0) + pg * 2048 + pp * 8;
                const char* const lb = l_img + cg * 2048 + pp * 8;
                auto frag = [&](const char* p0) -> bf16x8 {
                    const bf16x4 t0 = __builtin_bit_cast(bf16x4, lds_tr_read(p0 + row0 * 32));
                    const bf16x4 t1 = __builtin_bit_cast(bf16x4, lds_tr_read(p0 + row1 * 32));
                    return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
                };
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 ah = frag(ab + ks * 4096), al = frag(ab + ks * 4096 + 1024);
                    const bf16x8 lh = frag(lb + ks * 4096), ll = frag(lb + ks * 4096 + 1024);
                    dw = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, lh, dw, 0, 0, 0);
                    dw2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, lh, dw2, 0, 0, 0);
                    dw2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, ll, dw2, 0, 0, 0);
                }
                dw += dw2;
            }
            if (NABUF == 1) __syncthreads();
            // accumulator row 4 g + i <-> position 4 g + i of the prototype cell <-> prototype 8 (g & 1) + 4 (g >> 1) + i
            // (the grouping tail builds its dUnits fragments in ACCUMULATOR row order: position li of a class cell is then
            // unit 8 ((li >> 2) & 1) + 4 (li >> 3) + (li & 3) of the cell, as for the prototype cells)
            const int cls = 16 * cg + (a.packed_tailT ? 8 * ((li >> 2) & 1) + 4 * (li >> 3) + (li & 3) : li);
            const spx_rsrc dwr = make_rsrc_pred((char*)a.a_out + ((((size_t)panel * ntiles + tile_g) * NPB + pb) * K) * 128);
            const uint32_t vo = cls < K ? (uint32_t)(cls * 128 + (16 * pg + 8 * (g & 1) + 4 * (g >> 1)) * 4) : SPX_OOB;
            buf_store_b128(__builtin_bit_cast(u32x4, dw), dwr, vo, 0);
        };
        // ROLLED loop, two blocks per iteration from the static slots 0 and 1 (static dDist buffers too), then one
        // rotation by two: the next pair moves to the front, the pair's G (fp32) enters the two vacated slots.  NPB / 2
        // iterations, so the rotation ends aligned: acc[i] = block i's G.
#pragma unroll 1
        for (int pb = 0; pb < NPB; pb += 2) {
            f32x16 gA, gB;
            block(pb, std::integral_constant<int, 0>{}, ddA, ddB, gA);
            if (DW && want_dw && pb < nv) dw_stage(pb, std::integral_constant<int, 0>{});
            block(pb + 1, std::integral_constant<int, 1>{}, ddB, ddA, gB);
            if (DW && want_dw && pb + 1 < nv) dw_stage(pb + 1, std::integral_constant<int, 1>{});
#pragma unroll
            for (int i = 0; i + 2 < NPB; ++i) acc[i] = acc[i + 2];
            acc[NPB - 2] = gA;
            acc[NPB - 1] = gB;
        }
        // ---- ONE power-of-two scale per tile puts G into fp16 (11 significant bits; a gradient has no fixed range): the largest
        // |G| of the tile sits just under 2^15.  G enters dX = 2 (rs x - P^T G) and crosses to kernel 2 as that single fp16
        // plane (both products are fp16 MFMAs; the bank and X are bf16-representable, i.e. exact in fp16); the exponent goes
        // to the side array behind the blobs, one word per (panel, tile).
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, m));
        if (lane == 0) gmaxs[wave] = gmax;
        __syncthreads();
        int ex_t = __builtin_amdgcn_frexp_expf(fmaxf(fmaxf(gmaxs[0], gmaxs[1]), fmaxf(gmaxs[2], gmaxs[3])));
        ex_t = ex_t < -100 ? -100 : (ex_t > 100 ? 100 : ex_t);
        const int e_t = __builtin_amdgcn_readfirstlane(15 - ex_t);       // |G| 2^e_t < 2^15
        const float gscale = __builtin_amdgcn_ldexpf(1.0f, e_t), sinv = __builtin_amdgcn_ldexpf(1.0f, -e_t);
        if (a.g_out && tid == 0) *(int32_t*)((char*)a.g_out + spx_gexp_offset(blob_total) + ((size_t)panel * ntiles + tile_g) * 4) = e_t;
        {
            const int lane = spx_opaque((int)threadIdx.x) & 63, r = lane & 31, h = lane >> 5;     // (see spx_opaque)
            f16x2 one2;
            one2[0] = (_Float16)1.0f;
            one2[1] = (_Float16)1.0f;
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) {
                u32x4 gw[2];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    f32x2 v;
                    v[0] = acc[pb][2 * k] * gscale;
                    v[1] = acc[pb][2 * k + 1] * gscale;
                    const uint32_t w = pack_f16x2(v);
                    gw[k >> 2][k & 3] = w;
                    // the row sum uses the SAME rounded G as the P^T.G product: dX = 2 sum_p G_p (x - p) then carries
                    // G's rounding relative to |x - p|, not to |p| (matters where a pixel sits on a prototype)
                    rs = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, w), one2, rs, false);
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[pb][4 * s2 + i] = __uint_as_float(gw[s2][i]);
                    if (a.g_out)
                        buf_store_b128(gw[s2], gr, spx_blob_slot(r, h, s2) * 16u, (uint32_t)(((wave * NPB + pb) * 2 + s2) * 1024));
                }
            }
        }
#ifdef SPX_DIAG_STAMPS
        dg_t2 = __builtin_amdgcn_s_memtime();
#endif
        auto g_frag = [&](int pb, int s2) -> f16x8 {       // the fp16 B fragment of k-step s2
            u32x4 w;
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = __float_as_uint(acc[pb][4 * s2 + i]);
            return __builtin_bit_cast(f16x8, w);
        };
        auto clear_acc = [&]() {
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[pb][i] = 0.0f;
        };
        if (!a.dx) {
            clear_acc();
            return;
        }
        // ---- phase 2: dX^T[ch x px] = 2 rs * x + (-2 P)^T . G, one 32-channel block per (rolled) iteration ----
        const float rs_tot = rs + __shfl_xor(rs, 32);
        if (h == 0) rss[32 * wave + r] = 2.0f * rs_tot;      // the finish below wants 2 rs (the P^T fragments carry -2 p)
        const bool first_of_scale = (panel == q_begin) || (pl.panel_ch0[panel - 1] != ch0);
        // A scale of more than 192 prototypes spans several panels, each adding its share of dX.  With fp32 features the sum
        // runs in dX itself; with bf16 features it runs in the caller's fp32 scratch a.dx_acc ([B][C][HW rounded up to 4]) and
        // only the scale's LAST panel writes dX, rounded once (through the bf16 buffer it was one rounding per panel).
        const bool last_of_scale = (panel + 1 == q_end) || (pl.panel_ch0[panel + 1] != ch0);
        const bool acc32 = ACC && !XF32 && !(first_of_scale && last_of_scale);     // workgroup-uniform

        constexpr int BT = spx_bwd_bt_bytes<NPB>();
        char* const bt = smem;                               // 2 x BT  (P^T fragments of one channel block)
        char* const tt0 = smem + 2 * BT;                     // fp32 transpose tile 0
        char* const tt1 = hlds;                              // fp32 transpose tile 1 (the head^T image is dead in phase 2)
        // (phase 2's own copies of the thread / lane index: addresses derived from them are recomputed here instead of being
        // hoisted to kernel entry and spilled around the panel loop, see spx_opaque)
        const int tid = spx_opaque((int)threadIdx.x), lane = tid & 63, r = lane & 31, h = lane >> 5;
        const int frow = tid >> 3, fseg = tid & 7;           // finish mapping: channel row, 16-px segment
        const int fpx = px0 + fseg * 16;
        // this thread's segment inside a 32-channel block (resources are re-based per block: offsets < 2 GiB)
        const uint32_t fvo = ((uint32_t)frow * HW + (uint32_t)fpx) * ESZ;
        constexpr int NV = XF32 ? 4 : 2;                     // 16-B vectors per 16 px
        constexpr int PV = 16 / NV;                          // pixels per vector
        constexpr int BTP = (NPB * 2 + 3) / 4;               // fragment passes of the stage loader

        // stage loader: fragment f = wave + 4 i  <->  (pb = f >> 1, s2 = f & 1) of channel block chb
        u32x4 bt_reg[BTP];
        auto bt_load = [&](int chb) {
#pragma unroll
            for (int i = 0; i < BTP; ++i) {
                const int f = wave + 4 * i;
                const uint32_t so = (uint32_t)((((panel * NPB) * 2 + f) * nchb + chb) * 1024);
                bt_reg[i] = buf_load_b128(btr, (f < 2 * nv) ? (uint32_t)lane * 16u : SPX_OOB, so);
            }
        };
        auto bt_write = [&](int buf) {
#pragma unroll
            for (int i = 0; i < BTP; ++i) {
                const int f = wave + 4 * i;
                if (f < 2 * NPB) *(u32x4*)(bt + buf * BT + f * 1024 + lane * 16) = bt_reg[i];
            }
        };
        // x (and the previous partial dX) of a finish segment, as raw 16-B vectors; loaded one block ahead.  A ragged image end
        // (H*W not a multiple of the vector) sends the image's LAST tile down the element-wise path (tile-uniform switch).
        const bool use_vec = VEC && !(RAG && !tile_full);
        u32x4 xw[VEC ? NV : 1], pw[VEC ? NV : 1];
        auto x_load = [&](int chb) {
            if (use_vec) {
                const bool ch_ok = chb * 32 + frow < Cs;
                const spx_rsrc xir = make_rsrc_pred(x_img + (size_t)(ch0 + chb * 32) * a.HW * ESZ);
                const spx_rsrc dxr = make_rsrc_pred((char*)a.dx + ((size_t)b * C + ch0 + chb * 32) * a.HW * ESZ);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    // a vector wholly inside the image as it lies; the one straddling the image end (H*W % PV != 0) from a
                    // window moved back to end at the image end, shifted into place below (see SpxXStager::make_ctx)
                    // a vector wholly inside the image as it lies (any alignment); the one straddling a ragged image end is
                    // fetched element by element below, in the image's last tile only
                    const uint32_t vo = (ch_ok && (fpx + (v + 1) * PV <= a.HW)) ? fvo + 16u * v : SPX_OOB;
                    xw[v] = buf_load_b128(xir, vo, 0);
                    // first panel of a scale: nothing to accumulate onto (dropped load returns 0)
                    pw[v] = buf_load_b128(dxr, (first_of_scale || acc32) ? SPX_OOB : vo, 0);
                }
            }
        };
        bt_load(0);
        x_load(0);
        if (DACT) __syncthreads();     // the dAct scratch of slower waves sits where the P^T stages go
        bt_write(0);
        __syncthreads();
        for (int chb = 0; chb < nchb; ++chb) {
            const bool ch_ok = chb * 32 + frow < Cs;
            const spx_rsrc xir = make_rsrc_pred(x_img + (size_t)(ch0 + chb * 32) * a.HW * ESZ);
            const spx_rsrc dxr = make_rsrc_pred((char*)a.dx + ((size_t)b * C + ch0 + chb * 32) * a.HW * ESZ);
            float xv[16], pv[16];
            if (use_vec) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (XF32) {
                            xv[4 * v + e] = __uint_as_float(xw[v][e]);
                            pv[4 * v + e] = __uint_as_float(pw[v][e]);
                        } else {
                            xv[8 * v + 2 * e] = __uint_as_float(xw[v][e] << 16);
                            xv[8 * v + 2 * e + 1] = __uint_as_float(xw[v][e] & 0xffff0000u);
                            pv[8 * v + 2 * e] = __uint_as_float(pw[v][e] << 16);
                            pv[8 * v + 2 * e + 1] = __uint_as_float(pw[v][e] & 0xffff0000u);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const uint32_t vo = (ch_ok && fpx + e < a.HW) ? fvo + (uint32_t)e * ESZ : SPX_OOB;
                    if (XF32) {
                        xv[e] = buf_load_f32(xir, vo, 0);
                        pv[e] = first_of_scale ? 0.0f : buf_load_f32(dxr, vo, 0);
                    } else {
                        xv[e] = __uint_as_float((uint32_t)buf_load_u16(xir, vo, 0) << 16);
                        pv[e] = (first_of_scale || acc32) ? 0.0f : __uint_as_float((uint32_t)buf_load_u16(dxr, vo, 0) << 16);
                    }
                }
            }
            // next block's operands: P^T fragments and x, in flight across this block's MFMAs, barrier and finish
            bt_load(chb + 1 < nchb ? chb + 1 : chb);
            x_load(chb + 1 < nchb ? chb + 1 : chb);
            f32x16 accx;
#pragma unroll
            for (int i = 0; i < 16; ++i) accx[i] = 0.0f;
            const char* cur = bt + (chb & 1) * BT + lane * 16;
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) {
                if (pb < nv) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const f16x8 pt = *(const f16x8*)(cur + (pb * 2 + s2) * 1024);
                        accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(pt, g_frag(pb, s2), accx, 0, 0, 0);
                    }
                }
            }
            if (acc32 && !first_of_scale) {        // the partial of the scale's earlier panels (fp32)
                // (everything of this path is formed inside its branch: the default instance sits at the register cliff)
                const uint32_t HWp = (HW + 3u) & ~3u;
                const spx_rsrc d32r = make_rsrc_pred((char*)a.dx_acc + ((size_t)b * C + ch0 + chb * 32) * HWp * 4);
                const uint32_t fvo32 = ((uint32_t)frow * HWp + (uint32_t)fpx) * 4u;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const u32x4 w = buf_load_b128(d32r, (ch_ok && (uint32_t)(fpx + 4 * v) < HWp) ? fvo32 + 16u * v : SPX_OOB, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) pv[4 * v + e] = __uint_as_float(w[e]);
                }
            }
            char* T = (chb & 1) ? tt1 : tt0;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                *(float*)(T + acc_row(reg, h) * SPX_T_ROW + (32 * wave + r) * 4) = accx[reg];
            bt_write((chb + 1) & 1);
            __syncthreads();
            float ov[16];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const f32x4 tv = *(const f32x4*)(T + frow * SPX_T_ROW + fseg * 64 + v * 16);
                const f32x4 rv = *(const f32x4*)(rss + fseg * 16 + v * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[4 * v + e] = __builtin_fmaf(sinv, __builtin_fmaf(rv[e], xv[4 * v + e], tv[e]), pv[4 * v + e]);   // (2 rs x - 2 P^T.G16) / scale
            }
            if (acc32 && !last_of_scale) {
                const uint32_t HWp = (HW + 3u) & ~3u;
                const spx_rsrc d32r = make_rsrc_pred((char*)a.dx_acc + ((size_t)b * C + ch0 + chb * 32) * HWp * 4);
                const uint32_t fvo32 = ((uint32_t)frow * HWp + (uint32_t)fpx) * 4u;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    u32x4 w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = __float_as_uint(ov[4 * v + e]);
                    buf_store_b128(w, d32r, (ch_ok && (uint32_t)(fpx + 4 * v) < HWp) ? fvo32 + 16u * v : SPX_OOB, 0);
                }
            } else if (use_vec) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const bool ok = ch_ok && (fpx + (v + 1) * PV <= a.HW);
                    u32x4 w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (XF32) {
                            w[e] = __float_as_uint(ov[4 * v + e]);
                        } else {
                            bf16x2 p;
                            p[0] = (__bf16)ov[8 * v + 2 * e];
                            p[1] = (__bf16)ov[8 * v + 2 * e + 1];
                            w[e] = __builtin_bit_cast(uint32_t, p);
                        }
                    }
                    buf_store_b128(w, dxr, ok ? fvo + 16u * v : SPX_OOB, 0);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const uint32_t vo = (ch_ok && fpx + e < a.HW) ? fvo + (uint32_t)e * ESZ : SPX_OOB;
                    if (XF32)
                        buf_store_f32(ov[e], dxr, vo, 0);
                    else
                        buf_store_u16(__builtin_bit_cast(uint16_t, (__bf16)ov[e]), dxr, vo, 0);
                }
            }
        }
        if (panel + 1 < q_end) clear_acc();            // the next panel (if any) accumulates from zero
        __syncthreads();   // T tiles / P^T stages are rewritten by the next panel's main loop
    };

    // Software pipeline inside a panel: SpxPipeline (spx_mainloop.h).  Nothing is kept in flight across the
    // epilogue (its register budget is the binding one).
#ifdef SPX_DIAG_STAMPS
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), t1 = 0, t3 = 0;
#endif
    for (int panel = q_begin; panel < q_end; ++panel) {
        const char* bank0 = a.packed_bank + (size_t)(panel * nchunks) * chunk_bytes;
        x2part = 0.0f;
        pipe.run_panel(acc, x2part, tc, smem, bank0, pl.panel_ch0[panel], Cs, lane, wave, tid,
                       [&]() { consts_issue(panel); }, consts_commit);
#ifdef SPX_DIAG_STAMPS
        t1 = __builtin_amdgcn_s_memtime();
#endif
        epilogue(panel);
        __syncthreads();     // the epilogue re-uses the staging LDS and the head / |p|^2 images
    }
#ifdef SPX_DIAG_STAMPS
    if (a.dbg && tid == 0) {
        t3 = __builtin_amdgcn_s_memtime();
        unsigned long long* d = a.dbg + (size_t)blockIdx.x * 4;
        d[0] = t0; d[1] = t1; d[2] = dg_t2; d[3] = t3;

    }
#endif
}

template <int NPB, int NCB, bool GATHER, bool DACT>
static hipError_t launch_bwd_gd(const SpxBwdArgs& a, int x_dtype, dim3 grid, hipStream_t s) {
    constexpr size_t lds = (size_t)spx_bwd_lds_bytes<NPB, NCB, DACT>();
    if (x_dtype == 1) {
        if (a.vec_ok == 2) hipLaunchKernelGGL((spx_bwd_kernel<NPB, NCB, true, 2, GATHER, DACT>), grid, dim3(256), lds, s, a);
        else if (a.vec_ok) hipLaunchKernelGGL((spx_bwd_kernel<NPB, NCB, true, 1, GATHER, DACT>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((spx_bwd_kernel<NPB, NCB, true, 0, GATHER, DACT>), grid, dim3(256), lds, s, a);
    } else {
        bool acc = false;
        if (a.dx_acc && a.dx) {
            for (int q = 1; q < a.plan.npanels; ++q) acc |= a.plan.panel_ch0[q] == a.plan.panel_ch0[q - 1];
        }
        if (acc) {
            if (a.vec_ok == 2) hipLaunchKernelGGL((spx_bwd_kernel<NPB, NCB, false, 2, GATHER, DACT, true>), grid, dim3(256), lds, s, a);
            else if (a.vec_ok) hipLaunchKernelGGL((spx_bwd_kernel<NPB, NCB, false, 1, GATHER, DACT, true>), grid, dim3(256), lds, s, a);
            else hipLaunchKernelGGL((spx_bwd_kernel<NPB, NCB, false, 0, GATHER, DACT, true>), grid, dim3(256), lds, s, a);
        }
        if (acc) {
        } else if (a.vec_ok == 2) hipLaunchKernelGGL((spx_bwd_kernel<NPB, NCB, false, 2, GATHER, DACT>), grid, dim3(256), lds, s, a);
        else if (a.vec_ok) hipLaunchKernelGGL((spx_bwd_kernel<NPB, NCB, false, 1, GATHER, DACT>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((spx_bwd_kernel<NPB, NCB, false, 0, GATHER, DACT>), grid, dim3(256), lds, s, a);
    }
    return hipGetLastError();
}
template <int NPB, int NCB>
static hipError_t launch_bwd_x(const SpxBwdArgs& a, int x_dtype, dim3 grid, hipStream_t s) {
    if (a.labels) return a.d_act ? launch_bwd_gd<NPB, NCB, true, true>(a, x_dtype, grid, s)
                                 : launch_bwd_gd<NPB, NCB, true, false>(a, x_dtype, grid, s);
    return a.d_act ? launch_bwd_gd<NPB, NCB, false, true>(a, x_dtype, grid, s)
                   : launch_bwd_gd<NPB, NCB, false, false>(a, x_dtype, grid, s);
}

template <int NPB>
static hipError_t spx_launch_bwd_tiles(const SpxBwdArgs& a, int x_dtype, hipStream_t s) {
    const spx_plan& pl = a.plan;
    dim3 grid((unsigned)(a.tiles_launch * a.B), (unsigned)(a.ngroups > 1 ? a.ngroups : 1));
    if (pl.ncb == 1) return launch_bwd_x<NPB, 1>(a, x_dtype, grid, s);
    if (pl.ncb == 2) return launch_bwd_x<NPB, 2>(a, x_dtype, grid, s);
    return launch_bwd_x<NPB, 5>(a, x_dtype, grid, s);
}
// one translation unit per panel height, so the variants compile in parallel (vector staging: see spx_launch_fwd_npb)
template <int NPB>
static hipError_t spx_launch_bwd_npb(const SpxBwdArgs& a0, int x_dtype, hipStream_t s) {
    SpxBwdArgs a = a0;
    a.vec_ok = a.HW < 8 ? 0 : (a.HW % 8 == 0 ? 1 : 2);      // the element-wise path only for images of fewer than 8 pixels
    a.tile_first = 0;
    a.tiles_launch = (a.HW + SPX_TILE_PX - 1) / SPX_TILE_PX;
    a.tile_mul = spx_tile_mul(a.tiles_launch, (long long)a.HW * (x_dtype == 1 ? 4 : 2));
    return spx_launch_bwd_tiles<NPB>(a, x_dtype, s);
}
