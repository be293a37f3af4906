// Fused, persistent backward of the prototype-distance path (replaces autograd through
// segmentation/model/model_multiscale.py:255-281, :324-330, :243-244) for banks of ONE panel (a single scale of
// at most 192 prototypes, at most 256 channels, a head of at most 32 rows): dX AND dPrototypes from one kernel, with
// neither the G blob nor a second / third pass over X in HBM.  (The two-kernel backward of spx_bwd_impl.h /
// spx_bank.hip stays for every other bank shape; the formulas are in its header.)
//
// One workgroup = 8 waves = ONE compute unit for the whole launch (grid = number of CUs): it walks the 128-pixel
// tiles t = blockIdx.x, blockIdx.x + gridDim.x, ... and keeps its partial dPrototypes^T.X sums - the 192 x 256
// accumulator tile, 196 KB - in registers from the first tile to the last (96 registers per lane in every wave), so
// the parameter gradient leaves the chip once per workgroup as an fp32 slab (summed in a fixed order by
// spx_bank_reduce_kernel, as before: no float atomics, run-to-run identical results).
//
// Wave (pg, ph): pixel group pg = wave & 3 (32 pixels = the MFMA column), prototype half ph = wave >> 2
// (blocks 3 ph .. 3 ph + 2 of the panel's 6).  Per tile:
//   A  main loop   D^T[96 protos x 32 px] += Bank . X, the forward's arithmetic (bit-identical relu mask), SpxPipeline
//                  in its 512-thread form;
//   B1 G = (dDist + (dLogits.W) act'(d)) [d > 0] in accumulator layout (lane = pixel), fp32, left in the accumulator
//      registers; the activations leave as the 16-bit blob of the two-kernel backward (d_LastLayer is still formed by
//      spx_bank_bwd from it); every wave publishes max |G|;
//   B2 ONE power-of-two scale per tile puts G into fp16 (11 significant bits, the precision the G blob had): written to
//      LDS in MFMA B-fragment order ("G16", 48 KB) - the only copy of G there is; rowsum(G) of the rounded values;
//   C  dX^T[ch x px] = sinv (2 rs x + (-2 P)^T . G16): fp16 MFMA, G16 fragments read back verbatim, (-2P)^T fragments
//      (fp16, exact: the bank is bf16-representable) streamed through LDS, two channel blocks per round (one per
//      prototype half's waves), result transposed through LDS so X is read and dX written in whole rows;
//   D  dBank^T partial: the tile's X rows (L2 hits: the main loop has just streamed them) are laid out [channel][px]
//      in LDS as fp16, G16 is read TRANSPOSED (ds_read_b64_tr_b16: the pixel becomes the MFMA k) and wave (pg, ph)
//      accumulates prototype blocks 3 ph .. 3 ph + 2 x channel blocks 2 pg, 2 pg + 1.  The persistent accumulators
//      live in the units of the current tile's scale: they are multiplied by the (power-of-two) ratio of consecutive
//      scales once per tile, so the MFMAs accumulate in place.
#pragma once
#include "spx_args.h"
#include "spx_mainloop.h"
#include <type_traits>

#define SPXF_THREADS 512
#define SPXF_T_ROW 528                      // fp32 transpose tile row: 128 px * 4 B + 16 B pad
#define SPXF_T_BYTES (32 * SPXF_T_ROW)
#define SPXF_XD_ROW 272                     // fp16 [channel][128 px] image of phase D: 256 B + 16 B pad (conflict-free ds_read_b128)

struct SpxBwdFArgs {
    spx_plan plan;
    const void* x;
    const char* packed_bank;
    const char* packed_bankT16;   // fp16 A-fragments of -2 bank^T, [chb][pb][s2][lane][8]
    const float* p2;
    const char* packed_headT;
    const float* d_dist;
    const float* d_logits;
    const int32_t* labels;        // class-gathered distance gradient (see SpxBwdArgs)
    const uint32_t* proto_key;
    const float* d_cls_dist;
    int J;
    void* dx;
    uint16_t* a_out;
    float* workspace;             // [gridDim.x] slabs of the parameter kernel's layout; NULL = bank frozen
    int B, HW;
    float eps;
    int act_fn;
    unsigned long long* dbg;      // diagnostic builds only (SPX_DIAG_STAMPS): per-workgroup phase clocks
};

typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

__device__ __forceinline__ f32x16 mfma_f16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// LDS carve (one workgroup per CU):
//   region S (shared by the phases)   A: 2 main-loop stages | head^T fragments behind them
//                                     C: (-2P)^T fragments of the round (2 channel blocks) | 2 transpose tiles
//                                     D: the fp16 [256][128 px] X image
//   region G                          G16: [pg][pb][s2] 1-KiB fragments
//   small                             |p|^2, class keys / plane offsets (gathered mode), rowsum partials, 2 rs, wave maxima
template <int NPB>
__host__ __device__ constexpr int spxf_region_s_bytes() {
    constexpr int a = 2 * spx_stage_bytes(NPB);
    constexpr int c = 2 * NPB * 2 * 1024 + 2 * SPXF_T_BYTES;
    constexpr int d = 256 * SPXF_XD_ROW;
    return (a > c ? (a > d ? a : d) : (c > d ? c : d));
}
template <int NPB>
__host__ __device__ constexpr int spxf_region_g_bytes() { return 4 * NPB * 2 * 1024; }
template <int NPB>
__host__ __device__ constexpr int spxf_lds_bytes() {
    return spxf_region_s_bytes<NPB>() + spxf_region_g_bytes<NPB>() + NPB * 2 * 2048 + 3 * NPB * 32 * 4 + 2 * SPX_TILE_PX * 4 + SPX_TILE_PX * 4 + 64;
}
static_assert(spxf_lds_bytes<6>() <= SPX_LDS_LIMIT, "fused backward LDS");

template <int NPB, bool XF32, int VM, bool GATHER>
__global__ __launch_bounds__(SPXF_THREADS, 2) void spx_bwdf_kernel(const SpxBwdFArgs a) {
    constexpr bool VEC = VM != 0, RAG = VM == 2;
    constexpr int NH = NPB / 2;                       // prototype blocks per wave
    constexpr int NCB = 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const spx_plan& pl = a.plan;
    const int tid0 = threadIdx.x, lane0 = tid0 & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int pg = wave & 3, ph = wave >> 2;
    const int pbw = ph * NH;                          // this wave's first prototype block
    const int tiles_per_img = (a.HW + SPX_TILE_PX - 1) / SPX_TILE_PX;
    const long long ntiles = (long long)a.B * tiles_per_img;
    const int Cs = pl.channels_per_scale;
    const int P = pl.num_prototypes, K = pl.num_classes;
    const int np = pl.panel_np[0];
    const int nv = (np + 31) >> 5;                    // prototype blocks holding >= 1 real prototype
    const int nchb = (Cs + 31) / 32;
    const uint32_t HW = (uint32_t)a.HW;
    constexpr int ESZ = XF32 ? 4 : 2;
#ifndef SPXF_XR
#define SPXF_XR 4
#endif
    constexpr int XR = SPXF_XR;
#ifndef SPXF_BATCH
#define SPXF_BATCH 0      // (every fragment read of a main-loop chunk issued up front: measured, no gain here, 28 registers)
#endif
    using Pipe = SpxPipeline<NPB, XF32, VM, XR, SPXF_THREADS, NH, SPXF_BATCH != 0>;
    using XSt = SpxXStager<XF32, VM, SPXF_THREADS>;

    constexpr int RS = spxf_region_s_bytes<NPB>();
    constexpr int head_lds = NPB * 2 * 2048;          // head^T fragments (hi, lo) of the panel: 2 class k-steps
    char* const RG = smem + RS;
    char* const hlds = RG + spxf_region_g_bytes<NPB>();       // head^T fragments: staged once per launch
    float* const p2s = (float*)(hlds + head_lds);
    uint32_t* const keys = (uint32_t*)(p2s + NPB * 32);
    uint32_t* const koff = keys + NPB * 32;
    float* const rsp = (float*)(koff + NPB * 32);     // [2][128] rowsum(G16) of each prototype half
    float* const rss = rsp + 2 * SPX_TILE_PX;         // [128] 2 * rowsum
    float* const gmaxs = rss + SPX_TILE_PX;           // [8] wave maxima of |G|
    // [pg][k-step][hi | lo] dLogits B fragments of the tile (1 KiB each): read in phase B1 only, so they share region G with
    // G16 (written in phase B2); the [32][K] scratch they are built from sits behind them
    char* const dls = RG;

    const spx_rsrc htp = make_rsrc_pred(a.packed_headT);
    const spx_rsrc p2p = make_rsrc_pred(a.p2);
    const spx_rsrc keyr = make_rsrc_pred(GATHER ? a.proto_key : nullptr);
    const bool have_dd = GATHER ? a.d_cls_dist != nullptr : a.d_dist != nullptr;
    const bool have_dl = a.d_logits != nullptr;
    const bool act_is_log = a.act_fn == 0;
    const float act_c1 = act_is_log ? -(1.0f - a.eps) : -1.0f;
    const bool want_bank = a.workspace != nullptr;

    // ---- persistent state: the d_bank partial (prototype blocks pbw + i x channel blocks 2 pg + t) in the units of
    // the current scale 2^e_cur, and colsum(G16) of the blocks this wave owns the sum of ----
    f32x16 accp[NH][2];
    float csum[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        csum[i] = 0.0f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) accp[i][t][e] = 0.0f;
    }
    int e_cur = 0;
    bool first = true;

    // ---- once per launch: head^T fragments, |p|^2 and the class keys ----
    {
        constexpr int HPASS = head_lds / (SPXF_THREADS * 16);
#pragma unroll
        for (int i = 0; i < HPASS; ++i)
            *(u32x4*)(hlds + i * SPXF_THREADS * 16 + tid0 * 16) = buf_load_b128(htp, have_dl ? (uint32_t)(i * SPXF_THREADS * 16 + tid0 * 16) : SPX_OOB, 0);
    }
    if (tid0 < NPB * 32) {
        p2s[tid0] = buf_load_f32(p2p, (uint32_t)tid0 * 4u, 0);
        if (GATHER) {
            const uint32_t kr = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(keyr, (uint32_t)tid0 * 4u, 0, 0);
            keys[tid0] = kr;
            koff[tid0] = (kr & 0xFFFFu) * HW * 4u;
        }
    }
    // tile geometry; an index past the last tile gives a context whose every access is dropped (the run-ahead requests of
    // the last tile)
    auto tile_ctx = [&](long long tile, int& b, int& px0) {
        const bool ok = tile < ntiles;
        b = ok ? (int)(tile / tiles_per_img) : 0;
        px0 = ok ? (int)(tile - (long long)b * tiles_per_img) * SPX_TILE_PX : (tiles_per_img + 1) * SPX_TILE_PX;
    };
    __syncthreads();
    __syncthreads();

#ifdef SPX_DIAG_STAMPS
    unsigned long long dgt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define SPXF_STAMP(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); dgt[k] += now_ - dg_last; dg_last = now_; __builtin_amdgcn_sched_barrier(0); }
#else
#define SPXF_STAMP(k)
#endif
    // ---- phase D of a tile, split so that it can run at the START of the next tile's iteration, between that tile's first
    // requests and its main loop (the MFMAs below then cover the fetch latency of the next tile, with no loop-carried
    // prefetch registers): d_stage = the previous tile's X rows (L2 hits) -> fp16 [channel][px] image, d_compute = the products
    // (a macro, not a lambda: an array of stagers passed by reference stays in memory - every load is then waited for and
    // stored to scratch on its own)
#define SPXF_D_STAGE_CHUNK(ST, TCP, C)                                                                                   \
    {                                                                                                                     \
        (ST).fix_ragged(TCP);                                                                                             \
        u32x4 v_;                                                                                                         \
        if (XF32) {                                                                                                       \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                               \
                f32x2 p_; /* the bf16 rounding of the forward first, then fp16 (exact) */                                 \
                p_[0] = (float)(__bf16)__uint_as_float((ST).xr[0][e >> 1][(2 * e) & 3]);                                 \
                p_[1] = (float)(__bf16)__uint_as_float((ST).xr[0][e >> 1][(2 * e + 1) & 3]);                             \
                v_[e] = pack_f16x2(p_);                                                                                   \
            }                                                                                                             \
        } else {                                                                                                          \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) v_[e] = pack_f16x2(unpack_bf16x2((ST).xr[0][0][e]));           \
        }                                                                                                                 \
        *(u32x4*)(smem + ((C) * 32 + (spx_opaque(tid0) >> 4)) * SPXF_XD_ROW + (spx_opaque(tid0) & 15) * 16) = v_;        \
    }
    auto d_compute = [&]() {
        const int lane = spx_opaque(lane0), r = lane & 31, h = lane >> 5;
        const int tg = lane >> 4, tli = lane & 15, tqq = tli >> 2, tpp = tli & 3;
        const int ts2 = tg & 1, tkh = tg >> 1;
        const char* const XD = smem;
        f16x2_t one2;
        one2[0] = (_Float16)1.0f;
        one2[1] = (_Float16)1.0f;
#pragma unroll 2
        for (int ks = 0; ks < SPX_TILE_PX / 16; ++ks) {
            const int kof = (ks * 16 + 8 * h) * 2;
            const int pxa = ks * 16 + 8 * tkh + tqq;
            const int wsel = pxa >> 5, ra = pxa & 31;
            const int fo0 = spx_blob_slot(ra, tpp & 1, ts2) * 16 + 8 * (tpp >> 1);
            const int fo1 = spx_blob_slot(ra + 4, tpp & 1, ts2) * 16 + 8 * (tpp >> 1);
            f16x8 xb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) xb[t] = *(const f16x8*)(XD + ((2 * pg + t) * 32 + r) * SPXF_XD_ROW + kof);
#pragma unroll
            for (int i = 0; i < NH; ++i) {
                const int fb = ((wsel * NPB + pbw + i) * 2 + ts2) * 1024;
                const s16x4 g0 = lds_tr_read(RG + fb + fo0);
                const s16x4 g1 = lds_tr_read(RG + fb + fo1);
                const f16x8 gf = __builtin_bit_cast(f16x8, __builtin_shufflevector(g0, g1, 0, 1, 2, 3, 4, 5, 6, 7));
                if (pg == i) {          // colsum(G16) of block pbw + i: lanes r, r + 32 cover the k-step of prototype row r
                    float s8 = csum[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        f16x2_t pr;               // (element-wise, as spx_bank.hip: a dword extracted from the transposed read's
                        pr[0] = gf[2 * e];        // result through a vector bit-cast came out as dword 0 / 2 twice - hipcc 7.2)
                        pr[1] = gf[2 * e + 1];
                        s8 = __builtin_amdgcn_fdot2(pr, one2, s8, false);
                    }
                    csum[i] = s8;
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) accp[i][t] = mfma_f16(gf, xb[t], accp[i][t]);
            }
        }
    };
#ifndef SPXF_STAGGER
#define SPXF_STAGGER 8
#endif
    // De-phase the workgroups: all of them run the same phase sequence at the same pace, and started together they would ask
    // HBM for the same kind of data at the same time (a burst of several MB per phase for the chip, then silence).  A start
    // offset of up to one tile time spreads the phases - and the memory requests - evenly.
    if (SPXF_STAGGER > 0 && ntiles > (long long)gridDim.x) {
        const int steps = (int)((blockIdx.x >> 3) % SPXF_STAGGER);
        for (int i = 0; i < steps * (64 / SPXF_STAGGER); ++i) __builtin_amdgcn_s_sleep(127);
    }
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#ifdef SPX_DIAG_STAMPS
        unsigned long long dg_last = __builtin_amdgcn_s_memtime();
#endif
        // this tile's own copies of the thread / lane index (see spx_opaque); every phase below takes its own again
        const int tid = spx_opaque(tid0), lane = tid & 63, r = lane & 31, h = lane >> 5;
        int b, px0;
        tile_ctx(tile, b, px0);
        const char* x_img = (const char*)a.x + (size_t)b * Cs * a.HW * ESZ;
        const SpxTileCtx tc = XSt::make_ctx(x_img, a.HW, px0, tid);
        const int px = px0 + 32 * pg + r;
        const bool px_ok = px < a.HW;
        const bool tile_full = px0 + SPX_TILE_PX <= a.HW;
        const uint32_t voff_d = px_ok ? ((uint32_t)(4 * h) * HW + (uint32_t)px) * 4u : SPX_OOB;
        uint32_t lab16 = 0xFFFEu, voff_c = SPX_OOB;
        const spx_rsrc cdr = make_rsrc_pred((GATHER && a.d_cls_dist) ? a.d_cls_dist + (size_t)b * a.J * a.HW : nullptr);
        if (GATHER && a.d_cls_dist) {
            const spx_rsrc labr = make_rsrc_pred(a.labels + (size_t)b * a.HW);
            const uint32_t l = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(labr, px_ok ? (uint32_t)px * 4u : SPX_OOB, 0, 0);
            lab16 = (px_ok && l < 0xFFFEu) ? l : 0xFFFEu;
            voff_c = px_ok ? (uint32_t)px * 4u : SPX_OOB;
        }

        // ---- one burst of requests, consumed in this order: the tile's dLogits ([128][K] floats, contiguous: every thread takes
        // a few dwords), its first X / bank chunks, the first block's dDist ----
        constexpr int DLP = (SPX_TILE_PX * 32 + SPXF_THREADS - 1) / SPXF_THREADS;       // K <= 32
        float dlv[DLP];
        {
            const int npx_t = a.HW - px0 < SPX_TILE_PX ? a.HW - px0 : SPX_TILE_PX;
            const spx_rsrc rs_ = make_rsrc_pred(have_dl ? a.d_logits + ((size_t)b * a.HW + px0) * K : nullptr);
            const int nvalid = npx_t * K;
#pragma unroll
            for (int it = 0; it < DLP; ++it) {
                const int i = tid + SPXF_THREADS * it;
                dlv[it] = buf_load_f32(rs_, (have_dl && i < nvalid) ? (uint32_t)i * 4u : SPX_OOB, 0);      // (no d_logits: a null base)
            }
        }
        // the PREVIOUS tile's X rows for its phase D (L2 hits), queued ahead of this tile's own requests
        const bool do_d = want_bank && tile != (long long)blockIdx.x;                   // workgroup-uniform
        int b_p, px0_p;
        tile_ctx(do_d ? tile - gridDim.x : ntiles, b_p, px0_p);
        const SpxTileCtx tcp = XSt::make_ctx((const char*)a.x + (size_t)b_p * Cs * a.HW * ESZ, a.HW, px0_p, tid);
        XSt xim[8];                    // (unconditional: with no previous tile the context drops every access)
#pragma unroll
        for (int c = 0; c < 8; ++c) xim[c].load(tcp, c * 32, Cs - c * 32);
        Pipe pipe;
        pipe.issue_prologue(tc, a.packed_bank, 0, Cs, tid, [] {});
        // dDist of a prototype block (accumulator layout: a row of 32 pixels per register); the first block's is requested
        // before the main loop, so phase B does not open with a memory round trip
        float ddA[16], ddB[16];
        auto load_ddist = [&](int pb, float (&dst)[16], int h) {
            if (GATHER) {
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
            const spx_rsrc ddr = make_rsrc_pred(a.d_dist + ((size_t)b * P + pb * 32) * a.HW);
            if (pb * 32 + 32 <= np) {
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
#ifndef SPXF_DDPRE
#define SPXF_DDPRE 1
#endif
        // ---- phase D of the previous tile: its products run while this tile's first chunks are on their way ----
#pragma unroll
        for (int c = 0; c < 8; ++c) SPXF_D_STAGE_CHUNK(xim[c], tcp, c)
        __syncthreads();
        if (do_d) d_compute();
        __syncthreads();          // region S and G16 go to this tile
        SPXF_STAMP(5)
#if SPXF_DDPRE
        // the first block's dDist: requested here, it has the whole main loop to arrive (phase B opens without a memory round trip)
        if (have_dd && pbw < nv) load_ddist(pbw, ddA, h);
#endif
        // the raw dLogits -> LDS (behind the fragment image in region G, idle between tiles) ...
        float* const dlraw = (float*)(RG + 16384);
#pragma unroll
        for (int it = 0; it < DLP; ++it) {
            const int i = tid + SPXF_THREADS * it;
            if (i < SPX_TILE_PX * K) dlraw[i] = dlv[it];
        }
        // ... and, after the main loop's first barrier, the split-bf16 B fragments of each pixel group (element j of k-step c
        // <-> class 16 c + 8 h + j), pre-scaled by the constant factor of act'(d); built by the prototype-half 0 waves
        auto dl_frags = [&]() {
            if (ph == 0) {
                const int lane = spx_opaque(lane0), r = lane & 31, h = lane >> 5;
                const float* const bsc = dlraw + pg * 32 * K;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    u32x4 hw_, lw_;
#pragma unroll
                    for (int j2 = 0; j2 < 4; ++j2) {
                        f32x2 v2;
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int cls = c * 16 + 8 * h + 2 * j2 + e;
                            v2[e] = (have_dl && cls < K) ? bsc[r * K + cls] : 0.0f;
                        }
                        uint32_t hi, lo;
                        split_bf16x2(v2 * act_c1, hi, lo);
                        hw_[j2] = hi;
                        lw_[j2] = lo;
                    }
                    *(u32x4*)(dls + ((pg * 2 + c) * 2) * 1024 + lane * 16) = hw_;
                    *(u32x4*)(dls + ((pg * 2 + c) * 2 + 1) * 1024 + lane * 16) = lw_;
                }
            }
        };
        // =============================== phase A: the x.p tile ===============================
        f32x16 acc[NH];
#pragma unroll
        for (int i = 0; i < NH; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
        float x2part = 0.0f;
        pipe.run_body(acc, x2part, tc, smem, a.packed_bank, 0, Cs, lane, wave, tid, dl_frags);
        const float x2 = x2part + __shfl_xor(x2part, 32);
        SPXF_STAMP(0)
#ifdef SPX_DIAG_STAMPS
        dgt[6] += pipe.dg_issue; dgt[7] += pipe.dg_compute; dgt[8] += pipe.dg_write; dgt[9] += pipe.dg_barrier;
#endif
#if !SPXF_DDPRE
        if (have_dd && pbw < nv) load_ddist(pbw, ddA, h);
#endif

        // =============================== phase B1: G (fp32, in place) and the activation blob ===============================
        const size_t blob0 = ((size_t)tile * 4) * NPB * 2 * 1024;                       // bytes; [tile][wave pg][pb][s2]
        const size_t blob_total = (size_t)ntiles * 4 * NPB * 2 * 1024;
        const spx_rsrc ar = make_rsrc(a.a_out ? (const char*)a.a_out + blob0 : nullptr);
        const spx_rsrc asr = make_rsrc(a.a_out ? (const char*)a.a_out + blob_total + blob0 / 8 : nullptr);
        if (a.a_out && tile == 0 && tid == 0)
            *(uint32_t*)((char*)a.a_out + spx_ablob_fmt_offset(blob_total)) = SPX_ABLOB_I16;
        float gmax = 0.0f;

        auto block = [&](int i, auto slot_c, float (&ddc)[16], float (&ddnext)[16]) {
            constexpr int SLOT = decltype(slot_c)::value;
            const int lane = spx_opaque(lane0), r = lane & 31, h = lane >> 5;
            const int pb = pbw + i;
            u32x4 anew[2];
            int ex_a = 0;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 4; ++j) anew[s2][j] = 0u;
            if (pb < nv) {
                if (have_dd && i + 1 < NH && pb + 1 < nv) load_ddist(pb + 1, ddnext, h);
                const bool full = pb * 32 + 32 <= np;
                f32x16 ga;
#pragma unroll
                for (int e = 0; e < 16; ++e) ga[e] = 0.0f;
                if (have_dl) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const char* wf = hlds + (pb * 2 + c) * 2048 + lane * 16;
                        const bf16x8 whi = *(const bf16x8*)wf;
                        const bf16x8 wlo = *(const bf16x8*)(wf + 1024);
                        const bf16x8 dlh = *(const bf16x8*)(dls + ((pg * 2 + c) * 2) * 1024 + lane * 16);
                        const bf16x8 dll = *(const bf16x8*)(dls + ((pg * 2 + c) * 2 + 1) * 1024 + lane * 16);
                        ga = mfma_bf16(whi, dlh, ga);
                        ga = mfma_bf16(wlo, dlh, ga);
                        ga = mfma_bf16(whi, dll, ga);
                    }
                }
                // pass 1: d_raw in place of the x.p tile; its extreme over the valid entries gives the block exponent of the
                // activation blob up front (a decreases with d for "log", |a| grows with d for "linear"), so pass 2 can pack
                // every value as soon as it is formed.  Straight-line selects only (a per-element `||` compiles to branches).
                const bool whole = full && tile_full;             // wave-uniform: no entry of the block is padding
                float dmin = 3.0e38f, dmax = 0.0f;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 p2v = *(const f32x4*)(p2s + pb * 32 + 8 * g4 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d_raw = __builtin_fmaf(-2.0f, acc[SLOT][4 * g4 + e], p2v[e]) + x2;
                        acc[SLOT][4 * g4 + e] = d_raw;
                        const float dv = relu_f32(d_raw);
                        dmin = fminf(dmin, dv);
                        dmax = fmaxf(dmax, dv);
                    }
                }
                // (padding rows / pixels have x.p = 0 and |p|^2 = 0 or come from dropped loads: their d_raw = |x|^2 or |p|^2 >= 0 is
                // finite, so they can only widen [dmin, dmax], i.e. make the block exponent one step too large for a partial
                // block - the codes stay in range; their a is zeroed below)
                float amax;
                if (act_is_log) {
                    const float t1 = dmin + 1.0f;
                    amax = __builtin_amdgcn_logf((t1 * t1) * __builtin_amdgcn_rcpf(t1 * (dmin + a.eps)));
                } else {
                    amax = 1.44269504089f * dmax;
                }
                // activation blob: int16 codes scaled per (pixel, block) by a power of two (spx_common.h, format 2)
                int ea = __builtin_amdgcn_frexp_expf(amax);
                ea = ea < -100 ? -100 : (ea > 100 ? 100 : ea);
                const float ascale_dn = __builtin_amdgcn_ldexpf(1.0f, -ea);
                ex_a = ea;
                // pass 2, four register pairs (8 prototype rows) at a time, in place
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    f32x2 av[4], gv[4];
                    if (act_is_log) {
                        f32x2 t1[4], m[4], rpv[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            f32x2 d;
                            d[0] = relu_f32(acc[SLOT][8 * g + 2 * k]);
                            d[1] = relu_f32(acc[SLOT][8 * g + 2 * k + 1]);
                            t1[k] = d + 1.0f;
                            m[k] = t1[k] * (d + a.eps);
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            f32x2 rp;
                            rp[0] = __builtin_amdgcn_rcpf(m[k][0]);
                            rp[1] = __builtin_amdgcn_rcpf(m[k][1]);
                            rpv[k] = rp;
                            t1[k] = t1[k] * t1[k];
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const f32x2 q = t1[k] * rpv[k];
                            av[k][0] = __builtin_amdgcn_logf(q[0]);
                            av[k][1] = __builtin_amdgcn_logf(q[1]);
                            f32x2 gap, ddp;
                            gap[0] = ga[8 * g + 2 * k]; gap[1] = ga[8 * g + 2 * k + 1];
                            ddp[0] = ddc[8 * g + 2 * k]; ddp[1] = ddc[8 * g + 2 * k + 1];
                            gv[k] = __builtin_elementwise_fma(gap, rpv[k], ddp);
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                av[k][e] = -1.44269504089f * relu_f32(acc[SLOT][8 * g + 2 * k + e]);
                                gv[k][e] = ga[8 * g + 2 * k + e] + ddc[8 * g + 2 * k + e];
                            }
                        }
                    }
                    if (whole) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            gv[k][0] = acc[SLOT][8 * g + 2 * k] > 0.0f ? gv[k][0] : 0.0f;
                            gv[k][1] = acc[SLOT][8 * g + 2 * k + 1] > 0.0f ? gv[k][1] : 0.0f;
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const bool valid = px_ok & (pb * 32 + acc_row(8 * g + 2 * k + e, h) < np);
                                gv[k][e] = (valid & (acc[SLOT][8 * g + 2 * k + e] > 0.0f)) ? gv[k][e] : 0.0f;
                                av[k][e] = valid ? av[k][e] : 0.0f;
                            }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        gmax = fmaxf(gmax, fmaxf(__builtin_fabsf(gv[k][0]), __builtin_fabsf(gv[k][1])));
                        const f32x2 an = av[k] * ascale_dn;
                        anew[g][k] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pknorm_i16(an[0], an[1]));
                        acc[SLOT][8 * g + 2 * k] = gv[k][0];       // G stays in the accumulator registers, fp32, until the tile's scale is known
                        acc[SLOT][8 * g + 2 * k + 1] = gv[k][1];
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[SLOT][e] = 0.0f;
            }
            if (a.a_out) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const uint32_t so = (uint32_t)(((pg * NPB + pb) * 2 + s2) * 1024);
                    buf_store_b128(anew[s2], ar, spx_blob_slot(r, h, s2) * 16u, so);
                }
                const float exw = __uint_as_float(128u | ((uint32_t)(ex_a + 128) << 8));
                buf_store_f32(exw, asr, (uint32_t)lane * 4u, (uint32_t)((pg * NPB + pb) * 256));
            }
        };
        static_assert(NH == 3, "phase B is written for three prototype blocks per wave");
        block(0, std::integral_constant<int, 0>{}, ddA, ddB);
        __builtin_amdgcn_sched_barrier(0);
        block(1, std::integral_constant<int, 1>{}, ddB, ddA);
        __builtin_amdgcn_sched_barrier(0);
        block(2, std::integral_constant<int, 2>{}, ddA, ddB);
        // wave maximum of |G|
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, m));
        if (spx_opaque(lane0) == 0) gmaxs[wave] = gmax;
        SPXF_STAMP(1)
        __syncthreads();
        SPXF_STAMP(2)

        // =============================== phase B2: the tile's scale, G16 -> LDS, rowsum ===============================
        float tmax = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; ++w) tmax = fmaxf(tmax, gmaxs[w]);
        int ex = __builtin_amdgcn_frexp_expf(tmax);                 // tmax = m 2^ex, m in [0.5, 1)
        ex = ex < -100 ? -100 : (ex > 100 ? 100 : ex);
        int e_t = 15 - ex;                                          // |G| 2^e_t < 2^15
        // a tile whose gradients are more than 2^40 below the previous tile's keeps that tile's scale + 40 (its values then
        // lose precision relative to themselves only, and the accumulators never multiply by more than 2^40)
        if (!first && e_t > e_cur + 40) e_t = e_cur + 40;
        e_t = __builtin_amdgcn_readfirstlane(e_t);
        const float gscale = __builtin_amdgcn_ldexpf(1.0f, e_t);
        const float sinv = __builtin_amdgcn_ldexpf(1.0f, -e_t);
        const float ratio = first ? 1.0f : __builtin_amdgcn_ldexpf(1.0f, e_t - e_cur);
        e_cur = e_t;
        first = false;
        {
            const int lane = spx_opaque(lane0), r = lane & 31, h = lane >> 5;
            float rs = 0.0f;
            f16x2_t one2;
            one2[0] = (_Float16)1.0f;
            one2[1] = (_Float16)1.0f;
#pragma unroll
            for (int i = 0; i < NH; ++i) {
                u32x4 gw[2];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    f32x2 v;
                    v[0] = acc[i][2 * k] * gscale;
                    v[1] = acc[i][2 * k + 1] * gscale;
                    const uint32_t w = pack_f16x2(v);
                    gw[k >> 2][k & 3] = w;
                    // the row sum uses the SAME rounded G as the P^T.G product (see spx_bwd_impl.h)
                    rs = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, w), one2, rs, false);
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    *(u32x4*)(RG + ((pg * NPB + pbw + i) * 2 + s2) * 1024 + spx_blob_slot(r, h, s2) * 16) = gw[s2];
            }
            const float rs_tot = rs + __shfl_xor(rs, 32);
            if (h == 0) rsp[ph * SPX_TILE_PX + 32 * pg + r] = rs_tot;
        }
        if (want_bank) {
#pragma unroll
            for (int i = 0; i < NH; ++i) {
                csum[i] *= ratio;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) accp[i][t][e] *= ratio;
            }
        }
        __syncthreads();

        // =============================== phase C: dX ===============================
        SPXF_STAMP(3)
        if (a.dx) {
            const int tid = spx_opaque(tid0), lane = tid & 63, r = lane & 31, h = lane >> 5;
            if (tid < SPX_TILE_PX) rss[tid] = 2.0f * (rsp[tid] + rsp[SPX_TILE_PX + tid]);
            // G16 fragments of this wave's 32 pixels (all six prototype blocks), read back verbatim: the B operands of every round
#ifndef SPXF_GFR
#define SPXF_GFR 1
#endif
#if SPXF_GFR
            f16x8 gfr[NPB * 2];
#pragma unroll
            for (int f = 0; f < NPB * 2; ++f)
                gfr[f] = *(const f16x8*)(RG + ((pg * NPB) * 2 + f) * 1024 + spx_blob_slot(r, h, f & 1) * 16);
#else
            const char* const gsrc0 = RG + (pg * NPB) * 2 * 1024 + spx_blob_slot(r, h, 0) * 16;
            const char* const gsrc1 = RG + (pg * NPB) * 2 * 1024 + spx_blob_slot(r, h, 1) * 16;
#endif
            constexpr int PTB = NPB * 2 * 1024;                  // (-2P)^T fragments of one channel block
            char* const pt = smem;                               // [2][PTB]
            char* const tt = smem + 2 * PTB;                     // [2] transpose tiles
            const spx_rsrc btr = make_rsrc_pred(a.packed_bankT16);
            const int nrounds = (nchb + 1) / 2;
            constexpr int PTP = (2 * PTB) / (SPXF_THREADS * 16);   // 16-B pieces per thread and round
            u32x4 pt_reg[PTP];
            auto pt_load = [&](int j) {
#pragma unroll
                for (int i = 0; i < PTP; ++i) {
                    const uint32_t off = (uint32_t)((i * SPXF_THREADS + tid) * 16);
                    const int chb = 2 * j + (off >= (uint32_t)PTB ? 1 : 0);
                    pt_reg[i] = buf_load_b128(btr, chb < nchb ? off : SPX_OOB, (uint32_t)(j * 2 * PTB));
                }
            };
            auto pt_write = [&]() {
#pragma unroll
                for (int i = 0; i < PTP; ++i) *(u32x4*)(pt + (i * SPXF_THREADS + tid) * 16) = pt_reg[i];
            };
            // finish mapping: transpose tile tsel (= the prototype half whose waves produced it), channel row, 16-px segment
            const int tsel = tid >> 8, frow = (tid & 255) >> 3, fseg = tid & 7;
            const int fpx = px0 + fseg * 16;
            const uint32_t fvo = ((uint32_t)frow * HW + (uint32_t)fpx) * ESZ;
            constexpr int NV = XF32 ? 4 : 2;
            constexpr int PV = 16 / NV;
            const bool use_vec = VEC && !(RAG && !tile_full);
            u32x4 xw[VEC ? NV : 1];
            auto x_load = [&](int j) {
                if (use_vec) {
                    const int chb = 2 * j + tsel;
                    const bool ch_ok = chb * 32 + frow < Cs;
                    const spx_rsrc xir = make_rsrc_pred(x_img + (size_t)(chb * 32) * a.HW * ESZ);
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const uint32_t vo = (ch_ok && (fpx + (v + 1) * PV <= a.HW)) ? fvo + 16u * v : SPX_OOB;
                        xw[v] = buf_load_b128(xir, vo, 0);
                    }
                }
            };
            pt_load(0);
            x_load(0);
            pt_write();
            __syncthreads();
            for (int j = 0; j < nrounds; ++j) {
                const int chb = 2 * j + tsel;
                const bool ch_ok = chb * 32 + frow < Cs;
                const spx_rsrc xir = make_rsrc_pred(x_img + (size_t)(chb * 32) * a.HW * ESZ);
                const spx_rsrc dxr = make_rsrc_pred((char*)a.dx + ((size_t)b * Cs + chb * 32) * a.HW * ESZ);
                float xv[16];
                if (use_vec) {
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (XF32) {
                                xv[4 * v + e] = __uint_as_float(xw[v][e]);
                            } else {
                                xv[8 * v + 2 * e] = __uint_as_float(xw[v][e] << 16);
                                xv[8 * v + 2 * e + 1] = __uint_as_float(xw[v][e] & 0xffff0000u);
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const uint32_t vo = (ch_ok && fpx + e < a.HW) ? fvo + (uint32_t)e * ESZ : SPX_OOB;
                        if (XF32) xv[e] = buf_load_f32(xir, vo, 0);
                        else xv[e] = __uint_as_float((uint32_t)buf_load_u16(xir, vo, 0) << 16);
                    }
                }
                const int jn = j + 1 < nrounds ? j + 1 : j;
                pt_load(jn);
                x_load(jn);
                // this wave's channel block of the round: 2 j + ph, all prototype blocks
                f32x16 accx;
#pragma unroll
                for (int e = 0; e < 16; ++e) accx[e] = 0.0f;
                const char* cur = pt + ph * PTB + lane * 16;
                // the round's fragments in two batches of NPB reads, each issued whole before its MFMAs (left to itself hipcc
                // emits read -> wait -> MFMA per fragment: one exposed LDS round trip per MFMA)
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    f16x8 ptf[NPB];
#pragma unroll
                    for (int f = 0; f < NPB; ++f) ptf[f] = *(const f16x8*)(cur + (hb * NPB + f) * 1024);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int f = 0; f < NPB; ++f) {
#if SPXF_GFR
                        accx = mfma_f16(ptf[f], gfr[hb * NPB + f], accx);
#else
                        accx = mfma_f16(ptf[f], *(const f16x8*)((((hb * NPB + f) & 1) ? gsrc1 : gsrc0) + (hb * NPB + f) * 1024), accx);
#endif
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                char* T = tt + ph * SPXF_T_BYTES;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    *(float*)(T + acc_row(reg, h) * SPXF_T_ROW + (32 * pg + r) * 4) = accx[reg];
                __syncthreads();
                const char* Tf = tt + tsel * SPXF_T_BYTES;
                float ov[16];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const f32x4 tv = *(const f32x4*)(Tf + frow * SPXF_T_ROW + fseg * 64 + v * 16);
                    const f32x4 rv = *(const f32x4*)(rss + fseg * 16 + v * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[4 * v + e] = sinv * __builtin_fmaf(rv[e], xv[4 * v + e], tv[e]);
                }
                if (use_vec) {
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
                        if (XF32) buf_store_f32(ov[e], dxr, vo, 0);
                        else buf_store_u16(__builtin_bit_cast(uint16_t, (__bf16)ov[e]), dxr, vo, 0);
                    }
                }
                pt_write();                                   // (every wave has read the round's fragments; placed behind the finish,
                                                              // the fragment loads have the whole round to arrive)
                __syncthreads();                              // the tiles are rewritten, the fragments read, next round
            }
        }

        SPXF_STAMP(4)
    }
    if (want_bank && (long long)blockIdx.x < ntiles) {
        // phase D of this workgroup's last tile
        long long last = blockIdx.x;
        while (last + gridDim.x < ntiles) last += gridDim.x;
        int b_p, px0_p;
        tile_ctx(last, b_p, px0_p);
        const int tid = spx_opaque(tid0);
        const SpxTileCtx tcp = XSt::make_ctx((const char*)a.x + (size_t)b_p * Cs * a.HW * ESZ, a.HW, px0_p, tid);
        XSt xim[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) xim[c].load(tcp, c * 32, Cs - c * 32);
#pragma unroll
        for (int c = 0; c < 8; ++c) SPXF_D_STAGE_CHUNK(xim[c], tcp, c)
        __syncthreads();
        d_compute();
    }
#ifdef SPX_DIAG_STAMPS
    if (a.dbg && lane0 == 0) {
        for (int k = 0; k < 10; ++k) a.dbg[((size_t)blockIdx.x * 8 + wave) * 10 + k] = dgt[k];
    }
#endif

    // ---- this workgroup's partial slab (the parameter kernel's layout: [row][dP cols | dW cols | colsum]) ----
    if (want_bank) {
        const int lane = spx_opaque(lane0), r = lane & 31, h = lane >> 5;
        const float sinv = first ? 0.0f : __builtin_amdgcn_ldexpf(1.0f, -e_cur);
        const int ws = nchb * 32 + pl.ncb * 32 + 32;
        float* slab = a.workspace + (size_t)blockIdx.x * (NPB * 32) * ws;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int pb = pbw + i;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int chb = 2 * pg + t;
                if (chb < nchb) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        slab[(size_t)(pb * 32 + acc_row(reg, h)) * ws + chb * 32 + r] = accp[i][t][reg] * sinv;
                }
            }
            if (pg == i) {
                const float s = csum[i] + __shfl_xor(csum[i], 32);
                if (h == 0) slab[(size_t)(pb * 32 + r) * ws + nchb * 32 + NCB * 32] = s * sinv;
            }
        }
    }
}

template <int NPB, bool GATHER>
static hipError_t spx_launch_bwdf_g(const SpxBwdFArgs& a, int x_dtype, int grid, hipStream_t s) {
    constexpr size_t lds = (size_t)spxf_lds_bytes<NPB>();
    const int vm = a.HW < 8 ? 0 : (a.HW % 8 == 0 ? 1 : 2);
    dim3 g((unsigned)grid), bl(SPXF_THREADS);
#ifdef SPXF_ONLY_MAIN      // development builds: one instance (bf16 features, whole 8-pixel pieces, P-wide dDist) compiles in seconds
    (void)x_dtype; (void)vm;
    hipLaunchKernelGGL((spx_bwdf_kernel<NPB, false, 1, false>), g, bl, lds, s, a);
    return hipGetLastError();
#else
    if (x_dtype == 1) {
        if (vm == 2) hipLaunchKernelGGL((spx_bwdf_kernel<NPB, true, 2, GATHER>), g, bl, lds, s, a);
        else if (vm == 1) hipLaunchKernelGGL((spx_bwdf_kernel<NPB, true, 1, GATHER>), g, bl, lds, s, a);
        else hipLaunchKernelGGL((spx_bwdf_kernel<NPB, true, 0, GATHER>), g, bl, lds, s, a);
    } else {
        if (vm == 2) hipLaunchKernelGGL((spx_bwdf_kernel<NPB, false, 2, GATHER>), g, bl, lds, s, a);
        else if (vm == 1) hipLaunchKernelGGL((spx_bwdf_kernel<NPB, false, 1, GATHER>), g, bl, lds, s, a);
        else hipLaunchKernelGGL((spx_bwdf_kernel<NPB, false, 0, GATHER>), g, bl, lds, s, a);
    }
    return hipGetLastError();
#endif
}
