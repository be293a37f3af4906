// Fused forward of the prototype-distance path:
//   distances = relu(|x|^2 - 2 x.p + |p|^2)                (model_multiscale.py:255-317)
//   activations = log((d+1)/(d+eps)) | -d                  (:324-330)
//   logits = activations . W^T                             (:243-244, :369-376)
// one launch, one pass over X, distances written once in the reference's [B,P,H,W] layout.
#pragma once
#include "spx_args.h"
#include "spx_mainloop.h"

#ifndef SPX_FWD_WAVES
#define SPX_FWD_WAVES 2
#endif
// X chunks in flight per workgroup (register ring).  A/B on MI355X: 2 beats 4 (0.74 vs 0.78 ms): the loads are not
// what the loop waits for (in-kernel stamps: < 200 cycles per chunk), the extra registers only cost scheduling freedom
#ifndef SPX_FWD_SPLIT
#define SPX_FWD_SPLIT 1
#endif
#ifndef SPX_FWD_XPANEL
#define SPX_FWD_XPANEL 0
#endif
#ifndef SPX_FWD_XRING
#define SPX_FWD_XRING(xf32) 2
#endif

// per-wave LDS scratch of the epilogue's distance-tile turn: 32 prototype rows x 32 pixels fp32, 160-B rows (the
// two half-waves land 32 banks apart: conflict-free writes); aliases the main-loop stages, free after the loop
#ifndef SPX_FWD_TROW
#define SPX_FWD_TROW 40
#endif
#define SPX_GATHER_CLASSES 1024         // classes of the in-kernel slot-count table of the gathered mode (4 KiB of idle stage memory)
#define SPX_FWD_TSCRATCH 8192           // per wave: >= 32 * SPX_FWD_TROW * 4 (distance turn) and 32 px x 64 values (block I/O)
// LDS carve (bytes): [stage 0][stage 1][head fragments of the current panel (NCB == 1 only)][|p|^2 of the panel]
#ifdef SPX_FWD_HEAD_L2
template <int NPB, int NCB>
__host__ __device__ constexpr int spx_fwd_head_lds_bytes() { return 0; }   // experiment: head fragments straight from L2
#else
template <int NPB, int NCB>
__host__ __device__ constexpr int spx_fwd_head_lds_bytes() { return NCB * NPB <= 6 ? NCB * NPB * 4096 : 0; }   // <= 24 KiB: resident in LDS
#endif
// Heads whose panel image does not fit (more than 6 class-block x prototype-block cells: 150 classes, or two class blocks over
// 192-prototype panels) pass ONE prototype block's fragments at a time (NCB x 4 KiB) through LDS, fetched once per workgroup a
// block ahead: as per-wave loads from L2 the same fragments were read four times per tile - 120 KiB per panel and wave for
// 150 classes, 386 MB per launch on the ADE crops.  Measured: that forward 115 -> 100 us; the 2 Mpx grouping forward (two class
// blocks x six prototype blocks) unchanged at 0.91 ms (profiles/EXPERIMENTS.md).
template <int NPB, int NCB>
__host__ __device__ constexpr int spx_fwd_head_blk_bytes() { return spx_fwd_head_lds_bytes<NPB, NCB>() == 0 ? NCB * 4096 : 0; }
// region 0 = the two main-loop stages, re-used after the loop as the waves' epilogue scratch
template <int NPB, int SPLIT>
__host__ __device__ constexpr int spx_fwd_region0_bytes() {
    return 2 * spx_stage_bytes(NPB) > 4 * SPLIT * SPX_FWD_TSCRATCH ? 2 * spx_stage_bytes(NPB) : 4 * SPLIT * SPX_FWD_TSCRATCH;
}
template <int NPB, int NCB, int SPLIT>
__host__ __device__ constexpr int spx_fwd_lds_bytes() {
    return spx_fwd_region0_bytes<NPB, SPLIT>() + spx_fwd_head_lds_bytes<NPB, NCB>() + spx_fwd_head_blk_bytes<NPB, NCB>() + 3 * NPB * 32 * 4 + NPB * 32 * 8;   // + |p|^2, class keys, slot plane offsets, push minima
}

// SPLIT = waves per 32-pixel group.  SPLIT 1: 4 waves, each with all NPB blocks of its pixels (<= 256 VGPRs, two
// waves per SIMD).  SPLIT 2: 8 waves, wave (pg = w & 3, ph = w >> 2) accumulates blocks [ph*NPB/2, (ph+1)*NPB/2) of
// pixel group pg in <= 128 VGPRs: four waves per SIMD instead of two hide the LDS / HBM / transcendental latencies
// that a two-wave SIMD leaves exposed (the kernel was issue-stalled, not bandwidth-bound: MFMA 17 % + VALU ~42 %
// busy); the two halves' logits partials meet in LDS at the end.
// GATHER: class-gathered distances (spx_dist_fwd_cls) instead of the P-wide map.
// (2-block panels with a one-block head fit 168 VGPRs: three waves per SIMD.)
// ACT: the [pixel][P] activation output is requested (kept out of the default instance: its code costs registers).
template <int NPB, int NCB, bool XF32, int VM, int SPLIT, bool GATHER, bool ACT>
__global__ __launch_bounds__(256 * SPLIT, (NPB == 2 && NCB == 1 ? 3 : SPX_FWD_WAVES) * SPLIT) void spx_fwd_kernel(const SpxFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = 256 * SPLIT, NH = NPB / SPLIT;
    static_assert(NPB % SPLIT == 0, "blocks must split evenly over the waves of a pixel group");
    const spx_plan& pl = a.plan;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave & 3, ph = wave >> 2;       // pixel group, prototype half
    const int r = lane & 31, h = lane >> 5;
    const int tiles_per_img = (a.HW + SPX_TILE_PX - 1) / SPX_TILE_PX;
    const int b = blockIdx.x / a.tiles_launch;
    const int tile_i = a.tile_first + (int)(((long long)(blockIdx.x % a.tiles_launch) * a.tile_mul) % a.tiles_launch);
    const int px0 = tile_i * SPX_TILE_PX;
    const int Cs = pl.channels_per_scale;
    const int C = pl.num_scales * Cs;
    const int P = pl.num_prototypes, K = pl.num_classes;
    const uint32_t HW = (uint32_t)a.HW;
    // scale-parallel launch: this workgroup's panels and its (partial) logits plane
    const int q_begin = a.ngroups > 1 ? a.group_first[blockIdx.y] : 0;
    const int q_end = a.ngroups > 1 ? a.group_first[blockIdx.y + 1] : pl.npanels;
    float* const logits_out = a.logits ? a.logits + (size_t)blockIdx.y * a.logits_group_stride : nullptr;
    constexpr int XR = SPX_FWD_XRING(XF32);
    using Pipe = SpxPipeline<NPB, XF32, VM, XR, NT, NH>;

#ifdef SPX_DIAG_STAGGER
    // experiment: de-phase the two workgroups that share a CU (blocks i and i+256 of the first dispatch round)
    if (blockIdx.x >= 256 && blockIdx.x < 512)
        for (int i = 0; i < SPX_DIAG_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
#endif
    const SpxTileCtx tc = SpxXStager<XF32, VM, NT>::make_ctx((const char*)a.x + (size_t)b * C * a.HW * (XF32 ? 4 : 2), a.HW, px0, tid);

    constexpr int chunk_bytes = NPB * 2 * 1024;
    constexpr int head_lds = spx_fwd_head_lds_bytes<NPB, NCB>();
    constexpr int region0 = spx_fwd_region0_bytes<NPB, SPLIT>();
    const int nchunks = (Cs + SPX_KC - 1) / SPX_KC;
    char* const wlds = smem + region0;
    constexpr int head_blk = spx_fwd_head_blk_bytes<NPB, NCB>();
    constexpr bool HSTREAM = head_blk != 0 && NT == 256;     // (the 8-wave experiment keeps the per-wave loads)
    float* const p2s = (float*)(wlds + head_lds + head_blk);
    uint32_t* const keys = (uint32_t*)(p2s + NPB * 32);     // GATHER: (class << 16) | slot per padded prototype row
    uint32_t* const koff = keys + NPB * 32;                 // GATHER: byte offset of the row's slot plane (slot * HW * 4)
    unsigned long long* const pmin = (unsigned long long*)(koff + NPB * 32);   // fused push: the workgroup's (value key, pixel) minimum per row

    Pipe pipe;
    f32x16 acc[NH];
    f32x16 accl[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) accl[cb][i] = 0.0f;
    // SPX_FWD_DUAL_HEAD: the one-class-block head keeps a second logits accumulator for k-step 1 of every block, so the six
    // head MFMAs of a block form two dependent chains of three instead of one of six (summed once, after the last panel)
#ifndef SPX_FWD_DUAL_HEAD
#define SPX_FWD_DUAL_HEAD 0
#endif
    constexpr bool DUAL = SPX_FWD_DUAL_HEAD && NCB == 1 && SPLIT == 1;
    f32x16 accl2;
#pragma unroll
    for (int i = 0; i < 16; ++i) accl2[i] = 0.0f;
#pragma unroll
    for (int pb = 0; pb < NH; ++pb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[pb][i] = 0.0f;
    float x2part = 0.0f;

    const int px = px0 + 32 * pg + r;          // this lane's pixel
    const bool px_ok = px < a.HW;
    const bool want_head = a.logits != nullptr;
    // per-lane byte offsets, fixed for the whole kernel (SPX_OOB = access dropped); row / block selection
    // rides on wave-uniform SGPR offsets
    const uint32_t voff_d = px_ok ? ((uint32_t)(4 * h) * HW + (uint32_t)px) * 4u : SPX_OOB;                 // [row][px]
    // wide distance stores: lane = (row lane>>3 of an 8-row group, 4 pixels 4*(lane&7)..+3 of the wave's 32)
    const int pxw = px0 + 32 * pg + 4 * (lane & 7);
    const uint32_t voff_dw = pxw < a.HW ? ((uint32_t)(lane >> 3) * HW + (uint32_t)pxw) * 4u : SPX_OOB;
    const uint32_t voff_a = px_ok ? ((uint32_t)px * (uint32_t)P + (uint32_t)(4 * h)) * 4u : SPX_OOB;        // [px][row]
    const spx_rsrc hr = make_rsrc_pred(a.packed_head);
    const spx_rsrc p2r = make_rsrc_pred(a.p2);
    // GATHER: this pixel's class (0xFFFE = none; padding rows carry class 0xFFFF) and its [px][slot] row offset
    uint32_t lab16 = 0xFFFEu, voff_c = SPX_OOB;
    const spx_rsrc keyr = make_rsrc_pred(GATHER ? a.proto_key : nullptr);
    const spx_rsrc cdr = make_rsrc_pred(GATHER ? a.cls_dist + (size_t)b * a.J * a.HW : nullptr);
    if (GATHER) {
        const spx_rsrc labr = make_rsrc_pred(a.labels + (size_t)b * a.HW);
        const uint32_t l = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(labr, px_ok ? (uint32_t)px * 4u : SPX_OOB, 0, 0);
        lab16 = (px_ok && l < 0xFFFEu) ? l : 0xFFFEu;
        if (a.push_keys) {
            // the push's raw labels (push_multiscale_optimization.py:74-83: one_hot over K + 1 values, the void column dropped)
            const int li = (int)l;
            int c = li;
            if (a.push_void >= 0) c = li == a.push_void ? -1 : (li < a.push_void ? li : li - 1);
            lab16 = (px_ok && c >= 0 && c < a.push_K) ? (uint32_t)c : 0xFFFEu;
        }
        voff_c = px_ok ? (uint32_t)px * 4u : SPX_OOB;          // [slot][px] planes: a wave's 32 pixels are one 128-B run
        if (a.cls_dist && blockIdx.y == 0) {
            // Slots no prototype maps to - every slot of a pixel without a class, the slots past its class's prototype count -
            // are written as zeros HERE (the caller hands over uninitialised planes: a memset of 40 B/px less per forward).
            // The counts come from the key table (1 + the largest slot of a class), built once per workgroup in the still idle
            // stage memory; a slot that gets a zero never gets a value, so the stores below cannot race with these.
            // (label classes are the key table's, not the plan's head rows: a table of SPX_GATHER_CLASSES entries; a class beyond
            // it keeps all its slots - the host side zero-fills for such tables itself)
            int* const cnt_s = (int*)smem;
            for (int i = tid; i < SPX_GATHER_CLASSES; i += NT) cnt_s[i] = 0;
            __syncthreads();
            for (int i = tid; i < pl.npanels * NPB * 32; i += NT) {
                const uint32_t key = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(keyr, (uint32_t)i * 4u, 0, 0);
                if ((key >> 16) < (uint32_t)SPX_GATHER_CLASSES) atomicMax(cnt_s + (key >> 16), (int)(key & 0xFFFFu) + 1);
            }
            __syncthreads();
            const int cnt = lab16 < (uint32_t)SPX_GATHER_CLASSES ? cnt_s[lab16] : (lab16 == 0xFFFEu ? 0 : a.J);
            if (px_ok && lane < 32)
                for (int j = cnt; j < a.J; ++j) buf_store_f32(0.0f, cdr, voff_c + (uint32_t)j * (uint32_t)a.HW * 4u, 0);
            __syncthreads();                                   // the stages are the pipeline's from here on
        }
    }

    // panel prologue: head fragments + |p|^2 of the panel -> LDS (read in the epilogue, after >= 1 barrier)
    constexpr int HPB = NT * 16;                        // bytes per pass of the whole workgroup
    constexpr int HPASS = (head_lds + HPB - 1) / HPB;
    static_assert(head_lds % HPB == 0 || head_lds < HPB, "head fragments: whole passes, or one partial pass");
    static_assert(NCB == 1 || (NPB * 4096) % HPB == 0, "a pass must not straddle two class blocks");
    const bool h_in = head_lds >= HPB || tid * 16 < head_lds;
    u32x4 hreg[HPASS > 0 ? HPASS : 1];
    float p2reg = 0.0f;
    uint32_t keyreg = 0xFFFFFFFFu;
    auto consts_issue = [&](int panel) {
#pragma unroll
        for (int i = 0; i < HPASS; ++i) {
            // packed head = [class block][panel][block][k-step][hi|lo]: the panel's share of class block cb is
            // NPB * 4 KiB; the LDS image is [class block][block][k-step][hi|lo]
            const int cb = (i * HPB) / (NPB * 4096), rest = (i * HPB) % (NPB * 4096);
            hreg[i] = buf_load_b128(hr, (want_head && h_in) ? (uint32_t)(rest + tid * 16) : SPX_OOB,
                                    (uint32_t)((cb * pl.npanels + panel) * NPB * 4096));
        }
        p2reg = buf_load_f32(p2r, tid < NPB * 32 ? (uint32_t)tid * 4u : SPX_OOB, (uint32_t)(panel * NPB * 32 * 4));
        if (GATHER) keyreg = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(keyr, tid < NPB * 32 ? (uint32_t)tid * 4u : SPX_OOB, (uint32_t)(panel * NPB * 32 * 4), 0);
    };
    auto consts_commit = [&]() {
#pragma unroll
        for (int i = 0; i < HPASS; ++i)
            if (h_in) *(u32x4*)(wlds + i * HPB + tid * 16) = hreg[i];
        if (tid < NPB * 32) p2s[tid] = p2reg;
        if (GATHER && tid < NPB * 32) {
            keys[tid] = keyreg;
            koff[tid] = (keyreg & 0xFFFFu) * HW * 4u;
            if (a.push_keys) pmin[tid] = ~0ull;
        }
    };

    // Panel epilogue as a ROLLED loop over the panel's 32-prototype blocks (the body is compiled once, with a
    // small fixed register footprint, instead of NPB unrolled copies); the block's accumulator tile is fetched
    // with a wave-uniform select over static register indices and cleared for the next panel.
    // HSTREAM: the next block's head fragments (thread t: bytes [16 t, 16 t + 16) of each class block's 4 KiB) on their way to LDS
    u32x4 wreg[HSTREAM ? NCB : 1];
    auto w_issue = [&](int panel, int pb) {
#pragma unroll
        for (int cb = 0; cb < (HSTREAM ? NCB : 0); ++cb)
            wreg[cb] = buf_load_b128(hr, want_head ? (uint32_t)tid * 16u : SPX_OOB, (uint32_t)(((cb * pl.npanels + panel) * NPB + pb) * 4096));
    };
    auto epilogue = [&](int panel) {
        const float x2 = x2part + __shfl_xor(x2part, 32);
        const int p0 = pl.panel_p0[panel], np = pl.panel_np[panel];
        if (HSTREAM && panel == q_begin) w_issue(panel, ph * NH);
#pragma unroll 1
        for (int pbl = 0; pbl < NH; ++pbl) {
            const int pb = ph * NH + pbl;              // block index inside the panel
            if (HSTREAM) {
                __syncthreads();                       // every wave is done with the previous block's fragments
#pragma unroll
                for (int cb = 0; cb < (HSTREAM ? NCB : 0); ++cb) *(u32x4*)(wlds + cb * 4096 + tid * 16) = wreg[cb];
                __syncthreads();
                if (pbl + 1 < NH) w_issue(panel, pb + 1);
                else if (panel + 1 < q_end) w_issue(panel + 1, ph * NH);
            }
            const f32x16 tile = tile_get<NH>(acc, pbl);
            if (pb * 32 < np) {
                const spx_rsrc dr = make_rsrc_pred(a.dist ? a.dist + ((size_t)b * P + p0 + pb * 32) * a.HW : nullptr);
                const spx_rsrc ar = make_rsrc_pred(ACT ? a.act + (size_t)b * a.HW * P + p0 + pb * 32 : nullptr);
                const bool full = pb * 32 + 32 <= np;       // wave-uniform: no per-row predication needed
                // all arithmetic first (one straight-line block), then the stores under wave-uniform conditions
                float dv[16], av[16];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    // rows 8*g4 + 4h + (0..3) of the block <-> registers 4*g4..4*g4+3
                    const f32x4 p2v = *(const f32x4*)(p2s + pb * 32 + 8 * g4 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dv[4 * g4 + e] = fmaxf(__builtin_fmaf(-2.0f, tile[4 * g4 + e], p2v[e]) + x2, 0.0f);
                }
                if (a.act_fn == 0) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) av[reg] = act_log(dv[reg], a.eps);
                } else {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) av[reg] = -dv[reg];
                }
                if (GATHER && a.push_keys) {
                    // Fused prototype push (push_multiscale_optimization.py:74-91): v = d + max_dist * (1 - mask) with the
                    // reference's rounding, its minimum over the wave's 32 pixels per prototype row (lowest pixel index on
                    // ties) as an integer minimum into the workgroup's LDS row table; the table meets the global one once
                    // per panel (below).  The P-wide map never exists.
                    float* const sc = (float*)(smem + wave * SPX_FWD_TSCRATCH);
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const u32x4 kv = *(const u32x4*)(keys + pb * 32 + 8 * g4 + 4 * h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float one_minus_mask = (kv[e] >> 16) == lab16 ? 0.0f : 1.0f;
                            const float v = px_ok ? dv[4 * g4 + e] + a.push_max * one_minus_mask : __builtin_inff();
                            sc[(e + 8 * g4 + 4 * h) * SPX_FWD_TROW + r] = v;
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = 8 * q + (lane >> 3);
                        const f32x4 v = *(const f32x4*)(sc + row * SPX_FWD_TROW + 4 * (lane & 7));
                        float best = v[0];
                        int bi = pxw;
#pragma unroll
                        for (int e = 1; e < 4; ++e)
                            if (v[e] < best) { best = v[e]; bi = pxw + e; }
#pragma unroll
                        for (int m = 1; m <= 4; m <<= 1) {
                            const float ov = __shfl_xor(best, m);
                            const int oi = __shfl_xor(bi, m);
                            if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
                        }
                        if ((lane & 7) == 0) {
                            const unsigned long long key = ((unsigned long long)float_key(best + 0.0f) << 32) | (uint32_t)bi;
                            __hip_atomic_fetch_min(pmin + pb * 32 + row, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                } else if (GATHER) {
                    // a lane stores the rows whose class is its pixel's class, at [px][slot]; everything else is
                    // dropped by the out-of-range offset (rows of one class are few: ~P/K of the 32 per block).
                    // Label maps are piecewise constant, so most (wave, block) pairs have no match at all: one
                    // ballot skips their 16 store instructions.
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
                        for (int reg = 0; reg < 16; ++reg) buf_store_f32(dv[reg], cdr, vo[reg], 0);
                    }
                } else if (a.dist && a.dist_vec) {
                    // store-issue is what bounds this epilogue (one VMEM instruction per 256 B with a pixel per
                    // lane): turn the tile through the wave's LDS scratch so a lane owns 4 consecutive pixels of
                    // one prototype row -> 4 16-B stores per block instead of 16 dword stores, same bytes
                    float* const sc = (float*)(smem + wave * SPX_FWD_TSCRATCH);
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) sc[((reg & 3) + 8 * (reg >> 2) + 4 * h) * SPX_FWD_TROW + r] = dv[reg];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = *(const f32x4*)(sc + (8 * q + (lane >> 3)) * SPX_FWD_TROW + 4 * (lane & 7));
                        const uint32_t vo = (full || (pb * 32 + 8 * q + (lane >> 3) < np)) ? voff_dw : SPX_OOB;
                        buf_store_b128(__builtin_bit_cast(u32x4, v), dr, vo, (uint32_t)(8 * q) * HW * 4u);
                    }
                } else if (a.dist) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int rb = (reg & 3) + 8 * (reg >> 2);
                        const uint32_t vo = (full || (pb * 32 + rb + 4 * h < np)) ? voff_d : SPX_OOB;
                        buf_store_f32(dv[reg], dr, vo, (uint32_t)rb * HW * 4u);
                    }
                }
                if (ACT) {
                    // [pixel][P] rows: with a pixel per lane a store instruction scatters 64 dwords over 64 rows
                    // (0.6 ms of pattern cost per 2 Mpx, measured).  Turned through the wave's scratch a lane owns
                    // one prototype column: each instruction writes two 128-B row pieces.
                    float* const sc = (float*)(smem + wave * SPX_FWD_TSCRATCH);
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) sc[r * 33 + (reg & 3) + 8 * (reg >> 2) + 4 * h] = av[reg];
                    const int col = lane & 31;                                  // prototype inside the block
                    const bool col_ok = pb * 32 + col < np;
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int pl_ = 2 * q + (lane >> 5);                    // pixel inside the wave's 32
                        const int pxs = px0 + 32 * pg + pl_;
                        const uint32_t vo = (col_ok && pxs < a.HW) ? ((uint32_t)pxs * (uint32_t)P + (uint32_t)col) * 4u : SPX_OOB;
                        buf_store_f32(sc[pl_ * 33 + col], ar, vo, 0);
                    }
                }
                if (want_head) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        bf16x8 ahi, alo;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            __bf16 hi, lo;
                            split_bf16(av[8 * s2 + j], hi, lo);
                            ahi[j] = hi;
                            alo[j] = lo;
                        }
#pragma unroll
                        for (int cb = 0; cb < NCB; ++cb) {
                            bf16x8 whi, wlo;
                            if (head_lds) {
                                const char* wf = wlds + ((cb * NPB + pb) * 2 + s2) * 2048 + lane * 16;
                                whi = *(const bf16x8*)wf;
                                wlo = *(const bf16x8*)(wf + 1024);
                            } else if (HSTREAM) {
                                const char* wf = wlds + (cb * 2 + s2) * 2048 + lane * 16;
                                whi = *(const bf16x8*)wf;
                                wlo = *(const bf16x8*)(wf + 1024);
                            } else {
                                const uint32_t so = (uint32_t)((((cb * pl.npanels + panel) * NPB + pb) * 2 + s2) * 2048);
                                whi = __builtin_bit_cast(bf16x8, buf_load_b128(hr, (uint32_t)lane * 16u, so));
                                wlo = __builtin_bit_cast(bf16x8, buf_load_b128(hr, (uint32_t)lane * 16u, so + 1024u));
                            }
                            if (DUAL && s2 == 1) {
                                accl2 = mfma_bf16(whi, ahi, accl2);
                                accl2 = mfma_bf16(wlo, ahi, accl2);
                                accl2 = mfma_bf16(whi, alo, accl2);
                            } else {
                                accl[cb] = mfma_bf16(whi, ahi, accl[cb]);
                                accl[cb] = mfma_bf16(wlo, ahi, accl[cb]);
                                accl[cb] = mfma_bf16(whi, alo, accl[cb]);
                            }
                        }
                    }
                }
            }
        }
        if (GATHER && a.push_keys) {
            // one global integer atomicMin per prototype row and workgroup - skipped when the row's current minimum (read
            // past the vector cache; it only ever decreases) is already lower: a single memory round trip per panel
            __syncthreads();
            if (tid < np) {
                unsigned long long* const gk = a.push_keys + (size_t)b * P + p0 + tid;
                const unsigned long long key = pmin[tid];
                if (key < __hip_atomic_load(gk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(gk, key);
            }
        }
        // the next panel (if any) accumulates from zero
        if (panel + 1 < q_end) {
#pragma unroll
            for (int pb = 0; pb < NH; ++pb)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[pb][i] = 0.0f;
        }
    };

#ifdef SPX_DIAG_STAMPS
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), t1 = 0, t2 = 0;
#endif
    // SPX_FWD_XPANEL: fetch the NEXT panel's first chunks and constants before the current panel's epilogue (2-block
    // panels have the register room).  Measured on the 4-scale bank at 2 Mpx: 0.705 vs 0.692 ms without - the fill is
    // not what the multi-panel forward waits for - so it is off.
    constexpr bool PREFETCH_NEXT = SPX_FWD_XPANEL && NPB == 2 && NCB <= 2;
    auto bank_of = [&](int panel) { return a.packed_bank + (size_t)(panel * nchunks) * chunk_bytes; };
    pipe.issue_prologue(tc, bank_of(q_begin), pl.panel_ch0[q_begin], Cs, tid, [&]() { consts_issue(q_begin); });
    for (int panel = q_begin; panel < q_end; ++panel) {
        x2part = 0.0f;
        pipe.run_body(acc, x2part, tc, smem, bank_of(panel), pl.panel_ch0[panel], Cs, lane, wave, tid, consts_commit);
        const bool more = panel + 1 < q_end;
        if (PREFETCH_NEXT && more)
            pipe.issue_prologue(tc, bank_of(panel + 1), pl.panel_ch0[panel + 1], Cs, tid, [&]() { consts_issue(panel + 1); });
#ifdef SPX_DIAG_STAMPS
        t1 = __builtin_amdgcn_s_memtime();
#endif
        epilogue(panel);
#ifdef SPX_DIAG_STAMPS
        t2 = __builtin_amdgcn_s_memtime();
#endif
        if (more) {
            __syncthreads();   // the next panel's body overwrites the head / |p|^2 / stage LDS
            if (!PREFETCH_NEXT)
                pipe.issue_prologue(tc, bank_of(panel + 1), pl.panel_ch0[panel + 1], Cs, tid, [&]() { consts_issue(panel + 1); });
        }
    }

    if (DUAL) {
#pragma unroll
        for (int i = 0; i < 16; ++i) accl[0][i] += accl2[i];
    }
    if (want_head && SPLIT == 2) {
        // the upper prototype half hands its logits partial to the lower one through its own scratch tile
        // (wave-private until here: its last transposed reads were issued before, LDS serves a wave in order)
        float* const sc = (float*)(smem + (4 + pg) * SPX_FWD_TSCRATCH);
        static_assert(NCB == 1 || SPLIT == 1, "the split kernel carries one class block");
        if (ph == 1) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) sc[reg * 64 + lane] = accl[0][reg];
        }
        __syncthreads();
        if (ph == 0) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) accl[0][reg] += sc[reg * 64 + lane];
        }
    }
    // A wave's 32 pixels x n values of a [pixel][n] fp32 tensor (logits, group activations) are ONE contiguous block of
    // 128 n bytes in memory.  Written value by value from the accumulator layout they are n dword stores per lane with
    // an n*4-byte lane stride (each instruction touches 32+ cache lines: 0.9 ms per 2 Mpx for n = 57, measured); turned
    // through the wave's LDS scratch they leave as n/2 fully coalesced dword stores.
    const int pxw0 = px0 + 32 * pg;
    float* const bsc = (float*)(smem + wave * SPX_FWD_TSCRATCH);
    auto block_flush = [&](float* gimg, int n) {      // bsc holds [32][n]; gimg = tensor base of image b
        const int nvalid = (a.HW - pxw0 < 32 ? (a.HW - pxw0 > 0 ? a.HW - pxw0 : 0) : 32) * n;
        const spx_rsrc rs = make_rsrc_pred(gimg + (size_t)pxw0 * n);
#pragma unroll 1
        for (int i = lane; i < 32 * n; i += 64) buf_store_f32(bsc[i], rs, i < nvalid ? (uint32_t)i * 4u : SPX_OOB, 0);
    };
    if (want_head && SPLIT == 2) __syncthreads();     // (split variant: the hand-off scratch above is read before it is reused)
    if (want_head && ph == 0 && a.packed_tail) {
        // grouping-head tail (model_multiscale_group.py:303-308): g = exp(units), logits = W_g . g.  The unit tiles
        // are the B operand of a second split-bf16 product, exactly as the activation tiles were for the head.
        const int K2 = a.K2;
        const spx_rsrc tr = make_rsrc(a.packed_tail);
        const bool blk = NCB <= 2;                    // 32 px x (<= 64) values fit the scratch
        const spx_rsrc gor = make_rsrc_pred(a.gact ? a.gact + (size_t)b * a.HW * K : nullptr);
        const uint32_t voff_g = px_ok ? ((uint32_t)px * (uint32_t)K + (uint32_t)(4 * h)) * 4u : SPX_OOB;    // [px][unit]
        f32x16 acc2;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc2[i] = 0.0f;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            float gv[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int u = cb * 32 + acc_row(reg, h);
                gv[reg] = u < K ? __builtin_amdgcn_exp2f(accl[cb][reg] * 1.44269504089f) : 0.0f;
                if (a.gact) {
                    if (blk) {
                        if (u < K) bsc[r * K + u] = gv[reg];
                    } else {
                        buf_store_f32(gv[reg], gor, u < K ? voff_g : SPX_OOB, (uint32_t)((cb * 32 + (reg & 3) + 8 * (reg >> 2)) * 4));
                    }
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 ghi, glo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    __bf16 hi, lo;
                    split_bf16(gv[8 * s2 + j], hi, lo);
                    ghi[j] = hi;
                    glo[j] = lo;
                }
                const uint32_t so = (uint32_t)((cb * 2 + s2) * 2048);
                const bf16x8 whi = __builtin_bit_cast(bf16x8, buf_load_b128(tr, (uint32_t)lane * 16u, so));
                const bf16x8 wlo = __builtin_bit_cast(bf16x8, buf_load_b128(tr, (uint32_t)lane * 16u, so + 1024u));
                acc2 = mfma_bf16(whi, ghi, acc2);
                acc2 = mfma_bf16(wlo, ghi, acc2);
                acc2 = mfma_bf16(whi, glo, acc2);
            }
        }
        if (a.gact && blk) block_flush(a.gact + (size_t)b * a.HW * K, K);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            if (acc_row(reg, h) < K2) bsc[r * K2 + acc_row(reg, h)] = acc2[reg];
        block_flush(a.logits + (size_t)b * a.HW * K2, K2);
    } else if (want_head && ph == 0) {
        if (a.ce_labels) {
            // fused cross entropy: logsumexp, the label's logit and the argmax of this lane's pixel while the logits
            // tile is in registers (rows = classes, the two lane halves hold different rows of the same pixel)
            const spx_rsrc clr = make_rsrc_pred(a.ce_labels + (size_t)b * a.HW);
            const int lab = (int)__builtin_amdgcn_raw_buffer_load_b32(clr, px_ok ? (uint32_t)px * 4u : SPX_OOB, 0, 0);
            const bool valid = px_ok && (uint32_t)lab < (uint32_t)K;
            float m = -3.0e38f;
            int best = 0x7fffffff;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int cls = cb * 32 + acc_row(reg, h);
                    if (cls < K) ce_best(accl[cb][reg], cls, m, best);
                }
            ce_best(__shfl_xor(m, 32), __shfl_xor(best, 32), m, best);
            float ssum = 0.0f, picked = 0.0f;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int cls = cb * 32 + acc_row(reg, h);
                    if (cls < K) {
                        ssum += ce_exp(accl[cb][reg] - m);
                        picked += cls == lab ? accl[cb][reg] : 0.0f;
                    }
                }
            ssum += __shfl_xor(ssum, 32);
            picked += __shfl_xor(picked, 32);
            const float lse = m + ce_log(ssum);
            const uint32_t vo1 = (px_ok && h == 0) ? (uint32_t)px * 4u : SPX_OOB;
            buf_store_f32(lse, make_rsrc_pred(a.ce_lse + (size_t)b * a.HW), vo1, 0);
            if (a.ce_pred) buf_store_f32(__int_as_float(best), make_rsrc_pred(a.ce_pred + (size_t)b * a.HW), vo1, 0);
            float lossv = (valid && h == 0) ? lse - picked : 0.0f, cnt = (valid && h == 0) ? 1.0f : 0.0f;
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) {
                lossv += __shfl_xor(lossv, off);
                cnt += __shfl_xor(cnt, off);
            }
            if (lane == 0) {
                float* const pp = a.ce_partials + (((size_t)b * tiles_per_img + tile_i) * 4 + pg) * 2;
                pp[0] = lossv;
                pp[1] = cnt;
            }
        }
        if (NCB <= 2) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int cls = cb * 32 + acc_row(reg, h);
                    if (cls < K) bsc[r * K + cls] = accl[cb][reg];
                }
            block_flush(logits_out + (size_t)b * a.HW * K, K);
        } else {
            const spx_rsrc lr = make_rsrc_pred(logits_out + (size_t)b * a.HW * K);
            const uint32_t voff_l = px_ok ? ((uint32_t)px * (uint32_t)K + (uint32_t)(4 * h)) * 4u : SPX_OOB;   // [px][class]
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int cls = cb * 32 + acc_row(reg, h);
                    buf_store_f32(accl[cb][reg], lr, cls < K ? voff_l : SPX_OOB, (uint32_t)((cb * 32 + (reg & 3) + 8 * (reg >> 2)) * 4));
                }
            }
        }
    }
#ifdef SPX_DIAG_STAMPS
    if (a.dbg && tid == 0) {
        unsigned long long t3 = __builtin_amdgcn_s_memtime();
        unsigned long long* d = a.dbg + (size_t)blockIdx.x * 4;
        d[0] = t0; d[1] = t1; d[2] = t2; d[3] = t3;
        unsigned long long* e = a.dbg + (size_t)gridDim.x * 4 + (size_t)blockIdx.x * 4;
        e[0] = pipe.dg_compute; e[1] = pipe.dg_write; e[2] = pipe.dg_barrier; e[3] = pipe.dg_issue;
    }
#endif
}

template <int NPB, int NCB, int SPLIT, bool GATHER, bool ACT>
static hipError_t launch_fwd_ga(const SpxFwdArgs& a, int x_dtype, dim3 grid, hipStream_t s) {
    constexpr size_t lds = (size_t)spx_fwd_lds_bytes<NPB, NCB, SPLIT>();
    const dim3 blk(256 * SPLIT);
    // a.vec_ok: 0 = element-wise staging, 1 = vector staging, 2 = vector staging with a ragged image end (H*W % 8 != 0)
    if (x_dtype == 1) {
        if (a.vec_ok == 2) hipLaunchKernelGGL((spx_fwd_kernel<NPB, NCB, true, 2, SPLIT, GATHER, ACT>), grid, blk, lds, s, a);
        else if (a.vec_ok) hipLaunchKernelGGL((spx_fwd_kernel<NPB, NCB, true, 1, SPLIT, GATHER, ACT>), grid, blk, lds, s, a);
        else hipLaunchKernelGGL((spx_fwd_kernel<NPB, NCB, true, 0, SPLIT, GATHER, ACT>), grid, blk, lds, s, a);
    } else {
        if (a.vec_ok == 2) hipLaunchKernelGGL((spx_fwd_kernel<NPB, NCB, false, 2, SPLIT, GATHER, ACT>), grid, blk, lds, s, a);
        else if (a.vec_ok) hipLaunchKernelGGL((spx_fwd_kernel<NPB, NCB, false, 1, SPLIT, GATHER, ACT>), grid, blk, lds, s, a);
        else hipLaunchKernelGGL((spx_fwd_kernel<NPB, NCB, false, 0, SPLIT, GATHER, ACT>), grid, blk, lds, s, a);
    }
    return hipGetLastError();
}
template <int NPB, int NCB, int SPLIT>
static hipError_t launch_fwd_x(const SpxFwdArgs& a, int x_dtype, dim3 grid, hipStream_t s) {
    if (a.labels) return a.act ? launch_fwd_ga<NPB, NCB, SPLIT, true, true>(a, x_dtype, grid, s)
                               : launch_fwd_ga<NPB, NCB, SPLIT, true, false>(a, x_dtype, grid, s);
    return a.act ? launch_fwd_ga<NPB, NCB, SPLIT, false, true>(a, x_dtype, grid, s)
                 : launch_fwd_ga<NPB, NCB, SPLIT, false, false>(a, x_dtype, grid, s);
}

// one translation unit per panel height (SPX_TU_NPB), so the variants compile in parallel
// Features rows need no alignment: gfx950 under ROCm serves 16-B buffer accesses at any byte address
// (tools/ubench/unaligned.hip), so the vector staging path only asks that every 8-pixel piece lies wholly inside or
// outside the image.  That holds for every tile except an image's last one when H*W is not a multiple of 8 - and it is
// odd for every grid the reference uses (65 x 65 crops, 129 x 257 images): the ONE piece per feature row that straddles
// the image end is loaded from a window moved back to end exactly at the image end and shifted into place in registers
// (SpxXStager::make_ctx / fix_ragged), so nothing is read past the tensor and the whole launch stays on the vector path.
template <int NPB>
static hipError_t spx_launch_fwd_tiles(const SpxFwdArgs& a, int x_dtype, hipStream_t s) {
    const spx_plan& pl = a.plan;
    dim3 grid((unsigned)(a.tiles_launch * a.B), (unsigned)(a.ngroups > 1 ? a.ngroups : 1));
    // SPX_FWD_SPLIT 2 = 8-wave workgroups (4 waves per SIMD): measured 0.84 vs 0.72 ms at the north-star shape, off
    if (pl.ncb == 1) return launch_fwd_x<NPB, 1, SPX_FWD_SPLIT>(a, x_dtype, grid, s);
    // 33..64 head rows (the grouping head: 3 groups x 19 / 21 classes)
    if (pl.ncb == 2) return launch_fwd_x<NPB, 2, 1>(a, x_dtype, grid, s);
    // up to 160 classes: 80 logits accumulators per lane
    return launch_fwd_x<NPB, 5, 1>(a, x_dtype, grid, s);
}
// one translation unit per panel height (SPX_TU_NPB), so the variants compile in parallel
template <int NPB>
static hipError_t spx_launch_fwd_npb(const SpxFwdArgs& a0, int x_dtype, hipStream_t s) {
    SpxFwdArgs a = a0;
    a.vec_ok = a.HW < 8 ? 0 : (a.HW % 8 == 0 ? 1 : 2);      // the element-wise path only for images of fewer than 8 pixels
    a.tile_first = 0;
    a.tiles_launch = (a.HW + SPX_TILE_PX - 1) / SPX_TILE_PX;
    a.tile_mul = spx_tile_mul(a.tiles_launch, (long long)a.HW * (x_dtype == 1 ? 4 : 2));
    return spx_launch_fwd_tiles<NPB>(a, x_dtype, s);
}
