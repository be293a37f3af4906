// The x.p contraction shared by the forward and the pixel-side backward kernel.
//
// One workgroup = 4 waves = one tile of 128 latent pixels of one image.  Each wave owns
// 32 pixels (the MFMA column) against all 32*NPB prototypes of the current panel (the MFMA
// rows): D^T[proto x pixel] += Bank[proto x k] . X[k x pixel], 32x32x16 bf16 MFMA,
// fp32 accumulate.  The prototype-major orientation makes the accumulator's lane index the
// pixel index, so distance rows are written as 128-B pixel runs of the reference's
// [B,P,H,W] layout and the tile can feed the head MFMA as a B operand without leaving registers.
//
// Per step (one K-chunk of 32 channels of one panel) the workgroup stages
//   - X[32 x 128 px]  -> LDS as bf16 [k][pixel] rows (320-B stride), read back with
//     ds_read_b64_tr_b16 so every lane receives the 8 k-values of its own pixel;
//   - the panel's bank chunk, already in MFMA A-fragment order in HBM/L2, -> LDS verbatim
//     (lane-linear image, conflict-free ds_read_b128).
//
// Everything inside the step is STRAIGHT-LINE code: the panel height NPB is a template constant (the plan pads
// panels to 2, 4 or 6 blocks), the chunk is always 32 channels (a scale whose width is 16 mod 32 gets a
// zero-filled tail), and tails are predicated by out-of-range buffer offsets instead of branches.  With
// runtime-conditional loads or MFMAs hipcc puts each of them in its own basic block, waits vmcnt(0) /
// lgkmcnt(0) in front of every use and the pipeline below degenerates into exposed latencies.
#pragma once
#include "spx_common.h"

#define SPX_KC 32                          // channels per staged chunk
#define SPX_STAGE_X_BYTES (SPX_KC * SPX_XROW * 2)

struct SpxTileCtx {
    const char* x_img;    // features of image b, channel 0, pixel 0
    uint32_t x_voff;      // this thread's byte offset inside a pass: (row0 * HW + px) * esz, or SPX_OOB
    uint32_t hw;          // pixels per image
    int px;               // first of this thread's 8 staged pixels
    int row0;             // this thread's row inside a pass of NT/16 rows
    uint32_t rot_bits;    // vector path: this thread's piece straddles the image end - loaded from a window moved back by
                          // this many bits' worth of elements and shifted into place at LDS-commit time (0 = ordinary piece)
    bool ragged;          // tile-uniform: some piece of this tile straddles the image end
};

// One staged K-chunk: acc[i] += Bank_chunk[pb0 + i] . X_chunk (i < NH: the wave's share of the panel's NPB blocks),
// x2 += |x|^2 partial (this lane's k-half).  pg = the wave's 32-pixel group inside the 128-pixel tile.
template <int NPB, int NH>
__device__ __forceinline__ void spx_compute_chunk(f32x16 (&acc)[NH], float& x2part, const char* xs, const char* as,
                                                  int lane, int pg, int pb0) {
    constexpr int NKS = SPX_KC / 16;
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    // transposed-read address of this lane: row 8*(g>>1)+q of the k-step, 4 pixels at 32*pg+16*(g&1)+4*pp
    const char* xb = xs + ((8 * (g >> 1) + q) * SPX_XROW + 32 * pg + 16 * (g & 1) + 4 * pp) * 2;
    const char* ab = as + lane * 16 + pb0 * (NKS * 1024);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const s16x4 t0 = lds_tr_read(xb + ks * (16 * SPX_XROW * 2));
        const s16x4 t1 = lds_tr_read(xb + ks * (16 * SPX_XROW * 2) + 4 * SPX_XROW * 2);
        const bf16x4 b0 = __builtin_bit_cast(bf16x4, t0);
        const bf16x4 b1 = __builtin_bit_cast(bf16x4, t1);
        const bf16x8 bfrag = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bf16x2 p2v;
            p2v[0] = bfrag[2 * e];
            p2v[1] = bfrag[2 * e + 1];
            x2part = __builtin_amdgcn_fdot2_f32_bf16(p2v, p2v, x2part, false);
        }
        // (issuing every fragment read of the chunk up front was measured: no gain, +48 VGPRs)
#pragma unroll
        for (int pb = 0; pb < NH; ++pb) {
            const bf16x8 afrag = *(const bf16x8*)(ab + (pb * NKS + ks) * 1024);
            acc[pb] = mfma_bf16(afrag, bfrag, acc[pb]);
        }
    }
}

// The same chunk with EVERY fragment read issued before the first MFMA (28 more registers for NH = 3): for a workgroup that
// is alone on its CU (the fused persistent backward).  Under register pressure hipcc otherwise emits read -> wait -> MFMA
// per fragment, i.e. one exposed LDS round trip per MFMA; with two co-resident workgroups the other one fills those gaps
// (measured there: no gain), a lone workgroup stalls on every one of them.
template <int NPB, int NH>
__device__ __forceinline__ void spx_compute_chunk_batched(f32x16 (&acc)[NH], float& x2part, const char* xs, const char* as,
                                                          int lane, int pg, int pb0) {
    constexpr int NKS = SPX_KC / 16;
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const char* xb = xs + ((8 * (g >> 1) + q) * SPX_XROW + 32 * pg + 16 * (g & 1) + 4 * pp) * 2;
    const char* ab = as + lane * 16 + pb0 * (NKS * 1024);
    s16x4 t[NKS][2];
    bf16x8 af[NKS][NH];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        t[ks][0] = lds_tr_read(xb + ks * (16 * SPX_XROW * 2));
        t[ks][1] = lds_tr_read(xb + ks * (16 * SPX_XROW * 2) + 4 * SPX_XROW * 2);
#pragma unroll
        for (int pb = 0; pb < NH; ++pb) af[ks][pb] = *(const bf16x8*)(ab + (pb * NKS + ks) * 1024);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const bf16x4 b0 = __builtin_bit_cast(bf16x4, t[ks][0]);
        const bf16x4 b1 = __builtin_bit_cast(bf16x4, t[ks][1]);
        const bf16x8 bfrag = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bf16x2 p2v;
            p2v[0] = bfrag[2 * e];
            p2v[1] = bfrag[2 * e + 1];
            x2part = __builtin_amdgcn_fdot2_f32_bf16(p2v, p2v, x2part, false);
        }
#pragma unroll
        for (int pb = 0; pb < NH; ++pb) acc[pb] = mfma_bf16(af[ks][pb], bfrag, acc[pb]);
    }
}

// X stager: the registers of one 32-channel chunk of the tile (XPASS passes of NT/16 rows x 8 px per thread;
// NT = threads of the workgroup, 256 or 512)
// VM: 0 = element-wise loads, 1 = 16-B vector loads (every piece wholly inside or outside the image), 2 = vector loads
// with the in-register fix for the piece that straddles a ragged image end
template <bool XF32, int VM, int NT = 256>
struct SpxXStager {
    static constexpr bool VEC = VM != 0, RAG = VM == 2;
    static constexpr int RPP = NT / 16;           // rows per pass
    static constexpr int XPASS = SPX_KC / RPP;
    static constexpr int ESZ = XF32 ? 4 : 2;
    u32x4 xr[XPASS][XF32 ? 2 : 1];

    __device__ __forceinline__ static SpxTileCtx make_ctx(const void* x_img, int hw, int px0, int tid) {
        SpxTileCtx t;
        t.x_img = (const char*)x_img;
        t.hw = (uint32_t)hw;
        t.px = px0 + (tid & 15) * 8;
        t.row0 = tid >> 4;
        const uint32_t off = ((uint32_t)t.row0 * (uint32_t)hw + (uint32_t)t.px) * ESZ;
        // vector path: a piece wholly inside the image is loaded as it lies (any byte alignment: gfx950 serves misaligned
        // 16-B buffer accesses), one wholly outside is dropped, and the one that straddles the image end (H*W not a multiple
        // of 8) comes from a window moved back to END at the image end (hw >= 8 is the launcher's condition for this path)
        if constexpr (RAG) {
            const int over = t.px + 8 - hw;                  // elements of the piece past the image
            t.rot_bits = (over > 0 && over < 8) ? (uint32_t)over * (ESZ * 8) : 0u;
            t.ragged = px0 + SPX_TILE_PX > hw;
            t.x_voff = over <= 0 ? off : (over < 8 ? off - (uint32_t)over * ESZ : SPX_OOB);
        } else {
            t.rot_bits = 0u;
            t.ragged = false;
            t.x_voff = (!VEC || t.px + 8 <= hw) ? off : SPX_OOB;
        }
        return t;
    }

    // 32 channels from ch_first, ch_left of them real (<= 0: a padding step, which stages zeros).  No branches.
    __device__ __forceinline__ void load(const SpxTileCtx& t, int ch_first, int ch_left) {
        // the resource is re-based per step so that every offset stays far below the 2 GiB predication limit
        const spx_rsrc xr_ = make_rsrc_pred(t.x_img + (size_t)ch_first * t.hw * ESZ);
#pragma unroll
        for (int i = 0; i < XPASS; ++i) {
            const uint32_t soff = (uint32_t)(RPP * i) * t.hw * ESZ;
            const bool row_ok = t.row0 + RPP * i < ch_left;
            if (VEC) {
                const uint32_t vo = row_ok ? t.x_voff : SPX_OOB;
                xr[i][0] = buf_load_b128(xr_, vo, soff);
                if (XF32) xr[i][1] = buf_load_b128(xr_, vo == SPX_OOB ? SPX_OOB : vo + 16u, soff);
            } else if (XF32) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool ok = row_ok && (uint32_t)(t.px + e) < t.hw;
                    xr[i][e >> 2][e & 3] = __builtin_amdgcn_raw_buffer_load_b32(xr_, ok ? t.x_voff + 4 * e : SPX_OOB, soff, 0);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t lo = buf_load_u16(xr_, (row_ok && (uint32_t)(t.px + 2 * e) < t.hw) ? t.x_voff + 4 * e : SPX_OOB, soff);
                    const uint32_t hi = buf_load_u16(xr_, (row_ok && (uint32_t)(t.px + 2 * e + 1) < t.hw) ? t.x_voff + 4 * e + 2 : SPX_OOB, soff);
                    xr[i][0][e] = lo | (hi << 16);
                }
            }
        }
    }

    // the straddling piece of a ragged tile: shift the moved-back window into place (bf16: one 16-B register; fp32: the
    // 32-B piece is a 256-bit value - shift the pair)
    __device__ __forceinline__ void fix_ragged(const SpxTileCtx& t) {
        if (!RAG || !t.ragged) return;                     // tile-uniform
#pragma unroll
        for (int i = 0; i < XPASS; ++i) {
            if (XF32) spx_shr256(xr[i][0], xr[i][XF32 ? 1 : 0], t.rot_bits);
            else xr[i][0] = spx_shr128(xr[i][0], t.rot_bits);
        }
    }

    __device__ __forceinline__ void write(char* xs, int tid) {
        const int piece = tid & 15, row0 = tid >> 4;
#pragma unroll
        for (int i = 0; i < XPASS; ++i) {
            u32x4 v;
            if (XF32) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bf16x2 p;
                    p[0] = (__bf16)__uint_as_float(xr[i][e >> 1][(2 * e) & 3]);
                    p[1] = (__bf16)__uint_as_float(xr[i][e >> 1][(2 * e + 1) & 3]);
                    v[e] = __builtin_bit_cast(uint32_t, p);
                }
            } else {
                v = xr[i][0];
            }
            *(u32x4*)(xs + (row0 + RPP * i) * (SPX_XROW * 2) + piece * 16) = v;
        }
    }
};

// bank stager: the panel's A fragments of one chunk (NPB * 2 KiB, lane-linear), copied verbatim
template <int NPB, int NT = 256>
struct SpxAStager {
    static constexpr int ABYTES = NPB * (SPX_KC / 16) * 1024;
    static constexpr int PASS_BYTES = NT * 16;
    static constexpr int APASS = (ABYTES + PASS_BYTES - 1) / PASS_BYTES;
    static_assert(ABYTES % 4096 == 0, "panel height must be even");
    u32x4 ar[APASS];

    // the last pass may cover only part of the workgroup (12 KiB over 512 threads): per-thread predicate
    __device__ __forceinline__ static bool in_range(int i, int tid) { return (i + 1) * PASS_BYTES <= ABYTES || i * PASS_BYTES + tid * 16 < ABYTES; }

    __device__ __forceinline__ void load(const char* bank_chunk, bool real, int tid) {
        const spx_rsrc br_ = make_rsrc_pred(bank_chunk);
        const uint32_t bvo = real ? (uint32_t)(tid * 16) : SPX_OOB;
#pragma unroll
        for (int i = 0; i < APASS; ++i) ar[i] = buf_load_b128(br_, in_range(i, tid) ? bvo : SPX_OOB, (uint32_t)(i * PASS_BYTES));
    }
    __device__ __forceinline__ void write(char* as, int tid) {
#pragma unroll
        for (int i = 0; i < APASS; ++i)
            if (in_range(i, tid)) *(u32x4*)(as + i * PASS_BYTES + tid * 16) = ar[i];
    }
};

// Main-loop driver of one panel.  Global loads run ahead of the MFMAs through register rings: XR chunks of X
// (the HBM stream: XR = 4 keeps 32 KiB per workgroup in flight) and 2 chunks of bank fragments (L2 hits).
// Chunk c computes from LDS[c & 1]; at its end chunk c+1 is written to LDS[(c+1) & 1]; one barrier per chunk.
// The body is branch-free and unrolled by XR with static ring indices, so hipcc counts vmcnt exactly; chunk
// indices past the panel's last real chunk load and stage zeros.
// NT threads per workgroup; each wave accumulates NH of the panel's NPB blocks (NT = 256: NH = NPB, one wave per
// 32-pixel group; NT = 512: NH = NPB / 2, two waves per pixel group, each with half of the prototype blocks).
template <int NPB, bool XF32, int VM, int XR, int NT = 256, int NH = NPB, bool BATCH = false>
struct SpxPipeline {
    SpxXStager<XF32, VM, NT> xs[XR];
    SpxAStager<NPB, NT> as_[2];
    static constexpr int CHUNK_BYTES = NPB * 2 * 1024;
    static constexpr int STAGE = SPX_STAGE_X_BYTES + CHUNK_BYTES;

    template <int I>
    __device__ __forceinline__ void step(f32x16 (&acc)[NH], float& x2part, const SpxTileCtx& tc, char* smem,
                                         const char* bank0, int ch0, int Cs, int c, int lane, int wave, int tid) {
        // c = chunk index, I = c mod XR (static)
#ifdef SPX_DIAG_STAMPS
        const unsigned long long s0 = __builtin_amdgcn_s_memtime();
#endif
        xs[I % XR].load(tc, ch0 + (c + XR) * SPX_KC, Cs - (c + XR) * SPX_KC);
        as_[I % 2].load(bank0 + (size_t)(c + 2) * CHUNK_BYTES, Cs - (c + 2) * SPX_KC > 0, tid);
        __builtin_amdgcn_sched_barrier(0);   // the loads issue HERE, not below the MFMAs
#ifdef SPX_DIAG_STAMPS
        const unsigned long long s0b = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        dg_issue += s0b - s0;
#endif
        char* cur = smem + (I % 2) * STAGE;
        if constexpr (BATCH) spx_compute_chunk_batched<NPB, NH>(acc, x2part, cur, cur + SPX_STAGE_X_BYTES, lane, wave & 3, (wave >> 2) * NH);
        else spx_compute_chunk<NPB, NH>(acc, x2part, cur, cur + SPX_STAGE_X_BYTES, lane, wave & 3, (wave >> 2) * NH);
#ifdef SPX_DIAG_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long s1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
#endif
        char* dst = smem + ((I + 1) % 2) * STAGE;
        xs[(I + 1) % XR].fix_ragged(tc);
        xs[(I + 1) % XR].write(dst, tid);
        as_[(I + 1) % 2].write(dst + SPX_STAGE_X_BYTES, tid);
#ifdef SPX_DIAG_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long s2 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
#endif
        __syncthreads();
#ifdef SPX_DIAG_STAMPS
        const unsigned long long s3 = __builtin_amdgcn_s_memtime();
        dg_compute += s1 - s0; dg_write += s2 - s1; dg_barrier += s3 - s2;
#endif
    }
#ifdef SPX_DIAG_STAMPS
    unsigned long long dg_compute = 0, dg_write = 0, dg_barrier = 0, dg_issue = 0;
#endif

    // consts_issue / consts_commit stage the panel's epilogue constants (head fragments, |p|^2) into LDS: the
    // loads are issued behind the pipeline's prologue loads and committed after the first barrier, so they share
    // the fill latency instead of adding a serial round trip in front of it.
    // A panel = issue_prologue (global loads of the first chunks + the epilogue constants into registers) followed by
    // run_body.  Kernels with register room issue the NEXT panel's prologue before the current panel's epilogue, so
    // its pipeline fill hides behind the epilogue (multi-scale banks run 4+ panels per tile).
    template <typename F>
    __device__ __forceinline__ void issue_prologue(const SpxTileCtx& tc, const char* bank0, int ch0, int Cs, int tid,
                                                   F consts_issue) {
        xs[0].load(tc, ch0, Cs);
        as_[0].load(bank0, true, tid);
#pragma unroll
        for (int i = 1; i < XR; ++i) xs[i].load(tc, ch0 + i * SPX_KC, Cs - i * SPX_KC);
        as_[1].load(bank0 + CHUNK_BYTES, Cs - SPX_KC > 0, tid);
        consts_issue();      // loads only: they ride behind the pipeline's own prologue loads
    }
    template <typename G>
    __device__ __forceinline__ void run_body(f32x16 (&acc)[NH], float& x2part, const SpxTileCtx& tc, char* smem,
                                             const char* bank0, int ch0, int Cs, int lane, int wave, int tid,
                                             G consts_commit) {
        const int nchunks = (Cs + SPX_KC - 1) / SPX_KC;
        const int nrounds = (nchunks + XR - 1) / XR;       // chunks are processed XR at a time
        xs[0].fix_ragged(tc);
        xs[0].write(smem, tid);
        as_[0].write(smem + SPX_STAGE_X_BYTES, tid);
        __syncthreads();
        consts_commit();     // LDS writes of the constants; ordered before their first use by the loop's barriers
        for (int rnd = 0; rnd < nrounds; ++rnd) {
            const int c = rnd * XR;
            step<0>(acc, x2part, tc, smem, bank0, ch0, Cs, c, lane, wave, tid);
            if (XR > 1) step<1 % XR>(acc, x2part, tc, smem, bank0, ch0, Cs, c + 1, lane, wave, tid);
            if (XR > 2) step<2 % XR>(acc, x2part, tc, smem, bank0, ch0, Cs, c + 2, lane, wave, tid);
            if (XR > 2) step<3 % XR>(acc, x2part, tc, smem, bank0, ch0, Cs, c + 3, lane, wave, tid);
        }
    }
    template <typename F, typename G>
    __device__ __forceinline__ void run_panel(f32x16 (&acc)[NH], float& x2part, const SpxTileCtx& tc, char* smem,
                                              const char* bank0, int ch0, int Cs, int lane, int wave, int tid,
                                              F consts_issue, G consts_commit) {
        issue_prologue(tc, bank0, ch0, Cs, tid, consts_issue);
        run_body(acc, x2part, tc, smem, bank0, ch0, Cs, lane, wave, tid, consts_commit);
    }
};

// L2 warm-up of a panel's whole X tile: every thread touches 128-B lines of the tile's channel rows (one dword
// each, result unused), so the tile streams HBM -> L2 in one burst (2 KiB..64 KiB in flight per workgroup) and
// the staged chunk loads that follow hit L2.  The pipeline itself keeps only two 8-KiB chunks in flight per
// workgroup, too little to cover HBM latency at two workgroups per CU (measured: the X stream alone ran at
// 2.6 TB/s).
template <bool XF32>
__device__ __forceinline__ float spx_touch_tile(const SpxTileCtx& t, int ch_first, int nch, int px0, int tid) {
    constexpr int ESZ = XF32 ? 4 : 2;
    constexpr int LPR = 128 * ESZ / 128;            // 128-B lines per channel row of the tile
    const int lines = nch * LPR;
    uint32_t acc = 0;
    const spx_rsrc r = make_rsrc_pred(t.x_img + (size_t)ch_first * t.hw * ESZ);
    for (int l = tid; l < lines; l += 256) {
        const int row = l / LPR, part = l - row * LPR;
        const int px = px0 + part * (128 / ESZ);
        const uint32_t vo = ((uint32_t)px < t.hw) ? (uint32_t)px * ESZ & ~3u : SPX_OOB;
        acc ^= __builtin_amdgcn_raw_buffer_load_b32(r, vo, (uint32_t)row * t.hw * ESZ, 0);
    }
    return __uint_as_float(acc & 1u);   // <= 1.4e-45: keeps the loads alive without changing any result
}

// LDS bytes of one main-loop stage
__host__ __device__ constexpr int spx_stage_bytes(int npb) { return SPX_STAGE_X_BYTES + npb * (SPX_KC / 16) * 1024; }
