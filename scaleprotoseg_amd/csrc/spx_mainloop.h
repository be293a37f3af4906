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
    uint32_t x_voff;      // this thread's byte offset inside a 16-row pass: (row0 * HW + px) * esz, or SPX_OOB
    uint32_t hw;          // pixels per image
    int px;               // first of this thread's 8 staged pixels
    int row0;             // this thread's row inside a 16-row pass
};

template <int NPB, bool XF32, bool VEC>
struct SpxStager {
    static constexpr int XPASS = SPX_KC / 16;
    static constexpr int ABYTES = NPB * (SPX_KC / 16) * 1024;
    static constexpr int APASS = ABYTES / 4096;
    static_assert(ABYTES % 4096 == 0, "panel height must be even");
    static constexpr int ESZ = XF32 ? 4 : 2;
    u32x4 xr[XPASS][XF32 ? 2 : 1];
    u32x4 ar[APASS];

    __device__ __forceinline__ static SpxTileCtx make_ctx(const void* x_img, int hw, int px0, int tid) {
        SpxTileCtx t;
        t.x_img = (const char*)x_img;
        t.hw = (uint32_t)hw;
        t.px = px0 + (tid & 15) * 8;
        t.row0 = tid >> 4;
        const uint32_t off = ((uint32_t)t.row0 * (uint32_t)hw + (uint32_t)t.px) * ESZ;
        // vector path: a piece is wholly inside or wholly outside the image (HW % 8 == 0)
        t.x_voff = (!VEC || t.px + 8 <= hw) ? off : SPX_OOB;
        return t;
    }

    // issue the global loads of one step: 32 channels from ch_first (ch_left of them real; <= 0 for a padding
    // step past the end of the panel, which then stages zeros), bank chunk at bank_chunk.  No branches.
    __device__ __forceinline__ void load(const SpxTileCtx& t, const char* bank_chunk, int ch_first, int ch_left, int tid) {
        // resources are re-based per step so that every offset stays far below the 2 GiB predication limit
        const spx_rsrc xr_ = make_rsrc_pred(t.x_img + (size_t)ch_first * t.hw * ESZ);
        const spx_rsrc br_ = make_rsrc_pred(bank_chunk);
        const uint32_t bvo = ch_left > 0 ? (uint32_t)(tid * 16) : SPX_OOB;
#pragma unroll
        for (int i = 0; i < XPASS; ++i) {
            const uint32_t soff = (uint32_t)(16 * i) * t.hw * ESZ;
            const bool row_ok = t.row0 + 16 * i < ch_left;
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
#pragma unroll
        for (int i = 0; i < APASS; ++i) ar[i] = buf_load_b128(br_, bvo, (uint32_t)(i * 4096));
    }

    // write the staged registers into LDS buffer `xs` (X image) / `as` (bank fragments)
    __device__ __forceinline__ void write(char* xs, char* as, int tid) {
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
            *(u32x4*)(xs + (row0 + 16 * i) * (SPX_XROW * 2) + piece * 16) = v;
        }
#pragma unroll
        for (int i = 0; i < APASS; ++i) *(u32x4*)(as + i * 4096 + tid * 16) = ar[i];
    }
};

// One staged K-chunk: acc[pb] += Bank_chunk[pb] . X_chunk, x2 += |x|^2 partial (this lane's k-half).
template <int NPB>
__device__ __forceinline__ void spx_compute_chunk(f32x16 (&acc)[NPB], float& x2part, const char* xs, const char* as,
                                                  int lane, int wave) {
    constexpr int NKS = SPX_KC / 16;
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    // transposed-read address of this lane: row 8*(g>>1)+q of the k-step, 4 pixels at 32*wave+16*(g&1)+4*pp
    const char* xb = xs + ((8 * (g >> 1) + q) * SPX_XROW + 32 * wave + 16 * (g & 1) + 4 * pp) * 2;
    const char* ab = as + lane * 16;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const s16x4 t0 = lds_tr_read(xb + ks * (16 * SPX_XROW * 2));
        const s16x4 t1 = lds_tr_read(xb + ks * (16 * SPX_XROW * 2) + 4 * SPX_XROW * 2);
        bf16x8 bfrag;
        {
            const bf16x4 b0 = __builtin_bit_cast(bf16x4, t0);
            const bf16x4 b1 = __builtin_bit_cast(bf16x4, t1);
            bfrag = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bf16x2 p2v;
            p2v[0] = bfrag[2 * e];
            p2v[1] = bfrag[2 * e + 1];
            x2part = __builtin_amdgcn_fdot2_f32_bf16(p2v, p2v, x2part, false);
        }
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) {
            const bf16x8 afrag = *(const bf16x8*)(ab + (pb * NKS + ks) * 1024);
            acc[pb] = mfma_bf16(afrag, bfrag, acc[pb]);
        }
    }
}

// LDS bytes of one main-loop stage
__host__ __device__ constexpr int spx_stage_bytes(int npb) { return SPX_STAGE_X_BYTES + npb * (SPX_KC / 16) * 1024; }
