// The x.p contraction shared by the forward and the pixel-side backward kernel.
//
// One workgroup = 4 waves = one tile of 128 latent pixels of one image.  Each wave owns
// 32 pixels (the MFMA column) against all <=192 prototypes of the current panel (the MFMA
// rows): D^T[proto x pixel] += Bank[proto x k] . X[k x pixel], 32x32x16 bf16 MFMA,
// fp32 accumulate.  The prototype-major orientation makes the accumulator's lane index the
// pixel index, so distance rows are written as 128-B pixel runs of the reference's
// [B,P,H,W] layout and the tile can feed the head MFMA as a B operand without leaving registers.
//
// Per step (one K-chunk of `kc` channels of one panel) the workgroup stages
//   - X[kc x 128 px]  -> LDS as bf16 [k][pixel] rows (320-B stride), read back with
//     ds_read_b64_tr_b16 so every lane receives the 8 k-values of its own pixel;
//   - the panel's bank chunk, already in MFMA A-fragment order in HBM/L2, -> LDS verbatim
//     (lane-linear image, conflict-free ds_read_b128).
// Staging is split (issue global loads for step+1, compute step, then write LDS), two LDS
// buffers, one barrier per step.
#pragma once
#include "spx_common.h"

struct SpxTileCtx {
    const char* x;        // features of image b, channel 0, pixel 0 (bytes)
    int hw;               // pixels per image
    int px0;              // first pixel of the tile
    int vec_ok;           // 16-B vector loads allowed (row starts 16-B aligned)
};

template <int NPB, bool XF32>
struct SpxStager {
    static constexpr int XPASS = 2;                 // kc <= 32 -> at most 2 row passes of 16
    static constexpr int APASS = (NPB * 2 + 3) / 4; // bank chunk <= NPB*2 KiB, 4 KiB per pass
    u32x4 xr[XPASS][XF32 ? 2 : 1];
    u32x4 ar[APASS];

    // issue the global loads of one step
    __device__ __forceinline__ void load(const SpxTileCtx& t, const spx_plan& pl, const char* bank_chunk,
                                         int ch_first, int tid) {
        const int piece = tid & 15, row0 = tid >> 4;
        const int px = t.px0 + piece * 8;
        const int npass = pl.kc >> 4;
#pragma unroll
        for (int i = 0; i < XPASS; ++i) {
            if (i < npass) {
                const size_t row = (size_t)(ch_first + row0 + 16 * i) * (size_t)t.hw;
                if (XF32) {
                    const float* src = (const float*)t.x + row + px;
                    if (t.vec_ok && px + 8 <= t.hw) {
                        xr[i][0] = *(const u32x4*)src;
                        xr[i][1] = *(const u32x4*)(src + 4);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float v = (px + e < t.hw) ? src[e] : 0.0f;
                            xr[i][e >> 2][e & 3] = __float_as_uint(v);
                        }
                    }
                } else {
                    const uint16_t* src = (const uint16_t*)t.x + row + px;
                    if (t.vec_ok && px + 8 <= t.hw) {
                        xr[i][0] = *(const u32x4*)src;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            uint32_t lo = (px + 2 * e < t.hw) ? src[2 * e] : 0u;
                            uint32_t hi = (px + 2 * e + 1 < t.hw) ? src[2 * e + 1] : 0u;
                            xr[i][0][e] = lo | (hi << 16);
                        }
                    }
                }
            }
        }
        const int abytes = pl.npb * (pl.kc >> 4) * 1024;
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const int off = i * 4096 + tid * 16;
            if (off < abytes) ar[i] = *(const u32x4*)(bank_chunk + off);
        }
    }

    // write the staged registers into LDS buffer `xs` (X image) / `as` (bank fragments)
    __device__ __forceinline__ void write(const spx_plan& pl, char* xs, char* as, int tid) {
        const int piece = tid & 15, row0 = tid >> 4;
        const int npass = pl.kc >> 4;
#pragma unroll
        for (int i = 0; i < XPASS; ++i) {
            if (i < npass) {
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
        }
        const int abytes = pl.npb * (pl.kc >> 4) * 1024;
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const int off = i * 4096 + tid * 16;
            if (off < abytes) *(u32x4*)(as + off) = ar[i];
        }
    }
};

// One staged K-chunk: acc[pb] += Bank_chunk[pb] . X_chunk, x2 += |x|^2 partial (this lane's k-half).
template <int NPB>
__device__ __forceinline__ void spx_compute_chunk(f32x16 (&acc)[NPB], float& x2part, const spx_plan& pl,
                                                  const char* xs, const char* as, int lane, int wave) {
    const int nks = pl.kc >> 4;
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    // transposed-read address of this lane: row 8*(g>>1)+q of the k-step, 4 pixels at 32*wave+16*(g&1)+4*pp
    const char* xb = xs + ((8 * (g >> 1) + q) * SPX_XROW + 32 * wave + 16 * (g & 1) + 4 * pp) * 2;
    const char* ab = as + lane * 16;
    for (int ks = 0; ks < nks; ++ks) {
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
            if (pb < pl.npb) {
                const bf16x8 afrag = *(const bf16x8*)(ab + (pb * nks + ks) * 1024);
                acc[pb] = mfma_bf16(afrag, bfrag, acc[pb]);
            }
        }
    }
}

// LDS bytes of one stage and of the whole main loop (two stages)
__host__ __device__ inline int spx_stage_bytes(int kc, int npb) {
    return kc * SPX_XROW * 2 + npb * (kc >> 4) * 1024;
}
