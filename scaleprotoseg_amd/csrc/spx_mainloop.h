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
    spx_rsrc xr;          // features of image b (buffer resource over channel 0, pixel 0)
    spx_rsrc xp;          // same base, range-checked (offsets >= 2 GiB are dropped): predicated element loads
    spx_rsrc br;          // packed bank fragments
    uint32_t x_voff;      // this thread's byte offset inside a 16-row pass: (row0 * HW + px) * esz
    uint32_t hw;          // pixels per image
    int px;               // first of this thread's 8 staged pixels
    int vec_ok;           // 16-B vector loads allowed (row starts 16-B aligned)
};

template <int NPB, bool XF32>
struct SpxStager {
    static constexpr int XPASS = 2;                 // kc <= 32 -> at most 2 row passes of 16
    static constexpr int APASS = (NPB * 2 + 3) / 4; // bank chunk <= NPB*2 KiB, 4 KiB per pass
    static constexpr int ESZ = XF32 ? 4 : 2;
    u32x4 xr[XPASS][XF32 ? 2 : 1];
    u32x4 ar[APASS];

    __device__ __forceinline__ static SpxTileCtx make_ctx(const void* x_img, const void* packed_bank, int hw, int px0,
                                                         int vec_ok, int tid) {
        SpxTileCtx t;
        t.xr = make_rsrc(x_img);
        t.xp = make_rsrc_pred(x_img);
        t.br = make_rsrc(packed_bank);
        t.hw = (uint32_t)hw;
        t.px = px0 + (tid & 15) * 8;
        t.x_voff = ((uint32_t)(tid >> 4) * (uint32_t)hw + (uint32_t)t.px) * ESZ;
        t.vec_ok = vec_ok;
        return t;
    }

    // issue the global loads of one step: X rows ch_first.., bank chunk at byte offset bank_off
    __device__ __forceinline__ void load(const SpxTileCtx& t, const spx_plan& pl, uint32_t bank_off, int ch_first,
                                         int tid) {
        const int npass = pl.kc >> 4;
        const bool full = t.vec_ok && (uint32_t)(t.px + 8) <= t.hw;
#pragma unroll
        for (int i = 0; i < XPASS; ++i) {
            if (i < npass) {
                const uint32_t soff = (uint32_t)(ch_first + 16 * i) * t.hw * ESZ;
                if (full) {
                    xr[i][0] = buf_load_b128(t.xr, t.x_voff, soff);
                    if (XF32) xr[i][1] = buf_load_b128(t.xr, t.x_voff + 16, soff);
                } else if (XF32) {
                    // unaligned / tail pieces: element loads, predicated by an out-of-range offset (no branches)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const bool ok = (uint32_t)(t.px + e) < t.hw;
                        xr[i][e >> 2][e & 3] = __builtin_amdgcn_raw_buffer_load_b32(t.xp, ok ? t.x_voff + 4 * e : SPX_OOB, soff, 0);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t lo = buf_load_u16(t.xp, ((uint32_t)(t.px + 2 * e) < t.hw) ? t.x_voff + 4 * e : SPX_OOB, soff);
                        const uint32_t hi = buf_load_u16(t.xp, ((uint32_t)(t.px + 2 * e + 1) < t.hw) ? t.x_voff + 4 * e + 2 : SPX_OOB, soff);
                        xr[i][0][e] = lo | (hi << 16);
                    }
                }
            }
        }
        const int abytes = pl.npb * (pl.kc >> 4) * 1024;
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const int off = i * 4096 + tid * 16;
            if (off < abytes) ar[i] = buf_load_b128(t.br, (uint32_t)off, bank_off);
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
