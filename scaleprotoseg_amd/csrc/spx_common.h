// Shared device helpers for the gfx950 prototype-distance kernels.
// CDNA4 only: 64-lane wavefronts, v_mfma_f32_32x32x16_bf16, ds_read_b64_tr_b16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/spx_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

#define SPX_TILE_PX 128          // latent pixels per workgroup tile (4 waves x 32)
#define SPX_XROW 160             // bf16 elements per LDS row of the [k][pixel] X image (128 + 32 pad:
                                 // 320-B stride puts the 4 rows of a transposed read on disjoint banks)
#define SPX_LDS_LIMIT (160 * 1024)

// ---- buffer (SRSRC) addressing: one 32-bit per-lane offset + a wave-uniform SGPR offset per access.
// The epilogues touch ~100 distinct rows per tile; with flat 64-bit pointers every one of them costs two
// VGPRs of address and two VALU adds, which is what pushed these kernels into scratch.
typedef __amdgpu_buffer_rsrc_t spx_rsrc;
__device__ __forceinline__ spx_rsrc make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0xFFFFFFFFu, 0x00020000);
}
// Resource whose range check is used for predication: valid offsets stay below 2 GiB, a lane that must not
// access memory passes SPX_OOB as its offset and the hardware drops the access (no exec-mask branches).
#define SPX_OOB 0x80000000u
__device__ __forceinline__ spx_rsrc make_rsrc_pred(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x80000000u, 0x00020000);
}
__device__ __forceinline__ float buf_load_f32(spx_rsrc r, uint32_t voff, uint32_t soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store_f32(float v, spx_rsrc r, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}
__device__ __forceinline__ uint16_t buf_load_u16(spx_rsrc r, uint32_t voff, uint32_t soff) {
    return (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store_u16(uint16_t v, spx_rsrc r, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_buffer_store_b16((short)v, r, voff, soff, 0);
}
__device__ __forceinline__ u32x4 buf_load_b128(spx_rsrc r, uint32_t voff, uint32_t soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}
// 16-B store.  The wave-uniform offset is ADDED INTO the VGPR offset, soffset stays the literal 0: with an SGPR
// soffset hipcc (ROCm 7.2) assumes a >64-bit buffer store has no "VALU overwrites vdata right after issue" hazard
// and may re-use the first data register in the very next instruction; on gfx950 the store then intermittently
// writes that new value for its late lane groups (seen as small integers in dword 0 of lanes 12-15/28-31/...).
// With a literal soffset the hazard recognizer pads the two wait states itself.  SPX_OOB + soff stays out of range.
__device__ __forceinline__ void buf_store_b128(u32x4 v, spx_rsrc r, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff + soff, 0, 0);
}
__device__ __forceinline__ u32x2 buf_load_b64(spx_rsrc r, uint32_t voff, uint32_t soff) {
    return __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store_b64(u32x2 v, spx_rsrc r, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}

// Logical right shift of a 128-bit value (4 dwords, little endian) by `bits` in [0, 128), zeros shifted in.  Used on the
// ONE 16-byte piece per feature row that straddles the end of an image whose H*W is not a multiple of the piece: the
// piece is loaded from a window moved back to end exactly at the row's end (never reading past the tensor) and shifted
// into place here, which also zero-fills the elements past the image.  Wave-uniformly skipped in tiles without such a piece.
__device__ __forceinline__ u32x4 spx_shr128(u32x4 v, uint32_t bits) {
    // bit part with four funnel shifts, dword part with selects: no 64-bit temporaries (this runs at 256 VGPRs)
    const uint32_t b = bits & 31u, k = bits >> 5;
    const uint32_t t0 = __builtin_amdgcn_alignbit(v[1], v[0], b), t1 = __builtin_amdgcn_alignbit(v[2], v[1], b),
                   t2 = __builtin_amdgcn_alignbit(v[3], v[2], b), t3 = v[3] >> b;
    u32x4 o;
    o[0] = k == 0 ? t0 : k == 1 ? t1 : k == 2 ? t2 : k == 3 ? t3 : 0u;
    o[1] = k == 0 ? t1 : k == 1 ? t2 : k == 2 ? t3 : 0u;
    o[2] = k == 0 ? t2 : k == 1 ? t3 : 0u;
    o[3] = k == 0 ? t3 : 0u;
    return o;
}
__device__ __forceinline__ u32x4 spx_shl128(u32x4 v, uint32_t bits) {       // bits in (0, 128]
    const uint64_t lo = (uint64_t)v[0] | ((uint64_t)v[1] << 32), hi = (uint64_t)v[2] | ((uint64_t)v[3] << 32);
    const uint32_t s = bits & 63u;
    uint64_t olo, ohi;
    if (bits >= 128u) {
        olo = 0; ohi = 0;
    } else if (bits >= 64u) {
        olo = 0;
        ohi = lo << s;
    } else {
        olo = lo << s;
        ohi = s ? (hi << s) | (lo >> (64u - s)) : hi;
    }
    u32x4 o;
    o[0] = (uint32_t)olo; o[1] = (uint32_t)(olo >> 32); o[2] = (uint32_t)ohi; o[3] = (uint32_t)(ohi >> 32);
    return o;
}
// the same for a 256-bit value (lo, hi), bits in [0, 256)
__device__ __forceinline__ void spx_shr256(u32x4& lo, u32x4& hi, uint32_t bits) {
    if (bits >= 128u) {
        lo = spx_shr128(hi, bits - 128u);
        hi = u32x4{0u, 0u, 0u, 0u};
    } else if (bits) {
        const u32x4 a = spx_shr128(lo, bits), c = spx_shl128(hi, 128u - bits);
        lo = u32x4{a[0] | c[0], a[1] | c[1], a[2] | c[2], a[3] | c[3]};
        hi = spx_shr128(hi, bits);
    }
}

// A value the optimiser must treat as new at this point: lane-constant addresses and predicates derived from it are
// recomputed where they are used instead of being hoisted out of the tile loop (where hipcc then spills them around the
// whole loop and reloads them behind a full vmcnt wait: cdna guide, persistent-kernel pitfalls).
__device__ __forceinline__ int spx_opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// Row of a 32x32 MFMA accumulator held in register `reg` of lane half `h`
// (cdna guide §3: row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31).
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// Runtime-indexed access to a register array of accumulator tiles: a wave-uniform branch chain over static indices.
// (A dynamically indexed ext_vector array goes to scratch, and hipcc turns a plain select chain back into one; the
// empty asm statements keep the cases as separate basic blocks.  Rotating the array instead costs 16*(N-1) moves.)
#define SPX_TILE_CASE(K)                         \
    if (N > K && i == K) {                       \
        v = t[N > K ? K : 0];                    \
        asm volatile("; tile case " #K ::: );   \
    }
template <int N>
__device__ __forceinline__ f32x16 tile_get(const f32x16 (&t)[N], int i) {
    f32x16 v = t[0];
    SPX_TILE_CASE(1) SPX_TILE_CASE(2) SPX_TILE_CASE(3) SPX_TILE_CASE(4) SPX_TILE_CASE(5)
    return v;
}
#undef SPX_TILE_CASE
#define SPX_TILE_CASE(K)                         \
    if (N > K && i == K) {                       \
        t[N > K ? K : 0] = v;                    \
        asm volatile("; tile case " #K ::: );   \
    }
template <int N>
__device__ __forceinline__ void tile_set(f32x16 (&t)[N], int i, const f32x16& v) {
    SPX_TILE_CASE(0) SPX_TILE_CASE(1) SPX_TILE_CASE(2) SPX_TILE_CASE(3) SPX_TILE_CASE(4) SPX_TILE_CASE(5)
}
#undef SPX_TILE_CASE

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// Head-gradient scratch of the backward (kernel 1 -> kernels 2 / 3), opaque to the caller (spx_bwd_head_scratch_bytes):
//   one class block (K <= 32):  d_W is formed INSIDE kernel 1 (spx_bwd_impl.h, "d_W stage"): per (panel, tile, prototype block)
//       an fp32 partial [K][32 prototypes] of sum_px (a / ln 2) * (c1 dLogits) over the tile's 128 pixels - 128 K bytes per
//       block and tile instead of the 16-bit activation blob's 8 KiB;
//   wider heads (K > 32: the partial would be larger than the blob): the activations cross as 16-bit MFMA fragment blobs of
//       a / ln 2 - int16 codes scaled per (pixel, 32-prototype block) by a power of two (the block's largest |a| keeps 15 bits:
//       absolute error 2^-16 of that maximum; the exponent words follow the blobs) - and kernel 2 forms d_W from them.
// In both forms ONE float behind the data (the "head scale") is what the fixed-order reduction multiplies its sums by:
// ln 2 / c1 resp. ln 2 (c1 = the constant factor of act'(d) that kernel 1 folds into its dLogits operand).
#define SPX_ABLOB_I16_ONE 32767.0f
// bytes from the start of a blob scratch to the head scale: [blobs | block exponents (blobs / 8) | scale word (16 B)]
__host__ __device__ inline size_t spx_ablob_scale_offset(size_t blob_total) { return blob_total + blob_total / 8; }
// ... and behind that word the exponents of the G blob: one int32 per (panel, tile), G16 = G * 2^e
__host__ __device__ inline size_t spx_gexp_offset(size_t blob_total) { return blob_total + blob_total / 8 + 16; }
// bytes of the d_W tile partials of a plan with one class block: [panel][tile][block][K][32] fp32 (+ 16 B: the head scale)
__host__ __device__ inline size_t spx_dw_partial_bytes(int npanels, size_t ntiles, int npb, int K) {
    return (size_t)npanels * ntiles * npb * K * 128;
}

// fp32 -> (hi, lo) bf16 pair with hi + lo == x to ~2^-17 relative.
__device__ __forceinline__ void split_bf16(float x, __bf16& hi, __bf16& lo) {
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}

// The same split for TWO values at once, as packed words (element 0 in the low half): one packed convert for the high
// parts, one packed subtract, one packed convert for the residuals (2.5 VALU per value instead of 4 when every value
// is converted on its own and paired afterwards).
__device__ __forceinline__ uint32_t pack_bf16x2(f32x2 v) {
    bf16x2 p;
    p[0] = (__bf16)v[0];
    p[1] = (__bf16)v[1];
    return __builtin_bit_cast(uint32_t, p);
}
__device__ __forceinline__ f32x2 unpack_bf16x2(uint32_t w) {
    f32x2 v;
    v[0] = __uint_as_float(w << 16);
    v[1] = __uint_as_float(w & 0xffff0000u);
    return v;
}
__device__ __forceinline__ void split_bf16x2(f32x2 v, uint32_t& hi, uint32_t& lo) {
    hi = pack_bf16x2(v);
    asm("" : "+v"(hi));      // keeps the unpack below on the packed word (otherwise element 0 is converted a second time on its own)
    lo = pack_bf16x2(v - unpack_bf16x2(hi));
}
// sum of the two bf16 halves of a packed word, added to acc (v_dot2_f32_bf16 against ones)
__device__ __forceinline__ float add_bf16x2(uint32_t w, float acc) {
    bf16x2 one2;
    one2[0] = (__bf16)1.0f;
    one2[1] = (__bf16)1.0f;
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w), one2, acc, false);
}
// the same with round-toward-zero: ONE instruction (v_cvt_pkrtz_f16_f32), exact for values with <= 11 significant bits (every
// bf16 value in fp16's normal range), and a value beyond fp16's range saturates at 65504 instead of becoming inf
__device__ __forceinline__ uint32_t pack_f16x2_rtz(f32x2 v) {
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(v[0], v[1]));
}
__device__ __forceinline__ uint32_t pack_f16x2(f32x2 v) {
    f16x2 p;
    p[0] = (_Float16)v[0];
    p[1] = (_Float16)v[1];
    return __builtin_bit_cast(uint32_t, p);
}

// Transposed LDS read: 4 k-rows x (this lane's pixel) -> 4 bf16, see cdna guide T10.
__device__ __forceinline__ s16x4 lds_tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

// relu as ONE instruction.  fmaxf(x, 0) - and the med3 builtin, which the compiler rewrites to it - costs a second one (a
// canonicalising v_max x, x) wherever x cannot be proven free of signalling NaNs, e.g. after packed fp32 arithmetic.
__device__ __forceinline__ float relu_f32(float x) {
    float d;
    asm("v_max_f32 %0, 0, %1" : "=v"(d) : "v"(x));
    return d;
}

__device__ __forceinline__ float act_log(float d, float eps) {
    // log((d+1)/(d+eps)), segmentation/model/model_multiscale.py:326.  The ratio lies in [1, 1/eps], so the raw
    // v_rcp_f32 / v_log_f32 (1 ulp each, no denormal fix-up code) are exact enough: 6 VALU per element.
    return __builtin_amdgcn_logf((d + 1.0f) * __builtin_amdgcn_rcpf(d + eps)) * 0.69314718056f;
}
__device__ __forceinline__ float act_log_grad(float d, float eps) {
    // d/dd log((d+1)/(d+eps)) = 1/(d+1) - 1/(d+eps) = -(1-eps) / ((d+1)(d+eps))
    return -(1.0f - eps) * __builtin_amdgcn_rcpf((d + 1.0f) * (d + eps));
}

// cross entropy helpers (segmentation/model/loss.py:9-48): running maximum with the LOWEST class index on ties
// (torch.argmax's contract) and softmax pieces through the raw exp2 / log2 units
__device__ __forceinline__ void ce_best(float v, int cls, float& m, int& best) {
    if (v > m || (v == m && cls < best)) {
        m = v;
        best = cls;
    }
}
__device__ __forceinline__ float ce_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504089f); }
__device__ __forceinline__ float ce_log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718056f; }

// monotone float -> uint32 key (total order incl. negatives), for packed (value,index) minima
__device__ __forceinline__ uint32_t float_key(float v) {
    uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}
