// Operand re-packing: fp32 parameters -> bf16 MFMA fragment order (one 1-KiB fragment = 64 lanes x 8 bf16,
// stored lane-linear so a wave fetches it with one coalesced 16-B-per-lane load and LDS copies are verbatim).
#include "spx_common.h"

// A-fragment element map of v_mfma_f32_32x32x16_bf16: lane l holds A[row l&31][k = 8*(l>>5) + j], j = 0..7.
// When the B operand is an accumulator tile (head / dX products) hardware k = 8h+j meets tile row
// 16*s2 + 8*(j>>2) + 4*h + (j&3)  (cdna guide §3 'An accumulator tile as the next MFMA's operand').
__device__ __forceinline__ int perm_row(int s2, int h, int j) { return 16 * s2 + 8 * (j >> 2) + 4 * h + (j & 3); }

// packed_bank [panel][chunk][pb][ks][lane][8], p2 [panel][npb*32], packed_bankT [panel][pb][s2][chb][lane][8] (fp16, holds -2 p)
__global__ void spx_pack_bank_kernel(const spx_plan pl, const float* __restrict__ bank, __bf16* __restrict__ pb_out,
                                     __bf16* __restrict__ pbT_out, float* __restrict__ p2_out) {
    const int Cs = pl.channels_per_scale;
    const int nks = 2, nchunks = (Cs + 31) / 32;
    const int nchb = (Cs + 31) / 32;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_fwd = pl.npanels * pl.npb * 32 * nchunks * 32 / 8;          // 8-element groups
    const int n_T = pl.npanels * pl.npb * 2 * nchb * 64;
    const int n_p2 = pl.npanels * pl.npb * 32;
    if (gid < n_fwd) {
        int t = gid;
        const int lane = t & 63; t >>= 6;
        const int ks = t % nks; t /= nks;
        const int pb = t % pl.npb; t /= pl.npb;
        const int chunk = t % nchunks; t /= nchunks;
        const int panel = t;
        const int row = pb * 32 + (lane & 31);
        const int c0 = chunk * 32 + ks * 16 + 8 * (lane >> 5);
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = 0.0f;
            if (row < pl.panel_np[panel] && c0 + j < Cs) f = bank[(size_t)(pl.panel_p0[panel] + row) * Cs + c0 + j];
            v[j] = (__bf16)f;
        }
        *(bf16x8*)(pb_out + (size_t)gid * 8) = v;
    }
    if (pbT_out && gid < n_T) {
        int t = gid;
        const int lane = t & 63; t >>= 6;
        const int chb = t % nchb; t /= nchb;
        const int s2 = t & 1; t >>= 1;
        const int pb = t % pl.npb; t /= pl.npb;
        const int panel = t;
        const int ch = chb * 32 + (lane & 31);
        // fp16: the dX product is an fp16 MFMA (G is a single fp16 plane); the bf16-rounded bank value is exact in fp16 within
        // |p| < 32768 (anything below 2^-24 flushes, larger values saturate)
        f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = pb * 32 + perm_row(s2, lane >> 5, j);
            float f = 0.0f;
            if (row < pl.panel_np[panel] && ch < Cs) f = (float)(__bf16)bank[(size_t)(pl.panel_p0[panel] + row) * Cs + ch];
            // the transposed image carries -2 p (exact: a power of two): the dX kernel then forms
            // dX = 2 rs x + (-2 P)^T.G with one packed fma per pixel pair instead of multiply, subtract and doubling
            f = -2.0f * f;
            f = f > 65504.0f ? 65504.0f : (f < -65504.0f ? -65504.0f : f);
            v[j] = (_Float16)f;
        }
        *(f16x8*)((_Float16*)pbT_out + (size_t)gid * 8) = v;
    }
    // |p|^2: one wave per padded prototype row (lane-strided channels, fixed-order butterfly sum)
    const int wid = gid >> 6, lane = gid & 63;
    if (wid < n_p2) {
        const int panel = wid / (pl.npb * 32), row = wid - panel * pl.npb * 32;
        float s = 0.0f;
        if (row < pl.panel_np[panel]) {
            const float* src = bank + (size_t)(pl.panel_p0[panel] + row) * Cs;
            for (int c = lane; c < Cs; c += 64) {
                const float f = (float)(__bf16)src[c];   // |p|^2 of the prototype the MFMA actually sees
                s = __builtin_fmaf(f, f, s);
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
        if (lane == 0) p2_out[wid] = s;
    }
}

// packed_head  [cb][panel][pb][s2][hi|lo][lane][8]      (A = W rows, k = permuted prototype rows)
// packed_headT [panel][pb][cstep][hi|lo][lane][8]        (A = W^T rows = prototypes, k = classes)
__global__ void spx_pack_head_kernel(const spx_plan pl, const float* __restrict__ W, __bf16* __restrict__ ph,
                                     __bf16* __restrict__ phT) {
    const int P = pl.num_prototypes, K = pl.num_classes;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_h = pl.ncb * pl.npanels * pl.npb * 2 * 64;        // (hi,lo) pairs of 8-groups
    const int n_t = pl.npanels * pl.npb * (pl.ncb * 2) * 64;
    if (gid < n_h) {
        int t = gid;
        const int lane = t & 63; t >>= 6;
        const int s2 = t & 1; t >>= 1;
        const int pb = t % pl.npb; t /= pl.npb;
        const int panel = t % pl.npanels; t /= pl.npanels;
        const int cb = t;
        const int cls = cb * 32 + (lane & 31);
        bf16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = pb * 32 + perm_row(s2, lane >> 5, j);
            float f = 0.0f;
            if (cls < K && row < pl.panel_np[panel]) f = W[(size_t)cls * P + pl.panel_p0[panel] + row];
            __bf16 a, b;
            split_bf16(f, a, b);
            hi[j] = a;
            lo[j] = b;
        }
        const size_t base = ((size_t)(gid >> 6) * 2) * 512 + (size_t)lane * 8;
        *(bf16x8*)(ph + base) = hi;
        *(bf16x8*)(ph + base + 512) = lo;
    }
    if (phT && gid < n_t) {
        int t = gid;
        const int lane = t & 63; t >>= 6;
        const int cstep = t % (pl.ncb * 2); t /= (pl.ncb * 2);
        const int pb = t % pl.npb; t /= pl.npb;
        const int panel = t;
        const int row = pb * 32 + (lane & 31);
        bf16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cls = cstep * 16 + 8 * (lane >> 5) + j;
            float f = 0.0f;
            if (cls < K && row < pl.panel_np[panel]) f = W[(size_t)cls * P + pl.panel_p0[panel] + row];
            __bf16 a, b;
            split_bf16(f, a, b);
            hi[j] = a;
            lo[j] = b;
        }
        const size_t base = ((size_t)(gid >> 6) * 2) * 512 + (size_t)lane * 8;
        *(bf16x8*)(phT + base) = hi;
        *(bf16x8*)(phT + base + 512) = lo;
    }
}

// Grouping-head tail  logits = W_g . exp(units)  (model_multiscale_group.py:303-308): W_g [K2, U] as
//   tail   [cb][s2][hi|lo][lane][8]   A = W_g rows (classes), k = the units of accumulator tile cb in the permuted
//                                    order an accumulator tile presents as B operand (forward)
//   tailT  [ub][c][hi|lo][lane][8]    A = W_g^T rows (units of block ub), k = classes 16 c + 8 h + j (backward)
// and the head^T fragments with the UNIT (k) index in accumulator order, for the backward whose dUnits operand is
// built from accumulator tiles instead of being loaded from memory:
//   headT_units [panel][pb][cstep][hi|lo][lane][8]
__global__ void spx_pack_tail_kernel(const spx_plan pl, const float* __restrict__ Wg, int K2, __bf16* __restrict__ pt,
                                     __bf16* __restrict__ ptT) {
    const int U = pl.num_classes;                    // head rows = group units
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = pl.ncb * 2 * 64;
    if (gid >= n) return;
    const int lane = gid & 63, s2 = (gid >> 6) & 1, cb = gid >> 7;
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cls = lane & 31, u = cb * 32 + perm_row(s2, lane >> 5, j);
        const float f = (cls < K2 && u < U) ? Wg[(size_t)cls * U + u] : 0.0f;
        __bf16 a, c;
        split_bf16(f, a, c);
        hi[j] = a;
        lo[j] = c;
    }
    *(bf16x8*)(pt + ((size_t)(gid >> 6) * 2) * 512 + (size_t)lane * 8) = hi;
    *(bf16x8*)(pt + ((size_t)(gid >> 6) * 2 + 1) * 512 + (size_t)lane * 8) = lo;
    if (ptT) {
        const int c = s2, ub = cb;                   // same index space: (ub, c) pairs
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cls = c * 16 + 8 * (lane >> 5) + j, u = ub * 32 + (lane & 31);
            const float f = (cls < K2 && u < U) ? Wg[(size_t)cls * U + u] : 0.0f;
            __bf16 a, d;
            split_bf16(f, a, d);
            hi[j] = a;
            lo[j] = d;
        }
        *(bf16x8*)(ptT + ((size_t)(gid >> 6) * 2) * 512 + (size_t)lane * 8) = hi;
        *(bf16x8*)(ptT + ((size_t)(gid >> 6) * 2 + 1) * 512 + (size_t)lane * 8) = lo;
    }
}

__global__ void spx_pack_headT_units_kernel(const spx_plan pl, const float* __restrict__ W, __bf16* __restrict__ phT) {
    const int P = pl.num_prototypes, K = pl.num_classes;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_t = pl.npanels * pl.npb * (pl.ncb * 2) * 64;
    if (gid >= n_t) return;
    int t = gid;
    const int lane = t & 63; t >>= 6;
    const int cstep = t % (pl.ncb * 2); t /= (pl.ncb * 2);
    const int pb = t % pl.npb; t /= pl.npb;
    const int panel = t;
    const int row = pb * 32 + (lane & 31);
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cls = (cstep >> 1) * 32 + perm_row(cstep & 1, lane >> 5, j);
        float f = 0.0f;
        if (cls < K && row < pl.panel_np[panel]) f = W[(size_t)cls * P + pl.panel_p0[panel] + row];
        __bf16 a, c;
        split_bf16(f, a, c);
        hi[j] = a;
        lo[j] = c;
    }
    const size_t base = ((size_t)(gid >> 6) * 2) * 512 + (size_t)lane * 8;
    *(bf16x8*)(phT + base) = hi;
    *(bf16x8*)(phT + base + 512) = lo;
}

hipError_t spx_launch_pack_tail(const spx_plan& pl, const float* Wg, int K2, void* pt, void* ptT, hipStream_t s) {
    const int n = pl.ncb * 2 * 64;
    hipLaunchKernelGGL(spx_pack_tail_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pl, Wg, K2, (__bf16*)pt, (__bf16*)ptT);
    return hipGetLastError();
}
hipError_t spx_launch_pack_headT_units(const spx_plan& pl, const float* W, void* phT, hipStream_t s) {
    const int n = pl.npanels * pl.npb * (pl.ncb * 2) * 64;
    hipLaunchKernelGGL(spx_pack_headT_units_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pl, W, (__bf16*)phT);
    return hipGetLastError();
}

hipError_t spx_launch_pack_bank(const spx_plan& pl, const float* bank, void* pb, void* pbT, float* p2, hipStream_t s) {
    const int Cs = pl.channels_per_scale, nchb = (Cs + 31) / 32;
    int n = pl.npanels * pl.npb * 32 * (((Cs + 31) / 32) * 32) / 8;
    const int nT = pl.npanels * pl.npb * 2 * nchb * 64;
    if (nT > n) n = nT;
    const int np2 = pl.npanels * pl.npb * 32 * 64;     // one wave per row
    if (np2 > n) n = np2;
    hipLaunchKernelGGL(spx_pack_bank_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pl, bank, (__bf16*)pb,
                       (__bf16*)pbT, p2);
    return hipGetLastError();
}

hipError_t spx_launch_pack_head(const spx_plan& pl, const float* W, void* ph, void* phT, hipStream_t s) {
    const int n = pl.ncb * pl.npanels * pl.npb * 2 * 64;   // == n_t
    hipLaunchKernelGGL(spx_pack_head_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pl, W, (__bf16*)ph,
                       (__bf16*)phT);
    return hipGetLastError();
}
