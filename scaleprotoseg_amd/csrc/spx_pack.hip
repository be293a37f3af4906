// Operand re-packing: fp32 parameters -> bf16 MFMA fragment order (one 1-KiB fragment = 64 lanes x 8 bf16,
// stored lane-linear so a wave fetches it with one coalesced 16-B-per-lane load and LDS copies are verbatim).
#include "spx_args.h"
#include "spx_common.h"

// A-fragment element map of v_mfma_f32_32x32x16_bf16: lane l holds A[row l&31][k = 8*(l>>5) + j], j = 0..7.
// When the B operand is an accumulator tile (head / dX products) hardware k = 8h+j meets tile row
// 16*s2 + 8*(j>>2) + 4*h + (j&3)  (cdna guide §3 'An accumulator tile as the next MFMA's operand').
__device__ __forceinline__ int perm_row(int s2, int h, int j) { return 16 * s2 + 8 * (j >> 2) + 4 * h + (j & 3); }

// Loads in these kernels are UNCONDITIONAL on a clamped index and the value is selected afterwards: a guarded load compiles to
// a branch with its own wait, eight serial memory round trips per thread in the unrolled element loops (the four pack kernels
// of a grouping step took 27 us that way, a third of it after this change).
__device__ __forceinline__ float ld_sel(const float* __restrict__ p, bool ok, size_t idx) {
    const float f = p[ok ? idx : 0];
    return ok ? f : 0.0f;
}

// packed_bank [panel][chunk][pb][ks][lane][8], p2 [panel][npb*32], packed_bankT [panel][pb][s2][chb][lane][8] (fp16, holds -2 p)
__device__ __forceinline__ void pack_bank_job(const spx_plan& pl, const float* __restrict__ bank, __bf16* __restrict__ pb_out,
                                              __bf16* __restrict__ pbT_out, float* __restrict__ p2_out, int gid) {
    const int Cs = pl.channels_per_scale;
    const int nks = 2, nchunks = (Cs + 31) / 32;
    const int nchb = (Cs + 31) / 32;
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
        const bool row_ok = row < pl.panel_np[panel];
        const size_t rbase = (size_t)(pl.panel_p0[panel] + row) * Cs + c0;
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)ld_sel(bank, row_ok && c0 + j < Cs, rbase + j);
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
        const int np = pl.panel_np[panel], p0 = pl.panel_p0[panel];
        // fp16: the dX product is an fp16 MFMA (G is a single fp16 plane); the bf16-rounded bank value is exact in fp16 within
        // |p| < 32768 (anything below 2^-24 flushes, larger values saturate)
        f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = pb * 32 + perm_row(s2, lane >> 5, j);
            float f = (float)(__bf16)ld_sel(bank, row < np && ch < Cs, (size_t)(p0 + row) * Cs + ch);
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
// packed_headT [panel][pb][cstep][hi|lo][lane][8]        (A = W^T rows = prototypes, k = classes; UNITS: the class (unit)
//              index in accumulator order, for the grouping backward whose dUnits operand is built from accumulator tiles)
template <bool UNITS>
__device__ __forceinline__ void pack_headT_job(const spx_plan& pl, const float* __restrict__ W, __bf16* __restrict__ phT, int gid) {
    const int P = pl.num_prototypes, K = pl.num_classes;
    const int n_t = pl.npanels * pl.npb * (pl.ncb * 2) * 64;
    if (gid >= n_t) return;
    int t = gid;
    const int lane = t & 63; t >>= 6;
    const int cstep = t % (pl.ncb * 2); t /= (pl.ncb * 2);
    const int pb = t % pl.npb; t /= pl.npb;
    const int panel = t;
    const int row = pb * 32 + (lane & 31);
    const bool row_ok = row < pl.panel_np[panel];
    const size_t col = (size_t)pl.panel_p0[panel] + row;
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cls = UNITS ? (cstep >> 1) * 32 + perm_row(cstep & 1, lane >> 5, j) : cstep * 16 + 8 * (lane >> 5) + j;
        const float f = ld_sel(W, cls < K && row_ok, (size_t)cls * P + col);
        __bf16 a, b;
        split_bf16(f, a, b);
        hi[j] = a;
        lo[j] = b;
    }
    const size_t base = ((size_t)(gid >> 6) * 2) * 512 + (size_t)lane * 8;
    *(bf16x8*)(phT + base) = hi;
    *(bf16x8*)(phT + base + 512) = lo;
}
__device__ __forceinline__ void pack_head_job(const spx_plan& pl, const float* __restrict__ W, __bf16* __restrict__ ph, int gid) {
    const int P = pl.num_prototypes, K = pl.num_classes;
    const int n_h = pl.ncb * pl.npanels * pl.npb * 2 * 64;        // (hi,lo) pairs of 8-groups
    if (gid >= n_h) return;
    int t = gid;
    const int lane = t & 63; t >>= 6;
    const int s2 = t & 1; t >>= 1;
    const int pb = t % pl.npb; t /= pl.npb;
    const int panel = t % pl.npanels; t /= pl.npanels;
    const int cb = t;
    const int cls = cb * 32 + (lane & 31);
    const int np = pl.panel_np[panel];
    const size_t rbase = (size_t)cls * P + pl.panel_p0[panel];
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = pb * 32 + perm_row(s2, lane >> 5, j);
        const float f = ld_sel(W, cls < K && row < np, rbase + row);
        __bf16 a, b;
        split_bf16(f, a, b);
        hi[j] = a;
        lo[j] = b;
    }
    const size_t base = ((size_t)(gid >> 6) * 2) * 512 + (size_t)lane * 8;
    *(bf16x8*)(ph + base) = hi;
    *(bf16x8*)(ph + base + 512) = lo;
}

// Grouping-head tail  logits = W_g . exp(units)  (model_multiscale_group.py:303-308): W_g [K2, U] as
//   tail   [cb][s2][hi|lo][lane][8]   A = W_g rows (classes), k = the units of accumulator tile cb in the permuted
//                                    order an accumulator tile presents as B operand (forward)
//   tailT  [ub][c][hi|lo][lane][8]    A = W_g^T rows (units of block ub), k = classes 16 c + 8 h + j (backward)
__device__ __forceinline__ void pack_tail_job(const spx_plan& pl, const float* __restrict__ Wg, int K2, __bf16* __restrict__ pt,
                                              __bf16* __restrict__ ptT, int gid) {
    const int U = pl.num_classes;                    // head rows = group units
    const int n = pl.ncb * 2 * 64;
    if (gid >= n) return;
    const int lane = gid & 63, s2 = (gid >> 6) & 1, cb = gid >> 7;
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cls = lane & 31, u = cb * 32 + perm_row(s2, lane >> 5, j);
        const float f = ld_sel(Wg, cls < K2 && u < U, (size_t)cls * U + u);
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
            const float f = ld_sel(Wg, cls < K2 && u < U, (size_t)cls * U + u);
            __bf16 a, d;
            split_bf16(f, a, d);
            hi[j] = a;
            lo[j] = d;
        }
        *(bf16x8*)(ptT + ((size_t)(gid >> 6) * 2) * 512 + (size_t)lane * 8) = hi;
        *(bf16x8*)(ptT + ((size_t)(gid >> 6) * 2 + 1) * 512 + (size_t)lane * 8) = lo;
    }
}

__global__ void spx_pack_bank_kernel(const spx_plan pl, const float* __restrict__ bank, __bf16* __restrict__ pb_out,
                                     __bf16* __restrict__ pbT_out, float* __restrict__ p2_out) {
    pack_bank_job(pl, bank, pb_out, pbT_out, p2_out, blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void spx_pack_head_kernel(const spx_plan pl, const float* __restrict__ W, __bf16* __restrict__ ph,
                                     __bf16* __restrict__ phT) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    pack_head_job(pl, W, ph, gid);
    if (phT) pack_headT_job<false>(pl, W, phT, gid);
}
__global__ void spx_pack_tail_kernel(const spx_plan pl, const float* __restrict__ Wg, int K2, __bf16* __restrict__ pt,
                                     __bf16* __restrict__ ptT) {
    pack_tail_job(pl, Wg, K2, pt, ptT, blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void spx_pack_headT_units_kernel(const spx_plan pl, const float* __restrict__ W, __bf16* __restrict__ phT) {
    pack_headT_job<true>(pl, W, phT, blockIdx.x * blockDim.x + threadIdx.x);
}

// Every operand of one forward (+ backward) in ONE launch: workgroup ranges [0, nb_bank) | head | head^T | tail.
__global__ void spx_pack_all_kernel(const SpxPackAllArgs a) {
    int blk = blockIdx.x;
    if (blk < a.nb_bank) {
        pack_bank_job(a.plan, a.bank, (__bf16*)a.packed_bank, (__bf16*)a.packed_bankT, a.p2, blk * 256 + threadIdx.x);
        return;
    }
    blk -= a.nb_bank;
    if (blk < a.nb_head) {
        pack_head_job(a.plan, a.W, (__bf16*)a.packed_head, blk * 256 + threadIdx.x);
        return;
    }
    blk -= a.nb_head;
    if (blk < a.nb_headT) {
        if (a.headT_units)
            pack_headT_job<true>(a.plan, a.W, (__bf16*)a.packed_headT, blk * 256 + threadIdx.x);
        else
            pack_headT_job<false>(a.plan, a.W, (__bf16*)a.packed_headT, blk * 256 + threadIdx.x);
        return;
    }
    blk -= a.nb_headT;
    pack_tail_job(a.plan, a.Wg, a.K2, (__bf16*)a.packed_tail, (__bf16*)a.packed_tailT, blk * 256 + threadIdx.x);
}

static int pack_bank_threads(const spx_plan& pl) {
    const int Cs = pl.channels_per_scale, nchb = (Cs + 31) / 32;
    int n = pl.npanels * pl.npb * 32 * (((Cs + 31) / 32) * 32) / 8;
    const int nT = pl.npanels * pl.npb * 2 * nchb * 64;
    if (nT > n) n = nT;
    const int np2 = pl.npanels * pl.npb * 32 * 64;     // one wave per row
    return np2 > n ? np2 : n;
}
hipError_t spx_launch_pack_all(SpxPackAllArgs a, hipStream_t s) {
    const spx_plan& pl = a.plan;
    a.nb_bank = (pack_bank_threads(pl) + 255) / 256;
    const int nh = pl.ncb * pl.npanels * pl.npb * 2 * 64;     // == the head^T count
    a.nb_head = a.packed_head ? (nh + 255) / 256 : 0;
    a.nb_headT = a.packed_headT ? (nh + 255) / 256 : 0;
    const int nb_tail = a.packed_tail ? (pl.ncb * 2 * 64 + 255) / 256 : 0;
    hipLaunchKernelGGL(spx_pack_all_kernel, dim3((unsigned)(a.nb_bank + a.nb_head + a.nb_headT + nb_tail)), dim3(256), 0, s, a);
    return hipGetLastError();
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
    hipLaunchKernelGGL(spx_pack_bank_kernel, dim3((pack_bank_threads(pl) + 255) / 256), dim3(256), 0, s, pl, bank, (__bf16*)pb,
                       (__bf16*)pbT, p2);
    return hipGetLastError();
}

hipError_t spx_launch_pack_head(const spx_plan& pl, const float* W, void* ph, void* phT, hipStream_t s) {
    const int n = pl.ncb * pl.npanels * pl.npb * 2 * 64;   // == n_t
    hipLaunchKernelGGL(spx_pack_head_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pl, W, (__bf16*)ph,
                       (__bf16*)phT);
    return hipGetLastError();
}
