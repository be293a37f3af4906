// Stand-alone pixel-wise cross entropy over [M, K] logits (segmentation/model/loss.py:9-48): the arithmetic of the
// fused epilogue (spx_fwd_impl.h) / prologue (spx_bwd_impl.h) for heads those kernels do not carry (more than 160
// classes, the grouping tail) and for callers that hold logits of their own.
#include "spx_args.h"
#include "spx_common.h"

#define SPX_CE_THREADS 256

// one pixel per thread: logsumexp, argmax (lowest index on ties), label's logit; per-wave (loss sum, count) partials
__global__ __launch_bounds__(SPX_CE_THREADS) void spx_ce_fwd_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                                  long long M, int K, float* __restrict__ lse_out,
                                                                  int32_t* __restrict__ pred, float* __restrict__ partials) {
    const long long m_ = (long long)blockIdx.x * SPX_CE_THREADS + threadIdx.x;
    const bool in = m_ < M;
    const float* row = logits + (in ? m_ : 0) * K;
    const int lab = in ? labels[m_] : -1;
    const bool valid = in && (unsigned)lab < (unsigned)K;
    float mx = -3.0e38f;
    int best = 0x7fffffff;
    for (int k = 0; k < K; ++k) ce_best(row[k], k, mx, best);
    float ssum = 0.0f;
    for (int k = 0; k < K; ++k) ssum += ce_exp(row[k] - mx);
    const float lse = mx + ce_log(ssum);
    if (in) {
        lse_out[m_] = lse;
        if (pred) pred[m_] = best;
    }
    float lossv = valid ? lse - row[lab] : 0.0f, cnt = valid ? 1.0f : 0.0f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lossv += __shfl_xor(lossv, off);
        cnt += __shfl_xor(cnt, off);
    }
    if ((threadIdx.x & 63) == 0) {
        float* const pp = partials + ((size_t)blockIdx.x * (SPX_CE_THREADS / 64) + (threadIdx.x >> 6)) * 2;
        pp[0] = lossv;
        pp[1] = cnt;
    }
}

// mean loss -> loss[0], (count, loss sum) -> aux[0..1] from the kernels' (sum, count) partial pairs: one workgroup, a fixed summation order (thread-strided
// pairs, wave butterfly, waves in order) - the reduce + divide pair of torch launches as one; 0 / 0 = nan as torch's mean
__global__ __launch_bounds__(SPX_CE_THREADS) void spx_ce_finish_kernel(const float* __restrict__ partials, long long n, float* __restrict__ loss, float* __restrict__ aux) {
    __shared__ float red[2][SPX_CE_THREADS / 64];
    float s0 = 0.0f, s1 = 0.0f, c0 = 0.0f, c1 = 0.0f;
    long long i = threadIdx.x;
    for (; i + SPX_CE_THREADS < n; i += 2 * SPX_CE_THREADS) {
        const float2 a = ((const float2*)partials)[i], b = ((const float2*)partials)[i + SPX_CE_THREADS];
        s0 += a.x; c0 += a.y; s1 += b.x; c1 += b.y;
    }
    if (i < n) {
        const float2 a = ((const float2*)partials)[i];
        s0 += a.x; c0 += a.y;
    }
    float sv = s0 + s1, cv = c0 + c1;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        sv += __shfl_xor(sv, off);
        cv += __shfl_xor(cv, off);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = sv;
        red[1][threadIdx.x >> 6] = cv;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float st = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3], ct = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        loss[0] = st / ct;
        aux[0] = ct;
        aux[1] = st;
    }
}
hipError_t spx_launch_ce_finish(const float* partials, long long n, float* loss, float* aux, hipStream_t s) {
    hipLaunchKernelGGL(spx_ce_finish_kernel, dim3(1), dim3(SPX_CE_THREADS), 0, s, partials, n, loss, aux);
    return hipGetLastError();
}
// labels - 1 as int32 (loss.py:32: 0 = void becomes -1) from int64 / int32 labels: one launch
__global__ __launch_bounds__(SPX_CE_THREADS) void spx_shift_labels_kernel(const void* __restrict__ in, int is64, long long n, int32_t* __restrict__ out) {
    const long long i = (long long)blockIdx.x * SPX_CE_THREADS + threadIdx.x;
    if (i >= n) return;
    const long long v = is64 ? ((const long long*)in)[i] : (long long)((const int32_t*)in)[i];
    const long long w = v - 1;
    out[i] = (w < -2147483647LL || w > 2147483647LL) ? -1 : (int32_t)w;       // anything outside int32 is no class
}
hipError_t spx_launch_shift_labels(const void* in, int is64, long long n, int32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(spx_shift_labels_kernel, dim3((unsigned)((n + SPX_CE_THREADS - 1) / SPX_CE_THREADS)), dim3(SPX_CE_THREADS), 0, s, in, is64, n, out);
    return hipGetLastError();
}

// d_logits[m, k] = coef * (softmax - onehot) on the non-ignored pixels, 0 elsewhere
__global__ __launch_bounds__(SPX_CE_THREADS) void spx_ce_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ lse,
                                                                  const int32_t* __restrict__ labels, const float* __restrict__ coef,
                                                                  long long M, int K, float* __restrict__ d_logits) {
    const long long i = (long long)blockIdx.x * SPX_CE_THREADS + threadIdx.x;
    if (i >= M * K) return;
    const long long m_ = i / K;
    const int k = (int)(i - m_ * K);
    const int lab = labels[m_];
    const bool valid = (unsigned)lab < (unsigned)K;
    d_logits[i] = valid ? *coef * (ce_exp(logits[i] - lse[m_]) - (k == lab ? 1.0f : 0.0f)) : 0.0f;
}

// out[i] = sum over the scale groups of parts[g][i], in group order (deterministic): the logits of a scale-parallel forward
__global__ __launch_bounds__(SPX_CE_THREADS) void spx_sum_groups_kernel(const float* __restrict__ parts, size_t n, int groups,
                                                                      float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * SPX_CE_THREADS + threadIdx.x;
    if (i >= n) return;
    float s = parts[i];
    for (int g = 1; g < groups; ++g) s += parts[(size_t)g * n + i];
    out[i] = s;
}
// the same for MANY groups (the per-workgroup partials of spx_pixel_outer): 8 slices of the group range per output summed in
// parallel (slice j takes groups j, j + 8, ...), combined through LDS in slice order - a fixed order, so still deterministic
__global__ __launch_bounds__(SPX_CE_THREADS) void spx_sum_many_groups_kernel(const float* __restrict__ parts, size_t n, int groups,
                                                                           float* __restrict__ out) {
    __shared__ float red[8][32];
    const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const size_t i = (size_t)blockIdx.x * 32 + el;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (i < n) {
        int g = sl;
        for (; g + 24 < groups; g += 32) {          // four independent loads in flight per round
            const float v0 = parts[(size_t)g * n + i], v1 = parts[(size_t)(g + 8) * n + i];
            const float v2 = parts[(size_t)(g + 16) * n + i], v3 = parts[(size_t)(g + 24) * n + i];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        }
        for (; g < groups; g += 8) s0 += parts[(size_t)g * n + i];
    }
    red[sl][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && i < n) {
        float t = red[0][el];
#pragma unroll
        for (int j = 1; j < 8; ++j) t += red[j][el];
        out[i] = t;
    }
}
hipError_t spx_launch_sum_groups(const float* parts, size_t n, int groups, float* out, hipStream_t s) {
    if (groups > 16)
        hipLaunchKernelGGL(spx_sum_many_groups_kernel, dim3((unsigned)((n + 31) / 32)), dim3(SPX_CE_THREADS), 0, s, parts, n, groups, out);
    else
        hipLaunchKernelGGL(spx_sum_groups_kernel, dim3((unsigned)((n + SPX_CE_THREADS - 1) / SPX_CE_THREADS)), dim3(SPX_CE_THREADS), 0, s,
                           parts, n, groups, out);
    return hipGetLastError();
}

// Grouping-head tail as its own kernel (segmentation/model/model_multiscale_group.py:303-308): per pixel
//   units = sum over the scale groups of the partial unit products, g = exp(units), logits = W_g . g
// in plain fp32, for launches whose unit product ran scale-parallel (the tail needs the SUMMED units) or that carry the
// cross entropy (computed here, on the logits the kernel has just formed).  A workgroup takes 64 pixels: their [64][U]
// unit rows are ONE contiguous block per group - summed, exponentiated and staged in LDS with coalesced accesses - then
// thread (pixel, class quarter) forms up to 8 logits (W_g rows broadcast from LDS, the g row of a lane at an odd stride).
#define SPX_TAIL_PX 64
__global__ __launch_bounds__(SPX_CE_THREADS) void spx_group_tail_kernel(const float* __restrict__ parts, int groups, long long M,
                                                                      int U, const float* __restrict__ Wg, int K2,
                                                                      float* __restrict__ gact, float* __restrict__ logits,
                                                                      const int32_t* __restrict__ labels, float* __restrict__ lse_out,
                                                                      int32_t* __restrict__ pred, float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float tail_s[];
    const int US = U | 1;                                 // odd row stride: conflict-free column walks
    float* const wg_s = tail_s;                           // [U][32]
    float* const g_s = wg_s + 32 * U;                     // [64][US]
    float* const l_s = g_s + SPX_TAIL_PX * US;            // [64][33]
    float* const q_s = l_s + SPX_TAIL_PX * 33;            // [64][4] x 3: the class quarters' maxima, arg maxima, exp sums
    const long long m0 = (long long)blockIdx.x * SPX_TAIL_PX;
    const int npx = (int)((M - m0) < SPX_TAIL_PX ? (M - m0) : SPX_TAIL_PX);
    const size_t base = (size_t)m0 * U;
    // Staging: every round has all its loads in flight before the first use (a rolled load -> LDS loop pays one memory round
    // trip per element and scale group: 14 x 4 of them for 57 units and four groups, most of this kernel's time as measured).
    // W_g transposed and padded to 32 classes ([u][32], zeros past K2): a thread's 8 classes of one unit are two 16-B reads,
    // and the product loop carries no per-class condition (a guarded LDS read is its own wait: 9 serial LDS round trips per
    // unit, 20 of this kernel's 31 us on the Cityscapes crops as first written)
    const int nw = 32 * U;
    for (int i0 = threadIdx.x; i0 < nw; i0 += 8 * SPX_CE_THREADS) {
        float w[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int i = i0 + t * SPX_CE_THREADS, u = i >> 5, k = i & 31;
            w[t] = (i < nw && k < K2) ? Wg[k * U + u] : 0.0f;
        }
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (i0 + t * SPX_CE_THREADS < nw) wg_s[i0 + t * SPX_CE_THREADS] = w[t];
    }
    const int ntot = npx * U;
    const unsigned inv_u = ((1u << 24) + (unsigned)U - 1u) / (unsigned)U;
    for (int i0 = threadIdx.x; i0 < ntot; i0 += 8 * SPX_CE_THREADS) {
        float un[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) un[t] = 0.0f;
        for (int g0 = 0; g0 < groups; g0 += 4) {            // scale order: deterministic
            float v[4][8];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float* const pg = parts + (size_t)(g0 + g < groups ? g0 + g : 0) * M * U + base;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int i = i0 + t * SPX_CE_THREADS;
                    v[g][t] = (g0 + g < groups && i < ntot) ? pg[i] : 0.0f;
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int t = 0; t < 8; ++t) un[t] += v[g][t];
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int i = i0 + t * SPX_CE_THREADS;
            if (i < ntot) {
                const float gv = ce_exp(un[t]);
                if (gact) gact[base + i] = gv;
                const int p = (int)(((unsigned)i * inv_u) >> 24);      // = i / U: exact for i < 64 U, U <= 160
                g_s[p * US + (i - p * U)] = gv;
            }
        }
    }
    for (int i = ntot + threadIdx.x; i < SPX_TAIL_PX * U; i += SPX_CE_THREADS) {     // rows past the end read as 0
        const int p = i / U;
        g_s[p * US + (i - p * U)] = 0.0f;
    }
    __syncthreads();
    const int p = threadIdx.x & (SPX_TAIL_PX - 1), kq = threadIdx.x / SPX_TAIL_PX;     // pixel, class quarter (8 classes)
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
    const float4* const wq = (const float4*)(wg_s + kq * 8);
#pragma unroll 4
    for (int u = 0; u < U; ++u) {
        const float gv = g_s[p * US + u];
        const float4 w0 = wq[u * 8], w1 = wq[u * 8 + 1];
        acc[0] = __builtin_fmaf(w0.x, gv, acc[0]);
        acc[1] = __builtin_fmaf(w0.y, gv, acc[1]);
        acc[2] = __builtin_fmaf(w0.z, gv, acc[2]);
        acc[3] = __builtin_fmaf(w0.w, gv, acc[3]);
        acc[4] = __builtin_fmaf(w1.x, gv, acc[4]);
        acc[5] = __builtin_fmaf(w1.y, gv, acc[5]);
        acc[6] = __builtin_fmaf(w1.z, gv, acc[6]);
        acc[7] = __builtin_fmaf(w1.w, gv, acc[7]);
    }
    // the quarter's share of the cross entropy from its registers: maximum / arg maximum first ...
    float qm = -3.0e38f;
    int qb = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = kq * 8 + j;
        if (k < K2) {
            l_s[p * 33 + k] = acc[j];
            ce_best(acc[j], k, qm, qb);
        }
    }
    q_s[p * 4 + kq] = qm;
    q_s[SPX_TAIL_PX * 4 + p * 4 + kq] = __int_as_float(qb);
    __syncthreads();
    for (int i = threadIdx.x; i < npx * K2; i += SPX_CE_THREADS) {       // coalesced [px][K2] rows
        const int pp = i / K2, k = i - pp * K2;
        logits[(size_t)m0 * K2 + i] = l_s[pp * 33 + k];
    }
    if (!labels) return;
    // ... then the pixel's maximum (quarters in class order: the lowest index wins ties) and the quarter's exp sum
    float mx = -3.0e38f;
    int best = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < 4; ++q) ce_best(q_s[p * 4 + q], __float_as_int(q_s[SPX_TAIL_PX * 4 + p * 4 + q]), mx, best);
    float qs = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (kq * 8 + j < K2) qs += ce_exp(acc[j] - mx);
    q_s[2 * SPX_TAIL_PX * 4 + p * 4 + kq] = qs;
    __syncthreads();
    if (threadIdx.x >= SPX_TAIL_PX) return;                                 // wave 0: one pixel per lane
    const bool in = p < npx;
    const int lab = in ? labels[m0 + p] : -1;
    const bool valid = in && (unsigned)lab < (unsigned)K2;
    const float* const qq = q_s + 2 * SPX_TAIL_PX * 4 + p * 4;
    const float ssum = ((qq[0] + qq[1]) + qq[2]) + qq[3];
    const float lse = mx + ce_log(ssum);
    if (in) {
        lse_out[m0 + p] = lse;
        if (pred) pred[m0 + p] = best;
    }
    float lossv = valid ? lse - l_s[p * 33 + lab] : 0.0f, cnt = valid ? 1.0f : 0.0f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lossv += __shfl_xor(lossv, off);
        cnt += __shfl_xor(cnt, off);
    }
    if (threadIdx.x == 0) {
        partials[(size_t)blockIdx.x * 2] = lossv;
        partials[(size_t)blockIdx.x * 2 + 1] = cnt;
        // the caller's buffer holds spx_ce_partials_flat(M) pairs (a multiple of 4, at most 3 more than the grid): the
        // last workgroup clears the rest (no separate memset node in front of the kernel)
        if (blockIdx.x == gridDim.x - 1)
            for (unsigned e = gridDim.x; e < ((gridDim.x + 3u) & ~3u); ++e) partials[(size_t)e * 2] = partials[(size_t)e * 2 + 1] = 0.0f;
    }
}
size_t spx_group_tail_partials(long long M) { return (size_t)((M + SPX_TAIL_PX - 1) / SPX_TAIL_PX); }
hipError_t spx_launch_group_tail(const float* parts, int groups, long long M, int U, const float* Wg, int K2, float* gact,
                                 float* logits, const int32_t* labels, float* lse, int32_t* pred, float* partials, hipStream_t s) {
    const unsigned grid = (unsigned)((M + SPX_TAIL_PX - 1) / SPX_TAIL_PX);
    const size_t lds = ((size_t)32 * U + (size_t)SPX_TAIL_PX * (U | 1) + SPX_TAIL_PX * 33 + 3 * SPX_TAIL_PX * 4) * sizeof(float);
    hipLaunchKernelGGL(spx_group_tail_kernel, dim3(grid), dim3(SPX_CE_THREADS), lds, s, parts, groups, M, U,
                       Wg, K2, gact, logits, labels, lse, pred, partials);
    return hipGetLastError();
}

// out[i][j] = sum over the pixels of a[m][i] * b[m][j] for tall-skinny fp32 operands (n1 * n2 <= 8192): d W_g = d_logits^T . g
// of the grouping tail (19 x 57) and its relatives.  Every workgroup walks its pixel range in 32-pixel tiles staged in LDS
// (coalesced), every thread keeps its outputs in registers, the per-workgroup partials are summed in workgroup order by
// spx_sum_groups_kernel: deterministic, and no single-workgroup-shaped library GEMM (2.9 ms for 2 Mpx as a.t() @ b).
#define SPX_OUTER_MAX 8192
#define SPX_OUTER_EPT (SPX_OUTER_MAX / SPX_CE_THREADS)
__global__ __launch_bounds__(SPX_CE_THREADS) void spx_pixel_outer_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                       long long M, int n1, int n2, float* __restrict__ parts) {
    extern __shared__ float po_s[];
    float* const a_s = po_s;                 // [32][n1]
    float* const b_s = po_s + 32 * n1;       // [32][n2]
    const int nout = n1 * n2;
    float acc[SPX_OUTER_EPT];
    int oi[SPX_OUTER_EPT], oj[SPX_OUTER_EPT];
#pragma unroll
    for (int t = 0; t < SPX_OUTER_EPT; ++t) {
        const int e = threadIdx.x + SPX_CE_THREADS * t;
        acc[t] = 0.0f;
        oi[t] = e < nout ? e / n2 : 0;
        oj[t] = e < nout ? e - oi[t] * n2 : 0;
    }
    const long long per = ((M + gridDim.x - 1) / gridDim.x + 31) / 32 * 32;
    const long long m_begin = (long long)blockIdx.x * per, m_end = m_begin + per < M ? m_begin + per : M;
    for (long long m0 = m_begin; m0 < m_end; m0 += 32) {
        const int np = (int)(m_end - m0 < 32 ? m_end - m0 : 32);
        __syncthreads();
        for (int i = threadIdx.x; i < 32 * n1; i += SPX_CE_THREADS) a_s[i] = i < np * n1 ? a[(size_t)m0 * n1 + i] : 0.0f;
        for (int i = threadIdx.x; i < 32 * n2; i += SPX_CE_THREADS) b_s[i] = i < np * n2 ? b[(size_t)m0 * n2 + i] : 0.0f;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < SPX_OUTER_EPT; ++t) {
            if (threadIdx.x + SPX_CE_THREADS * t < nout) {
                float s_ = acc[t];
                for (int p = 0; p < 32; ++p) s_ = __builtin_fmaf(a_s[p * n1 + oi[t]], b_s[p * n2 + oj[t]], s_);
                acc[t] = s_;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < SPX_OUTER_EPT; ++t) {
        const int e = threadIdx.x + SPX_CE_THREADS * t;
        if (e < nout) parts[(size_t)blockIdx.x * nout + e] = acc[t];
    }
}
// The same product for n1 <= 32, n2 <= 64 (d W_g of the Cityscapes / Pascal grouping tails) on the fp32 matrix pipe:
// v_mfma_f32_32x32x2_f32 with the PIXEL as k - a wave's k-step is two pixels, its operands the two pixels' rows read straight
// from memory (adjacent rows: one contiguous piece per instruction), fp32 in, fp32 accumulate (no operand rounding).  The four
// waves of a workgroup take interleaved pixel pairs; their accumulators are summed through LDS in wave order.
typedef float spx_f32x16 __attribute__((ext_vector_type(16)));
#define SPX_OUTER_UNROLL 8
__global__ __launch_bounds__(SPX_CE_THREADS) void spx_pixel_outer_mfma_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                            long long M, int n1, int n2, float* __restrict__ parts) {
    __shared__ float red[4][2][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, kh = lane >> 5;
    const long long per = ((M + gridDim.x - 1) / gridDim.x + 31) / 32 * 32;
    const long long m_begin = (long long)blockIdx.x * per, m_end = m_begin + per < M ? m_begin + per : M;
    spx_f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
    const bool ia = i < n1, ib0 = i < n2, ib1 = 32 + i < n2;
    for (long long m = m_begin + 2 * wave + kh; m < m_end + kh; m += 8 * SPX_OUTER_UNROLL) {
        float av[SPX_OUTER_UNROLL], b0[SPX_OUTER_UNROLL], b1[SPX_OUTER_UNROLL];
#pragma unroll
        for (int u = 0; u < SPX_OUTER_UNROLL; ++u) {          // every load is issued before the first MFMA
            const long long px = m + 8 * u;
            const bool ok = px < m_end;
            av[u] = (ok && ia) ? a[px * n1 + i] : 0.0f;
            b0[u] = (ok && ib0) ? b[px * n2 + i] : 0.0f;
            b1[u] = (ok && ib1) ? b[px * n2 + 32 + i] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < SPX_OUTER_UNROLL; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], b0[u], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], b1[u], acc1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        red[wave][0][r][lane] = acc0[r];
        red[wave][1][r][lane] = acc1[r];
    }
    __syncthreads();
    // accumulator register r of lane l: row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31
    for (int e = threadIdx.x; e < 2 * 16 * 64; e += SPX_CE_THREADS) {
        const int blk = e >> 10, r = (e >> 6) & 15, l = e & 63;
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = 32 * blk + (l & 31);
        if (row < n1 && col < n2)
            parts[(size_t)blockIdx.x * n1 * n2 + row * n2 + col] =
                ((red[0][blk][r][l] + red[1][blk][r][l]) + red[2][blk][r][l]) + red[3][blk][r][l];
    }
}
int spx_pixel_outer_blocks(long long M) {
    long long n = (M + 127) / 128;          // a workgroup per 128 pixels until the chip is well covered
    return (int)(n < 1 ? 1 : (n > 1024 ? 1024 : n));
}
hipError_t spx_launch_pixel_outer(const float* a, const float* b, long long M, int n1, int n2, float* out, float* parts, hipStream_t s) {
    const int blocks = spx_pixel_outer_blocks(M);
    if (n1 <= 32 && n2 <= 64)
        hipLaunchKernelGGL(spx_pixel_outer_mfma_kernel, dim3((unsigned)blocks), dim3(SPX_CE_THREADS), 0, s, a, b, M, n1, n2, parts);
    else
        hipLaunchKernelGGL(spx_pixel_outer_kernel, dim3((unsigned)blocks), dim3(SPX_CE_THREADS), (size_t)32 * (n1 + n2) * sizeof(float), s,
                           a, b, M, n1, n2, parts);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return spx_launch_sum_groups(parts, (size_t)n1 * n2, blocks, out, s);
}

hipError_t spx_launch_ce_fwd(const float* logits, const int32_t* labels, long long M, int K, float* lse, int32_t* pred,
                             float* partials, hipStream_t s) {
    const unsigned grid = (unsigned)((M + SPX_CE_THREADS - 1) / SPX_CE_THREADS);
    hipLaunchKernelGGL(spx_ce_fwd_kernel, dim3(grid), dim3(SPX_CE_THREADS), 0, s, logits, labels, M, K, lse, pred, partials);
    return hipGetLastError();
}
hipError_t spx_launch_ce_bwd(const float* logits, const float* lse, const int32_t* labels, const float* coef, long long M, int K,
                             float* d_logits, hipStream_t s) {
    const long long n = M * K;
    const unsigned grid = (unsigned)((n + SPX_CE_THREADS - 1) / SPX_CE_THREADS);
    hipLaunchKernelGGL(spx_ce_bwd_kernel, dim3(grid), dim3(SPX_CE_THREADS), 0, s, logits, lse, labels, coef, M, K, d_logits);
    return hipGetLastError();
}

// g = exp(units) of the grouping head as a stand-alone elementwise kernel (compute_group on activations that did not come
// out of the fused forward, and the heads wider than the fused kernels; segmentation/model/model_multiscale_group.py:283-303):
// forward out = exp(x); backward (g, y given) out = g * y.
__global__ __launch_bounds__(SPX_CE_THREADS) void spx_exp_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                               const float* __restrict__ y, float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * SPX_CE_THREADS + threadIdx.x;
    if (i >= n) return;
    out[i] = g ? g[i] * y[i] : expf(x[i]);
}
hipError_t spx_launch_exp(const float* x, const float* g, const float* y, float* out, long long n, hipStream_t s) {
    hipLaunchKernelGGL(spx_exp_kernel, dim3((unsigned)((n + SPX_CE_THREADS - 1) / SPX_CE_THREADS)), dim3(SPX_CE_THREADS), 0, s, x, g, y, out, n);
    return hipGetLastError();
}

// Dense [U, P] form of the per-class group projections (segmentation/model/model_multiscale_group.py:249-269, :283-303: class
// j's weight [g_j, n_j] acts on the prototypes of class j): out[u][p] = W_j[row_local(u)][col_local(p)] where unit u and prototype
// p belong to the same block j, 0 elsewhere.  Every element is written (no zero fill, no concatenation of the weights first:
// torch's zeros + cat + index_put were three launches per step); the block pointers travel in the kernel arguments.
__global__ __launch_bounds__(SPX_CE_THREADS) void spx_group_dense_kernel(const SpxGroupDenseArgs a) {
    const int i = blockIdx.x * SPX_CE_THREADS + threadIdx.x;
    if (i >= a.U * a.P) return;
    const int u = i / a.P, p = i - u * a.P;
    const int jb = a.row_block[u];
    const bool ok = jb >= 0 && jb == a.col_block[p];
    const int j = ok ? jb : 0;
    const float v = a.ptrs[j][ok ? a.row_local[u] * a.ncols[j] + a.col_local[p] : 0];
    a.out[i] = ok ? v : 0.0f;
}
hipError_t spx_launch_group_dense(const SpxGroupDenseArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(spx_group_dense_kernel, dim3((unsigned)((a.U * a.P + SPX_CE_THREADS - 1) / SPX_CE_THREADS)), dim3(SPX_CE_THREADS), 0, s, a);
    return hipGetLastError();
}
// its adjoint: the weights' gradients, flat in block order, gathered from d_out at the (row, col) of every weight element
__global__ __launch_bounds__(SPX_CE_THREADS) void spx_group_dense_bwd_kernel(const float* __restrict__ d_out, const int32_t* __restrict__ rows,
                                                                           const int32_t* __restrict__ cols, long long n, int P,
                                                                           float* __restrict__ d_flat) {
    const long long i = (long long)blockIdx.x * SPX_CE_THREADS + threadIdx.x;
    if (i >= n) return;
    d_flat[i] = d_out[(size_t)rows[i] * P + cols[i]];
}
hipError_t spx_launch_group_dense_bwd(const float* d_out, const int32_t* rows, const int32_t* cols, long long n, int P, float* d_flat,
                                      hipStream_t s) {
    hipLaunchKernelGGL(spx_group_dense_bwd_kernel, dim3((unsigned)((n + SPX_CE_THREADS - 1) / SPX_CE_THREADS)), dim3(SPX_CE_THREADS), 0, s,
                       d_out, rows, cols, n, P, d_flat);
    return hipGetLastError();
}
