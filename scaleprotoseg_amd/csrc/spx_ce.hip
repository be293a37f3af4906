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
    float s0 = 0.0f, s1 = 0.0f;
    if (i < n) {
        int g = sl;
        for (; g + 8 < groups; g += 16) {
            s0 += parts[(size_t)g * n + i];
            s1 += parts[(size_t)(g + 8) * n + i];
        }
        if (g < groups) s0 += parts[(size_t)g * n + i];
    }
    red[sl][el] = s0 + s1;
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
    extern __shared__ float tail_s[];
    const int US = U | 1;                                 // odd row stride: conflict-free column walks
    float* const wg_s = tail_s;                           // [K2][U]
    float* const g_s = wg_s + K2 * U;                     // [64][US]
    float* const l_s = g_s + SPX_TAIL_PX * US;            // [64][33]
    const long long m0 = (long long)blockIdx.x * SPX_TAIL_PX;
    const int npx = (int)((M - m0) < SPX_TAIL_PX ? (M - m0) : SPX_TAIL_PX);
    for (int i = threadIdx.x; i < K2 * U; i += SPX_CE_THREADS) wg_s[i] = Wg[i];
    const size_t base = (size_t)m0 * U;
    for (int i = threadIdx.x; i < SPX_TAIL_PX * U; i += SPX_CE_THREADS) {
        const int p = i / U, u = i - p * U;
        float gv = 0.0f;
        if (p < npx) {
            float un = parts[base + i];
            for (int g = 1; g < groups; ++g) un += parts[(size_t)g * M * U + base + i];     // scale order: deterministic
            gv = ce_exp(un);
            if (gact) gact[base + i] = gv;
        }
        g_s[p * US + u] = gv;
    }
    __syncthreads();
    const int p = threadIdx.x & (SPX_TAIL_PX - 1), kq = threadIdx.x / SPX_TAIL_PX;     // pixel, class quarter (8 classes)
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
    for (int u = 0; u < U; ++u) {
        const float gv = g_s[p * US + u];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kq * 8 + j;
            if (k < K2) acc[j] = __builtin_fmaf(wg_s[k * U + u], gv, acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = kq * 8 + j;
        if (k < K2) l_s[p * 33 + k] = acc[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < npx * K2; i += SPX_CE_THREADS) {       // coalesced [px][K2] rows
        const int pp = i / K2, k = i - pp * K2;
        logits[(size_t)m0 * K2 + i] = l_s[pp * 33 + k];
    }
    if (!labels || threadIdx.x >= SPX_TAIL_PX) return;                      // wave 0: one pixel per lane
    const bool in = p < npx;
    const int lab = in ? labels[m0 + p] : -1;
    const bool valid = in && (unsigned)lab < (unsigned)K2;
    float mx = -3.0e38f;
    int best = 0x7fffffff;
    for (int k = 0; k < K2; ++k) ce_best(l_s[p * 33 + k], k, mx, best);
    float ssum = 0.0f;
    for (int k = 0; k < K2; ++k) ssum += ce_exp(l_s[p * 33 + k] - mx);
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
    }
}
size_t spx_group_tail_partials(long long M) { return (size_t)((M + SPX_TAIL_PX - 1) / SPX_TAIL_PX); }
hipError_t spx_launch_group_tail(const float* parts, int groups, long long M, int U, const float* Wg, int K2, float* gact,
                                 float* logits, const int32_t* labels, float* lse, int32_t* pred, float* partials, hipStream_t s) {
    const unsigned grid = (unsigned)((M + SPX_TAIL_PX - 1) / SPX_TAIL_PX);
    const size_t lds = ((size_t)K2 * U + (size_t)SPX_TAIL_PX * (U | 1) + SPX_TAIL_PX * 33) * sizeof(float);
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
int spx_pixel_outer_blocks(long long M) {
    long long n = (M + 127) / 128;          // a workgroup per 128 pixels (4 LDS tiles) until the chip is well covered
    return (int)(n < 1 ? 1 : (n > 1024 ? 1024 : n));
}
hipError_t spx_launch_pixel_outer(const float* a, const float* b, long long M, int n1, int n2, float* out, float* parts, hipStream_t s) {
    const int blocks = spx_pixel_outer_blocks(M);
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
