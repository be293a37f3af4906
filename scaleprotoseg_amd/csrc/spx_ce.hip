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
hipError_t spx_launch_sum_groups(const float* parts, size_t n, int groups, float* out, hipStream_t s) {
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
