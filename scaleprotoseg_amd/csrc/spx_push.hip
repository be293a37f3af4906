// Prototype-push reductions (segmentation/push_multiscale_optimization.py:74-91, :135-137).
// HBM-bound streaming minima: coalesced reads of one distance row, (value,index) packed into one
// 64-bit key so that a lexicographic minimum is a plain integer minimum — wavefront shuffles inside a
// wave, LDS across the 4 waves, one atomicMin per workgroup (integer min is order-independent, so the
// result is deterministic).
#include "spx_common.h"

#define SPX_PUSH_CHUNK 8192   // pixels per workgroup

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, m);
    hi = __shfl_xor(hi, m);
    return ((uint64_t)hi << 32) | lo;
}

__global__ __launch_bounds__(256) void spx_push_argmin_kernel(const float* __restrict__ dist,
                                                              const int32_t* __restrict__ labels,
                                                              const float* __restrict__ ident, int P, int K, int HW,
                                                              int void_class, float max_dist,
                                                              unsigned long long* __restrict__ scratch) {
    __shared__ float s_ident[160];
    __shared__ unsigned long long s_min[4];
    const int p = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x;
    for (int k = tid; k < K; k += 256) s_ident[k] = ident[(size_t)p * K + k];
    __syncthreads();
    const float* row = dist + ((size_t)b * P + p) * HW;
    const int32_t* lab = labels + (size_t)b * HW;
    const int begin = blockIdx.x * SPX_PUSH_CHUNK;
    const int end = min(begin + SPX_PUSH_CHUNK, HW);
    unsigned long long best = ~0ull;
    for (int i = begin + tid; i < end; i += 256) {
        int l = lab[i];
        float m = 0.0f;
        if (void_class >= 0) {
            if (l != void_class) {
                const int c = l < void_class ? l : l - 1;
                if (c >= 0 && c < K) m = s_ident[c];
            }
        } else if (l >= 0 && l < K) {
            m = s_ident[l];
        }
        // reference arithmetic, kept bit for bit: distances + max_dist * (1 - mask)   (:86-88)
        const float v = row[i] + max_dist * (1.0f - m);
        const unsigned long long key = ((unsigned long long)float_key(v) << 32) | (uint32_t)i;
        best = key < best ? key : best;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned long long o = shfl_xor_u64(best, m);
        best = o < best ? o : best;
    }
    if ((tid & 63) == 0) s_min[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        unsigned long long v = s_min[0];
        for (int w = 1; w < 4; ++w) v = s_min[w] < v ? s_min[w] : v;
        atomicMin(scratch + (size_t)b * P + p, v);
    }
}

__global__ void spx_push_finalize_kernel(const unsigned long long* __restrict__ scratch, int n,
                                         int64_t* __restrict__ idx, float* __restrict__ val) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const unsigned long long k = scratch[i];
        idx[i] = (int64_t)(uint32_t)k;
        val[i] = key_float((uint32_t)(k >> 32));
    }
}

// best[p] = argmin_n values[n][p], lowest n on ties (torch.argmin(dim=0) on concatenated minima, :135-137)
__global__ void spx_argmin_images_kernel(const float* __restrict__ values, int N, int P, int64_t* __restrict__ best) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    float bv = values[p];
    int bn = 0;
    for (int n = 1; n < N; ++n) {
        const float v = values[(size_t)n * P + p];
        if (v < bv) { bv = v; bn = n; }
    }
    best[p] = bn;
}

hipError_t spx_launch_push_argmin(const float* dist, const int32_t* labels, const float* ident, int B, int P, int K,
                                  int HW, int void_class, float max_dist, int64_t* idx, float* val,
                                  uint64_t* scratch, hipStream_t s) {
    hipError_t e = hipMemsetAsync(scratch, 0xFF, (size_t)B * P * sizeof(uint64_t), s);
    if (e != hipSuccess) return e;
    dim3 grid((HW + SPX_PUSH_CHUNK - 1) / SPX_PUSH_CHUNK, P, B);
    hipLaunchKernelGGL(spx_push_argmin_kernel, grid, dim3(256), 0, s, dist, labels, ident, P, K, HW, void_class,
                       max_dist, (unsigned long long*)scratch);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int n = B * P;
    hipLaunchKernelGGL(spx_push_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, s,
                       (const unsigned long long*)scratch, n, idx, val);
    return hipGetLastError();
}

hipError_t spx_launch_argmin_images(const float* values, int N, int P, int64_t* best, hipStream_t s) {
    hipLaunchKernelGGL(spx_argmin_images_kernel, dim3((P + 255) / 256), dim3(256), 0, s, values, N, P, best);
    return hipGetLastError();
}
