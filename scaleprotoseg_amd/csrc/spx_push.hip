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
#pragma clang fp contract(off)       // rounded as the reference rounds it, also for a fractional mask
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

// Vector path (HW % 4 == 0): a workgroup takes SPX_PUSH_PX pixels x SPX_PUSH_PB prototypes.  A thread reads the labels
// of 4 consecutive pixels ONCE for the 8 prototypes, issues the 8 float4 row loads together, and takes the 8 masks of a
// pixel as two ds_read_b128 from a class-major copy of the identity block.  The running minimum is a (value, index)
// pair under `<` (a thread's indices only grow, so ties keep the lowest); the 64-bit key appears only in the final
// wave / workgroup / atomic reduction.  One prototype per workgroup with scalar loads ran at 2.0 TB/s.
#define SPX_PUSH_PB 8
#define SPX_PUSH_IT 4
#define SPX_PUSH_PX (256 * 4 * SPX_PUSH_IT)

__global__ __launch_bounds__(256) void spx_push_argmin_vec_kernel(const float* __restrict__ dist, const int32_t* __restrict__ labels,
                                                                  const float* __restrict__ ident, int P, int K, int HW,
                                                                  int void_class, float max_dist,
                                                                  unsigned long long* __restrict__ scratch) {
#pragma clang fp contract(off)       // distances + max_dist * (1 - mask), rounded as the reference rounds it (:86-88)
    extern __shared__ float s_mask[];                      // [K + 1][PB]: row K = "no class" (mask 0)
    __shared__ unsigned long long s_min[4][SPX_PUSH_PB];
    const int p0 = blockIdx.y * SPX_PUSH_PB, b = blockIdx.z, tid = threadIdx.x;
    for (int i = tid; i < (K + 1) * SPX_PUSH_PB; i += 256) {
        const int c = i / SPX_PUSH_PB, pp = i - c * SPX_PUSH_PB;
        s_mask[i] = (c < K && p0 + pp < P) ? ident[(size_t)(p0 + pp) * K + c] : 0.0f;
    }
    __syncthreads();
    const int32_t* lab = labels + (size_t)b * HW;
    const float* rows[SPX_PUSH_PB];
#pragma unroll
    for (int pp = 0; pp < SPX_PUSH_PB; ++pp) rows[pp] = dist + ((size_t)b * P + min(p0 + pp, P - 1)) * HW;
    float bv[SPX_PUSH_PB];
    int bi[SPX_PUSH_PB];
#pragma unroll
    for (int pp = 0; pp < SPX_PUSH_PB; ++pp) {
        bv[pp] = __builtin_inff();
        bi[pp] = -1;
    }
    const int begin = blockIdx.x * SPX_PUSH_PX;
#pragma unroll 1
    for (int it = 0; it < SPX_PUSH_IT; ++it) {
        const int i = begin + (it * 256 + tid) * 4;
        const bool in = i < HW;                            // HW % 4 == 0: all four pixels or none
        const int is = in ? i : 0;
        const int4 l4 = *(const int4*)(lab + is);
        float4 d4[SPX_PUSH_PB];
#pragma unroll
        for (int pp = 0; pp < SPX_PUSH_PB; ++pp) d4[pp] = *(const float4*)(rows[pp] + is);
        const int ls[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int c = ls[e];
            if (void_class >= 0) c = (c == void_class) ? -1 : (c < void_class ? c : c - 1);
            c = (c >= 0 && c < K) ? c : K;
            const float4 m0 = *(const float4*)(s_mask + c * SPX_PUSH_PB), m1 = *(const float4*)(s_mask + c * SPX_PUSH_PB + 4);
            const float m[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
            for (int pp = 0; pp < SPX_PUSH_PB; ++pp) {
                const float dv = e == 0 ? d4[pp].x : e == 1 ? d4[pp].y : e == 2 ? d4[pp].z : d4[pp].w;
                const float v = dv + max_dist * (1.0f - m[pp]);
                const bool take = in && (v < bv[pp] || bi[pp] < 0);
                bv[pp] = take ? v : bv[pp];
                bi[pp] = take ? i + e : bi[pp];
            }
        }
    }
#pragma unroll
    for (int pp = 0; pp < SPX_PUSH_PB; ++pp) {
        // -0.0 and +0.0 compare equal above; give them one key as well
        unsigned long long best = bi[pp] >= 0 ? (((unsigned long long)float_key(bv[pp] + 0.0f) << 32) | (uint32_t)bi[pp]) : ~0ull;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned long long o = shfl_xor_u64(best, m);
            best = o < best ? o : best;
        }
        if ((tid & 63) == 0) s_min[tid >> 6][pp] = best;
    }
    __syncthreads();
    if (tid < SPX_PUSH_PB && p0 + tid < P) {
        unsigned long long v = s_min[0][tid];
        for (int w = 1; w < 4; ++w) v = s_min[w][tid] < v ? s_min[w][tid] : v;
        atomicMin(scratch + (size_t)b * P + p0 + tid, v);
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
    if (HW % 4 == 0) {
        dim3 grid((HW + SPX_PUSH_PX - 1) / SPX_PUSH_PX, (P + SPX_PUSH_PB - 1) / SPX_PUSH_PB, B);
        hipLaunchKernelGGL(spx_push_argmin_vec_kernel, grid, dim3(256), (size_t)(K + 1) * SPX_PUSH_PB * 4, s, dist, labels, ident,
                           P, K, HW, void_class, max_dist, (unsigned long long*)scratch);
    } else {
        dim3 grid((HW + SPX_PUSH_CHUNK - 1) / SPX_PUSH_CHUNK, P, B);
        hipLaunchKernelGGL(spx_push_argmin_kernel, grid, dim3(256), 0, s, dist, labels, ident, P, K, HW, void_class,
                           max_dist, (unsigned long long*)scratch);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int n = B * P;
    hipLaunchKernelGGL(spx_push_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, s,
                       (const unsigned long long*)scratch, n, idx, val);
    return hipGetLastError();
}

hipError_t spx_launch_push_finalize(const uint64_t* scratch, int n, int64_t* idx, float* val, hipStream_t s) {
    hipLaunchKernelGGL(spx_push_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const unsigned long long*)scratch, n, idx, val);
    return hipGetLastError();
}

hipError_t spx_launch_argmin_images(const float* values, int N, int P, int64_t* best, hipStream_t s) {
    hipLaunchKernelGGL(spx_argmin_images_kernel, dim3((P + 255) / 256), dim3(256), 0, s, values, N, P, best);
    return hipGetLastError();
}
