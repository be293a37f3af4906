// Evaluation-time full-resolution maps: bilinear upsample (align_corners = False) of a
// [N, C, h, w] map to [H, W] fused with the per-pixel argmin / argmax over C - the reference materialises the
// upsampled [C, H, W] tensor (1.6-1.9 GB per Cityscapes image) and copies it to the host first
// (segmentation/eval_valid_multiscale.py:229-234, :375-383).  The source map (<= tens of MB) stays cache-resident:
// every output pixel reads its 4 neighbours of each channel; neighbouring outputs share them.
#include "spx_common.h"

// torch's area_pixel_compute_source_index for align_corners = False (no explicit scale factor): scale = in / out,
// src = scale * (dst + 0.5) - 0.5 clamped at 0; i0 = floor, i1 = min(i0 + 1, in - 1), lambda1 = src - i0.
// The arithmetic is the CPU kernel's OPERATION BY OPERATION, fused multiply-adds where the shipped x86 builds have them
// (aten/src/ATen/native/cpu/UpSampleKernel.cpp; restated and pinned bit for bit in oracle/ppnet_oracle.py::
// upsample_bilinear_restated): the values, and with them the arg{min,max} indices, are bit-identical to the reference's
// F.interpolate + min / max on the host (this library is compiled with -ffp-contract=off: every fma below is explicit).
__device__ __forceinline__ void src_index(int dst, float scale, int in_size, int& i0, int& i1, float& l0, float& l1) {
    float s = __builtin_fmaf(scale, (float)dst + 0.5f, -0.5f);
    s = s < 0.0f ? 0.0f : s;
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
    l0 = 1.0f - l1;
}

__global__ __launch_bounds__(256) void spx_upsample_argext_kernel(const float* __restrict__ src, int C, int h, int w,
                                                                  int H, int W, float sh, float sw, int take_max,
                                                                  int64_t* __restrict__ idx, float* __restrict__ val) {
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n = blockIdx.z;
    if (ox >= W || oy >= H) return;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    src_index(oy, sh, h, y0, y1, ly0, ly1);
    src_index(ox, sw, w, x0, x1, lx0, lx1);
    const float* base = src + (size_t)n * C * h * w;
    const int o00 = y0 * w + x0, o01 = y0 * w + x1, o10 = y1 * w + x0, o11 = y1 * w + x1;
    float best = 0.0f;
    int bi = 0;
    const int hw = h * w;
#pragma unroll 4
    for (int c = 0; c < C; ++c) {
        const float* p = base + (size_t)c * hw;
        // rows first (t0, t1), then the two rows blended; in each blend the first product is fused, the second rounded
        const float t0 = __builtin_fmaf(p[o00], lx0, p[o01] * lx1), t1 = __builtin_fmaf(p[o10], lx0, p[o11] * lx1);
        const float v = __builtin_fmaf(t0, ly0, t1 * ly1);
        const bool better = take_max ? (v > best) : (v < best);
        if (c == 0 || better) {      // strict comparison: ties keep the lowest channel index
            best = v;
            bi = c;
        }
    }
    const size_t o = ((size_t)n * H + oy) * W + ox;
    idx[o] = bi;
    if (val) val[o] = best;
}

hipError_t spx_launch_upsample_argext(const float* src, int N, int C, int h, int w, int H, int W, int take_max,
                                      int64_t* idx, float* val, hipStream_t s) {
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    dim3 grid((unsigned)((W + 63) / 64), (unsigned)((H + 3) / 4), (unsigned)N);
    hipLaunchKernelGGL(spx_upsample_argext_kernel, grid, dim3(256), 0, s, src, C, h, w, H, W, sh, sw, take_max, idx, val);
    return hipGetLastError();
}
