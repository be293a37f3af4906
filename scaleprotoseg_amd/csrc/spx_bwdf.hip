// Fused persistent backward: instances + launcher + the fp16 (-2 bank)^T pack (see spx_bwdf_impl.h).
#include "spx_bwdf_impl.h"

// fp16 A-fragments of -2 bank^T for the fused backward's dX product, [chb][pb][s2][lane][8]: lane l holds
// row = channel chb * 32 + (l & 31), element j <-> prototype row pb * 32 + perm(s2, l >> 5, j) (the order in which
// an accumulator tile presents its rows as a B operand, see spx_pack.hip).  Exact for a bank that is
// bf16-representable in fp16's range (|p| < 32768, anything below 2^-24 flushes); larger values saturate.
__global__ void spx_pack_bankT16_kernel(const spx_plan pl, const float* __restrict__ bank, _Float16* __restrict__ out) {
    const int Cs = pl.channels_per_scale, nchb = (Cs + 31) / 32;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = nchb * pl.npb * 2 * 64;
    if (gid >= n) return;
    int t = gid;
    const int lane = t & 63; t >>= 6;
    const int s2 = t & 1; t >>= 1;
    const int pb = t % pl.npb; t /= pl.npb;
    const int chb = t;
    const int ch = chb * 32 + (lane & 31);
    f16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = pb * 32 + 16 * s2 + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
        float f = 0.0f;
        if (row < pl.panel_np[0] && ch < Cs) f = (float)(__bf16)bank[(size_t)(pl.panel_p0[0] + row) * Cs + ch];
        f = -2.0f * f;
        f = f > 65504.0f ? 65504.0f : (f < -65504.0f ? -65504.0f : f);
        v[j] = (_Float16)f;
    }
    *(f16x8*)(out + (size_t)gid * 8) = v;
}

hipError_t spx_launch_pack_bankT16(const spx_plan& pl, const float* bank, void* out, hipStream_t s) {
    const int n = ((pl.channels_per_scale + 31) / 32) * pl.npb * 2 * 64;
    hipLaunchKernelGGL(spx_pack_bankT16_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pl, bank, (_Float16*)out);
    return hipGetLastError();
}

bool spx_bwdf_supported(const spx_plan& pl) {
    return pl.npanels == 1 && pl.npb == 6 && pl.ncb == 1 && pl.num_scales == 1 && pl.channels_per_scale <= 256;
}

// workgroups of the persistent launch: one per compute unit, never more than tiles
int spx_bwdf_grid(int B, int HW) {
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256;
        ncu = v;
    }
    const long long tiles = (long long)B * ((HW + SPX_TILE_PX - 1) / SPX_TILE_PX);
    return (int)(tiles < ncu ? tiles : ncu);
}

hipError_t spx_launch_bwdf(const SpxBwdFArgs& a, int x_dtype, int grid, hipStream_t s) {
    if (a.labels) return spx_launch_bwdf_g<6, true>(a, x_dtype, grid, s);
    return spx_launch_bwdf_g<6, false>(a, x_dtype, grid, s);
}

// called by the C ABI (spx_api.hip), which does not see the kernel's argument block
int spx_bwdf_run(const spx_plan& pl, const void* x, int x_dtype, int B, int HW, const void* packed_bank, const void* packed_bankT16,
                 const float* p2, const void* packed_headT, const float* d_dist, const int32_t* labels, const uint32_t* proto_key,
                 int J, const float* d_cls, const float* d_logits, void* dx, void* a_out, float* workspace, float eps, int act_fn,
                 int grid, hipStream_t s, unsigned long long* dbg) {
    SpxBwdFArgs a;
    a.dbg = dbg;
    a.plan = pl;
    a.x = x;
    a.packed_bank = (const char*)packed_bank;
    a.packed_bankT16 = (const char*)packed_bankT16;
    a.p2 = p2;
    a.packed_headT = (const char*)packed_headT;
    a.d_dist = d_dist;
    a.d_logits = d_logits;
    a.labels = labels;
    a.proto_key = proto_key;
    a.d_cls_dist = d_cls;
    a.J = J;
    a.dx = dx;
    a.a_out = (uint16_t*)a_out;
    a.workspace = workspace;
    a.B = B;
    a.HW = HW;
    a.eps = eps;
    a.act_fn = act_fn;
    return spx_launch_bwdf(a, x_dtype, grid, s) == hipSuccess ? 0 : 1;
}
