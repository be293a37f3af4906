// Fused forward of the prototype-distance path:
//   distances = relu(|x|^2 - 2 x.p + |p|^2)                (model_multiscale.py:255-317)
//   activations = log((d+1)/(d+eps)) | -d                  (:324-330)
//   logits = activations . W^T                             (:243-244, :369-376)
// one launch, one pass over X, distances written once in the reference's [B,P,H,W] layout.
#include "spx_args.h"
#include "spx_mainloop.h"

template <int NPB, int NCB, bool XF32>
__global__ __launch_bounds__(256, 2) void spx_fwd_kernel(const SpxFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const spx_plan& pl = a.plan;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_per_img = (a.HW + SPX_TILE_PX - 1) / SPX_TILE_PX;
    const int b = blockIdx.x / tiles_per_img;
    const int px0 = (blockIdx.x % tiles_per_img) * SPX_TILE_PX;
    const int C = pl.num_scales * pl.channels_per_scale;
    const int P = pl.num_prototypes, K = pl.num_classes;

    SpxTileCtx tc;
    tc.x = (const char*)a.x + (size_t)b * C * a.HW * (XF32 ? 4 : 2);
    tc.hw = a.HW;
    tc.px0 = px0;
    tc.vec_ok = a.vec_ok;

    const int stage = spx_stage_bytes(pl.kc, pl.npb);
    const int xs_bytes = pl.kc * SPX_XROW * 2;
    const int nchunks = pl.channels_per_scale / pl.kc;
    const int nks = pl.kc >> 4;
    const int chunk_bytes = pl.npb * nks * 1024;
    const int total = pl.npanels * nchunks;

    SpxStager<NPB, XF32> st;
    f32x16 acc[NPB];
    f32x16 accl[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) accl[cb][i] = 0.0f;
    float x2part = 0.0f;

    const int px = px0 + 32 * wave + r;        // this lane's pixel
    const bool px_ok = px < a.HW;
    const bool want_head = a.logits != nullptr;

    st.load(tc, pl, a.packed_bank, pl.panel_ch0[0], tid);
    st.write(pl, smem, smem + xs_bytes, tid);
    __syncthreads();

    int buf = 0;
    for (int step = 0; step < total; ++step) {
        const int panel = step / nchunks, chunk = step - panel * nchunks;
        if (chunk == 0) {
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[pb][i] = 0.0f;
            x2part = 0.0f;
        }
        const bool more = step + 1 < total;
        if (more) {
            const int np_ = (step + 1) / nchunks, nc_ = (step + 1) - np_ * nchunks;
            st.load(tc, pl, a.packed_bank + (size_t)(step + 1) * chunk_bytes, pl.panel_ch0[np_] + nc_ * pl.kc, tid);
        }
        char* cur = smem + buf * stage;
        spx_compute_chunk<NPB>(acc, x2part, pl, cur, cur + xs_bytes, lane, wave);
        if (more) {
            char* nxt = smem + (buf ^ 1) * stage;
            st.write(pl, nxt, nxt + xs_bytes, tid);
        }
        __syncthreads();
        buf ^= 1;

        if (chunk == nchunks - 1) {
            // ---------------- panel epilogue ----------------
            const float x2 = x2part + __shfl_xor(x2part, 32);
            const int p0 = pl.panel_p0[panel], np = pl.panel_np[panel];
            const float* p2p = a.p2 + panel * pl.npb * 32;
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) {
                if (pb < pl.npb && pb * 32 < np) {
                    float av[16];
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int row0 = pb * 32 + 8 * g4 + 4 * h;   // rows row0..row0+3 <-> regs 4*g4..4*g4+3
                        const f32x4 p2v = *(const f32x4*)(p2p + row0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int reg = 4 * g4 + e;
                            float d = __builtin_fmaf(-2.0f, acc[pb][reg], p2v[e]) + x2;
                            d = fmaxf(d, 0.0f);
                            const int pl_row = row0 + e;
                            if (a.dist && px_ok && pl_row < np)
                                a.dist[((size_t)b * P + (p0 + pl_row)) * a.HW + px] = d;
                            av[reg] = a.act_fn == 0 ? act_log(d, a.eps) : -d;
                        }
                        if (a.act && px_ok) {
                            float* dst = a.act + ((size_t)b * a.HW + px) * P + p0 + row0;
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (row0 + e < np) dst[e] = av[4 * g4 + e];
                        }
                    }
                    if (want_head) {
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) {
                            bf16x8 ahi, alo;
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                __bf16 hi, lo;
                                split_bf16(av[8 * s2 + j], hi, lo);
                                ahi[j] = hi;
                                alo[j] = lo;
                            }
#pragma unroll
                            for (int cb = 0; cb < NCB; ++cb) {
                                if (cb < pl.ncb) {
                                    const char* wf = a.packed_head +
                                        ((((size_t)cb * pl.npanels + panel) * pl.npb + pb) * 2 + s2) * 2048 + lane * 16;
                                    const bf16x8 whi = *(const bf16x8*)wf;
                                    const bf16x8 wlo = *(const bf16x8*)(wf + 1024);
                                    accl[cb] = mfma_bf16(whi, ahi, accl[cb]);
                                    accl[cb] = mfma_bf16(wlo, ahi, accl[cb]);
                                    accl[cb] = mfma_bf16(whi, alo, accl[cb]);
                                }
                            }
                        }
                    }
                }
            }
        }
    }

    if (want_head && px_ok) {
        float* dst = a.logits + ((size_t)b * a.HW + px) * K;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            if (cb < pl.ncb) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int cls = cb * 32 + acc_row(reg, h);
                    if (cls < K) dst[cls] = accl[cb][reg];
                }
            }
        }
    }
}

template <int NPB, int NCB>
static hipError_t launch_fwd_x(const SpxFwdArgs& a, int x_dtype, dim3 grid, size_t lds, hipStream_t s) {
    if (x_dtype == 1)
        hipLaunchKernelGGL((spx_fwd_kernel<NPB, NCB, true>), grid, dim3(256), lds, s, a);
    else
        hipLaunchKernelGGL((spx_fwd_kernel<NPB, NCB, false>), grid, dim3(256), lds, s, a);
    return hipGetLastError();
}

hipError_t spx_launch_fwd(const SpxFwdArgs& a, int x_dtype, hipStream_t s) {
    const spx_plan& pl = a.plan;
    const int tiles = (a.HW + SPX_TILE_PX - 1) / SPX_TILE_PX;
    dim3 grid((unsigned)(tiles * a.B));
    const size_t lds = 2 * (size_t)spx_stage_bytes(pl.kc, pl.npb);
    const bool small_p = pl.npb <= 2;
    const bool small_k = pl.ncb <= 1;
    if (small_p && small_k) return launch_fwd_x<2, 1>(a, x_dtype, grid, lds, s);
    if (small_k) return launch_fwd_x<6, 1>(a, x_dtype, grid, lds, s);
    if (small_p) return launch_fwd_x<2, 5>(a, x_dtype, grid, lds, s);
    return launch_fwd_x<6, 5>(a, x_dtype, grid, lds, s);
}
