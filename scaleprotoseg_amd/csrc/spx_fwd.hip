// Fused forward of the prototype-distance path:
//   distances = relu(|x|^2 - 2 x.p + |p|^2)                (model_multiscale.py:255-317)
//   activations = log((d+1)/(d+eps)) | -d                  (:324-330)
//   logits = activations . W^T                             (:243-244, :369-376)
// one launch, one pass over X, distances written once in the reference's [B,P,H,W] layout.
#include "spx_args.h"
#include "spx_mainloop.h"

#ifndef SPX_FWD_WAVES
#define SPX_FWD_WAVES 2
#endif

// LDS carve (bytes): [stage 0][stage 1][head fragments of the current panel][|p|^2 of the panel]
__host__ __device__ inline int spx_fwd_head_lds_bytes(const spx_plan& pl) {
    const int b = pl.ncb * pl.npb * 4096;
    return b <= 32768 ? b : 0;          // large heads (ADE: 5 class blocks) stream their fragments from L2 instead
}
__host__ __device__ inline int spx_fwd_lds_bytes(const spx_plan& pl) {
    return 2 * spx_stage_bytes(pl.kc, pl.npb) + spx_fwd_head_lds_bytes(pl) + pl.npb * 32 * 4;
}

template <int NPB, int NCB, bool XF32, bool ACT_LOG>
__global__ __launch_bounds__(256, SPX_FWD_WAVES) void spx_fwd_kernel(const SpxFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const spx_plan& pl = a.plan;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_per_img = (a.HW + SPX_TILE_PX - 1) / SPX_TILE_PX;
    const int b = blockIdx.x / tiles_per_img;
    const int px0 = (blockIdx.x % tiles_per_img) * SPX_TILE_PX;
    const int C = pl.num_scales * pl.channels_per_scale;
    const int P = pl.num_prototypes, K = pl.num_classes;
    const uint32_t HW = (uint32_t)a.HW;

    const SpxTileCtx tc = SpxStager<NPB, XF32>::make_ctx(
        (const char*)a.x + (size_t)b * C * a.HW * (XF32 ? 4 : 2), a.packed_bank, a.HW, px0, a.vec_ok, tid);

    const int stage = spx_stage_bytes(pl.kc, pl.npb);
    const int xs_bytes = pl.kc * SPX_XROW * 2;
    const int nchunks = pl.channels_per_scale / pl.kc;
    const int nks = pl.kc >> 4;
    const uint32_t chunk_bytes = (uint32_t)(pl.npb * nks * 1024);
    const int total = pl.npanels * nchunks;
    const int head_lds = spx_fwd_head_lds_bytes(pl);
    char* const wlds = smem + 2 * stage;
    float* const p2s = (float*)(wlds + head_lds);

    SpxStager<NPB, XF32> stA, stB;
    f32x16 acc[NPB];
    f32x16 accl[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) accl[cb][i] = 0.0f;
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[pb][i] = 0.0f;
    float x2part = 0.0f;

    const int px = px0 + 32 * wave + r;        // this lane's pixel
    const bool px_ok = px < a.HW;
    const bool want_head = a.logits != nullptr;
    // per-lane byte offsets, fixed for the whole kernel (SPX_OOB = access dropped); row / block selection
    // rides on wave-uniform SGPR offsets
    const uint32_t voff_d = px_ok ? ((uint32_t)(4 * h) * HW + (uint32_t)px) * 4u : SPX_OOB;                 // [row][px]
    const uint32_t voff_a = px_ok ? ((uint32_t)px * (uint32_t)P + (uint32_t)(4 * h)) * 4u : SPX_OOB;        // [px][row]
    const spx_rsrc hr = make_rsrc(a.packed_head);
    const spx_rsrc p2r = make_rsrc(a.p2);

    // panel prologue: head fragments + |p|^2 of the panel -> LDS (read in the epilogue, after >= 1 barrier)
    auto stage_panel_consts = [&](int panel) {
        if (want_head && head_lds) {
            const int per_cb = pl.npb * 4096;
            for (int off = tid * 16; off < head_lds; off += 256 * 16) {
                const int cb = off / per_cb, rem = off - cb * per_cb;
                const uint32_t so = (uint32_t)((cb * pl.npanels + panel) * per_cb);
                *(u32x4*)(wlds + off) = buf_load_b128(hr, (uint32_t)rem, so);
            }
        }
        if (tid < pl.npb * 32) p2s[tid] = buf_load_f32(p2r, (uint32_t)tid * 4u, (uint32_t)(panel * pl.npb * 32 * 4));
    };

    // Panel epilogue as a ROLLED loop over the panel's 32-prototype blocks: the block being finished is always
    // acc[0] and the accumulator array is rotated after each block (NPB-1 register-tile moves), so the body is
    // compiled once with a small, fixed register footprint instead of NPB unrolled copies.
    auto epilogue = [&](int panel) {
        const float x2 = x2part + __shfl_xor(x2part, 32);
        const int p0 = pl.panel_p0[panel], np = pl.panel_np[panel];
#pragma unroll 1
        for (int pb = 0; pb < pl.npb; ++pb) {
            if (pb * 32 < np) {
                const spx_rsrc dr = make_rsrc_pred(a.dist ? a.dist + ((size_t)b * P + p0 + pb * 32) * a.HW : nullptr);
                const spx_rsrc ar = make_rsrc_pred(a.act ? a.act + (size_t)b * a.HW * P + p0 + pb * 32 : nullptr);
                const bool full = pb * 32 + 32 <= np;       // wave-uniform: no per-row predication needed
                // all arithmetic first (one straight-line block), then the stores under wave-uniform conditions
                float dv[16], av[16];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    // rows 8*g4 + 4h + (0..3) of the block <-> registers 4*g4..4*g4+3
                    const f32x4 p2v = *(const f32x4*)(p2s + pb * 32 + 8 * g4 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int reg = 4 * g4 + e;
                        const float d = fmaxf(__builtin_fmaf(-2.0f, acc[0][reg], p2v[e]) + x2, 0.0f);
                        dv[reg] = d;
                        av[reg] = ACT_LOG ? act_log(d, a.eps) : -d;
                    }
                }
                if (a.dist) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int rb = (reg & 3) + 8 * (reg >> 2);
                        const uint32_t vo = (full || (pb * 32 + rb + 4 * h < np)) ? voff_d : SPX_OOB;
                        buf_store_f32(dv[reg], dr, vo, (uint32_t)rb * HW * 4u);
                    }
                }
                if (a.act) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int rb = (reg & 3) + 8 * (reg >> 2);
                        const uint32_t vo = (full || (pb * 32 + rb + 4 * h < np)) ? voff_a : SPX_OOB;
                        buf_store_f32(av[reg], ar, vo, (uint32_t)(rb * 4));
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
                                bf16x8 whi, wlo;
                                if (head_lds) {
                                    const char* wf = wlds + ((cb * pl.npb + pb) * 2 + s2) * 2048 + lane * 16;
                                    whi = *(const bf16x8*)wf;
                                    wlo = *(const bf16x8*)(wf + 1024);
                                } else {
                                    const uint32_t so = (uint32_t)((((cb * pl.npanels + panel) * pl.npb + pb) * 2 + s2) * 2048);
                                    whi = __builtin_bit_cast(bf16x8, buf_load_b128(hr, (uint32_t)lane * 16u, so));
                                    wlo = __builtin_bit_cast(bf16x8, buf_load_b128(hr, (uint32_t)lane * 16u, so + 1024u));
                                }
                                accl[cb] = mfma_bf16(whi, ahi, accl[cb]);
                                accl[cb] = mfma_bf16(wlo, ahi, accl[cb]);
                                accl[cb] = mfma_bf16(whi, alo, accl[cb]);
                            }
                        }
                    }
                }
            }
            // rotate: next block -> acc[0]; the vacated slot is zero (= the next panel's initial accumulator)
#pragma unroll
            for (int i = 0; i + 1 < NPB; ++i) acc[i] = acc[i + 1];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[NPB - 1][i] = 0.0f;
        }
    };

    // One step of the software pipeline.  Global loads run two steps ahead of the MFMAs: while step i computes
    // from LDS[i&1], the registers of `nxt` (loaded during step i-1) are written to LDS[(i+1)&1] and `far`
    // issues the loads of step i+2.
    auto do_step = [&](int step, SpxStager<NPB, XF32>& far, SpxStager<NPB, XF32>& nxt) {
        const int panel = step / nchunks, chunk = step - panel * nchunks;
        if (chunk == 0) {
            x2part = 0.0f;
            stage_panel_consts(panel);
        }
        if (step + 2 < total) {
            const int np_ = (step + 2) / nchunks, nc_ = (step + 2) - np_ * nchunks;
            far.load(tc, pl, (uint32_t)(step + 2) * chunk_bytes, pl.panel_ch0[np_] + nc_ * pl.kc, tid);
        }
        char* cur = smem + (step & 1) * stage;
        spx_compute_chunk<NPB>(acc, x2part, pl, cur, cur + xs_bytes, lane, wave);
        if (step + 1 < total) {
            char* dst = smem + ((step + 1) & 1) * stage;
            nxt.write(pl, dst, dst + xs_bytes, tid);
        }
        __syncthreads();
        if (chunk == nchunks - 1) {
            epilogue(panel);
            if (step + 1 < total) __syncthreads();   // next panel's prologue overwrites the head / |p|^2 LDS
        }
    };

    stA.load(tc, pl, 0u, pl.panel_ch0[0], tid);
    if (total > 1) {
        const int np_ = 1 / nchunks, nc_ = 1 - np_ * nchunks;
        stB.load(tc, pl, chunk_bytes, pl.panel_ch0[np_] + nc_ * pl.kc, tid);
    }
    stA.write(pl, smem, smem + xs_bytes, tid);
    __syncthreads();
    for (int step = 0; step < total; step += 2) {
        do_step(step, stA, stB);
        if (step + 1 < total) do_step(step + 1, stB, stA);
    }

    if (want_head) {
        const spx_rsrc lr = make_rsrc_pred(a.logits + (size_t)b * a.HW * K);
        const uint32_t voff_l = px_ok ? ((uint32_t)px * (uint32_t)K + (uint32_t)(4 * h)) * 4u : SPX_OOB;   // [px][class]
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            if (cb < pl.ncb) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int cls = cb * 32 + acc_row(reg, h);
                    buf_store_f32(accl[cb][reg], lr, cls < K ? voff_l : SPX_OOB, (uint32_t)((cb * 32 + (reg & 3) + 8 * (reg >> 2)) * 4));
                }
            }
        }
    }
}

template <int NPB, int NCB>
static hipError_t launch_fwd_x(const SpxFwdArgs& a, int x_dtype, dim3 grid, size_t lds, hipStream_t s) {
    const bool lg = a.act_fn == 0;
    if (x_dtype == 1) {
        if (lg) hipLaunchKernelGGL((spx_fwd_kernel<NPB, NCB, true, true>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((spx_fwd_kernel<NPB, NCB, true, false>), grid, dim3(256), lds, s, a);
    } else {
        if (lg) hipLaunchKernelGGL((spx_fwd_kernel<NPB, NCB, false, true>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((spx_fwd_kernel<NPB, NCB, false, false>), grid, dim3(256), lds, s, a);
    }
    return hipGetLastError();
}

hipError_t spx_launch_fwd(const SpxFwdArgs& a, int x_dtype, hipStream_t s) {
    const spx_plan& pl = a.plan;
    const int tiles = (a.HW + SPX_TILE_PX - 1) / SPX_TILE_PX;
    dim3 grid((unsigned)(tiles * a.B));
    const size_t lds = (size_t)spx_fwd_lds_bytes(pl);
    const bool small_p = pl.npb <= 2;
    const bool small_k = pl.ncb <= 1;
    if (small_p && small_k) return launch_fwd_x<2, 1>(a, x_dtype, grid, lds, s);
    if (small_k) return launch_fwd_x<6, 1>(a, x_dtype, grid, lds, s);
    if (small_p) return launch_fwd_x<2, 5>(a, x_dtype, grid, lds, s);
    return launch_fwd_x<6, 5>(a, x_dtype, grid, lds, s);
}
