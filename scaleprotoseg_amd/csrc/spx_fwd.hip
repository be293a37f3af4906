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
// X chunks in flight per workgroup (register ring).  A/B on MI355X: 2 beats 4 (0.74 vs 0.78 ms): the loads are not
// what the loop waits for (in-kernel stamps: < 200 cycles per chunk), the extra registers only cost scheduling freedom
#ifndef SPX_FWD_XRING
#define SPX_FWD_XRING(xf32) 2
#endif

// LDS carve (bytes): [stage 0][stage 1][head fragments of the current panel (NCB == 1 only)][|p|^2 of the panel]
#ifdef SPX_FWD_HEAD_L2
template <int NPB, int NCB>
__host__ __device__ constexpr int spx_fwd_head_lds_bytes() { return 0; }   // experiment: head fragments straight from L2
#else
template <int NPB, int NCB>
__host__ __device__ constexpr int spx_fwd_head_lds_bytes() { return NCB == 1 ? NPB * 4096 : 0; }
#endif
template <int NPB, int NCB>
__host__ __device__ constexpr int spx_fwd_lds_bytes() {
    return 2 * spx_stage_bytes(NPB) + spx_fwd_head_lds_bytes<NPB, NCB>() + NPB * 32 * 4;
}

template <int NPB, int NCB, bool XF32, bool VEC>
__global__ __launch_bounds__(256, SPX_FWD_WAVES) void spx_fwd_kernel(const SpxFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const spx_plan& pl = a.plan;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int tiles_per_img = (a.HW + SPX_TILE_PX - 1) / SPX_TILE_PX;
    const int b = blockIdx.x / tiles_per_img;
    const int px0 = (blockIdx.x % tiles_per_img) * SPX_TILE_PX;
    const int Cs = pl.channels_per_scale;
    const int C = pl.num_scales * Cs;
    const int P = pl.num_prototypes, K = pl.num_classes;
    const uint32_t HW = (uint32_t)a.HW;
    constexpr int XR = SPX_FWD_XRING(XF32);
    using Pipe = SpxPipeline<NPB, XF32, VEC, XR>;

    const SpxTileCtx tc = SpxXStager<XF32, VEC>::make_ctx((const char*)a.x + (size_t)b * C * a.HW * (XF32 ? 4 : 2), a.HW, px0, tid);

    constexpr int stage = spx_stage_bytes(NPB);
    constexpr int chunk_bytes = NPB * 2 * 1024;
    constexpr int head_lds = spx_fwd_head_lds_bytes<NPB, NCB>();
    const int nchunks = (Cs + SPX_KC - 1) / SPX_KC;
    char* const wlds = smem + 2 * stage;
    float* const p2s = (float*)(wlds + head_lds);

    Pipe pipe;
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
    const spx_rsrc hr = make_rsrc_pred(a.packed_head);
    const spx_rsrc p2r = make_rsrc_pred(a.p2);

    // panel prologue: head fragments + |p|^2 of the panel -> LDS (read in the epilogue, after >= 1 barrier)
    constexpr int HPASS = head_lds / 4096;
    u32x4 hreg[HPASS > 0 ? HPASS : 1];
    float p2reg = 0.0f;
    auto consts_issue = [&](int panel) {
#pragma unroll
        for (int i = 0; i < HPASS; ++i)
            hreg[i] = buf_load_b128(hr, want_head ? (uint32_t)(i * 4096 + tid * 16) : SPX_OOB, (uint32_t)(panel * head_lds));
        p2reg = buf_load_f32(p2r, tid < NPB * 32 ? (uint32_t)tid * 4u : SPX_OOB, (uint32_t)(panel * NPB * 32 * 4));
    };
    auto consts_commit = [&]() {
#pragma unroll
        for (int i = 0; i < HPASS; ++i) *(u32x4*)(wlds + i * 4096 + tid * 16) = hreg[i];
        if (tid < NPB * 32) p2s[tid] = p2reg;
    };

    // Panel epilogue as a ROLLED loop over the panel's 32-prototype blocks (the body is compiled once, with a
    // small fixed register footprint, instead of NPB unrolled copies); the block's accumulator tile is fetched
    // with a wave-uniform select over static register indices and cleared for the next panel.
    auto epilogue = [&](int panel) {
        const float x2 = x2part + __shfl_xor(x2part, 32);
        const int p0 = pl.panel_p0[panel], np = pl.panel_np[panel];
#pragma unroll 1
        for (int pb = 0; pb < NPB; ++pb) {
            const f32x16 tile = tile_get<NPB>(acc, pb);
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
                    for (int e = 0; e < 4; ++e) dv[4 * g4 + e] = fmaxf(__builtin_fmaf(-2.0f, tile[4 * g4 + e], p2v[e]) + x2, 0.0f);
                }
                if (a.act_fn == 0) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) av[reg] = act_log(dv[reg], a.eps);
                } else {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) av[reg] = -dv[reg];
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
                            bf16x8 whi, wlo;
                            if (head_lds) {
                                const char* wf = wlds + ((cb * NPB + pb) * 2 + s2) * 2048 + lane * 16;
                                whi = *(const bf16x8*)wf;
                                wlo = *(const bf16x8*)(wf + 1024);
                            } else {
                                const uint32_t so = (uint32_t)((((cb * pl.npanels + panel) * NPB + pb) * 2 + s2) * 2048);
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
        // the next panel accumulates from zero
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[pb][i] = 0.0f;
    };

#ifdef SPX_DIAG_STAMPS
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), t1 = 0, t2 = 0;
#endif
    for (int panel = 0; panel < pl.npanels; ++panel) {
        const char* bank0 = a.packed_bank + (size_t)(panel * nchunks) * chunk_bytes;
        x2part = 0.0f;
        pipe.run_panel(acc, x2part, tc, smem, bank0, pl.panel_ch0[panel], Cs, lane, wave, tid,
                       [&]() { consts_issue(panel); }, consts_commit);
#ifdef SPX_DIAG_STAMPS
        t1 = __builtin_amdgcn_s_memtime();
#endif
        epilogue(panel);
#ifdef SPX_DIAG_STAMPS
        t2 = __builtin_amdgcn_s_memtime();
#endif
        if (panel + 1 < pl.npanels) __syncthreads();   // next panel's prologue overwrites the head / |p|^2 / stage LDS
    }

    if (want_head) {
        const spx_rsrc lr = make_rsrc_pred(a.logits + (size_t)b * a.HW * K);
        const uint32_t voff_l = px_ok ? ((uint32_t)px * (uint32_t)K + (uint32_t)(4 * h)) * 4u : SPX_OOB;   // [px][class]
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int cls = cb * 32 + acc_row(reg, h);
                buf_store_f32(accl[cb][reg], lr, cls < K ? voff_l : SPX_OOB, (uint32_t)((cb * 32 + (reg & 3) + 8 * (reg >> 2)) * 4));
            }
        }
    }
#ifdef SPX_DIAG_STAMPS
    if (a.dbg && tid == 0) {
        unsigned long long t3 = __builtin_amdgcn_s_memtime();
        unsigned long long* d = a.dbg + (size_t)blockIdx.x * 4;
        d[0] = t0; d[1] = t1; d[2] = t2; d[3] = t3;
        unsigned long long* e = a.dbg + (size_t)gridDim.x * 4 + (size_t)blockIdx.x * 4;
        e[0] = pipe.dg_compute; e[1] = pipe.dg_write; e[2] = pipe.dg_barrier; e[3] = 0;
    }
#endif
}

template <int NPB, int NCB>
static hipError_t launch_fwd_x(const SpxFwdArgs& a, int x_dtype, dim3 grid, hipStream_t s) {
    constexpr size_t lds = (size_t)spx_fwd_lds_bytes<NPB, NCB>();
    if (x_dtype == 1) {
        if (a.vec_ok) hipLaunchKernelGGL((spx_fwd_kernel<NPB, NCB, true, true>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((spx_fwd_kernel<NPB, NCB, true, false>), grid, dim3(256), lds, s, a);
    } else {
        if (a.vec_ok) hipLaunchKernelGGL((spx_fwd_kernel<NPB, NCB, false, true>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((spx_fwd_kernel<NPB, NCB, false, false>), grid, dim3(256), lds, s, a);
    }
    return hipGetLastError();
}

hipError_t spx_launch_fwd(const SpxFwdArgs& a, int x_dtype, hipStream_t s) {
    const spx_plan& pl = a.plan;
    const int tiles = (a.HW + SPX_TILE_PX - 1) / SPX_TILE_PX;
    dim3 grid((unsigned)(tiles * a.B));
    if (pl.ncb == 1) {
        if (pl.npb == 2) return launch_fwd_x<2, 1>(a, x_dtype, grid, s);
        if (pl.npb == 4) return launch_fwd_x<4, 1>(a, x_dtype, grid, s);
        return launch_fwd_x<6, 1>(a, x_dtype, grid, s);
    }
    if (pl.npb == 2) return launch_fwd_x<2, 5>(a, x_dtype, grid, s);
    if (pl.npb == 4) return launch_fwd_x<4, 5>(a, x_dtype, grid, s);
    return launch_fwd_x<6, 5>(a, x_dtype, grid, s);
}
