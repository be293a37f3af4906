// Backward of the prototype-distance path (replaces autograd through
// segmentation/model/model_multiscale.py:255-281, :324-330, :243-244).
//
// With d_raw = |x|^2 - 2 x.p + |p|^2, d = relu(d_raw), a = act(d), logits = a.W^T:
//   gA = dAct + dLogits.W                                 [pixel x proto]
//   G  = (dDist + gA * act'(d)) * [d_raw > 0]
//   dX[m, c in s] = 2 (x[m,c] * rowsum_s(G)[m] - (G.P)[m,c])
//   dP[p, c]      = 2 (p[p,c] * colsum(G)[p]   - (G^T.X)[p,c])
//   dW[k, p]      = (dLogits^T . a)[k,p]
//
// Kernel 1 (pixel side, spx_bwd_kernel): same tiling and main loop as the forward (the x.p tile is
// recomputed with identical arithmetic, so the relu mask is the forward's bit for bit — cheaper than
// re-reading the fp32 distance map: 2*P*C flop/px on the matrix pipe vs 4*P bytes/px of HBM), then G, dX,
// and bf16 copies of G and a in prototype-major [P_pad][B*HWp] order for kernel 2.
// Kernel 2 (parameter side, spx_bank_bwd_kernel): pixel-split MFMA reduction G^T.X and a^T.dLogits with
// per-workgroup fp32 partial slabs; kernel 3 sums the slabs in a fixed order (no float atomics).
#include "spx_args.h"
#include "spx_mainloop.h"

#ifndef SPX_BWD_WAVES
#define SPX_BWD_WAVES 1
#endif

// ------------------------------------------------------------------------------------------------
// kernel 1: pixel side
// ------------------------------------------------------------------------------------------------
template <int NPB, int NCB, int NCHB, bool XF32>
__global__ __launch_bounds__(256, SPX_BWD_WAVES) void spx_bwd_kernel(const SpxBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const spx_plan& pl = a.plan;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_per_img = (a.HW + SPX_TILE_PX - 1) / SPX_TILE_PX;
    const int b = blockIdx.x / tiles_per_img;
    const int px0 = (blockIdx.x % tiles_per_img) * SPX_TILE_PX;
    const int Cs = pl.channels_per_scale;
    const int C = pl.num_scales * Cs;
    const int P = pl.num_prototypes, K = pl.num_classes;
    const int nchb = (Cs + 31) / 32;
    const int ncstep = pl.ncb * 2;
    const uint32_t HW = (uint32_t)a.HW;
    const size_t Mp = (size_t)a.B * a.HWp;
    constexpr int ESZ = XF32 ? 4 : 2;

    const char* x_img = (const char*)a.x + (size_t)b * C * a.HW * ESZ;
    const SpxTileCtx tc = SpxStager<NPB, XF32>::make_ctx(x_img, a.packed_bank, a.HW, px0, a.vec_ok, tid);

    const int stage = spx_stage_bytes(pl.kc, pl.npb);
    const int xs_bytes = pl.kc * SPX_XROW * 2;
    const int nchunks = Cs / pl.kc;
    const int nks = pl.kc >> 4;
    const uint32_t chunk_bytes = (uint32_t)(pl.npb * nks * 1024);
    const int total = pl.npanels * nchunks;

    const int px = px0 + 32 * wave + r;
    const bool px_ok = px < a.HW;
    const bool px_pad_ok = px < a.HWp;
    // per-lane byte offsets (fixed for the kernel); row/block selection rides on wave-uniform SGPR offsets
    const uint32_t voff_d = ((uint32_t)(4 * h) * HW + (uint32_t)px) * 4u;             // [row][px] fp32 maps
    const uint32_t voff_a = ((uint32_t)px * (uint32_t)P + (uint32_t)(4 * h)) * 4u;    // dAct [px][row]
    const uint32_t voff_x = ((uint32_t)(4 * h) * HW + (uint32_t)px) * ESZ;            // X / dX [ch][px]
    const uint32_t voff_g = (uint32_t)(((size_t)(4 * h) * Mp + (size_t)b * a.HWp + px) * 2);   // G/a scratch rows
    const spx_rsrc htr = make_rsrc(a.packed_headT);
    const spx_rsrc btr = make_rsrc(a.packed_bankT);
    const spx_rsrc xir = make_rsrc(x_img);
    const spx_rsrc dxr = make_rsrc(a.dx ? (char*)a.dx + (size_t)b * C * a.HW * ESZ : nullptr);

    // dLogits of this lane's pixel as split-bf16 B fragments: element j of k-step c <-> class 16c + 8h + j
    bf16x8 dlhi[NCB * 2], dllo[NCB * 2];
    {
        const spx_rsrc lr = make_rsrc(a.d_logits ? a.d_logits + (size_t)b * a.HW * K : nullptr);
        const uint32_t voff_l = ((uint32_t)px * (uint32_t)K + (uint32_t)(8 * h)) * 4u;
#pragma unroll
        for (int c = 0; c < NCB * 2; ++c) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = 0.0f;
                const int cls = c * 16 + 8 * h + j;
                if (a.d_logits && px_ok && c < ncstep && cls < K) v = buf_load_f32(lr, voff_l, (uint32_t)((c * 16 + j) * 4));
                __bf16 hi, lo;
                split_bf16(v, hi, lo);
                dlhi[c][j] = hi;
                dllo[c][j] = lo;
            }
        }
    }

    SpxStager<NPB, XF32> st;
    f32x16 acc[NPB];
    float x2part = 0.0f;

    st.load(tc, pl, 0u, pl.panel_ch0[0], tid);
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
            st.load(tc, pl, (uint32_t)(step + 1) * chunk_bytes, pl.panel_ch0[np_] + nc_ * pl.kc, tid);
        }
        char* cur = smem + buf * stage;
        spx_compute_chunk<NPB>(acc, x2part, pl, cur, cur + xs_bytes, lane, wave);
        if (more) {
            char* nxt = smem + (buf ^ 1) * stage;
            st.write(pl, nxt, nxt + xs_bytes, tid);
        }
        __syncthreads();
        buf ^= 1;
        if (chunk != nchunks - 1) continue;

        // ---------------- panel epilogue, phase 1: G ----------------
        const float x2 = x2part + __shfl_xor(x2part, 32);
        const int p0 = pl.panel_p0[panel], np = pl.panel_np[panel];
        const int ch0 = pl.panel_ch0[panel];
        const spx_rsrc p2r = make_rsrc(a.p2 + panel * pl.npb * 32);
        bf16x8 gpk[NPB][2];
        float rs = 0.0f;
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) gpk[pb][s2][j] = (__bf16)0.0f;
            const size_t grow = (size_t)panel * pl.npb * 32 + pb * 32;
            const spx_rsrc gr = make_rsrc(a.g_out ? a.g_out + grow * Mp : nullptr);
            const spx_rsrc ar = make_rsrc(a.a_out ? a.a_out + grow * Mp : nullptr);
            if (pb < pl.npb && pb * 32 < np) {
                f32x16 ga;
#pragma unroll
                for (int i = 0; i < 16; ++i) ga[i] = 0.0f;
                if (a.d_logits) {
#pragma unroll
                    for (int c = 0; c < NCB * 2; ++c) {
                        if (c < ncstep) {
                            const uint32_t so = (uint32_t)(((panel * pl.npb + pb) * ncstep + c) * 2048);
                            const bf16x8 whi = __builtin_bit_cast(bf16x8, buf_load_b128(htr, (uint32_t)lane * 16u, so));
                            const bf16x8 wlo = __builtin_bit_cast(bf16x8, buf_load_b128(htr, (uint32_t)lane * 16u, so + 1024u));
                            ga = mfma_bf16(whi, dlhi[c], ga);
                            ga = mfma_bf16(wlo, dlhi[c], ga);
                            ga = mfma_bf16(whi, dllo[c], ga);
                        }
                    }
                }
                const spx_rsrc ddr = make_rsrc(a.d_dist ? a.d_dist + ((size_t)b * P + p0 + pb * 32) * a.HW : nullptr);
                const spx_rsrc dar = make_rsrc(a.d_act ? a.d_act + (size_t)b * a.HW * P + p0 + pb * 32 : nullptr);
                float gv[16];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int row0 = 8 * g4 + 4 * h;
                    const u32x4 p2u = buf_load_b128(p2r, (uint32_t)(16 * h), (uint32_t)((pb * 32 + 8 * g4) * 4));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int reg = 4 * g4 + e;
                        const bool valid = px_ok && (pb * 32 + row0 + e < np);
                        const float d_raw = __builtin_fmaf(-2.0f, acc[pb][reg], __uint_as_float(p2u[e])) + x2;
                        const float d = fmaxf(d_raw, 0.0f);
                        float dact;   // act'(d)
                        float aval;
                        if (a.act_fn == 0) {
                            dact = -(1.0f - a.eps) * __fdividef(1.0f, (d + 1.0f) * (d + a.eps));
                            aval = act_log(d, a.eps);
                        } else {
                            dact = -1.0f;
                            aval = -d;
                        }
                        float gtot = ga[reg];
                        float dd = 0.0f;
                        if (valid) {
                            if (a.d_act) gtot += buf_load_f32(dar, voff_a, (uint32_t)((8 * g4 + e) * 4));
                            if (a.d_dist) dd = buf_load_f32(ddr, voff_d, (uint32_t)(8 * g4 + e) * HW * 4u);
                        }
                        const float G = (valid && d_raw > 0.0f) ? dd + gtot * dact : 0.0f;
                        rs += G;
                        gv[reg] = G;
                        if (px_pad_ok) {
                            const uint32_t so = (uint32_t)((size_t)(8 * g4 + e) * Mp * 2);
                            if (a.g_out) buf_store_u16(__builtin_bit_cast(uint16_t, (__bf16)G), gr, voff_g, so);
                            if (a.a_out) buf_store_u16(__builtin_bit_cast(uint16_t, (__bf16)(valid ? aval : 0.0f)), ar, voff_g, so);
                        }
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int j = 0; j < 8; ++j) gpk[pb][s2][j] = (__bf16)gv[8 * s2 + j];
                __builtin_amdgcn_sched_barrier(0);   // keep later blocks' loads out of this block (VGPR budget)
            } else if (pb < pl.npb && px_pad_ok) {
                // wholly padded prototype block: kernel 2 still reads these rows -> keep them finite (zero)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const uint32_t so = (uint32_t)((size_t)((reg & 3) + 8 * (reg >> 2)) * Mp * 2);
                    if (a.g_out) buf_store_u16(0, gr, voff_g, so);
                    if (a.a_out) buf_store_u16(0, ar, voff_g, so);
                }
            }
        }
        if (!a.dx) continue;

        // ---------------- phase 2: dX^T[ch x px] = 2 (rs * x - P^T . G) ----------------
        const float rs_tot = rs + __shfl_xor(rs, 32);
        const bool first_of_scale = (panel == 0) || (pl.panel_ch0[panel - 1] != ch0);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int CG = NCHB < 4 ? NCHB : 4;          // channel blocks per pass (<= 64 accumulator registers)
#pragma unroll
        for (int cg = 0; cg < NCHB; cg += CG) {
            if (cg >= nchb) break;
            f32x16 accx[CG];
#pragma unroll
            for (int c = 0; c < CG; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) accx[c][i] = 0.0f;
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) {
                if (pb < pl.npb && pb * 32 < np) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                        for (int c = 0; c < CG; ++c) {
                            const int chb = cg + c;
                            if (chb < nchb) {
                                const uint32_t so = (uint32_t)(((((panel * pl.npb + pb) * 2 + s2) * nchb) + chb) * 1024);
                                const bf16x8 pf = __builtin_bit_cast(bf16x8, buf_load_b128(btr, (uint32_t)lane * 16u, so));
                                accx[c] = mfma_bf16(pf, gpk[pb][s2], accx[c]);
                            }
                        }
                    }
                }
            }
            if (px_ok) {
#pragma unroll
                for (int c = 0; c < CG; ++c) {
                    const int chb = cg + c;
                    if (chb < nchb) {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const int chl = chb * 32 + (reg & 3) + 8 * (reg >> 2);     // + 4h rides in voff_x
                            if (chl + 4 * h < Cs) {
                                const uint32_t so = (uint32_t)(ch0 + chl) * HW * ESZ;
                                float xv, prev = 0.0f;
                                if (XF32) {
                                    xv = buf_load_f32(xir, voff_x, so);
                                    if (!first_of_scale) prev = buf_load_f32(dxr, voff_x, so);
                                } else {
                                    xv = (float)__builtin_bit_cast(__bf16, buf_load_u16(xir, voff_x, so));
                                    if (!first_of_scale) prev = (float)__builtin_bit_cast(__bf16, buf_load_u16(dxr, voff_x, so));
                                }
                                const float v = prev + 2.0f * (rs_tot * xv - accx[c][reg]);
                                if (XF32)
                                    buf_store_f32(v, dxr, voff_x, so);
                                else
                                    buf_store_u16(__builtin_bit_cast(uint16_t, (__bf16)v), dxr, voff_x, so);
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int NPB, int NCB, int NCHB>
static hipError_t launch_bwd_x(const SpxBwdArgs& a, int x_dtype, dim3 grid, size_t lds, hipStream_t s) {
    if (x_dtype == 1)
        hipLaunchKernelGGL((spx_bwd_kernel<NPB, NCB, NCHB, true>), grid, dim3(256), lds, s, a);
    else
        hipLaunchKernelGGL((spx_bwd_kernel<NPB, NCB, NCHB, false>), grid, dim3(256), lds, s, a);
    return hipGetLastError();
}

hipError_t spx_launch_bwd(const SpxBwdArgs& a, int x_dtype, hipStream_t s) {
    const spx_plan& pl = a.plan;
    const int tiles = (a.HW + SPX_TILE_PX - 1) / SPX_TILE_PX;
    dim3 grid((unsigned)(tiles * a.B));
    const size_t lds = 2 * (size_t)spx_stage_bytes(pl.kc, pl.npb);
    const bool small_p = pl.npb <= 2;
    const bool small_c = pl.channels_per_scale <= 64;
    if (pl.ncb <= 1) {
        if (small_p && small_c) return launch_bwd_x<2, 1, 2>(a, x_dtype, grid, lds, s);
        if (small_p) return launch_bwd_x<2, 1, 8>(a, x_dtype, grid, lds, s);
        if (small_c) return launch_bwd_x<6, 1, 2>(a, x_dtype, grid, lds, s);
        return launch_bwd_x<6, 1, 8>(a, x_dtype, grid, lds, s);
    }
    if (small_c) return launch_bwd_x<6, 5, 2>(a, x_dtype, grid, lds, s);
    return launch_bwd_x<6, 5, 8>(a, x_dtype, grid, lds, s);
}

// ------------------------------------------------------------------------------------------------
// kernel 2: parameter side   S[q][row][col] = sum_px Gq[row][px] * Xs[col][px]   (+ a^T.dLogits, colsum G)
// ------------------------------------------------------------------------------------------------
#define SPX_BK_PX 64          // pixels per K-chunk
#define SPX_BK_ROW 144        // LDS row stride in bytes (128 + 16: conflict-free ds_read_b128 over 16 rows)

__host__ __device__ inline int spx_bk_wstride(const spx_plan& pl) {
    return ((pl.channels_per_scale + 31) / 32) * 32 + pl.ncb * 32 + 32;   // [dP cols | dW cols | colsum + pad]
}
int spx_bank_bwd_nsplit(const spx_plan& pl, int B, int HW) {
    const long long chunks = (long long)B * ((HW + SPX_BK_PX - 1) / SPX_BK_PX);
    long long n = 512 / pl.npanels;
    if (n < 1) n = 1;
    if (n > chunks) n = chunks;
    return (int)n;
}
size_t spx_bank_bwd_ws_floats(const spx_plan& pl, int nsplit) {
    return (size_t)nsplit * pl.npanels * pl.npb * 32 * spx_bk_wstride(pl);
}

template <int NPB, int NCB, bool XF32>
__global__ __launch_bounds__(256, 1) void spx_bank_bwd_kernel(const SpxBankBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const spx_plan& pl = a.plan;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int q = blockIdx.y, split = blockIdx.x;
    const int Cs = pl.channels_per_scale, K = pl.num_classes;
    const int C = pl.num_scales * Cs;
    const int nchb = (Cs + 31) / 32;
    const int rows = pl.npb * 32;
    const int ch0 = pl.panel_ch0[q];
    const size_t Mp = (size_t)a.B * a.HWp;
    const int nci = (a.HW + SPX_BK_PX - 1) / SPX_BK_PX;
    const long long total = (long long)a.B * nci;
    const long long per = (total + a.nsplit - 1) / a.nsplit;
    const long long c_begin = split * per;
    const long long c_end = (c_begin + per < total) ? c_begin + per : total;
    const bool want_w = a.d_W != nullptr;
    const bool want_p = a.d_bank != nullptr;

    char* Gs = smem;                                  // [rows][144 B]
    char* As = Gs + rows * SPX_BK_ROW;                // [rows][144 B]
    char* Xs = As + rows * SPX_BK_ROW;                // [nchb*32][144 B]
    char* Ls = Xs + nchb * 32 * SPX_BK_ROW;           // [ncb*32][144 B] dLogits^T (bf16)

    // staging registers: 16-B pieces (8 px) of G, a, X rows
    constexpr int GP = NPB;                // rows*8/256 pieces per thread for G (and for a)
    constexpr int XP = 8;                  // <= 256 rows * 8 / 256
    u32x4 gr[GP], ar[GP], xr[XP][XF32 ? 2 : 1];
    f32x16 accp[NPB][2];
    f32x16 accw[2][NCB];
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) accp[pb][t][i] = 0.0f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int i = 0; i < 16; ++i) accw[t][cb][i] = 0.0f;
    float csum[GP];
#pragma unroll
    for (int i = 0; i < GP; ++i) csum[i] = 0.0f;

    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const int piece = tid & 7, prow = tid >> 3;      // piece of 8 px, row within a pass of 32 rows

    auto issue = [&](long long c) {
        const int b = (int)(c / nci);
        const int px0 = (int)(c - (long long)b * nci) * SPX_BK_PX;
        const int px = px0 + piece * 8;
        const size_t goff = (size_t)b * a.HWp + px;
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            const int row = prow + 32 * i;
            gr[i] = zero4;
            ar[i] = zero4;
            if (i < pl.npb && px < a.HWp) {
                const size_t o = ((size_t)q * rows + row) * Mp + goff;
                if (want_p) gr[i] = *(const u32x4*)(a.g_in + o);
                if (want_w) ar[i] = *(const u32x4*)(a.a_in + o);
            }
        }
        if (want_p) {
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const int row = prow + 32 * i;
#pragma unroll
                for (int w = 0; w < (XF32 ? 2 : 1); ++w) xr[i][w] = zero4;
                if (row < Cs) {
                    const size_t o = ((size_t)b * C + ch0 + row) * a.HW + px;
                    if (XF32) {
                        const float* src = (const float*)a.x + o;
                        if (a.vec_ok && px + 8 <= a.HW) {
                            xr[i][0] = *(const u32x4*)src;
                            xr[i][1] = *(const u32x4*)(src + 4);
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                xr[i][e >> 2][e & 3] = (px + e < a.HW) ? __float_as_uint(src[e]) : 0u;
                        }
                    } else {
                        const uint16_t* src = (const uint16_t*)a.x + o;
                        if (a.vec_ok && px + 8 <= a.HW) {
                            xr[i][0] = *(const u32x4*)src;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const uint32_t lo = (px + 2 * e < a.HW) ? src[2 * e] : 0u;
                                const uint32_t hi = (px + 2 * e + 1 < a.HW) ? src[2 * e + 1] : 0u;
                                xr[i][0][e] = lo | (hi << 16);
                            }
                        }
                    }
                }
            }
        }
    };
    auto commit = [&](long long c) {
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            if (i < pl.npb) {
                const int row = prow + 32 * i;
                *(u32x4*)(Gs + row * SPX_BK_ROW + piece * 16) = gr[i];
                *(u32x4*)(As + row * SPX_BK_ROW + piece * 16) = ar[i];
                // colsum(G) partial of this thread's 8 px of row `row`
                const bf16x8 gv8 = __builtin_bit_cast(bf16x8, gr[i]);
                float sum8 = 0.0f;
#pragma unroll
                for (int j = 0; j < 8; ++j) sum8 += (float)gv8[j];
                csum[i] += sum8;
            }
        }
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int row = prow + 32 * i;
            if (row < nchb * 32) {
                u32x4 v;
                if (XF32) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        bf16x2 p;
                        p[0] = (__bf16)__uint_as_float(xr[i][e >> 1][(2 * e) & 3]);
                        p[1] = (__bf16)__uint_as_float(xr[i][e >> 1][(2 * e + 1) & 3]);
                        v[e] = __builtin_bit_cast(uint32_t, p);
                    }
                } else {
                    v = xr[i][0];
                }
                *(u32x4*)(Xs + row * SPX_BK_ROW + piece * 16) = v;
            }
        }
        if (want_w) {
            // dLogits of the chunk's 64 px (contiguous [64][K] floats) -> bf16 [class][px]
            const int b = (int)(c / nci);
            const int px0 = (int)(c - (long long)b * nci) * SPX_BK_PX;
            const float* src = a.d_logits + ((size_t)b * a.HW + px0) * K;
            const int n = SPX_BK_PX * K;
            for (int e = tid; e < n; e += 256) {
                const int p = e / K, cls = e - p * K;
                const float v = (px0 + p < a.HW) ? src[e] : 0.0f;
                *(uint16_t*)(Ls + cls * SPX_BK_ROW + p * 2) = __builtin_bit_cast(uint16_t, (__bf16)v);
            }
        }
    };

    if (want_w) {   // padded class rows of the dLogits^T image stay zero for the whole kernel
        for (int e = tid; e < pl.ncb * 32 * SPX_BK_PX; e += 256) {
            const int cls = e / SPX_BK_PX, p = e - cls * SPX_BK_PX;
            if (cls >= K) *(uint16_t*)(Ls + cls * SPX_BK_ROW + p * 2) = 0;
        }
    }
    if (c_begin < c_end) issue(c_begin);
    for (long long c = c_begin; c < c_end; ++c) {
        commit(c);
        __syncthreads();
        if (c + 1 < c_end) issue(c + 1);
        // ---- MFMA over the chunk's 64 px (4 k-steps) ----
#pragma unroll
        for (int ks = 0; ks < SPX_BK_PX / 16; ++ks) {
            const int koff = (ks * 16 + 8 * h) * 2;
            if (want_p) {
                bf16x8 xb[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int chb = wave + 4 * t;
                    if (chb < nchb) xb[t] = *(const bf16x8*)(Xs + (chb * 32 + r) * SPX_BK_ROW + koff);
                }
#pragma unroll
                for (int pb = 0; pb < NPB; ++pb) {
                    if (pb < pl.npb) {
                        const bf16x8 gf = *(const bf16x8*)(Gs + (pb * 32 + r) * SPX_BK_ROW + koff);
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            if (wave + 4 * t < nchb) accp[pb][t] = mfma_bf16(gf, xb[t], accp[pb][t]);
                    }
                }
            }
            if (want_w) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int pb = wave + 4 * t;
                    if (pb < pl.npb) {
                        const bf16x8 af = *(const bf16x8*)(As + (pb * 32 + r) * SPX_BK_ROW + koff);
#pragma unroll
                        for (int cb = 0; cb < NCB; ++cb) {
                            if (cb < pl.ncb) {
                                const bf16x8 lf = *(const bf16x8*)(Ls + (cb * 32 + r) * SPX_BK_ROW + koff);
                                accw[t][cb] = mfma_bf16(af, lf, accw[t][cb]);
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- write this workgroup's partial slab ----
    const int ws = spx_bk_wstride(pl);
    float* slab = a.workspace + ((size_t)split * pl.npanels + q) * rows * ws;
    if (want_p) {
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) {
            if (pb < pl.npb) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int chb = wave + 4 * t;
                    if (chb < nchb) {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg)
                            slab[(size_t)(pb * 32 + acc_row(reg, h)) * ws + chb * 32 + r] = accp[pb][t][reg];
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            if (i < pl.npb) {
                float s = csum[i];
                s += __shfl_xor(s, 1);
                s += __shfl_xor(s, 2);
                s += __shfl_xor(s, 4);
                if (piece == 0) slab[(size_t)(prow + 32 * i) * ws + nchb * 32 + pl.ncb * 32] = s;
            }
        }
    }
    if (want_w) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int pb = wave + 4 * t;
            if (pb < pl.npb) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    if (cb < pl.ncb) {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg)
                            slab[(size_t)(pb * 32 + acc_row(reg, h)) * ws + nchb * 32 + cb * 32 + r] = accw[t][cb][reg];
                    }
                }
            }
        }
    }
}

// kernel 3: fixed-order sum of the slabs, + the p * colsum(G) term
__global__ void spx_bank_reduce_kernel(const SpxBankBwdArgs a) {
    const spx_plan& pl = a.plan;
    const int Cs = pl.channels_per_scale, K = pl.num_classes, P = pl.num_prototypes;
    const int nchb = (Cs + 31) / 32;
    const int rows = pl.npb * 32;
    const int ws = spx_bk_wstride(pl);
    const int ncols = Cs + K;
    const long long n = (long long)pl.npanels * rows * ncols;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n) return;
    const int col = (int)(gid % ncols);
    const int row = (int)((gid / ncols) % rows);
    const int q = (int)(gid / ((long long)ncols * rows));
    if (row >= pl.panel_np[q]) return;
    const int p = pl.panel_p0[q] + row;
    const size_t slab_stride = (size_t)pl.npanels * rows * ws;
    const float* base = a.workspace + ((size_t)q * rows + row) * ws;
    if (col < Cs) {
        if (!a.d_bank) return;
        float s = 0.0f, cs = 0.0f;
        for (int j = 0; j < a.nsplit; ++j) {
            s += base[(size_t)j * slab_stride + col];
            cs += base[(size_t)j * slab_stride + nchb * 32 + pl.ncb * 32];
        }
        a.d_bank[(size_t)p * Cs + col] = 2.0f * (a.bank[(size_t)p * Cs + col] * cs - s);
    } else {
        if (!a.d_W) return;
        const int k = col - Cs;
        float s = 0.0f;
        for (int j = 0; j < a.nsplit; ++j) s += base[(size_t)j * slab_stride + nchb * 32 + k];
        a.d_W[(size_t)k * P + p] = s;
    }
}

template <int NPB, int NCB>
static hipError_t launch_bank_x(const SpxBankBwdArgs& a, int x_dtype, dim3 grid, size_t lds, hipStream_t s) {
    if (x_dtype == 1)
        hipLaunchKernelGGL((spx_bank_bwd_kernel<NPB, NCB, true>), grid, dim3(256), lds, s, a);
    else
        hipLaunchKernelGGL((spx_bank_bwd_kernel<NPB, NCB, false>), grid, dim3(256), lds, s, a);
    return hipGetLastError();
}

hipError_t spx_launch_bank_bwd(const SpxBankBwdArgs& a, int x_dtype, hipStream_t s) {
    const spx_plan& pl = a.plan;
    const int rows = pl.npb * 32, nchb = (pl.channels_per_scale + 31) / 32;
    const size_t lds = (size_t)(2 * rows + nchb * 32 + pl.ncb * 32) * SPX_BK_ROW;
    dim3 grid((unsigned)a.nsplit, (unsigned)pl.npanels);
    hipError_t e;
    if (pl.ncb <= 1)
        e = pl.npb <= 2 ? launch_bank_x<2, 1>(a, x_dtype, grid, lds, s) : launch_bank_x<6, 1>(a, x_dtype, grid, lds, s);
    else
        e = launch_bank_x<6, 5>(a, x_dtype, grid, lds, s);
    if (e != hipSuccess) return e;
    const long long n = (long long)pl.npanels * rows * (pl.channels_per_scale + pl.num_classes);
    hipLaunchKernelGGL(spx_bank_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}
