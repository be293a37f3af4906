// KLD loss over class-gathered distances (reference: segmentation/model/loss.py:51-146).
//
// Input: vals [B, J, HW] (slot planes of spx_dist_fwd_cls), labels [B, HW] (class 0..K-1, else none).  A segment is
// (image, class); per segment and slot j the reference takes log_softmax of the slot's distances over the segment's
// pixels, then the symmetric KL of every slot pair, which is a function of the segment Gram matrix
//     A[seg][j][k] = sum_px p_j(px) * l_k(px),   l = log_softmax, p = exp(l)
// (kept as A[j][k] - A[j][j]: see spx_kld_pairs_kernel)
// Four streaming passes over vals (each 4 J bytes per pixel, nothing else): segment max, segment sum-exp, A, and the
// gradient (which needs no further reduction: see spx_kld_backward_kernel).  Segment reductions use per-workgroup LDS
// tables and INTEGER atomics (ordered float keys for the max, 64-bit fixed point for the sums), so results do not
// depend on the order of arrival: the loss is run-to-run bit-identical like the rest of the path.
#include "spx_common.h"
#include <algorithm>
#include <type_traits>

#define SPX_KLD_TABLE_LDS (60 * 1024)      // LDS budget of the per-class tables of the pair and gradient passes (class blocks beyond it)

#define SPX_KLD_MIN_WGS 512      // tile rows shrink (64 -> 32 -> 16) until the reduction passes launch at least this many workgroups
#define SPX_KLD_THREADS 256
#define SPX_KLD_PX_PER_WG 2048
#define SPX_KLD_MAXJ 16
#define SPX_KLD_TILE 64                 // W given: a workgroup's tile (64 x 64 pixels: four 16-column strips of 16 steps of 4 rows)

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const int lo = __shfl_xor((int)__double2loint(v), m), hi = __shfl_xor((int)__double2hiint(v), m);
        v += __hiloint2double(hi, lo);
    }
    return v;      // fixed butterfly order: deterministic
}
// Sum over the 64 lanes in a fixed order without the LDS crossbar: four DPP adds leave every lane of a 16-lane row with its
// row's sum, the four row sums are read back and added in row order.  ~8 vector instructions per value, against 12
// ds_bpermute + 6 double adds for the butterfly above: the pair pass publishes 132 such sums per class run.
__device__ __forceinline__ float wave_sum_f32(float v) {
    auto dpp_add = [](float x, auto ctrl) {
        return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    v = dpp_add(v, std::integral_constant<int, 0xB1>{});     // quad_perm [1,0,3,2]
    v = dpp_add(v, std::integral_constant<int, 0x4E>{});     // quad_perm [2,3,0,1]
    v = dpp_add(v, std::integral_constant<int, 0x141>{});    // row_half_mirror
    v = dpp_add(v, std::integral_constant<int, 0x140>{});    // row_mirror
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return ((r0 + r1) + r2) + r3;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// The pixels a wave visits in the reduction passes.  W > 0: the pixels are rows of W and a wave walks DOWN a 16-pixel-wide
// column strip in steps of 16 x 4 pixel blocks (label maps are coherent in both directions; a compact block crosses far
// fewer class boundaries than the same pixels taken along one row).  W == 0: a linear walk.  This lane's pixel of step s is
// first + s*stride (s < nsteps, real for s < nvalid).  The passes keep per-thread partial results while all 64 pixels of a
// step share one class and reduce + publish them (wave butterfly, one LDS integer atomic per entry) only when the class
// changes or the walk ends; steps whose pixels are not all of one class take the per-lane atomic path.
// The J plane values of one pixel, loaded UNCONDITIONALLY (a padded slot re-reads slot J-1, a lane without a pixel reads
// pixel `px_safe`): a load under a per-lane condition becomes its own basic block with a full s_waitcnt in front of
// its use, i.e. one exposed memory round trip per slot instead of one per pixel step (measured: 10 round trips per
// step made the pair-sum pass 170 us for 84 MB).
template <int JT>
__device__ __forceinline__ void spx_kld_load_planes(float (&raw)[JT], const float* __restrict__ v, int J, int HW, int px_safe) {
#pragma unroll
    for (int j = 0; j < JT; ++j) raw[j] = v[(size_t)min(j, J - 1) * HW + px_safe];
}

struct SpxKldWalk {
    int first, stride, nsteps, nvalid;
};
__device__ __forceinline__ SpxKldWalk spx_kld_walk(int HW, int W, int trows, int lane, int wave) {
    SpxKldWalk w;
    if (W > 0) {
        // a step of a wave = a 16-column x 4-row block, the wave walks DOWN its 16-column strip (trows rows: 64 = 16 steps on
        // large maps, fewer on small ones so that the chip fills), the four waves of a workgroup sit side by side.  A compact block lies inside ONE label region far more often
        // than a 64 x 1 row segment does (a 64-pixel row of a map with 16-pixel regions is never of one class; 613 us -> see
        // profiles/EXPERIMENTS.md for the pair pass at 2 Mpx), and every load instruction still moves four whole 64-B pieces.
        const int tiles_x = (W + SPX_KLD_TILE - 1) / SPX_KLD_TILE, H = HW / W;
        const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
        const int col = tx * SPX_KLD_TILE + wave * 16 + (lane & 15), row = ty * trows + (lane >> 4);
        const int hend = min(H, (ty + 1) * trows);
        w.nsteps = (hend - ty * trows + 3) / 4;
        if (tx * SPX_KLD_TILE + wave * 16 >= W) w.nsteps = 0;             // wave-uniform
        w.first = row * W + col;
        w.stride = 4 * W;
        w.nvalid = (col < W && row < hend) ? (hend - row + 3) / 4 : 0;
    } else {
        constexpr int PX_PER_WAVE = SPX_KLD_PX_PER_WG / (SPX_KLD_THREADS / 64);
        const int px0 = blockIdx.x * SPX_KLD_PX_PER_WG + wave * PX_PER_WAVE;
        w.nsteps = px0 < HW ? min(PX_PER_WAVE / 64, (HW - px0 + 63) / 64) : 0;
        w.first = px0 + lane;
        w.stride = 64;
        w.nvalid = w.first < HW ? (HW - w.first + 63) / 64 : 0;
    }
    return w;
}

// One step's inputs of a lane: its pixel's label and J plane values.  Fetched UNCONDITIONALLY (a lane without a pixel at that
// step reads pixel 0 and ignores it) and ONE STEP AHEAD: the label and the planes leave together, and the next step's round
// trip runs under this step's arithmetic - a wave's walk was a chain of two dependent round trips per step (label, then
// planes) with nothing else to issue.
template <int JT>
struct SpxKldStep {
    int c;
    float d[JT];
};
template <int JT>
__device__ __forceinline__ void spx_kld_fetch(SpxKldStep<JT>& o, const float* __restrict__ v, const int32_t* __restrict__ lab,
                                              const SpxKldWalk& w, int step, int J, int HW) {
    const int px = step < w.nvalid ? w.first + step * w.stride : 0;
    o.c = lab[px];
    spx_kld_load_planes(o.d, v, J, HW, px);
}

// pass 0: smax_key[b][c][j] = max over the segment's pixels of vals (ordered-uint key of the float)
__global__ __launch_bounds__(SPX_KLD_THREADS) void spx_kld_max_kernel(const float* __restrict__ vals, const int32_t* __restrict__ labels,
                                                                     int J, int HW, int W, int trows, int K, unsigned int* __restrict__ smax_key,
                                                                     unsigned int* __restrict__ counts, unsigned int* __restrict__ range_keys) {
    extern __shared__ unsigned long long kld_smem[];
    unsigned int* tab = (unsigned int*)kld_smem;          // [K][J]
    unsigned int* cnt = tab + K * J;                      // [K] pixels of the class in this workgroup's range
    unsigned int* rng = cnt + K;                          // [2] keys of max(v) and max(-v) over every value that enters a segment
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < K * J + K + 2; i += SPX_KLD_THREADS) tab[i] = 0u;
    float vmx = -3.0e38f, vmn = 3.0e38f;
    __syncthreads();
    const float* v = vals + (size_t)b * J * HW;
    const int32_t* lab = labels + (size_t)b * HW;
    const SpxKldWalk w = spx_kld_walk(HW, W, trows, lane, wave);
    float m[SPX_KLD_MAXJ];
#pragma unroll
    for (int j = 0; j < SPX_KLD_MAXJ; ++j) m[j] = -3.0e38f;
    int cur = -1;                                          // class of the running maxima (wave-uniform)
    unsigned int run = 0;                                  // its pixels so far
    auto publish = [&]() {
        if (cur < 0) return;
#pragma unroll
        for (int j = 0; j < SPX_KLD_MAXJ; ++j)
            if (j < J) {
                const float wm = wave_max_f32(m[j]);
                if (lane == 0) atomicMax(&tab[cur * J + j], float_key(wm));
                m[j] = -3.0e38f;
            }
        if (lane == 0) atomicAdd(&cnt[cur], run);
        run = 0;
    };
    SpxKldStep<SPX_KLD_MAXJ> nx;
    spx_kld_fetch(nx, v, lab, w, 0, J, HW);
    for (int step = 0; step < w.nsteps; ++step) {
        const SpxKldStep<SPX_KLD_MAXJ> cs = nx;
        spx_kld_fetch(nx, v, lab, w, step + 1, J, HW);
        const int c = step < w.nvalid ? cs.c : -1;
        const bool ok = c >= 0 && c < K;
        // lanes without a class (void pixels, lanes past the map) contribute neutral values either way: a step is uniform when
        // the lanes that HAVE a class agree on it (void borders and ragged tile edges do not send it down the per-lane path)
        const unsigned long long okm = __builtin_amdgcn_ballot_w64(ok);
        const int c0 = okm ? __builtin_amdgcn_readlane(c, __builtin_ffsll((long long)okm) - 1) : -1;
        const bool uniform = __builtin_amdgcn_ballot_w64(ok && c != c0) == 0;
        float d[SPX_KLD_MAXJ];
#pragma unroll
        for (int j = 0; j < SPX_KLD_MAXJ; ++j) d[j] = cs.d[j];
        if (range_keys) {
#pragma unroll
            for (int j = 0; j < SPX_KLD_MAXJ; ++j) vmn = fminf(vmn, (ok && j < J) ? d[j] : 3.0e38f);
        }
#pragma unroll
        for (int j = 0; j < SPX_KLD_MAXJ; ++j) d[j] = (ok && j < J) ? d[j] : -3.0e38f;
        if (range_keys) {
#pragma unroll
            for (int j = 0; j < SPX_KLD_MAXJ; ++j) vmx = fmaxf(vmx, d[j]);
        }
        if (uniform) {
            if (okm == 0) continue;                        // a step without a class pixel
            if (c0 != cur) {
                publish();
                cur = c0;
            }
            run += (unsigned)__builtin_popcountll(okm);
#pragma unroll
            for (int j = 0; j < SPX_KLD_MAXJ; ++j) m[j] = fmaxf(m[j], d[j]);
        } else if (ok) {
            atomicAdd(&cnt[c], 1u);
#pragma unroll
            for (int j = 0; j < SPX_KLD_MAXJ; ++j)
                if (j < J) atomicMax(&tab[c * J + j], float_key(d[j]));
        }
    }
    publish();
    if (range_keys) {
        const float wmx = wave_max_f32(vmx), wmn = -wave_max_f32(-vmn);
        if (lane == 0 && wmx >= wmn) {                     // (a wave without a class pixel leaves max < min)
            atomicMax(&rng[0], float_key(wmx));
            atomicMax(&rng[1], float_key(-wmn));
        }
    }
    __syncthreads();
    for (int i = tid; i < K * J; i += SPX_KLD_THREADS)
        if (tab[i]) atomicMax(&smax_key[(size_t)b * K * J + i], tab[i]);
    if (range_keys && tid < 2 && rng[tid]) atomicMax(&range_keys[tid], rng[tid]);
    if (counts)
        for (int i = tid; i < K; i += SPX_KLD_THREADS)
            if (cnt[i]) atomicAdd(&counts[(size_t)b * K + i], cnt[i]);
}

// pass 1: ssum_fx[b][c][j] = sum exp(d - smax) in 2^40 fixed point (every term is in (0, 1], the maximum contributes 1)
__global__ __launch_bounds__(SPX_KLD_THREADS) void spx_kld_sumexp_kernel(const float* __restrict__ vals, const int32_t* __restrict__ labels,
                                                                        int J, int HW, int W, int trows, int K, const unsigned int* __restrict__ smax_key,
                                                                        unsigned long long* __restrict__ ssum_fx) {
    extern __shared__ unsigned long long kld_smem[];
    unsigned long long* tab = kld_smem;                    // [K][J]
    float* sm = (float*)(tab + K * J);                     // [K][J] segment maxima (a pixel of the segment exists => its key is set)
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < K * J; i += SPX_KLD_THREADS) {
        tab[i] = 0ull;
        sm[i] = key_float(smax_key[(size_t)b * K * J + i]);
    }
    __syncthreads();
    const float* v = vals + (size_t)b * J * HW;
    const int32_t* lab = labels + (size_t)b * HW;
    const double FX = 1099511627776.0;                     // 2^40
    const SpxKldWalk w = spx_kld_walk(HW, W, trows, lane, wave);
    float acc[SPX_KLD_MAXJ];                               // <= 16 terms (steps of the walk) of (0, 1] each: fp32 is ample
#pragma unroll
    for (int j = 0; j < SPX_KLD_MAXJ; ++j) acc[j] = 0.0f;
    int cur = -1;
    auto publish = [&]() {
        if (cur < 0) return;
#pragma unroll
        for (int j = 0; j < SPX_KLD_MAXJ; ++j)
            if (j < J) {
                const double s = wave_sum_f64((double)acc[j]);
                if (lane == 0) atomicAdd(&tab[cur * J + j], (unsigned long long)(s * FX + 0.5));
                acc[j] = 0.0f;
            }
    };
    SpxKldStep<SPX_KLD_MAXJ> nx;
    spx_kld_fetch(nx, v, lab, w, 0, J, HW);
    for (int step = 0; step < w.nsteps; ++step) {
        const SpxKldStep<SPX_KLD_MAXJ> cs = nx;
        spx_kld_fetch(nx, v, lab, w, step + 1, J, HW);
        const int c = step < w.nvalid ? cs.c : -1;
        const bool ok = c >= 0 && c < K;
        // lanes without a class (void pixels, lanes past the map) contribute neutral values either way: a step is uniform when
        // the lanes that HAVE a class agree on it (void borders and ragged tile edges do not send it down the per-lane path)
        const unsigned long long okm = __builtin_amdgcn_ballot_w64(ok);
        const int c0 = okm ? __builtin_amdgcn_readlane(c, __builtin_ffsll((long long)okm) - 1) : -1;
        const bool uniform = __builtin_amdgcn_ballot_w64(ok && c != c0) == 0;
        float e[SPX_KLD_MAXJ];
#pragma unroll
        for (int j = 0; j < SPX_KLD_MAXJ; ++j) e[j] = cs.d[j];
        const float* smc = sm + (ok ? c : 0) * J;
#pragma unroll
        for (int j = 0; j < SPX_KLD_MAXJ; ++j) e[j] = (ok && j < J) ? __expf(e[j] - smc[min(j, J - 1)]) : 0.0f;
        if (uniform) {
            if (okm == 0) continue;
            if (c0 != cur) {
                publish();
                cur = c0;
            }
#pragma unroll
            for (int j = 0; j < SPX_KLD_MAXJ; ++j) acc[j] += e[j];
        } else if (ok) {
#pragma unroll
            for (int j = 0; j < SPX_KLD_MAXJ; ++j)
                if (j < J) atomicAdd(&tab[c * J + j], (unsigned long long)((double)e[j] * FX + 0.5));
        }
    }
    publish();
    __syncthreads();
    for (int i = tid; i < K * J; i += SPX_KLD_THREADS)
        if (tab[i]) atomicAdd(&ssum_fx[(size_t)b * K * J + i], tab[i]);
}

// pass 2: A_fx[b][c][j][k] = sum_px p_j * (l_k - l_j) in signed fixed point (scale given by the caller from the pixel
// count): the Gram matrix MINUS its row's diagonal entry, i.e. -KL(j || k) of the segment.  The loss and the gradient
// only ever use A[j][k] - A[j][j] (rows of dLoss/dA sum to zero), and in this form the sums are small numbers built
// from small terms instead of differences of large ones, so a thread can carry them in fp32 registers across its
// pixels (see SpxKldWalk).
template <int JT>
__global__ __launch_bounds__(SPX_KLD_THREADS) void spx_kld_pairs_kernel(const float* __restrict__ vals, const int32_t* __restrict__ labels,
                                                                       int J, int HW, int W, int trows, int Kall, int KB, const float* __restrict__ lse,
                                                                       const double* __restrict__ scale_p, unsigned long long* __restrict__ A_fx) {
    extern __shared__ unsigned long long kld_smem[];
    // the tables cover the class block [c_lo, c_lo + K) of blockIdx.z (one block when all Kall classes fit the LDS; the
    // reference's ADE / COCO banks - 150 / 182 classes x 12 slots - take 3 - 4): a pixel of another block's class is a pixel
    // without a class here, and every segment is summed by exactly one block
    const int c_lo = blockIdx.z * KB, K = min(KB, Kall - c_lo);
    unsigned long long* tab = kld_smem;                    // [K][J][J], two's complement
    float* ls = (float*)(tab + KB * J * J);                // [K][J]
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double scale = *scale_p;                         // device-side: the host never reads the data (no sync)
    for (int i = tid; i < K * J * J; i += SPX_KLD_THREADS) tab[i] = 0ull;
    for (int i = tid; i < K * J; i += SPX_KLD_THREADS) ls[i] = lse[((size_t)b * Kall + c_lo) * J + i];
    __syncthreads();
    const float* v = vals + (size_t)b * J * HW;
    const int32_t* lab = labels + (size_t)b * HW;
    const SpxKldWalk w = spx_kld_walk(HW, W, trows, lane, wave);
    float acc[JT][JT];
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int k = 0; k < JT; ++k) acc[j][k] = 0.0f;
    int cur = -1;                                          // class the accumulators belong to (wave-uniform)
    auto publish = [&]() {
        if (cur < 0) return;
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int k = 0; k < JT; ++k)
                if (j != k && j < J && k < J) {
                    const double s = (double)wave_sum_f32(acc[j][k]);
                    if (lane == 0) atomicAdd(&tab[(cur * J + j) * J + k], (unsigned long long)(long long)llrint(s * scale));
                    acc[j][k] = 0.0f;
                }
    };
    SpxKldStep<JT> nx;
    spx_kld_fetch(nx, v, lab, w, 0, J, HW);
    for (int step = 0; step < w.nsteps; ++step) {
        const SpxKldStep<JT> cs = nx;
        spx_kld_fetch(nx, v, lab, w, step + 1, J, HW);
        const int c = step < w.nvalid ? cs.c - c_lo : -1;
        const bool ok = c >= 0 && c < K;
        // lanes without a class (void pixels, lanes past the map) contribute neutral values either way: a step is uniform when
        // the lanes that HAVE a class agree on it (void borders and ragged tile edges do not send it down the per-lane path)
        const unsigned long long okm = __builtin_amdgcn_ballot_w64(ok);
        const int c0 = okm ? __builtin_amdgcn_readlane(c, __builtin_ffsll((long long)okm) - 1) : -1;
        const bool uniform = __builtin_amdgcn_ballot_w64(ok && c != c0) == 0;
        float l[JT], p[JT];          // JT = J rounded up to a multiple of 4: static indices, padded slots contribute 0
#pragma unroll
        for (int j = 0; j < JT; ++j) l[j] = cs.d[j];
        const float* lsc = ls + (ok ? c : 0) * J;
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            l[j] = (ok && j < J) ? l[j] - lsc[min(j, J - 1)] : 0.0f;
            p[j] = (ok && j < J) ? __expf(l[j]) : 0.0f;
        }
        if (uniform) {
            if (okm == 0) continue;                        // a step without a class pixel
            if (c0 != cur) {
                publish();
                cur = c0;
            }
#pragma unroll
            for (int j = 0; j < JT; ++j)
#pragma unroll
                for (int k = 0; k < JT; ++k)
                    if (j != k) acc[j][k] = fmaf(p[j], l[k] - l[j], acc[j][k]);
        } else if (ok) {
#pragma unroll
            for (int j = 0; j < JT; ++j)
#pragma unroll
                for (int k = 0; k < JT; ++k)
                    if (j != k && j < J && k < J) {
                        const double t = (double)p[j] * (double)(l[k] - l[j]);
                        atomicAdd(&tab[(c * J + j) * J + k], (unsigned long long)(long long)llrint(t * scale));
                    }
        }
    }
    publish();
    __syncthreads();
    for (int i = tid; i < K * J * J; i += SPX_KLD_THREADS)
        if (tab[i]) atomicAdd(&A_fx[((size_t)b * Kall + c_lo) * J * J + i], tab[i]);
}

// pass 3: gradient.  With Cf = dLoss/d(Gram) (per segment; built below from the caller's dLoss/dA) and sum_px p_j = 1:
//   dLoss/dd_m(px) = p_m * [ sum_k Cf[m][k] (l_k - A[m][k]) - sum_j Cf[j][m] ] + sum_j Cf[j][m] p_j
// - per pixel, given the segment's A and Cf: no reduction.
template <int JT>
__global__ __launch_bounds__(SPX_KLD_THREADS) void spx_kld_backward_kernel(const float* __restrict__ vals, const int32_t* __restrict__ labels,
                                                                          int J, int HW, int Kall, int KB, const float* __restrict__ lse,
                                                                          const float* __restrict__ A, const float* __restrict__ Cf,
                                                                          const float* __restrict__ cf_scale, int ppw,
                                                                          float* __restrict__ grad) {
    extern __shared__ unsigned long long kld_smem[];
    // class block of blockIdx.z (see spx_kld_pairs_kernel): this workgroup writes the gradient of the pixels whose class lies
    // in its block; block 0 also the zeros of the pixels without a class
    const int c_lo = blockIdx.z * KB, K = min(KB, Kall - c_lo);
    // tables with rows padded to JT floats (16-B aligned: a pixel reads a row with ds_read_b128; lanes of one class broadcast)
    float* sC = (float*)kld_smem;                          // [K][JT][JT]  Cf, diagonal fixed below
    float* sT = sC + KB * JT * JT;                         // [K][JT][JT]  first A, then Cf transposed
    float* sL = sT + KB * JT * JT;                         // [K][JT]      lse
    float* sR = sL + KB * JT;                              // [K][JT]      sum_j Cf[j][m] + sum_k Cf[m][k] A[m][k]
    const int b = blockIdx.y, tid = threadIdx.x;
    const float cs = cf_scale ? *cf_scale : 1.0f;          // Cf may arrive unnormalised with its factor in device memory
    for (int i = tid; i < K * JT * JT; i += SPX_KLD_THREADS) {
        const int c = i / (JT * JT), m = (i / JT) % JT, k = i % JT;
        const bool in = m < J && k < J;
        const size_t src = ((size_t)b * Kall + c_lo) * J * J + ((size_t)c * J + m) * J + k;
        sT[i] = in ? A[src] : 0.0f;
        sC[i] = in ? Cf[src] * cs : 0.0f;
    }
    for (int i = tid; i < K * JT; i += SPX_KLD_THREADS) sL[i] = (i % JT) < J ? lse[((size_t)b * Kall + c_lo) * J + (i / JT) * J + (i % JT)] : 0.0f;
    __syncthreads();
    // A holds A[j][k] - A[j][j] (pass 2), so dLoss/d(Gram) has the off-diagonal entries of Cf and zero row sums;
    // with zero row sums the A[m][m] offset drops out of sum_k Cf[m][k] (l_k - A[m][k]) below
    for (int i = tid; i < K * JT; i += SPX_KLD_THREADS) {
        const int c = i / JT, m = i % JT;
        float sum = 0.0f;
        for (int k = 0; k < J; ++k)
            if (k != m) sum += sC[(c * JT + m) * JT + k];
        if (m < J) sC[(c * JT + m) * JT + m] = -sum;
    }
    __syncthreads();
    // per (class, m): everything of the gradient that does not depend on the pixel
    for (int i = tid; i < K * JT; i += SPX_KLD_THREADS) {
        const int c = i / JT, m = i % JT;
        float sum = 0.0f;
        for (int j = 0; j < J; ++j) sum += sC[(c * JT + j) * JT + m];
        for (int k = 0; k < J; ++k) sum += sC[(c * JT + m) * JT + k] * sT[(c * JT + m) * JT + k];
        sR[i] = sum;
    }
    __syncthreads();
    for (int i = tid; i < K * JT * JT; i += SPX_KLD_THREADS) {
        const int c = i / (JT * JT), m = (i / JT) % JT, k = i % JT;
        sT[i] = sC[(c * JT + k) * JT + m];                 // Cf^T: the second dot product reads rows too
    }
    __syncthreads();
    const float* v = vals + (size_t)b * J * HW;
    float* g = grad + (size_t)b * J * HW;
    const int32_t* lab = labels + (size_t)b * HW;
    const int px_end = min(HW, (int)(blockIdx.x + 1) * ppw);
    // label and planes of the NEXT pixel of the thread are in flight while this one is worked on (see SpxKldStep)
    int px = blockIdx.x * ppw + tid;
    int craw_n = lab[px < px_end ? px : 0];
    float l_n[JT];
    spx_kld_load_planes(l_n, v, J, HW, px < px_end ? px : 0);
    for (; px < px_end; px += SPX_KLD_THREADS) {
        const int craw = craw_n, c = craw - c_lo;
        float l[JT], p[JT];
#pragma unroll
        for (int j = 0; j < JT; ++j) l[j] = l_n[j];
        {
            const int pn = px + SPX_KLD_THREADS < px_end ? px + SPX_KLD_THREADS : 0;
            craw_n = lab[pn];
            spx_kld_load_planes(l_n, v, J, HW, pn);
        }
        const bool ok = c >= 0 && c < K;
        if (!ok && !(blockIdx.z == 0 && (craw < 0 || craw >= Kall))) continue;      // another block's pixel
        const float* slc = sL + (ok ? c : 0) * JT;
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            l[j] = (ok && j < J) ? l[j] - slc[j] : 0.0f;
            p[j] = (ok && j < J) ? __expf(l[j]) : 0.0f;
        }
        const int cb = (ok ? c : 0) * JT;
#pragma unroll
        for (int m = 0; m < JT; ++m) {
            if (m < J) {
                // dLoss/dd_m = p_m (sum_k Cf[m][k] l_k - R[m]) + sum_k Cf[k][m] p_k      (padded entries are zero)
                float s1 = -sR[cb + m], s2 = 0.0f;
#pragma unroll
                for (int k4 = 0; k4 < JT / 4; ++k4) {
                    const f32x4 cr = *(const f32x4*)(sC + (cb + m) * JT + 4 * k4);
                    const f32x4 ct = *(const f32x4*)(sT + (cb + m) * JT + 4 * k4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s1 = fmaf(cr[e], l[4 * k4 + e], s1);
                        s2 = fmaf(ct[e], p[4 * k4 + e], s2);
                    }
                }
                g[(size_t)m * HW + px] = ok ? fmaf(p[m], s1, s2) : 0.0f;
            }
        }
    }
}

// lse[i] = smax + log(sum exp(v - smax)) of segment slot i from passes 0 and 1; 0 where the segment has no pixel
__global__ void spx_kld_lse_kernel(const unsigned int* __restrict__ smax_key, const unsigned long long* __restrict__ ssum_fx, int n,
                                   float* __restrict__ lse, const unsigned int* __restrict__ range_keys, int HW,
                                   double* __restrict__ scale_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && scale_out) {
        // fixed-point scale of the pair sums: a power of two such that HW terms of size <= 2 (value range) + 32 stay inside int64
        double span = 32.0;
        if (range_keys[0] && range_keys[1]) span += (double)(key_float(range_keys[0]) + key_float(range_keys[1]));   // max - min, in fp32
        *scale_out = exp2(floor(log2(2305843009213693952.0 / ((double)HW * span))));
    }
    if (i >= n) return;
    const unsigned int k = smax_key[i];
    const double s = (double)ssum_fx[i] * (1.0 / 1099511627776.0);
    lse[i] = (k != 0u && s > 0.0) ? (float)((double)key_float(k) + log(s)) : 0.0f;
}
hipError_t spx_launch_kld_lse(const uint32_t* keys, const uint64_t* ssum_fx, int n, float* lse, const uint32_t* range_keys, int HW,
                              double* scale_out, hipStream_t s) {
    hipLaunchKernelGGL(spx_kld_lse_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, keys, (const unsigned long long*)ssum_fx, n, lse,
                       range_keys, HW, scale_out);
    return hipGetLastError();
}

// The [B, K, J, J]-sized algebra between the pair sums and the gradient pass (loss.py:113-142).  One workgroup per segment:
//   A = a_fx / scale;  kld_jk = (A_jj + A_kk - A_jk - A_kj) / 2;  valid = pair_ok[class][j][k] and the segment has >= 2 pixels;
//   e = valid * exp(-kld);  Cf' = (es - diag(rowsum es)) / 2 with es = e + e^T  (dLoss/dA WITHOUT the 1/n of the mean);
//   part[seg] = (sum e, number of valid entries).
// A second, one-workgroup kernel sums the parts in a fixed order: loss = sum e / n (0 when n = 0) and inv = 1 / max(n, 1), the
// factor the gradient pass applies to Cf'.  (One workgroup for everything took 222 us for 10 crops x 19 classes x 12 x 12.)
__global__ __launch_bounds__(256) void spx_kld_gram_kernel(const long long* __restrict__ a_fx, const double* __restrict__ scale,
                                                           const unsigned int* __restrict__ counts,
                                                           const unsigned char* __restrict__ pair_ok, int K, int J,
                                                           float* __restrict__ A, float* __restrict__ Cf, double* __restrict__ part) {
    __shared__ float sA[256], sE[256];
    __shared__ double ssum[256], scnt[256];
    const int tid = threadIdx.x, seg = blockIdx.x, JJ = J * J;
    const int j = tid / J, k = tid - j * J;
    const bool in = tid < JJ;
    const double inv_scale = 1.0 / *scale;
    float a = 0.0f;
    if (in) {
        a = (float)((double)a_fx[(size_t)seg * JJ + tid] * inv_scale);
        A[(size_t)seg * JJ + tid] = a;
    }
    sA[tid] = a;
    __syncthreads();
    float e = 0.0f;
    bool valid = false;
    if (in) {
        valid = pair_ok[(size_t)(seg % K) * JJ + tid] && counts[seg] >= 2u;
        const float kld = 0.5f * (((sA[j * J + j] + sA[k * J + k]) - sA[j * J + k]) - sA[k * J + j]);
        e = valid ? expf(-kld) : 0.0f;
    }
    sE[tid] = e;
    ssum[tid] = (double)e;
    scnt[tid] = valid ? 1.0 : 0.0;
    __syncthreads();
    if (in) {
        float v = sE[j * J + k] + sE[k * J + j];
        if (j == k) {
            float row = 0.0f;
            for (int m = 0; m < J; ++m) row += sE[j * J + m] + sE[m * J + j];
            v -= row;
        }
        Cf[(size_t)seg * JJ + tid] = 0.5f * v;
    }
    for (int m = 128; m >= 1; m >>= 1) {
        if (tid < m) {
            ssum[tid] += ssum[tid + m];
            scnt[tid] += scnt[tid + m];
        }
        __syncthreads();
    }
    if (tid == 0) {
        part[2 * seg] = ssum[0];
        part[2 * seg + 1] = scnt[0];
    }
}
__global__ __launch_bounds__(256) void spx_kld_gram_finish_kernel(const double* __restrict__ part, int nseg, float* __restrict__ loss) {
    __shared__ double ssum[256], scnt[256];
    const int tid = threadIdx.x;
    double se = 0.0, sn = 0.0;
    for (int i = tid; i < nseg; i += 256) {
        se += part[2 * i];
        sn += part[2 * i + 1];
    }
    ssum[tid] = se;
    scnt[tid] = sn;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if (tid < m) {
            ssum[tid] += ssum[tid + m];
            scnt[tid] += scnt[tid + m];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float inv = 1.0f / (float)(scnt[0] < 1.0 ? 1.0 : scnt[0]);
        loss[0] = (float)ssum[0] * inv;
        loss[1] = inv;
    }
}
hipError_t spx_launch_kld_gram_loss(const int64_t* a_fx, const double* scale, const uint32_t* counts, const uint8_t* pair_ok, int nseg, int K,
                                    int J, float* A, float* Cf, double* part, float* loss, hipStream_t s) {
    hipLaunchKernelGGL(spx_kld_gram_kernel, dim3((unsigned)nseg), dim3(256), 0, s, (const long long*)a_fx, scale, counts, pair_ok, K, J, A, Cf, part);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(spx_kld_gram_finish_kernel, dim3(1), dim3(256), 0, s, (const double*)part, nseg, loss);
    return hipGetLastError();
}

// (pass 0: t0 = counts, t1 = range keys)
hipError_t spx_launch_kld(int pass, const float* vals, const int32_t* labels, int B, int J, int HW, int W, int K, const float* t0,
                          const float* t1, const float* t2, const double* scale, void* out, hipStream_t s, const float* cf_scale) {
    dim3 grid((unsigned)((HW + SPX_KLD_PX_PER_WG - 1) / SPX_KLD_PX_PER_WG), (unsigned)B);
    dim3 blk(SPX_KLD_THREADS);
    // rows of a workgroup's tile: 64 (16 steps per wave) on large maps; on small ones (training crops) 32 or 16, so that there
    // are enough workgroups to fill the chip.  (Two workgroups per CU are enough since the passes fetch a step ahead: at 2 Mpx
    // 64-row tiles = 512 workgroups run the max / sum-exp / pair passes in 27 / 22 / 39 us, 32-row tiles = 1024 in 39 / 25 / 52 -
    // half the class-run publishes per pixel; 128-row tiles = 256 workgroups in 35 / 38 / 59.)
    int trows = SPX_KLD_TILE;
    if (W > 0 && pass != 3) {
        const int tiles_x = (W + SPX_KLD_TILE - 1) / SPX_KLD_TILE, H = HW / W;
        while (trows > 16 && (long long)B * tiles_x * ((H + trows - 1) / trows) < SPX_KLD_MIN_WGS) trows >>= 1;
        grid.x = (unsigned)(tiles_x * ((H + trows - 1) / trows));
    }
    if (pass == 0)
        hipLaunchKernelGGL(spx_kld_max_kernel, grid, blk, (size_t)(K * J + K + 2) * 4, s, vals, labels, J, HW, W, trows, K, (unsigned int*)out, (unsigned int*)t0, (unsigned int*)t1);
    else if (pass == 1)
        hipLaunchKernelGGL(spx_kld_sumexp_kernel, grid, blk, (size_t)K * J * 12, s, vals, labels, J, HW, W, trows, K, (const unsigned int*)t0, (unsigned long long*)out);
    else if (pass == 2) {
        // class blocks (grid.z) so that a block's tables fit SPX_KLD_TABLE_LDS
        const size_t per_class = (size_t)J * J * 8 + (size_t)J * 4;
        const int KB = (int)std::min<size_t>((size_t)K, std::max<size_t>(1, SPX_KLD_TABLE_LDS / per_class));
        grid.z = (unsigned)((K + KB - 1) / KB);
        const size_t lds = (size_t)KB * per_class;
        unsigned long long* o = (unsigned long long*)out;
        if (J <= 4) hipLaunchKernelGGL(spx_kld_pairs_kernel<4>, grid, blk, lds, s, vals, labels, J, HW, W, trows, K, KB, t0, scale, o);
        else if (J <= 8) hipLaunchKernelGGL(spx_kld_pairs_kernel<8>, grid, blk, lds, s, vals, labels, J, HW, W, trows, K, KB, t0, scale, o);
        else if (J <= 12) hipLaunchKernelGGL(spx_kld_pairs_kernel<12>, grid, blk, lds, s, vals, labels, J, HW, W, trows, K, KB, t0, scale, o);
        else hipLaunchKernelGGL(spx_kld_pairs_kernel<16>, grid, blk, lds, s, vals, labels, J, HW, W, trows, K, KB, t0, scale, o);
    } else {
        const int JT = J <= 4 ? 4 : (J <= 8 ? 8 : (J <= 12 ? 12 : 16));
        const size_t per_class = (size_t)(2 * JT * JT + 2 * JT) * 4;
        const int KB = (int)std::min<size_t>((size_t)K, std::max<size_t>(1, SPX_KLD_TABLE_LDS / per_class));
        grid.z = (unsigned)((K + KB - 1) / KB);
        const size_t lds = (size_t)KB * per_class;
        float* o = (float*)out;
        // pixels per workgroup: 2048 on large maps; small maps (training crops) get enough workgroups to fill the chip - a thread
        // then takes one pixel instead of walking eight in sequence behind the table set-up (80 -> ~20 us at 10 x 65 x 65)
        // (2 Mpx, same box: 512 / 1024 / 2048 / 4096 / 8192 pixels per workgroup = 65 / 51 / 45 / 55 / 82 us)
        int ppw = SPX_KLD_PX_PER_WG;
        while (ppw > SPX_KLD_THREADS && (long long)B * ((HW + ppw - 1) / ppw) < 512) ppw >>= 1;
        grid.x = (unsigned)((HW + ppw - 1) / ppw);
        if (J <= 4) hipLaunchKernelGGL(spx_kld_backward_kernel<4>, grid, blk, lds, s, vals, labels, J, HW, K, KB, t0, t1, t2, cf_scale, ppw, o);
        else if (J <= 8) hipLaunchKernelGGL(spx_kld_backward_kernel<8>, grid, blk, lds, s, vals, labels, J, HW, K, KB, t0, t1, t2, cf_scale, ppw, o);
        else if (J <= 12) hipLaunchKernelGGL(spx_kld_backward_kernel<12>, grid, blk, lds, s, vals, labels, J, HW, K, KB, t0, t1, t2, cf_scale, ppw, o);
        else hipLaunchKernelGGL(spx_kld_backward_kernel<16>, grid, blk, lds, s, vals, labels, J, HW, K, KB, t0, t1, t2, cf_scale, ppw, o);
    }
    return hipGetLastError();
}
