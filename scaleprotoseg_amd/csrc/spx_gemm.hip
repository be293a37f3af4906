// Heads wider than the distance kernels carry (more than 160 rows, a grouping tail over more than 32 classes) and the
// head product behind a user-supplied similarity: fp32 products on the [pixel][P] activations,
//     C[i][j] = sum_k A(i, k) . B(j, k)
// with each operand given by (row stride, k stride), one of them 1 - so the three products of a linear layer
// (y = a . w^T, d_a = g . w, d_w = g^T . a: segmentation/model/model_multiscale.py:243-244 and its autograd,
// model_multiscale_group.py:283-308) are ONE kernel family with no transposed copies.
//
// v_mfma_f32_32x32x2_f32: fp32 operands, fp32 accumulate - the products carry no bf16 rounding at all, so these heads
// have the reference's fp32 arithmetic up to summation order.  Workgroup = 4 waves = a 128 x 128 tile of C, wave = 64 x 64
// (2 x 2 MFMA tiles, 64 accumulator registers); k runs in chunks of 16 through two LDS buffers ([row][k] images with an odd
// 17-float row stride: a fragment read `row = lane & 31, k = 2 kk + (lane >> 5)` touches 64 distinct banks); the next
// chunk's global loads are in flight in registers while the current one feeds the matrix pipe (4 MFMAs = 256 cycles per 4
// ds_read_b32).  Long contractions over few output tiles (d_w: k = pixels) are split over workgroups (grid.z) into
// workspace slabs that a second kernel sums in slab order: deterministic, no float atomics.
#include "spx_args.h"
#include "spx_common.h"

#define SPX_G_TK 32          // k ranges of a split contraction are multiples of this (both chunk sizes divide it)


typedef float spx_f4u __attribute__((ext_vector_type(4), aligned(4)));      // rows of [pixel][n] tensors are only 4-B aligned in general

#ifdef SPX_DIAG
static int g_gemm_force_wm = 0, g_gemm_force_splits = 0;      // diagnostic builds only (spx_diag_set_gemm; splits < 0: linear block -> tile map)
#else
static constexpr int g_gemm_force_wm = 0, g_gemm_force_splits = 0;
#endif

struct SpxGemmArgs {
    const float* A; long long ras, kas;
    const float* B; long long rbs, kbs;
    float* C; long long ldc;
    const float* E; long long lde;          // flags & 4: C = acc * exp(E[i][j])
    int M, N, K;
    int flags;                              // 1: A elements enter as exp(A); 2: B elements enter as exp(B)
    int splits, kper;                       // k range per split (multiple of 32)
    int ni, nj, ntiles;                     // tile grid of this launch: ni x nj tiles x splits
    int linear_map;                         // experiments: tile = block index
    float* ws;                              // [splits][M][N] partials when splits > 1
};

// one operand tile: TI rows x TK k into registers (8 floats per thread), then into its LDS image.  The values stay untouched
// in registers until commit() (the exponential of the grouping tail included), so a load can stay in flight across two chunks.
template <bool KCONT, int TI, int TK>
struct SpxGemmStager {
    static constexpr int LD = TK + 1;
    float r[8];
    unsigned valid;     // bit e: element e is inside the operand (others are zero)
    int row, kk;        // KCONT: row of the tile, first of this thread's 8 k;   else: k row and first of 8 tile rows
    __device__ __forceinline__ void init(int tid) {
        if (KCONT) { row = tid / (TK / 8); kk = (tid % (TK / 8)) * 8; }
        else { kk = tid / (TI / 8); row = (tid % (TI / 8)) * 8; }
    }
    __device__ __forceinline__ void load(const float* __restrict__ P, long long rs, long long ks, int r0, int nrows, int k0, int kend) {
        // KCONT: 8 consecutive k of one row;  else: 8 consecutive rows of one k.  (Measured against unconditional 16-B buffer
        // loads with out-of-range predication + masking at commit: those ran 30-40 % slower on every shape.)
        const int gi = r0 + row, k = k0 + kk;
        const float* p = KCONT ? P + (long long)gi * rs + k : P + (long long)k * ks + gi;
        const bool whole = KCONT ? (gi < nrows && k + 8 <= kend) : (k < kend && gi + 8 <= nrows);
        if (whole) {
            const spx_f4u v0 = *(const spx_f4u*)p, v1 = *(const spx_f4u*)(p + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { r[e] = v0[e]; r[4 + e] = v1[e]; }
            valid = 0xFFu;
        } else {
            valid = 0u;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = KCONT ? (gi < nrows && k + e < kend) : (k < kend && gi + e < nrows);
                r[e] = 0.0f;
                if (ok) { r[e] = p[e]; valid |= 1u << e; }
            }
        }
    }
    __device__ __forceinline__ void commit(float* __restrict__ img, bool ex) {
        if (ex) {
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = (valid >> e) & 1u ? expf(r[e]) : 0.0f;
        }
        if (KCONT) {
#pragma unroll
            for (int e = 0; e < 8; ++e) img[row * LD + kk + e] = r[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) img[(row + e) * LD + kk] = r[e];
        }
    }
};

// WM = MFMA tiles per wave and dimension: 2 -> 128 x 128 workgroup tiles (64 x 64 per wave) in k-chunks of 16, 1 -> 64 x 64
// (32 x 32 per wave) in k-chunks of 32: the small tile halves the operand re-use but quarters the work quantum, for shapes
// whose 128-tiles would leave most CUs waiting for a few (8450 x 450: 268 big tiles on 256 CUs = two rounds for 5 % more
// work than one).  Global loads run TWO chunks ahead of the matrix pipe through two register sets (a chunk's MFMAs take
// ~1 us, an HBM round trip under load ~2 us: one chunk of lead left a lone workgroup at half speed).
template <bool AK, bool BK, int WM>
__global__ __launch_bounds__(256) void spx_gemm_kernel(const SpxGemmArgs a) {
    constexpr int TI = 64 * WM, TK = WM == 2 ? 16 : 32, LD = TK + 1;
    __shared__ float As[2][TI * LD];
    __shared__ float Bs[2][TI * LD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wi = wave >> 1, wj = wave & 1;
    // XCD-aware block -> tile map.  Workgroup w runs on XCD w mod 8, each with its own L2.  Tiles are numbered
    // L = (slab * ni + i) * nj + j (tiles that share an A panel - and, in a split contraction, a k slab - are neighbours) and
    // XCD x takes the contiguous range [x * per, (x + 1) * per): an operand panel is then fetched into ONE L2 instead of
    // all eight.  Measured on the ADE / COCO head shapes (tools/probes/gemm_sweep.py, "linear" columns): no difference -
    // their operands (<= 61 MB) sit in the memory-side cache and the kernels are bound by their own issue, not by L2 fills;
    // kept because it costs nothing and is the right map once the operands outgrow that cache.
    const int per = (a.ntiles + 7) / 8;
    const int L = a.linear_map ? (int)blockIdx.x : (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if (L >= a.ntiles || (!a.linear_map && (int)(blockIdx.x >> 3) >= per)) return;
    const int bj = L % a.nj, bi = (L / a.nj) % a.ni, bz = L / (a.nj * a.ni);
    const int i0 = bi * TI, j0 = bj * TI;
    const int kbeg = bz * a.kper;
    const int kend = (kbeg + a.kper < a.K) ? kbeg + a.kper : a.K;
    const bool exa = a.flags & 1, exb = a.flags & 2;

    f32x16 acc[WM][WM];
#pragma unroll
    for (int u = 0; u < WM; ++u)
#pragma unroll
        for (int v = 0; v < WM; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][v][r] = 0.0f;

    SpxGemmStager<AK, TI, TK> sa[2];
    SpxGemmStager<BK, TI, TK> sb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        sa[q].init(tid);
        sb[q].init(tid);
    }
    auto fetch = [&](int q, int k0) {       // (a chunk past the end loads nothing: every element is out of range)
        sa[q].load(a.A, a.ras, a.kas, i0, a.M, k0, kend);
        sb[q].load(a.B, a.rbs, a.kbs, j0, a.N, k0, kend);
    };
    auto commit = [&](int q, int buf) {
        sa[q].commit(As[buf], exa);
        sb[q].commit(Bs[buf], exb);
    };
    auto compute = [&](int buf) {
        const float* ap = As[buf] + (wi * 32 * WM + (lane & 31)) * LD + (lane >> 5);
        const float* bp = Bs[buf] + (wj * 32 * WM + (lane & 31)) * LD + (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < TK / 2; ++kk) {
            float av[WM], bv[WM];
#pragma unroll
            for (int u = 0; u < WM; ++u) {
                av[u] = ap[u * 32 * LD + 2 * kk];
                bv[u] = bp[u * 32 * LD + 2 * kk];
            }
#pragma unroll
            for (int u = 0; u < WM; ++u)
#pragma unroll
                for (int v = 0; v < WM; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[v], acc[u][v], 0, 0, 0);
        }
    };
    // chunk c lives in LDS buffer c & 1; register set q holds chunk c + 1 + q's data on its way there
    fetch(0, kbeg);
    commit(0, 0);
    fetch(0, kbeg + TK);
    fetch(1, kbeg + 2 * TK);
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += 2 * TK) {
        // even chunk: buffer 0 feeds the pipe; set 0 (chunk + 1) goes to buffer 1 and is re-armed with chunk + 3
        commit(0, 1);
        fetch(0, k0 + 3 * TK);
        compute(0);
        __syncthreads();
        if (k0 + TK >= kend) break;
        // odd chunk: buffer 1 feeds the pipe; set 1 (chunk + 2) goes to buffer 0 and is re-armed with chunk + 4
        commit(1, 0);
        fetch(1, k0 + 4 * TK);
        compute(1);
        __syncthreads();
    }

    // accumulator tile: register r of lane l is row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31: a store instruction
    // writes two 128-B row pieces
    float* const out = a.splits > 1 ? a.ws + (size_t)bz * (size_t)a.M * (size_t)a.N : a.C;
    const long long ldo = a.splits > 1 ? a.N : a.ldc;
    const bool mulexp = (a.flags & 4) && a.splits == 1;
#pragma unroll
    for (int u = 0; u < WM; ++u)
#pragma unroll
        for (int v = 0; v < WM; ++v) {
            const int j = j0 + (wj * WM + v) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + (wi * WM + u) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (i < a.M && j < a.N) {
                    float val = acc[u][v][r];
                    if (mulexp) val *= expf(a.E[(long long)i * a.lde + j]);
                    out[(long long)i * ldo + j] = val;
                }
            }
        }
}

// C[i][j] = sum over slabs, in slab order
__global__ __launch_bounds__(256) void spx_gemm_reduce_kernel(const float* __restrict__ ws, int splits, long long MN, int N, float* __restrict__ C, long long ldc) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= MN) return;
    float s = 0.0f;
    for (int k = 0; k < splits; ++k) s += ws[(size_t)k * MN + g];
    C[(g / N) * ldc + (g % N)] = s;
}

// Tile and split policy: a pure function of the shape (results never depend on the machine state).
struct SpxGemmPlan { int wm, splits, kper; };
#ifdef SPX_DIAG
void spx_gemm_force(int wm, int splits) { g_gemm_force_wm = wm; g_gemm_force_splits = splits; }
#endif
static SpxGemmPlan spx_gemm_plan(int M, int N, int K, int flags) {
    if (g_gemm_force_wm) {
        int sfor = g_gemm_force_splits < 0 ? -g_gemm_force_splits : g_gemm_force_splits;
        if (sfor < 1 || (flags & 4)) sfor = 1;
        int kp = ((K + sfor - 1) / sfor + SPX_G_TK - 1) / SPX_G_TK * SPX_G_TK;
        return SpxGemmPlan{g_gemm_force_wm, (K + kp - 1) / kp, kp};
    }
    // Measured on MI355X over the wide-head shapes (tools/probes/gemm_sweep.py): a workgroup alone on a CU runs at about half
    // the matrix pipe's rate (its barriers and LDS commits are exposed), so the launch should put ~8 workgroups on every CU;
    // the 64-tiles beat the 128-tiles until there are several thousand of them, and a contraction is split (slabs summed in
    // order by a second kernel) until that many workgroups exist, keeping >= 512 k per slab.
    const long long t64 = (long long)((M + 63) / 64) * ((N + 63) / 64);
    if (t64 >= 8192) return SpxGemmPlan{2, 1, (K + SPX_G_TK - 1) / SPX_G_TK * SPX_G_TK};
    int sp = 1;
    while (!(flags & 4) && sp < 32 && t64 * sp < 2048 && K / (2 * sp) >= 512) sp *= 2;
    int kp = ((K + sp - 1) / sp + SPX_G_TK - 1) / SPX_G_TK * SPX_G_TK;
    return SpxGemmPlan{1, (K + kp - 1) / kp, kp};
}

size_t spx_gemm_workspace(int M, int N, int K, int flags) {
    const SpxGemmPlan p = spx_gemm_plan(M, N, K, flags);
    return p.splits > 1 ? (size_t)p.splits * M * N * sizeof(float) : 0;
}

template <int WM>
static void spx_launch_gemm_wm(const SpxGemmArgs& a_, bool ak, bool bk, hipStream_t s) {
    constexpr int TI = 64 * WM;
    SpxGemmArgs a = a_;
    a.ni = (a.M + TI - 1) / TI;
    a.nj = (a.N + TI - 1) / TI;
    a.ntiles = a.ni * a.nj * a.splits;
    a.linear_map = g_gemm_force_splits < 0;
    const dim3 grid((unsigned)(((a.ntiles + 7) / 8) * 8));
    if (ak && bk) hipLaunchKernelGGL((spx_gemm_kernel<true, true, WM>), grid, dim3(256), 0, s, a);
    else if (ak) hipLaunchKernelGGL((spx_gemm_kernel<true, false, WM>), grid, dim3(256), 0, s, a);
    else if (bk) hipLaunchKernelGGL((spx_gemm_kernel<false, true, WM>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((spx_gemm_kernel<false, false, WM>), grid, dim3(256), 0, s, a);
}

hipError_t spx_launch_gemm(const float* A, long long ras, long long kas, const float* B, long long rbs, long long kbs,
                           float* C, long long ldc, int M, int N, int K, int flags, const float* E, long long lde,
                           float* ws, hipStream_t s) {
    SpxGemmArgs a;
    a.A = A; a.ras = ras; a.kas = kas;
    a.B = B; a.rbs = rbs; a.kbs = kbs;
    a.C = C; a.ldc = ldc;
    a.E = E; a.lde = lde;
    a.M = M; a.N = N; a.K = K;
    a.flags = flags;
    const SpxGemmPlan p = spx_gemm_plan(M, N, K, flags);
    a.splits = p.splits;
    a.kper = p.kper;
    a.ws = ws;
    const bool ak = kas == 1, bk = kbs == 1;
    if (p.wm == 2) spx_launch_gemm_wm<2>(a, ak, bk, s);
    else spx_launch_gemm_wm<1>(a, ak, bk, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (a.splits > 1) {
        const long long MN = (long long)M * N;
        hipLaunchKernelGGL(spx_gemm_reduce_kernel, dim3((unsigned)((MN + 255) / 256)), dim3(256), 0, s, (const float*)ws, a.splits, MN, N, C, ldc);
        e = hipGetLastError();
    }
    return e;
}
