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

#define SPX_G_TI 128
#define SPX_G_TK 16
#define SPX_G_LD 17

typedef float spx_f4u __attribute__((ext_vector_type(4), aligned(4)));      // rows of [pixel][n] tensors are only 4-B aligned in general

struct SpxGemmArgs {
    const float* A; long long ras, kas;
    const float* B; long long rbs, kbs;
    float* C; long long ldc;
    const float* E; long long lde;          // flags & 4: C = acc * exp(E[i][j])
    int M, N, K;
    int flags;                              // 1: A elements enter as exp(A); 2: B elements enter as exp(B)
    int splits, kper;                       // k range per split (multiple of 16)
    float* ws;                              // [splits][M][N] partials when splits > 1
};

// one operand tile: 128 rows x 16 k into registers (8 floats per thread), then into its LDS image
template <bool KCONT>
struct SpxGemmStager {
    float r[8];
    int row, kk;        // KCONT: row of the tile, first of this thread's 8 k;   else: k row (0..15) and first of 8 tile rows
    __device__ __forceinline__ void init(int tid) {
        if (KCONT) { row = tid >> 1; kk = (tid & 1) * 8; }
        else { kk = tid >> 4; row = (tid & 15) * 8; }
    }
    __device__ __forceinline__ void load(const float* __restrict__ P, long long rs, long long ks, int r0, int nrows, int k0, int kend, bool ex) {
        if (KCONT) {
            const int gi = r0 + row, k = k0 + kk;
            const float* p = P + (long long)gi * rs + k;
            if (gi < nrows && k + 8 <= kend) {
                const spx_f4u v0 = *(const spx_f4u*)p, v1 = *(const spx_f4u*)(p + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { r[e] = v0[e]; r[4 + e] = v1[e]; }
                if (ex) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) r[e] = expf(r[e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = 0.0f;
                    if (gi < nrows && k + e < kend) { v = p[e]; if (ex) v = expf(v); }
                    r[e] = v;
                }
            }
        } else {
            const int k = k0 + kk, gi = r0 + row;
            const float* p = P + (long long)k * ks + gi;
            if (k < kend && gi + 8 <= nrows) {
                const spx_f4u v0 = *(const spx_f4u*)p, v1 = *(const spx_f4u*)(p + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { r[e] = v0[e]; r[4 + e] = v1[e]; }
                if (ex) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) r[e] = expf(r[e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = 0.0f;
                    if (k < kend && gi + e < nrows) { v = p[e]; if (ex) v = expf(v); }
                    r[e] = v;
                }
            }
        }
    }
    __device__ __forceinline__ void commit(float* __restrict__ img) const {
        if (KCONT) {
#pragma unroll
            for (int e = 0; e < 8; ++e) img[row * SPX_G_LD + kk + e] = r[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) img[(row + e) * SPX_G_LD + kk] = r[e];
        }
    }
};

template <bool AK, bool BK>
__global__ __launch_bounds__(256) void spx_gemm_kernel(const SpxGemmArgs a) {
    __shared__ float As[2][SPX_G_TI * SPX_G_LD];
    __shared__ float Bs[2][SPX_G_TI * SPX_G_LD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wi = wave >> 1, wj = wave & 1;
    const int i0 = blockIdx.x * SPX_G_TI, j0 = blockIdx.y * SPX_G_TI;
    const int kbeg = blockIdx.z * a.kper;
    const int kend = (kbeg + a.kper < a.K) ? kbeg + a.kper : a.K;
    const bool exa = a.flags & 1, exb = a.flags & 2;

    f32x16 acc[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][v][r] = 0.0f;

    SpxGemmStager<AK> sa;
    SpxGemmStager<BK> sb;
    sa.init(tid);
    sb.init(tid);
    if (kbeg < kend) {
        sa.load(a.A, a.ras, a.kas, i0, a.M, kbeg, kend, exa);
        sb.load(a.B, a.rbs, a.kbs, j0, a.N, kbeg, kend, exb);
        sa.commit(As[0]);
        sb.commit(Bs[0]);
    }
    __syncthreads();
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += SPX_G_TK) {
        const bool more = k0 + SPX_G_TK < kend;
        if (more) {
            sa.load(a.A, a.ras, a.kas, i0, a.M, k0 + SPX_G_TK, kend, exa);
            sb.load(a.B, a.rbs, a.kbs, j0, a.N, k0 + SPX_G_TK, kend, exb);
        }
        const float* ap = As[buf] + (wi * 64 + (lane & 31)) * SPX_G_LD + (lane >> 5);
        const float* bp = Bs[buf] + (wj * 64 + (lane & 31)) * SPX_G_LD + (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < SPX_G_TK / 2; ++kk) {
            const float a0 = ap[2 * kk], a1 = ap[32 * SPX_G_LD + 2 * kk];
            const float b0 = bp[2 * kk], b1 = bp[32 * SPX_G_LD + 2 * kk];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) {
            sa.commit(As[buf ^ 1]);
            sb.commit(Bs[buf ^ 1]);
        }
        __syncthreads();
        buf ^= 1;
    }

    // accumulator tile: register r of lane l is row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31: a store instruction
    // writes two 128-B row pieces
    float* const out = a.splits > 1 ? a.ws + (size_t)blockIdx.z * (size_t)a.M * (size_t)a.N : a.C;
    const long long ldo = a.splits > 1 ? a.N : a.ldc;
    const bool mulexp = (a.flags & 4) && a.splits == 1;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int j = j0 + wj * 64 + v * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + wi * 64 + u * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
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

// split policy: a pure function of the shape (so results do not depend on the machine state)
static void spx_gemm_split(int M, int N, int K, int flags, int& splits, int& kper) {
    const long long tiles = (long long)((M + SPX_G_TI - 1) / SPX_G_TI) * ((N + SPX_G_TI - 1) / SPX_G_TI);
    int s = 1;
    if (tiles < 128 && K >= 1024 && !(flags & 4)) {      // (the exp-scaled epilogue is applied by the product kernel itself)
        s = (int)(512 / tiles);
        const int smax = K / 256;
        if (s > smax) s = smax;
        if (s > 64) s = 64;
        if (s < 1) s = 1;
    }
    int kp = (K + s - 1) / s;
    kp = (kp + SPX_G_TK - 1) / SPX_G_TK * SPX_G_TK;
    if (kp < SPX_G_TK) kp = SPX_G_TK;
    s = (K + kp - 1) / kp;
    if (s < 1) s = 1;
    splits = s;
    kper = kp;
}

size_t spx_gemm_workspace(int M, int N, int K, int flags) {
    int s, kp;
    spx_gemm_split(M, N, K, flags, s, kp);
    return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
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
    spx_gemm_split(M, N, K, flags, a.splits, a.kper);
    a.ws = ws;
    const dim3 grid((M + SPX_G_TI - 1) / SPX_G_TI, (N + SPX_G_TI - 1) / SPX_G_TI, a.splits);
    const bool ak = kas == 1, bk = kbs == 1;
    if (ak && bk) hipLaunchKernelGGL((spx_gemm_kernel<true, true>), grid, dim3(256), 0, s, a);
    else if (ak) hipLaunchKernelGGL((spx_gemm_kernel<true, false>), grid, dim3(256), 0, s, a);
    else if (bk) hipLaunchKernelGGL((spx_gemm_kernel<false, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((spx_gemm_kernel<false, false>), grid, dim3(256), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (a.splits > 1) {
        const long long MN = (long long)M * N;
        hipLaunchKernelGGL(spx_gemm_reduce_kernel, dim3((unsigned)((MN + 255) / 256)), dim3(256), 0, s, (const float*)ws, a.splits, MN, N, C, ldc);
        e = hipGetLastError();
    }
    return e;
}
