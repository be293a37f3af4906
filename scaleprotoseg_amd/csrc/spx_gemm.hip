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

// ---------------------------------------------------------------------------------------------------------------------
// The same product on the bf16 matrix pipe at fp32 accuracy ("bf16x3"): every fp32 operand is split into three bf16 planes
// x = h + m + l (8 + 8 + 8 significant bits = the fp32 mantissa, each plane the round-to-nearest bf16 of the remainder), and
//     x . y  =  h h' + (h m' + m h') + (h l' + l h' + m m')  (+ m l' + l m' + l l':  <= 2^-23 |x y|, dropped)
// runs as SIX v_mfma_f32_32x32x16_bf16 per fp32 k-step of 16: 6 x 8 passes against the 8 x 16 passes of
// v_mfma_f32_32x32x2_f32 over the same k - 2.7x the fp32 pipe's rate with the same fp32 accumulation, the dropped terms
// below the summation-order noise (tests/test_gpu_gemm.py holds both kernels to the same 2e-6 |A|.|B| bound).
// Workgroup = 4 waves = 128 x 128 of C (wave: 64 x 64 = 2 x 2 MFMA tiles), k in chunks of 32 (a k-contiguous operand is then
// fetched in whole 128-B row pieces) through ONE LDS buffer of three bf16 planes per operand (48 KiB: three workgroups per CU):
//   k-contiguous operand:   [plane][row][32 k], 64-B rows, the row's four 16-B slots XOR-swizzled by (row >> 2) & 3 - a fragment
//                           (row = lane & 31, k = 8 (lane >> 5) + j) is one ds_read_b128 and the 16 lanes of a read group fall on 16
//                           distinct slots without padding: 48 KiB for two such operands, THREE workgroups per CU;
//   row-contiguous operand: [plane][k][128 rows], 256-B rows, 8-B slots XOR-swizzled by 8 (k & 3) - a thread's 8 consecutive
//                           rows are one 16-B write, and the fragment comes back through two ds_read_b64_tr_b16 (k becomes the
//                           register index; the swizzle keeps the four k rows x eight slots of a 32-lane half on distinct banks).
typedef __attribute__((ext_vector_type(8))) short s16x8;
#define SPX_G3_TI 128
#define SPX_G3_TK 32
#define SPX_G3_KS 64                      // k-contiguous image: row stride in bytes (32 k, unpadded: 16-B slots XOR-swizzled by (row >> 2) & 3)
#define SPX_G3_RS 256                     // row-contiguous image: k-row stride in bytes (128 rows, unpadded: 8-B slots XOR-swizzled by 8 (k & 3))
template <bool KCONT> __host__ __device__ constexpr int spx_g3_plane() { return KCONT ? 128 * SPX_G3_KS : 32 * SPX_G3_RS; }
template <bool KCONT> __host__ __device__ constexpr int spx_g3_oper() { return 3 * spx_g3_plane<KCONT>(); }

__device__ __forceinline__ void split3_bf16x8(const float (&r)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 a = (__bf16)r[e];
        const float r1 = r[e] - (float)a;
        const __bf16 b = (__bf16)r1;
        h[e] = a;
        m[e] = b;
        l[e] = (__bf16)(r1 - (float)b);
    }
}

// One operand's chunk: 128 rows x 32 k as two pieces of 256 threads x 8 floats (k-contiguous: 8 k of one row, rows +64 for the
// second piece;  row-contiguous: 8 rows of one k, k +16), fetched with UNCONDITIONAL 16-B buffer loads: the resource starts at
// the chunk's first element and ends with the tensor, a lane without a valid element passes an out-of-range offset, and a
// lane whose window crosses the end of its row range keeps a count of valid elements and zeroes the rest at commit.  (The
// fp32-pipe kernel's guarded element loads put a vmcnt(0) between the pieces on every ragged tile: eight serial memory round
// trips per chunk, the 128-tiles of N = 450 ran at a third of the rate of its interior tiles.)
template <bool KCONT>
struct SpxGemm3Operand {
    u32x4 v[2][2];
    int nval[2];
    int row, kk;
    uint32_t voff, pstride;
    __device__ __forceinline__ void init(int tid, long long rs, long long ks) {
        if (KCONT) { row = tid >> 2; kk = (tid & 3) * 8; voff = (uint32_t)((row * rs + kk) * 4); pstride = (uint32_t)(64 * rs * 4); }
        else { kk = tid >> 4; row = (tid & 15) * 8; voff = (uint32_t)((kk * ks + row) * 4); pstride = (uint32_t)(16 * ks * 4); }
    }
    __device__ __forceinline__ void load(const float* __restrict__ P, long long rs, long long ks, int r0, int nrows, int K, int k0, int kend) {
        const long long RS = KCONT ? rs : 1, KS = KCONT ? 1 : ks;
        const long long total = (long long)(nrows - 1) * RS + (long long)(K - 1) * KS + 1;
        const long long base = (long long)r0 * RS + (long long)k0 * KS;
        long long rem = (total - base) * 4;
        rem = rem < 0 ? 0 : (rem > 0x7FFFFFFFLL ? 0x7FFFFFFFLL : rem);
        const spx_rsrc rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(rem ? P + base : P), 0, (uint32_t)rem, 0x00020000);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int gi = r0 + row + (KCONT ? 64 * p : 0), k = k0 + kk + (KCONT ? 0 : 16 * p);
            int n = KCONT ? kend - k : nrows - gi;
            n = n < 0 ? 0 : (n > 8 ? 8 : n);
            if (KCONT ? gi >= nrows : k >= kend) n = 0;
            const uint32_t off = n ? voff + p * pstride : SPX_OOB;
            v[p][0] = buf_load_b128(rsrc, off, 0);
            v[p][1] = buf_load_b128(rsrc, off + 16, 0);
            nval[p] = n;
        }
    }
    __device__ __forceinline__ void commit(char* __restrict__ img, bool ex) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float r[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = __uint_as_float(v[p][e >> 2][e & 3]);
            if (ex) {
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] = expf(r[e]);
            }
            if (__builtin_amdgcn_ballot_w64(nval[p] != 8) != 0) {        // wave-uniform: interior chunks skip the masking
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] = e < nval[p] ? r[e] : 0.0f;
            }
            bf16x8 h, m, l;
            split3_bf16x8(r, h, m, l);
            // KCONT: 8 k of one row;  else: 8 rows of one k - 16 contiguous bytes of the image either way
            const int rw = row + 64 * p;
            const int kr = kk + 16 * p;
            char* const dst = img + (KCONT ? rw * SPX_G3_KS + (((kk >> 3) ^ ((rw >> 2) & 3)) << 4) : kr * SPX_G3_RS + (((row >> 2) ^ (8 * (kr & 3))) << 3));
            *(bf16x8*)dst = h;
            *(bf16x8*)(dst + spx_g3_plane<KCONT>()) = m;
            *(bf16x8*)(dst + 2 * spx_g3_plane<KCONT>()) = l;
        }
    }
};

// fragment of k-step ks (16 k) of the 32 rows starting at r0 of one plane image
template <bool KCONT>
__device__ __forceinline__ bf16x8 g3_frag(const char* __restrict__ plane, int r0, int ks, int lane) {
    if (KCONT) {
        const int rw = r0 + (lane & 31);
        return *(const bf16x8*)(plane + rw * SPX_G3_KS + (((2 * ks + (lane >> 5)) ^ ((rw >> 2) & 3)) << 4));
    }
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pq = i & 3;
    // 8-B slot (r0 + 16 (g & 1)) / 4 + pq of k row 16 ks + 8 (g >> 1) + q, swizzled by the row's k & 3 = q (r0 is a multiple of 32:
    // the eight slots of a 32-lane half keep bits 3-4 for the swizzle, so its 4 k rows x 8 slots fall on 32 distinct bank pairs)
    const char* const p0 = plane + (16 * ks + 8 * (g >> 1) + q) * SPX_G3_RS + ((((r0 >> 2) + 4 * (g & 1) + pq) ^ (8 * q)) << 3);
    const s16x4 lo = lds_tr_read(p0), hi = lds_tr_read(p0 + 4 * SPX_G3_RS);
    s16x8 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
    return __builtin_bit_cast(bf16x8, v);
}

template <bool AK, bool BK>
__global__ __launch_bounds__(256, 3) void spx_gemm3_kernel(const SpxGemmArgs a) {
    constexpr int TI = SPX_G3_TI, TK = SPX_G3_TK;
    extern __shared__ __attribute__((aligned(16))) char g3_lds[];        // [A | B][plane image]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wi = wave >> 1, wj = wave & 1;
    const int per = (a.ntiles + 7) / 8;                                    // XCD-aware block -> tile map (see spx_gemm_kernel)
    const int L = a.linear_map ? (int)blockIdx.x : (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if (L >= a.ntiles || (!a.linear_map && (int)(blockIdx.x >> 3) >= per)) return;
    const int bj = L % a.nj, bi = (L / a.nj) % a.ni, bz = L / (a.nj * a.ni);
    const int i0 = bi * TI, j0 = bj * TI;
    const int kbeg = bz * a.kper;
    const int kend = (kbeg + a.kper < a.K) ? kbeg + a.kper : a.K;
    const bool exa = a.flags & 1, exb = a.flags & 2;

    f32x16 acc[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][v][r] = 0.0f;

    SpxGemm3Operand<AK> sa;
    SpxGemm3Operand<BK> sb;
    sa.init(tid, a.ras, a.kas);
    sb.init(tid, a.rbs, a.kbs);
    char* const ia = g3_lds;
    char* const ib = g3_lds + spx_g3_oper<AK>();
    // One LDS buffer, one register set: chunk c + 1 is in flight in registers while chunk c feeds the matrix pipe; the two
    // workgroups of a CU fill each other's commit phases.  (A second register set - two chunks of lead - measured no faster.)
    sa.load(a.A, a.ras, a.kas, i0, a.M, a.K, kbeg, kend);
    sb.load(a.B, a.rbs, a.kbs, j0, a.N, a.K, kbeg, kend);
    for (int k0 = kbeg; k0 < kend; k0 += TK) {
        __syncthreads();                     // the previous chunk's fragment reads are done
        sa.commit(ia, exa);
        sb.commit(ib, exb);
        __syncthreads();
        sa.load(a.A, a.ras, a.kas, i0, a.M, a.K, k0 + TK, kend);       // (a chunk past the end loads nothing)
        sb.load(a.B, a.rbs, a.kbs, j0, a.N, a.K, k0 + TK, kend);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[2][3], fb[2][3];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    fa[u][pl] = g3_frag<AK>(ia + pl * spx_g3_plane<AK>(), (wi * 2 + u) * 32, ks, lane);
                    fb[u][pl] = g3_frag<BK>(ib + pl * spx_g3_plane<BK>(), (wj * 2 + u) * 32, ks, lane);
                }
            // smallest terms first into the accumulator
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    f32x16 c = acc[u][v];
                    c = mfma_bf16(fa[u][2], fb[v][0], c);       // l h'
                    c = mfma_bf16(fa[u][0], fb[v][2], c);       // h l'
                    c = mfma_bf16(fa[u][1], fb[v][1], c);       // m m'
                    c = mfma_bf16(fa[u][1], fb[v][0], c);       // m h'
                    c = mfma_bf16(fa[u][0], fb[v][1], c);       // h m'
                    c = mfma_bf16(fa[u][0], fb[v][0], c);       // h h'
                    acc[u][v] = c;
                }
        }
    }

    float* const out = a.splits > 1 ? a.ws + (size_t)bz * (size_t)a.M * (size_t)a.N : a.C;
    const long long ldo = a.splits > 1 ? a.N : a.ldc;
    const bool mulexp = (a.flags & 4) && a.splits == 1;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int j = j0 + (wj * 2 + v) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + (wi * 2 + u) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
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
    // four slabs in flight per round (a rolled sum pays one memory round trip per slab); slab order is kept: a fixed order
    float s = 0.0f;
    int k = 0;
    for (; k + 4 <= splits; k += 4) {
        const float v0 = ws[(size_t)k * MN + g], v1 = ws[(size_t)(k + 1) * MN + g];
        const float v2 = ws[(size_t)(k + 2) * MN + g], v3 = ws[(size_t)(k + 3) * MN + g];
        s = (((s + v0) + v1) + v2) + v3;
    }
    for (; k < splits; ++k) s += ws[(size_t)k * MN + g];
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
    // The bf16x3 kernel (128-tiles) unless its tiles would be mostly padding (N = 150: 1.7x) or there is next to no work.
    // Splits, from the sweeps over the ADE / COCO head shapes (tools/probes/gemm3_sweep.py, profiles/EXPERIMENTS.md): slabs of
    // ~420 k for contractions over prototypes (k-contiguous operands: three workgroups per CU) and ~700 k for contractions over
    // pixels, at most ~2000 workgroups, at least ~400, and for the pixel contractions never between one and one and a half rounds
    // of the 512 workgroup slots (536 workgroups took as long as 268: the last 24 run alone).
    const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128);
    const double waste = (double)(t128 * 128 * 128) / ((double)M * (double)N);
    if (waste <= 1.5 && t128 >= 8 && K >= 64) {
        int sp = 1;
        if (!(flags & 4)) {
            const int slab = K > 4096 ? 700 : 420;        // pixel contractions (few tiles, long k) / prototype contractions
            sp = (K + slab / 2) / slab;
            sp = sp < 1 ? 1 : (sp > 16 ? 16 : sp);
            while (sp > 1 && t128 * sp > 2000) --sp;
            while (sp < 16 && t128 * sp < 400 && K / (sp + 1) >= 128) ++sp;
            if (K > 4096)
                while (sp < 16 && t128 * sp > 512 && t128 * sp < 800 && K / (sp + 1) >= 128) ++sp;
        }
        int kp = ((K + sp - 1) / sp + SPX_G_TK - 1) / SPX_G_TK * SPX_G_TK;
        return SpxGemmPlan{3, (K + kp - 1) / kp, kp};
    }
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

static void spx_launch_gemm3(const SpxGemmArgs& a_, bool ak, bool bk, hipStream_t s) {
    SpxGemmArgs a = a_;
    a.ni = (a.M + SPX_G3_TI - 1) / SPX_G3_TI;
    a.nj = (a.N + SPX_G3_TI - 1) / SPX_G3_TI;
    a.ntiles = a.ni * a.nj * a.splits;
    a.linear_map = g_gemm_force_splits < 0;
    const dim3 grid((unsigned)(((a.ntiles + 7) / 8) * 8));
    if (ak && bk) hipLaunchKernelGGL((spx_gemm3_kernel<true, true>), grid, dim3(256), spx_g3_oper<true>() + spx_g3_oper<true>(), s, a);
    else if (ak) hipLaunchKernelGGL((spx_gemm3_kernel<true, false>), grid, dim3(256), spx_g3_oper<true>() + spx_g3_oper<false>(), s, a);
    else if (bk) hipLaunchKernelGGL((spx_gemm3_kernel<false, true>), grid, dim3(256), spx_g3_oper<false>() + spx_g3_oper<true>(), s, a);
    else hipLaunchKernelGGL((spx_gemm3_kernel<false, false>), grid, dim3(256), spx_g3_oper<false>() + spx_g3_oper<false>(), s, a);
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
    if (p.wm == 3) spx_launch_gemm3(a, ak, bk, s);
    else if (p.wm == 2) spx_launch_gemm_wm<2>(a, ak, bk, s);
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
