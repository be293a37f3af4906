// Backward, parameter side (kernels 2 and 3 of the backward; kernel 1 = spx_bwd_impl.h, whose header describes the
// G / a fragment blobs consumed here).
#include "spx_args.h"
#include "spx_common.h"
#include <type_traits>

// ------------------------------------------------------------------------------------------------
// kernel 2: parameter side   S[q][row][col] = sum_px Gq[row][px] * Xs[col][px]   (+ a^T.dLogits, colsum G)
//
// One workgroup = 8 waves (2 per SIMD) owns one (panel, pixel split) and walks its 64-px chunks.  Per chunk the
// G / a fragment blobs of the two kernel-1 waves covering it are copied VERBATIM into LDS and read back with
// ds_read_b64_tr_b16 so that the pixel becomes the MFMA k (A operand = G^T / a^T rows); X rows [channel][px]
// are the B operand (ds_read_b128); dLogits is transposed to bf16 [class][px] while staged.  Wave w accumulates
// the 32x32 tiles (prototype block pb in its half, channel blocks 2(w&3), 2(w&3)+1); the tiles live in
// registers for the whole launch and leave as one fp32 slab per workgroup.  Straight-line chunk body: panel
// height, class blocks and the 8 channel blocks are compile-time (channels beyond Cs are zero rows).
// ------------------------------------------------------------------------------------------------
// Pixels per K-chunk (template parameter CPX): 64 = half a kernel-1 tile, ONE LDS buffer, commit -> barrier -> compute ->
// barrier per chunk; 32 = one kernel-1 wave, TWO LDS buffers: chunk c computes from one buffer while chunk c+1 is
// converted into the other and chunk c+2's loads are in flight, one barrier per chunk (the loads then have a whole
// iteration to land instead of the MFMA phase only, and the conversion VALU of one wave runs under the other's MFMAs).
#define SPX_BK_THREADS 512
// Default since the end of round 2 (with the int16 activation blob the commit phase got heavier): the pipelined + staggered
// variant for the 6-block panels without the k-step split (the north-star bank), 0.783 -> 0.740 ms on one box.
#define SPX_BANK_PIPE 1
#define SPX_BANK_STAGGER 1
// LDS row stride of the [channel][px] images: 2 CPX + 16 bytes (144 / 80: conflict-free ds_read_b128)
__host__ __device__ constexpr int spx_bk_row(int cpx) { return 2 * cpx + 16; }

__host__ __device__ inline int spx_bk_wstride(const spx_plan& pl) {
    return ((pl.channels_per_scale + 31) / 32) * 32 + pl.ncb * 32 + 32;   // [dP cols | dW cols | colsum + pad]
}
// The chunk walk is launched in up to two parts (see spx_launch_fwd_npb: vector staging wants whole 8-pixel pieces): the
// chunks that lie wholly inside their image, and - when H*W is not a multiple of 8 - each image's ragged last chunk(s) on
// the element-wise path.  Both parts write partial slabs; the reduction kernel sums all of them.
static void spx_bank_parts(const spx_plan& pl, int B, int HW, int& nci, int& nci_vec, int& slabs_vec, int& slabs_tail) {
    nci = 2 * ((HW + SPX_TILE_PX - 1) / SPX_TILE_PX);           // 64-px chunks per image (the blobs are tile-granular)
    nci_vec = HW >= 8 ? nci : 0;                                 // the vector staging path handles a ragged image end itself (H*W >= 8)
    long long cap = 256 / pl.npanels;
    if (cap < 1) cap = 1;
    const long long cv = (long long)B * nci_vec, ct = (long long)B * (nci - nci_vec);
    slabs_vec = (int)(cv < cap ? cv : cap);
    slabs_tail = (int)(ct < cap ? ct : cap);
}
int spx_bank_bwd_nsplit(const spx_plan& pl, int B, int HW) {
    int nci, nci_vec, sv, st;
    spx_bank_parts(pl, B, HW, nci, nci_vec, sv, st);
    return sv + st;
}
size_t spx_bank_bwd_ws_floats(const spx_plan& pl, int nsplit) {
    return (size_t)nsplit * pl.npanels * pl.npb * 32 * spx_bk_wstride(pl);
}
// LDS carve: [G16 | X] (d_bank instances: both fp16) then [a hi | a lo | dLogits^T hi | dLogits^T lo] (d_W instances)
template <int NPB, int NCB, bool DO_P, bool DO_W, int CPX>
__host__ __device__ constexpr int spx_bk_buf_bytes() {
    constexpr int fb = (CPX / 32) * NPB * 2 * 1024;
    return (DO_P ? fb + 256 * spx_bk_row(CPX) : 0) + (DO_W ? 2 * fb + 2 * NCB * 32 * spx_bk_row(CPX) : 0);
}
template <int NPB, int NCB, bool DO_P, bool DO_W, int CPX>
__host__ __device__ constexpr int spx_bk_lds_bytes() {
    constexpr int red = (2 * 3 * 2 * 1024 + 2 * 3 * 64) * 4;          // the k-step-split reduction re-uses the staging area
    constexpr int n = (CPX == 32 ? 2 : 1) * spx_bk_buf_bytes<NPB, NCB, DO_P, DO_W, CPX>();
    return (n > red ? n : red) + 16;        // + the fp16 exponent of the chunk in each buffer
}

// DO_P / DO_W: which of the two products this instance carries.  One launch does both for small heads; for the
// 5-block head (80 d_W accumulators per lane next to 96 for d_bank) the launcher runs two instances, each with its
// own operands only (G + X, or a + dLogits): no byte is read twice and neither instance spills.
// KSPLIT: scales of <= 64 channels (see the wave roles below).
template <int NPB, int NCB, bool XF32, int VM, bool DO_P, bool DO_W, bool KSPLIT, int CPX>
__global__ __launch_bounds__(SPX_BK_THREADS, 2) void spx_bank_bwd_kernel(const SpxBankBwdArgs a) {
    constexpr bool VEC = VM != 0, RAG = VM == 2;
    constexpr int SPX_BK_PX = CPX, SPX_BK_ROW = spx_bk_row(CPX);
    constexpr int NW1 = CPX / 32;                     // kernel-1 waves per chunk
    constexpr bool PIPE = CPX == 32;
    static_assert(!(PIPE && KSPLIT), "the k-step split needs four pixel k-steps per chunk");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const spx_plan& pl = a.plan;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int q = blockIdx.y, split = blockIdx.x;      // split: this launch's workgroup index; its slab is slab_first + split
    const int Cs = pl.channels_per_scale, K = pl.num_classes;
    const int C = pl.num_scales * Cs;
    const int nchb = (Cs + 31) / 32;
    constexpr int rows = NPB * 32;
    const int ch0 = pl.panel_ch0[q];
    const int tiles_per_img = (a.HW + SPX_TILE_PX - 1) / SPX_TILE_PX;
    const size_t ntiles = (size_t)a.B * tiles_per_img;
    static_assert(CPX == 64 || CPX == 32, "chunk width");
    const int nci = a.nci_launch * (64 / CPX);               // chunks per image IN THIS LAUNCH (the launcher counts 64-px chunks)
    const int ci0 = a.ci_first * (64 / CPX);
    const long long total = (long long)a.B * nci;
    // chunk c belongs to workgroup c mod nsplit: the workgroups running at one time then cover CONSECUTIVE chunks, i.e.
    // one contiguous stretch of every X row (16-32 KB) and of the blobs, instead of 128-B pieces 16 KB apart
    // (DRAM page locality).  Every slab still sums a fixed chunk set in a fixed order: results stay deterministic.
    const long long cstep = a.nslabs;
    // Pipelined variant (32-px chunks = 64-B halves of the 128-B lines of an X row): chunks 2j and 2j + 1 go to two workgroups
    // of the SAME XCD (workgroup w runs on XCD w mod 8), so the second half of a line hits that XCD's L2 instead of being
    // fetched into another one.  A bijection of [0, nslabs) whenever nslabs is a multiple of 16; the slab index stays `split`.
    int phase = split;
    if (CPX == 32 && (a.nslabs & 15) == 0) {
        const int xcd = split & 7, slot = split >> 3;
        phase = 2 * ((slot >> 1) * 8 + xcd) + (slot & 1);
    }
    const long long c_begin = phase;
    const long long c_end = total;                     // exclusive bound of this workgroup's walk c_begin, c_begin + cstep, ...
    const bool want_w = DO_W && a.d_W != nullptr;
    const bool want_p = DO_P && a.d_bank != nullptr;
    constexpr int ESZ = XF32 ? 4 : 2;
    constexpr int NFRAG = NW1 * NPB * 2;              // fragments per chunk and image: kernel-1 waves x NPB x 2 k-steps
    constexpr int FBYTES = NFRAG * 1024;

    // The G blob is ONE fp16 plane with a power-of-two scale per kernel-1 tile (exponent in the side array behind the blobs):
    // it is copied verbatim and enters the d_bank product as it is (fp16 MFMA against X, which is bf16-representable, i.e.
    // exact in fp16).  The activation blob (int16 / fp16 codes) is split into an exact bf16 hi + lo pair while it is committed
    // to LDS, in the blob's own lane order (the split is elementwise).
    constexpr int BUF = spx_bk_buf_bytes<NPB, NCB, DO_P, DO_W, CPX>();
    int32_t* const ebuf = (int32_t*)(smem + spx_bk_lds_bytes<NPB, NCB, DO_P, DO_W, CPX>() - 16);     // [2]
    char *Gs, *Xs, *As, *As2, *Ls, *Ls2;
    int cur_buf = 0;
    auto use_buffer = [&](int buf) {
        cur_buf = buf;
        Gs = smem + buf * BUF;                        // G16 fragments
        Xs = Gs + (DO_P ? FBYTES : 0);                // [256][row]  X rows as fp16 (rows >= Cs are zero)
        As = Xs + (DO_P ? 256 * SPX_BK_ROW : 0);      // a fragments: bf16 high part of the fp16 blob
        As2 = As + FBYTES;                            // ... and the bf16 residual (a = hi + lo exactly)
        Ls = As2 + FBYTES;                            // [NCB*32][row] dLogits^T, bf16 high part
        Ls2 = Ls + NCB * 32 * SPX_BK_ROW;             // ... and the bf16 residual: dLogits enters d_W as hi + lo (~2^-17)
    };
    use_buffer(0);

    // wave roles
    const int cpair = wave & 3;                       // channel blocks 2*cpair, 2*cpair+1
    constexpr int PH = NPB / 2;                       // prototype blocks per half
    const int pb0 = (wave >> 2) * PH;                 // this wave's prototype blocks pb0 .. pb0+PH-1
    const bool w_role = wave < NPB;                   // waves 0..NPB-1 also own the dW tiles of block `wave`
    // Scales of <= 64 channels (every ScaleProtoSeg gin: Cs = 64) have only channel blocks 0 and 1: the four cpair
    // waves of a prototype half then split the chunk's four pixel k-steps instead (wave cpair takes k-step cpair for
    // both channel blocks) and their partial tiles are summed through LDS once, after the last chunk.
    constexpr bool ksplit = KSPLIT;
    // colsum(G) of prototype block i of the half: per k-step owner (split); otherwise the wave cs_owner(i), chosen among the
    // waves with the least other work (waves 0..NPB-1 also carry the d_W tiles; waves wave and wave + 4 share a SIMD)
    // (per SIMD beyond the bank MFMAs: two d_W tiles + one colsum on SIMDs 0 and 1, one d_W tile + two colsums on 2 and 3)
    const int cs_first = wave >> 2;                   // half 0: blocks -> cpair 0, 2, 3; half 1: cpair 1, 2, 3
    auto cs_owner = [&](int i) { return i == 0 ? cs_first : i + 1; };

    f32x16 accp[PH][2];
    f32x16 accw[NCB];
    float csum[PH];
    // the d_bank accumulators (and colsum) live in the units of the chunk added last (2^e_acc); a chunk of a tile with another
    // fp16 exponent first multiplies them by the power-of-two ratio (see chunk_scale)
    int e_acc = 0;
    bool have_e = false;
#pragma unroll
    for (int i = 0; i < PH; ++i) {
        csum[i] = 0.0f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) accp[i][t][e] = 0.0f;
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) accw[cb][e] = 0.0f;

    // staging registers (512 threads)
    constexpr int FP = (FBYTES / 16 + SPX_BK_THREADS - 1) / SPX_BK_THREADS;   // 16-B pieces per thread per image
    constexpr int PPR = CPX / 8;                                                // 8-px pieces per X row
    constexpr int RPP = SPX_BK_THREADS / PPR;                                   // rows per staging pass
    constexpr int XPT = 256 / RPP;                                              // X pieces per thread
    // staging registers: one set per chunk in flight (the pipelined variant keeps two chunks of loads in flight)
    struct Stage {
        u32x4 gr[FP], ar[FP], xr[XPT][XF32 ? 2 : 1];
        uint32_t exw[FP];                             // block exponents of each staged piece (one lane of one block): G | activation << 8, +128 each
        float lr_[(CPX * 32 * NCB + SPX_BK_THREADS - 1) / SPX_BK_THREADS];
        int32_t gexp;                                 // fp16 exponent of the chunk's G16 (kernel-1 tile uniform)
        uint32_t rot_bits;                            // this thread's X piece straddles the image end (see SpxXStager::make_ctx)
        bool ragged;                                  // chunk-uniform: some piece of the chunk does
    };
    Stage stg[PIPE ? 2 : 1];
    // piece -> (fragment, lane) of the blob: the inverse of spx_blob_slot.  Fixed per thread for the whole launch.
    uint32_t gsc_off[FP];
#pragma unroll
    for (int i = 0; i < FP; ++i) {
        const int off = (i * SPX_BK_THREADS + tid) * 16;
        const int frag = off >> 10, slot = (off >> 4) & 63, s2 = frag & 1;
        const int t = (slot - 8 * s2) & 63;
        const int rr = (t >> 3) * 4 + (t & 3), hh = (t & 7) >> 2;
        gsc_off[i] = off < FBYTES ? (uint32_t)(((frag >> 1) * 64 + rr + 32 * hh) * 4) : SPX_OOB;
    }
    const size_t blob_total = (size_t)pl.npanels * ntiles * 4 * NPB * 2 * 1024;
    constexpr int LPT = (SPX_BK_PX * 32 * NCB + SPX_BK_THREADS - 1) / SPX_BK_THREADS;   // dLogits elements per thread (upper bound)
    const int piece = tid % PPR, prow = tid / PPR;   // X staging: piece of 8 px, row within a pass of RPP rows

    // The walk issues its chunks strictly in order (c_begin, c_begin + cstep, ...; past the end the last one again), so
    // the (image, chunk-in-image) position advances incrementally: one 64-bit division per launch instead of one per chunk.
    long long nx_c = c_begin;
    int nx_b = (int)(c_begin / nci);
    int nx_ci = (int)(c_begin - (long long)nx_b * nci);
    const int step_b = (int)(cstep / nci), step_ci = (int)(cstep - (long long)step_b * nci);
    auto issue = [&](Stage& st) {
        const int b = nx_b;
        const int ci = ci0 + nx_ci;
        if (nx_c + cstep < c_end) {
            nx_c += cstep;
            nx_b += step_b;
            nx_ci += step_ci;
            if (nx_ci >= nci) {
                nx_ci -= nci;
                ++nx_b;
            }
        }
        constexpr int CPT = SPX_TILE_PX / CPX;            // chunks per kernel-1 tile
        const size_t tile_g = (size_t)b * tiles_per_img + ci / CPT;
        // the chunk's fragments are contiguous: kernel-1 waves NW1 (ci % CPT) ... of the tile
        const size_t blob0 = ((((size_t)q * ntiles + tile_g) * 4 + NW1 * (ci % CPT)) * NPB * 2) * 1024;
        const spx_rsrc grs = make_rsrc_pred(a.g_in ? (const char*)a.g_in + blob0 : nullptr);
        const spx_rsrc ars = make_rsrc_pred(a.a_in ? (const char*)a.a_in + blob0 : nullptr);
        // the block exponents of the activation blob sit behind it
        const char* const exsrc = (want_w && a.a_in) ? (const char*)a.a_in : nullptr;
        const spx_rsrc exs = make_rsrc_pred(exsrc ? exsrc + blob_total + blob0 / 8 : nullptr);
        st.gexp = (want_p && a.g_in) ? *(const int32_t*)((const char*)a.g_in + spx_gexp_offset(blob_total) + ((size_t)q * ntiles + tile_g) * 4) : 0;
#pragma unroll
        for (int i = 0; i < FP; ++i) {
            const uint32_t off = (uint32_t)((i * SPX_BK_THREADS + tid) * 16);
            st.gr[i] = buf_load_b128(grs, (want_p && off < (uint32_t)FBYTES) ? off : SPX_OOB, 0);
            st.exw[i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(exs, exsrc ? gsc_off[i] : SPX_OOB, 0, 0);
            st.ar[i] = buf_load_b128(ars, (want_w && off < (uint32_t)FBYTES) ? off : SPX_OOB, 0);
        }
        const int px = ci * SPX_BK_PX + piece * 8;
        {
            const int over = px + 8 - a.HW;
            st.rot_bits = (RAG && over > 0 && over < 8) ? (uint32_t)over * (ESZ * 8) : 0u;
            st.ragged = RAG && (ci * SPX_BK_PX + SPX_BK_PX > a.HW);
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int row = prow + RPP * i;
            // rebase per (image, row block): offsets from the tensor base can exceed 4 GiB for large batches
            const spx_rsrc xb = make_rsrc_pred((const char*)a.x + ((size_t)b * C + ch0 + RPP * i) * a.HW * ESZ);
            const uint32_t vo = ((uint32_t)prow * (uint32_t)a.HW + (uint32_t)px) * ESZ;
            const bool row_ok = want_p && row < Cs;
            if (VEC) {
                const int over = px + 8 - a.HW;                  // elements of the piece past the image
                const uint32_t v = !row_ok ? SPX_OOB : (over <= 0 ? vo : ((RAG && over < 8) ? vo - (uint32_t)over * ESZ : SPX_OOB));
                st.xr[i][0] = buf_load_b128(xb, v, 0);
                if (XF32) st.xr[i][1] = buf_load_b128(xb, v == SPX_OOB ? SPX_OOB : v + 16, 0);
            } else if (XF32) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    st.xr[i][e >> 2][e & 3] = __float_as_uint(buf_load_f32(xb, (row_ok && px + e < a.HW) ? vo + 4 * e : SPX_OOB, 0));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t lo = buf_load_u16(xb, (row_ok && px + 2 * e < a.HW) ? vo + 4 * e : SPX_OOB, 0);
                    const uint32_t hi = buf_load_u16(xb, (row_ok && px + 2 * e + 1 < a.HW) ? vo + 4 * e + 2 : SPX_OOB, 0);
                    st.xr[i][0][e] = lo | (hi << 16);
                }
            }
        }
        // dLogits of the chunk's 64 px: a contiguous [64][K] fp32 block
        const int px0 = ci * SPX_BK_PX;
        const spx_rsrc lb = make_rsrc_pred(a.d_logits ? a.d_logits + ((size_t)b * a.HW + px0) * K : nullptr);
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int e = i * SPX_BK_THREADS + tid;
            const int p = e / K;
            st.lr_[i] = buf_load_f32(lb, (want_w && e < SPX_BK_PX * K && px0 + p < a.HW) ? (uint32_t)e * 4u : SPX_OOB, 0);
        }
    };

    auto commit = [&](Stage& st) {
#pragma unroll
        for (int i = 0; i < FP; ++i) {
            const int off = (i * SPX_BK_THREADS + tid) * 16;
            if (off < FBYTES) {
                if (DO_P) *(u32x4*)(Gs + off) = st.gr[i];
                if (DO_W) {
                    // the activation blob holds block-scaled int16 codes: every element becomes a bf16 hi + lo pair HERE, with
                    // the whole workgroup and in the blob's own lane order (the split is elementwise), so the waves of the
                    // head product only read fragments
                    u32x4 ahw, alw;
                    const float asc = __builtin_amdgcn_ldexpf(1.0f / SPX_ABLOB_I16_ONE, (int)((st.exw[i] >> 8) & 0xffu) - 128);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x2 v;
                        v[0] = (float)(short)(st.ar[i][j] & 0xffffu);            // 15 bits + sign: hi + lo is exact
                        v[1] = (float)((int)st.ar[i][j] >> 16);
                        uint32_t hi, lo;
                        split_bf16x2(v * asc, hi, lo);
                        ahw[j] = hi;
                        alw[j] = lo;
                    }
                    *(u32x4*)(As + off) = ahw;
                    *(u32x4*)(As2 + off) = alw;
                }
            }
        }
        if (DO_P && tid == 0) ebuf[cur_buf] = st.gexp;
        if (RAG && st.ragged) {                              // chunk-uniform: shift the moved-back window into place
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                if (XF32) spx_shr256(st.xr[i][0], st.xr[i][XF32 ? 1 : 0], st.rot_bits);
                else st.xr[i][0] = spx_shr128(st.xr[i][0], st.rot_bits);
            }
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            u32x4 v;
            if (XF32) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f32x2 p;                        // the forward's bf16 rounding first, then fp16 (exact; saturating)
                    p[0] = (float)(__bf16)__uint_as_float(st.xr[i][e >> 1][(2 * e) & 3]);
                    p[1] = (float)(__bf16)__uint_as_float(st.xr[i][e >> 1][(2 * e + 1) & 3]);
                    v[e] = pack_f16x2_rtz(p);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = pack_f16x2_rtz(unpack_bf16x2(st.xr[i][0][e]));
            }
            if (DO_P) *(u32x4*)(Xs + (prow + RPP * i) * SPX_BK_ROW + piece * 16) = v;
        }
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int e = i * SPX_BK_THREADS + tid;
            if (DO_W && e < SPX_BK_PX * K) {
                const int p = e / K, cls = e - p * K;
                __bf16 hi, lo;
                split_bf16(st.lr_[i], hi, lo);
                *(uint16_t*)(Ls + cls * SPX_BK_ROW + p * 2) = __builtin_bit_cast(uint16_t, hi);
                *(uint16_t*)(Ls2 + cls * SPX_BK_ROW + p * 2) = __builtin_bit_cast(uint16_t, lo);
            }
        }
    };

    // padded class rows of the dLogits^T image stay zero for the whole kernel
    for (int buf = 0; buf < (PIPE ? 2 : 1); ++buf) {
        use_buffer(buf);
        for (int e = tid; DO_W && e < NCB * 32 * SPX_BK_PX; e += SPX_BK_THREADS) {
            const int cls = e / SPX_BK_PX, p = e - cls * SPX_BK_PX;
            if (cls >= K) {
                *(uint16_t*)(Ls + cls * SPX_BK_ROW + p * 2) = 0;
                *(uint16_t*)(Ls2 + cls * SPX_BK_ROW + p * 2) = 0;
            }
        }
    }
    use_buffer(0);

    // transposed-read lane map of an A fragment (rows = prototypes of block pb, k = 16 px of a k-step):
    // 16-lane group g: prototype sub-block s2 = g & 1, k-half g >> 1; lane 4 qq + pp: pixel row qq, prototype quad pp
    const int tg = lane >> 4, tli = lane & 15, tqq = tli >> 2, tpp = tli & 3;
    const int ts2 = tg & 1, tkh = tg >> 1;

    // d_bank part of one pixel k-step: this wave's prototype blocks x channel blocks 2 cp, 2 cp + 1
    // bring the accumulators to the units of the chunk about to be added (a power-of-two multiply, exact); returns false for
    // a chunk whose gradients are more than 2^80 below what the accumulators are scaled for: it could not change them
    auto chunk_scale = [&]() -> bool {
        const int e_c = __builtin_amdgcn_readfirstlane(ebuf[cur_buf]);
        if (have_e && e_c == e_acc) return true;
        if (have_e && e_c - e_acc > 80) return false;
        const float ratio = have_e ? __builtin_amdgcn_ldexpf(1.0f, e_c - e_acc) : 1.0f;
#pragma unroll
        for (int i = 0; i < PH; ++i) {
            csum[i] *= ratio;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) accp[i][t][e] *= ratio;
        }
        e_acc = e_c;
        have_e = true;
        return true;
    };
    auto bank_part = [&](int ks, int cp) {
        const int koff = (ks * 16 + 8 * h) * 2;
        // this lane's pixel rows of the k-step: px = 16 ks + 8 tkh + tqq (+4): kernel-1 wave px >> 5, lane px & 31
        const int pxa = ks * 16 + 8 * tkh + tqq;
        const int wsel = pxa >> 5, ra = pxa & 31;
        const int fo0 = spx_blob_slot(ra, tpp & 1, ts2) * 16 + 8 * (tpp >> 1);
        const int fo1 = spx_blob_slot(ra + 4, tpp & 1, ts2) * 16 + 8 * (tpp >> 1);
        f16x8 xb[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) xb[t] = *(const f16x8*)(Xs + ((2 * cp + t) * 32 + r) * SPX_BK_ROW + koff);
#pragma unroll
        for (int i = 0; i < PH; ++i) {
            const int fb = ((wsel * NPB + pb0 + i) * 2 + ts2) * 1024;
            const s16x4 g0 = lds_tr_read(Gs + fb + fo0);
            const s16x4 g1 = lds_tr_read(Gs + fb + fo1);
            const f16x8 gf = __builtin_bit_cast(f16x8, __builtin_shufflevector(g0, g1, 0, 1, 2, 3, 4, 5, 6, 7));
            if (ksplit || cpair == cs_owner(i)) {   // colsum(G): every lane holds 8 px of its prototype row; lanes r, r+32 cover the k-step
                f16x2 one2;
                one2[0] = (_Float16)1.0f;
                one2[1] = (_Float16)1.0f;
                float s8 = csum[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f16x2 pr;                   // (element by element: a dword taken from the transposed read's result through a
                    pr[0] = gf[2 * e];          // vector bit-cast came out as dword 0 / 2 twice - hipcc 7.2)
                    pr[1] = gf[2 * e + 1];
                    s8 = __builtin_amdgcn_fdot2(pr, one2, s8, false);
                }
                csum[i] = s8;
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) accp[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gf, xb[t], accp[i][t], 0, 0, 0);
        }
    };
    // d_W part of one pixel k-step (waves 0 .. NPB-1: prototype block `wave`, every class block)
    auto head_part = [&](int ks) {
        const int koff = (ks * 16 + 8 * h) * 2;
        const int pxa = ks * 16 + 8 * tkh + tqq;
        const int wsel = pxa >> 5, ra = pxa & 31;
        const int fo0 = spx_blob_slot(ra, tpp & 1, ts2) * 16 + 8 * (tpp >> 1);
        const int fo1 = spx_blob_slot(ra + 4, tpp & 1, ts2) * 16 + 8 * (tpp >> 1);
        const int fb = ((wsel * NPB + wave) * 2 + ts2) * 1024;
        // a = hi + lo (split at commit time): d_W = (hi + lo) . (dl_hi + dl_lo) carries ~2^-12 of a's fp16 rounding
        // instead of bf16's 2^-9
        const bf16x4 a0 = __builtin_bit_cast(bf16x4, lds_tr_read(As + fb + fo0));
        const bf16x4 a1 = __builtin_bit_cast(bf16x4, lds_tr_read(As + fb + fo1));
        const bf16x4 b0 = __builtin_bit_cast(bf16x4, lds_tr_read(As2 + fb + fo0));
        const bf16x4 b1 = __builtin_bit_cast(bf16x4, lds_tr_read(As2 + fb + fo1));
        const bf16x8 af = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8 af2 = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const bf16x8 lf = *(const bf16x8*)(Ls + (cb * 32 + r) * SPX_BK_ROW + koff);
            const bf16x8 lf2 = *(const bf16x8*)(Ls2 + (cb * 32 + r) * SPX_BK_ROW + koff);
            accw[cb] = mfma_bf16(af, lf, accw[cb]);
            accw[cb] = mfma_bf16(af2, lf, accw[cb]);
            accw[cb] = mfma_bf16(af, lf2, accw[cb]);
        }
    };
    auto compute = [&]() {
        const bool add_p = DO_P && chunk_scale();
        if constexpr (KSPLIT) {
            if (add_p) bank_part(cpair, 0);
            if (DO_W && w_role) {
#pragma unroll
                for (int ks = 0; ks < SPX_BK_PX / 16; ++ks) head_part(ks);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < SPX_BK_PX / 16; ++ks) {
                if (add_p) bank_part(ks, cpair);
                if (DO_W && w_role) head_part(ks);
            }
        }
    };
    if constexpr (!PIPE) {
        if (c_begin < c_end) issue(stg[0]);
        for (long long c = c_begin; c < c_end; c += cstep) {
            commit(stg[0]);
            __syncthreads();
            issue(stg[0]);                               // always issue (branch-free); past the end the last chunk is re-read and ignored
            __builtin_amdgcn_sched_barrier(0);
            compute();
            __syncthreads();
        }
    } else if (c_begin < c_end) {
        // three-stage software pipeline: chunk c computes from one LDS buffer | chunk c+1 is converted into the other |
        // the loads of chunks c+1 / c+2 are in flight in two register sets; one barrier per chunk.  Indices past the
        // walk's end re-read the last chunk (branch-free, results unused).
        //
        // SPX_BANK_STAGGER: waves w and w + 4 share a SIMD and run the same program.  The upper four ("late") take the two
        // phases of every barrier interval in the opposite order - conversion first, MFMA phase second - so one partner's
        // VALU work runs under the other's MFMA chains:
        //     early:  C0 M1 | C1 M2 | C2 M3 | ...          (C = MFMA phase of a chunk, M = conversion of a chunk, | = barrier)
        //     late:   M1 C0 | M2 C1 | M3 C2 | ...
        // In every interval all waves read one buffer and write the other, so both orders are legal.  The late stream is the
        // SAME rolled body (C, then M) with the barrier between the two instead of after them and the conversions running
        // one chunk further ahead (one pre-rolled M1); its two load sets are named the other way round, so the body indexes
        // the register sets statically.  Every wave executes one barrier per interval.
        const bool late = SPX_BANK_STAGGER && wave >= 4;
        const int la = late ? 1 : 0;                 // how many chunks further ahead this wave converts
        issue(stg[0]);
        commit(stg[0]);                              // buffer 0 <- the first chunk
        if (!late) {
            issue(stg[0]);                           // chunk 1
            issue(stg[1]);                           // chunk 2
        } else {
            issue(stg[1]);                           // chunk 1
            issue(stg[0]);                           // chunk 2
            use_buffer(1);
            commit(stg[1]);                          // M1, before the first barrier
            issue(stg[1]);                           // chunk 3
        }
        __syncthreads();
        for (long long c = c_begin; c < c_end; c += 2 * cstep) {
            use_buffer(0);
            compute();                               // chunk c
            if (late) __syncthreads();
            use_buffer(1 - la);
            commit(stg[0]);                          // chunk c + 1 (late: c + 2)
            issue(stg[0]);                           // chunk c + 3 (late: c + 4)
            if (!late) __syncthreads();
            if (c + cstep < c_end) {                 // workgroup-uniform
                use_buffer(1);
                compute();                           // chunk c + 1
                if (late) __syncthreads();
                use_buffer(la);
                commit(stg[1]);                      // chunk c + 2 (late: c + 3)
                issue(stg[1]);                       // chunk c + 4 (late: c + 5)
                if (!late) __syncthreads();
            }
        }
    }

    // ---- k-step split: sum the four cpair partials of each prototype half through LDS (staging is dead now) ----
    if (DO_P && ksplit && want_p) {
        float* const red = (float*)smem;                       // [half][cpair-1][t][reg][lane]
        float* const redc = red + 2 * 3 * 2 * 1024;            // [half][cpair-1][lane]
        const int half = wave >> 2;
#pragma unroll
        for (int i = 0; i < PH; ++i) {
            __syncthreads();
            if (cpair > 0) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        red[(((half * 3 + cpair - 1) * 2 + t) * 16 + reg) * 64 + lane] = accp[i][t][reg];
                redc[(half * 3 + cpair - 1) * 64 + lane] = csum[i];
            }
            __syncthreads();
            if (cpair == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {          // fixed order: results stay run-to-run identical
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg)
                            accp[i][t][reg] += red[(((half * 3 + c) * 2 + t) * 16 + reg) * 64 + lane];
                    csum[i] += redc[(half * 3 + c) * 64 + lane];
                }
            }
        }
    }

    // ---- write this workgroup's partial slab ----
    const int ws = spx_bk_wstride(pl);
    float* slab = a.workspace + ((size_t)(a.slab_first + split) * pl.npanels + q) * rows * ws;
    const float sinv_acc = have_e ? __builtin_amdgcn_ldexpf(1.0f, -e_acc) : 0.0f;      // accumulator units -> true values
    if (want_p && (!ksplit || cpair == 0)) {
#pragma unroll
        for (int i = 0; i < PH; ++i) {
            const int pb = pb0 + i;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int chb = ksplit ? t : 2 * cpair + t;
                if (chb < nchb) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        slab[(size_t)(pb * 32 + acc_row(reg, h)) * ws + chb * 32 + r] = accp[i][t][reg] * sinv_acc;
                }
            }
            if (ksplit ? cpair == 0 : cpair == cs_owner(i)) {
                const float s = csum[i] + __shfl_xor(csum[i], 32);
                if (h == 0) slab[(size_t)(pb * 32 + r) * ws + nchb * 32 + NCB * 32] = s * sinv_acc;
            }
        }
    }
    if (want_w && w_role) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                slab[(size_t)(wave * 32 + acc_row(reg, h)) * ws + nchb * 32 + cb * 32 + r] = accw[cb][reg];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// kernel 2, LDS-DMA form (spx_bank_dma_kernel): the d_bank product for bf16 features whose pixel count is a multiple of 64,
// scales wider than 64 channels, no d_W duty (heads of one class block: d_W comes from kernel 1's tile partials).
//
// What bounds this product is how X is FETCHED (timing-only builds of this kernel, round 4, north star, the 1.07 GB X stream
// alone): as the 64-B row pieces of a 32-pixel chunk 0.43 ms (2.5 TB/s), as 128-B pieces 0.28 ms, as 256-B pieces 0.22 ms;
// the 0.8 GB of G fragments (contiguous) 0.19 ms; the MFMA phase with its operand reads 0.41 ms.  The register-staged kernel
// above reads 64-B pieces (two chunks of loads in flight is all its 256 registers hold, and its LDS stage is one chunk).
// Here NO operand passes through registers on its way in: everything arrives by LDS-DMA (buffer_load ... lds, 1 KiB per wave
// instruction).  X comes in UNITS of 64 pixels - 128-B row pieces, whole cache lines - into a ring of three 32-KiB images;
// the G fragments (verbatim) per 32-pixel chunk into a ring of four; a step = one chunk = the MFMA phase of the register-
// staged kernel.  Per step a wave issues its pieces of the G chunk three steps ahead and of half an X unit two units ahead,
// waits with a counted vmcnt for everything issued three or more steps ago (the pieces of the last two steps stay in flight),
// and the one barrier per step publishes every wave's pieces and frees the slots of the step before.  X becomes fp16 (the G
// plane's MFMA type; exact for bf16 values; round toward zero saturates instead of overflowing) while its B fragments are
// read, not in a commit pass.  An LDS-DMA writes lane-linear, so the X image is plain [row][128 B]; its bank swizzle (16-B
// piece p of row r at position p ^ ((r >> 1) & 7): conflict-free ds_read_b128 of one piece index by 32 rows) is applied on
// the SOURCE address.  Same wave roles, accumulators, exponent handling, slab layout and fixed summation order as above.
// ------------------------------------------------------------------------------------------------
#define SPX_BD_GSLOTS 4
#define SPX_BD_XSLOTS 3
#define SPX_BD_XUNIT (256 * 128)
template <int NPB>
__host__ __device__ constexpr int spx_bd_lds_bytes() {
    // G ring, X ring, per G slot the piece that carries the tile's fp16 exponent, the landing pad of dropped pieces
    return SPX_BD_GSLOTS * NPB * 2 * 1024 + SPX_BD_XSLOTS * SPX_BD_XUNIT + (SPX_BD_GSLOTS + 1) * 1024;
}

template <int NPB>
__global__ __launch_bounds__(SPX_BK_THREADS, 2) void spx_bank_dma_kernel(const SpxBankBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const spx_plan& pl = a.plan;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int q = blockIdx.y, split = blockIdx.x;
    const int Cs = pl.channels_per_scale;
    const int C = pl.num_scales * Cs;
    const int nchb = (Cs + 31) / 32;
    constexpr int rows = NPB * 32;
    constexpr int GB = NPB * 2 * 1024;
    constexpr int NGP = NPB * 2 + 1;                   // 1-KiB pieces of a G chunk: its fragments + the exponent piece
    constexpr int PWG = (NGP + 7) / 8;                 // ... per wave (the surplus ones are dropped loads)
    constexpr int PW = PWG + 2;                        // + this wave's two 8-row pieces of half an X unit: DMAs per wave and step
    char* const gring = smem;
    char* const xring = smem + SPX_BD_GSLOTS * GB;
    char* const epad = xring + SPX_BD_XSLOTS * SPX_BD_XUNIT;   // [G slot][1 KiB]: word 0 = the fp16 exponent of the chunk's kernel-1 tile
    char* const trash = epad + SPX_BD_GSLOTS * 1024;
    const int ch0 = pl.panel_ch0[q];
    const int tiles_per_img = (a.HW + SPX_TILE_PX - 1) / SPX_TILE_PX;
    const size_t ntiles = (size_t)a.B * tiles_per_img;
    const int nui = a.nci_launch;                      // 64-px units per image (the launcher counts exactly those)
    const long long total = (long long)a.B * nui;
    const long long cstep = a.nslabs;
    const long long u_begin = split;
    const int n_units = u_begin < total ? (int)((total - u_begin + cstep - 1) / cstep) : 0;
    const int n_steps = 2 * n_units;
    const size_t blob_total = (size_t)pl.npanels * ntiles * 4 * NPB * 2 * 1024;
    const int32_t* const gexp = (const int32_t*)((const char*)a.g_in + spx_gexp_offset(blob_total)) + (size_t)q * ntiles;

    // wave roles (as in spx_bank_bwd_kernel)
    const int cpair = wave & 3;
    constexpr int PH = NPB / 2;
    const int pb0 = (wave >> 2) * PH;
    const int cs_first = wave >> 2;
    auto cs_owner = [&](int i) { return i == 0 ? cs_first : i + 1; };
    f32x16 accp[PH][2];
    float csum[PH];
    int e_acc = 0;
    bool have_e = false;
#pragma unroll
    for (int i = 0; i < PH; ++i) {
        csum[i] = 0.0f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) accp[i][t][e] = 0.0f;
    }
    // rows past the scale's channels are never written by a DMA: they stay zero for the whole launch
    for (int o = tid * 16; o < spx_bd_lds_bytes<NPB>(); o += SPX_BK_THREADS * 16) *(u32x4*)(smem + o) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();

    // buffer_load_dwordx4 ... lds through inline asm: hipcc then keeps its own wait-count bookkeeping out of it (through the
    // builtin it waits vmcnt(0) in front of LDS reads it cannot tell apart from the DMA's target); ordering is this kernel's:
    // counted vmcnt + barrier below.  Descriptor as make_rsrc_pred builds it; M0 = the LDS byte address of the piece.
    auto dma = [&](const char* base, uint32_t vo, char* dst) {
        const uint64_t ba = (uint64_t)base;
        u32x4 rs;
        rs[0] = __builtin_amdgcn_readfirstlane((uint32_t)ba);
        rs[1] = __builtin_amdgcn_readfirstlane((uint32_t)(ba >> 32) & 0xffffu);
        rs[2] = 0x80000000u;
        rs[3] = 0x00020000u;
        const uint32_t la = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)dst);
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(la), "v"(vo), "s"(rs) : "memory");
    };
    // a cursor over this workgroup's units u_begin, u_begin + cstep, ...: (image, unit in image), advanced without a division;
    // past the walk's end it stays on the last unit (those pieces land in slots nobody reads any more)
    struct Cursor {
        int k, b, ui;
    };
    const int step_b = (int)(cstep / nui), step_ui = (int)(cstep - (long long)step_b * nui);
    auto advance = [&](Cursor& c) {
        ++c.k;
        if (c.k < n_units) {
            c.b += step_b;
            c.ui += step_ui;
            if (c.ui >= nui) {
                c.ui -= nui;
                ++c.b;
            }
        }
    };
    Cursor cg, cx;
    cg.k = cx.k = 0;
    cg.b = cx.b = (int)(u_begin / nui);
    cg.ui = cx.ui = (int)(u_begin - (long long)cg.b * nui);
    // the G chunk (unit cursor cg, half e) into G slot gs: fragments of kernel-1 wave 2 (ui & 1) + e of tile ui >> 1, and the
    // tile's exponent (an ordinary load of it would make hipcc drain every DMA in flight at its use: lane 0 fetches 16 bytes
    // from the tile's word on - the scratch is padded behind the array -, the other lanes drop)
    auto issue_g = [&](int gs, int e) {
        const size_t tile_g = (size_t)cg.b * tiles_per_img + (cg.ui >> 1);
        const char* const gsrc = (const char*)a.g_in + ((((size_t)q * ntiles + tile_g) * 4 + 2 * (cg.ui & 1) + e) * NPB * 2) * 1024;
#pragma unroll
        for (int i = 0; i < PWG; ++i) {
            const int p = wave + 8 * i;                 // wave-uniform
            const bool is_g = p < NPB * 2, is_e = p == NPB * 2;
            dma(is_g ? gsrc + (size_t)p * 1024 : (const char*)(gexp + tile_g),
                is_g ? (uint32_t)lane * 16u : ((is_e && lane == 0) ? 0u : SPX_OOB),
                is_g ? gring + gs * GB + p * 1024 : (is_e ? epad + gs * 1024 : trash));
        }
    };
    // half e of the X unit (cursor cx) into X slot xs: 16 pieces of 8 rows x 128 B, two per wave.  Lane l <-> row 8 p + (l >> 3),
    // position l & 7 of the row's eight 16-B pieces; the piece it fetches is the one that belongs there under the swizzle.
    auto issue_x = [&](int xs, int e) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = 16 * e + wave + 8 * i;        // piece of the unit: rows 8 p .. 8 p + 7
            const int row = 8 * p + (lane >> 3);
            const int px = cx.ui * 64 + 8 * ((lane & 7) ^ ((row >> 1) & 7));
            dma((const char*)a.x + ((size_t)cx.b * C + ch0 + 8 * p) * a.HW * 2,
                (row < Cs && px + 8 <= a.HW) ? ((uint32_t)(lane >> 3) * (uint32_t)a.HW + (uint32_t)px) * 2u : SPX_OOB,
                xring + xs * SPX_BD_XUNIT + p * 1024);
        }
    };

    const int tg = lane >> 4, tli = lane & 15, tqq = tli >> 2, tpp = tli & 3;
    const int ts2 = tg & 1, tkh = tg >> 1;
    // the MFMA phase of one 32-pixel chunk: G slot gs, half e of X slot xs
    auto compute = [&](int gs, int xs, int e) {
        const char* const Gs = gring + gs * GB;
        const char* const Xs = xring + xs * SPX_BD_XUNIT;
        // bring the accumulators to the units of the chunk about to be added (see chunk_scale above)
        const int e_c = __builtin_amdgcn_readfirstlane(*(const int32_t*)(epad + gs * 1024));
        if (have_e && e_c - e_acc > 80) return;
        if (!have_e || e_c != e_acc) {
            const float ratio = have_e ? __builtin_amdgcn_ldexpf(1.0f, e_c - e_acc) : 1.0f;
#pragma unroll
            for (int i = 0; i < PH; ++i) {
                csum[i] *= ratio;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int k = 0; k < 16; ++k) accp[i][t][k] *= ratio;
            }
            e_acc = e_c;
            have_e = true;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int pxa = ks * 16 + 8 * tkh + tqq;
            const int fo0 = spx_blob_slot(pxa, tpp & 1, ts2) * 16 + 8 * (tpp >> 1);
            const int fo1 = spx_blob_slot(pxa + 4, tpp & 1, ts2) * 16 + 8 * (tpp >> 1);
            f16x8 xb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const u32x4 raw = *(const u32x4*)(Xs + ((2 * cpair + t) * 32 + r) * 128 + (((4 * e + 2 * ks + h) ^ ((r >> 1) & 7)) * 16));
                u32x4 w;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    w[k] = pack_f16x2_rtz(unpack_bf16x2(raw[k]));
                xb[t] = __builtin_bit_cast(f16x8, w);
            }
#pragma unroll
            for (int i = 0; i < PH; ++i) {
                const int fb = ((pb0 + i) * 2 + ts2) * 1024;
                const s16x4 g0 = lds_tr_read(Gs + fb + fo0);
                const s16x4 g1 = lds_tr_read(Gs + fb + fo1);
                const f16x8 gf = __builtin_bit_cast(f16x8, __builtin_shufflevector(g0, g1, 0, 1, 2, 3, 4, 5, 6, 7));
                if (cpair == cs_owner(i)) {
                    f16x2 one2;
                    one2[0] = (_Float16)1.0f;
                    one2[1] = (_Float16)1.0f;
                    float s8 = csum[i];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        f16x2 pr;                   // (element by element: see bank_part above)
                        pr[0] = gf[2 * k];
                        pr[1] = gf[2 * k + 1];
                        s8 = __builtin_amdgcn_fdot2(pr, one2, s8, false);
                    }
                    csum[i] = s8;
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) accp[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gf, xb[t], accp[i][t], 0, 0, 0);
            }
        }
    };
    if (n_steps > 0) {
        // run-in: the first unit whole, then the G chunks of steps 1 and 2 and the second unit (issue order = what the first
        // counted wait lets stay in flight)
        issue_g(0, 0);
        issue_x(0, 0);
        issue_x(0, 1);
        issue_g(1, 1);
        advance(cg);
        issue_g(2, 0);
        advance(cx);
        issue_x(1, 0);
        issue_x(1, 1);
        // cg: unit 1, its half 1 goes next (step 0 issues the chunk of step 3); cx: unit 2 next
        advance(cx);
        int gs = 0, xs = 0;
        for (int s = 0; s < n_steps; ++s) {
            const int e = s & 1;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");   // own pieces of steps <= s - ... landed (two steps' worth stay in flight)
            __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): this wave's LDS reads of the step before are done
            __builtin_amdgcn_s_barrier();
            // the chunk of step s + 3 (half (s + 3) & 1 = 1 - e of the G cursor's unit) and half e of the unit two ahead
            issue_g((gs + 3) & 3, 1 - e);
            if (e == 0) advance(cg);                    // (step s + 4 starts the next unit)
            issue_x(xs >= 1 ? xs - 1 : 2, e);           // X slot (j + 2) % 3 = (xs + 2) % 3
            if (e == 1) advance(cx);
            compute(gs, xs, e);
            gs = (gs + 1) & 3;
            if (e == 1) xs = xs == 2 ? 0 : xs + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the run-ahead pieces of the walk's end: landed before the LDS is released
    }

    // ---- write this workgroup's partial slab (layout of spx_bank_bwd_kernel) ----
    if (!a.d_bank) return;
    const int ws = spx_bk_wstride(pl);
    float* slab = a.workspace + ((size_t)(a.slab_first + split) * pl.npanels + q) * rows * ws;
    const float sinv_acc = have_e ? __builtin_amdgcn_ldexpf(1.0f, -e_acc) : 0.0f;
#pragma unroll
    for (int i = 0; i < PH; ++i) {
        const int pb = pb0 + i;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int chb = 2 * cpair + t;
            if (chb < nchb) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    slab[(size_t)(pb * 32 + acc_row(reg, h)) * ws + chb * 32 + r] = accp[i][t][reg] * sinv_acc;
            }
        }
        if (cpair == cs_owner(i)) {
            const float s = csum[i] + __shfl_xor(csum[i], 32);
            if (h == 0) slab[(size_t)(pb * 32 + r) * ws + nchb * 32 + pl.ncb * 32] = s * sinv_acc;
        }
    }
}

// kernel 3: fixed-order sum of the slabs, + the p * colsum(G) term.  Four independent partial sums per output
// (slab j goes to partial j & 3) keep several loads in flight per thread; the order is fixed, so results are
// run-to-run identical.
#define SPX_RED_ELEMS 64     // output elements per workgroup (a wave reads 256-B row pieces: 128-B pieces streamed at half the rate)
#define SPX_RED_PARTS 4      // slab ranges summed in parallel per element (256 threads)
#define SPX_RED_FLY 8        // slabs in flight per thread
__global__ __launch_bounds__(SPX_RED_ELEMS * SPX_RED_PARTS) void spx_bank_reduce_kernel(const SpxBankBwdArgs a) {
    __shared__ float red_s[SPX_RED_PARTS][SPX_RED_ELEMS], red_c[SPX_RED_PARTS][SPX_RED_ELEMS];
    const spx_plan& pl = a.plan;
    const int Cs = pl.channels_per_scale, K = pl.num_classes, P = pl.num_prototypes;
    const int nchb = (Cs + 31) / 32;
    const int rows = pl.npb * 32;
    const int ws = spx_bk_wstride(pl);
    const int ncols = Cs + K;
    const long long n = (long long)pl.npanels * rows * ncols;
    const int el = threadIdx.x % SPX_RED_ELEMS, part = threadIdx.x / SPX_RED_ELEMS;
    const long long gid = (long long)blockIdx.x * SPX_RED_ELEMS + el;
    const bool in = gid < n;
    const int col = in ? (int)(gid % ncols) : 0;
    const int row = in ? (int)((gid / ncols) % rows) : 0;
    const int q = in ? (int)(gid / ((long long)ncols * rows)) : 0;
    const bool is_p = col < Cs;
    const bool live = in && row < pl.panel_np[q] && (is_p ? a.d_bank != nullptr : a.d_W != nullptr);
    const size_t slab_stride = (size_t)pl.npanels * rows * ws;
    const float* base = a.workspace + ((size_t)q * rows + row) * ws;
    const int c0 = is_p ? col : nchb * 32 + (col - Cs);
    const int c1 = nchb * 32 + pl.ncb * 32;          // colsum column
    // this thread's slab range; inside it slab j goes to partial (j - j0) % SPX_RED_FLY: a fixed order, so results are
    // run-to-run identical.  The colsum column is read unconditionally (head columns ignore it): no load sits under a branch.
    const int per = (a.nsplit + SPX_RED_PARTS - 1) / SPX_RED_PARTS;
    const int j0 = part * per, j1 = min(a.nsplit, j0 + per);
    float s[SPX_RED_FLY], cs[SPX_RED_FLY];
#pragma unroll
    for (int u = 0; u < SPX_RED_FLY; ++u) s[u] = cs[u] = 0.0f;
    if (live) {
        int j = j0;
        for (; j + SPX_RED_FLY <= j1; j += SPX_RED_FLY) {
            float v[SPX_RED_FLY], w[SPX_RED_FLY];
#pragma unroll
            for (int u = 0; u < SPX_RED_FLY; ++u) {
                v[u] = base[(size_t)(j + u) * slab_stride + c0];
                w[u] = base[(size_t)(j + u) * slab_stride + c1];
            }
#pragma unroll
            for (int u = 0; u < SPX_RED_FLY; ++u) {
                s[u] += v[u];
                cs[u] += w[u];
            }
        }
        for (; j < j1; ++j) {
            s[(j - j0) % SPX_RED_FLY] += base[(size_t)j * slab_stride + c0];
            cs[(j - j0) % SPX_RED_FLY] += base[(size_t)j * slab_stride + c1];
        }
    }
    red_s[part][el] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    red_c[part][el] = ((cs[0] + cs[1]) + (cs[2] + cs[3])) + ((cs[4] + cs[5]) + (cs[6] + cs[7]));
    __syncthreads();
    if (part != 0 || !live) return;
    float st = 0.0f, ct = 0.0f;
#pragma unroll
    for (int i = 0; i < SPX_RED_PARTS; ++i) {
        st += red_s[i][el];
        ct += red_c[i][el];
    }
    const int p = pl.panel_p0[q] + row;
    if (is_p) {
        a.d_bank[(size_t)p * Cs + col] = 2.0f * (a.bank[(size_t)p * Cs + col] * ct - st);
    } else {
        // the head scale kernel 1 left behind its tile partials / its activation blob (spx_common.h)
        const size_t ntl = (size_t)a.B * ((a.HW + SPX_TILE_PX - 1) / SPX_TILE_PX);
        const size_t off = pl.ncb == 1 ? spx_dw_partial_bytes(pl.npanels, ntl, pl.npb, K) : spx_ablob_scale_offset(ntl * pl.npanels * 4 * pl.npb * 2 * 1024);
        a.d_W[(size_t)(col - Cs) * P + p] = *(const float*)((const char*)a.a_in + off) * st;
    }
}

// d_W of the heads with one class block, level 1: kernel 1 left one fp32 partial [block][K][32 prototypes] per (panel, tile)
// (spx_bwd_impl.h, "d_W stage").  Workgroup (j, q) adds the partials of tiles j, j + nsplit, ... of panel q - a fixed set in
// a fixed order (eight independent partial sums per output, combined pairwise) - into the d_W columns of slab j, which the
// parameter kernel leaves alone for these heads; kernel 3 then sums the slabs as it does for d_bank.  A streaming read of
// 128 K bytes per block and tile (north star: 239 MB).
__global__ __launch_bounds__(256) void spx_dw_reduce_kernel(const SpxBankBwdArgs a) {
    const spx_plan& pl = a.plan;
    const int j = blockIdx.x, q = blockIdx.y;
    const int K = pl.num_classes, NPB = pl.npb;
    const int nchb = (pl.channels_per_scale + 31) / 32;
    const int rows = NPB * 32, ws = spx_bk_wstride(pl);
    const long long ntiles = (long long)a.B * ((a.HW + SPX_TILE_PX - 1) / SPX_TILE_PX);
    const int nvb = (pl.panel_np[q] + 31) >> 5;                  // blocks of the panel holding >= 1 real prototype
    const int nvec = nvb * K * 8;                                // 16-B vectors of a partial's valid blocks (they are contiguous)
    const size_t tstride = (size_t)NPB * K * 32;                 // floats per (panel, tile)
    const float* const base = (const float*)a.a_in + (size_t)q * ntiles * tstride;
    float* const slab = a.workspace + ((size_t)j * pl.npanels + q) * rows * ws;
    for (int v = threadIdx.x; v < nvec; v += 256) {
        f32x4 s[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        long long t = j;
        for (; t + 7ll * a.nsplit < ntiles; t += 8ll * a.nsplit) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += *(const f32x4*)(base + (size_t)(t + (long long)u * a.nsplit) * tstride + (size_t)v * 4);
        }
        for (int u = 0; t < ntiles; t += a.nsplit, ++u) s[u] += *(const f32x4*)(base + (size_t)t * tstride + (size_t)v * 4);
        const f32x4 tot = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
        const int pb = v / (K * 8), rem = v - pb * (K * 8), cls = rem >> 3, p4 = rem & 7;
#pragma unroll
        for (int e = 0; e < 4; ++e) slab[(size_t)(pb * 32 + p4 * 4 + e) * ws + nchb * 32 + cls] = tot[e];
    }
}

// 32-px chunks with two LDS buffers and two register load sets (CPX = 32: a three-stage pipeline, one barrier per chunk).
// History of the measurements on MI355X (north-star shape, A/B on one box each): on its own the pipeline equals or loses to the
// 64-px single-buffered loop (0.865 vs 0.865 ms early in round 2, 0.832 vs 0.742 later: more barriers and shorter MFMA chains per
// chunk); WITH the wave stagger (partners on a SIMD take the two phases in opposite order) it hides the conversion VALU of the
// commit phase under the MFMA chains: 0.721-0.730 vs 0.742 with the fp16 activation blob, 0.737-0.742 vs 0.783 with the int16
// one - so both are on for the 6-block panels that do not use the k-step split.  (The PMC fetch count of the pipelined kernel is
// 1.07 GB higher: the two 64-B halves of every 128-B line of an X row go to two workgroups, i.e. two L2s.  Giving both halves to ONE
// workgroup - walk units of two consecutive 32-px chunks - was tried: 0.74 -> 0.94 ms, the workgroups running at one time then touch
// only every other 64 B of a twice as long stretch.  The second fetch of a line is served by the memory-side cache.)
template <int NPB, int NCB, bool DO_P, bool DO_W, bool KSPLIT>
static hipError_t launch_bank_k(const SpxBankBwdArgs& a, int x_dtype, dim3 grid, hipStream_t s) {
    // 32-px double-buffered chunks wherever the k-step split is not in use and two buffers fit the LDS
    // (one-class-block heads only: with 32 more d_W accumulators the pipelined instance spills)
    constexpr int CPX = (SPX_BANK_PIPE && !KSPLIT && NPB == 6 && NCB == 1 && 2 * spx_bk_buf_bytes<NPB, NCB, DO_P, DO_W, 32>() <= SPX_LDS_LIMIT) ? 32 : 64;
    constexpr size_t lds = (size_t)spx_bk_lds_bytes<NPB, NCB, DO_P, DO_W, CPX>();
    static_assert(lds <= SPX_LDS_LIMIT, "bank kernel LDS");
    if (x_dtype == 1) {
        if (a.vec_ok == 2) hipLaunchKernelGGL((spx_bank_bwd_kernel<NPB, NCB, true, 2, DO_P, DO_W, KSPLIT, CPX>), grid, dim3(SPX_BK_THREADS), lds, s, a);
        else if (a.vec_ok) hipLaunchKernelGGL((spx_bank_bwd_kernel<NPB, NCB, true, 1, DO_P, DO_W, KSPLIT, CPX>), grid, dim3(SPX_BK_THREADS), lds, s, a);
        else hipLaunchKernelGGL((spx_bank_bwd_kernel<NPB, NCB, true, 0, DO_P, DO_W, KSPLIT, CPX>), grid, dim3(SPX_BK_THREADS), lds, s, a);
    } else {
        if (a.vec_ok == 2) hipLaunchKernelGGL((spx_bank_bwd_kernel<NPB, NCB, false, 2, DO_P, DO_W, KSPLIT, CPX>), grid, dim3(SPX_BK_THREADS), lds, s, a);
        else if (a.vec_ok) hipLaunchKernelGGL((spx_bank_bwd_kernel<NPB, NCB, false, 1, DO_P, DO_W, KSPLIT, CPX>), grid, dim3(SPX_BK_THREADS), lds, s, a);
        else hipLaunchKernelGGL((spx_bank_bwd_kernel<NPB, NCB, false, 0, DO_P, DO_W, KSPLIT, CPX>), grid, dim3(SPX_BK_THREADS), lds, s, a);
    }
    return hipGetLastError();
}
template <int NPB, int NCB, bool DO_P, bool DO_W>
static hipError_t launch_bank_pw(const SpxBankBwdArgs& a, int x_dtype, dim3 grid, hipStream_t s) {
    // <= 64 channels per scale and a d_bank product to do: the k-step split variant
    if (DO_P && a.plan.channels_per_scale <= 64) return launch_bank_k<NPB, NCB, DO_P, DO_W, true>(a, x_dtype, grid, s);
    return launch_bank_k<NPB, NCB, DO_P, DO_W, false>(a, x_dtype, grid, s);
}
template <int NPB, int NCB>
static hipError_t launch_bank_x(const SpxBankBwdArgs& a, int x_dtype, dim3 grid, hipStream_t s) {
    if constexpr (NCB == 1) {
        // one class block: d_W comes from kernel 1's tile partials (spx_dw_reduce_kernel), this kernel carries d_bank only
        if (!a.d_bank) return hipSuccess;
        if (x_dtype == 0 && a.vec_ok == 1 && a.HW % 64 == 0 && a.plan.channels_per_scale > 64) {
            constexpr size_t lds = (size_t)spx_bd_lds_bytes<NPB>();
            static_assert(lds <= SPX_LDS_LIMIT, "bank kernel LDS");
            hipLaunchKernelGGL((spx_bank_dma_kernel<NPB>), grid, dim3(SPX_BK_THREADS), lds, s, a);
            return hipGetLastError();
        }
        return launch_bank_pw<NPB, NCB, true, false>(a, x_dtype, grid, s);
    } else if constexpr (NCB < 5) {
        if (!a.d_bank) return launch_bank_pw<NPB, NCB, false, true>(a, x_dtype, grid, s);
        return launch_bank_pw<NPB, NCB, true, true>(a, x_dtype, grid, s);
    } else {
        hipError_t e = hipSuccess;
        if (a.d_bank) e = launch_bank_pw<NPB, NCB, true, false>(a, x_dtype, grid, s);
        if (e == hipSuccess && a.d_W) e = launch_bank_pw<NPB, NCB, false, true>(a, x_dtype, grid, s);
        return e;
    }
}

static hipError_t spx_launch_bank_part(const SpxBankBwdArgs& a, int x_dtype, hipStream_t s) {
    const spx_plan& pl = a.plan;
    dim3 grid((unsigned)a.nslabs, (unsigned)pl.npanels);
    if (pl.ncb == 1)
        return pl.npb == 2 ? launch_bank_x<2, 1>(a, x_dtype, grid, s) : pl.npb == 4 ? launch_bank_x<4, 1>(a, x_dtype, grid, s) : launch_bank_x<6, 1>(a, x_dtype, grid, s);
    if (pl.ncb == 2)
        return pl.npb == 2 ? launch_bank_x<2, 2>(a, x_dtype, grid, s) : pl.npb == 4 ? launch_bank_x<4, 2>(a, x_dtype, grid, s) : launch_bank_x<6, 2>(a, x_dtype, grid, s);
    return pl.npb == 2 ? launch_bank_x<2, 5>(a, x_dtype, grid, s) : pl.npb == 4 ? launch_bank_x<4, 5>(a, x_dtype, grid, s) : launch_bank_x<6, 5>(a, x_dtype, grid, s);
}

hipError_t spx_launch_bank_bwd(const SpxBankBwdArgs& a0, int x_dtype, hipStream_t s) {
    SpxBankBwdArgs a = a0;
    const spx_plan& pl = a.plan;
    const int rows = pl.npb * 32;
    int nci, nci_vec, sv, st;
    spx_bank_parts(pl, a.B, a.HW, nci, nci_vec, sv, st);
    a.nsplit = sv + st;
    hipError_t e = hipSuccess;
    if (sv > 0) {
        a.vec_ok = a.HW % 8 == 0 ? 1 : 2; a.ci_first = 0; a.nci_launch = nci_vec; a.slab_first = 0; a.nslabs = sv;
        e = spx_launch_bank_part(a, x_dtype, s);
    }
    if (e == hipSuccess && st > 0) {
        a.vec_ok = 0; a.ci_first = nci_vec; a.nci_launch = nci - nci_vec; a.slab_first = sv; a.nslabs = st;
        e = spx_launch_bank_part(a, x_dtype, s);
    }
    if (e != hipSuccess) return e;
    if (pl.ncb == 1 && a.d_W) {
        hipLaunchKernelGGL(spx_dw_reduce_kernel, dim3((unsigned)a.nsplit, (unsigned)pl.npanels), dim3(256), 0, s, a);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    const long long n = (long long)pl.npanels * rows * (pl.channels_per_scale + pl.num_classes);
    hipLaunchKernelGGL(spx_bank_reduce_kernel, dim3((unsigned)((n + SPX_RED_ELEMS - 1) / SPX_RED_ELEMS)),
                       dim3(SPX_RED_ELEMS * SPX_RED_PARTS), 0, s, a);
    return hipGetLastError();
}
