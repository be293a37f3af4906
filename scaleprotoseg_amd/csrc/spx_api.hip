// C ABI of libspx_hip.so (see include/spx_hip.h).  Host-side validation + kernel launches; no torch types,
// no allocation, no synchronisation (graph-capturable).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "spx_args.h"

static thread_local char g_err[512] = "";
#ifdef SPX_DIAG
// diagnostic builds only (-DSPX_DIAG, never the product library: include/spx_hip.h promises no process-global state): the
// buffer of the in-kernel phase clocks (SPX_DIAG_STAMPS) and overrides of the tile permutation / product-kernel tiling
static unsigned long long* g_dbg = nullptr;
static int g_tile_mul_req = 0;                 // 0 = automatic (below), 1 = identity, > 1 = that multiplier
#else
static constexpr unsigned long long* g_dbg = nullptr;
static constexpr int g_tile_mul_req = 0;
#endif

// Block -> tile permutation of the pixel kernels: tile = (block * mul) mod tiles with mul coprime to the tile count, so that
// the workgroups running at one time are spread over the whole pixel range instead of covering one contiguous stretch of every
// feature plane (see profiles/EXPERIMENTS.md, plane strides).
static int gcd_i(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }
int spx_tile_mul(int tiles_launch, long long plane_bytes) {
    // Automatic: only where the feature planes are a multiple of 1 MiB apart (power-of-two grids such as 1024 x 2048: plane
    // stride 4 MiB).  There neighbouring tiles of all channels fall on the same few DRAM banks at the same time and spreading
    // the co-running tiles 64 KiB apart measured -3.5 % (forward) / -1.6 % (pixel backward); on every other stride tried it
    // changed nothing or cost up to 15 % (1016 x 2048), so it stays off there (tools/probes/stride_sweep.py).
    int m = g_tile_mul_req;
    if (m == 0) m = (plane_bytes > 0 && plane_bytes % (1ll << 20) == 0) ? 257 : 1;
    if (m <= 1 || tiles_launch < 4 * m) return 1;
    while (gcd_i(m, tiles_launch) != 1) ++m;
    return m;
}

static int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}
static int hip_status(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    return fail("%s: %s", what, hipGetErrorString(e));
}
static int check_plan(const spx_plan* pl) {
    if (!pl) return fail("plan is NULL");
    if (pl->npanels < 1 || pl->npanels > SPX_MAX_PANELS) return fail("plan: npanels %d out of range", pl->npanels);
    if (pl->npb < 1 || pl->npb > 6) return fail("plan: npb %d out of range", pl->npb);
    if (pl->ncb < 1 || pl->ncb > 5) return fail("plan: ncb %d out of range", pl->ncb);
    if (pl->kc != 32) return fail("plan: kc %d", pl->kc);
    if (pl->npb != 2 && pl->npb != 4 && pl->npb != 6) return fail("plan: npb %d (must be 2, 4 or 6)", pl->npb);
    if (pl->ncb != 1 && pl->ncb != 2 && pl->ncb != 5) return fail("plan: ncb %d (must be 1, 2 or 5)", pl->ncb);
    return 0;
}
static int x_vec_ok(const void* x, int x_dtype, int HW) {
    (void)x; (void)x_dtype; (void)HW;
    return 0;   /* decided per launch by the launchers (full tiles: vector staging at any alignment; ragged tail: element-wise) */
}

// Scale-parallel launch (grid.y = scale): for pixel grids that do not fill the chip (the reference's training crops:
// 10 x 65 x 65 px = 331 tiles on 512-768 workgroup slots, each walking every panel of every scale serially) the
// panels of different scales are independent work - different channels of X and dX, different prototype rows - and run
// as separate workgroups.  Only the logits (a sum over all prototypes) need a second step, a fixed-order sum of the
// per-scale partials.  Returns the number of groups (1 = keep the single walk) and their first panels.
#ifndef SPX_SPLIT_MAX_TILES
#define SPX_SPLIT_MAX_TILES 1024
#endif
int spx_split_groups(const spx_plan& pl, int B, int HW, int32_t* group_first) {
    const long long tiles = (long long)B * ((HW + SPX_TILE_PX - 1) / SPX_TILE_PX);
    int g = 0;
    for (int q = 0; q < pl.npanels; ++q)
        if (q == 0 || pl.panel_ch0[q] != pl.panel_ch0[q - 1]) group_first[g++] = q;
    group_first[g] = pl.npanels;
    if (g < 2 || tiles > SPX_SPLIT_MAX_TILES) {
        group_first[0] = 0;
        group_first[1] = pl.npanels;
        return 1;
    }
    return g;
}

hipError_t spx_launch_fwd_npb2(const SpxFwdArgs& a, int x_dtype, hipStream_t s);
hipError_t spx_launch_fwd_npb4(const SpxFwdArgs& a, int x_dtype, hipStream_t s);
hipError_t spx_launch_fwd_npb6(const SpxFwdArgs& a, int x_dtype, hipStream_t s);
hipError_t spx_launch_bwd_npb2(const SpxBwdArgs& a, int x_dtype, hipStream_t s);
hipError_t spx_launch_bwd_npb4(const SpxBwdArgs& a, int x_dtype, hipStream_t s);
hipError_t spx_launch_bwd_npb6(const SpxBwdArgs& a, int x_dtype, hipStream_t s);
hipError_t spx_launch_fwd(const SpxFwdArgs& a, int x_dtype, hipStream_t s) {
    return a.plan.npb == 2 ? spx_launch_fwd_npb2(a, x_dtype, s) : a.plan.npb == 4 ? spx_launch_fwd_npb4(a, x_dtype, s) : spx_launch_fwd_npb6(a, x_dtype, s);
}
hipError_t spx_launch_bwd(const SpxBwdArgs& a, int x_dtype, hipStream_t s) {
    return a.plan.npb == 2 ? spx_launch_bwd_npb2(a, x_dtype, s) : a.plan.npb == 4 ? spx_launch_bwd_npb4(a, x_dtype, s) : spx_launch_bwd_npb6(a, x_dtype, s);
}

extern "C" {

int spx_version(void) { return SPX_ABI_VERSION; }
#ifdef SPX_DIAG
void spx_diag_set_debug_buffer(void* p) { g_dbg = (unsigned long long*)p; }
void spx_diag_set_tile_mul(int m) { g_tile_mul_req = m; }       /* 1 = identity */
void spx_diag_set_gemm(int wm, int splits) { spx_gemm_force(wm, splits); }     /* 0 = the shape-derived default */
#endif
const char* spx_last_error(void) { return g_err; }

int spx_make_plan(int32_t P, int32_t K, int32_t S, int32_t Cs, const int32_t* lo, const int32_t* hi, spx_plan* out) {
    if (!out || !lo || !hi) return fail("spx_make_plan: NULL argument");
    if (P < 1 || K < 1 || S < 1 || Cs < 1) return fail("spx_make_plan: non-positive size (P=%d K=%d S=%d Cs=%d)", P, K, S, Cs);
    if (Cs % 16) return fail("spx_make_plan: channels per scale (%d) must be a multiple of 16 (MFMA k-step)", Cs);
    if (K > 160) return fail("spx_make_plan: num_classes %d > 160 not supported by the fused head", K);
    memset(out, 0, sizeof(*out));
    out->num_prototypes = P;
    out->num_classes = K;
    out->num_scales = S;
    out->channels_per_scale = Cs;
    out->kc = 32;                              /* chunks of 32 channels; a 16-channel tail is zero-filled */
    out->ncb = (K <= 32) ? 1 : (K <= 64) ? 2 : 5;   /* the kernels are specialised for 1, 2 or 5 class blocks */
    int per_max = 1, covered = 0;
    for (int s = 0; s < S; ++s) {
        const int n = hi[s] - lo[s];
        if (lo[s] < 0 || hi[s] > P || n < 0) return fail("spx_make_plan: scale %d range (%d,%d) outside [0,%d]", s, lo[s], hi[s], P);
        if (s && lo[s] != hi[s - 1]) return fail("spx_make_plan: scale ranges must be contiguous");
        covered += n;
        if (n == 0) continue;
        const int np_s = (n + 191) / 192;
        const int per = (n + np_s - 1) / np_s;
        if (per > per_max) per_max = per;
    }
    // the reference's F.linear rejects a distance map narrower than the head (P % S != 0): same here
    if (covered != P) return fail("spx_make_plan: scale table covers %d prototypes but the bank has %d (reference needs P %% S == 0)", covered, P);
    out->npb = ((per_max + 63) / 64) * 2;      /* panel height in 32-prototype blocks: 2, 4 or 6 */
    const int cap = out->npb * 32;
    int q = 0;
    for (int s = 0; s < S; ++s) {
        for (int p = lo[s]; p < hi[s]; p += cap) {
            if (q >= SPX_MAX_PANELS) return fail("spx_make_plan: more than %d panels", SPX_MAX_PANELS);
            out->panel_ch0[q] = s * Cs;
            out->panel_p0[q] = p;
            out->panel_np[q] = (hi[s] - p < cap) ? hi[s] - p : cap;
            ++q;
        }
    }
    if (q == 0) return fail("spx_make_plan: empty bank");
    out->npanels = q;
    return 0;
}

size_t spx_bwd_scratch_bytes(const spx_plan* pl, int32_t B, int32_t HW) { return spx_bwd_scratch_elems(*pl, B, HW) * 2; }
size_t spx_bwd_dx_scratch_bytes(const spx_plan* pl, int32_t x_dtype, int32_t B, int32_t HW) {
    if (x_dtype != 0) return 0;                                     /* fp32 features accumulate in dX itself */
    for (int q = 1; q < pl->npanels; ++q)
        if (pl->panel_ch0[q] == pl->panel_ch0[q - 1])               /* some scale spans several panels */
            return (size_t)B * pl->num_scales * pl->channels_per_scale * (((size_t)HW + 3) / 4 * 4) * sizeof(float);
    return 0;
}
size_t spx_bwd_head_scratch_bytes(const spx_plan* pl, int32_t B, int32_t HW) {
    if (pl->ncb != 1) return spx_bwd_scratch_elems(*pl, B, HW) * 2;              /* the activation blob */
    const size_t tiles = (size_t)B * ((HW + SPX_TILE_PX - 1) / SPX_TILE_PX);
    return spx_dw_partial_bytes(pl->npanels, tiles, pl->npb, pl->num_classes) + 16;   /* d_W tile partials + the head scale */
}

size_t spx_packed_bank_bytes(const spx_plan* pl) {
    return (size_t)pl->npanels * pl->npb * 32 * (((pl->channels_per_scale + 31) / 32) * 32) * 2;
}
size_t spx_packed_bankT_bytes(const spx_plan* pl) {
    return (size_t)pl->npanels * pl->npb * 2 * ((pl->channels_per_scale + 31) / 32) * 1024;
}
size_t spx_packed_p2_bytes(const spx_plan* pl) { return (size_t)pl->npanels * pl->npb * 32 * 4; }
size_t spx_packed_head_bytes(const spx_plan* pl) { return (size_t)pl->ncb * pl->npanels * pl->npb * 2 * 2048; }
size_t spx_packed_headT_bytes(const spx_plan* pl) { return (size_t)pl->npanels * pl->npb * (pl->ncb * 2) * 2048; }

int spx_pack_bank(const spx_plan* pl, const float* bank, void* pb, void* pbT, float* p2, void* stream) {
    if (check_plan(pl)) return 1;
    if (!bank || !pb || !p2) return fail("spx_pack_bank: NULL buffer");
    return hip_status(spx_launch_pack_bank(*pl, bank, pb, pbT, p2, (hipStream_t)stream), "spx_pack_bank");
}

int spx_pack_head(const spx_plan* pl, const float* W, void* ph, void* phT, void* stream) {
    if (check_plan(pl)) return 1;
    if (!W || !ph) return fail("spx_pack_head: NULL buffer");
    return hip_status(spx_launch_pack_head(*pl, W, ph, phT, (hipStream_t)stream), "spx_pack_head");
}

size_t spx_packed_tail_bytes(const spx_plan* pl) { return (size_t)pl->ncb * 2 * 2048; }

int spx_pack_all(const spx_plan* pl, const float* bank, const float* W, const float* Wg, int32_t K2, void* packed_bank,
                 void* packed_bankT, float* p2, void* packed_head, void* packed_headT, void* packed_tail, void* packed_tailT,
                 void* stream) {
    if (check_plan(pl)) return 1;
    if (!bank || !packed_bank || !p2) return fail("spx_pack_all: NULL bank buffer");
    if ((packed_head || packed_headT) && !W) return fail("spx_pack_all: head outputs without a head");
    if (packed_headT && !packed_head) return fail("spx_pack_all: packed_headT without packed_head");
    if ((packed_tail || packed_tailT) && (!Wg || !W || !packed_tail)) return fail("spx_pack_all: tail outputs need W, Wg and packed_tail");
    if (packed_tail && (K2 < 1 || K2 > 32)) return fail("spx_pack_all: %d tail classes (the fused tail carries at most 32)", K2);
    SpxPackAllArgs a{};
    a.plan = *pl;
    a.bank = bank; a.W = W; a.Wg = Wg; a.K2 = K2;
    a.headT_units = packed_tail != nullptr;      // the grouping backward reads head^T in accumulator (unit) order
    a.packed_bank = packed_bank; a.packed_bankT = packed_bankT; a.p2 = p2;
    a.packed_head = packed_head; a.packed_headT = packed_headT; a.packed_tail = packed_tail; a.packed_tailT = packed_tailT;
    return hip_status(spx_launch_pack_all(a, (hipStream_t)stream), "spx_pack_all");
}

int spx_pack_group_tail(const spx_plan* pl, const float* Wg, int32_t K2, void* packed_tail, void* packed_tailT, void* stream) {
    if (check_plan(pl)) return 1;
    if (!Wg || !packed_tail) return fail("spx_pack_group_tail: NULL buffer");
    if (K2 < 1 || K2 > 32) return fail("spx_pack_group_tail: %d classes (the fused tail carries at most 32)", K2);
    return hip_status(spx_launch_pack_tail(*pl, Wg, K2, packed_tail, packed_tailT, (hipStream_t)stream), "spx_pack_group_tail");
}

int spx_pack_headT_units(const spx_plan* pl, const float* W, void* packed_headT_units, void* stream) {
    if (check_plan(pl)) return 1;
    if (!W || !packed_headT_units) return fail("spx_pack_headT_units: NULL buffer");
    return hip_status(spx_launch_pack_headT_units(*pl, W, packed_headT_units, (hipStream_t)stream), "spx_pack_headT_units");
}

struct SpxTailFwd { const void* packed_tail; int32_t K2; float* gact; };
struct SpxTailBwd { const void* packed_tailT; int32_t K2; const float* gact; float* d_units; const float* d_gact; };

static int dist_fwd_impl(const spx_plan* pl, const void* x, int32_t x_dtype, int32_t B, int32_t HW,
                         const void* packed_bank, const float* packed_p2, const void* packed_head, float* distances,
                         const int32_t* labels, const uint32_t* proto_key, int32_t J, float* cls_dist,
                         float* activations, float* logits, float epsilon, int32_t act_fn, void* stream,
                         SpxTailFwd tail = SpxTailFwd{nullptr, 0, nullptr}, const spx_ce* ce = nullptr, void* split_ws = nullptr,
                         bool keep_partials = false, unsigned long long* push_keys = nullptr, float push_max = 0.0f, int push_void = -1, int push_K = 0) {
    if (check_plan(pl)) return 1;
    if (!x || !packed_bank || !packed_p2) return fail("spx_dist_fwd: NULL operand");
    if (x_dtype != 0 && x_dtype != 1) return fail("spx_dist_fwd: x_dtype %d (0 = bf16, 1 = fp32)", x_dtype);
    if (B < 1 || HW < 1) return fail("spx_dist_fwd: empty input (B=%d HW=%d)", B, HW);
    if (act_fn != 0 && act_fn != 1) return fail("spx_dist_fwd: act_fn %d", act_fn);
    if (logits && !packed_head) return fail("spx_dist_fwd: logits requested without a packed head");
    const long long tiles = (long long)B * ((HW + SPX_TILE_PX - 1) / SPX_TILE_PX);
    if (tiles > 0x7fffffffLL) return fail("spx_dist_fwd: too many pixel tiles");
    if ((long long)pl->num_prototypes * HW >= (1LL << 29)) return fail("spx_dist_fwd: P*HW too large for 32-bit offsets");
    if ((long long)pl->num_scales * pl->channels_per_scale * HW * (x_dtype ? 4 : 2) >= (1LL << 32)) return fail("spx_dist_fwd: one image of features exceeds 4 GiB");
    SpxFwdArgs a;
    a.plan = *pl;
    a.x = x;
    a.packed_bank = (const char*)packed_bank;
    a.p2 = packed_p2;
    a.packed_head = (const char*)packed_head;
    a.dist = distances;
    a.act = activations;
    a.logits = logits;
    a.B = B;
    a.HW = HW;
    a.vec_ok = x_vec_ok(x, x_dtype, HW);
    a.labels = labels;
    a.proto_key = proto_key;
    a.cls_dist = cls_dist;
    a.J = J;
    a.push_keys = push_keys;
    a.push_max = push_max;
    a.push_void = push_void;
    a.push_K = push_K;
    a.packed_tail = (const char*)tail.packed_tail;
    a.gact = tail.gact;
    a.K2 = tail.K2;
    a.dist_vec = distances && ((uintptr_t)distances & 15) == 0 && HW % 4 == 0;
    a.ce_labels = nullptr;
    a.ce_lse = nullptr;
    a.ce_pred = nullptr;
    a.ce_partials = nullptr;
    if (ce) {
        if (!logits) return fail("spx_dist_fwd_ce: the cross entropy needs the logits output");
        if (!ce->labels || !ce->lse || !ce->partials) return fail("spx_dist_fwd_ce: NULL labels / lse / partials");
        a.ce_labels = ce->labels;
        a.ce_lse = ce->lse;
        a.ce_pred = ce->pred;
        a.ce_partials = ce->partials;
    }
    a.eps = epsilon;
    a.act_fn = act_fn;
    a.dbg = g_dbg;
    // scale-parallel launch: always when no logits are asked for; with logits when the caller brought the workspace for
    // the per-scale partials (and neither a grouping tail nor the fused cross entropy rides on the logits tile)
    a.ngroups = 1;
    a.logits_group_stride = 0;
    const int groups = spx_split_groups(*pl, B, HW, a.group_first);
    if (groups > 1 && !tail.packed_tail && !ce && (!logits || split_ws)) {
        a.ngroups = groups;
        if (logits) {
            a.logits = (float*)split_ws;
            a.logits_group_stride = (size_t)B * HW * pl->num_classes;
        }
    }
    if (hip_status(spx_launch_fwd(a, x_dtype, (hipStream_t)stream), "spx_dist_fwd")) return 1;
    if (a.ngroups > 1 && logits && !keep_partials)
        return hip_status(spx_launch_sum_groups((const float*)split_ws, a.logits_group_stride, groups, logits, (hipStream_t)stream), "spx_dist_fwd (sum of the scale partials)");
    return 0;
}

int spx_dist_fwd(const spx_plan* pl, const void* x, int32_t x_dtype, int32_t B, int32_t HW, const void* packed_bank,
                 const float* packed_p2, const void* packed_head, float* distances, float* activations,
                 float* logits, float epsilon, int32_t act_fn, void* stream) {
    return dist_fwd_impl(pl, x, x_dtype, B, HW, packed_bank, packed_p2, packed_head, distances, nullptr, nullptr, 0,
                         nullptr, activations, logits, epsilon, act_fn, stream);
}

int spx_dist_fwd_group(const spx_plan* pl, const void* x, int32_t x_dtype, int32_t B, int32_t HW,
                       const void* packed_bank, const float* packed_p2, const void* packed_head,
                       const void* packed_tail, int32_t K2, float* distances, float* activations,
                       float* group_activations, float* logits, float epsilon, int32_t act_fn, void* stream) {
    if (!packed_head || !packed_tail || !logits) return fail("spx_dist_fwd_group: NULL head / tail / logits");
    if (K2 < 1 || K2 > 32) return fail("spx_dist_fwd_group: %d classes (at most 32)", K2);
    return dist_fwd_impl(pl, x, x_dtype, B, HW, packed_bank, packed_p2, packed_head, distances, nullptr, nullptr, 0,
                         nullptr, activations, logits, epsilon, act_fn, stream, SpxTailFwd{packed_tail, K2, group_activations});
}

static int check_cls(const char* who, const int32_t* labels, const uint32_t* proto_key, int32_t J, int32_t HW) {
    if (!labels || !proto_key) return fail("%s: NULL labels / proto_key", who);
    if (J < 1 || J > 0xFFFF) return fail("%s: J %d out of range", who, J);
    if ((long long)HW * J >= (1LL << 29)) return fail("%s: HW*J too large for 32-bit offsets", who);
    return 0;
}

size_t spx_group_tail_workspace_bytes(const spx_plan* pl, int32_t B, int32_t HW) {
    return pl ? (size_t)B * HW * pl->num_classes * sizeof(float) : 0;
}

int spx_dist_fwd_group_ws(const spx_plan* pl, const void* x, int32_t x_dtype, int32_t B, int32_t HW,
                          const void* packed_bank, const float* packed_p2, const void* packed_head,
                          const float* Wg, int32_t K2, float* distances, float* activations,
                          float* group_activations, float* logits, const spx_ce* ce, void* workspace,
                          float epsilon, int32_t act_fn, void* stream) {
    if (!packed_head || !Wg || !logits || !workspace) return fail("spx_dist_fwd_group_ws: NULL head / W_g / logits / workspace");
    if (K2 < 1 || K2 > 32) return fail("spx_dist_fwd_group_ws: %d classes (at most 32)", K2);
    if (check_plan(pl)) return 1;
    if (ce && (!ce->labels || !ce->lse || !ce->partials)) return fail("spx_dist_fwd_group_ws: NULL labels / lse / partials");
    // the unit product as the kernel's \"logits\" into the workspace, then the tail kernel.  NOT scale-parallel: measured on
    // the Cityscapes crops (10 x 65 x 65, 57 units) the per-scale partial units cost more traffic (4 x [M][57] written and
    // re-read: +30 us) than the shorter panel walk saves
    const int groups = 1;
    float* const parts = (float*)workspace;
    if (dist_fwd_impl(pl, x, x_dtype, B, HW, packed_bank, packed_p2, packed_head, distances, nullptr, nullptr, 0, nullptr,
                      activations, parts, epsilon, act_fn, stream, SpxTailFwd{nullptr, 0, nullptr}, nullptr, nullptr, true))
        return 1;
    const long long M = (long long)B * HW;
    // the tail writes one (sum, count) pair per 64 pixels and clears the rest of the spx_ce_partials_flat(M) pairs itself
    return hip_status(spx_launch_group_tail(parts, groups, M, pl->num_classes, Wg, K2, group_activations, logits,
                                            ce ? ce->labels : nullptr, ce ? ce->lse : nullptr, ce ? ce->pred : nullptr,
                                            ce ? ce->partials : nullptr, (hipStream_t)stream), "spx_dist_fwd_group_ws (tail)");
}

int spx_dist_fwd_cls(const spx_plan* pl, const void* x, int32_t x_dtype, int32_t B, int32_t HW,
                     const void* packed_bank, const float* packed_p2, const void* packed_head,
                     const int32_t* labels, const uint32_t* proto_key, int32_t J, float* class_distances,
                     float* activations, float* logits, float epsilon, int32_t act_fn, void* stream) {
    if (check_cls("spx_dist_fwd_cls", labels, proto_key, J, HW)) return 1;
    if (!class_distances) return fail("spx_dist_fwd_cls: NULL class_distances");
    return dist_fwd_impl(pl, x, x_dtype, B, HW, packed_bank, packed_p2, packed_head, nullptr, labels, proto_key, J,
                         class_distances, activations, logits, epsilon, act_fn, stream);
}

int spx_dist_push_min(const spx_plan* pl, const void* x, int32_t x_dtype, int32_t B, int32_t HW, const void* packed_bank,
                      const float* packed_p2, const int32_t* labels, int32_t void_class, int32_t K, const uint32_t* proto_key,
                      float max_dist, int64_t* indices, float* values, uint64_t* scratch, void* stream) {
    if (check_cls("spx_dist_push_min", labels, proto_key, 1, HW)) return 1;
    if (K < 1 || K > 0xFFFD) return fail("spx_dist_push_min: K %d out of range", K);
    if (!indices || !values || !scratch) return fail("spx_dist_push_min: NULL output / scratch");
    if (check_plan(pl)) return 1;
    const size_t n = (size_t)B * pl->num_prototypes;
    if (hip_status(hipMemsetAsync(scratch, 0xFF, n * sizeof(uint64_t), (hipStream_t)stream), "spx_dist_push_min (scratch fill)")) return 1;
    if (dist_fwd_impl(pl, x, x_dtype, B, HW, packed_bank, packed_p2, nullptr, nullptr, labels, proto_key, 1, nullptr, nullptr, nullptr,
                      1e-4f, 0, stream, SpxTailFwd{nullptr, 0, nullptr}, nullptr, nullptr, false, (unsigned long long*)scratch, max_dist, void_class, K))
        return 1;
    return hip_status(spx_launch_push_finalize(scratch, (int)n, indices, values, (hipStream_t)stream), "spx_dist_push_min (decode)");
}

static int dist_bwd_impl(const spx_plan* pl, const void* x, int32_t x_dtype, int32_t B, int32_t HW,
                         const void* packed_bank, const void* packed_bankT, const float* packed_p2,
                         const void* packed_headT, const float* d_dist, const int32_t* labels,
                         const uint32_t* proto_key, int32_t J, const float* d_cls_dist, const float* d_act,
                         const float* d_logits, void* dx, void* dx_acc, void* g_out, void* a_out, float epsilon, int32_t act_fn,
                         void* stream, SpxTailBwd tail = SpxTailBwd{nullptr, 0, nullptr, nullptr, nullptr}, const spx_ce* ce = nullptr) {
    if (check_plan(pl)) return 1;
    if (!x || !packed_bank || !packed_p2) return fail("spx_dist_bwd: NULL operand");
    if (x_dtype != 0 && x_dtype != 1) return fail("spx_dist_bwd: x_dtype %d", x_dtype);
    if (B < 1 || HW < 1) return fail("spx_dist_bwd: empty input");
    if (dx && !packed_bankT) return fail("spx_dist_bwd: dx requested without packed bank^T");
    if ((d_logits || ce) && !packed_headT) return fail("spx_dist_bwd: d_logits given without packed head^T");
    if (ce && d_logits) return fail("spx_dist_bwd_ce: the fused cross entropy replaces d_logits");
    if (ce && (!ce->labels || !ce->lse || !ce->logits || !ce->coef)) return fail("spx_dist_bwd_ce: NULL labels / lse / logits / coef");
    if ((long long)pl->num_prototypes * HW >= (1LL << 29)) return fail("spx_dist_bwd: P*HW too large for 32-bit offsets");
    const long long tiles = (long long)B * ((HW + SPX_TILE_PX - 1) / SPX_TILE_PX);
    if (tiles > 0x7fffffffLL) return fail("spx_dist_bwd: too many pixel tiles");
    if ((long long)pl->num_scales * pl->channels_per_scale * HW * (x_dtype ? 4 : 2) >= (1LL << 32)) return fail("spx_dist_bwd: one image of features exceeds 4 GiB");
    if (g_out && pl->channels_per_scale > 256) return fail("spx_dist_bwd: G scratch requested but spx_bank_bwd supports channels_per_scale <= 256 (got %d)", pl->channels_per_scale);
    SpxBwdArgs a;
    a.plan = *pl;
    a.x = x;
    a.packed_bank = (const char*)packed_bank;
    a.packed_bankT = (const char*)packed_bankT;
    a.p2 = packed_p2;
    a.packed_headT = (const char*)packed_headT;
    a.d_dist = d_dist;
    a.d_act = d_act;
    a.d_logits = d_logits;
    a.labels = labels;
    a.proto_key = proto_key;
    a.d_cls_dist = d_cls_dist;
    a.J = J;
    a.packed_tailT = (const char*)tail.packed_tailT;
    a.gact = tail.gact;
    a.d_units = tail.d_units;
    a.d_gact = tail.d_gact;
    a.K2 = tail.K2;
    a.ce_labels = ce ? ce->labels : nullptr;
    a.ce_logits = ce ? ce->logits : nullptr;
    a.ce_lse = ce ? ce->lse : nullptr;
    a.ce_coef = ce ? ce->coef : nullptr;
    a.ce_dlogits_out = ce ? ce->d_logits_out : nullptr;
    a.dx = dx;
    a.dx_acc = (float*)dx_acc;
    a.g_out = (uint16_t*)g_out;
    a.a_out = (uint16_t*)a_out;
    a.B = B;
    a.HW = HW;
    a.vec_ok = x_vec_ok(x, x_dtype, HW);
    a.eps = epsilon;
    a.act_fn = act_fn;
    a.dbg = g_dbg;
    a.ngroups = spx_split_groups(*pl, B, HW, a.group_first);      // the backward needs no cross-scale step at all
    return hip_status(spx_launch_bwd(a, x_dtype, (hipStream_t)stream), "spx_dist_bwd");
}

int spx_dist_bwd(const spx_plan* pl, const void* x, int32_t x_dtype, int32_t B, int32_t HW, const void* packed_bank,
                 const void* packed_bankT, const float* packed_p2, const void* packed_headT, const float* d_dist,
                 const float* d_act, const float* d_logits, void* dx, void* dx_acc, void* g_out, void* a_out, float epsilon,
                 int32_t act_fn, void* stream) {
    return dist_bwd_impl(pl, x, x_dtype, B, HW, packed_bank, packed_bankT, packed_p2, packed_headT, d_dist, nullptr,
                         nullptr, 0, nullptr, d_act, d_logits, dx, dx_acc, g_out, a_out, epsilon, act_fn, stream);
}

int spx_dist_bwd_group(const spx_plan* pl, const void* x, int32_t x_dtype, int32_t B, int32_t HW,
                       const void* packed_bank, const void* packed_bankT, const float* packed_p2,
                       const void* packed_headT_units, const void* packed_tailT, int32_t K2,
                       const float* group_activations, const float* d_dist, const float* d_act,
                       const float* d_logits, const float* d_group_activations, float* d_units, void* dx, void* dx_acc, void* g_out,
                       void* a_out, float epsilon, int32_t act_fn, void* stream) {
    if (!packed_headT_units || !packed_tailT || !group_activations || !d_logits || !d_units)
        return fail("spx_dist_bwd_group: NULL tail operand");
    if (K2 < 1 || K2 > 32) return fail("spx_dist_bwd_group: %d classes (at most 32)", K2);
    return dist_bwd_impl(pl, x, x_dtype, B, HW, packed_bank, packed_bankT, packed_p2, packed_headT_units, d_dist,
                         nullptr, nullptr, 0, nullptr, d_act, d_logits, dx, dx_acc, g_out, a_out, epsilon, act_fn, stream,
                         SpxTailBwd{packed_tailT, K2, group_activations, d_units, d_group_activations});
}

int spx_dist_bwd_group_ce(const spx_plan* pl, const void* x, int32_t x_dtype, int32_t B, int32_t HW,
                          const void* packed_bank, const void* packed_bankT, const float* packed_p2,
                          const void* packed_headT_units, const void* packed_tailT, int32_t K2,
                          const float* group_activations, const float* d_dist, const float* d_act, const spx_ce* ce,
                          const float* d_group_activations, float* d_units, void* dx, void* dx_acc, void* g_out, void* a_out, float epsilon,
                          int32_t act_fn, void* stream) {
    if (!ce) return fail("spx_dist_bwd_group_ce: NULL ce");
    if (!packed_headT_units || !packed_tailT || !group_activations || !d_units)
        return fail("spx_dist_bwd_group_ce: NULL tail operand");
    if (K2 < 1 || K2 > 32) return fail("spx_dist_bwd_group_ce: %d classes (at most 32)", K2);
    return dist_bwd_impl(pl, x, x_dtype, B, HW, packed_bank, packed_bankT, packed_p2, packed_headT_units, d_dist,
                         nullptr, nullptr, 0, nullptr, d_act, nullptr, dx, dx_acc, g_out, a_out, epsilon, act_fn, stream,
                         SpxTailBwd{packed_tailT, K2, group_activations, d_units, d_group_activations}, ce);
}

int spx_dist_bwd_cls(const spx_plan* pl, const void* x, int32_t x_dtype, int32_t B, int32_t HW,
                     const void* packed_bank, const void* packed_bankT, const float* packed_p2,
                     const void* packed_headT, const int32_t* labels, const uint32_t* proto_key, int32_t J,
                     const float* d_class_distances, const float* d_act, const float* d_logits, void* dx,
                     void* dx_acc, void* g_out, void* a_out, float epsilon, int32_t act_fn, void* stream) {
    if (check_cls("spx_dist_bwd_cls", labels, proto_key, J, HW)) return 1;
    // d_class_distances may be NULL (no gradient reaches the gathered distances)
    return dist_bwd_impl(pl, x, x_dtype, B, HW, packed_bank, packed_bankT, packed_p2, packed_headT, nullptr,
                         d_class_distances ? labels : nullptr, proto_key, J, d_class_distances, d_act, d_logits, dx, dx_acc,
                         g_out, a_out, epsilon, act_fn, stream);
}

int32_t spx_fwd_split_groups(const spx_plan* pl, int32_t B, int32_t HW) {
    int32_t gf[SPX_MAX_PANELS + 1];
    return pl ? spx_split_groups(*pl, B, HW, gf) : 1;
}
size_t spx_fwd_split_workspace_bytes(const spx_plan* pl, int32_t B, int32_t HW) {
    const int g = spx_fwd_split_groups(pl, B, HW);
    return g > 1 ? (size_t)g * B * HW * pl->num_classes * sizeof(float) : 0;
}
int spx_dist_fwd_ws(const spx_plan* pl, const void* x, int32_t x_dtype, int32_t B, int32_t HW, const void* packed_bank,
                    const float* packed_p2, const void* packed_head, const int32_t* labels_cls, const uint32_t* proto_key,
                    int32_t J, float* class_distances, float* distances, float* activations, float* logits,
                    void* split_workspace, float epsilon, int32_t act_fn, void* stream) {
    if (labels_cls) {
        if (check_cls("spx_dist_fwd_ws", labels_cls, proto_key, J, HW)) return 1;
        if (!class_distances) return fail("spx_dist_fwd_ws: NULL class_distances");
        if (distances) return fail("spx_dist_fwd_ws: class-gathered and P-wide distances are exclusive");
    }
    return dist_fwd_impl(pl, x, x_dtype, B, HW, packed_bank, packed_p2, packed_head, distances, labels_cls, proto_key,
                         labels_cls ? J : 0, labels_cls ? class_distances : nullptr, activations, logits, epsilon, act_fn,
                         stream, SpxTailFwd{nullptr, 0, nullptr}, nullptr, split_workspace);
}

size_t spx_ce_partials(int32_t B, int32_t HW) { return (size_t)B * ((HW + SPX_TILE_PX - 1) / SPX_TILE_PX) * 4; }
size_t spx_ce_partials_flat(int64_t M) { return (size_t)((M + 255) / 256) * 4; }

int spx_dist_fwd_ce(const spx_plan* pl, const void* x, int32_t x_dtype, int32_t B, int32_t HW, const void* packed_bank,
                    const float* packed_p2, const void* packed_head, const int32_t* labels_cls, const uint32_t* proto_key,
                    int32_t J, float* class_distances, float* distances, float* activations, float* logits,
                    const spx_ce* ce, float epsilon, int32_t act_fn, void* stream) {
    if (!ce) return fail("spx_dist_fwd_ce: NULL ce");
    if (labels_cls) {
        if (check_cls("spx_dist_fwd_ce", labels_cls, proto_key, J, HW)) return 1;
        if (!class_distances) return fail("spx_dist_fwd_ce: NULL class_distances");
        if (distances) return fail("spx_dist_fwd_ce: class-gathered and P-wide distances are exclusive");
    }
    return dist_fwd_impl(pl, x, x_dtype, B, HW, packed_bank, packed_p2, packed_head, distances, labels_cls, proto_key,
                         labels_cls ? J : 0, labels_cls ? class_distances : nullptr, activations, logits, epsilon, act_fn,
                         stream, SpxTailFwd{nullptr, 0, nullptr}, ce);
}

int spx_dist_bwd_ce(const spx_plan* pl, const void* x, int32_t x_dtype, int32_t B, int32_t HW, const void* packed_bank,
                    const void* packed_bankT, const float* packed_p2, const void* packed_headT, const int32_t* labels_cls,
                    const uint32_t* proto_key, int32_t J, const float* d_dist, const float* d_class_distances,
                    const float* d_act, const spx_ce* ce, void* dx, void* dx_acc, void* g_out, void* a_out, float epsilon,
                    int32_t act_fn, void* stream) {
    if (!ce) return fail("spx_dist_bwd_ce: NULL ce");
    if (labels_cls && check_cls("spx_dist_bwd_ce", labels_cls, proto_key, J, HW)) return 1;
    if (labels_cls && d_dist) return fail("spx_dist_bwd_ce: class-gathered and P-wide distance gradients are exclusive");
    return dist_bwd_impl(pl, x, x_dtype, B, HW, packed_bank, packed_bankT, packed_p2, packed_headT, d_dist,
                         (labels_cls && d_class_distances) ? labels_cls : nullptr, proto_key, labels_cls ? J : 0,
                         labels_cls ? d_class_distances : nullptr, d_act, nullptr, dx, dx_acc, g_out, a_out, epsilon, act_fn, stream,
                         SpxTailBwd{nullptr, 0, nullptr, nullptr, nullptr}, ce);
}

size_t spx_pixel_outer_workspace_bytes(int64_t M, int32_t n1, int32_t n2) {
    return (size_t)spx_pixel_outer_blocks(M) * n1 * n2 * sizeof(float);
}
int spx_pixel_outer(const float* a, const float* b, int64_t M, int32_t n1, int32_t n2, float* out, void* workspace, void* stream) {
    if (!a || !b || !out || !workspace) return fail("spx_pixel_outer: NULL buffer");
    if (M < 1 || n1 < 1 || n2 < 1 || (long long)n1 * n2 > 8192) return fail("spx_pixel_outer: bad sizes (M=%lld, %d x %d; n1*n2 <= 8192)", (long long)M, n1, n2);
    return hip_status(spx_launch_pixel_outer(a, b, M, n1, n2, out, (float*)workspace, (hipStream_t)stream), "spx_pixel_outer");
}

size_t spx_rows_gemm_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t flags) {
    if (M < 1 || N < 1 || K < 1) return 0;
    return spx_gemm_workspace(M, N, K, flags);
}
int spx_rows_gemm(const float* A, int64_t ras, int64_t kas, const float* B, int64_t rbs, int64_t kbs, float* C, int64_t ldc,
                  int32_t M, int32_t N, int32_t K, int32_t flags, const float* E, int64_t lde, void* workspace, void* stream) {
    if (!A || !B || !C) return fail("spx_rows_gemm: NULL operand");
    if (M < 1 || N < 1 || K < 1) return fail("spx_rows_gemm: bad sizes (M=%d N=%d K=%d)", M, N, K);
    if ((ras != 1 && kas != 1) || (rbs != 1 && kbs != 1) || ras < 1 || kas < 1 || rbs < 1 || kbs < 1)
        return fail("spx_rows_gemm: each operand needs one unit stride (A: %lld, %lld; B: %lld, %lld)", (long long)ras, (long long)kas,
                    (long long)rbs, (long long)kbs);
    if (ldc < N) return fail("spx_rows_gemm: ldc (%lld) < N (%d)", (long long)ldc, N);
    if (flags & ~7) return fail("spx_rows_gemm: unknown flags 0x%x", flags);
    if ((flags & 4) && (!E || lde < N)) return fail("spx_rows_gemm: flag 4 needs E with lde >= N");
    if (spx_gemm_workspace(M, N, K, flags) && !workspace) return fail("spx_rows_gemm: workspace needed (spx_rows_gemm_workspace_bytes)");
    return hip_status(spx_launch_gemm(A, ras, kas, B, rbs, kbs, C, ldc, M, N, K, flags, E, lde, (float*)workspace, (hipStream_t)stream),
                      "spx_rows_gemm");
}

int spx_exp(const float* x, float* y, int64_t n, void* stream) {
    if (!x || !y || n < 1 || (n + 255) / 256 > 0x7fffffffLL) return fail("spx_exp: bad arguments");
    return hip_status(spx_launch_exp(x, nullptr, nullptr, y, n, (hipStream_t)stream), "spx_exp");
}
int spx_exp_bwd(const float* g, const float* y, float* dx, int64_t n, void* stream) {
    if (!g || !y || !dx || n < 1 || (n + 255) / 256 > 0x7fffffffLL) return fail("spx_exp_bwd: bad arguments");
    return hip_status(spx_launch_exp(nullptr, g, y, dx, n, (hipStream_t)stream), "spx_exp_bwd");
}

int spx_group_dense(const float* const* block_ptrs, const int32_t* block_cols, int32_t nblocks, const int32_t* row_block,
                    const int32_t* row_local, const int32_t* col_block, const int32_t* col_local, int32_t U, int32_t P, float* out,
                    void* stream) {
    if (!block_ptrs || !block_cols || !row_block || !row_local || !col_block || !col_local || !out) return fail("spx_group_dense: NULL buffer");
    if (nblocks < 1 || nblocks > SPX_GROUP_BLOCKS_MAX) return fail("spx_group_dense: %d blocks (1..%d)", nblocks, SPX_GROUP_BLOCKS_MAX);
    if (U < 1 || P < 1 || (long long)U * P > 0x7fffffffLL) return fail("spx_group_dense: bad sizes (U=%d P=%d)", U, P);
    SpxGroupDenseArgs a{};
    for (int j = 0; j < nblocks; ++j) {
        if (!block_ptrs[j] || block_cols[j] < 1) return fail("spx_group_dense: block %d is empty", j);
        a.ptrs[j] = block_ptrs[j];
        a.ncols[j] = block_cols[j];
    }
    a.row_block = row_block; a.row_local = row_local; a.col_block = col_block; a.col_local = col_local;
    a.U = U; a.P = P; a.out = out;
    return hip_status(spx_launch_group_dense(a, (hipStream_t)stream), "spx_group_dense");
}
int spx_group_dense_bwd(const float* d_out, const int32_t* flat_row, const int32_t* flat_col, int64_t n, int32_t P, float* d_flat,
                        void* stream) {
    if (!d_out || !flat_row || !flat_col || !d_flat) return fail("spx_group_dense_bwd: NULL buffer");
    if (n < 1 || P < 1 || (n + 255) / 256 > 0x7fffffffLL) return fail("spx_group_dense_bwd: bad sizes");
    return hip_status(spx_launch_group_dense_bwd(d_out, flat_row, flat_col, n, P, d_flat, (hipStream_t)stream), "spx_group_dense_bwd");
}

int spx_ce_fwd(const float* logits, const int32_t* labels, int64_t M, int32_t K, float* lse, int32_t* pred, float* partials,
               void* stream) {
    if (!logits || !labels || !lse || !partials) return fail("spx_ce_fwd: NULL buffer");
    if (M < 1 || K < 1 || (M + 255) / 256 > 0x7fffffffLL) return fail("spx_ce_fwd: bad sizes (M=%lld K=%d)", (long long)M, K);
    return hip_status(spx_launch_ce_fwd(logits, labels, M, K, lse, pred, partials, (hipStream_t)stream), "spx_ce_fwd");
}
int spx_ce_finish(const float* partials, int64_t n_pairs, float* loss, float* count_sum, void* stream) {
    if (!partials || !loss || !count_sum || n_pairs < 1) return fail("spx_ce_finish: bad arguments");
    return hip_status(spx_launch_ce_finish(partials, n_pairs, loss, count_sum, (hipStream_t)stream), "spx_ce_finish");
}
int spx_shift_labels(const void* labels, int32_t is_int64, int64_t n, int32_t* out, void* stream) {
    if (!labels || !out || n < 1 || (n + 255) / 256 > 0x7fffffffLL) return fail("spx_shift_labels: bad arguments");
    return hip_status(spx_launch_shift_labels(labels, is_int64, n, out, (hipStream_t)stream), "spx_shift_labels");
}
int spx_ce_bwd(const float* logits, const float* lse, const int32_t* labels, const float* coef, int64_t M, int32_t K,
               float* d_logits, void* stream) {
    if (!logits || !lse || !labels || !coef || !d_logits) return fail("spx_ce_bwd: NULL buffer");
    if (M < 1 || K < 1 || (M * K + 255) / 256 > 0x7fffffffLL) return fail("spx_ce_bwd: bad sizes (M=%lld K=%d)", (long long)M, K);
    return hip_status(spx_launch_ce_bwd(logits, lse, labels, coef, M, K, d_logits, (hipStream_t)stream), "spx_ce_bwd");
}

size_t spx_bank_bwd_workspace_bytes(const spx_plan* pl, int32_t B, int32_t HW) {
    const int nsplit = spx_bank_bwd_nsplit(*pl, B, HW);
    return spx_bank_bwd_ws_floats(*pl, nsplit) * sizeof(float);
}

int spx_bank_bwd(const spx_plan* pl, const void* x, int32_t x_dtype, int32_t B, int32_t HW, const float* bank,
                 const void* g_in, const void* a_in, const float* d_logits, float* d_bank, float* d_W,
                 void* workspace, void* stream) {
    if (check_plan(pl)) return 1;
    if (!x || !bank || !workspace) return fail("spx_bank_bwd: NULL operand");
    if (x_dtype != 0 && x_dtype != 1) return fail("spx_bank_bwd: x_dtype %d", x_dtype);
    if (d_bank && !g_in) return fail("spx_bank_bwd: d_bank requested without G");
    if (d_W && !a_in) return fail("spx_bank_bwd: d_W requested without the head-gradient scratch of spx_dist_bwd");
    if (d_W && pl->ncb > 1 && !d_logits) return fail("spx_bank_bwd: d_W of a head wider than 32 rows needs d_logits");
    if (pl->channels_per_scale > 256) return fail("spx_bank_bwd: Cs %d > 256 not supported", pl->channels_per_scale);
    SpxBankBwdArgs a;
    a.plan = *pl;
    a.x = x;
    a.bank = bank;
    a.g_in = (const uint16_t*)g_in;
    a.a_in = (const uint16_t*)a_in;
    a.d_logits = d_logits;
    a.d_bank = d_bank;
    a.d_W = d_W;
    a.workspace = (float*)workspace;
    a.B = B;
    a.HW = HW;
    a.vec_ok = x_vec_ok(x, x_dtype, HW);
    a.nsplit = spx_bank_bwd_nsplit(*pl, B, HW);
    return hip_status(spx_launch_bank_bwd(a, x_dtype, (hipStream_t)stream), "spx_bank_bwd");
}

int spx_push_argmin(const float* distances, const int32_t* labels, const float* class_identity, int32_t B, int32_t P,
                    int32_t K, int32_t HW, int32_t void_class, float max_dist, int64_t* indices, float* values,
                    uint64_t* scratch, void* stream) {
    if (!distances || !labels || !class_identity || !indices || !values || !scratch)
        return fail("spx_push_argmin: NULL buffer");
    if (B < 1 || P < 1 || K < 1 || HW < 1) return fail("spx_push_argmin: empty input");
    if (K > 160) return fail("spx_push_argmin: num_classes %d > 160", K);
    if (P > 65535 || B > 65535) return fail("spx_push_argmin: grid too large");
    return hip_status(spx_launch_push_argmin(distances, labels, class_identity, B, P, K, HW, void_class, max_dist,
                                             indices, values, scratch, (hipStream_t)stream),
                      "spx_push_argmin");
}

int spx_argmin_images(const float* values, int32_t N, int32_t P, int64_t* best, void* stream) {
    if (!values || !best) return fail("spx_argmin_images: NULL buffer");
    if (N < 1 || P < 1) return fail("spx_argmin_images: empty input");
    return hip_status(spx_launch_argmin_images(values, N, P, best, (hipStream_t)stream), "spx_argmin_images");
}

int spx_upsample_argext(const float* src, int32_t N, int32_t C, int32_t h, int32_t w, int32_t H, int32_t W,
                        int32_t take_max, int64_t* indices, float* values, void* stream) {
    if (!src || !indices) return fail("spx_upsample_argext: NULL buffer");
    if (N < 1 || C < 1 || h < 1 || w < 1 || H < 1 || W < 1) return fail("spx_upsample_argext: empty input");
    if ((long long)h * w >= (1LL << 31) || (long long)N > 65535) return fail("spx_upsample_argext: map too large");
    return hip_status(spx_launch_upsample_argext(src, N, C, h, w, H, W, take_max ? 1 : 0, indices, values,
                                                 (hipStream_t)stream), "spx_upsample_argext");
}

static int kld_check(const char* who, const float* vals, const int32_t* labels, int32_t B, int32_t J, int32_t HW, int32_t K,
                     const void* out, int pairs) {
    if (!vals || !labels || !out) return fail("%s: NULL buffer", who);
    if (B < 1 || B > 65535 || HW < 1 || K < 1 || J < 1 || J > 16) return fail("%s: bad sizes (B=%d HW=%d K=%d J=%d; J <= 16)", who, B, HW, K, J);
    if ((long long)J * HW >= (1LL << 31)) return fail("%s: J*HW too large", who);
    (void)pairs;        // (the pair and gradient passes tile their per-class tables over class blocks: no K * J * J limit)
    if ((long long)K * J * 12 + (long long)K * 4 + 8 > 60 * 1024) return fail("%s: K*J = %d exceeds the segment tables of the reduction passes", who, K * J);
    return 0;
}

static int kld_check_w(const char* fn, int32_t HW, int32_t W) {
    if (W < 0 || (W > 0 && HW % W != 0)) return fail("%s: W must be 0 or divide HW", fn);
    return 0;
}
int spx_kld_segment_max(const float* vals, const int32_t* labels, int32_t B, int32_t J, int32_t HW, int32_t W, int32_t K,
                        uint32_t* smax_keys, uint32_t* counts, uint32_t* range_keys, void* stream) {
    if (kld_check("spx_kld_segment_max", vals, labels, B, J, HW, K, smax_keys, 0) || kld_check_w("spx_kld_segment_max", HW, W)) return 1;
    return hip_status(spx_launch_kld(0, vals, labels, B, J, HW, W, K, (const float*)counts, (const float*)range_keys, nullptr, nullptr, smax_keys, (hipStream_t)stream), "spx_kld_segment_max");
}
int spx_kld_segment_sumexp(const float* vals, const int32_t* labels, int32_t B, int32_t J, int32_t HW, int32_t W, int32_t K,
                           const uint32_t* smax_keys, uint64_t* ssum_fx, void* stream) {
    if (kld_check("spx_kld_segment_sumexp", vals, labels, B, J, HW, K, ssum_fx, 0) || !smax_keys) return smax_keys ? 1 : fail("spx_kld_segment_sumexp: NULL smax_keys");
    if (kld_check_w("spx_kld_segment_sumexp", HW, W)) return 1;
    return hip_status(spx_launch_kld(1, vals, labels, B, J, HW, W, K, (const float*)smax_keys, nullptr, nullptr, nullptr, ssum_fx, (hipStream_t)stream), "spx_kld_segment_sumexp");
}
int spx_kld_segment_lse(const uint32_t* smax_keys, const uint64_t* ssum_fx, int32_t n, float* lse, const uint32_t* range_keys, int32_t HW,
                        double* scale, void* stream) {
    if (!smax_keys || !ssum_fx || !lse) return fail("spx_kld_segment_lse: NULL pointer");
    if (n < 0) return fail("spx_kld_segment_lse: n < 0");
    if (scale && (!range_keys || HW < 1)) return fail("spx_kld_segment_lse: the scale needs range_keys and HW");
    if (n == 0) return 0;
    return hip_status(spx_launch_kld_lse(smax_keys, ssum_fx, n, lse, range_keys, HW, scale, (hipStream_t)stream), "spx_kld_segment_lse");
}
int spx_kld_gram_loss(const int64_t* a_fx, const double* scale, const uint32_t* counts, const uint8_t* pair_ok, int32_t nseg, int32_t K,
                      int32_t J, float* A, float* Cf, double* partials, float* loss, void* stream) {
    if (!a_fx || !scale || !counts || !pair_ok || !A || !Cf || !partials || !loss) return fail("spx_kld_gram_loss: NULL pointer");
    if (nseg < 1 || K < 1 || J < 1 || J > 16 || nseg % K) return fail("spx_kld_gram_loss: bad sizes (nseg=%d K=%d J=%d)", nseg, K, J);
    return hip_status(spx_launch_kld_gram_loss(a_fx, scale, counts, pair_ok, nseg, K, J, A, Cf, partials, loss, (hipStream_t)stream), "spx_kld_gram_loss");
}
int spx_kld_pair_sums(const float* vals, const int32_t* labels, int32_t B, int32_t J, int32_t HW, int32_t W, int32_t K,
                      const float* lse, const double* scale, int64_t* a_fx, void* stream) {
    if (kld_check("spx_kld_pair_sums", vals, labels, B, J, HW, K, a_fx, 1) || !lse || !scale) return (lse && scale) ? 1 : fail("spx_kld_pair_sums: NULL lse / scale");
    if (kld_check_w("spx_kld_pair_sums", HW, W)) return 1;
    return hip_status(spx_launch_kld(2, vals, labels, B, J, HW, W, K, lse, nullptr, nullptr, scale, a_fx, (hipStream_t)stream), "spx_kld_pair_sums");
}
int spx_kld_backward(const float* vals, const int32_t* labels, int32_t B, int32_t J, int32_t HW, int32_t K,
                     const float* lse, const float* A, const float* Cf, const float* cf_scale, float* grad, void* stream) {
    if (kld_check("spx_kld_backward", vals, labels, B, J, HW, K, grad, 1)) return 1;
    if (!lse || !A || !Cf) return fail("spx_kld_backward: NULL table");
    return hip_status(spx_launch_kld(3, vals, labels, B, J, HW, 0, K, lse, A, Cf, nullptr, grad, (hipStream_t)stream, cf_scale), "spx_kld_backward");
}

}  // extern "C"
