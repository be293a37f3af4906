// Kernel argument blocks + launcher prototypes shared by the kernel translation units and the C ABI.
#pragma once
#include "spx_common.h"

struct SpxFwdArgs {
    spx_plan plan;
    const void* x;
    const char* packed_bank;
    const float* p2;
    const char* packed_head;
    float* dist;
    float* act;
    float* logits;
    int B, HW, vec_ok;
    int tile_first, tiles_launch;   // this launch covers tiles [tile_first, tile_first + tiles_launch) of every image
    int tile_mul;                   // block -> tile permutation: tile = (block * tile_mul) mod tiles_launch (1 = identity), see spx_tile_mul
    int dist_vec;              // distances 16-B aligned and HW % 4 == 0: 16-B stores of 4 pixels of a row
    // class-gathered distances (spx_dist_fwd_cls): every pixel keeps only the distances to its own class's prototypes
    const int32_t* labels;     // [B, HW] class per pixel (anything outside 0..0xFFFD = none)
    const uint32_t* proto_key; // [npanels][32 npb] (class << 16) | slot per padded prototype row, 0xFFFFFFFF = none
    float* cls_dist;           // [B, J, HW] slot planes
    int J;
    // fused prototype push (spx_dist_push_min): instead of the slot planes, the class-masked minimum of every prototype's
    // distance row over the image, as (ordered float key << 32 | flat pixel index) integer minima
    unsigned long long* push_keys;   // [B, P], caller-initialised to all ones; NULL = off
    float push_max;                  // the reference's max_dist (1e10)
    int push_void, push_K;           // label decode of spx_push_argmin: labels are raw (void_class dropped from 0..K), see spx_hip.h
    // grouping-head tail (spx_dist_fwd_group): logits = W_g . exp(units), units = the head product
    const char* packed_tail;   // W_g A-fragments (spx_pack_group_tail); NULL = no tail
    float* gact;               // [B*HW, U] exp(units) (optional)
    int K2;                    // classes of the tail (<= 32)
    // scale-parallel launch for small pixel grids (grid.y = scale group): workgroup (tile, g) walks only the panels
    // [group_first[g], group_first[g+1]) of scale g and writes its partial logits to logits + g * logits_group_stride
    int ngroups;               // 1 = off
    size_t logits_group_stride;
    int32_t group_first[SPX_MAX_PANELS + 1];
    // fused cross entropy (spx_dist_fwd_ce): statistics of the logits tile, see spx_hip.h
    const int32_t* ce_labels;  // [B, HW]; NULL = off
    float* ce_lse;             // [B*HW]
    int32_t* ce_pred;          // [B*HW] or NULL
    float* ce_partials;        // [tiles * 4 waves][2]
    float eps;
    int act_fn;
    unsigned long long* dbg;   // diagnostic builds only (SPX_DIAG_STAMPS): per-workgroup phase clocks
};
struct SpxBwdArgs {
    spx_plan plan;
    const void* x;
    const char* packed_bank;
    const char* packed_bankT;
    const float* p2;
    const char* packed_headT;
    const float* d_dist;
    const float* d_act;
    const float* d_logits;
    const int32_t* labels;      // class-gathered mode (spx_dist_bwd_cls): see SpxFwdArgs
    const uint32_t* proto_key;
    const float* d_cls_dist;    // [B, J, HW]
    int J;
    // grouping-head tail (spx_dist_bwd_group): d_logits is [B*HW, K2]; dUnits = (W_g^T . dLogits) * exp(units)
    const char* packed_tailT;   // W_g^T A-fragments; NULL = no tail
    const float* gact;          // [B*HW, U] exp(units) of the forward
    float* d_units;             // [B*HW, U] written for the parameter kernel
    const float* d_gact;        // [B*HW, U] gradient on g = exp(units) itself (may be NULL)
    int K2;
    int ngroups;                // scale-parallel launch (see SpxFwdArgs); 1 = off
    int32_t group_first[SPX_MAX_PANELS + 1];
    // fused cross entropy (spx_dist_bwd_ce): d_logits = coef * (softmax(logits) - onehot(label)) formed in the prologue
    const int32_t* ce_labels;   // [B, HW]; NULL = off (then d_logits above is used)
    const float* ce_logits;     // [B*HW, K] the forward's logits
    const float* ce_lse;        // [B*HW]
    const float* ce_coef;       // device scalar
    float* ce_dlogits_out;      // [B*HW, K] written for the parameter kernel (may be NULL)
    void* dx;
    float* dx_acc;              // fp32 [B][C][HW rounded up to 4]: partial dX of scales that span several panels (bf16 features); may be NULL
    uint16_t* g_out;
    uint16_t* a_out;            // head-gradient scratch (spx_common.h): fp32 d_W tile partials (one class block) or the activation blob
    int B, HW, vec_ok;
    int tile_first, tiles_launch;   // this launch covers tiles [tile_first, tile_first + tiles_launch) of every image
    int tile_mul;                   // see SpxFwdArgs
    float eps;
    int act_fn;
    unsigned long long* dbg;
};
struct SpxBankBwdArgs {
    spx_plan plan;
    const void* x;
    const float* bank;
    const uint16_t* g_in;
    const uint16_t* a_in;       // kernel 1's head-gradient scratch
    const float* d_logits;
    float* d_bank;
    float* d_W;
    float* workspace;
    int B, HW, vec_ok, nsplit;
    int ci_first, nci_launch;   // this launch covers chunks [ci_first, ci_first + nci_launch) of every image ...
    int slab_first, nslabs;     // ... with nslabs workgroups writing slabs [slab_first, slab_first + nslabs); nsplit = all slabs
};
// Slot (16-B unit) of lane (r = pixel, h) inside the 1-KiB fragment blob of k-step s2.  Kernel 2 copies the blobs
// verbatim into LDS and reads them back with ds_read_b64_tr_b16 (pixel = k): the 32 lanes of a half-wave then
// touch 4 consecutive pixels x both lane halves x both k-steps, which this permutation spreads over all 16
// 16-B bank groups (the plain slot r + 32 h would put them 4-way on the same banks).
__host__ __device__ inline uint32_t spx_blob_slot(int r, int h, int s2) {
    return (uint32_t)(((r >> 2) * 8 + (r & 3) + 4 * h + 8 * s2) & 63);
}

int spx_tile_mul(int tiles_launch, long long plane_bytes);
hipError_t spx_launch_fwd(const SpxFwdArgs& a, int x_dtype, hipStream_t s);        // a.labels != NULL: class-gathered variant
hipError_t spx_launch_bwd(const SpxBwdArgs& a, int x_dtype, hipStream_t s);
hipError_t spx_launch_bank_bwd(const SpxBankBwdArgs& a, int x_dtype, hipStream_t s);
int spx_bank_bwd_nsplit(const spx_plan& pl, int B, int HW);
size_t spx_bwd_scratch_elems(const spx_plan& pl, int B, int HW);
size_t spx_bank_bwd_ws_floats(const spx_plan& pl, int nsplit);
hipError_t spx_launch_pack_bank(const spx_plan& pl, const float* bank, void* pb, void* pbT, float* p2, hipStream_t s);
hipError_t spx_launch_pack_head(const spx_plan& pl, const float* W, void* ph, void* phT, hipStream_t s);
hipError_t spx_launch_pack_tail(const spx_plan& pl, const float* Wg, int K2, void* pt, void* ptT, hipStream_t s);
hipError_t spx_launch_pack_headT_units(const spx_plan& pl, const float* W, void* phT, hipStream_t s);
// every operand of one forward (+ backward) in one launch (spx_pack_all); the nb_* workgroup ranges are filled by the launcher
struct SpxPackAllArgs {
    spx_plan plan;
    const float* bank;
    const float* W;            // head [K, P] or NULL
    const float* Wg;           // group tail [K2, K] or NULL
    int K2, headT_units;       // headT_units: head^T with the unit index in accumulator order (grouping backward)
    void *packed_bank, *packed_bankT;
    float* p2;
    void *packed_head, *packed_headT, *packed_tail, *packed_tailT;
    int nb_bank, nb_head, nb_headT;
};
hipError_t spx_launch_pack_all(SpxPackAllArgs a, hipStream_t s);
hipError_t spx_launch_ce_finish(const float* partials, long long n, float* loss, float* aux, hipStream_t s);
hipError_t spx_launch_shift_labels(const void* in, int is64, long long n, int32_t* out, hipStream_t s);
// dense form of the per-class group projections (spx_group_dense)
#define SPX_GROUP_BLOCKS_MAX 192
struct SpxGroupDenseArgs {
    const float* ptrs[SPX_GROUP_BLOCKS_MAX];     // device pointer of every block's weight [g_j, n_j]
    int ncols[SPX_GROUP_BLOCKS_MAX];             // n_j
    const int32_t *row_block, *row_local, *col_block, *col_local;
    int U, P;
    float* out;
};
hipError_t spx_launch_group_dense(const SpxGroupDenseArgs& a, hipStream_t s);
hipError_t spx_launch_group_dense_bwd(const float* d_out, const int32_t* rows, const int32_t* cols, long long n, int P, float* d_flat,
                                      hipStream_t s);
hipError_t spx_launch_push_argmin(const float* dist, const int32_t* labels, const float* ident, int B, int P, int K,
                                  int HW, int void_class, float max_dist, int64_t* idx, float* val,
                                  uint64_t* scratch, hipStream_t s);
hipError_t spx_launch_argmin_images(const float* values, int N, int P, int64_t* best, hipStream_t s);
hipError_t spx_launch_upsample_argext(const float* src, int N, int C, int h, int w, int H, int W, int take_max,
                                      int64_t* idx, float* val, hipStream_t s);
hipError_t spx_launch_sum_groups(const float* parts, size_t n, int groups, float* out, hipStream_t s);
int spx_split_groups(const spx_plan& pl, int B, int HW, int32_t* group_first);
hipError_t spx_launch_group_tail(const float* parts, int groups, long long M, int U, const float* Wg, int K2, float* gact,
                                 float* logits, const int32_t* labels, float* lse, int32_t* pred, float* partials, hipStream_t s);
hipError_t spx_launch_push_finalize(const uint64_t* scratch, int n, int64_t* idx, float* val, hipStream_t s);
#ifdef SPX_DIAG
void spx_gemm_force(int wm, int splits);
#endif
size_t spx_gemm_workspace(int M, int N, int K, int flags);
hipError_t spx_launch_gemm(const float* A, long long ras, long long kas, const float* B, long long rbs, long long kbs,
                           float* C, long long ldc, int M, int N, int K, int flags, const float* E, long long lde,
                           float* ws, hipStream_t s);
int spx_pixel_outer_blocks(long long M);
hipError_t spx_launch_pixel_outer(const float* a, const float* b, long long M, int n1, int n2, float* out, float* parts, hipStream_t s);
hipError_t spx_launch_ce_fwd(const float* logits, const int32_t* labels, long long M, int K, float* lse, int32_t* pred,
                             float* partials, hipStream_t s);
hipError_t spx_launch_ce_bwd(const float* logits, const float* lse, const int32_t* labels, const float* coef, long long M, int K,
                             float* d_logits, hipStream_t s);
hipError_t spx_launch_exp(const float* x, const float* g, const float* y, float* out, long long n, hipStream_t s);
hipError_t spx_launch_kld_lse(const uint32_t* keys, const uint64_t* ssum_fx, int n, float* lse, const uint32_t* range_keys, int HW,
                              double* scale_out, hipStream_t s);
hipError_t spx_launch_kld_gram_loss(const int64_t* a_fx, const double* scale, const uint32_t* counts, const uint8_t* pair_ok, int nseg, int K,
                                    int J, float* A, float* Cf, double* part, float* loss, hipStream_t s);
hipError_t spx_launch_kld(int pass, const float* vals, const int32_t* labels, int B, int J, int HW, int W, int K, const float* t0,
                          const float* t1, const float* t2, const double* scale, void* out, hipStream_t s, const float* cf_scale = nullptr);
