// A C-ABI consumer of libspx_hip.so without Python or torch: HIP runtime for memory, include/spx_hip.h for the
// operators.  Runs the forward (distances + logits) and the push argmin on a small two-scale problem and checks
// them against plain loops over the same numbers (model_multiscale.py:255-330, :243-244;
// push_multiscale_optimization.py:74-91).  Exit code 0 and "c-abi consumer ok" on success.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "spx_hip.h"

#define HIP_OK(x)                                                                     \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
            return 2;                                                                 \
        }                                                                             \
    } while (0)
#define SPX_OK(x)                                                                     \
    do {                                                                              \
        if ((x) != 0) {                                                               \
            std::fprintf(stderr, "%s: %s\n", #x, spx_last_error());                   \
            return 3;                                                                 \
        }                                                                             \
    } while (0)

static uint32_t rng_state = 12345u;
static float rnd() {   // uniform [0, 1) with 8 mantissa bits: exactly representable in bf16
    rng_state = rng_state * 1664525u + 1013904223u;
    return (float)(rng_state >> 24) / 256.0f;
}
static uint16_t bf16_bits(float v) {   // v is bf16-representable: truncation is exact
    uint32_t u;
    std::memcpy(&u, &v, 4);
    return (uint16_t)(u >> 16);
}
template <typename T>
static T* dev_copy(const std::vector<T>& h) {
    T* d = nullptr;
    if (hipMalloc((void**)&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

int main() {
    const int P = 40, K = 5, S = 2, Cs = 32, C = S * Cs, B = 2, H = 9, W = 13, HW = H * W;
    const float eps = 1e-4f;
    if (spx_version() != SPX_ABI_VERSION) {
        std::fprintf(stderr, "ABI %d != header %d\n", spx_version(), SPX_ABI_VERSION);
        return 1;
    }
    const int32_t lo[S] = {0, 20}, hi[S] = {20, 40};
    spx_plan plan;
    SPX_OK(spx_make_plan(P, K, S, Cs, lo, hi, &plan));

    std::vector<float> x((size_t)B * C * HW), bank((size_t)P * Cs), Wl((size_t)K * P);
    for (auto& v : x) v = rnd();
    for (auto& v : bank) v = rnd();
    for (auto& v : Wl) v = rnd() - 0.5f;
    std::vector<uint16_t> xb(x.size());
    for (size_t i = 0; i < x.size(); ++i) xb[i] = bf16_bits(x[i]);
    std::vector<int32_t> target((size_t)B * HW);
    for (auto& t : target) t = (int32_t)(rnd() * (K + 1));         // 0 = void, 1..K
    std::vector<float> ident((size_t)P * K, 0.0f);
    for (int p = 0; p < P; ++p) ident[(size_t)p * K + p % K] = 1.0f;

    uint16_t* d_x = dev_copy(xb);
    float* d_bank = dev_copy(bank);
    float* d_W = dev_copy(Wl);
    int32_t* d_t = dev_copy(target);
    float* d_ident = dev_copy(ident);
    if (!d_x || !d_bank || !d_W || !d_t || !d_ident) return 2;
    void *p_bank, *p_head;
    float *p_p2, *d_dist, *d_logits, *d_val;
    int64_t* d_idx;
    uint64_t* d_scr;
    HIP_OK(hipMalloc(&p_bank, spx_packed_bank_bytes(&plan)));
    HIP_OK(hipMalloc(&p_head, spx_packed_head_bytes(&plan)));
    HIP_OK(hipMalloc((void**)&p_p2, spx_packed_p2_bytes(&plan)));
    HIP_OK(hipMalloc((void**)&d_dist, (size_t)B * P * HW * 4));
    HIP_OK(hipMalloc((void**)&d_logits, (size_t)B * HW * K * 4));
    HIP_OK(hipMalloc((void**)&d_idx, (size_t)B * P * 8));
    HIP_OK(hipMalloc((void**)&d_val, (size_t)B * P * 4));
    HIP_OK(hipMalloc((void**)&d_scr, (size_t)B * P * 8));

    SPX_OK(spx_pack_bank(&plan, d_bank, p_bank, nullptr, p_p2, nullptr));
    SPX_OK(spx_pack_head(&plan, d_W, p_head, nullptr, nullptr));
    SPX_OK(spx_dist_fwd(&plan, d_x, 0, B, HW, p_bank, p_p2, p_head, d_dist, nullptr, d_logits, eps, 0, nullptr));
    SPX_OK(spx_push_argmin(d_dist, d_t, d_ident, B, P, K, HW, 0, 1e10f, d_idx, d_val, d_scr, nullptr));
    HIP_OK(hipDeviceSynchronize());

    std::vector<float> dist((size_t)B * P * HW), logits((size_t)B * HW * K), val((size_t)B * P);
    std::vector<int64_t> idx((size_t)B * P);
    HIP_OK(hipMemcpy(dist.data(), d_dist, dist.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(logits.data(), d_logits, logits.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(idx.data(), d_idx, idx.size() * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(val.data(), d_val, val.size() * 4, hipMemcpyDeviceToHost));

    // an argument error must be reported, not executed
    if (spx_dist_fwd(&plan, d_x, 7, B, HW, p_bank, p_p2, p_head, d_dist, nullptr, d_logits, eps, 0, nullptr) == 0 ||
        std::strlen(spx_last_error()) == 0) {
        std::fprintf(stderr, "bad x_dtype was accepted\n");
        return 1;
    }

    int bad = 0;
    double max_dd = 0.0, max_dl = 0.0;
    for (int b = 0; b < B; ++b)
        for (int px = 0; px < HW; ++px) {
            std::vector<double> act(P);
            for (int p = 0; p < P; ++p) {
                const int s = p < hi[0] ? 0 : 1;
                double d = 0.0;
                for (int c = 0; c < Cs; ++c) {
                    const double xv = x[((size_t)b * C + s * Cs + c) * HW + px], pv = bank[(size_t)p * Cs + c];
                    d += (xv - pv) * (xv - pv);
                }
                const double got = dist[((size_t)b * P + p) * HW + px];
                max_dd = std::fmax(max_dd, std::fabs(got - d) / (1.0 + d));
                // the reference expands (x-p)^2 in fp32 and clamps at 0; the bank enters the MFMA as bf16
                act[p] = std::log((got + 1.0) / (got + eps));
            }
            for (int k = 0; k < K; ++k) {
                double l = 0.0;
                for (int p = 0; p < P; ++p) l += act[p] * Wl[(size_t)k * P + p];
                max_dl = std::fmax(max_dl, std::fabs(l - logits[((size_t)b * HW + px) * K + k]));
            }
        }
    if (max_dd > 1e-4) { std::fprintf(stderr, "distances off by %g\n", max_dd); ++bad; }
    if (max_dl > 2e-3) { std::fprintf(stderr, "logits off by %g\n", max_dl); ++bad; }
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < P; ++p) {
            float best = 0.0f;
            int bi = -1;
            for (int px = 0; px < HW; ++px) {
                const int t = target[(size_t)b * HW + px];
                const float m = (t >= 1 && t <= K) ? ident[(size_t)p * K + t - 1] : 0.0f;
                const float v = dist[((size_t)b * P + p) * HW + px] + 1e10f * (1.0f - m);
                if (bi < 0 || v < best) { best = v; bi = px; }
            }
            if (idx[(size_t)b * P + p] != bi || val[(size_t)b * P + p] != best) {
                std::fprintf(stderr, "argmin (%d,%d): got (%lld, %g) want (%d, %g)\n", b, p, (long long)idx[(size_t)b * P + p],
                             val[(size_t)b * P + p], bi, best);
                ++bad;
            }
        }
    if (bad) return 1;
    std::printf("c-abi consumer ok: max rel distance error %.2e, max logit error %.2e\n", max_dd, max_dl);
    return 0;
}
