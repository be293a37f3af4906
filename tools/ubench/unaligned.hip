// Probe: do buffer loads / stores of 4, 8 and 16 bytes work at 2-byte-aligned addresses on gfx950 (ROCm unaligned access mode)?
// hipcc --offload-arch=gfx950 -O2 -o unaligned unaligned.hip && ./unaligned
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
__device__ __amdgpu_buffer_rsrc_t mk(const void* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x80000000u, 0x00020000); }
__global__ void k_load(const uint8_t* src, uint32_t off, uint32_t* out) {
    const uint32_t vo = off + threadIdx.x * 16;
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(mk(src), vo, 0, 0);
    u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(mk(src), vo, 0, 0);
    uint32_t d = __builtin_amdgcn_raw_buffer_load_b32(mk(src), vo, 0, 0);
    uint32_t* o = out + threadIdx.x * 8;
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; o[4] = w[0]; o[5] = w[1]; o[6] = d; o[7] = 0;
}
__global__ void k_store(uint8_t* dst, uint32_t off) {
    const uint32_t vo = off + threadIdx.x * 16;
    u32x4 v = {0x03020100u + threadIdx.x, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu};
    __builtin_amdgcn_raw_buffer_store_b128(v, mk(dst), vo, 0, 0);
}
int main() {
    const int N = 4096;
    std::vector<uint8_t> h(N);
    for (int i = 0; i < N; ++i) h[i] = (uint8_t)(i * 7 + 3);
    uint8_t *d, *d2; uint32_t* o;
    hipMalloc(&d, N); hipMalloc(&d2, N); hipMalloc(&o, 64 * 8 * 4);
    hipMemcpy(d, h.data(), N, hipMemcpyHostToDevice);
    int bad_total = 0;
    for (uint32_t off : {0u, 4u, 8u, 2u, 6u, 10u, 14u, 1u}) {
        k_load<<<1, 64>>>(d, off, o);
        std::vector<uint32_t> r(64 * 8);
        if (hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("load off %u: FAULT\n", off); return 1; }
        int bad = 0;
        for (int t = 0; t < 64; ++t) {
            uint8_t ex[16]; memcpy(ex, &h[off + t * 16], 16);
            bad += memcmp(ex, &r[t * 8], 16) != 0; bad += memcmp(ex, &r[t * 8 + 4], 8) != 0; bad += memcmp(ex, &r[t * 8 + 6], 4) != 0;
        }
        printf("load  off %2u: %s (%d mismatches)\n", off, bad ? "WRONG" : "ok", bad);
        bad_total += bad;
        hipMemset(d2, 0xEE, N);
        k_store<<<1, 64>>>(d2, off);
        std::vector<uint8_t> s(N);
        if (hipMemcpy(s.data(), d2, N, hipMemcpyDeviceToHost) != hipSuccess) { printf("store off %u: FAULT\n", off); return 1; }
        bad = 0;
        for (int i = 0; i < N; ++i) {
            int rel = i - (int)off; uint8_t ex = 0xEE;
            if (rel >= 0 && rel < 64 * 16) { int t = rel / 16, b = rel % 16; ex = (uint8_t)b; if (b < 4) { uint32_t w0 = 0x03020100u + t; ex = (uint8_t)(w0 >> (8 * b)); } }
            bad += s[i] != ex;
        }
        printf("store off %2u: %s (%d mismatches)\n", off, bad ? "WRONG" : "ok", bad);
        bad_total += bad;
    }
    printf("total mismatches %d\n", bad_total);
    return 0;
}
