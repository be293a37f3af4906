// Micro-benchmark: achievable HBM read bandwidth of the tile access pattern of the prototype kernels
// (C channel rows x TPX pixels per workgroup, 16-B loads) versus a contiguous stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__global__ void k_contig(const u32x4* __restrict__ x, size_t n, uint32_t* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t s = 0;
    for (; i + 3 * stride < n; i += 4 * stride) {
        u32x4 a = x[i], b = x[i + stride], c = x[i + 2 * stride], d = x[i + 3 * stride];
        s ^= a[0] ^ b[1] ^ c[2] ^ d[3];
    }
    if (s == 0x12345) out[0] = s;
}

// one workgroup = one tile of TPX pixels x C rows (bf16): rows of TPX*2 bytes, row stride HW*2 bytes
template <int TPX, int INFLIGHT>
__global__ void k_tile(const char* __restrict__ x, int C, int HW, uint32_t* out) {
    const int tid = threadIdx.x;
    constexpr int PIECES = TPX * 2 / 16;             // 16-B pieces per row
    constexpr int ROWS_PER_PASS = 256 / PIECES;
    const int piece = tid % PIECES, row0 = tid / PIECES;
    const size_t px0 = (size_t)blockIdx.x * TPX;
    const char* base = x + (px0 + piece * 8) * 2;
    uint32_t s = 0;
    for (int r = row0; r < C; r += ROWS_PER_PASS * INFLIGHT) {
        u32x4 v[INFLIGHT];
#pragma unroll
        for (int i = 0; i < INFLIGHT; ++i) {
            int rr = r + i * ROWS_PER_PASS;
            v[i] = rr < C ? *(const u32x4*)(base + (size_t)rr * HW * 2) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < INFLIGHT; ++i) s ^= v[i][0] ^ v[i][3];
    }
    if (s == 0x12345) out[0] = s;
}

template <typename F>
static float time_ms(F f, int reps = 10) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const int C = 256, HW = 1024 * 2048;
    const size_t bytes = (size_t)C * HW * 2;
    char* x;
    uint32_t* out;
    hipMalloc(&x, bytes);
    hipMalloc(&out, 4);
    hipMemset(x, 1, bytes);
    printf("bytes %.2f GB\n", bytes / 1e9);
    for (int blocks : {2048, 8192}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(k_contig, dim3(blocks), dim3(256), 0, 0, (const u32x4*)x, bytes / 16, out); });
        printf("contiguous, %d blocks: %.3f ms  %.2f TB/s\n", blocks, ms, bytes / ms / 1e9);
    }
#define RUN(TPX, INF)                                                                                          \
    {                                                                                                          \
        float ms = time_ms([&] { hipLaunchKernelGGL((k_tile<TPX, INF>), dim3(HW / TPX), dim3(256), 0, 0, x, C, HW, out); }); \
        printf("tile %4d px, %2d loads in flight/thread: %.3f ms  %.2f TB/s\n", TPX, INF, ms, bytes / ms / 1e9);      \
    }
    RUN(128, 2) RUN(128, 4) RUN(128, 8) RUN(128, 16)
    RUN(256, 2) RUN(256, 4) RUN(256, 8) RUN(256, 16)
    RUN(512, 4) RUN(512, 8) RUN(512, 16)
    RUN(64, 4) RUN(64, 16)
    return 0;
}
