// Micro-benchmark: ceiling of the pixel-side backward kernel's (K1) HBM traffic shape with no arithmetic.
// Per workgroup (128 pixels, 4 waves, two workgroups per CU as in K1):
//   phase A  read the X tile: C = 256 bf16 rows of 256 B (16-B loads)                                   64 KB
//   phase B  read dDist: P = 190 fp32 rows of 512 B, a pixel per lane (dword loads, 16 in flight)       95 KB
//            write the G and a blobs: 2 x 48 KB contiguous, 16-B stores                                 96 KB
//   phase C  read the X tile again, write the dX tile (256 rows of 256 B, 16-B stores)                 128 KB
// = 383 KB per workgroup, 6.4 GB per launch at 2 Mpx: K1's own byte count (profiles/traffic.json: 6.7 GB).
// Measured (MI355X): A 0.18 ms, B 0.605 ms, C 0.40 ms, all three in one kernel 1.14 ms = 5.65 TB/s.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int PHASES>   // bit 0: A, bit 1: B, bit 2: C
__global__ __launch_bounds__(256, 2) void k1_shape(const char* __restrict__ x, const float* __restrict__ dd, char* __restrict__ blobs,
                                                   char* __restrict__ dx, int C, int P, int HW, uint32_t* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t px0 = (size_t)blockIdx.x * 128;
    const int piece = tid & 15, row0 = tid >> 4;          // 16 pieces of 16 B per 256-B row, 16 rows per pass
    uint32_t s = 0;
    if (PHASES & 1) {
        const char* base = x + (px0 + piece * 8) * 2;
        for (int r = row0; r < C; r += 64) {
            u32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = *(const u32x4*)(base + (size_t)(r + 16 * i) * HW * 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) s ^= v[i][0] ^ v[i][3];
        }
    }
    if (PHASES & 2) {
        const float* base = dd + px0 + 32 * wave + (lane & 31);
        char* bl = blobs + ((size_t)blockIdx.x * 4 + wave) * 6 * 2 * 2048;       // G then a, per wave 6 blocks x 2 KB each
        for (int pb = 0; pb < 6; ++pb) {
            float v[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = pb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                v[reg] = base[(size_t)(row < P ? row : P - 1) * HW];
            }
            uint32_t q = 0;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) q ^= __float_as_uint(v[reg]);
            s ^= q;
            const u32x4 w = {q, q + 1, q + 2, q + 3};
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                *(u32x4*)(bl + (pb * 2 + k) * 1024 + lane * 16) = w;                       // G fragments
                *(u32x4*)(bl + 6 * 2048 + (pb * 2 + k) * 1024 + lane * 16) = w;            // a fragments
            }
        }
    }
    if (PHASES & 4) {
        const char* base = x + (px0 + piece * 8) * 2;
        char* ob = dx + (px0 + piece * 8) * 2;
        for (int r = row0; r < C; r += 64) {
            u32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = *(const u32x4*)(base + (size_t)(r + 16 * i) * HW * 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i][0] ^= s;
                *(u32x4*)(ob + (size_t)(r + 16 * i) * HW * 2) = v[i];
            }
        }
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
    const int C = 256, P = 190, HW = 1024 * 2048, tiles = HW / 128;
    const size_t xb = (size_t)C * HW * 2, db = (size_t)P * HW * 4, bb = (size_t)tiles * 4 * 6 * 2 * 2048;    // G + a: 2 x 48 KB per tile
    char *x, *blobs, *dx;
    float* dd;
    uint32_t* out;
    hipMalloc(&x, xb);
    hipMalloc(&dx, xb);
    hipMalloc(&dd, db);
    hipMalloc(&blobs, bb);
    hipMalloc(&out, 4);
    hipMemset(x, 1, xb);
    hipMemset(dd, 0, db);
    const double bA = (double)xb, bB = (double)db + (double)bb, bC = 2.0 * xb;
#define RUN(PH, BYTES, NAME)                                                                                           \
    {                                                                                                                  \
        float ms = time_ms([&] { hipLaunchKernelGGL(k1_shape<PH>, dim3(tiles), dim3(256), 0, 0, x, dd, blobs, dx, C, P, HW, out); }); \
        printf("%-34s %.3f ms  %.2f GB  %.2f TB/s\n", NAME, ms, (BYTES) / 1e9, (BYTES) / ms / 1e9);                    \
    }
    RUN(1, bA, "A: X tile read");
    RUN(2, bB, "B: dDist read + blob writes");
    RUN(4, bC, "C: X read + dX write");
    RUN(7, bA + bB + bC, "A+B+C: K1's traffic, no arithmetic");
    return 0;
}
