// Issue rate of v_mfma_i32_16x16x64_i8 / v_mfma_f32_16x16x32_f16 in a loop shaped like the bits GEMM's k-step (24 accumulators,
// 4 A operands x 6 B operands), one or two waves per SIMD.  Build twice: default (accumulators in AGPRs) and with
// -mllvm -amdgpu-mfma-vgpr-form=1 (accumulators in VGPRs, as libbmf_hip.so is built).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

template <int KIND>
__global__ __launch_bounds__(256, 2) void rate(const i32x4* a, const i32x4* b, i32x4* d, int iters, long long* out) {
    i32x4 av[4], bv[6];
    for (int i = 0; i < 4; ++i) av[i] = a[(threadIdx.x + 7 * i) & 63];
    for (int i = 0; i < 6; ++i) bv[i] = b[(threadIdx.x + 5 * i) & 63];
    i32x4 acc[4][6];
    f32x4 facc[4][6];
    for (int m = 0; m < 4; ++m)
        for (int n = 0; n < 6; ++n) { acc[m][n] = i32x4{0, 0, 0, 0}; facc[m][n] = f32x4{0, 0, 0, 0}; }
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 6; ++n) {
                    if (KIND == 0) acc[m][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av[m], bv[n], acc[m][n], 0, 0, 0);
                    else facc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[m]), __builtin_bit_cast(f16x8, bv[n]), facc[m][n], 0, 0, 0);
                }
        asm volatile("" ::: "memory");
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    i32x4 s = {0, 0, 0, 0};
    for (int m = 0; m < 4; ++m)
        for (int n = 0; n < 6; ++n) { s += acc[m][n]; s[0] += (int)facc[m][n][1]; }
    d[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

int main() {
    i32x4 *da, *db, *dd; long long* dc;
    hipMalloc(&da, 64 * 16); hipMalloc(&db, 64 * 16); hipMalloc(&dd, 1024 * 256 * 16); hipMalloc(&dc, 16);
    std::vector<unsigned char> ha(64 * 16), hb(64 * 16);
    srand(1);
    for (auto& v : ha) v = (rand() % 100) < 8;        // A: 0/1 at the density of the benchmark's X
    for (auto& v : hb) v = (unsigned char)rand();
    hipMemcpy(da, ha.data(), 64 * 16, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), 64 * 16, hipMemcpyHostToDevice);
    const char* names[2] = {"mfma_i32_16x16x64_i8 ", "mfma_f32_16x16x32_f16"};
    for (int rep = 0; rep < 3; ++rep)
        for (int blocks = 256; blocks <= 512; blocks += 256)
            for (int kind = 0; kind < 2; ++kind) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                const int iters = 8000;
                hipEventRecord(e0);
                if (kind == 0) rate<0><<<blocks, 256>>>(da, db, dd, iters, dc); else rate<1><<<blocks, 256>>>(da, db, dd, iters, dc);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                long long cyc[2]; hipMemcpy(cyc, dc, 16, hipMemcpyDeviceToHost);
                if (rep > 0)
                    printf("%s %d waves/SIMD: %.3f ms, %.2f shader cycles per MFMA of one wave, %.2f GHz, chip %.1f G MFMA/s = %.2f cycles per MFMA per SIMD\n",
                           names[kind], blocks / 256, ms, (double)cyc[0] / (iters * 48.0), (double)cyc[0] / cyc[1] * 0.1,
                           (double)blocks * 4 * iters * 48 / ms / 1e6, 1024.0 * ((double)cyc[0] / cyc[1] * 0.1e9) / ((double)blocks * 4 * iters * 48 / (ms * 1e-3)));
            }
    return 0;
}
