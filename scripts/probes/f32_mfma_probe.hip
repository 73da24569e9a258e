// Issue rate and dependent latency of v_mfma_f32_32x32x2_f32: NACC independent accumulators per wave, 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NACC>
__global__ __launch_bounds__(256, 2) void rate(const float* a, float* d, int iters, long long* out) {
    float av[4], bv[4];
    for (int i = 0; i < 4; ++i) { av[i] = a[(threadIdx.x + 7 * i) & 63]; bv[i] = a[(threadIdx.x + 5 * i + 64) & 127]; }
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 16 / NACC; ++rep)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(rep + n) & 3], bv[rep & 3], acc[n], 0, 0, 0);
        asm volatile("" ::: "memory");
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) s += acc[n][1];
    d[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}
template <int NACC> void run(int blocks, const float* da, float* dd, long long* dc) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        rate<NACC><<<blocks, 256>>>(da, dd, iters, dc);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc[2]; hipMemcpy(cyc, dc, 16, hipMemcpyDeviceToHost);
    printf("%d accumulators, %d waves/SIMD: %.2f shader cycles per MFMA of one wave, %.2f GHz, %.1f cycles per MFMA per SIMD, %.1f TFLOP/s\n", NACC, blocks / 256,
           (double)cyc[0] / (iters * 16.0), (double)cyc[0] / cyc[1] * 0.1, 1024.0 * ((double)cyc[0] / cyc[1] * 0.1e9) / ((double)blocks * 4 * iters * 16 / (ms * 1e-3)),
           (double)blocks * 4 * iters * 16 * 4096 / (ms * 1e-3) / 1e12);
}
int main() {
    float *da, *dd; long long* dc;
    hipMalloc(&da, 128 * 4); hipMalloc(&dd, 1024 * 256 * 4); hipMalloc(&dc, 16);
    float h[128]; for (int i = 0; i < 128; ++i) h[i] = 0.001f * (i % 17);
    hipMemcpy(da, h, sizeof(h), hipMemcpyHostToDevice);
    for (int blocks = 256; blocks <= 512; blocks += 256) { run<1>(blocks, da, dd, dc); run<2>(blocks, da, dd, dc); run<4>(blocks, da, dd, dc); }
    return 0;
}
