// v_mfma_scale_f32_16x16x128_f8f6f4 with an FP4 (e2m1) A operand and an FP6 (e2m3) B operand, unit block scales:
//   (1) semantics: which (lane, bit field) holds element (row, k) of A and (k, col) of B; are sums of small integers exact
//       (bits x digits d/8, |d| <= 15, fp32 accumulate) including the e2m3 subnormal codes;
//   (2) issue rate next to v_mfma_i32_16x16x64_i8 in a loop shaped like the bits GEMM's k-step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void one(const i32x8* a, const i32x8* b, f32x4* d, int scale) {
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[threadIdx.x], b[threadIdx.x], c, 4, 2, 0, scale, 0, scale);
    d[threadIdx.x] = c;
}

template <int KIND>
__global__ __launch_bounds__(256, 2) void rate(const i32x8* a, const i32x8* b, i32x4* d, int iters, long long* out) {
    constexpr int NB = KIND == 0 ? 6 : 5;   // 3 int8 planes x 2 column tiles / 5 fp6 planes x ... (per 64 / 128 reduction indices)
    i32x8 av[4], bv[NB];
    for (int i = 0; i < 4; ++i) av[i] = a[(threadIdx.x + 7 * i) & 63];
    for (int i = 0; i < NB; ++i) bv[i] = b[(threadIdx.x + 5 * i) & 63];
    i32x4 acc[4][NB];
    f32x4 facc[4][NB];
    for (int m = 0; m < 4; ++m)
        for (int n = 0; n < NB; ++n) { acc[m][n] = i32x4{0, 0, 0, 0}; facc[m][n] = f32x4{0, 0, 0, 0}; }
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    if (KIND == 0) {
                        i32x4 x = {av[m][0], av[m][1], av[m][2], av[m][3]}, y = {bv[n][0], bv[n][1], bv[n][2], bv[n][3]};
                        acc[m][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, y, acc[m][n], 0, 0, 0);
                    } else
                        facc[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av[m], bv[n], facc[m][n], 4, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                }
        asm volatile("" ::: "memory");
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    i32x4 s = {0, 0, 0, 0};
    for (int m = 0; m < 4; ++m)
        for (int n = 0; n < NB; ++n) { s += acc[m][n]; s[0] += (int)facc[m][n][1]; }
    d[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

static int e2m3_code(int dgt) {   // value dgt / 8, |dgt| <= 15: sign, 2 exponent bits, 3 mantissa bits (e = 0: subnormal m/8; e = 1: 1 + m/8)
    const int s = dgt < 0, v = abs(dgt);
    return (s << 5) | (v < 8 ? v : (1 << 3) | (v - 8));
}

int main() {
    // ---- semantics ----
    std::vector<int> A(16 * 128), B(128 * 16);   // A[r][k] in {0,1}; B[k][c] digits
    srand(3);
    for (auto& v : A) v = (rand() % 100) < 30;
    for (auto& v : B) v = rand() % 31 - 15;
    for (int acode = 0; acode < 2; ++acode) {    // FP4 code of a set bit: 0b0010 (1.0, normal) / 0b0001 (0.5, subnormal)
        std::vector<unsigned> ha(64 * 8, 0u), hb(64 * 8, 0u);
        for (int l = 0; l < 64; ++l) {
            const int rc = l & 15, kb = 32 * (l >> 4);
            for (int j = 0; j < 32; ++j) {
                if (A[rc * 128 + kb + j]) ha[l * 8 + (4 * j) / 32] |= (acode == 0 ? 2u : 1u) << ((4 * j) % 32);
                const unsigned code = (unsigned)e2m3_code(B[(kb + j) * 16 + rc]);
                const int bit = 6 * j;
                hb[l * 8 + bit / 32] |= code << (bit % 32);
                if (bit % 32 > 26) hb[l * 8 + bit / 32 + 1] |= code >> (32 - bit % 32);
            }
        }
        i32x8 *da, *db; f32x4* dd;
        hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dd, 64 * 16);
        hipMemcpy(da, ha.data(), 64 * 32, hipMemcpyHostToDevice);
        hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice);
        for (int scale : {0x7f7f7f7f, 0}) {
            one<<<1, 64>>>(da, db, dd, scale);
            float hd[256];
            hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
            int bad = 0; double worst = 0;
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < 4; ++i) {
                    const int row = (l >> 4) * 4 + i, col = l & 15;
                    long ref = 0;
                    for (int k = 0; k < 128; ++k) ref += (long)A[row * 128 + k] * B[k * 16 + col];
                    const double want = (double)ref / 8.0 * (acode == 0 ? 1.0 : 0.5);
                    if ((double)hd[l * 4 + i] != want) { ++bad; if (bad < 4) printf("  mismatch lane %d reg %d: got %g want %g\n", l, i, hd[l * 4 + i], want); }
                    worst = fmax(worst, fabs((double)hd[l * 4 + i] - want));
                }
            printf("A code %s, scale word 0x%08x: %d of 256 outputs differ from the exact integer sums (max abs diff %g)\n",
                   acode == 0 ? "0b0010 (1.0)" : "0b0001 (0.5, subnormal)", (unsigned)scale, bad, worst);
        }
    }
    // ---- rate ----
    i32x8 *da, *db; i32x4* dd; long long* dc;
    hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dd, 1024 * 256 * 16); hipMalloc(&dc, 16);
    std::vector<unsigned> ha(64 * 8), hb(64 * 8);
    for (auto& v : ha) { v = 0; for (int j = 0; j < 8; ++j) if ((rand() % 100) < 8) v |= 2u << (4 * j); }
    for (auto& v : hb) v = (unsigned)rand() & 0x7df7df7du;
    hipMemcpy(da, ha.data(), 64 * 32, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice);
    const char* names[2] = {"mfma_i32_16x16x64_i8 (24 per step)       ", "mfma_scale_f32_16x16x128 fp4 x fp6 (20/step)"};
    for (int rep = 0; rep < 3; ++rep)
        for (int blocks = 256; blocks <= 512; blocks += 256)
            for (int kind = 0; kind < 2; ++kind) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                const int iters = 8000;
                const double per_it = 2.0 * 4 * (kind == 0 ? 6 : 5);
                hipEventRecord(e0);
                if (kind == 0) rate<0><<<blocks, 256>>>(da, db, dd, iters, dc); else rate<1><<<blocks, 256>>>(da, db, dd, iters, dc);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                long long cyc[2]; hipMemcpy(cyc, dc, 16, hipMemcpyDeviceToHost);
                if (rep > 0)
                    printf("%s %d waves/SIMD: %.3f ms, %.2f shader cycles per MFMA of one wave, %.2f GHz, %.2f cycles per MFMA per SIMD\n",
                           names[kind], blocks / 256, ms, (double)cyc[0] / (iters * per_it), (double)cyc[0] / cyc[1] * 0.1,
                           1024.0 * ((double)cyc[0] / cyc[1] * 0.1e9) / ((double)blocks * 4 * iters * per_it / (ms * 1e-3)));
            }
    return 0;
}
