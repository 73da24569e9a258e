// Probe of v_smfmac_i32_16x16x128_i8 (2:4 structured-sparse A) on gfx950: operand layout and issue rate.
// Build: hipcc --offload-arch=gfx950 -O3 -o smfmac_probe smfmac_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

// one wave: D = smfmac(A, B, 0, idx); every operand given per lane from memory
__global__ void one(const i32x4* a, const i32x8* b, const int* idx, i32x4* d) {
    i32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_smfmac_i32_16x16x128_i8(a[threadIdx.x], b[threadIdx.x], acc, idx[threadIdx.x], 0, 0);
    d[threadIdx.x] = acc;
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8p;
typedef __attribute__((ext_vector_type(4))) float f32x4p;
// KIND 0: mfma_i32_16x16x64_i8, 1: smfmac_i32_16x16x128_i8, 2: mfma_f32_16x16x32_bf16.  out[0] = shader cycles, out[1] = 100 MHz ticks
template <int KIND>
__global__ __launch_bounds__(256) void rate(const i32x4* a, const i32x8* b, i32x4* d, int iters, long long* out) {
    i32x4 av = a[threadIdx.x & 63];
    i32x8 bv = b[threadIdx.x & 63];
    i32x4 bd = {bv[0], bv[1], bv[2], bv[3]};
    i32x4 acc[8];
    f32x4p facc[8];
    for (int i = 0; i < 8; ++i) { acc[i] = i32x4{0, 0, 0, 0}; facc[i] = f32x4p{0, 0, 0, 0}; }
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 1) acc[i] = __builtin_amdgcn_smfmac_i32_16x16x128_i8(av, bv, acc[i], 0x44444444, 0, 0);
            else if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bd, acc[i], 0, 0, 0);
            else facc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8p, av), __builtin_bit_cast(bf16x8p, bd), facc[i], 0, 0, 0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    i32x4 s = acc[0];
    for (int i = 1; i < 8; ++i) s += acc[i];
    for (int i = 0; i < 8; ++i) s[0] += (int)facc[i][0];
    d[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

int main() {
    // ---- layout: A one-hot in compressed slot s (0..15) of lane L, index value v (0..3) for that slot; B[k][c] coded per k
    std::vector<int> hidx(64);
    i32x4* da; i32x8* db; int* didx; i32x4* dd;
    hipMalloc(&da, 64 * sizeof(i32x4)); hipMalloc(&db, 64 * sizeof(i32x8)); hipMalloc(&didx, 64 * 4); hipMalloc(&dd, 64 * sizeof(i32x4));
    // B: lane (c = lane & 15, g = lane >> 4) holds 32 bytes; we do not know which k they are.  Code each (lane, byte) uniquely in
    // two passes: pass 0 value = byte index + 1 (1..32), pass 1 value = g + 1.  Row r of D then tells which B byte(s) a given A
    // slot multiplied: D[r][c] = sum over selected k of B[k][c].
    printf("A slot -> (B lane group, B byte) for idx field value v\n");
    for (int slot = 0; slot < 16; ++slot) {
        for (int v = 0; v < 4; ++v) {
            int res[2][4];
            for (int lg = 0; lg < 4; ++lg) {   // A lane group under test
                for (int pass = 0; pass < 2; ++pass) {
                    std::vector<unsigned char> ha(64 * 16, 0), hb(64 * 32, 0);
                    for (int l = 0; l < 64; ++l) {
                        for (int by = 0; by < 32; ++by) hb[l * 32 + by] = pass == 0 ? (unsigned char)(by + 1) : (unsigned char)((l >> 4) + 1);
                    }
                    int lane = lg * 16 + 3;  // row 3
                    ha[lane * 16 + slot] = 1;
                    for (int l = 0; l < 64; ++l) hidx[l] = 0;
                    // index bits of slot: assume 2 bits per slot, slot s at bits [2s, 2s+2)
                    for (int l = 0; l < 64; ++l) {
                        unsigned w = 0;
                        for (int s2 = 0; s2 < 16; ++s2) w |= (unsigned)((s2 == slot ? v : (s2 & 1 ? 3 : 2)) & 3) << (2 * s2);
                        hidx[l] = (int)w;
                    }
                    hipMemcpy(da, ha.data(), 64 * 16, hipMemcpyHostToDevice);
                    hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice);
                    hipMemcpy(didx, hidx.data(), 64 * 4, hipMemcpyHostToDevice);
                    one<<<1, 64>>>(da, db, didx, dd);
                    int hd[64 * 4];
                    hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
                    // D layout: col = lane & 15, row = 4 * (lane >> 4) + i ; row 3 -> lane group 0, i = 3; column 5
                    res[pass][lg] = hd[(0 * 16 + 5) * 4 + 3];
                }
            }
            printf("slot %2d v %d :", slot, v);
            for (int lg = 0; lg < 4; ++lg) printf("  A-lanegroup %d -> B byte %2d of B-lanegroup %d |", lg, res[0][lg] - 1, res[1][lg] - 1);
            printf("\n");
        }
    }
    // ---- rate: random operands (power matters), 1 and 2 waves per SIMD, kinds interleaved, after a warm-up
    {
        std::vector<unsigned char> ha(64 * 16), hb(64 * 32);
        srand(1);
        for (auto& v : ha) v = rand() & 1;            // A: 0/1 like the bits GEMM
        for (auto& v : hb) v = (unsigned char)rand();  // B: random digits
        hipMemcpy(da, ha.data(), 64 * 16, hipMemcpyHostToDevice);
        hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice);
    }
    long long* dc; hipMalloc(&dc, 16);
    i32x4* dbig; hipMalloc(&dbig, 256 * 1024 * sizeof(i32x4));
    const char* names[3] = {"mfma_i32_16x16x64_i8   ", "smfmac_i32_16x16x128_i8", "mfma_f32_16x16x32_bf16 "};
    for (int rep = 0; rep < 4; ++rep)
        for (int blocks = 256; blocks <= 512; blocks += 256)
            for (int kind = 0; kind < 3; ++kind) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                const int iters = 40000;
                hipEventRecord(e0);
                if (kind == 0) rate<0><<<blocks, 256>>>(da, db, dbig, iters, dc);
                else if (kind == 1) rate<1><<<blocks, 256>>>(da, db, dbig, iters, dc);
                else rate<2><<<blocks, 256>>>(da, db, dbig, iters, dc);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                long long cyc[2]; hipMemcpy(cyc, dc, 16, hipMemcpyDeviceToHost);
                const double per_wave = (double)cyc[0] / (iters * 8.0), ghz = (double)cyc[0] / (double)cyc[1] * 0.1;
                if (rep > 0)
                    printf("%s %d waves/SIMD: %.3f ms, %.2f shader cycles per MFMA per SIMD at %.2f GHz, chip %.1f G MFMA/s\n", names[kind], blocks / 256, ms,
                           per_wave / (blocks / 256), ghz, (double)blocks * 4 * iters * 8 / ms / 1e6);
            }
    return 0;
}
