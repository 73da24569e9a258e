// What an LDS-DMA piece, a ds_read_b128 and a bit-expansion pair cost the ISSUING wave between back-to-back v_mfma_i32_16x16x64_i8,
// at ONE wave per SIMD (the question behind a 64 x 64 wave tile for the bits GEMM: every stall of the only wave idles the matrix
// pipe).  One iteration = 96 MFMAs (a 128-index stage of a 64 x 64 x 3-plane wave tile) with NDMA 1-KiB pieces, NLDS fragment
// reads and NVALU shift/and pairs spread evenly between them.  Build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) int i32x4;

// FORM 0: SGPR base + 32-bit lane offset; 1: 64-bit lane address
// SHARED 1: every workgroup streams the same 256 KiB (L2 hits, like the factor panel); 0: 64 KiB of its own per wave (64 MiB in all:
// Infinity Cache)
template <int NDMA, int NLDS, int NVALU, int FORM, int WAVES_PER_SIMD, int SHARED = 1>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void probe(const char* __restrict__ src, size_t src_bytes, const i32x4* ab, i32x4* d, int iters, long long* out) {
    __shared__ __attribute__((aligned(16))) char smem[64 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    i32x4 av[4], bv[6];
    for (int i = 0; i < 4; ++i) av[i] = ab[(threadIdx.x + 7 * i) & 63];
    for (int i = 0; i < 6; ++i) bv[i] = ab[64 + ((threadIdx.x + 5 * i) & 63)];
    i32x4 acc[4][6];
    for (int m = 0; m < 4; ++m)
        for (int n = 0; n < 6; ++n) acc[m][n] = i32x4{0, 0, 0, 0};
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned m0v = lds0 + wave * 16384;   // each wave its own 16 KiB: pieces round-robin inside
    const unsigned rd = lds0 + wave * 16384 + lane * 16;
    const char* base = src + ((size_t)(SHARED ? 0 : blockIdx.x) * 4 + wave) * (1 << 16);   // 64 KiB per wave (L2 / Infinity Cache resident, like the panel), walked piece by piece, wrapping
    unsigned off = lane * 16;
    unsigned w = ab[threadIdx.x & 63][0];
    i32x4 frag[4];
    for (int i = 0; i < 4; ++i) frag[i] = i32x4{0, 0, 0, 0};
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 96; ++q) {
            const int m = (q / 6) & 3, n = q % 6;
            acc[m][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av[m], bv[n], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (NDMA > 0 && (q * NDMA) / 96 != ((q + 1) * NDMA) / 96) {
                unsigned keep;
                const unsigned dst = m0v + (((q * NDMA) / 96) & 15) * 1024;
                if (FORM == 0)
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "s"(dst), "v"(off), "s"(base) : "memory");
                else {
                    const char* p = base + off;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "s"(dst), "v"(p) : "memory");
                }
                off = (off + 1024) & ((1 << 16) - 1);
            }
            if (NLDS > 0 && (q * NLDS) / 96 != ((q + 1) * NLDS) / 96) {
                const int f = ((q * NLDS) / 96) & 3;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(frag[f]) : "v"(rd), "n"(0));
            }
            if (NVALU > 0 && (q * NVALU) / 96 != ((q + 1) * NVALU) / 96) {
                const int e = ((q * NVALU) / 96) & 3;
                av[(q >> 4) & 3][e] = (int)((w >> (q & 7)) & 0x01010101u);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (NDMA > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA > 16 ? 16 : NDMA) : "memory");
        if (NLDS > 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(frag[i]));
        }
        w = w * 1664525u + 1013904223u;
        asm volatile("" ::: "memory");
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    i32x4 s = frag[0] + frag[1] + frag[2] + frag[3];
    for (int m = 0; m < 4; ++m)
        for (int n = 0; n < 6; ++n) s += acc[m][n];
    d[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

template <int NDMA, int NLDS, int NVALU, int FORM, int WPS, int SHARED = 1>
void run(const char* name, const char* src, size_t bytes, const i32x4* ab, i32x4* d, long long* dc) {
    const int iters = 3000, blocks = 256 * WPS;
    float best = 1e9f; long long cyc[2] = {0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        probe<NDMA, NLDS, NVALU, FORM, WPS, SHARED><<<blocks, 256>>>(src, bytes, ab, d, iters, dc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; hipMemcpy(cyc, dc, 16, hipMemcpyDeviceToHost); }
    }
    const double per_it = (double)cyc[0] / iters;
    printf("%-44s waves/SIMD %d: %8.1f cycles per 96-MFMA iteration (pipe needs 1536/wave), busy %.1f %%, clock %.2f GHz, %.3f ms\n", name, WPS, per_it,
           100.0 * 1536.0 * WPS / per_it, (double)cyc[0] / cyc[1] * 0.1, best);
}

int main() {
    const size_t bytes = (size_t)2048 * (1 << 18);   // 512 MiB
    char* src; i32x4 *ab, *d; long long* dc;
    hipMalloc(&src, bytes); hipMalloc(&ab, 128 * 16); hipMalloc(&d, 512 * 256 * 16); hipMalloc(&dc, 16);
    hipMemset(src, 1, bytes);
    std::vector<unsigned char> h(128 * 16);
    srand(1);
    for (int i = 0; i < 64 * 16; ++i) h[i] = (rand() % 100) < 8;
    for (int i = 64 * 16; i < 128 * 16; ++i) h[i] = (unsigned char)rand();
    hipMemcpy(ab, h.data(), 128 * 16, hipMemcpyHostToDevice);
#define RUN(nd, nl, nv, form, wps, name) run<nd, nl, nv, form, wps>(name, src, bytes, ab, d, dc)
    RUN(0, 0, 0, 0, 1, "bare MFMAs");
    RUN(0, 0, 32, 0, 1, "+ 32 shift/and pairs (64 x 64 tile)");
    RUN(0, 0, 64, 0, 1, "+ 64 shift/and pairs (64 x 32 tile)");
    RUN(0, 24, 0, 0, 1, "+ 24 ds_read_b128");
    RUN(3, 0, 0, 0, 1, "+ 3 DMA pieces (SGPR base)");
    RUN(7, 0, 0, 0, 1, "+ 7 DMA pieces (SGPR base)");
    RUN(7, 0, 0, 1, 1, "+ 7 DMA pieces (64-bit lane address)");
    RUN(14, 0, 0, 0, 1, "+ 14 DMA pieces (SGPR base)");
    RUN(7, 24, 32, 0, 1, "+ 7 DMA + 24 ds_read + 32 pairs (64 x 64 tile)");
    run<3, 0, 0, 0, 1, 0>("+ 3 DMA pieces, source in the Infinity Cache", src, bytes, ab, d, dc);
    run<7, 0, 0, 0, 1, 0>("+ 7 DMA pieces, source in the Infinity Cache", src, bytes, ab, d, dc);
    run<14, 0, 0, 0, 1, 0>("+ 14 DMA pieces, source in the Infinity Cache", src, bytes, ab, d, dc);
    return 0;
}
