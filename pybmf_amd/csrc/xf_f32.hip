// K1/K2 for a real-valued X (WNMF on non-Boolean data): out = A . F with exact-fp32 MFMA.
//
//   X  @ V   replaces  multiply(W, X) @ V      PyBMF/models/WNMF.py:105
//   X^T @ U  replaces  multiply(W, X).T @ U    PyBMF/models/WNMF.py:98      (A = a transposed copy of X)
//
// v_mfma_f32_32x32x2_f32 takes ONE fp32 per lane per operand (lane (r, h): A[r][k=h], B[k=h][c]).  The reduction
// order inside a group of 8 indices is free as long as both operands agree, so every lane loads 4 consecutive
// floats (16 B) of its row -- half h takes floats 4h..4h+3 of the group -- and MFMA step t multiplies element t of
// both halves.  A streams from HBM (it is read exactly once), the transposed factor FT[j][c] is L2-resident and is
// read straight into registers; no LDS.  The reduction is split over `splits` workgroups per row tile (slabs).
#include "common.h"

namespace {

template <int NT>
__global__ __launch_bounds__(256) void xf_f32_kernel(const float* __restrict__ A, int64_t lda, int groups_total,
                                                      int groups_per_split, const float* __restrict__ FT, int64_t ldft,
                                                      float* __restrict__ out, int64_t slab_stride, int n_row_tiles,
                                                      const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    constexpr int NC = 32 * NT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int split = blockIdx.x / n_row_tiles;
    const int tile = blockIdx.x - split * n_row_tiles;
    const int g0 = split * groups_per_split;
    const int g1 = min(g0 + groups_per_split, groups_total);
    const int64_t row_base = (int64_t)tile * 128 + wave * 32;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;

    const float* ap = A + (row_base + r) * lda + 4 * h;
    const float* bp = FT + (int64_t)r * ldft + 4 * h;

    int g = g0;
    for (; g + 4 <= g1; g += 4) {  // 4 groups of 8 reduction indices in flight
        f32x4 a[4], b[4][NT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const f32x4*>(ap + 8 * (int64_t)(g + u));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                b[u][nt] = *reinterpret_cast<const f32x4*>(bp + (int64_t)(32 * nt) * ldft + 8 * (int64_t)(g + u));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t], b[u][nt][t], acc[nt], 0, 0, 0);
    }
    for (; g < g1; ++g) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(ap + 8 * (int64_t)g);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(bp + (int64_t)(32 * nt) * ldft + 8 * (int64_t)g);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc[nt], 0, 0, 0);
        }
    }
    float* o = out + (int64_t)split * slab_stride;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t row = row_base + (i & 3) + 8 * (i >> 2) + 4 * h;
            o[row * NC + 32 * nt + r] = acc[nt][i];
        }
}

// LDS-staged flavour (reduction length a multiple of 64): the direct-to-register kernel above makes every wave walk 32
// rows 128 bytes at a time, so with ~4000 waves in flight DRAM sees ~140k interleaved streams of single cache lines and
// delivers 2 TB/s.  Here a workgroup brings a stage of 128 rows x 64 floats into LDS by LDS-DMA, 256 contiguous bytes of a
// row per quarter-wave, double buffered (64 KiB: two workgroups per CU), and the waves read it back in MFMA order.  The
// 16-byte chunks of a row are stored XOR-swizzled with the row number (applied on the DMA source address, the LDS image must
// stay lane-linear) so that the rows a ds_read_b128 touches fall on distinct banks.  The factor operand comes from L2.
template <int NT>
__global__ __launch_bounds__(256) void xf_f32_lds_kernel(const float* __restrict__ A, int64_t lda, int stages_total,
                                                          int stages_per_split, const float* __restrict__ FT, int64_t ldft,
                                                          float* __restrict__ out, int64_t slab_stride, int n_row_tiles,
                                                          const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    constexpr int NC = 32 * NT;
    constexpr int SF = 64;                    // floats of a row per stage
    constexpr int STAGE_BYTES = 128 * SF * 4;  // 32 KiB
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int split = blockIdx.x / n_row_tiles;
    const int tile = blockIdx.x - split * n_row_tiles;
    const int s0 = split * stages_per_split;
    const int s1 = min(s0 + stages_per_split, stages_total);
    const int64_t tile_row = (int64_t)tile * 128;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;

    // DMA piece q (1 KiB) = rows 4q .. 4q+3 of the stage; lane l: row 4q + (l >> 4), LDS chunk l & 15 <- source chunk (l & 15) ^ (row & 15)
    const int d_row = lane >> 4, d_chunk = lane & 15;
    auto issue = [&](int stage, int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = wave * 8 + i;
            const int row = 4 * q + d_row;
            const float* src = A + (tile_row + row) * lda + (int64_t)stage * SF + ((d_chunk ^ (row & 15)) << 2);
            char* dst = smem + buf * STAGE_BYTES + q * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    const float* bp = FT + (int64_t)r * ldft + 4 * h;
    const int my_row = wave * 32 + r;  // row of the stage this lane feeds to the MFMA

    if (s0 < s1) issue(s0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = s0; s < s1; ++s) {
        const int cur = (s - s0) & 1;
        if (s + 1 < s1) issue(s + 1, cur ^ 1);
        const char* buf = smem + cur * STAGE_BYTES + my_row * (SF * 4);
#pragma unroll
        for (int u0 = 0; u0 < SF / 8; u0 += 4) {  // 4 groups of 8 reduction indices at a time
            f32x4 a[4], b[4][NT];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = *reinterpret_cast<const f32x4*>(buf + (((2 * (u0 + u) + h) ^ (r & 15)) << 4));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    b[u][nt] = *reinterpret_cast<const f32x4*>(bp + (int64_t)(32 * nt) * ldft + (int64_t)s * SF + 8 * (u0 + u));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t], b[u][nt][t], acc[nt], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    float* o = out + (int64_t)split * slab_stride;
    const int64_t row_base = tile_row + wave * 32;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t row = row_base + (i & 3) + 8 * (i >> 2) + 4 * h;
            o[row * NC + 32 * nt + r] = acc[nt][i];
        }
}

}  // namespace

int bmf_xf_f32_launch(const float* A, int64_t rows_pad, int64_t lda, int64_t red, const float* FT, int64_t ldft, int kp, float* out,
                      int64_t slab_stride, int splits, const int32_t* stop, hipStream_t s) {
    BMF_REQUIRE(A && FT && out, "bmf_xf_f32: null pointer");
    BMF_REQUIRE(rows_pad > 0 && rows_pad % 128 == 0, "bmf_xf_f32: rows_pad must be a positive multiple of 128");
    BMF_REQUIRE(red > 0 && red % 8 == 0, "bmf_xf_f32: red=%lld must be a positive multiple of 8", (long long)red);
    BMF_REQUIRE(lda >= red && lda % 4 == 0 && ldft >= red && ldft % 4 == 0, "bmf_xf_f32: lda/ldft must be >= red and multiples of 4");
    BMF_REQUIRE(kp == 32 || kp == 64, "bmf_xf_f32: kp must be 32 or 64");
    BMF_REQUIRE(splits >= 1 && splits <= red / 8, "bmf_xf_f32: splits out of range");
    BMF_REQUIRE(slab_stride >= rows_pad * kp, "bmf_xf_f32: slab_stride too small");
    BMF_REQUIRE(bmf_aligned16(A) && bmf_aligned16(FT), "bmf_xf_f32: pointers must be 16-byte aligned");
    const int n_row_tiles = (int)(rows_pad / 128);
    dim3 grid((unsigned)(n_row_tiles * splits)), block(256);
    if (red % 64 == 0 && bmf_aligned16(out)) {
        const int stages = (int)(red / 64);
        const int sps = (stages + splits - 1) / splits;
        if (kp == 32)
            BMF_LAUNCH(xf_f32_lds_kernel<1>, grid, block, 0, s, A, lda, stages, sps, FT, ldft, out, slab_stride, n_row_tiles, stop);
        else
            BMF_LAUNCH(xf_f32_lds_kernel<2>, grid, block, 0, s, A, lda, stages, sps, FT, ldft, out, slab_stride, n_row_tiles, stop);
        BMF_LAUNCH_CHECK();
        return BMF_OK;
    }
    const int groups = (int)(red / 8);
    const int gps = (groups + splits - 1) / splits;
    if (kp == 32)
        BMF_LAUNCH(xf_f32_kernel<1>, grid, block, 0, s, A, lda, groups, gps, FT, ldft, out, slab_stride, n_row_tiles, stop);
    else
        BMF_LAUNCH(xf_f32_kernel<2>, grid, block, 0, s, A, lda, groups, gps, FT, ldft, out, slab_stride, n_row_tiles, stop);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_xf_f32(const float* A, int64_t rows_pad, int64_t lda, int64_t red, const float* FT, int64_t ldft, int kp,
                          float* out, int64_t slab_stride, int splits, void* stream) {
    return bmf_xf_f32_launch(A, rows_pad, lda, red, FT, ldft, kp, out, slab_stride, splits, nullptr, (hipStream_t)stream);
}
