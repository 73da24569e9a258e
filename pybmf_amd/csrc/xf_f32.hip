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
                                                      float* __restrict__ out, int64_t slab_stride, int n_row_tiles) {
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

}  // namespace

extern "C" int bmf_xf_f32(const float* A, int64_t rows_pad, int64_t lda, int64_t red, const float* FT, int64_t ldft, int kp,
                          float* out, int64_t slab_stride, int splits, void* stream) {
    BMF_REQUIRE(A && FT && out, "bmf_xf_f32: null pointer");
    BMF_REQUIRE(rows_pad > 0 && rows_pad % 128 == 0, "bmf_xf_f32: rows_pad must be a positive multiple of 128");
    BMF_REQUIRE(red > 0 && red % 8 == 0, "bmf_xf_f32: red=%lld must be a positive multiple of 8", (long long)red);
    BMF_REQUIRE(lda >= red && lda % 4 == 0 && ldft >= red && ldft % 4 == 0, "bmf_xf_f32: lda/ldft must be >= red and multiples of 4");
    BMF_REQUIRE(kp == 32 || kp == 64, "bmf_xf_f32: kp must be 32 or 64");
    BMF_REQUIRE(splits >= 1 && splits <= red / 8, "bmf_xf_f32: splits out of range");
    BMF_REQUIRE(slab_stride >= rows_pad * kp, "bmf_xf_f32: slab_stride too small");
    BMF_REQUIRE(bmf_aligned16(A) && bmf_aligned16(FT), "bmf_xf_f32: pointers must be 16-byte aligned");
    const int groups = (int)(red / 8);
    const int gps = (groups + splits - 1) / splits;
    const int n_row_tiles = (int)(rows_pad / 128);
    dim3 grid((unsigned)(n_row_tiles * splits)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (kp == 32)
        BMF_LAUNCH(xf_f32_kernel<1>, grid, block, 0, s, A, lda, groups, gps, FT, ldft, out, slab_stride, n_row_tiles);
    else
        BMF_LAUNCH(xf_f32_kernel<2>, grid, block, 0, s, A, lda, groups, gps, FT, ldft, out, slab_stride, n_row_tiles);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}
