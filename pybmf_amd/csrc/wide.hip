// Rank 64 < k <= 128: the pieces that couple the two 64-column BLOCKS of a factor.
//
// Every kernel of the multiplicative-update path is written for a padded rank of 32 or 64 (one 64-bit word of factor bits per row,
// two 32-column MFMA tiles).  A wider factor is held as two blocks F = [F_0 | F_1] of 64 columns each; the per-column work -- the
// bits GEMMs X V_b and X^T U_b, the digit planes, the fp64 element-wise update -- runs per block through the kernels that exist,
// and only what mixes the blocks is new (reference: PyBMF/models/BinaryMFPenalty.py:136-163 has no rank limit):
//   * bmf_fg_f32:          den_b = sum_b' F_b' G[b'][b], the re-associated denominator  multiply(W, U V^T) V = U (V^T V)
//                          (:142,157; WNMF.py:99,106) -- one 64 x 64 block of the Gram matrix per call, accumulating;
//   * bmf_gram_cross:      G[a][b] = F_a^T F_b (partial sums per row range, summed by bmf_reduce_slabs);
//   * bmf_cover_count_wide: TP / FP of the Boolean product over all 128 factors (utils/common.py:110-151, metrics.py:56-68);
//   * bmf_resid_sums_wide: sum |X - U V^T| and sum (X - U V^T)^2 with the product over all 128 columns (metrics.py:149-160) on
//                          the fp16 MFMA kernel of mae.hip instantiated at K = 128.
// These are correctness rows (SURVEY 8f has no configuration with k > 64): simple tilings, no tuning.
#include "common.h"

int bmf_mae_wide_launch(const uint32_t* XT, int64_t ldxt, int64_t m_pad, int64_t n_pad, const float* UA, const float* UB, const float* VA,
                        const float* VB, uint16_t* ws, double* sums, int x_tiled, hipStream_t s);

namespace {

// out[rows_pad][64] (+)= F[rows_pad][64] . G[64][ldg]: one block = 128 rows = 4 waves x 32 rows, exact-fp32 MFMA (the F G product
// of mu_epilogue_kernel, epilogue.hip, as a kernel of its own)
__global__ __launch_bounds__(256) void fg_f32_kernel(const float* __restrict__ F, const float* __restrict__ G, int ldg, float* __restrict__ out,
                                                      int accumulate) {
    constexpr int KP = 64, KH = 32, NT = 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 32;
    float av[KH], gv[NT][KH];
    const float* ap = F + (row0 + c) * KP + KH * h;
#pragma unroll
    for (int s = 0; s < KH; s += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(ap + s);
        av[s] = v[0]; av[s + 1] = v[1]; av[s + 2] = v[2]; av[s + 3] = v[3];
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < KH; ++s) gv[nt][s] = G[(KH * h + s) * ldg + 32 * nt + c];
    f32x16 fg[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) fg[nt][i] = 0.f;
#pragma unroll
    for (int s = 0; s < KH; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) fg[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], gv[nt][s], fg[nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t idx = (row0 + (i & 3) + 8 * (i >> 2) + 4 * h) * KP + 32 * nt + c;
            out[idx] = accumulate ? out[idx] + fg[nt][i] : fg[nt][i];
        }
}

// slabs[block][64][64] = partial A^T B over the block's row range (gram_partial_kernel of util.hip with two operands)
__global__ __launch_bounds__(256) void gram_cross_kernel(const float* __restrict__ A, const float* __restrict__ B, int64_t rows_pad,
                                                          float* __restrict__ slabs, int gram_blocks) {
    constexpr int NT = 2, KP = 64;
    __shared__ float sh[4][KP * KP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int64_t nwaves = (int64_t)gram_blocks * 4;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
    const int64_t pairs = rows_pad / 2;
    const int64_t per = (pairs + nwaves - 1) / nwaves;
    const int64_t p0 = gw * per, p1 = min(p0 + per, pairs);
    f32x16 acc[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    for (int64_t p = p0; p < p1; ++p) {
        float va[NT], vb[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            va[t] = A[(2 * p + h) * KP + 32 * t + c];
            vb[t] = B[(2 * p + h) * KP + 32 * t + c];
        }
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[a], vb[b], acc[a][b], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) sh[wave][(32 * a + (i & 3) + 8 * (i >> 2) + 4 * h) * KP + 32 * b + c] = acc[a][b][i];
    __syncthreads();
    float* o = slabs + (int64_t)blockIdx.x * KP * KP;
    for (int i = threadIdx.x; i < KP * KP; i += 256) o[i] = (sh[0][i] + sh[1][i]) + (sh[2][i] + sh[3][i]);
}

// Cover count over 128 factors: a block keeps a chunk of CW words of all 128 bit-columns in LDS (64 KiB); a wave takes one row of X
// at a time, walks the set bits of the row's two 64-bit words (wave-uniform) and ORs the selected bit-columns, two words per lane.
constexpr int CW = 128;
__global__ __launch_bounds__(256) void cover_wide_kernel(const uint32_t* __restrict__ X, int64_t ldx, int64_t words, int64_t rows_pad,
                                                          const uint64_t* __restrict__ rbA, const uint64_t* __restrict__ rbB,
                                                          const uint32_t* __restrict__ cbA, const uint32_t* __restrict__ cbB, int64_t ldcb,
                                                          int rows_per_block, unsigned long long* __restrict__ counts) {
    __shared__ __attribute__((aligned(16))) uint32_t vt[128][CW];
    __shared__ unsigned red[2][4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t w0 = (int64_t)blockIdx.x * CW;
    const int nw = (int)min((int64_t)CW, words - w0);
    for (int p = threadIdx.x; p < 128 * (CW / 4); p += 256) {
        const int l = p / (CW / 4), pw = (p % (CW / 4)) * 4;
        const uint32_t* src = l < 64 ? cbA + (int64_t)l * ldcb : cbB + (int64_t)(l - 64) * ldcb;
        *reinterpret_cast<u32x4*>(&vt[l][pw]) = pw < nw ? *reinterpret_cast<const u32x4*>(src + w0 + pw) : u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, rows_pad);
    const bool lane_on = 2 * lane < nw;   // words is a multiple of 4
    unsigned tp = 0, fp = 0;
    for (int64_t r = r0 + wave; r < r1; r += 4) {
        const unsigned long long ua64 = rbA[r], ub64 = rbB[r];
        unsigned u[4] = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ua64), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ua64 >> 32)),
                         (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ub64), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ub64 >> 32))};
        if ((u[0] | u[1] | u[2] | u[3]) == 0u) continue;
        const uint2 x = lane_on ? *reinterpret_cast<const uint2*>(X + r * ldx + w0 + 2 * lane) : uint2{0u, 0u};
        uint2 pd = {0u, 0u};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned w = u[q];
            while (w) {
                const int l = 32 * q + __builtin_ctz(w);
                w &= w - 1;
                const uint2 v = *reinterpret_cast<const uint2*>(&vt[l][2 * lane]);
                pd.x |= v.x;
                pd.y |= v.y;
            }
        }
        if (!lane_on) pd = uint2{0u, 0u};
        tp += __popc(x.x & pd.x) + __popc(x.y & pd.y);
        fp += __popc(~x.x & pd.x) + __popc(~x.y & pd.y);
    }
    tp = wave_sum(tp);
    fp = wave_sum(fp);
    if (lane == 0) { red[0][wave] = tp; red[1][wave] = fp; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const unsigned long long t = (unsigned long long)red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        if (t) atomicAdd(&counts[threadIdx.x], t);
    }
}

}  // namespace

extern "C" int bmf_fg_f32(const float* F, int64_t rows_pad, const float* G, int ldg, float* out, int accumulate, void* stream) {
    BMF_REQUIRE(F && G && out, "bmf_fg_f32: null pointer");
    BMF_REQUIRE(rows_pad > 0 && rows_pad % 128 == 0 && ldg >= 64, "bmf_fg_f32: rows_pad must be a positive multiple of 128, ldg >= 64");
    BMF_REQUIRE(bmf_aligned16(F), "bmf_fg_f32: F must be 16-byte aligned");
    BMF_LAUNCH(fg_f32_kernel, dim3((unsigned)(rows_pad / 128)), dim3(256), 0, (hipStream_t)stream, F, G, ldg, out, accumulate);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_gram_cross(const float* A, const float* B, int64_t rows_pad, float* slabs, int blocks, void* stream) {
    BMF_REQUIRE(A && B && slabs, "bmf_gram_cross: null pointer");
    BMF_REQUIRE(rows_pad > 0 && rows_pad % 2 == 0 && blocks >= 1 && blocks <= 1024, "bmf_gram_cross: rows_pad must be even, blocks 1..1024");
    BMF_LAUNCH(gram_cross_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, A, B, rows_pad, slabs, blocks);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_cover_count_wide(const uint32_t* Xbits, int64_t rows_pad, int64_t ldx, int64_t words, const uint64_t* rowbitsA,
                                    const uint64_t* rowbitsB, const uint32_t* colbitsA, const uint32_t* colbitsB, int64_t ldcb,
                                    unsigned long long* counts, void* stream) {
    BMF_REQUIRE(Xbits && rowbitsA && rowbitsB && colbitsA && colbitsB && counts, "bmf_cover_count_wide: null pointer");
    BMF_REQUIRE(rows_pad > 0 && words > 0 && words % 4 == 0 && ldx >= words && ldx % 4 == 0 && ldcb >= words && ldcb % 4 == 0,
                "bmf_cover_count_wide: words / ldx / ldcb must be multiples of 4, ldx and ldcb >= words");
    BMF_REQUIRE(bmf_aligned16(Xbits) && bmf_aligned16(colbitsA) && bmf_aligned16(colbitsB), "bmf_cover_count_wide: pointers must be 16-byte aligned");
    const unsigned chunks = (unsigned)((words + CW - 1) / CW);
    int64_t groups = 1024 / chunks > 0 ? 1024 / chunks : 1;
    const int64_t units = (rows_pad + 3) / 4;
    if (groups > units) groups = units;
    const int rows_per_block = (int)(((units + groups - 1) / groups) * 4);
    groups = (rows_pad + rows_per_block - 1) / rows_per_block;
    BMF_LAUNCH(cover_wide_kernel, dim3(chunks, (unsigned)groups), dim3(256), 0, (hipStream_t)stream, Xbits, ldx, words, rows_pad, rowbitsA, rowbitsB,
               colbitsA, colbitsB, ldcb, rows_per_block, counts);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_resid_sums_wide(const uint32_t* XTbits, int64_t ldxt, int64_t m_pad, int64_t n_pad, const float* UA, const float* UB,
                                   const float* VA, const float* VB, uint16_t* ws, double* sums, int x_tiled, void* stream) {
    return bmf_mae_wide_launch(XTbits, ldxt, m_pad, n_pad, UA, UB, VA, VB, ws, sums, x_tiled, (hipStream_t)stream);
}
