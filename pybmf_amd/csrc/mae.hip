// K5'': sum |X - U V^T| over the cells of a Boolean X -- the MAE column of evaluate(..., metrics=['RMSE', 'MAE'])
// (PyBMF/utils/metrics.py:156-160, called every iteration by BinaryMFPenalty.py:71,97 / WNMF.py:60,83) -- on the bf16 MFMA.
//
// The exact-fp32 residual pass (residual.hip) costs 2 m n k flop at the fp32-MFMA rate: 1.85 ms at 100k x 20k, k = 64,
// twice the rest of the iteration.  MAE is a sum of 2e9 absolute values, so the product only has to be right to ~1e-5 per
// cell and unbiased: both factors are split into two bf16 addends (row-major copies made by split_rows_kernel) and
// P = Uh Vh^T + Uh Vl^T + Ul Vh^T accumulates in fp32 (16 significant bits per operand, the dropped Ul Vl^T term is 2^-16
// of P), at 16x the fp32-MFMA rate.
//
// Tiling: a wave keeps the A fragments of 64 rows of U (K = kp, both addends) in registers; the workgroup's four waves (256
// rows) share stages of 64 rows of V staged in LDS by LDS-DMA (double buffered; the 16-byte k-groups of a row are XOR-swizzled
// with the row number on the DMA source address).  Per 64 x 16 output tile: 3 x (kp / 32) x 4 MFMAs, then 16 cells per lane:
// bit extract, convert, |x - p|, add.  X is read in the transposed orientation: the lane's column j is one row of X^T and
// the wave's 64 rows are two words of it.  Padded rows / columns are zero in both factors and in X, so they add nothing.
#include "common.h"

namespace {

// F (rows_pad x kp fp32) -> H, L (rows_pad x kp bf16 each): F = H + L + O(2^-16 F)
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ F, int64_t total, uint16_t* __restrict__ H,
                                                          uint16_t* __restrict__ Lo, const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (int64_t)gridDim.x * 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(F + i);
        uint16_t h[4], l[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            h[q] = bf16_bits(v[q]);
            l[q] = bf16_bits(v[q] - bf16_to_f32(h[q]));
        }
        *reinterpret_cast<uint2*>(H + i) = uint2{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
        *reinterpret_cast<uint2*>(Lo + i) = uint2{(unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16)};
    }
}

template <int KP>
__global__ __launch_bounds__(256) void mae_kernel(const uint32_t* __restrict__ XTbits, int64_t ldxt, int64_t n_pad,
                                                   const uint16_t* __restrict__ Uh, const uint16_t* __restrict__ Ul,
                                                   const uint16_t* __restrict__ Vh, const uint16_t* __restrict__ Vl,
                                                   int stages_per_block, double* __restrict__ sum,
                                                   const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    constexpr int KS = KP / 32;            // k-steps of 32
    constexpr int ROWB = KP * 2;           // bytes of one row of one addend
    constexpr int CH = ROWB / 16;          // 16-byte k-groups per row (4 or 8)
    constexpr int STAGE_BYTES = 2 * 64 * ROWB;  // [addend][64 rows of V][ROWB]
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
    __shared__ double red[4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * 64;  // this wave's 64 rows of U
    const int total_stages = (int)(n_pad / 64);
    const int s0 = blockIdx.y * stages_per_block;
    const int s1 = min(s0 + stages_per_block, total_stages);

    // A fragments: lane (c, g) holds U[i0 + 16 mt + c][32 ks + 8 g .. + 7] of both addends
    u32x4 ah[4][KS], al[4][KS];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int64_t off = (i0 + 16 * mt + c) * KP + 32 * ks + 8 * g;
            ah[mt][ks] = *reinterpret_cast<const u32x4*>(Uh + off);
            al[mt][ks] = *reinterpret_cast<const u32x4*>(Ul + off);
        }

    // DMA: a stage = 64 rows of V x 2 addends; piece q (1 KiB) = 1024 / ROWB rows of one addend; lane l: row (l / CH), LDS chunk
    // l % CH <- source chunk (l % CH) ^ (row % CH)
    constexpr int ROWS_PER_PIECE = 1024 / ROWB;          // 8 (kp = 64) or 16 (kp = 32)
    constexpr int PIECES = 2 * 64 / ROWS_PER_PIECE;      // per stage
    constexpr int PER_WAVE = PIECES / 4;
    const int d_row = lane / CH, d_chunk = lane % CH;
    auto issue = [&](int stage, int buf) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int q = wave * PER_WAVE + i;
            const int term = q / (64 / ROWS_PER_PIECE), rq = q - term * (64 / ROWS_PER_PIECE);
            const int row = rq * ROWS_PER_PIECE + d_row;
            const uint16_t* src = (term ? Vl : Vh) + ((int64_t)stage * 64 + row) * KP + ((d_chunk ^ (row % CH)) << 3);
            char* dst = smem + buf * STAGE_BYTES + q * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    float acc_abs = 0.f;
    double total = 0.0;
    if (s0 < s1) issue(s0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = s0; s < s1; ++s) {
        const int cur = (s - s0) & 1;
        if (s + 1 < s1) issue(s + 1, cur ^ 1);
        const char* buf = smem + cur * STAGE_BYTES;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {  // 16 columns j at a time
            const int jrow = 16 * jt + c;  // row of V inside the stage = this lane's column j
            // X^T row j, the two words that cover rows i0 .. i0 + 63
            const uint2 xw = *reinterpret_cast<const uint2*>(XTbits + ((int64_t)s * 64 + jrow) * ldxt + (i0 >> 5));
            u32x4 bh[KS], bl[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int chunk = ((ks * 4 + g) ^ (jrow % CH)) << 4;
                bh[ks] = *reinterpret_cast<const u32x4*>(buf + jrow * ROWB + chunk);
                bl[ks] = *reinterpret_cast<const u32x4*>(buf + 64 * ROWB + jrow * ROWB + chunk);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                f32x4 p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    p = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, al[mt][ks]), __builtin_bit_cast(bf16x8, bh[ks]), p, 0, 0, 0);
                    p = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[mt][ks]), __builtin_bit_cast(bf16x8, bl[ks]), p, 0, 0, 0);
                    p = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[mt][ks]), __builtin_bit_cast(bf16x8, bh[ks]), p, 0, 0, 0);
                }
                // D layout: column = lane & 15 (this lane's j), rows 16 mt + 4 g + reg
                const unsigned w = mt < 2 ? xw.x : xw.y;
                const int b0 = 16 * (mt & 1) + 4 * g;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float x = (float)((w >> (b0 + q)) & 1u);
                    acc_abs += fabsf(x - p[q]);
                }
            }
        }
        total += (double)acc_abs;  // keep the fp32 partial short: one stage = 1024 cells per lane
        acc_abs = 0.f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    total = wave_sum(total);
    if (lane == 0) red[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sum, ((red[0] + red[1]) + red[2]) + red[3]);
}

}  // namespace

int bmf_mae_launch(const uint32_t* XTbits, int64_t ldxt, int64_t m_pad, int64_t n_pad, const float* U, const float* V, int kp,
                   uint16_t* ws, double* sum, const int32_t* stop, hipStream_t s) {
    BMF_REQUIRE(XTbits && U && V && ws && sum, "bmf_mae_sum: null pointer");
    BMF_REQUIRE(m_pad > 0 && m_pad % 256 == 0 && n_pad > 0 && n_pad % 64 == 0, "bmf_mae_sum: m_pad must be a multiple of 256, n_pad of 64");
    BMF_REQUIRE(ldxt * 32 >= m_pad && ldxt % 2 == 0, "bmf_mae_sum: ldxt must be even and cover m_pad");
    BMF_REQUIRE(kp == 32 || kp == 64, "bmf_mae_sum: kp must be 32 or 64");
    BMF_REQUIRE(bmf_aligned16(U) && bmf_aligned16(V) && bmf_aligned16(ws) && (((uintptr_t)XTbits) & 7u) == 0, "bmf_mae_sum: alignment");
    uint16_t* Uh = ws;
    uint16_t* Ul = Uh + m_pad * kp;
    uint16_t* Vh = Ul + m_pad * kp;
    uint16_t* Vl = Vh + n_pad * kp;
    const int64_t tu = m_pad * kp, tv = n_pad * kp;
    auto blocks = [](int64_t total) { const int64_t b = (total / 4 + 255) / 256; return (unsigned)(b < 2048 ? b : 2048); };
    BMF_LAUNCH(split_rows_kernel, dim3(blocks(tu)), dim3(256), 0, s, U, tu, Uh, Ul, stop);
    BMF_LAUNCH(split_rows_kernel, dim3(blocks(tv)), dim3(256), 0, s, V, tv, Vh, Vl, stop);
    const int row_blocks = (int)(m_pad / 256);
    const int stages = (int)(n_pad / 64);
    int groups = (1024 + row_blocks - 1) / row_blocks;  // ~4 workgroups per CU in total
    if (groups > stages) groups = stages;
    const int per = (stages + groups - 1) / groups;
    groups = (stages + per - 1) / per;
    dim3 grid((unsigned)row_blocks, (unsigned)groups), block(256);
    if (kp == 32) BMF_LAUNCH(mae_kernel<32>, grid, block, 0, s, XTbits, ldxt, n_pad, Uh, Ul, Vh, Vl, per, sum, stop);
    else BMF_LAUNCH(mae_kernel<64>, grid, block, 0, s, XTbits, ldxt, n_pad, Uh, Ul, Vh, Vl, per, sum, stop);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_mae_sum(const uint32_t* XTbits, int64_t ldxt, int64_t m_pad, int64_t n_pad, const float* U, const float* V,
                           int kp, uint16_t* ws, double* sum, void* stream) {
    return bmf_mae_launch(XTbits, ldxt, m_pad, n_pad, U, V, kp, ws, sum, nullptr, (hipStream_t)stream);
}
