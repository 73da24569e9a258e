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

#include <utility>

namespace {

// scheduling hint: after each of N MFMAs let V VALU instructions through (the immediates must be constant expressions)
template <int N, int V, int... I>
__device__ __forceinline__ void interleave_mfma_valu(std::integer_sequence<int, I...>) {
    ((void)I, ..., (void)0);
    ((__builtin_amdgcn_sched_group_barrier(0x008, 1, 0), __builtin_amdgcn_sched_group_barrier(0x002, V, 0), (void)I), ...);
}

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

template <int KP, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void mae_kernel(const uint32_t* __restrict__ XTbits, int64_t ldxt, int64_t n_pad,
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
    __shared__ double red[WAVES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int64_t i0 = ((int64_t)blockIdx.x * WAVES + wave) * 64;  // this wave's 64 rows of U
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
    constexpr int PER_WAVE = PIECES / WAVES;
    static_assert(PIECES % WAVES == 0, "stage must split evenly over the waves");
    // per-lane parts of the source addresses are 32-bit element offsets computed once; the stage-dependent part is wave-uniform
    // (SGPR arithmetic) -- 64-bit per-lane address math inside the stage loop made the kernel VALU-bound
    const int d_row = lane / CH, d_chunk = lane % CH;
    unsigned d_off[PER_WAVE];
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int q = wave * PER_WAVE + i;
        const int rq = q % (64 / ROWS_PER_PIECE);
        const int row = rq * ROWS_PER_PIECE + d_row;
        d_off[i] = (unsigned)(row * KP + ((d_chunk ^ (row % CH)) << 3));
    }
    auto issue = [&](int stage, int buf) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int q = wave * PER_WAVE + i;
            const int term = q / (64 / ROWS_PER_PIECE);
            const uint16_t* base = (term ? Vl : Vh) + (int64_t)stage * 64 * KP;  // wave-uniform
            char* dst = smem + buf * STAGE_BYTES + q * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + d_off[i]),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    float acc_abs = 0.f;
    double total = 0.0;
    // X^T row j, the two words that cover rows i0 .. i0 + 63, for the four column tiles of a stage: fetched one stage ahead
    // (a global load inside the tile loop would sit on the critical path of every tile)
    uint2 xw[4], xn[4];
    unsigned x_off[4];  // word offsets inside a stage's 64 rows of X^T (n_pad * ldxt words fit 32 bits by a wide margin)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) x_off[jt] = (unsigned)((16 * jt + c) * ldxt + (i0 >> 5));
    auto load_x = [&](int stage, uint2 (&dst)[4]) {
        const uint32_t* base = XTbits + (int64_t)stage * 64 * ldxt;  // wave-uniform
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) dst[jt] = *reinterpret_cast<const uint2*>(base + x_off[jt]);
    };
    if (s0 < s1) {
        issue(s0, 0);
        load_x(s0, xw);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = s0; s < s1; ++s) {
        const int cur = (s - s0) & 1;
        if (s + 1 < s1) {
            issue(s + 1, cur ^ 1);
            load_x(s + 1, xn);
        }
        const char* buf = smem + cur * STAGE_BYTES;
        // software pipeline over the four 16-column tiles of the stage: the MFMAs of tile jt + 1 are issued interleaved with
        // the element-wise work on tile jt (otherwise the matrix pipe idles through every element-wise phase of the wave)
        auto products = [&](int jt, f32x4 (&p)[4]) {
            const int jrow = 16 * jt + c;  // row of V inside the stage = this lane's column j
            u32x4 bh[KS], bl[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int chunk = ((ks * 4 + g) ^ (jrow % CH)) << 4;
                bh[ks] = *reinterpret_cast<const u32x4*>(buf + jrow * ROWB + chunk);
                bl[ks] = *reinterpret_cast<const u32x4*>(buf + 64 * ROWB + jrow * ROWB + chunk);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) p[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            // the three products of a k-step go round the four row groups: consecutive MFMAs never share an accumulator
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    p[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, al[mt][ks]), __builtin_bit_cast(bf16x8, bh[ks]), p[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    p[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[mt][ks]), __builtin_bit_cast(bf16x8, bl[ks]), p[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    p[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[mt][ks]), __builtin_bit_cast(bf16x8, bh[ks]), p[mt], 0, 0, 0);
            }
        };
        auto reduce = [&](int jt, const f32x4 (&p)[4]) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                // D layout: column = lane & 15 (this lane's j), rows 16 mt + 4 g + reg
                const unsigned w = mt < 2 ? xw[jt].x : xw[jt].y;
                const int b0 = 16 * (mt & 1) + 4 * g;
                // the 4 bits of rows 4g .. 4g+3 -> one byte each (multiply by 0x00204081 puts bit q at bit 8q), then the
                // byte -> float converts: 6 ops per 4 cells instead of 8
                const unsigned spread = (((w >> b0) & 0xFu) * 0x00204081u) & 0x01010101u;
                acc_abs += fabsf((float)(spread & 0xFFu) - p[mt][0]);
                acc_abs += fabsf((float)((spread >> 8) & 0xFFu) - p[mt][1]);
                acc_abs += fabsf((float)((spread >> 16) & 0xFFu) - p[mt][2]);
                acc_abs += fabsf((float)(spread >> 24) - p[mt][3]);
            }
        };
        f32x4 pa[4], pb[4];
        products(0, pa);
        __builtin_amdgcn_sched_barrier(0);
        products(1, pb);
        reduce(0, pa);
        interleave_mfma_valu<12 * KS, 4>(std::make_integer_sequence<int, 12 * KS>{});
        __builtin_amdgcn_sched_barrier(0);
        products(2, pa);
        reduce(1, pb);
        interleave_mfma_valu<12 * KS, 4>(std::make_integer_sequence<int, 12 * KS>{});
        __builtin_amdgcn_sched_barrier(0);
        products(3, pb);
        reduce(2, pa);
        interleave_mfma_valu<12 * KS, 4>(std::make_integer_sequence<int, 12 * KS>{});
        __builtin_amdgcn_sched_barrier(0);
        reduce(3, pb);
        total += (double)acc_abs;  // keep the fp32 partial short: one stage = 64 cells per lane
        acc_abs = 0.f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) xw[jt] = xn[jt];
    }
    total = wave_sum(total);
    if (lane == 0) red[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) t += red[w];
        atomicAdd(sum, t);
    }
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
    // 4 waves = 256 rows of U per workgroup (8 waves / 512 rows halve the V traffic through L2 but leave one workgroup per
    // CU: measured 886 vs 686 us)
    const int row_blocks = (int)(m_pad / 256);
    const int stages = (int)(n_pad / 64);
    int groups = (1024 + row_blocks - 1) / row_blocks;  // ~4 workgroups per CU in total
    if (groups > stages) groups = stages;
    const int per = (stages + groups - 1) / groups;
    groups = (stages + per - 1) / per;
    dim3 grid((unsigned)row_blocks, (unsigned)groups), block(256);
    if (kp == 32) BMF_LAUNCH((mae_kernel<32, 4>), grid, block, 0, s, XTbits, ldxt, n_pad, Uh, Ul, Vh, Vl, per, sum, stop);
    else BMF_LAUNCH((mae_kernel<64, 4>), grid, block, 0, s, XTbits, ldxt, n_pad, Uh, Ul, Vh, Vl, per, sum, stop);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_mae_sum(const uint32_t* XTbits, int64_t ldxt, int64_t m_pad, int64_t n_pad, const float* U, const float* V,
                           int kp, uint16_t* ws, double* sum, void* stream) {
    return bmf_mae_launch(XTbits, ldxt, m_pad, n_pad, U, V, kp, ws, sum, nullptr, (hipStream_t)stream);
}
