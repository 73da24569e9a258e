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
// rows) share stages of 64 rows of V and the matching 64 x 256-bit tile of X^T, brought into a ring of three LDS slots by
// LDS-DMA two stages ahead (the 16-byte k-groups of a row of V are XOR-swizzled with the row number on the DMA source
// address); one bare s_barrier per stage.  Per 64 x 16 output tile: 3 x (kp / 32) x 4 MFMAs, then 16 cells per lane: bit
// extract, packed fp8 -> f32 convert, |x - p|, add.  The MFMAs of a tile are interleaved with the element-wise work on the
// tile before it while the LDS reads for the tile after it are in flight (hand-placed ds_read / s_waitcnt).  X is read in the
// transposed orientation: the lane's column j is one row of X^T and the wave's 64 rows are two words of it.  Padded rows /
// columns are zero in both factors and in X, so they add nothing.  Workgroup ids are mapped XCD-aware (see the kernel).
//
// Where the time goes at 100k x 20k, k = 64 (0.65 ms; MFMA pipe alone 0.36 ms): MFMA stream 0.40, + LDS reads 0.09, + the two
// DMA streams 0.09, + element-wise 0.08 -- measured by leaving each out (BMF_EXP_MAE_* macros, timing only).
#include "common.h"

#include <utility>

#ifndef BMF_MAE_V
#define BMF_MAE_V 2
#endif

namespace {

// scheduling hint: after each of N MFMAs let V VALU instructions through (the immediates must be constant expressions)
template <int N, int V, int... I>
__device__ __forceinline__ void interleave_mfma_valu(std::integer_sequence<int, I...>) {
    ((void)I, ..., (void)0);
    ((__builtin_amdgcn_sched_group_barrier(0x008, 1, 0), __builtin_amdgcn_sched_group_barrier(0x002, V, 0), (void)I), ...);
}

typedef __attribute__((ext_vector_type(2))) float f32x2;

// the cells of X enter the accumulators as 0 / X_ONE (a byte 0x40 = 2.0 as OCP fp8 e4m3: ONE bit per cell, so that a rotate and an
// AND turn four bits of a word of X^T into four fp8 bytes), U is scaled by -X_ONE to match: the MFMAs leave X_ONE (x - p)
constexpr float X_ONE = 2.0f;

// scale * F (rows_pad x kp fp32, scale a power of two) -> H, L (rows_pad x kp bf16 each): scale * F = H + L + O(2^-16 F)
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ F, int64_t total, float scale,
                                                          uint16_t* __restrict__ H, uint16_t* __restrict__ Lo,
                                                          const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (int64_t)gridDim.x * 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(F + i) * scale;
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

// scale * F (rows_pad x kp fp32) -> H (fp16, one addend, saturated at the fp16 maximum): the operands of the single-product pass.
// Both factors in ONE launch (blocks [0, blocks0) take F0), and block 0 zeroes the two sums of the log row when asked to: the pass
// used to be preceded by three tiny launches (two conversions, one zeroing kernel), ~5 us of stream time each.
__global__ __launch_bounds__(256) void to_f16_pair_kernel(const float* __restrict__ F0, int64_t total0, float scale0, uint16_t* __restrict__ H0,
                                                           int blocks0, const float* __restrict__ F1, int64_t total1, float scale1,
                                                           uint16_t* __restrict__ H1, double* __restrict__ zero2,
                                                           const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    if (zero2 && blockIdx.x == 0 && threadIdx.x < 2) zero2[threadIdx.x] = 0.0;
    const bool first = (int)blockIdx.x < blocks0;
    const float* F = first ? F0 : F1;
    uint16_t* H = first ? H0 : H1;
    const int64_t total = first ? total0 : total1;
    const float scale = first ? scale0 : scale1;
    const int64_t b = first ? blockIdx.x : blockIdx.x - blocks0, nb = first ? blocks0 : (int)gridDim.x - blocks0;
    for (int64_t i = (b * 256 + threadIdx.x) * 4; i < total; i += nb * 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(F + i) * scale;
        uint16_t h[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) h[q] = __builtin_bit_cast(uint16_t, (_Float16)fminf(fmaxf(v[q], -65504.f), 65504.f));
        *reinterpret_cast<uint2*>(H + i) = uint2{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
    }
}

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8m;

// ONE = true: one fp16 addend per factor and one product (11 significant bits per operand, ~2e-4 |P| per cell, unbiased): for
// matrices of >= 2^24 cells, where the error of the SUM is that over sqrt(cells).  ONE = false: two bf16 addends, three products.
template <int KP, int WAVES, bool ONE>
__global__ __launch_bounds__(WAVES * 64) void mae_kernel(const uint32_t* __restrict__ XTbits, int64_t ldxt, int64_t n_pad,
                                                   const uint16_t* __restrict__ Uh, const uint16_t* __restrict__ Ul,
                                                   const uint16_t* __restrict__ Vh, const uint16_t* __restrict__ Vl,
                                                   int row_blocks, int rb_per_xcd, int stages_per_group, double* __restrict__ sum,
                                                   const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    constexpr int KS = KP / 32;            // k-steps of 32
    constexpr int ROWB = KP * 2;           // bytes of one row of one addend
    constexpr int CH = ROWB / 16;          // 16-byte k-groups per row (4 or 8)
    constexpr int X_BYTES = 64 * 32;       // the workgroup's 256 bits of 64 rows of X^T
    constexpr int NADD = ONE ? 1 : 2;      // addends of V in a stage
    constexpr int STAGE_BYTES = NADD * 64 * ROWB + X_BYTES;  // [addend][64 rows of V][ROWB], then [64 rows of X^T][32 bytes]
    constexpr int RING = 3;                // stages in LDS: the one in use and the two behind it in flight
    __shared__ __attribute__((aligned(16))) char smem[RING * STAGE_BYTES];
    __shared__ double red[WAVES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int total_stages = (int)(n_pad / 64);
    // Workgroup -> (row block, column range), XCD-aware.  A workgroup covers 256 rows of U = 32 bytes of every row of X^T it
    // touches, so four neighbouring row blocks share each 128-byte line of X^T, and every row block of a column range streams
    // the same rows of V.  Consecutive workgroup ids go round the 8 XCDs (each with its own L2): with the plain (row block,
    // range) grid the four sharers of a line sat on four different XCDs and X^T crossed the fabric four times (1 GB per pass at
    // the bench size -- the pass was bound by that, not by the MFMAs).  Here the ids that share an XCD (id % 8) get a contiguous
    // run of row blocks, dispatched range by range: line sharers start together on one L2.
    const int per_xcd = gridDim.x >> 3;  // = rb_per_xcd * groups
    const int local = blockIdx.x >> 3;
    const int rb = (int)(blockIdx.x & 7) * rb_per_xcd + local % rb_per_xcd;
    (void)per_xcd;
    const int s0 = (local / rb_per_xcd) * stages_per_group;
    const int s1 = min(total_stages, s0 + stages_per_group);
    if (rb >= row_blocks || s0 >= s1) return;
    float acc_abs = 0.f;
    double total = 0.0;
    const int64_t i0 = ((int64_t)rb * WAVES + wave) * 64;  // this wave's 64 rows of U

    // A fragments: lane (c, g) holds row c of row block mt, k = 32 ks + 8 g .. + 7, of both addends.  WHICH row of U that is, is
    // free, and chosen so that the four cells a lane gets back per block (tile rows 4 g + q, q = 0 .. 3, of its column) are four
    // bits EIGHT apart in one word of X^T: tile row 4 g' + q of block mt = row 32 (mt >> 1) + 8 q + 4 (mt & 1) + g' of the wave's 64.
    u32x4 ah[4][KS], al[4][KS];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int64_t off = (i0 + 32 * (mt >> 1) + 8 * (c & 3) + 4 * (mt & 1) + (c >> 2)) * KP + 32 * ks + 8 * g;
            ah[mt][ks] = *reinterpret_cast<const u32x4*>(Uh + off);
            al[mt][ks] = ONE ? u32x4{0u, 0u, 0u, 0u} : *reinterpret_cast<const u32x4*>(Ul + off);
        }

    // DMA: a stage = 64 rows of V x 2 addends + the X^T tile; V piece q (1 KiB) = 1024 / ROWB rows of one addend; lane l: row
    // (l / CH), LDS chunk l % CH <- source chunk (l % CH) ^ (row % CH).  X^T: wave w brings rows 16 w .. 16 w + 15, two 16-byte
    // halves per row (lanes 0 .. 31).  Through LDS the X words ride the same ring as V (two stages ahead); fetched into
    // registers one stage ahead they were an HBM round trip per stage that the single barrier wait exposed (0.12 ms of the pass).
    constexpr int ROWS_PER_PIECE = 1024 / ROWB;          // 8 (kp = 64) or 16 (kp = 32)
    constexpr int PIECES = NADD * 64 / ROWS_PER_PIECE;   // per stage
    constexpr int PER_WAVE = PIECES / WAVES;
    static_assert(PIECES % WAVES == 0, "stage must split evenly over the waves");
    static_assert(WAVES == 4, "the X^T tile is brought in by four waves, 16 rows each");
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
    const unsigned x_src = (unsigned)((16 * wave + (lane >> 1)) * ldxt + rb * 8 + (lane & 1) * 4);  // word offset inside a stage
    auto issue = [&](int stage, int slot) {
#ifndef BMF_EXP_MAE_NO_DMA
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int q = wave * PER_WAVE + i;
            const int term = q / (64 / ROWS_PER_PIECE);
            const uint16_t* base = (term ? Vl : Vh) + (int64_t)stage * 64 * KP;  // wave-uniform
            char* dst = smem + slot * STAGE_BYTES + q * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + d_off[i]),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
#endif
#ifndef BMF_EXP_MAE_NO_X
        if (lane < 32) {
            const uint32_t* base = XTbits + (int64_t)stage * 64 * ldxt;  // wave-uniform
            char* dst = smem + slot * STAGE_BYTES + NADD * 64 * ROWB + wave * 512;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + x_src),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
#endif
    };
    constexpr int OPS_PER_STAGE = PER_WAVE + 1;  // vector-memory operations one wave issues per stage

    // LDS reads are hand-placed ds_read / s_waitcnt (inline asm, as in xf_bits.hip): with plain loads the compiler orders every
    // LDS read after the LDS-DMA it might alias -- s_waitcnt vmcnt(0) right after the next stage's DMA was issued -- and puts
    // lgkmcnt(0) in front of the first MFMA of a phase, i.e. after the *next* tile's reads were issued.
    // B fragments of one 16-column tile: lane (c, g) reads V[16 jt + c][32 ks + 8 g .. + 7] of both addends.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    unsigned b_addr[4][KS];  // per tile of a stage, per k-step: byte address in slot 0 (high addend; low = + 64 rows)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) b_addr[jt][ks] = lds0 + (unsigned)((16 * jt + c) * ROWB + (((ks * 4 + g) ^ ((16 * jt + c) % CH)) << 4));
    auto load_b = [&](unsigned slot_off, int jt, u32x4 (&bh)[KS], u32x4 (&bl)[KS]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const unsigned addr = b_addr[jt][ks] + slot_off;
#ifdef BMF_EXP_MAE_NO_LDS
            asm volatile("" : "=v"(bh[ks]) : "v"(addr));
            asm volatile("" : "=v"(bl[ks]) : "v"(addr));
#else
            asm volatile("ds_read_b128 %0, %1" : "=v"(bh[ks]) : "v"(addr));
            if constexpr (!ONE) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bl[ks]) : "v"(addr), "n"(64 * ROWB));
            else asm volatile("" : "=v"(bl[ks]));
#endif
        }
    };
    // X^T row j = 16 jt + c of the stage, the two words that cover this wave's rows i0 .. i0 + 63
    const unsigned x_addr = lds0 + (unsigned)(NADD * 64 * ROWB + c * 32 + wave * 8);
    auto load_xw = [&](unsigned slot_off, uint2 (&x)[4]) {
        const unsigned addr = x_addr + slot_off;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x[jt]) : "v"(addr), "n"(jt * 16 * 32));
    };
    auto wait_b = [&](u32x4 (&bh)[KS], u32x4 (&bl)[KS]) {  // all outstanding LDS reads have landed; ties the fragments to the wait
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            asm volatile("" : "+v"(bh[ks]));
            asm volatile("" : "+v"(bl[ks]));
        }
    };
    // the accumulators arrive holding -x (init_p), so that the MFMAs leave p - x and the element-wise part is one |.| + add per cell
    auto products = [&](const u32x4 (&bh)[KS], const u32x4 (&bl)[KS], f32x4 (&p)[4]) {
#ifdef BMF_EXP_MAE_NO_MFMA
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) p[mt] = __builtin_bit_cast(f32x4, bh[mt & (KS - 1)] ^ bl[mt & (KS - 1)] ^ ah[mt][0]);
        return;
#endif
        if constexpr (ONE) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    p[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8m, ah[mt][ks]), __builtin_bit_cast(f16x8m, bh[ks]), p[mt], 0, 0, 0);
            return;
        }
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
    // X_ONE x of one tile into the accumulators.  D layout: column = lane & 15 (this lane's j), tile rows 4 g + q of block mt = bits
    // 8 q + 4 (mt & 1) + g of word mt >> 1 (see the A fragments): a rotate brings them to bit 6 of the four bytes, the AND leaves
    // 0x40 = 2.0 as OCP fp8 (e4m3) or 0, two packed fp8 -> f32 converts.  Four full-rate instructions per four cells (the first
    // version -- bit-field extract, two integer multiplies, AND -- had a quarter-rate v_mul_lo_u32 in it: 36 cycles instead of 16).
    const unsigned rot[2] = {(unsigned)(g - 6) & 31u, (unsigned)(g - 2) & 31u};
    auto init_p = [&](const uint2 x, f32x4 (&p)[4]) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const unsigned w = mt < 2 ? x.x : x.y;
            const unsigned spread = __builtin_amdgcn_alignbit(w, w, rot[mt & 1]) & 0x40404040u;
            const f32x2 x01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)spread, false);
            const f32x2 x23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)spread, true);
            p[mt] = f32x4{x01[0], x01[1], x23[0], x23[1]};
        }
    };
    auto reduce = [&](const f32x4 (&p)[4]) {
#ifdef BMF_EXP_MAE_NO_REDUCE
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc_abs += p[mt][0] + p[mt][1] + p[mt][2] + p[mt][3];
        return;
#endif
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            acc_abs += fabsf(p[mt][0]);
            acc_abs += fabsf(p[mt][1]);
            acc_abs += fabsf(p[mt][2]);
            acc_abs += fabsf(p[mt][3]);
        }
    };
    // One phase = the MFMAs of one tile interleaved with the element-wise work on the tile before it, while the B fragments of
    // the tile after it are on their way from LDS (read one phase ahead into the other register set: with the reads issued
    // right before their MFMAs every tile exposed an LDS round trip).  The pipeline runs across stages: the element-wise
    // work on a stage's last tile overlaps the first MFMAs of the next stage.
#define BMF_MAE_PHASE(LOADOFF, LOADJT, BNEXT_H, BNEXT_L, BH, BL, PNEW, XNEW, POLD)                  \
    load_b(LOADOFF, LOADJT, BNEXT_H, BNEXT_L);                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    init_p(XNEW, PNEW);                                                                             \
    products(BH, BL, PNEW);                                                                         \
    reduce(POLD);                                                                                   \
    interleave_mfma_valu<(ONE ? 4 : 12) * KS, (ONE ? 4 : 2)>(std::make_integer_sequence<int, (ONE ? 4 : 12) * KS>{}); \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    wait_b(BNEXT_H, BNEXT_L);                                                                       \
    __builtin_amdgcn_sched_barrier(0);

    // prologue: the first two stages land before the first barrier
    issue(s0, 0);
    if (s0 + 1 < s1) issue(s0 + 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {  // the A fragments are complete as well: no pending load may reach into the stage loop
            asm volatile("" : "+v"(ah[mt][ks]));
            asm volatile("" : "+v"(al[mt][ks]));
        }
    __syncthreads();
    u32x4 b0h[KS], b0l[KS], b1h[KS], b1l[KS];
    f32x4 pa[4], pb[4];
    uint2 xw[4], xwn[4];   // X words of the stage in hand / of the next one (read from LDS one phase before their first use)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) pb[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    load_b(0u, 0, b0h, b0l);
    load_xw(0u, xw);
    wait_b(b0h, b0l);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) { asm volatile("" : "+v"(xw[jt].x)); asm volatile("" : "+v"(xw[jt].y)); }
    __builtin_amdgcn_sched_barrier(0);
    int slot = 0;  // ring slot of stage s
    for (int s = s0; s < s1; ++s) {
        const int slot1 = slot == RING - 1 ? 0 : slot + 1, slot2 = slot1 == RING - 1 ? 0 : slot1 + 1;
        const unsigned off = (unsigned)(slot * STAGE_BYTES), off1 = (unsigned)(slot1 * STAGE_BYTES);
        // slot2 held stage s - 1: every wave finished its reads of it before the barrier of that stage
        const bool ahead = s + 2 < s1;
        if (ahead) issue(s + 2, slot2);
        BMF_MAE_PHASE(off, 1, b1h, b1l, b0h, b0l, pa, xw[0], pb)
        BMF_MAE_PHASE(off, 2, b0h, b0l, b1h, b1l, pb, xw[1], pa)
        BMF_MAE_PHASE(off, 3, b1h, b1l, b0h, b0l, pa, xw[2], pb)
        // all of this stage's LDS reads are in registers (wait_b) and stage s + 1 has landed -- this wave's share: everything but
        // the operations of stage s + 2 just issued; the barrier makes it everyone's.  One bare barrier per stage.
#ifndef BMF_EXP_MAE_NO_SYNC
        if (ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS_PER_STAGE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#endif
        __builtin_amdgcn_sched_barrier(0);
        load_xw(off1, xwn);   // the X words of stage s + 1 (an idle slot when there is no next stage), with the fragments of its tile 0
        BMF_MAE_PHASE(off1, 0, b0h, b0l, b1h, b1l, pb, xw[3], pa)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {   // landed with that phase's wait; the copy must not move above it
            asm volatile("" : "+v"(xwn[jt].x));
            asm volatile("" : "+v"(xwn[jt].y));
            xw[jt] = xwn[jt];
        }
        total += (double)acc_abs;  // keep the fp32 partial short: one stage = 64 cells per lane
        acc_abs = 0.f;
        slot = slot1;
    }
    reduce(pb);
    total += (double)acc_abs;
#undef BMF_MAE_PHASE
    total = wave_sum(total);
    if (lane == 0) red[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) t += red[w];
        atomicAdd(sum, t * (double)(1.0f / X_ONE));
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The single-product pass on v_mfma_f32_32x32x16_f16 (round 3; the default at >= 2^24 cells).
//
// The pass is bound by the SIMD's vector ISSUE, not by the matrix pipe: per cell one instruction gets x into the accumulator and
// one adds |x - p| to the sum, each 4 cycles per 64 cells, and every MFMA holds the issue port for 8 cycles whatever its shape
// (MI355X_MICROARCH, 'vector-instruction ISSUE cost').  The 16x16x32 kernel above pays 8 MFMAs per 1024 cells (64 cycles of issue
// beside 128 of element-wise work), this one 4 (32 beside 128): 160 cycles per 1024 cells instead of 192.  What else changed
// against mae_kernel: (i) x enters as one rotate + AND + two packed fp8 converts per four cells (see init below), (ii) the
// element-wise instructions are pinned between the MFMAs by hand: left to the compiler the 16 dependent |.| adds of a tile were
// emitted in one lump after three tiles' MFMAs (instruction selection places pure nodes next to their use, and the use of the
// running sum is at the end of the stage; sched_group_barrier only sees the order it is given), so matrix and vector work alternated
// instead of overlapping -- 78 cycles per 256 cells measured against 68 of issue.
//
// Tiling: a wave = 64 rows of U (two 32-row blocks, A fragments in registers) x one stage of 64 rows of V (two 32-column tiles);
// phase = one 32-column tile = 8 MFMAs (2 row blocks x KP / 16 k-steps), 64 element-wise instructions on the OTHER accumulator set:
// 32 adds of the previous tile's |x - p|, then 32 that load the same registers with X_ONE x of the tile after this one.  LDS ring,
// DMA and workgroup map as in mae_kernel.
typedef __attribute__((ext_vector_type(16))) float f32x16;

// SQ: sum[1] += sum (x - p)^2 as well (one more vector instruction per cell; the K = 128 instance of wide.hip uses it)
template <int KP, bool XTILED, bool SQ = false>
__global__ __launch_bounds__(256) void mae32_kernel(const uint32_t* __restrict__ XTbits, int64_t ldxt, int64_t n_pad,
                                                    const uint16_t* __restrict__ Uh, const uint16_t* __restrict__ Vh, int row_blocks,
                                                    int rb_per_xcd, int stages_per_group, double* __restrict__ sum,
                                                    const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    constexpr int KS = KP / 16;            // k-steps of 16
    constexpr int ROWB = KP * 2;           // bytes of one row of V
    constexpr int CH = ROWB / 16;          // 16-byte k-groups per row (4 or 8)
    constexpr int X_BYTES = 64 * 32;       // the workgroup's 256 bits of 64 rows of X^T
    constexpr int STAGE_BYTES = 64 * ROWB + X_BYTES;
#ifndef BMF_MAE32_RING
#define BMF_MAE32_RING 3
#endif
    constexpr int RING = BMF_MAE32_RING;   // stages in LDS: the one in use and RING - 1 behind it in flight
    constexpr int WAVES = 4;
    __shared__ __attribute__((aligned(16))) char smem[RING * STAGE_BYTES];
    __shared__ double red[WAVES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int total_stages = (int)(n_pad / 64);
    const int local = blockIdx.x >> 3;     // XCD-aware workgroup map: see mae_kernel
    const int rb = (int)(blockIdx.x & 7) * rb_per_xcd + local % rb_per_xcd;
    const int s0 = (local / rb_per_xcd) * stages_per_group;
    const int s1 = min(total_stages, s0 + stages_per_group);
    if (rb >= row_blocks || s0 >= s1) return;
    const int64_t i0 = ((int64_t)rb * WAVES + wave) * 64;  // this wave's 64 rows of U

    // A fragments: lane (r, h) holds tile row r of row block mt, k = 16 ks + 8 h .. + 7.  WHICH row of U a tile row is, is free, and
    // chosen so that the four cells a lane gets back in registers 4 G .. 4 G + 3 (tile rows q + 8 G + 4 h, q = 0 .. 3, of its column)
    // are four bits EIGHT apart in one word of X^T: tile row q + 4 h' + 8 G = row 32 mt + 8 q + 2 G + h' of the wave's 64.
    u32x4 a[2][KS];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int64_t row = i0 + 32 * mt + 8 * (c & 3) + 2 * (c >> 3) + ((c >> 2) & 1);
            a[mt][ks] = *reinterpret_cast<const u32x4*>(Uh + row * KP + 16 * ks + 8 * h);
        }

    // DMA of a stage: as in mae_kernel (one addend): V piece q (1 KiB) = 1024 / ROWB rows, 16-byte k-groups XOR-swizzled with the
    // row number on the source address; X^T: wave w brings rows 16 w .. 16 w + 15, two 16-byte halves per row (lanes 0 .. 31)
    constexpr int ROWS_PER_PIECE = 1024 / ROWB;
    constexpr int PIECES = 64 / ROWS_PER_PIECE;
    constexpr int PER_WAVE = PIECES / WAVES;
    static_assert(PIECES % WAVES == 0, "stage must split evenly over the waves");
    const int d_row = lane / CH, d_chunk = lane % CH;
    unsigned d_off[PER_WAVE];
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int row = (wave * PER_WAVE + i) * ROWS_PER_PIECE + d_row;
        d_off[i] = (unsigned)(row * KP + ((d_chunk ^ (row % CH)) << 3));
    }
    // X^T tile of a stage, word offset inside it.  Plain rows: 32 bytes of each of 64 rows ldxt words apart (64 lines of 64 DRAM pages
    // per stage, each line shared with three neighbouring row blocks).  XTILED (the bmf_tile_bits copy the int8 GEMM streams):
    // block (j / 256, i / 512) = 256 rows of X^T x 16 words, contiguous: the stage is 32 bytes of each 64-byte row of ONE 4-KiB
    // piece of it, the other half belongs to the neighbouring row block (same XCD: an L2 hit for one of the two).
    const unsigned x_src = XTILED ? (unsigned)((16 * wave + (lane >> 1)) * 16 + (rb & 1) * 8 + (lane & 1) * 4)
                                  : (unsigned)((16 * wave + (lane >> 1)) * ldxt + rb * 8 + (lane & 1) * 4);
    const int64_t x_groups = ldxt >> 4;   // XTILED: blocks per 256-row tile of X^T
    auto issue = [&](int stage, int slot) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
#ifdef BMF_EXP_M32_NO_VDMA
            if (stage >= 0) break;
#endif
            const uint16_t* base = Vh + (int64_t)stage * 64 * KP;  // wave-uniform
            char* dst = smem + slot * STAGE_BYTES + (wave * PER_WAVE + i) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + d_off[i]),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
#ifndef BMF_EXP_M32_NO_XDMA
        if (lane < 32)
#else
        if (lane < 32 && stage < 0)
#endif
        {
            const uint32_t* base = XTILED ? XTbits + (((int64_t)(stage >> 2) * x_groups + (rb >> 1)) * 256 + (stage & 3) * 64) * 16
                                          : XTbits + (int64_t)stage * 64 * ldxt;  // wave-uniform
            char* dst = smem + slot * STAGE_BYTES + 64 * ROWB + wave * 512;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + x_src),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    constexpr int OPS_PER_STAGE = PER_WAVE + 1;

    // B fragments of one 32-column tile: lane (c, h) reads V[32 jt + c][16 ks + 8 h .. + 7]; hand-placed ds_read / s_waitcnt as above
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    unsigned b_addr[2][KS];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) b_addr[jt][ks] = lds0 + (unsigned)((32 * jt + c) * ROWB + (((2 * ks + h) ^ ((32 * jt + c) % CH)) << 4));
    auto load_b = [&](unsigned slot_off, int jt, u32x4 (&b)[KS]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const unsigned addr = b_addr[jt][ks] + slot_off;
            asm volatile("ds_read_b128 %0, %1" : "=v"(b[ks]) : "v"(addr));
        }
    };
    // X^T row j = 32 jt + c of the stage, the two words that cover this wave's rows
    const unsigned x_addr = lds0 + (unsigned)(64 * ROWB + c * 32 + wave * 8);
    auto load_xw = [&](unsigned slot_off, uint2 (&x)[2]) {
        const unsigned addr = x_addr + slot_off;
        asm volatile("ds_read_b64 %0, %1" : "=v"(x[0]) : "v"(addr));
        asm volatile("ds_read_b64 %0, %1 offset:1024" : "=v"(x[1]) : "v"(addr));
    };
    // bit 8 q + 2 G + h of the word -> bit 6 of byte q: 0x40 = 2.0 as OCP fp8 (e4m3) -- X_ONE -- or 0
    unsigned rot[4];
#pragma unroll
    for (int G = 0; G < 4; ++G) rot[G] = (unsigned)(2 * G + h - 6) & 31u;

    float acc0 = 0.f, acc1 = 0.f, sq0 = 0.f, sq1 = 0.f;
    double total = 0.0, total_sq = 0.0;
    // One step of a phase: 8 element-wise instructions on register group(s) of the idle accumulator set, then one MFMA into the busy
    // set.  The empty asm statements are chained nodes that take the values as operands: they pin the adds (pure nodes otherwise
    // placed at the use of the sum) and the MFMAs to their step; sched_barrier keeps the machine scheduler from undoing it.
#define BMF_PIN4(P, G) asm volatile("" : "+v"(P[4 * (G)]), "+v"(P[4 * (G) + 1]), "+v"(P[4 * (G) + 2]), "+v"(P[4 * (G) + 3]))
    auto add8 = [&](f32x16 (&p)[2], int step) {   // step 0 .. 3: registers 8 (step & 1) .. + 7 of block step >> 1
        f32x16& q = p[step >> 1];
        const int o = 8 * (step & 1);
        acc0 += fabsf(q[o + 0]); acc1 += fabsf(q[o + 1]); acc0 += fabsf(q[o + 2]); acc1 += fabsf(q[o + 3]);
        acc0 += fabsf(q[o + 4]); acc1 += fabsf(q[o + 5]); acc0 += fabsf(q[o + 6]); acc1 += fabsf(q[o + 7]);
        asm volatile("" : "+v"(acc0), "+v"(acc1));
        if constexpr (SQ) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) { sq0 = fmaf(q[o + e], q[o + e], sq0); sq1 = fmaf(q[o + e + 1], q[o + e + 1], sq1); }
            asm volatile("" : "+v"(sq0), "+v"(sq1));
        }
    };
    auto init8 = [&](f32x16 (&p)[2], const uint2 x, int step) {   // step 0 .. 3: groups 2 (step & 1), + 1 of block step >> 1
        f32x16& q = p[step >> 1];
        const unsigned w = (step >> 1) ? x.y : x.x;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int G = 2 * (step & 1) + t;
            const unsigned spread = __builtin_amdgcn_alignbit(w, w, rot[G]) & 0x40404040u;
            const f32x2 x01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)spread, false);
            const f32x2 x23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)spread, true);
            q[4 * G + 0] = x01[0]; q[4 * G + 1] = x01[1]; q[4 * G + 2] = x23[0]; q[4 * G + 3] = x23[1];
        }
        asm volatile("" : "+v"(q));
    };
    auto mfma = [&](f32x16 (&p)[2], const u32x4 (&b)[KS], int step) {   // step 0 .. 2 KS - 1
        const int ks = step >> 1, mt = step & 1;
        p[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8m, a[mt][ks]), __builtin_bit_cast(f16x8m, b[ks]), p[mt], 0, 0, 0);
        asm volatile("" : "+v"(p[mt]));
    };
#ifdef BMF_EXP_M32_NO_VALU   // timing-only flavours (wrong results): scripts/r03/mae32_ablation.sh
#define BMF_M32_VALU(...)
#else
#define BMF_M32_VALU(...) __VA_ARGS__
#endif
#ifdef BMF_EXP_M32_NO_MFMA
#define BMF_M32_MFMA(...)
#else
#define BMF_M32_MFMA(...) __VA_ARGS__
#endif
    // the element-wise steps are spread over the 2 KS MFMA steps: with KS = 4 one per MFMA, with KS = 2 two per MFMA, with KS = 8 (the
    // K = 128 instance of wide.hip) two MFMAs per step
#define BMF_MAE32_PHASE(BUSY, B, IDLE, XNEXT, WAIT_AT_HALF)                                           \
    _Pragma("unroll") for (int st = 0; st < 8; ++st) {                                                \
        if (st == 4 && (WAIT_AT_HALF)) {                                                              \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                        \
            asm volatile("" : "+v"(XNEXT.x), "+v"(XNEXT.y));                                          \
        }                                                                                             \
        BMF_M32_VALU(if (st < 4) add8(IDLE, st); else init8(IDLE, XNEXT, st - 4);)                    \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        BMF_M32_MFMA(if (KS == 8) { mfma(BUSY, B, 2 * st); mfma(BUSY, B, 2 * st + 1); }               \
                     else if (KS == 4) mfma(BUSY, B, st); else if (st & 1) mfma(BUSY, B, st >> 1);)    \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }
    auto tie_b = [&](u32x4 (&b)[KS]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(b[ks]));
    };

    // prologue: the first RING - 1 stages are requested, the first has landed before the first barrier
#pragma unroll
    for (int q = 0; q < RING - 1; ++q) issue(min(s0 + q, s1 - 1), q);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 2) * OPS_PER_STAGE) : "memory");
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(a[mt][ks]));
    __syncthreads();
    u32x4 bA[KS], bB[KS];
    f32x16 pa[2], pb[2];
    uint2 xw[2], xwn[2];
    load_b(0u, 0, bA);
    load_xw(0u, xw);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    tie_b(bA);
    asm volatile("" : "+v"(xw[0].x), "+v"(xw[0].y), "+v"(xw[1].x), "+v"(xw[1].y));
#pragma unroll
    for (int i = 0; i < 16; ++i) { pb[0][i] = 0.f; pb[1][i] = 0.f; }
#pragma unroll
    for (int st = 0; st < 4; ++st) init8(pa, xw[0], st);
    __builtin_amdgcn_sched_barrier(0);
    int slot = 0;
    for (int s = s0; s < s1; ++s) {
        const int slot1 = slot == RING - 1 ? 0 : slot + 1, slot2 = slot == 0 ? RING - 1 : slot - 1;   // of stage s + 1 / of stage s - 1 (free)
        const unsigned off = (unsigned)(slot * STAGE_BYTES), off1 = (unsigned)(slot1 * STAGE_BYTES);
        // slot2 held stage s - 1: every wave finished its reads of it before the barrier of that stage.  Branch-free: past the end the
        // last stage is fetched again into the free slot, so that the counted wait below is the same in every iteration (a branch here
        // splits the loop body into basic blocks and the element-wise work of the first tile ends up behind it)
#ifndef BMF_EXP_M32_NO_DMA
        issue(min(s + RING - 1, s1 - 1), slot2);
#endif
        // tile (s, 0): MFMAs into pa; pb: |.| of tile (s - 1, 1), then x of tile (s, 1)
        load_b(off, 1, bB);
        __builtin_amdgcn_sched_barrier(0);
        BMF_MAE32_PHASE(pa, bA, pb, xw[1], false)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tie_b(bB);
        // all of this stage's LDS reads are in registers and stage s + 1 has landed (this wave's share; the barrier makes it everyone's)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 2) * OPS_PER_STAGE) : "memory");
#ifndef BMF_EXP_M32_NO_BAR
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // tile (s, 1): MFMAs into pb; pa: |.| of tile (s, 0), then x of tile (s + 1, 0) (an idle slot's bytes after the last stage: the
        // registers they load are never used)
        load_xw(off1, xwn);
        load_b(off1, 0, bA);
        __builtin_amdgcn_sched_barrier(0);
        BMF_MAE32_PHASE(pb, bB, pa, xwn[0], true)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tie_b(bA);
        asm volatile("" : "+v"(xwn[1].x), "+v"(xwn[1].y));
        xw[1] = xwn[1];
        total += (double)acc0 + (double)acc1;   // keep the fp32 partials short: one stage = 64 cells per lane
        acc0 = 0.f;
        acc1 = 0.f;
        if constexpr (SQ) { total_sq += (double)sq0 + (double)sq1; sq0 = 0.f; sq1 = 0.f; }
        slot = slot1;
    }
#pragma unroll
    for (int st = 0; st < 4; ++st) add8(pb, st);
    total += (double)acc0 + (double)acc1;
    if constexpr (SQ) total_sq += (double)sq0 + (double)sq1;
#undef BMF_MAE32_PHASE
#undef BMF_PIN4
    total = wave_sum(total);
    if (lane == 0) red[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) t += red[w];
        atomicAdd(sum, t * (double)(1.0f / X_ONE));
    }
    if constexpr (SQ) {
        total_sq = wave_sum(total_sq);
        __syncthreads();
        if (lane == 0) red[wave] = total_sq;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) t += red[w];
            atomicAdd(sum + 1, t * (double)(1.0f / (X_ONE * X_ONE)));
        }
    }
}

// [F0 | F1] (two rows_pad x 64 fp32 blocks) -> H (rows_pad x 128 fp16, saturated), times scale: the operands of the K = 128 instance
__global__ __launch_bounds__(256) void to_f16_wide_kernel(const float* __restrict__ F0, const float* __restrict__ F1, int64_t rows, float scale,
                                                           uint16_t* __restrict__ H) {
    const int64_t total = rows * 32;   // float4 pieces: 16 per block row
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i >> 5;
        const int q = (int)(i & 31);
        const float* src = (q < 16 ? F0 : F1) + row * 64 + 4 * (q & 15);
        const f32x4 v = *reinterpret_cast<const f32x4*>(src) * scale;
        uint16_t h[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = __builtin_bit_cast(uint16_t, (_Float16)fminf(fmaxf(v[e], -65504.f), 65504.f));
        *reinterpret_cast<uint2*>(H + row * 128 + 4 * q) = uint2{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
    }
}

}  // namespace

// sums[0] += sum |X - U V^T|, sums[1] += sum (X - U V^T)^2 for factors of two 64-column blocks each (64 < k <= 128): the single-product
// fp16 pass at K = 128.  ws: (m_pad + n_pad) * 128 uint16.
int bmf_mae_wide_launch(const uint32_t* XT, int64_t ldxt, int64_t m_pad, int64_t n_pad, const float* UA, const float* UB, const float* VA,
                        const float* VB, uint16_t* ws, double* sums, int x_tiled, hipStream_t s) {
    BMF_REQUIRE(XT && UA && UB && VA && VB && ws && sums, "bmf_resid_sums_wide: null pointer");
    BMF_REQUIRE(m_pad > 0 && m_pad % 256 == 0 && n_pad > 0 && n_pad % 64 == 0, "bmf_resid_sums_wide: m_pad must be a multiple of 256, n_pad of 64");
    BMF_REQUIRE(ldxt * 32 >= m_pad && ldxt % 4 == 0, "bmf_resid_sums_wide: ldxt must be a multiple of 4 words and cover m_pad");
    BMF_REQUIRE(!x_tiled || (n_pad % 256 == 0 && ldxt % 16 == 0 && ldxt * 32 == m_pad), "bmf_resid_sums_wide: the tiled X^T needs n_pad %% 256 == 0 and ldxt == m_pad / 32, a multiple of 16");
    BMF_REQUIRE(bmf_aligned16(XT) && bmf_aligned16(UA) && bmf_aligned16(UB) && bmf_aligned16(VA) && bmf_aligned16(VB) && bmf_aligned16(ws),
                "bmf_resid_sums_wide: alignment");
    uint16_t* Uh = ws;
    uint16_t* Vh = ws + m_pad * 128;
    auto blocks = [](int64_t rows) { const int64_t b = (rows * 32 + 255) / 256; return (unsigned)(b < 2048 ? b : 2048); };
    BMF_LAUNCH(to_f16_wide_kernel, dim3(blocks(m_pad)), dim3(256), 0, s, UA, UB, m_pad, -X_ONE, Uh);
    BMF_LAUNCH(to_f16_wide_kernel, dim3(blocks(n_pad)), dim3(256), 0, s, VA, VB, n_pad, 1.0f, Vh);
    const int row_blocks = (int)(m_pad / 256), stages = (int)(n_pad / 64);
    const int rb_per_xcd = (row_blocks + 7) / 8;
    int groups = (6 * 2 * bmf_cu_count() + row_blocks - 1) / row_blocks;
    if (groups > stages) groups = stages;
    if (groups < 1) groups = 1;
    const int per = (stages + groups - 1) / groups;
    groups = (stages + per - 1) / per;
    dim3 grid((unsigned)(8 * rb_per_xcd * groups)), block(256);
    if (x_tiled) BMF_LAUNCH((mae32_kernel<128, true, true>), grid, block, 0, s, XT, ldxt, n_pad, Uh, Vh, row_blocks, rb_per_xcd, per, sums, nullptr);
    else BMF_LAUNCH((mae32_kernel<128, false, true>), grid, block, 0, s, XT, ldxt, n_pad, Uh, Vh, row_blocks, rb_per_xcd, per, sums, nullptr);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

int bmf_mae_launch(const uint32_t* XTbits, int64_t ldxt, int64_t m_pad, int64_t n_pad, const float* U, const float* V, int kp,
                   uint16_t* ws, double* sum, const int32_t* stop, hipStream_t s, int one_product, int x_tiled, int zero_sums) {
    BMF_REQUIRE(XTbits && U && V && ws && sum, "bmf_mae_sum: null pointer");
    BMF_REQUIRE(m_pad > 0 && m_pad % 256 == 0 && n_pad > 0 && n_pad % 64 == 0, "bmf_mae_sum: m_pad must be a multiple of 256, n_pad of 64");
    BMF_REQUIRE(ldxt * 32 >= m_pad && ldxt % 4 == 0, "bmf_mae_sum: ldxt must be a multiple of 4 words and cover m_pad");
    BMF_REQUIRE(kp == 32 || kp == 64, "bmf_mae_sum: kp must be 32 or 64");
    BMF_REQUIRE(bmf_aligned16(U) && bmf_aligned16(V) && bmf_aligned16(ws) && bmf_aligned16(XTbits), "bmf_mae_sum: alignment");
    uint16_t* Uh = ws;
    uint16_t* Ul = Uh + m_pad * kp;
    uint16_t* Vh = Ul + m_pad * kp;
    uint16_t* Vl = Vh + n_pad * kp;
    const int64_t tu = m_pad * kp, tv = n_pad * kp;
    auto blocks = [](int64_t total) { const int64_t b = (total / 4 + 255) / 256; return (unsigned)(b < 2048 ? b : 2048); };
    // one_product < 0: by size -- a single fp16 product once the sum runs over >= 2^24 cells (measured at 1.5e6 cells: 2e-6 of the
    // sum; the per-cell error, ~2e-4 |P|, mostly averages out); the three-product bf16 split otherwise (per-cell accuracy)
    const bool one = one_product < 0 ? (m_pad * n_pad >= (1 << 24)) : one_product != 0;
    BMF_REQUIRE(!x_tiled || (one && n_pad % 256 == 0 && ldxt % 16 == 0 && ldxt * 32 == m_pad),
                "bmf_mae_sum_tiled: the tiled X^T needs the single-product pass, n_pad %% 256 == 0 and ldxt == m_pad / 32, a multiple of 16");
    if (one) {
        BMF_LAUNCH(to_f16_pair_kernel, dim3(blocks(tu) + blocks(tv)), dim3(256), 0, s, U, tu, -X_ONE, Uh, (int)blocks(tu), V, tv, 1.0f, Vh,
                   zero_sums ? sum : nullptr, stop);
    } else {
        if (zero_sums) BMF_HIP_CHECK(hipMemsetAsync(sum, 0, 2 * sizeof(double), s));
        BMF_LAUNCH(split_rows_kernel, dim3(blocks(tu)), dim3(256), 0, s, U, tu, -X_ONE, Uh, Ul, stop);
        BMF_LAUNCH(split_rows_kernel, dim3(blocks(tv)), dim3(256), 0, s, V, tv, 1.0f, Vh, Vl, stop);
    }
    // 4 waves = 256 rows of U per workgroup (8 waves / 512 rows halve the V traffic through L2 but leave one workgroup per
    // CU: measured 886 vs 686 us)
    // 4 waves = 256 rows of U per workgroup, two workgroups per CU.  8 waves / 512 rows (one workgroup per CU, the same two
    // waves per SIMD) halve the V traffic from L2 but tie eight waves to every stage barrier: 0.75 vs 0.69 ms.
    const int row_blocks = (int)(m_pad / 256), stages = (int)(n_pad / 64);
    const int rb_per_xcd = (row_blocks + 7) / 8;
    // column ranges: about six workgroups per resident slot (2 per CU), so that the last round is short
    int groups = (6 * 2 * bmf_cu_count() + row_blocks - 1) / row_blocks;
    if (groups > stages) groups = stages;
    if (groups < 1) groups = 1;
    const int per = (stages + groups - 1) / groups;
    groups = (stages + per - 1) / per;
    dim3 grid((unsigned)(8 * rb_per_xcd * groups)), block(256);
    if (one) {   // (one fp16 product per cell: the 32x32x16 form, mae32_kernel; the 16x16 form it replaced lost its switch in round 5)
        if (kp == 32 && x_tiled) BMF_LAUNCH((mae32_kernel<32, true>), grid, block, 0, s, XTbits, ldxt, n_pad, Uh, Vh, row_blocks, rb_per_xcd, per, sum, stop);
        else if (kp == 32) BMF_LAUNCH((mae32_kernel<32, false>), grid, block, 0, s, XTbits, ldxt, n_pad, Uh, Vh, row_blocks, rb_per_xcd, per, sum, stop);
        else if (x_tiled) BMF_LAUNCH((mae32_kernel<64, true>), grid, block, 0, s, XTbits, ldxt, n_pad, Uh, Vh, row_blocks, rb_per_xcd, per, sum, stop);
        else BMF_LAUNCH((mae32_kernel<64, false>), grid, block, 0, s, XTbits, ldxt, n_pad, Uh, Vh, row_blocks, rb_per_xcd, per, sum, stop);
    } else if (kp == 32) BMF_LAUNCH((mae_kernel<32, 4, false>), grid, block, 0, s, XTbits, ldxt, n_pad, Uh, Ul, Vh, Vl, row_blocks, rb_per_xcd, per, sum, stop);
    else BMF_LAUNCH((mae_kernel<64, 4, false>), grid, block, 0, s, XTbits, ldxt, n_pad, Uh, Ul, Vh, Vl, row_blocks, rb_per_xcd, per, sum, stop);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_mae_sum(const uint32_t* XTbits, int64_t ldxt, int64_t m_pad, int64_t n_pad, const float* U, const float* V,
                           int kp, uint16_t* ws, double* sum, void* stream) {
    return bmf_mae_launch(XTbits, ldxt, m_pad, n_pad, U, V, kp, ws, sum, nullptr, (hipStream_t)stream, -1, 0, 0);
}

extern "C" int bmf_mae_sum_ex(const uint32_t* XTbits, int64_t ldxt, int64_t m_pad, int64_t n_pad, const float* U, const float* V,
                              int kp, uint16_t* ws, double* sum, int one_product, void* stream) {
    return bmf_mae_launch(XTbits, ldxt, m_pad, n_pad, U, V, kp, ws, sum, nullptr, (hipStream_t)stream, one_product, 0, 0);
}

extern "C" int bmf_mae_sum_tiled(const uint32_t* XTtiled, int64_t ldxt, int64_t m_pad, int64_t n_pad, const float* U, const float* V,
                                 int kp, uint16_t* ws, double* sum, void* stream) {
    return bmf_mae_launch(XTtiled, ldxt, m_pad, n_pad, U, V, kp, ws, sum, nullptr, (hipStream_t)stream, 1, 1, 0);
}
