// Updates whose m x n product goes through an element-wise non-linearity before it is contracted again:
//
//   PNLPF (sigmoid link)        PyBMF/models/PNLPF.py:61-91
//     S = lam (U V^T - 1/2),  sig = sigmoid(S),  d = sig (1 - sig)
//     num_U = lam (W o X o d) V + 3 reg U^2        den_U = lam (W o sig o d) V + 2 reg U^3 + reg U      (V likewise, transposed)
//   WNMF, Kullback-Leibler      PyBMF/models/WNMF.py:111-129
//     num_U = ((W o X) / (U V^T)) V                den_U = 1 V   (column sums of V, the same for every row)
//
// Re-association does not apply, so the pass is tile-fused and the m x n intermediate never exists: per 32 x 32 tile
//   1. P^T = B A^T on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), transposed like the residual pass so that a lane owns
//      ONE row i of X / A and 16 columns j -- i.e. 16 bits of one word of X;
//   2. g1 = lam x d (or x / p), g2 = lam sig d, element-wise in registers;
//   3. out[i][:] += sum_j g[i][j] B[j][:] on the same MFMA: the C/D registers of step 1 are exactly the A operand of step 3
//      if the reduction order is taken as "the j this lane holds in register s" (the order is free as long as both operands
//      agree), and the B operand of step s is then row j0 + jr(s, h) of B -- a plain coalesced load.  No LDS, no shuffle.
// The row block's two accumulators stay in registers over the whole sweep of j; the column range can be split over
// blockIdx.y into slabs (summed in slab order by the consumer).
//
// Algorithmic work: 2 m n k (P) + 2 m n k per contraction = 6 m n k (sigmoid link) / 4 m n k (KL) per factor update, exact fp32:
// bound by the fp32 MFMA (157 TFLOP/s).
#include "common.h"

#include <cstdlib>
#include <type_traits>

namespace {

__device__ __forceinline__ void sigmoid_parts(float s, float& sig, float& d) {
    // sig = sigmoid(s), d = sig (1 - sig) without cancellation: with e = exp(-|s|), d = e / (1 + e)^2
    const float e = __expf(-fabsf(s));
    const float r = __builtin_amdgcn_rcpf(1.0f + e);  // 1 ulp; an IEEE division is ~10 instructions per cell
    sig = s >= 0.f ? r : e * r;
    d = e * r * r;
}

template <int KP, int LINK>
__global__ __launch_bounds__(256) void link_pass_kernel(const uint32_t* __restrict__ Xbits, int64_t ldx, int rows, int cols,
                                                         const float* __restrict__ A, const float* __restrict__ B,
                                                         float lam, int col_tiles_per_block, float* __restrict__ num,
                                                         float* __restrict__ den, int64_t slab_stride) {
    constexpr int KH = KP / 2, NT = KP / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * 32;  // 32 rows of X per wave
    const int col_tiles = (cols + 31) / 32;
    const int jt0 = blockIdx.y * col_tiles_per_block;
    const int jt1 = min(jt0 + col_tiles_per_block, col_tiles);

    float a[KH];
    {
        const float* ap = A + (i0 + c) * KP + KH * h;
#pragma unroll
        for (int s = 0; s < KH; s += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(ap + s);
            a[s] = v[0]; a[s + 1] = v[1]; a[s + 2] = v[2]; a[s + 3] = v[3];
        }
    }
    const bool row_ok = (i0 + c) < rows;
    f32x16 o1[NT], o2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) { o1[nt][i] = 0.f; o2[nt][i] = 0.f; }

    for (int jt = jt0; jt < jt1; ++jt) {
        const int64_t j0 = (int64_t)jt * 32;
        float b[KH];
        const float* bp = B + (j0 + c) * KP + KH * h;
#pragma unroll
        for (int s = 0; s < KH; s += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(bp + s);
            b[s] = v[0]; b[s + 1] = v[1]; b[s + 2] = v[2]; b[s + 3] = v[3];
        }
        const unsigned xw = Xbits[(i0 + c) * ldx + jt];
        // the B operand of the second contraction: row j0 + jr(s, h), columns 32 nt + c  (requested early: L2 latency)
        float bv[16][NT];
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[s][nt] = B[(j0 + (s & 3) + 8 * (s >> 2) + 4 * h) * KP + 32 * nt + c];

        f32x16 p;
#pragma unroll
        for (int i = 0; i < 16; ++i) p[i] = 0.f;
#pragma unroll
        for (int s = 0; s < KH; ++s) p = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s], a[s], p, 0, 0, 0);

        float g1[16], g2[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int jr = (i & 3) + 8 * (i >> 2) + 4 * h;
            const bool ok = row_ok && (j0 + jr) < cols;
            const bool x = (xw >> jr) & 1u;
            if (LINK == BMF_LINK_SIGMOID) {
                float sig, d;
                sigmoid_parts(lam * (p[i] - 0.5f), sig, d);
                g1[i] = (ok && x) ? lam * d : 0.f;
                g2[i] = ok ? lam * sig * d : 0.f;
            } else {
                g1[i] = (ok && x && p[i] > 0.f) ? __builtin_amdgcn_rcpf(p[i]) : 0.f;
                g2[i] = 0.f;
            }
        }
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                o1[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1[s], bv[s][nt], o1[nt], 0, 0, 0);
                if (LINK == BMF_LINK_SIGMOID) o2[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(g2[s], bv[s][nt], o2[nt], 0, 0, 0);
            }
    }
    // C/D layout: lane (c, h) holds column 32 nt + c of rows i0 + (reg & 3) + 8 (reg >> 2) + 4 h
    float* on = num + (int64_t)blockIdx.y * slab_stride;
    float* od = den ? den + (int64_t)blockIdx.y * slab_stride : nullptr;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t row = i0 + (i & 3) + 8 * (i >> 2) + 4 * h;
            on[row * KP + 32 * nt + c] = o1[nt][i];
            if (LINK == BMF_LINK_SIGMOID && od) od[row * KP + 32 * nt + c] = o2[nt][i];
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same pass on the bf16 MFMA (v_mfma_f32_32x32x16_bf16, 16x the fp32-MFMA rate) with split operands.  P goes through the
// non-linearity (x lamda inside a sigmoid, or a reciprocal), so it is kept at fp32 accuracy: three bf16 addends per factor, the
// six products >= 2^-24.  The second contraction is linear and sums ~n terms with independent rounding: two addends per
// operand, three products (2^-16 per product).
//   P^T = B A^T:  A-operand = rows j of F_other (row-major bf16 copies), B-operand = this wave's 32 rows of F_self, kept in
//                 registers for the whole sweep.  Same C/D layout as the fp32 MFMA: lane (c, h) owns row i0 + c and the 16
//                 columns jr(reg, h) = (reg & 3) + 8 (reg >> 2) + 4 h.
//   out += G B:   G is split in registers (v_cvt_pk_bf16_f32) and packed 8 values per k-chunk q = reg / 8: these ARE the A
//                 fragments of a 32x32x16 MFMA if the reduction index of chunk q is taken as t <-> column jr(8 q + t, h).  The
//                 matching B fragment -- F_other[j0 + jr(8 q + t, h)][kk], t = 0..7 -- is one 16-byte load from a copy of
//                 F_other stored in exactly that order (link_split_kernel: [32-row block][q][h][kk][t]).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int link_jr(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// F (rows_pad x KP fp32) -> three row-major bf16 addends RH + RM + RL = F exactly (operands of P, which goes through the
// non-linearity and is kept at fp32 accuracy) and two permuted addends PH, PL (operand of the linear contraction); all
// rows_pad x KP bf16
// (round 4) The operands of P are now TWO fp16 addends of the factor scaled by a power of two S (max |F| S in [2^14, 2^15)): hi = f16(F S),
// lo = f16(F S - hi) -- 22 significant bits relative to the factor's largest entry, so that P = (hi hi' + hi lo' + lo hi') / (S S') is right to
// 2^-22 with THREE MFMAs per k-step where the three-bf16-addend form needed six for 2^-24.  These passes are limited by what the matrix pipe may
// draw (the clock falls as its utilisation rises, profiles/r04_pmc_link.md): fewer MFMAs is the lever that is left.  ws keeps its five-array
// layout: [0] hi, [1] lo (fp16, row-major), [2] the scale (two floats: S, 1 / S) and the max word in its first 16 bytes, [3], [4] the permuted bf16
// hi / lo of the contraction.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8l;
__global__ __launch_bounds__(256) void link_max_kernel(const float* __restrict__ F, int64_t total, unsigned* __restrict__ maxbits) {
    __shared__ float sh[4];
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(F[i]));
    m = fmaxf(m, __shfl_xor(m, 32, 64)); m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 8, 64));
    m = fmaxf(m, __shfl_xor(m, 4, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64)); m = fmaxf(m, __shfl_xor(m, 1, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {   // one atomic per workgroup (4096 of them on one address were 50 us; non-negative floats order like their bits)
        m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
        if (m > 0.f) atomicMax(maxbits, __float_as_uint(m));
    }
}

__device__ __forceinline__ float link_scale_from_max(unsigned maxbits) {
    const float mx = __uint_as_float(maxbits);
    if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.0f;
    int ex;
    (void)frexpf(mx, &ex);            // mx = m 2^ex, m in [0.5, 1)
    return ldexpf(1.0f, 15 - ex);     // mx S in [2^14, 2^15)
}

// Column maxima of |F| (kp <= 64 columns; non-negative floats order like their bit patterns): thread t takes column t % kp.
__global__ __launch_bounds__(256) void link_colmax_kernel(const float* __restrict__ F, int64_t rows_pad, int kp, unsigned* __restrict__ colmaxbits) {
    __shared__ float sh[256];
    const int col = threadIdx.x % kp, sub = threadIdx.x / kp, nsub = 256 / kp;
    float m = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * nsub + sub; r < rows_pad; r += (int64_t)gridDim.x * nsub) m = fmaxf(m, fabsf(F[r * kp + col]));
    sh[threadIdx.x] = m;
    __syncthreads();
    if (sub == 0) {
        for (int q = 1; q < nsub; ++q) m = fmaxf(m, sh[q * kp + col]);
        if (m > 0.f && m < 3.0e38f) atomicMax(colmaxbits + col, __float_as_uint(m));
    }
}

// Per-column power-of-two scales S_k (factor A) and T_k (factor B) with S_k T_k = C for every column, so that the scaled product is C P:
//   a_k < 2^ea_k, b_k < 2^eb_k the column maxima;  S_k = 2^(15 - ea_k) puts A's column into [2^14, 2^15);  C = 2^(30 - M), M = max_k (ea_k + eb_k);
//   T_k = C / S_k = 2^(15 - M + ea_k), so b_k T_k <= 2^15 as well.  A column pair's precision (2^-22 of a_k b_k ... scaled) is proportional to its
//   largest possible contribution to P -- unlike one scale per factor, where a small column of one factor lost its lo addend even when the
//   partner's column was large.  Columns with a_k b_k = 0 contribute nothing: scale 0 (written as zeros).
// words of a workspace's third array: [0] S (unused here), [1] this factor's share of 1 / C, [2] legacy global max, [4, 4 + kp) column max bits,
// [4 + kp, 4 + 2 kp) column scales.
__global__ void link_pair_scales_kernel(unsigned* __restrict__ wa, unsigned* __restrict__ wb, int kp) {
    __shared__ int se[64];
    const int k = threadIdx.x;
    int ea = 0, eb = 0;
    bool live = false;
    if (k < kp) {
        const float a = __uint_as_float(wa[4 + k]), b = __uint_as_float(wb[4 + k]);
        live = a > 0.f && b > 0.f;
        if (live) { (void)frexpf(a, &ea); (void)frexpf(b, &eb); }
        se[k] = live ? ea + eb : -100000;
    }
    __syncthreads();
    int M = -100000;
    for (int j = 0; j < kp; ++j) M = max(M, se[j]);
    if (k < kp) {
        reinterpret_cast<float*>(wa)[4 + kp + k] = live ? ldexpf(1.0f, 15 - ea) : 0.f;
        reinterpret_cast<float*>(wb)[4 + kp + k] = live ? ldexpf(1.0f, 15 - M + ea) : 0.f;
    }
    if (k == 0) {
        const float invC = M > -100000 ? ldexpf(1.0f, M - 30) : 1.0f;
        reinterpret_cast<float*>(wa)[0] = 1.0f; reinterpret_cast<float*>(wa)[1] = invC;
        reinterpret_cast<float*>(wb)[0] = 1.0f; reinterpret_cast<float*>(wb)[1] = 1.0f;
    }
}

__global__ __launch_bounds__(256) void link_split_kernel(const float* __restrict__ F, int64_t rows_pad, int kp,
                                                          uint16_t* __restrict__ RH, uint16_t* __restrict__ RM,
                                                          uint16_t* __restrict__ RL, uint16_t* __restrict__ PH,
                                                          uint16_t* __restrict__ PL, const float* __restrict__ colscale) {
    const int64_t total = rows_pad * kp;
    // colscale != nullptr: one power-of-two scale per column, made by link_pair_scales_kernel for this factor AND its partner (their product is the
    // same for every column; 0 = the column contributes nothing and is written as zeros); else one scale for the whole factor from its maximum
    const float S = colscale ? 1.0f : link_scale_from_max(reinterpret_cast<const unsigned*>(RL)[2]);
    if (!colscale && blockIdx.x == 0 && threadIdx.x == 0) { reinterpret_cast<float*>(RL)[0] = S; reinterpret_cast<float*>(RL)[1] = 1.0f / S; }
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        // idx walks the PERMUTED layout (coalesced writes): ((((jb * 2 + q) * 2 + h) * kp + kk) * 8 + t)
        const int t = (int)(idx & 7);
        const int64_t r1 = idx >> 3;
        const int kk = (int)(r1 % kp);
        const int64_t r2 = r1 / kp;
        const int h = (int)(r2 & 1), q = (int)((r2 >> 1) & 1);
        const int64_t jb = r2 >> 2;
        const int64_t row = jb * 32 + link_jr(8 * q + t, h);
        const float f = F[row * kp + kk];
        const uint16_t hi = bf16_bits(f);
        const uint16_t mid = bf16_bits(f - bf16_to_f32(hi));
        PH[idx] = hi;
        PL[idx] = mid;
        const float fs = f * (colscale ? colscale[kk] : S);
        const _Float16 h16 = (_Float16)fs;
        const _Float16 l16 = (_Float16)(fs - (float)h16);
        RH[row * kp + kk] = __builtin_bit_cast(uint16_t, h16);
        RM[row * kp + kk] = __builtin_bit_cast(uint16_t, l16);
    }
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {  // a in the low half
    return (unsigned)bf16_bits(a) | ((unsigned)bf16_bits(b) << 16);
}

// ---- the element-wise part of a tile, written for the instruction count (round 4: the passes were 54 % VALU-busy against 38 % MFMA,
// ~520 vector instructions per lane and tile; profiles/r04_pmc_link.md) ----
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v)); }   // one v_cvt_pk_bf16_f32

// sig = sigmoid(s), d = sig (1 - sig), sd = sig d for s = lam (p - 1/2); c1 = -lam log2(e), c0 = lam log2(e) / 2.
// e = 2^min(-s log2 e, 100) = exp(-s) needs no |s|: e r r with r = 1 / (1 + e) has no cancellation at either end, and the clamp keeps
// e r r finite (2^100 -> d = 2^-100) where exp(-s) would overflow to inf * 0.  Scalar f32 on purpose (and link.hip is built with
// -fno-slp-vectorize): beside MFMAs a v_pk_*_f32 costs more issue time than the two instructions it replaces
// (MI355X_MICROARCH.md, per-instruction constants).
__device__ __forceinline__ void sigmoid_cell(float p, float c1, float c0, float& r, float& d) {
    const float e = __builtin_amdgcn_exp2f(fminf(fmaf(p, c1, c0), 100.f));
    r = __builtin_amdgcn_rcpf(1.0f + e);
    d = (e * r) * r;
}

// all-ones / all-zeros word from bit b of xs: v_bfe_i32 (sign-extended 1-bit field).  As a builtin the compiler turned it into and + compare +
// select (three instructions where two do: the field and the AND that applies it).
__device__ __forceinline__ unsigned bit_mask(unsigned xs, int b) {
    int m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(xs), "s"(b));
    return (unsigned)m;
}

// hi / lo bf16 words of a pair: hi = bf16(g), lo = bf16(g - hi)
__device__ __forceinline__ void split_pair(f32x2 g, unsigned& hi, unsigned& lo) {
    hi = cvt_pk_bf16(g);
    lo = cvt_pk_bf16(f32x2{g[0] - __uint_as_float(hi << 16), g[1] - __uint_as_float(hi & 0xffff0000u)});
}

// ---------------------------------------------------------------------------------------------------------------------
// The pass software-pipelined INSIDE every wave (round 4): four waves and 128 rows per workgroup as in
// link_pass16_kernel, but iteration t runs P(t) on the matrix pipe while the SAME wave's vector unit works through the element-wise
// part of tile t - 1 (P -> g, packed), placed between the MFMAs by sched_group_barrier -- ~12 (6) vector instructions per MFMA gap --
// and then the contraction of tile t - 1.  Nothing of a tile's element-wise work is left to a partner wave (the pairing of an
// M wave with a V wave on one SIMD stretches the M half 2-2.8x, profiles/r04_link_phase_stamps.txt).  Tiles of F_other in a ring of
// three LDS buffers (P reads tile t, the contraction tile t - 1, the next one is written meanwhile), one bare barrier per tile.
template <int KP, int LINK>
__global__ __launch_bounds__(256, 2) void link_pass16sp_kernel(const uint32_t* __restrict__ Xbits, int64_t ldx, int64_t rows_pad,
                                                             const uint16_t* __restrict__ ARH, const uint16_t* __restrict__ ARM,
                                                             const uint16_t* __restrict__ ARL, const uint16_t* __restrict__ BRH,
                                                             const uint16_t* __restrict__ BRM, const uint16_t* __restrict__ BRL,
                                                             const uint16_t* __restrict__ BPH, const uint16_t* __restrict__ BPL,
                                                             float lam, int col_tiles, int col_tiles_per_block, float* __restrict__ num,
                                                             float* __restrict__ den, int64_t slab_stride) {
    constexpr int KS = KP / 16, NT = KP / 32;
    constexpr int ROWB = KP * 2, CH = ROWB / 16, ARR = 32 * ROWB, TILE_BYTES = 5 * ARR, PIECES = ARR / 16;
    __shared__ __attribute__((aligned(16))) char smem[3 * TILE_BYTES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
    const int jt0 = blockIdx.y * col_tiles_per_block;
    const int ntile = min(jt0 + col_tiles_per_block, col_tiles) - jt0;
    if (ntile <= 0) return;   // (block-uniform)

    u32x4 ah[KS], al[KS];   // this lane's row of F_self: fp16 hi / lo of the scaled factor
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int64_t off = (i0 + c) * KP + 16 * ks + 8 * h;
        ah[ks] = *reinterpret_cast<const u32x4*>(ARH + off);
        al[ks] = *reinterpret_cast<const u32x4*>(ARM + off);
    }
    const float pinv = reinterpret_cast<const float*>(ARL)[1] * reinterpret_cast<const float*>(BRL)[1];   // 1 / (S_self S_other)
    const float c1 = -lam * 1.44269504088896f * pinv, c0 = 0.5f * lam * 1.44269504088896f;   // on the SCALED product
    f32x16 o1[NT], o2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) { o1[nt][i] = 0.f; o2[nt][i] = 0.f; }

    const int pt = threadIdx.x;
    const bool p_on = pt < PIECES;
    const int p_row = pt / CH, p_chunk = pt % CH;
    const int p_lds = p_row * ROWB + ((p_chunk ^ (p_row % CH)) << 4);
    u32x4 stage[5];
    auto fetch = [&](int t) {   // (t clamped by the caller: a redundant fetch of the last tile keeps the loop body free of branches)
        if (!p_on) return;
        const int64_t jt = jt0 + t;
        const int64_t rm = (jt * 32 + p_row) * KP + p_chunk * 8, pm = jt * 32 * KP + pt * 8;
        stage[0] = *reinterpret_cast<const u32x4*>(BRH + rm);
        stage[1] = *reinterpret_cast<const u32x4*>(BRM + rm);
        stage[3] = *reinterpret_cast<const u32x4*>(BPH + pm);
        stage[4] = *reinterpret_cast<const u32x4*>(BPL + pm);
    };
    auto stash = [&](int buf) {
        if (!p_on) return;
        char* b = smem + buf * TILE_BYTES;
#pragma unroll
        for (int a = 0; a < 2; ++a) *reinterpret_cast<u32x4*>(b + a * ARR + p_lds) = stage[a];
        *reinterpret_cast<u32x4*>(b + 3 * ARR + pt * 16) = stage[3];
        *reinterpret_cast<u32x4*>(b + 4 * ARR + pt * 16) = stage[4];
    };
    auto barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
#define BMF_MF(a_, b_, acc_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8l, a_), __builtin_bit_cast(f16x8l, b_), acc_, 0, 0, 0)
#define BMF_MM(a_, b_, acc_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), acc_, 0, 0, 0)
    auto p_tile = [&](const char* tb, f32x16& p) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int off = c * ROWB + (((2 * ks + h) ^ (c % CH)) << 4);
            const u32x4 bh = *reinterpret_cast<const u32x4*>(tb + off);
            const u32x4 bl = *reinterpret_cast<const u32x4*>(tb + ARR + off);
            if (ks == 0) {
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                p = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8l, bl), __builtin_bit_cast(f16x8l, ah[0]), zero, 0, 0, 0);
            } else BMF_MF(bl, ah[ks], p);
            BMF_MF(bh, al[ks], p);
            BMF_MF(bh, ah[ks], p);
        }
    };
    u32x4 g1h[2], g1l[2], g2h[2], g2l[2];
    auto e_tile = [&](const f32x16& p, unsigned xw) {   // P -> g, packed (see link_pass16_kernel)
        const unsigned xs = xw >> (4 * h);
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const int b0 = (i & 3) + 8 * (i >> 2);
            const unsigned m0 = bit_mask(xs, b0), m1 = bit_mask(xs, b0 + 1);
            f32x2 ga, gb;
            if (LINK == BMF_LINK_SIGMOID) {
                float r0, d0, r1, d1;
                sigmoid_cell(p[i], c1, c0, r0, d0);
                sigmoid_cell(p[i + 1], c1, c0, r1, d1);
                ga = f32x2{__uint_as_float(__float_as_uint(d0) & m0), __uint_as_float(__float_as_uint(d1) & m1)};
                gb = f32x2{r0 * d0, r1 * d1};
            } else {
                const float r0 = p[i] > 0.f ? __builtin_amdgcn_rcpf(p[i] * pinv) : 0.f, r1 = p[i + 1] > 0.f ? __builtin_amdgcn_rcpf(p[i + 1] * pinv) : 0.f;
                ga = f32x2{__uint_as_float(__float_as_uint(r0) & m0), __uint_as_float(__float_as_uint(r1) & m1)};
                gb = f32x2{0.f, 0.f};
            }
            const int q = i >> 3, w = (i & 7) >> 1;
            unsigned wh, wl;
            split_pair(ga, wh, wl);
            g1h[q][w] = wh; g1l[q][w] = wl;
            if (LINK == BMF_LINK_SIGMOID) {
                split_pair(gb, wh, wl);
                g2h[q][w] = wh; g2l[q][w] = wl;
            }
        }
    };
    auto c_tile = [&](const char* tb) {   // out += g B over the tile whose permuted arrays sit at tb
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            u32x4 vh[NT], vl[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int off = (((q * 2 + h) * KP) + 32 * nt + c) * 16;
                vh[nt] = *reinterpret_cast<const u32x4*>(tb + 3 * ARR + off);
                vl[nt] = *reinterpret_cast<const u32x4*>(tb + 4 * ARR + off);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                BMF_MM(g1l[q], vh[nt], o1[nt]); BMF_MM(g1h[q], vl[nt], o1[nt]); BMF_MM(g1h[q], vh[nt], o1[nt]);
                if (LINK == BMF_LINK_SIGMOID) { BMF_MM(g2l[q], vh[nt], o2[nt]); BMF_MM(g2h[q], vl[nt], o2[nt]); BMF_MM(g2h[q], vh[nt], o2[nt]); }
            }
        }
    };

    // prologue: tiles 0 and 1 into the ring, tile 2 into registers; P(0)
    fetch(0); stash(0);
    fetch(min(1, ntile - 1)); stash(1);
    fetch(min(2, ntile - 1));
    barrier();
    const uint32_t* xrow = Xbits + (i0 + c) * ldx + jt0;
    f32x16 p_prev, p_cur;
    unsigned xw_prev = xrow[0];
    p_tile(smem, p_prev);
    for (int t = 1; t < ntile; ++t) {
        // tile t + 1 goes into the buffer tile t - 2 left (its last reader, the contraction of iteration t - 1, is behind the barrier);
        // tile t + 2 is requested.  Past the end the last tile is fetched and stashed again into a buffer nobody reads.
        stash((t + 1) % 3);
        fetch(min(t + 2, ntile - 1));
        const unsigned xw_cur = xrow[t];
        // P(t) and the element-wise part of tile t - 1, woven by hand: one MFMA, a third of a cell pair's vector work (sigmoid of one cell,
        // sigmoid of the other, the two bf16 splits), fenced so that the scheduler keeps the order (left to itself, or steered by
        // sched_group_barrier, it issued the MFMAs first and the ~300 vector instructions after them).  3 KS MFMAs over 24 sub-steps.
        {
            const char* tb = smem + (t % 3) * TILE_BYTES;
            const unsigned xs = xw_prev >> (4 * h);
            u32x4 bh, bl, nh, nl;
            auto operands = [&](int ks, u32x4& oh, u32x4& ol) {
                const int off = c * ROWB + (((2 * ks + h) ^ (c % CH)) << 4);
                oh = *reinterpret_cast<const u32x4*>(tb + off);
                ol = *reinterpret_cast<const u32x4*>(tb + ARR + off);
            };
            operands(0, bh, bl);
            float dd[2], sd[2];
            auto cell = [&](int i, int e) {        // sigmoid parts (or the reciprocal) of cell i + e, masked by its X bit
                const int b0 = (i & 3) + 8 * (i >> 2) + e;
                const unsigned mk = bit_mask(xs, b0);
                if (LINK == BMF_LINK_SIGMOID) {
                    float r, d;
                    sigmoid_cell(p_prev[i + e], c1, c0, r, d);
                    dd[e] = __uint_as_float(__float_as_uint(d) & mk);
                    sd[e] = r * d;
                } else {
                    const float r = p_prev[i + e] > 0.f ? __builtin_amdgcn_rcpf(p_prev[i + e] * pinv) : 0.f;
                    dd[e] = __uint_as_float(__float_as_uint(r) & mk);
                    sd[e] = 0.f;
                }
            };
            auto pack = [&](int i) {
                const int q = i >> 3, w = (i & 7) >> 1;
                unsigned wh, wl;
                split_pair(f32x2{dd[0], dd[1]}, wh, wl);
                g1h[q][w] = wh; g1l[q][w] = wl;
                if (LINK == BMF_LINK_SIGMOID) {
                    split_pair(f32x2{sd[0], sd[1]}, wh, wl);
                    g2h[q][w] = wh; g2l[q][w] = wl;
                }
            };
            constexpr int NM = 3 * KS;             // MFMAs of P
            constexpr int HALF = 12;               // 4 cell pairs x 3 sub-steps: cells 0..7 = k-chunk q = 0 of the contraction
            auto sub_step = [&](int st) {
                const int pair = st / 3, sub = st % 3;
                if (sub == 0) cell(2 * pair, 0);
                else if (sub == 1) cell(2 * pair, 1);
                else pack(2 * pair);
            };
            // first half of the element-wise work (cells 0..7) between the MFMAs of P(t)
#pragma unroll
            for (int st = 0; st < HALF; ++st) {
#pragma unroll
                for (int j = st * NM / HALF; j < (st + 1) * NM / HALF; ++j) {
                    const int ks = j / 3, w3 = j % 3;
                    if (w3 == 0 && ks + 1 < KS) operands(ks + 1, nh, nl);   // the next k-step's operands: requested three MFMAs ahead
                    if (j == 0) {
                        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        p_cur = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8l, bl), __builtin_bit_cast(f16x8l, ah[0]), zero, 0, 0, 0);
                    } else if (w3 == 0) BMF_MF(bl, ah[ks], p_cur);
                    else if (w3 == 1) BMF_MF(bh, al[ks], p_cur);
                    else { BMF_MF(bh, ah[ks], p_cur); if (ks + 1 < KS) { bh = nh; bl = nl; } }
                }
                __builtin_amdgcn_sched_barrier(0);
                sub_step(st);
                __builtin_amdgcn_sched_barrier(0);
            }
            // second half (cells 8..15 = chunk q = 1) between the MFMAs of the contraction's chunk q = 0, whose g is complete
            const char* tbc = smem + ((t - 1) % 3) * TILE_BYTES;
            u32x4 vh[2][NT], vl[2][NT];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int off = (((q * 2 + h) * KP) + 32 * nt + c) * 16;
                    vh[q][nt] = *reinterpret_cast<const u32x4*>(tbc + 3 * ARR + off);
                    vl[q][nt] = *reinterpret_cast<const u32x4*>(tbc + 4 * ARR + off);
                }
            constexpr int PER = LINK == BMF_LINK_SIGMOID ? 6 : 3, NCQ = NT * PER;   // MFMAs of one chunk of the contraction
            auto c_mfma = [&](int q, int idx) {
                const int nt = idx / PER, w = idx % PER;
                if (w == 0) BMF_MM(g1l[q], vh[q][nt], o1[nt]);
                else if (w == 1) BMF_MM(g1h[q], vl[q][nt], o1[nt]);
                else if (w == 2) BMF_MM(g1h[q], vh[q][nt], o1[nt]);
                else if (w == 3) BMF_MM(g2l[q], vh[q][nt], o2[nt]);
                else if (w == 4) BMF_MM(g2h[q], vl[q][nt], o2[nt]);
                else BMF_MM(g2h[q], vh[q][nt], o2[nt]);
            };
#pragma unroll
            for (int st = 0; st < HALF; ++st) {
#pragma unroll
                for (int j = st * NCQ / HALF; j < (st + 1) * NCQ / HALF; ++j) c_mfma(0, j);
                __builtin_amdgcn_sched_barrier(0);
                sub_step(HALF + st);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < NCQ; ++j) c_mfma(1, j);
        }
        p_prev = p_cur;
        xw_prev = xw_cur;
        barrier();
    }
    e_tile(p_prev, xw_prev);
    c_tile(smem + ((ntile - 1) % 3) * TILE_BYTES);
#undef BMF_MM
#undef BMF_MF
    float* on = num + (int64_t)blockIdx.y * slab_stride;
    float* od = den ? den + (int64_t)blockIdx.y * slab_stride : nullptr;
    const float oscale = LINK == BMF_LINK_SIGMOID ? lam : 1.0f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t row = i0 + link_jr(i, h);
            on[row * KP + 32 * nt + c] = oscale * o1[nt][i];
            if (LINK == BMF_LINK_SIGMOID && od) od[row * KP + 32 * nt + c] = oscale * o2[nt][i];
        }
}

// scalar sums of the same tile pass: sums[0] += sum |x - f(p)|, sums[1] += sum (x - f(p))^2 with f = the link (identity for
// KL), sums[2] += sum (x log(x / p) - x + p) with 0 log 0 = 0 (the KL objective, WNMF.py:143-145)
template <int KP, int LINK>
__global__ __launch_bounds__(256) void link_sums_kernel(const uint32_t* __restrict__ Xbits, int64_t ldx, int rows, int cols,
                                                         const float* __restrict__ A, const float* __restrict__ B,
                                                         float lam, int col_tiles_per_block, const uint32_t* __restrict__ Obits, double* __restrict__ sums) {
    constexpr int KH = KP / 2;
    __shared__ double red[4][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
    const int col_tiles = (cols + 31) / 32;
    const int jt0 = blockIdx.y * col_tiles_per_block;
    const int jt1 = min(jt0 + col_tiles_per_block, col_tiles);
    float a[KH];
    const float* ap = A + (i0 + c) * KP + KH * h;
#pragma unroll
    for (int s = 0; s < KH; s += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(ap + s);
        a[s] = v[0]; a[s + 1] = v[1]; a[s + 2] = v[2]; a[s + 3] = v[3];
    }
    const bool row_ok = (i0 + c) < rows;
    double s_abs = 0.0, s_sq = 0.0, s_kl = 0.0;
    for (int jt = jt0; jt < jt1; ++jt) {
        const int64_t j0 = (int64_t)jt * 32;
        float b[KH];
        const float* bp = B + (j0 + c) * KP + KH * h;
#pragma unroll
        for (int s = 0; s < KH; s += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(bp + s);
            b[s] = v[0]; b[s + 1] = v[1]; b[s + 2] = v[2]; b[s + 3] = v[3];
        }
        const unsigned xw = Xbits[(i0 + c) * ldx + jt];
        const unsigned ow = Obits ? Obits[(i0 + c) * ldx + jt] : 0xffffffffu;  // observed cells (KL objective only)
        f32x16 p;
#pragma unroll
        for (int i = 0; i < 16; ++i) p[i] = 0.f;
#pragma unroll
        for (int s = 0; s < KH; ++s) p = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s], a[s], p, 0, 0, 0);
        float t_abs = 0.f, t_sq = 0.f, t_kl = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int jr = (i & 3) + 8 * (i >> 2) + 4 * h;
            const bool ok = row_ok && (j0 + jr) < cols;
            const float x = (float)((xw >> jr) & 1u);
            float f = p[i];
            if (LINK == BMF_LINK_SIGMOID) {
                float d;
                sigmoid_parts(lam * (p[i] - 0.5f), f, d);
            }
            const float r = ok ? x - f : 0.f;
            t_abs += fabsf(r);
            t_sq = fmaf(r, r, t_sq);
            if (LINK == BMF_LINK_KL && ok && ((ow >> jr) & 1u)) t_kl += (x != 0.f) ? (p[i] - 1.0f - __logf(fmaxf(p[i], 1e-37f))) : p[i];
        }
        s_abs += (double)t_abs;
        s_sq += (double)t_sq;
        s_kl += (double)t_kl;
    }
    s_abs = wave_sum(s_abs);
    s_sq = wave_sum(s_sq);
    s_kl = wave_sum(s_kl);
    if (lane == 0) { red[wave][0] = s_abs; red[wave][1] = s_sq; red[wave][2] = s_kl; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int q = threadIdx.x;
        atomicAdd(&sums[q], ((red[0][q] + red[1][q]) + red[2][q]) + red[3][q]);
    }
}

// bmf_link_sums on the bf16 MFMA: P from the three-addend splits (six products, fp32 accuracy), the rest as above
template <int KP, int LINK>
__global__ __launch_bounds__(256) void link_sums16_kernel(const uint32_t* __restrict__ Xbits, int64_t ldx, int rows, int cols,
                                                           const uint16_t* __restrict__ ARH, const uint16_t* __restrict__ ARM,
                                                           const uint16_t* __restrict__ ARL, const uint16_t* __restrict__ BRH,
                                                           const uint16_t* __restrict__ BRM, const uint16_t* __restrict__ BRL,
                                                           float lam, int col_tiles_per_block, const uint32_t* __restrict__ Obits, double* __restrict__ sums) {
    constexpr int KS = KP / 16;
    constexpr int ROWB = KP * 2, CH = ROWB / 16, ARR = 32 * ROWB, TILE_BYTES = 2 * ARR, PIECES = ARR / 16;  // fp16 hi / lo of the other factor's 32 rows
    __shared__ __attribute__((aligned(16))) char smem[2 * TILE_BYTES];
    __shared__ double red[4][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
    const int col_tiles = (cols + 31) / 32;
    const int jt0 = blockIdx.y * col_tiles_per_block;
    const int jt1 = min(jt0 + col_tiles_per_block, col_tiles);
    u32x4 ah[KS], al[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int64_t off = (i0 + c) * KP + 16 * ks + 8 * h;
        ah[ks] = *reinterpret_cast<const u32x4*>(ARH + off);
        al[ks] = *reinterpret_cast<const u32x4*>(ARM + off);
    }
    const float pinv = reinterpret_cast<const float*>(ARL)[1] * reinterpret_cast<const float*>(BRL)[1];   // 1 / (S_U S_V): P comes out scaled
    const float c1 = -lam * 1.44269504088896f, c0 = 0.5f * lam * 1.44269504088896f;   // t = -lam (p - 1/2) log2 e
    const bool row_ok = (i0 + c) < rows;
    double s_abs = 0.0, s_sq = 0.0, s_kl = 0.0;
    const int pt = threadIdx.x;
    const bool p_on = pt < PIECES;
    const int p_row = pt / CH, p_chunk = pt % CH;
    const int p_lds = p_row * ROWB + ((p_chunk ^ (p_row % CH)) << 4);
    u32x4 stage[2];
    auto fetch = [&](int jt) {
        if (!p_on) return;
        const int64_t rm = ((int64_t)jt * 32 + p_row) * KP + p_chunk * 8;
        stage[0] = *reinterpret_cast<const u32x4*>(BRH + rm);
        stage[1] = *reinterpret_cast<const u32x4*>(BRM + rm);
    };
    auto stash = [&](int buf) {
        if (!p_on) return;
#pragma unroll
        for (int a = 0; a < 2; ++a) *reinterpret_cast<u32x4*>(smem + buf * TILE_BYTES + a * ARR + p_lds) = stage[a];
    };
    if (jt0 < jt1) {
        fetch(jt0);
        stash(0);
    }
    __syncthreads();
    int cur = 0;
    for (int jt = jt0; jt < jt1; ++jt) {
        const int64_t j0 = (int64_t)jt * 32;
        if (jt + 1 < jt1) fetch(jt + 1);
        const char* tb = smem + cur * TILE_BYTES;
        u32x4 bh[KS], bl[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int off = c * ROWB + (((2 * ks + h) ^ (c % CH)) << 4);
            bh[ks] = *reinterpret_cast<const u32x4*>(tb + off);
            bl[ks] = *reinterpret_cast<const u32x4*>(tb + ARR + off);
        }
        const unsigned xw = Xbits[(i0 + c) * ldx + jt];
        const unsigned ow = Obits ? Obits[(i0 + c) * ldx + jt] : 0xffffffffu;  // observed cells (KL objective only)
        f32x16 p;
#pragma unroll
        for (int i = 0; i < 16; ++i) p[i] = 0.f;
#define BMF_MF(a_, b_, acc_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8l, a_), __builtin_bit_cast(f16x8l, b_), acc_, 0, 0, 0)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {   // smallest first: lo hi', hi lo', hi hi'
            BMF_MF(bl[ks], ah[ks], p);
            BMF_MF(bh[ks], al[ks], p);
            BMF_MF(bh[ks], ah[ks], p);
        }
#undef BMF_MF
        float t_abs = 0.f, t_sq = 0.f, t_kl = 0.f;
        // Interior tiles (all 32 x 32 cells inside the matrix: all but the last row block and the last column tile) need no masks; the
        // cell of register i is bit (i & 3) + 8 (i >> 2) of xw >> 4 h, and x as a float is that bit spread over the word & 1.0f.
        const bool interior = i0 + 32 <= rows && j0 + 32 <= cols;   // wave-uniform
        const unsigned xs = xw >> (4 * h), os = ow >> (4 * h);
        auto cells = [&](auto edge_tag) {
            constexpr bool EDGE = decltype(edge_tag)::value;
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const int b0 = (i & 3) + 8 * (i >> 2);
                const f32x2 pv = {p[i] * pinv, p[i + 1] * pinv};
                const f32x2 x = {__uint_as_float(bit_mask(xs, b0) & 0x3f800000u),
                                 __uint_as_float(bit_mask(xs, b0 + 1) & 0x3f800000u)};
                f32x2 f = pv;
                if (LINK == BMF_LINK_SIGMOID) {   // sigmoid(s) = 1 / (1 + 2^min(-s log2 e, 100)), see sigmoid_cell
                    f = f32x2{__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fminf(fmaf(pv[0], c1, c0), 100.f))),
                              __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fminf(fmaf(pv[1], c1, c0), 100.f)))};
                }
                f32x2 r = {x[0] - f[0], x[1] - f[1]};
                if (EDGE) {
                    const int jr = link_jr(i, h);   // (i, i + 1) -> jr, jr + 1
                    if (!(row_ok && (j0 + jr) < cols)) r[0] = 0.f;
                    if (!(row_ok && (j0 + jr + 1) < cols)) r[1] = 0.f;
                }
                t_abs += fabsf(r[0]) + fabsf(r[1]);
                t_sq = fmaf(r[1], r[1], fmaf(r[0], r[0], t_sq));
                if (LINK == BMF_LINK_KL) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        bool on = (os >> (b0 + e)) & 1u;
                        if (EDGE) on = on && row_ok && (j0 + link_jr(i + e, h)) < cols;
                        if (on) t_kl += (x[e] != 0.f) ? (pv[e] - 1.0f - __logf(fmaxf(pv[e], 1e-37f))) : pv[e];
                    }
                }
            }
        };
        if (interior) cells(std::false_type{});
        else cells(std::true_type{});
        s_abs += (double)t_abs;
        s_sq += (double)t_sq;
        s_kl += (double)t_kl;
        if (jt + 1 < jt1) stash(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    s_abs = wave_sum(s_abs);
    s_sq = wave_sum(s_sq);
    s_kl = wave_sum(s_kl);
    if (lane == 0) { red[wave][0] = s_abs; red[wave][1] = s_sq; red[wave][2] = s_kl; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int q = threadIdx.x;
        atomicAdd(&sums[q], ((red[0][q] + red[1][q]) + red[2][q]) + red[3][q]);
    }
}

// out[r][c] = sum_i F[i][c] for every row r < out_rows (the KL denominator 1 V, WNMF.py:116,124).  Column sums in two steps, fp64, fixed
// order: partial[b][c] over a row range per block (all CUs busy, coalesced), then one block adds the partials; then the fill.
// (Round 2 summed with kp / 4 blocks, each walking all rows: 408 us for the 100k x 64 factor, 13 % of a WNMF-KL iteration.)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ F, int64_t rows, int kp, int64_t rows_per_block,
                                                              double* __restrict__ partial) {
    __shared__ double sh[256];
    const int c = threadIdx.x % kp, sub = threadIdx.x / kp, nsub = 256 / kp;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, rows);
    double acc = 0.0;
    for (int64_t r = r0 + sub; r < r1; r += nsub) acc += (double)F[r * kp + c];
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (sub == 0) {
        double t = 0.0;
        for (int q = 0; q < nsub; ++q) t += sh[q * kp + c];
        partial[(int64_t)blockIdx.x * kp + c] = t;
    }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const double* __restrict__ partial, int blocks, int kp, float* __restrict__ colsum) {
    __shared__ double sh[256];
    const int c = threadIdx.x % kp, sub = threadIdx.x / kp, nsub = 256 / kp;   // group `sub` adds a contiguous range of partials
    const int per = (blocks + nsub - 1) / nsub;
    const int b0 = sub * per, b1 = min(b0 + per, blocks);
    double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
    int b = b0;
    for (; b + 4 <= b1; b += 4) {   // four loads in flight
        t0 += partial[(int64_t)b * kp + c];
        t1 += partial[(int64_t)(b + 1) * kp + c];
        t2 += partial[(int64_t)(b + 2) * kp + c];
        t3 += partial[(int64_t)(b + 3) * kp + c];
    }
    for (; b < b1; ++b) t0 += partial[(int64_t)b * kp + c];
    sh[threadIdx.x] = (t0 + t1) + (t2 + t3);
    __syncthreads();
    if (sub == 0) {
        double t = 0.0;
        for (int q = 0; q < nsub; ++q) t += sh[q * kp + c];
        colsum[c] = (float)t;
    }
}

__global__ __launch_bounds__(256) void fill_rows_kernel(const float* __restrict__ vec, int kp, int64_t total, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) out[i] = vec[i % kp];
}

// Column splits of a pass: workgroups = row blocks x splits, 512 of them resident (two per CU).  Every workgroup of a launch takes
// the same time, so a launch runs in whole rounds: 1564 workgroups (100k rows, 2 splits) = 3.05 rounds cost 4, 1099 (20k rows, 7
// splits) = 2.15 rounds cost 3 -- a quarter of the pass idle (round 4, profiles/r04_pmc_link.md).  Fewest splits (each is a slab of
// rows x k floats to write and re-read) whose last round is >= 90 % full, else the fullest.
int splits_for(int64_t rows, int64_t cols) {
    const int64_t row_blocks = (rows + 127) / 128, col_tiles = (cols + 31) / 32;
    const int64_t smax = col_tiles / 8 < 1 ? 1 : (col_tiles / 8 > 16 ? 16 : col_tiles / 8);
    int best = 1;
    double best_fill = 0.0;
    for (int64_t s = 1; s <= smax; ++s) {
        const int64_t per = (col_tiles + s - 1) / s;
        if ((col_tiles + per - 1) / per != s) continue;   // this many splits leave one empty
        const int64_t wgs = row_blocks * s, rounds = (wgs + 511) / 512;
        const double fill = (double)wgs / (double)(rounds * 512);
        if (fill >= 0.9) return (int)s;
        if (fill > best_fill) { best_fill = fill; best = (int)s; }
    }
    return best;
}

}  // namespace

extern "C" int bmf_link_splits(int64_t rows, int64_t cols) {
    if (rows < 1 || cols < 1) {
        bmf_set_error("bmf_link_splits: rows and cols must be positive");
        return BMF_ERR_BAD_ARG;
    }
    return splits_for(rows, cols);
}

extern "C" int bmf_link_pass(const uint32_t* Xbits, int64_t rows_pad, int64_t ldx, int32_t rows, int32_t cols,
                             const float* Fself, const float* Fother, int64_t other_pad, int kp, int link, double lamda,
                             float* num, float* den, int64_t slab_stride, int splits, void* stream) {
    BMF_REQUIRE(Xbits && Fself && Fother && num, "bmf_link_pass: null pointer");
    BMF_REQUIRE(link == BMF_LINK_SIGMOID || link == BMF_LINK_KL, "bmf_link_pass: link must be BMF_LINK_SIGMOID or BMF_LINK_KL");
    BMF_REQUIRE(link != BMF_LINK_SIGMOID || den, "bmf_link_pass: the sigmoid link needs a den buffer");
    BMF_REQUIRE(rows >= 1 && cols >= 1 && rows <= rows_pad && rows_pad % 128 == 0, "bmf_link_pass: bad rows/rows_pad");
    BMF_REQUIRE(cols <= other_pad && other_pad % 32 == 0 && ldx * 32 >= cols, "bmf_link_pass: other_pad / ldx do not cover cols");
    BMF_REQUIRE(kp == 32 || kp == 64, "bmf_link_pass: kp must be 32 or 64");
    BMF_REQUIRE(splits == splits_for(rows, cols), "bmf_link_pass: splits=%d, this shape needs %d (bmf_link_splits)", splits,
                splits_for(rows, cols));
    BMF_REQUIRE(slab_stride >= rows_pad * kp, "bmf_link_pass: slab_stride too small");
    BMF_REQUIRE(bmf_aligned16(Fself) && bmf_aligned16(Fother), "bmf_link_pass: factors must be 16-byte aligned");
    const int col_tiles = (cols + 31) / 32;
    const int per = (col_tiles + splits - 1) / splits;
    dim3 grid((unsigned)(rows_pad / 128), (unsigned)splits), block(256);
    hipStream_t s = (hipStream_t)stream;
    const float lam = (float)lamda;
#define BMF_LINK_CASE(KP_, L_)                                                                                         \
    if (kp == KP_ && link == L_)                                                                                       \
        BMF_LAUNCH((link_pass_kernel<KP_, L_>), grid, block, 0, s, Xbits, ldx, rows, cols, Fself, Fother, lam, per, num, den, slab_stride);
    BMF_LINK_CASE(32, BMF_LINK_SIGMOID) BMF_LINK_CASE(64, BMF_LINK_SIGMOID) BMF_LINK_CASE(32, BMF_LINK_KL) BMF_LINK_CASE(64, BMF_LINK_KL)
#undef BMF_LINK_CASE
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_link_split(const float* F, int64_t rows_pad, int kp, uint16_t* ws, void* stream) {
    BMF_REQUIRE(F && ws, "bmf_link_split: null pointer");
    BMF_REQUIRE(rows_pad > 0 && rows_pad % 32 == 0 && (kp == 32 || kp == 64), "bmf_link_split: rows_pad must be a multiple of 32, kp 32 or 64");
    BMF_REQUIRE(bmf_aligned16(ws), "bmf_link_split: ws must be 16-byte aligned");
    const int64_t n = rows_pad * kp;
    const int64_t blocks = (n + 255) / 256;
    // the power-of-two scale of the fp16 operands: max |F| first (one word in the third array, which the split no longer fills)
    BMF_HIP_CHECK(hipMemsetAsync(ws + 2 * n, 0, 16, (hipStream_t)stream));
    BMF_LAUNCH(link_max_kernel, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(256), 0, (hipStream_t)stream, F, n,
               reinterpret_cast<unsigned*>(ws + 2 * n) + 2);
    BMF_LAUNCH(link_split_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, F, rows_pad, kp, ws,
               ws + n, ws + 2 * n, ws + 3 * n, ws + 4 * n, nullptr);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_link_split_pair(const float* A, int64_t a_pad, const float* B, int64_t b_pad, int kp, uint16_t* wsA, uint16_t* wsB, void* stream) {
    BMF_REQUIRE(A && B && wsA && wsB, "bmf_link_split_pair: null pointer");
    BMF_REQUIRE(a_pad > 0 && a_pad % 32 == 0 && b_pad > 0 && b_pad % 32 == 0 && (kp == 32 || kp == 64), "bmf_link_split_pair: rows must be multiples of 32, kp 32 or 64");
    BMF_REQUIRE(bmf_aligned16(wsA) && bmf_aligned16(wsB), "bmf_link_split_pair: workspaces must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int64_t na = a_pad * kp, nb = b_pad * kp;
    unsigned* wa = reinterpret_cast<unsigned*>(wsA + 2 * na);
    unsigned* wb = reinterpret_cast<unsigned*>(wsB + 2 * nb);
    const size_t head = (size_t)(4 + 2 * kp) * sizeof(unsigned);
    BMF_HIP_CHECK(hipMemsetAsync(wa, 0, head, s));
    BMF_HIP_CHECK(hipMemsetAsync(wb, 0, head, s));
    const int rows_per_block = 256 / kp;
    auto cm_blocks = [&](int64_t rows) { const int64_t b = (rows + rows_per_block - 1) / rows_per_block; return (unsigned)(b < 256 ? b : 256); };
    BMF_LAUNCH(link_colmax_kernel, dim3(cm_blocks(a_pad)), dim3(256), 0, s, A, a_pad, kp, wa + 4);
    BMF_LAUNCH(link_colmax_kernel, dim3(cm_blocks(b_pad)), dim3(256), 0, s, B, b_pad, kp, wb + 4);
    BMF_LAUNCH(link_pair_scales_kernel, dim3(1), dim3(64), 0, s, wa, wb, kp);
    const int64_t ba = (na + 255) / 256, bb = (nb + 255) / 256;
    BMF_LAUNCH(link_split_kernel, dim3((unsigned)(ba < 4096 ? ba : 4096)), dim3(256), 0, s, A, a_pad, kp, wsA, wsA + na, wsA + 2 * na, wsA + 3 * na,
               wsA + 4 * na, reinterpret_cast<const float*>(wa) + 4 + kp);
    BMF_LAUNCH(link_split_kernel, dim3((unsigned)(bb < 4096 ? bb : 4096)), dim3(256), 0, s, B, b_pad, kp, wsB, wsB + nb, wsB + 2 * nb, wsB + 3 * nb,
               wsB + 4 * nb, reinterpret_cast<const float*>(wb) + 4 + kp);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_link_pass16(const uint32_t* Xbits, int64_t rows_pad, int64_t ldx, int32_t rows, int32_t cols,
                               const uint16_t* ws_self, const uint16_t* ws_other, int64_t other_pad, int kp, int link,
                               double lamda, float* num, float* den, int64_t slab_stride, int splits, void* stream) {
    BMF_REQUIRE(Xbits && ws_self && ws_other && num, "bmf_link_pass16: null pointer");
    BMF_REQUIRE(link == BMF_LINK_SIGMOID || link == BMF_LINK_KL, "bmf_link_pass16: link must be BMF_LINK_SIGMOID or BMF_LINK_KL");
    BMF_REQUIRE(link != BMF_LINK_SIGMOID || den, "bmf_link_pass16: the sigmoid link needs a den buffer");
    BMF_REQUIRE(rows >= 1 && cols >= 1 && rows <= rows_pad && rows_pad % 128 == 0, "bmf_link_pass16: bad rows/rows_pad");
    BMF_REQUIRE(cols <= other_pad && other_pad % 32 == 0 && ldx * 32 >= cols, "bmf_link_pass16: other_pad / ldx do not cover cols");
    BMF_REQUIRE(kp == 32 || kp == 64, "bmf_link_pass16: kp must be 32 or 64");
    BMF_REQUIRE(splits == splits_for(rows, cols), "bmf_link_pass16: splits=%d, this shape needs %d (bmf_link_splits)", splits,
                splits_for(rows, cols));
    BMF_REQUIRE(slab_stride >= rows_pad * kp, "bmf_link_pass16: slab_stride too small");
    BMF_REQUIRE(bmf_aligned16(ws_self) && bmf_aligned16(ws_other), "bmf_link_pass16: workspaces must be 16-byte aligned");
    const int col_tiles = (cols + 31) / 32;
    const int per = (col_tiles + splits - 1) / splits;
    dim3 grid((unsigned)(rows_pad / 128), (unsigned)splits), block(256);
    hipStream_t s = (hipStream_t)stream;
    const float lam = (float)lamda;
    const int64_t ns = rows_pad * kp, no = other_pad * kp;
    const uint16_t *ARH = ws_self, *ARM = ws_self + ns, *ARL = ws_self + 2 * ns;
    const uint16_t *BRH = ws_other, *BRM = ws_other + no, *BRL = ws_other + 2 * no, *BPH = ws_other + 3 * no, *BPL = ws_other + 4 * no;
    // the pass software-pipelined inside every wave (round 4): the element-wise part woven between the MFMAs of P and of the contraction's
    // first chunk.  (The two-wave-group form it replaced -- eight waves, opposite phases, 5.41 vs 5.19 ms per sigmoid update pair and
    // 4.79 vs 3.55 ms for KL -- was removed in round 5; HISTORY.md, profiles/r04_pmc_link.md.)
#define BMF_LINK_CASE(KP_, L_)                                                                                         \
    if (kp == KP_ && link == L_)                                                                                       \
        BMF_LAUNCH((link_pass16sp_kernel<KP_, L_>), grid, block, 0, s, Xbits, ldx, rows_pad, ARH, ARM, ARL, BRH, BRM, BRL, BPH, BPL, lam, \
                   col_tiles, per, num, den, slab_stride);
    BMF_LINK_CASE(32, BMF_LINK_SIGMOID) BMF_LINK_CASE(64, BMF_LINK_SIGMOID) BMF_LINK_CASE(32, BMF_LINK_KL) BMF_LINK_CASE(64, BMF_LINK_KL)
#undef BMF_LINK_CASE
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_link_sums(const uint32_t* Xbits, int64_t m_pad, int64_t ldx, int32_t m, int32_t n, const float* U,
                             const float* V, int64_t n_pad, int kp, int link, double lamda, const uint32_t* Obits, double* sums, void* stream) {
    BMF_REQUIRE(Xbits && U && V && sums, "bmf_link_sums: null pointer");
    BMF_REQUIRE(link == BMF_LINK_SIGMOID || link == BMF_LINK_KL, "bmf_link_sums: link must be BMF_LINK_SIGMOID or BMF_LINK_KL");
    BMF_REQUIRE(m >= 1 && n >= 1 && m <= m_pad && m_pad % 128 == 0 && n <= n_pad && n_pad % 32 == 0 && ldx * 32 >= n,
                "bmf_link_sums: bad shape");
    BMF_REQUIRE(kp == 32 || kp == 64, "bmf_link_sums: kp must be 32 or 64");
    const int splits = splits_for(m, n);
    const int col_tiles = (n + 31) / 32;
    const int per = (col_tiles + splits - 1) / splits;
    dim3 grid((unsigned)((m + 127) / 128), (unsigned)splits), block(256);
    hipStream_t s = (hipStream_t)stream;
    const float lam = (float)lamda;
#define BMF_LINK_CASE(KP_, L_) \
    if (kp == KP_ && link == L_) BMF_LAUNCH((link_sums_kernel<KP_, L_>), grid, block, 0, s, Xbits, ldx, m, n, U, V, lam, per, Obits, sums);
    BMF_LINK_CASE(32, BMF_LINK_SIGMOID) BMF_LINK_CASE(64, BMF_LINK_SIGMOID) BMF_LINK_CASE(32, BMF_LINK_KL) BMF_LINK_CASE(64, BMF_LINK_KL)
#undef BMF_LINK_CASE
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_link_sums16(const uint32_t* Xbits, int64_t m_pad, int64_t ldx, int32_t m, int32_t n, const uint16_t* wsU,
                               const uint16_t* wsV, int64_t n_pad, int kp, int link, double lamda, const uint32_t* Obits, double* sums, void* stream) {
    BMF_REQUIRE(Xbits && wsU && wsV && sums, "bmf_link_sums16: null pointer");
    BMF_REQUIRE(link == BMF_LINK_SIGMOID || link == BMF_LINK_KL, "bmf_link_sums16: link must be BMF_LINK_SIGMOID or BMF_LINK_KL");
    BMF_REQUIRE(m >= 1 && n >= 1 && m <= m_pad && m_pad % 128 == 0 && n <= n_pad && n_pad % 32 == 0 && ldx * 32 >= n,
                "bmf_link_sums16: bad shape");
    BMF_REQUIRE(kp == 32 || kp == 64, "bmf_link_sums16: kp must be 32 or 64");
    BMF_REQUIRE(bmf_aligned16(wsU) && bmf_aligned16(wsV), "bmf_link_sums16: workspaces must be 16-byte aligned");
    const int splits = splits_for(m, n);
    const int col_tiles = (n + 31) / 32;
    const int per = (col_tiles + splits - 1) / splits;
    dim3 grid((unsigned)((m + 127) / 128), (unsigned)splits), block(256);
    hipStream_t s = (hipStream_t)stream;
    const float lam = (float)lamda;
    const int64_t nu = m_pad * kp, nv = n_pad * kp;
#define BMF_LINK_CASE(KP_, L_)                                                                                                    \
    if (kp == KP_ && link == L_)                                                                                                  \
        BMF_LAUNCH((link_sums16_kernel<KP_, L_>), grid, block, 0, s, Xbits, ldx, m, n, wsU, wsU + nu, wsU + 2 * nu, wsV, wsV + nv, \
                   wsV + 2 * nv, lam, per, Obits, sums);
    BMF_LINK_CASE(32, BMF_LINK_SIGMOID) BMF_LINK_CASE(64, BMF_LINK_SIGMOID) BMF_LINK_CASE(32, BMF_LINK_KL) BMF_LINK_CASE(64, BMF_LINK_KL)
#undef BMF_LINK_CASE
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_colsum_fill(const float* F, int64_t rows, int kp, float* colsum, float* out, int64_t out_rows, void* stream) {
    BMF_REQUIRE(F && colsum && out, "bmf_colsum_fill: null pointer");
    BMF_REQUIRE(rows >= 1 && out_rows >= 1 && (kp == 32 || kp == 64), "bmf_colsum_fill: bad shape");
    hipStream_t s = (hipStream_t)stream;
    BMF_REQUIRE(bmf_aligned16(out), "bmf_colsum_fill: out must be 16-byte aligned (its head is the scratch of the partial sums)");
    // partial sums live in `out` until the fill overwrites it: blocks * kp doubles <= out_rows * kp floats
    int64_t pblocks = (rows + 63) / 64;
    if (pblocks > 512) pblocks = 512;
    if (pblocks > out_rows / 2) pblocks = out_rows / 2;
    if (pblocks < 1) pblocks = 1;
    const int64_t rpb = (rows + pblocks - 1) / pblocks;
    pblocks = (rows + rpb - 1) / rpb;
    BMF_REQUIRE(pblocks * 2 <= out_rows || (pblocks == 1 && out_rows >= 2), "bmf_colsum_fill: out is too small to hold the partial sums");
    BMF_LAUNCH(colsum_partial_kernel, dim3((unsigned)pblocks), dim3(256), 0, s, F, rows, kp, rpb, reinterpret_cast<double*>(out));
    BMF_LAUNCH(colsum_final_kernel, dim3(1), dim3(256), 0, s, reinterpret_cast<const double*>(out), (int)pblocks, kp, colsum);
    const int64_t total = out_rows * kp;
    const int64_t blocks = (total + 255) / 256;
    BMF_LAUNCH(fill_rows_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s, colsum, kp, total, out);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

// ---- one whole iteration of the link models' loop per call (see bmf_hip.h) ----
static int link_side(const bmf_link_loop* st, bool is_v, double reg, void* stream) {
    const int kp = st->kp;
    const uint32_t* bits = is_v ? st->XTbits : st->Xbits;
    const int64_t rows_pad = is_v ? st->n_pad : st->m_pad, ldx = is_v ? st->ldxt : st->ldx, opad = is_v ? st->m_pad : st->n_pad;
    const int rows = is_v ? st->n : st->m, cols = is_v ? st->m : st->n, splits = is_v ? st->splitsV : st->splitsU;
    uint16_t* ws_s = is_v ? st->wsV : st->wsU;
    const uint16_t* ws_o = is_v ? st->wsU : st->wsV;
    float* num = is_v ? st->numV : st->numU;
    float* den_slabs = is_v ? st->denV_slabs : st->denU_slabs;
    bmf_epilogue_args e = is_v ? st->epiV : st->epiU;
    const bmf_epilogue_args& eo = is_v ? st->epiU : st->epiV;
    e.reg = reg;
    const int64_t stride = rows_pad * kp;
    int rc = bmf_link_pass16(bits, rows_pad, ldx, rows, cols, ws_s, ws_o, opad, kp, st->link, st->lamda, num, den_slabs, stride, splits, stream);
    if (rc != BMF_OK) return rc;
    if (st->link == BMF_LINK_SIGMOID) rc = bmf_reduce_slabs(den_slabs, stride, splits, stride, const_cast<float*>(e.den), nullptr, stream);
    else rc = bmf_colsum_fill(eo.F, cols, kp, st->colsum, const_cast<float*>(e.den), rows_pad, stream);   // column sums of the other factor
    if (rc != BMF_OK) return rc;
    rc = bmf_mu_epilogue(&e, stream);
    if (rc != BMF_OK) return rc;
    // the operand copies of BOTH factors: the column scales of a pair depend on both (link_pair_scales_kernel)
    return bmf_link_split_pair(st->epiU.F, st->m_pad, st->epiV.F, st->n_pad, kp, st->wsU, st->wsV, stream);
}

extern "C" int bmf_link_iterate(const bmf_link_loop* st, double reg, int with_update, double* host_row, void* stream) {
    BMF_REQUIRE(st && host_row, "bmf_link_iterate: null pointer");
    BMF_REQUIRE(st->struct_bytes == (int32_t)sizeof(bmf_link_loop), "bmf_link_iterate: struct_bytes=%d, library expects %d", st->struct_bytes,
                (int)sizeof(bmf_link_loop));
    BMF_REQUIRE(st->link == BMF_LINK_SIGMOID || st->link == BMF_LINK_KL, "bmf_link_iterate: link must be BMF_LINK_SIGMOID or BMF_LINK_KL");
    BMF_REQUIRE(st->Xbits && st->XTbits && st->wsU && st->wsV && st->numU && st->numV && st->sums && st->counts && st->Up64 && st->Vp64 &&
                    st->epiU.F64 && st->epiV.F64 && st->epiU.den && st->epiV.den && (st->link == BMF_LINK_KL ? st->colsum != nullptr
                                                                                                                : (st->denU_slabs && st->denV_slabs)),
                "bmf_link_iterate: null device pointer in state");
    hipStream_t s = (hipStream_t)stream;
    if (with_update) {
        BMF_HIP_CHECK(hipMemcpyAsync(st->Up64, st->epiU.F64, (size_t)st->m_pad * st->kp * sizeof(double), hipMemcpyDeviceToDevice, s));
        BMF_HIP_CHECK(hipMemcpyAsync(st->Vp64, st->epiV.F64, (size_t)st->n_pad * st->kp * sizeof(double), hipMemcpyDeviceToDevice, s));
        int rc = link_side(st, true, reg, stream);    // V, then U with the new V (Gauss-Seidel)
        if (rc != BMF_OK) return rc;
        rc = link_side(st, false, reg, stream);
        if (rc != BMF_OK) return rc;
    }
    BMF_HIP_CHECK(hipMemsetAsync(st->sums, 0, 4 * sizeof(double), s));
    BMF_HIP_CHECK(hipMemsetAsync(st->counts, 0, 4 * sizeof(unsigned long long), s));
    int rc = bmf_link_sums16(st->Xbits, st->m_pad, st->ldx, st->m, st->n, st->wsU, st->wsV, st->n_pad, st->kp, st->link, st->lamda, st->Obits, st->sums, stream);
    if (rc != BMF_OK) return rc;
    rc = bmf_cover_count(st->Xbits, st->m_pad, st->ldx, st->n_pad / 32, st->epiU.rowbits, st->epiV.colbits, st->n_pad / 32, st->kp, st->counts, nullptr, stream);
    if (rc != BMF_OK) return rc;
    // gather: [0] = sums[2] (KL objective), [3], [4] = sums[0], sums[1]
    return bmf_masked_scalars(st->sums + 2, st->epiU.partials, st->nbU, st->epiV.partials, st->nbV, st->sums, st->counts, host_row, stream);
}
