// K1/K2: out = bits(A) . F  -- the two big contractions of the multiplicative update on a Boolean X.
//
//   X  @ V   (A = X bits,   panel of V)   replaces  multiply(W, X) @ V      PyBMF/models/BinaryMFPenalty.py:139
//   X^T @ U  (A = X^T bits, panel of U)   replaces  multiply(W, X).T @ U    PyBMF/models/BinaryMFPenalty.py:154
//
// Design (gfx950):
//   * A is a bit matrix (1 bit per cell: 250 MB for 100k x 20k), so the kernel is MFMA-bound, not HBM-bound.
//     Each lane loads the bits of "its" rows straight into VGPRs (4 B per row per 128 reduction indices) and expands
//     them to 16-bit {0, 2.0} operands with one shift + one AND per dword:  (w << s) & 0x40004000 puts bit b in
//     bit 14 of the low half and bit b+16 in bit 14 of the high half; 0x4000 is 2.0 in bf16 and in fp16, and the
//     result is scaled by 0.5 (exact) on the way out.
//   * The factor is fed as a panel of T 16-bit addends: bf16 (F = t0 + t1 + t2; T = 3 reproduces fp32 exactly) or two
//     fp16 addends of the column-scaled factor (bmf_make_panel_f16: 22 significant bits relative to the column maximum,
//     2/3 of the MFMA work; the default).  All T products accumulate into the same fp32 MFMA accumulator.  The panel is
//     stored position-permuted (common.h: panel_pos) so that each lane's B fragment is 16 contiguous bytes.
//   * A panel stage (T x kp x 128 elements) is brought into LDS by LDS-DMA (global_load_lds, 16 B/lane) into a ring of
//     three buffers, two stages ahead; one bare s_barrier per stage.  The XOR swizzle that makes the ds_read_b128
//     fragment reads conflict-free is applied on the DMA *source* address (the LDS image must stay lane-linear).
//   * v_mfma_f32_16x16x32_{f16,bf16}, 64-wide waves: each wave owns 64 rows x kp columns (4 x 2NT accumulator tiles); a
//     workgroup of 8 waves shares one panel stage.  B fragments are double-buffered in registers with hand-placed
//     ds_read_b128 / s_waitcnt (inline asm) so that the fetch of k-step i+1 -- across the stage barrier too -- runs under
//     the MFMAs of k-step i, and sched_group_barrier spreads the bit-expansion VALU ops between the MFMAs.
//   * One persistent workgroup per CU; the (row tile, stage) space is cut stream-K fashion into equal slices.  A tile's
//     partial results go to slabs that the consumer sums in fixed order (deterministic, no atomics).
//
// Measured on MI355X at 100k x 20k, k = 64, fp16 x 2 (profiles/): 0.34 ms per launch = 755 TFLOP/s
// algorithmic (1.51 PFLOP/s of MFMA work).  What the rest costs, from builds with one ingredient removed: panel DMA 14 %,
// LDS fragment reads 8 %, barrier 8 %, bit expansion 7 %; the MFMA-only skeleton of this loop runs at 0.28 ms.  Things
// that were tried and lost: two workgroups per CU (needs <= 128 VGPRs: no fragment double-buffering, 10 % slower), 128 rows
// per wave (256-VGPR cap, VALU can no longer interleave, 6 % slower), 4 waves x 128 rows with one wave per SIMD and all
// 512 registers (20 % slower: a single wave cannot keep the matrix pipe fed), 12 waves x 64 rows = three waves per SIMD at
// 168 VGPRs (4 % slower), X words loaded before the DMA is issued (6 % slower).
#include "common.h"

#include <utility>

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// scheduling hint: one MFMA, then this MFMA's share of the 8 VALU ops that belong to a group of M MFMAs (the immediates
// of sched_group_barrier must be integer constant expressions, hence the pack expansion)
template <int M, int I>
__device__ __forceinline__ void interleave_one() {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    constexpr int c = ((I % M + 1) * 8) / M - ((I % M) * 8) / M;
    if constexpr (c > 0) __builtin_amdgcn_sched_group_barrier(0x002, c, 0);
}
template <int M, int... I>
__device__ __forceinline__ void interleave_mfma_valu(std::integer_sequence<int, I...>) {
    (interleave_one<M, I>(), ...);
}

// one MFMA on 16-byte A / B fragments of either 16-bit format (0x4000 reads as 2.0 in both bf16 and fp16)
template <bool F16>
__device__ __forceinline__ f32x4 mfma_16x16x32(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ f32x16 mfma_32x32x16(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// F16 = the panel holds fp16 addends of the column-scaled factor (bmf_make_panel_f16); colscale[c] = 0.5 / 2^e_c undoes the
// scale (and the 2.0 of the expanded bits) on the way out.  colscale == nullptr: plain 0.5 (bf16 panels).
//
// One workgroup per CU (measured: two co-resident workgroups need <= 128 VGPRs, which costs the in-register prefetch of
// the B fragments and is 10 % slower).  The panel stages live in a ring of RING LDS buffers filled by LDS-DMA LOOK =
// RING - 1 stages ahead; a stage's buffer is complete and visible one barrier before its first read, so the B fragments
// of the next stage's first k-step are fetched into registers *before* the end-of-stage barrier and the MFMA pipe
// does not drain at stage boundaries.
// MT = 16-row groups per wave (4: 64 rows, 8: 128 rows).  With 128 rows per wave a stage's DMA, LDS reads and barrier are
// amortised over twice the MFMAs (measured cost of those three at MT = 4: 14 % + 8 % + 8 % of the kernel).
template <int NT, int T, int WAVES, bool F16, int MT>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void xf_bits_kernel(const uint32_t* __restrict__ A, int64_t ldw, int stages,
                                                              const uint16_t* __restrict__ P, int64_t ldp,
                                                              float* __restrict__ out, int64_t slab_stride,
                                                              int units_per_wg, int64_t total_units, int slots,
                                                              const float* __restrict__ colscale,
                                                              const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;  // early stop tripped: the state is frozen, skip the work (wave-uniform)
    constexpr int NC = NT * 32;               // panel columns
    constexpr int LROWS = T * NC;             // 256-byte LDS rows per stage
    constexpr int STAGE_BYTES = LROWS * 256;  // T*NC*128 bf16
    constexpr int PIECES = LROWS / 4;  // 1 KiB DMA pieces per stage
    constexpr int DMA_PER_WAVE = (PIECES + WAVES - 1) / WAVES;
    // three buffers: the copy of the next stage's X words into the working registers at the end of a stage makes the
    // compiler wait for every outstanding load there anyway, so a deeper ring buys nothing (measured)
    constexpr int RING = 3;
    constexpr int LOOK = RING - 1;  // DMA look-ahead in stages
    (void)LOOK;
    static_assert(RING * STAGE_BYTES <= 160 * 1024, "stage ring must fit the 160 KiB LDS");
    __shared__ __attribute__((aligned(16))) char smem[RING * STAGE_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#if BMF_SHAPE16
    const int r = lane & 15, h = lane >> 4;  // 16x16x32: row/column inside a 16-tile, k-group 0..3
#else
    const int r = lane & 31, h = lane >> 5;
#endif
    // per-lane DMA source: LDS row (4*q + lane/16), 16-byte chunk (lane%16) of that row holds source chunk
    // (lane%16) ^ (column & 15)
    const int d_sub = lane >> 4, d_chunk = lane & 15;
    // LDS offset of this lane's B fragment row for (nt, t): ((t*NC + tile*nt + r) * 256); chunk = ch ^ (r & 15)
    const int b_row = r * 256;
    const int b_sw = r & 15;

    // per-lane 32-bit byte offsets (recomputed at every issue: 3 VALU ops are cheaper than 4 live registers here) on
    // wave-uniform 64-bit bases (SGPRs): keeps the address math out of the VGPR budget
    auto issue_dma = [&](int stage, int buf) {
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {
            const int q = wave + WAVES * i;               // wave-uniform 1 KiB piece: LDS rows 4q .. 4q+3 (= t*NC + j)
            if (PIECES % WAVES != 0 && q >= PIECES) break;
#ifdef BMF_EXP_PANEL_WRAP  // timing experiment only: every workgroup re-reads the same 8 stages (L2-resident panel)
            const char* base = reinterpret_cast<const char*>(P + (int64_t)(4 * q) * ldp + (int64_t)(stage & 7) * 128);
#else
            const char* base = reinterpret_cast<const char*>(P + (int64_t)(4 * q) * ldp + (int64_t)stage * 128);
#endif
            char* dst = smem + buf * STAGE_BYTES + q * 1024;  // wave-uniform base; hardware adds lane*16
            const int jcol = (4 * q + d_sub) & (NC - 1);
            const unsigned d_off = (unsigned)(d_sub * ldp + ((d_chunk ^ (jcol & 15)) << 3)) * 2u;
#ifdef BMF_EXP_NODMA
            if (stage < 0)
#endif
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + d_off),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    // stream-K: the (row tile, stage) space is linearised tile-major and cut into equal contiguous slices, one per
    // workgroup; a slice may end one tile and start the next.  Partial tiles go to slab `slot` = how many workgroups
    // started on that tile before this one, so the consumer's slab-order sum is deterministic.
    int64_t u = (int64_t)blockIdx.x * units_per_wg;
    const int64_t u_end = min(u + units_per_wg, total_units);
    while (u < u_end) {
    const int tile = (int)(u / stages);
    const int s0 = (int)(u - (int64_t)tile * stages);
    const int s1 = (int)min((int64_t)stages, s0 + (u_end - u));
    const int first_wg = (int)(((int64_t)tile * stages) / units_per_wg);
    const int slot = (int)blockIdx.x - first_wg;
    const int64_t row_base = (int64_t)tile * (WAVES * 16 * MT) + wave * (16 * MT);

#if BMF_SHAPE16
    // ---- 16x16x32 flavour: MT x (2*NT) accumulator tiles of 16x16 (16*MT rows x kp columns per wave) ----
    constexpr int NT16 = 2 * NT;
    f32x4 acc[MT][NT16];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT16; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[mt][nt][i] = 0.f;

    // lane (r, g = h) reads word g of the stage for rows row_base + 16*mt + r: uniform base + 32-bit lane offset
    const uint32_t* a_base = A + row_base * ldw;
    const unsigned a_off = (unsigned)(r * ldw + h);
    unsigned aw[MT], an[MT];  // X words of this stage and of the next one
    auto load_a = [&](int stage, unsigned (&dst)[MT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            // the base is wave-uniform; saying so (readfirstlane) gets the SGPR-base + 32-bit-offset addressing mode
            // instead of MT 64-bit VGPR addresses
            const uint64_t b = reinterpret_cast<uint64_t>(a_base + (int64_t)(16 * mt) * ldw + 4 * (int64_t)stage);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
            dst[mt] = reinterpret_cast<const uint32_t*>(((uint64_t)hi << 32) | lo)[a_off];
        }
    };

    // B fragments are read with hand-placed ds_read_b128 / s_waitcnt (inline asm): the compiler's own waitcnt insertion
    // turns every cross-iteration fragment dependency into lgkmcnt(0) right after the next fetch was issued, which
    // serialises LDS latency with the MFMAs.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto fetch_b = [&](int slot, int ks, u32x4 (&dst)[NT16][T]) {
        const unsigned addr = lds0 + (unsigned)(slot * STAGE_BYTES) + (unsigned)(b_row + (((ks * 4 + h) ^ b_sw) << 4));
#pragma unroll
        for (int nt = 0; nt < NT16; ++nt)
#pragma unroll
            for (int t = 0; t < T; ++t)
#ifdef BMF_EXP_NOLDS
                asm volatile("" : "+v"(dst[nt][t]) : "v"(addr));
#else
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[nt][t]) : "v"(addr), "n"((t * NC + 16 * nt) * 256));
#endif
    };
    auto wait_b = [&](u32x4 (&dst)[NT16][T]) {  // all outstanding LDS reads have landed; ties the fragments to the wait
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int nt = 0; nt < NT16; ++nt)
#pragma unroll
            for (int t = 0; t < T; ++t) asm volatile("" : "+v"(dst[nt][t]));
    };

    // prologue: stages s0 and s0+1 land before the first barrier, the rest of the look-ahead window stays in flight
    load_a(s0, aw);
    issue_dma(s0, 0);
    if (s0 + 1 < s1) issue_dma(s0 + 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int l = 2; l < LOOK; ++l)
        if (s0 + l < s1) issue_dma(s0 + l, l);
    u32x4 b0[NT16][T], b1[NT16][T];  // even / odd k-steps
#ifdef BMF_EXP_NOLDS
#pragma unroll
    for (int nt = 0; nt < NT16; ++nt)
#pragma unroll
        for (int t = 0; t < T; ++t) b0[nt][t] = b1[nt][t] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
#endif
    fetch_b(0, 0, b0);
    wait_b(b0);

    int cur = 0;  // ring slot of stage s
    for (int s = s0; s < s1; ++s) {
        int nxt = cur + 1, far = cur + LOOK;
        if (nxt >= RING) nxt -= RING;
        if (far >= RING) far -= RING;
        // stage s + 2 goes into the buffer stage s - 1 was read from (everybody is past the barrier that ended it); the X
        // words of stage s + 1 ride along.  Both have this whole stage to land.
        if (s + LOOK < s1) issue_dma(s + LOOK, far);
        if (s + 1 < s1) load_a(s + 1, an);

        auto k_step = [&](int ks, u32x4 (&bc)[NT16][T], u32x4 (&bx)[NT16][T]) {
            // fetch the next k-step's fragments (from the next stage's buffer at the end: it has been complete and
            // visible since the previous barrier), run this k-step's MFMAs under that latency, then collect
            if (ks < 3) fetch_b(cur, ks + 1, bx);
            else if (s + 1 < s1) fetch_b(nxt, 0, bx);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                u32x4 av;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int bit = 4 * ks + e;  // low half <- bit, high half <- bit + 16
#ifdef BMF_EXP_NOVALU
                    av[e] = aw[mt];
#else
                    av[e] = (bit <= 14 ? (aw[mt] << (14 - bit)) : (aw[mt] >> (bit - 14))) & 0x40004000u;
#endif
                }
#pragma unroll
                for (int t = 0; t < T; ++t)  // addend outermost: consecutive MFMAs hit different accumulators
#pragma unroll
                    for (int nt = 0; nt < NT16; ++nt) acc[mt][nt] = mfma_16x16x32<F16>(av, bc[nt][t], acc[mt][nt]);
            }
            // issue order inside the block: the 8 shift/and ops that expand the next row group's bits are spread
            // between the MFMAs of the current one (left to itself the scheduler clusters them and the matrix pipe
            // idles behind each cluster)
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
            interleave_mfma_valu<NT16 * T>(std::make_integer_sequence<int, (MT - 1) * NT16 * T>{});
            __builtin_amdgcn_sched_group_barrier(0x008, NT16 * T, 0);
            __builtin_amdgcn_sched_barrier(0);
            wait_b(bx);
        };
        k_step(0, b0, b1);
        k_step(1, b1, b0);
        k_step(2, b0, b1);
        k_step(3, b1, b0);

        // This wave's pieces of stage s + 2 (issued at the top of this stage) must be in LDS before the barrier publishes
        // the stage.  Bare s_barrier: __syncthreads() would add an lgkmcnt(0)/vmcnt(0) fence of its own.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef BMF_EXP_NOBAR
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) aw[mt] = an[mt];
        cur = nxt;
    }

    // C/D layout of 16x16 MFMA: column = lane & 15, row = 4*(lane >> 4) + reg
    float* o = out + (int64_t)slot * slab_stride;
    float osc[NT16];
#pragma unroll
    for (int nt = 0; nt < NT16; ++nt) osc[nt] = colscale ? colscale[16 * nt + r] : 0.5f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT16; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t row = row_base + 16 * mt + 4 * h + i;
#ifdef BMF_EXP_NOSTORE  // timing experiment only
                if (acc[mt][nt][i] == 12345.678f)
#endif
                o[row * NC + 16 * nt + r] = osc[nt] * acc[mt][nt][i];
            }
    if (s1 == stages) {  // last contributor of this tile: the slab slots nobody writes must read as zero
        for (int z = slot + 1; z < slots; ++z) {
            float* oz = out + (int64_t)z * slab_stride;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT16; ++nt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int64_t row = row_base + 16 * mt + 4 * h + i;
                        oz[row * NC + 16 * nt + r] = 0.f;
                    }
        }
    }
#else
    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    // per-lane A pointers (bits of row row_base + 32*mt + r, word 2*h of the stage)
    const uint32_t* a_ptr0 = A + (row_base + r) * ldw + 2 * h;
    const uint32_t* a_ptr1 = a_ptr0 + 32 * ldw;

    u32x2 a_cur[2], a_nxt[2];
    if (s0 < s1) {
        issue_dma(s0, 0);
        a_cur[0] = *reinterpret_cast<const u32x2*>(a_ptr0 + 4 * (int64_t)s0);
        a_cur[1] = *reinterpret_cast<const u32x2*>(a_ptr1 + 4 * (int64_t)s0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int s = s0; s < s1; ++s) {
        const int cur = (s - s0) & 1;
        if (s + 1 < s1) {
            issue_dma(s + 1, cur ^ 1);
            a_nxt[0] = *reinterpret_cast<const u32x2*>(a_ptr0 + 4 * (int64_t)(s + 1));
            a_nxt[1] = *reinterpret_cast<const u32x2*>(a_ptr1 + 4 * (int64_t)(s + 1));
        }
        const char* buf = smem + cur * STAGE_BYTES;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const unsigned w0 = q ? a_cur[0].y : a_cur[0].x;
            const unsigned w1 = q ? a_cur[1].y : a_cur[1].x;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int ch = ((q * 4 + ks) * 2 + h) ^ b_sw;
                u32x4 b[NT][T];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int t = 0; t < T; ++t)
                        b[nt][t] = *reinterpret_cast<const u32x4*>(buf + (t * NC + 32 * nt) * 256 + b_row + ch * 16);
                u32x4 a0, a1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int bit = 4 * ks + i;  // low half <- bit, high half <- bit + 16
                    a0[i] = (bit <= 14 ? (w0 << (14 - bit)) : (w0 >> (bit - 14))) & 0x40004000u;
                    a1[i] = (bit <= 14 ? (w1 << (14 - bit)) : (w1 >> (bit - 14))) & 0x40004000u;
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        acc[0][nt] = mfma_32x32x16<F16>(a0, b[nt][t], acc[0][nt]);
                        acc[1][nt] = mfma_32x32x16<F16>(a1, b[nt][t], acc[1][nt]);
                    }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        a_cur[0] = a_nxt[0];
        a_cur[1] = a_nxt[1];
    }

    // C/D layout of 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
    float* o = out + (int64_t)slot * slab_stride;
    float osc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) osc[nt] = colscale ? colscale[32 * nt + r] : 0.5f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t row = row_base + 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * h;
                o[row * NC + 32 * nt + r] = osc[nt] * acc[mt][nt][i];
            }
    if (s1 == stages) {  // last contributor of this tile: the slab slots nobody writes must read as zero
        for (int z = slot + 1; z < slots; ++z) {
            float* oz = out + (int64_t)z * slab_stride;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int64_t row = row_base + 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * h;
                        oz[row * NC + 32 * nt + r] = 0.f;
                    }
        }
    }
#endif
    u += s1 - s0;
    }  // stream-K slice loop
}

struct Plan {
    int num_wgs, units_per_wg, slots;
    int64_t total;
};

// one persistent workgroup per CU
Plan make_plan(int64_t rows_pad, int stages, int terms, int kp, int tile_rows) {
    Plan p;
    const int n_row_tiles = (int)(rows_pad / tile_rows);
    p.total = (int64_t)n_row_tiles * stages;
    (void)terms;
    (void)kp;
    int64_t g = (int64_t)bmf_cu_count();
    if (g > p.total) g = p.total;
    p.units_per_wg = (int)((p.total + g - 1) / g);
    p.num_wgs = (int)((p.total + p.units_per_wg - 1) / p.units_per_wg);
    int slots = 1;
    for (int t = 0; t < n_row_tiles; ++t) {
        const int first = (int)(((int64_t)t * stages) / p.units_per_wg);
        const int last = (int)((((int64_t)(t + 1)) * stages - 1) / p.units_per_wg);
        if (last - first + 1 > slots) slots = last - first + 1;
    }
    p.slots = slots;
    return p;
}

template <int NT, int T, int WAVES, bool F16, int MT>
int launch(const uint32_t* A, int64_t ldw, int stages, const uint16_t* P, int64_t ldp, float* out, int64_t slab_stride,
           const Plan& pl, int slots, const float* colscale, const int32_t* stop, hipStream_t s) {
    dim3 grid((unsigned)pl.num_wgs), block(WAVES * 64);
    BMF_LAUNCH((xf_bits_kernel<NT, T, WAVES, F16, MT>), grid, block, 0, s, A, ldw, stages, P, ldp, out, slab_stride,
                       pl.units_per_wg, pl.total, slots, colscale, stop);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

}  // namespace

// rows per workgroup tile: 8 waves x 128 rows when the register budget allows it (2 addends) and the padding fits
static int tile_rows_for(int64_t rows_pad, int terms, int kp) {
    (void)kp;
#if BMF_SHAPE16 && defined(BMF_GEMM_MT8)
    // 128 rows per wave halves the per-MFMA cost of DMA, LDS reads and barriers, but the wave then sits at the 256-VGPR cap
    // and loses the second A-operand register set that lets VALU and MFMA interleave: measured 6 % slower.  Kept for A/B.
    return (terms == 2 && rows_pad % 1024 == 0) ? 1024 : 512;
#else
    (void)rows_pad;
    (void)terms;
    return 512;
#endif
}

extern "C" int bmf_xf_bits_slots(int64_t rows_pad, int64_t red_words, int terms, int kp) {
    if (rows_pad <= 0 || rows_pad % BMF_ROW_PAD || red_words <= 0 || red_words % 4 || (kp != 32 && kp != 64) || terms < 1 ||
        terms > 3) {
        bmf_set_error("bmf_xf_bits_slots: bad arguments");
        return BMF_ERR_BAD_ARG;
    }
    return make_plan(rows_pad, (int)(red_words / 4), terms, kp, tile_rows_for(rows_pad, terms, kp)).slots;
}

int bmf_xf_bits_launch(const uint32_t* Abits, int64_t rows_pad, int64_t ldw, int64_t red_words, const uint16_t* panel,
                        int64_t ldp, int terms, int kp, float* out, int64_t slab_stride, int splits, int panel_kind,
                        const float* colscale, const int32_t* stop, hipStream_t s) {
    BMF_REQUIRE(Abits && panel && out, "bmf_xf_bits: null pointer");
    BMF_REQUIRE(rows_pad > 0 && rows_pad % BMF_ROW_PAD == 0, "bmf_xf_bits: rows_pad=%lld must be a positive multiple of %d",
                (long long)rows_pad, BMF_ROW_PAD);
    BMF_REQUIRE(red_words > 0 && red_words % 4 == 0, "bmf_xf_bits: red_words=%lld must be a positive multiple of 4",
                (long long)red_words);
    BMF_REQUIRE(ldw >= red_words && ldw % 4 == 0, "bmf_xf_bits: ldw=%lld must be >= red_words and a multiple of 4",
                (long long)ldw);
    BMF_REQUIRE(ldp >= 32 * red_words && ldp % 8 == 0, "bmf_xf_bits: ldp=%lld must be >= 32*red_words and a multiple of 8",
                (long long)ldp);
    BMF_REQUIRE(kp == 32 || kp == 64, "bmf_xf_bits: kp=%d must be 32 or 64", kp);
    BMF_REQUIRE(terms >= 1 && terms <= 3, "bmf_xf_bits: terms=%d must be 1..3", terms);
    BMF_REQUIRE(panel_kind == BMF_PANEL_BF16 || (panel_kind == BMF_PANEL_F16 && colscale && terms == 2),
                "bmf_xf_bits: panel_kind must be BMF_PANEL_BF16, or BMF_PANEL_F16 with terms == 2 and a colscale vector");
    BMF_REQUIRE(slab_stride >= rows_pad * kp, "bmf_xf_bits: slab_stride too small");
    BMF_REQUIRE(bmf_aligned16(Abits) && bmf_aligned16(panel) && bmf_aligned16(out), "bmf_xf_bits: pointers must be 16-byte aligned");
    const int stages = (int)(red_words / 4);
    const int tile_rows = tile_rows_for(rows_pad, terms, kp);
    const Plan pl = make_plan(rows_pad, stages, terms, kp, tile_rows);
    BMF_REQUIRE(splits >= pl.slots, "bmf_xf_bits: splits=%d but this shape needs %d slab slots (bmf_xf_bits_slots)", splits, pl.slots);
#define BMF_XF_CASE(NT_, T_, F16_, MT_)                                                                             \
    if (kp == 32 * NT_ && terms == T_ && (panel_kind == BMF_PANEL_F16) == F16_ && tile_rows == 128 * MT_)           \
        return launch<NT_, T_, 8, F16_, MT_>(Abits, ldw, stages, panel, ldp, out, slab_stride, pl, splits, colscale, stop, s);
    BMF_XF_CASE(1, 2, true, 4) BMF_XF_CASE(2, 2, true, 4) BMF_XF_CASE(1, 2, true, 8) BMF_XF_CASE(2, 2, true, 8)
    BMF_XF_CASE(1, 1, false, 4) BMF_XF_CASE(1, 2, false, 4) BMF_XF_CASE(1, 3, false, 4)
    BMF_XF_CASE(2, 1, false, 4) BMF_XF_CASE(2, 2, false, 4) BMF_XF_CASE(2, 3, false, 4)
#if BMF_SHAPE16
    BMF_XF_CASE(1, 2, false, 8) BMF_XF_CASE(2, 2, false, 8)
#endif
#undef BMF_XF_CASE
    bmf_set_error("bmf_xf_bits: unsupported kp/terms");
    return BMF_ERR_UNSUPPORTED;
}

extern "C" int bmf_xf_bits(const uint32_t* Abits, int64_t rows_pad, int64_t ldw, int64_t red_words,
                           const uint16_t* panel, int64_t ldp, int terms, int kp, float* out, int64_t slab_stride,
                           int splits, void* stream) {
    return bmf_xf_bits_launch(Abits, rows_pad, ldw, red_words, panel, ldp, terms, kp, out, slab_stride, splits, BMF_PANEL_BF16,
                              nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int bmf_xf_bits_f16(const uint32_t* Abits, int64_t rows_pad, int64_t ldw, int64_t red_words,
                               const uint16_t* panel, int64_t ldp, const float* colscale, int kp, float* out,
                               int64_t slab_stride, int splits, void* stream) {
    return bmf_xf_bits_launch(Abits, rows_pad, ldw, red_words, panel, ldp, 2, kp, out, slab_stride, splits, BMF_PANEL_F16,
                              colscale, nullptr, (hipStream_t)stream);
}
