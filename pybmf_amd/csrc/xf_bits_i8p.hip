// The int8 bits GEMM of xf_bits_i8.hip as ONE eight-wave workgroup per CU whose two wave groups work in ANTI-PHASE (round 5,
// variant 4 of bmf_xf_bits_i8_variant; same operands, same digit planes, same exact int32 accumulation and fp64 recombination).
//
//   X  @ V   (A = X bits,   limb panel of V)   replaces  multiply(W, X) @ V      PyBMF/models/BinaryMFPenalty.py:139
//   X^T @ U  (A = X^T bits, limb panel of U)   replaces  multiply(W, X).T @ U    PyBMF/models/BinaryMFPenalty.py:154
//
// Why: in the default kernel (4 waves, 256 x 32 tile, two workgroups per CU) the two waves of a SIMD belong to two workgroups that
// drift against each other; each wave interleaves its own loads, waits and barrier with its own 48 matrix instructions per stage, and
// the matrix pipe ends up 72-75 % busy (9.4 ns per instruction and SIMD where the instruction stream alone needs 7.6:
// profiles/r05_i8_smfmac.md).  Here a workgroup owns 512 rows x 32 columns; waves 0-3 (group A) and 4-7 (group B) share the SIMDs
// pairwise and alternate between two kinds of phase, a workgroup barrier in between:
//   phase 2u:     A COMPUTES stage u (48 matrix instructions + the bit expansion in their shadow, operands in registers)
//                 B LOADS stage u  (its twelve digit-plane fragments LDS -> registers, its X words at a group start, its share of the
//                                   LDS-DMA pieces, the waits)
//   phase 2u + 1: A loads stage u + 1 (and issues the plane pieces of stage u + 4) | B computes stage u
// so the instructions that stall an issuing wave sit beside the partner's matrix instructions by construction.  The digit planes of
// a stage serve eight waves (half the plane bytes per matrix instruction); waves 0-3 issue the plane pieces, waves 4-7 the X pieces.
//
// LDS (one workgroup per CU): plane ring 4 x 12 KiB + X words 2 groups x 32 KiB = 112 KiB.  Plane ring: stage u lives in buffer
// u % 4; A reads it in phase 2u - 1, B in phase 2u, A refills it (stage u + 4) in phase 2u + 1 and waits, at the end of its compute
// phase 2u, for stage u + 1 (issued in phase 2u - 5; the six pieces of stages u + 2, u + 3 may stay in flight).  X words (the tiled
// copy of the bit matrix, bmf_tile_bits: 16 KiB per 256 rows and group of four stages): two buffers of a whole group; B requests
// group gq + 2 in phases 2, 4, 6 of group gq (both groups have read buffer gq & 1 by the end of phase 0) and waits at the end of
// phase 6 for group gq + 1, which A reads in phase 7.
#include "common.h"
#include "i8_plan.h"

#include <type_traits>
#include <utility>

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int M, int V, int I>
__device__ __forceinline__ void pp_interleave_one() {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    constexpr int c = ((I % M + 1) * V) / M - ((I % M) * V) / M;
    if constexpr (c > 0) __builtin_amdgcn_sched_group_barrier(0x002, c, 0);
}
template <int M, int V, int... I>
__device__ __forceinline__ void pp_interleave(std::integer_sequence<int, I...>) {
    (pp_interleave_one<M, V, I>(), ...);
}

template <int L>
__global__ __launch_bounds__(512, 1) void xf_bits_i8p_kernel(const uint32_t* __restrict__ A, int stages, const int8_t* __restrict__ P, int64_t ldp,
                                                              int kp, int col_base, int halves, float* __restrict__ out, int64_t slab_stride,
                                                              int u_len, int64_t total_units, int n_slices, int slots,
                                                              const float* __restrict__ colscale, const int32_t* __restrict__ stop,
                                                              SlicePerm perm) {
    if (stop && *stop != 0) return;
    static_assert(L == 3, "three digit planes");
    constexpr int TILE_ROWS = 512;
    constexpr int STAGE_BYTES = L * 32 * 128;               // 12 KiB
    constexpr int DMA_PER_WAVE = STAGE_BYTES / 1024 / 4;    // plane pieces per issuing wave and stage (waves 0-3)
    constexpr int RING = 4;
    constexpr int XB = 256 * 64;                            // the X words of 256 rows and one group of four stages: 16 KiB
    constexpr int XG_BYTES = 2 * XB;                        // a 512-row tile's group
    static_assert(RING * STAGE_BYTES + 2 * XG_BYTES <= 160 * 1024, "the plane ring and two X groups must fit the 160 KiB LDS");
    __shared__ __attribute__((aligned(16))) char smem[RING * STAGE_BYTES + 2 * XG_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0..7 = 64-row group of the tile
    const bool grp_a = wave < 4;
    const int r = lane & 15, g = lane >> 4;
    const int bx = blockIdx.x & 7, bi = blockIdx.x >> 3;
    const int half = bi % halves;
    const int bslice = (bi / halves) * 8 + bx;
    if (bslice >= 512) return;
    const int slice = perm.p[bslice];
    if (slice >= n_slices) return;
    const int col0 = col_base + 32 * half;

    const int64_t u0 = (int64_t)slice * u_len;
    const int64_t u1 = min(u0 + u_len, total_units);
    if (u0 >= u1) return;
    const int n_groups = (int)((u1 - u0) >> 2);
    const int n_units = n_groups << 2;
    const int groups_per_tile = stages >> 2;

    // ---- plane pieces (waves 0-3): piece q = wave + 4 i (1 KiB): LDS rows 8q .. 8q+7 (row R = limb * 32 + column); lane i fills
    // physical 16-byte chunk i & 7 of row 8q + (i >> 3) with source chunk (i & 7) ^ ((R >> 1) & 7) ----
    const int8_t* dsrc[DMA_PER_WAVE];
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) {
        const int q = (wave & 3) + 4 * i;
        const int limb = q >> 2, j0 = (q & 3) * 8, d_row = lane >> 3, d_chunk = lane & 7;
        const int R = 8 * q + d_row;
        dsrc[i] = P + (int64_t)(limb * kp + col0 + j0 + d_row) * ldp + ((d_chunk ^ ((R >> 1) & 7)) << 4);
    }
    auto issue_dma = [&](int stage, int buf) {
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dsrc[i] + (int64_t)stage * 128),
                                             (__attribute__((address_space(3))) void*)(smem + buf * STAGE_BYTES + ((wave & 3) + 4 * i) * 1024), 16, 0, 0);
    };
    auto lds0_of = [](char* p_) { return (unsigned)(size_t)(__attribute__((address_space(3))) char*)p_; };
    const unsigned lds0 = lds0_of(smem);
    // B fragment of (16-column tile nt, limb l), k-step ks: row l*32 + 16 nt + r, physical chunk (4 ks + g) ^ (r >> 1)
    const unsigned b_addr0 = lds0 + (unsigned)(r * 128) + (unsigned)(((g ^ (r >> 1)) & 7) << 4);
    const unsigned b_addr1 = lds0 + (unsigned)(r * 128) + (unsigned)((((4 + g) ^ (r >> 1)) & 7) << 4);

    // ---- X pieces (waves 4-7): issuing wave j = wave - 4 fetches the words of the computing waves 2 j and 2 j + 1, both in 256-row
    // block j >> 1 of the tile; per computing wave w' four 1-KiB pieces at (w' * 4 + p) * 1024 of the block (rows 16 p .. 16 p + 15 of
    // its 64, 64 bytes each: the tiled layout IS the LDS layout) ----
    int tile = (int)(u0 / stages);
    int st_cur = (int)(u0 - (int64_t)tile * stages);
    const int64_t n_tiles_a = total_units / stages;
    int xq_tile = tile, xq_grp = st_cur >> 2;   // the next group to request; clamps at the last group of the matrix
    auto x_block_ptr = [&](int tl, int grp, int blk) {
        return reinterpret_cast<const char*>(A) + (((int64_t)(2 * tl + blk)) * groups_per_tile + grp) * (int64_t)XB;
    };
    auto advance_xq = [&]() {
        const bool tile_last = xq_grp + 1 == groups_per_tile;
        const bool at_end = tile_last && (xq_tile + 1 == (int)n_tiles_a);
        if (!at_end) {
            xq_grp = tile_last ? 0 : xq_grp + 1;
            xq_tile += tile_last ? 1 : 0;
        }
    };
    const int xj = wave & 3;
    const int x_blk = xj >> 1;
    char* const x_lds = smem + RING * STAGE_BYTES;
    auto issue_x = [&](int xbuf, int k) {   // piece k = 0..7 of the group (xq_tile, xq_grp) into X buffer xbuf
        const int wq = ((2 * xj) & 3) + (k >> 2);
        const int off = (wq * 4 + (k & 3)) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(x_block_ptr(xq_tile, xq_grp, x_blk) + off + lane * 16),
                                         (__attribute__((address_space(3))) void*)(x_lds + xbuf * XG_BYTES + x_blk * XB + off), 16, 0, 0);
    };
    // this computing wave's words: block wave >> 2, rows 64 (wave & 3) .. + 63 of it; lane (r, g): words 4 g .. 4 g + 3 of row 16 mt + r
    const unsigned x_rd = lds0 + (unsigned)(RING * STAGE_BYTES + (wave >> 2) * XB + (64 * (wave & 3) + r) * 64 + g * 16);
    u32x4 aq[4];
    auto read_x = [&](int xbuf) {
        const unsigned b_ = x_rd + (unsigned)(xbuf * XG_BYTES);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(aq[mt]) : "v"(b_), "n"(16 * 64 * mt));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) asm volatile("" : "+v"(aq[mt]));
    };

    float osc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) osc[nt] = colscale[col0 + 16 * nt + r];
    i32x4 acc[4][2][L];
    auto zero_acc = [&]() {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int l = 0; l < L; ++l) acc[mt][nt][l] = i32x4{0, 0, 0, 0};
    };
    // C/D layout of the 16x16 MFMA: column = lane & 15, row = 4 (lane >> 4) + i.  The digit planes are recombined in fp64.
    auto write_tile = [&](int tl, bool last_of_tile) {
        const int64_t tu = (int64_t)tl * stages;
        const int slot = slice - (int)(tu / u_len);
        const int64_t row_base = (int64_t)tl * TILE_ROWS + wave * 64;
        float* o = out + (int64_t)slot * slab_stride;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t row = row_base + 16 * mt + 4 * g + i;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    long long v = 0;
#pragma unroll
                    for (int l = L - 1; l >= 0; --l) v = v * 256 + acc[mt][nt][l][i];
                    o[row * kp + col0 + 16 * nt + r] = (float)((double)v * (double)osc[nt]);
                    if (last_of_tile)   // last contributor of this tile: the slab slots nobody writes must read as zero
                        for (int z = slot + 1; z < slots; ++z) out[(int64_t)z * slab_stride + row * kp + col0 + 16 * nt + r] = 0.f;
                }
            }
    };

    // ---- prologue: plane stages 0..3; the X words of groups 0 and 1 of the run ----
    int st_dma = st_cur;
    int n_dma = 0;
    auto next_dma = [&](int buf) {
        if (grp_a) issue_dma(st_dma, buf);
        ++n_dma;
        const int nx = st_dma + 1 == stages ? 0 : st_dma + 1;
        st_dma = n_dma < n_units ? nx : st_dma;
    };
    next_dma(0);
    next_dma(1);
    next_dma(2);
    next_dma(3);
    if (!grp_a) {
#pragma unroll
        for (int k = 0; k < 8; ++k) issue_x(0, k);
    }
    advance_xq();
    if (!grp_a) {
#pragma unroll
        for (int k = 0; k < 8; ++k) issue_x(1, k);
    }
    advance_xq();   // (xq now names group 2 of the run, requested during group 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    read_x(0);
    zero_acc();

    // the twelve fragments of a stage: b[ks][nt][l]
    i32x4 bf[2][2][L];
#define BMF_PP_FETCH(tag, slot, ks, nt, l)                                                                                \
    asm volatile("ds_read_b128 %0, %1 offset:%2 ; " tag : "=v"(bf[ks][nt][l]) : "v"((ks) ? b_addr1 : b_addr0),            \
                 "n"((slot) * STAGE_BYTES + ((l) * 32 + 16 * (nt)) * 128))
#define BMF_PP_LOAD(tag, slot)                                                                                            \
    do {                                                                                                                  \
        BMF_PP_FETCH(tag, slot, 0, 0, 0); BMF_PP_FETCH(tag, slot, 0, 1, 0); BMF_PP_FETCH(tag, slot, 0, 0, 1);             \
        BMF_PP_FETCH(tag, slot, 0, 1, 1); BMF_PP_FETCH(tag, slot, 0, 0, 2); BMF_PP_FETCH(tag, slot, 0, 1, 2);             \
        BMF_PP_FETCH(tag, slot, 1, 0, 0); BMF_PP_FETCH(tag, slot, 1, 1, 0); BMF_PP_FETCH(tag, slot, 1, 0, 1);             \
        BMF_PP_FETCH(tag, slot, 1, 1, 1); BMF_PP_FETCH(tag, slot, 1, 0, 2); BMF_PP_FETCH(tag, slot, 1, 1, 2);             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                \
        _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_) _Pragma("unroll") for (int nt_ = 0; nt_ < 2; ++nt_)            \
            _Pragma("unroll") for (int l_ = 0; l_ < L; ++l_) asm volatile("" : "+v"(bf[ks_][nt_][l_]));                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
    } while (0)
    // one k-step (64 indices) of stage tq: per 16-row group the 4 expansion dwords and 2 L matrix instructions; the 8 shift / and ops
    // that expand the next row group's bits are spread between the matrix instructions of the current one (as in the default kernel)
#define BMF_PP_KSTEP(tq, ks)                                                                                              \
    do {                                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                                \
            const unsigned w = aq[mt][tq];                                                                                \
            i32x4 av;                                                                                                     \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) av[e] = (int)((w >> (4 * (ks) + e)) & 0x01010101u);              \
            _Pragma("unroll") for (int l = 0; l < L; ++l) _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                 \
                acc[mt][nt][l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bf[ks][nt][l], acc[mt][nt][l], 0, 0, 0);       \
        }                                                                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);                                                                \
        pp_interleave<2 * L, 8>(std::make_integer_sequence<int, 3 * 2 * L>{});                                            \
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * L, 0);                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
    } while (0)
#define BMF_PP_COMPUTE(tq) do { BMF_PP_KSTEP(tq, 0); BMF_PP_KSTEP(tq, 1); } while (0)
#define BMF_PP_BARRIER() do { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#define BMF_PP_STAGE_A(t)                                                                                                 \
    do {                                                                                                                  \
        BMF_PP_COMPUTE(t);                                                                                                \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA_PER_WAVE) : "memory");                                           \
        BMF_PP_BARRIER(); /* ---- end of phase 2u */                                                                      \
        if ((t) == 3 && flush) {                                                                                          \
            write_tile(tile, tile_end);                                                                                   \
            zero_acc();                                                                                                   \
        }                                                                                                                 \
        next_dma(t);                                                                                                      \
        if ((t) == 3 && gq + 1 < n_groups) read_x(xbuf ^ 1);                                                              \
        BMF_PP_LOAD("A", ((t) + 1) & 3);                                                                                  \
        BMF_PP_BARRIER(); /* ---- end of phase 2u + 1 */                                                                  \
    } while (0)
#define BMF_PP_STAGE_B(t)                                                                                                 \
    do {                                                                                                                  \
        if ((t) == 0 && gq > 0) read_x(xbuf);                                                                             \
        if ((t) == 1) { issue_x(xbuf, 0); issue_x(xbuf, 1); issue_x(xbuf, 2); }                                           \
        if ((t) == 2) { issue_x(xbuf, 3); issue_x(xbuf, 4); issue_x(xbuf, 5); }                                           \
        if ((t) == 3) { issue_x(xbuf, 6); issue_x(xbuf, 7); }                                                             \
        BMF_PP_LOAD("B", t);                                                                                              \
        if ((t) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                    \
        BMF_PP_BARRIER(); /* ---- end of phase 2u */                                                                      \
        BMF_PP_COMPUTE(t);                                                                                                \
        if ((t) == 3 && flush) {                                                                                          \
            write_tile(tile, tile_end);                                                                                   \
            zero_acc();                                                                                                   \
        }                                                                                                                 \
        BMF_PP_BARRIER(); /* ---- end of phase 2u + 1 */                                                                  \
    } while (0)
    // (two loops, one per wave group: with a branch per stage the accumulators and fragments met in phis after every stage and the
    // register allocator spilled -- see xf_bits_i8s.hip)
    if (grp_a) {
        BMF_PP_LOAD("A", 0);
        BMF_PP_BARRIER();   // (phase -1)
        for (int gq = 0; gq < n_groups; ++gq) {
            const int xbuf = gq & 1;
            const bool tile_end = st_cur + 4 == stages;
            const bool flush = tile_end || gq + 1 == n_groups;
            BMF_PP_STAGE_A(0);
            BMF_PP_STAGE_A(1);
            BMF_PP_STAGE_A(2);
            BMF_PP_STAGE_A(3);
            tile += tile_end ? 1 : 0;
            st_cur = tile_end ? 0 : st_cur + 4;
        }
    } else {
        BMF_PP_BARRIER();   // (phase -1)
        for (int gq = 0; gq < n_groups; ++gq) {
            const int xbuf = gq & 1;
            const bool tile_end = st_cur + 4 == stages;
            const bool flush = tile_end || gq + 1 == n_groups;
            BMF_PP_STAGE_B(0);
            BMF_PP_STAGE_B(1);
            BMF_PP_STAGE_B(2);
            BMF_PP_STAGE_B(3);
            advance_xq();
            tile += tile_end ? 1 : 0;
            st_cur = tile_end ? 0 : st_cur + 4;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#undef BMF_PP_FETCH
#undef BMF_PP_LOAD
#undef BMF_PP_KSTEP
#undef BMF_PP_COMPUTE
#undef BMF_PP_BARRIER
#undef BMF_PP_STAGE_A
#undef BMF_PP_STAGE_B
}

}  // namespace

// variant 4 of the whole-factor launch (kp = 64, both column halves): `pl` = make_plan_i8(rows_pad, stages, 64, 4)
int bmf_xf_bits_i8p_launch(const uint32_t* A, int a_tiled, int stages, const int8_t* P, int64_t ldp, int limbs, float* out, int64_t slab_stride,
                           const PlanI8& pl, int slots, const float* colscale, const int32_t* stop, hipStream_t s) {
    BMF_REQUIRE(a_tiled, "bmf_xf_bits_i8 (variant 4): the anti-phase kernel reads the tiled copy of the bit matrix (bmf_tile_bits)");
    BMF_REQUIRE(limbs == 3, "bmf_xf_bits_i8 (variant 4): three digit planes only");
    BMF_REQUIRE(pl.n_big == 0 && pl.u_big == pl.u_small, "bmf_xf_bits_i8 (variant 4): equal slices expected");
    BMF_LAUNCH((xf_bits_i8p_kernel<3>), dim3((unsigned)pl.grid), dim3(512), 0, s, A, stages, P, ldp, 64, 0, 2, out, slab_stride, pl.u_small, pl.total,
               pl.n_slices, slots, colscale, stop, pl.perm);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}
