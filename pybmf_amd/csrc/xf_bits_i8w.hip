// The int8 bits GEMM for a whole 64-column factor: out = bits(A) . F, F as three (two) signed 8-bit digit planes, exact int32
// accumulation -- the same arithmetic, operand layouts and stream-K scheme as xf_bits_i8.hip, with another wave tile.
//
//   X  @ V   replaces  multiply(W, X) @ V      PyBMF/models/BinaryMFPenalty.py:139
//   X^T @ U  replaces  multiply(W, X).T @ U    PyBMF/models/BinaryMFPenalty.py:154
//
// Why a second kernel.  In xf_bits_i8.hip a wave owns 64 rows x 32 columns: every 32-bit word of X is expanded into MFMA A
// operands ((w >> s) & 0x01010101: a shift and an AND per dword) for 6 MFMAs, 1.33 expansion instructions per MFMA, and with two
// such waves per SIMD the vector-issue port is 87 % subscribed before any wait (profiles/r03_i8_stall_attribution.md section 4).
// Expansion work per MFMA depends only on the tile's WIDTH (8 / (3 C) for C 16-column tiles), fragment reads from LDS per MFMA only
// on its HEIGHT (1 / R for R 16-row tiles).  Here a wave owns 32 rows x 64 columns (R = 2, C = 4): 0.67 expansion instructions per
// MFMA, the port 68 % subscribed, at twice the B-fragment reads (LDS array ~50 % busy instead of ~25 %).  The accumulators are the
// same 96 registers, so two waves still fit a SIMD; 64 rows x 64 columns (192 accumulator registers) would leave one wave per
// SIMD, whose every LDS-DMA issue (~60 cycles, 7 per stage) then idles the matrix pipe.
//
// A workgroup = 8 waves (two per SIMD) = 256 rows x 64 columns: it owns whole rows, so X is fetched once per row tile (the
// 32-column kernel's second column half re-reads it through L2) and a row tile has one stream of partial slabs instead of two.
// One workgroup per CU: ring of 4 stages x (L x 64 rows x 128 B) + the X words of one group of four stages.
#include "common.h"
#include "i8_plan.h"

#include <type_traits>
#include <utility>

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int N>
using ic = std::integral_constant<int, N>;

template <int N, int... I>
__device__ __forceinline__ void copy_words(u32x4 (&dst)[N], const u32x4 (&src)[N], std::integer_sequence<int, I...>) {
    ((dst[I] = src[I]), ...);
}

// L: digit planes; R: 16-row groups per wave (a wave owns 16 R rows x 64 columns); WAVES per workgroup (8: two per SIMD, 4: one).
template <int L, int R, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void xf_bits_i8w_kernel(const uint32_t* __restrict__ A, int64_t ldw, int a_tiled, int stages,
                                                              const int8_t* __restrict__ P, int64_t ldp, float* __restrict__ out,
                                                              int64_t slab_stride, int n_big, int u_big, int u_small, int64_t total_units,
                                                              int n_slices, int slots, const float* __restrict__ colscale,
                                                              const int32_t* __restrict__ stop, SlicePerm perm) {
    if (stop && *stop != 0) return;
    constexpr int KP = 64;
    constexpr int WROWS = 16 * R;              // rows per wave
    constexpr int TILE_ROWS = WAVES * WROWS;   // 256 or 512
    constexpr int LROWS = L * 64;              // 128-byte LDS rows per stage: row R = limb * 64 + column
    constexpr int STAGE_BYTES = LROWS * 128;
    constexpr int PIECES = STAGE_BYTES / 1024;
    constexpr int DMA_PER_WAVE = PIECES / WAVES;
    static_assert(DMA_PER_WAVE <= 6, "one panel piece per unit, units 1..6");
    constexpr int XP = R;                      // X pieces (1 KiB = 16 rows x 64 B) per wave and group of four stages
    static_assert(PIECES % WAVES == 0, "every wave issues the same number of DMA pieces (the vmcnt bookkeeping counts on it)");
    constexpr int RING = 4;
    constexpr int XG_BYTES = TILE_ROWS * 64;   // the X words of one group of four stages: TILE_ROWS rows x 16 words
    static_assert(RING * STAGE_BYTES + XG_BYTES <= 160 * 1024, "ring + X words must fit the 160 KiB LDS");
    __shared__ __attribute__((aligned(16))) char smem[RING * STAGE_BYTES + XG_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // = 32-row group of the tile
    const int r = lane & 15, g = lane >> 4;
    const int bslice = blockIdx.x;
    if (bslice >= 512) return;
    const int slice = perm.p[bslice];
    if (slice >= n_slices) return;           // (0xFFFF: this workgroup has no slice)

    // this workgroup's run of (row tile, stage) units: whole groups of four stages (slice lengths % 4 == 0, stages % 4 == 0)
    const int64_t big_end = (int64_t)n_big * u_big;
    const int64_t u0 = slice < n_big ? (int64_t)slice * u_big : big_end + (int64_t)(slice - n_big) * u_small;
    const int64_t u1 = min(u0 + (slice < n_big ? u_big : u_small), total_units);
    if (u0 >= u1) return;
    const int n_groups = (int)((u1 - u0) >> 2);
    const int n_units = n_groups << 2;

    // DMA piece q = wave + WAVES i (1 KiB): LDS rows 8q .. 8q+7; lane i fills physical 16-byte chunk i & 7 of row R = 8q + (i >> 3)
    // with source chunk (i & 7) ^ ((R >> 1) & 7).  Panel row of LDS row R: limb * KP + column = R.  (R ldp < 192 * 2^24 < 2^32)
    unsigned d_off[DMA_PER_WAVE];
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) {
        const int Rr = 8 * (wave + WAVES * i) + (lane >> 3);
        d_off[i] = (unsigned)((int64_t)Rr * ldp) + (unsigned)(((lane & 7) ^ ((Rr >> 1) & 7)) << 4);
    }

    auto lds0_of = [](char* p_) { return (unsigned)(size_t)(__attribute__((address_space(3))) char*)p_; };
    const unsigned lds0 = lds0_of(smem);
    unsigned d_m0[DMA_PER_WAVE];   // LDS destination of piece i inside a ring buffer (wave-uniform)
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) d_m0[i] = lds0 + (unsigned)((wave + WAVES * i) * 1024);
    const unsigned x_m0 = lds0 + (unsigned)(RING * STAGE_BYTES + wave * (XP * 1024));
    // B fragment of (16-column tile nt, limb l), k-step ks, ring slot sl: row l*64 + 16 nt + r, physical chunk (4 ks + g) ^ (r >> 1).
    // The slot's byte offset does not fit the 16-bit immediate of ds_read for sl >= 2: one address register per (sl >> 1, ks).
    unsigned b_addr[2][2];
#pragma unroll
    for (int sh = 0; sh < 2; ++sh)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            b_addr[sh][ks] = lds0 + (unsigned)(sh * 2 * STAGE_BYTES + r * 128 + ((((ks * 4 + g) ^ (r >> 1)) & 7) << 4));

    // X words: lane (r, g) uses, for each of its two 16-row groups, the 16 bytes [4g, 4g+4) words of a group of four stages -- word t
    // in stage t (the panel is stored in the matching order, bmf_panel_pos_i8).  One wave-uniform pointer walks the groups.  Layouts
    // of the bit matrix as in xf_bits_i8.hip: plain rows, or TILED (bmf_tile_bits: the 256 rows x 16 words of a group contiguous).
    int tile = (int)(u0 / stages);
    int st_cur = (int)(u0 - (int64_t)tile * stages);   // first stage of the group being computed (multiple of 4)
    const int64_t n_tiles_a = total_units / stages;
    // (tiled: 256-row blocks in (block row, group) order; a 512-row tile is two block rows, a wave's rows lie in one of them)
    constexpr int B256 = TILE_ROWS / 256;
    const int wblk = (wave * WROWS) >> 8, wrow = (wave * WROWS) & 255;
    const int64_t grps = stages >> 2;
    const uint32_t* a_ptr = a_tiled ? A + ((((int64_t)tile * B256 + wblk) * grps + (st_cur >> 2)) * 256 + wrow) * 16
                                    : A + ((int64_t)tile * TILE_ROWS + wave * WROWS) * ldw + 4 * (int64_t)st_cur;
    const uint32_t* const a_last = a_tiled ? A + ((((n_tiles_a - 1) * B256 + wblk) * grps + (grps - 1)) * 256 + wrow) * 16
                                           : A + ((n_tiles_a - 1) * TILE_ROWS + wave * WROWS) * ldw + 4 * (int64_t)(stages - 4);
    int a_st = st_cur;
    const int64_t a_tile_step = a_tiled ? ((B256 - 1) * grps + 1) * 4096 : TILE_ROWS * ldw - 4 * (int64_t)(stages - 4);   // last group of a tile -> first of the next
    const int64_t a_group_step = a_tiled ? 4096 : 16;
    auto advance_a = [&]() {
        const bool tile_last = a_st + 4 == stages;
        const uint32_t* nx = a_ptr + (tile_last ? a_tile_step : a_group_step);
        a_st = tile_last ? 0 : a_st + 4;
        a_ptr = a_ptr == a_last ? a_ptr : nx;
    };
    // Through LDS by LDS-DMA: a group's words (16 KiB per workgroup, 2 KiB = two pieces per wave) are requested at the top of the
    // previous group and read back in its last stage; each wave fetches and reads only its own 32 rows, so no barrier is involved,
    // and ONE buffer is enough (the words in use live in registers).
    // piece p (0, 1) of this wave: rows 16 p .. 16 p + 15 of its 32, lane i -> row i >> 2, 16-byte chunk i & 3
    unsigned x_src[XP];
#pragma unroll
    for (int p_ = 0; p_ < XP; ++p_)
        x_src[p_] = a_tiled ? (unsigned)(p_ * 1024 + lane * 16) : (unsigned)((16 * p_ + (lane >> 2)) * ldw + 4 * (lane & 3)) * 4u;
    const unsigned x_rd = lds0 + (unsigned)(RING * STAGE_BYTES + (WROWS * wave + r) * 64 + g * 16);
    u32x4 aq[R], an[R];
    auto fetch_x = [&](u32x4 (&dst)[R]) {   // issue only; a later counted wait covers it, then tie_x()
#pragma unroll
        for (int mt = 0; mt < R; ++mt) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[mt]) : "v"(x_rd), "n"(16 * 64 * mt));
    };
    auto tie_x = [&](u32x4 (&dst)[R]) {
#pragma unroll
        for (int mt = 0; mt < R; ++mt) asm volatile("" : "+v"(dst[mt]));
    };

    // output scales, fetched before the pipeline starts (a load inside the loop would make the compiler drain the DMA queue)
    float osc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) osc[nt] = colscale[16 * nt + r];
    i32x4 acc[R][4][L];
    auto zero_acc = [&]() {
#pragma unroll
        for (int mt = 0; mt < R; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int l = 0; l < L; ++l) acc[mt][nt][l] = i32x4{0, 0, 0, 0};
    };
    // C/D layout of the 16x16 MFMA: column = lane & 15, row = 4 (lane >> 4) + i.  The digit planes are recombined in fp64.
    auto write_tile = [&](int tl, bool last_of_tile) {
        const int64_t tu = (int64_t)tl * stages;   // the logical slice that holds the tile's first unit
        const int first_wg = tu < big_end ? (int)(tu / u_big) : n_big + (int)((tu - big_end) / u_small);
        const int slot = slice - first_wg;
        const int64_t row_base = (int64_t)tl * TILE_ROWS + wave * WROWS;
        float* o = out + (int64_t)slot * slab_stride;
#pragma unroll
        for (int mt = 0; mt < R; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    long long v = 0;
#pragma unroll
                    for (int l = L - 1; l >= 0; --l) v = v * 256 + acc[mt][nt][l][i];
                    const int64_t row = row_base + 16 * mt + 4 * g + i;
#ifdef BMF_EXP_NOSTORE  // timing experiment only
                    if (v == 0x7fffffffffffll)
#endif
                    o[row * KP + 16 * nt + r] = (float)((double)v * (double)osc[nt]);
                }
        if (last_of_tile) {  // last contributor of this tile: the slab slots nobody writes must read as zero
            for (int z = slot + 1; z < slots; ++z) {
                float* oz = out + (int64_t)z * slab_stride;
#pragma unroll
                for (int mt = 0; mt < R; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int64_t row = row_base + 16 * mt + 4 * g + i;
                            oz[row * KP + 16 * nt + r] = 0.f;
                        }
            }
        }
    };

    // ---- LDS-DMA, hand-placed: SGPR base + per-lane 32-bit offset (no address arithmetic on the vector unit), M0 = LDS destination ----
    auto dma16 = [&](const void* sbase, unsigned m0v, unsigned voff) {
#ifndef BMF_EXP_NODMA
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(m0v), "v"(voff), "s"(sbase) : "memory");
#endif
    };
    int st_dma = st_cur;   // stage of the next DMA to issue; n_dma counts them (past the end the last stage is re-fetched)
    int n_dma = 0;
    const int8_t* d_base = P;   // panel base of the stage being fetched (wave-uniform)
    unsigned d_buf = 0;
    auto begin_dma = [&](int buf) {   // stage st_dma -> ring buffer buf; the pieces follow with dma_piece()
        d_base = P + (int64_t)st_dma * 128;
        d_buf = (unsigned)(buf * STAGE_BYTES);
        ++n_dma;
        const int nx = st_dma + 1 == stages ? 0 : st_dma + 1;
        st_dma = n_dma < n_units ? nx : st_dma;
    };
    auto dma_piece = [&](int i) { dma16(d_base, d_m0[i] + d_buf, d_off[i]); };
    auto issue_x = [&](int p_) {   // piece p_ of the group at a_ptr
#ifndef BMF_EXP_NOALOAD
        dma16(a_ptr, x_m0 + (unsigned)(p_ * 1024), x_src[p_]);
#endif
    };

    // ---- prologue: stages 0..2 of the run and the X words of the first group ----
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        begin_dma(b);
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) dma_piece(i);
    }
#pragma unroll
    for (int p_ = 0; p_ < XP; ++p_) issue_x(p_);
    advance_a();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    fetch_x(aq);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    tie_x(aq);
    zero_acc();

    // B fragments.  A UNIT = one k-step (64 reduction indices) of one 16-column tile = 2 L MFMAs on L fragments.  Four fragment
    // buffers, one per 16-column tile: the reads of unit u + 3 are issued before the MFMAs of unit u, and the wait in front of those is
    // COUNTED -- the 3 L youngest reads (units u + 1 .. u + 3) stay in flight, so a unit's fragments have 4 L MFMAs (two waves: 8 L)
    // to arrive and the LDS sees a steady stream of reads instead of eight waves' bursts.
    i32x4 bf[4][L];
    auto fetch_b = [&](auto SL, auto KS, auto NT) {
        constexpr int sl = decltype(SL)::value, ks = decltype(KS)::value, nt = decltype(NT)::value;
#ifndef BMF_EXP_NOLDS
        const unsigned ad = b_addr[sl >> 1][ks];
        i32x4 (&dst)[L] = bf[nt];   // (asm operands inside a generic lambda must be its own locals)
#pragma unroll
        for (int l = 0; l < L; ++l)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[l]) : "v"(ad), "n"((sl & 1) * STAGE_BYTES + (l * 64 + 16 * nt) * 128));
#else
        i32x4 (&dst)[L] = bf[nt];
#pragma unroll
        for (int l = 0; l < L; ++l) asm volatile("" : "+v"(dst[l]));
#endif
    };
    auto wait_b = [&](auto NT) {
        constexpr int nt = decltype(NT)::value;
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(3 * L) : "memory");
        i32x4 (&dst)[L] = bf[nt];
#pragma unroll
        for (int l = 0; l < L; ++l) asm volatile("" : "+v"(dst[l]));
    };
#ifdef BMF_EXP_NOLDS
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int l = 0; l < L; ++l) bf[nt][l] = i32x4{0x01020304, 0x05060708, 0x01020304, 0x05060708};
#endif
    // A operands of one k-step: dword e of row group mt = bits 4 ks + e (+ 8 b for byte b) of the stage's word
    i32x4 avA[R], avB[R];   // avA: k-step 0 of a stage, avB: k-step 1
    auto expand1 = [&](unsigned w, int sh) {
#ifdef BMF_EXP_NOVALU  // timing experiment only (wrong results)
        return (int)w;
#else
        return (int)((w >> sh) & 0x01010101u);
#endif
    };
#pragma unroll
    for (int mt = 0; mt < R; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) avA[mt][e] = expand1(aq[mt][0], e);
    fetch_b(ic<0>{}, ic<0>{}, ic<0>{});
    fetch_b(ic<0>{}, ic<0>{}, ic<1>{});
    fetch_b(ic<0>{}, ic<0>{}, ic<2>{});

    for (int gq = 0; gq < n_groups; ++gq) {
        auto unit = [&](auto T, auto U) {
            constexpr int t = decltype(T)::value, u = decltype(U)::value, ks = u >> 2, nt = u & 3;
            // fragments of unit u + 3 (the next stage's buffer has been complete and visible since the previous barrier)
            if constexpr (u + 3 < 8) fetch_b(ic<t>{}, ic<((u + 3) >> 2)>{}, ic<((u + 3) & 3)>{});
            else fetch_b(ic<((t + 1) & 3)>{}, ic<0>{}, ic<(u + 3 - 8)>{});
            wait_b(ic<nt>{});
            if (t == 3 && u == 3) tie_x(an);   // (older than every read still in flight here)
            __builtin_amdgcn_sched_barrier(0);
            // two dwords of the NEXT k-step's A operands (k-step 1 of this stage: bits 4..7 of the same words; k-step 0 of the next
            // stage: the next word, from the next group's words at t == 3), spread between this unit's MFMAs
            i32x4 (&ac)[R] = ks == 0 ? avA : avB;
            i32x4 (&ax)[R] = ks == 0 ? avB : avA;
#pragma unroll
            for (int q = 0; q < R; ++q) {
                constexpr int dbase = R * nt;
                const int d = dbase + q, mt = d >> 2, e = d & 3;
                const unsigned w = ks == 0 ? aq[mt][t] : (t == 3 ? an[mt][0] : aq[mt][(t + 1) & 3]);
                ax[mt][e] = expand1(w, (ks == 0 ? 4 : 0) + e);
            }
#pragma unroll
            for (int mt = 0; mt < R; ++mt)
#pragma unroll
                for (int l = 0; l < L; ++l)
                    acc[mt][nt][l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ac[mt], bf[nt][l], acc[mt][nt][l], 0, 0, 0);
#ifndef BMF_EXP_NOSCHED
#pragma unroll
            for (int q = 0; q < 2 * R; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            }
            if constexpr (R * L > 2 * R) __builtin_amdgcn_sched_group_barrier(0x008, R * L - 2 * R, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
            // this stage's panel pieces, one at a time behind the MFMAs of a unit (units 1, 3, 5, or 1..6 when a wave has more than
            // three); then (t == 0) the NEXT group's X words, half behind unit 6 and half behind unit 7
            if constexpr (DMA_PER_WAVE <= 3) {
                if constexpr ((u & 1) == 1 && (u >> 1) < DMA_PER_WAVE) dma_piece(u >> 1);
            } else {
                if constexpr (u >= 1 && u - 1 < DMA_PER_WAVE) dma_piece(u - 1);
            }
            if constexpr (t == 0 && u == 6) {
#pragma unroll
                for (int p_ = 0; p_ < XP / 2; ++p_) issue_x(p_);
            }
            if constexpr (t == 0 && u == 7) {
#pragma unroll
                for (int p_ = XP / 2; p_ < XP; ++p_) issue_x(p_);
                advance_a();
            }
        };
        auto stage = [&](auto T) {
            constexpr int t = decltype(T)::value;
            begin_dma((t + 3) & 3);      // stage t + 3 goes into the buffer stage t - 1 was read from
            if (t == 3) fetch_x(an);     // this wave's own pieces of the next group, complete since the wait of t == 2
            unit(T, ic<0>{}); unit(T, ic<1>{}); unit(T, ic<2>{}); unit(T, ic<3>{});
            unit(T, ic<4>{}); unit(T, ic<5>{}); unit(T, ic<6>{}); unit(T, ic<7>{});
            // End of stage u: stage u + 2 is fetched from (B fragments) from the middle of stage u + 1 on, so this wave's pieces of it --
            // issued during stage u - 1 -- must have landed before the barrier.  What was issued since may stay in flight: this stage's
            // pieces and, in t == 0 / 1, the two X pieces, which are issued AFTER the panel pieces of t == 0 so that they are
            // younger than them: vmcnt counts in issue order, and this way the X pieces (an HBM round trip) are only forced to
            // complete by the wait of t == 2, more than two stages after their issue.
            //   issue order: D0 X X | D1 | D2 | D3;  end of t = 0 needs D3' (younger: D0 X X), t = 1 needs D0 (X X D1), t = 2 needs D1 (D2)
            if (t <= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE + XP) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE) : "memory");
#ifndef BMF_EXP_NOBAR
            __builtin_amdgcn_s_barrier();
#endif
            asm volatile("" ::: "memory");
        };
        stage(ic<0>{});
        stage(ic<1>{});
        stage(ic<2>{});
        stage(ic<3>{});
        copy_words(aq, an, std::make_integer_sequence<int, R>{});   // (a loop here, even under "unroll", sends the arrays to scratch)
        const bool tile_end = st_cur + 4 == stages;
        if (tile_end || gq + 1 == n_groups) {
            write_tile(tile, tile_end);
            zero_acc();
        }
        tile += tile_end ? 1 : 0;
        st_cur = tile_end ? 0 : st_cur + 4;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus DMAs of the last stages
}

template <int L, int R, int WAVES>
int launch_i8w(const uint32_t* A, int64_t ldw, int a_tiled, int stages, const int8_t* P, int64_t ldp, float* out, int64_t slab_stride,
               const PlanI8& pl, int slots, const float* colscale, const int32_t* stop, hipStream_t s) {
    BMF_LAUNCH((xf_bits_i8w_kernel<L, R, WAVES>), dim3((unsigned)pl.grid), dim3(64 * WAVES), 0, s, A, ldw, a_tiled, stages, P, ldp, out, slab_stride,
               pl.n_big, pl.u_big, pl.u_small, pl.total, pl.n_slices, slots, colscale, stop, pl.perm);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

}  // namespace

// variant (bmf_i8_use_wide): 1 = 8 waves x (32 rows x 64 columns), two per SIMD; 2 = 4 waves x (64 x 64), one per SIMD;
// 3 = 4 waves x (128 x 64), one per SIMD, 512-row tiles
int bmf_xf_bits_i8w_launch(int variant, const uint32_t* A, int64_t ldw, int a_tiled, int stages, const int8_t* P, int64_t ldp, int limbs, float* out,
                           int64_t slab_stride, const PlanI8& pl, int slots, const float* colscale, const int32_t* stop, hipStream_t s) {
#define BMF_W_ARGS A, ldw, a_tiled, stages, P, ldp, out, slab_stride, pl, slots, colscale, stop, s
    // 4 = the anti-phase eight-wave kernel of xf_bits_i8p.hip (512 rows x 32 columns per workgroup, the column halves kept)
    if (variant == 4) return bmf_xf_bits_i8p_launch(A, a_tiled, stages, P, ldp, limbs, out, slab_stride, pl, slots, colscale, stop, s);
#ifndef BMF_W_VARIANTS
#define BMF_W_VARIANTS 3   // bit v - 1: variant v is compiled in (3: hipcc cannot allocate its 384 accumulator registers: -amdgpu-mfma-vgpr-form=1 crashes, without it 1900 registers spill)
#endif
#if BMF_W_VARIANTS & 2
    if (variant == 2) return limbs == 3 ? launch_i8w<3, 4, 4>(BMF_W_ARGS) : launch_i8w<2, 4, 4>(BMF_W_ARGS);
#endif
#if BMF_W_VARIANTS & 4
    if (variant == 3) return limbs == 3 ? launch_i8w<3, 8, 4>(BMF_W_ARGS) : launch_i8w<2, 8, 4>(BMF_W_ARGS);
#endif
#if BMF_W_VARIANTS & 1
    if (variant == 1) return limbs == 3 ? launch_i8w<3, 2, 8>(BMF_W_ARGS) : launch_i8w<2, 2, 8>(BMF_W_ARGS);
#endif
    bmf_set_error("bmf_xf_bits_i8w: variant %d is not compiled in", variant);
    return BMF_ERR_BAD_ARG;
#undef BMF_W_ARGS
}

int bmf_xf_bits_i8w_occupancy(int variant, int limbs, int* out) {
    (void)variant; (void)limbs;
    *out = 1;   // one workgroup per CU by design (LDS: ring + X words > 80 KiB)
    return BMF_OK;
}
