// K1/K2 on the 2:4 structured-sparse integer matrix instruction: out = bits(A) . F with F as signed 8-bit digit planes, EXACT
// int32 accumulation, HALF the matrix instructions of xf_bits_i8.hip for rows of A that are sparse enough.
//
//   X  @ V   (A = X bits,   digit planes of V)   replaces  multiply(W, X) @ V      PyBMF/models/BinaryMFPenalty.py:139
//   X^T @ U  (A = X^T bits, digit planes of U)   replaces  multiply(W, X).T @ U    PyBMF/models/BinaryMFPenalty.py:154
//
// v_smfmac_i32_16x16x128_i8 multiplies a 16 x 128 A whose every aligned group of four reduction indices holds at most TWO
// non-zeros (stored as the two values + two 2-bit positions) with a dense 128 x 16 B, at the issue rate of the dense
// v_mfma_i32_16x16x64_i8 (scripts/probes/smfmac_probe.hip).  A Boolean matrix of density p violates "two of four" in a share
// ~ 4 p^3 of its groups; the rest of the work of this file is about those:
//
//   * bmf_s24_pack: the "S24" form of a bit matrix -- per group of four cells the two value bits and two 2-bit positions of its
//     FIRST TWO ones (1.5 bits per cell), tiled like bmf_tile_bits -- plus the list of the ones that did not fit (third and fourth
//     of a group: "overflow"), as CSR over the packed rows.  Rows are packed through a row selection, so that the caller can keep
//     rows with many overflow ones out of this form altogether and hand them to the dense kernel (bmf_s24_classify).
//   * xf_bits_i8s_kernel: the stream-K GEMM of xf_bits_i8.hip (same plan, same slab slots, same LDS ring of digit-plane stages,
//     same fp64 recombination of the planes) on the S24 form; output rows go through the row map of the selection.
//   * s24_overflow_kernel: the overflow ones, exact: one wave per row gathers the quantised factor rows q = rint(F 2^e) -- the
//     SAME integer the digit planes hold, recomputed from the fp64 master with the builder's arithmetic -- sums them in int64 and
//     adds the scaled sum to slab slot 0 of that row (after the GEMM, same stream).
//
// Operand layout of the instruction (measured, probe above): lane (row r, group a) of A holds 16 compressed bytes ("slots") and 16
// two-bit positions (slot s at bits [2s, 2s+2) of the index register); slot pair (2j, 2j+1) selects among bytes 4j .. 4j+3 of
//   a = 0: B lane group 0 bytes 0..15 (slots 0..7), B lane group 1 bytes 0..15 (slots 8..15)
//   a = 1: B lane groups 2 / 3, bytes 0..15         a = 2: B lane groups 0 / 1, bytes 16..31         a = 3: groups 2 / 3, bytes 16..31
// With B lane (column c, group g) holding the two 16-byte fragments of the dense kernel's two k-steps (plane order
// bmf_panel_pos_i8: chunk ks * 4 + g of the stage, byte 4 (s & 3) + b for bit 8 b + s of word (g, t)), a "group of four" is the
// four bits {s, s + 8, s + 16, s + 24} of one 32-bit word of the bit matrix, and lives in A lane group (g >> 1) + 2 (s >> 2), slots
// 8 (g & 1) + 2 (s & 3) + {0, 1}.
#include "common.h"
#include "i8_plan.h"
#include "s24.h"

#include <type_traits>
#include <utility>

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

// One workgroup = 4 waves (one per SIMD) = 256 rows x 32 columns x L planes, two workgroups per CU, persistent, stream-K over
// (row tile, stage) in whole groups of four stages -- the skeleton of xf_bits_i8_kernel.  What differs:
//   * a stage (128 reduction indices) is ONE instruction per 16 x 16 tile and plane: 24 per wave and stage (dense: 48);
//   * B fragments are 32 bytes per lane (both 16-byte chunks of the dense kernel's two k-steps), fetched per (plane, column tile)
//     through a ring of three register sets two fragments ahead of their use, across the stage barrier;
//   * the A side of a group of four stages is 24 bytes per lane and 16-row group: four index dwords (one per stage) and two dwords
//     of value bits (one per stage pair), through LDS by LDS-DMA like the dense kernel's X words -- 24 KiB per workgroup and
//     group, ONE buffer (a group's words are in registers before the next group's DMA is issued; every wave fetches and reads
//     only its own rows);
//   * value bits expand with the dense kernel's (w >> s) & 0x01010101: 4 dwords per stage and row group instead of 8.
template <int L>
__global__ __launch_bounds__(256, 2) void xf_bits_i8s_kernel(const uint32_t* __restrict__ A, int stages,
                                                              const int8_t* __restrict__ P, int64_t ldp, int kp, int col_base, int halves,
                                                              float* __restrict__ out, int64_t slab_stride, int n_big, int u_big, int u_small,
                                                              int64_t total_units, int n_slices, int slots,
                                                              const float* __restrict__ colscale, const int32_t* __restrict__ rowmap,
                                                              const int32_t* __restrict__ stop, SlicePerm perm) {
    if (stop && *stop != 0) return;
    constexpr int TILE_ROWS = 256;
    constexpr int LROWS = L * 32;             // 128-byte LDS rows per stage
    constexpr int STAGE_BYTES = LROWS * 128;
    constexpr int PIECES = STAGE_BYTES / 1024;
    constexpr int DMA_PER_WAVE = PIECES / 4;
    static_assert(PIECES % 4 == 0, "every wave issues the same number of DMA pieces (the vmcnt bookkeeping counts on it)");
    constexpr int RING = 4;
    constexpr int XI_BYTES = TILE_ROWS * 64;  // index dwords of a group: 256 rows x 4 lane groups x 4 stages
    constexpr int XV_BYTES = TILE_ROWS * 32;  // value bits of a group:   256 rows x 4 lane groups x 2 stage pairs
    constexpr int XG_BYTES = XI_BYTES + XV_BYTES;
    static_assert(XG_BYTES == BMF_S24_GROUP_BYTES, "the S24 block of a (row tile, group)");
    static_assert(2 * (RING * STAGE_BYTES + XG_BYTES) <= 160 * 1024, "two workgroups' rings must fit the 160 KiB LDS");
    __shared__ __attribute__((aligned(16))) char smem[RING * STAGE_BYTES + XG_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // = 64-row group of the tile
    const int r = lane & 15, g = lane >> 4;
    // block -> (column half, slice): blocks b, b + 8, ... share an XCD (round-robin dispatch; a speed assumption only)
    const int bx = blockIdx.x & 7, bi = blockIdx.x >> 3;
    const int half = bi % halves;
    const int bslice = (bi / halves) * 8 + bx;
    if (bslice >= 512) return;
    const int slice = perm.p[bslice];
    if (slice >= n_slices) return;           // (0xFFFF: this workgroup has no slice)
    const int col0 = col_base + 32 * half;   // this workgroup's 32 columns of the kp-wide factor / output

    const int64_t big_end = (int64_t)n_big * u_big;
    const int64_t u0 = slice < n_big ? (int64_t)slice * u_big : big_end + (int64_t)(slice - n_big) * u_small;
    const int64_t u1 = min(u0 + (slice < n_big ? u_big : u_small), total_units);
    if (u0 >= u1) return;
    const int n_groups = (int)((u1 - u0) >> 2);
    const int n_units = n_groups << 2;

    // digit-plane stages -> LDS ring, exactly as in xf_bits_i8_kernel: DMA piece q = wave + 4 i (1 KiB): LDS rows 8q .. 8q+7 (row
    // R = limb * 32 + column); lane i fills physical 16-byte chunk i & 7 of row 8q + (i >> 3) with source chunk (i & 7) ^ ((R >> 1) & 7)
    const int8_t* dsrc[DMA_PER_WAVE];
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) {
        const int q = wave + 4 * i;
        const int limb = q >> 2, j0 = (q & 3) * 8, d_row = lane >> 3, d_chunk = lane & 7;
        const int R = 8 * q + d_row;
        dsrc[i] = P + (int64_t)(limb * kp + col0 + j0 + d_row) * ldp + ((d_chunk ^ ((R >> 1) & 7)) << 4);
    }
    auto issue_dma = [&](int stage, int buf) {
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i)
#ifdef BMF_EXP_NODMA   // timing experiments only (wrong results): -DBMF_EXP_NODMA / NOXDMA / NOBAR / NOLDS / NOVALU / NOMFMA
            if (stage < 0)
#endif
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dsrc[i] + (int64_t)stage * 128),
                                             (__attribute__((address_space(3))) void*)(smem + buf * STAGE_BYTES + (wave + 4 * i) * 1024), 16, 0, 0);
    };
    auto lds0_of = [](char* p_) { return (unsigned)(size_t)(__attribute__((address_space(3))) char*)p_; };
    const unsigned lds0 = lds0_of(smem);
    // B fragment of (16-column tile nt, limb l): row l*32 + 16 nt + r, physical chunks g ^ (r >> 1) (bytes 0..15 of the lane's 32)
    // and (4 + g) ^ (r >> 1) (bytes 16..31)
    const unsigned b_addr0 = lds0 + (unsigned)(r * 128) + (unsigned)(((g ^ (r >> 1)) & 7) << 4);
    const unsigned b_addr1 = lds0 + (unsigned)(r * 128) + (unsigned)((((4 + g) ^ (r >> 1)) & 7) << 4);

    // S24 words of this wave's 64 rows: block (tile, group) is 24 KiB = [256 rows][4 lane groups][4 stages] index dwords, then
    // [256 rows][4 lane groups][2 stage pairs] value dwords; one wave-uniform pointer walks the blocks
    int tile = (int)(u0 / stages);
    int st_cur = (int)(u0 - (int64_t)tile * stages);   // first stage of the group being computed (multiple of 4)
    const int64_t n_tiles_a = total_units / stages;
    const char* a_ptr = reinterpret_cast<const char*>(A) + ((int64_t)tile * (stages >> 2) + (st_cur >> 2)) * XG_BYTES;
    const char* const a_last = reinterpret_cast<const char*>(A) + (n_tiles_a * (stages >> 2) - 1) * XG_BYTES;
    auto advance_a = [&]() { a_ptr = a_ptr == a_last ? a_ptr : a_ptr + XG_BYTES; };
    char* const x_lds = smem + RING * STAGE_BYTES;
    // piece p of this wave: 0..3 = index dwords of its rows 16 p .. 16 p + 15 (1 KiB each), 4..5 = the two halves of its value dwords
    auto issue_x = [&](int p_) {
        const int src_off = p_ < 4 ? wave * 4096 + p_ * 1024 : XI_BYTES + wave * 2048 + (p_ - 4) * 1024;
#ifdef BMF_EXP_NOXDMA
        if (p_ < 0)
#endif
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_ptr + src_off + lane * 16),
                                         (__attribute__((address_space(3))) void*)(x_lds + src_off), 16, 0, 0);
    };
    const unsigned xi_rd = lds0 + (unsigned)(RING * STAGE_BYTES + (64 * wave + r) * 64 + g * 16);
    const unsigned xv_rd = lds0 + (unsigned)(RING * STAGE_BYTES + XI_BYTES + (64 * wave + r) * 32 + g * 8);
    u32x4 aqi[4];   // [16-row group][stage]: position dwords
    u32x2 aqv[4];   // [16-row group][stage pair]: value bits
    auto read_x = [&]() {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(aqi[mt]) : "v"(xi_rd), "n"(16 * 64 * mt));
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(aqv[mt]) : "v"(xv_rd), "n"(16 * 32 * mt));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            asm volatile("" : "+v"(aqi[mt]));
            asm volatile("" : "+v"(aqv[mt]));
        }
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
    // C/D layout of the 16x16 result: column = lane & 15, row = 4 (lane >> 4) + i.  The digit planes are recombined in fp64.  Rows go
    // through `rowmap` (packed row -> row of `out`; negative: a padding row of the packed form, nothing to write).
    auto write_tile = [&](int tl, bool last_of_tile) {
        const int64_t tu = (int64_t)tl * stages;
        const int first_wg = tu < big_end ? (int)(tu / u_big) : n_big + (int)((tu - big_end) / u_small);
        const int slot = slice - first_wg;
        const int64_t row_base = (int64_t)tl * TILE_ROWS + wave * 64;
        float* o = out + (int64_t)slot * slab_stride;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t prow = row_base + 16 * mt + 4 * g + i;
                const int64_t row = rowmap ? (int64_t)rowmap[prow] : prow;
                if (row < 0) continue;
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

    // ---- prologue: stages 0..2 of the run and the S24 words of the first group ----
    int st_dma = st_cur;
    int n_dma = 0;
    auto next_dma = [&](int buf) {
        issue_dma(st_dma, buf);
        ++n_dma;
        const int nx = st_dma + 1 == stages ? 0 : st_dma + 1;
        st_dma = n_dma < n_units ? nx : st_dma;
    };
    next_dma(0);
    next_dma(1);
    next_dma(2);
    next_dma(3);
#pragma unroll
    for (int p_ = 0; p_ < 6; ++p_) issue_x(p_);
    advance_a();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    read_x();
    zero_acc();

    // fragment f = 0..5 of a stage = (limb f >> 1, column tile f & 1); one register set per fragment, fetched a WHOLE STAGE ahead:
    // right after the instructions of (stage t, fragment f) have issued, set f takes (stage t + 1, fragment f).  (First version: a ring
    // of three sets two fragments ahead -- 8 instructions = 128 pipe cycles of cover against an LDS latency of ~500 cycles under this
    // load: every fragment was waited for.)
    i32x4 blo[2 * L], bhi[2 * L];
#ifdef BMF_EXP_NOLDS
#define BMF_FETCH_F(slot, f, ri) do { asm volatile("" : "+v"(blo[ri]), "+v"(bhi[ri])); } while (0)
#else
#define BMF_FETCH_F(slot, f, ri)                                                                                          \
    do {                                                                                                                  \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(blo[ri]) : "v"(b_addr0),                                      \
                     "n"((slot) * STAGE_BYTES + (((f) >> 1) * 32 + 16 * ((f) & 1)) * 128));                               \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bhi[ri]) : "v"(b_addr1),                                      \
                     "n"((slot) * STAGE_BYTES + (((f) >> 1) * 32 + 16 * ((f) & 1)) * 128));                               \
    } while (0)
#endif
    constexpr int NF = 2 * L;   // fragments per stage
#ifdef BMF_EXP_NOLDS
    for (int i = 0; i < NF; ++i) blo[i] = bhi[i] = i32x4{0x01020304, 0x05060708, 0x01020304, 0x05060708};
#endif
#pragma unroll
    for (int f = 0; f < NF; ++f) BMF_FETCH_F(0, f, f);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // buffer 0 is in registers: stage 4 may go into it
    __builtin_amdgcn_s_barrier();
    auto expand = [&](unsigned w, int sh) {
        i32x4 av;
#pragma unroll
#ifdef BMF_EXP_NOVALU
        for (int e = 0; e < 4; ++e) av[e] = (int)w;
#else
        for (int e = 0; e < 4; ++e) av[e] = (int)((w >> (sh + e)) & 0x01010101u);
#endif
        return av;
    };

    for (int gq = 0; gq < n_groups; ++gq) {
        i32x4 av[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) av[mt] = expand(aqv[mt][0], 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {   // (fully unrolled: t, and with it every ring slot, is a constant in each copy)
            // Stage t + 4 goes into the buffer of stage t ITSELF: its fragments have been in registers since the barrier that ended the
            // stage before (every wave drains its fragment reads there).  Three stages of lead for the plane pieces instead of two: with
            // two, a stage could not be shorter than half the L2 -> LDS latency under load (~1.6 us), which was the ~0.8 us per stage --
            // 196-200 us per launch -- that every earlier form of this kernel measured, whatever else was changed
            // (profiles/r05_i8_smfmac.md).
            next_dma(t);
            // the NEXT group's S24 words into the X buffer (this group's are in registers): three pieces in each of the first two
            // stages, AFTER the stage's panel pieces (see the waits below)
            if (t == 0) { issue_x(0); issue_x(1); issue_x(2); }
            if (t == 1) { issue_x(3); issue_x(4); issue_x(5); advance_a(); }
            i32x4 avn[4];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                // (stage t, fragment f) is the oldest of the six fragments in flight: the five younger ones may stay there
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (NF - 1)) : "memory");
                asm volatile("" : "+v"(blo[f]), "+v"(bhi[f]));
                __builtin_amdgcn_sched_barrier(0);
                const i32x8 b8 = __builtin_shufflevector(blo[f], bhi[f], 0, 1, 2, 3, 4, 5, 6, 7);
                const int l = f >> 1, nt = f & 1;
                // the next stage's value bytes: row group f's four dwords ride between the instructions of fragments 0..3
                if (t < 3 && f < 4) avn[f] = expand(aqv[f][(t + 1) >> 1], 4 * ((t + 1) & 1));
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#ifdef BMF_EXP_NOMFMA
                    asm volatile("" : "+v"(acc[mt][nt][l]) : "v"(av[mt]), "v"(b8), "v"(aqi[mt][t]));
#else
                    acc[mt][nt][l] = __builtin_amdgcn_smfmac_i32_16x16x128_i8(av[mt], b8, acc[mt][nt][l], (int)aqi[mt][t], 0, 0);
#endif
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                // the same fragment of the NEXT stage into the set just consumed (its buffer has been complete and visible since the
                // previous barrier)
                BMF_FETCH_F((t + 1) & 3, f, f);
            }
            if (t < 3) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) av[mt] = avn[mt];
            }
            // End of stage u: the next stage's fragments are all requested -- drain them (the buffer they come from is overwritten
            // right after the barrier); and this wave's pieces of stage u + 2 -- issued at the top of stage u - 2 -- must have landed.
            // Issue order of a group: D X X X | D X X X | D | D.  Younger than those pieces, and allowed to stay in flight: at t = 0:
            // D(t3') D X X X; t = 1: D X X X D X X X; t = 2: X X X D X X X D; t = 3: only D D -- the S24 pieces of t = 1 must be complete
            // by the end of the group.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (t == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA_PER_WAVE + 3) : "memory");
            else if (t == 1 || t == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA_PER_WAVE + 6) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA_PER_WAVE) : "memory");
#ifndef BMF_EXP_NOBAR
            __builtin_amdgcn_s_barrier();
#endif
            asm volatile("" ::: "memory");
        }
        read_x();   // the next group's words: this wave's own pieces, complete since the wait of t == 3
        const bool tile_end = st_cur + 4 == stages;
        if (tile_end || gq + 1 == n_groups) {
            write_tile(tile, tile_end);
            zero_acc();
        }
        tile += tile_end ? 1 : 0;
        st_cur = tile_end ? 0 : st_cur + 4;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the surplus DMAs and fragment reads of the last stages
#undef BMF_FETCH_F
}

// The same kernel as ONE workgroup of EIGHT waves per CU (round 5, second form).  Why: the counters of the 4-wave form say it is bound by
// its data-movement skeleton, not by the matrix pipe or by power (pipe 51 % busy at 1.96 GHz; with the matrix instructions removed it
// keeps 70-80 % of its time: profiles/r05_i8_smfmac.md).  Here
//   * a workgroup owns 512 rows x 32 columns: the 12 KiB of digit planes of a stage serve eight waves instead of four -- half the
//     plane bytes through LDS-DMA per matrix instruction;
//   * the DMA roles are split between the waves: waves 0-3 issue the plane pieces (three per stage each, L2 latency, a counted vmcnt
//     per stage as before), waves 4-7 issue the S24 pieces (twelve per group each, HBM latency) and wait for them ONCE per group --
//     the planes' counted waits no longer inherit the HBM latency of S24 pieces ahead of them in one in-order vmcnt queue;
//   * with all of the CU's 160 KiB of LDS for one workgroup the S24 words are double-buffered by group (2 x 48 KiB beside the 48-KiB
//     plane ring): a group's words are requested a whole group (four stages) before they are read.
// The price is the lockstep of eight waves at every stage barrier (the dense kernel lost 9 % to it in round 2).
// PP = 1 (form 2): the two wave groups in anti-phase, see the comment in the body.  Measured: forms 0, 1 and 2 all take 197-212 us
// at the headline shape (profiles/r05_i8_smfmac.md, second series) -- a wave's per-stage chain of non-matrix work bounds them all.
template <int L, int PP>
__global__ __launch_bounds__(512, 1) void xf_bits_i8s8_kernel(const uint32_t* __restrict__ A, int stages,
                                                               const int8_t* __restrict__ P, int64_t ldp, int kp, int col_base, int halves,
                                                               float* __restrict__ out, int64_t slab_stride, int u_len,
                                                               int64_t total_units, int n_slices, int slots,
                                                               const float* __restrict__ colscale, const int32_t* __restrict__ rowmap,
                                                               const int32_t* __restrict__ stop, SlicePerm perm) {
    if (stop && *stop != 0) return;
    static_assert(L == 3, "three digit planes");
    constexpr int TILE_ROWS = 512;
    constexpr int STAGE_BYTES = L * 32 * 128;   // 12 KiB
    constexpr int DMA_PER_WAVE = STAGE_BYTES / 1024 / 4;   // plane pieces per issuing wave and stage (waves 0-3)
    constexpr int RING = 4;
    constexpr int XB = BMF_S24_GROUP_BYTES;     // one 256-row S24 block of a group: 16 KiB of position dwords + 8 KiB of value dwords
    constexpr int XG_BYTES = 2 * XB;            // a 512-row tile's group
    static_assert(RING * STAGE_BYTES + 2 * XG_BYTES <= 160 * 1024, "the plane ring and two S24 groups must fit the 160 KiB LDS");
    __shared__ __attribute__((aligned(16))) char smem[RING * STAGE_BYTES + 2 * XG_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0..7 = 64-row group of the tile
    const bool plane_role = wave < 4;                                    // waves 0-3 fetch planes, waves 4-7 fetch S24 words
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

    // ---- plane pieces (waves 0-3): piece q = wave + 4 i, as in the four-wave kernel ----
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
#ifdef BMF_EXP_NODMA
            if (stage < 0)
#endif
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dsrc[i] + (int64_t)stage * 128),
                                             (__attribute__((address_space(3))) void*)(smem + buf * STAGE_BYTES + ((wave & 3) + 4 * i) * 1024), 16, 0, 0);
    };
    auto lds0_of = [](char* p_) { return (unsigned)(size_t)(__attribute__((address_space(3))) char*)p_; };
    const unsigned lds0 = lds0_of(smem);
    const unsigned b_addr0 = lds0 + (unsigned)(r * 128) + (unsigned)(((g ^ (r >> 1)) & 7) << 4);
    const unsigned b_addr1 = lds0 + (unsigned)(r * 128) + (unsigned)((((4 + g) ^ (r >> 1)) & 7) << 4);

    // ---- S24 pieces (waves 4-7): issuing wave j = wave - 4 fetches the words of the computing waves 2 j and 2 j + 1, both in 256-row
    // block j >> 1 of the tile; per computing wave w' = (c & 3): four 1-KiB pieces of position dwords at w' * 4096 + p * 1024 and two of
    // value dwords at 16384 + w' * 2048 + p * 1024 of its 24-KiB block (the S24 layout) ----
    int tile = (int)(u0 / stages);
    int st_cur = (int)(u0 - (int64_t)tile * stages);
    const int64_t n_tiles_a = total_units / stages;
    // (tile, group) of the next S24 group to request; clamps at the last group of the matrix (re-fetching it is harmless)
    int xq_tile = tile, xq_grp = st_cur >> 2;
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
    const int xj = wave & 3;                       // issuing wave's index
    const int x_blk = xj >> 1;                     // 256-row block of its two computing waves
    auto x_piece_off = [&](int k) {                // piece k = 0..11 of an issuing wave: byte offset inside its 24-KiB block
        const int c = k / 6, p_ = k % 6;
        const int wq = ((2 * xj) & 3) + c;
        return p_ < 4 ? wq * 4096 + p_ * 1024 : BMF_S24_IDX_BYTES + wq * 2048 + (p_ - 4) * 1024;
    };
    char* const x_lds = smem + RING * STAGE_BYTES;
    auto issue_x = [&](int xbuf, int k) {          // piece k of the group (xq_tile, xq_grp) into S24 buffer xbuf
        const int off = x_piece_off(k);
#ifdef BMF_EXP_NOXDMA
        if (k < 0)
#endif
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(x_block_ptr(xq_tile, xq_grp, x_blk) + off + lane * 16),
                                         (__attribute__((address_space(3))) void*)(x_lds + xbuf * XG_BYTES + x_blk * XB + off), 16, 0, 0);
    };
    // this computing wave's words: 256-row block wave >> 2, rows 64 (wave & 3) .. + 63 of it
    const unsigned xi_rd = lds0 + (unsigned)(RING * STAGE_BYTES + (wave >> 2) * XB + (64 * (wave & 3) + r) * 64 + g * 16);
    const unsigned xv_rd = lds0 + (unsigned)(RING * STAGE_BYTES + (wave >> 2) * XB + BMF_S24_IDX_BYTES + (64 * (wave & 3) + r) * 32 + g * 8);
    u32x4 aqi[4];
    u32x2 aqv[4];
    auto read_x = [&](int xbuf) {
        const unsigned bi_ = xi_rd + (unsigned)(xbuf * XG_BYTES), bv_ = xv_rd + (unsigned)(xbuf * XG_BYTES);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(aqi[mt]) : "v"(bi_), "n"(16 * 64 * mt));
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(aqv[mt]) : "v"(bv_), "n"(16 * 32 * mt));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            asm volatile("" : "+v"(aqi[mt]));
            asm volatile("" : "+v"(aqv[mt]));
        }
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
    auto write_tile = [&](int tl, bool last_of_tile) {
        const int64_t tu = (int64_t)tl * stages;
        const int slot = slice - (int)(tu / u_len);
        const int64_t row_base = (int64_t)tl * TILE_ROWS + wave * 64;
        float* o = out + (int64_t)slot * slab_stride;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t prow = row_base + 16 * mt + 4 * g + i;
                const int64_t row = rowmap ? (int64_t)rowmap[prow] : prow;
                if (row < 0) continue;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    long long v = 0;
#pragma unroll
                    for (int l = L - 1; l >= 0; --l) v = v * 256 + acc[mt][nt][l][i];
                    o[row * kp + col0 + 16 * nt + r] = (float)((double)v * (double)osc[nt]);
                    if (last_of_tile)
                        for (int z = slot + 1; z < slots; ++z) out[(int64_t)z * slab_stride + row * kp + col0 + 16 * nt + r] = 0.f;
                }
            }
    };

    // ---- prologue: plane stages 0..2; the S24 words of groups 0 and 1 of the run ----
    int st_dma = st_cur;
    int n_dma = 0;
    auto next_dma = [&](int buf) {
        if (plane_role) issue_dma(st_dma, buf);
        ++n_dma;
        const int nx = st_dma + 1 == stages ? 0 : st_dma + 1;
        st_dma = n_dma < n_units ? nx : st_dma;
    };
    next_dma(0);
    next_dma(1);
    next_dma(2);
    next_dma(3);
    if (!plane_role) {
#pragma unroll
        for (int k = 0; k < 12; ++k) issue_x(0, k);
    }
    advance_xq();
    if (!plane_role) {
#pragma unroll
        for (int k = 0; k < 12; ++k) issue_x(1, k);
    }
    advance_xq();   // (xq now names group 2 of the run, requested during group 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    read_x(0);
    zero_acc();

    i32x4 blo[2 * L], bhi[2 * L];   // one register set per fragment, fetched a whole stage ahead (see the four-wave kernel)
#ifdef BMF_EXP_NOLDS
#define BMF_FETCH_F8(slot, f, ri) do { asm volatile("" : "+v"(blo[ri]), "+v"(bhi[ri])); } while (0)
#else
#define BMF_FETCH_F8(slot, f, ri)                                                                                         \
    do {                                                                                                                  \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(blo[ri]) : "v"(b_addr0),                                      \
                     "n"((slot) * STAGE_BYTES + (((f) >> 1) * 32 + 16 * ((f) & 1)) * 128));                               \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bhi[ri]) : "v"(b_addr1),                                      \
                     "n"((slot) * STAGE_BYTES + (((f) >> 1) * 32 + 16 * ((f) & 1)) * 128));                               \
    } while (0)
#endif
    constexpr int NF = 2 * L;
#ifdef BMF_EXP_NOLDS
    for (int i = 0; i < NF; ++i) blo[i] = bhi[i] = i32x4{0x01020304, 0x05060708, 0x01020304, 0x05060708};
#endif
    auto expand = [&](unsigned w, int sh) {
        i32x4 av;
#pragma unroll
#ifdef BMF_EXP_NOVALU
        for (int e = 0; e < 4; ++e) av[e] = (int)w;
#else
        for (int e = 0; e < 4; ++e) av[e] = (int)((w >> (sh + e)) & 0x01010101u);
#endif
        return av;
    };

    if constexpr (PP) {
        // ---- ping-pong (form 2): waves 0-3 (group A) and 4-7 (group B) share the SIMDs pairwise and work in ANTI-PHASE, a workgroup
        // barrier between the phases.  In a phase one group issues nothing but the 24 matrix instructions of a stage, back to back, from
        // registers; the other group does everything else for ITS next stage -- the six fragment reads, the value-bit expansion, its
        // LDS-DMA pieces, the waits, and at a tile end the write-out.  The instructions that stall an issuing wave (a DMA piece costs it
        // 30-100 cycles) then sit beside the partner's matrix instructions instead of between the wave's own.
        //   phase 2u:     A computes stage u            | B loads stage u (reads X words at a group start, issues S24 pieces)
        //   phase 2u + 1: A loads stage u + 1 (issues the plane pieces of stage u + 4) | B computes stage u
        // Plane ring: stage u lives in buffer u % 4; A reads it in phase 2u - 1, B in phase 2u, and A refills it (stage u + 4) in phase
        // 2u + 1.  A waits for its pieces at the end of its compute phase 2u: stage u + 1 (issued in phase 2u - 5) must have landed,
        // the six pieces of stages u + 2, u + 3 may stay in flight.  S24 words: as in the lockstep form, two buffers of a whole group;
        // B requests group gq + 2 in phases 2, 4, 6 of group gq (both groups have read buffer gq & 1 by the end of phase 0) and waits
        // at the end of phase 6 for group gq + 1, which A reads in phase 7.
        const bool grp_a = plane_role;
        i32x4 av[4];
// (the fragment reads carry a group tag in their text: identical asm statements of the two branches would be merged behind a phi of
// their immediates, which then are no constants any more; and the four stages of a group are written out -- BMF_PP_STAGE(0..3) -- so
// that every ring slot is a literal)
#define BMF_PP_FETCH(tag, slot, f)                                                                                        \
        do {                                                                                                              \
            asm volatile("ds_read_b128 %0, %1 offset:%2 ; " tag : "=v"(blo[f]) : "v"(b_addr0),                            \
                         "n"((slot) * STAGE_BYTES + (((f) >> 1) * 32 + 16 * ((f) & 1)) * 128));                           \
            asm volatile("ds_read_b128 %0, %1 offset:%2 ; " tag : "=v"(bhi[f]) : "v"(b_addr1),                            \
                         "n"((slot) * STAGE_BYTES + (((f) >> 1) * 32 + 16 * ((f) & 1)) * 128));                           \
        } while (0)
#ifdef BMF_EXP_NOLDS
#define BMF_PP_FETCH6(tag, slot) do { } while (0)
#else
#define BMF_PP_FETCH6(tag, slot)                                                                                          \
        do {                                                                                                              \
            BMF_PP_FETCH(tag, slot, 0); BMF_PP_FETCH(tag, slot, 1); BMF_PP_FETCH(tag, slot, 2);                           \
            BMF_PP_FETCH(tag, slot, 3); BMF_PP_FETCH(tag, slot, 4); BMF_PP_FETCH(tag, slot, 5);                           \
        } while (0)
#endif
#define BMF_PP_LOAD(tag, slot, tq)                                                                                        \
        do {                                                                                                              \
            BMF_PP_FETCH6(tag, slot);                                                                                     \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) av[mt] = expand(aqv[mt][(tq) >> 1], 4 * ((tq) & 1));          \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
            _Pragma("unroll") for (int f = 0; f < NF; ++f) asm volatile("" : "+v"(blo[f]), "+v"(bhi[f]));                 \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
        } while (0)
#ifdef BMF_EXP_NOMFMA
#define BMF_PP_MMA(mt, nt, l, tq) asm volatile("" : "+v"(acc[mt][nt][l]) : "v"(av[mt]), "v"(b8), "v"(aqi[mt][tq]))
#else
#define BMF_PP_MMA(mt, nt, l, tq) acc[mt][nt][l] = __builtin_amdgcn_smfmac_i32_16x16x128_i8(av[mt], b8, acc[mt][nt][l], (int)aqi[mt][tq], 0, 0)
#endif
#define BMF_PP_COMPUTE(tq)                                                                                                \
        do {                                                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
            _Pragma("unroll") for (int f = 0; f < NF; ++f) {                                                              \
                const i32x8 b8 = __builtin_shufflevector(blo[f], bhi[f], 0, 1, 2, 3, 4, 5, 6, 7);                          \
                _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) BMF_PP_MMA(mt, f & 1, f >> 1, tq);                        \
            }                                                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
        } while (0)
#ifdef BMF_EXP_NOBAR
#define BMF_PP_BARRIER() asm volatile("" ::: "memory")
#else
#define BMF_PP_BARRIER() do { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#endif
#define BMF_PP_STAGE_A(t)                                                                                                 \
        do {                                                                                                              \
            BMF_PP_COMPUTE(t);                                                                                            \
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA_PER_WAVE) : "memory");                                       \
            BMF_PP_BARRIER(); /* ---- end of phase 2u */                                                                  \
            if ((t) == 3 && flush) {                                                                                      \
                write_tile(tile, tile_end);                                                                               \
                zero_acc();                                                                                               \
            }                                                                                                             \
            next_dma(t);                                                                                                  \
            if ((t) == 3 && gq + 1 < n_groups) read_x(xbuf ^ 1);                                                          \
            BMF_PP_LOAD("A", ((t) + 1) & 3, ((t) + 1) & 3);                                                               \
            BMF_PP_BARRIER(); /* ---- end of phase 2u + 1 */                                                              \
        } while (0)
#define BMF_PP_STAGE_B(t)                                                                                                 \
        do {                                                                                                              \
            if ((t) == 0 && gq > 0) read_x(xbuf);                                                                         \
            if ((t) >= 1) {                                                                                               \
                _Pragma("unroll") for (int k = 0; k < 4; ++k) issue_x(xbuf, 4 * ((t) - 1) + k);                            \
            }                                                                                                             \
            BMF_PP_LOAD("B", t, t);                                                                                       \
            if ((t) == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                                               \
            BMF_PP_BARRIER(); /* ---- end of phase 2u */                                                                  \
            BMF_PP_COMPUTE(t);                                                                                            \
            if ((t) == 3 && flush) {                                                                                      \
                write_tile(tile, tile_end);                                                                               \
                zero_acc();                                                                                               \
            }                                                                                                             \
            BMF_PP_BARRIER(); /* ---- end of phase 2u + 1 */                                                              \
        } while (0)
        // (two loops, not one loop with a branch per stage: the accumulators, fragments and value bytes would otherwise meet in phis
        // after every stage, and the register allocator answered that with 700 spills)
        if (grp_a) {
            BMF_PP_LOAD("A", 0, 0);
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
#undef BMF_PP_STAGE_A
#undef BMF_PP_STAGE_B
#undef BMF_PP_FETCH
#undef BMF_PP_FETCH6
#undef BMF_PP_LOAD
#undef BMF_PP_MMA
#undef BMF_PP_COMPUTE
#undef BMF_PP_BARRIER
    } else {
#pragma unroll
        for (int f = 0; f < NF; ++f) BMF_FETCH_F8(0, f, f);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int gq = 0; gq < n_groups; ++gq) {
            const int xbuf = gq & 1;   // the S24 buffer this group's words came from: free again once every wave has read it
            i32x4 av[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) av[mt] = expand(aqv[mt][0], 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                next_dma(t);   // stage t + 4 into the buffer of stage t itself (in registers since the last barrier): three stages of lead
                // group gq + 2's S24 words into the buffer this group was read from -- after the barrier of stage 0, behind which every
                // wave has its words in registers: four pieces in each of stages 1, 2, 3
                if (!plane_role && t >= 1) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) issue_x(xbuf, 4 * (t - 1) + k);
                }
                i32x4 avn[4];
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (NF - 1)) : "memory");
                    asm volatile("" : "+v"(blo[f]), "+v"(bhi[f]));
                    __builtin_amdgcn_sched_barrier(0);
                    const i32x8 b8 = __builtin_shufflevector(blo[f], bhi[f], 0, 1, 2, 3, 4, 5, 6, 7);
                    const int l = f >> 1, nt = f & 1;
                    if (t < 3 && f < 4) avn[f] = expand(aqv[f][(t + 1) >> 1], 4 * ((t + 1) & 1));
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#ifdef BMF_EXP_NOMFMA
                        asm volatile("" : "+v"(acc[mt][nt][l]) : "v"(av[mt]), "v"(b8), "v"(aqi[mt][t]));
#else
                        acc[mt][nt][l] = __builtin_amdgcn_smfmac_i32_16x16x128_i8(av[mt], b8, acc[mt][nt][l], (int)aqi[mt][t], 0, 0);
#endif
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    BMF_FETCH_F8((t + 1) & 3, f, f);
                }
                if (t < 3) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) av[mt] = avn[mt];
                }
                // End of stage u: drain the next stage's fragment reads (their buffer is overwritten after the barrier).  Plane waves: their
                // pieces of stage u + 2 (issued at the top of stage u - 2) must have landed; the six of the last two stages may stay in
                // flight.  S24 waves wait once per group, at t = 3: the next group's twelve pieces (requested a whole
                // group ago) must have landed; the twelve requested during this group may stay in flight.
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (plane_role) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA_PER_WAVE) : "memory");
                else if (t == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
#ifndef BMF_EXP_NOBAR
                __builtin_amdgcn_s_barrier();
#endif
                asm volatile("" ::: "memory");
            }
            advance_xq();
            read_x(xbuf ^ 1);   // the next group's words (complete and visible since the barrier of t = 3)
            const bool tile_end = st_cur + 4 == stages;
            if (tile_end || gq + 1 == n_groups) {
                write_tile(tile, tile_end);
                zero_acc();
            }
            tile += tile_end ? 1 : 0;
            st_cur = tile_end ? 0 : st_cur + 4;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
#undef BMF_FETCH_F8
}

}  // namespace

namespace {

// overflow ones of a word: per bit lane s, the cells {s, s + 8, s + 16, s + 24}; a group with three ones leaves one over, with four, two
__device__ __forceinline__ unsigned s24_word_extra(uint32_t x) {
    const unsigned t0 = x & 255u, t1 = (x >> 8) & 255u, t2 = (x >> 16) & 255u, t3 = x >> 24;
    const unsigned ge3 = (t0 & t1 & t2) | (t0 & t1 & t3) | (t0 & t2 & t3) | (t1 & t2 & t3);
    return (unsigned)__builtin_popcount(ge3) + (unsigned)__builtin_popcount(t0 & t1 & t2 & t3);
}

// counts[row] = overflow ones of that row of the bit matrix (one wave per row)
__global__ __launch_bounds__(256) void s24_count_kernel(const uint32_t* __restrict__ bits, int64_t rows, int64_t ldw, int64_t red_words,
                                                         int32_t* __restrict__ counts) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    unsigned n = 0;
    for (int64_t w = lane; w < red_words; w += 64) n += s24_word_extra(bits[row * ldw + w]);
    n = wave_sum(n);
    if (lane == 0) counts[row] = (int32_t)n;
}

// One thread per (packed row, 512-block, word pair h): the idx / val dwords of lane groups h and h + 2, the overflow ones of its
// eight words, optionally the words with those ones cleared (`kept`, the plain layout of `bits` over the PACKED rows).
__global__ __launch_bounds__(256) void s24_pack_kernel(const uint32_t* __restrict__ bits, int64_t ldw, int groups, const int32_t* __restrict__ rowsel,
                                                        int64_t n_threads, uint32_t* __restrict__ s24, const int64_t* __restrict__ ovf_ptr,
                                                        int32_t* __restrict__ ovf_cursor, int32_t* __restrict__ ovf_idx,
                                                        uint32_t* __restrict__ kept, int64_t ldk) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_threads; i += (int64_t)gridDim.x * 256) {
        const int h = (int)(i & 1);
        const int64_t rg = i >> 1;                 // (tile * groups + grp) * 256 + row
        const int row = (int)(rg & 255);
        const int64_t tg = rg >> 8;
        const int64_t tile = tg / groups, grp = tg - tile * groups;
        const int64_t prow = tile * 256 + row;
        const int64_t src = rowsel ? (int64_t)rowsel[prow] : prow;
        uint32_t w[2][4], idx[2][4], val[2][2], kp_[2][4];
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (src >= 0) v = *reinterpret_cast<const u32x4*>(bits + src * ldw + 16 * grp + 4 * (2 * h + gi));
#pragma unroll
            for (int t = 0; t < 4; ++t) w[gi][t] = v[t];
        }
        const unsigned extra = bmf_s24_encode_pair(w, idx, val, kp_);
        char* blk = reinterpret_cast<char*>(s24) + tg * (int64_t)BMF_S24_GROUP_BYTES;
#pragma unroll
        for (int ai = 0; ai < 2; ++ai) {
            const int a = h + 2 * ai;
            *reinterpret_cast<u32x4*>(blk + row * 64 + a * 16) = u32x4{idx[ai][0], idx[ai][1], idx[ai][2], idx[ai][3]};
            *reinterpret_cast<u32x2*>(blk + BMF_S24_IDX_BYTES + row * 32 + a * 8) = u32x2{val[ai][0], val[ai][1]};
        }
        if (kept) {
#pragma unroll
            for (int gi = 0; gi < 2; ++gi)
                *reinterpret_cast<u32x4*>(kept + prow * ldk + 16 * grp + 4 * (2 * h + gi)) = u32x4{kp_[gi][0], kp_[gi][1], kp_[gi][2], kp_[gi][3]};
        }
        if (ovf_idx && extra) {
            int64_t at = ovf_ptr[prow] + atomicAdd(ovf_cursor + prow, (int)extra);
#pragma unroll
            for (int gi = 0; gi < 2; ++gi)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    uint32_t over = w[gi][t] & ~kp_[gi][t];
                    while (over) {
                        const int bit = __builtin_ctz(over);
                        over &= over - 1;
                        ovf_idx[at++] = (int32_t)(512 * grp + 128 * (2 * h + gi) + 32 * t + bit);
                    }
                }
        }
    }
}

// The overflow ones of the packed rows, exact.  One wave per packed row, lane = factor column (kp <= 64): q = the integer the
// digit planes hold for (factor row j, column c) -- rint(clamp(F64 * 2^e)), e from the colscale the GEMM uses, the arithmetic of
// make_panel_i8_kernel -- summed in int64 over the row's list, scaled once, added to slab slot 0 of the row's output row.
__global__ __launch_bounds__(256) void s24_overflow_kernel(const int64_t* __restrict__ ovf_ptr, const int32_t* __restrict__ ovf_idx,
                                                            const int32_t* __restrict__ rowmap, int64_t prows, const double* __restrict__ F64,
                                                            int64_t ldf, const float* __restrict__ colscale, int kp, int col0, int ncols,
                                                            float* __restrict__ out, const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    const int64_t prow = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (prow >= prows) return;
    const int lane = threadIdx.x & 63;
    const int64_t b = ovf_ptr[prow], e = ovf_ptr[prow + 1];
    if (b == e) return;
    const int64_t row = rowmap ? (int64_t)rowmap[prow] : prow;
    if (row < 0 || lane >= ncols) return;
    const int c = col0 + lane;
    const double osc = (double)colscale[c], sc = 1.0 / osc;
    long long acc = 0;
    int64_t p = b;
    for (; p + 4 <= e; p += 4) {   // four gathers in flight
        const int j0 = ovf_idx[p], j1 = ovf_idx[p + 1], j2 = ovf_idx[p + 2], j3 = ovf_idx[p + 3];
        const double f0 = F64[(int64_t)j0 * ldf + c], f1 = F64[(int64_t)j1 * ldf + c], f2 = F64[(int64_t)j2 * ldf + c], f3 = F64[(int64_t)j3 * ldf + c];
        acc += (long long)__double2int_rn(fmax(fmin(f0 * sc, 8355711.0), -8355711.0));
        acc += (long long)__double2int_rn(fmax(fmin(f1 * sc, 8355711.0), -8355711.0));
        acc += (long long)__double2int_rn(fmax(fmin(f2 * sc, 8355711.0), -8355711.0));
        acc += (long long)__double2int_rn(fmax(fmin(f3 * sc, 8355711.0), -8355711.0));
    }
    for (; p < e; ++p) acc += (long long)__double2int_rn(fmax(fmin(F64[(int64_t)ovf_idx[p] * ldf + c] * sc, 8355711.0), -8355711.0));
    float* o = out + row * kp + c;
    *o = (float)((double)*o + (double)acc * osc);
}

}  // namespace

extern "C" int64_t bmf_s24_bytes(int64_t rows_pad, int64_t red_words) {
    if (rows_pad <= 0 || rows_pad % 256 || red_words <= 0 || red_words % 16) return -1;
    return (rows_pad / 256) * (red_words / 16) * (int64_t)BMF_S24_GROUP_BYTES;
}

extern "C" int bmf_s24_count(const uint32_t* bits, int64_t rows, int64_t ldw, int64_t red_words, int32_t* counts, void* stream) {
    BMF_REQUIRE(bits && counts, "bmf_s24_count: null pointer");
    BMF_REQUIRE(rows > 0 && red_words > 0 && ldw >= red_words, "bmf_s24_count: bad shape");
    BMF_LAUNCH(s24_count_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, bits, rows, ldw, red_words, counts);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_s24_pack(const uint32_t* bits, int64_t ldw, int64_t red_words, const int32_t* rowsel, int64_t rows_pad_s, uint32_t* s24,
                            const int64_t* ovf_ptr, int32_t* ovf_cursor, int32_t* ovf_idx, uint32_t* kept, int64_t ldk, void* stream) {
    BMF_REQUIRE(bits && s24, "bmf_s24_pack: null pointer");
    BMF_REQUIRE(rows_pad_s > 0 && rows_pad_s % 256 == 0 && red_words > 0 && red_words % 16 == 0 && ldw >= red_words && ldw % 4 == 0,
                "bmf_s24_pack: rows_pad_s must be a multiple of 256, red_words of 16, ldw >= red_words and a multiple of 4");
    BMF_REQUIRE(!ovf_idx || (ovf_ptr && ovf_cursor), "bmf_s24_pack: the overflow list needs its row pointers and a zeroed cursor array");
    BMF_REQUIRE(!kept || (ldk >= red_words && ldk % 4 == 0 && bmf_aligned16(kept)), "bmf_s24_pack: bad `kept` buffer");
    BMF_REQUIRE(bmf_aligned16(bits) && bmf_aligned16(s24), "bmf_s24_pack: pointers must be 16-byte aligned");
    const int groups = (int)(red_words / 16);
    const int64_t n_threads = rows_pad_s * groups * 2;
    const int64_t blocks = (n_threads + 255) / 256;
    BMF_LAUNCH(s24_pack_kernel, dim3((unsigned)(blocks < 262144 ? blocks : 262144)), dim3(256), 0, (hipStream_t)stream, bits, ldw, groups, rowsel,
               n_threads, s24, ovf_ptr, ovf_cursor, ovf_idx, kept, ldk);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

// Which form serves bmf_xf_bits_i8s: 0 = four waves, 256-row tiles, two workgroups per CU (the first form); 1 = eight waves, 512-row
// tiles, one workgroup per CU, DMA roles split (the second; rows_pad_s % 512 == 0).  The slab-slot count of a shape depends on it.
static int& i8s_form() {
    static int form = 0;
    return form;
}
extern "C" int bmf_xf_bits_i8s_form(int v) {
    const int prev = i8s_form();
    if (v < 0) return prev;
    BMF_REQUIRE(v <= 2, "bmf_xf_bits_i8s_form: form %d does not exist (0, 1, 2)", v);
    i8s_form() = v;
    return prev;
}

extern "C" int bmf_xf_bits_i8s_slots(int64_t rows_pad_s, int64_t red_words, int kp) {
    const int tile = i8s_form() >= 1 ? 512 : 256;
    if (rows_pad_s <= 0 || rows_pad_s % tile || red_words <= 0 || red_words % 16 || (kp != 32 && kp != 64)) {
        bmf_set_error("bmf_xf_bits_i8s_slots: bad arguments");
        return BMF_ERR_BAD_ARG;
    }
    return make_plan_i8(rows_pad_s, (int)(red_words / 4), kp, i8s_form() >= 1 ? 4 : 0).slots;
}

// out rows: rowmap[packed row] (or the packed row itself); the other rows and columns of `out` are not touched.  `splits` = the slab
// slots of `out`, >= bmf_xf_bits_i8s_slots(rows_pad_s, red_words, ncols): the slots this launch does not reach are zero-filled for
// its rows.
int bmf_xf_bits_i8s_launch(const uint32_t* s24, int64_t rows_pad_s, int64_t red_words, const int8_t* panel, int64_t ldp, const float* colscale,
                           int kp, int col0, int ncols, float* out, int64_t slab_stride, int splits, const int32_t* rowmap, const int32_t* stop,
                           hipStream_t s) {
    BMF_REQUIRE(s24 && panel && out && colscale, "bmf_xf_bits_i8s: null pointer");
    BMF_REQUIRE(rows_pad_s > 0 && rows_pad_s % 256 == 0, "bmf_xf_bits_i8s: rows_pad_s=%lld must be a positive multiple of 256", (long long)rows_pad_s);
    BMF_REQUIRE(red_words > 0 && red_words % 16 == 0, "bmf_xf_bits_i8s: red_words=%lld must be a positive multiple of 16 (reduction padded to 512)", (long long)red_words);
    BMF_REQUIRE(ldp >= 32 * red_words && ldp % 16 == 0, "bmf_xf_bits_i8s: ldp=%lld must be >= 32*red_words and a multiple of 16", (long long)ldp);
    BMF_REQUIRE(kp == 32 || kp == 64, "bmf_xf_bits_i8s: kp=%d must be 32 or 64", kp);
    BMF_REQUIRE((ncols == 32 || ncols == 64) && col0 >= 0 && col0 % 32 == 0 && col0 + ncols <= kp, "bmf_xf_bits_i8s: bad column range [%d, %d) of %d",
                col0, col0 + ncols, kp);
    BMF_REQUIRE(red_words * 32 < (1 << 24), "bmf_xf_bits_i8s: reduction length %lld would overflow the int32 accumulators", (long long)red_words * 32);
    BMF_REQUIRE(bmf_aligned16(s24) && bmf_aligned16(panel) && bmf_aligned16(out), "bmf_xf_bits_i8s: pointers must be 16-byte aligned");
    const int stages = (int)(red_words / 4);
    if (i8s_form() >= 1) {
        BMF_REQUIRE(rows_pad_s % 512 == 0, "bmf_xf_bits_i8s: the eight-wave form tiles the rows by 512 (rows_pad_s=%lld)", (long long)rows_pad_s);
        const PlanI8 pl8 = make_plan_i8(rows_pad_s, stages, ncols, 4);
        BMF_REQUIRE(splits >= pl8.slots, "bmf_xf_bits_i8s: splits=%d but this shape needs %d slab slots (bmf_xf_bits_i8s_slots)", splits, pl8.slots);
        BMF_REQUIRE(pl8.n_big == 0 && pl8.u_big == pl8.u_small, "bmf_xf_bits_i8s: the eight-wave form takes equal slices");
        if (i8s_form() == 2)
            BMF_LAUNCH((xf_bits_i8s8_kernel<3, 1>), dim3((unsigned)pl8.grid), dim3(512), 0, s, s24, stages, panel, ldp, kp, col0, ncols / 32, out,
                       slab_stride, pl8.u_small, pl8.total, pl8.n_slices, splits, colscale, rowmap, stop, pl8.perm);
        else
            BMF_LAUNCH((xf_bits_i8s8_kernel<3, 0>), dim3((unsigned)pl8.grid), dim3(512), 0, s, s24, stages, panel, ldp, kp, col0, ncols / 32, out,
                       slab_stride, pl8.u_small, pl8.total, pl8.n_slices, splits, colscale, rowmap, stop, pl8.perm);
        BMF_LAUNCH_CHECK();
        return BMF_OK;
    }
    const PlanI8 pl = make_plan_i8(rows_pad_s, stages, ncols, 0);
    BMF_REQUIRE(splits >= pl.slots, "bmf_xf_bits_i8s: splits=%d but this shape needs %d slab slots (bmf_xf_bits_i8s_slots)", splits, pl.slots);
    BMF_LAUNCH((xf_bits_i8s_kernel<3>), dim3((unsigned)pl.grid), dim3(256), 0, s, s24, stages, panel, ldp, kp, col0, ncols / 32, out, slab_stride,
               pl.n_big, pl.u_big, pl.u_small, pl.total, pl.n_slices, splits, colscale, rowmap, stop, pl.perm);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_xf_bits_i8s(const uint32_t* s24, int64_t rows_pad_s, int64_t red_words, const int8_t* panel, int64_t ldp, const float* colscale,
                               int kp, float* out, int64_t slab_stride, int splits, const int32_t* rowmap, void* stream) {
    return bmf_xf_bits_i8s_launch(s24, rows_pad_s, red_words, panel, ldp, colscale, kp, 0, kp, out, slab_stride, splits, rowmap, nullptr,
                                  (hipStream_t)stream);
}

extern "C" int bmf_xf_bits_i8s_occupancy(void) {
    int n = 0;
    hipError_t e = i8s_form() == 2   ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, xf_bits_i8s8_kernel<3, 1>, 512, 0)
                   : i8s_form() == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, xf_bits_i8s8_kernel<3, 0>, 512, 0)
                                   : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, xf_bits_i8s_kernel<3>, 256, 0);
    if (e != hipSuccess) {
        bmf_set_error("hipOccupancyMaxActiveBlocksPerMultiprocessor failed: %s", hipGetErrorString(e));
        return BMF_ERR_HIP;
    }
    return n;
}

int bmf_s24_overflow_launch(const int64_t* ovf_ptr, const int32_t* ovf_idx, const int32_t* rowmap, int64_t prows, const double* F64, int64_t ldf,
                            const float* colscale, int kp, int col0, int ncols, float* out, const int32_t* stop, hipStream_t s) {
    BMF_REQUIRE(ovf_ptr && ovf_idx && F64 && colscale && out, "bmf_s24_overflow: null pointer");
    BMF_REQUIRE(prows > 0 && (kp == 32 || kp == 64) && ldf >= kp && col0 >= 0 && ncols > 0 && col0 + ncols <= kp, "bmf_s24_overflow: bad shape");
    BMF_LAUNCH(s24_overflow_kernel, dim3((unsigned)((prows + 3) / 4)), dim3(256), 0, s, ovf_ptr, ovf_idx, rowmap, prows, F64, ldf, colscale, kp,
               col0, ncols, out, stop);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_s24_overflow(const int64_t* ovf_ptr, const int32_t* ovf_idx, const int32_t* rowmap, int64_t prows, const double* F64, int64_t ldf,
                                const float* colscale, int kp, float* out, void* stream) {
    return bmf_s24_overflow_launch(ovf_ptr, ovf_idx, rowmap, prows, F64, ldf, colscale, kp, 0, kp, out, nullptr, (hipStream_t)stream);
}
