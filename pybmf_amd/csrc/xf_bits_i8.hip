// K1/K2 on the integer matrix cores: out = bits(A) . F with F as signed 8-bit limbs and EXACT int32 accumulation.
//
//   X  @ V   (A = X bits,   limb panel of V)   replaces  multiply(W, X) @ V      PyBMF/models/BinaryMFPenalty.py:139
//   X^T @ U  (A = X^T bits, limb panel of U)   replaces  multiply(W, X).T @ U    PyBMF/models/BinaryMFPenalty.py:154
//
// Why integers.  X is 0/1 in any format, so the only inexact step of a floating-point bits GEMM is the fp32 accumulation over
// 20 000 - 100 000 terms -- and that noise, amplified by the update dynamics, is what the trajectory's distance from the fp64
// reference consists of (profiles/r02_parity_trace_c3_*.txt: 8e-5 at the worst iteration with fp16 x 2 operands, 6e-5 even with
// fp32-exact bf16 x 3 operands, against a 1e-4 gate; 8e-6 with this kernel).  With the column-scaled factor rounded to a 24-bit
// integer q = rint(F 2^e_c), |q| < 0.996 * 2^23, and written in balanced base-256 digits q = d2 2^16 + d1 2^8 + d0 (d in [-128, 127]), each
// digit plane is an int8 matrix, v_mfma_i32_16x16x64_i8 sums bit x digit products in int32 without any rounding
// (|sum| <= 128 * 100 352 < 2^24), and the three planes are recombined in fp64 at the end: the result is the exact product
// of X with the quantised factor.  The i8 MFMA also runs at twice the bf16 rate per clock, so three limbs cost 3/4 of the
// MFMA time of two fp16 addends.
//
// Design (gfx950):
//   * A operand from bits in registers: lane (r, g) fetches 16 bytes (words 4g .. 4g+3 of a 512-index block) of row r with one
//     load and uses word t in stage t of the block; (w >> s) & 0x01010101, s = 0..7, turns a word into the 8 dwords = 32
//     bytes of 0/1 that the two k-steps (64 indices each) of a stage consume.  The reduction order is free as long as both
//     operands agree, so the panel is stored in the order the bits fall out (panel_pos_i8).
//   * B operand: the L digit planes of a stage (L x kp rows of 128 bytes) go to LDS by LDS-DMA into a ring of LOOK + 1
//     buffers, LOOK stages ahead; ds_read_b128 fragments, conflict-free through an XOR swizzle applied on the DMA source.
//     A stage is only ~0.7 us of MFMA work, less than a DMA round trip under load, so the end-of-stage wait is a COUNTED
//     vmcnt that leaves the youngest stages' DMAs in flight, and the pipeline runs through row-tile boundaries.
//   * Three digit planes need three accumulator sets, so a wave owns 64 rows x 32 columns x L planes (96 accumulator VGPRs)
//     and a workgroup of 4 waves a tile of 256 rows x 32 columns; two such workgroups per CU.
//   * Persistent workgroups, stream-K over (row tile, stage) as in xf_bits.hip.  The slice -> workgroup map is
//     XCD-aware: slices are sorted by the stage they start at and dealt to the 8 XCDs in runs, so the workgroups that
//     share an L2 walk (nearly) the same panel stages at the same time and the panel is served from L2 instead of the fabric.
//     X words are loaded non-temporally (each is used once) so that they do not evict the panel.
#include "common.h"
#include "i8_plan.h"

#include <algorithm>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

#ifndef BMF_I8_XDMA
#define BMF_I8_XDMA 1   // X words through LDS by LDS-DMA (1) or straight into registers by global loads (0)
#endif

#ifdef BMF_EXP_STAMP   // diagnostic build only (scripts/build_flavour.sh): per-workgroup start / end stamps of the last launch
__device__ unsigned long long bmf_dbg_stamps[512 * 4];
extern "C" int bmf_debug_read_stamps(unsigned long long* out_host) {
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(bmf_dbg_stamps), sizeof(unsigned long long) * 512 * 4) == hipSuccess ? 0 : -2;
}
#endif

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int M, int V, int I>
__device__ __forceinline__ void interleave_one_i8() {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    constexpr int c = ((I % M + 1) * V) / M - ((I % M) * V) / M;
    if constexpr (c > 0) __builtin_amdgcn_sched_group_barrier(0x002, c, 0);
}
template <int M, int V, int... I>
__device__ __forceinline__ void interleave_mfma_valu_i8(std::integer_sequence<int, I...>) {
    (interleave_one_i8<M, V, I>(), ...);
}

// The DMA look-ahead is 3 stages and the ring has 4 buffers = the 4 stages of a group, so every ring index is a compile-time
// constant of the unrolled group body.  The B fragments of a stage's first k-step are fetched during the previous stage, so a
// stage must have landed one barrier earlier than its first MFMA: a DMA issued at the top of stage u (for stage u + 3) has two
// whole stages to land.
//
// One workgroup = 4 waves (one per SIMD) = 256 rows x 32 columns x L planes; TWO workgroups per CU.  A workgroup's barrier,
// DMA issue, bookkeeping and tile write-out then stall only one of the two waves of each SIMD -- the other belongs to a
// workgroup that is somewhere else in its own stage (measured with all 8 waves in one workgroup, in lockstep: barrier 11 %,
// DMA issue 9 % of the kernel).  With kp = 64 the two column halves of the same row tile are two workgroups that the block
// map puts on one XCD, so the X words the second one asks for are L2 hits.
//
// A stage is only 48 MFMAs per wave, so the scalar bookkeeping around it counts (measured: ~40 % of a lone wave's time in
// the first version): slices are whole groups of 4 stages, the group body is branch-free (DMAs and X-word loads are always
// issued -- past the end they re-fetch a valid stage into a free buffer -- which also keeps the vmcnt arithmetic static), and
// the only conditional work, the tile write-out, sits at the end of a group.
template <int L>
__global__ __launch_bounds__(256, 2) void xf_bits_i8_kernel(const uint32_t* __restrict__ A, int64_t ldw, int a_tiled, int stages,
                                                             const int8_t* __restrict__ P, int64_t ldp, int kp, int col_base, int halves,
                                                             float* __restrict__ out, int64_t slab_stride, int n_big, int u_big, int u_small,
                                                             int64_t total_units, int n_slices, int slots,
                                                             const float* __restrict__ colscale,
                                                             const int32_t* __restrict__ stop, SlicePerm perm) {
    if (stop && *stop != 0) return;
#ifdef BMF_EXP_STAMP
    const unsigned long long stamp_t0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int TILE_ROWS = 256;
    constexpr int LROWS = L * 32;             // 128-byte LDS rows per stage
    constexpr int STAGE_BYTES = LROWS * 128;
    constexpr int PIECES = STAGE_BYTES / 1024;
    constexpr int DMA_PER_WAVE = PIECES / 4;
    static_assert(PIECES % 4 == 0, "every wave issues the same number of DMA pieces (the vmcnt bookkeeping counts on it)");
    constexpr int RING = 4;
#if BMF_I8_XDMA
    constexpr int XG_BYTES = TILE_ROWS * 64;  // the X words of one group of four stages: 256 rows x 16 words
    constexpr int X_BYTES = 2 * XG_BYTES;     // two groups: the one in use and the next
#else
    constexpr int X_BYTES = 0;
#endif
    static_assert(2 * (RING * STAGE_BYTES + X_BYTES) <= 160 * 1024, "two workgroups' rings must fit the 160 KiB LDS");
    __shared__ __attribute__((aligned(16))) char smem[RING * STAGE_BYTES + X_BYTES];

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

    // this workgroup's run of (row tile, stage) units: whole groups of four stages (slice lengths % 4 == 0, stages % 4 == 0).  Slices
    // come in two lengths: the first n_big logical slices are u_big units long, the rest u_small (see build_plan_i8).
    const int64_t big_end = (int64_t)n_big * u_big;
    const int64_t u0 = slice < n_big ? (int64_t)slice * u_big : big_end + (int64_t)(slice - n_big) * u_small;
    const int64_t u1 = min(u0 + (slice < n_big ? u_big : u_small), total_units);
    if (u0 >= u1) return;
    const int n_groups = (int)((u1 - u0) >> 2);
    const int n_units = n_groups << 2;
    // Static wave priority (round 5, scripts/r05_i8_prio_ab.sh, profiles/r05_i8_prio_ab.txt): the first-dispatched workgroup of a CU --
    // the one with the big slice, which the SIMD's age-based arbitration favours anyway -- runs at priority 1, its partner at 0, set
    // once (no flips in the loop).  With the big share at 0.66 instead of 0.63: sustained 1 741 -> 1 759 it/s (+1.0 %; the share alone
    // +0.7 %).  The opposite -- priority for the later-dispatched workgroup, shares 0.50 - 0.63 -- lost 2 - 8 %.
#ifndef BMF_EXP_PRIO_BIG
#define BMF_EXP_PRIO_BIG 1
#endif
    if (slice < n_big) __builtin_amdgcn_s_setprio(BMF_EXP_PRIO_BIG);
#ifdef BMF_EXP_PRIO_SMALL   // (timing experiments only)
    if (slice >= n_big) __builtin_amdgcn_s_setprio(BMF_EXP_PRIO_SMALL);
#endif

    // DMA piece q = wave + 4 i (1 KiB): LDS rows 8q .. 8q+7 (row R = limb * 32 + column); lane i fills physical 16-byte chunk
    // i & 7 of row 8q + (i >> 3) with source chunk (i & 7) ^ ((R >> 1) & 7).  Panel row of LDS row R: limb * kp + col0 + column.
    // (SGPR base + per-lane 32-bit offset, hand-placed: scripts/probes/dma_issue_probe.hip prices a piece at ~29 cycles of the
    // issuing wave in this form against ~42 with a 64-bit lane address, and no vector instruction is spent on addresses;
    // -DBMF_I8_DMA_BUILTIN=1 keeps the round-3 form for A/B)
#ifndef BMF_I8_DMA_BUILTIN
#define BMF_I8_DMA_BUILTIN 1   // (measured round 4: the hand-placed form is no faster here, 245 vs 241 us in the microbenchmark: two waves per SIMD hide the issue)
#endif
    const int8_t* dsrc[DMA_PER_WAVE];
    unsigned d_off[DMA_PER_WAVE];
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) {
        const int q = wave + 4 * i;
        const int limb = q >> 2, j0 = (q & 3) * 8, d_row = lane >> 3, d_chunk = lane & 7;
        const int R = 8 * q + d_row;
        dsrc[i] = P + (int64_t)(limb * kp + col0 + j0 + d_row) * ldp + ((d_chunk ^ ((R >> 1) & 7)) << 4);
        d_off[i] = (unsigned)((int64_t)(limb * kp + col0 + j0 + d_row) * ldp) + (unsigned)((d_chunk ^ ((R >> 1) & 7)) << 4);   // < 192 * 2^24
    }
    [[maybe_unused]] auto dma16 = [&](const void* sbase, unsigned m0v, unsigned voff) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(m0v), "v"(voff), "s"(sbase) : "memory");
    };
    [[maybe_unused]] const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto issue_dma = [&](int stage, int buf) {
#if !BMF_I8_DMA_BUILTIN
        const int8_t* sb = P + (int64_t)stage * 128;
#endif
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {
#ifdef BMF_EXP_NODMA  // timing experiment only (wrong results)
            if (stage < 0)
#endif
#if BMF_I8_DMA_BUILTIN
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dsrc[i] + (int64_t)stage * 128),
                                             (__attribute__((address_space(3))) void*)(smem + buf * STAGE_BYTES + (wave + 4 * i) * 1024), 16, 0, 0);
#else
            dma16(sb, lds_base + (unsigned)(buf * STAGE_BYTES + (wave + 4 * i) * 1024), d_off[i]);
#endif
        }
    };

    auto lds0_of = [](char* p_) { return (unsigned)(size_t)(__attribute__((address_space(3))) char*)p_; };
    const unsigned lds0 = lds0_of(smem);
    // B fragment of (16-column tile nt, limb l), k-step ks: row l*32 + 16 nt + r, physical chunk (4 ks + g) ^ (r >> 1)
    const unsigned b_lane = lds0 + (unsigned)(r * 128);
    const int b_sw = r >> 1;
    auto wait_b = [&](i32x4 (&dst)[2][L]) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int l = 0; l < L; ++l) asm volatile("" : "+v"(dst[nt][l]));
    };

    // X words.  Lane (r, g) uses, for each of its four 16-row groups, the 16 bytes [4g, 4g+4) words of a group of four stages:
    // word t in stage t of the group (the panel is stored in the matching order, panel_pos_i8).  One wave-uniform pointer walks
    // the groups: + 64 bytes per group, + the rest of a 256-row tile at a tile end; it stops advancing on the last group of the
    // matrix (re-reading it is harmless).
    // Two layouts of the bit matrix: plain rows (ldw words each), or TILED (bmf_tile_bits): the 256 rows x 16 words a workgroup
    // needs for one group of stages stored as one contiguous 16-KiB block, blocks in (row tile, group) order -- then every DMA
    // piece is 1 KiB of consecutive bytes (whole 128-byte lines, one DRAM page) instead of sixteen 64-byte pieces of sixteen rows.
    int tile = (int)(u0 / stages);
    int st_cur = (int)(u0 - (int64_t)tile * stages);   // first stage of the group being computed (multiple of 4)
    const int64_t n_tiles_a = total_units / stages;
    const uint32_t* a_ptr = a_tiled ? A + (((int64_t)tile * (stages >> 2) + (st_cur >> 2)) * TILE_ROWS + wave * 64) * 16
                                    : A + ((int64_t)tile * TILE_ROWS + wave * 64) * ldw + 4 * (int64_t)st_cur;
    const uint32_t* const a_last = a_tiled ? A + ((n_tiles_a * (stages >> 2) - 1) * TILE_ROWS + wave * 64) * 16
                                           : A + ((n_tiles_a - 1) * TILE_ROWS + wave * 64) * ldw + 4 * (int64_t)(stages - 4);
    int a_st = st_cur;
    const int64_t a_tile_step = a_tiled ? TILE_ROWS * 16 : TILE_ROWS * ldw - 4 * (int64_t)(stages - 4);   // last group of a tile -> first of the next
    const int64_t a_group_step = a_tiled ? TILE_ROWS * 16 : 16;
    auto advance_a = [&]() {
        const bool tile_last = a_st + 4 == stages;
        const uint32_t* nx = a_ptr + (tile_last ? a_tile_step : a_group_step);
        a_st = tile_last ? 0 : a_st + 4;
        a_ptr = a_ptr == a_last ? a_ptr : nx;
    };
    u32x4 aq[4];
#if BMF_I8_XDMA
    // Through LDS, by LDS-DMA, like the panel: a group's words (16 KiB per workgroup) are requested during the previous group --
    // two 1-KiB pieces per wave in its stages 0 and 1 -- and have landed by its last wait; each wave fetches and reads only its
    // own 64 rows, so no barrier is involved.  As plain loads into registers (BMF_I8_XDMA = 0) they were an HBM round trip that
    // the in-order vmcnt waits of the panel pipeline force to complete within two stages, and 4 loads + 16 register moves of
    // issue per group: together 15 % of the kernel.
    // piece p (0..3) of this wave: rows 16 p .. 16 p + 15 of its 64, lane i -> row i >> 2, 16-byte chunk i & 3
    unsigned x_src[4];
#pragma unroll
    for (int p_ = 0; p_ < 4; ++p_)
        x_src[p_] = a_tiled ? (unsigned)(p_ * 1024 + lane * 16) : (unsigned)((16 * p_ + (lane >> 2)) * ldw + 4 * (lane & 3)) * 4u;
    char* const x_lds = smem + RING * STAGE_BYTES;
    auto issue_x = [&](int gbuf, int p_) {   // piece p_ of the group at a_ptr into X buffer gbuf
        const char* base = reinterpret_cast<const char*>(a_ptr);   // wave-uniform
        char* dst = x_lds + gbuf * XG_BYTES + (wave * 4 + p_) * 1024;
#ifndef BMF_EXP_NOALOAD
#if BMF_I8_DMA_BUILTIN
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + x_src[p_]),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
#else
        dma16(base, (unsigned)(size_t)(__attribute__((address_space(3))) char*)dst, x_src[p_]);
#endif
#endif
    };
    const unsigned x_rd = lds0_of(smem) + (unsigned)(RING * STAGE_BYTES + (64 * wave + r) * 64 + g * 16);
    auto read_x = [&](int gbuf) {   // this lane's words of the group in X buffer gbuf -> aq (waits for them)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(aq[mt]) : "v"(x_rd + (unsigned)(gbuf * XG_BYTES)), "n"(16 * 64 * mt));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) asm volatile("" : "+v"(aq[mt]));
    };
#else
    // Straight into registers: one 16-byte load per 16-row group and group of stages.  The loads are hand-written asm so that the
    // compiler's waitcnt insertion does not see them: the counted waits at the end of every stage (below) cover them.
    u32x4 an[4];
    unsigned a_off[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) a_off[mt] = (unsigned)((16 * mt + r) * (a_tiled ? 16 : ldw) + 4 * g) * 4u;
    auto load_a = [&](u32x4 (&dst)[4]) {   // loads the group at a_ptr, then advances
        const uint64_t b = reinterpret_cast<uint64_t>(a_ptr);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
        const uint64_t sb = ((uint64_t)hi << 32) | lo;
        // (s_nop 4: the base may have just been written by v_readfirstlane, and an SGPR written by a VALU instruction needs 5 wait
        // states before a vector-memory instruction reads it as its scalar base -- the compiler cannot see that this asm is one)
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(dst[0]) : "v"(a_off[0]), "s"(sb) : "memory");
#pragma unroll
        for (int mt = 1; mt < 4; ++mt) asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst[mt]) : "v"(a_off[mt]), "s"(sb) : "memory");
        advance_a();
    };
#endif

    // output scales, fetched before the pipeline starts (a load inside the loop would make the compiler drain the DMA queue)
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
        const int64_t tu = (int64_t)tl * stages;   // the logical slice that holds the tile's first unit
        const int first_wg = tu < big_end ? (int)(tu / u_big) : n_big + (int)((tu - big_end) / u_small);
        const int slot = slice - first_wg;
        const int64_t row_base = (int64_t)tl * TILE_ROWS + wave * 64;
        float* o = out + (int64_t)slot * slab_stride;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    long long v = 0;
#pragma unroll
                    for (int l = L - 1; l >= 0; --l) v = v * 256 + acc[mt][nt][l][i];
                    const int64_t row = row_base + 16 * mt + 4 * g + i;
#ifdef BMF_EXP_NOSTORE  // timing experiment only
                    if (v == 0x7fffffffffffll)
#endif
                    o[row * kp + col0 + 16 * nt + r] = (float)((double)v * (double)osc[nt]);
                }
        if (last_of_tile) {  // last contributor of this tile: the slab slots nobody writes must read as zero
            for (int z = slot + 1; z < slots; ++z) {
                float* oz = out + (int64_t)z * slab_stride;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int64_t row = row_base + 16 * mt + 4 * g + i;
                            oz[row * kp + col0 + 16 * nt + r] = 0.f;
                        }
            }
        }
    };

    // ---- prologue: stages 0..2 of the run and the X words of the first group ----
    int st_dma = st_cur;   // stage of the next DMA to issue; n_dma counts them (past the end the last stage is re-fetched)
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
#if BMF_I8_XDMA
#pragma unroll
    for (int p_ = 0; p_ < 4; ++p_) issue_x(0, p_);
    advance_a();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    read_x(0);
#else
    load_a(aq);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) asm volatile("" : "+v"(aq[mt]));
    __syncthreads();
#endif
    zero_acc();
    i32x4 b0[2][L], b1[2][L];
#ifdef BMF_EXP_NOLDS
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int l = 0; l < L; ++l) b0[nt][l] = b1[nt][l] = i32x4{0x01020304, 0x05060708, 0x01020304, 0x05060708};
#endif
    // fragment reads with the ring slot and (tile, limb) offset folded into the immediate
#ifndef BMF_EXP_NOLDS
#define BMF_FETCH_B(slot, ks, dst)                                                                                       \
    do {                                                                                                                 \
        const unsigned addr_ = b_lane + (unsigned)(((((ks) * 4 + g) ^ b_sw) & 7) << 4);                                   \
        _Pragma("unroll") for (int nt_ = 0; nt_ < 2; ++nt_) _Pragma("unroll") for (int l_ = 0; l_ < L; ++l_)              \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[nt_][l_]) : "v"(addr_),                              \
                         "n"((slot) * STAGE_BYTES + (l_ * 32 + 16 * nt_) * 128));                                        \
    } while (0)
#else
#define BMF_FETCH_B(slot, ks, dst)                                                                                       \
    do {                                                                                                                 \
        _Pragma("unroll") for (int nt_ = 0; nt_ < 2; ++nt_) _Pragma("unroll") for (int l_ = 0; l_ < L; ++l_)              \
            asm volatile("" : "+v"(dst[nt_][l_]));                                                                       \
    } while (0)
#endif
    BMF_FETCH_B(0, 0, b0);
    wait_b(b0);

    for (int gq = 0; gq < n_groups; ++gq) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {   // (fully unrolled: t, and with it every ring slot, is a constant in each copy)
            next_dma((t + 3) & 3);     // stage t + 3 goes into the buffer stage t - 1 was read from
#if BMF_I8_XDMA
            // the NEXT group's X words, into the other X buffer (free since this group's words went to registers): two pieces in
            // each of the first two stages, AFTER the stage's panel pieces (see the waits below)
            if (t == 0) { issue_x((gq + 1) & 1, 0); issue_x((gq + 1) & 1, 1); }
            if (t == 1) { issue_x((gq + 1) & 1, 2); issue_x((gq + 1) & 1, 3); advance_a(); }
#elif !defined(BMF_EXP_NOALOAD)
            if (t == 0) load_a(an);    // the NEXT group's X words; issued AFTER this stage's DMA (see the wait below)
#endif

            auto k_step = [&](int ks, i32x4 (&bc)[2][L], i32x4 (&bx)[2][L]) {
                // fetch the next k-step's fragments (from the next stage's buffer at the end: it has been complete and visible
                // since the previous barrier), run this k-step's MFMAs under that latency, then collect
                if (ks == 0) BMF_FETCH_B(t, 1, bx);
                else BMF_FETCH_B((t + 1) & 3, 0, bx);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const unsigned w = aq[mt][t];
                    i32x4 av;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#ifdef BMF_EXP_NOVALU  // timing experiment only (wrong results)
                        av[e] = (int)w;
#else
                        av[e] = (int)((w >> (4 * ks + e)) & 0x01010101u);
#endif
#pragma unroll
                    for (int l = 0; l < L; ++l)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc[mt][nt][l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bc[nt][l], acc[mt][nt][l], 0, 0, 0);
                }
                // the 8 shift/and ops that expand the next row group's bits are spread between the MFMAs of the current one
#ifndef BMF_EXP_NOSCHED
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                interleave_mfma_valu_i8<2 * L, 8>(std::make_integer_sequence<int, 3 * 2 * L>{});
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * L, 0);
#endif
                __builtin_amdgcn_sched_barrier(0);
                wait_b(bx);
            };
            k_step(0, b0, b1);
            k_step(1, b1, b0);

            // End of stage u: stage u + 2 is fetched from (B fragments) during stage u + 1, so this wave's pieces of it -- issued at
            // the top of stage u - 1 -- must have landed before the barrier.  What was issued since may stay in flight: this
            // stage's DMA pieces and the four X-word loads of t == 0, which are issued right AFTER that stage's DMA so that they
            // are younger than it: vmcnt counts in issue order, and this way the loads (an HBM round trip each) are only forced
            // to complete by the wait of t == 2, three stages after their issue, instead of one stage earlier.
#if BMF_I8_XDMA
            // (X pieces, two after the panel pieces of t == 0 and of t == 1: younger than the DMA of t - 1 are, at the end of
            // t = 0: D0 X X; t = 1: X X D1 X X; t = 2: X X D2; t = 3: D3 -- so the X pieces are complete by the end of t = 3)
            if (t == 0 || t == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE + 2) : "memory");
            else if (t == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE + 4) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE) : "memory");
#else
            if (t <= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE + 4) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE) : "memory");
#endif
#ifndef BMF_EXP_NOBAR
            __builtin_amdgcn_s_barrier();
#endif
            asm volatile("" ::: "memory");
        }
#if BMF_I8_XDMA
        read_x((gq + 1) & 1);   // the next group's words: this wave's own pieces, complete since the wait of t == 3
#else
        // the X words of the next group were requested at t == 0 and the waits of t = 1..3 covered them: tie the registers here
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            asm volatile("" : "+v"(an[mt]));
            aq[mt] = an[mt];
        }
#endif
        const bool tile_end = st_cur + 4 == stages;
        if (tile_end || gq + 1 == n_groups) {
            write_tile(tile, tile_end);
            zero_acc();
        }
        tile += tile_end ? 1 : 0;
        st_cur = tile_end ? 0 : st_cur + 4;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus DMAs of the last stages
#undef BMF_FETCH_B
#ifdef BMF_EXP_STAMP
    if (threadIdx.x == 0 && blockIdx.x < 512) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        bmf_dbg_stamps[4 * blockIdx.x + 0] = stamp_r0;
        bmf_dbg_stamps[4 * blockIdx.x + 1] = r1;
        bmf_dbg_stamps[4 * blockIdx.x + 2] = t1 - stamp_t0;
        bmf_dbg_stamps[4 * blockIdx.x + 3] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
}

// Share of a CU's work that goes to the FIRST of its two workgroups.  Equal slices would be the obvious cut, but the two
// workgroups of a CU do not run at equal speed: the SIMD arbitrates between its two waves by age, so the workgroup that was
// dispatched first runs nearly unimpeded and the second one gets the leftover issue slots until the first has finished --
// measured with per-workgroup stamps (-DBMF_EXP_STAMP, profiles/r03_i8_workgroup_stamps.txt) at the headline shape: the 256
// workgroups with blockIdx < 256 take 161-182 us, their 250 partners (always blockIdx + 256: the dispatcher deals the first 256
// workgroups one per CU) 238-258 us, and for the last third of the kernel every CU runs ONE workgroup at the rate a lone wave per
// SIMD sustains.  Cutting the work so that both finish together removes that tail.  A speed assumption only: any cut is correct.
#ifdef BMF_EXP_SHARE   // (timing experiments: -DBMF_EXP_SHARE=63)
static double old_share() { return BMF_EXP_SHARE / 100.0; }
#else
static double old_share() { return 0.66; }   // (0.63 until round 5; re-swept with the static priority: profiles/r05_i8_prio_ab.txt)
#endif   // (swept 0.50 - 0.68 in one box: profiles/r03_i8_share_sweep.txt)

#ifndef BMF_I8_WG_PER_CU
#define BMF_I8_WG_PER_CU 2
#endif
// ncols = width of the column range one launch covers (32 or 64, a multiple of 32 inside the kp-wide factor); wide: the plan of
// the 64-column kernel (xf_bits_i8w.hip) -- ONE workgroup per CU that owns whole rows, so no column halves and equal slices
PlanI8 build_plan_i8(int64_t rows_pad, int stages, int ncols, int cus, int wide) {
    PlanI8 p;
    // (wide == 4: the eight-wave sparse kernel of xf_bits_i8s.hip -- 512 rows x 32 columns per workgroup, one workgroup per CU, the
    // two column halves of a row tile on two CUs of one XCD, equal slices)
    const int halves = (wide && wide != 4) ? 1 : ncols / 32;
    const int wg_per_cu = wide ? 1 : BMF_I8_WG_PER_CU;
    const int n_row_tiles = (int)(rows_pad / ((wide == 3 || wide == 4) ? 512 : 256));
    p.total = (int64_t)n_row_tiles * stages;
    // two workgroups per CU; with kp = 64 they are the two column halves of one slice
    int64_t gsz = wg_per_cu * (int64_t)cus / halves;   // slices (per column half)
    if (gsz > 512) gsz = 512;
    const int64_t groups = p.total / 4;                        // whole groups of four stages (stages % 4 == 0, so is the total)
    const double share = old_share();
    // block b -> bslice = (b >> 3) / halves * 8 + (b & 7): the first gsz / 2 bslices belong to the first-dispatched workgroup of
    // their CU ("old"), the rest to the second ("young")
    const int n_old = (int)(gsz / 2);
    if (wg_per_cu == 2 && gsz % 16 == 0 && groups >= 4 * gsz && share > 0.5) {
        const int64_t g_big = (int64_t)(share * 2.0 * (double)groups / (double)gsz + 0.999);        // groups per big slice
        p.n_big = (int)std::min<int64_t>(n_old, (groups + g_big - 1) / g_big);
        const int64_t rest = groups - std::min<int64_t>(groups, (int64_t)p.n_big * g_big);
        const int64_t g_small = std::max<int64_t>(1, (rest + (gsz - n_old) - 1) / (gsz - n_old));
        p.u_big = (int)(4 * g_big);
        p.u_small = (int)(4 * g_small);
        p.n_slices = p.n_big + (int)((rest + g_small - 1) / g_small);
    } else {   // small problems (and the one-workgroup-per-CU flavour): equal slices
        if (gsz > groups) gsz = groups;
        const int64_t g = (groups + gsz - 1) / gsz;
        p.n_big = 0;
        p.u_big = p.u_small = (int)(4 * g);
        p.n_slices = (int)((groups + g - 1) / g);
    }
    const int64_t big_end = (int64_t)p.n_big * p.u_big;
    auto slice_of = [&](int64_t u) { return u < big_end ? (int)(u / p.u_big) : p.n_big + (int)((u - big_end) / p.u_small); };
    auto start_of = [&](int l) { return l < p.n_big ? (int64_t)l * p.u_big : big_end + (int64_t)(l - p.n_big) * p.u_small; };
    int slots = 1;
    for (int t = 0; t < n_row_tiles; ++t) slots = std::max(slots, slice_of((int64_t)(t + 1) * stages - 1) - slice_of((int64_t)t * stages) + 1);
    p.slots = slots;
    // XCD-aware slice -> workgroup map: workgroups b, b + 8, b + 16, ... share an XCD (round-robin dispatch; a speed
    // assumption only -- any map is correct).  Slices sorted by their starting stage are dealt to the XCDs in runs, so the
    // workgroups that share an L2 walk (nearly) the same panel stages at the same time.  Big slices go to the old bslices, small
    // ones to the young (with equal slices: one class).
    for (int b = 0; b < 512; ++b) p.perm.p[b] = 0xFFFF;
    // `lead`: the class's starting stages are taken `lead` stages EARLIER for the sort, i.e. the XCD that gets the big slices
    // starting around stage w gets the small ones starting around w + lead.  A small slice is walked more slowly than a big one
    // (that is why it is small), so with lead = u_big - u_small the two fronts of an XCD END at the same panel stage and are never
    // further apart than `lead` stages: the faster front runs into lines the slower one fetched a moment ago, instead of both
    // fetching the panel range once each (measured before this: fabric traffic of a launch 0.42 -> 0.69 GB when the slices
    // became uneven).
    auto deal = [&](int l0, int l1, int b0, int b1, int64_t lead) {   // logical slices [l0, l1) onto bslices [b0, b1), b0 % 8 == 0
        std::vector<int> order;
        for (int l = l0; l < l1; ++l) order.push_back(l);
        auto key = [&](int l) { return ((start_of(l) - lead) % stages + stages) % stages; };
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key(a) < key(b); });
        size_t j = 0;
        for (int x = 0; x < 8; ++x)
            for (int b = b0 + x; b < b1 && j < order.size(); b += 8) p.perm.p[b] = (uint16_t)order[j++];
        // (leftovers when the class has more slices than bslices cannot happen: the counts above are bounded by the class sizes)
    };
#ifdef BMF_EXP_LEAD   // (timing experiments)
    constexpr int lead_on = BMF_EXP_LEAD;
#else
    constexpr int lead_on = 50;   // percent of u_big - u_small (0 / 50 / 100 A/B'd in round 3: HISTORY.md; 25 / 75 / 100 again at share 0.66)
#endif
    if (p.n_big > 0) {
        deal(0, p.n_big, 0, n_old, 0);
        deal(p.n_big, p.n_slices, n_old, (int)gsz, (int64_t)(p.u_big - p.u_small) * lead_on / 100);
    } else {
        deal(0, p.n_slices, 0, (p.n_slices + 7) / 8 * 8, 0);
    }
    int last = 0;
    for (int b = 0; b < 512; ++b)
        if (p.perm.p[b] != 0xFFFF) last = b;
    p.grid = (last / 8 + 1) * 8 * halves;
    return p;
}

// The plan of a shape is built once (sorting 512 slices costs more host time than enqueueing the kernel: an iteration of a
// row-sharded run at 1/8 of the headline rows is host-paced) and kept; a handful of shapes per process.  Returned BY VALUE, copied
// under the lock (another thread may evict the entry); the key includes the CU count the plan was cut for, so a process that
// drives GPUs of different sizes does not reuse one device's plan (and slab-slot count) on another.
}  // namespace
PlanI8 make_plan_i8(int64_t rows_pad, int stages, int ncols, int wide) {
    struct Entry { int64_t rows_pad; int stages, ncols, cus, wide; PlanI8 plan; };
    static std::mutex mu;
    static std::deque<Entry> cache;
    const int cus = bmf_cu_count_current();
    std::lock_guard<std::mutex> lock(mu);
    for (const Entry& e : cache)
        if (e.rows_pad == rows_pad && e.stages == stages && e.ncols == ncols && e.cus == cus && e.wide == wide) return e.plan;
    if (cache.size() >= 64) cache.pop_front();
    cache.push_back(Entry{rows_pad, stages, ncols, cus, wide, build_plan_i8(rows_pad, stages, ncols, cus, wide)});
    return cache.back().plan;
}

// Which kernel takes a launch that covers a whole 64-column factor: 0 = the 32-column kernel of this file (two column halves, the
// default: profiles/r04_i8_wide_tile.md has the A/B), 1 / 2 = a variant of the 64-column kernel (xf_bits_i8w.hip).  Set by
// bmf_xf_bits_i8_variant(); the slab-slot count of a shape depends on it, so switch it before the buffers of an engine are sized,
// not in the middle of a run.
static int& i8_variant() {
    static int variant = 0;
    return variant;
}
int bmf_i8_use_wide(int ncols, int kp) { return (ncols == 64 && kp == 64) ? i8_variant() : 0; }
extern "C" int bmf_xf_bits_i8_variant(int v) {
    const int prev = i8_variant();
    if (v < 0) return prev;
    BMF_REQUIRE(v <= 2 || v == 4, "bmf_xf_bits_i8_variant: variant %d does not exist (0, 1, 2, 4)", v);
    i8_variant() = v;
    return prev;
}
namespace {

// One block = 128 factor rows = lane group g = blk & 3 of 512-block blk >> 2 (see bmf_panel_pos_i8_dev): its bytes land in
// eight 16-byte segments per (limb, column) row -- stage t, k-step ks -> offset 128 t + (4 ks + g) 16.
template <int KP>
__global__ __launch_bounds__(256) void make_panel_i8_kernel(const double* __restrict__ F64, int64_t ldf,
                                                             const float* __restrict__ scale, int8_t* __restrict__ panel,
                                                             int64_t ldp, int limbs, const int32_t* __restrict__ stop,
                                                             const float* __restrict__ flags, float* __restrict__ colscale_out,
                                                             int panel_blocks, const float* __restrict__ rslabs, int rcount, int rn,
                                                             float* __restrict__ rout32, double* __restrict__ rout64) {
    if (stop && *stop != 0) {
        // (after the stop: the fp64 Gram block of a row-sharded run is still all-reduced every iteration until the host notices --
        // leave zeros there, not a sum that grows by the world size each time)
        if ((int)blockIdx.x >= panel_blocks && rout64) {
            const int i = ((int)blockIdx.x - panel_blocks) * 16 + (threadIdx.x & 15);
            if ((threadIdx.x >> 4) == 0 && i < rn) rout64[i] = 0.0;
        }
        return;
    }
    // Blocks past `panel_blocks` (iteration driver): sum the k x k Gram slabs of the launch before -- out[i] = sum_b rslabs[b * rn + i],
    // fp64, slab order, the arithmetic of reduce_slabs_kernel (util.hip) -- so that the conditional rebuild below and that
    // reduction are one launch instead of two 5-us ones.  16 outputs x 16 slab groups per block.
    if ((int)blockIdx.x >= panel_blocks) {
        __shared__ double rsh[16][16];
        const int o = threadIdx.x & 15, gq = threadIdx.x >> 4;
        const int i = ((int)blockIdx.x - panel_blocks) * 16 + o;
        double acc = 0.0;
        if (i < rn) {
            const int per = (rcount + 15) / 16;
            const int b0 = gq * per, b1 = min(b0 + per, rcount);
            const float* p = rslabs + (int64_t)b0 * rn + i;
            int b = b0;
            for (; b + 4 <= b1; b += 4) {
                const float v0 = p[0], v1 = p[rn], v2 = p[2 * rn], v3 = p[3 * rn];
                p += 4 * rn;
                acc = (((acc + (double)v0) + (double)v1) + (double)v2) + (double)v3;
            }
            for (; b < b1; ++b) { acc += (double)*p; p += rn; }
        }
        rsh[gq][o] = acc;
        __syncthreads();
        if (gq == 0 && i < rn) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += rsh[q][o];
            if (rout32) rout32[i] = (float)t;
            if (rout64) rout64[i] = t;
        }
        return;
    }
    // Conditional form (iteration driver): the epilogue has already built the planes with predicted column scales; `flags` ([KP],
    // from the column-scale step) says for which columns the prediction was off.  None: nothing to do.  A few: rebuild those columns
    // only, byte by byte (a column is 1/KP of the factor: strided 8-byte reads, single-byte writes -- ~2 us per column at the headline
    // shape).  Many (the first iteration, a fresh start): rebuild everything with the exact scales, and hand those to the GEMM.
    if (flags) {
        int nflag = 0;
        for (int i = 0; i < KP; ++i) nflag += flags[i] != 0.f ? 1 : 0;   // (uniform: every thread reads the same words)
        if (nflag == 0) return;
        if (nflag <= 8) {
            const int rl = threadIdx.x & 127;
            const int64_t row = (int64_t)blockIdx.x * 128 + rl;
            const int64_t pos = (((int64_t)blockIdx.x >> 2) << 9) + bmf_panel_pos_i8_dev(128 * (int)(blockIdx.x & 3) + rl);
            int seen = 0;
            for (int jc = 0; jc < KP; ++jc) {
                if (flags[jc] == 0.f) continue;
                if ((seen++ & 1) != (int)(threadIdx.x >> 7)) continue;   // the two 128-thread halves take alternate flagged columns
                int q = __double2int_rn(fmax(fmin(F64[row * ldf + jc] * (double)scale[jc], 8355711.0), -8355711.0));
                if (limbs == 2) {
                    q = (q + 128) >> 8;
                    const int d1 = ((q + 128) & 255) - 128;
                    const int d2 = (q - d1) >> 8;
                    panel[(int64_t)(0 * KP + jc) * ldp + pos] = (int8_t)d1;
                    panel[(int64_t)(1 * KP + jc) * ldp + pos] = (int8_t)d2;
                } else {
                    const int d0 = ((q + 128) & 255) - 128;
                    const int q1 = (q - d0) >> 8;
                    const int d1 = ((q1 + 128) & 255) - 128;
                    const int d2 = (q1 - d1) >> 8;
                    panel[(int64_t)(0 * KP + jc) * ldp + pos] = (int8_t)d0;
                    panel[(int64_t)(1 * KP + jc) * ldp + pos] = (int8_t)d1;
                    panel[(int64_t)(2 * KP + jc) * ldp + pos] = (int8_t)d2;
                }
            }
            return;
        }
        if (blockIdx.x == 0 && threadIdx.x < KP) colscale_out[threadIdx.x] = (limbs == 2 ? 256.0f : 1.0f) / scale[threadIdx.x];
    }
    // The tile is filled a dword at a time: dword d of a (limb, column) row holds the digits of the four rows 32 (d >> 3) + (d & 7)
    // + 8 b, b = 0..3 (see bmf_panel_pos_i8_dev), so a thread takes one column and those four rows -- four coalesced loads across
    // the column lanes -- and stores one packed word per limb.  (Byte stores, one per digit, were 16 x the LDS instructions with
    // 4-way bank conflicts: 31 us for U at the headline shape against 15 us of HBM time.)  Rows of 33 dwords: odd, so the
    // column-per-lane stores cover all banks.
    constexpr int LROWW = 33;
    __shared__ unsigned tile[3 * KP * LROWW];
    const int64_t row0 = (int64_t)blockIdx.x * 128;
    const int g = blockIdx.x & 3;
    constexpr int SUBS = 256 / KP;   // threads per column
    const int j = threadIdx.x % KP, sub = threadIdx.x / KP;
    const double sc = (double)scale[j];
    constexpr int ITEMS = 32 / SUBS;   // dwords per thread: all their loads are issued before the first digit is taken (one block has
    double f[ITEMS][4];                // only 12 waves' worth of work: with a load round trip per dword the kernel was latency-bound)
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int d = sub + it * SUBS;
        const int rbase = 32 * (d >> 3) + (d & 7);
#pragma unroll
        for (int b = 0; b < 4; ++b) f[it][b] = F64[(row0 + rbase + 8 * b) * ldf + j];
    }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int d = sub + it * SUBS;
        unsigned w0 = 0u, w1 = 0u, w2 = 0u;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            int q = __double2int_rn(fmax(fmin(f[it][b] * sc, 8355711.0), -8355711.0));
            if (limbs == 2) {  // 15 significant bits: drop the lowest digit (round to a multiple of 256)
                q = (q + 128) >> 8;
                const int d1 = ((q + 128) & 255) - 128;
                const int d2 = (q - d1) >> 8;
                w0 |= (unsigned)(d1 & 255) << (8 * b);
                w1 |= (unsigned)(d2 & 255) << (8 * b);
            } else {
                const int d0 = ((q + 128) & 255) - 128;
                const int q1 = (q - d0) >> 8;
                const int d1 = ((q1 + 128) & 255) - 128;
                const int d2 = (q1 - d1) >> 8;
                w0 |= (unsigned)(d0 & 255) << (8 * b);
                w1 |= (unsigned)(d1 & 255) << (8 * b);
                w2 |= (unsigned)(d2 & 255) << (8 * b);
            }
        }
        tile[(0 * KP + j) * LROWW + d] = w0;
        tile[(1 * KP + j) * LROWW + d] = w1;
        if (limbs == 3) tile[(2 * KP + j) * LROWW + d] = w2;
    }
    __syncthreads();
    const int pieces = limbs * KP * 8;  // 16-byte segments
    const int64_t blk512 = (row0 >> 9) << 9;
    for (int p = threadIdx.x; p < pieces; p += 256) {
        const int rowi = p >> 3, seg = p & 7;  // rowi = limb*KP + j; seg = 2 t + ks
        const unsigned* tp = tile + rowi * LROWW + 4 * seg;
        const uint4 v = {tp[0], tp[1], tp[2], tp[3]};
        *reinterpret_cast<uint4*>(panel + (int64_t)rowi * ldp + blk512 + 128 * (seg >> 1) + ((seg & 1) * 4 + g) * 16) = v;
    }
}

// scale[c] = 2^e_c with max|F[:, c]| 2^e_c in [2^22, 0.996 * 2^23] (else [2^21, 2^22)); scale[kp + c] = 2^-e_c (the GEMM's colscale)
// (body: bmf_colscale_i8_block in common.h -- the iteration driver runs it as extra blocks of the Gram launch)
__global__ __launch_bounds__(256) void colscale_i8_kernel(const float* __restrict__ blockmax, int nblk, int kp, int limbs,
                                                           float* __restrict__ scale, const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    __shared__ float sh[256];
    bmf_colscale_i8_block(blockmax, nblk, kp, limbs, scale, (int)blockIdx.x, sh);
}

template <int L>
int launch_i8(const uint32_t* A, int64_t ldw, int a_tiled, int stages, const int8_t* P, int64_t ldp, int kp, int col0, int ncols, float* out,
              int64_t slab_stride, const PlanI8& pl, int slots, const float* colscale, const int32_t* stop, hipStream_t s) {
    BMF_LAUNCH((xf_bits_i8_kernel<L>), dim3((unsigned)pl.grid), dim3(256), 0, s, A, ldw, a_tiled, stages, P, ldp, kp, col0, ncols / 32, out, slab_stride,
               pl.n_big, pl.u_big, pl.u_small, pl.total, pl.n_slices, slots, colscale, stop, pl.perm);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

}  // namespace

extern "C" int bmf_panel_pos_i8(int cl) { return (cl < 0 || cl > 511) ? -1 : bmf_panel_pos_i8_dev(cl); }

// resident workgroups per CU the runtime grants the GEMM kernel (2 by design: registers and LDS are budgeted for it)
extern "C" int bmf_xf_bits_i8_occupancy(int limbs) {
    int n = 0;
    hipError_t e = limbs == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, xf_bits_i8_kernel<2>, 256, 0)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, xf_bits_i8_kernel<3>, 256, 0);
    if (e != hipSuccess) {
        bmf_set_error("hipOccupancyMaxActiveBlocksPerMultiprocessor failed: %s", hipGetErrorString(e));
        return BMF_ERR_HIP;
    }
    return n;
}

extern "C" int bmf_xf_bits_i8_slots(int64_t rows_pad, int64_t red_words, int kp) {
    if (rows_pad <= 0 || rows_pad % BMF_ROW_PAD || red_words <= 0 || red_words % 16 || (kp != 32 && kp != 64)) {
        bmf_set_error("bmf_xf_bits_i8_slots: bad arguments");
        return BMF_ERR_BAD_ARG;
    }
    return make_plan_i8(rows_pad, (int)(red_words / 4), kp, bmf_i8_use_wide(kp, kp)).slots;
}

int bmf_blockmax_launch(const float* F, int64_t rows_pad, int64_t ldf, int kp, float* ws, const int32_t* stop, hipStream_t s);

// col0, ncols: the column range of the kp-wide factor this launch computes (the whole factor: 0, kp); the other columns of `out`
// are not touched.  `splits` must cover bmf_xf_bits_i8_slots(rows_pad, red_words, ncols).
int bmf_xf_bits_i8_launch(const uint32_t* Abits, int64_t rows_pad, int64_t ldw, int64_t red_words, const int8_t* panel, int64_t ldp,
                          int limbs, const float* colscale, int kp, int col0, int ncols, float* out, int64_t slab_stride, int splits,
                          int a_tiled, const int32_t* stop, hipStream_t s) {
    BMF_REQUIRE(Abits && panel && out && colscale, "bmf_xf_bits_i8: null pointer");
    BMF_REQUIRE(rows_pad > 0 && rows_pad % BMF_ROW_PAD == 0, "bmf_xf_bits_i8: rows_pad=%lld must be a positive multiple of %d",
                (long long)rows_pad, BMF_ROW_PAD);
    BMF_REQUIRE(red_words > 0 && red_words % 16 == 0, "bmf_xf_bits_i8: red_words=%lld must be a positive multiple of 16 (reduction padded to 512)", (long long)red_words);
    BMF_REQUIRE(ldw >= red_words && ldw % 4 == 0, "bmf_xf_bits_i8: ldw=%lld must be >= red_words and a multiple of 4", (long long)ldw);
    BMF_REQUIRE(ldp >= 32 * red_words && ldp % 16 == 0, "bmf_xf_bits_i8: ldp=%lld must be >= 32*red_words and a multiple of 16", (long long)ldp);
    BMF_REQUIRE(kp == 32 || kp == 64, "bmf_xf_bits_i8: kp=%d must be 32 or 64", kp);
    BMF_REQUIRE((ncols == 32 || ncols == 64) && col0 >= 0 && col0 % 32 == 0 && col0 + ncols <= kp, "bmf_xf_bits_i8: bad column range [%d, %d) of %d",
                col0, col0 + ncols, kp);
    BMF_REQUIRE(limbs == 2 || limbs == 3, "bmf_xf_bits_i8: limbs=%d must be 2 or 3", limbs);
    BMF_REQUIRE(red_words * 32 < (1 << 24), "bmf_xf_bits_i8: reduction length %lld would overflow the int32 accumulators",
                (long long)red_words * 32);
    BMF_REQUIRE(slab_stride >= rows_pad * kp, "bmf_xf_bits_i8: slab_stride too small");
    BMF_REQUIRE(bmf_aligned16(Abits) && bmf_aligned16(panel) && bmf_aligned16(out), "bmf_xf_bits_i8: pointers must be 16-byte aligned");
    const int stages = (int)(red_words / 4);
    const int wide = bmf_i8_use_wide(ncols, kp);
    const PlanI8 pl = make_plan_i8(rows_pad, stages, ncols, wide);
    BMF_REQUIRE(splits >= pl.slots, "bmf_xf_bits_i8: splits=%d but this shape needs %d slab slots (bmf_xf_bits_i8_slots)", splits, pl.slots);
    if (wide) return bmf_xf_bits_i8w_launch(wide, Abits, ldw, a_tiled, stages, panel, ldp, limbs, out, slab_stride, pl, splits, colscale, stop, s);
    if (limbs == 3) return launch_i8<3>(Abits, ldw, a_tiled, stages, panel, ldp, kp, col0, ncols, out, slab_stride, pl, splits, colscale, stop, s);
    return launch_i8<2>(Abits, ldw, a_tiled, stages, panel, ldp, kp, col0, ncols, out, slab_stride, pl, splits, colscale, stop, s);
}

extern "C" int bmf_xf_bits_i8(const uint32_t* Abits, int64_t rows_pad, int64_t ldw, int64_t red_words, const int8_t* panel, int64_t ldp,
                              int limbs, const float* colscale, int kp, float* out, int64_t slab_stride, int splits, int a_tiled,
                              void* stream) {
    return bmf_xf_bits_i8_launch(Abits, rows_pad, ldw, red_words, panel, ldp, limbs, colscale, kp, 0, kp, out, slab_stride, splits, a_tiled,
                                 nullptr, (hipStream_t)stream);
}

namespace {
// tiled[((tile * groups + grp) * 256 + row) * 16 + w] = bits[(tile * 256 + row) * ldw + 16 * grp + w]; one 16-byte piece per thread
__global__ __launch_bounds__(256) void tile_bits_kernel(const uint32_t* __restrict__ bits, int64_t ldw, int groups, int64_t pieces,
                                                         uint32_t* __restrict__ tiled) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < pieces; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i & 3);
        const int64_t rowg = i >> 2;                 // (tile * groups + grp) * 256 + row
        const int row = (int)(rowg & 255);
        const int64_t tg = rowg >> 8;
        const int64_t tile = tg / groups, grp = tg - tile * groups;
        *reinterpret_cast<u32x4*>(tiled + i * 4) = *reinterpret_cast<const u32x4*>(bits + (tile * 256 + row) * ldw + 16 * grp + 4 * q);
    }
}
}  // namespace

extern "C" int bmf_tile_bits(const uint32_t* bits, int64_t rows_pad, int64_t ldw, int64_t red_words, uint32_t* tiled, void* stream) {
    BMF_REQUIRE(bits && tiled, "bmf_tile_bits: null pointer");
    BMF_REQUIRE(rows_pad > 0 && rows_pad % 256 == 0 && red_words > 0 && red_words % 16 == 0 && ldw >= red_words && ldw % 4 == 0,
                "bmf_tile_bits: rows_pad must be a multiple of 256, red_words of 16, ldw >= red_words and a multiple of 4");
    BMF_REQUIRE(bmf_aligned16(bits) && bmf_aligned16(tiled), "bmf_tile_bits: pointers must be 16-byte aligned");
    const int64_t pieces = rows_pad * (red_words / 4);
    const int64_t blocks = (pieces + 255) / 256;
    BMF_LAUNCH(tile_bits_kernel, dim3((unsigned)(blocks < 65535 ? blocks : 65535)), dim3(256), 0, (hipStream_t)stream, bits, ldw,
               (int)(red_words / 16), pieces, tiled);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

// have_scale: the column scales are already in `scale` (the iteration driver derives them beside the Gram kernel).
// flags != nullptr: conditional rebuild -- `scale` is then the exact scale [kp], `flags` the kp per-column words of the fused
// column-scale step and `colscale_out` [kp] receives 1 / scale when everything is rebuilt (see make_panel_i8_kernel).
// rslabs != nullptr (with flags): rn / 16 extra blocks sum the rcount Gram slabs of rn floats into rout32 / rout64 (either may be null).
int bmf_panel_i8_launch(const double* F64, const float* F, int64_t rows_pad, int64_t ldf, int kp, int limbs, int8_t* panel, int64_t ldp,
                        float* ws, float* scale, bool have_blockmax, const int32_t* stop, hipStream_t s, bool have_scale,
                        const float* flags, float* colscale_out, const float* rslabs, int rcount, int rn, float* rout32, double* rout64) {
    BMF_REQUIRE(F64 && panel && ws && scale && (have_blockmax || F), "bmf_make_panel_i8: null pointer");
    BMF_REQUIRE(rows_pad > 0 && rows_pad % 512 == 0, "bmf_make_panel_i8: rows_pad must be a multiple of 512");
    BMF_REQUIRE((kp == 32 || kp == 64) && ldf >= kp, "bmf_make_panel_i8: kp must be 32 or 64 and ldf >= kp");
    BMF_REQUIRE(limbs == 2 || limbs == 3, "bmf_make_panel_i8: limbs must be 2 or 3");
    BMF_REQUIRE(ldp >= rows_pad && ldp % 16 == 0, "bmf_make_panel_i8: ldp must be >= rows_pad and a multiple of 16");
    BMF_REQUIRE(bmf_aligned16(panel), "bmf_make_panel_i8: panel must be 16-byte aligned");
    const int nblk = (int)(rows_pad / 128);
    if (!have_blockmax) {
        int rc = bmf_blockmax_launch(F, rows_pad, ldf, kp, ws, stop, s);
        if (rc != BMF_OK) return rc;
    }
    if (!have_scale) BMF_LAUNCH(colscale_i8_kernel, dim3((unsigned)(kp / 4)), dim3(256), 0, s, ws, nblk, kp, limbs, scale, stop);
    BMF_REQUIRE(!flags || (have_scale && colscale_out), "bmf_make_panel_i8: the conditional form needs the scale and colscale_out");
    BMF_REQUIRE(!rslabs || (flags && rcount >= 1 && rn >= 1 && rn % 16 == 0 && (rout32 || rout64)), "bmf_make_panel_i8: bad slab-reduction arguments");
    const unsigned grid = (unsigned)(nblk + (rslabs ? rn / 16 : 0));
    if (kp == 32) BMF_LAUNCH(make_panel_i8_kernel<32>, dim3(grid), dim3(256), 0, s, F64, ldf, scale, panel, ldp, limbs, stop, flags, colscale_out, nblk, rslabs, rcount, rn, rout32, rout64);
    else BMF_LAUNCH(make_panel_i8_kernel<64>, dim3(grid), dim3(256), 0, s, F64, ldf, scale, panel, ldp, limbs, stop, flags, colscale_out, nblk, rslabs, rcount, rn, rout32, rout64);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_make_panel_i8(const double* F64, const float* F, int64_t rows_pad, int64_t ldf, int kp, int limbs, int8_t* panel,
                                 int64_t ldp, float* ws, float* scale, void* stream) {
    return bmf_panel_i8_launch(F64, F, rows_pad, ldf, kp, limbs, panel, ldp, ws, scale, false, nullptr, (hipStream_t)stream, false, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr);
}
