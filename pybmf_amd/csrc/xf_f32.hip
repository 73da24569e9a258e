// K1/K2 for a real-valued X (WNMF on non-Boolean data): out = A . F with exact-fp32 MFMA.
//
//   X  @ V   replaces  multiply(W, X) @ V      PyBMF/models/WNMF.py:105
//   X^T @ U  replaces  multiply(W, X).T @ U    PyBMF/models/WNMF.py:98      (A = a transposed copy of X)
//
// v_mfma_f32_32x32x2_f32 takes ONE fp32 per lane per operand (lane (r, h): A[r][k=h], B[k=h][c]).  The reduction
// order inside a group of 8 indices is free as long as both operands agree, so every lane loads 4 consecutive
// floats (16 B) of its row -- half h takes floats 4h..4h+3 of the group -- and MFMA step t multiplies element t of
// both halves.  A streams from HBM (it is read exactly once), the transposed factor FT[j][c] is L2-resident and is
// read straight into registers; no LDS.  The reduction is split over `splits` workgroups per row tile (slabs).
#include "common.h"

#include <cstdlib>

#ifndef BMF_F32_BARRIER
#define BMF_F32_BARRIER 1   // 1: the waves of a workgroup also meet at a barrier every stage: not needed for correctness (the quarters are
                            // private), but it keeps their four DMA streams on the same 16-KiB block: 103 vs 110 us (k = 32), 165 vs 190 (k = 64)
#endif

#ifndef BMF_F32_RESID_BARRIER
#define BMF_F32_RESID_BARRIER 0   // the fused contraction + residual pass is matrix-pipe-bound: see the A/B in its header
#endif

namespace {

template <int NT>
__global__ __launch_bounds__(256) void xf_f32_kernel(const float* __restrict__ A, int64_t lda, int groups_total,
                                                      int groups_per_split, const float* __restrict__ FT, int64_t ldft,
                                                      float* __restrict__ out, int64_t slab_stride, int n_row_tiles,
                                                      const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    constexpr int NC = 32 * NT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int split = blockIdx.x / n_row_tiles;
    const int tile = blockIdx.x - split * n_row_tiles;
    const int g0 = split * groups_per_split;
    const int g1 = min(g0 + groups_per_split, groups_total);
    const int64_t row_base = (int64_t)tile * 128 + wave * 32;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;

    const float* ap = A + (row_base + r) * lda + 4 * h;
    const float* bp = FT + (int64_t)r * ldft + 4 * h;

    int g = g0;
    for (; g + 4 <= g1; g += 4) {  // 4 groups of 8 reduction indices in flight
        f32x4 a[4], b[4][NT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const f32x4*>(ap + 8 * (int64_t)(g + u));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                b[u][nt] = *reinterpret_cast<const f32x4*>(bp + (int64_t)(32 * nt) * ldft + 8 * (int64_t)(g + u));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t], b[u][nt][t], acc[nt], 0, 0, 0);
    }
    for (; g < g1; ++g) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(ap + 8 * (int64_t)g);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(bp + (int64_t)(32 * nt) * ldft + 8 * (int64_t)g);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc[nt], 0, 0, 0);
        }
    }
    float* o = out + (int64_t)split * slab_stride;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t row = row_base + (i & 3) + 8 * (i >> 2) + 4 * h;
            o[row * NC + 32 * nt + r] = acc[nt][i];
        }
}

// LDS-staged flavour (reduction length a multiple of 64): the direct-to-register kernel above makes every wave walk 32
// rows 128 bytes at a time, so with ~4000 waves in flight DRAM sees ~140k interleaved streams of single cache lines and
// delivers 2 TB/s.  Here a workgroup of 4 waves owns a tile of 64 rows; wave (rw, kh) takes rows 32 rw .. + 31 and, of every
// stage of 64 reduction indices, the floats 32 kh .. + 31 -- its QUARTER of the stage (32 rows x 128 bytes = 4 KiB) -- which it
// streams through its own ring of four LDS buffers filled by LDS-DMA three stages ahead; two workgroups per CU.  A quarter is
// read by the wave that fetched it and by nobody else, so no barrier is NEEDED in the loop: each wave is an independent pipeline
// paced by a COUNTED vmcnt that leaves its youngest DMAs in flight (a stage is only ~1000 cycles of MFMA work per wave, less than
// a DRAM round trip under load).  History at 20096 x 5120, k = 32 (400 MB): one vmcnt(0) + __syncthreads per stage 122 us;
// counted waits, 4-deep ring 107 us; factor fragments in lane order instead of 16 bytes of 64 cache lines per load 97 us (the L1
// was busier with the L2-resident factor than with A); of which 81 us without MFMAs and 83 us without HBM (re-fetching one block):
// pure LDS-DMA streaming of this pattern tops out near 5 TB/s on this part, and with compute mixed in the loop is latency-bound
// (bytes in flight / DRAM round trip): 4.0-4.5 TB/s.  The two reduction halves are added through LDS at the end.
// The 16-byte chunks of a quarter row are stored XOR-swizzled with bits 1..3 of the row number (applied on the DMA source, the
// LDS image must stay lane-linear): the 16 lanes a ds_read_b128 serves at a time then cover all 64 banks once.
#ifndef BMF_F32_RING
#define BMF_F32_RING 4      // LDS buffers per wave for k <= 32 (3: three workgroups per CU, look-ahead 2; 4: two, look-ahead 3)
#endif
#ifndef BMF_F32_INTERLEAVE
#define BMF_F32_INTERLEAVE 0   // 1: the loads of a slot are issued between its MFMAs instead of in front of them
#endif
#ifndef BMF_F32_DMA_AUX
#define BMF_F32_DMA_AUX 0   // cache policy bits of the LDS-DMA of A (2 = nt)
#endif
template <int NT>
__global__ __launch_bounds__(256, (NT == 1 && BMF_F32_RING == 3) ? 3 : 2) void xf_f32_ring_kernel(const float* __restrict__ A, int64_t lda, int stages_total,
                                                              int stages_per_split, const float* __restrict__ FT, int64_t ldft,
                                                              float* __restrict__ out, int64_t slab_stride, int n_row_tiles,
                                                              int a_tiled, int b_frag, const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    constexpr int NC = 32 * NT;
    constexpr int SF = 64, TR = 64;            // floats of a row per stage, rows per tile
    constexpr int STAGE_BYTES = TR * SF * 4;   // 16 KiB
    constexpr int RING = NT == 1 ? BMF_F32_RING : 4;   // = look-ahead + 1; with 3 the buffer of a stage is its slot of the unrolled loop
    constexpr int LA = RING - 1;
    constexpr int DPW = STAGE_BYTES / 1024 / 4;  // DMA instructions per wave per stage
    static_assert(RING == 3 || RING == 4, "ring of 3 or 4 buffers");
    static_assert(16 * NT * 256 * 4 * 2 <= RING * STAGE_BYTES, "the final exchange of the two reduction halves reuses the ring");
    __shared__ __attribute__((aligned(16))) char smem[RING * STAGE_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rw = wave & 1, kh = wave >> 1;
    const int r = lane & 31, h = lane >> 5;
    const int split = blockIdx.x / n_row_tiles;
    const int tile = blockIdx.x - split * n_row_tiles;
    const int s0 = split * stages_per_split;
    const int s1 = min(s0 + stages_per_split, stages_total);
    const int64_t tile_row = (int64_t)tile * TR;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;

    if (s0 < s1) {
        // DMA piece i of a wave (1 KiB) = rows 8 i .. 8 i + 7 of its quarter; lane l: row 8 i + (l >> 3), LDS chunk l & 7 <- source chunk
        // (l & 7) ^ ((row >> 1) & 7).  A tiled copy of A (bmf_tile_f32) holds every quarter as 4 contiguous KiB that already ARE the
        // LDS image.  Stages past the end re-fetch the last one into a free buffer: the vmcnt arithmetic stays static.
        const int wq = 2 * kh + rw;               // quarter index inside a tiled block
        const float* dma_src[DPW];
        const int64_t stage_stride = a_tiled ? TR * SF : SF;
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            const int rl = 8 * i + (lane >> 3);
#ifdef BMF_F32_EXP_STAGEMAJOR
            dma_src[i] = a_tiled ? A + (int64_t)tile * (TR * SF) + wq * 1024 + i * 256 + lane * 4
#else
            dma_src[i] = a_tiled ? A + (int64_t)tile * stages_total * (TR * SF) + wq * 1024 + i * 256 + lane * 4
#endif
                                 : A + (tile_row + 32 * rw + rl) * lda + 32 * kh + (((lane & 7) ^ ((rl >> 1) & 7)) << 2);
        }
        char* const my_ring = smem + wave * (RING * 4096);   // this wave's four quarter buffers
        auto issue_dma = [&](int stage, int buf) {
#ifdef BMF_F32_EXP_L2ONLY   // ablation: every stage re-fetches the split's first block (L2 hits instead of HBM)
            const int st = s0;
#else
            int st = min(max(stage, s0), s1 - 1);
#endif
#ifdef BMF_F32_EXP_ROT        // timing experiment: every workgroup starts its run of stages somewhere else
            st += (tile * 7 + split * 3) % (s1 - s0); if (st >= s1) st -= s1 - s0;
#endif
#ifdef BMF_F32_EXP_STAGEMAJOR // timing experiment: blocks of one stage of all tiles are neighbours (the data is then wrong)
            const int64_t stage_stride = (int64_t)n_row_tiles * (TR * SF);
#endif
#pragma unroll
            for (int i = 0; i < DPW; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dma_src[i] + (int64_t)st * stage_stride),
                                                 (__attribute__((address_space(3))) void*)(my_ring + buf * 4096 + i * 1024), 16, 0, BMF_F32_DMA_AUX);
        };
        // Read from the transposed factor FT[j][c] these are 16 bytes of 64 different cache lines per instruction, and the L1 / TA
        // then spends more cycles on the (L2-resident!) factor than on the DMA of A: 107 us per launch at 20096 x 5120, k = 32,
        // 95 us with the MFMAs removed.  In fragment order (bmf_frag_f32) every instruction is one contiguous KiB.
        const float* bp = b_frag ? FT + (kh * 4) * (NT * 256) + lane * 4 : FT + (int64_t)r * ldft + 32 * kh + 4 * h;
        const int64_t b_stage = b_frag ? 8 * NT * 256 : SF, b_u = b_frag ? NT * 256 : 8, b_nt = b_frag ? 256 : 32 * ldft;
        // The loaded fragments are written by the memory system long after the load instruction has issued, which the compiler
        // does not know: any register copy it places between the load and the counted wait moves stale data (the first version of
        // this loop had a `b_cur = b_nxt` sunk below the next loads).  So there are THREE fragment sets, each written at exactly one
        // place of a loop unrolled by three -- slot k (stage s) loads the fragments of stage s + 2 into set (k + 2) % 3 and
        // multiplies with set k -- and no prologue: the loop starts three stages early with the compute switched off.
        // Loads retire in order.  Issue order per slot: fragments of s + 2, then the DMAs of s + 3; at the top of slot s everything
        // younger than the fragments of s -- DMAs of s + 1, fragments of s + 1, DMAs of s + 2 -- may stay in flight.
        f32x4 bq[3][4][NT];
        // A fragments: row r of the quarter, chunk (2 u + h) ^ ((r >> 1) & 7)
        unsigned a_off[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a_off[u] = (unsigned)(r * 128 + (((2 * u + h) ^ ((r >> 1) & 7)) << 4));
        const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)my_ring;

        for (int sb = s0 - 3; sb < s1; sb += 3) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int s = sb + k;
                const bool live = s >= s0 && s < s1;   // wave-uniform
                if (live) {
#if defined(BMF_F32_EXP_NOB) && defined(BMF_F32_EXP_NODMA)   // ablations: without the factor loads / without the DMA of A
                    constexpr int YOUNGER = 0;
#elif defined(BMF_F32_EXP_NOB)
                    constexpr int YOUNGER = (LA - 1) * DPW;
#elif defined(BMF_F32_EXP_NODMA)
                    constexpr int YOUNGER = 4 * NT;
#else
                    constexpr int YOUNGER = (LA - 1) * DPW + 4 * NT;
#endif
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");   // this wave's quarter of stage s, its fragments
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(bq[k][u][nt]));
#if BMF_F32_BARRIER
                    __builtin_amdgcn_s_barrier();
#endif
                }
#if BMF_F32_INTERLEAVE
                // The loads of the coming stages go out BETWEEN the MFMAs: a dependent MFMA cannot issue for 64 cycles anyway, and a
                // vector-memory instruction placed in that shadow costs the wave nothing, where the same eight instructions in front
                // of the chain held it (and, with two waves per SIMD, often the matrix pipe) for their 30 - 100 issue cycles each.
                // Same instructions, same order among the loads (the vmcnt arithmetic is unchanged), same arithmetic.
                f32x4 a[4];
                if (live) {
                    const unsigned abase = lds_base + (unsigned)((RING == 3 ? k : (s - s0) & 3) * 4096);
#pragma unroll
                    for (int u = 0; u < 4; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(a[u]) : "v"(abase + a_off[u]));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(a[u]));
                }
                const float* p = bp + (int64_t)min(max(s + 2, s0), s1 - 1) * b_stage;
                const int dst = min(max(s + LA, s0), s1 - 1), dbuf = RING == 3 ? (k + 2) % 3 : (s + LA - s0) & 3;
#pragma unroll
                for (int j = 0; j < 8; ++j) {   // pair j of the sixteen MFMAs, then load j of the slot (old order: the four factor loads, the four DMAs)
                    if (live) {
#pragma unroll
                        for (int t = 2 * (j & 1); t < 2 * (j & 1) + 2; ++t)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j >> 1][t], bq[k][j >> 1][nt][t], acc[nt], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (s < s1) {
                        static_assert(DPW == 4, "one DMA piece per pair of MFMAs of the second half");
                        if (j < 4) {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(bq[(k + 2) % 3][j][nt]) : "v"(p + nt * b_nt + j * b_u) : "memory");
                        } else {
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dma_src[j - 4] + (int64_t)dst * stage_stride),
                                                             (__attribute__((address_space(3))) void*)(my_ring + dbuf * 4096 + (j - 4) * 1024), 16, 0, BMF_F32_DMA_AUX);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#else
                if (s < s1) {
                    const float* p = bp + (int64_t)min(max(s + 2, s0), s1 - 1) * b_stage;
#ifndef BMF_F32_EXP_NOB
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(bq[(k + 2) % 3][u][nt]) : "v"(p + nt * b_nt + u * b_u) : "memory");
#else
                    if (s == s0 - 3) bq[(k + 2) % 3][0][0] = *reinterpret_cast<const f32x4*>(p);
#endif
#ifndef BMF_F32_EXP_NODMA
                    issue_dma(s + LA, RING == 3 ? (k + 2) % 3 : (s + LA - s0) & 3);   // into the buffer of stage s - 1, which this wave has finished reading
#endif
                }
                if (live) {
                    f32x4 a[4];
                    const unsigned abase = lds_base + (unsigned)((RING == 3 ? k : (s - s0) & 3) * 4096);
#pragma unroll
                    for (int u = 0; u < 4; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(a[u]) : "v"(abase + a_off[u]));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(a[u]));
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
#ifdef BMF_F32_EXP_NOMFMA   // ablation: one MFMA per stage instead of sixteen
                                if (u == 0 && t == 0)
#endif
                                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t], bq[k][u][nt][t], acc[nt], 0, 0, 0);
                            }
                }
            }
        }
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus DMAs / factor loads of the last stages
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(bq[k][u][nt]));
        __syncthreads();
        // the two reduction halves of a row group meet in LDS
        float* ex = reinterpret_cast<float*>(smem) + rw * (16 * NT * 64);
        if (kh == 1) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) ex[(nt * 16 + i) * 64 + lane] = acc[nt][i];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[nt][i] += ex[(nt * 16 + i) * 64 + lane];
        }
    }
    if (kh == 0) {
        float* o = out + (int64_t)split * slab_stride;
        const int64_t row_base = tile_row + 32 * rw;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t row = row_base + (i & 3) + 8 * (i >> 2) + 4 * h;
                o[row * NC + 32 * nt + r] = acc[nt][i];
            }
    }
}

// The tiled contraction for k <= 32 on the bf16 matrix instruction with BOTH operands split three ways (round 5): the passes of config #2
// are co-bound by the LDS-DMA stream of A (81 us alone at 20096 x 5120) and the 16 dependent v_mfma_f32_32x32x2_f32 per wave and stage
// (83 us alone: 16 x 64 cycles), and the two overlap imperfectly in an in-order wave (90 - 97 us).  x = hi + mid + lo exactly
// (bmf_split3_bf16), six products per k-step of 16 on v_mfma_f32_32x32x16_bf16: 12 x 32 cycles per wave and stage, and the same
// fp32 result to 2^-23.  The register grouping of the fp32 kernel serves unchanged -- k-step ks takes registers (u = 2 ks, t) and
// (u = 2 ks + 1, t) of both operands, the same reduction index in the same slot on both sides.  A is split in the kernel (~90 vector
// instructions per stage); the factor arrives pre-split ("frag3", six 16-byte pieces per lane and stage-half instead of four: split
// here as well the vector work would be the new bound): real_update_kernel / bmf_frag_bf16x3 write it.
//   frag3[(((st * 2 + kh) * 2 + ks) * 3 + split) * 256 + lane * 4 + w],  split 0 / 1 / 2 = hi / mid / lo, dword w = elements 2 w, 2 w + 1 of
//   the k-step's eight: element j = F[64 st + 32 kh + 8 (2 ks + (j >> 2)) + 4 h + (j & 3)][r],  lane = 32 h + r.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8q;
#define BMF_BF3_PRODUCTS(accv, AH, AM, AL, BH, BM, BL)                                                                              \
    do {                                                                                                                           \
        accv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8q, AL), __builtin_bit_cast(bf16x8q, BH), accv, 0, 0, 0); \
        accv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8q, AM), __builtin_bit_cast(bf16x8q, BM), accv, 0, 0, 0); \
        accv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8q, AH), __builtin_bit_cast(bf16x8q, BL), accv, 0, 0, 0); \
        accv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8q, AM), __builtin_bit_cast(bf16x8q, BH), accv, 0, 0, 0); \
        accv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8q, AH), __builtin_bit_cast(bf16x8q, BM), accv, 0, 0, 0); \
        accv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8q, AH), __builtin_bit_cast(bf16x8q, BH), accv, 0, 0, 0); \
    } while (0)

__global__ __launch_bounds__(256, 2) void xf_f32_bf3_ring_kernel(const float* __restrict__ A, int stages_total, int stages_per_split,
                                                                 const uint32_t* __restrict__ F3, float* __restrict__ out, int64_t slab_stride,
                                                                 int n_row_tiles, const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    constexpr int NC = 32;
    constexpr int SF = 64, TR = 64;
    constexpr int STAGE_BYTES = TR * SF * 4;
#ifndef BMF_BF3_RING
#define BMF_BF3_RING 4   // (5: 80 KiB per workgroup, four stages ahead -- timing experiment)
#endif
    constexpr int RING = BMF_BF3_RING, LA = RING - 1;
    constexpr int DPW = STAGE_BYTES / 1024 / 4;
    constexpr int NB = 6;   // pieces of the split factor per lane and stage: (k-step, hi / mid / lo)
    __shared__ __attribute__((aligned(16))) char smem[RING * STAGE_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rw = wave & 1, kh = wave >> 1;
    const int r = lane & 31, h = lane >> 5;
    const int split = blockIdx.x / n_row_tiles;
    const int tile = blockIdx.x - split * n_row_tiles;
    const int s0 = split * stages_per_split;
    const int s1 = min(s0 + stages_per_split, stages_total);
    const int64_t tile_row = (int64_t)tile * TR;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    if (s0 < s1) {
        const int wq = 2 * kh + rw;
        const float* dma_src[DPW];
#pragma unroll
        for (int i = 0; i < DPW; ++i) dma_src[i] = A + (int64_t)tile * stages_total * (TR * SF) + wq * 1024 + i * 256 + lane * 4;
        char* const my_ring = smem + wave * (RING * 4096);
        auto issue_dma = [&](int stage, int buf) {
            const int st = min(max(stage, s0), s1 - 1);
#pragma unroll
            for (int i = 0; i < DPW; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dma_src[i] + (int64_t)st * (TR * SF)),
                                                 (__attribute__((address_space(3))) void*)(my_ring + buf * 4096 + i * 1024), 16, 0, BMF_F32_DMA_AUX);
        };
        const uint32_t* bp = F3 + kh * (NB * 256) + lane * 4;
        const int64_t b_stage = 2 * NB * 256;
        // three fragment sets, each written at one place of a loop unrolled by three (see xf_f32_ring_kernel)
        u32x4 bq[3][NB];
        unsigned a_off[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a_off[u] = (unsigned)(r * 128 + (((2 * u + h) ^ ((r >> 1) & 7)) << 4));
        const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)my_ring;

        // (the loop starts LA stages early -- rounded up to the three fragment sets -- with the compute switched off; at the top of slot s
        // everything younger than the fragments of s may stay in flight: the DMA of slot s - 2, the fragments and the DMA of slot s - 1)
        for (int sb = s0 - 3 * ((LA + 2) / 3); sb < s1; sb += 3) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int s = sb + k;
                const bool live = s >= s0 && s < s1;   // wave-uniform
                if (live) {
#if defined(BMF_BF3_EXP_NOB) && defined(BMF_BF3_EXP_NODMA)
                    constexpr int YOUNGER = 0;
#elif defined(BMF_BF3_EXP_NOB)
                    constexpr int YOUNGER = (LA - 1) * DPW;
#elif defined(BMF_BF3_EXP_NODMA)
                    constexpr int YOUNGER = NB;
#else
                    constexpr int YOUNGER = 2 * DPW + NB;
#endif
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");   // this wave's quarter of stage s, its fragments
#pragma unroll
                    for (int q = 0; q < NB; ++q) asm volatile("" : "+v"(bq[k][q]));
                }
                if (s < s1) {
                    const uint32_t* p = bp + (int64_t)min(max(s + 2, s0), s1 - 1) * b_stage;
#pragma unroll
                    for (int q = 0; q < NB; ++q)
#ifdef BMF_BF3_EXP_NOB    // timing experiments only (wrong results): -DBMF_BF3_EXP_NOB / NODMA / NOSPLIT / NOMFMA
                        asm volatile("v_mov_b32 %0, 0" : "=&v"(bq[(k + 2) % 3][q][0]) : "v"(p + q * 256) : "memory");
#else
                        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(bq[(k + 2) % 3][q]) : "v"(p + q * 256) : "memory");
#endif
#ifndef BMF_BF3_EXP_NODMA
                    if (s + LA >= s0) issue_dma(s + LA, (s + LA - s0) % RING);
#endif
                }
                if (live) {
                    f32x4 a[4];
                    const unsigned abase = lds_base + (unsigned)(((s - s0) % RING) * 4096);
#pragma unroll
                    for (int u = 0; u < 4; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(a[u]) : "v"(abase + a_off[u]));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(a[u]));
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const float x[8] = {a[2 * ks][0], a[2 * ks][1], a[2 * ks][2], a[2 * ks][3], a[2 * ks + 1][0], a[2 * ks + 1][1], a[2 * ks + 1][2], a[2 * ks + 1][3]};
                        u32x4 ah, am, al;
#ifdef BMF_BF3_EXP_NOSPLIT
                        ah = u32x4{__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])};
                        am = u32x4{__float_as_uint(x[4]), __float_as_uint(x[5]), __float_as_uint(x[6]), __float_as_uint(x[7])};
                        al = ah;
#else
                        bmf_split3_bf16(x, ah, am, al);
#endif
#ifdef BMF_BF3_EXP_NOMFMA
                        asm volatile("" : "+v"(acc) : "v"(ah), "v"(am), "v"(al), "v"(bq[k][3 * ks]), "v"(bq[k][3 * ks + 1]), "v"(bq[k][3 * ks + 2]));
#else
                        BMF_BF3_PRODUCTS(acc, ah, am, al, bq[k][3 * ks], bq[k][3 * ks + 1], bq[k][3 * ks + 2]);
#endif
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus DMAs / factor loads of the last stages
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int q = 0; q < NB; ++q) asm volatile("" : "+v"(bq[k][q]));
        __syncthreads();
        // the two reduction halves of a row group meet in LDS
        float* ex = reinterpret_cast<float*>(smem) + rw * (16 * 64);
        if (kh == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) ex[i * 64 + lane] = acc[i];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] += ex[i * 64 + lane];
        }
    }
    if (kh == 0) {
        float* o = out + (int64_t)split * slab_stride;
        const int64_t row_base = tile_row + 32 * rw;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t row = row_base + (i & 3) + 8 * (i >> 2) + 4 * h;
            o[row * NC + r] = acc[i];
        }
    }
}

// frag3 of a 32-column factor (layout above): one (stage-half, k-step) = three 16-byte pieces per thread
__global__ __launch_bounds__(256) void frag_bf16x3_kernel(const float* __restrict__ F, int64_t items, uint32_t* __restrict__ frag3,
                                                           const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (int64_t)gridDim.x * 256) {
        const int lane = (int)(i & 63), r = lane & 31, h = lane >> 5;
        const int64_t g = i >> 6;              // (st * 2 + kh) * 2 + ks
        const int ks = (int)(g & 1), kh = (int)((g >> 1) & 1);
        const int64_t st = g >> 2;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = F[(64 * st + 32 * kh + 8 * (2 * ks + (j >> 2)) + 4 * h + (j & 3)) * 32 + r];
        u32x4 hi, mid, lo;
        bmf_split3_bf16(x, hi, mid, lo);
        uint32_t* o = frag3 + (g * 3) * 256 + lane * 4;
        *reinterpret_cast<u32x4*>(o) = hi;
        *reinterpret_cast<u32x4*>(o + 256) = mid;
        *reinterpret_cast<u32x4*>(o + 512) = lo;
    }
}

// The same contraction with the RESIDUAL SUMS of the pass folded in (round 3; k <= 32): out = A F as above, and
//   sums[0] += sum |A - G F^T|,  sums[1] += sum (A - G F^T)^2     over the cells of A
// for a second factor G with one row per row of A.  WNMF on real-valued X reads X three times per iteration (X V, X^T U, the residual
// pass for MAE); with A = X^T, F = U, G = V the third pass rides in the second: the quarter a wave holds as the A operand of
// out += A_q U_q IS the block of cells whose residual it can form, in the layout of that product's own accumulator -- lane (r, h)
// has row j = r of X^T and the floats i = 8 u + 4 h + t, and P^T = U_q V_q^T with M = i, N = j leaves exactly those cells in the
// lane's 16 registers (accumulators started at -x: the residual pass of resid_f32.hip, transposed).
// The residual product runs on the bf16 MFMA with both factors split into two bf16 addends (hi hi + hi lo + lo hi: the product is
// right to 2^-16 per cell, unbiased -- sums of 1e8 cells do not see it; the MAE pass of the Boolean path, mae.hip, does the same):
// 6 MFMAs of 32 cycles per quarter-stage.  A first version used the exact-fp32 MFMA for it as well (16 more MFMAs of 64 cycles):
// the matrix pipe then needed 2.0 us per stage against the 2.2 us the LDS-DMA stream delivers one in, the two no longer overlapped
// (ablation: 108 us of stream + 36 us per product, 181 us in all), and the fused pass was no faster than the two passes it replaces.
// Costs now: U a second time from L2 as bf16 pairs in row-fragment order (bmf_frag_rows_bf16, 4 KiB per wave and stage), the wave's
// 32 rows of V split in registers once per tile, 6 MFMAs and 32 element-wise instructions per stage.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8r;
#ifndef BMF_F32_RESID_RING
#define BMF_F32_RESID_RING BMF_F32_RING
#endif
// BF3: the contraction out += A_q U_q on the bf16 matrix instruction with both operands split three ways (xf_f32_bf3_ring_kernel above): F3 =
// the factor's frag3 order instead of FT.
template <bool BF3>
__global__ __launch_bounds__(256, BMF_F32_RESID_RING == 3 ? 3 : 2) void xf_f32_resid_ring_kernel(const float* __restrict__ A, int stages_total, int stages_per_split,
                                                                    const float* __restrict__ FT, const uint32_t* __restrict__ F3, const uint32_t* __restrict__ Frf,
                                                                    const float* __restrict__ Grow, float* __restrict__ out, int64_t slab_stride,
                                                                    int n_row_tiles, double* __restrict__ sums, const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    constexpr int NT = 1, KP = 32, VL = 4;   // VL: 16-byte pieces of the split factor per lane and stage: (k-step, hi / lo)
    constexpr int NC = 32 * NT;
    constexpr int SF = 64, TR = 64;
    constexpr int STAGE_BYTES = TR * SF * 4;
    constexpr int RING = BMF_F32_RESID_RING, LA = RING - 1;
    constexpr int DPW = STAGE_BYTES / 1024 / 4;
    static_assert(RING == 3 || RING == 4, "ring of 3 or 4 buffers");
    __shared__ __attribute__((aligned(16))) char smem[RING * STAGE_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rw = wave & 1, kh = wave >> 1;
    const int r = lane & 31, h = lane >> 5;
    const int split = blockIdx.x / n_row_tiles;
    const int tile = blockIdx.x - split * n_row_tiles;
    const int s0 = split * stages_per_split;
    const int s1 = min(s0 + stages_per_split, stages_total);
    const int64_t tile_row = (int64_t)tile * TR;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
    double s_abs = 0.0, s_sq = 0.0;

    if (s0 < s1) {
        // this lane's row of G (row 32 rw + r of the tile) as the B operand of v_mfma_f32_32x32x16_bf16: k = 16 ks + 8 h .. + 7, split
        // into bf16 hi + lo once per tile
        u32x4 gh[2], gl[2];
        {
            const float* gp = Grow + (tile_row + 32 * rw + r) * KP + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(gp + 16 * ks), v1 = *reinterpret_cast<const f32x4*>(gp + 16 * ks + 4);
                const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint16_t h0 = bf16_bits(v[2 * q]), h1 = bf16_bits(v[2 * q + 1]);
                    const uint16_t l0 = bf16_bits(v[2 * q] - bf16_to_f32(h0)), l1 = bf16_bits(v[2 * q + 1] - bf16_to_f32(h1));
                    gh[ks][q] = (unsigned)h0 | ((unsigned)h1 << 16);
                    gl[ks][q] = (unsigned)l0 | ((unsigned)l1 << 16);
                }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) { asm volatile("" : "+v"(gh[ks])); asm volatile("" : "+v"(gl[ks])); }   // the loads' wait sits before the pipeline starts
        }
        const int wq = 2 * kh + rw;               // quarter index inside a tiled block
        const float* dma_src[DPW];
#pragma unroll
        for (int i = 0; i < DPW; ++i) dma_src[i] = A + (int64_t)tile * stages_total * (TR * SF) + wq * 1024 + i * 256 + lane * 4;
        char* const my_ring = smem + wave * (RING * 4096);
        auto issue_dma = [&](int stage, int buf) {
            const int st = min(max(stage, s0), s1 - 1);
#pragma unroll
            for (int i = 0; i < DPW; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dma_src[i] + (int64_t)st * (TR * SF)),
                                                 (__attribute__((address_space(3))) void*)(my_ring + buf * 4096 + i * 1024), 16, 0, BMF_F32_DMA_AUX);
        };
        const float* bp = FT + (kh * 4) * (NT * 256) + lane * 4;          // fragment order of the contraction (bmf_frag_f32)
        const int64_t b_stage = 8 * NT * 256, b_u = NT * 256;
        constexpr int NB3 = 6;                                            // BF3: pieces (k-step, hi / mid / lo) of frag3 per lane and stage-half
        const uint32_t* bp3 = F3 + kh * (NB3 * 256) + lane * 4;
        constexpr int NLOAD = BF3 ? NB3 : 4 * NT;
        const uint32_t* fp = Frf + kh * (VL * 256) + lane * 4;            // bmf_frag_rows_bf16: rows 32 kh .. of a stage, pieces (ks, hi / lo)
        // three fragment sets, each written at one place of a loop unrolled by three (see xf_f32_ring_kernel); per slot the loads of
        // stage s + 2 -- 4 NT pieces for the contraction, VL for the residual product -- go out before the DMAs of stage s + 3
        f32x4 bq[3][4][NT];
        u32x4 bq3[3][NB3];
        u32x4 fq[3][VL];
        unsigned a_off[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a_off[u] = (unsigned)(r * 128 + (((2 * u + h) ^ ((r >> 1) & 7)) << 4));
        const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)my_ring;

        for (int sb = s0 - 3; sb < s1; sb += 3) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int s = sb + k;
                const bool live = s >= s0 && s < s1;   // wave-uniform
                if (live) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LA - 1) * DPW + NLOAD + VL) : "memory");   // this wave's quarter of stage s, its fragments
                    if constexpr (BF3) {
#pragma unroll
                        for (int q = 0; q < NB3; ++q) asm volatile("" : "+v"(bq3[k][q]));
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(bq[k][u][nt]));
                    }
#pragma unroll
                    for (int q = 0; q < VL; ++q) asm volatile("" : "+v"(fq[k][q]));
#if BMF_F32_RESID_BARRIER
                    __builtin_amdgcn_s_barrier();
#endif
                }
                if (s < s1) {
                    const int sn = min(max(s + 2, s0), s1 - 1);
                    if constexpr (BF3) {
                        const uint32_t* p3 = bp3 + (int64_t)sn * (2 * NB3 * 256);
#pragma unroll
                        for (int q = 0; q < NB3; ++q) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(bq3[(k + 2) % 3][q]) : "v"(p3 + q * 256) : "memory");
                    } else {
                        const float* p = bp + (int64_t)sn * b_stage;
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(bq[(k + 2) % 3][u][nt]) : "v"(p + nt * 256 + u * b_u) : "memory");
                    }
                    const uint32_t* p2 = fp + (int64_t)sn * (2 * VL * 256);
#pragma unroll
                    for (int q = 0; q < VL; ++q) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(fq[(k + 2) % 3][q]) : "v"(p2 + q * 256) : "memory");
                    issue_dma(s + LA, RING == 3 ? (k + 2) % 3 : (s + LA - s0) & 3);
                }
                if (live) {
                    f32x4 a[4];
                    const unsigned abase = lds_base + (unsigned)((RING == 3 ? k : (s - s0) & 3) * 4096);
#pragma unroll
                    for (int u = 0; u < 4; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(a[u]) : "v"(abase + a_off[u]));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(a[u]));
                    // the residual of the quarter: accumulator = -x (cell (j = r, i = 8 u + 4 h + t) in register 4 u + t), + U_q V_q^T
                    f32x16 res;
#pragma unroll
                    for (int e = 0; e < 16; ++e) res[e] = -a[e >> 2][e & 3];
                    if constexpr (BF3) {
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            const float x[8] = {a[2 * ks][0], a[2 * ks][1], a[2 * ks][2], a[2 * ks][3], a[2 * ks + 1][0], a[2 * ks + 1][1], a[2 * ks + 1][2], a[2 * ks + 1][3]};
                            u32x4 ah, am, al;
                            bmf_split3_bf16(x, ah, am, al);
                            BMF_BF3_PRODUCTS(acc[0], ah, am, al, bq3[k][3 * ks], bq3[k][3 * ks + 1], bq3[k][3 * ks + 2]);
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int t = 0; t < 4; ++t)
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t], bq[k][u][nt][t], acc[nt], 0, 0, 0);
                    }
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {   // pieces of fq: 2 ks = hi, 2 ks + 1 = lo
                        res = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8r, fq[k][2 * ks + 1]), __builtin_bit_cast(bf16x8r, gh[ks]), res, 0, 0, 0);
                        res = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8r, fq[k][2 * ks]), __builtin_bit_cast(bf16x8r, gl[ks]), res, 0, 0, 0);
                        res = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8r, fq[k][2 * ks]), __builtin_bit_cast(bf16x8r, gh[ks]), res, 0, 0, 0);
                    }
                    float pa = 0.f, ps = 0.f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        pa += fabsf(res[e]);
                        ps = fmaf(res[e], res[e], ps);
                    }
                    s_abs += (double)pa;
                    s_sq += (double)ps;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus DMAs / factor loads of the last stages
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if constexpr (BF3) {
#pragma unroll
                for (int q = 0; q < NB3; ++q) asm volatile("" : "+v"(bq3[k][q]));
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(bq[k][u][nt]));
            }
#pragma unroll
            for (int q = 0; q < VL; ++q) asm volatile("" : "+v"(fq[k][q]));
        }
        __syncthreads();
        float* ex = reinterpret_cast<float*>(smem) + rw * (16 * NT * 64);
        if (kh == 1) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) ex[(nt * 16 + i) * 64 + lane] = acc[nt][i];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[nt][i] += ex[(nt * 16 + i) * 64 + lane];
        }
    }
    if (kh == 0) {
        float* o = out + (int64_t)split * slab_stride;
        const int64_t row_base = tile_row + 32 * rw;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t row = row_base + (i & 3) + 8 * (i >> 2) + 4 * h;
                o[row * NC + 32 * nt + r] = acc[nt][i];
            }
    }
    s_abs = wave_sum(s_abs);
    s_sq = wave_sum(s_sq);
    __syncthreads();   // the exchange above has been read
    double (*red)[2] = reinterpret_cast<double (*)[2]>(smem);
    if (lane == 0) { red[wave][0] = s_abs; red[wave][1] = s_sq; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const double t = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
        if (t != 0.0) atomicAdd(&sums[threadIdx.x], t);
    }
}

// A 32-column factor as bf16 pairs in the order the fused kernel's lanes consume it: stage st (64 rows of F), row half kh, piece
// q = 2 ks + (0: hi, 1: lo), lane (r, h) -> the eight bf16 of row 64 st + 32 kh + r, columns 16 ks + 8 h .. + 7:
//   frag[(((st * 2 + kh) * 4 + q) * 64 + 32 h + r) * 4 + w] = bf16(F[row][16 ks + 8 h + 2 w]) | bf16(F[row][.. + 2 w + 1]) << 16,  F = hi + lo
__global__ __launch_bounds__(256) void frag_rows_bf16_kernel(const float* __restrict__ F, int64_t pieces, uint32_t* __restrict__ frag,
                                                              const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < pieces; i += (int64_t)gridDim.x * 256) {
        const int lane = (int)(i & 63), r = lane & 31, h = lane >> 5;
        const int64_t g = i >> 6;
        const int q = (int)(g & 3), ks = q >> 1, lo = q & 1;
        const int kh = (int)((g >> 2) & 1);
        const int64_t st = g >> 3;
        const float* src = F + (64 * st + 32 * kh + r) * 32 + 16 * ks + 8 * h;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
        const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        u32x4 o;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint16_t b0 = bf16_bits(v[2 * w]), b1 = bf16_bits(v[2 * w + 1]);
            if (lo) {
                b0 = bf16_bits(v[2 * w] - bf16_to_f32(b0));
                b1 = bf16_bits(v[2 * w + 1] - bf16_to_f32(b1));
            }
            o[w] = (unsigned)b0 | ((unsigned)b1 << 16);
        }
        *reinterpret_cast<u32x4*>(frag + i * 4) = o;
    }
}

// Block (tile, st) = rows 64 tile .. + 63, floats 64 st .. + 63, as four quarters q = 2 kh + rw (rows 32 rw .. + 31, floats 32 kh .. + 31)
// of 4 contiguous KiB each, stored as the swizzled LDS image of the ring kernels:
//   tiled[((((tile * stages + st) * 4 + q) * 32 + rl) * 8 + c) * 4 + e] = X[(64 tile + 32 rw + rl) * lda + 64 st + 32 kh + 4 (c ^ ((rl >> 1) & 7)) + e]
// one 16-byte chunk per thread
__global__ __launch_bounds__(256) void tile_f32_kernel(const float* __restrict__ X, int64_t lda, int stages, int64_t chunks,
                                                        float* __restrict__ tiled) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i & 7), rl = (int)((i >> 3) & 31), q = (int)((i >> 8) & 3);
        const int rw = q & 1, kh = q >> 1;
        const int64_t blk = i >> 10;
        const int64_t tile = blk / stages, st = blk - tile * stages;
        *reinterpret_cast<f32x4*>(tiled + i * 4) =
            *reinterpret_cast<const f32x4*>(X + (tile * 64 + 32 * rw + rl) * lda + st * 64 + 32 * kh + 4 * (c ^ ((rl >> 1) & 7)));
    }
}

// The factor in the order the ring kernel's lanes consume it: stage st (64 reduction indices = rows of F), reduction half kh,
// group u, column tile nt, lane (r, h) -> the four rows 64 st + 32 kh + 8 u + 4 h + t of column 32 nt + r:
//   frag[((((st * 2 + kh) * 4 + u) * NT + nt) * 64 + 32 h + r) * 4 + t] = F[(64 st + 32 kh + 8 u + 4 h + t) * kp + 32 nt + r]
__global__ __launch_bounds__(256) void frag_f32_kernel(const float* __restrict__ F, int kp, int64_t pieces, float* __restrict__ frag,
                                                        const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    const int NT = kp / 32;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < pieces; i += (int64_t)gridDim.x * 256) {
        const int lane = (int)(i & 63), r = lane & 31, h = lane >> 5;
        const int64_t g = i >> 6;
        const int nt = (int)(g % NT);
        const int64_t g2 = g / NT;
        const int u = (int)(g2 & 3), kh = (int)((g2 >> 2) & 1);
        const int64_t st = g2 >> 3;
        const float* src = F + (64 * st + 32 * kh + 8 * u + 4 * h) * kp + 32 * nt + r;
        f32x4 v = {src[0], src[kp], src[2 * kp], src[3 * kp]};
        *reinterpret_cast<f32x4*>(frag + i * 4) = v;
    }
}

}  // namespace

int bmf_frag_f32_launch(const float* F, int64_t rows_pad, int kp, float* frag, const int32_t* stop, hipStream_t s) {
    BMF_REQUIRE(F && frag, "bmf_frag_f32: null pointer");
    BMF_REQUIRE(rows_pad > 0 && rows_pad % 64 == 0 && (kp == 32 || kp == 64), "bmf_frag_f32: rows_pad must be a positive multiple of 64, kp 32 or 64");
    BMF_REQUIRE(bmf_aligned16(frag), "bmf_frag_f32: frag must be 16-byte aligned");
    const int64_t pieces = rows_pad * kp / 4;
    const int64_t blocks = (pieces + 255) / 256;
    BMF_LAUNCH(frag_f32_kernel, dim3((unsigned)(blocks < 65535 ? blocks : 65535)), dim3(256), 0, s, F, kp, pieces, frag, stop);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_frag_f32(const float* F, int64_t rows_pad, int kp, float* frag, void* stream) {
    return bmf_frag_f32_launch(F, rows_pad, kp, frag, nullptr, (hipStream_t)stream);
}

int bmf_xf_f32_launch(const float* A, int64_t rows_pad, int64_t lda, int64_t red, const float* FT, int64_t ldft, int kp, float* out,
                      int64_t slab_stride, int splits, int a_tiled, int b_frag, const int32_t* stop, hipStream_t s) {
    BMF_REQUIRE(A && FT && out, "bmf_xf_f32: null pointer");
    BMF_REQUIRE(rows_pad > 0 && rows_pad % 128 == 0, "bmf_xf_f32: rows_pad must be a positive multiple of 128");
    BMF_REQUIRE(red > 0 && red % 8 == 0, "bmf_xf_f32: red=%lld must be a positive multiple of 8", (long long)red);
    BMF_REQUIRE(lda >= red && lda % 4 == 0 && (b_frag || (ldft >= red && ldft % 4 == 0)), "bmf_xf_f32: lda/ldft must be >= red and multiples of 4");
    BMF_REQUIRE(!b_frag || a_tiled, "bmf_xf_f32: a factor in fragment order goes with a tiled A");
    BMF_REQUIRE(kp == 32 || kp == 64, "bmf_xf_f32: kp must be 32 or 64");
    BMF_REQUIRE(splits >= 1 && splits <= red / 8, "bmf_xf_f32: splits out of range");
    BMF_REQUIRE(slab_stride >= rows_pad * kp, "bmf_xf_f32: slab_stride too small");
    BMF_REQUIRE(bmf_aligned16(A) && bmf_aligned16(FT), "bmf_xf_f32: pointers must be 16-byte aligned");
    dim3 block(256);
    BMF_REQUIRE(!a_tiled || (red % 64 == 0 && bmf_aligned16(out)), "bmf_xf_f32_tiled: the reduction length must be a multiple of 64, out 16-byte aligned");
    if (red % 64 == 0 && bmf_aligned16(out)) {
        const int stages = (int)(red / 64);
        const int sps = (stages + splits - 1) / splits;
        const int tiles64 = (int)(rows_pad / 64);
        dim3 grid64((unsigned)(tiles64 * splits));
        if (kp == 32)
            BMF_LAUNCH(xf_f32_ring_kernel<1>, grid64, block, 0, s, A, lda, stages, sps, FT, ldft, out, slab_stride, tiles64, a_tiled, b_frag, stop);
        else
            BMF_LAUNCH(xf_f32_ring_kernel<2>, grid64, block, 0, s, A, lda, stages, sps, FT, ldft, out, slab_stride, tiles64, a_tiled, b_frag, stop);
        BMF_LAUNCH_CHECK();
        return BMF_OK;
    }
    const int n_row_tiles = (int)(rows_pad / 128);
    dim3 grid((unsigned)(n_row_tiles * splits));
    const int groups = (int)(red / 8);
    const int gps = (groups + splits - 1) / splits;
    if (kp == 32)
        BMF_LAUNCH(xf_f32_kernel<1>, grid, block, 0, s, A, lda, groups, gps, FT, ldft, out, slab_stride, n_row_tiles, stop);
    else
        BMF_LAUNCH(xf_f32_kernel<2>, grid, block, 0, s, A, lda, groups, gps, FT, ldft, out, slab_stride, n_row_tiles, stop);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_xf_f32(const float* A, int64_t rows_pad, int64_t lda, int64_t red, const float* FT, int64_t ldft, int kp,
                          float* out, int64_t slab_stride, int splits, void* stream) {
    return bmf_xf_f32_launch(A, rows_pad, lda, red, FT, ldft, kp, out, slab_stride, splits, 0, 0, nullptr, (hipStream_t)stream);
}

extern "C" int bmf_xf_f32_tiled(const float* Atiled, int64_t rows_pad, int64_t red, const float* Ffrag, int kp, float* out,
                                int64_t slab_stride, int splits, void* stream) {
    return bmf_xf_f32_launch(Atiled, rows_pad, red, red, Ffrag, 0, kp, out, slab_stride, splits, 1, 1, nullptr, (hipStream_t)stream);
}

/* out = A F and the residual sums of the same pass (see xf_f32_resid_ring_kernel): Atiled = bmf_tile_f32 of A (rows_pad x red),
 * Ffrag = bmf_frag_f32 of F (red x 32), Frf = bmf_frag_rows_bf16 of F, Grow = the second factor, rows_pad x 32 plain rows.
 * sums[0..1] are ADDED to. */
// Both fragment orders of a 32-column factor in ONE launch (bmf_frag_f32 -> frag, bmf_frag_rows_bf16 -> frag_bf), and block 0 zeroes
// the residual sums: what precedes the fused contraction + residual pass in the real-valued WNMF loop was three ~5-us launches.
__global__ __launch_bounds__(256) void frag_pair_kernel(const float* __restrict__ F, int64_t pieces, float* __restrict__ frag,
                                                         uint32_t* __restrict__ frag_bf, double* __restrict__ zero4,
                                                         const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    if (zero4 && blockIdx.x == 0 && threadIdx.x < 4) zero4[threadIdx.x] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < pieces; i += (int64_t)gridDim.x * 256) {
        const int lane = (int)(i & 63), r = lane & 31, h = lane >> 5;
        const int64_t g = i >> 6;
        {   // frag_f32_kernel with kp = 32 (NT = 1)
            const int u = (int)(g & 3), kh = (int)((g >> 2) & 1);
            const int64_t st = g >> 3;
            const float* src = F + (64 * st + 32 * kh + 8 * u + 4 * h) * 32 + r;
            *reinterpret_cast<f32x4*>(frag + i * 4) = f32x4{src[0], src[32], src[64], src[96]};
        }
        {   // frag_rows_bf16_kernel
            const int q = (int)(g & 3), ks = q >> 1, lo = q & 1;
            const int kh = (int)((g >> 2) & 1);
            const int64_t st = g >> 3;
            const float* src = F + (64 * st + 32 * kh + r) * 32 + 16 * ks + 8 * h;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
            const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            u32x4 o;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                uint16_t b0 = bf16_bits(v[2 * w]), b1 = bf16_bits(v[2 * w + 1]);
                if (lo) {
                    b0 = bf16_bits(v[2 * w] - bf16_to_f32(b0));
                    b1 = bf16_bits(v[2 * w + 1] - bf16_to_f32(b1));
                }
                o[w] = (unsigned)b0 | ((unsigned)b1 << 16);
            }
            *reinterpret_cast<u32x4*>(frag_bf + i * 4) = o;
        }
    }
}

int bmf_frag_pair_launch(const float* F, int64_t rows_pad, float* frag, uint32_t* frag_bf, double* zero4, const int32_t* stop, hipStream_t s) {
    BMF_REQUIRE(F && frag && frag_bf, "bmf_frag_pair: null pointer");
    BMF_REQUIRE(rows_pad > 0 && rows_pad % 64 == 0, "bmf_frag_pair: rows_pad must be a positive multiple of 64");
    const int64_t pieces = rows_pad * 32 / 4;
    const int64_t blocks = (pieces + 255) / 256;
    BMF_LAUNCH(frag_pair_kernel, dim3((unsigned)(blocks < 65535 ? blocks : 65535)), dim3(256), 0, s, F, pieces, frag, frag_bf, zero4, stop);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

int bmf_frag_rows_bf16_launch(const float* F, int64_t rows_pad, int kp, uint32_t* frag, const int32_t* stop, hipStream_t s) {
    BMF_REQUIRE(F && frag, "bmf_frag_rows_bf16: null pointer");
    BMF_REQUIRE(rows_pad > 0 && rows_pad % 64 == 0 && kp == 32, "bmf_frag_rows_bf16: rows_pad must be a positive multiple of 64, kp 32");
    BMF_REQUIRE(bmf_aligned16(F) && bmf_aligned16(frag), "bmf_frag_rows_bf16: pointers must be 16-byte aligned");
    const int64_t pieces = rows_pad * kp / 4;
    const int64_t blocks = (pieces + 255) / 256;
    BMF_LAUNCH(frag_rows_bf16_kernel, dim3((unsigned)(blocks < 65535 ? blocks : 65535)), dim3(256), 0, s, F, pieces, frag, stop);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_frag_rows_bf16(const float* F, int64_t rows_pad, int kp, uint32_t* frag, void* stream) {
    return bmf_frag_rows_bf16_launch(F, rows_pad, kp, frag, nullptr, (hipStream_t)stream);
}

// F3 != NULL: the contraction on the bf16 matrix instruction with three-way split operands (F3 = bmf_frag_bf16x3 of F; Ffrag is then unused)
int bmf_xf_f32_resid_launch(const float* Atiled, int64_t rows_pad, int64_t red, const float* Ffrag, const uint32_t* Frf, const float* Grow, int kp,
                            float* out, int64_t slab_stride, int splits, double* sums, const int32_t* stop, hipStream_t s, const uint32_t* F3) {
    BMF_REQUIRE(Atiled && (Ffrag || F3) && Frf && Grow && out && sums, "bmf_xf_f32_tiled_resid: null pointer");
    BMF_REQUIRE(kp == 32, "bmf_xf_f32_tiled_resid: kp must be 32 (the fused pass holds two products' operands in registers)");
    BMF_REQUIRE(rows_pad > 0 && rows_pad % 64 == 0 && red > 0 && red % 64 == 0, "bmf_xf_f32_tiled_resid: rows_pad and red must be positive multiples of 64");
    BMF_REQUIRE(splits >= 1 && splits <= red / 64 && slab_stride >= rows_pad * kp, "bmf_xf_f32_tiled_resid: bad splits / slab_stride");
    BMF_REQUIRE(bmf_aligned16(Atiled) && bmf_aligned16(Ffrag) && bmf_aligned16(F3) && bmf_aligned16(Frf) && bmf_aligned16(Grow) && bmf_aligned16(out),
                "bmf_xf_f32_tiled_resid: pointers must be 16-byte aligned");
    const int stages = (int)(red / 64);
    const int sps = (stages + splits - 1) / splits;
    const int tiles64 = (int)(rows_pad / 64);
    if (F3)
        BMF_LAUNCH(xf_f32_resid_ring_kernel<true>, dim3((unsigned)(tiles64 * splits)), dim3(256), 0, s, Atiled, stages, sps, Ffrag, F3, Frf, Grow, out, slab_stride,
                   tiles64, sums, stop);
    else
        BMF_LAUNCH(xf_f32_resid_ring_kernel<false>, dim3((unsigned)(tiles64 * splits)), dim3(256), 0, s, Atiled, stages, sps, Ffrag, F3, Frf, Grow, out, slab_stride,
                   tiles64, sums, stop);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_xf_f32_tiled_resid(const float* Atiled, int64_t rows_pad, int64_t red, const float* Ffrag, const uint32_t* Frf, const float* Grow,
                                      int kp, float* out, int64_t slab_stride, int splits, double* sums, void* stream) {
    return bmf_xf_f32_resid_launch(Atiled, rows_pad, red, Ffrag, Frf, Grow, kp, out, slab_stride, splits, sums, nullptr, (hipStream_t)stream, nullptr);
}

// ---- the bf16 x 3 forms (round 5): frag3 producer, contraction, contraction + residual sums ----
int bmf_frag_bf16x3_launch(const float* F, int64_t rows_pad, uint32_t* frag3, const int32_t* stop, hipStream_t s) {
    BMF_REQUIRE(F && frag3, "bmf_frag_bf16x3: null pointer");
    BMF_REQUIRE(rows_pad > 0 && rows_pad % 64 == 0 && bmf_aligned16(frag3), "bmf_frag_bf16x3: rows_pad must be a positive multiple of 64, frag3 16-byte aligned");
    const int64_t items = rows_pad / 64 * 4 * 64;   // (stage, half, k-step) x lanes
    const int64_t blocks = (items + 255) / 256;
    BMF_LAUNCH(frag_bf16x3_kernel, dim3((unsigned)(blocks < 65535 ? blocks : 65535)), dim3(256), 0, s, F, items, frag3, stop);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_frag_bf16x3(const float* F, int64_t rows_pad, uint32_t* frag3, void* stream) {
    return bmf_frag_bf16x3_launch(F, rows_pad, frag3, nullptr, (hipStream_t)stream);
}

int bmf_xf_f32_bf3_launch(const float* Atiled, int64_t rows_pad, int64_t red, const uint32_t* F3, float* out, int64_t slab_stride, int splits,
                          const int32_t* stop, hipStream_t s) {
    BMF_REQUIRE(Atiled && F3 && out, "bmf_xf_f32_tiled_bf3: null pointer");
    BMF_REQUIRE(rows_pad > 0 && rows_pad % 64 == 0 && red > 0 && red % 64 == 0, "bmf_xf_f32_tiled_bf3: rows_pad and red must be positive multiples of 64");
    BMF_REQUIRE(splits >= 1 && splits <= red / 64 && slab_stride >= rows_pad * 32, "bmf_xf_f32_tiled_bf3: bad splits / slab_stride");
    BMF_REQUIRE(bmf_aligned16(Atiled) && bmf_aligned16(F3) && bmf_aligned16(out), "bmf_xf_f32_tiled_bf3: pointers must be 16-byte aligned");
    const int stages = (int)(red / 64);
    const int sps = (stages + splits - 1) / splits;
    const int tiles64 = (int)(rows_pad / 64);
    BMF_LAUNCH(xf_f32_bf3_ring_kernel, dim3((unsigned)(tiles64 * splits)), dim3(256), 0, s, Atiled, stages, sps, F3, out, slab_stride, tiles64, stop);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_xf_f32_tiled_bf3(const float* Atiled, int64_t rows_pad, int64_t red, const uint32_t* F3, float* out, int64_t slab_stride, int splits,
                                    void* stream) {
    return bmf_xf_f32_bf3_launch(Atiled, rows_pad, red, F3, out, slab_stride, splits, nullptr, (hipStream_t)stream);
}

extern "C" int bmf_xf_f32_tiled_resid_bf3(const float* Atiled, int64_t rows_pad, int64_t red, const uint32_t* F3, const uint32_t* Frf, const float* Grow,
                                          float* out, int64_t slab_stride, int splits, double* sums, void* stream) {
    return bmf_xf_f32_resid_launch(Atiled, rows_pad, red, nullptr, Frf, Grow, 32, out, slab_stride, splits, sums, nullptr, (hipStream_t)stream, F3);
}

extern "C" int bmf_tile_f32(const float* X, int64_t rows_pad, int64_t lda, int64_t red, float* tiled, void* stream) {
    BMF_REQUIRE(X && tiled, "bmf_tile_f32: null pointer");
    BMF_REQUIRE(rows_pad > 0 && rows_pad % 64 == 0 && red > 0 && red % 64 == 0 && lda >= red && lda % 4 == 0,
                "bmf_tile_f32: rows_pad and red must be positive multiples of 64, lda >= red and a multiple of 4");
    BMF_REQUIRE(bmf_aligned16(X) && bmf_aligned16(tiled), "bmf_tile_f32: pointers must be 16-byte aligned");
    const int64_t chunks = rows_pad * (red / 4);
    const int64_t blocks = (chunks + 255) / 256;
    BMF_LAUNCH(tile_f32_kernel, dim3((unsigned)(blocks < 65535 ? blocks : 65535)), dim3(256), 0, (hipStream_t)stream, X, lda,
               (int)(red / 64), chunks, tiled);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}
