// K6: Boolean cover count -- TP / FP of the Boolean product of the thresholded factors against X, all in bits.
//
//   pd = min(1, (U > u) @ (V > v)^T)        PyBMF/utils/common.py:110-151 (get_prediction_with_threshold)
//   TP = sum(X o pd), FP = sum(max(pd - X, 0))  PyBMF/utils/metrics.py:56-68   (FN, TN follow from sum(X) and m*n)
//
// Row i of pd is the OR of the bit-columns V_l (one n-bit vector per factor l) over the factors l set in
// ubits[i].  A block keeps a chunk of CH words of all k bit-columns in LDS; a wave takes one X row at a time,
// walks the set bits of that row's k-bit word on the scalar unit (the word is wave-uniform) and ORs the
// selected bit-columns 128 bits per lane; TP/FP are popcounts against the X bits.  Integer arithmetic: exact.
// Cost is data dependent: popcount(ubits[i]) LDS reads per row-chunk instead of k.
#include "common.h"

#include <algorithm>
#include <cstdlib>

namespace {

constexpr int RB = 8, NW = 8;  // rows per batch, waves per block
constexpr int CH = 256;  // words per column chunk: 64 columns-of-bits x 256 words x 4 B = 64 KiB of LDS

__global__ __launch_bounds__(512) void cover_kernel(const uint32_t* __restrict__ X, int64_t ldx, int64_t words,
                                                     const uint64_t* __restrict__ rowbits,
                                                     const uint32_t* __restrict__ colbits, int64_t ldcb, int kp,
                                                     int64_t rows_pad, int rows_per_block, int y_big, int rows_big,
                                                     unsigned long long* __restrict__ counts,
                                                     const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    __shared__ __attribute__((aligned(16))) uint32_t vt[BMF_MAX_KP + 1][CH];  // row 64 stays zero (padding of the 4-way walk)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t w0 = (int64_t)blockIdx.x * CH;
    const int nw = (int)min((int64_t)CH, words - w0);

    // LDS fill: 64 bit-columns x 1 KiB, 16-byte pieces, 4 independent loads in flight per thread
    {
        constexpr int PIECES = BMF_MAX_KP * CH / 4;  // 4096 pieces of 16 B
#pragma unroll
        for (int base = 0; base < PIECES; base += 4 * 512) {
            u32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = base + q * 512 + threadIdx.x;
                const int l = p / (CH / 4), pw = (p % (CH / 4)) * 4;
                v[q] = (l < kp && pw < nw) ? *reinterpret_cast<const u32x4*>(colbits + (int64_t)l * ldcb + w0 + pw)
                                           : u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = base + q * 512 + threadIdx.x;
                *reinterpret_cast<u32x4*>(&vt[p / (CH / 4)][(p % (CH / 4)) * 4]) = v[q];
            }
        }
        if (threadIdx.x < CH / 4) *reinterpret_cast<u32x4*>(&vt[BMF_MAX_KP][threadIdx.x * 4]) = u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();

    // row groups of two sizes: the first y_big groups (the workgroups dispatched first, which the SIMDs' age-ordered arbitration
    // favours over their later partners -- see build_plan_i8 in xf_bits_i8.hip) take rows_big rows each, the rest rows_per_block
    const int by = (int)blockIdx.y;
    const int64_t r0 = by < y_big ? (int64_t)by * rows_big : (int64_t)y_big * rows_big + (int64_t)(by - y_big) * rows_per_block;
    const int64_t r1 = min(r0 + (by < y_big ? rows_big : rows_per_block), rows_pad);
    if (r0 >= r1) return;
    const bool lane_on = 4 * lane < nw;  // words is a multiple of 4
    unsigned tp = 0, fp = 0;
    // rows are taken in batches of RB per wave; the X words and k-bit words of the NEXT batch are requested before
    // the current batch is processed, so the (dependent) set-bit walk overlaps the global-load latency
    // The kernel is instruction-issue bound (one row x 8192 columns per ~100 wave instructions), so the loop is kept
    // lean: 32-bit offsets, no bounds checks (the launcher makes every block's row count a multiple of 64 = 8 waves
    // x 8 rows, and all rows < rows_pad exist), X rows prefetched one batch ahead.  The k-bit words of 64 rows come
    // from ONE vector load (lane j holds the word of the wave's j-th row) and are broadcast with v_readlane.
    const int nrows_wave = (int)((r1 - r0) / NW);  // rows this wave owns: r0 + wave + NW*j
    const uint32_t* xw = X + (r0 + wave) * ldx + w0 + (lane_on ? 4 * lane : 0);
    const unsigned row_step = (unsigned)(NW * ldx);  // words between two rows of this wave
    for (int g = 0; g < nrows_wave; g += 64) {
        const int jl = min(g + lane, nrows_wave - 1);
        const unsigned long long uvec = rowbits[r0 + wave + (int64_t)NW * jl];
        const unsigned uv_lo = (g + lane < nrows_wave) ? (unsigned)uvec : 0u;
        const unsigned uv_hi = (g + lane < nrows_wave) ? (unsigned)(uvec >> 32) : 0u;
        const int nj = min(64, nrows_wave - g);  // multiple of RB
        const uint32_t* xg = xw + (size_t)g * row_step;
        u32x4 x_cur[RB], x_nxt[RB];
#pragma unroll
        for (int b = 0; b < RB; ++b) x_cur[b] = *reinterpret_cast<const u32x4*>(xg + (size_t)b * row_step);
        for (int j0 = 0; j0 < nj; j0 += RB) {
            const int jn = (j0 + RB < nj) ? j0 + RB : j0;  // last batch re-reads itself (cheap, keeps the loop uniform)
#pragma unroll
            for (int b = 0; b < RB; ++b) x_nxt[b] = *reinterpret_cast<const u32x4*>(xg + (size_t)(jn + b) * row_step);
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                unsigned ulo = __builtin_amdgcn_readlane(uv_lo, j0 + b);
                unsigned uhi = __builtin_amdgcn_readlane(uv_hi, j0 + b);
                if ((ulo | uhi) == 0u) continue;
                u32x4 pd = {0u, 0u, 0u, 0u};
                // two set bits per trip from each 32-bit half (row 64 = zeros pads): independent LDS reads in flight
                while (ulo | uhi) {
                    const int l0 = ulo ? __builtin_ctz(ulo) : BMF_MAX_KP;
                    ulo &= ulo - 1;
                    const int l1 = ulo ? __builtin_ctz(ulo) : BMF_MAX_KP;
                    ulo &= ulo - 1;
                    const int l2 = uhi ? 32 + __builtin_ctz(uhi) : BMF_MAX_KP;
                    uhi &= uhi - 1;
                    const int l3 = uhi ? 32 + __builtin_ctz(uhi) : BMF_MAX_KP;
                    uhi &= uhi - 1;
                    const u32x4 v0 = *reinterpret_cast<const u32x4*>(&vt[l0][4 * lane]);
                    const u32x4 v1 = *reinterpret_cast<const u32x4*>(&vt[l1][4 * lane]);
                    const u32x4 v2 = *reinterpret_cast<const u32x4*>(&vt[l2][4 * lane]);
                    const u32x4 v3 = *reinterpret_cast<const u32x4*>(&vt[l3][4 * lane]);
                    pd |= (v0 | v1) | (v2 | v3);
                }
                if (!lane_on) pd = u32x4{0u, 0u, 0u, 0u};  // lanes beyond the chunk re-read lane 0's words: mask them out
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    tp += __popc(x_cur[b][q] & pd[q]);
                    fp += __popc(~x_cur[b][q] & pd[q]);
                }
            }
#pragma unroll
            for (int b = 0; b < RB; ++b) x_cur[b] = x_nxt[b];
        }
    }
    // one atomic pair per BLOCK: device-scope atomics on one address serialise at ~12 ns each, so a pair per wave
    // (8192 of them) was a fixed ~90 us tail
    __shared__ unsigned red[2][NW];
    tp = wave_sum(tp);
    fp = wave_sum(fp);
    if (lane == 0) { red[0][wave] = tp; red[1][wave] = fp; }
    __syncthreads();
    if (threadIdx.x < 2) {
        unsigned long long t = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[threadIdx.x][w];
        if (t) atomicAdd(&counts[threadIdx.x], t);
    }
}

// The same count with a whole row chunk of up to 640 words per wave-row (round 4).  cover_kernel keeps a 256-word chunk of every
// bit-column in LDS (64 KiB) and a lane owns 4 of its words; at the headline shape a row is 640 words = 2.5 chunks, so the
// blocks of the third chunk column run with half their lanes off, and the set-bit walk (scalar bit extraction + one LDS read per
// set factor bit) is paid three times per row.  Here ONE workgroup per CU holds up to 640 words of all kp bit-columns (the whole
// 160 KiB of LDS; the block reduction at the end reuses it), a lane owns 4 N4 + TW words of the chunk in N4 + 1 segments (segment
// s: words 256 s + 4 lane ..., the tail segment 2 or 4 words per lane), so every X load and every LDS read is a full-width wave
// instruction, and a row's k-bit word is walked ONCE: ~40 % fewer wave instructions per row at 640 words.  16 waves per workgroup,
// two rows per wave in flight ahead of the two being counted.
template <int N4, int TW>
__global__ __launch_bounds__(1024) void cover_wide_kernel(const uint32_t* __restrict__ X, int64_t ldx, int64_t words, int chunk_words,
                                                           const uint64_t* __restrict__ rowbits, const uint32_t* __restrict__ colbits,
                                                           int64_t ldcb, int kp, int64_t rows_pad, int rows_per_block,
                                                           unsigned long long* __restrict__ counts, const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    constexpr int CW = 256 * N4 + 64 * TW;   // words of a bit-column held in LDS
#ifndef BMF_COVER_RBW
#define BMF_COVER_RBW 2
#endif
    constexpr int NWV = 16, RBW = BMF_COVER_RBW;   // waves per block, rows per batch
    constexpr int NS = N4 + (TW ? 1 : 0);    // segments
    static_assert(BMF_MAX_KP * CW * 4 <= 160 * 1024 && (TW == 0 || TW == 2 || TW == 4) && NS >= 1, "chunk must fit the LDS");
    __shared__ __attribute__((aligned(16))) uint32_t vt[BMF_MAX_KP * CW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t w0 = (int64_t)blockIdx.x * chunk_words;
    const int nw = (int)min((int64_t)min(CW, chunk_words), words - w0);   // multiple of 4

    // LDS fill: kp bit-columns x CW words, 16-byte pieces (columns >= kp and words >= nw read as zero)
    {
        constexpr int PPC = CW / 4;                   // pieces per column
        constexpr int PIECES = BMF_MAX_KP * PPC;
        for (int base = 0; base < PIECES; base += 4 * 1024) {
            u32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = base + q * 1024 + (int)threadIdx.x;
                const int l = p / PPC, pw = (p % PPC) * 4;
                v[q] = (p < PIECES && l < kp && pw < nw) ? *reinterpret_cast<const u32x4*>(colbits + (int64_t)l * ldcb + w0 + pw) : u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = base + q * 1024 + (int)threadIdx.x;
                if (p < PIECES) *reinterpret_cast<u32x4*>(&vt[4 * p]) = v[q];
            }
        }
    }
    __syncthreads();

    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = min(r0 + rows_per_block, rows_pad);
    unsigned tp = 0, pp = 0;   // |X & P|, |P|
    if (r0 < r1) {
        // lane's word offset inside the chunk, per segment, and whether those words exist (off lanes re-read words 0.. and are masked)
        int soff[NS];
        bool son[NS];
#pragma unroll
        for (int sg = 0; sg < NS; ++sg) {
            const int wpl = (sg < N4) ? 4 : TW;
            const int o = 256 * sg + wpl * lane;
            son[sg] = o < nw;
            soff[sg] = son[sg] ? o : 0;
        }
        const int nrows_wave = (int)((r1 - r0) / NWV);   // rows this wave owns: r0 + wave + NWV * j (a multiple of RBW: rows_per_block % 64 == 0)
        const uint32_t* xw = X + (r0 + wave) * ldx + w0;
        const unsigned row_step = (unsigned)(NWV * ldx);
        for (int g = 0; g < nrows_wave; g += 64) {
            const int jl = min(g + lane, nrows_wave - 1);
            const unsigned long long uvec = rowbits[r0 + wave + (int64_t)NWV * jl];
            const unsigned uv_lo = (g + lane < nrows_wave) ? (unsigned)uvec : 0u;
            const unsigned uv_hi = (g + lane < nrows_wave) ? (unsigned)(uvec >> 32) : 0u;
            const int nj = min(64, nrows_wave - g);
            const uint32_t* xg = xw + (size_t)g * row_step;
            u32x4 x_cur[RBW][NS], x_nxt[RBW][NS];   // (the tail segment uses .xy when TW == 2)
            auto load_rows = [&](int j, u32x4 (&dst)[RBW][NS]) {
#pragma unroll
                for (int b = 0; b < RBW; ++b)
#pragma unroll
                    for (int sg = 0; sg < NS; ++sg) {
                        // (non-temporal: every word of X is used once per launch and should not displace the factor panels in L2)
                        const uint32_t* p_ = xg + (size_t)(j + b) * row_step + soff[sg];
#ifdef BMF_COVER_PLAIN_LOADS
                        if (sg < N4 || TW == 4) dst[b][sg] = *reinterpret_cast<const u32x4*>(p_);
                        else {
                            const u32x2 t = *reinterpret_cast<const u32x2*>(p_);
                            dst[b][sg] = u32x4{t[0], t[1], 0u, 0u};
                        }
#else
                        if (sg < N4 || TW == 4) dst[b][sg] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p_));
                        else {
                            const u32x2 t = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p_));
                            dst[b][sg] = u32x4{t[0], t[1], 0u, 0u};
                        }
#endif
                    }
            };
            load_rows(0, x_cur);
            for (int j0 = 0; j0 < nj; j0 += RBW) {
                const int jn = (j0 + RBW < nj) ? j0 + RBW : j0;   // the last batch re-reads itself (keeps the loop uniform)
                load_rows(jn, x_nxt);
#pragma unroll
                for (int b = 0; b < RBW; ++b) {
                    const unsigned ulo = __builtin_amdgcn_readlane(uv_lo, j0 + b);
                    const unsigned uhi = __builtin_amdgcn_readlane(uv_hi, j0 + b);
                    unsigned long long u = ((unsigned long long)uhi << 32) | ulo;
                    if (u == 0ull) continue;
                    u32x4 pd[NS];
#pragma unroll
                    for (int sg = 0; sg < NS; ++sg) pd[sg] = u32x4{0u, 0u, 0u, 0u};
                    // two set bits per trip; a missing second bit re-reads the first column (pd | v = pd then)
                    while (u) {
                        const int l0 = __builtin_ctzll(u);
                        u &= u - 1;
                        const int l1 = u ? __builtin_ctzll(u) : l0;
                        u &= u - 1;
                        const uint32_t* c0 = vt + l0 * CW;
                        const uint32_t* c1 = vt + l1 * CW;
#pragma unroll
                        for (int sg = 0; sg < NS; ++sg) {
                            if (sg < N4 || TW == 4) {
                                const u32x4 v0 = *reinterpret_cast<const u32x4*>(c0 + soff[sg]);
                                const u32x4 v1 = *reinterpret_cast<const u32x4*>(c1 + soff[sg]);
                                pd[sg] |= v0 | v1;
                            } else {
                                const u32x2 v0 = *reinterpret_cast<const u32x2*>(c0 + soff[sg]);
                                const u32x2 v1 = *reinterpret_cast<const u32x2*>(c1 + soff[sg]);
                                pd[sg][0] |= v0[0] | v1[0];
                                pd[sg][1] |= v0[1] | v1[1];
                            }
                        }
                    }
#pragma unroll
                    for (int sg = 0; sg < NS; ++sg) {
                        const int nq = (sg < N4 || TW == 4) ? 4 : 2;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (q < nq) {
                                const unsigned pw_ = son[sg] ? pd[sg][q] : 0u;   // lanes beyond the chunk re-read words 0..: mask them out
                                tp += __popc(x_cur[b][sg][q] & pw_);
                                pp += __popc(pw_);
                            }
                    }
                }
#pragma unroll
                for (int b = 0; b < RBW; ++b)
#pragma unroll
                    for (int sg = 0; sg < NS; ++sg) x_cur[b][sg] = x_nxt[b][sg];
            }
        }
    }
    // one atomic pair per block (see cover_kernel); the bit-columns are dead: their LDS is the reduction buffer
    tp = wave_sum(tp);
    pp = wave_sum(pp);
    __syncthreads();
    if (lane == 0) { vt[wave] = tp; vt[NWV + wave] = pp; }
    __syncthreads();
    if (threadIdx.x < 2) {
        unsigned long long t = 0, a_ = 0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { t += vt[w]; a_ += vt[NWV + w]; }
        const unsigned long long val = threadIdx.x == 0 ? t : a_ - t;   // TP, FP = |P| - TP
        if (val) atomicAdd(&counts[threadIdx.x], val);
    }
}

// Materialise the Boolean product as bits: out[i][w] = OR_{l in rowbits[i]} colbits[l][w]  (the X_pd the reference builds
// with csr @ csr, PyBMF/utils/common.py:147-149).  Thread per (row, word); used once at the end of fit(), not in the loop.
__global__ __launch_bounds__(256) void product_bits_kernel(const uint64_t* __restrict__ rowbits, int64_t rows,
                                                            const uint32_t* __restrict__ colbits, int64_t ldcb, int64_t words,
                                                            uint32_t* __restrict__ out, int64_t ldo) {
    const int64_t total = rows * words;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t row = idx / words, w = idx - row * words;
        unsigned long long u = rowbits[row];
        unsigned pd = 0u;
        while (u) {
            const int l = __builtin_ctzll(u);
            u &= u - 1;
            pd |= colbits[(int64_t)l * ldcb + w];
        }
        out[row * ldo + w] = pd;
    }
}

// per-row confusion counts of two bit matrices: tp[r] = |G_r & P_r|, fp[r] = |~G_r & P_r| (one wave per row)
__global__ __launch_bounds__(256) void confusion_rows_kernel(const uint32_t* __restrict__ G, int64_t ldg,
                                                              const uint32_t* __restrict__ P, int64_t ldp, int64_t rows,
                                                              int64_t words, uint32_t* __restrict__ tp,
                                                              uint32_t* __restrict__ fp) {
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += nwaves) {
        unsigned a = 0u, b = 0u;
        for (int64_t w = lane; w < words; w += 64) {
            const uint32_t g = G[r * ldg + w], p = P[r * ldp + w];
            a += __popc(g & p);
            b += __popc(~g & p);
        }
        a = wave_sum(a);
        b = wave_sum(b);
        if (lane == 0) {
            tp[r] = a;
            fp[r] = b;
        }
    }
}

}  // namespace

extern "C" int bmf_confusion_rows(const uint32_t* Gbits, int64_t ldg, const uint32_t* Pbits, int64_t ldp, int64_t rows,
                                  int64_t words, uint32_t* tp, uint32_t* fp, void* stream) {
    BMF_REQUIRE(Gbits && Pbits && tp && fp, "bmf_confusion_rows: null pointer");
    BMF_REQUIRE(rows >= 1 && words >= 1 && ldg >= words && ldp >= words, "bmf_confusion_rows: bad shape");
    const int64_t blocks = (rows + 3) / 4;
    BMF_LAUNCH(confusion_rows_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, Gbits, ldg,
               Pbits, ldp, rows, words, tp, fp);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

namespace {
template <int N4, int TW>
void launch_cover_wide(dim3 grid, hipStream_t s, const uint32_t* Xbits, int64_t ldx, int64_t words, int chunk_words, const uint64_t* rowbits,
                       const uint32_t* colbits, int64_t ldcb, int kp, int64_t rows_pad, int rows_per_block, unsigned long long* counts,
                       const int32_t* stop) {
    BMF_LAUNCH((cover_wide_kernel<N4, TW>), grid, dim3(1024), 0, s, Xbits, ldx, words, chunk_words, rowbits, colbits, ldcb, kp, rows_pad,
               rows_per_block, counts, stop);
}
}  // namespace

int bmf_cover_launch(const uint32_t* Xbits, int64_t rows_pad, int64_t ldx, int64_t words, const uint64_t* rowbits,
                     const uint32_t* colbits, int64_t ldcb, int kp, unsigned long long* counts, const int32_t* stop,
                     hipStream_t s) {
    // Whole-row chunks (cover_wide_kernel) wherever a block's rows fill its 16 waves; the 256-word kernel of rounds 1-3 below that
    if (words >= 128) {
        const int chunks_w = (int)((words + 639) / 640);
        int cw = (int)(((words + chunks_w - 1) / chunks_w + 127) / 128 * 128);   // <= 640, a multiple of 128
        const int n4 = cw / 256 > 2 ? 2 : cw / 256, rem = cw - 256 * n4;
        const int tw = rem == 0 ? 0 : (rem <= 128 ? 2 : 4);
        const int cus = bmf_cu_count_current();
        int64_t groups = cus / chunks_w > 0 ? cus / chunks_w : 1;   // one workgroup per CU, one round
        const int64_t units = rows_pad / 64;
        if (groups > units) groups = units;
        const int rows_per_block = (int)(((units + groups - 1) / groups) * 64);
        groups = (rows_pad + rows_per_block - 1) / rows_per_block;
        dim3 grid((unsigned)chunks_w, (unsigned)groups);
#define BMF_CW_CASE(N4_, TW_) \
    if (n4 == N4_ && tw == TW_) launch_cover_wide<N4_, TW_>(grid, s, Xbits, ldx, words, cw, rowbits, colbits, ldcb, kp, rows_pad, rows_per_block, counts, stop);
        BMF_CW_CASE(0, 2) BMF_CW_CASE(0, 4) BMF_CW_CASE(1, 0) BMF_CW_CASE(1, 2) BMF_CW_CASE(1, 4) BMF_CW_CASE(2, 0) BMF_CW_CASE(2, 2)
#undef BMF_CW_CASE
        BMF_LAUNCH_CHECK();
        return BMF_OK;
    }
    const unsigned chunks = (unsigned)((words + CH - 1) / CH);
    // at most 512 blocks (two per CU, 66.5 KiB of LDS each, one round: a 513th block would run alone); every block
    // owns a multiple of 64 rows (8 waves x batches of 8 rows), so the kernel needs no bounds checks
    int64_t groups = 512 / chunks > 0 ? 512 / chunks : 1;
    int64_t units = rows_pad / 64;  // rows_pad is a multiple of 64
    if (groups > units) groups = units;
    const double share = 0.63;   // (swept with the GEMM's share in round 3: profiles/r03_i8_share_sweep.txt)
    int y_big = 0, rows_big = 0, rows_per_block;
    if (share > 0.5 && groups >= 8 && units >= 4 * groups) {
        // the first half of the blocks in dispatch order (x fastest) are the first workgroups of their CUs
        y_big = (int)(groups / 2);
        const int64_t u_big = (int64_t)(share * 2.0 * (double)units / (double)groups + 0.999);
        rows_big = (int)(u_big * 64);
        const int64_t rest = units - std::min<int64_t>(units, (int64_t)y_big * u_big);
        const int64_t n_small = groups - y_big;
        rows_per_block = (int)(std::max<int64_t>(1, (rest + n_small - 1) / n_small) * 64);
    } else {
        rows_per_block = (int)(((units + groups - 1) / groups) * 64);
        groups = (rows_pad + rows_per_block - 1) / rows_per_block;
    }
    dim3 grid(chunks, (unsigned)groups), block(512);
    BMF_LAUNCH(cover_kernel, grid, block, 0, s, Xbits, ldx, words, rowbits, colbits, ldcb, kp, rows_pad,
                       rows_per_block, y_big, rows_big, counts, stop);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_cover_count(const uint32_t* Xbits, int64_t rows_pad, int64_t ldx, int64_t words,
                               const uint64_t* rowbits, const uint32_t* colbits, int64_t ldcb, int kp,
                               unsigned long long* counts, const int32_t* stop, void* stream) {
    BMF_REQUIRE(Xbits && rowbits && colbits && counts, "bmf_cover_count: null pointer");
    BMF_REQUIRE(rows_pad > 0 && rows_pad % 64 == 0, "bmf_cover_count: rows_pad must be a positive multiple of 64");
    BMF_REQUIRE(words > 0 && words % 4 == 0 && ldx >= words && ldx % 4 == 0, "bmf_cover_count: words/ldx must be multiples of 4, ldx >= words");
    BMF_REQUIRE(ldcb >= words && ldcb % 4 == 0, "bmf_cover_count: ldcb must be >= words and a multiple of 4");
    BMF_REQUIRE(bmf_aligned16(colbits), "bmf_cover_count: colbits must be 16-byte aligned");
    BMF_REQUIRE(kp >= 1 && kp <= BMF_MAX_KP, "bmf_cover_count: kp must be 1..64");
    BMF_REQUIRE(bmf_aligned16(Xbits), "bmf_cover_count: Xbits must be 16-byte aligned");
    return bmf_cover_launch(Xbits, rows_pad, ldx, words, rowbits, colbits, ldcb, kp, counts, stop, (hipStream_t)stream);
}

extern "C" int bmf_boolean_product_bits(const uint64_t* rowbits, int64_t rows, const uint32_t* colbits, int64_t ldcb, int kp,
                                        int64_t words, uint32_t* out, int64_t ldo, void* stream) {
    BMF_REQUIRE(rowbits && colbits && out, "bmf_boolean_product_bits: null pointer");
    BMF_REQUIRE(rows > 0 && words > 0 && ldcb >= words && ldo >= words, "bmf_boolean_product_bits: bad shape");
    BMF_REQUIRE(kp >= 1 && kp <= BMF_MAX_KP, "bmf_boolean_product_bits: kp must be 1..64");
    const int64_t blocks = (rows * words + 255) / 256;
    BMF_LAUNCH(product_bits_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream,
                       rowbits, rows, colbits, ldcb, words, out, ldo);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}
