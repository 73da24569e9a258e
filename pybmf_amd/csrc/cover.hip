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

namespace {

constexpr int CH = 256;  // words per column chunk: 64 columns-of-bits x 256 words x 4 B = 64 KiB of LDS

__global__ __launch_bounds__(256) void cover_kernel(const uint32_t* __restrict__ X, int64_t ldx, int64_t words,
                                                     const uint64_t* __restrict__ rowbits,
                                                     const uint32_t* __restrict__ colbits, int64_t ldcb, int kp,
                                                     int64_t rows_pad, int rows_per_block,
                                                     unsigned long long* __restrict__ counts,
                                                     const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    __shared__ __attribute__((aligned(16))) uint32_t vt[BMF_MAX_KP][CH];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t w0 = (int64_t)blockIdx.x * CH;
    const int nw = (int)min((int64_t)CH, words - w0);

    // LDS fill: 64 bit-columns x 1 KiB, 16-byte pieces, 4 independent loads in flight per thread
    {
        constexpr int PIECES = BMF_MAX_KP * CH / 4;  // 4096 pieces of 16 B
#pragma unroll
        for (int base = 0; base < PIECES; base += 4 * 256) {
            u32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = base + q * 256 + threadIdx.x;
                const int l = p / (CH / 4), pw = (p % (CH / 4)) * 4;
                v[q] = (l < kp && pw < nw) ? *reinterpret_cast<const u32x4*>(colbits + (int64_t)l * ldcb + w0 + pw)
                                           : u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = base + q * 256 + threadIdx.x;
                *reinterpret_cast<u32x4*>(&vt[p / (CH / 4)][(p % (CH / 4)) * 4]) = v[q];
            }
        }
    }
    __syncthreads();

    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = min(r0 + rows_per_block, rows_pad);
    const bool lane_on = 4 * lane < nw;  // words is a multiple of 4
    unsigned tp = 0, fp = 0;
    // rows are taken in batches of RB per wave; the X words and k-bit words of the NEXT batch are requested before
    // the current batch is processed, so the (dependent) set-bit walk overlaps the global-load latency
    constexpr int RB = 4;
    const uint32_t* xp = X + w0 + 4 * lane;
    auto load_batch = [&](int64_t base, unsigned long long (&u)[RB], u32x4 (&x)[RB]) {
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int64_t i = base + 4 * b;  // waves interleave rows: wave w takes rows r0 + w, r0 + w + 4, ...
            const bool in = i < r1;
            u[b] = in ? rowbits[i] : 0ull;
            x[b] = (in && lane_on) ? *reinterpret_cast<const u32x4*>(xp + i * ldx) : u32x4{0u, 0u, 0u, 0u};
        }
    };
    unsigned long long u_cur[RB], u_nxt[RB];
    u32x4 x_cur[RB], x_nxt[RB];
    load_batch(r0 + wave, u_cur, x_cur);
    for (int64_t base = r0 + wave; base < r1; base += 4 * RB) {
        load_batch(base + 4 * RB, u_nxt, x_nxt);
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            unsigned ulo = __builtin_amdgcn_readfirstlane((unsigned)u_cur[b]);
            unsigned uhi = __builtin_amdgcn_readfirstlane((unsigned)(u_cur[b] >> 32));
            if ((ulo | uhi) == 0u) continue;
            u32x4 pd = {0u, 0u, 0u, 0u};
            while (ulo) {
                const int l = __builtin_ctz(ulo);
                ulo &= ulo - 1;
                pd |= *reinterpret_cast<const u32x4*>(&vt[l][4 * lane]);
            }
            while (uhi) {
                const int l = 32 + __builtin_ctz(uhi);
                uhi &= uhi - 1;
                pd |= *reinterpret_cast<const u32x4*>(&vt[l][4 * lane]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                tp += __popc(x_cur[b][q] & pd[q]);
                fp += __popc(~x_cur[b][q] & pd[q]);
            }
        }
#pragma unroll
        for (int b = 0; b < RB; ++b) { u_cur[b] = u_nxt[b]; x_cur[b] = x_nxt[b]; }
    }
    tp = wave_sum(tp);
    fp = wave_sum(fp);
    if (lane == 0) {
        if (tp) atomicAdd(&counts[0], (unsigned long long)tp);
        if (fp) atomicAdd(&counts[1], (unsigned long long)fp);
    }
}

}  // namespace

int bmf_cover_launch(const uint32_t* Xbits, int64_t rows_pad, int64_t ldx, int64_t words, const uint64_t* rowbits,
                     const uint32_t* colbits, int64_t ldcb, int kp, unsigned long long* counts, const int32_t* stop,
                     hipStream_t s) {
    const unsigned chunks = (unsigned)((words + CH - 1) / CH);
    // two resident blocks per CU (64 KiB of LDS each) in a single round: the 64 KiB LDS fill is paid once per block
    int64_t groups = rows_pad / 64;
    const int64_t want = (512 + chunks - 1) / chunks;
    if (groups > want) groups = want;
    if (groups < 1) groups = 1;
    const int rows_per_block = (int)((rows_pad + groups - 1) / groups);
    groups = (rows_pad + rows_per_block - 1) / rows_per_block;
    dim3 grid(chunks, (unsigned)groups), block(256);
    hipLaunchKernelGGL(cover_kernel, grid, block, 0, s, Xbits, ldx, words, rowbits, colbits, ldcb, kp, rows_pad,
                       rows_per_block, counts, stop);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_cover_count(const uint32_t* Xbits, int64_t rows_pad, int64_t ldx, int64_t words,
                               const uint64_t* rowbits, const uint32_t* colbits, int64_t ldcb, int kp,
                               unsigned long long* counts, const int32_t* stop, void* stream) {
    BMF_REQUIRE(Xbits && rowbits && colbits && counts, "bmf_cover_count: null pointer");
    BMF_REQUIRE(rows_pad > 0, "bmf_cover_count: rows_pad must be positive");
    BMF_REQUIRE(words > 0 && words % 4 == 0 && ldx >= words && ldx % 4 == 0, "bmf_cover_count: words/ldx must be multiples of 4, ldx >= words");
    BMF_REQUIRE(ldcb >= words && ldcb % 4 == 0, "bmf_cover_count: ldcb must be >= words and a multiple of 4");
    BMF_REQUIRE(bmf_aligned16(colbits), "bmf_cover_count: colbits must be 16-byte aligned");
    BMF_REQUIRE(kp >= 1 && kp <= BMF_MAX_KP, "bmf_cover_count: kp must be 1..64");
    BMF_REQUIRE(bmf_aligned16(Xbits), "bmf_cover_count: Xbits must be 16-byte aligned");
    return bmf_cover_launch(Xbits, rows_pad, ldx, words, rowbits, colbits, ldcb, kp, counts, stop, (hipStream_t)stream);
}
