// K4: fused multiplicative-update epilogue for one factor F (rows x k), lane = factor column.
//
//   PENALTY:  den = F G + 2 reg F^3 + reg F ; den==0 -> eps ; Fn = F o ((num + 3 reg F^2) / den) ; Fn==0 -> eps
//             PyBMF/models/BinaryMFPenalty.py:136-163 (update_U / update_V), with the m x n product re-associated:
//             multiply(W, U @ V.T) @ V  ==  U (V^T V) for W = 'full'.
//   WNMF   :  den = F G ; den==0 -> eps ; Fn = F o (num / den)                    PyBMF/models/WNMF.py:96-109
//   PREPARE:  Fn = F (iteration-0 bookkeeping)
//
// and, from the new factor, everything the following kernels need without another pass over it:
//   * its bf16 panel (T addends, position-permuted) for the next bits GEMM,
//   * the thresholded Boolean factor, both as one k-bit word per row (ballot) and as bit-columns
//     (PyBMF/utils/common.py:64-79 binarize, strict '>'),
//   * per-block fp64 partials of sum((Fn^2 - Fn)^2) (BinaryMFPenalty.py:182-186) and of sum(Fn o num), the
//     <X, U V^T> term of the trace form of rec_error (BinaryMFPenalty.py:175-179).
//
// Precision: the factor itself is kept in fp64 (F64, the master copy) and the element-wise part of the update -- the
// penalty terms, the ratio, the clamps, the regulariser sum -- is evaluated in fp64.  Only the two contractions feeding
// it (num from the bits GEMM, F G here) are fp32-accurate, which is enough: they are sums of O(1) positive terms.  This
// matters once lambda is large (the reference's default schedule reaches 1e10): entries converge to 1 like 1 - (2/3)^t
// and reg_error = lambda/2 * sum (u^2-u)^2 must keep falling below `tol` -- fp32 entries cannot get closer to 1 than
// 6e-8.  An fp32 shadow copy (F) is written alongside for the Gram / residual / MAE kernels.
#include "common.h"

#ifdef BMF_EPI_STAMP   // diagnostic build only (scripts/build_flavour.sh NAME -DBMF_EPI_STAMP epilogue.hip): phase stamps of wave 0 of workgroup 0
__device__ unsigned long long bmf_epi_stamps[16];
extern "C" int bmf_debug_epi_stamps(unsigned long long* out_host) {
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(bmf_epi_stamps), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -2;
}
#define BMF_EPI_STAMP_AT(i)                                                                              \
    do {                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                    \
        if (blockIdx.x == 0 && threadIdx.x == 0) bmf_epi_stamps[i] = __builtin_amdgcn_s_memtime();      \
        __builtin_amdgcn_sched_barrier(0);                                                               \
    } while (0)
#define BMF_EPI_STAMP_NW(i)   /* no wait: the time the wave REACHES this point */                     \
    do {                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if (blockIdx.x == 0 && threadIdx.x == 0) bmf_epi_stamps[i] = __builtin_amdgcn_s_memtime();      \
        __builtin_amdgcn_sched_barrier(0);                                                               \
    } while (0)
#else
#define BMF_EPI_STAMP_AT(i) do { } while (0)
#define BMF_EPI_STAMP_NW(i) do { } while (0)
#endif

namespace {

constexpr int LDS_ROW = 264;  // bytes per (term, column) row of the staged panel tile: 256 + 8 pad (2-way max on b16 writes)

// One block = 128 rows (one panel permutation block) = 4 waves x 32 rows.
//
// F G (the re-associated denominator, 32 x kp per wave) runs on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): lane
// (c, h) feeds A = F[row c][32h + s] from the fp32 shadow and B = G[32h + s][32nt + c], 32 steps per 32-column tile.
// The result arrives with the C/D layout -- lane (c, h) owns column 32nt + c of the 16 rows (i&3) + 8(i>>2) + 4h --
// and the element-wise fp64 update is done right there, 16 rows x NT columns per lane.
template <int T, int NT>
__global__ __launch_bounds__(256, 2) void mu_epilogue_kernel(bmf_epilogue_args a) {
    if (a.stop && *a.stop != 0) return;
    constexpr int KP = 32 * NT;
    __shared__ __attribute__((aligned(16))) char tile[(T > 0 ? T : 1) * (T > 0 ? KP * LDS_ROW : 16)];  // T = 0: no panel
    __shared__ double red[4][2];
    __shared__ float cmax[4][KP];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int k = a.k;
    const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 32;  // first row of this wave
    const double reg = a.reg;
    const bool update = a.mode != BMF_MODE_PREPARE;

    // ---- all of this wave's inputs are requested up front (independent loads in flight; they land during the MFMAs) ----
    double fv[16][NT];
    float nv[16][NT];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            fv[i][nt] = a.F64[(row0 + (i & 3) + 8 * (i >> 2) + 4 * h) * KP + 32 * nt + c];
            nv[i][nt] = 0.f;
        }
    if (a.num) {
        // plain slabs [rows_pad][KP], or 32-column blocks [KP / 32][rows_pad][32] (the sharded exchange buffer)
        const int64_t ld = a.num_block_stride ? 32 : KP, bs = a.num_block_stride ? a.num_block_stride : 32;
        for (int sp = 0; sp < a.splits; ++sp) {
            const float* np_ = a.num + (int64_t)sp * a.slab_stride + (row0 + 4 * h) * ld + c;
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) nv[i][nt] += np_[(int64_t)((i & 3) + 8 * (i >> 2)) * ld + bs * nt];
        }
    }

    // ---- F G on the matrix cores ----
    f32x16 fg[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) fg[nt][i] = 0.f;
    if (update && !a.den) {
        constexpr int KH = KP / 2;  // reduction indices per lane half
        float av[KH], gv[NT][KH];
        const float* ap = a.F + (row0 + c) * KP + KH * h;
#pragma unroll
        for (int s = 0; s < KH; s += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(ap + s);
            av[s] = v[0]; av[s + 1] = v[1]; av[s + 2] = v[2]; av[s + 3] = v[3];
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int s = 0; s < KH; ++s) gv[nt][s] = a.G[(KH * h + s) * KP + 32 * nt + c];
#pragma unroll
        for (int s = 0; s < KH; ++s)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) fg[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], gv[nt][s], fg[nt], 0, 0, 0);
    }

    // ---- element-wise update in fp64, C/D layout: element (i, nt) = row (i&3) + 8(i>>2) + 4h, column 32nt + c ----
    double reg_acc = 0.0, dot_acc = 0.0;
    unsigned colword[NT];
    float cm[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        colword[nt] = 0u;
        cm[nt] = 0.f;
    }

#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int rl = (i & 3) + 8 * (i >> 2) + 4 * h;  // row inside the wave's 32
        const int64_t r = row0 + rl;
        const bool row_ok = r < a.rows;
        unsigned long long ball[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = 32 * nt + c;
            const bool ok = row_ok && col < k;
            const int64_t idx = r * KP + col;
            const double f = fv[i][nt];
            const float num = nv[i][nt];
            double fn = f;
            if (update) {
                double den = a.den ? (double)a.den[idx] : (double)fg[nt][i];
                double nume = (double)num;
                if (a.mode == BMF_MODE_PENALTY) {
                    const double f2 = f * f;
                    nume = nume + 3.0 * reg * f2;
                    den = den + (2.0 * reg * (f2 * f) + reg * f);
                }
                if (den == 0.0) den = BMF_EPS_D;
                fn = f * (nume / den);
                if (a.mode == BMF_MODE_PENALTY && fn == 0.0) fn = BMF_EPS_D;
            }
            if (!ok) fn = 0.0;
            const float fn32 = (float)fn;
            if (update) a.F64[idx] = fn;
            a.F[idx] = fn32;  // fp32 shadow (also refreshed in PREPARE mode)
            cm[nt] = fmaxf(cm[nt], fabsf(fn32));

            const double d = fn * fn - fn;
            reg_acc += d * d;
            dot_acc += fn * (double)num;

            const bool bit = ok && (fn > (double)a.thr);
            ball[nt] = __ballot(bit);
            colword[nt] |= (bit ? 1u : 0u) << rl;

            // bf16 addends into the LDS tile at the permuted position of row (wave*32 + rl)
            if constexpr (T > 0) {
                const int pos = panel_pos(wave * 32 + rl);
                float rem = fn32;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const uint16_t b = bf16_bits(rem);
                    *reinterpret_cast<uint16_t*>(tile + (t * KP + col) * LDS_ROW + 2 * pos) = b;
                    rem -= bf16_to_f32(b);
                }
            }
        }
        // k-bit row words: lanes 0-31 answered for row rl(h=0), lanes 32-63 for that row + 4
        if (lane == 0) {
            unsigned long long lo = (unsigned)ball[0], hi = (unsigned)(ball[0] >> 32);
            if (NT == 2) {
                lo |= (unsigned long long)(unsigned)ball[NT - 1] << 32;
                hi |= (unsigned long long)(unsigned)(ball[NT - 1] >> 32) << 32;
            }
            const int64_t ra = row0 + (i & 3) + 8 * (i >> 2);
            a.rowbits[ra] = lo;
            a.rowbits[ra + 4] = hi;
        }
    }
    // bit-columns: the two lane halves hold complementary row sets of the same column
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const unsigned w = colword[nt] | __shfl_xor(colword[nt], 32, 64);
        if (h == 0) a.colbits[(int64_t)(32 * nt + c) * a.ldcb + (row0 >> 5)] = w;
    }

    const double rs = wave_sum(reg_acc);
    const double ds = wave_sum(dot_acc);
    if (lane == 0) {
        red[wave][0] = rs;
        red[wave][1] = ds;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float mx = fmaxf(cm[nt], __shfl_xor(cm[nt], 32, 64));
        if (h == 0) cmax[wave][32 * nt + c] = mx;
    }
    __syncthreads();
    if (a.blockmax && threadIdx.x < KP)
        a.blockmax[(int64_t)blockIdx.x * KP + threadIdx.x] =
            fmaxf(fmaxf(cmax[0][threadIdx.x], cmax[1][threadIdx.x]), fmaxf(cmax[2][threadIdx.x], cmax[3][threadIdx.x]));
    if (threadIdx.x == 0) {
        a.partials[2 * blockIdx.x + 0] = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
        a.partials[2 * blockIdx.x + 1] = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
    }
    // staged panel tile -> global, 8 bytes per thread, 32 threads per 256-byte (term, column) row
    if constexpr (T > 0) {
        const int64_t blk0 = (int64_t)blockIdx.x * 128;
        constexpr int pieces = T * KP * 32;
        for (int p = threadIdx.x; p < pieces; p += 256) {
            const int rowi = p >> 5, off = (p & 31) * 8;  // rowi = t*KP + j
            const uint2 v = *reinterpret_cast<const uint2*>(tile + rowi * LDS_ROW + off);
            *reinterpret_cast<uint2*>(reinterpret_cast<char*>(a.panel + (int64_t)rowi * a.ldp + blk0) + off) = v;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The same update with the int8 digit planes of the new factor (the operand of the next bits GEMM, xf_bits_i8.hip) emitted HERE,
// from the fp64 values while they are in registers -- instead of by a second kernel that re-reads the fp64 master
// (make_panel_i8_kernel: 27 + 9 us per iteration at the headline shape).
//
// The planes need a per-column power-of-two scale 2^e_c that puts the column maximum just under 2^23, and the maximum of the NEW
// column is only known once the whole factor is updated.  So the planes are built with a PREDICTED scale: the one the previous
// iteration's maximum implies.  The column-scale step that follows (extra blocks of the Gram launch,
// bmf_colscale_i8_fused_block) checks the prediction against the maxima this kernel leaves in `blockmax`: if some column
// overflowed three balanced digits, or lost more than one bit, it raises a flag and the stand-alone builder rebuilds all planes
// with the exact scale (a no-op launch otherwise).  Digits: q = rint(F 2^e), |q| <= 8 355 711, balanced base
// 256 -- identical to make_panel_i8_kernel, so a rebuilt and a predicted plane set differ only in e.
//
// Where the bytes go needs no staging: in the C/D layout of the 32x32 MFMA lane (c, h) of wave w owns, for its column, the rows
// rl = (i & 3) + 8 (i >> 2) + 4 h, i = 0..15, of the wave's 32 -- and in the plane order (bmf_panel_pos_i8_dev: stage t = w, k-step
// ks = h, byte 4 (s & 3) + b with s & 3 = i & 3, b = i >> 2) those are exactly the 16 bytes of ONE 16-byte segment.  Each lane
// assembles its segments in registers and stores them, one 16-byte store per (digit, 32-column tile).
//
// Registers: the element-wise part walks the 16 rows of a lane in eight chunks of two through a ring of four register sets (loads
// issued three chunks ahead), which leaves room for the 24 registers of plane segments beside the accumulators at two waves per SIMD.
// MODE, LIMBS: compile-time (the run-time forms kept the element-wise part full of branches, and the register allocator spilled
// around them)
template <int NT, int MODE, int LIMBS>
__global__ __launch_bounds__(256, 2) void mu_epilogue_i8_kernel(bmf_epilogue_args a) {
#ifdef BMF_EPI_STAMP
    if (blockIdx.x == 0 && threadIdx.x == 0) { bmf_epi_stamps[0] = __builtin_amdgcn_s_memtime(); bmf_epi_stamps[14] = __builtin_amdgcn_s_memrealtime(); }
#endif
    if (a.stop && *a.stop != 0) return;
    BMF_EPI_STAMP_AT(1);   // stop word read
    constexpr int KP = 32 * NT;
    __shared__ double red[4][2];
    __shared__ float cmax[4][KP];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int k = a.k;
    // Block -> 128-row block, XCD-aware: the four blocks of a 512-row group write the four 16-byte quarters of every 64-byte piece of
    // the digit planes.  Consecutive block ids go round the 8 XCDs (each with its own L2), so with the plain map those quarters
    // would leave four different L2s as four partial writes; here the ids that share an XCD (id % 8) take the four blocks of one
    // group one after the other, and the piece is merged in one L2.  (The tail that does not fill a round of 32 ids maps plainly.)
    const int nblk_all = (int)gridDim.x;
    const int bid = (int)blockIdx.x;
    const int blk = bid < (nblk_all & ~31) ? (((bid >> 5) * 8 + (bid & 7)) << 2) + ((bid >> 3) & 3) : bid;
    const int64_t row0 = (int64_t)blk * 128 + wave * 32;
    const double reg = a.reg;
    constexpr bool update = MODE != BMF_MODE_PREPARE;
    constexpr int limbs = LIMBS;

    // ---- operands of F G (exact-fp32 MFMA): requested first, consumed after the first chunks' loads have been issued too ----
    constexpr int KH = KP / 2;   // reduction indices per lane half
    float av[KH], gv[NT][KH];
    if constexpr (update) {
        const float* ap = a.F + (row0 + c) * KP + KH * h;
#pragma unroll
        for (int s = 0; s < KH; s += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(ap + s);
            av[s] = v[0]; av[s + 1] = v[1]; av[s + 2] = v[2]; av[s + 3] = v[3];
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int s = 0; s < KH; ++s) gv[nt][s] = a.G[(KH * h + s) * KP + 32 * nt + c];
    }
    f32x16 fg[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) fg[nt][i] = 0.f;

    // plane scale of this lane's columns (the predicted 2^e_c)
    double psc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) psc[nt] = (double)a.plane_scale[32 * nt + c];

    double reg_acc = 0.0, dot_acc = 0.0;
    unsigned colword[NT];
    float cm[NT];
    unsigned seg[3][NT][4];   // [digit][column tile][dword of the 16-byte segment]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        colword[nt] = 0u;
        cm[nt] = 0.f;
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int j = 0; j < 4; ++j) seg[l][nt][j] = 0u;
    }

    const int64_t ld = a.num_block_stride ? 32 : KP, bs = a.num_block_stride ? a.num_block_stride : 32;
    const unsigned loff = (unsigned)(4 * h * KP + c);          // lane part of an element offset in F64 / F (row 4 h, column c)
    const unsigned noff = (unsigned)(4 * h * (int)ld + c);     // ... in the numerator slabs
    // chunk q2 = rows i = 2 q2 + jj, jj = 0, 1  (i & 3 = 2 (q2 & 1) + jj, i >> 2 = q2 >> 1, rl = (i & 3) + 8 (i >> 2) + 4 h)
    constexpr int CR = 2;   // rows of a lane per chunk
    auto load_chunk = [&](int q2, double (&fv)[CR][NT], float (&nv)[CR][NT]) {
#pragma unroll
        for (int jj = 0; jj < CR; ++jj) {
            const int i = CR * q2 + jj;
            // wave-uniform base of the chunk's 8-row group + the lane's 32-bit offset: scalar-base addressing, ONE offset register
            // for all elements (64-bit per-element addresses were most of the register pressure of the first version)
            const double* fq = a.F64 + (row0 + 8 * (i >> 2)) * KP;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                fv[jj][nt] = fq[loff + (i & 3) * KP + 32 * nt];
                nv[jj][nt] = 0.f;
            }
        }
        if (a.num)
            for (int sp = 0; sp < a.splits; ++sp) {
#pragma unroll
                for (int jj = 0; jj < CR; ++jj) {
                    const int i = CR * q2 + jj;
                    const float* nq = a.num + (int64_t)sp * a.slab_stride + (row0 + 8 * (i >> 2)) * ld;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) nv[jj][nt] += nq[noff + (i & 3) * (int)ld + (int)bs * nt];
                }
            }
    };
    auto do_chunk = [&](int q2, const double (&fv)[CR][NT], const float (&nv)[CR][NT]) {
#pragma unroll
        for (int jj = 0; jj < CR; ++jj) {
            const int i = CR * q2 + jj;
            const int j = i & 3, q = i >> 2;
            const int rl = j + 8 * q + 4 * h;
            const int64_t r = row0 + rl;
            const bool row_ok = r < a.rows;
            double* fq = a.F64 + (row0 + 8 * q) * KP;   // wave-uniform bases, see load_chunk
            float* sq = a.F + (row0 + 8 * q) * KP;
            unsigned long long ball[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = 32 * nt + c;
                const bool ok = row_ok && col < k;
                const unsigned eoff = loff + j * KP + 32 * nt;
                const double f = fv[jj][nt];
                const float num = nv[jj][nt];
                double fn = f;
                if constexpr (update) {
                    double den = (double)fg[nt][i];
                    double nume = (double)num;
                    if constexpr (MODE == BMF_MODE_PENALTY) {
                        const double f2 = f * f;
                        nume = nume + 3.0 * reg * f2;
                        den = den + (2.0 * reg * (f2 * f) + reg * f);
                    }
                    if (den == 0.0) den = BMF_EPS_D;
#if defined(BMF_EXP_EPI_NODIV)      // timing experiments only (wrong / differently rounded results)
                    fn = f * (nume * den);
#elif defined(BMF_EXP_EPI_FASTDIV)
                    {
                        double r = __builtin_amdgcn_rcp(den);
                        r = fma(fma(-den, r, 1.0), r, r);
                        r = fma(fma(-den, r, 1.0), r, r);
                        const double q0 = nume * r;
                        fn = f * fma(fma(-den, q0, nume), r, q0);
                    }
#else
                    fn = f * (nume / den);
#endif
                    if (MODE == BMF_MODE_PENALTY && fn == 0.0) fn = BMF_EPS_D;
                }
                if (!ok) fn = 0.0;
                const float fn32 = (float)fn;
                if constexpr (update) fq[eoff] = fn;
                sq[eoff] = fn32;
                cm[nt] = fmaxf(cm[nt], fabsf(fn32));

                const double d = fn * fn - fn;
                reg_acc += d * d;
                dot_acc += fn * (double)num;

                const bool bit = ok && (fn > (double)a.thr);
                ball[nt] = __ballot(bit);
                colword[nt] |= (bit ? 1u : 0u) << rl;

                // digits of q = rint(fn 2^e): byte (i >> 2) of dword (i & 3) of this lane's segment (see the header comment)
#ifdef BMF_EXP_EPI_NODIGITS
                int qi = 0;
#else
                int qi = __double2int_rn(fmax(fmin(fn * psc[nt], 8355711.0), -8355711.0));
#endif
                if constexpr (limbs == 2) {
                    qi = (qi + 128) >> 8;
                    const int d1 = ((qi + 128) & 255) - 128;
                    const int d2 = (qi - d1) >> 8;
                    seg[0][nt][j] |= (unsigned)(d1 & 255) << (8 * q);
                    seg[1][nt][j] |= (unsigned)(d2 & 255) << (8 * q);
                } else {
                    const int d0 = ((qi + 128) & 255) - 128;
                    const int q1 = (qi - d0) >> 8;
                    const int d1 = ((q1 + 128) & 255) - 128;
                    const int d2 = (q1 - d1) >> 8;
                    seg[0][nt][j] |= (unsigned)(d0 & 255) << (8 * q);
                    seg[1][nt][j] |= (unsigned)(d1 & 255) << (8 * q);
                    seg[2][nt][j] |= (unsigned)(d2 & 255) << (8 * q);
                }
                __builtin_amdgcn_sched_barrier(0);   // one element at a time (see below)
            }
            if (lane == 0) {
                unsigned long long lo = (unsigned)ball[0], hi = (unsigned)(ball[0] >> 32);
                if (NT == 2) {
                    lo |= (unsigned long long)(unsigned)ball[NT - 1] << 32;
                    hi |= (unsigned long long)(unsigned)(ball[NT - 1] >> 32) << 32;
                }
                const int64_t ra = row0 + j + 8 * q;
                a.rowbits[ra] = lo;
                a.rowbits[ra + 4] = hi;
            }
            // one row at a time: interleaving the fp64 divisions of all the elements of a chunk (what the scheduler does on its own)
            // needs ~12 temporaries per element and spills
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    {
        // A ring of four register sets, the loads of a chunk issued THREE chunks ahead: with one chunk of look-ahead a wave made
        // eight dependent memory round trips of ~2 us each under load and the kernel was latency-bound (72 us for U at the
        // headline shape, against 62 + 27 for the two kernels it replaces).  (sched_barriers: the scheduler must not hoist further.)
        double fr[4][CR][NT];
        float nr[4][CR][NT];
        load_chunk(0, fr[0], nr[0]);
        load_chunk(1, fr[1], nr[1]);
        load_chunk(2, fr[2], nr[2]);
        __builtin_amdgcn_sched_barrier(0);
        BMF_EPI_STAMP_AT(2);   // operands of F G, plane scales and the first three chunks landed
        if constexpr (update) {   // F G while those loads are in flight
#pragma unroll
            for (int s = 0; s < KH; ++s)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) fg[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], gv[nt][s], fg[nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        BMF_EPI_STAMP_AT(3);   // F G
#pragma unroll
        for (int q2 = 0; q2 < 16 / CR; ++q2) {
            if (q2 + 3 < 16 / CR) load_chunk(q2 + 3, fr[(q2 + 3) & 3], nr[(q2 + 3) & 3]);
            __builtin_amdgcn_sched_barrier(0);
            do_chunk(q2, fr[q2 & 3], nr[q2 & 3]);
            __builtin_amdgcn_sched_barrier(0);
            BMF_EPI_STAMP_NW(4 + q2);   // chunk q2 done (issue side)
        }
    }

    // the digit planes: segment (stage t = wave, k-step ks = h) of the 128-byte stage row of (digit, column), block-local group g
    {
        const int g = blk & 3;
        const int64_t blk512 = ((int64_t)blk >> 2) << 9;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int l = 0; l < 3; ++l)
                if (l < limbs) {
                    const u32x4 v = {seg[l][nt][0], seg[l][nt][1], seg[l][nt][2], seg[l][nt][3]};
                    *reinterpret_cast<u32x4*>(a.planes + (int64_t)(l * KP + 32 * nt + c) * a.ldp + blk512 + 128 * wave + (4 * h + g) * 16) = v;
                }
    }

    // bit-columns: the two lane halves hold complementary row sets of the same column
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const unsigned w = colword[nt] | __shfl_xor(colword[nt], 32, 64);
        if (h == 0) a.colbits[(int64_t)(32 * nt + c) * a.ldcb + (row0 >> 5)] = w;
    }
    const double rs = wave_sum(reg_acc);
    const double ds = wave_sum(dot_acc);
    if (lane == 0) {
        red[wave][0] = rs;
        red[wave][1] = ds;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float mx = fmaxf(cm[nt], __shfl_xor(cm[nt], 32, 64));
        if (h == 0) cmax[wave][32 * nt + c] = mx;
    }
    __syncthreads();
    if (threadIdx.x < KP)
        a.blockmax[(int64_t)blk * KP + threadIdx.x] =
            fmaxf(fmaxf(cmax[0][threadIdx.x], cmax[1][threadIdx.x]), fmaxf(cmax[2][threadIdx.x], cmax[3][threadIdx.x]));
    if (threadIdx.x == 0) {
        a.partials[2 * blk + 0] = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
        a.partials[2 * blk + 1] = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
    }
    BMF_EPI_STAMP_AT(12);   // planes, bit-columns, block sums
#ifdef BMF_EPI_STAMP
    if (blockIdx.x == 0 && threadIdx.x == 0) bmf_epi_stamps[15] = __builtin_amdgcn_s_memrealtime();
#endif
}

}  // namespace

extern "C" int bmf_mu_epilogue(const bmf_epilogue_args* a, void* stream) {
    BMF_REQUIRE(a, "bmf_mu_epilogue: null args");
    BMF_REQUIRE(a->F && a->F64 && (a->panel || a->terms == 0) && a->rowbits && a->colbits && a->partials, "bmf_mu_epilogue: null pointer");
    BMF_REQUIRE(a->rows_pad > 0 && a->rows_pad % 128 == 0, "bmf_mu_epilogue: rows_pad must be a multiple of 128");
    BMF_REQUIRE(a->rows >= 1 && a->rows <= a->rows_pad, "bmf_mu_epilogue: rows out of range");
    BMF_REQUIRE((a->kp == 32 || a->kp == 64) && a->k >= 1 && a->k <= a->kp, "bmf_mu_epilogue: need 1 <= k <= kp, kp in {32,64}");
    BMF_REQUIRE(a->mode >= 0 && a->mode <= 2, "bmf_mu_epilogue: bad mode");
    BMF_REQUIRE(a->mode == BMF_MODE_PREPARE || ((a->G || a->den) && a->num), "bmf_mu_epilogue: update modes need num and G (or den)");
    BMF_REQUIRE(!a->num || (a->splits >= 1 && a->slab_stride >= a->rows_pad * a->kp), "bmf_mu_epilogue: bad slab description");
    BMF_REQUIRE(a->terms >= 0 && a->terms <= 3, "bmf_mu_epilogue: terms must be 0..3 (0 = no bf16 panel)");
    BMF_REQUIRE(a->ldp >= a->rows_pad && a->ldp % 4 == 0, "bmf_mu_epilogue: ldp must be >= rows_pad and a multiple of 4");
    BMF_REQUIRE(a->ldcb >= a->rows_pad / 32, "bmf_mu_epilogue: ldcb too small");
    BMF_REQUIRE(a->terms == 0 || ((uintptr_t)a->panel & 7u) == 0, "bmf_mu_epilogue: panel must be 8-byte aligned");
    BMF_REQUIRE(bmf_aligned16(a->F), "bmf_mu_epilogue: F must be 16-byte aligned");
    dim3 grid((unsigned)(a->rows_pad / 128)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (a->planes) {   // int8 digit planes emitted in-line (see mu_epilogue_i8_kernel)
        BMF_REQUIRE(a->terms == 0 && (a->limbs == 2 || a->limbs == 3) && a->plane_scale && a->blockmax,
                    "bmf_mu_epilogue: planes need terms == 0, limbs 2 or 3, plane_scale and blockmax");
        BMF_REQUIRE(a->rows_pad % 512 == 0 && a->ldp >= a->rows_pad && a->ldp % 16 == 0 && bmf_aligned16(a->planes),
                    "bmf_mu_epilogue: planes need rows_pad %% 512 == 0, ldp >= rows_pad, ldp %% 16 == 0 and a 16-byte aligned pointer");
        BMF_REQUIRE(!a->den, "bmf_mu_epilogue: planes cannot be combined with a precomputed denominator (den)");
#define BMF_EPI8_CASE(NT_, M_, L_) \
    if (a->kp == 32 * NT_ && a->mode == M_ && a->limbs == L_) BMF_LAUNCH((mu_epilogue_i8_kernel<NT_, M_, L_>), grid, block, 0, s, *a);
        BMF_EPI8_CASE(1, 0, 2) BMF_EPI8_CASE(1, 0, 3) BMF_EPI8_CASE(1, 1, 2) BMF_EPI8_CASE(1, 1, 3) BMF_EPI8_CASE(1, 2, 2) BMF_EPI8_CASE(1, 2, 3)
        BMF_EPI8_CASE(2, 0, 2) BMF_EPI8_CASE(2, 0, 3) BMF_EPI8_CASE(2, 1, 2) BMF_EPI8_CASE(2, 1, 3) BMF_EPI8_CASE(2, 2, 2) BMF_EPI8_CASE(2, 2, 3)
#undef BMF_EPI8_CASE
        BMF_LAUNCH_CHECK();
        return BMF_OK;
    }
#define BMF_EPI_CASE(T_, NT_) \
    if (a->terms == T_ && a->kp == 32 * NT_) BMF_LAUNCH((mu_epilogue_kernel<T_, NT_>), grid, block, 0, s, *a);
    BMF_EPI_CASE(0, 1) BMF_EPI_CASE(0, 2) BMF_EPI_CASE(1, 1) BMF_EPI_CASE(2, 1) BMF_EPI_CASE(3, 1) BMF_EPI_CASE(1, 2) BMF_EPI_CASE(2, 2) BMF_EPI_CASE(3, 2)
#undef BMF_EPI_CASE
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}
