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
// One block = 128 rows (one panel permutation block), 4 waves x 32 consecutive rows.  F G is done on the VALU with
// G's column in registers and the row broadcast through v_readlane (k <= 64): ~2k instructions per row.
//
// Precision: the factor itself is kept in fp64 (F64, the master copy) and the element-wise part of the update -- the
// penalty terms, the ratio, the clamps, the regulariser sum -- is evaluated in fp64.  Only the two contractions feeding
// it (num from the bits GEMM, F G here) are fp32-accurate, which is enough: they are sums of O(1) positive terms.  This
// matters once lambda is large (the reference's default schedule reaches 1e10): entries converge to 1 like 1 - (2/3)^t
// and reg_error = lambda/2 * sum (u^2-u)^2 must keep falling below `tol` -- fp32 entries cannot get closer to 1 than
// 6e-8.  An fp32 shadow copy (F) is written alongside for the Gram / residual / MAE kernels.
#include "common.h"

namespace {

constexpr int LDS_ROW = 264;  // bytes per (term, column) row of the staged panel tile: 256 + 8 pad (2-way max on b16 writes)

template <int T>
__global__ __launch_bounds__(256) void mu_epilogue_kernel(bmf_epilogue_args a) {
    if (a.stop && *a.stop != 0) return;
    __shared__ __attribute__((aligned(16))) char tile[T * BMF_MAX_KP * LDS_ROW];
    __shared__ double red[4][2];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kp = a.kp, k = a.k;
    const int64_t row0 = (int64_t)blockIdx.x * 128;
    const bool col_ok = lane < k;
    const bool col_in = lane < kp;

    // column `lane` of G in registers (zero beyond kp)
    float Gc[BMF_MAX_KP];
#pragma unroll
    for (int l = 0; l < BMF_MAX_KP; ++l) Gc[l] = (col_in && l < kp && a.mode != BMF_MODE_PREPARE) ? a.G[l * kp + lane] : 0.f;

    const double reg = a.reg;
    double reg_acc = 0.0, dot_acc = 0.0;
    unsigned colword = 0;

    // all 32 rows of this wave are loaded up front (independent loads in flight), the loop below is pure ALU
    double fbuf[32];
    float nbuf[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int64_t r = row0 + wave * 32 + i;
        fbuf[i] = col_in ? a.F64[r * kp + lane] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) nbuf[i] = 0.f;
    if (a.num && col_in) {
        for (int s = 0; s < a.splits; ++s) {
            const float* np_ = a.num + (int64_t)s * a.slab_stride + (row0 + wave * 32) * kp + lane;
#pragma unroll
            for (int i = 0; i < 32; ++i) nbuf[i] += np_[(int64_t)i * kp];
        }
    }

#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int rl = wave * 32 + i;  // row inside the 128-block
        const int64_t r = row0 + rl;
        const bool row_ok = r < a.rows;
        const bool ok = row_ok && col_ok;
        const double f = fbuf[i];
        const float f32 = (float)f, num = nbuf[i];
        double fn = f;
        if (a.mode != BMF_MODE_PREPARE) {
            float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;  // (F G)[r][lane] in fp32, four independent chains
#pragma unroll
            for (int l = 0; l < BMF_MAX_KP; l += 4) {
                d0 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(f32), l + 0)), Gc[l + 0], d0);
                d1 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(f32), l + 1)), Gc[l + 1], d1);
                d2 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(f32), l + 2)), Gc[l + 2], d2);
                d3 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(f32), l + 3)), Gc[l + 3], d3);
            }
            double den = (double)((d0 + d1) + (d2 + d3));
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
        if (col_in && a.mode != BMF_MODE_PREPARE) a.F64[r * kp + lane] = fn;
        if (col_in) a.F[r * kp + lane] = fn32;  // fp32 shadow (also refreshed in PREPARE mode)

        const double d = fn * fn - fn;
        reg_acc += d * d;
        dot_acc += fn * (double)num;

        const bool bit = ok && (fn > (double)a.thr);
        const unsigned long long rb = __ballot(bit);
        if (lane == 0) a.rowbits[r] = rb;
        colword |= (bit ? 1u : 0u) << i;

        // bf16 addends into the LDS tile at the permuted position
        if (col_in) {
            const int pos = panel_pos(rl);
            float rem = fn32;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const uint16_t b = bf16_bits(rem);
                *reinterpret_cast<uint16_t*>(tile + (t * BMF_MAX_KP + lane) * LDS_ROW + 2 * pos) = b;
                rem -= bf16_to_f32(b);
            }
        }
    }
    if (col_in) a.colbits[(int64_t)lane * a.ldcb + (row0 >> 5) + wave] = colword;

    const double rs = wave_sum(reg_acc);
    const double ds = wave_sum(dot_acc);
    if (lane == 0) {
        red[wave][0] = rs;
        red[wave][1] = ds;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.partials[2 * blockIdx.x + 0] = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
        a.partials[2 * blockIdx.x + 1] = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
    }
    // staged panel tile -> global, 8 bytes per thread, 32 threads per 256-byte (term, column) row
    const int pieces = T * kp * 32;
    for (int p = threadIdx.x; p < pieces; p += 256) {
        const int rowi = p >> 5, off = (p & 31) * 8;  // rowi = t*kp + j
        const int t = rowi / kp, j = rowi - t * kp;
        const uint2 v = *reinterpret_cast<const uint2*>(tile + (t * BMF_MAX_KP + j) * LDS_ROW + off);
        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(a.panel + ((int64_t)t * kp + j) * a.ldp + row0) + off) = v;
    }
}

}  // namespace

extern "C" int bmf_mu_epilogue(const bmf_epilogue_args* a, void* stream) {
    BMF_REQUIRE(a, "bmf_mu_epilogue: null args");
    BMF_REQUIRE(a->F && a->F64 && a->panel && a->rowbits && a->colbits && a->partials, "bmf_mu_epilogue: null pointer");
    BMF_REQUIRE(a->rows_pad > 0 && a->rows_pad % 128 == 0, "bmf_mu_epilogue: rows_pad must be a multiple of 128");
    BMF_REQUIRE(a->rows >= 1 && a->rows <= a->rows_pad, "bmf_mu_epilogue: rows out of range");
    BMF_REQUIRE((a->kp == 32 || a->kp == 64) && a->k >= 1 && a->k <= a->kp, "bmf_mu_epilogue: need 1 <= k <= kp, kp in {32,64}");
    BMF_REQUIRE(a->mode >= 0 && a->mode <= 2, "bmf_mu_epilogue: bad mode");
    BMF_REQUIRE(a->mode == BMF_MODE_PREPARE || (a->G && a->num), "bmf_mu_epilogue: update modes need G and num");
    BMF_REQUIRE(!a->num || (a->splits >= 1 && a->slab_stride >= a->rows_pad * a->kp), "bmf_mu_epilogue: bad slab description");
    BMF_REQUIRE(a->terms >= 1 && a->terms <= 3, "bmf_mu_epilogue: terms must be 1..3");
    BMF_REQUIRE(a->ldp >= a->rows_pad && a->ldp % 4 == 0, "bmf_mu_epilogue: ldp must be >= rows_pad and a multiple of 4");
    BMF_REQUIRE(a->ldcb >= a->rows_pad / 32, "bmf_mu_epilogue: ldcb too small");
    BMF_REQUIRE(((uintptr_t)a->panel & 7u) == 0, "bmf_mu_epilogue: panel must be 8-byte aligned");
    dim3 grid((unsigned)(a->rows_pad / 128)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (a->terms == 1) BMF_LAUNCH(mu_epilogue_kernel<1>, grid, block, 0, s, *a);
    if (a->terms == 2) BMF_LAUNCH(mu_epilogue_kernel<2>, grid, block, 0, s, *a);
    if (a->terms == 3) BMF_LAUNCH(mu_epilogue_kernel<3>, grid, block, 0, s, *a);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}
