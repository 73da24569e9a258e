// Whole-iteration driver for WNMF (Frobenius, all-ones mask) on a REAL-VALUED X -- BASELINE config #2 -- with no host in the loop
// (PyBMF/models/WNMF.py:51-109: _fit loop, update; :133-144: error).
//
// Per iteration, all enqueued on one stream:  V epilogue (numerator X^T U, Gram U^T U) -> V^T (the transposed operand of the
// next contraction) -> V^T V -> X V -> U epilogue -> U^T -> U^T U -> X^T U (for the NEXT V update) -> [residual pass for MAE] ->
// finalize: error by the trace form 1/2 (sum X^2 - 2 <U, X V> + <U^T U, V^T V>) from by-products of the update (<U, X V> is a
// partial sum of the U epilogue), RMSE, MAE, and the stopping rule (models/BaseModelTools.py:326-334) on the device: a raised
// flag turns every later kernel of this driver into a no-op, so max_iter + 1 iterations can be enqueued blindly.
// X is read twice per iteration (three times with MAE); the host-driven loop of round 1 read it four times and synchronised
// every iteration.
#include "common.h"

#include <cstdlib>

int bmf_residual_launch_f32(const float* X, int64_t m_pad, int64_t ldx, int m, int n, const float* U, const float* V, int kp, double* sums,
                            const int32_t* stop, hipStream_t s);
int bmf_xf_f32_launch(const float* A, int64_t rows_pad, int64_t lda, int64_t red, const float* FT, int64_t ldft, int kp, float* out,
                      int64_t slab_stride, int splits, int a_tiled, int b_frag, const int32_t* stop, hipStream_t s);
int bmf_frag_f32_launch(const float* F, int64_t rows_pad, int kp, float* frag, const int32_t* stop, hipStream_t s);
int bmf_frag_rows_f32_launch(const float* V, int64_t rows_pad, int kp, float* frag, const int32_t* stop, hipStream_t s);
int bmf_residual_tiled_launch(const float* Xtiled, int64_t m_pad, int64_t n_pad, const float* U, const float* V, int kp, double* sums,
                              const int32_t* stop, hipStream_t s);
int bmf_xf_f32_resid_launch(const float* Atiled, int64_t rows_pad, int64_t red, const float* Ffrag, const uint32_t* Frf, const float* Grow, int kp,
                            float* out, int64_t slab_stride, int splits, double* sums, const int32_t* stop, hipStream_t s, const uint32_t* F3);
int bmf_xf_f32_bf3_launch(const float* Atiled, int64_t rows_pad, int64_t red, const uint32_t* F3, float* out, int64_t slab_stride, int splits,
                          const int32_t* stop, hipStream_t s);
int bmf_frag_rows_bf16_launch(const float* F, int64_t rows_pad, int kp, uint32_t* frag, const int32_t* stop, hipStream_t s);
int bmf_frag_pair_launch(const float* F, int64_t rows_pad, float* frag, uint32_t* frag_bf, double* zero4, const int32_t* stop, hipStream_t s);

namespace {

// FT[j][r] = F[r][j]: 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ F, int64_t rows_pad, int kp, float* __restrict__ FT,
                                                         const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int64_t r0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    for (int q = ty; q < 32; q += 8) tile[q][tx] = F[(r0 + q) * kp + c0 + tx];
    __syncthreads();
    for (int q = ty; q < 32; q += 8) FT[(int64_t)(c0 + q) * rows_pad + r0 + tx] = tile[tx][q];
}

__global__ __launch_bounds__(1024) void real_finalize_kernel(bmf_wnmf_real_state st, int iter, int max_iter) {
    const int sflag = *st.stop;
    if (sflag != 0 && iter > sflag) return;
    __shared__ double sh[1024];
    const int kk = st.kp * st.kp;
    double b = 0.0;
    for (int i = threadIdx.x; i < kk; i += 1024) b += st.GU64[i] * st.GV64[i];
    sh[threadIdx.x] = b;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    const double cross = sh[0];
    __syncthreads();
    double dot = 0.0;   // <U, X V>: the second partial of the U epilogue's blocks
    for (int i = threadIdx.x; i < (int)(st.m_pad / 128); i += 1024) dot += st.partU[2 * i + 1];
    sh[threadIdx.x] = dot;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    dot = sh[0];
    const double err = 0.5 * (st.sum_x2 - 2.0 * dot + cross);
    double* row = st.log + (int64_t)iter * BMF_LOG_COLS;
    row[BMF_LOG_ITER] = (double)iter;
    row[BMF_LOG_ERROR] = err;
    row[BMF_LOG_REC] = err;
    row[BMF_LOG_RMSE] = sqrt(fmax(2.0 * err, 0.0) / st.cells);
    row[BMF_LOG_MAE] = st.with_mae ? st.sums[0] / st.cells : __builtin_nan("");
    row[BMF_LOG_VALID] = 1.0;
    int stop_now = 0;
    if (iter >= 1) {   // WNMF watches the error (WNMF.py:72,89)
        const double diff = fabs(st.scal[1] - err);
        if (err <= st.tol) stop_now = 1;
        if (iter > max_iter) stop_now = 1;
        if (diff < st.min_diff) stop_now = 1;
    }
    st.scal[1] = err;
    row[BMF_LOG_STOP] = (double)stop_now;
    if (stop_now) *st.stop = iter;
}

// The whole factor update of the tiled, k <= 32 loop in ONE launch (round 4): what used to be the shared epilogue, the fragment
// re-order(s) of the new factor and the Gram partials -- three launches of 17 + 5 + 6 us and the gaps between them, per side.
// A block walks groups of 128 rows (wave w: rows 32 w .. + 31 of the group) with the lane layout of mu_epilogue_kernel: F G on the
// exact-fp32 MFMA leaves lane (c, h) with column c of the 16 rows (i & 3) + 8 (i >> 2) + 4 h, the fp64 update is done there
// (same arithmetic, same order: the new factor is bit-identical to the epilogue's), and then, from the registers,
//   * the fp32 shadow and the fp64 master,
//   * the order of the ring kernels' B operand (bmf_frag_f32): registers 4 u .. 4 u + 3 of a lane ARE piece (group, w, u) of it,
//   * with frag_bf, the bf16 hi / lo pairs in row order (bmf_frag_rows_bf16) through a 32 x 32 tile in LDS,
//   * the Gram partial: 16 MFMAs with the new values as both operands (lane (c, h), register i: the same row in both k slots),
//     kept in the accumulator over the block's groups and left as ONE slab per block (slab order = block order: deterministic),
//   * <Fn, num> per group (partials[2 g + 1], the <U, X V> term of the error's trace form).
// The numerator slabs are read four at a time (the epilogue's loop waited for each slab's 16 loads in turn: 13 round trips to L2
// for the 13 slabs of X^T U at 20000 x 5000) and added in slab order.
struct real_update_args {
    double* F64; float* F; int64_t rows_pad; int32_t rows, k;
    const float* num; int64_t slab_stride; int32_t splits, update;
    const float* G;
    float* frag; uint32_t* frag_bf;
    uint32_t* frag3;   // optional: the bf16 x 3 order of the new factor (xf_f32.hip: bmf_frag_bf16x3), operand of the bf16 contractions
    float* gram_slabs; double* partials; double* zero4;
    const int32_t* stop;
};

__global__ __launch_bounds__(256, 2) void real_update_kernel(real_update_args a) {
    if (a.stop && *a.stop != 0) return;
    constexpr int KP = 32, TS = 36;   // TS: row stride of the transposition tile in floats (16-byte aligned, conflict-free b128 reads)
    __shared__ __attribute__((aligned(16))) float sh[4][32 * TS];
    __shared__ double red[4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    if (a.zero4 && blockIdx.x == 0 && threadIdx.x < 4) a.zero4[threadIdx.x] = 0.0;

    float gv[16];   // B operand of F G: G[16 h + s][c]
    if (a.update) {
#pragma unroll
        for (int s = 0; s < 16; ++s) gv[s] = a.G[(16 * h + s) * KP + c];
    }
    f32x16 gram;
#pragma unroll
    for (int i = 0; i < 16; ++i) gram[i] = 0.f;

    const int groups = (int)(a.rows_pad / 128);
    for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int64_t row0 = (int64_t)grp * 128 + wave * 32;
        double fv[16];
        float nv[16];
        const double* fp = a.F64 + (row0 + 4 * h) * KP + c;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            fv[i] = fp[((i & 3) + 8 * (i >> 2)) * KP];
            nv[i] = 0.f;
        }
        float av[16];
        if (a.update) {
            const float* ap = a.F + (row0 + c) * KP + 16 * h;
#pragma unroll
            for (int s = 0; s < 16; s += 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(ap + s);
                av[s] = v[0]; av[s + 1] = v[1]; av[s + 2] = v[2]; av[s + 3] = v[3];
            }
        }
        if (a.num) {
            const float* np_ = a.num + (row0 + 4 * h) * KP + c;
            int sp = 0;
            for (; sp + 4 <= a.splits; sp += 4) {
                float t[4][16];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < 16; ++i) t[q][i] = np_[(int64_t)(sp + q) * a.slab_stride + ((i & 3) + 8 * (i >> 2)) * KP];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < 16; ++i) nv[i] += t[q][i];
            }
            for (; sp < a.splits; ++sp)
#pragma unroll
                for (int i = 0; i < 16; ++i) nv[i] += np_[(int64_t)sp * a.slab_stride + ((i & 3) + 8 * (i >> 2)) * KP];
        }
        f32x16 fg;
#pragma unroll
        for (int i = 0; i < 16; ++i) fg[i] = 0.f;
        if (a.update) {
#pragma unroll
            for (int s = 0; s < 16; ++s) fg = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], gv[s], fg, 0, 0, 0);
        }
        double dot_acc = 0.0;
        float fn32[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rl = (i & 3) + 8 * (i >> 2) + 4 * h;
            const int64_t r = row0 + rl;
            const bool ok = r < a.rows && c < a.k;
            double fn = fv[i];
            if (a.update) {
                double den = (double)fg[i];
                if (den == 0.0) den = BMF_EPS_D;
                fn = fv[i] * ((double)nv[i] / den);
            }
            if (!ok) fn = 0.0;
            fn32[i] = (float)fn;
            if (a.update) a.F64[r * KP + c] = fn;
            a.F[r * KP + c] = fn32[i];
            dot_acc += fn * (double)nv[i];
        }
        // the B operand order of the contraction kernels
        float* fo = a.frag + ((int64_t)(grp * 4 + wave) * 4) * 256 + lane * 4;
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(fo + u * 256) = f32x4{fn32[4 * u], fn32[4 * u + 1], fn32[4 * u + 2], fn32[4 * u + 3]};
        if (a.frag3) {   // registers 8 ks .. 8 ks + 7 of a lane ARE the eight elements of k-step ks of stage-half (grp, wave): split, store
            uint32_t* f3 = a.frag3 + ((int64_t)(grp * 4 + wave) * 6) * 256 + lane * 4;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const float x[8] = {fn32[8 * ks], fn32[8 * ks + 1], fn32[8 * ks + 2], fn32[8 * ks + 3], fn32[8 * ks + 4], fn32[8 * ks + 5], fn32[8 * ks + 6], fn32[8 * ks + 7]};
                u32x4 hi, mid, lo;
                bmf_split3_bf16(x, hi, mid, lo);
                *reinterpret_cast<u32x4*>(f3 + (3 * ks) * 256) = hi;
                *reinterpret_cast<u32x4*>(f3 + (3 * ks + 1) * 256) = mid;
                *reinterpret_cast<u32x4*>(f3 + (3 * ks + 2) * 256) = lo;
            }
        }
        // Gram of the new rows
#pragma unroll
        for (int i = 0; i < 16; ++i) gram = __builtin_amdgcn_mfma_f32_32x32x2f32(fn32[i], fn32[i], gram, 0, 0, 0);
        if (a.frag_bf) {   // bf16 pairs in row order: lane (r = c, h) takes row r, columns 16 ks + 8 h .. + 7
            float* tl = sh[wave];
#pragma unroll
            for (int i = 0; i < 16; ++i) tl[((i & 3) + 8 * (i >> 2) + 4 * h) * TS + c] = fn32[i];
            __builtin_amdgcn_wave_barrier();
            uint32_t* bo = a.frag_bf + ((int64_t)(grp * 4 + wave) * 4) * 256 + lane * 4;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(tl + c * TS + 16 * ks + 8 * h), v1 = *reinterpret_cast<const f32x4*>(tl + c * TS + 16 * ks + 8 * h + 4);
                const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                u32x4 hi, lo;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const uint16_t h0 = bf16_bits(v[2 * w]), h1 = bf16_bits(v[2 * w + 1]);
                    const uint16_t l0 = bf16_bits(v[2 * w] - bf16_to_f32(h0)), l1 = bf16_bits(v[2 * w + 1] - bf16_to_f32(h1));
                    hi[w] = (unsigned)h0 | ((unsigned)h1 << 16);
                    lo[w] = (unsigned)l0 | ((unsigned)l1 << 16);
                }
                *reinterpret_cast<u32x4*>(bo + (2 * ks) * 256) = hi;
                *reinterpret_cast<u32x4*>(bo + (2 * ks + 1) * 256) = lo;
            }
            __builtin_amdgcn_wave_barrier();
        }
        const double ds = wave_sum(dot_acc);
        __syncthreads();   // red[] of the previous group has been read
        if (lane == 0) red[wave] = ds;
        __syncthreads();
        if (threadIdx.x == 0) {
            a.partials[2 * grp + 0] = 0.0;
            a.partials[2 * grp + 1] = ((red[0] + red[1]) + red[2]) + red[3];
        }
    }
    // the block's Gram slab: the four waves' accumulators meet in LDS (C/D layout: col = c, row = (i & 3) + 8 (i >> 2) + 4 h)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) sh[wave][((i & 3) + 8 * (i >> 2) + 4 * h) * KP + c] = gram[i];
    __syncthreads();
    float* o = a.gram_slabs + (int64_t)blockIdx.x * KP * KP;
    for (int i = threadIdx.x; i < KP * KP; i += 256) o[i] = (sh[0][i] + sh[1][i]) + (sh[2][i] + sh[3][i]);
}

__global__ void zero_sums_kernel(double* sums, const int32_t* stop) {
    if (stop && *stop != 0) return;
    if (threadIdx.x < 4) sums[threadIdx.x] = 0.0;
}

}  // namespace

#define BMF_TRY(expr)                  \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != BMF_OK) return rc_; \
    } while (0)

static int check_real_state(const bmf_wnmf_real_state* st, const char* who) {
    BMF_REQUIRE(st, "%s: null state", who);
    BMF_REQUIRE(st->struct_bytes == (int32_t)sizeof(bmf_wnmf_real_state), "%s: struct_bytes=%d, library expects %d", who, st->struct_bytes,
                (int)sizeof(bmf_wnmf_real_state));
    BMF_REQUIRE(st->m >= 1 && st->n >= 1 && st->k >= 1 && (st->kp == 32 || st->kp == 64) && st->k <= st->kp, "%s: bad m, n, k, kp", who);
    BMF_REQUIRE(st->m_pad % 128 == 0 && st->n_pad % 128 == 0 && st->m_pad >= st->m && st->n_pad >= st->n, "%s: m_pad / n_pad must be multiples of 128", who);
    BMF_REQUIRE(st->X && st->XT && st->U64 && st->V64 && st->U && st->V && st->UT && st->VT && st->Mslab && st->Nslab && st->gram_slabs && st->GU &&
                    st->GV && st->GU64 && st->GV64 && st->partU && st->partV && st->rowbits && st->colbits && st->sums && st->scal && st->log && st->stop,
                "%s: null device pointer in state", who);
    BMF_REQUIRE((st->Xtiled == nullptr) == (st->XTtiled == nullptr), "%s: Xtiled and XTtiled go together", who);
    BMF_REQUIRE(!(st->Xtiled && st->with_mae) || st->Vrf, "%s: the tiled residual pass needs Vrf", who);
    BMF_REQUIRE((st->UT3 == nullptr) == (st->VT3 == nullptr), "%s: UT3 and VT3 go together", who);
    BMF_REQUIRE(!st->UT3 || (st->Xtiled && st->kp == 32), "%s: the bf16 x 3 orders (UT3 / VT3) go with the tiled matrix and kp == 32", who);
    BMF_REQUIRE(st->splits_xv >= 1 && st->splits_xtu >= 1 && st->gram_blocks >= 1 && st->gram_blocks <= 1024 && st->log_rows >= 1, "%s: bad splits / blocks", who);
    return BMF_OK;
}

static int epilogue(const bmf_wnmf_real_state* st, bool is_u, int mode, hipStream_t s) {
    bmf_epilogue_args a = {};
    a.F64 = is_u ? st->U64 : st->V64; a.F = is_u ? st->U : st->V;
    a.rows_pad = is_u ? st->m_pad : st->n_pad; a.rows = is_u ? st->m : st->n; a.k = st->k; a.kp = st->kp;
    a.num = is_u ? st->Mslab : st->Nslab; a.slab_stride = a.rows_pad * st->kp; a.splits = is_u ? st->splits_xv : st->splits_xtu;
    a.G = is_u ? st->GV : st->GU; a.reg = 0.0; a.mode = mode; a.thr = 0.5f; a.terms = 0;
    a.panel = nullptr; a.ldp = a.rows_pad; a.rowbits = st->rowbits; a.colbits = st->colbits; a.ldcb = st->ldcb;
    a.partials = is_u ? st->partU : st->partV; a.stop = st->stop;
    return bmf_mu_epilogue(&a, s);
}

static int gram(const bmf_wnmf_real_state* st, bool is_u, hipStream_t s) {
    const int kk = st->kp * st->kp;
    BMF_TRY(bmf_gram_partial(is_u ? st->U : st->V, is_u ? st->m_pad : st->n_pad, st->kp, st->kp, st->gram_slabs, st->gram_blocks, s));
    return bmf_reduce_slabs(st->gram_slabs, kk, st->gram_blocks, kk, is_u ? st->GU : st->GV, is_u ? st->GU64 : st->GV64, s);
}

static int transpose(const float* F, int64_t rows_pad, int kp, float* FT, const int32_t* stop, hipStream_t s) {
    BMF_LAUNCH(transpose_kernel, dim3((unsigned)(rows_pad / 32), (unsigned)(kp / 32)), dim3(256), 0, s, F, rows_pad, kp, FT, stop);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

// the fused update of one side (real_update_kernel) and the reduction of its Gram slabs; `bf`: also the bf16 row order + zeroed sums
static bool fused_update_ok(const bmf_wnmf_real_state* st) {
    return st->Xtiled && st->XTtiled && st->kp == 32;
}

static int fused_update(const bmf_wnmf_real_state* st, bool is_u, int mode, bool bf, hipStream_t s) {
    real_update_args a = {};
    a.F64 = is_u ? st->U64 : st->V64; a.F = is_u ? st->U : st->V;
    a.rows_pad = is_u ? st->m_pad : st->n_pad; a.rows = is_u ? st->m : st->n; a.k = st->k;
    a.num = is_u ? st->Mslab : st->Nslab; a.slab_stride = a.rows_pad * st->kp; a.splits = is_u ? st->splits_xv : st->splits_xtu;
    a.update = mode != BMF_MODE_PREPARE;
    a.G = is_u ? st->GV : st->GU;
    a.frag = is_u ? st->UT : st->VT; a.frag_bf = bf ? (uint32_t*)st->Urf : nullptr;
    a.frag3 = is_u ? st->UT3 : st->VT3;
    a.gram_slabs = st->gram_slabs; a.partials = is_u ? st->partU : st->partV; a.zero4 = bf ? st->sums : nullptr;
    a.stop = st->stop;
    const int groups = (int)(a.rows_pad / 128), blocks = groups < st->gram_blocks ? groups : st->gram_blocks;
    BMF_LAUNCH(real_update_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    BMF_LAUNCH_CHECK();
    const int kk = st->kp * st->kp;
    return bmf_reduce_slabs(st->gram_slabs, kk, blocks, kk, is_u ? st->GU : st->GV, is_u ? st->GU64 : st->GV64, s);
}

// everything of an iteration after the V update: V^T, V^T V, X V, U (update or, at iteration 0, bookkeeping), U^T, U^T U, X^T U, MAE
static int after_v(const bmf_wnmf_real_state* st, int u_mode, hipStream_t s) {
    const int kp = st->kp;
    const bool tiled = st->Xtiled && st->XTtiled;
    const bool fused = fused_update_ok(st);   // the V update has then left V's fragment order and V^T V already
    // the factor operand of the contraction: fragment order for the tiled kernels, plain transpose otherwise
    if (!fused) {
        if (tiled) BMF_TRY(bmf_frag_f32_launch(st->V, st->n_pad, kp, st->VT, st->stop, s));
        else BMF_TRY(transpose(st->V, st->n_pad, kp, st->VT, st->stop, s));
        BMF_TRY(gram(st, false, s));
    }
    // (with the factors' bf16 x 3 orders -- UT3, VT3, written by the fused update -- the contractions run on the bf16 matrix instruction)
    const bool bf3 = fused && st->UT3 && st->VT3;
    if (bf3) BMF_TRY(bmf_xf_f32_bf3_launch(st->Xtiled, st->m_pad, st->n_pad, st->VT3, st->Mslab, st->m_pad * kp, st->splits_xv, st->stop, s));
    else
    BMF_TRY(bmf_xf_f32_launch(tiled ? st->Xtiled : st->X, st->m_pad, st->n_pad, st->n_pad, st->VT, st->n_pad, kp, st->Mslab, st->m_pad * kp,
                              st->splits_xv, tiled, tiled, st->stop, s));
    // the residual sums of (U, V) ride in the X^T U pass when they can (X is then read twice per iteration, not three times)
    const bool fuse_resid = tiled && st->with_mae && st->Urf && kp == 32;
    if (fused) BMF_TRY(fused_update(st, true, u_mode, fuse_resid, s));   // U, both orders of it, zeroed sums, U^T U
    else {
        BMF_TRY(epilogue(st, true, u_mode, s));
        if (fuse_resid) BMF_TRY(bmf_frag_pair_launch(st->U, st->m_pad, st->UT, (uint32_t*)st->Urf, st->sums, st->stop, s));   // both orders of U + zeroed sums: one launch
        else if (tiled) BMF_TRY(bmf_frag_f32_launch(st->U, st->m_pad, kp, st->UT, st->stop, s));
        else BMF_TRY(transpose(st->U, st->m_pad, kp, st->UT, st->stop, s));
        BMF_TRY(gram(st, true, s));
    }
    if (fuse_resid) {
        BMF_TRY(bmf_xf_f32_resid_launch(st->XTtiled, st->n_pad, st->m_pad, st->UT, (const uint32_t*)st->Urf, st->V, kp, st->Nslab, st->n_pad * kp, st->splits_xtu, st->sums,
                                        st->stop, s, bf3 ? st->UT3 : nullptr));
        return BMF_OK;
    }
    if (bf3) BMF_TRY(bmf_xf_f32_bf3_launch(st->XTtiled, st->n_pad, st->m_pad, st->UT3, st->Nslab, st->n_pad * kp, st->splits_xtu, st->stop, s));
    else
    BMF_TRY(bmf_xf_f32_launch(tiled ? st->XTtiled : st->XT, st->n_pad, st->m_pad, st->m_pad, st->UT, st->m_pad, kp, st->Nslab, st->n_pad * kp,
                              st->splits_xtu, tiled, tiled, st->stop, s));
    if (st->with_mae) {
        BMF_LAUNCH(zero_sums_kernel, dim3(1), dim3(64), 0, s, st->sums, st->stop);
        if (tiled) {
            BMF_TRY(bmf_frag_rows_f32_launch(st->V, st->n_pad, kp, st->Vrf, st->stop, s));
            BMF_TRY(bmf_residual_tiled_launch(st->Xtiled, st->m_pad, st->n_pad, st->U, st->Vrf, kp, st->sums, st->stop, s));
        }
        else BMF_TRY(bmf_residual_launch_f32(st->X, st->m_pad, st->n_pad, st->m, st->n, st->U, st->V, kp, st->sums, st->stop, s));
    }
    return BMF_OK;
}

// the V side of an iteration: the shared epilogue, or the fused update that also leaves V's fragment order and V^T V
static int update_v(const bmf_wnmf_real_state* st, int mode, hipStream_t s) {
    if (fused_update_ok(st)) return fused_update(st, false, mode, false, s);
    return epilogue(st, false, mode, s);
}

extern "C" int bmf_wnmf_real_prepare(const bmf_wnmf_real_state* st, void* stream) {
    BMF_TRY(check_real_state(st, "bmf_wnmf_real_prepare"));
    hipStream_t s = (hipStream_t)stream;
    BMF_TRY(update_v(st, BMF_MODE_PREPARE, s));          // shadows of the initial V
    BMF_TRY(after_v(st, BMF_MODE_PREPARE, s));           // <U0, X V0>, the Grams, X^T U0, MAE of the initial state
    BMF_LAUNCH(real_finalize_kernel, dim3(1), dim3(1024), 0, s, *st, 0, 0);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_wnmf_real_run(const bmf_wnmf_real_state* st, int32_t iter0, int32_t iter1, int32_t max_iter, void* stream) {
    BMF_TRY(check_real_state(st, "bmf_wnmf_real_run"));
    BMF_REQUIRE(iter0 >= 1 && iter1 >= iter0 && iter1 <= st->log_rows, "bmf_wnmf_real_run: bad iteration range [%d,%d) for %d log rows", iter0, iter1,
                st->log_rows);
    hipStream_t s = (hipStream_t)stream;
    for (int it = iter0; it < iter1; ++it) {
        BMF_TRY(update_v(st, BMF_MODE_WNMF, s));          // V <- V o (X^T U) / (V (U^T U))          WNMF.py:98-101
        BMF_TRY(after_v(st, BMF_MODE_WNMF, s));           // U <- U o (X V) / (U (V^T V)) with the new V  :105-108
        BMF_LAUNCH(real_finalize_kernel, dim3(1), dim3(1024), 0, s, *st, it, (int)max_iter);
        BMF_LAUNCH_CHECK();
    }
    return BMF_OK;
}
