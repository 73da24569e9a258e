// proximal (PALM / iPALM) factor steps of ELBMF and PRIMP on the same contractions as the multiplicative
// update -- the gradient (F G^T - X) G = F (G^T G) - X G comes from the bits GEMM (X G, slabs) and the k x k Gram of the
// other factor; only the element-wise epilogue differs.
//
//   ELBMF  PyBMF/models/ELBMF.py:177-210   update_U: L = max(||G^T G||_2, 1e-4); eta = 1/(1.1 L) (beta = 0) or
//          2 (1 - beta) / (1 + 2 beta) / L; Fe = F + beta (F - F_before); Fn = prox(Fe - eta grad(Fe), l1 eta, l2 eta),
//          prox = elastic-net prox towards {0, 1}, negatives -> 0; get_integrality_gap :166-174.
//   PRIMP  PyBMF/models/PRIMP.py:51-88     elbmf_step_ipalm: L = max(||G^T G||_F, 1e-4), the same step, then proxelbmfnn
//          (max 0) followed by _proxelbmfnn (min 1); the inertial term extrapolates from a fixed anchor (the reference's
//          loop never advances Uold).
//
// bmf_sym_norms: spectral and Frobenius norm of the (symmetric positive semi-definite) k x k Gram, on the device, so that the
// step size never visits the host.  Spectral norm = repeated squaring (A <- A^2 / tr(A^2), 32 times: the dominant
// eigen-direction is amplified 2^32-fold) followed by a Rayleigh quotient of the original matrix; the quotient's error is
// second order in what is left of the other directions, <= |l1 - l2| (l2/l1)^(2^33), i.e. below 1e-10 relative for any gap.
#include "common.h"

namespace {

constexpr int NORM_SQUARINGS = 32;

// one workgroup, 256 threads, each owns a 4 x 4 tile of the 64 x 64 (or 2 x 2 of 32 x 32) matrix; all fp64 in LDS.
// Every iterate is symmetric, so both operands of a tile come from ROW q of the current matrix (A[i][q] = A[q][i]): two aligned
// 32-byte reads per reduction index instead of eight scattered 8-byte ones.  The squarings stop as soon as the iterate is rank one
// to machine precision: it is kept at trace 1, so tr(A^2) = ||A||_F^2 = 1 exactly then, and further squarings would reproduce it
// (a Gram matrix with a 1 % gap between its two largest eigenvalues gets there in 12).  Per squaring ~2 us (1024 fp64 FMAs per
// thread); round 2's version -- eight LDS reads per index, a 10-barrier tree for the trace, always 32 squarings -- took 228 us per
// call, a third of an ELBMF iteration at the headline shape.
template <int KP>
__device__ __forceinline__ void sym_norms_body(const double* __restrict__ G, double* __restrict__ out) {
    constexpr int TS = KP / 16;  // tile side per thread
    __shared__ __attribute__((aligned(32))) double A[KP][KP], B[KP][KP], G0[KP][KP];   // G0: the normalised input, kept for the quotient
    __shared__ double red[2][4];
    __shared__ int jmax_s;
    const int t = threadIdx.x, ti = (t >> 4) * TS, tj = (t & 15) * TS;
    const int wave = t >> 6;

    // sum over the block of two values at once: wave reduction, then the four waves through LDS (fixed order)
    auto block_sum2 = [&](double& v0, double& v1) {
        v0 = wave_sum(v0);
        v1 = wave_sum(v1);
        __syncthreads();   // (the previous call's readers are done)
        if ((t & 63) == 0) { red[0][wave] = v0; red[1][wave] = v1; }
        __syncthreads();
        v0 = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        v1 = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
    };

    double fro = 0.0, tr = 0.0;
    for (int e = t; e < KP * KP; e += 256) {
        const double g = G[e];
        A[e / KP][e % KP] = g;
        fro += g * g;
        if (e / KP == e % KP) tr += g;
    }
    block_sum2(fro, tr);
    if (!(tr > 0.0)) {  // the zero matrix (an all-zero factor)
        if (t == 0) {
            out[0] = 0.0;
            out[1] = sqrt(fro);
        }
        return;
    }
    // symmetrised on the way in (G is a Gram matrix; its fp64 slab sums are symmetric already, this makes the row trick safe for any input)
    double sym[TS][TS];
#pragma unroll
    for (int a = 0; a < TS; ++a)
#pragma unroll
        for (int b = 0; b < TS; ++b) sym[a][b] = 0.5 * (A[ti + a][tj + b] + A[tj + b][ti + a]) / tr;
    __syncthreads();
#pragma unroll
    for (int a = 0; a < TS; ++a)
#pragma unroll
        for (int b = 0; b < TS; ++b) {
            A[ti + a][tj + b] = sym[a][b];
            G0[ti + a][tj + b] = sym[a][b];
        }
    __syncthreads();

    double (*cur)[KP] = A, (*nxt)[KP] = B;
    int n_sq = 0;
    for (int it = 0; it < NORM_SQUARINGS; ++it) {
        ++n_sq;
        double c[TS][TS];
#pragma unroll
        for (int a = 0; a < TS; ++a)
#pragma unroll
            for (int b = 0; b < TS; ++b) c[a][b] = 0.0;
#pragma unroll 4
        for (int q = 0; q < KP; ++q) {
            double x[TS], y[TS];
#pragma unroll
            for (int a = 0; a < TS; ++a) x[a] = cur[q][ti + a];   // = cur[ti + a][q]
#pragma unroll
            for (int b = 0; b < TS; ++b) y[b] = cur[q][tj + b];
#pragma unroll
            for (int a = 0; a < TS; ++a)
#pragma unroll
                for (int b = 0; b < TS; ++b) c[a][b] = fma(x[a], y[b], c[a][b]);
        }
        double d = 0.0, unused = 0.0;
#pragma unroll
        for (int a = 0; a < TS; ++a)
#pragma unroll
            for (int b = 0; b < TS; ++b)
                if (ti + a == tj + b) d += c[a][b];
        block_sum2(d, unused);
        const double trace = d;  // > 0: the squared matrix of a non-zero symmetric matrix has a positive trace
        const double inv = 1.0 / trace;
        // the products c[a][b] and c[b][a] of the mirrored tile are sums of the same terms in the same order: nxt is symmetric bit for bit
#pragma unroll
        for (int a = 0; a < TS; ++a)
#pragma unroll
            for (int b = 0; b < TS; ++b) nxt[ti + a][tj + b] = c[a][b] * inv;
        __syncthreads();
        double (*tmp)[KP] = cur;
        cur = nxt;
        nxt = tmp;
        // tr(A^2) = 1 - 2 eps + O(eps^2) at trace 1, eps = the weight of the other directions in the iterate that was squared; the one
        // just written has eps^2, its dominant column is off the eigenvector by O(eps^2) and the quotient below by O(eps^4): at
        // eps <= 5e-6 that is 1e-21 relative.  (Waiting for tr(A^2) >= 1 - 1e-14 cost one squaring more for nothing.)  Block-uniform.
        if (trace >= 1.0 - 1.0e-5) break;
    }
    // v = the column of the amplified matrix with the largest diagonal entry (first wave: lane j looks at entry j, arg-max by
    // shuffles -- one thread walking the diagonal was 64 dependent LDS round trips); Rayleigh quotient of the ORIGINAL matrix, from the
    // normalised copy in LDS (times the trace it was divided by)
    if (t < 64) {
        double dv = t < KP ? cur[t][t] : -1.0;
        int dj = t;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_xor(dv, o, 64);
            const int oj = __shfl_xor(dj, o, 64);
            if (ov > dv || (ov == dv && oj < dj)) { dv = ov; dj = oj; }
        }
        if (t == 0) jmax_s = dj;
    }
    __syncthreads();
    const int jm = jmax_s;
    double num = 0.0, den = 0.0;
    if (t < KP) {
        double w = 0.0;
#pragma unroll 8
        for (int q = 0; q < KP; ++q) w = fma(G0[q][t], cur[q][jm], w);   // (G0 is symmetric: row t read as column t, conflict-free)
        num = cur[t][jm] * w;
        den = cur[t][jm] * cur[t][jm];
    }
    block_sum2(num, den);
    if (t == 0) {
        out[0] = tr * (num / den);
        out[1] = sqrt(fro);
#ifdef BMF_EXP_NORM_COUNT   // diagnostic build: the number of squarings instead of the Frobenius norm
        out[1] = (double)n_sq;
#endif
    }
    (void)n_sq;
}
template <int KP>
__global__ __launch_bounds__(256) void sym_norms_kernel(const double* __restrict__ G, double* __restrict__ out) {
    sym_norms_body<KP>(G, out);
}
// two matrices in one launch (one workgroup each): the Grams of both factors of an ELBMF iteration
template <int KP>
__global__ __launch_bounds__(256) void sym_norms2_kernel(const double* __restrict__ G0, double* __restrict__ out0, const double* __restrict__ G1,
                                                          double* __restrict__ out1) {
    if (blockIdx.x == 0) sym_norms_body<KP>(G0, out0);
    else sym_norms_body<KP>(G1, out1);
}

__device__ __forceinline__ double sgn(double x) { return (double)((x > 0.0) - (x < 0.0)); }
// proxelbmf (PRIMP.py:55-56) = the inner expression of prox (ELBMF.py:203-208)
__device__ __forceinline__ double prox_core(double x, double kai, double lam) {
    const double p = x <= 0.5 ? x - kai * sgn(x) : x - kai * sgn(x - 1.0) + lam;
    return p / (1.0 + lam);
}

// One block = 128 rows = 4 waves x 32 rows, same tiling as mu_epilogue_kernel (epilogue.hip): Fe G on the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32), the element-wise step in fp64 in the C/D layout of the product.
template <int NT>
__global__ __launch_bounds__(256) void palm_epilogue_kernel(bmf_palm_args a) {
    if (a.stop && *a.stop != 0) return;
    constexpr int KP = 32 * NT;
    constexpr int KH = KP / 2;
    __shared__ double red[4][2];
    __shared__ float cmax[4][KP];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 32;
    const double beta = a.beta;

    // step size from the norm of the other factor's Gram (ELBMF.py:184-185, PRIMP.py:73-80)
    const double L = fmax(a.norms[a.norm_kind == BMF_NORM_SPECTRAL ? 0 : 1], 1e-4);
    const double eta = beta == 0.0 ? 1.0 / (1.1 * L) : 2.0 * (1.0 - beta) / (1.0 + 2.0 * beta) / L;
    const double kai = a.l1 * eta, lam = a.l2 * eta;

    // ---- Fe G: A = (float)Fe[row c][KH h + s] from the fp64 masters, B = G[KH h + s][32 nt + c] ----
    f32x16 fg[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) fg[nt][i] = 0.f;
    if (!a.den) {
        const double* fp = a.F64 + (row0 + c) * KP + KH * h;
        const double* pp = a.Fprev64 + (row0 + c) * KP + KH * h;
#pragma unroll
        for (int s0 = 0; s0 < KH; s0 += 8) {
            float av[8], gv[NT][8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const double f = fp[s0 + s];
                av[s] = (float)(beta == 0.0 ? f : f + beta * (f - pp[s0 + s]));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) gv[nt][s] = a.G[(KH * h + s0 + s) * KP + 32 * nt + c];
            }
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) fg[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], gv[nt][s], fg[nt], 0, 0, 0);
        }
    }

    double gap_acc = 0.0;
    unsigned colword[NT];
    float cm[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        colword[nt] = 0u;
        cm[nt] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int rl = (i & 3) + 8 * (i >> 2) + 4 * h;
        const int64_t r = row0 + rl;
        const bool row_ok = r < a.rows;
        unsigned long long ball[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = 32 * nt + c;
            const bool ok = row_ok && col < a.k;
            const int64_t idx = r * KP + col;
            const double f = a.F64[idx];
            const double p = a.Fprev64[idx];
            float num = 0.f;
            for (int sp = 0; sp < a.splits; ++sp) num += a.num[(int64_t)sp * a.slab_stride + idx];
            const double fe = beta == 0.0 ? f : f + beta * (f - p);
            // all-ones mask: Fe (G^T G) - X G;  a mask / weight matrix: (W o (Fe G^T)) G - (W o X) G from the masked pass (a.den, a.num)
            const double grad = (a.den ? (double)a.den[idx] : (double)fg[nt][i]) - (double)num;
            double x = fe - eta * grad;
            double fn;
            if (a.variant == BMF_PALM_ELBMF) {
                fn = prox_core(x, kai, lam);
                if (fn < 0.0) fn = 0.0;
            } else {  // PRIMP: proxelbmfnn then _proxelbmfnn
                x = fmax(prox_core(x, kai, lam), 0.0);
                fn = fmin(prox_core(x, kai, lam), 1.0);
            }
            if (!ok) fn = 0.0;
            const float fn32 = (float)fn;
            a.F64[idx] = fn;
            if (a.advance_prev) a.Fprev64[idx] = ok ? f : 0.0;  // ELBMF: the caller's F_last becomes the old current factor
            a.F[idx] = fn32;
            cm[nt] = fmaxf(cm[nt], fabsf(fn32));
            if (ok) {
                const double dist = fn < 0.5 ? fabs(fn) : fabs(fn - 1.0);
                gap_acc += a.gap_l1 * dist + a.gap_l2 * dist * dist;
            }
            const bool bit = ok && (fn > (double)a.thr);
            ball[nt] = __ballot(bit);
            colword[nt] |= (bit ? 1u : 0u) << rl;
        }
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
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const unsigned w = colword[nt] | __shfl_xor(colword[nt], 32, 64);
        if (h == 0) a.colbits[(int64_t)(32 * nt + c) * a.ldcb + (row0 >> 5)] = w;
    }
    const double gs = wave_sum(gap_acc);
    if (lane == 0) red[wave][0] = gs;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float mx = fmaxf(cm[nt], __shfl_xor(cm[nt], 32, 64));
        if (h == 0) cmax[wave][32 * nt + c] = mx;
    }
    __syncthreads();
    if (a.blockmax && threadIdx.x < KP)
        a.blockmax[(int64_t)blockIdx.x * KP + threadIdx.x] =
            fmaxf(fmaxf(cmax[0][threadIdx.x], cmax[1][threadIdx.x]), fmaxf(cmax[2][threadIdx.x], cmax[3][threadIdx.x]));
    if (threadIdx.x == 0) a.partials[blockIdx.x] = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
}

// The same step (beta = 0, all-ones mask) in the shape of mu_epilogue_i8_kernel (epilogue.hip): the 16 rows of a lane walked in eight
// chunks of two through a ring of four register sets, loads issued three chunks ahead, and the int8 digit planes of the new factor
// emitted from the fp64 values with the PREDICTED column scale (checked afterwards, bmf_colscale_i8_fused_block).  The first form
// above makes sixteen dependent memory round trips per lane -- stores to F64 between the loads of F64, so nothing can be hoisted --
// and took ~100 us for ANY number of rows (V at the headline shape: 158 blocks, 100 us); with beta = 0 the extrapolated point is the
// factor itself, so the F G operand comes from the fp32 shadow, and the previous iterate is only written.
// HASBETA: the inertial form -- the previous iterate rides in the ring too, and the F G operand is the extrapolated point, converted
// from the two fp64 masters.  `planes`, `blockmax` and `dotpart` are optional at run time (PRIMP and the stand-alone step leave them out).
template <int NT, bool HASBETA>
__global__ __launch_bounds__(256, 2) void palm_epilogue_i8_kernel(bmf_palm_args a) {
    if (a.stop && *a.stop != 0) return;
    constexpr int KP = 32 * NT;
    __shared__ double red[4][2];
    __shared__ float cmax[4][KP];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int k = a.k;
    // block -> 128-row block, XCD-aware as in mu_epilogue_i8_kernel (the four blocks of a 512-row group write the four quarters of
    // every 64-byte piece of the planes: they go to ids that share an XCD)
    const int nblk_all = (int)gridDim.x;
    const int bid = (int)blockIdx.x;
    const int blk = bid < (nblk_all & ~31) ? (((bid >> 5) * 8 + (bid & 7)) << 2) + ((bid >> 3) & 3) : bid;
    const int64_t row0 = (int64_t)blk * 128 + wave * 32;

    const double L = fmax(a.norms[a.norm_kind == BMF_NORM_SPECTRAL ? 0 : 1], 1e-4);
    const double beta = HASBETA ? a.beta : 0.0;
    const double eta = HASBETA ? 2.0 * (1.0 - beta) / (1.0 + 2.0 * beta) / L : 1.0 / (1.1 * L);
    const double kai = a.l1 * eta, lam = a.l2 * eta;
    const bool primp = a.variant != BMF_PALM_ELBMF;
    const bool advance = a.advance_prev != 0;

    constexpr int KH = KP / 2;
    float av[KH], gv[NT][KH];
    {
        if constexpr (HASBETA) {
            const double* fp = a.F64 + (row0 + c) * KP + KH * h;
            const double* pp = a.Fprev64 + (row0 + c) * KP + KH * h;
#pragma unroll
            for (int s = 0; s < KH; s += 4) {   // (four at a time: left alone the compiler loads all 2 KH doubles before it converts one)
#pragma unroll
                for (int q = 0; q < 4; ++q) av[s + q] = (float)(fp[s + q] + beta * (fp[s + q] - pp[s + q]));
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            const float* ap = a.F + (row0 + c) * KP + KH * h;
#pragma unroll
            for (int s = 0; s < KH; s += 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(ap + s);
                av[s] = v[0]; av[s + 1] = v[1]; av[s + 2] = v[2]; av[s + 3] = v[3];
            }
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
    double psc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) psc[nt] = a.planes ? (double)a.plane_scale[32 * nt + c] : 0.0;

    double gap_acc = 0.0, dot_acc = 0.0;
    unsigned colword[NT];
    float cm[NT];
    unsigned seg[3][NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        colword[nt] = 0u;
        cm[nt] = 0.f;
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int j = 0; j < 4; ++j) seg[l][nt][j] = 0u;
    }
    const unsigned loff = (unsigned)(4 * h * KP + c);
    constexpr int CR = 2;
    auto load_chunk = [&](int q2, double (&fv)[CR][NT], float (&nv)[CR][NT], double (&pv)[HASBETA ? CR : 1][NT]) {
#pragma unroll
        for (int jj = 0; jj < CR; ++jj) {
            const int i = CR * q2 + jj;
            const double* fq = a.F64 + (row0 + 8 * (i >> 2)) * KP;
            const double* pq = a.Fprev64 + (row0 + 8 * (i >> 2)) * KP;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                fv[jj][nt] = fq[loff + (i & 3) * KP + 32 * nt];
                if constexpr (HASBETA) pv[jj][nt] = pq[loff + (i & 3) * KP + 32 * nt];
                nv[jj][nt] = 0.f;
            }
        }
        for (int sp = 0; sp < a.splits; ++sp) {
#pragma unroll
            for (int jj = 0; jj < CR; ++jj) {
                const int i = CR * q2 + jj;
                const float* nq = a.num + (int64_t)sp * a.slab_stride + (row0 + 8 * (i >> 2)) * KP;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) nv[jj][nt] += nq[loff + (i & 3) * KP + 32 * nt];
            }
        }
    };
    auto do_chunk = [&](int q2, const double (&fv)[CR][NT], const float (&nv)[CR][NT], const double (&pv)[HASBETA ? CR : 1][NT]) {
#pragma unroll
        for (int jj = 0; jj < CR; ++jj) {
            const int i = CR * q2 + jj;
            const int j = i & 3, q = i >> 2;
            const int rl = j + 8 * q + 4 * h;
            const int64_t r = row0 + rl;
            const bool row_ok = r < a.rows;
            double* fq = a.F64 + (row0 + 8 * q) * KP;
            double* pq = a.Fprev64 + (row0 + 8 * q) * KP;
            float* sq = a.F + (row0 + 8 * q) * KP;
            unsigned long long ball[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = 32 * nt + c;
                const bool ok = row_ok && col < k;
                const unsigned eoff = loff + j * KP + 32 * nt;
                const double f = fv[jj][nt];
                const double grad = (double)fg[nt][i] - (double)nv[jj][nt];
                double fe = f;
                if constexpr (HASBETA) fe = f + beta * (f - pv[jj][nt]);
                double x = fe - eta * grad;
                double fn;
                if (!primp) {
                    fn = prox_core(x, kai, lam);
                    if (fn < 0.0) fn = 0.0;
                } else {
                    x = fmax(prox_core(x, kai, lam), 0.0);
                    fn = fmin(prox_core(x, kai, lam), 1.0);
                }
                if (!ok) fn = 0.0;
                const float fn32 = (float)fn;
                fq[eoff] = fn;
                if (advance) pq[eoff] = ok ? f : 0.0;
                sq[eoff] = fn32;
                cm[nt] = fmaxf(cm[nt], fabsf(fn32));
                if (ok) {
                    const double dist = fn < 0.5 ? fabs(fn) : fabs(fn - 1.0);
                    gap_acc += a.gap_l1 * dist + a.gap_l2 * dist * dist;
                }
                const bool bit = ok && (fn > (double)a.thr);
                ball[nt] = __ballot(bit);
                colword[nt] |= (bit ? 1u : 0u) << rl;
                // digits of q = rint(fn 2^e), balanced base 256: byte (i >> 2) of dword (i & 3) of this lane's segment (epilogue.hip)
                // (no planes asked for: the scale is 0 and the digits are zeros nobody stores)
                const int qi = __double2int_rn(fmax(fmin(fn * psc[nt], 8355711.0), -8355711.0));
                const int d0 = ((qi + 128) & 255) - 128;
                const int q1 = (qi - d0) >> 8;
                const int d1 = ((q1 + 128) & 255) - 128;
                const int d2 = (q1 - d1) >> 8;
                seg[0][nt][j] |= (unsigned)(d0 & 255) << (8 * q);
                seg[1][nt][j] |= (unsigned)(d1 & 255) << (8 * q);
                seg[2][nt][j] |= (unsigned)(d2 & 255) << (8 * q);
                {   // <F, X G> of the state this step starts from (padding: f = 0).  Product and sum are pinned: left alone the compiler keeps
                    // the products of a whole chunk alive to add them later (256 registers + 46 spilled against 186)
                    double pr_ = f * (double)nv[jj][nt];
                    asm volatile("" : "+v"(pr_));
                    dot_acc += pr_;
                    asm volatile("" : "+v"(dot_acc));
                }
                __builtin_amdgcn_sched_barrier(0);
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
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    {
        double fr[4][CR][NT], pr[4][HASBETA ? CR : 1][NT];
        float nr[4][CR][NT];
        load_chunk(0, fr[0], nr[0], pr[0]);
        load_chunk(1, fr[1], nr[1], pr[1]);
        load_chunk(2, fr[2], nr[2], pr[2]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KH; ++s)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) fg[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], gv[nt][s], fg[nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q2 = 0; q2 < 16 / CR; ++q2) {
            if (q2 + 3 < 16 / CR) load_chunk(q2 + 3, fr[(q2 + 3) & 3], nr[(q2 + 3) & 3], pr[(q2 + 3) & 3]);
            __builtin_amdgcn_sched_barrier(0);
            do_chunk(q2, fr[q2 & 3], nr[q2 & 3], pr[q2 & 3]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (a.planes) {
        const int g = blk & 3;
        const int64_t blk512 = ((int64_t)blk >> 2) << 9;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const u32x4 v = {seg[l][nt][0], seg[l][nt][1], seg[l][nt][2], seg[l][nt][3]};
                *reinterpret_cast<u32x4*>(a.planes + (int64_t)(l * KP + 32 * nt + c) * a.ldp + blk512 + 128 * wave + (4 * h + g) * 16) = v;
            }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const unsigned w = colword[nt] | __shfl_xor(colword[nt], 32, 64);
        if (h == 0) a.colbits[(int64_t)(32 * nt + c) * a.ldcb + (row0 >> 5)] = w;
    }
    const double gs = wave_sum(gap_acc);
    const double ds = wave_sum(dot_acc);
    if (lane == 0) {
        red[wave][0] = gs;
        red[wave][1] = ds;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float mx = fmaxf(cm[nt], __shfl_xor(cm[nt], 32, 64));
        if (h == 0) cmax[wave][32 * nt + c] = mx;
    }
    __syncthreads();
    if (a.blockmax && threadIdx.x < KP)
        a.blockmax[(int64_t)blk * KP + threadIdx.x] =
            fmaxf(fmaxf(cmax[0][threadIdx.x], cmax[1][threadIdx.x]), fmaxf(cmax[2][threadIdx.x], cmax[3][threadIdx.x]));
    if (threadIdx.x == 0) {
        a.partials[blk] = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
        if (a.dotpart) a.dotpart[blk] = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
    }
}

// out[0] = sum part[0..n): the cross term <U, X V> of a log row, completed by the U step of the NEXT iteration (or by
// bmf_palm_finish_row from dot_slabs partials)
__global__ __launch_bounds__(256) void palm_dot_finish_kernel(const double* __restrict__ part, int n, double* __restrict__ out) {
    __shared__ double red[4];
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) a += part[i];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + red[2]) + red[3];
}

// out[e] = (float)(F[e] + beta (F[e] - Fprev[e])): the point at which an inertial step evaluates its gradient, as the fp32 operand
// of the masked pass
__global__ __launch_bounds__(256) void palm_extrapolate_kernel(const double* __restrict__ F64, const double* __restrict__ Fprev64, double beta,
                                                                int64_t n, float* __restrict__ out) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const double f = F64[e];
        out[e] = (float)(beta == 0.0 ? f : f + beta * (f - Fprev64[e]));
    }
}

// partial[b] = sum over block b of F64[e] * (sum_s slabs[s][e]): <F, X G> for the trace form of ||X - U V^T||^2
__global__ __launch_bounds__(256) void dot_slabs_kernel(const double* __restrict__ F64, const float* __restrict__ slabs,
                                                         int64_t stride, int splits, int64_t n, double* __restrict__ partial) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        float s = 0.f;
        for (int sp = 0; sp < splits; ++sp) s += slabs[(int64_t)sp * stride + e];
        acc += F64[e] * (double)s;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// out[0] = sum dotpart, out[1] = <GU, GV>, out[2] = sum partU, out[3] = sum partV, out[4], out[5] = counts[0], counts[1] (which are
// reset): the scalars of one PALM iteration gathered by ONE launch (they were five torch reductions and a fill: ~40 us of stream time)
__global__ __launch_bounds__(256) void palm_scalars_kernel(const double* __restrict__ dotpart, int nd, const double* __restrict__ GU,
                                                            const double* __restrict__ GV, int kk, const double* __restrict__ partU, int nu,
                                                            const double* __restrict__ partV, int nv, unsigned long long* __restrict__ counts,
                                                            double* __restrict__ out) {
    __shared__ double red[4][4];
    const int t = threadIdx.x;
    double a = 0.0, b = 0.0, c = 0.0, d = 0.0;
    for (int i = t; i < nd; i += 256) a += dotpart[i];
    // (eight loads in flight: 16 dependent trips per thread for a 64 x 64 Gram pair were most of this kernel's 11 us)
    int i = t;
    for (; i + 7 * 256 < kk; i += 8 * 256) {
        double gu[8], gv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { gu[q] = GU[i + 256 * q]; gv[q] = GV[i + 256 * q]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) b += gu[q] * gv[q];
    }
    for (; i < kk; i += 256) b += GU[i] * GV[i];
    for (i = t; i + 3 * 256 < nu; i += 4 * 256) c += (partU[i] + partU[i + 256]) + (partU[i + 512] + partU[i + 768]);
    for (; i < nu; i += 256) c += partU[i];
    for (i = t; i < nv; i += 256) d += partV[i];
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c); d = wave_sum(d);
    if ((t & 63) == 0) { red[0][t >> 6] = a; red[1][t >> 6] = b; red[2][t >> 6] = c; red[3][t >> 6] = d; }
    __syncthreads();
    if (t < 4) out[t] = ((red[t][0] + red[t][1]) + red[t][2]) + red[t][3];
    if (t == 0 && counts) {
        out[4] = (double)counts[0];
        out[5] = (double)counts[1];
        counts[0] = 0ull;
        counts[1] = 0ull;
    }
}

}  // namespace

extern "C" int bmf_palm_scalars(const double* dotpart, int nd, const double* GU64, const double* GV64, int kk, const double* partU, int nu,
                                const double* partV, int nv, unsigned long long* counts, double* out, void* stream) {
    BMF_REQUIRE(dotpart && GU64 && GV64 && partU && partV && out, "bmf_palm_scalars: null pointer");
    BMF_REQUIRE(nd >= 1 && kk >= 1 && nu >= 1 && nv >= 1, "bmf_palm_scalars: bad lengths");
    BMF_LAUNCH(palm_scalars_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, dotpart, nd, GU64, GV64, kk, partU, nu, partV, nv, counts, out);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_sym_norms(const double* G64, int kp, double* out, void* stream) {
    BMF_REQUIRE(G64 && out, "bmf_sym_norms: null pointer");
    BMF_REQUIRE(kp == 32 || kp == 64, "bmf_sym_norms: kp must be 32 or 64");
    if (kp == 32) BMF_LAUNCH(sym_norms_kernel<32>, dim3(1), dim3(256), 0, (hipStream_t)stream, G64, out);
    else BMF_LAUNCH(sym_norms_kernel<64>, dim3(1), dim3(256), 0, (hipStream_t)stream, G64, out);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_palm_epilogue(const bmf_palm_args* a, void* stream) {
    BMF_REQUIRE(a, "bmf_palm_epilogue: null args");
    BMF_REQUIRE(a->F64 && a->Fprev64 && a->F && a->num && (a->G || a->den) && a->norms && a->rowbits && a->colbits && a->partials,
                "bmf_palm_epilogue: null pointer");
    BMF_REQUIRE(!a->den || a->splits == 1, "bmf_palm_epilogue: with den, num is one array (splits must be 1)");
    BMF_REQUIRE(a->rows_pad > 0 && a->rows_pad % 128 == 0, "bmf_palm_epilogue: rows_pad must be a multiple of 128");
    BMF_REQUIRE(a->rows >= 1 && a->rows <= a->rows_pad, "bmf_palm_epilogue: rows out of range");
    BMF_REQUIRE((a->kp == 32 || a->kp == 64) && a->k >= 1 && a->k <= a->kp, "bmf_palm_epilogue: need 1 <= k <= kp, kp in {32,64}");
    BMF_REQUIRE(a->splits >= 1 && a->slab_stride >= a->rows_pad * a->kp, "bmf_palm_epilogue: bad slab description");
    BMF_REQUIRE(a->variant == BMF_PALM_ELBMF || a->variant == BMF_PALM_PRIMP, "bmf_palm_epilogue: variant must be BMF_PALM_ELBMF or _PRIMP");
    BMF_REQUIRE(a->norm_kind == BMF_NORM_SPECTRAL || a->norm_kind == BMF_NORM_FROBENIUS, "bmf_palm_epilogue: bad norm_kind");
    BMF_REQUIRE(a->beta >= 0.0 && a->beta < 1.0, "bmf_palm_epilogue: beta must be in [0, 1)");
    BMF_REQUIRE(a->ldcb >= a->rows_pad / 32, "bmf_palm_epilogue: ldcb too small");
    dim3 grid((unsigned)(a->rows_pad / 128)), block(256);
    if (a->planes) {
        BMF_REQUIRE(!a->den && a->blockmax && a->plane_scale, "bmf_palm_epilogue: planes need blockmax and plane_scale, and exclude den");
        BMF_REQUIRE(a->rows_pad % 512 == 0 && a->ldp >= a->rows_pad && a->ldp % 16 == 0 && bmf_aligned16(a->planes),
                    "bmf_palm_epilogue: planes need rows_pad %% 512 == 0, ldp >= rows_pad, ldp %% 16 == 0, 16-byte alignment");
    }
    BMF_REQUIRE(!a->dotpart || !a->den, "bmf_palm_epilogue: dotpart is not available with den");
    if (!a->den) {   // the ring form (all-ones mask); the first form stays for the masked gradient
        BMF_REQUIRE(bmf_aligned16(a->F), "bmf_palm_epilogue: F must be 16-byte aligned");
        const bool hb = a->beta != 0.0;
        if (a->kp == 32) { if (hb) BMF_LAUNCH((palm_epilogue_i8_kernel<1, true>), grid, block, 0, (hipStream_t)stream, *a); else BMF_LAUNCH((palm_epilogue_i8_kernel<1, false>), grid, block, 0, (hipStream_t)stream, *a); }
        else { if (hb) BMF_LAUNCH((palm_epilogue_i8_kernel<2, true>), grid, block, 0, (hipStream_t)stream, *a); else BMF_LAUNCH((palm_epilogue_i8_kernel<2, false>), grid, block, 0, (hipStream_t)stream, *a); }
        BMF_LAUNCH_CHECK();
        return BMF_OK;
    }
    if (a->kp == 32) BMF_LAUNCH(palm_epilogue_kernel<1>, grid, block, 0, (hipStream_t)stream, *a);
    else BMF_LAUNCH(palm_epilogue_kernel<2>, grid, block, 0, (hipStream_t)stream, *a);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_palm_extrapolate(const double* F64, const double* Fprev64, double beta, int64_t n, float* out, void* stream) {
    BMF_REQUIRE(F64 && Fprev64 && out, "bmf_palm_extrapolate: null pointer");
    BMF_REQUIRE(n >= 1 && beta >= 0.0 && beta < 1.0, "bmf_palm_extrapolate: need n >= 1 and beta in [0, 1)");
    const int64_t blocks = (n + 255) / 256;
    BMF_LAUNCH(palm_extrapolate_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, F64, Fprev64, beta, n, out);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_dot_slabs(const double* F64, const float* slabs, int64_t stride, int splits, int64_t n, double* partial,
                             int blocks, void* stream) {
    BMF_REQUIRE(F64 && slabs && partial, "bmf_dot_slabs: null pointer");
    BMF_REQUIRE(splits >= 1 && n >= 1 && stride >= n && blocks >= 1 && blocks <= 65535, "bmf_dot_slabs: bad arguments");
    BMF_LAUNCH(dot_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, F64, slabs, stride, splits, n, partial);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

// ---- one ELBMF iteration per call (bmf_palm_state) ---------------------------------------------------------------------------------
int bmf_panel_i8_launch(const double* F64, const float* F, int64_t rows_pad, int64_t ldf, int kp, int limbs, int8_t* panel, int64_t ldp,
                        float* ws, float* scale, bool have_blockmax, const int32_t* stop, hipStream_t s, bool have_scale,
                        const float* flags, float* colscale_out, const float* rslabs, int rcount, int rn, float* rout32, double* rout64);
int bmf_gram_partial_launch(const float* F, int64_t rows_pad, int64_t ldf, int kp, float* slabs, int blocks, const float* blockmax, int limbs,
                            float* scale, const int32_t* stop, hipStream_t s, int fused);

// `fused`: the step emits the digit planes itself with the predicted column scales
static int palm_step(const bmf_palm_state* st, bool u_side, double l1, double l2, double gap_l1, double gap_l2, bool fused, void* stream,
                     int advance_prev = 1) {
    bmf_palm_args a{};
    a.F64 = u_side ? st->U64 : st->V64;
    a.Fprev64 = u_side ? st->Up64 : st->Vp64;
    a.F = u_side ? st->U : st->V;
    a.rows_pad = u_side ? st->m_pad : st->n_pad;
    a.rows = u_side ? st->m : st->n;
    a.k = st->k;
    a.kp = st->kp;
    a.splits = u_side ? st->splits_xv : st->splits_xtu;
    a.num = u_side ? st->Mslab : st->Nslab;
    a.slab_stride = a.rows_pad * st->kp;
    a.G = u_side ? st->GV : st->GU;
    a.norms = u_side ? st->normsV : st->normsU;
    a.norm_kind = st->norm_kind;
    a.variant = st->variant;
    a.beta = st->beta;
    a.l1 = l1; a.l2 = l2; a.gap_l1 = gap_l1; a.gap_l2 = gap_l2;
    a.advance_prev = advance_prev;
    a.thr = u_side ? st->thr_u : st->thr_v;
    a.rowbits = u_side ? st->ubits : st->vbits;
    a.colbits = u_side ? st->ucolbits : st->vcolbits;
    a.ldcb = a.rows_pad / 32;
    a.partials = u_side ? st->partU : st->partV;
    a.blockmax = nullptr;
    if (fused) {
        a.blockmax = u_side ? st->wsU : st->wsV;
        a.planes = u_side ? st->Upanel : st->Vpanel;
        a.ldp = a.rows_pad;
        a.plane_scale = u_side ? st->scaleU : st->scaleV;
        // (U side: the per-block <U, X V> partials of the state this step starts from go where the gap partials of the V side are
        // not: the tail of dotpart is unused by the fused path, which needs m_pad / 128 <= dot_blocks entries -- see palm_fused)
        if (u_side) a.dotpart = st->dotpart;
    }
    return bmf_palm_epilogue(&a, stream);
}

// planes, Gram (fp32 + fp64) of one factor.  Fused: the planes exist already -- the extra blocks of the Gram launch check the
// predicted column scales against the new maxima, and the conditional rebuild and the reduction of the Gram slabs are one launch
// (the sequence of the multiplicative-update loop, api.hip).
static int palm_derive(const bmf_palm_state* st, bool u_side, bool fused, void* stream) {
    const int kp = st->kp, kk = kp * kp;
    const int64_t rows_pad = u_side ? st->m_pad : st->n_pad;
    const double* F64 = u_side ? st->U64 : st->V64;
    const float* F = u_side ? st->U : st->V;
    int8_t* panel = u_side ? st->Upanel : st->Vpanel;
    float* ws = u_side ? st->wsU : st->wsV;
    float* scale = u_side ? st->scaleU : st->scaleV;
    float* G = u_side ? st->GU : st->GV;
    double* G64 = u_side ? st->GU64 : st->GV64;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (fused) {
        if ((rc = bmf_gram_partial_launch(F, rows_pad, kp, kp, st->gram_slabs, st->gram_blocks, ws, 3, scale, nullptr, s, 1)) != BMF_OK) return rc;
        return bmf_panel_i8_launch(F64, F, rows_pad, kp, kp, 3, panel, rows_pad, ws, scale + 2 * kp, true, nullptr, s, true, scale + 3 * kp, scale + kp,
                                   st->gram_slabs, st->gram_blocks, kk, G, G64);
    }
    if ((rc = bmf_make_panel_i8(F64, F, rows_pad, kp, kp, 3, panel, rows_pad, ws, scale, stream)) != BMF_OK) return rc;
    if ((rc = bmf_gram_partial(F, rows_pad, kp, kp, st->gram_slabs, st->gram_blocks, stream)) != BMF_OK) return rc;
    return bmf_reduce_slabs(st->gram_slabs, kk, st->gram_blocks, kk, G, G64, stream);
}

static bool palm_fused(const bmf_palm_state* st) {
    return st->m_pad / 128 <= st->dot_blocks;
}

static int palm_check_state(const bmf_palm_state* st, int it, const char* who, int variant = BMF_PALM_ELBMF) {
    BMF_REQUIRE(st, "%s: null state", who);
    BMF_REQUIRE(st->struct_bytes == (int32_t)sizeof(bmf_palm_state), "%s: struct_bytes=%d, library expects %d", who, st->struct_bytes,
                (int)sizeof(bmf_palm_state));
    BMF_REQUIRE(st->variant == variant, "%s: the state's variant is %d (ELBMF's loop: bmf_palm_iterate; PRIMP's: bmf_primp_iterate)", who, st->variant);
    BMF_REQUIRE(st->Xbits && st->Xtiled && st->XTtiled && st->U64 && st->V64 && st->Up64 && st->Vp64 && st->U && st->V && st->Upanel && st->Vpanel &&
                    st->scaleU && st->scaleV && st->wsU && st->wsV && st->Mslab && st->Nslab && st->gram_slabs && st->GU && st->GV && st->GU64 &&
                    st->GV64 && st->normsU && st->normsV && st->partU && st->partV && st->dotpart && st->ubits && st->vbits && st->ucolbits &&
                    st->vcolbits && st->counts && st->log,
                "%s: null pointer in the state", who);
    BMF_REQUIRE(st->m_pad > 0 && st->n_pad > 0 && st->m_pad % 512 == 0 && st->n_pad % 512 == 0, "%s: m_pad, n_pad must be multiples of 512", who);
    BMF_REQUIRE(st->kp == 32 || st->kp == 64, "%s: kp must be 32 or 64", who);
    BMF_REQUIRE(st->log_rows >= 2 && st->dot_blocks >= 1 && it >= 0, "%s: need log_rows >= 2, dot_blocks >= 1 and it >= 0", who);
    return BMF_OK;
}

extern "C" int bmf_palm_row_lag(const bmf_palm_state* st) {
    int rc = palm_check_state(st, 0, "bmf_palm_row_lag");
    if (rc != BMF_OK) return rc;
    return palm_fused(st) ? 1 : 0;
}

extern "C" int bmf_palm_finish_row(const bmf_palm_state* st, int it, void* stream) {
    int rc = palm_check_state(st, it, "bmf_palm_finish_row");
    if (rc != BMF_OK) return rc;
    if (!palm_fused(st)) return BMF_OK;
    const int64_t nu = st->m_pad * st->kp;
    if ((rc = bmf_dot_slabs(st->U64, st->Mslab, nu, st->splits_xv, nu, st->dotpart, st->dot_blocks, stream)) != BMF_OK) return rc;
    BMF_LAUNCH(palm_dot_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, st->dotpart, st->dot_blocks, st->log + 8 * (int64_t)(it % st->log_rows));
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_palm_iterate(const bmf_palm_state* st, int it, double l1, double l2, double gap_l1, double gap_l2, int phase, void* stream) {
    int rc = palm_check_state(st, it, "bmf_palm_iterate");
    if (rc != BMF_OK) return rc;
    BMF_REQUIRE(phase >= 1 && phase <= 3, "bmf_palm_iterate: phase must be 1 (head), 2 (tail) or 3 (both)");
    const int kp = st->kp;
    const bool fused = palm_fused(st);
    hipStream_t s = (hipStream_t)stream;
    if (phase & 1) {
        // both steps read the state of the previous iteration (ELBMF.py:124-125): nothing derived is touched until both have run
        if ((rc = palm_step(st, true, l1, l2, gap_l1, gap_l2, fused, stream)) != BMF_OK) return rc;
        if (fused && it > 0) {   // <U, X V> of the state the step started from = the cross term of the previous row
            BMF_LAUNCH(palm_dot_finish_kernel, dim3(1), dim3(256), 0, s, st->dotpart, (int)(st->m_pad / 128),
                       st->log + 8 * (int64_t)((it - 1) % st->log_rows));
            BMF_LAUNCH_CHECK();
        }
    }
    if (!(phase & 2)) return BMF_OK;
    if ((rc = palm_step(st, false, l1, l2, gap_l1, gap_l2, fused, stream)) != BMF_OK) return rc;
    if ((rc = palm_derive(st, true, fused, stream)) != BMF_OK) return rc;
    if ((rc = palm_derive(st, false, fused, stream)) != BMF_OK) return rc;
    // the norms of both Grams (the step sizes of the NEXT iteration) in one launch
    if (kp == 32) BMF_LAUNCH(sym_norms2_kernel<32>, dim3(2), dim3(256), 0, s, st->GU64, st->normsU, st->GV64, st->normsV);
    else BMF_LAUNCH(sym_norms2_kernel<64>, dim3(2), dim3(256), 0, s, st->GU64, st->normsU, st->GV64, st->normsV);
    BMF_LAUNCH_CHECK();
    // X^T U (uses the planes of U) and X V (planes of V)
    rc = bmf_xf_bits_i8(st->XTtiled, st->n_pad, st->m_pad / 32, st->m_pad / 32, st->Upanel, st->m_pad, 3, st->scaleU + kp, kp, st->Nslab,
                        st->n_pad * kp, st->splits_xtu, 1, stream);
    if (rc != BMF_OK) return rc;
    // (the cover count between the two GEMMs)
    rc = bmf_cover_count(st->Xbits, st->m_pad, st->ldx, st->n_pad / 32, st->ubits, st->vcolbits, st->n_pad / 32, kp, st->counts, nullptr, stream);
    if (rc != BMF_OK) return rc;
    rc = bmf_xf_bits_i8(st->Xtiled, st->m_pad, st->n_pad / 32, st->n_pad / 32, st->Vpanel, st->n_pad, 3, st->scaleV + kp, kp, st->Mslab,
                        st->m_pad * kp, st->splits_xv, 1, stream);
    if (rc != BMF_OK) return rc;
    // the log row (`log` may be host memory the device can write: the row then needs no copy).  Fused: its cross term <U, X V> comes
    // from the U step of the next iteration (or bmf_palm_finish_row); otherwise from a pass over U and X V here.
    double* row = st->log + 8 * (int64_t)(it % st->log_rows);
    int nd = 0;
    if (!fused) {
        const int64_t nu = st->m_pad * kp;
        if ((rc = bmf_dot_slabs(st->U64, st->Mslab, nu, st->splits_xv, nu, st->dotpart, st->dot_blocks, stream)) != BMF_OK) return rc;
        nd = st->dot_blocks;
    }
    BMF_LAUNCH(palm_scalars_kernel, dim3(1), dim3(256), 0, s, st->dotpart, nd, st->GU64, st->GV64, kp * kp, st->partU, (int)(st->m_pad / 128), st->partV,
               (int)(st->n_pad / 128), st->counts, row);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

// One iteration of PRIMP's loop (PyBMF/models/PRIMP.py:96-131) per call: Gauss-Seidel -- the V step sees the new U (:114-115) -- with
// the inertial term anchored at Up64 / Vp64 for the whole run (the reference never advances its "previous iterate", :96-110), the
// box prox applied twice (palm_epilogue, variant PRIMP), Frobenius step sizes, and the objective ||X - U V^T||_F^2 = sum X - 2 <U, X V>
// + <U^T U, V^T V> as words 0 and 1 of log row it % log_rows (host memory the device can write: the caller reads it one iteration
// late and never waits for the stream).  No digit planes out of the step (the anchored, inertial form), no cover count.
extern "C" int bmf_primp_iterate(const bmf_palm_state* st, int it, double l1, double l2, void* stream) {
    int rc = palm_check_state(st, it, "bmf_primp_iterate", BMF_PALM_PRIMP);
    if (rc != BMF_OK) return rc;
    const int kp = st->kp;
    hipStream_t s = (hipStream_t)stream;
    for (int side = 0; side < 2; ++side) {
        const bool u_side = side == 0;
        // the step consumes X V / V^T V (U side) or X^T U / U^T U of the NEW U (V side); the anchor stays (advance_prev = 0)
        if ((rc = palm_step(st, u_side, l1, l2, 0.0, 0.0, false, stream, 0)) != BMF_OK) return rc;
        if ((rc = palm_derive(st, u_side, false, stream)) != BMF_OK) return rc;
        if ((rc = bmf_sym_norms(u_side ? st->GU64 : st->GV64, kp, u_side ? st->normsU : st->normsV, stream)) != BMF_OK) return rc;
        rc = u_side ? bmf_xf_bits_i8(st->XTtiled, st->n_pad, st->m_pad / 32, st->m_pad / 32, st->Upanel, st->m_pad, 3, st->scaleU + kp, kp, st->Nslab,
                                     st->n_pad * kp, st->splits_xtu, 1, stream)
                    : bmf_xf_bits_i8(st->Xtiled, st->m_pad, st->n_pad / 32, st->n_pad / 32, st->Vpanel, st->n_pad, 3, st->scaleV + kp, kp, st->Mslab,
                                     st->m_pad * kp, st->splits_xv, 1, stream);
        if (rc != BMF_OK) return rc;
    }
    const int64_t nu = st->m_pad * kp;
    if ((rc = bmf_dot_slabs(st->U64, st->Mslab, nu, st->splits_xv, nu, st->dotpart, st->dot_blocks, stream)) != BMF_OK) return rc;
    BMF_LAUNCH(palm_scalars_kernel, dim3(1), dim3(256), 0, s, st->dotpart, st->dot_blocks, st->GU64, st->GV64, kp * kp, st->partU, (int)(st->m_pad / 128),
               st->partV, (int)(st->n_pad / 128), (unsigned long long*)nullptr, st->log + 8 * (int64_t)(it % st->log_rows));
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}
