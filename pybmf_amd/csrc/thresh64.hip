// The thresholding objective of BinaryMFThreshold in fp64 end to end (PyBMF/models/BinaryMFThreshold.py:150-227).
//
//   F(u, v)  = 1/2 || W o (X - Us Vs^T) ||_F^2,  Us = sigmoid(lamda (U - u)), Vs = sigmoid(lamda (V - v))       :150-171
//   dF(u, v) = ( sum R o (dXdx(U, u) Vs^T),  sum R o (Us dXdx(V, v)^T) )  with R = W o (X - Us Vs^T)              :174-207
//
// The line search (PyBMF/solvers/line_search.py) is a chain of comparisons of F values that differ by min_diff = 1e-3 on
// F ~ 1e4, and at lamda = 100 the sigmoid turns a 6e-8 rounding of a factor entry into a 1e-6 change of a cell: with fp32
// factors on the device the search took different branches than the reference after a few iterations (round 1: row count
// +-2, (u, v) within 5e-3).  Here the factors arrive as fp64, the transform, the products and every sum are fp64, and the
// block partials are added in a fixed order, so the search reproduces the reference's decisions (tests: same row count,
// (u, v) to 1e-6).  The matrices of this model are small (MovieLens-1M: 6040 x 3706, k = 16 -- 7e8 flop per evaluation),
// so the fp64 VALU rate (78 TFLOP/s) is not what limits it; an evaluation is bound by its launches.
#include "common.h"

namespace {

__device__ __forceinline__ void sigmoid_pair(double z, double lam, double& s, double& d) {
    if (z >= 0) s = 1.0 / (1.0 + exp(-z));
    else { const double e = exp(z); s = e / (1.0 + e); }
    d = lam * s * (1.0 - s);   // = lam exp(-z) s^2 (BinaryMFThreshold.py:211-227) without the overflow for z < -709
}

__global__ __launch_bounds__(256) void transform64_kernel(const double* __restrict__ F, int64_t rows_pad, int rows, int k, int kp,
                                                           double x, double lam, double* __restrict__ S, double* __restrict__ D) {
    const int64_t total = rows_pad * kp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / kp;
        const int j = (int)(i - r * kp);
        double s = 0.0, d = 0.0;
        if (r < rows && j < k) sigmoid_pair((F[i] - x) * lam, lam, s, d);
        S[i] = s;
        if (D) D[i] = d;
    }
}

// both factors in one launch (an evaluation is bound by its launches): blocks [0, gu) do U, the rest V
__global__ __launch_bounds__(256) void transform64_pair_kernel(const double* __restrict__ U, int64_t m_pad, int m, double u,
                                                                double* __restrict__ Us, double* __restrict__ dUs, int gu,
                                                                const double* __restrict__ V, int64_t n_pad, int n, double v,
                                                                double* __restrict__ Vs, double* __restrict__ dVs, int k, int kp, double lam) {
    const bool is_u = (int)blockIdx.x < gu;
    const double* F = is_u ? U : V;
    double* S = is_u ? Us : Vs;
    double* D = is_u ? dUs : dVs;
    const int rows = is_u ? m : n;
    const double x = is_u ? u : v;
    const int64_t total = (is_u ? m_pad : n_pad) * kp;
    const int64_t b = is_u ? blockIdx.x : blockIdx.x - gu, nb = is_u ? gu : gridDim.x - gu;
    for (int64_t i = b * 256 + threadIdx.x; i < total; i += nb * 256) {
        const int64_t r = i / kp;
        const int j = (int)(i - r * kp);
        double s = 0.0, d = 0.0;
        if (r < rows && j < k) sigmoid_pair((F[i] - x) * lam, lam, s, d);
        S[i] = s;
        if (D) D[i] = d;
    }
}

// One block = a 64 x 64 tile of cells, thread (ty, tx) = rows 4 ty .. 4 ty + 3, columns 4 tx .. 4 tx + 3.  The factor tiles go
// through LDS 16 latent dimensions at a time, transposed ([kk][row]) so that a thread's four rows / columns are one 32-byte read.
// partial[block][0..3] = sum |r|, sum r^2, sum r (dUs Vs^T), sum r (Us dVs^T) over the block's real cells.
template <bool GRAD>
__global__ __launch_bounds__(256) void dense64_kernel(const uint32_t* __restrict__ Xbits, int64_t ldx, int m, int n,
                                                       const double* __restrict__ Us, const double* __restrict__ dUs,
                                                       const double* __restrict__ Vs, const double* __restrict__ dVs, int kp, int k,
                                                       double* __restrict__ partial) {
    __shared__ double a_t[16][64], b_t[16][64], da_t[GRAD ? 16 : 1][64], db_t[GRAD ? 16 : 1][64];
    __shared__ double red[4][4];
    const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
    const int64_t i0 = (int64_t)blockIdx.y * 64, j0 = (int64_t)blockIdx.x * 64;
    double p[4][4], gu[4][4], gv[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) p[a][b] = gu[a][b] = gv[a][b] = 0.0;
    for (int k0 = 0; k0 < k; k0 += 16) {   // latent dimensions >= k are zero padding: whole chunks of them are skipped
        __syncthreads();
        for (int e = t; e < 64 * 16; e += 256) {   // e = kk * 64 + row: consecutive threads write consecutive LDS words (with e = row * 16
            const int kk = e >> 6, row = e & 63;    // + kk the transposed store was a 16-way bank conflict); the tiles come from L2
            a_t[kk][row] = Us[(i0 + row) * kp + k0 + kk];
            b_t[kk][row] = Vs[(j0 + row) * kp + k0 + kk];
            if (GRAD) {
                da_t[kk][row] = dUs[(i0 + row) * kp + k0 + kk];
                db_t[kk][row] = dVs[(j0 + row) * kp + k0 + kk];
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < 16; ++kk) {
            double a[4], b[4], da[4], db[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[q] = a_t[kk][4 * ty + q];
                b[q] = b_t[kk][4 * tx + q];
                da[q] = GRAD ? da_t[kk][4 * ty + q] : 0.0;
                db[q] = GRAD ? db_t[kk][4 * tx + q] : 0.0;
            }
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) {
                    p[x][y] = fma(a[x], b[y], p[x][y]);
                    if (GRAD) {
                        gu[x][y] = fma(da[x], b[y], gu[x][y]);
                        gv[x][y] = fma(a[x], db[y], gv[x][y]);
                    }
                }
        }
    }
    double s_abs = 0.0, s_sq = 0.0, s_gu = 0.0, s_gv = 0.0;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const int64_t i = i0 + 4 * ty + x;
        const unsigned w = i < m ? Xbits[i * ldx + ((j0 + 4 * tx) >> 5)] : 0u;
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const int64_t j = j0 + 4 * tx + y;
            if (i < m && j < n) {
                const double xv = (double)((w >> ((int)(j & 31))) & 1u);
                const double r = xv - p[x][y];
                s_abs += fabs(r);
                s_sq += r * r;
                if (GRAD) {
                    s_gu += r * gu[x][y];
                    s_gv += r * gv[x][y];
                }
            }
        }
    }
    s_abs = wave_sum(s_abs); s_sq = wave_sum(s_sq); s_gu = wave_sum(s_gu); s_gv = wave_sum(s_gv);
    if ((t & 63) == 0) { red[t >> 6][0] = s_abs; red[t >> 6][1] = s_sq; red[t >> 6][2] = s_gu; red[t >> 6][3] = s_gv; }
    __syncthreads();
    if (t < 4) partial[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + t] = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
}

// out[c] = sum_b partial[b][c], c < ncol <= 4, in a fixed order (deterministic): 1024 threads, thread t sums column t & 3 of the
// blocks t >> 2, t >> 2 + 256, ...; then a tree over the 256 partial sums of each column
__global__ __launch_bounds__(1024) void sum_partials_kernel(const double* __restrict__ partial, int64_t nblk, int stride, int ncol,
                                                             double* __restrict__ out) {
    __shared__ double red[4][256];
    const int c = threadIdx.x & 3, q = threadIdx.x >> 2;
    double acc = 0.0;
    if (c < ncol)
        for (int64_t b = q; b < nblk; b += 256) acc += partial[b * stride + c];
    red[c][q] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (q < o) red[c][q] += red[c][q + o];
        __syncthreads();
    }
    if (q == 0 && c < ncol) out[c] = red[c][0];
}

// masked variant: over the observed cells of a segmented CSR list (see bmf_masked_pass, csrc/masked.hip), shaped like its segment
// kernel: the cell list of a segment is one vector load (lane = cell); G lanes take a cell (G = 16 / 32 when the factors are that
// narrow, else 64), so a wave handles 64 / G cells per step with four steps' gathers in flight, and a cell's dot products are DPP sums
// over its own 16-lane row(s) -- the first version walked the cells one by one, each with one to three __shfl_xor trees on doubles
// (twelve dependent ds_bpermute each): 200 us per F evaluation for a million cells, three times the DENSE evaluation of 22 million.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// sum over the G lanes of this lane's group, in every lane of the group
template <int G>
__device__ __forceinline__ double group_sum(double v) {
    v += dpp_f64<0xB1>(v);    // quad_perm [1, 0, 3, 2]
    v += dpp_f64<0x4E>(v);    // quad_perm [2, 3, 0, 1]
    v += dpp_f64<0x141>(v);   // row_half_mirror
    v += dpp_f64<0x140>(v);   // row_mirror: the 16-lane row's sum
    if constexpr (G >= 32) v += __shfl_xor(v, 16, 64);
    if constexpr (G >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
template <int KP, bool GRAD, int G>
__global__ __launch_bounds__(256) void masked64_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                        const float* __restrict__ val, const float* __restrict__ wgt,
                                                        const int32_t* __restrict__ seg_row, const int64_t* __restrict__ seg_beg,
                                                        int nseg, const double* __restrict__ Us, const double* __restrict__ dUs,
                                                        const double* __restrict__ Vs, const double* __restrict__ dVs,
                                                        double* __restrict__ partial) {
    static_assert(G == 16 || G == 32 || G == 64, "lanes per cell");
    constexpr int CPS = 64 / G, NQ = 4;
    __shared__ double red[4][3];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = lane / G, cl = lane % G;
    const bool on = cl < KP;
    double f = 0.0, g1 = 0.0, g2 = 0.0;   // per lane group
    for (int sg = blockIdx.x * 4 + wave; sg < nseg; sg += gridDim.x * 4) {
        const int r = seg_row[sg];
        const int64_t base = seg_beg[sg];
        const int cnt = (int)min((int64_t)64, ptr[r + 1] - base);
        const double u = on ? Us[(int64_t)r * KP + cl] : 0.0;
        const double du = (GRAD && on) ? dUs[(int64_t)r * KP + cl] : 0.0;
        const int64_t me = base + min(lane, cnt - 1);
        const int my_j = idx[me];
        const float my_x = val[me];
        const float my_w = wgt ? wgt[me] : 1.f;
        for (int q0 = 0; q0 < cnt; q0 += NQ * CPS) {
            double x[NQ], w[NQ], v[NQ], dv[GRAD ? NQ : 1];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int cell = q0 + q * CPS + grp;
                const int qq = min(cell, cnt - 1);   // tail: repeat the last cell with weight 0
                const int j = __shfl(my_j, qq, 64);
                x[q] = (double)__shfl(my_x, qq, 64);
                const float wq = __shfl(my_w, qq, 64);
                w[q] = cell < cnt ? (double)wq : 0.0;
                v[q] = on ? Vs[(int64_t)j * KP + cl] : 0.0;
                if (GRAD) dv[q] = on ? dVs[(int64_t)j * KP + cl] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const double rr = w[q] * (x[q] - group_sum<G>(u * v[q]));
                f += rr * rr;
                if (GRAD) {
                    g1 += rr * group_sum<G>(du * v[q]);
                    g2 += rr * group_sum<G>(u * dv[q]);
                }
            }
        }
    }
    // one lane per group carries the group's sums
    f = wave_sum(cl == 0 ? f : 0.0);
    g1 = wave_sum(cl == 0 ? g1 : 0.0);
    g2 = wave_sum(cl == 0 ? g2 : 0.0);
    if (lane == 0) { red[wave][0] = f; red[wave][1] = g1; red[wave][2] = g2; }
    __syncthreads();
    if (threadIdx.x < 3) partial[(int64_t)blockIdx.x * 4 + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

}  // namespace

extern "C" int bmf_thresh_transform64(const double* F, int64_t rows_pad, int32_t rows, int k, int kp, double x, double lamda,
                                      double* S, double* D, void* stream) {
    BMF_REQUIRE(F && S, "bmf_thresh_transform64: null pointer");
    BMF_REQUIRE(rows >= 1 && rows <= rows_pad && (kp == 32 || kp == 64) && k >= 1 && k <= kp, "bmf_thresh_transform64: bad shape");
    const unsigned g = (unsigned)((rows_pad * kp + 255) / 256);
    BMF_LAUNCH(transform64_kernel, dim3(g < 2048 ? g : 2048), dim3(256), 0, (hipStream_t)stream, F, rows_pad, rows, k, kp, x, lamda, S, D);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int64_t bmf_thresh_eval64_work(int64_t m_pad, int64_t n_pad, int kp) {
    if (m_pad <= 0 || n_pad <= 0 || m_pad % 64 || n_pad % 64 || (kp != 32 && kp != 64)) return BMF_ERR_BAD_ARG;
    return (2 * m_pad + 2 * n_pad) * kp + (m_pad / 64) * (n_pad / 64) * 4;
}

extern "C" int bmf_thresh_eval64(const uint32_t* Xbits, int64_t m_pad, int64_t ldx, int32_t m, int32_t n, const double* U64,
                                 int64_t n_pad, const double* V64, int k, int kp, double u, double v, double lamda, int want_grad,
                                 double* work, double* out, void* stream) {
    BMF_REQUIRE(Xbits && U64 && V64 && work && out, "bmf_thresh_eval64: null pointer");
    BMF_REQUIRE(m >= 1 && n >= 1 && m <= m_pad && n <= n_pad && m_pad % 64 == 0 && n_pad % 64 == 0, "bmf_thresh_eval64: bad shape");
    BMF_REQUIRE(ldx * 32 >= n_pad, "bmf_thresh_eval64: ldx does not cover n_pad");
    BMF_REQUIRE((kp == 32 || kp == 64) && k >= 1 && k <= kp, "bmf_thresh_eval64: need 1 <= k <= kp, kp in {32,64}");
    hipStream_t s = (hipStream_t)stream;
    // work = [Us | dUs | Vs | dVs | block partials]
    double* Us = work;
    double* dUs = Us + m_pad * kp;
    double* Vs = dUs + m_pad * kp;
    double* dVs = Vs + n_pad * kp;
    double* partial = dVs + n_pad * kp;
    const unsigned gu = (unsigned)((m_pad * kp + 255) / 256), gv = (unsigned)((n_pad * kp + 255) / 256);
    const unsigned bu = gu < 1024 ? gu : 1024, bv = gv < 1024 ? gv : 1024;
    BMF_LAUNCH(transform64_pair_kernel, dim3(bu + bv), dim3(256), 0, s, U64, m_pad, m, u, Us, want_grad ? dUs : nullptr, (int)bu, V64, n_pad, n, v, Vs,
               want_grad ? dVs : nullptr, k, kp, lamda);
    dim3 grid((unsigned)(n_pad / 64), (unsigned)(m_pad / 64)), block(256);
    if (want_grad) BMF_LAUNCH(dense64_kernel<true>, grid, block, 0, s, Xbits, ldx, m, n, Us, dUs, Vs, dVs, kp, k, partial);
    else BMF_LAUNCH(dense64_kernel<false>, grid, block, 0, s, Xbits, ldx, m, n, Us, dUs, Vs, dVs, kp, k, partial);
    BMF_LAUNCH(sum_partials_kernel, dim3(1), dim3(1024), 0, s, partial, (int64_t)grid.x * grid.y, 4, 4, out);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

static int masked_thresh64_launch(const int64_t* ptr, const int32_t* idx, const float* val, const float* wgt, const int32_t* seg_row,
                                  const int64_t* seg_beg, int32_t nseg, const double* Us, const double* dUs, const double* Vs, const double* dVs,
                                  int kp, int kcols, double* partial, int32_t partial_blocks, double* out, void* stream, const char* who) {
    BMF_REQUIRE(ptr && idx && val && seg_row && seg_beg && Us && Vs && partial && out, "%s: null pointer", who);
    BMF_REQUIRE((dUs == nullptr) == (dVs == nullptr), "%s: give both derivative factors or neither", who);
    BMF_REQUIRE((kp == 32 || kp == 64) && kcols >= 1 && kcols <= kp, "%s: kp must be 32 or 64, kcols 1..kp", who);
    BMF_REQUIRE(nseg >= 1 && partial_blocks >= 1 && partial_blocks <= 65535, "%s: bad nseg / partial_blocks", who);
    hipStream_t s = (hipStream_t)stream;
    const bool grad = dUs != nullptr;
    dim3 grid((unsigned)partial_blocks), block(256);
    // lanes per cell: the narrowest group that holds the kcols real columns (the padding columns of the transformed factors are zero)
    const int g = kp == 64 ? 64 : (kcols <= 16 ? 16 : 32);
#define BMF_MT64(KP_, GR_, G_) BMF_LAUNCH((masked64_kernel<KP_, GR_, G_>), grid, block, 0, s, ptr, idx, val, wgt, seg_row, seg_beg, nseg, Us, dUs, Vs, dVs, partial)
    if (g == 16) { if (grad) BMF_MT64(32, true, 16); else BMF_MT64(32, false, 16); }
    else if (g == 32) { if (grad) BMF_MT64(32, true, 32); else BMF_MT64(32, false, 32); }
    else { if (grad) BMF_MT64(64, true, 64); else BMF_MT64(64, false, 64); }
#undef BMF_MT64
    BMF_LAUNCH(sum_partials_kernel, dim3(1), dim3(1024), 0, s, partial, (int64_t)partial_blocks, 4, 3, out);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_masked_thresh64(const int64_t* ptr, const int32_t* idx, const float* val, const float* wgt,
                                   const int32_t* seg_row, const int64_t* seg_beg, int32_t nseg, const double* Us, const double* dUs,
                                   const double* Vs, const double* dVs, int kp, double* partial, int32_t partial_blocks, double* out,
                                   void* stream) {
    return masked_thresh64_launch(ptr, idx, val, wgt, seg_row, seg_beg, nseg, Us, dUs, Vs, dVs, kp, kp, partial, partial_blocks, out, stream,
                                  "bmf_masked_thresh64");
}

extern "C" int bmf_masked_thresh64_k(const int64_t* ptr, const int32_t* idx, const float* val, const float* wgt,
                                     const int32_t* seg_row, const int64_t* seg_beg, int32_t nseg, const double* Us, const double* dUs,
                                     const double* Vs, const double* dVs, int kp, int kcols, double* partial, int32_t partial_blocks,
                                     double* out, void* stream) {
    return masked_thresh64_launch(ptr, idx, val, wgt, seg_row, seg_beg, nseg, Us, dUs, Vs, dVs, kp, kcols, partial, partial_blocks, out, stream,
                                  "bmf_masked_thresh64_k");
}
