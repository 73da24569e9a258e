// K5' / K7: tile-fused residual pass  R = X - A B^T  over the real m x n cells, never materialised.
//
//   sums[0] += sum |R|, sums[1] += sum R^2                       -> MAE / RMSE  PyBMF/utils/metrics.py:149-160
//   (A, B) = (U, V)                                               (evaluate(..., metrics=['RMSE','MAE']) BinaryMFPenalty.py:71,97)
//   (A, B) = (sigmoid(lam(U-u)), sigmoid(lam(V-v)))               -> F(u,v) = 0.5 * sums[1]   BinaryMFThreshold.py:150-171
//   with gradient operands dA = dXdx(U,u), dB = dXdx(V,v):
//   sums[2] += sum R o (dA B^T), sums[3] += sum R o (A dB^T)      -> dF(u,v)                  BinaryMFThreshold.py:174-207
//
// Exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), computed transposed so that the X bits a lane needs are one
// contiguous stream:  D[j][i] = sum_k B[j][k] A[i][k]; lane (c, h) holds column i = i0 + c and 16 rows j, i.e. 16
// bits of one 32-bit word of X row i.  A wave keeps its 32 A rows in registers (the reduction index is split over
// the two lane halves: half h owns k in [kp/2*h, kp/2*(h+1))) and streams B rows from L2; fp64 accumulation.
#include "common.h"

namespace {

template <int KP, bool GRAD, bool REALX = false>
__global__ __launch_bounds__(256) void residual_kernel(const uint32_t* __restrict__ Xbits, int64_t ldx, int m, int n,
                                                        const float* __restrict__ A, const float* __restrict__ B,
                                                        const float* __restrict__ dA, const float* __restrict__ dB,
                                                        int col_tiles_per_block, double* __restrict__ sums,
                                                        const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    constexpr int KH = KP / 2;  // reduction indices per lane half
    __shared__ double red[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * 32;  // 32 rows of X per wave
    const int jt0 = blockIdx.y * col_tiles_per_block;
    const int jt1 = min(jt0 + col_tiles_per_block, (n + 31) / 32);

    // this lane's row of A (and dA): KH floats
    float a[KH], da[GRAD ? KH : 1];
    {
        const float* ap = A + (i0 + c) * KP + KH * h;
#pragma unroll
        for (int s = 0; s < KH; s += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(ap + s);
            a[s] = v[0]; a[s + 1] = v[1]; a[s + 2] = v[2]; a[s + 3] = v[3];
        }
        if constexpr (GRAD) {
            const float* dp = dA + (i0 + c) * KP + KH * h;
#pragma unroll
            for (int s = 0; s < KH; s += 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(dp + s);
                da[s] = v[0]; da[s + 1] = v[1]; da[s + 2] = v[2]; da[s + 3] = v[3];
            }
        }
    }
    const bool row_ok = (i0 + c) < m;
    double s_abs = 0.0, s_sq = 0.0, s_g1 = 0.0, s_g2 = 0.0;

    for (int jt = jt0; jt < jt1; ++jt) {
        const int64_t j0 = (int64_t)jt * 32;
        float b[KH], db[GRAD ? KH : 1];
        const float* bp = B + (j0 + c) * KP + KH * h;
#pragma unroll
        for (int s = 0; s < KH; s += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(bp + s);
            b[s] = v[0]; b[s + 1] = v[1]; b[s + 2] = v[2]; b[s + 3] = v[3];
        }
        if constexpr (GRAD) {
            const float* dp = dB + (j0 + c) * KP + KH * h;
#pragma unroll
            for (int s = 0; s < KH; s += 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(dp + s);
                db[s] = v[0]; db[s + 1] = v[1]; db[s + 2] = v[2]; db[s + 3] = v[3];
            }
        }
        // X of this lane's row: one 32-bit word of bits, or (REALX) 4 x 4 consecutive floats -- columns j0 + 8q + 4h + 0..3
        unsigned xw = 0u;
        f32x4 xr[4];
        if constexpr (REALX) {
            const float* xp = reinterpret_cast<const float*>(Xbits) + (i0 + c) * ldx + j0 + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) xr[q] = *reinterpret_cast<const f32x4*>(xp + 8 * q);
        } else {
            xw = Xbits[(i0 + c) * ldx + jt];
        }

        f32x16 p, q1, q2;
#pragma unroll
        for (int i = 0; i < 16; ++i) { p[i] = 0.f; q1[i] = 0.f; q2[i] = 0.f; }
#pragma unroll
        for (int s = 0; s < KH; ++s) {
            // MFMA row index = B row (lane & 31 of the A-operand), column index = A row (lane & 31 of the B-operand)
            p = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s], a[s], p, 0, 0, 0);
            if constexpr (GRAD) {
                q1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s], da[s], q1, 0, 0, 0);   // (dA B^T)^T
                q2 = __builtin_amdgcn_mfma_f32_32x32x2f32(db[s], a[s], q2, 0, 0, 0);   // (A dB^T)^T
            }
        }
        float t_abs = 0.f, t_sq = 0.f, t_g1 = 0.f, t_g2 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int jr = (i & 3) + 8 * (i >> 2) + 4 * h;  // row of D = column j0 + jr of X
            const bool ok = row_ok && (j0 + jr) < n;
            const float x = REALX ? xr[i >> 2][i & 3] : (float)((xw >> jr) & 1u);
            const float r = ok ? (x - p[i]) : 0.f;
            t_abs += fabsf(r);
            t_sq = fmaf(r, r, t_sq);
            if constexpr (GRAD) {
                t_g1 = fmaf(r, q1[i], t_g1);
                t_g2 = fmaf(r, q2[i], t_g2);
            }
        }
        s_abs += (double)t_abs;
        s_sq += (double)t_sq;
        if constexpr (GRAD) { s_g1 += (double)t_g1; s_g2 += (double)t_g2; }
    }
    s_abs = wave_sum(s_abs);
    s_sq = wave_sum(s_sq);
    if constexpr (GRAD) { s_g1 = wave_sum(s_g1); s_g2 = wave_sum(s_g2); }
    if (lane == 0) { red[wave][0] = s_abs; red[wave][1] = s_sq; red[wave][2] = s_g1; red[wave][3] = s_g2; }
    __syncthreads();
    if (threadIdx.x < (GRAD ? 4 : 2)) {
        const int q = threadIdx.x;
        atomicAdd(&sums[q], ((red[0][q] + red[1][q]) + red[2][q]) + red[3][q]);
    }
}

// out[i][j] = sum_k A[i][k] B[j][k]  (the real-valued X_pd = U V^T of get_prediction(boolean=False), PyBMF/utils/common.py:98-107),
// written as a dense m x n fp32 matrix; same tile mapping as the residual pass (lane (c, h): row i0 + c, 16 columns).
template <int KP>
__global__ __launch_bounds__(256) void product_kernel(const float* __restrict__ A, const float* __restrict__ B, int m, int n,
                                                       int col_tiles_per_block, float* __restrict__ out, int64_t ldo) {
    constexpr int KH = KP / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
    const int jt0 = blockIdx.y * col_tiles_per_block;
    const int jt1 = min(jt0 + col_tiles_per_block, (n + 31) / 32);
    float a[KH];
    const float* ap = A + (i0 + c) * KP + KH * h;
#pragma unroll
    for (int s = 0; s < KH; s += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(ap + s);
        a[s] = v[0]; a[s + 1] = v[1]; a[s + 2] = v[2]; a[s + 3] = v[3];
    }
    for (int jt = jt0; jt < jt1; ++jt) {
        const int64_t j0 = (int64_t)jt * 32;
        float b[KH];
        const float* bp = B + (j0 + c) * KP + KH * h;
#pragma unroll
        for (int s = 0; s < KH; s += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(bp + s);
            b[s] = v[0]; b[s + 1] = v[1]; b[s + 2] = v[2]; b[s + 3] = v[3];
        }
        f32x16 p;
#pragma unroll
        for (int i = 0; i < 16; ++i) p[i] = 0.f;
#pragma unroll
        for (int s = 0; s < KH; ++s) p = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s], a[s], p, 0, 0, 0);
        if (i0 + c < m) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t j = j0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (j < n) out[(i0 + c) * ldo + j] = p[i];
            }
        }
    }
}

// elementwise transform for the thresholding objective: S = sigmoid(lam (F - x)), D = lam * S * (1 - S)
// (= lam exp(-lam(F-x)) sigmoid(lam(F-x))^2 of BinaryMFThreshold.py:211-227 in an overflow-free form), fp64 math.
__global__ __launch_bounds__(256) void thresh_transform_kernel(const float* __restrict__ F, int64_t rows_pad, int rows,
                                                                int k, int kp, double x, double lam,
                                                                float* __restrict__ S, float* __restrict__ D) {
    const int64_t total = rows_pad * kp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / kp;
        const int j = (int)(i - r * kp);
        float s = 0.f, d = 0.f;
        if (r < rows && j < k) {
            const double z = ((double)F[i] - x) * lam;
            double sg;
            if (z >= 0) sg = 1.0 / (1.0 + exp(-z));
            else { const double e = exp(z); sg = e / (1.0 + e); }
            s = (float)sg;
            d = (float)(lam * sg * (1.0 - sg));
        }
        S[i] = s;
        if (D) D[i] = d;
    }
}

}  // namespace

int bmf_residual_launch(const uint32_t* Xbits, int64_t m_pad, int64_t ldx, int m, int n, const float* A, const float* B,
                        const float* dA, const float* dB, int kp, double* sums, const int32_t* stop, hipStream_t s) {
    const int row_blocks = (int)((m + 127) / 128);  // 4 waves x 32 rows
    const int col_tiles = (n + 31) / 32;
    // enough blocks to fill the chip ~4x; each block walks a contiguous range of column tiles
    int col_groups = (1024 + row_blocks - 1) / row_blocks;
    if (col_groups > col_tiles) col_groups = col_tiles;
    if (col_groups < 1) col_groups = 1;
    const int per = (col_tiles + col_groups - 1) / col_groups;
    col_groups = (col_tiles + per - 1) / per;
    dim3 grid((unsigned)row_blocks, (unsigned)col_groups), block(256);
    const bool grad = dA != nullptr;
    if (kp == 32) {
        if (grad) BMF_LAUNCH((residual_kernel<32, true>), grid, block, 0, s, Xbits, ldx, m, n, A, B, dA, dB, per, sums, stop);
        else BMF_LAUNCH((residual_kernel<32, false>), grid, block, 0, s, Xbits, ldx, m, n, A, B, dA, dB, per, sums, stop);
    } else {
        if (grad) BMF_LAUNCH((residual_kernel<64, true>), grid, block, 0, s, Xbits, ldx, m, n, A, B, dA, dB, per, sums, stop);
        else BMF_LAUNCH((residual_kernel<64, false>), grid, block, 0, s, Xbits, ldx, m, n, A, B, dA, dB, per, sums, stop);
    }
    BMF_LAUNCH_CHECK();
    (void)m_pad;
    return BMF_OK;
}

extern "C" int bmf_residual_sums(const uint32_t* Xbits, int64_t m_pad, int64_t ldx, int32_t m, int32_t n, const float* U,
                                 const float* V, int kp, double* sums, const int32_t* stop, void* stream) {
    BMF_REQUIRE(Xbits && U && V && sums, "bmf_residual_sums: null pointer");
    BMF_REQUIRE(m >= 1 && n >= 1 && m <= m_pad && m_pad % 128 == 0, "bmf_residual_sums: bad m/m_pad");
    BMF_REQUIRE(ldx * 32 >= n, "bmf_residual_sums: ldx does not cover n");
    BMF_REQUIRE(kp == 32 || kp == 64, "bmf_residual_sums: kp must be 32 or 64");
    BMF_REQUIRE(bmf_aligned16(U) && bmf_aligned16(V), "bmf_residual_sums: factors must be 16-byte aligned");
    return bmf_residual_launch(Xbits, m_pad, ldx, m, n, U, V, nullptr, nullptr, kp, sums, stop, (hipStream_t)stream);
}

extern "C" int bmf_thresh_eval(const uint32_t* Xbits, int64_t m_pad, int64_t ldx, int32_t m, int32_t n, const float* U,
                               int64_t n_pad, const float* V, int k, int kp, double u, double v, double lamda,
                               int want_grad, float* work, double* out, void* stream) {
    BMF_REQUIRE(Xbits && U && V && work && out, "bmf_thresh_eval: null pointer");
    BMF_REQUIRE(m >= 1 && n >= 1 && m <= m_pad && n <= n_pad && m_pad % 128 == 0 && n_pad % 32 == 0, "bmf_thresh_eval: bad shape");
    BMF_REQUIRE(ldx * 32 >= n, "bmf_thresh_eval: ldx does not cover n");
    BMF_REQUIRE((kp == 32 || kp == 64) && k >= 1 && k <= kp, "bmf_thresh_eval: need 1 <= k <= kp, kp in {32,64}");
    BMF_REQUIRE(bmf_aligned16(work), "bmf_thresh_eval: work must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    // work = [Us | dUs | Vs | dVs], (2*m_pad + 2*n_pad) * kp floats
    float* Us = work;
    float* dUs = Us + m_pad * kp;
    float* Vs = dUs + m_pad * kp;
    float* dVs = Vs + n_pad * kp;
    BMF_HIP_CHECK(hipMemsetAsync(out, 0, 4 * sizeof(double), s));
    const unsigned gu = (unsigned)((m_pad * kp + 255) / 256), gv = (unsigned)((n_pad * kp + 255) / 256);
    BMF_LAUNCH(thresh_transform_kernel, dim3(gu < 2048 ? gu : 2048), dim3(256), 0, s, U, m_pad, m, k, kp, u, lamda, Us,
                       want_grad ? dUs : nullptr);
    BMF_LAUNCH(thresh_transform_kernel, dim3(gv < 2048 ? gv : 2048), dim3(256), 0, s, V, n_pad, n, k, kp, v, lamda, Vs,
                       want_grad ? dVs : nullptr);
    BMF_LAUNCH_CHECK();
    return bmf_residual_launch(Xbits, m_pad, ldx, m, n, Us, Vs, want_grad ? dUs : nullptr, want_grad ? dVs : nullptr, kp, out,
                               nullptr, s);
}

/* real-valued X (WNMF on non-Boolean data): X is m_pad x ldx floats, ldx a multiple of 32 covering n, zero padded */
int bmf_residual_launch_f32(const float* X, int64_t m_pad, int64_t ldx, int m, int n, const float* U, const float* V, int kp, double* sums,
                            const int32_t* stop, hipStream_t s) {
    BMF_REQUIRE(X && U && V && sums, "bmf_residual_sums_f32: null pointer");
    BMF_REQUIRE(m >= 1 && n >= 1 && m <= m_pad && m_pad % 128 == 0, "bmf_residual_sums_f32: bad m/m_pad");
    BMF_REQUIRE(ldx >= n && ldx % 32 == 0, "bmf_residual_sums_f32: ldx must be a multiple of 32 covering n");
    BMF_REQUIRE(kp == 32 || kp == 64, "bmf_residual_sums_f32: kp must be 32 or 64");
    BMF_REQUIRE(bmf_aligned16(X) && bmf_aligned16(U) && bmf_aligned16(V), "bmf_residual_sums_f32: pointers must be 16-byte aligned");
    const int row_blocks = (int)((m + 127) / 128);
    const int col_tiles = (n + 31) / 32;
    int col_groups = (1024 + row_blocks - 1) / row_blocks;
    if (col_groups > col_tiles) col_groups = col_tiles;
    const int per = (col_tiles + col_groups - 1) / col_groups;
    col_groups = (col_tiles + per - 1) / per;
    dim3 grid((unsigned)row_blocks, (unsigned)col_groups), block(256);
    const uint32_t* Xw = reinterpret_cast<const uint32_t*>(X);
    if (kp == 32)
        BMF_LAUNCH((residual_kernel<32, false, true>), grid, block, 0, s, Xw, ldx, m, n, U, V, nullptr, nullptr, per, sums, stop);
    else
        BMF_LAUNCH((residual_kernel<64, false, true>), grid, block, 0, s, Xw, ldx, m, n, U, V, nullptr, nullptr, per, sums, stop);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_residual_sums_f32(const float* X, int64_t m_pad, int64_t ldx, int32_t m, int32_t n, const float* U,
                                     const float* V, int kp, double* sums, void* stream) {
    return bmf_residual_launch_f32(X, m_pad, ldx, m, n, U, V, kp, sums, nullptr, (hipStream_t)stream);
}

extern "C" int bmf_real_product(const float* U, int64_t m_pad, int32_t m, const float* V, int64_t n_pad, int32_t n, int kp,
                                float* out, int64_t ldo, void* stream) {
    BMF_REQUIRE(U && V && out, "bmf_real_product: null pointer");
    BMF_REQUIRE(m >= 1 && n >= 1 && m <= m_pad && n <= n_pad && m_pad % 128 == 0 && n_pad % 32 == 0, "bmf_real_product: bad shape");
    BMF_REQUIRE(kp == 32 || kp == 64, "bmf_real_product: kp must be 32 or 64");
    BMF_REQUIRE(ldo >= n, "bmf_real_product: ldo < n");
    BMF_REQUIRE(bmf_aligned16(U) && bmf_aligned16(V), "bmf_real_product: factors must be 16-byte aligned");
    const int row_blocks = (int)((m + 127) / 128);
    const int col_tiles = (n + 31) / 32;
    int col_groups = (1024 + row_blocks - 1) / row_blocks;
    if (col_groups > col_tiles) col_groups = col_tiles;
    const int per = (col_tiles + col_groups - 1) / col_groups;
    col_groups = (col_tiles + per - 1) / per;
    dim3 grid((unsigned)row_blocks, (unsigned)col_groups), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (kp == 32) BMF_LAUNCH(product_kernel<32>, grid, block, 0, s, U, V, m, n, per, out, ldo);
    else BMF_LAUNCH(product_kernel<64>, grid, block, 0, s, U, V, m, n, per, out, ldo);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

/* S = sigmoid(lam (F - x)), D = lam S (1 - S) (D may be NULL): the element-wise transform of the thresholding objective
 * (PyBMF/models/BinaryMFThreshold.py:163-164,211-227), fp64 math, rows >= `rows` and columns >= k written as 0. */
extern "C" int bmf_thresh_transform(const float* F, int64_t rows_pad, int32_t rows, int k, int kp, double x, double lamda,
                                    float* S, float* D, void* stream) {
    BMF_REQUIRE(F && S, "bmf_thresh_transform: null pointer");
    BMF_REQUIRE(rows >= 1 && rows <= rows_pad && (kp == 32 || kp == 64) && k >= 1 && k <= kp, "bmf_thresh_transform: bad shape");
    const unsigned g = (unsigned)((rows_pad * kp + 255) / 256);
    BMF_LAUNCH(thresh_transform_kernel, dim3(g < 2048 ? g : 2048), dim3(256), 0, (hipStream_t)stream, F, rows_pad, rows, k, kp, x, lamda, S, D);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}
