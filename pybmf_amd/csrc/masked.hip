// Masked multiplicative update: the two sparse contractions over the OBSERVED cells of X.
//
//   num = (W o X)  F_other        den = (W o (F_self F_other^T)) F_other      (and the transposed pair for the other factor)
//   PyBMF/models/BinaryMFPenalty.py:139-142,154-157 and PyBMF/models/WNMF.py:98-106 with W = 'mask' (the pattern of stored
//   entries of a csr X_train, explicit zeros included -- models/ContinuousModel.py:52-63) or an explicit weight matrix.
//
// With a general mask the product (U V^T) cannot be re-associated; but only the observed cells matter,
// so this is SDDMM + SpMM fused over a CSR (or, for the other factor, CSC) list of observed cells: for cell e = (r, j) with
// value x_e and weight w_e:   p_e = <F_self[r], F_other[j]>,  num[r] += w_e x_e F_other[j],  den[r] += w_e p_e F_other[j],
// and (optionally) sums += { w_e (x_e - p_e)^2 , w_e |x_e - p_e| }  -> rec_error over the observed cells (:175-179).
// One wave per row, lane = factor column; the cell list of a row is fetched 64 cells at a time by one vector load and
// broadcast with v_readlane; four cells are in flight per trip (independent 256-byte gathers of F_other rows).
#include "common.h"

namespace {

template <int KP>
__global__ __launch_bounds__(256) void masked_pass_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                           const float* __restrict__ val, const float* __restrict__ wgt,
                                                           int rows, const float* __restrict__ Fself,
                                                           const float* __restrict__ Fother, float* __restrict__ num,
                                                           float* __restrict__ den, double* __restrict__ sums) {
    __shared__ double red[4][2];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool on = lane < KP;
    double s2 = 0.0, s1 = 0.0;  // wave-uniform partial sums (every lane carries the same value)
    for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
        const float u = on ? Fself[(int64_t)r * KP + lane] : 0.f;
        const int64_t e0 = ptr[r], e1 = ptr[r + 1];
        float nacc = 0.f, dacc = 0.f;
        for (int64_t base = e0; base < e1; base += 64) {
            const int cnt = (int)min((int64_t)64, e1 - base);
            const int64_t me = min(base + lane, e1 - 1);
            const int my_j = idx[me];
            const float my_x = val[me];
            const float my_w = wgt ? wgt[me] : 1.f;
            for (int q0 = 0; q0 < cnt; q0 += 4) {
                int j[4];
                float x[4], w[4], v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int qq = min(q0 + q, cnt - 1);  // tail: repeat the last cell with weight 0
                    j[q] = __builtin_amdgcn_readlane(my_j, qq);
                    x[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_x), qq));
                    w[q] = (q0 + q < cnt) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), qq)) : 0.f;
                    v[q] = on ? Fother[(int64_t)j[q] * KP + lane] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float p = wave_sum(u * v[q]);
                    nacc = fmaf(w[q] * x[q], v[q], nacc);
                    dacc = fmaf(w[q] * p, v[q], dacc);
                    const double d = (double)x[q] - (double)p;
                    s2 += (double)w[q] * d * d;
                    s1 += (double)w[q] * fabs(d);
                }
            }
        }
        if (on) {
            num[(int64_t)r * KP + lane] = nacc;
            den[(int64_t)r * KP + lane] = dacc;
        }
    }
    if (sums) {  // one atomic pair per block (same-address atomics serialise at ~12 ns each)
        if (lane == 0) { red[wave][0] = s2; red[wave][1] = s1; }
        __syncthreads();
        if (threadIdx.x < 2) atomicAdd(&sums[threadIdx.x], ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x]);
    }
}

}  // namespace

extern "C" int bmf_masked_pass(const int64_t* ptr, const int32_t* idx, const float* val, const float* wgt, int32_t rows,
                               const float* Fself, const float* Fother, int kp, float* num, float* den, double* sums,
                               void* stream) {
    BMF_REQUIRE(ptr && idx && val && Fself && Fother && num && den, "bmf_masked_pass: null pointer");
    BMF_REQUIRE(rows >= 1, "bmf_masked_pass: rows must be positive");
    BMF_REQUIRE(kp == 32 || kp == 64, "bmf_masked_pass: kp must be 32 or 64");
    const int blocks = (rows + 3) / 4;
    dim3 grid((unsigned)(blocks < 4096 ? blocks : 4096)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (kp == 32) BMF_LAUNCH(masked_pass_kernel<32>, grid, block, 0, s, ptr, idx, val, wgt, rows, Fself, Fother, num, den, sums);
    else BMF_LAUNCH(masked_pass_kernel<64>, grid, block, 0, s, ptr, idx, val, wgt, rows, Fself, Fother, num, den, sums);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}
