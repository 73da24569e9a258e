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
//
// Recommender-style data has power-law rows (one user may hold thousands of cells), and the per-cell work is a dependent
// gather -> dot -> FMA chain, so a wave per row is latency-bound on the longest row (0.8 ms at MovieLens-1M shape).  Work
// is therefore cut into SEGMENTS of at most 64 consecutive cells of one row (table built on the host once): pass 1 gives
// every segment to a wave (lane = factor column; the segment's cell list is one vector load, broadcast with v_readlane;
// eight independent 128/256-byte gathers in flight), pass 2 adds the segment partials of each row in segment order --
// deterministic, no float atomics on the outputs.
#include "common.h"

namespace {

// LINK = 0: the plain product above.  LINK = BMF_LINK_SIGMOID (PNLPF under a mask, PyBMF/models/PNLPF.py:61-91): with s = lamda (p - 1/2),
// sig = sigmoid(s), d = sig (1 - sig):  num[r] += lamda w x d F_other[j],  den[r] += lamda w sig d F_other[j], and the sums are
// taken against the link prediction sig (rec_error of the inherited loop: 0.5 sum W o (X - sigmoid(S))^2, BinaryMFPenalty.py:175).
// G = lanes per cell: 64 (the wave takes one cell at a time, lane = factor column), or 32 / 16 when the factor is that narrow
// (k <= 32 / k <= 16): the wave then takes 2 / 4 cells per step, one per lane group -- a quarter of the gather instructions and of
// the dependent steps per segment, and the dot product of a cell is the DPP sum of its own 16-lane row(s), no read-out needed.
template <int KP, int LINK, int G>
__global__ __launch_bounds__(256) void masked_segments_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                               const float* __restrict__ val, const float* __restrict__ wgt,
                                                               const int32_t* __restrict__ seg_row,
                                                               const int64_t* __restrict__ seg_beg, int nseg,
                                                               const float* __restrict__ Fself,
                                                               const float* __restrict__ Fother, float* __restrict__ part,
                                                               double* __restrict__ sums, float lamda) {
    static_assert(G == 16 || G == 32 || G == 64, "lanes per cell");
    constexpr int CPS = 64 / G;   // cells per step
    __shared__ double red[4][2];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = lane / G, cl = lane % G;
    const bool on = cl < KP;
    double s2 = 0.0, s1 = 0.0;  // per lane group (every lane of a group carries the same value)
    for (int sg = blockIdx.x * 4 + wave; sg < nseg; sg += gridDim.x * 4) {
        const int r = seg_row[sg];
        const int64_t base = seg_beg[sg];
        const int cnt = (int)min((int64_t)64, ptr[r + 1] - base);
        const float u = on ? Fself[(int64_t)r * KP + cl] : 0.f;
        const int64_t me = base + min(lane, cnt - 1);
        const int my_j = idx[me];
        const float my_x = val[me];
        const float my_w = wgt ? wgt[me] : 1.f;
        float nacc = 0.f, dacc = 0.f;
        for (int q0 = 0; q0 < cnt; q0 += 8 * CPS) {
            float x[8], w[8], v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int cell = q0 + q * CPS + grp;
                const int qq = min(cell, cnt - 1);  // tail: repeat the last cell with weight 0
                int j;
                if constexpr (G == 64) {
                    j = __builtin_amdgcn_readlane(my_j, qq);
                    x[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_x), qq));
                    w[q] = (cell < cnt) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), qq)) : 0.f;
                } else {   // lane groups look at different cells: fetch this group's cell from the lane that loaded it
                    j = __shfl(my_j, qq, 64);
                    x[q] = __shfl(my_x, qq, 64);
                    const float wq = __shfl(my_w, qq, 64);
                    w[q] = (cell < cnt) ? wq : 0.f;
                }
                v[q] = on ? Fother[(int64_t)j * KP + cl] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float p;
                if constexpr (G == 64) {
                    p = wave_sum_dpp<(KP <= 32 ? 32 : 64)>(u * v[q]);   // (lanes >= KP hold 0)
                } else {
                    p = u * v[q];
                    p += bmf_dpp_f32<0xB1>(p);
                    p += bmf_dpp_f32<0x4E>(p);
                    p += bmf_dpp_f32<0x141>(p);
                    p += bmf_dpp_f32<0x140>(p);              // every lane of a 16-lane row: the row's sum
                    if constexpr (G == 32) p += __shfl_xor(p, 16, 64);
                }
                float cn = w[q] * x[q], cd = w[q] * p, pred = p;
                if constexpr (LINK == BMF_LINK_SIGMOID) {
                    const float sarg = lamda * (p - 0.5f);
                    const float e = __expf(-fabsf(sarg));            // sigma (1 - sigma) = e / (1 + e)^2 without cancellation
                    const float r1 = 1.0f / (1.0f + e);
                    pred = sarg >= 0.f ? r1 : e * r1;
                    const float dsig = e * r1 * r1;
                    cn = lamda * w[q] * x[q] * dsig;
                    cd = lamda * w[q] * pred * dsig;
                }
                if constexpr (LINK == BMF_LINK_KL) {
                    // WNMF, Kullback-Leibler loss under a weight matrix (models/WNMF.py:111-129): the numerator is (W o X / U V^T) F_other, the
                    // denominator O F_other uses the ALL-ONES matrix (the caller supplies the column sums); sums[0] takes TWICE the objective
                    // sum w (x log(x / p) - x + p), 0 log 0 = 0 (:143-145), so that the caller's 0.5 sums[0] is the error as for the other models
                    const float pp = fmaxf(p, 1e-30f);
                    cn = x[q] != 0.f ? w[q] * x[q] / pp : 0.f;
                    cd = 0.f;
                    const double kl = (x[q] != 0.f ? (double)x[q] * log((double)x[q] / (double)pp) : 0.0) - (double)x[q] + (double)p;
                    nacc = fmaf(cn, v[q], nacc);
                    s2 += 2.0 * (double)w[q] * kl;
                    continue;
                }
                nacc = fmaf(cn, v[q], nacc);
                dacc = fmaf(cd, v[q], dacc);
                const double d = (double)x[q] - (double)pred;
                s2 += (double)w[q] * d * d;
                s1 += (double)w[q] * fabs(d);
            }
        }
        if constexpr (G < 64) {   // the groups' partial sums of a column -> lanes 0 .. G - 1
            if constexpr (G == 16) {
                nacc += __shfl_xor(nacc, 16, 64);
                dacc += __shfl_xor(dacc, 16, 64);
            }
            nacc += __shfl_xor(nacc, 32, 64);
            dacc += __shfl_xor(dacc, 32, 64);
        }
        if (lane < KP) {   // (columns G .. KP - 1 of a narrow factor are padding: zeros)
            part[(int64_t)sg * 2 * KP + lane] = lane < G ? nacc : 0.f;
            part[(int64_t)sg * 2 * KP + KP + lane] = lane < G ? dacc : 0.f;
        }
    }
    if (sums) {  // one atomic pair per block (same-address atomics serialise at ~12 ns each)
        if constexpr (G < 64) {   // one lane per group carries the group's sums; the wave's total is their sum
            s2 = wave_sum(cl == 0 ? s2 : 0.0);
            s1 = wave_sum(cl == 0 ? s1 : 0.0);
        }
        if (lane == 0) { red[wave][0] = s2; red[wave][1] = s1; }
        __syncthreads();
        if (threadIdx.x < 2) atomicAdd(&sums[threadIdx.x], ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x]);
    }
}

// The same pass for a factor of TWO 64-column blocks (64 < k <= 128, pybmf_amd/wide.py): p_e is the dot product over both blocks, the
// numerator / denominator partials are kept per block (part0, part1: the layout of the one-block kernel each, so masked_rows_kernel
// serves both).  Plain product only (no link).
__global__ __launch_bounds__(256) void masked_segments_wide_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                                    const float* __restrict__ val, const float* __restrict__ wgt,
                                                                    const int32_t* __restrict__ seg_row,
                                                                    const int64_t* __restrict__ seg_beg, int nseg,
                                                                    const float* __restrict__ Fself0, const float* __restrict__ Fself1,
                                                                    const float* __restrict__ Fother0, const float* __restrict__ Fother1,
                                                                    float* __restrict__ part0, float* __restrict__ part1,
                                                                    double* __restrict__ sums) {
    constexpr int KP = 64;
    __shared__ double red[4][2];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double s2 = 0.0, s1 = 0.0;
    for (int sg = blockIdx.x * 4 + wave; sg < nseg; sg += gridDim.x * 4) {
        const int r = seg_row[sg];
        const int64_t base = seg_beg[sg];
        const int cnt = (int)min((int64_t)64, ptr[r + 1] - base);
        const float u0 = Fself0[(int64_t)r * KP + lane], u1 = Fself1[(int64_t)r * KP + lane];
        const int64_t me = base + min(lane, cnt - 1);
        const int my_j = idx[me];
        const float my_x = val[me];
        const float my_w = wgt ? wgt[me] : 1.f;
        float n0 = 0.f, d0 = 0.f, n1 = 0.f, d1 = 0.f;
        for (int q0 = 0; q0 < cnt; q0 += 8) {
            float x[8], w[8], v0[8], v1[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int cell = q0 + q;
                const int qq = min(cell, cnt - 1);  // tail: repeat the last cell with weight 0
                const int j = __builtin_amdgcn_readlane(my_j, qq);
                x[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_x), qq));
                w[q] = (cell < cnt) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), qq)) : 0.f;
                v0[q] = Fother0[(int64_t)j * KP + lane];
                v1[q] = Fother1[(int64_t)j * KP + lane];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float p = wave_sum_dpp<64>(fmaf(u1, v1[q], u0 * v0[q]));   // (padding columns of both blocks hold 0)
                const float cn = w[q] * x[q], cd = w[q] * p;
                n0 = fmaf(cn, v0[q], n0);
                d0 = fmaf(cd, v0[q], d0);
                n1 = fmaf(cn, v1[q], n1);
                d1 = fmaf(cd, v1[q], d1);
                const double d = (double)x[q] - (double)p;
                s2 += (double)w[q] * d * d;
                s1 += (double)w[q] * fabs(d);
            }
        }
        part0[(int64_t)sg * 2 * KP + lane] = n0;
        part0[(int64_t)sg * 2 * KP + KP + lane] = d0;
        part1[(int64_t)sg * 2 * KP + lane] = n1;
        part1[(int64_t)sg * 2 * KP + KP + lane] = d1;
    }
    if (sums) {
        if (lane == 0) { red[wave][0] = s2; red[wave][1] = s1; }
        __syncthreads();
        if (threadIdx.x < 2) atomicAdd(&sums[threadIdx.x], ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x]);
    }
}

// num[r] / den[r] = sum of the row's segment partials, in segment order (a row without observed cells gets zeros)
template <int KP>
__global__ __launch_bounds__(256) void masked_rows_kernel(const int64_t* __restrict__ row_seg_ptr, int rows,
                                                           const float* __restrict__ part, float* __restrict__ num,
                                                           float* __restrict__ den) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int r = (int)(i / KP), c = (int)(i % KP);
    if (r >= rows) return;
    float a = 0.f, b = 0.f;
    // eight segments' loads in flight (a user with thousands of cells has dozens of segments: one dependent load per segment made this
    // small kernel 19 us at MovieLens-1M shape); the order of the additions is fixed
    const int64_t s1 = row_seg_ptr[r + 1];
    int64_t sg = row_seg_ptr[r];
    for (; sg + 8 <= s1; sg += 8) {
        float ta[8], tb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            ta[q] = part[(sg + q) * 2 * KP + c];
            tb[q] = part[(sg + q) * 2 * KP + KP + c];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            a += ta[q];
            b += tb[q];
        }
    }
    for (; sg < s1; ++sg) {
        a += part[sg * 2 * KP + c];
        b += part[sg * 2 * KP + KP + c];
    }
    num[i] = a;
    den[i] = b;
}

// Thresholding objective over the observed cells (PyBMF/models/BinaryMFThreshold.py:150-207 with a mask W):
//   out[0] += sum_e (w_e (x_e - p_e))^2          (F = 0.5 * out[0]; note the squared weight, :169-170)
//   out[1] += sum_e w_e (x_e - p_e) <dUs[i], Vs[j]>      out[2] += sum_e w_e (x_e - p_e) <Us[i], dVs[j]>     (dF, :195-206)
// with p_e = <Us[i], Vs[j]>, Us = sigmoid(lam (U - u)), dUs = dXdx(U, u) etc. (prepared by the transform kernel).
template <int KP, bool GRAD>
__global__ __launch_bounds__(256) void masked_thresh_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                             const float* __restrict__ val, const float* __restrict__ wgt,
                                                             const int32_t* __restrict__ seg_row,
                                                             const int64_t* __restrict__ seg_beg, int nseg,
                                                             const float* __restrict__ Us, const float* __restrict__ dUs,
                                                             const float* __restrict__ Vs, const float* __restrict__ dVs,
                                                             double* __restrict__ out) {
    __shared__ double red[4][3];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool on = lane < KP;
    double f = 0.0, g1 = 0.0, g2 = 0.0;
    for (int sg = blockIdx.x * 4 + wave; sg < nseg; sg += gridDim.x * 4) {
        const int r = seg_row[sg];
        const int64_t base = seg_beg[sg];
        const int cnt = (int)min((int64_t)64, ptr[r + 1] - base);
        const float u = on ? Us[(int64_t)r * KP + lane] : 0.f;
        const float du = (GRAD && on) ? dUs[(int64_t)r * KP + lane] : 0.f;
        const int64_t me = base + min(lane, cnt - 1);
        const int my_j = idx[me];
        const float my_x = val[me];
        const float my_w = wgt ? wgt[me] : 1.f;
        for (int q0 = 0; q0 < cnt; q0 += 4) {
            float x[4], w[4], v[4], dv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int qq = min(q0 + q, cnt - 1);
                const int j = __builtin_amdgcn_readlane(my_j, qq);
                x[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_x), qq));
                w[q] = (q0 + q < cnt) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), qq)) : 0.f;
                v[q] = on ? Vs[(int64_t)j * KP + lane] : 0.f;
                dv[q] = (GRAD && on) ? dVs[(int64_t)j * KP + lane] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double rr = (double)w[q] * ((double)x[q] - (double)wave_sum_dpp<(KP <= 32 ? 32 : 64)>(u * v[q]));
                f += rr * rr;
                if (GRAD) {
                    g1 += rr * (double)wave_sum_dpp<(KP <= 32 ? 32 : 64)>(du * v[q]);
                    g2 += rr * (double)wave_sum_dpp<(KP <= 32 ? 32 : 64)>(u * dv[q]);
                }
            }
        }
    }
    if (lane == 0) { red[wave][0] = f; red[wave][1] = g1; red[wave][2] = g2; }
    __syncthreads();
    if (threadIdx.x < 3) atomicAdd(&out[threadIdx.x], ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x]);
}

// Confusion counts over the observed cells only (task='prediction': PyBMF/utils/evaluate_utils.py:32-44 gathers X_pd at the
// stored entries, then utils/metrics.py:56-77 on the two 1-D vectors): pd_e = (rowbits_self[i] & rowbits_other[j]) != 0,
// gt_e = (x_e != 0).  counts += {TP, FP, FN, TN}.  Thread per cell, one atomic quadruple per block.
__global__ __launch_bounds__(256) void masked_counts_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                             const float* __restrict__ val, int rows, int64_t nnz,
                                                             const int32_t* __restrict__ cell_row,
                                                             const uint64_t* __restrict__ bits_self,
                                                             const uint64_t* __restrict__ bits_other,
                                                             const uint64_t* __restrict__ bits_self1,
                                                             const uint64_t* __restrict__ bits_other1,
                                                             unsigned long long* __restrict__ counts) {
    __shared__ unsigned red[4][4];
    unsigned c[4] = {0u, 0u, 0u, 0u};
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * 256) {
        bool pd = (bits_self[cell_row[e]] & bits_other[idx[e]]) != 0ull;
        if (bits_self1) pd = pd || (bits_self1[cell_row[e]] & bits_other1[idx[e]]) != 0ull;   // second block of 64 factors (64 < k <= 128)
        const bool gt = val[e] != 0.f;
        // 0: TP, 1: FP, 2: FN, 3: TN  (four predicated adds: a dynamically indexed c[] lives in scratch memory)
        c[0] += (gt && pd) ? 1u : 0u;
        c[1] += (!gt && pd) ? 1u : 0u;
        c[2] += (gt && !pd) ? 1u : 0u;
        c[3] += (!gt && !pd) ? 1u : 0u;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned t = wave_sum(c[q]);
        if (lane == 0) red[wave][q] = t;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const unsigned long long t = (unsigned long long)red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (t) atomicAdd(&counts[threadIdx.x], t);
    }
    (void)ptr; (void)rows;
}

// The scalars of one masked iteration in one launch: out[0] = sums[0] (residual sum of squares over the observed cells, from the pass),
// out[1] = sum_b partU[2 b], out[2] = sum_b partV[2 b] (the regulariser partials of the two epilogues), out[3], out[4] = sums2[0], sums2[1]
// (whole-matrix |.| and (.)^2 sums), out[5], out[6] = counts[0], counts[1] (cover count); sums2 and counts are reset for the next
// iteration.  They were a dozen small torch launches per iteration -- more host time than the kernels take at MovieLens-1M size.
__global__ __launch_bounds__(256) void masked_scalars_kernel(const double* __restrict__ sums, const double* __restrict__ partU, int nbU,
                                                              const double* __restrict__ partV, int nbV, double* __restrict__ sums2,
                                                              unsigned long long* __restrict__ counts, double* __restrict__ out) {
    __shared__ double red[2][4];
    const int t = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int i = t; i < nbU; i += 256) a += partU[2 * i];
    for (int i = t; i < nbV; i += 256) b += partV[2 * i];
    a = wave_sum(a);
    b = wave_sum(b);
    if ((t & 63) == 0) { red[0][t >> 6] = a; red[1][t >> 6] = b; }
    __syncthreads();
    if (t == 0) {
        out[0] = sums[0];
        out[1] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        out[2] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        out[3] = sums2 ? sums2[0] : 0.0;
        out[4] = sums2 ? sums2[1] : 0.0;
        out[5] = counts ? (double)counts[0] : 0.0;   // exact: counts < 2^53
        out[6] = counts ? (double)counts[1] : 0.0;
        if (sums2) { sums2[0] = 0.0; sums2[1] = 0.0; }
        if (counts) { counts[0] = 0ull; counts[1] = 0ull; }
        // word 7 last, behind a system-scope fence: a host that polls it (out may be pinned host memory) then sees the other seven
        __threadfence_system();
        out[7] = 0.0;
    }
}

}  // namespace

extern "C" int bmf_masked_counts(const int32_t* cell_row, const int32_t* idx, const float* val, int64_t nnz,
                                 const uint64_t* bits_self, const uint64_t* bits_other, unsigned long long* counts,
                                 void* stream) {
    BMF_REQUIRE(cell_row && idx && val && bits_self && bits_other && counts, "bmf_masked_counts: null pointer");
    BMF_REQUIRE(nnz >= 1, "bmf_masked_counts: no observed cells");
    const int64_t blocks = (nnz + 255) / 256;
    // (at most one workgroup per CU: every workgroup ends with an atomic quadruple on the same four words, ~12 ns each in turn -- 2048
    // of them were 25 of this kernel's 29 us at 850 k cells)
    BMF_LAUNCH(masked_counts_kernel, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(256), 0, (hipStream_t)stream, nullptr, idx, val,
               0, nnz, cell_row, bits_self, bits_other, nullptr, nullptr, counts);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_masked_counts_wide(const int32_t* cell_row, const int32_t* idx, const float* val, int64_t nnz, const uint64_t* bits_self0,
                                      const uint64_t* bits_self1, const uint64_t* bits_other0, const uint64_t* bits_other1,
                                      unsigned long long* counts, void* stream) {
    BMF_REQUIRE(cell_row && idx && val && bits_self0 && bits_self1 && bits_other0 && bits_other1 && counts, "bmf_masked_counts_wide: null pointer");
    BMF_REQUIRE(nnz >= 1, "bmf_masked_counts_wide: no observed cells");
    const int64_t blocks = (nnz + 255) / 256;
    BMF_LAUNCH(masked_counts_kernel, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(256), 0, (hipStream_t)stream, nullptr, idx, val,
               0, nnz, cell_row, bits_self0, bits_other0, bits_self1, bits_other1, counts);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

// The "confusion sums" of a REAL-valued ground truth against the Boolean product of the thresholded factors, over the whole matrix,
// with the arithmetic the reference's metrics apply to two csr matrices under task='reconstruction' (utils/evaluate_utils.py:46-51,
// utils/metrics.py:56-77,161-170): out[0] += TP = sum gt pd, [1] FP = sum max(pd - gt, 0), [2] FN = sum max(gt - pd, 0),
// [3] TN = sum (1 - gt)(1 - pd)  (TP of the inverted pair), [4] sum gt, [5] sum pd; pd = (rowbits_u[i] & rowbits_v[j]) != 0.
namespace {
__global__ __launch_bounds__(256) void real_confusion_kernel(const float* __restrict__ X, int64_t ld, int m, int n,
                                                              const uint64_t* __restrict__ ubits, const uint64_t* __restrict__ vbits,
                                                              double* __restrict__ out) {
    __shared__ double red[4][6];
    double a[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int i = blockIdx.x; i < m; i += gridDim.x) {
        const uint64_t ub = ubits[i];
        const float* xr = X + (int64_t)i * ld;
        for (int j = threadIdx.x; j < n; j += 256) {
            const double gt = (double)xr[j];
            const double pd = (ub & vbits[j]) != 0ull ? 1.0 : 0.0;
            a[0] += gt * pd;
            a[1] += fmax(pd - gt, 0.0);
            a[2] += fmax(gt - pd, 0.0);
            a[3] += (1.0 - gt) * (1.0 - pd);
            a[4] += gt;
            a[5] += pd;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const double t = wave_sum(a[q]);
        if (lane == 0) red[wave][q] = t;
    }
    __syncthreads();
    if (threadIdx.x < 6) atomicAdd(&out[threadIdx.x], ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x]);
}
}  // namespace

extern "C" int bmf_real_confusion(const float* X, int64_t ld, int32_t m, int32_t n, const uint64_t* ubits, const uint64_t* vbits, double* out,
                                  void* stream) {
    BMF_REQUIRE(X && ubits && vbits && out, "bmf_real_confusion: null pointer");
    BMF_REQUIRE(m >= 1 && n >= 1 && ld >= n, "bmf_real_confusion: bad shape");
    BMF_LAUNCH(real_confusion_kernel, dim3((unsigned)(m < 1024 ? m : 1024)), dim3(256), 0, (hipStream_t)stream, X, ld, (int)m, (int)n, ubits, vbits, out);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_masked_thresh(const int64_t* ptr, const int32_t* idx, const float* val, const float* wgt,
                                 const int32_t* seg_row, const int64_t* seg_beg, int32_t nseg, const float* Us,
                                 const float* dUs, const float* Vs, const float* dVs, int kp, double* out, void* stream) {
    BMF_REQUIRE(ptr && idx && val && seg_row && seg_beg && Us && Vs && out, "bmf_masked_thresh: null pointer");
    BMF_REQUIRE((dUs == nullptr) == (dVs == nullptr), "bmf_masked_thresh: give both derivative factors or neither");
    BMF_REQUIRE(kp == 32 || kp == 64, "bmf_masked_thresh: kp must be 32 or 64");
    BMF_REQUIRE(nseg >= 1, "bmf_masked_thresh: no observed cells");
    const int blocks = (nseg + 3) / 4;
    dim3 grid((unsigned)(blocks < 8192 ? blocks : 8192)), block(256);
    hipStream_t s = (hipStream_t)stream;
    const bool grad = dUs != nullptr;
#define BMF_MT(KP_, G_) BMF_LAUNCH((masked_thresh_kernel<KP_, G_>), grid, block, 0, s, ptr, idx, val, wgt, seg_row, seg_beg, nseg, Us, dUs, Vs, dVs, out)
    if (kp == 32) { if (grad) BMF_MT(32, true); else BMF_MT(32, false); }
    else { if (grad) BMF_MT(64, true); else BMF_MT(64, false); }
#undef BMF_MT
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

static int masked_pass_launch(const int64_t* ptr, const int32_t* idx, const float* val, const float* wgt, int32_t rows,
                              const int32_t* seg_row, const int64_t* seg_beg, int32_t nseg, const int64_t* row_seg_ptr,
                              const float* Fself, const float* Fother, int kp, float* part, float* num, float* den,
                              double* sums, int link, double lamda, int kcols, hipStream_t s, const char* who) {
    BMF_REQUIRE(ptr && idx && val && seg_row && seg_beg && row_seg_ptr && Fself && Fother && part && num && den,
                "%s: null pointer", who);
    BMF_REQUIRE(kcols >= 1 && kcols <= kp, "%s: kcols must be 1..kp", who);
    BMF_REQUIRE(rows >= 1 && nseg >= 0, "%s: rows must be positive, nseg non-negative", who);
    BMF_REQUIRE(kp == 32 || kp == 64, "%s: kp must be 32 or 64", who);
    BMF_REQUIRE(link == 0 || link == BMF_LINK_SIGMOID || link == BMF_LINK_KL, "%s: link must be 0, BMF_LINK_SIGMOID or BMF_LINK_KL", who);
    if (nseg > 0) {
        const int blocks = (nseg + 3) / 4;
        dim3 grid((unsigned)(blocks < 8192 ? blocks : 8192)), block(256);
#define BMF_MS(KP_, LK_, G_) BMF_LAUNCH((masked_segments_kernel<KP_, LK_, G_>), grid, block, 0, s, ptr, idx, val, wgt, seg_row, seg_beg, nseg, Fself, Fother, part, sums, (float)lamda)
        // lanes per cell: the narrowest group that holds the k real columns (kcols; the padding columns of both factors are zero)
        const int g = kp == 32 ? (kcols <= 16 ? 16 : 32) : 64;
        if (kp == 32) {
            if (g == 16) { if (link == BMF_LINK_KL) BMF_MS(32, BMF_LINK_KL, 16); else if (link) BMF_MS(32, BMF_LINK_SIGMOID, 16); else BMF_MS(32, 0, 16); }
            else if (g == 32) { if (link == BMF_LINK_KL) BMF_MS(32, BMF_LINK_KL, 32); else if (link) BMF_MS(32, BMF_LINK_SIGMOID, 32); else BMF_MS(32, 0, 32); }
            else { if (link == BMF_LINK_KL) BMF_MS(32, BMF_LINK_KL, 64); else if (link) BMF_MS(32, BMF_LINK_SIGMOID, 64); else BMF_MS(32, 0, 64); }
        } else { if (link == BMF_LINK_KL) BMF_MS(64, BMF_LINK_KL, 64); else if (link) BMF_MS(64, BMF_LINK_SIGMOID, 64); else BMF_MS(64, 0, 64); }
#undef BMF_MS
    }
    const int64_t total = (int64_t)rows * kp;
    dim3 grid2((unsigned)((total + 255) / 256)), block2(256);
    if (kp == 32) BMF_LAUNCH(masked_rows_kernel<32>, grid2, block2, 0, s, row_seg_ptr, rows, part, num, den);
    else BMF_LAUNCH(masked_rows_kernel<64>, grid2, block2, 0, s, row_seg_ptr, rows, part, num, den);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_masked_pass(const int64_t* ptr, const int32_t* idx, const float* val, const float* wgt, int32_t rows,
                               const int32_t* seg_row, const int64_t* seg_beg, int32_t nseg, const int64_t* row_seg_ptr,
                               const float* Fself, const float* Fother, int kp, float* part, float* num, float* den,
                               double* sums, void* stream) {
    return masked_pass_launch(ptr, idx, val, wgt, rows, seg_row, seg_beg, nseg, row_seg_ptr, Fself, Fother, kp, part, num, den, sums, 0, 0.0, kp,
                              (hipStream_t)stream, "bmf_masked_pass");
}

extern "C" int bmf_masked_link_pass(const int64_t* ptr, const int32_t* idx, const float* val, const float* wgt, int32_t rows,
                                    const int32_t* seg_row, const int64_t* seg_beg, int32_t nseg, const int64_t* row_seg_ptr,
                                    const float* Fself, const float* Fother, int kp, float* part, float* num, float* den,
                                    double* sums, int link, double lamda, void* stream) {
    return masked_pass_launch(ptr, idx, val, wgt, rows, seg_row, seg_beg, nseg, row_seg_ptr, Fself, Fother, kp, part, num, den, sums, link, lamda, kp,
                              (hipStream_t)stream, "bmf_masked_link_pass");
}

extern "C" int bmf_masked_link_pass_k(const int64_t* ptr, const int32_t* idx, const float* val, const float* wgt, int32_t rows,
                                      const int32_t* seg_row, const int64_t* seg_beg, int32_t nseg, const int64_t* row_seg_ptr,
                                      const float* Fself, const float* Fother, int kp, int kcols, float* part, float* num, float* den,
                                      double* sums, int link, double lamda, void* stream) {
    return masked_pass_launch(ptr, idx, val, wgt, rows, seg_row, seg_beg, nseg, row_seg_ptr, Fself, Fother, kp, part, num, den, sums, link, lamda, kcols,
                              (hipStream_t)stream, "bmf_masked_link_pass_k");
}

extern "C" int bmf_masked_pass_wide(const int64_t* ptr, const int32_t* idx, const float* val, const float* wgt, int32_t rows,
                                    const int32_t* seg_row, const int64_t* seg_beg, int32_t nseg, const int64_t* row_seg_ptr,
                                    const float* Fself0, const float* Fself1, const float* Fother0, const float* Fother1, float* part0,
                                    float* part1, float* num0, float* num1, float* den0, float* den1, double* sums, void* stream) {
    BMF_REQUIRE(ptr && idx && val && seg_row && seg_beg && row_seg_ptr && Fself0 && Fself1 && Fother0 && Fother1 && part0 && part1 && num0 &&
                    num1 && den0 && den1,
                "bmf_masked_pass_wide: null pointer");
    BMF_REQUIRE(rows >= 1 && nseg >= 0, "bmf_masked_pass_wide: rows must be positive, nseg non-negative");
    hipStream_t s = (hipStream_t)stream;
    if (nseg > 0) {
        const int blocks = (nseg + 3) / 4;
        BMF_LAUNCH(masked_segments_wide_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s, ptr, idx, val, wgt, seg_row, seg_beg,
                   nseg, Fself0, Fself1, Fother0, Fother1, part0, part1, sums);
    }
    const int64_t total = (int64_t)rows * 64;
    dim3 grid2((unsigned)((total + 255) / 256)), block2(256);
    BMF_LAUNCH(masked_rows_kernel<64>, grid2, block2, 0, s, row_seg_ptr, rows, part0, num0, den0);
    BMF_LAUNCH(masked_rows_kernel<64>, grid2, block2, 0, s, row_seg_ptr, rows, part1, num1, den1);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_masked_scalars(const double* sums, const double* partU, int nbU, const double* partV, int nbV, double* sums2,
                                  unsigned long long* counts, double* out, void* stream) {
    BMF_REQUIRE(sums && partU && partV && out, "bmf_masked_scalars: null pointer");
    BMF_REQUIRE(nbU >= 1 && nbV >= 1, "bmf_masked_scalars: bad lengths");
    BMF_LAUNCH(masked_scalars_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sums, partU, nbU, partV, nbV, sums2, counts, out);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

// ---- one whole masked iteration per call (see bmf_hip.h) ----
static int masked_side_pass(const bmf_masked_loop* st, const bmf_masked_side* sd, const float* Fself, const float* Fother, float* num, float* den,
                            double* sums, void* stream) {
    return bmf_masked_link_pass_k(sd->ptr, sd->idx, sd->val, sd->wgt, sd->rows, sd->seg_row, sd->seg_beg, sd->nseg, sd->row_seg_ptr, Fself, Fother,
                                  st->kp, st->k, sd->part, num, den, sums, st->link, st->lamda, stream);
}

extern "C" int bmf_masked_iterate(const bmf_masked_loop* st, double reg, int with_update, double* host_row, void* stream) {
    BMF_REQUIRE(st && host_row, "bmf_masked_iterate: null pointer");
    BMF_REQUIRE(st->struct_bytes == (int32_t)sizeof(bmf_masked_loop), "bmf_masked_iterate: struct_bytes=%d, library expects %d", st->struct_bytes,
                (int)sizeof(bmf_masked_loop));
    BMF_REQUIRE(st->link == 0 || st->link == BMF_LINK_SIGMOID, "bmf_masked_iterate: link must be 0 or BMF_LINK_SIGMOID");
    BMF_REQUIRE(st->sums && st->sums2 && st->counts && st->Up64 && st->Vp64 && st->epiU.F64 && st->epiV.F64 && st->epiU.num && st->epiV.num &&
                    st->epiU.den && st->epiV.den,
                "bmf_masked_iterate: null device pointer in state");
    hipStream_t s = (hipStream_t)stream;
    if (with_update) {
        BMF_HIP_CHECK(hipMemcpyAsync(st->Up64, st->epiU.F64, (size_t)st->epiU.rows_pad * st->kp * sizeof(double), hipMemcpyDeviceToDevice, s));
        BMF_HIP_CHECK(hipMemcpyAsync(st->Vp64, st->epiV.F64, (size_t)st->epiV.rows_pad * st->kp * sizeof(double), hipMemcpyDeviceToDevice, s));
        bmf_epilogue_args ev = st->epiV, eu = st->epiU;
        ev.reg = reg; eu.reg = reg;
        int rc = bmf_mu_epilogue(&ev, stream);                                                                         // V <- ...
        if (rc != BMF_OK) return rc;
        rc = masked_side_pass(st, &st->csr, eu.F, ev.F, const_cast<float*>(eu.num), const_cast<float*>(eu.den), nullptr, stream);   // U-side sums with the new V
        if (rc != BMF_OK) return rc;
        rc = bmf_mu_epilogue(&eu, stream);                                                                             // U <- ...
        if (rc != BMF_OK) return rc;
        BMF_HIP_CHECK(hipMemsetAsync(st->sums, 0, 4 * sizeof(double), s));
        rc = masked_side_pass(st, &st->csc, ev.F, eu.F, const_cast<float*>(ev.num), const_cast<float*>(ev.den), st->sums, stream);  // for the next V update + rec_error
        if (rc != BMF_OK) return rc;
    }
    double* sums2 = nullptr;
    unsigned long long* counts = nullptr;
    if (st->Xbits) {
        int rc = st->link ? bmf_link_sums(st->Xbits, st->x_m_pad, st->ldx, st->m, st->n, st->epiU.F, st->epiV.F, st->x_n_pad, st->kp, st->link, st->lamda,
                                          nullptr, st->sums2, stream)
                          : bmf_residual_sums(st->Xbits, st->x_m_pad, st->ldx, st->m, st->n, st->epiU.F, st->epiV.F, st->kp, st->sums2, nullptr, stream);
        if (rc != BMF_OK) return rc;
        rc = bmf_cover_count(st->Xbits, st->x_m_pad, st->ldx, st->x_n_pad / 32, st->epiU.rowbits, st->epiV.colbits, st->x_n_pad / 32, st->kp, st->counts,
                             nullptr, stream);
        if (rc != BMF_OK) return rc;
        sums2 = st->sums2; counts = st->counts;
    } else if (st->Xreal) {
        const int rc = bmf_residual_sums_f32(st->Xreal, st->r_m_pad, st->r_n_pad, st->m, st->n, st->epiU.F, st->epiV.F, st->kp, st->sums2, stream);
        if (rc != BMF_OK) return rc;
        sums2 = st->sums2;
    }
    return bmf_masked_scalars(st->sums, st->epiU.partials, st->nbU, st->epiV.partials, st->nbV, sums2, counts, host_row, stream);
}
