// The thresholding objective of BinaryMFThreshold for the all-ones mask (W = 'full') in TRACE form, for a BATCH of (u, v) pairs
// (PyBMF/models/BinaryMFThreshold.py:150-207; the candidates of one Wolfe search, PyBMF/solvers/line_search.py:30-62).
//
//   F(u, v)  = 1/2 || X - Us Vs^T ||_F^2 = 1/2 ( sum X  -  2 sum_{X_ij = 1} <Us_i, Vs_j>  +  < Us^T Us, Vs^T Vs > )
//   dF_u     = sum_{X_ij = 1} <dUs_i, Vs_j>  -  < Vs^T Vs, Us^T dUs >         (the reference's sign convention, :174-207)
//   dF_v     = sum_{X_ij = 1} <Us_i, dVs_j>  -  < Us^T Us, Vs^T dVs >
//   Us = sigmoid(lamda (U - u)), dUs = lamda Us (1 - Us), likewise V.
//
// X is Boolean, so every X-dependent term is a sum over its ONES of a dot product of two k-vectors, and everything else is a
// k x k Gram matrix: O(nnz k + (m + n) k^2) per pair instead of the m n k of the tile product (thresh64.hip) -- at MovieLens-1M
// shape 40 x less arithmetic -- and, above all, MANY pairs per launch: one wave instruction serves 64 / kc pairs (kc = 16, 32 or
// 64 lanes per pair), the list of ones is walked once for all of them.  A Wolfe search of the reference evaluates F at
// alpha = 2, 1, 1/2, ... until the Armijo test holds (~25 halvings at MovieLens-1M shape), one dependent launch-and-wait each; its
// candidate sequence is known beforehand, so the host asks for the whole chain in ONE call (solvers/line_search.py here).
// fp64 throughout, block partials added in a fixed order: the same decisions as the reference (golden g4 / g8 / g12; the trace
// form agrees with the literal one to 3e-15 relative in fp64).
#include "common.h"

#include <cstdlib>

#define BMF_TRACE_MAX_GROUPS 8
#define BMF_TRACE_MAX_SEGMENTS(m) ((int64_t)8 * (m) + 64)   // segment slots the workspace holds (a matrix with up to ~900 ones per row on average)

namespace {

struct TracePairs {
    double u[32], v[32];
};

__device__ __forceinline__ void sigmoid_pair_t(double z, double lam, double& s, double& d) {   // (as thresh64.hip)
    if (z >= 0) s = 1.0 / (1.0 + exp(-z));
    else { const double e = exp(z); s = e / (1.0 + e); }
    d = lam * s * (1.0 - s);
}

// S[p][row][c], c < kc: sigmoid-transformed factor of pair p = blockIdx.y (zero for padding rows / columns; one extra all-zero row
// `rows_alloc - 1` that the cell pass points missing cells at).  Blocks [0, gu) of a pair do U, the rest V.  kc = 1 << kshift.
__global__ __launch_bounds__(256) void trace_transform_kernel(const double* __restrict__ U, int64_t ldu, int m, int64_t mrows, const double* __restrict__ V,
                                                               int n, int64_t nrows, int k, int kshift, double lam, TracePairs pr,
                                                               double* __restrict__ Us, double* __restrict__ dUs, double* __restrict__ Vs,
                                                               double* __restrict__ dVs, int gu) {
    const bool is_u = (int)blockIdx.x < gu;
    const int p = (int)blockIdx.y;
    const double* F = is_u ? U : V;
    const int rows = is_u ? m : n;
    const int per_pair = (int)((is_u ? mrows : nrows) << kshift);   // < 2^31: checked by the launcher
    double* S = (is_u ? Us : Vs) + (int64_t)p * per_pair;
    double* D = (is_u ? dUs : dVs);
    if (D) D += (int64_t)p * per_pair;
    const double x = is_u ? pr.u[p] : pr.v[p];
    const int b = is_u ? (int)blockIdx.x : (int)blockIdx.x - gu, nb = is_u ? gu : (int)gridDim.x - gu;
    for (int i = b * 256 + (int)threadIdx.x; i < per_pair; i += nb * 256) {
        const int r = i >> kshift, c = i & ((1 << kshift) - 1);
        double s_ = 0.0, d_ = 0.0;
        if (r < rows && c < k) sigmoid_pair_t((F[(int64_t)r * ldu + c] - x) * lam, lam, s_, d_);
        S[i] = s_;
        if (D) D[i] = d_;
    }
}

// The ONES of X, cut into SEGMENTS of at most 128 cells of one row (seg_row, seg_beg, seg_end: built once per fit, longest first).
// A wave takes one segment: lane (pl, c), pl = lane / KC, serves column c of pairs g * PPI + pl, g < G <= GMAX -- per cell (i, j) and
// group ONE load of Vs (one of dVs) and one (three) FMAs.  The wave fetches 64 column indices with one load and walks them T at a
// time (v_readlane: the indices are wave-uniform; missing cells point at the zero row), so T * G independent loads are in flight per
// lane.  cellpart[p][kind][segment] = sum over the segment's ones of <Us_i, Vs_j>, <dUs_i, Vs_j>, <Us_i, dVs_j>.
template <int KC, bool GRAD, int GMAX>
__global__ __launch_bounds__(256) void trace_cells_kernel(const int32_t* __restrict__ seg_row, const int64_t* __restrict__ seg_beg,
                                                           const int32_t* __restrict__ seg_len, const int32_t* __restrict__ idx, int nseg,
                                                           int64_t mrows, int64_t nrows, int G, const double* __restrict__ Us,
                                                           const double* __restrict__ dUs, const double* __restrict__ Vs, const double* __restrict__ dVs,
                                                           double* __restrict__ cellpart) {
    constexpr int PPI = 64 / KC;
    constexpr int T = (GRAD ? 16 : 32) / GMAX;   // cells per trip
    constexpr int NV = GRAD ? 3 : 1;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sg = (int)blockIdx.x * 4 + wave;
    if (sg >= nseg) return;
    const int i = seg_row[sg];
    const int64_t w0 = seg_beg[sg];
    const int len = seg_len[sg];   // <= 128
    const int pl = lane / KC, c = lane % KC;
    const int64_t upair = mrows * KC, vpair = nrows * KC;
    double us[GMAX], dus[GMAX], a0[GMAX], a1[GMAX], a2[GMAX];
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        const int64_t p = g * PPI + pl;
        us[g] = g < G ? Us[p * upair + (int64_t)i * KC + c] : 0.0;
        dus[g] = (GRAD && g < G) ? dUs[p * upair + (int64_t)i * KC + c] : 0.0;
        a0[g] = a1[g] = a2[g] = 0.0;
    }
    const int zrow = (int)(nrows - 1);   // all-zero row
    const double* vlane = Vs + (int64_t)pl * vpair + c;
    const double* dlane = dVs + (int64_t)pl * vpair + c;
    const int myj0 = lane < len ? idx[w0 + lane] : zrow;
    const int myj1 = 64 + lane < len ? idx[w0 + 64 + lane] : zrow;
    for (int st = 0; st < len; st += 64) {
        const int myj = st == 0 ? myj0 : myj1;
        const int nst = min(64, len - st);
        for (int t0 = 0; t0 < nst; t0 += T) {
            double vv[T][GMAX], dv[T][GMAX];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int64_t jo = (int64_t)__builtin_amdgcn_readlane(myj, t0 + t) * KC;   // (t0 + t < 64: T divides 64)
#pragma unroll
                for (int g = 0; g < GMAX; ++g)
                    if (g < G) {
                        vv[t][g] = vlane[(int64_t)(g * PPI) * vpair + jo];
                        if (GRAD) dv[t][g] = dlane[(int64_t)(g * PPI) * vpair + jo];
                    }
            }
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int g = 0; g < GMAX; ++g)
                    if (g < G) {
                        a0[g] = fma(us[g], vv[t][g], a0[g]);
                        if (GRAD) {
                            a1[g] = fma(dus[g], vv[t][g], a1[g]);
                            a2[g] = fma(us[g], dv[t][g], a2[g]);
                        }
                    }
        }
    }
    // the KC columns of a pair are summed in a fixed order (butterfly inside the pair's lanes), lane c == 0 of a pair writes
#pragma unroll
    for (int g = 0; g < GMAX; ++g)
        if (g < G) {
#pragma unroll
            for (int o = KC / 2; o > 0; o >>= 1) {
                a0[g] += __shfl_xor(a0[g], o, 64);
                if (GRAD) { a1[g] += __shfl_xor(a1[g], o, 64); a2[g] += __shfl_xor(a2[g], o, 64); }
            }
            if (c == 0) {
                const int64_t p = g * PPI + pl;
                cellpart[(p * 3 + 0) * nseg + sg] = a0[g];
                if (GRAD) { cellpart[(p * 3 + 1) * nseg + sg] = a1[g]; cellpart[(p * 3 + 2) * nseg + sg] = a2[g]; }
            }
        }
    (void)NV;
}

template <int KC, bool GRAD>
__global__ __launch_bounds__(256) void trace_gram_kernel(int64_t mrows, int64_t nrows, int npairs_alloc, const double* __restrict__ Us,
                                                          const double* __restrict__ dUs, const double* __restrict__ Vs, const double* __restrict__ dVs,
                                                          double* __restrict__ grampart, int uchunks, int vchunks) {
    constexpr int CR = 4096 / KC;           // rows per chunk
    __shared__ double sh[2 * CR * KC];      // the S and D chunks (64 KiB)
    const int gb = (int)blockIdx.x;
    const int per_pair_blocks = (GRAD ? 2 : 1) * (uchunks + vchunks);
    const int p = gb / per_pair_blocks;
    int r = gb % per_pair_blocks;
    const bool second = r >= uchunks + vchunks;   // the H matrices (S^T D)
    if (second) r -= uchunks + vchunks;
    const bool is_u = r < uchunks;
    const int chunk = is_u ? r : r - uchunks;
    const int64_t ralloc = is_u ? mrows : nrows;
    const double* S = (is_u ? Us : Vs) + (int64_t)p * ralloc * KC;
    const double* D = second ? ((is_u ? dUs : dVs) + (int64_t)p * ralloc * KC) : S;
    const int64_t r0 = (int64_t)chunk * CR;
    const int nr = (int)min((int64_t)CR, ralloc - r0);
    double* sS = sh;
    double* sD = sh + CR * KC;
    for (int e = threadIdx.x; e < CR * KC; e += 256) {
        const bool in = e < nr * KC;
        sS[e] = in ? S[r0 * KC + e] : 0.0;
        sD[e] = in ? D[r0 * KC + e] : 0.0;
    }
    __syncthreads();
    constexpr int EPT = KC * KC / 256;   // elements per thread (KC = 16: one)
    double acc[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) acc[e] = 0.0;
#pragma unroll 8
    for (int row = 0; row < CR; ++row) {   // (unrolled: the LDS reads of eight rows in flight; one accumulator, rows in order)
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int el = threadIdx.x + 256 * e;
            acc[e] = fma(sS[row * KC + el / KC], sD[row * KC + el % KC], acc[e]);
        }
    }
    // grampart[q][p][chunk][KC * KC], q = 0: Us^T Us, 1: Vs^T Vs, 2: Us^T dUs, 3: Vs^T dVs; chunk slots: max(uchunks, vchunks)
    const int qm = (second ? 2 : 0) + (is_u ? 0 : 1);
    const int cmax = uchunks > vchunks ? uchunks : vchunks;
    double* out = grampart + (((int64_t)qm * npairs_alloc + p) * cmax + chunk) * (KC * KC);
#pragma unroll
    for (int e = 0; e < EPT; ++e) out[threadIdx.x + 256 * e] = acc[e];
}

// One block of 1024 threads per pair: the ordered sums, F and dF.  out[4 p + 0..3] = (seq, 2 F, dF_u, dF_v) -- words 1..3 the slots of
// bmf_thresh_eval64, word 0 the call's sequence number written behind them -- in pinned host memory; the LAST block to finish (a device
// counter) writes the sequence word out[4 npairs] behind a system-scope fence: the host waits for that word AND for every pair's stamp.  Every sum has a fixed shape: thread-strided partial sums, then a
// binary tree over the threads.
template <int KC>
__global__ __launch_bounds__(1024) void trace_final_kernel(const double* __restrict__ cellpart, const double* __restrict__ grampart, int nseg, int npairs,
                                                            int npairs_alloc, int uchunks, int vchunks, double sum_x, int want_grad,
                                                            double* __restrict__ out, double seq, unsigned* __restrict__ counter) {
    __shared__ double sh[3][1024];
    __shared__ double gq[4][4][KC * KC > 256 ? 1 : 256];   // KC = 16: the four quarter-sums of every element of the four matrices
    const int p = (int)blockIdx.x, t = threadIdx.x;
    auto tree = [&]() {
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if (t < o) { sh[0][t] += sh[0][t + o]; sh[1][t] += sh[1][t + o]; sh[2][t] += sh[2][t + o]; }
            __syncthreads();
        }
    };
    // cell sums over the segments
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    const double* cp = cellpart + (int64_t)p * 3 * nseg;   // [p][kind][segment]
    for (int i = t; i < nseg; i += 1024) {
        s0 += cp[i];
        if (want_grad) { s1 += cp[(int64_t)nseg + i]; s2 += cp[2 * (int64_t)nseg + i]; }
    }
    sh[0][t] = s0; sh[1][t] = s1; sh[2][t] = s2;
    tree();
    const double a = sh[0][0], du1 = sh[1][0], dv1 = sh[2][0];
    __syncthreads();
    // < GU, GV >, < GV, HU >, < GU, HV >: an element's chunks are summed in four quarters (by four threads), the quarters in order
    const int cmax = uchunks > vchunks ? uchunks : vchunks;
    const int nmat = want_grad ? 4 : 2;
    double b = 0.0, bu = 0.0, bv = 0.0;
    if constexpr (KC * KC <= 256) {
        const int el = t & 255, qt = t >> 8;   // qt: quarter of the chunks
        if (el < KC * KC) {
            for (int q = 0; q < nmat; ++q) {
                const int nch = (q & 1) ? vchunks : uchunks;
                const int c0 = (nch * qt) / 4, c1 = (nch * (qt + 1)) / 4;
                const double* g = grampart + ((int64_t)q * npairs_alloc + p) * cmax * (KC * KC) + el;
                double s = 0.0;
                for (int ch = c0; ch < c1; ++ch) s += g[(int64_t)ch * (KC * KC)];
                gq[q][qt][el] = s;
            }
        }
        __syncthreads();
        if (t < KC * KC) {
            const double gu = ((gq[0][0][t] + gq[0][1][t]) + gq[0][2][t]) + gq[0][3][t];
            const double gv = ((gq[1][0][t] + gq[1][1][t]) + gq[1][2][t]) + gq[1][3][t];
            b = gu * gv;
            if (want_grad) {
                bu = gv * (((gq[2][0][t] + gq[2][1][t]) + gq[2][2][t]) + gq[2][3][t]);
                bv = gu * (((gq[3][0][t] + gq[3][1][t]) + gq[3][2][t]) + gq[3][3][t]);
            }
        }
    } else {
        auto gel = [&](int q, int el) {
            const int nch = (q & 1) ? vchunks : uchunks;
            const double* g = grampart + ((int64_t)q * npairs_alloc + p) * cmax * (KC * KC) + el;
            double s = 0.0;
            for (int ch = 0; ch < nch; ++ch) s += g[(int64_t)ch * (KC * KC)];
            return s;
        };
        for (int el = t; el < KC * KC; el += 1024) {
            const double gu = gel(0, el), gv = gel(1, el);
            b += gu * gv;
            if (want_grad) {
                bu += gv * gel(2, el);
                bv += gu * gel(3, el);
            }
        }
    }
    sh[0][t] = b; sh[1][t] = bu; sh[2][t] = bv;
    tree();
    if (t == 0) {
        out[4 * p + 1] = sum_x - 2.0 * a + sh[0][0];
        out[4 * p + 2] = want_grad ? du1 - sh[1][0] : 0.0;
        out[4 * p + 3] = want_grad ? dv1 - sh[2][0] : 0.0;
        __threadfence_system();
        // word 0 of the pair = the call's sequence number, behind the pair's three results: a host that polls checks EVERY pair's stamp,
        // not only the word after the last pair -- writes to host memory that leave from different blocks (and land in different cache
        // lines) need not become visible in the order of the device-side fences (seen once in ~10^5 calls as a result that was the
        // PREVIOUS call's value at the same slot: a nearby trial point, so the search took a slightly different path)
        out[4 * p + 0] = seq;
        __threadfence_system();
        const unsigned done = atomicAdd(counter, 1u);
        if (done == (unsigned)npairs - 1) {
            *counter = 0u;   // ready for the next call on this stream
            __threadfence_system();
            out[4 * npairs] = seq;
        }
    }
}

int kc_of(int k) { return k <= 16 ? 16 : (k <= 32 ? 32 : 64); }

}  // namespace

extern "C" int bmf_thresh_trace64_max_pairs(int k) {
    if (k < 1 || k > 64) return BMF_ERR_BAD_ARG;
    const int n = BMF_TRACE_MAX_GROUPS * (64 / kc_of(k));
    return n < 32 ? n : 32;
}

// doubles of workspace for up to max_pairs pairs; the first word is the completion counter of the final kernel
extern "C" int64_t bmf_thresh_trace64_work(int32_t m, int32_t n, int k, int max_pairs) {
    if (m < 1 || n < 1 || k < 1 || k > 64 || max_pairs < 1 || max_pairs > bmf_thresh_trace64_max_pairs(k)) return BMF_ERR_BAD_ARG;
    const int kc = kc_of(k), ppi = 64 / kc;
    const int64_t pa = (int64_t)((max_pairs + ppi - 1) / ppi) * ppi;
    const int64_t mrows = m + 1, nrows = n + 1;
    const int64_t cr = 4096 / kc;
    const int64_t uch = (mrows + cr - 1) / cr, vch = (nrows + cr - 1) / cr, cmax = uch > vch ? uch : vch;
    return 2 + 2 * pa * (mrows + nrows) * kc + BMF_TRACE_MAX_SEGMENTS(m) * pa * 3 + 4 * pa * cmax * kc * kc;
}

// F and dF at n_pairs points (u, v) in one enqueue.  The ONES of X as a list of column indices `idx` (int32, device, row after
// row) cut into nseg segments of at most 128 cells of ONE row each: seg_row[s] (int32), seg_beg[s] (int64 offset into idx),
// seg_len[s] (int32, 1..128) -- best in descending order of length (the long segments start first); nseg <= 8 m + 64; U64, V64: the fp64 factors with leading dimension ldf; uv_host: 2 n_pairs doubles (u0, v0, u1, v1, ...), read
// before the call returns; sum_x = the number of ones; work: bmf_thresh_trace64_work doubles, zero-filled once by the caller;
// out_host: 4 n_pairs + 1 doubles of pinned host memory -- out[4 p + 1] = 2 F, out[4 p + 2..3] = dF (want_grad), out[4 p] = `seq`
// written behind them, and the last word = `seq` once everything before it has left the device.  A polling host waits for all
// n_pairs + 1 stamps (or synchronises the stream).
extern "C" int bmf_thresh_trace64(const int32_t* seg_row, const int64_t* seg_beg, const int32_t* seg_len, int32_t nseg, const int32_t* idx, int32_t m, int32_t n, const double* U64, const double* V64, int64_t ldf,
                                  int k, const double* uv_host, int32_t n_pairs, double lamda, double sum_x, int want_grad, double* work,
                                  double* out_host, double seq, void* stream) {
    BMF_REQUIRE(seg_row && seg_beg && seg_len && idx && U64 && V64 && uv_host && work && out_host, "bmf_thresh_trace64: null pointer");
    BMF_REQUIRE(nseg >= 1 && nseg <= BMF_TRACE_MAX_SEGMENTS(m), "bmf_thresh_trace64: nseg=%d outside 1..%lld", nseg, (long long)BMF_TRACE_MAX_SEGMENTS(m));
    BMF_REQUIRE(m >= 1 && n >= 1 && k >= 1 && k <= 64 && ldf >= k, "bmf_thresh_trace64: bad shape");
    BMF_REQUIRE(n_pairs >= 1 && n_pairs <= bmf_thresh_trace64_max_pairs(k), "bmf_thresh_trace64: n_pairs=%d outside 1..%d for k=%d", n_pairs,
                bmf_thresh_trace64_max_pairs(k), k);
    hipStream_t s = (hipStream_t)stream;
    const int kc = kc_of(k), ppi = 64 / kc;
    const int G = (n_pairs + ppi - 1) / ppi, pa = G * ppi;
    const int64_t mrows = (int64_t)m + 1, nrows = (int64_t)n + 1;
    const int cr = 4096 / kc;
    const int uch = (int)((mrows + cr - 1) / cr), vch = (int)((nrows + cr - 1) / cr), cmax = uch > vch ? uch : vch;
    TracePairs pr;
    for (int p = 0; p < 32; ++p) {   // (padding pairs repeat the last real one: their results are never read)
        const int q = p < n_pairs ? p : n_pairs - 1;
        pr.u[p] = uv_host[2 * q];
        pr.v[p] = uv_host[2 * q + 1];
    }
    unsigned* counter = reinterpret_cast<unsigned*>(work);
    double* Us = work + 2;
    double* dUs = Us + (int64_t)pa * mrows * kc;
    double* Vs = dUs + (int64_t)pa * mrows * kc;
    double* dVs = Vs + (int64_t)pa * nrows * kc;
    double* cellpart = dVs + (int64_t)pa * nrows * kc;
    double* grampart = cellpart + BMF_TRACE_MAX_SEGMENTS(m) * pa * 3;
    BMF_REQUIRE(mrows * kc < (1ll << 31) && nrows * kc < (1ll << 31), "bmf_thresh_trace64: factor too large");
    const int kshift = kc == 16 ? 4 : (kc == 32 ? 5 : 6);
    int gu = (int)((mrows * kc + 255) / 256), gv = (int)((nrows * kc + 255) / 256);
    if (gu > 1024) gu = 1024;
    if (gv > 1024) gv = 1024;
    BMF_LAUNCH(trace_transform_kernel, dim3((unsigned)(gu + gv), (unsigned)pa), dim3(256), 0, s, U64, ldf, m, mrows, V64, n, nrows, k, kshift, lamda, pr, Us,
               want_grad ? dUs : nullptr, Vs, want_grad ? dVs : nullptr, gu);
    const int gram_blocks = pa * (want_grad ? 2 : 1) * (uch + vch);
    // The cell pass gathers 8 kc bytes of Vs per cell and pair; the transformed V of ALL pairs (11 MB for 24 pairs at MovieLens-1M
    // shape) does not fit an XCD's 4-MiB L2, so the pass runs in slices of GP groups (8 pairs at kc = 16: 3.8 MB) whose rows of Vs
    // stay L2-resident (1, 2, 4 or 8 groups per pass were measured at that shape, 24 pairs: 8 -> 2 took the pass from 350 to
    // the time on record in DESIGN section 8).
    constexpr int gp_env = 2;
#define BMF_TRACE_CELLS(KC_, GR_, GM_)                                                                                                        \
    BMF_LAUNCH((trace_cells_kernel<KC_, GR_, GM_>), dim3((unsigned)((nseg + 3) / 4)), dim3(256), 0, s, seg_row, seg_beg, seg_len, idx, nseg, mrows, nrows, gq, \
               Us + (int64_t)g0 * ppi * mrows * KC_, dUs + (int64_t)g0 * ppi * mrows * KC_, Vs + (int64_t)g0 * ppi * nrows * KC_,                         \
               dVs + (int64_t)g0 * ppi * nrows * KC_, cellpart + (int64_t)g0 * ppi * 3 * nseg)
#define BMF_TRACE_MAIN(KC_)                                                                                                                   \
    if (kc == KC_) {                                                                                                                          \
        if (want_grad) BMF_LAUNCH((trace_gram_kernel<KC_, true>), dim3((unsigned)gram_blocks), dim3(256), 0, s, mrows, nrows, pa, Us, dUs, Vs, dVs, grampart, uch, vch); \
        else BMF_LAUNCH((trace_gram_kernel<KC_, false>), dim3((unsigned)gram_blocks), dim3(256), 0, s, mrows, nrows, pa, Us, dUs, Vs, dVs, grampart, uch, vch); \
        for (int g0 = 0; g0 < G; g0 += gp_env) {                                                                                              \
            const int gq = G - g0 < gp_env ? G - g0 : gp_env;                                                                                 \
            const int gmax = gq <= 1 ? 1 : (gq <= 2 ? 2 : (gq <= 4 ? 4 : 8));                                                                 \
            if (want_grad) {                                                                                                                  \
                if (gmax == 1) BMF_TRACE_CELLS(KC_, true, 1); else if (gmax == 2) BMF_TRACE_CELLS(KC_, true, 2);                              \
                else if (gmax == 4) BMF_TRACE_CELLS(KC_, true, 4); else BMF_TRACE_CELLS(KC_, true, 8);                                        \
            } else {                                                                                                                          \
                if (gmax == 1) BMF_TRACE_CELLS(KC_, false, 1); else if (gmax == 2) BMF_TRACE_CELLS(KC_, false, 2);                            \
                else if (gmax == 4) BMF_TRACE_CELLS(KC_, false, 4); else BMF_TRACE_CELLS(KC_, false, 8);                                      \
            }                                                                                                                                 \
        }                                                                                                                                     \
        BMF_LAUNCH((trace_final_kernel<KC_>), dim3((unsigned)n_pairs), dim3(1024), 0, s, cellpart, grampart, nseg, n_pairs, pa, uch, vch, sum_x, want_grad, \
                   out_host, seq, counter);                                                                                                   \
    }
    BMF_TRACE_MAIN(16) BMF_TRACE_MAIN(32) BMF_TRACE_MAIN(64)
#undef BMF_TRACE_CELLS
#undef BMF_TRACE_MAIN
    (void)cmax;
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}
