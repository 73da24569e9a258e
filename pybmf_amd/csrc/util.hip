// Small support kernels: bit packing, popcount, stand-alone panel builder, slab reduction, k x k Gram.
#include "common.h"

#include <string.h>

// ---------------------------------------------------------------------------------------------------
// error string
// ---------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void bmf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* bmf_last_error(void) { return g_err; }
extern "C" int bmf_version(void) { return BMF_ABI_VERSION; }
extern "C" int bmf_struct_bytes(int which) {
    switch (which) {
        case 0: return (int)sizeof(bmf_epilogue_args);
        case 1: return (int)sizeof(bmf_palm_args);
        case 2: return (int)sizeof(bmf_penalty_state);
        case 3: return (int)sizeof(bmf_wnmf_real_state);
        case 4: return (int)sizeof(bmf_palm_state);
        case 5: return (int)sizeof(bmf_masked_loop);
        case 6: return (int)sizeof(bmf_masked_side);
        case 7: return (int)sizeof(bmf_link_loop);
        default: return -1;
    }
}
extern "C" int bmf_panel_pos(int cl) { return (cl < 0 || cl > 127) ? -1 : panel_pos(cl); }

namespace {

// ---------------------------------------------------------------------------------------------------
// pack: one wave packs 64 consecutive columns of one row per ballot (HBM-bound byte scan, coalesced)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_rows_u8_kernel(const uint8_t* __restrict__ X, int64_t rows, int64_t cols,
                                                            int64_t ldx, uint32_t* __restrict__ bits, int64_t ldw) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t pairs = (cols + 63) / 64;  // 64-bit groups per row that hold real columns
    const int64_t total = rows * pairs;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t g = wave; g < total; g += nwaves) {
        const int64_t row = g / pairs, pr = g - row * pairs;
        const int64_t col = pr * 64 + lane;
        const bool on = (col < cols) && (X[row * ldx + col] != 0);
        const unsigned long long m = __ballot(on);
        if (lane == 0) {
            bits[row * ldw + 2 * pr] = (uint32_t)m;
            bits[row * ldw + 2 * pr + 1] = (uint32_t)(m >> 32);
        }
    }
}

__global__ __launch_bounds__(256) void popcount_kernel(const uint32_t* __restrict__ bits, int64_t rows, int64_t words,
                                                        int64_t ldw, unsigned long long* __restrict__ count) {
    const int64_t total = rows * words;
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / words, w = i - r * words;
        c += __popc(bits[r * ldw + w]);
    }
    // wave reduce (64-bit via two 32-bit halves is unnecessary: per-thread counts are small) then one atomic per wave
    unsigned lo = (unsigned)c;  // per-thread count < 2^32 for any realistic grid
    lo = wave_sum(lo);
    if ((threadIdx.x & 63) == 0) atomicAdd(count, (unsigned long long)lo);
}

// ---------------------------------------------------------------------------------------------------
// stand-alone panel builder: F (rows_pad x ldf fp32) -> panel[t][j][pos]
// one block per 128 rows; thread handles (row, 16 columns) -- simple, only used at set-up and in tests
// ---------------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(256) void make_panel_kernel(const float* __restrict__ F, int64_t ldf, int kp,
                                                          uint16_t* __restrict__ panel, int64_t ldp) {
    const int64_t row0 = (int64_t)blockIdx.x * 128;
    for (int idx = threadIdx.x; idx < 128 * kp; idx += 256) {
        const int j = idx / 128, cl = idx - j * 128;  // consecutive threads -> consecutive rows of one column
        float f = F[(row0 + cl) * ldf + j];
        const int pos = panel_pos(cl);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const uint16_t b = bf16_bits(f);
            panel[((int64_t)t * kp + j) * ldp + row0 + pos] = b;
            f -= bf16_to_f32(b);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// fp16 panel: F[:, c] * 2^e_c = hi + lo (two fp16 addends, 22 significant bits relative to the column), e_c chosen so
// that the column maximum lands in [2^14, 2^15).  Step 1: per-128-row-block column maxima (written by the update epilogue
// as a by-product, or by blockmax_kernel for a stand-alone factor) -> colscale_kernel turns them into the two scale
// vectors: scale[c] = 2^e_c for the builder, scale[kp + c] = 0.5 / 2^e_c for the GEMM output.  No atomics: max is
// order-independent anyway, and two tiny kernels beat a ticketed tail (measured: 40 us -> 4 us at 100k rows).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void blockmax_kernel(const float* __restrict__ F, int64_t ldf, int kp,
                                                        float* __restrict__ blockmax, const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    __shared__ float sh[256];
    const int c = threadIdx.x % kp, sub = threadIdx.x / kp, nsub = 256 / kp;
    const int64_t r0 = (int64_t)blockIdx.x * 128;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
    for (int r = sub; r < 128; r += 4 * nsub) {  // 128 / nsub rows per thread, 4 independent loads in flight
        const float v0 = F[(r0 + r) * ldf + c], v1 = F[(r0 + r + nsub) * ldf + c], v2 = F[(r0 + r + 2 * nsub) * ldf + c],
                    v3 = F[(r0 + r + 3 * nsub) * ldf + c];
        m0 = fmaxf(m0, fabsf(v0)); m1 = fmaxf(m1, fabsf(v1)); m2 = fmaxf(m2, fabsf(v2)); m3 = fmaxf(m3, fabsf(v3));
    }
    sh[threadIdx.x] = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    __syncthreads();
    if (sub == 0) {
        float m = sh[c];
        for (int q = 1; q < nsub; ++q) m = fmaxf(m, sh[q * kp + c]);
        blockmax[(int64_t)blockIdx.x * kp + c] = m;
    }
}

// grid = kp / 4 blocks; block b owns columns 4b .. 4b+3; thread (c = t % 4, sub = t / 4) strides over the row blocks
__global__ __launch_bounds__(256) void colscale_kernel(const float* __restrict__ blockmax, int nblk, int kp,
                                                        float* __restrict__ scale, const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    __shared__ float sh[256];
    const int cl = threadIdx.x & 3, sub = threadIdx.x >> 2;
    const int c = blockIdx.x * 4 + cl;
    float m0 = 0.f, m1 = 0.f;
    int b = sub;
    for (; b + 64 < nblk; b += 128) {
        m0 = fmaxf(m0, blockmax[(int64_t)b * kp + c]);
        m1 = fmaxf(m1, blockmax[(int64_t)(b + 64) * kp + c]);
    }
    if (b < nblk) m0 = fmaxf(m0, blockmax[(int64_t)b * kp + c]);
    sh[threadIdx.x] = fmaxf(m0, m1);
    __syncthreads();
    for (int o = 128; o >= 4; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x < 4) {
        const float m = sh[threadIdx.x];
        int e = 0;
        if (m > 0.f && m <= 3.0e38f) {
            int ex;
            (void)frexpf(m, &ex);  // m = f * 2^ex, f in [0.5, 1)
            e = min(max(15 - ex, -110), 110);
        }
        scale[c] = ldexpf(1.0f, e);
        scale[kp + c] = ldexpf(0.5f, -e);
    }
}

// Step 2: coalesced read of 128 rows, split, transpose through LDS to the position-permuted panel rows.
template <int KP>
__global__ __launch_bounds__(256) void make_panel_f16_kernel(const float* __restrict__ F, int64_t ldf,
                                                              const float* __restrict__ scale,
                                                              uint16_t* __restrict__ panel, int64_t ldp,
                                                              const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    constexpr int LROW = 256 + 8;  // bytes per (term, column) row of the tile (+8: spreads the 2-byte scatter over banks)
    __shared__ __attribute__((aligned(16))) char tile[2 * KP * LROW];
    const int64_t row0 = (int64_t)blockIdx.x * 128;
    for (int idx = threadIdx.x; idx < 128 * KP; idx += 256) {
        const int rl = idx / KP, j = idx - rl * KP;  // consecutive threads -> consecutive columns of one row
        const float f = F[(row0 + rl) * ldf + j] * scale[j];
        const _Float16 hi = (_Float16)f;
        const _Float16 lo = (_Float16)(f - (float)hi);
        const int pos = panel_pos(rl);
        *reinterpret_cast<uint16_t*>(tile + (0 * KP + j) * LROW + 2 * pos) = __builtin_bit_cast(uint16_t, hi);
        *reinterpret_cast<uint16_t*>(tile + (1 * KP + j) * LROW + 2 * pos) = __builtin_bit_cast(uint16_t, lo);
    }
    __syncthreads();
    constexpr int pieces = 2 * KP * 32;
    for (int p = threadIdx.x; p < pieces; p += 256) {
        const int rowi = p >> 5, off = (p & 31) * 8;  // rowi = t*KP + j
        const uint2 v = *reinterpret_cast<const uint2*>(tile + rowi * LROW + off);
        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(panel + (int64_t)rowi * ldp + row0) + off) = v;
    }
}

// ---------------------------------------------------------------------------------------------------
// out[i] = sum_b slabs[b*stride + i], fp64 accumulation in slab order (deterministic)
// block = 64 outputs x 16 slab groups
// ---------------------------------------------------------------------------------------------------
// (out32 may be the first slab: every element's slabs are read before its sum is written, by the same thread or after a barrier)
__global__ __launch_bounds__(1024) void reduce_slabs_kernel(const float* slabs, int64_t stride, int count,
                                                             int64_t n, float* out32,
                                                             double* __restrict__ out64) {
    constexpr int NG = 16;  // slab groups per block: 64 outputs x 16 groups = 1024 threads
    __shared__ double sh[NG][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + o;
    double acc = 0.0;
    if (i < n) {
        // group g sums a contiguous range of slabs so that the final order is slab order
        const int per = (count + NG - 1) / NG;
        const int b0 = g * per, b1 = min(b0 + per, count);
        const float* p = slabs + (int64_t)b0 * stride + i;
        int b = b0;
        for (; b + 4 <= b1; b += 4) {  // 4 independent loads in flight
            const float v0 = p[0], v1 = p[stride], v2 = p[2 * stride], v3 = p[3 * stride];
            p += 4 * stride;
            acc = (((acc + (double)v0) + (double)v1) + (double)v2) + (double)v3;
        }
        for (; b < b1; ++b) { acc += (double)*p; p += stride; }
    }
    sh[g][o] = acc;
    __syncthreads();
    if (g == 0 && i < n) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < NG; ++q) t += sh[q][o];
        if (out32) out32[i] = (float)t;
        if (out64) out64[i] = t;
    }
}

// wide form for long vectors (the X^T U slabs): one thread per 4 consecutive outputs, float4 loads, slab order
__global__ __launch_bounds__(256) void reduce_slabs_wide_kernel(const float* slabs, int64_t stride, int count,
                                                                 int64_t n4, float* out32,
                                                                 double* __restrict__ out64) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float* p = slabs + 4 * i;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int b = 0;
        for (; b + 4 <= count; b += 4) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(p);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + stride);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + 2 * stride);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(p + 3 * stride);
            p += 4 * stride;
            a0 = (((a0 + (double)v0[0]) + (double)v1[0]) + (double)v2[0]) + (double)v3[0];
            a1 = (((a1 + (double)v0[1]) + (double)v1[1]) + (double)v2[1]) + (double)v3[1];
            a2 = (((a2 + (double)v0[2]) + (double)v1[2]) + (double)v2[2]) + (double)v3[2];
            a3 = (((a3 + (double)v0[3]) + (double)v1[3]) + (double)v2[3]) + (double)v3[3];
        }
        for (; b < count; ++b) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p);
            p += stride;
            a0 += (double)v[0]; a1 += (double)v[1]; a2 += (double)v[2]; a3 += (double)v[3];
        }
        if (out32) *reinterpret_cast<f32x4*>(out32 + 4 * i) = f32x4{(float)a0, (float)a1, (float)a2, (float)a3};
        if (out64) { out64[4 * i] = a0; out64[4 * i + 1] = a1; out64[4 * i + 2] = a2; out64[4 * i + 3] = a3; }
    }
}

// ---------------------------------------------------------------------------------------------------
// Gram partials with exact-fp32 MFMA (v_mfma_f32_32x32x2_f32):  G = F^T F.
// Lane (c = lane & 31, h = lane >> 5) of a 32x32x2 MFMA holds A[i=c][k=h] and B[k=h][j=c]; with both operands
// taken from rows r+h of F, tile (ti, tj) accumulates sum_r F[r][32ti+i] F[r][32tj+j].  Padded rows are zero.
// Each wave owns a contiguous range of row pairs; the 4 waves of a block are summed through LDS.
// ---------------------------------------------------------------------------------------------------
// Blocks past `gram_blocks` (the iteration driver adds kp / 4 of them when the factor goes to int8 digit planes) derive the column
// scales of those planes instead -- a 5-us kernel of its own otherwise, in line between the epilogue and the plane builder.
template <int NT>
__global__ __launch_bounds__(256) void gram_partial_kernel(const float* __restrict__ F, int64_t rows_pad, int64_t ldf,
                                                            float* __restrict__ slabs, int gram_blocks, const float* __restrict__ blockmax,
                                                            int nblk, int limbs, float* __restrict__ scale, const int32_t* __restrict__ stop,
                                                            int fused) {
    constexpr int KP = 32 * NT;
    __shared__ float sh[4][KP * KP];
    if ((int)blockIdx.x >= gram_blocks) {
        if (stop && *stop != 0) return;
        if (fused) bmf_colscale_i8_fused_block(blockmax, nblk, KP, limbs, scale, (int)blockIdx.x - gram_blocks, &sh[0][0]);
        else bmf_colscale_i8_block(blockmax, nblk, KP, limbs, scale, (int)blockIdx.x - gram_blocks, &sh[0][0]);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int64_t nwaves = (int64_t)gram_blocks * 4;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
    const int64_t pairs = rows_pad / 2;
    const int64_t per = (pairs + nwaves - 1) / nwaves;
    const int64_t p0 = gw * per, p1 = min(p0 + per, pairs);

    f32x16 acc[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    const float* fp = F + (2 * p0 + h) * ldf + c;
    int64_t p = p0;
    for (; p + 4 <= p1; p += 4) {  // 4 row pairs in flight
        float v[4][NT];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t) v[u][t] = fp[(int64_t)(2 * u) * ldf + 32 * t];
        fp += 8 * ldf;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = a; b < NT; ++b)   // (the tile below the diagonal: see the write-out)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u][a], v[u][b], acc[a][b], 0, 0, 0);
    }
    for (; p < p1; ++p) {
        float v[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) v[t] = fp[32 * t];
        fp += 2 * ldf;
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int b = a; b < NT; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[a], v[b], acc[a][b], 0, 0, 0);
    }
    // C/D layout: col = lane & 31, row = (i & 3) + 8*(i >> 2) + 4*h.  The tile below the diagonal is the transpose of the one above
    // it, product by product and in the same order (F[r][32 + i] F[r][j] = F[r][j] F[r][32 + i]): it is not computed -- a quarter
    // of the MFMAs at kp = 64 -- the write-out mirrors the other, and G is symmetric bit for bit.
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = a; b < NT; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h;
                sh[wave][row * KP + 32 * b + c] = acc[a][b][i];
                if (b != a) sh[wave][(32 * b + c) * KP + row] = acc[a][b][i];
            }
    __syncthreads();
    float* o = slabs + (int64_t)blockIdx.x * KP * KP;
    for (int i = threadIdx.x; i < KP * KP; i += 256) o[i] = (sh[0][i] + sh[1][i]) + (sh[2][i] + sh[3][i]);
}

}  // namespace

extern "C" int bmf_pack_rows_u8(const uint8_t* X, int64_t rows, int64_t cols, int64_t ldx, uint32_t* bits, int64_t ldw,
                                void* stream) {
    BMF_REQUIRE(X && bits, "bmf_pack_rows_u8: null pointer");
    BMF_REQUIRE(rows > 0 && cols > 0 && ldx >= cols, "bmf_pack_rows_u8: bad shape rows=%lld cols=%lld ldx=%lld",
                (long long)rows, (long long)cols, (long long)ldx);
    BMF_REQUIRE(ldw % 2 == 0 && ldw * 32 >= cols, "bmf_pack_rows_u8: ldw=%lld must be even and cover cols", (long long)ldw);
    BMF_REQUIRE(((uintptr_t)bits & 7u) == 0, "bmf_pack_rows_u8: bits must be 8-byte aligned");
    const int64_t groups = rows * ((cols + 63) / 64);
    const int64_t blocks = (groups + 3) / 4;
    const unsigned grid = (unsigned)(blocks < 8192 ? blocks : 8192);
    BMF_LAUNCH(pack_rows_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, X, rows, cols, ldx, bits, ldw);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_popcount(const uint32_t* bits, int64_t rows, int64_t words, int64_t ldw, unsigned long long* count,
                            void* stream) {
    BMF_REQUIRE(bits && count, "bmf_popcount: null pointer");
    BMF_REQUIRE(rows > 0 && words > 0 && ldw >= words, "bmf_popcount: bad shape");
    const int64_t total = rows * words;
    const int64_t blocks = (total + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 2048 ? blocks : 2048);
    BMF_LAUNCH(popcount_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, bits, rows, words, ldw, count);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_make_panel(const float* F, int64_t rows_pad, int64_t ldf, int kp, int terms, uint16_t* panel,
                              int64_t ldp, void* stream) {
    BMF_REQUIRE(F && panel, "bmf_make_panel: null pointer");
    BMF_REQUIRE(rows_pad > 0 && rows_pad % 128 == 0, "bmf_make_panel: rows_pad must be a multiple of 128");
    BMF_REQUIRE((kp == 32 || kp == 64) && ldf >= kp, "bmf_make_panel: kp must be 32 or 64 and ldf >= kp");
    BMF_REQUIRE(ldp >= rows_pad, "bmf_make_panel: ldp < rows_pad");
    BMF_REQUIRE(terms >= 1 && terms <= 3, "bmf_make_panel: terms must be 1..3");
    dim3 grid((unsigned)(rows_pad / 128)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (terms == 1) BMF_LAUNCH(make_panel_kernel<1>, grid, block, 0, s, F, ldf, kp, panel, ldp);
    if (terms == 2) BMF_LAUNCH(make_panel_kernel<2>, grid, block, 0, s, F, ldf, kp, panel, ldp);
    if (terms == 3) BMF_LAUNCH(make_panel_kernel<3>, grid, block, 0, s, F, ldf, kp, panel, ldp);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

// have_blockmax: ws already holds the per-128-row-block column maxima (the update epilogue wrote them)
int bmf_blockmax_launch(const float* F, int64_t rows_pad, int64_t ldf, int kp, float* ws, const int32_t* stop, hipStream_t s) {
    BMF_REQUIRE(F && ws && rows_pad > 0 && rows_pad % 128 == 0 && (kp == 32 || kp == 64) && ldf >= kp, "bmf_blockmax: bad arguments");
    BMF_LAUNCH(blockmax_kernel, dim3((unsigned)(rows_pad / 128)), dim3(256), 0, s, F, ldf, kp, ws, stop);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

int bmf_panel_f16_launch(const float* F, int64_t rows_pad, int64_t ldf, int kp, uint16_t* panel, int64_t ldp, float* ws,
                         float* scale, bool have_blockmax, const int32_t* stop, hipStream_t s) {
    BMF_REQUIRE(F && panel && ws && scale, "bmf_make_panel_f16: null pointer");
    BMF_REQUIRE(rows_pad > 0 && rows_pad % 128 == 0, "bmf_make_panel_f16: rows_pad must be a multiple of 128");
    BMF_REQUIRE((kp == 32 || kp == 64) && ldf >= kp, "bmf_make_panel_f16: kp must be 32 or 64 and ldf >= kp");
    BMF_REQUIRE(ldp >= rows_pad && ldp % 4 == 0, "bmf_make_panel_f16: ldp must be >= rows_pad and a multiple of 4");
    BMF_REQUIRE(((uintptr_t)panel & 7u) == 0, "bmf_make_panel_f16: panel must be 8-byte aligned");
    const int nblk = (int)(rows_pad / 128);
    dim3 grid((unsigned)nblk), block(256);
    if (!have_blockmax) BMF_LAUNCH(blockmax_kernel, grid, block, 0, s, F, ldf, kp, ws, stop);
    BMF_LAUNCH(colscale_kernel, dim3((unsigned)(kp / 4)), block, 0, s, ws, nblk, kp, scale, stop);
    if (kp == 32) BMF_LAUNCH(make_panel_f16_kernel<32>, grid, block, 0, s, F, ldf, scale, panel, ldp, stop);
    else BMF_LAUNCH(make_panel_f16_kernel<64>, grid, block, 0, s, F, ldf, scale, panel, ldp, stop);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_make_panel_f16(const float* F, int64_t rows_pad, int64_t ldf, int kp, uint16_t* panel, int64_t ldp,
                                  float* ws, float* scale, void* stream) {
    return bmf_panel_f16_launch(F, rows_pad, ldf, kp, panel, ldp, ws, scale, false, nullptr, (hipStream_t)stream);
}

extern "C" int bmf_reduce_slabs(const float* slabs, int64_t stride, int count, int64_t n, float* out32, double* out64,
                                void* stream) {
    BMF_REQUIRE(slabs && (out32 || out64), "bmf_reduce_slabs: null pointer");
    BMF_REQUIRE(count >= 1 && n >= 1 && stride >= n, "bmf_reduce_slabs: bad count/n/stride");
    if (n >= 65536 && n % 4 == 0 && stride % 4 == 0 && bmf_aligned16(slabs) && (!out32 || bmf_aligned16(out32))) {
        const int64_t n4 = n / 4, blocks = (n4 + 255) / 256;
        BMF_LAUNCH(reduce_slabs_wide_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                           (hipStream_t)stream, slabs, stride, count, n4, out32, out64);
    } else {
        dim3 grid((unsigned)((n + 63) / 64)), block(1024);
        BMF_LAUNCH(reduce_slabs_kernel, grid, block, 0, (hipStream_t)stream, slabs, stride, count, n, out32, out64);
    }
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

// blockmax != nullptr: kp / 4 extra blocks derive the int8 column scales (see the kernel)
// fused: the planes exist already, built with the predicted scale in scale[0, kp) (bmf_colscale_i8_fused_block; scale: 4 * kp floats)
int bmf_gram_partial_launch(const float* F, int64_t rows_pad, int64_t ldf, int kp, float* slabs, int blocks, const float* blockmax, int limbs,
                            float* scale, const int32_t* stop, hipStream_t s, int fused) {
    BMF_REQUIRE(F && slabs, "bmf_gram_partial: null pointer");
    BMF_REQUIRE(rows_pad > 0 && rows_pad % 2 == 0, "bmf_gram_partial: rows_pad must be even");
    BMF_REQUIRE((kp == 32 || kp == 64) && ldf >= kp, "bmf_gram_partial: kp must be 32 or 64 and ldf >= kp");
    BMF_REQUIRE(blocks >= 1 && blocks <= 1024, "bmf_gram_partial: blocks must be 1..1024");
    BMF_REQUIRE(!blockmax || (scale && rows_pad % 128 == 0 && (limbs == 2 || limbs == 3)), "bmf_gram_partial: bad column-scale arguments");
    dim3 grid((unsigned)(blocks + (blockmax ? kp / 4 : 0))), block(256);
    const int nblk = (int)(rows_pad / 128);
    if (kp == 32) BMF_LAUNCH(gram_partial_kernel<1>, grid, block, 0, s, F, rows_pad, ldf, slabs, blocks, blockmax, nblk, limbs, scale, stop, fused);
    else BMF_LAUNCH(gram_partial_kernel<2>, grid, block, 0, s, F, rows_pad, ldf, slabs, blocks, blockmax, nblk, limbs, scale, stop, fused);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_gram_partial(const float* F, int64_t rows_pad, int64_t ldf, int kp, float* slabs, int blocks,
                                void* stream) {
    return bmf_gram_partial_launch(F, rows_pad, ldf, kp, slabs, blocks, nullptr, 0, nullptr, nullptr, (hipStream_t)stream, 0);
}


// ---- 0.5 * sum(W o (A - B)^2) for dense fp64 arrays: rec_error(X_gt, X_pd, W) with an explicit prediction ---------------------
namespace {
constexpr int SQD_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sqdiff_partial_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                              const double* __restrict__ W, int64_t n, double* __restrict__ partial) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double d = A[i] - B[i];
        acc += (W ? W[i] : 1.0) * d * d;
    }
    __shared__ double sh[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
}
// out[0] += sum of the block partials, in block order (deterministic)
__global__ __launch_bounds__(256) void sqdiff_final_kernel(const double* __restrict__ partial, int blocks, double* __restrict__ out) {
    __shared__ double sh[256];
    double acc = 0.0;
    for (int b = threadIdx.x; b < blocks; b += 256) acc += partial[b];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] += sh[0];
}
}  // namespace

extern "C" int64_t bmf_sqdiff_work(void) { return SQD_BLOCKS; }

extern "C" int bmf_sqdiff_sum(const double* A, const double* B, const double* W, int64_t n, double* work, double* out, void* stream) {
    BMF_REQUIRE(A && B && work && out, "bmf_sqdiff_sum: null pointer");
    BMF_REQUIRE(n >= 0, "bmf_sqdiff_sum: n must be >= 0");
    if (n == 0) return BMF_OK;
    const int64_t want = (n + 255) / 256;
    const int blocks = (int)(want < SQD_BLOCKS ? want : SQD_BLOCKS);
    hipStream_t s = (hipStream_t)stream;
    BMF_LAUNCH(sqdiff_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, s, A, B, W, n, work);
    BMF_LAUNCH(sqdiff_final_kernel, dim3(1), dim3(256), 0, s, work, blocks, out);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}
