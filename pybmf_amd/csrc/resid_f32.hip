// Residual pass over a real-valued X held TILED (bmf_tile_f32):  sums[0] += sum |X - U V^T|,  sums[1] += sum (X - U V^T)^2
//
//   RMSE / MAE of WNMF on non-Boolean data            PyBMF/utils/metrics.py:149-160, PyBMF/models/WNMF.py:132-144
//
// Same plumbing as xf_f32_ring_kernel (xf_f32.hip): a workgroup of 4 waves owns 64 rows of X and walks the 64-column blocks of
// that row tile; wave (rw, cw) takes the 32 x 32 sub-tile rows 32 rw, columns 32 cw of every block -- one quarter (4 contiguous
// KiB) of the tiled copy -- through its own ring of four LDS buffers filled by LDS-DMA three stages ahead, counted vmcnt, no
// barrier in the loop, two workgroups per CU.  P^T = V_sub U_sub^T on the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32; lane (r, h) then holds row i = r of X and the 16 columns j = 8 g + 4 h + q, i.e. four 16-byte
// pieces of its LDS row), with the accumulator started at -X so that the MFMA result is the negated residual.  U_sub stays in
// registers for the whole walk; the 32 rows of V a stage needs (L2-resident, in fragment order: bmf_frag_rows_f32) are fetched
// two stages ahead by hand-placed loads issued before the stage's DMAs (loads retire in order: see xf_f32.hip).  The zero padding of X, U and V makes the residual
// of padded cells exactly 0, so no masks.  fp32 partial sums over the 16 cells of a lane per stage, fp64 from there on.
#include "common.h"

#ifndef BMF_F32_BARRIER
#define BMF_F32_BARRIER 0
#endif

namespace {

template <int KP>
__global__ __launch_bounds__(256, 2) void resid_f32_ring_kernel(const float* __restrict__ Xt, int stages_total, int stages_per_split,
                                                                 const float* __restrict__ U, const float* __restrict__ Vr,
                                                                 double* __restrict__ sums, int n_row_tiles,
                                                                 const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    constexpr int KH = KP / 2;                 // reduction indices per lane half
    constexpr int SF = 64, TR = 64;
    constexpr int STAGE_BYTES = TR * SF * 4;   // 16 KiB
    constexpr int RING = 4;
    constexpr int DPW = STAGE_BYTES / 1024 / 4;
    constexpr int VL = KH / 4;                 // 16-byte loads of V per lane per stage
    __shared__ __attribute__((aligned(16))) char smem[RING * STAGE_BYTES];
    __shared__ double red[4][2];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rw = wave & 1, cw = wave >> 1;
    const int r = lane & 31, h = lane >> 5;
    const int split = blockIdx.x / n_row_tiles;
    const int tile = blockIdx.x - split * n_row_tiles;
    const int s0 = split * stages_per_split;
    const int s1 = min(s0 + stages_per_split, stages_total);
    double s_abs = 0.0, s_sq = 0.0;

    if (s0 < s1) {
        // this lane's row of U: reduction indices KH h .. KH h + KH - 1 (the B operand of every MFMA of the walk)
        float u[KH];
        {
            const float* up = U + ((int64_t)tile * TR + 32 * rw + r) * KP + KH * h;
#pragma unroll
            for (int s = 0; s < KH; s += 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(up + s);
                u[s] = v[0]; u[s + 1] = v[1]; u[s + 2] = v[2]; u[s + 3] = v[3];
            }
            // consumed here, so that the compiler's own wait for these loads sits before the DMA pipeline starts (inside the loop
            // it would have to be a vmcnt(0))
#pragma unroll
            for (int s = 0; s < KH; ++s) asm volatile("" : "+v"(u[s]));
        }
        // wave-private quarter stages, as in xf_f32_ring_kernel: quarter q = 2 cw + rw of block (tile, st) is 4 contiguous KiB
        const int wq = 2 * cw + rw;
        const float* dma_src[DPW];
#pragma unroll
        for (int i = 0; i < DPW; ++i) dma_src[i] = Xt + (int64_t)tile * stages_total * (TR * SF) + wq * 1024 + i * 256 + lane * 4;
        char* const my_ring = smem + wave * (RING * 4096);
        auto issue_dma = [&](int stage) {
            const int st = min(max(stage, s0), s1 - 1);
            const int buf = (stage - s0) & (RING - 1);
#pragma unroll
            for (int i = 0; i < DPW; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dma_src[i] + (int64_t)st * (TR * SF)),
                                                 (__attribute__((address_space(3))) void*)(my_ring + buf * 4096 + i * 1024), 16, 0, 0);
        };
        // V rows of a stage in fragment order (bmf_frag_rows_f32): one contiguous KiB per load instruction.  Three fragment sets,
        // each written at one place of a loop unrolled by three, no prologue, loads issued before the DMAs of the slot: see
        // xf_f32_ring_kernel for why.
        const float* vp = Vr + cw * (VL * 256) + lane * 4;
        f32x4 vq[3][VL];
        unsigned x_off[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) x_off[g] = (unsigned)(r * 128 + (((2 * g + h) ^ ((r >> 1) & 7)) << 4));
        const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)my_ring;

        for (int sb = s0 - 3; sb < s1; sb += 3) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int s = sb + k;
                const bool live = s >= s0 && s < s1;   // wave-uniform
                if (live) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPW + VL) : "memory");
#pragma unroll
                    for (int q = 0; q < VL; ++q) asm volatile("" : "+v"(vq[k][q]));
#if BMF_F32_BARRIER
                    __builtin_amdgcn_s_barrier();
#endif
                }
                if (s < s1) {
                    const float* p = vp + (int64_t)min(max(s + 2, s0), s1 - 1) * (2 * VL * 256);
#pragma unroll
                    for (int q = 0; q < VL; ++q) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(vq[(k + 2) % 3][q]) : "v"(p + q * 256) : "memory");
                    issue_dma(s + 3);
                }
                if (live) {
                    f32x4 x[4];
                    const unsigned xbase = lds_base + (unsigned)(((s - s0) & (RING - 1)) * 4096);
#pragma unroll
                    for (int g = 0; g < 4; ++g) asm volatile("ds_read_b128 %0, %1" : "=v"(x[g]) : "v"(xbase + x_off[g]));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(x[g]));
                    f32x16 acc;
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] = -x[e >> 2][e & 3];
#pragma unroll
                    for (int kk = 0; kk < KH; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vq[k][kk >> 2][kk & 3], u[kk], acc, 0, 0, 0);
                    float pa = 0.f, ps = 0.f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        pa += fabsf(acc[e]);
                        ps = fmaf(acc[e], acc[e], ps);
                    }
                    s_abs += (double)pa;
                    s_sq += (double)ps;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus DMAs / V loads of the last stages
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int q = 0; q < VL; ++q) asm volatile("" : "+v"(vq[k][q]));
    }
    s_abs = wave_sum(s_abs);
    s_sq = wave_sum(s_sq);
    if (lane == 0) { red[wave][0] = s_abs; red[wave][1] = s_sq; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const double t = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
        if (t != 0.0) atomicAdd(&sums[threadIdx.x], t);
    }
}

// V in the order the residual kernel's lanes consume it: stage st (64 rows of V), column half cw, piece q, lane (r, h) ->
//   frag[(((st * 2 + cw) * (kp / 8) + q) * 64 + 32 h + r) * 4 + t] = V[(64 st + 32 cw + r) * kp + (kp / 2) h + 4 q + t]
__global__ __launch_bounds__(256) void frag_rows_f32_kernel(const float* __restrict__ V, int kp, int64_t pieces, float* __restrict__ frag,
                                                             const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;
    const int VL = kp / 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < pieces; i += (int64_t)gridDim.x * 256) {
        const int lane = (int)(i & 63), r = lane & 31, h = lane >> 5;
        const int64_t g = i >> 6;
        const int q = (int)(g % VL);
        const int64_t g2 = g / VL;
        const int cw = (int)(g2 & 1);
        const int64_t st = g2 >> 1;
        *reinterpret_cast<f32x4*>(frag + i * 4) = *reinterpret_cast<const f32x4*>(V + (64 * st + 32 * cw + r) * kp + (kp / 2) * h + 4 * q);
    }
}

}  // namespace

int bmf_frag_rows_f32_launch(const float* V, int64_t rows_pad, int kp, float* frag, const int32_t* stop, hipStream_t s) {
    BMF_REQUIRE(V && frag, "bmf_frag_rows_f32: null pointer");
    BMF_REQUIRE(rows_pad > 0 && rows_pad % 64 == 0 && (kp == 32 || kp == 64), "bmf_frag_rows_f32: rows_pad must be a positive multiple of 64, kp 32 or 64");
    BMF_REQUIRE(bmf_aligned16(V) && bmf_aligned16(frag), "bmf_frag_rows_f32: pointers must be 16-byte aligned");
    const int64_t pieces = rows_pad * kp / 4;
    const int64_t blocks = (pieces + 255) / 256;
    BMF_LAUNCH(frag_rows_f32_kernel, dim3((unsigned)(blocks < 65535 ? blocks : 65535)), dim3(256), 0, s, V, kp, pieces, frag, stop);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_frag_rows_f32(const float* V, int64_t rows_pad, int kp, float* frag, void* stream) {
    return bmf_frag_rows_f32_launch(V, rows_pad, kp, frag, nullptr, (hipStream_t)stream);
}

/* Xtiled: bmf_tile_f32 of the zero-padded m_pad x n_pad X (both multiples of 64); Vfrag: bmf_frag_rows_f32 of V (n_pad x kp).
 * sums[0..1] are ADDED to (zeroed by the caller). */
int bmf_residual_tiled_launch(const float* Xtiled, int64_t m_pad, int64_t n_pad, const float* U, const float* Vfrag, int kp, double* sums,
                              const int32_t* stop, hipStream_t s) {
    const float* V = Vfrag;
    BMF_REQUIRE(Xtiled && U && V && sums, "bmf_residual_sums_f32_tiled: null pointer");
    BMF_REQUIRE(m_pad > 0 && m_pad % 64 == 0 && n_pad > 0 && n_pad % 64 == 0, "bmf_residual_sums_f32_tiled: m_pad and n_pad must be positive multiples of 64");
    BMF_REQUIRE(kp == 32 || kp == 64, "bmf_residual_sums_f32_tiled: kp must be 32 or 64");
    BMF_REQUIRE(bmf_aligned16(Xtiled) && bmf_aligned16(U) && bmf_aligned16(V), "bmf_residual_sums_f32_tiled: pointers must be 16-byte aligned");
    const int tiles = (int)(m_pad / 64), stages = (int)(n_pad / 64);
    int splits = (1024 + tiles - 1) / tiles;                 // ~1024 workgroups, at least ~8 stages each
    if (splits > stages / 8) splits = stages / 8 > 0 ? stages / 8 : 1;
    const int sps = (stages + splits - 1) / splits;
    splits = (stages + sps - 1) / sps;
    dim3 grid((unsigned)(tiles * splits)), block(256);
    if (kp == 32) BMF_LAUNCH(resid_f32_ring_kernel<32>, grid, block, 0, s, Xtiled, stages, sps, U, V, sums, tiles, stop);
    else BMF_LAUNCH(resid_f32_ring_kernel<64>, grid, block, 0, s, Xtiled, stages, sps, U, V, sums, tiles, stop);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

extern "C" int bmf_residual_sums_f32_tiled(const float* Xtiled, int64_t m_pad, int64_t n_pad, const float* U, const float* Vfrag, int kp,
                                           double* sums, void* stream) {
    return bmf_residual_tiled_launch(Xtiled, m_pad, n_pad, U, Vfrag, kp, sums, nullptr, (hipStream_t)stream);
}
