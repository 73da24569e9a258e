// Shared host/device helpers for libbmf_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/bmf_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define BMF_EPS_F 2.220446049250313e-16f  // np.finfo(np.float64).eps, representable in fp32 (2^-52)
#define BMF_EPS_D 2.220446049250313e-16

// ---- error plumbing (host) ----
void bmf_set_error(const char* fmt, ...);
#define BMF_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            bmf_set_error(__VA_ARGS__);        \
            return BMF_ERR_BAD_ARG;            \
        }                                      \
    } while (0)
#define BMF_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            bmf_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return BMF_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)
#define BMF_LAUNCH_CHECK() BMF_HIP_CHECK(hipGetLastError())
// hipGetLastError() is sticky per thread: an unrelated runtime call made earlier by the host framework can leave an error
// behind that a post-launch check would mis-attribute to our kernel.  Clear it right before every launch.
#define BMF_LAUNCH(...)              \
    do {                             \
        (void)hipGetLastError();     \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

static inline bool bmf_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// compute units of the current device (256 on MI355X)
static inline int bmf_cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

// compute units of the CURRENT device, looked up per call through a small per-device cache (a process may drive several GPUs)
static inline int bmf_cu_count_current() {
    static int cache[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return bmf_cu_count();
    if (cache[dev] == 0) {
        hipDeviceProp_t prop;
        cache[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cache[dev];
}

// ---- panel permutation ----
// Inside one 128-block of reduction indices, local index cl = 32*wq + bit (wq = which of the 4 words of the stage, bit =
// bit inside the word).  The bits GEMM expands a 32-bit word into MFMA A-fragments with  (w << s) & 0x40004000, which
// yields, for k-step ks (0..3) of that word, fragment element j = 2*(bit & 3) + (bit >> 4) from bit = 4*ks + (j >> 1) +
// 16*(j & 1).  The panel stores the factor in the order in which B-fragments (16 contiguous bytes = 8 elements) are read:
//   BMF_SHAPE16 = 0 (v_mfma_f32_32x32x16_bf16): lane half h = wq >> 1, sub-block q = wq & 1: chunk (q*4 + ks)*2 + h
//   BMF_SHAPE16 = 1 (v_mfma_f32_16x16x32_bf16): lane group g = wq (0..3):                    chunk ks*4 + g
#ifndef BMF_SHAPE16
#define BMF_SHAPE16 1
#endif
__host__ __device__ static inline int panel_pos(int cl) {
    const int wq = cl >> 5, bit = cl & 31;
    const int rem = bit & 15;
    const int ks = rem >> 2;
    const int j = 2 * (rem & 3) + (bit >> 4);
#if BMF_SHAPE16
    return (ks * 4 + wq) * 8 + j;
#else
    const int h = wq >> 1, q = wq & 1;
    return ((q * 4 + ks) * 2 + h) * 8 + j;
#endif
}

// int8 limb panels (xf_bits_i8.hip): inside a 512-block of reduction indices, cl = 128*g + 32*t + bit: lane group g of the
// GEMM fetches words [4g, 4g+4) of the block in one 16-byte load and uses word t in stage t of the block; it expands that word
// with (w >> s) & 0x01010101, s = 4*ks + e, into dword e of k-step ks, whose byte b is bit s + 8*b.  The panel stores stage t
// as 128 contiguous bytes, B fragments being 16 contiguous bytes per (ks, g).
__host__ __device__ static inline int bmf_panel_pos_i8_dev(int cl) {
    const int g = cl >> 7, t = (cl >> 5) & 3, bit = cl & 31;
    const int b = bit >> 3, s = bit & 7;
    return 128 * t + ((s >> 2) * 4 + g) * 16 + 4 * (s & 3) + b;
}

// ---- device helpers ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Sum over the lanes of a wave, returned wave-uniform, WITHOUT the LDS: __shfl_xor is ds_bpermute_b32 (an LDS instruction, six dependent
// ones per sum) -- in a kernel that reduces once per observed cell they were the whole cost.  Four DPP steps (quad swaps, half-row and
// row mirror) leave every lane of a 16-lane row with its row's sum; the rows are then read out with v_readlane.  LANES: how many
// lanes carry data (the rest must hold 0): 16, 32 or 64.
template <int CTRL>
__device__ __forceinline__ float bmf_dpp_f32(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
template <int LANES>
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += bmf_dpp_f32<0xB1>(v);    // quad_perm [1, 0, 3, 2]
    v += bmf_dpp_f32<0x4E>(v);    // quad_perm [2, 3, 0, 1]
    v += bmf_dpp_f32<0x141>(v);   // row_half_mirror
    v += bmf_dpp_f32<0x140>(v);   // row_mirror
    float t = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    if constexpr (LANES > 16) t += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    if constexpr (LANES > 32) {
        t += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
        t += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    }
    return t;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ unsigned wave_sum(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Column scales of the int8 digit planes from the per-128-row column maxima the epilogue leaves in `blockmax` ([nblk][kp]): block
// `blk` (256 threads, `sh`: 256 floats of LDS) does columns 4 blk .. 4 blk + 3.  scale[c] = 2^e_c with max|F[:, c]| 2^e_c in
// [2^22, 0.996 * 2^23] (else [2^21, 2^22)); scale[kp + c] = 2^-e_c (the GEMM's colscale; two limbs: the lowest digit is dropped).
__device__ __forceinline__ void bmf_colscale_i8_block(const float* __restrict__ blockmax, int nblk, int kp, int limbs,
                                                      float* __restrict__ scale, int blk, float* sh) {
    const int cl = threadIdx.x & 3, sub = threadIdx.x >> 2;
    const int c = blk * 4 + cl;
    float m0 = 0.f;
    for (int b = sub; b < nblk; b += 64) m0 = fmaxf(m0, blockmax[(int64_t)b * kp + c]);
    sh[threadIdx.x] = m0;
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
            const float f = frexpf(m, &ex);  // m = f * 2^ex, f in [0.5, 1)
            // 23 bits -- unless the column maximum would land above the largest number three balanced digits can hold
            // (127 * 65793 = 8 355 711 = 0.996 * 2^23): then 22
#ifdef BMF_EXP_QMAX_FRAC   // experiment: quantise to |q| <= BMF_EXP_QMAX_FRAC * 2^BMF_EXP_QMAX_EXP (emulates a narrower digit format)
            e = min(max((f > BMF_EXP_QMAX_FRAC ? BMF_EXP_QMAX_EXP - 1 : BMF_EXP_QMAX_EXP) - ex, -100), 100);
#else
            e = min(max((f > 0.99599f ? 22 : 23) - ex, -100), 100);
#endif
        }
        scale[c] = ldexpf(1.0f, e);
        scale[kp + c] = ldexpf(1.0f, (limbs == 2 ? 8 : 0) - e);
    }
}

// The same step when the digit planes were ALREADY built, by the epilogue, with a predicted scale (epilogue.hip, mu_epilogue_i8_kernel).
// `scale` is 4 * kp floats:
//   [0, kp)        in : the scale the epilogue just used;  out: the prediction for the NEXT epilogue = the exact scale of the
//                       new maxima (a column whose maximum then grows past what three digits hold, or shrinks by more than one
//                       bit, costs a rebuild; once the factors settle that is rare, and no precision is given away to a guard bit)
//   [kp, 2 kp)     out: the GEMM's colscale for the planes as they stand (1 / the scale used)
//   [2 kp, 3 kp)   out: the exact scale of the new maxima (what the stand-alone builder uses if it has to rebuild)
//   [3 kp, 4 kp)   out: per column, 1.0 if its prediction was off (the builder then rebuilds THAT column with the exact scale, and
//                       its colscale above is already the exact one), else 0.0
// A prediction is kept while the column's maximum, scaled, is at most the largest number three balanced digits hold
// (8 355 711) and at least 2^21 (at most one bit below the exact scale's range); an all-zero column is always fine.
__device__ __forceinline__ void bmf_colscale_i8_fused_block(const float* __restrict__ blockmax, int nblk, int kp, int limbs,
                                                            float* __restrict__ scale, int blk, float* sh) {
    const int cl = threadIdx.x & 3, sub = threadIdx.x >> 2;
    const int c = blk * 4 + cl;
    float m0 = 0.f;
    for (int b = sub; b < nblk; b += 64) m0 = fmaxf(m0, blockmax[(int64_t)b * kp + c]);
    sh[threadIdx.x] = m0;
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
            const float f = frexpf(m, &ex);
#ifdef BMF_EXP_QMAX_FRAC   // (experiment, as above)
            e = min(max((f > BMF_EXP_QMAX_FRAC ? BMF_EXP_QMAX_EXP - 1 : BMF_EXP_QMAX_EXP) - ex, -100), 100);
#else
            e = min(max((f > 0.99599f ? 22 : 23) - ex, -100), 100);
#endif
        }
        const float used = scale[c];
        const float v = m * used;
#ifdef BMF_EXP_QMAX_FRAC
        const bool ok = used > 0.f && used <= 3.0e38f && (m == 0.f || (v <= BMF_EXP_QMAX_FRAC * (float)(1 << BMF_EXP_QMAX_EXP) && v >= (float)(1 << (BMF_EXP_QMAX_EXP - 2))));
#else
        const bool ok = used > 0.f && used <= 3.0e38f && (m == 0.f || (v <= 8355711.0f && v >= 2097152.0f));
#endif
        const float fresh = ldexpf(1.0f, e);
        scale[kp + c] = (limbs == 2 ? 256.0f : 1.0f) / (ok ? used : fresh);
        scale[2 * kp + c] = fresh;
        scale[3 * kp + c] = ok ? 0.0f : 1.0f;
        scale[c] = fresh;
    }
}

__device__ __forceinline__ uint16_t bf16_bits(float x) {
    __bf16 b = (__bf16)x;  // round-to-nearest-even (v_cvt_pk_bf16_f32)
    return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float(((unsigned)b) << 16); }

// Eight floats as three bf16 addends each, x = hi + mid + lo EXACTLY: every addend is the top 8 significant bits of what is left
// (truncation: 24 = 8 + 8 + 8), packed in pairs like a v_mfma_f32_32x32x16_bf16 operand (element 2 w in the low half of dword w).
// Six products a_hi b_hi + a_hi b_mid + a_mid b_hi + a_hi b_lo + a_mid b_mid + a_lo b_hi of two such triples reproduce the fp32
// product to 2^-23: what v_mfma_f32_32x32x2_f32 computes, at 6 / 16 of its matrix-pipe time (config #2, xf_f32.hip).
__device__ __forceinline__ void bmf_split3_bf16(const float (&x)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
    unsigned xb[8], mb[8], lb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        xb[j] = __float_as_uint(x[j]);
        const float r1 = x[j] - __uint_as_float(xb[j] & 0xffff0000u);
        mb[j] = __float_as_uint(r1);
        lb[j] = __float_as_uint(r1 - __uint_as_float(mb[j] & 0xffff0000u));
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {   // (x1[31:16] << 16) | x0[31:16]
        hi[w] = __builtin_amdgcn_perm(xb[2 * w + 1], xb[2 * w], 0x07060302u);
        mid[w] = __builtin_amdgcn_perm(mb[2 * w + 1], mb[2 * w], 0x07060302u);
        lo[w] = __builtin_amdgcn_perm(lb[2 * w + 1], lb[2 * w], 0x07060302u);
    }
}

// Timer hooks used by the iteration driver (api.hip)
void bmf_timer_begin(hipStream_t s);
void bmf_timer_end(hipStream_t s);
