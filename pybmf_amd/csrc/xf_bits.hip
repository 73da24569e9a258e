// K1/K2: out = bits(A) . F  -- the two big contractions of the multiplicative update on a Boolean X.
//
//   X  @ V   (A = X bits,   panel of V)   replaces  multiply(W, X) @ V      PyBMF/models/BinaryMFPenalty.py:139
//   X^T @ U  (A = X^T bits, panel of U)   replaces  multiply(W, X).T @ U    PyBMF/models/BinaryMFPenalty.py:154
//
// Design (gfx950):
//   * A is a bit matrix (1 bit per cell: 250 MB for 100k x 20k), so the kernel is MFMA-bound, not HBM-bound.
//     Each lane loads the bits of "its" row straight into VGPRs (8 B per 128 reduction indices) and expands
//     them to bf16 {0, 2.0} operands with one shift + one AND per dword:  (w << s) & 0x40004000 puts bit b in
//     bit 14 of the low half and bit b+16 in bit 14 of the high half; 0x4000 is bf16 2.0, and the final result
//     is scaled by 0.5 (exact).
//   * The factor is fed as a bf16 panel split into T addends (F = t0 + t1 + t2; T = 3 reproduces fp32 exactly),
//     all T products accumulate into the same fp32 MFMA accumulator.  The panel is stored position-permuted
//     (common.h: panel_pos) so that each lane's B fragment is 16 contiguous bytes.
//   * The panel stage (T x kp x 128 bf16) is brought into LDS by LDS-DMA (global_load_lds, 16 B/lane), double
//     buffered, one barrier per stage; the XOR swizzle that makes the ds_read_b128 fragment reads conflict-free
//     is applied on the DMA *source* address (the LDS image must stay lane-linear).
//   * v_mfma_f32_32x32x16_bf16, 64-wide waves: each wave owns 64 rows x kp columns (2 x NT accumulator tiles);
//     a workgroup of WAVES waves shares one panel stage.  The reduction is split over `splits` workgroups per
//     row tile (slabs, summed in fixed order by the consumer: deterministic, no atomics).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// one MFMA on 16-byte A / B fragments of either 16-bit format (0x4000 reads as 2.0 in both bf16 and fp16)
template <bool F16>
__device__ __forceinline__ f32x4 mfma_16x16x32(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ f32x16 mfma_32x32x16(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// F16 = the panel holds fp16 addends of the column-scaled factor (bmf_make_panel_f16); colscale[c] = 0.5 / 2^e_c undoes the
// scale (and the 2.0 of the expanded bits) on the way out.  colscale == nullptr: plain 0.5 (bf16 panels).
template <int NT, int T, int WAVES, bool F16>
__global__ __launch_bounds__(WAVES * 64) void xf_bits_kernel(const uint32_t* __restrict__ A, int64_t ldw, int stages,
                                                              const uint16_t* __restrict__ P, int64_t ldp,
                                                              float* __restrict__ out, int64_t slab_stride,
                                                              int units_per_wg, int64_t total_units, int slots,
                                                              const float* __restrict__ colscale,
                                                              const int32_t* __restrict__ stop) {
    if (stop && *stop != 0) return;  // early stop tripped: the state is frozen, skip the work (wave-uniform)
    constexpr int NC = NT * 32;               // panel columns
    constexpr int LROWS = T * NC;             // 256-byte LDS rows per stage
    constexpr int STAGE_BYTES = LROWS * 256;  // T*NC*128 bf16
    constexpr int DMA_PER_WAVE = LROWS / 4 / WAVES;
    static_assert(LROWS % (4 * WAVES) == 0, "stage must split evenly over the waves");
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#if BMF_SHAPE16
    const int r = lane & 15, h = lane >> 4;  // 16x16x32: row/column inside a 16-tile, k-group 0..3
#else
    const int r = lane & 31, h = lane >> 5;
#endif
    // per-lane DMA source: LDS row (4*q + lane/16), 16-byte chunk (lane%16) of that row holds source chunk
    // (lane%16) ^ (column & 15)
    const int d_sub = lane >> 4, d_chunk = lane & 15;
    // LDS offset of this lane's B fragment row for (nt, t): ((t*NC + tile*nt + r) * 256); chunk = ch ^ (r & 15)
    const int b_row = r * 256;
    const int b_sw = r & 15;

    auto issue_dma = [&](int stage, int buf) {
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {
            const int q = wave * DMA_PER_WAVE + i;        // wave-uniform 1 KiB piece
            const int lrow = 4 * q + d_sub;               // = t*NC + j
            const int j = lrow & (NC - 1);
            const uint16_t* src = P + (int64_t)lrow * ldp + (int64_t)stage * 128 + ((d_chunk ^ (j & 15)) << 3);
            char* dst = smem + buf * STAGE_BYTES + q * 1024;  // wave-uniform base; hardware adds lane*16
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    // stream-K: the (row tile, stage) space is linearised tile-major and cut into equal contiguous slices, one per
    // workgroup; a slice may end one tile and start the next.  Partial tiles go to slab `slot` = how many workgroups
    // started on that tile before this one, so the consumer's slab-order sum is deterministic.
    int64_t u = (int64_t)blockIdx.x * units_per_wg;
    const int64_t u_end = min(u + units_per_wg, total_units);
    while (u < u_end) {
    const int tile = (int)(u / stages);
    const int s0 = (int)(u - (int64_t)tile * stages);
    const int s1 = (int)min((int64_t)stages, s0 + (u_end - u));
    const int first_wg = (int)(((int64_t)tile * stages) / units_per_wg);
    const int slot = (int)blockIdx.x - first_wg;
    const int64_t row_base = (int64_t)tile * (WAVES * 64) + wave * 64;

#if BMF_SHAPE16
    // ---- 16x16x32 flavour: 4 x (2*NT) accumulator tiles of 16x16 (same 64 rows x kp columns per wave) ----
    constexpr int NT16 = 2 * NT;
    f32x4 acc[4][NT16];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT16; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[mt][nt][i] = 0.f;

    // lane (r, g = h) reads word g of the stage for rows row_base + 16*mt + r
    const uint32_t* a_ptr = A + (row_base + r) * ldw + h;
    unsigned a_cur[4], a_nxt[4];
    if (s0 < s1) {
        issue_dma(s0, 0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a_cur[mt] = a_ptr[(int64_t)(16 * mt) * ldw + 4 * (int64_t)s0];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int s = s0; s < s1; ++s) {
        const int cur = (s - s0) & 1;
        if (s + 1 < s1) {
            issue_dma(s + 1, cur ^ 1);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a_nxt[mt] = a_ptr[(int64_t)(16 * mt) * ldw + 4 * (int64_t)(s + 1)];
        }
        const char* buf = smem + cur * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int ch = (ks * 4 + h) ^ b_sw;
            u32x4 b[NT16][T];
#pragma unroll
            for (int nt = 0; nt < NT16; ++nt)
#pragma unroll
                for (int t = 0; t < T; ++t)
                    b[nt][t] = *reinterpret_cast<const u32x4*>(buf + (t * NC + 16 * nt) * 256 + b_row + ch * 16);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                u32x4 av;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int bit = 4 * ks + e;  // low half <- bit, high half <- bit + 16
                    av[e] = (bit <= 14 ? (a_cur[mt] << (14 - bit)) : (a_cur[mt] >> (bit - 14))) & 0x40004000u;
                }
#pragma unroll
                for (int nt = 0; nt < NT16; ++nt)
#pragma unroll
                    for (int t = 0; t < T; ++t) acc[mt][nt] = mfma_16x16x32<F16>(av, b[nt][t], acc[mt][nt]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a_cur[mt] = a_nxt[mt];
    }

    // C/D layout of 16x16 MFMA: column = lane & 15, row = 4*(lane >> 4) + reg
    float* o = out + (int64_t)slot * slab_stride;
    float osc[NT16];
#pragma unroll
    for (int nt = 0; nt < NT16; ++nt) osc[nt] = colscale ? colscale[16 * nt + r] : 0.5f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT16; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t row = row_base + 16 * mt + 4 * h + i;
                o[row * NC + 16 * nt + r] = osc[nt] * acc[mt][nt][i];
            }
    if (s1 == stages) {  // last contributor of this tile: the slab slots nobody writes must read as zero
        for (int z = slot + 1; z < slots; ++z) {
            float* oz = out + (int64_t)z * slab_stride;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT16; ++nt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int64_t row = row_base + 16 * mt + 4 * h + i;
                        oz[row * NC + 16 * nt + r] = 0.f;
                    }
        }
    }
#else
    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    // per-lane A pointers (bits of row row_base + 32*mt + r, word 2*h of the stage)
    const uint32_t* a_ptr0 = A + (row_base + r) * ldw + 2 * h;
    const uint32_t* a_ptr1 = a_ptr0 + 32 * ldw;

    u32x2 a_cur[2], a_nxt[2];
    if (s0 < s1) {
        issue_dma(s0, 0);
        a_cur[0] = *reinterpret_cast<const u32x2*>(a_ptr0 + 4 * (int64_t)s0);
        a_cur[1] = *reinterpret_cast<const u32x2*>(a_ptr1 + 4 * (int64_t)s0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int s = s0; s < s1; ++s) {
        const int cur = (s - s0) & 1;
        if (s + 1 < s1) {
            issue_dma(s + 1, cur ^ 1);
            a_nxt[0] = *reinterpret_cast<const u32x2*>(a_ptr0 + 4 * (int64_t)(s + 1));
            a_nxt[1] = *reinterpret_cast<const u32x2*>(a_ptr1 + 4 * (int64_t)(s + 1));
        }
        const char* buf = smem + cur * STAGE_BYTES;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const unsigned w0 = q ? a_cur[0].y : a_cur[0].x;
            const unsigned w1 = q ? a_cur[1].y : a_cur[1].x;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int ch = ((q * 4 + ks) * 2 + h) ^ b_sw;
                u32x4 b[NT][T];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int t = 0; t < T; ++t)
                        b[nt][t] = *reinterpret_cast<const u32x4*>(buf + (t * NC + 32 * nt) * 256 + b_row + ch * 16);
                u32x4 a0, a1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int bit = 4 * ks + i;  // low half <- bit, high half <- bit + 16
                    a0[i] = (bit <= 14 ? (w0 << (14 - bit)) : (w0 >> (bit - 14))) & 0x40004000u;
                    a1[i] = (bit <= 14 ? (w1 << (14 - bit)) : (w1 >> (bit - 14))) & 0x40004000u;
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        acc[0][nt] = mfma_32x32x16<F16>(a0, b[nt][t], acc[0][nt]);
                        acc[1][nt] = mfma_32x32x16<F16>(a1, b[nt][t], acc[1][nt]);
                    }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        a_cur[0] = a_nxt[0];
        a_cur[1] = a_nxt[1];
    }

    // C/D layout of 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
    float* o = out + (int64_t)slot * slab_stride;
    float osc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) osc[nt] = colscale ? colscale[32 * nt + r] : 0.5f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t row = row_base + 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * h;
                o[row * NC + 32 * nt + r] = osc[nt] * acc[mt][nt][i];
            }
    if (s1 == stages) {  // last contributor of this tile: the slab slots nobody writes must read as zero
        for (int z = slot + 1; z < slots; ++z) {
            float* oz = out + (int64_t)z * slab_stride;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int64_t row = row_base + 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * h;
                        oz[row * NC + 32 * nt + r] = 0.f;
                    }
        }
    }
#endif
    u += s1 - s0;
    }  // stream-K slice loop
}

struct Plan {
    int num_wgs, units_per_wg, slots;
    int64_t total;
};

int cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

// one workgroup per CU (two when the double-buffered panel stage leaves room in the 160 KiB LDS)
Plan make_plan(int64_t rows_pad, int stages, int terms, int kp) {
    Plan p;
    const int n_row_tiles = (int)(rows_pad / 512);
    p.total = (int64_t)n_row_tiles * stages;
    const int lds = 2 * terms * kp * 256;
    const int occ = (2 * lds <= 160 * 1024) ? 2 : 1;
    int64_t g = (int64_t)cu_count() * occ;
    if (g > p.total) g = p.total;
    p.units_per_wg = (int)((p.total + g - 1) / g);
    p.num_wgs = (int)((p.total + p.units_per_wg - 1) / p.units_per_wg);
    int slots = 1;
    for (int t = 0; t < n_row_tiles; ++t) {
        const int first = (int)(((int64_t)t * stages) / p.units_per_wg);
        const int last = (int)((((int64_t)(t + 1)) * stages - 1) / p.units_per_wg);
        if (last - first + 1 > slots) slots = last - first + 1;
    }
    p.slots = slots;
    return p;
}

template <int NT, int T, int WAVES, bool F16>
int launch(const uint32_t* A, int64_t ldw, int stages, const uint16_t* P, int64_t ldp, float* out, int64_t slab_stride,
           const Plan& pl, int slots, const float* colscale, const int32_t* stop, hipStream_t s) {
    dim3 grid((unsigned)pl.num_wgs), block(WAVES * 64);
    BMF_LAUNCH((xf_bits_kernel<NT, T, WAVES, F16>), grid, block, 0, s, A, ldw, stages, P, ldp, out, slab_stride,
                       pl.units_per_wg, pl.total, slots, colscale, stop);
    BMF_LAUNCH_CHECK();
    return BMF_OK;
}

}  // namespace

extern "C" int bmf_xf_bits_slots(int64_t rows_pad, int64_t red_words, int terms, int kp) {
    if (rows_pad <= 0 || rows_pad % BMF_ROW_PAD || red_words <= 0 || red_words % 4 || (kp != 32 && kp != 64) || terms < 1 ||
        terms > 3) {
        bmf_set_error("bmf_xf_bits_slots: bad arguments");
        return BMF_ERR_BAD_ARG;
    }
    return make_plan(rows_pad, (int)(red_words / 4), terms, kp).slots;
}

int bmf_xf_bits_launch(const uint32_t* Abits, int64_t rows_pad, int64_t ldw, int64_t red_words, const uint16_t* panel,
                        int64_t ldp, int terms, int kp, float* out, int64_t slab_stride, int splits, int panel_kind,
                        const float* colscale, const int32_t* stop, hipStream_t s) {
    BMF_REQUIRE(Abits && panel && out, "bmf_xf_bits: null pointer");
    BMF_REQUIRE(rows_pad > 0 && rows_pad % BMF_ROW_PAD == 0, "bmf_xf_bits: rows_pad=%lld must be a positive multiple of %d",
                (long long)rows_pad, BMF_ROW_PAD);
    BMF_REQUIRE(red_words > 0 && red_words % 4 == 0, "bmf_xf_bits: red_words=%lld must be a positive multiple of 4",
                (long long)red_words);
    BMF_REQUIRE(ldw >= red_words && ldw % 4 == 0, "bmf_xf_bits: ldw=%lld must be >= red_words and a multiple of 4",
                (long long)ldw);
    BMF_REQUIRE(ldp >= 32 * red_words && ldp % 8 == 0, "bmf_xf_bits: ldp=%lld must be >= 32*red_words and a multiple of 8",
                (long long)ldp);
    BMF_REQUIRE(kp == 32 || kp == 64, "bmf_xf_bits: kp=%d must be 32 or 64", kp);
    BMF_REQUIRE(terms >= 1 && terms <= 3, "bmf_xf_bits: terms=%d must be 1..3", terms);
    BMF_REQUIRE(panel_kind == BMF_PANEL_BF16 || (panel_kind == BMF_PANEL_F16 && colscale && terms == 2),
                "bmf_xf_bits: panel_kind must be BMF_PANEL_BF16, or BMF_PANEL_F16 with terms == 2 and a colscale vector");
    BMF_REQUIRE(slab_stride >= rows_pad * kp, "bmf_xf_bits: slab_stride too small");
    BMF_REQUIRE(bmf_aligned16(Abits) && bmf_aligned16(panel) && bmf_aligned16(out), "bmf_xf_bits: pointers must be 16-byte aligned");
    const int stages = (int)(red_words / 4);
    const Plan pl = make_plan(rows_pad, stages, terms, kp);
    BMF_REQUIRE(splits >= pl.slots, "bmf_xf_bits: splits=%d but this shape needs %d slab slots (bmf_xf_bits_slots)", splits, pl.slots);
    if (panel_kind == BMF_PANEL_F16) {
        if (kp == 32) return launch<1, 2, 8, true>(Abits, ldw, stages, panel, ldp, out, slab_stride, pl, splits, colscale, stop, s);
        return launch<2, 2, 8, true>(Abits, ldw, stages, panel, ldp, out, slab_stride, pl, splits, colscale, stop, s);
    }
#define BMF_XF_CASE(NT_, T_)                                                                                        \
    if (kp == 32 * NT_ && terms == T_)                                                                              \
        return launch<NT_, T_, 8, false>(Abits, ldw, stages, panel, ldp, out, slab_stride, pl, splits, colscale, stop, s);
    BMF_XF_CASE(1, 1) BMF_XF_CASE(1, 2) BMF_XF_CASE(1, 3) BMF_XF_CASE(2, 1) BMF_XF_CASE(2, 2) BMF_XF_CASE(2, 3)
#undef BMF_XF_CASE
    bmf_set_error("bmf_xf_bits: unsupported kp/terms");
    return BMF_ERR_UNSUPPORTED;
}

extern "C" int bmf_xf_bits(const uint32_t* Abits, int64_t rows_pad, int64_t ldw, int64_t red_words,
                           const uint16_t* panel, int64_t ldp, int terms, int kp, float* out, int64_t slab_stride,
                           int splits, void* stream) {
    return bmf_xf_bits_launch(Abits, rows_pad, ldw, red_words, panel, ldp, terms, kp, out, slab_stride, splits, BMF_PANEL_BF16,
                              nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int bmf_xf_bits_f16(const uint32_t* Abits, int64_t rows_pad, int64_t ldw, int64_t red_words,
                               const uint16_t* panel, int64_t ldp, const float* colscale, int kp, float* out,
                               int64_t slab_stride, int splits, void* stream) {
    return bmf_xf_bits_launch(Abits, rows_pad, ldw, red_words, panel, ldp, 2, kp, out, slab_stride, splits, BMF_PANEL_F16,
                              colscale, nullptr, (hipStream_t)stream);
}
