/*
 * bmf_hip.h -- C ABI of libbmf_hip.so: the MI355X (gfx950) kernels behind PyBMF's continuous-relaxation
 * multiplicative-update hot path (BinaryMFPenalty / WNMF / BinaryMFThreshold).
 *
 * PyBMF (the reference, /root/reference @ 2024_10_08) is pure Python and has no FFI of its own; the
 * "operator surface" it defines for this path is a set of NumPy call sites.  Every entry point below
 * names the reference call site(s) it replaces (file:line under PyBMF/).  INTEGRATION.md shows the
 * ctypes binding a PyBMF maintainer would add.
 *
 * Conventions
 *   - plain C, no C++ types, no torch types; all pointers are DEVICE pointers unless the name ends in
 *     _host; `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - every function returns 0 (BMF_OK) or a negative BMF_ERR_* code and never throws; the message for
 *     the last failure on the calling thread is bmf_last_error().
 *   - all launches are asynchronous on `stream`; nothing here allocates, frees or synchronises
 *     (graph-capturable), except bmf_timer_* which say so.
 *   - reference letters: X (m x n data), U (m x k), V (n x k).  Factors are stored row-major fp32 with
 *     leading dimension kp (k rounded up to 32, kp <= 64; a rank 64 < k <= 128 is held as two 64-column blocks: the last section of
 *     this header) and BMF_ROW_PAD-padded row counts; padded rows
 *     and columns hold zeros and stay zero.
 *
 * Data layouts
 *   bit matrix   : uint32 words, row-major, `ldw` words per row; bit c of row r is
 *                  (bits[r*ldw + c/32] >> (c%32)) & 1  (numpy.packbits(..., bitorder='little') viewed as
 *                  little-endian uint32).  Row count padded to BMF_ROW_PAD, columns padded (with zero bits)
 *                  to a multiple of BMF_RED_PAD.
 *   factor panel : the transposed factor split into `terms` bf16 addends (F = t0 + t1 + t2, each bf16,
 *                  3 terms reproduce the fp32 value exactly), panel[t][j][p] for term t, factor column j,
 *                  position p; leading dimension ldp (elements).  Inside each aligned block of 128
 *                  reduction indices the order is permuted to the order in which the MFMA kernel consumes
 *                  bits: index c = 128*b + cl is stored at p = 128*b + bmf_panel_pos(cl) (see below).
 */
#ifndef BMF_HIP_H
#define BMF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BMF_OK 0
#define BMF_ERR_BAD_ARG (-1)
#define BMF_ERR_HIP (-2)
#define BMF_ERR_UNSUPPORTED (-3)
#define BMF_ERR_COMM (-4) /* RCCL missing / an RCCL call or the host all-reduce callback failed */

#define BMF_ROW_PAD 512 /* row padding of factors, slabs and bit matrices */
#define BMF_RED_PAD 128 /* bit-column (reduction) padding */
#define BMF_MAX_KP 64
#define BMF_LOG_COLS 16

/* epilogue modes */
#define BMF_MODE_PREPARE 0 /* no update: rebuild panels/bits/partials from F as it is */
#define BMF_MODE_PENALTY 1 /* BinaryMFPenalty update (penalty terms, both eps clamps) */
#define BMF_MODE_WNMF 2    /* WNMF Frobenius update (no penalty, only denom==0 -> eps) */

/* log row columns written by bmf_penalty_finalize (fp64) */
enum {
    BMF_LOG_ITER = 0, BMF_LOG_ERROR, BMF_LOG_REC, BMF_LOG_REG, BMF_LOG_REGERR, BMF_LOG_RMSE, BMF_LOG_MAE,
    BMF_LOG_TP, BMF_LOG_FP, BMF_LOG_FN, BMF_LOG_TN, BMF_LOG_VALID, BMF_LOG_STOP
};

/* 100 * round + revision; bumped whenever a struct below changes size or meaning (400: round 4 -- bmf_masked_loop, the scale
 * contract of the fused digit planes: plane_scale / scaleU / scaleV hold 4 * kp floats; 501: bmf_wnmf_real_state.UT3 / VT3) */
#define BMF_ABI_VERSION 501
int bmf_version(void);
const char* bmf_last_error(void);
/* sizeof() of the argument structs as THIS library was compiled, so that a binding can refuse a mismatch before the first call
 * (the structs that are passed without a struct_bytes field of their own included): which = 0 bmf_epilogue_args, 1 bmf_palm_args,
 * 2 bmf_penalty_state, 3 bmf_wnmf_real_state, 4 bmf_palm_state, 5 bmf_masked_loop, 6 bmf_masked_side, 7 bmf_link_loop; -1 for an unknown id. */
int bmf_struct_bytes(int which);

/* position of local reduction index cl (0..127) inside its 128-block of a factor panel (host helper) */
int bmf_panel_pos(int cl);

/* ---- data preparation ------------------------------------------------------------------------------ */

/* Pack a dense 0/1 byte matrix (nonzero = 1) into a bit matrix.  Replaces the dense fp64 X_train the
 * reference keeps (models/ContinuousModel.py:169-203).  X: rows x cols, leading dim ldx bytes.
 * bits: rows x ldw words (ldw even); each row gets ceil(cols/64)*2 words written (bits beyond cols are 0), the
 * rest of the row is left untouched, so a column block of a wider bit matrix can be filled in place. */
int bmf_pack_rows_u8(const uint8_t* X, int64_t rows, int64_t cols, int64_t ldx, uint32_t* bits, int64_t ldw,
                     void* stream);

/* Number of set bits of a rows x words bit matrix, added to *count (device uint64; caller zeroes it).
 * Gives sum(X) = ||X||_F^2 for the trace form of rec_error (models/BinaryMFPenalty.py:175-179). */
int bmf_popcount(const uint32_t* bits, int64_t rows, int64_t words, int64_t ldw, unsigned long long* count,
                 void* stream);

/* Build a factor panel from a row-major fp32 factor F (rows_pad x ldf) -- stand-alone form of what
 * bmf_mu_epilogue does in-line; used for initial factors and by the tests. */
int bmf_make_panel(const float* F, int64_t rows_pad, int64_t ldf, int kp, int terms, uint16_t* panel, int64_t ldp,
                   void* stream);

/* fp16 panel (BMF_PANEL_F16): F[:, c] * 2^e_c = hi + lo as two fp16 addends (22 significant bits relative to the column
 * maximum), e_c = the power of two that brings max|F[:, c]| into [2^14, 2^15).  Three small kernels: column maxima per
 * 128-row block, scales, split.  panel: [2][kp][ldp] fp16 in the same position-permuted order as the bf16 panel.
 * scale (out, 2*kp floats): scale[c] = 2^e_c, scale[kp + c] = 0.5 / 2^e_c -- the `colscale` argument of bmf_xf_bits_f16.
 * ws: (rows_pad / 128) * kp floats of device scratch. */
#define BMF_PANEL_BF16 0
#define BMF_PANEL_F16 1
int bmf_make_panel_f16(const float* F, int64_t rows_pad, int64_t ldf, int kp, uint16_t* panel, int64_t ldp, float* ws,
                       float* scale, void* stream);

/* ---- the two big contractions ------------------------------------------------------------------------ */

/* out[s][r][j], s < splits: partial sums of  sum_c A[r][c] * F[c][j]  for a 0/1 bit matrix A; the full product is the
 * sum of the `splits` slabs (slab_stride floats apart) added in slab order (deterministic, no atomics).
 *   A = X bits   , panel of V:  X @ V       = multiply(W, X) @ V      models/BinaryMFPenalty.py:139, WNMF.py:105
 *   A = X^T bits , panel of U:  X^T @ U     = multiply(W, X).T @ U    models/BinaryMFPenalty.py:154, WNMF.py:98
 * bf16 MFMA (v_mfma_f32_32x32x16_bf16), bits expanded to bf16 in registers, fp32 accumulation.  The (row tile, reduction
 * stage) space is cut stream-K fashion into equal slices, one per persistent workgroup (one or two per CU); a row
 * tile therefore receives a shape- and device-dependent number of partial results: bmf_xf_bits_slots() says how many
 * slabs the caller must provide at least (`splits` >= that; surplus slabs are written as zeros).
 * rows_pad % 512 == 0, red_words % 4 == 0 (reduction length in 32-bit words), ldw >= red_words, ldw % 4 == 0,
 * ldp >= 32*red_words and ldp % 8 == 0, kp in {32, 64}, terms in {1,2,3}. */
int bmf_xf_bits_slots(int64_t rows_pad, int64_t red_words, int terms, int kp); /* >= 1, or a negative BMF_ERR_* */
int bmf_xf_bits(const uint32_t* Abits, int64_t rows_pad, int64_t ldw, int64_t red_words, const uint16_t* panel,
                int64_t ldp, int terms, int kp, float* out, int64_t slab_stride, int splits, void* stream);
/* The same contraction on an fp16 panel (bmf_make_panel_f16; v_mfma_f32_16x16x32_f16, two addends): out[:, c] = colscale[c] *
 * (bits(A) . panel)[:, c], colscale = scale + kp of the panel builder. */
int bmf_xf_bits_f16(const uint32_t* Abits, int64_t rows_pad, int64_t ldw, int64_t red_words, const uint16_t* panel,
                    int64_t ldp, const float* colscale, int kp, float* out, int64_t slab_stride, int splits, void* stream);

/* The same contraction on the INTEGER matrix cores, exact: the factor as `limbs` planes of signed 8-bit digits
 * (bmf_make_panel_i8), v_mfma_i32_16x16x64_i8 with int32 accumulation, planes recombined in fp64:
 * out[:, c] = colscale[c] * sum_l 256^l (bits(A) . plane_l)[:, c] -- no rounding before the final conversion to fp32, so the
 * result is the exact product of A with the quantised factor.  panel: [limbs][kp][ldp] int8 (ldp bytes, >= the padded
 * reduction length, % 16 == 0), inside each 512-block in the order bmf_panel_pos_i8 (cl in 0..511).  colscale = scale + kp of the builder.
 * Reduction length < 2^24, padded to a multiple of 512 (red_words % 16 == 0).  Slab slots: bmf_xf_bits_i8_slots (this kernel tiles rows by 256). */
#define BMF_PANEL_I8 2
int bmf_panel_pos_i8(int cl);
int bmf_xf_bits_i8_slots(int64_t rows_pad, int64_t red_words, int kp); /* >= 1, or a negative BMF_ERR_* */
int bmf_xf_bits_i8_occupancy(int limbs); /* workgroups per CU the runtime grants the kernel (designed for 2); needs a GPU */
/* Which kernel serves launches over a whole 64-column factor (kp == 64): 0 = 64 rows x 32 columns per wave, two 4-wave workgroups
 * per CU (the default); 1 = 32 rows x 64 columns per wave, one 8-wave workgroup per CU; 2 = 64 x 64 per wave, one wave per SIMD
 * (both in csrc/xf_bits_i8w.hip); 4 = 64 x 32 per wave in ONE 8-wave workgroup per CU on 512-row tiles whose two wave groups alternate
 * between a matrix phase and a load phase (csrc/xf_bits_i8p.hip; needs the tiled bit matrix and three planes).  Results are the same
 * exact sums, slab boundaries differ; none is faster than 0 (profiles/r04_i8_wide_tile.md, r05_i8_antiphase.md).  v < 0 only
 * queries.  Returns the previous value, or a negative BMF_ERR_*.  bmf_xf_bits_i8_slots depends on it: set it before sizing slab arrays. */
int bmf_xf_bits_i8_variant(int v);
int bmf_xf_bits_i8(const uint32_t* Abits, int64_t rows_pad, int64_t ldw, int64_t red_words, const int8_t* panel, int64_t ldp,
                   int limbs, const float* colscale, int kp, float* out, int64_t slab_stride, int splits, int a_tiled, void* stream);
/* The bit matrix re-laid for that kernel (a_tiled = 1): the 256 rows x 16 words one workgroup consumes per group of four stages
 * become one contiguous 16-KiB block, blocks in (row tile, group) order:
 *   tiled[((tile * (red_words / 16) + grp) * 256 + row) * 16 + w] = bits[(tile * 256 + row) * ldw + 16 * grp + w].
 * Every 1-KiB DMA piece of the kernel is then consecutive bytes instead of sixteen 64-byte pieces of sixteen rows.
 * tiled: rows_pad * red_words words.  rows_pad % 256 == 0, red_words % 16 == 0. */
int bmf_tile_bits(const uint32_t* bits, int64_t rows_pad, int64_t ldw, int64_t red_words, uint32_t* tiled, void* stream);
/* ---- the same contraction on the 2:4 structured-sparse integer matrix instruction (csrc/xf_bits_i8s.hip, csrc/s24.h) ----------
 * v_smfmac_i32_16x16x128_i8 takes an A with at most two non-zeros per aligned group of four reduction indices (here: the bits
 * {s, s+8, s+16, s+24} of one 32-bit word) at twice the reduction length per instruction.  The "S24" form of a bit matrix keeps the
 * first two ones of every group (value bits + 2-bit positions, 1.5 bits per cell; layout: csrc/s24.h); the ones that do not fit
 * ("overflow") go to a CSR list and are added exactly by bmf_s24_overflow.  Rows with many overflow ones are kept out of the form by
 * a row selection and served by the dense kernel (bmf_xf_bits_i8_rows).  Same call sites as bmf_xf_bits_i8: BinaryMFPenalty.py:139,154.
 *
 * bmf_s24_bytes: size of the S24 form of rows_pad (% 256) rows x red_words (% 16) words; -1 on bad arguments.
 * bmf_s24_count: counts[row] = overflow ones of each of `rows` rows of a plain bit matrix (ldw words per row).
 * bmf_s24_pack: packed row p = row rowsel[p] of `bits` (rowsel == NULL: row p; a negative entry: an all-zero padding row), p <
 *   rows_pad_s.  ovf_idx != NULL: the overflow ones of packed row p are written (as reduction indices, any order) to
 *   ovf_idx[ovf_ptr[p] ..), ovf_ptr = exclusive prefix sums of the rows' counts, ovf_cursor = rows_pad_s zeroed int32 of scratch.
 *   kept != NULL: the bit matrix of the packed rows with the overflow ones cleared (plain layout, ldk words per row; test aid). */
int64_t bmf_s24_bytes(int64_t rows_pad, int64_t red_words);
int bmf_s24_count(const uint32_t* bits, int64_t rows, int64_t ldw, int64_t red_words, int32_t* counts, void* stream);
int bmf_s24_pack(const uint32_t* bits, int64_t ldw, int64_t red_words, const int32_t* rowsel, int64_t rows_pad_s, uint32_t* s24,
                 const int64_t* ovf_ptr, int32_t* ovf_cursor, int32_t* ovf_idx, uint32_t* kept, int64_t ldk, void* stream);
/* out[s][rowmap[p]][j] (rowmap == NULL: row p; a negative entry: skipped), three digit planes, the slab contract of bmf_xf_bits_i8;
 * slots of `out` beyond those this launch reaches are zero-filled for its rows (splits >= bmf_xf_bits_i8s_slots). */
int bmf_xf_bits_i8s_slots(int64_t rows_pad_s, int64_t red_words, int kp);
int bmf_xf_bits_i8s_occupancy(void);
/* Which form of the kernel bmf_xf_bits_i8s launches: 0 = four waves / 256-row tiles / two workgroups per CU (default), 1 = eight waves /
 * 512-row tiles / one workgroup per CU with the DMA roles split between the waves (rows_pad_s % 512 == 0), 2 = form 1 with its two wave
 * groups in anti-phase (one issues only matrix instructions while the other loads; two barriers per stage).  Same results (sums of
 * slabs), measured within 8 % of each other (profiles/r05_i8_smfmac.md).  v < 0 only queries; returns the previous value.
 * bmf_xf_bits_i8s_slots depends on it: set it before sizing slab arrays. */
int bmf_xf_bits_i8s_form(int v);
int bmf_xf_bits_i8s(const uint32_t* s24, int64_t rows_pad_s, int64_t red_words, const int8_t* panel, int64_t ldp, const float* colscale,
                    int kp, float* out, int64_t slab_stride, int splits, const int32_t* rowmap, void* stream);
/* out[0][rowmap[p]][c] += colscale[c] * sum over the overflow ones j of packed row p of q(j, c), q = rint(F64[j][c] / colscale[c])
 * clamped to three balanced digits -- the integer the digit planes hold (bmf_make_panel_i8) -- summed in int64.  After the GEMM
 * launches that write slot 0 of those rows, on the same stream. */
int bmf_s24_overflow(const int64_t* ovf_ptr, const int32_t* ovf_idx, const int32_t* rowmap, int64_t prows, const double* F64, int64_t ldf,
                     const float* colscale, int kp, float* out, void* stream);
/* int8 limb panel of a factor: q = rint(F64[:, c] 2^e_c), e_c = the power of two that puts max|F[:, c]| in [2^22, 0.996 * 2^23]
 * (the largest number three balanced digits hold is 127 * 65793; a column maximum above it takes [2^21, 2^22)), written
 * in balanced base-256 digits q = d2 2^16 + d1 2^8 + d0 (limbs = 3; limbs = 2 keeps d2, d1 of q rounded to a multiple of 256).
 * F: the fp32 shadow of F64 (column maxima are taken from it); rows_pad % 512 == 0; ws: (rows_pad / 128) * kp floats;
 * scale (out): 2 * kp floats. */
int bmf_make_panel_i8(const double* F64, const float* F, int64_t rows_pad, int64_t ldf, int kp, int limbs, int8_t* panel,
                      int64_t ldp, float* ws, float* scale, void* stream);

/* Same contraction for a real-valued fp32 A (WNMF on non-Boolean data): exact-fp32 MFMA
 * (v_mfma_f32_32x32x2_f32).  A: rows_pad x lda floats, reduction length red (multiple of 8, zero padded),
 * FT: the transposed factor, FT[j][c] fp32 with leading dim ldft. */
int bmf_xf_f32(const float* A, int64_t rows_pad, int64_t lda, int64_t red, const float* FT, int64_t ldft, int kp,
               float* out, int64_t slab_stride, int splits, void* stream);
/* A real-valued matrix re-laid for the LDS-ring kernels: block (tile, st) = rows 64 tile .. + 63, floats 64 st .. + 63 is one
 * contiguous 16 KiB made of four quarters q = 2 kh + rw (rows 32 rw .. + 31, floats 32 kh .. + 31) of 4 KiB, each stored as the
 * swizzled LDS image the wave that owns it reads:
 *   tiled[((((tile * (red / 64) + st) * 4 + q) * 32 + rl) * 8 + c) * 4 + e]
 *       = X[(64 tile + 32 rw + rl) * lda + 64 st + 32 kh + 4 (c ^ ((rl >> 1) & 7)) + e],
 * so that a wave fetches its share of a stage as 4 consecutive KiB instead of 32 row pieces of 128 bytes.
 * rows_pad % 64 == 0, red % 64 == 0; tiled: rows_pad * red floats. */
int bmf_tile_f32(const float* X, int64_t rows_pad, int64_t lda, int64_t red, float* tiled, void* stream);
/* The factor operand of bmf_xf_f32_tiled, in the order the kernel's lanes consume it (one contiguous KiB per load instruction
 * instead of 16 bytes of 64 cache lines): F is rows_pad x kp fp32 row-major (NOT transposed), rows_pad % 64 == 0, NT = kp / 32,
 *   frag[((((st * 2 + kh) * 4 + u) * NT + nt) * 64 + 32 h + r) * 4 + t] = F[(64 st + 32 kh + 8 u + 4 h + t) * kp + 32 nt + r]. */
int bmf_frag_f32(const float* F, int64_t rows_pad, int kp, float* frag, void* stream);
/* out = A . F like bmf_xf_f32, from the tiled copy of A (bmf_tile_f32) and the fragment-ordered factor (bmf_frag_f32 of the
 * red x kp factor); same numbers as bmf_xf_f32 on A and F^T bit for bit. */
int bmf_xf_f32_tiled(const float* Atiled, int64_t rows_pad, int64_t red, const float* Ffrag, int kp, float* out,
                     int64_t slab_stride, int splits, void* stream);
/* The same contraction with the residual sums of the pass folded in (kp == 32): out = A F, and sums[0] += sum |A - G F^T|, sums[1] +=
 * sum (A - G F^T)^2 over the cells of A, for a second factor G (Grow: rows_pad x 32 fp32, plain rows).  Frf = bmf_frag_rows_bf16 of F:
 * the residual product runs on the bf16 MFMA with both factors split into two bf16 addends (right to 2^-16 per cell).
 * With A = X^T, F = U, G = V this is X^T U plus the RMSE / MAE sums of WNMF.error (WNMF.py:132-144) in ONE read of X. */
int bmf_xf_f32_tiled_resid(const float* Atiled, int64_t rows_pad, int64_t red, const float* Ffrag, const uint32_t* Frf, const float* Grow, int kp,
                           float* out, int64_t slab_stride, int splits, double* sums, void* stream);
/* The same contractions (kp = 32, tiled A) on v_mfma_f32_32x32x16_bf16 with BOTH operands split three ways into bf16 by truncation
 * (x = hi + mid + lo exactly; six products per k-step reproduce the fp32 product to 2^-23): 12 x 32 cycles of matrix pipe per wave and
 * 64 x 64 stage-quarter instead of 16 x 64 -- the passes of config #2 then run at the rate of the LDS-DMA stream of A.
 * bmf_frag_bf16x3: frag3 (rows_pad x 48 words) of F (rows_pad x 32 fp32); bmf_xf_f32_tiled_bf3 = bmf_xf_f32_tiled with F3 = frag3 of the
 * factor; bmf_xf_f32_tiled_resid_bf3 = bmf_xf_f32_tiled_resid likewise (Frf, Grow, sums as there).  Same call sites: WNMF.py:98,105. */
int bmf_frag_bf16x3(const float* F, int64_t rows_pad, uint32_t* frag3, void* stream);
int bmf_xf_f32_tiled_bf3(const float* Atiled, int64_t rows_pad, int64_t red, const uint32_t* F3, float* out, int64_t slab_stride, int splits,
                         void* stream);
int bmf_xf_f32_tiled_resid_bf3(const float* Atiled, int64_t rows_pad, int64_t red, const uint32_t* F3, const uint32_t* Frf, const float* Grow,
                               float* out, int64_t slab_stride, int splits, double* sums, void* stream);
/* frag (rows_pad * kp / 4 pieces of 16 bytes = rows_pad * kp uint32): F (rows_pad x 32 fp32) as bf16 pairs hi + lo in the row-fragment
 * order of bmf_xf_f32_tiled_resid (the formula is in csrc/xf_f32.hip at frag_rows_bf16_kernel). */
int bmf_frag_rows_bf16(const float* F, int64_t rows_pad, int kp, uint32_t* frag, void* stream);

/* ---- k x k Gram ---------------------------------------------------------------------------------------- */

/* slabs[b] = partial F^T F over a row range (fp32 MFMA); blocks = number of partial slabs (<= 1024).
 * Together with bmf_reduce_slabs this is U^T U / V^T V, i.e. the re-associated denominators
 * (U V^T)^T U = V (U^T U)  models/BinaryMFPenalty.py:142,157; WNMF.py:99,106. */
int bmf_gram_partial(const float* F, int64_t rows_pad, int64_t ldf, int kp, float* slabs, int blocks, void* stream);

/* out32[i] / out64[i] = sum_b slabs[b*stride + i] (fp64 accumulation, fixed order); either output may be NULL. */
int bmf_reduce_slabs(const float* slabs, int64_t stride, int count, int64_t n, float* out32, double* out64,
                     void* stream);

/* ---- fused multiplicative-update epilogue ----------------------------------------------------------------- */

typedef struct {
    double* F64;       /* rows_pad x kp factor, fp64 master copy, updated in place */
    float* F;          /* rows_pad x kp fp32 shadow of F64: must equal (float)F64 on entry to an update (it is the MFMA
                          operand of F G); rewritten by every call, PREPARE mode included; 16-byte aligned */
    int64_t rows_pad;  /* multiple of 128 */
    int32_t rows;      /* real rows */
    int32_t k, kp;
    const float* num;  /* numerator slabs from bmf_xf_bits/_f32: [splits][rows_pad][kp]; NULL in PREPARE mode w/o dot */
    int64_t slab_stride;
    int32_t splits;
    const float* G;    /* kp x kp Gram of the OTHER factor (fp32) */
    double reg;        /* lambda of this iteration (PENALTY mode) */
    int32_t mode;      /* BMF_MODE_* */
    float thr;         /* threshold for the Boolean bits (strict >) */
    int32_t terms;     /* bf16 addends of the panel built in-line; 0 = none (fp16 panels are built afterwards) */
    uint16_t* panel;   /* out: panel of the updated factor [terms][kp][ldp] */
    int64_t ldp;
    uint64_t* rowbits; /* out: [rows_pad], bit j = F[r][j] > thr (0 for padded rows/cols) */
    uint32_t* colbits; /* out: [kp][ldcb] words, bit r of word r/32 = F[r][j] > thr */
    int64_t ldcb;
    double* partials;  /* out: [rows_pad/128][2] = { sum (F^2-F)^2 , sum F_new * num } per block */
    const int32_t* stop; /* optional device flag: kernel is a no-op when *stop != 0 */
    const float* den;  /* optional, rows_pad x kp: the contraction part of the denominator, precomputed (masked path:
                          (W o (F F_other^T)) F_other from bmf_masked_pass); when given, G is not used */
    float* blockmax;   /* optional out: [rows_pad/128][kp] column maxima of the new factor per 128-row block (input of the
                          fp16 / int8 panel builders) */
    int64_t num_block_stride; /* 0: num is [splits][rows_pad][kp]; else num is stored in 32-column blocks, [kp/32][rows_pad][32]
                          with this many elements between blocks (the sharded exchange buffer); splits must then be 1 */
    int8_t* planes;    /* optional out (terms must be 0, blockmax given, rows_pad % 512 == 0): the int8 digit planes of the new factor,
                          [limbs][kp][ldp] in the order of bmf_make_panel_i8, built in-line with the column scales in plane_scale */
    const float* plane_scale; /* [kp]: 2^e_c used for `planes` -- a prediction from the previous iteration's column maxima; the
                          caller checks it against `blockmax` afterwards and rebuilds with bmf_make_panel_i8 when it was off */
    int32_t limbs;     /* digits per entry in `planes`: 2 or 3 */
    int32_t _pad0;
} bmf_epilogue_args;

/* One factor update, fused:  F <- F o (num + 3 reg F^2) / (F G + 2 reg F^3 + reg F), denom==0 -> eps,
 * F==0 -> eps (models/BinaryMFPenalty.py:136-163; WNMF.py:96-109 in WNMF mode), element-wise part in fp64 on the fp64
 * master copy (the two contractions num and F G are fp32-accurate), plus everything the next
 * kernels need from the new factor: its bf16 panel, its thresholded bits (utils/common.py:64-79 binarize),
 * the regulariser sum (BinaryMFPenalty.py:182-186) and sum(F_new o num) for the trace form of rec_error. */
int bmf_mu_epilogue(const bmf_epilogue_args* args, void* stream);

/* ---- masked update (W = 'mask' / a weight matrix): sparse contractions over the observed cells ------------------------- */

/* CSR list of observed cells of one orientation (rows = rows of F_self): ptr[rows+1], idx[nnz] (row index into F_other),
 * val[nnz] (x_e), wgt[nnz] (w_e, NULL = 1).  For every row r:
 *   num[r][:] = sum_e w_e x_e F_other[idx_e][:]                 = (W o X) F_other        models/BinaryMFPenalty.py:139,154
 *   den[r][:] = sum_e w_e <F_self[r], F_other[idx_e]> F_other[idx_e][:] = (W o (F_self F_other^T)) F_other   :142,157
 * and, if sums != NULL:  sums[0] += sum_e w_e (x_e - p_e)^2, sums[1] += sum_e w_e |x_e - p_e|   (rec_error :175-179).
 * Rows are cut into segments of at most 64 consecutive cells (load balance on power-law rows): seg_row[nseg], seg_beg[nseg]
 * (first cell of the segment), row_seg_ptr[rows+1] (segments of row r = [row_seg_ptr[r], row_seg_ptr[r+1]), in cell order);
 * part = 2*nseg*kp floats of scratch.  Segment partials are added per row in segment order (deterministic).
 * F_self / F_other / num / den: row-major fp32 with leading dimension kp.  Call it with the CSC list to update V. */
int bmf_masked_pass(const int64_t* ptr, const int32_t* idx, const float* val, const float* wgt, int32_t rows,
                    const int32_t* seg_row, const int64_t* seg_beg, int32_t nseg, const int64_t* row_seg_ptr,
                    const float* Fself, const float* Fother, int kp, float* part, float* num, float* den, double* sums,
                    void* stream);

/* The same pass with the product sent through a link first (BMF_LINK_SIGMOID: PNLPF under a mask / weight matrix,
 * models/PNLPF.py:61-91): with s = lamda (p_e - 1/2), sig = sigmoid(s), d = sig (1 - sig)
 *   num[r][:] = lamda sum_e w_e x_e d_e F_other[idx_e][:]      = link_lamda * multiply(W, multiply(X, d_sig)) @ V     :65,81
 *   den[r][:] = lamda sum_e w_e sig_e d_e F_other[idx_e][:]    = link_lamda * multiply(W, multiply(sig, d_sig)) @ V   :68,84
 *   sums[0] += sum_e w_e (x_e - sig_e)^2, sums[1] += sum_e w_e |x_e - sig_e|     (rec_error against the link prediction).
 * BMF_LINK_KL (WNMF, Kullback-Leibler loss under a weight matrix, models/WNMF.py:111-129): num[r][:] = sum_e w_e x_e / p_e F_other[idx_e][:]
 * = ((W o X) / (U V^T)) F_other; den is left at zero (the reference's denominator O F_other uses the all-ones matrix: the column sums of
 * F_other, supplied by the caller); sums[0] += 2 sum_e w_e (x_e log(x_e / p_e) - x_e + p_e), 0 log 0 = 0 (:143-145; twice, so that
 * 0.5 sums[0] is the error as for the other models).  link = 0 is bmf_masked_pass. */
int bmf_masked_link_pass(const int64_t* ptr, const int32_t* idx, const float* val, const float* wgt, int32_t rows,
                         const int32_t* seg_row, const int64_t* seg_beg, int32_t nseg, const int64_t* row_seg_ptr,
                         const float* Fself, const float* Fother, int kp, float* part, float* num, float* den, double* sums,
                         int link, double lamda, void* stream);
/* The same, told how many of the kp columns are real (kcols <= kp; columns kcols .. kp - 1 of BOTH factors must be zero, as every
 * engine keeps its padding): with kp = 32 a wave then takes two cells per step (kcols <= 32) or four (kcols <= 16), one per group of 32 /
 * 16 lanes, instead of one -- a quarter of the gather instructions and dependent steps at k = 16. */
int bmf_masked_link_pass_k(const int64_t* ptr, const int32_t* idx, const float* val, const float* wgt, int32_t rows,
                           const int32_t* seg_row, const int64_t* seg_beg, int32_t nseg, const int64_t* row_seg_ptr,
                           const float* Fself, const float* Fother, int kp, int kcols, float* part, float* num, float* den,
                           double* sums, int link, double lamda, void* stream);

/* Confusion counts over the observed cells only (task='prediction': utils/evaluate_utils.py:32-44 + utils/metrics.py:56-77):
 * for cell e = (cell_row[e], idx[e]) with value val[e]: pd = (bits_self[row] & bits_other[col]) != 0, gt = val != 0;
 * counts[0..3] += TP, FP, FN, TN (device uint64, caller zeroes).  bits_*: one k-bit word per factor row (rowbits). */
int bmf_masked_counts(const int32_t* cell_row, const int32_t* idx, const float* val, int64_t nnz, const uint64_t* bits_self,
                      const uint64_t* bits_other, unsigned long long* counts, void* stream);
/* 64 < k <= 128 (two blocks of 64 factor columns, pybmf_amd/wide.py; the reference has no rank limit, BinaryMFPenalty.py:32):
 * bmf_masked_pass_wide = bmf_masked_pass with p_e the dot product over both blocks and the numerators / denominators per block
 * (part0 / part1: [max(nseg, 1)][2][64] floats of scratch each; num*, den*: [rows][64]); bmf_masked_counts_wide = bmf_masked_counts
 * with two k-bit words per factor row, pd = ((self0 & other0) | (self1 & other1)) != 0. */
int bmf_masked_pass_wide(const int64_t* ptr, const int32_t* idx, const float* val, const float* wgt, int32_t rows, const int32_t* seg_row,
                         const int64_t* seg_beg, int32_t nseg, const int64_t* row_seg_ptr, const float* Fself0, const float* Fself1,
                         const float* Fother0, const float* Fother1, float* part0, float* part1, float* num0, float* num1, float* den0,
                         float* den1, double* sums, void* stream);
int bmf_masked_counts_wide(const int32_t* cell_row, const int32_t* idx, const float* val, int64_t nnz, const uint64_t* bits_self0,
                           const uint64_t* bits_self1, const uint64_t* bits_other0, const uint64_t* bits_other1, unsigned long long* counts,
                           void* stream);
/* The same question for a REAL-valued ground truth over the WHOLE matrix (task='reconstruction' on data that is not 0 / 1: the
 * reference's metrics are arithmetic on two csr matrices, utils/metrics.py:56-77): X fp32 row-major (ld floats per row), pd =
 * (ubits[i] & vbits[j]) != 0; out[0..5] += sum gt pd (TP), sum max(pd - gt, 0) (FP), sum max(gt - pd, 0) (FN), sum (1 - gt)(1 - pd)
 * (TN), sum gt, sum pd  (device doubles, caller zeroes; fp64 atomics across workgroups). */
int bmf_real_confusion(const float* X, int64_t ld, int32_t m, int32_t n, const uint64_t* ubits, const uint64_t* vbits, double* out, void* stream);

/* sum += sum |X - U V^T| over all cells of a Boolean X (the MAE numerator, utils/metrics.py:156-160) on the bf16 MFMA: both
 * factors split into two bf16 addends, P = Uh Vh^T + Uh Vl^T + Ul Vh^T in fp32 (the product is right to 2^-16, which a sum
 * of absolute values over m n cells does not see), 16x the rate of the exact-fp32 pass of bmf_residual_sums.
 * XTbits: the TRANSPOSED bit matrix (n_pad rows x ldxt words).  U: m_pad x kp, V: n_pad x kp fp32, zero padded (padded cells
 * then add 0).  m_pad % 256 == 0, n_pad % 64 == 0, ldxt % 4 == 0, XTbits 16-byte aligned (its rows are fetched by 16-byte
 * LDS-DMA).  ws: 2 * (m_pad + n_pad) * kp uint16 of scratch.  `sum`: device fp64, caller zeroes. */
int bmf_mae_sum(const uint32_t* XTbits, int64_t ldxt, int64_t m_pad, int64_t n_pad, const float* U, const float* V, int kp,
                uint16_t* ws, double* sum, void* stream);
/* The same with the operand precision spelled out: one_product = 0: the three-product bf16 split above; 1: ONE fp16 addend per
 * factor and a single product (a third of the MFMA work; per-cell error ~2e-4 |P|, unbiased -- the error of the sum is that
 * over sqrt(cells)); < 0: chosen by size (single product from 2^24 padded cells on), which is what bmf_mae_sum and the
 * iteration driver do. */
int bmf_mae_sum_ex(const uint32_t* XTbits, int64_t ldxt, int64_t m_pad, int64_t n_pad, const float* U, const float* V, int kp,
                   uint16_t* ws, double* sum, int one_product, void* stream);
/* The single-product pass reading X^T from its bmf_tile_bits copy (XTtiled = bmf_tile_bits(XTbits, n_pad, ldxt, ldxt, .), the copy
 * the int8 GEMM streams): a stage's words are then half of one contiguous 4-KiB piece instead of 32 bytes of each of 64 rows.
 * Needs n_pad % 256 == 0 and ldxt == m_pad / 32, a multiple of 16.  The iteration driver uses it when the state has XTtiled. */
int bmf_mae_sum_tiled(const uint32_t* XTtiled, int64_t ldxt, int64_t m_pad, int64_t n_pad, const float* U, const float* V, int kp,
                      uint16_t* ws, double* sum, void* stream);

/* ---- Boolean cover count ------------------------------------------------------------------------------------ */

/* counts[0] += TP = sum X and pd, counts[1] += FP = sum (not X) and pd, with pd[i][j] = OR_l ubits[i][l] & V_l[j]
 * = min(1, (U>u) @ (V>v)^T)  (utils/common.py:110-151) scored as utils/metrics.py:56-68.  Integer, exact.
 * Xbits: m_pad x ldx words; rowbits: per X row; colbits: [kp][ldcb] per factor bit-column over X columns. */
int bmf_cover_count(const uint32_t* Xbits, int64_t rows_pad, int64_t ldx, int64_t words, const uint64_t* rowbits,
                    const uint32_t* colbits, int64_t ldcb, int kp, unsigned long long* counts, const int32_t* stop,
                    void* stream);

/* Per-row confusion counts of two bit matrices of equal shape (ground truth G, prediction P):
 * tp[r] = |G_r and P_r|, fp[r] = |not G_r and P_r|  -- TP / FP with axis=1 of utils/metrics.py:56-68 (pass the transposed
 * bit matrices for axis=0); FN = rowsum(G) - TP, TN follows.  Feeds coverage_score / weighted_error (metrics.py:182-201). */
int bmf_confusion_rows(const uint32_t* Gbits, int64_t ldg, const uint32_t* Pbits, int64_t ldp, int64_t rows, int64_t words,
                       uint32_t* tp, uint32_t* fp, void* stream);

/* out[i][w] = OR over the factors l set in rowbits[i] of colbits[l][w]: the Boolean product itself as a bit matrix
 * (self.X_pd of the reference, utils/common.py:147-149), rows x words, leading dimension ldo words. */
int bmf_boolean_product_bits(const uint64_t* rowbits, int64_t rows, const uint32_t* colbits, int64_t ldcb, int kp,
                             int64_t words, uint32_t* out, int64_t ldo, void* stream);

/* out[i][j] = sum_k U[i][k] V[j][k] as a dense m x n fp32 matrix (leading dimension ldo): the real-valued prediction
 * get_prediction(U, V, boolean=False) of utils/common.py:98-107 / self.X_pd of WNMF (models/WNMF.py:47), exact-fp32 MFMA.
 * U: m_pad x kp, V: n_pad x kp (fp32, zero padded). */
int bmf_real_product(const float* U, int64_t m_pad, int32_t m, const float* V, int64_t n_pad, int32_t n, int kp, float* out,
                     int64_t ldo, void* stream);

/* ---- residual pass (MAE / direct rec_error) ------------------------------------------------------------------- */

/* sums[0] += sum |X - U V^T|, sums[1] += sum (X - U V^T)^2 over the real m x n cells (fp64 device accumulators,
 * caller zeroes them).  Needed for MAE (utils/metrics.py:156-160), which has no trace form.  Exact-fp32 MFMA on
 * 32x32 tiles, the m x n product is never materialised.  Xbits: m_pad x ldx words. */
int bmf_residual_sums(const uint32_t* Xbits, int64_t m_pad, int64_t ldx, int32_t m, int32_t n, const float* U,
                      const float* V, int kp, double* sums, const int32_t* stop, void* stream);

/* out[0] += sum_i W[i] (A[i] - B[i])^2 over n elements of dense fp64 arrays (W may be NULL = all ones): twice the reference's
 * rec_error(X_gt, X_pd, W) = 0.5 * sum(multiply(W, power(X_gt - X_pd, 2))) (models/BinaryMFPenalty.py:175-179) for a caller that
 * hands in an explicit prediction X_pd instead of factors (PNLPF's error() does, models/PNLPF.py:1,50-58).  Block partials added
 * in a fixed order.  work: bmf_sqdiff_work() doubles of scratch; out: device fp64, the caller zeroes it (calls accumulate, so a
 * large matrix can be fed in row chunks). */
int64_t bmf_sqdiff_work(void);
int bmf_sqdiff_sum(const double* A, const double* B, const double* W, int64_t n, double* work, double* out, void* stream);

/* Same sums for a real-valued fp32 X (m_pad x ldx floats, ldx % 32 == 0, zero padded): WNMF on non-Boolean data. */
int bmf_residual_sums_f32(const float* X, int64_t m_pad, int64_t ldx, int32_t m, int32_t n, const float* U, const float* V,
                          int kp, double* sums, void* stream);
/* V (rows_pad x kp fp32 row-major, rows_pad % 64 == 0) in the order the tiled residual pass consumes it:
 *   frag[(((st * 2 + cw) * (kp / 8) + q) * 64 + 32 h + r) * 4 + t] = V[(64 st + 32 cw + r) * kp + (kp / 2) h + 4 q + t]. */
int bmf_frag_rows_f32(const float* V, int64_t rows_pad, int kp, float* frag, void* stream);
/* The same sums over the tiled copy of the zero-padded X (bmf_tile_f32; m_pad, n_pad multiples of 64; U: m_pad x kp zero padded;
 * Vfrag: bmf_frag_rows_f32 of the zero-padded n_pad x kp V): sums[0] += sum |X - U V^T|, sums[1] += sum (X - U V^T)^2 -- ADDED to
 * what sums holds (the caller zeroes it). */
int bmf_residual_sums_f32_tiled(const float* Xtiled, int64_t m_pad, int64_t n_pad, const float* U, const float* Vfrag, int kp,
                                double* sums, void* stream);

/* ---- whole-iteration driver (BinaryMFPenalty._fit loop body, models/BinaryMFPenalty.py:81-115) ---------------- */

typedef struct {
    int32_t struct_bytes; /* sizeof(bmf_penalty_state), checked */
    int32_t m, n, k, kp, terms;
    int32_t mode;         /* BMF_MODE_PENALTY or BMF_MODE_WNMF */
    int32_t with_mae;     /* run bmf_residual_sums each iteration */
    int64_t m_pad, n_pad; /* multiples of BMF_ROW_PAD */
    const uint32_t* Xbits;  int64_t ldx;  /* m_pad x ldx words, ldx = n_pad/32 */
    const uint32_t* XTbits; int64_t ldxt; /* n_pad x ldxt words, ldxt = m_pad/32 */
    double* U64; double* V64;             /* m_pad x kp, n_pad x kp: fp64 master factors */
    float* U; float* V;                   /* fp32 shadows (inputs of the Gram / residual kernels) */
    uint16_t* Upanel; uint16_t* Vpanel;   /* [terms][kp][m_pad], [terms][kp][n_pad] */
    float* Mslab; int32_t splits_xv;  int32_t _pad0; /* X V   : [splits_xv ][m_pad][kp] */
    float* Nslab; int32_t splits_xtu; int32_t _pad1; /* X^T U : [splits_xtu][n_pad][kp] */
    float* Nred;                          /* [n_pad][kp]: X^T U summed over slabs (the fp32 all-reduce buffer) */
    float* gram_slabs; int32_t gram_blocks; int32_t _pad2;
    float* GU; float* GV;                 /* kp x kp fp32 */
    double* comm;                         /* fp64 all-reduce block: [0]=sum U o (XV) [1]=sum(U^2-U)^2 [2]=TP [3]=FP
                                             [4]=sum|X-UV^T| [5]=sum(X-UV^T)^2 [6..7] spare, [8 .. 8+kp*kp) = U^T U */
    double* GV64;                         /* kp*kp */
    double* partU; double* partV;         /* epilogue block partials: [m_pad/128][2], [n_pad/128][2] */
    double* scal;                         /* [8]: [0]=sum(V^2-V)^2 (replicated), [1]=previous reg_error, rest spare */
    uint64_t* ubits; uint32_t* ucolbits; int64_t lduc; /* [m_pad], [kp][m_pad/32] */
    uint64_t* vbits; uint32_t* vcolbits; int64_t ldvc; /* [n_pad], [kp][n_pad/32] */
    unsigned long long* counts;           /* [4] local TP, FP, spare */
    double* log;                          /* [log_rows][BMF_LOG_COLS] */
    int32_t log_rows; int32_t _pad3;
    int32_t* stop;                        /* device flag: 0 running, else the iteration that tripped early stop */
    double sum_x;                         /* sum(X) over ALL ranks */
    double cells;                         /* m_total * n */
    double tol, min_diff;                 /* early-stop parameters (models/BaseModelTools.py:326-334) */
    float thr_u, thr_v;                   /* 0.5 / 0.5 for BinaryMFPenalty */
    int32_t panel_kind;                   /* BMF_PANEL_BF16: `terms` bf16 addends, panels built inside the epilogue;
                                             BMF_PANEL_F16: two column-scaled fp16 addends (terms must be 2);
                                             BMF_PANEL_I8: `terms` (2 or 3) int8 digit planes, exact integer accumulation
                                             (Upanel / Vpanel then hold int8 [terms][kp][m_pad | n_pad]) */
    int32_t updates_only;                 /* non-zero: skip the Boolean cover count (and the MAE pass) -- the factor updates and the
                                             error terms only; TP / FP of the log rows are then 0 (bench "updates_only" leg) */
    float* scaleU; float* scaleV;         /* BMF_PANEL_F16: [2*kp] each, outputs of bmf_make_panel_f16.  BMF_PANEL_I8: [4*kp] each, zeroed by the
                                             caller before bmf_penalty_prepare: [0,kp) the scale predicted for the next epilogue's digit planes,
                                             [kp,2kp) the GEMM's colscale of the planes as they stand, [2kp,3kp) the exact scale of the current
                                             column maxima, [3kp, 4kp) per-column "prediction was off" flags (csrc/common.h) */
    float* panel_ws;                      /* max(m_pad, n_pad) / 128 * kp floats, BMF_PANEL_F16 only */
    uint16_t* mae_ws;                     /* optional, 2 * (m_pad + n_pad) * kp: with it the MAE pass runs on the bf16 MFMA
                                             (bmf_mae_sum); NULL = the exact-fp32 residual pass */
    const uint32_t* Xtiled;               /* optional (BMF_PANEL_I8): bmf_tile_bits copies of Xbits / XTbits for the int8 GEMM; NULL = it */
    const uint32_t* XTtiled;              /* reads the plain bit matrices */
    int32_t nred_blocks;                  /* 0 / 1: Nred is [n_pad][kp]; 2 (kp = 64, BMF_PANEL_I8): Nred is stored in 32-column blocks
                                             [2][n_pad][32] and X^T U can be computed block by block (bmf_penalty_update_xtu), so
                                             that the all-reduce of one block runs under the GEMM of the next */
    int32_t exchange_overlap;             /* row-sharded loop: the scalar part of a step under the numerator's all-reduce?  1 = no,
                                             2 = yes -- the caller's decision, which MUST be the same on every rank (the two forms issue
                                             different collectives: decide from rank-invariant quantities, e.g. the largest shard);
                                             0 = the library decides from THIS state (one rank, or equal shards) */
} bmf_penalty_state;

/* Build panels, bits, Grams, partial sums and X V, X^T U from the initial U, V (iteration-0 bookkeeping,
 * models/BinaryMFPenalty.py:68-75).  After it (and the caller's all-reduce of Nred / comm when sharded)
 * bmf_penalty_finalize(st, 0, reg0) writes log row 0. */
int bmf_penalty_prepare(const bmf_penalty_state* st, void* stream);

/* One multiplicative-update iteration with regulariser `reg`: V epilogue, X V, U epilogue, Grams, cover count,
 * X^T U of the NEW U (the numerator of the next V update, so that one exchange per iteration carries everything --
 * SURVEY section 8e), Grams, cover count.  Leaves local partial results in Nred / comm. */
int bmf_penalty_update(const bmf_penalty_state* st, double reg, void* stream);

/* The same iteration in phases, for the sharded loop: _head = V update, V^T V, X V, U update and the scalar part (U^T U, cover
 * counts, MAE sums, gather): afterwards the fp64 block `comm` is complete.  _xtu(block) = X^T U for column block `block` of
 * st->nred_blocks (or everything: block = -1) into Nred.  The caller starts the all-reduce of block b (block 0 together with
 * `comm`) right after enqueueing it, so that it runs under the GEMM of block b + 1.  With mode = BMF_MODE_PREPARE semantics:
 * bmf_penalty_prepare does both phases. */
int bmf_penalty_update_head(const bmf_penalty_state* st, double reg, void* stream);
int bmf_penalty_update_xtu(const bmf_penalty_state* st, int32_t block, void* stream);

/* Turn the (all-reduced) comm block into log row `iter`: error, rec_error (trace form), reg_error, RMSE, MAE,
 * TP/FP/FN/TN; evaluates the early-stop rule on the device and sets *stop (BaseModelTools.py:299-343). */
int bmf_penalty_finalize(const bmf_penalty_state* st, int32_t iter, double reg_used, int32_t max_iter, void* stream);

/* Single-GPU convenience: for it in [iter0, iter1): update(regs_host[it - iter0]); finalize(it). */
int bmf_penalty_run(const bmf_penalty_state* st, int32_t iter0, int32_t iter1, const double* regs_host,
                    int32_t max_iter, void* stream);

/* ---- the exchange of the row-sharded loop, issued from C (SURVEY 8b "bmf_allreduce", 8e) ------------------------------- */

/* One communicator per rank (= per process and GPU).  It owns the RCCL communicator, a side stream on which the collectives
 * run, and the events that fence that stream against the caller's compute stream.  RCCL is loaded at run time
 * (dlopen "librccl.so.1": inside a PyTorch process this resolves to the copy torch has loaded); a single-GPU caller never
 * loads it.  Not thread-safe per handle.  These calls create / destroy streams and events: not graph-capturable. */
typedef struct bmf_comm bmf_comm;
#define BMF_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */
#define BMF_COMM_RCCL 1
#define BMF_COMM_HOST 2
#define BMF_DTYPE_F32 0
#define BMF_DTYPE_F64 1
/* rank 0 makes the id (ncclGetUniqueId) and hands its 128 bytes to every rank by whatever channel the host framework has
 * (the drop-in classes broadcast it through the existing torch.distributed group), then every rank calls bmf_comm_create
 * with its own device current (ncclCommInitRank: collective, blocks until all ranks have called). */
int bmf_comm_available(void);   /* 1: RCCL loads in this process; 0: it does not (bmf_last_error).  No GPU, no peers: agree on it across
                                 * the ranks BEFORE bmf_comm_create, whose ncclCommInitRank is a collective */
int bmf_comm_unique_id(void* id_host);
int bmf_comm_create(const void* id_host, int32_t world, int32_t rank, bmf_comm** out);
/* A communicator whose all-reduce is a host function: fn(user, buf, count, dtype, stream) must sum the `count` elements at
 * device pointer `buf` over the ranks in place, ordered after the work already enqueued on `stream`, and be complete (or
 * stream-ordered on `stream`) when it returns; 0 = success.  For frameworks that bring their own transport, and for the
 * tests (two ranks on ONE GPU over gloo -- RCCL refuses that); the sequencing of the loop is the same code for both kinds. */
typedef int (*bmf_allreduce_fn)(void* user, void* buf, int64_t count, int32_t dtype, void* stream);
int bmf_comm_create_host(bmf_allreduce_fn fn, void* user, int32_t world, int32_t rank, bmf_comm** out);
int bmf_comm_destroy(bmf_comm* comm);
/* any out pointer may be NULL; rccl_version: ncclGetVersion() of the loaded library, 0 for BMF_COMM_HOST */
int bmf_comm_info(const bmf_comm* comm, int32_t* kind, int32_t* world, int32_t* rank, int32_t* rccl_version);

/* The fused exchange as one call: sum f32_buf[0..n32) and f64_buf[0..n64) over the ranks, in place, as ONE grouped RCCL launch
 * (ncclGroupStart / End) ordered on `stream`.  Either count may be 0.  Replaces nothing in the reference (PyBMF is
 * single-process); it is the collective of SURVEY 8e: X_p^T U_p (fp32) and [scalars, U_p^T U_p] (fp64). */
int bmf_allreduce(bmf_comm* comm, float* f32_buf, int64_t n32, double* f64_buf, int64_t n64, void* stream);

/* Row-sharded forms of bmf_penalty_prepare + finalize(0) and of bmf_penalty_run (models/BinaryMFPenalty.py:68-75, 81-115):
 * the whole loop is enqueued from C, collectives included, nothing returns to the host inside an iteration.  Per iteration:
 *     head (V update .. U update, scalar part)                                   compute stream
 *     nred_blocks <= 1:  X^T U; { all-reduce(Nred), all-reduce(comm block) } as ONE grouped launch on the compute stream itself
 *                        (nothing is left to hide it under; a cross-stream fence costs more than the scalars' all-reduce)
 *     nred_blocks == 2:  X^T U block 0; { all-reduce(Nred block 0), all-reduce(comm block) } grouped, on the side stream, under
 *                        X^T U block 1; all-reduce(Nred block 1); the compute stream waits for the side stream
 *     finalize (log row, stopping rule -- on all-reduced values, so every rank takes
 *     the same decision)
 * After the device-side stop flag is raised the kernels are no-ops but the collectives still run (every rank issues the same
 * sequence); callers enqueue a bounded number of iterations per call and look at the flag in between (engine.MUEngine.run). */
int bmf_penalty_prepare_sharded(const bmf_penalty_state* st, bmf_comm* comm, double reg0, int32_t max_iter, void* stream);
int bmf_penalty_run_sharded(const bmf_penalty_state* st, bmf_comm* comm, int32_t iter0, int32_t iter1, const double* regs_host,
                            int32_t max_iter, void* stream);
/* 1 when that loop enqueues the scalar part of a step (cover count, MAE sums, gather) BEHIND the X^T U GEMM, under the all-reduce of the
 * numerator on the communicator's side stream; 0 when everything stays in stream order (one rank, or a scalar part shorter than the
 * three stream crossings the overlap costs: csrc/api.hip::overlap_exchange; BMF_EXCHANGE_OVERLAP=0|1 overrides). */
int bmf_exchange_overlaps(const bmf_penalty_state* st, const bmf_comm* comm);
/* The shard-size rule of that decision (1 = overlap) for a row count the caller knows to be the same on every rank -- the padded
 * rows of the LARGEST shard; a caller with more than one rank evaluates it there and passes the answer in st->exchange_overlap. */
int bmf_exchange_overlap_rule(int with_mae, int64_t m_pad);

/* Event timing of the exchange inside bmf_penalty_run_sharded: bmf_comm_timing(comm, max_steps) starts recording (0 stops and
 * frees the events); bmf_comm_timing_read synchronises and returns, summed over the recorded steps, `exposed_ms` = what the
 * compute stream waited for the collectives after its own last kernel, and `span_ms` = from the start of the X^T U phase to
 * the end of the exchange.  Three timing events per step (~6 us of stream time each): for measurement legs only. */
int bmf_comm_timing(bmf_comm* comm, int32_t max_steps);
int bmf_comm_timing_read(bmf_comm* comm, int32_t* steps, double* exposed_ms, double* span_ms);

/* ---- whole-iteration driver for WNMF on a real-valued X (models/WNMF.py:51-109, 133-144; BASELINE config #2) ------------ */

typedef struct {
    int32_t struct_bytes; /* sizeof(bmf_wnmf_real_state), checked */
    int32_t m, n, k, kp;
    int32_t with_mae;     /* run the residual pass each iteration (the MAE column) */
    int64_t m_pad, n_pad; /* multiples of 128 */
    const float* X;       /* m_pad x n_pad fp32, zero padded */
    const float* XT;      /* n_pad x m_pad: the transposed copy */
    double* U64; double* V64;   /* fp64 master factors, m_pad x kp / n_pad x kp */
    float* U; float* V;         /* fp32 shadows */
    float* UT; float* VT;       /* kp x m_pad / kp x n_pad: transposed shadows (operands of bmf_xf_f32) */
    float* Mslab; int32_t splits_xv;  int32_t _pad0; /* X V   : [splits_xv ][m_pad][kp] */
    float* Nslab; int32_t splits_xtu; int32_t _pad1; /* X^T U : [splits_xtu][n_pad][kp] */
    float* gram_slabs; int32_t gram_blocks; int32_t _pad2;
    float* GU; float* GV; double* GU64; double* GV64;   /* kp x kp */
    double* partU; double* partV;                       /* epilogue partials, [m_pad/128][2], [n_pad/128][2] */
    uint64_t* rowbits; uint32_t* colbits; int64_t ldcb; /* scratch for the epilogue's Boolean by-products: max(m_pad, n_pad) words,
                                                           kp x ldcb words, ldcb >= max(m_pad, n_pad) / 32 */
    double* sums;         /* [4]: residual pass (sum |R|, sum R^2) */
    double* scal;         /* [8]: [1] = previous error */
    double* log;          /* [log_rows][BMF_LOG_COLS]: iter, error (= rec_error), RMSE, MAE, valid, stop */
    int32_t log_rows; int32_t _pad3;
    int32_t* stop;        /* device flag: 0 running, else the iteration that tripped the stopping rule */
    double sum_x2;        /* sum X^2 */
    double cells;         /* m * n */
    double tol, min_diff; /* models/BaseModelTools.py:326-334; WNMF watches the error */
    const float* Xtiled;  /* bmf_tile_f32 of X / of XT, or both NULL: the contractions and the residual pass then stream contiguous */
    const float* XTtiled; /* 16-KiB blocks instead of 256-byte row pieces; UT / VT then hold the factors in fragment order          */
    float* Vrf;           /* n_pad x kp floats: V in the row-fragment order of the tiled residual pass (needed with Xtiled + with_mae) */
    float* Urf;           /* optional, m_pad x kp words (bmf_frag_rows_bf16 of U, rebuilt every iteration): with it (and Xtiled, with_mae, kp == 32) the residual sums ride in the X^T U pass
                             (bmf_xf_f32_tiled_resid: X is read twice per iteration instead of three times); NULL = a pass of their own */
    uint32_t* UT3;        /* optional, both or neither, Xtiled and kp == 32 only: m_pad x 48 / n_pad x 48 words -- the factors' bf16 x 3 order          */
    uint32_t* VT3;        /* (bmf_frag_bf16x3), rebuilt by every update; with them X V and X^T U run on the bf16 matrix instruction (round 5) */
} bmf_wnmf_real_state;

/* Log row 0 and everything the first update needs (X^T U, U^T U) from the initial factors (WNMF.py:57-63). */
int bmf_wnmf_real_prepare(const bmf_wnmf_real_state* st, void* stream);
/* for it in [iter0, iter1): V update, U update (Gauss-Seidel), log row `it`, stopping rule on the device -- all enqueued on
 * `stream`, nothing returns to the host (the C-side loop SURVEY 8b calls bmf_wnmf_run). */
int bmf_wnmf_real_run(const bmf_wnmf_real_state* st, int32_t iter0, int32_t iter1, int32_t max_iter, void* stream);

/* ---- thresholding objective (models/BinaryMFThreshold.py:150-207) ------------------------------------------------ */

/* Thresholding objective and gradient in one tile-fused pass (the call zeroes out[0..3]):
 *   out[1] = sum (X - Us Vs^T)^2  with Us = sigmoid(lam (U - u)), Vs = sigmoid(lam (V - v))  =>  F(u,v) = 0.5 * out[1]
 *   out[0] = sum |X - Us Vs^T|
 *   if want_grad: out[2], out[3] = the reference's dF 2-vector (sum R o (dXdx(U,u) Vs^T), sum R o (Us dXdx(V,v)^T)).
 * work: (2*m_pad + 2*n_pad) * kp floats of scratch.  U: m_pad x kp, V: n_pad x kp (fp32, padded with zeros). */
int bmf_thresh_eval(const uint32_t* Xbits, int64_t m_pad, int64_t ldx, int32_t m, int32_t n, const float* U,
                    int64_t n_pad, const float* V, int k, int kp, double u, double v, double lamda, int want_grad,
                    float* work, double* out, void* stream);

/* The two halves of the thresholding objective as separate entry points, for the masked variant (W = 'mask' / weights):
 * bmf_thresh_transform: S = sigmoid(lam (F - x)), D = lam S (1 - S) (D may be NULL) for one factor;
 * bmf_masked_thresh: over the observed cells e = (i, j) of a segmented CSR list (see bmf_masked_pass), with
 *   p_e = <Us[i], Vs[j]>:  out[0] += sum (w_e (x_e - p_e))^2  (F = 0.5 out[0]),
 *   out[1] += sum w_e (x_e - p_e) <dUs[i], Vs[j]>,  out[2] += sum w_e (x_e - p_e) <Us[i], dVs[j]>   (dUs = dVs = NULL: F only).
 * out: 3 device doubles, zeroed by the caller. */
int bmf_thresh_transform(const float* F, int64_t rows_pad, int32_t rows, int k, int kp, double x, double lamda, float* S,
                         float* D, void* stream);
int bmf_masked_thresh(const int64_t* ptr, const int32_t* idx, const float* val, const float* wgt, const int32_t* seg_row,
                      const int64_t* seg_beg, int32_t nseg, const float* Us, const float* dUs, const float* Vs,
                      const float* dVs, int kp, double* out, void* stream);

/* The same objective in fp64 END TO END, for the line search: it compares F values that differ by min_diff = 1e-3 on F ~ 1e4,
 * and with fp32 factors on the device it took other branches than the reference after a few iterations.  U64 / V64: the fp64
 * factors (m_pad x kp, n_pad x kp, zero padded; m_pad, n_pad multiples of 64, ldx * 32 >= n_pad).  Transform, products and sums in
 * fp64, block partials added in a fixed order (deterministic).  out[0..3] as bmf_thresh_eval.  work: bmf_thresh_eval64_work()
 * doubles of scratch.  (PyBMF/models/BinaryMFThreshold.py:150-227) */
int64_t bmf_thresh_eval64_work(int64_t m_pad, int64_t n_pad, int kp);
int bmf_thresh_eval64(const uint32_t* Xbits, int64_t m_pad, int64_t ldx, int32_t m, int32_t n, const double* U64, int64_t n_pad,
                      const double* V64, int k, int kp, double u, double v, double lamda, int want_grad, double* work, double* out,
                      void* stream);

/* The same objective for the all-ones mask (W = 'full') in TRACE form, for a batch of (u, v) pairs in one enqueue
 * (csrc/thresh_trace.hip): F = 1/2 (sum X - 2 sum_{X_ij = 1} <Us_i, Vs_j> + <Us^T Us, Vs^T Vs>), the gradient likewise -- O(nnz k +
 * (m + n) k^2) per pair instead of m n k, and the candidate chain of one Wolfe search (PyBMF/solvers/line_search.py:30-62) in
 * one launch-and-wait.  idx: the column indices of the ones of X, row after row (int32, device), cut into nseg <= 8 m + 64 segments of
 * at most 128 cells of one row: seg_row (int32), seg_beg (int64 offsets into idx), seg_len (int32), longest first for speed;
 * U64 / V64: fp64 factors, leading dimension ldf >= k; uv_host: 2 n_pairs doubles (u0, v0, u1, v1, ...), host memory, read before
 * the call returns; n_pairs <= bmf_thresh_trace64_max_pairs(k) (32 for k <= 16, 16 for k <= 32, 8 above); sum_x: the number of ones;
 * work: bmf_thresh_trace64_work(m, n, k, max_pairs) doubles, device, zero-filled ONCE by the caller; out_host: 4 n_pairs + 1 doubles
 * of PINNED host memory: out[4 p + 1] = 2 F(u_p, v_p), out[4 p + 2], out[4 p + 3] = dF (want_grad), out[4 p] = seq written behind
 * those three, and out[4 n_pairs] = seq written last of all, each behind a system-scope fence: a host that polls instead of
 * synchronising the stream waits for ALL n_pairs + 1 stamps (pairs are written by different workgroups; the order in which their
 * writes reach host memory is not the order of the device-side fences).  fp64, fixed-order sums. */
int bmf_thresh_trace64_max_pairs(int k);
int64_t bmf_thresh_trace64_work(int32_t m, int32_t n, int k, int max_pairs);
int bmf_thresh_trace64(const int32_t* seg_row, const int64_t* seg_beg, const int32_t* seg_len, int32_t nseg, const int32_t* idx, int32_t m, int32_t n, const double* U64, const double* V64, int64_t ldf,
                       int k, const double* uv_host, int32_t n_pairs, double lamda, double sum_x, int want_grad, double* work,
                       double* out_host, double seq, void* stream);
/* fp64 forms of bmf_thresh_transform / bmf_masked_thresh (the masked objective, W = 'mask' / weights).  partial: 4 *
 * partial_blocks doubles of scratch (one row per workgroup, summed in order); out: 3 doubles (written, not accumulated). */
int bmf_thresh_transform64(const double* F, int64_t rows_pad, int32_t rows, int k, int kp, double x, double lamda, double* S,
                           double* D, void* stream);
int bmf_masked_thresh64(const int64_t* ptr, const int32_t* idx, const float* val, const float* wgt, const int32_t* seg_row,
                        const int64_t* seg_beg, int32_t nseg, const double* Us, const double* dUs, const double* Vs,
                        const double* dVs, int kp, double* partial, int32_t partial_blocks, double* out, void* stream);
/* The same, told how many of the kp columns are real (kcols; the padding columns of both transformed factors are zero): with kp = 32
 * a wave takes two (kcols <= 32) or four (kcols <= 16) cells per step instead of one. */
int bmf_masked_thresh64_k(const int64_t* ptr, const int32_t* idx, const float* val, const float* wgt, const int32_t* seg_row,
                          const int64_t* seg_beg, int32_t nseg, const double* Us, const double* dUs, const double* Vs, const double* dVs,
                          int kp, int kcols, double* partial, int32_t partial_blocks, double* out, void* stream);

/* ---- updates through an element-wise link (PNLPF, WNMF with the Kullback-Leibler loss) ---------------------------------- */

#define BMF_LINK_SIGMOID 1 /* PNLPF: prediction sigmoid(lamda (U V^T - 1/2)), models/PNLPF.py:54-58 */
#define BMF_LINK_KL 2      /* WNMF beta_loss='kullback-leibler', models/WNMF.py:111-129 */

/* One tile-fused pass for one factor (rows = rows of F_self; call it with X^T bits and (V, U) for the other one):
 *   BMF_LINK_SIGMOID: S = lamda (F_self F_other^T - 1/2), sig = sigmoid(S), d = sig (1 - sig)
 *     num = lamda (X o d) F_other      = link_lamda * multiply(W, multiply(X, d_sig)) @ V        models/PNLPF.py:65,81
 *     den = lamda (sig o d) F_other    = link_lamda * multiply(W, multiply(sig, d_sig)) @ V      models/PNLPF.py:68,84
 *   BMF_LINK_KL:      num = (X / (F_self F_other^T)) F_other = (WX / UV) @ V                     models/WNMF.py:117,125
 *     (den is not written: it is the column-sum vector of F_other, bmf_colsum_fill)
 * for the all-ones mask W.  The m x n intermediates are never materialised; exact-fp32 MFMA throughout.  The column range
 * is cut into `splits` = bmf_link_splits(rows, cols) slabs: num / den are [splits][rows_pad][kp] (stride slab_stride), to be
 * summed in slab order.  Xbits: rows_pad x ldx words; F_self: rows_pad x kp, F_other: other_pad x kp (fp32, zero padded). */
int bmf_link_splits(int64_t rows, int64_t cols);
int bmf_link_pass(const uint32_t* Xbits, int64_t rows_pad, int64_t ldx, int32_t rows, int32_t cols, const float* Fself,
                  const float* Fother, int64_t other_pad, int kp, int link, double lamda, float* num, float* den,
                  int64_t slab_stride, int splits, void* stream);

/* The same pass on the 16-bit MFMAs (16x the fp32-MFMA rate) with split operands.  The product P that goes through the link is built
 * from TWO fp16 addends per factor of the factor scaled by a power of two S (max |F| S in [2^14, 2^15): hi = f16(F S), lo = f16(F S - hi);
 * three products hi hi' + hi lo' + lo hi', right to 2^-22 relative to the factors' largest entries; round 4 -- until then three bf16
 * addends and six products), the linear contraction from two bf16 addends (2^-16 per product).
 * bmf_link_split makes the copies of one factor the pass needs in ws (5 * rows_pad * kp uint16, 16-byte aligned): [0] fp16 hi and [1] fp16
 * lo, row-major; [2] the first 16 bytes hold {float S, float 1 / S, uint32 bits of max |F|}; [3], [4] the bf16 hi / lo in the
 * reduction order of the contraction.  Call it for a factor whenever that factor changed.  bmf_link_pass16 = bmf_link_pass with the
 * factors given as workspaces. */
int bmf_link_split(const float* F, int64_t rows_pad, int kp, uint16_t* ws, void* stream);
/* The same for BOTH factors of a product at once, with one power-of-two scale per COLUMN pair instead of one per factor: S_k for column k of
 * A, T_k for column k of B, S_k T_k = C for every k (so the kernels divide by one constant), A's columns at full fp16 range, B's at a range
 * proportional to the column pair's largest possible contribution -- a small column of one factor next to a large one of the other keeps its
 * precision (what bmf_link_split loses).  This is what the engines and bmf_link_iterate call; re-run it whenever EITHER factor changed. */
int bmf_link_split_pair(const float* A, int64_t a_pad, const float* B, int64_t b_pad, int kp, uint16_t* wsA, uint16_t* wsB, void* stream);
int bmf_link_pass16(const uint32_t* Xbits, int64_t rows_pad, int64_t ldx, int32_t rows, int32_t cols, const uint16_t* ws_self,
                    const uint16_t* ws_other, int64_t other_pad, int kp, int link, double lamda, float* num, float* den,
                    int64_t slab_stride, int splits, void* stream);

/* bmf_link_sums with the factors given as bmf_link_split workspaces (P from the fp16 hi / lo splits, 2^-22). */
int bmf_link_sums16(const uint32_t* Xbits, int64_t m_pad, int64_t ldx, int32_t m, int32_t n, const uint16_t* wsU,
                    const uint16_t* wsV, int64_t n_pad, int kp, int link, double lamda, const uint32_t* Obits, double* sums, void* stream);

/* Scalars of the same model (caller zeroes sums[0..2], device fp64): with f = sigmoid(lamda (p - 1/2)) or f = p (KL),
 *   sums[0] += sum |x - f|, sums[1] += sum (x - f)^2     -> MAE / RMSE / rec_error = 0.5 sums[1]  (PNLPF via BinaryMFPenalty.py:175)
 *   sums[2] += sum (x log(x / p) - x + p), 0 log 0 = 0   -> the KL objective                      (WNMF.py:143-145)
 * Obits (may be NULL = every cell): bits of the observed cells, laid out like Xbits; restricts sums[2] to them (the W of
 * WNMF.error; RMSE / MAE are whole-matrix scores under task='reconstruction' whatever the mask). */
int bmf_link_sums(const uint32_t* Xbits, int64_t m_pad, int64_t ldx, int32_t m, int32_t n, const float* U, const float* V,
                  int64_t n_pad, int kp, int link, double lamda, const uint32_t* Obits, double* sums, void* stream);

/* colsum[c] = sum_i F[i][c] (fp64 accumulation, fixed order), out[r][c] = colsum[c] for r < out_rows: the KL denominator
 * O @ V of WNMF.py:118,126 as a rows x kp array for bmf_mu_epilogue's `den`. */
int bmf_colsum_fill(const float* F, int64_t rows, int kp, float* colsum, float* out, int64_t out_rows, void* stream);

/* The scalars of one masked iteration gathered by one launch: out[0] = sums[0]; out[1] = sum_b partU[2 b]; out[2] = sum_b partV[2 b];
 * out[3], out[4] = sums2[0], sums2[1]; out[5], out[6] = counts[0], counts[1]; out[7] = 0.  sums2 and counts (either may be NULL) are
 * reset to zero afterwards, ready for the next iteration's residual pass and cover count.  out: 8 doubles on the device (or pinned host
 * memory).  (error / rec_error / reg_error of BinaryMFPenalty.py:166-186 under a mask, and the scores of evaluate().) */
int bmf_masked_scalars(const double* sums, const double* partU, int nbU, const double* partV, int nbV, double* sums2,
                       unsigned long long* counts, double* out, void* stream);

/* ---- one whole iteration of the masked loop per call (round 4) -------------------------------------------------------------
 * BinaryMFPenalty / WNMF / PNLPF under W = 'mask' or a weight matrix (models/BinaryMFPenalty.py:81-115 with the contractions
 * over the observed cells): the previous iterate is kept (a loop that reads its scalars one iteration late returns it when the
 * stopping rule trips), then V epilogue, U-side pass, U epilogue, V-side pass (+ residual sums over the observed cells), the
 * whole-matrix sums, the cover count and the gather of the eight scalars of bmf_masked_scalars into `host_row` (pinned host memory;
 * word 7 is written last).  Everything is enqueued on `stream`; nothing returns to the host.  link: 0 or BMF_LINK_SIGMOID. */
typedef struct {
    const int64_t* ptr; const int32_t* idx; const float* val; const float* wgt; /* segmented CSR of one orientation (bmf_masked_pass) */
    const int32_t* seg_row; const int64_t* seg_beg; const int64_t* row_seg_ptr;
    float* part;          /* [max(nseg, 1)][2][kp] scratch */
    int32_t rows, nseg;
} bmf_masked_side;

typedef struct {
    int32_t struct_bytes; /* sizeof(bmf_masked_loop), checked */
    int32_t m, n, k, kp, link;
    double lamda;
    bmf_masked_side csr, csc;     /* rows of X for the U pass, columns for the V pass */
    bmf_epilogue_args epiU, epiV; /* as for bmf_mu_epilogue with num / den set; reg is taken from the call */
    double* sums;                 /* [4]: residual sums of the V-side pass (zeroed by the call) */
    double* Up64; double* Vp64;   /* the previous iterate (m_pad x kp, n_pad x kp fp64) */
    /* whole-matrix scores: on the bit matrix (Xbits != NULL), on a real-valued copy (Xreal != NULL), or none */
    const uint32_t* Xbits; int64_t x_m_pad, ldx, x_n_pad;
    const float* Xreal; int64_t r_m_pad, r_n_pad;
    double* sums2; unsigned long long* counts;   /* accumulators, reset by the gather */
    int32_t nbU, nbV;             /* blocks of epiU.partials / epiV.partials */
} bmf_masked_loop;

/* with_update = 0: the scalars of the current state only (log row 0). */
int bmf_masked_iterate(const bmf_masked_loop* st, double reg, int with_update, double* host_row, void* stream);

/* ---- one whole iteration of the link models' loop per call (round 4; PNLPF models/PNLPF.py:61-91 through BinaryMFPenalty._fit
 * :81-115, WNMF Kullback-Leibler models/WNMF.py:51-129; Boolean X, all-ones mask, the 16-bit MFMA flavour, one GPU) -------------
 * The previous iterate is kept, then the V side (tile-fused pass over X^T, its denominator -- the slab sum for the sigmoid link, the
 * column sums of U for KL --, the fp64 epilogue, the split of the new V), the U side likewise, the scalar pass (bmf_link_sums16),
 * the cover count, and one gather launch that writes eight doubles into `host_row` (pinned host memory; word 7 last):
 *   [0] the KL objective sum, [1], [2] the regulariser partials of U and V, [3] sum |x - f|, [4] sum (x - f)^2, [5] TP, [6] FP.
 * Everything is enqueued on `stream`; nothing returns to the host. */
typedef struct {
    int32_t struct_bytes; /* sizeof(bmf_link_loop), checked */
    int32_t m, n, k, kp, link;
    int32_t splitsU, splitsV;          /* bmf_link_splits(m, n), bmf_link_splits(n, m) */
    double lamda;
    const uint32_t* Xbits; const uint32_t* XTbits; int64_t m_pad, n_pad, ldx, ldxt;
    uint16_t* wsU; uint16_t* wsV;      /* bmf_link_split workspaces of U and V (kept current by the call) */
    float* numU; float* numV;          /* [splitsU][m_pad][kp], [splitsV][n_pad][kp] */
    float* denU_slabs; float* denV_slabs;   /* the same shapes (sigmoid link; NULL for KL) */
    float* colsum;                     /* [kp] scratch (KL) */
    bmf_epilogue_args epiU, epiV;      /* as for bmf_mu_epilogue with num / den / splits set; reg is taken from the call */
    double* Up64; double* Vp64;        /* the previous iterate */
    double* sums;                      /* [4] accumulators of bmf_link_sums16 */
    const uint32_t* Obits;             /* optional: observed cells of the KL objective (W = 'mask') */
    unsigned long long* counts;        /* [4] accumulators of the cover count */
    int32_t nbU, nbV;                  /* blocks of epiU.partials / epiV.partials */
} bmf_link_loop;

/* with_update = 0: the scalars of the current state only (log row 0). */
int bmf_link_iterate(const bmf_link_loop* st, double reg, int with_update, double* host_row, void* stream);

/* ---- proximal (PALM / iPALM) factor steps: ELBMF and PRIMP ------------------------------------- */

#define BMF_PALM_ELBMF 1     /* prox + clamp at 0                              models/ELBMF.py:199-210 */
#define BMF_PALM_PRIMP 2     /* proxelbmfnn (max 0) then _proxelbmfnn (min 1)  models/PRIMP.py:51-64,84-87 */
#define BMF_NORM_SPECTRAL 0  /* L = ||G^T G||_2   ELBMF.py:184 */
#define BMF_NORM_FROBENIUS 1 /* L = ||G^T G||_F   PRIMP.py:73 */

/* out[0] = spectral norm, out[1] = Frobenius norm of a symmetric positive semi-definite kp x kp matrix (fp64, row-major) --
 * the Lipschitz constants of the PALM step sizes, computed on the device so the step never visits the host
 * (np.linalg.norm(V.T @ V, ord=2), ELBMF.py:184; VVt.norm(), PRIMP.py:73).  One workgroup; repeated squaring + a Rayleigh
 * quotient, relative error < 1e-10. */
int bmf_sym_norms(const double* G64, int kp, double* out, void* stream);

typedef struct {
    double* F64;         /* rows_pad x kp factor being stepped (fp64 master), updated in place */
    double* Fprev64;     /* rows_pad x kp: the point the inertial term extrapolates from (U_{t-1}); see advance_prev */
    float* F;            /* out: fp32 shadow of the new factor */
    int64_t rows_pad;    /* multiple of 128 */
    int32_t rows, k, kp;
    int32_t splits;
    const float* num;    /* X G from bmf_xf_bits*: [splits][rows_pad][kp] slabs */
    int64_t slab_stride;
    const float* G;      /* kp x kp fp32 Gram G^T G of the OTHER factor */
    const double* norms; /* bmf_sym_norms output for that Gram */
    int32_t norm_kind;   /* BMF_NORM_* */
    int32_t variant;     /* BMF_PALM_* */
    double beta;         /* inertial coefficient in [0, 1); 0 = PALM */
    double l1, l2;       /* prox parameters BEFORE the step size: kai = l1 eta, lamda = l2 eta (ELBMF.py:193, PRIMP.py:84) */
    double gap_l1, gap_l2; /* weights of the integrality gap sum written to partials (ELBMF.py:166-174) */
    int32_t advance_prev; /* 1: Fprev64 <- the old F64 (ELBMF's returned U_last); 0: leave it (PRIMP's fixed anchor) */
    float thr;           /* threshold of the Boolean bits (strict >) */
    uint64_t* rowbits;   /* out, as in bmf_epilogue_args */
    uint32_t* colbits;
    int64_t ldcb;
    double* partials;    /* out: [rows_pad/128] integrality gap per 128-row block */
    float* blockmax;     /* optional out: [rows_pad/128][kp] column maxima per block (panel builders) */
    const int32_t* stop; /* optional device flag */
    const float* den;    /* optional, rows_pad x kp: (W o (Fe G^T)) G over the observed cells of a mask / weight matrix W, from
                            bmf_masked_pass run on Fe (bmf_palm_extrapolate); `num` is then (W o X) G from the same pass (one array,
                            splits = 1) and G is not used: gradient = den - num = multiply(W, Fe G^T - X) G  (ELBMF.py:190) */
    int8_t* planes;      /* optional: emit the int8 digit planes of the new factor here ([3][kp][ldp], the layout of bmf_make_panel_i8) with */
    int64_t ldp;         /*   the PREDICTED column scales plane_scale[kp] (2^e_c), as bmf_epilogue_args.planes does; needs beta = 0, no den, */
    const float* plane_scale; /* blockmax, rows_pad % 512 == 0.  The caller checks the prediction afterwards (bmf_palm_iterate does). */
    double* dotpart;     /* optional, with planes: [rows_pad/128] per-block sums of F_old o num = <F, X G> of the state BEFORE this step */
} bmf_palm_args;

/* One proximal gradient step of one factor:  Fe = F + beta (F - Fprev);  Fn = prox(Fe - eta (Fe G - num), l1 eta, l2 eta),
 * eta = 1 / (1.1 L) (beta = 0) or 2 (1 - beta) / (1 + 2 beta) / L, L = max(norm, 1e-4)
 * (update_U, models/ELBMF.py:177-196; elbmf_step_ipalm, models/PRIMP.py:71-88 with W = all ones, so that
 * multiply(W, U V^T - X) V = U (V^T V) - X V).  Element-wise part in fp64 on the fp64 master copy. */
int bmf_palm_epilogue(const bmf_palm_args* args, void* stream);

/* out[e] = (float)(F64[e] + beta (F64[e] - Fprev64[e])), e < n: the extrapolated point of an inertial step as the fp32 `Fself`
 * operand of bmf_masked_pass (U = U_last + beta (U_last - U_before_last), ELBMF.py:188). */
int bmf_palm_extrapolate(const double* F64, const double* Fprev64, double beta, int64_t n, float* out, void* stream);

/* partial[b] = sum over grid-strided elements of F64[e] * (sum_s slabs[s*stride + e]), b < blocks: <F, X G>, the cross term of
 * ||X - U V^T||_F^2 = sum X - 2 <U, X V> + <U^T U, V^T V>  (err of ELBMF.py:128, fn of PRIMP.py:118). */
int bmf_dot_slabs(const double* F64, const float* slabs, int64_t stride, int splits, int64_t n, double* partial, int blocks,
                  void* stream);
/* The scalars of one PALM iteration in one launch: out[0] = sum dotpart[0..nd), out[1] = <GU64, GV64> (kk entries), out[2] = sum partU,
 * out[3] = sum partV (the per-block integrality gaps of bmf_palm_epilogue), out[4], out[5] = counts[0], counts[1] of the cover count,
 * which are reset (counts may be NULL).  out: 6 doubles on the device. */
int bmf_palm_scalars(const double* dotpart, int nd, const double* GU64, const double* GV64, int kk, const double* partU, int nu,
                     const double* partV, int nv, unsigned long long* counts, double* out, void* stream);

/* One whole ELBMF iteration (the body of iPALM's loop, models/ELBMF.py:121-160, under the all-ones mask) enqueued by ONE call: the
 * two proximal steps from the state of the previous iteration (Jacobi: the V step sees the old U, :124-125), everything derived
 * from the new factors (digit planes, Grams, their norms, X^T U and X V on the int8 GEMM), and the scalars of the log row
 * (err :128, the two integrality gaps :131, the cover count behind the scores :144-155).  Nothing visits the host; the caller
 * reads row `it % log_rows` of `log` when it likes, so it can enqueue iteration t + 1 before it has seen the scalars of t (the
 * stopping rule :158-160 then lags one iteration: Up64 / Vp64 hold the factors of t once t + 1 has run).
 * Row layout (8 doubles): <U, X V>, <U^T U, V^T V>, U gap, V gap, TP, FP, 0, 0  (err = sum X - 2 row[0] + row[1]). */
typedef struct {
    int32_t struct_bytes; /* sizeof(bmf_palm_state): layout check */
    int32_t m, n, k, kp;
    int32_t variant;      /* BMF_PALM_ELBMF (PRIMP drives bmf_palm_epilogue itself: its anchor never advances) */
    int32_t norm_kind;    /* BMF_NORM_* */
    int32_t splits_xv, splits_xtu, gram_blocks, dot_blocks, log_rows;
    int64_t m_pad, n_pad; /* multiples of 256 */
    const uint32_t* Xbits; int64_t ldx;          /* m_pad x ldx words (cover count) */
    const uint32_t* Xtiled; const uint32_t* XTtiled; /* bmf_tile_bits copies of X and X^T (the int8 GEMM's bit operand) */
    double *U64, *V64, *Up64, *Vp64;             /* masters and previous iterates, m_pad x kp / n_pad x kp */
    float *U, *V;                                /* fp32 shadows */
    int8_t *Upanel, *Vpanel;                     /* [3][kp][m_pad], [3][kp][n_pad] digit planes */
    float *scaleU, *scaleV;                      /* 4 * kp floats each (bmf_make_panel_i8's scale / the predicted-scale block) */
    float *wsU, *wsV;                            /* (m_pad / 128) * kp and (n_pad / 128) * kp floats: per-block column maxima */
    float *Mslab, *Nslab;                        /* X V: [splits_xv][m_pad][kp]; X^T U: [splits_xtu][n_pad][kp] */
    float* gram_slabs;                           /* [gram_blocks][kp][kp] */
    float *GU, *GV; double *GU64, *GV64;         /* kp x kp Grams */
    double *normsU, *normsV;                     /* 2 doubles each (bmf_sym_norms) */
    double *partU, *partV;                       /* m_pad / 128, n_pad / 128: per-block integrality gaps */
    double* dotpart;                             /* dot_blocks doubles */
    uint64_t *ubits, *vbits; uint32_t *ucolbits, *vcolbits;   /* thresholded factors, as in bmf_penalty_state */
    unsigned long long* counts;                  /* 4 */
    double* log;                                 /* [log_rows][8]; device memory, or host memory the device can write (pinned) */
    double beta;
    float thr_u, thr_v;
} bmf_palm_state;
/* phase: 1 = head (the U step; with beta = 0 it also completes row it - 1: <U, X V> of the previous state falls out of the step
 * that consumes X V, so no separate pass computes it), 2 = tail (everything else), 3 = both.  bmf_palm_row_lag(st) = 1 when rows
 * complete one head late (beta = 0), else 0 (row `it` is complete when the tail of `it` is); bmf_palm_finish_row completes row `it`
 * when no further iteration will be enqueued (lag = 1 only; a no-op otherwise). */
int bmf_palm_iterate(const bmf_palm_state* st, int it, double l1, double l2, double gap_l1, double gap_l2, int phase, void* stream);
int bmf_palm_row_lag(const bmf_palm_state* st);
int bmf_palm_finish_row(const bmf_palm_state* st, int it, void* stream);
/* One iteration of PRIMP's loop (models/PRIMP.py:96-131) on a state with variant = BMF_PALM_PRIMP: U step, everything derived from the
 * new U (planes, Gram, Frobenius norm, X^T U), V step, everything derived from the new V (..., X V), and words 0 / 1 of log row
 * it % log_rows = <U, X V>, <U^T U, V^T V> of the new pair (objective = sum X - 2 word0 + word1).  Up64 / Vp64 are the anchors of the
 * inertial term and are never advanced (:96-110).  Everything is enqueued; nothing returns to the host. */
int bmf_primp_iterate(const bmf_palm_state* st, int it, double l1, double l2, void* stream);

/* ---- rank 64 < k <= 128: what couples the two 64-column blocks of a factor (csrc/wide.hip) --------------------------------------
 * A wider factor is held as two blocks F = [F_0 | F_1] of 64 columns each (every array of the k <= 64 path once per block); the
 * contractions, digit planes and the fp64 epilogue run per block, the epilogue with a precomputed denominator (`den`). */

/* out[rows_pad][64] (+)= F[rows_pad][64] . G[64][ldg] on the exact-fp32 MFMA: one term of the re-associated denominator
 * den_b = sum_b' F_b' G[b'][b]  (multiply(W, U V^T) V = U (V^T V), models/BinaryMFPenalty.py:142,157, WNMF.py:99,106).
 * rows_pad % 128 == 0; accumulate != 0 adds to `out`. */
int bmf_fg_f32(const float* F, int64_t rows_pad, const float* G, int ldg, float* out, int accumulate, void* stream);
/* slabs[b][64][64], b < blocks: partial sums of A^T B over row ranges (A, B: rows_pad x 64 fp32); summed by bmf_reduce_slabs.
 * A = B gives a diagonal block of the Gram matrix, A != B a cross block. */
int bmf_gram_cross(const float* A, const float* B, int64_t rows_pad, float* slabs, int blocks, void* stream);
/* counts[0] += TP, counts[1] += FP of the Boolean product over 128 factors against X (utils/common.py:110-151,
 * utils/metrics.py:56-68): rowbitsA / B = the two 64-bit words of factor bits per row, colbitsA / B = the bit-columns of the two
 * blocks ([64][ldcb] words each). */
int bmf_cover_count_wide(const uint32_t* Xbits, int64_t rows_pad, int64_t ldx, int64_t words, const uint64_t* rowbitsA,
                         const uint64_t* rowbitsB, const uint32_t* colbitsA, const uint32_t* colbitsB, int64_t ldcb,
                         unsigned long long* counts, void* stream);
/* sums[0] += sum |X - U V^T|, sums[1] += sum (X - U V^T)^2 over all cells, U = [UA | UB], V = [VA | VB] (fp32 shadows, rows_pad x 64
 * each, zero padded), one fp16 product per cell (utils/metrics.py:149-160).  XTbits: the transposed bit matrix, or its bmf_tile_bits
 * copy with x_tiled = 1.  ws: (m_pad + n_pad) * 128 uint16.  m_pad % 256 == 0, n_pad % 64 == 0. */
int bmf_resid_sums_wide(const uint32_t* XTbits, int64_t ldxt, int64_t m_pad, int64_t n_pad, const float* UA, const float* UB,
                        const float* VA, const float* VB, uint16_t* ws, double* sums, int x_tiled, void* stream);

/* ---- kernel timing (bench.py roofline leg) ----------------------------------------------------------------------- */

/* When enabled, bmf_xf_bits launches made through bmf_penalty_update are bracketed by hipEvents on `stream`
 * (created by bmf_timer_enable, which therefore must not be called during graph capture).  bmf_timer_read
 * synchronises the events and returns the number of timed launches and their total milliseconds. */
int bmf_timer_enable(int max_launches);
/* Bracket only every `every`-th launch from now on (default 1): keeps the instrumentation out of the measured rate. */
int bmf_timer_stride(int every);
int bmf_timer_read(int* launches, double* total_ms);
int bmf_timer_disable(void);

#ifdef __cplusplus
}
#endif
#endif /* BMF_HIP_H */
