/* Host-side AddressSanitizer run of the C-ABI's argument-checking layer (CPU only -- every call below must be
 * refused before anything touches a GPU).  Built and run by `make -C pybmf_amd/csrc asan`:
 *   hipcc -fsanitize=address -fno-gpu-sanitize ... -> build_asan/libbmf_hip_asan.so ; clang -fsanitize=address this file.
 * Exit code 0 = every bad call came back with a negative code and a message, and ASan saw no invalid access on the way. */
#include <stdio.h>
#include <string.h>
#include "../../include/bmf_hip.h"

static int fails = 0;
#define EXPECT_REFUSED(call)                                                      \
    do {                                                                          \
        int rc_ = (call);                                                         \
        const char* msg_ = bmf_last_error();                                      \
        if (rc_ >= 0 || !msg_ || !msg_[0]) {                                      \
            fprintf(stderr, "NOT refused (rc %d): %s\n", rc_, #call);            \
            ++fails;                                                              \
        }                                                                         \
    } while (0)

int main(void) {
    char junk[64];
    memset(junk, 0, sizeof junk);
    uint32_t* bits = (uint32_t*)junk;  /* a host pointer: only its non-NULL-ness / alignment may be looked at */
    float* f = (float*)junk;
    if (bmf_version() < 100) return 2;
    EXPECT_REFUSED(bmf_pack_rows_u8(NULL, 8, 8, 8, NULL, 2, NULL));
    EXPECT_REFUSED(bmf_popcount(NULL, 1, 1, 1, NULL, NULL));
    EXPECT_REFUSED(bmf_make_panel(NULL, 128, 32, 32, 3, NULL, 128, NULL));
    EXPECT_REFUSED(bmf_make_panel_f16(NULL, 128, 32, 32, NULL, 128, NULL, NULL, NULL));
    EXPECT_REFUSED(bmf_make_panel_i8(NULL, NULL, 512, 32, 32, 3, NULL, 512, NULL, NULL, NULL));
    EXPECT_REFUSED(bmf_xf_bits(NULL, 512, 4, 4, NULL, 128, 3, 64, NULL, 512 * 64, 1, NULL));
    EXPECT_REFUSED(bmf_xf_bits(bits, 500, 4, 4, (const uint16_t*)junk, 128, 3, 64, f, 512 * 64, 1, NULL));
    EXPECT_REFUSED(bmf_xf_bits_f16(bits, 512, 4, 4, (const uint16_t*)junk, 128, NULL, 64, f, 512 * 64, 1, NULL));
    EXPECT_REFUSED(bmf_xf_bits_i8(bits, 512, 16, 16, (const int8_t*)junk, 512, 4, f, 64, f, 512 * 64, 1, 0, NULL));
    EXPECT_REFUSED(bmf_xf_bits_i8(bits, 512, 16, 12, (const int8_t*)junk, 512, 3, f, 64, f, 512 * 64, 1, 0, NULL));
    EXPECT_REFUSED(bmf_xf_bits_i8_slots(500, 16, 64));
    EXPECT_REFUSED(bmf_xf_bits_slots(512, 4, 3, 48));
    EXPECT_REFUSED(bmf_tile_bits(bits, 100, 16, 16, bits, NULL));
    EXPECT_REFUSED(bmf_xf_f32(NULL, 128, 64, 64, NULL, 64, 32, NULL, 128 * 32, 1, NULL));
    EXPECT_REFUSED(bmf_gram_partial(NULL, 512, 64, 64, NULL, 4, NULL));
    EXPECT_REFUSED(bmf_reduce_slabs(NULL, 16, 1, 16, NULL, NULL, NULL));
    EXPECT_REFUSED(bmf_mu_epilogue(NULL, NULL));
    {
        bmf_epilogue_args a;
        memset(&a, 0, sizeof a);
        EXPECT_REFUSED(bmf_mu_epilogue(&a, NULL));
        bmf_palm_args p;
        memset(&p, 0, sizeof p);
        EXPECT_REFUSED(bmf_palm_epilogue(&p, NULL));
        bmf_penalty_state st;
        memset(&st, 0, sizeof st);
        EXPECT_REFUSED(bmf_penalty_prepare(&st, NULL));
        st.struct_bytes = (int32_t)sizeof st;
        EXPECT_REFUSED(bmf_penalty_update(&st, 1.0, NULL));
        EXPECT_REFUSED(bmf_penalty_update_head(&st, 1.0, NULL));
        EXPECT_REFUSED(bmf_penalty_update_xtu(&st, 0, NULL));
        EXPECT_REFUSED(bmf_penalty_finalize(&st, 0, 1.0, 10, NULL));
        double regs[2] = {1.0, 1.0};
        EXPECT_REFUSED(bmf_penalty_run(&st, 1, 3, regs, 10, NULL));
    }
    EXPECT_REFUSED(bmf_palm_epilogue(NULL, NULL));
    EXPECT_REFUSED(bmf_sym_norms(NULL, 64, NULL, NULL));
    EXPECT_REFUSED(bmf_dot_slabs(NULL, NULL, 16, 1, 16, NULL, 1, NULL));
    EXPECT_REFUSED(bmf_masked_pass(NULL, NULL, NULL, NULL, 1, NULL, NULL, 1, NULL, NULL, NULL, 32, NULL, NULL, NULL, NULL, NULL));
    EXPECT_REFUSED(bmf_masked_link_pass_k(NULL, NULL, NULL, NULL, 1, NULL, NULL, 1, NULL, NULL, NULL, 32, 16, NULL, NULL, NULL, NULL, 0, 0.0, NULL));
    EXPECT_REFUSED(bmf_masked_scalars(NULL, NULL, 1, NULL, 1, NULL, NULL, NULL, NULL));
    EXPECT_REFUSED(bmf_masked_counts(NULL, NULL, NULL, 1, NULL, NULL, NULL, NULL));
    EXPECT_REFUSED(bmf_mae_sum(NULL, 4, 256, 64, NULL, NULL, 32, NULL, NULL, NULL));
    EXPECT_REFUSED(bmf_mae_sum_ex(bits, 4, 257, 64, f, f, 32, (uint16_t*)junk, (double*)junk, 1, NULL));
    EXPECT_REFUSED(bmf_cover_count(NULL, 512, 4, 4, NULL, NULL, 4, 32, NULL, NULL, NULL));
    EXPECT_REFUSED(bmf_confusion_rows(NULL, 1, NULL, 1, 1, 1, NULL, NULL, NULL));
    EXPECT_REFUSED(bmf_boolean_product_bits(NULL, 1, NULL, 1, 32, 1, NULL, 1, NULL));
    EXPECT_REFUSED(bmf_real_product(NULL, 128, 1, NULL, 128, 1, 32, NULL, 1, NULL));
    EXPECT_REFUSED(bmf_residual_sums(NULL, 512, 4, 1, 1, NULL, NULL, 32, NULL, NULL, NULL));
    EXPECT_REFUSED(bmf_residual_sums_f32(NULL, 128, 32, 1, 1, NULL, NULL, 32, NULL, NULL));
    EXPECT_REFUSED(bmf_thresh_eval(NULL, 512, 4, 1, 1, NULL, 512, NULL, 1, 32, 0.5, 0.5, 10.0, 0, NULL, NULL, NULL));
    EXPECT_REFUSED(bmf_thresh_transform(NULL, 128, 1, 1, 32, 0.5, 10.0, NULL, NULL, NULL));
    EXPECT_REFUSED(bmf_masked_thresh(NULL, NULL, NULL, NULL, NULL, NULL, 1, NULL, NULL, NULL, NULL, 32, NULL, NULL));
    EXPECT_REFUSED(bmf_thresh_eval64(NULL, 512, 16, 1, 1, NULL, 512, NULL, 1, 32, 0.5, 0.5, 10.0, 0, NULL, NULL, NULL));
    EXPECT_REFUSED(bmf_thresh_transform64(NULL, 128, 1, 1, 32, 0.5, 10.0, NULL, NULL, NULL));
    EXPECT_REFUSED(bmf_masked_thresh64(NULL, NULL, NULL, NULL, NULL, NULL, 1, NULL, NULL, NULL, NULL, 32, NULL, 1, NULL, NULL));
    EXPECT_REFUSED(bmf_masked_thresh64_k(NULL, NULL, NULL, NULL, NULL, NULL, 1, NULL, NULL, NULL, NULL, 32, 16, NULL, 1, NULL, NULL));
    if (bmf_thresh_eval64_work(100, 64, 32) >= 0) ++fails;
    EXPECT_REFUSED(bmf_link_pass(NULL, 512, 4, 1, 1, NULL, NULL, 512, 32, 1, 10.0, NULL, NULL, 512 * 32, 1, NULL));
    EXPECT_REFUSED(bmf_link_split(NULL, 512, 32, NULL, NULL));
    EXPECT_REFUSED(bmf_link_pass16(NULL, 512, 4, 1, 1, NULL, NULL, 512, 32, 1, 10.0, NULL, NULL, 512 * 32, 1, NULL));
    EXPECT_REFUSED(bmf_link_sums(NULL, 512, 4, 1, 1, NULL, NULL, 512, 32, 1, 10.0, NULL, NULL, NULL));
    EXPECT_REFUSED(bmf_link_sums16(NULL, 512, 4, 1, 1, NULL, NULL, 512, 32, 1, 10.0, NULL, NULL, NULL));
    EXPECT_REFUSED(bmf_colsum_fill(NULL, 1, 32, NULL, NULL, 1, NULL));
    /* round 3 */
    EXPECT_REFUSED(bmf_mae_sum_tiled(bits, 4, 256, 64, f, f, 64, (uint16_t*)junk, (double*)junk, NULL));          /* n_pad % 256, ldxt % 16 */
    EXPECT_REFUSED(bmf_palm_scalars(NULL, 1, NULL, NULL, 1, NULL, 1, NULL, 1, NULL, NULL, NULL));
    {
        bmf_palm_state ps;
        memset(&ps, 0, sizeof ps);
        EXPECT_REFUSED(bmf_palm_iterate(NULL, 0, 0.0, 0.0, 0.0, 0.0, 3, NULL));
        EXPECT_REFUSED(bmf_palm_iterate(&ps, 0, 0.0, 0.0, 0.0, 0.0, 3, NULL));           /* struct_bytes = 0 */
        ps.struct_bytes = (int32_t)sizeof ps;
        ps.variant = BMF_PALM_ELBMF;
        EXPECT_REFUSED(bmf_palm_iterate(&ps, 0, 0.0, 0.0, 0.0, 0.0, 3, NULL));        /* null pointers */
        EXPECT_REFUSED(bmf_palm_finish_row(&ps, 0, NULL));
        EXPECT_REFUSED(bmf_palm_row_lag(NULL));
    }
    EXPECT_REFUSED(bmf_xf_f32_tiled_resid(f, 64, 64, f, (const uint32_t*)junk, f, 64, f, 64 * 64, 1, (double*)junk, NULL));   /* kp must be 32 */
    EXPECT_REFUSED(bmf_frag_rows_bf16(f, 64, 64, (uint32_t*)junk, NULL));
    EXPECT_REFUSED(bmf_fg_f32(NULL, 128, NULL, 64, NULL, 0, NULL));
    EXPECT_REFUSED(bmf_fg_f32(f, 100, f, 64, f, 0, NULL));
    EXPECT_REFUSED(bmf_gram_cross(NULL, NULL, 512, NULL, 4, NULL));
    EXPECT_REFUSED(bmf_cover_count_wide(NULL, 512, 4, 4, NULL, NULL, NULL, NULL, 4, NULL, NULL));
    EXPECT_REFUSED(bmf_resid_sums_wide(bits, 4, 100, 64, f, f, f, f, (uint16_t*)junk, (double*)junk, 0, NULL));
    EXPECT_REFUSED(bmf_comm_create(NULL, 2, 0, NULL));
    EXPECT_REFUSED(bmf_penalty_run_sharded(NULL, NULL, 1, 2, NULL, 10, NULL));
    EXPECT_REFUSED(bmf_timer_enable(0));
    EXPECT_REFUSED(bmf_timer_stride(0));
    EXPECT_REFUSED(bmf_timer_read(NULL, NULL));
    if (bmf_panel_pos(200) != -1 || bmf_panel_pos_i8(512) != -1) ++fails;
    printf("abi_asan_driver: %d entry point(s) not refused\n", fails);
    return fails ? 1 : 0;
}
