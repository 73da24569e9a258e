// Stream-K plan shared by the two int8 bits-GEMM kernels (xf_bits_i8.hip: 64 rows x 32 columns per wave, two 4-wave workgroups
// per CU; xf_bits_i8w.hip: 32 rows x 64 columns per wave, one 8-wave workgroup per CU).
#pragma once
#include <stdint.h>

struct SlicePerm {
    uint16_t p[512];  // p[bslice] = logical stream-K slice of that workgroup (0xFFFF: none)
};

struct PlanI8 {
    int n_slices, grid, n_big, u_big, u_small, slots;
    int64_t total;
    SlicePerm perm;
};

// ncols = width of the column range one launch covers (32 or 64).  wide = 0: the 32-column kernel (two workgroups per CU, slices
// of two lengths); wide >= 1: a variant of the 64-column kernel (one workgroup per CU that owns whole rows, equal slices;
// variant 3 tiles the rows by 512; 4: the eight-wave sparse kernel, 512-row tiles, column halves kept).
PlanI8 make_plan_i8(int64_t rows_pad, int stages, int ncols, int wide);

// variant of the 64-column kernel a launch over `ncols` columns of a kp-wide factor runs on, or 0 for the 32-column kernel
// (bmf_xf_bits_i8_variant sets it)
int bmf_i8_use_wide(int ncols, int kp);

int bmf_xf_bits_i8p_launch(const uint32_t* A, int a_tiled, int stages, const int8_t* P, int64_t ldp, int limbs, float* out, int64_t slab_stride,
                           const PlanI8& pl, int slots, const float* colscale, const int32_t* stop, hipStream_t s);
int bmf_xf_bits_i8w_launch(int variant, const uint32_t* A, int64_t ldw, int a_tiled, int stages, const int8_t* P, int64_t ldp, int limbs, float* out,
                           int64_t slab_stride, const PlanI8& pl, int slots, const float* colscale, const int32_t* stop, hipStream_t s);
int bmf_xf_bits_i8w_occupancy(int variant, int limbs, int* out);
