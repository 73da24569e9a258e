// Operand / result layout of v_mfma_f64_16x16x4_f64: A[i][k] = 100 i + k, B[k][j] = 1 if k == kk else 0 with j-dependent weight, so that
// D[i][j] identifies (i, j) -- prints which (i, j) each (lane, register) of D holds, for the assumed A / B operand layouts.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) double f64x4;
__global__ void probe(double* out) {
    const int lane = threadIdx.x, li = lane & 15, lq = lane >> 4;
    // assumed: A lane (i = li, k = lq); B lane (j = li, k = lq)
    const double a = (lq == 0) ? (double)(li + 1) : 0.0;          // A[i][0] = i + 1
    const double b = (lq == 0) ? (double)(1000 * (li + 1)) : 0.0;  // B[0][j] = 1000 (j + 1)
    f64x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}
int main() {
    double* d; hipMalloc(&d, 256 * 8);
    probe<<<1, 64>>>(d);
    double h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int lane = 0; lane < 64; lane += 5)
        for (int r = 0; r < 4; ++r) {
            const long v = (long)h[lane * 4 + r];   // = (i + 1) * 1000 * (j + 1)
            // decode: find i, j in 0..15 with (i+1)(j+1)*1000 == v -- ambiguous in general, so print candidates under the two hypotheses
            const int j_h = lane & 15, q = lane >> 4;
            const int i_a = 4 * q + r, i_b = 4 * r + q;
            printf("lane %2d reg %d: %8ld   hyp A (i = 4 q + r = %2d): %8d   hyp B (i = 4 r + q = %2d): %8d\n", lane, r, v, i_a, (i_a + 1) * 1000 * (j_h + 1), i_b,
                   (i_b + 1) * 1000 * (j_h + 1));
        }
    return 0;
}
