#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(2))) float f32x2;
__global__ void k(const unsigned* in, float* out) {
    unsigned w = in[threadIdx.x];
    f32x2 a = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 0);
    f32x2 b = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 1);
    f32x2 c = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 2);
    f32x2 d = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 3);
    out[threadIdx.x * 8 + 0] = a[0]; out[threadIdx.x * 8 + 1] = a[1];
    out[threadIdx.x * 8 + 2] = b[0]; out[threadIdx.x * 8 + 3] = b[1];
    out[threadIdx.x * 8 + 4] = c[0]; out[threadIdx.x * 8 + 5] = c[1];
    out[threadIdx.x * 8 + 6] = d[0]; out[threadIdx.x * 8 + 7] = d[1];
}
int main() {
    unsigned h[64]; for (int i = 0; i < 64; ++i) h[i] = 0x22222222u & (0x11111111u * (i & 15)) * 2u | (i == 1 ? 0x76543210u : 0u);
    h[0] = 0x22222222u; h[1] = 0x76543210u; h[2] = 0xFEDCBA98u; h[3] = 0x20020020u;
    unsigned* di; float* dout; hipMalloc(&di, sizeof h); hipMalloc(&dout, 64 * 8 * 4);
    hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 64>>>(di, dout);
    float o[512]; hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; ++i) { printf("%08x:", h[i]); for (int j = 0; j < 8; ++j) printf(" %g", o[i * 8 + j]); printf("\n"); }
    return 0;
}
