// what do the packed fp8 -> f32 converts of gfx950 return for single-bit bytes?  (OCP vs FNUZ bias, denormals)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(float* out) {
    const int w = 0x40010140;  // bytes: 0x40, 0x01, 0x01, 0x40
    f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8(w, false);
    f32x2 b = __builtin_amdgcn_cvt_pk_f32_fp8(w, true);
    f32x2 c = __builtin_amdgcn_cvt_pk_f32_bf8(w, false);
    f32x2 d = __builtin_amdgcn_cvt_pk_f32_bf8(w, true);
    out[0] = a.x; out[1] = a.y; out[2] = b.x; out[3] = b.y;
    out[4] = c.x; out[5] = c.y; out[6] = d.x; out[7] = d.y;
}
int main() {
    float* d; float h[8];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(1), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) printf("%d %.10g\n", i, h[i]);
    return 0;
}
