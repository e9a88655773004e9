// Probe: does v_mfma_f32_32x32x16_f16 honour fp16 subnormal INPUTS, or flush them to zero?
// A = all `a`, B = all `b` (fp16), D = 16 * a * b expected in every element.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_denorm.hip -o mfma_denorm ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void probe(float a, float b, float* out) {
    f16x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (_Float16)a; bv[i] = (_Float16)b; }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)av[0]; out[2] = (float)bv[0]; }
}

int main() {
    float* d; hipMalloc(&d, 64);
    const float cases[][2] = {{1.0f, 1.0f}, {ldexpf(1.f, -14), 1.0f}, {ldexpf(1.f, -15), 1.0f}, {ldexpf(1.f, -20), 1.0f},
                              {ldexpf(1.f, -24), 1.0f}, {1.0f, ldexpf(1.f, -20)}, {ldexpf(1.5f, -16), 1024.0f},
                              {ldexpf(1.f, -10), ldexpf(1.f, -10)}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, c[0], c[1], d);
        float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("a = %.6e (as fp16 %.6e)  b = %.6e (as fp16 %.6e): mfma = %.9e  expected %.9e\n", c[0], h[1], c[1], h[2], h[0],
               16.0 * (double)h[1] * (double)h[2]);
    }
    return 0;
}
