// Probe: which fp32 -> fp16 conversion instructions of gfx950 produce fp16 SUBNORMAL results, and which flush them to zero?
// Build: hipcc --offload-arch=gfx950 -O3 cvt_denorm.hip -o cvt_denorm ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

__global__ void probe(float x, float y, unsigned* out) {
    unsigned r0, r1, r2, r3, r4;
    asm volatile("v_cvt_f16_f32_e32 %0, %1" : "=v"(r0) : "v"(x));
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r1) : "v"(x), "v"(y));
    asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(r2) : "v"(x), "v"(y));
    r3 = 0;
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(r3) : "v"(x), "v"(1.0f));
    r4 = 0;
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,0]" : "+v"(r4) : "v"(x), "v"(2.0f), "v"(x));   // 2x - x
    if (threadIdx.x == 0) { out[0] = r0 & 0xffff; out[1] = r1; out[2] = r2; out[3] = r3 & 0xffff; out[4] = r4 & 0xffff; }
}

static float h2f(unsigned h) {
    const int e = (h >> 10) & 31, m = h & 1023;
    const float v = e ? ldexpf(1.f + m / 1024.f, e - 15) : ldexpf(m / 1024.f, -14);
    return (h & 0x8000) ? -v : v;
}

int main() {
    unsigned* d; hipMalloc(&d, 64);
    const float xs[] = {1.0f + ldexpf(1.f, -11), 1.0f + ldexpf(3.f, -11), 1.0f + ldexpf(1.f, -11) + ldexpf(1.f, -20), 1.0f + ldexpf(1.f, -11) - ldexpf(1.f, -20), -(1.0f + ldexpf(1.f, -11)), 1.0f, ldexpf(1.f, -14), ldexpf(1.f, -15), ldexpf(1.25f, -17), ldexpf(1.f, -20), ldexpf(1.f, -24)};
    for (float x : xs) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, x, 3.0f * x, d);
        unsigned h[5]; hipMemcpy(h, d, 20, hipMemcpyDeviceToHost);
        printf("x = %.9e: cvt_f16_f32 %.9e | cvt_pk_f16_f32 (x, 3x) %.9e %.9e | cvt_pkrtz %.9e %.9e | fma_mixlo(x*1) %.9e | fma_mixlo(2x - x) %.9e\n",
               x, h2f(h[0]), h2f(h[1] & 0xffff), h2f(h[1] >> 16), h2f(h[2] & 0xffff), h2f(h[2] >> 16), h2f(h[3]), h2f(h[4]));
    }
    return 0;
}
