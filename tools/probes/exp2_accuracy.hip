// Probe: relative error of v_exp_f32 (__builtin_amdgcn_exp2f) and of the library exp2f over x in [-40, 0].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(float* a, float* b, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = -40.0f * i / n;
    a[i] = __builtin_amdgcn_exp2f(x);
    b[i] = exp2f(x);
}
int main() {
    const int n = 1 << 20;
    float *da, *db; hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, n);
    static float a[1 << 20], b[1 << 20];
    hipMemcpy(a, da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b, db, n * 4, hipMemcpyDeviceToHost);
    double ea = 0, eb = 0;
    for (int i = 0; i < n; ++i) {
        const float x = -40.0f * i / n;
        const double r = exp2((double)x);
        ea = fmax(ea, fabs(a[i] - r) / r); eb = fmax(eb, fabs(b[i] - r) / r);
    }
    printf("max relative error on [-40, 0]: v_exp_f32 %.3e   exp2f %.3e   (fp32 epsilon 5.96e-08)\n", ea, eb);
    return 0;
}
