// Probe: accumulation accuracy of v_mfma_f32_32x32x16_f16.  One 32x32x16 block, random fp16 operands (and a hi + lo split of
// fp32 operands: a_lo b_hi + a_hi b_lo + a_hi b_hi chained through the accumulator), against the float64 value.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_accuracy.hip -o mfma_accuracy ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// A [32][16], B [16][32] fp32 in global; out [32][32]
__global__ void k(const float* A, const float* B, float* out1, float* out3, float cinit) {
    const int lane = threadIdx.x, l31 = lane & 31, kh = lane >> 5;
    f16x8 ah, al, bh, bl;
    for (int i = 0; i < 8; ++i) {
        const float a = A[l31 * 16 + 8 * kh + i], b = B[(8 * kh + i) * 32 + l31];
        ah[i] = (_Float16)a; al[i] = (_Float16)(a - (float)ah[i]);
        bh[i] = (_Float16)b; bl[i] = (_Float16)(b - (float)bh[i]);
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = cinit;
    f32x16 c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
    f32x16 c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c3, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c3, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
        out1[row * 32 + l31] = c1[r];
        out3[row * 32 + l31] = c3[r];
    }
}

static float h(float x) { return (float)(_Float16)x; }

int main() {
    float A[512], B[512], *dA, *dB, *d1, *d3, o1[1024], o3[1024];
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&d1, 4096); hipMalloc(&d3, 4096);
    srand(1);
    for (float cinit : {0.0f, 100.0f}) {
        for (int i = 0; i < 512; ++i) { A[i] = (rand() / (float)RAND_MAX - 0.5f) * 4.f; B[i] = (rand() / (float)RAND_MAX - 0.5f) * 4.f; }
        hipMemcpy(dA, A, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B, 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, d1, d3, cinit);
        hipMemcpy(o1, d1, 4096, hipMemcpyDeviceToHost); hipMemcpy(o3, d3, 4096, hipMemcpyDeviceToHost);
        double e1 = 0, e3 = 0, e3s = 0, mag = 0;
        for (int m = 0; m < 32; ++m)
            for (int n = 0; n < 32; ++n) {
                double x1 = cinit, x3 = cinit, xs = cinit, sa = 0;
                for (int kk = 0; kk < 16; ++kk) {
                    const float a = A[m * 16 + kk], b = B[kk * 32 + n];
                    const double ahh = h(a), all = h(a - h(a)), bhh = h(b), bll = h(b - h(b));
                    x1 += ahh * bhh;                                  // what one product should give exactly
                    xs += ahh * bhh + all * bhh + ahh * bll;          // what the three products should give exactly
                    x3 += (double)a * b;                              // the fp32 operands' product
                    sa += fabs((double)a * b);
                }
                e1 = fmax(e1, fabs(o1[m * 32 + n] - x1)); e3s = fmax(e3s, fabs(o3[m * 32 + n] - xs)); e3 = fmax(e3, fabs(o3[m * 32 + n] - x3));
                mag = fmax(mag, sa);
            }
        printf("c = %g: sum|a b| up to %.2f | 1 product vs exact sum of fp16 products: %.3e | 3 products vs exact 3-product sum: %.3e | 3 products vs fp32 operands: %.3e\n",
               cinit, mag, e1, e3s, e3);
    }
    return 0;
}
