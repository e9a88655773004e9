// Shared host/device helpers for libstreamflow_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/streamflow_hip.h"

namespace sf {

// thread-local error string (sf_last_error)
char* err_buf();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SF_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return SF_OK;
}

// erf(x) in fp32 as x*P(x^2)/Q(x^2) on [-4,4] (the rational minimax fit used by Eigen/XLA's float erf):
// branch-free, 12 FMAs + one reciprocal; |error| <= 4.5e-7 absolute, the same class as libm erff's fp32
// rounding.  ocml's erff has two data-dependent branches (both taken in a wave of mixed magnitudes) and made
// the GELU epilogues VALU-bound.
__device__ __forceinline__ float erf_fast(float x) {
    x = fminf(fmaxf(x, -4.0f), 4.0f);
    const float x2 = x * x;
    float p = -2.72614225801306e-10f;
    p = fmaf(p, x2, 2.77068142495902e-08f);
    p = fmaf(p, x2, -2.10102402082508e-06f);
    p = fmaf(p, x2, -5.69250639462346e-05f);
    p = fmaf(p, x2, -7.34990630326855e-04f);
    p = fmaf(p, x2, -2.95459980854025e-03f);
    p = fmaf(p, x2, -1.60960333262415e-02f);
    float q = -1.45660718464996e-05f;
    q = fmaf(q, x2, -2.13374055278905e-04f);
    q = fmaf(q, x2, -1.68282697438203e-03f);
    q = fmaf(q, x2, -7.37332916720468e-03f);
    q = fmaf(q, x2, -1.42647390514189e-02f);
    return (x * p) * __builtin_amdgcn_rcpf(q);
}

// exact (erf-based) GELU, as F.gelu / nn.GELU default: 0.5 x (1 + erf(x / sqrt(2)))
// (x / 2 is clamped below where the erf argument saturates: 1 + erf_fast(-4) is 0 only to the fit's 4.5e-7, which times an
// unbounded |x| / 2 would be an error growing with |x|; clamped it stays <= 1.3e-6 for every x)
__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * fmaxf(x, -5.6568542f) * (1.0f + erf_fast(x * 0.70710678118654752440f));
}

// a * b rounded ONCE to fp32, opaque to the optimiser.  hipcc folds "x = a * b; h = (half)x" into v_fma_mixlo_f16 (h =
// RN16 of the EXACT product) while "x - (float)h" elsewhere uses RN16(RN32(a b)): two roundings of one value that differ
// when RN32(a b) sits on an fp16 tie.  Once in ~2^13 values the hi that is multiplied is then not the hi that lo was
// computed against and a hi + lo split is off by a whole fp16 ulp (measured: csrc/encoder.hip's three-product attention,
// 4.7e-5 on two queries in 1560 instead of 4e-7; `#pragma clang fp contract(off)` does not stop the fold).
__device__ __forceinline__ float mul_rn(float a, float b) {
    float r = a * b;
    asm volatile("" : "+v"(r));
    return r;
}

// Two GELUs at a time on packed fp32 (v_pk_fma_f32 / v_pk_mul_f32): the same operations in the same order as
// gelu_erf, i.e. bit-identical results at half the VALU issue slots.  The GEMM epilogues are VALU-bound on small-K
// shapes (M384 K256: 186 us with the GELU epilogue vs 141 us without), and a lane holds its outputs in pairs anyway.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float a) { f32x2 v; v[0] = a; v[1] = a; return v; }
__device__ __forceinline__ f32x2 erf_fast2(f32x2 x) {
    x = __builtin_elementwise_min(__builtin_elementwise_max(x, splat2(-4.0f)), splat2(4.0f));
    const f32x2 x2 = x * x;
    f32x2 p = splat2(-2.72614225801306e-10f);
    p = __builtin_elementwise_fma(p, x2, splat2(2.77068142495902e-08f));
    p = __builtin_elementwise_fma(p, x2, splat2(-2.10102402082508e-06f));
    p = __builtin_elementwise_fma(p, x2, splat2(-5.69250639462346e-05f));
    p = __builtin_elementwise_fma(p, x2, splat2(-7.34990630326855e-04f));
    p = __builtin_elementwise_fma(p, x2, splat2(-2.95459980854025e-03f));
    p = __builtin_elementwise_fma(p, x2, splat2(-1.60960333262415e-02f));
    f32x2 q = splat2(-1.45660718464996e-05f);
    q = __builtin_elementwise_fma(q, x2, splat2(-2.13374055278905e-04f));
    q = __builtin_elementwise_fma(q, x2, splat2(-1.68282697438203e-03f));
    q = __builtin_elementwise_fma(q, x2, splat2(-7.37332916720468e-03f));
    q = __builtin_elementwise_fma(q, x2, splat2(-1.42647390514189e-02f));
    f32x2 r;
    r[0] = __builtin_amdgcn_rcpf(q[0]);
    r[1] = __builtin_amdgcn_rcpf(q[1]);
    return (x * p) * r;
}
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    return splat2(0.5f) * __builtin_elementwise_max(x, splat2(-5.6568542f)) * (splat2(1.0f) + erf_fast2(x * splat2(0.70710678118654752440f)));
}

// GELU for results that leave as fp16 in the f16x2 / f16 arithmetic (kFast below: the k-octet GEMM epilogue, the fp16-row
// output of the depthwise kernel): a pure polynomial -- erf(z) ~ z Q(z^2) on |z| <= 3 (degree-8 minimax
// Q, |err| <= 2.2e-5, erf(3) = 1 - 2.2e-5 beyond), constants folded so that gelu(x) = h + h * (xc * D(xc^2)), h = x / 2,
// xc = x clamped to +-3 sqrt 2: 14 issue slots per PAIR (2 v_med3, 10 v_pk_fma / v_pk_mul, no v_rcp) against 28 for the
// rational form above.  |gelu error| <= 5.2e-5 absolute for EVERY x (h is clamped below at -3/sqrt 2, so that the residue of
// erf(3) != 1 stays 4.7e-5 instead of growing like 1.1e-5 |x|: ADVICE r2) and <= 1.1e-5 relative for x > 0 -- a tenth of the
// fp16 rounding (2^-11 relative) that every activation of these modes receives on its way into the next contraction.
__device__ __forceinline__ f32x2 gelu_poly2(f32x2 x) {
    f32x2 xc;
    xc[0] = __builtin_amdgcn_fmed3f(x[0], -4.2426405f, 4.2426405f);
    xc[1] = __builtin_amdgcn_fmed3f(x[1], -4.2426405f, 4.2426405f);
    const f32x2 t = xc * xc;
    f32x2 p = splat2(1.12535e-10f);
    p = __builtin_elementwise_fma(p, t, splat2(-1.074371e-08f));
    p = __builtin_elementwise_fma(p, t, splat2(4.5365834e-07f));
    p = __builtin_elementwise_fma(p, t, splat2(-1.12924145e-05f));
    p = __builtin_elementwise_fma(p, t, splat2(0.0001871811f));
    p = __builtin_elementwise_fma(p, t, splat2(-0.0022188f));
    p = __builtin_elementwise_fma(p, t, splat2(0.019636236f));
    p = __builtin_elementwise_fma(p, t, splat2(-0.13269384f));
    p = __builtin_elementwise_fma(p, t, splat2(0.79780626f));
    const f32x2 h = splat2(0.5f) * __builtin_elementwise_max(x, splat2(-4.2426405f));
    return __builtin_elementwise_fma(h, xc * p, h);
}
template <bool kFast>
__device__ __forceinline__ f32x2 gelu2(f32x2 x) {
    if constexpr (kFast) return gelu_poly2(x);
    else return gelu_erf2(x);
}

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// MI355X dispatches workgroup b of a 1-D grid to XCD b % 8 (8 XCDs with private 4 MiB L2s; observed, used for speed
// only).  Returns a work-item id such that every XCD walks ONE CONTIGUOUS range of ids (bijective for any grid size),
// so that consecutive ids -- which the caller orders to share operand tiles -- meet in the same L2.
__device__ __forceinline__ int xcd_linear_id(int b, int nwg) {
    constexpr int kXcd = 8;
    const int xcd = b % kXcd, local = b / kXcd;
    const int q = nwg / kXcd, r = nwg % kXcd;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

}  // namespace sf

#define SF_REQUIRE(cond, ...)                                         \
    do {                                                              \
        if (!(cond)) return sf::fail(SF_ERR_BAD_ARG, __VA_ARGS__);    \
    } while (0)
