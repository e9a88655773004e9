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

// exact (erf) GELU, as F.gelu / nn.GELU default
__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace sf

#define SF_REQUIRE(cond, ...)                                         \
    do {                                                              \
        if (!(cond)) return sf::fail(SF_ERR_BAD_ARG, __VA_ARGS__);    \
    } while (0)
