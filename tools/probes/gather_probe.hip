// Probe: what does FETCH_SIZE (rocprofv3 --pmc) report for a 4-byte / 2-byte GATHER of known size?
// MI355X_MICROARCH.md calibrates the counter only for wide coalesced streams (reports exactly half the bytes).  The
// lookup kernels gather footprint rows of 10 cells (40 B fp32 / 20 B fp16) at a row stride of w cells; this kernel reads
// exactly that pattern from a buffer far larger than the Infinity Cache, once, so the bytes are known:
//   requested bytes = footprints * 100 * esize;   64-byte sectors touched = counted on the host for the same addresses.
// Run:  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -o g -- ./gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <set>
#include <vector>

template <typename T>
__global__ void gather(const T* vol, const int* x0s, const int* y0s, float* out, long long slice, int w, int n) {
    // one footprint (10 x 10 cells) per 100 consecutive lanes, as in corr_lookup_kernel's gather phase
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long fp = idx / 100;
    const int cell = (int)(idx % 100);
    if (fp >= n) return;
    const T v = vol[fp * slice + (long long)(y0s[fp] + cell / 10) * w + x0s[fp] + cell % 10];
    if ((float)v > 60000.0f) out[0] = 1.f;             // keep the load alive (representable in fp16: not provably false)
}

template <typename T>
void run(const char* name, int w, int h) {
    const int n = 1 << 20;                               // footprints; slices of w*h cells each: 1M x 7040 x esize = 14-28 GB
    const long long slice = (long long)w * h;
    T* vol; int *x0, *y0; float* out;
    hipMalloc(&vol, (size_t)n * slice * sizeof(T));
    hipMemset(vol, 0, (size_t)n * slice * sizeof(T));
    std::vector<int> hx(n), hy(n);
    long long sectors = 0;
    srand(1);
    for (int i = 0; i < n; ++i) {
        hx[i] = rand() % (w - 10); hy[i] = rand() % (h - 10);
        for (int r = 0; r < 10; ++r) {
            const long long b0 = ((long long)i * slice + (long long)(hy[i] + r) * w + hx[i]) * sizeof(T);
            const long long b1 = b0 + 10 * sizeof(T) - 1;
            sectors += b1 / 64 - b0 / 64 + 1;
        }
    }
    hipMalloc(&x0, n * 4); hipMalloc(&y0, n * 4); hipMalloc(&out, 4);
    hipMemcpy(x0, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(y0, hy.data(), n * 4, hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    const long long threads = (long long)n * 100;
    hipLaunchKernelGGL(gather<T>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, vol, x0, y0, out, slice, w, n);
    hipDeviceSynchronize();
    printf("%s: footprints %d, requested %.1f MB, 64-byte sectors touched %.1f MB (x%.2f), 128-byte lines %.1f MB\n", name, n,
           n * 100.0 * sizeof(T) / 1e6, sectors * 64.0 / 1e6, sectors * 64.0 / (n * 100.0 * sizeof(T)), 0.0);
    hipFree(vol); hipFree(x0); hipFree(y0); hipFree(out);
}

int main() {
    run<float>("fp32 footprints (40-byte rows, 512-byte row stride)", 128, 55);
    run<_Float16>("fp16 footprints (20-byte rows, 256-byte row stride)", 128, 55);
    return 0;
}
