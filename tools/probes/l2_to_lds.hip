// Probe: how fast can a CU pull L2-resident operand tiles, by path?
//   mode 0: buffer_load_dwordx4 ... lds  (LDS-DMA, 1 KB per wave-instruction)
//   mode 1: buffer_load_dwordx4 -> VGPR  (consumed by an XOR so it cannot be dropped)
//   mode 2: buffer_load_dwordx4 -> VGPR -> ds_write_b128 (register staging)
//   mode 3: buffer_load_dword ... lds    (LDS-DMA, 256 B per wave-instruction)
// Every workgroup of an XCD streams the same 2 MB window (L2-resident, far larger than the 32 KB L1), each wave from
// its own offset.  Prints bytes per clock per CU (clock from s_memtime vs wall) for 1..4 workgroups of 256 threads per CU.
// Build: hipcc --offload-arch=gfx950 -O3 l2_to_lds.hip -o l2_to_lds ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;
constexpr int kWin = 2 << 20;            // bytes per XCD window
constexpr int kUnroll = 8;               // loads in flight per wave between waits

template <int MODE>
__global__ __launch_bounds__(256) void pull(const char* base, int iters, unsigned* sink) {
    __shared__ __attribute__((aligned(1024))) char smem[4 * kUnroll * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x % 8;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base) + (size_t)xcd * kWin, 0, kWin, 0x00020000);
    // each wave walks the window in 1 KB (mode 3: 256 B) pieces from its own start
    unsigned pos = ((blockIdx.x / 8) * 4 + wave) * 37u * 1024u;
    u32x4 acc = {0u, 0u, 0u, 0u};
    char* sb = smem + wave * kUnroll * 1024;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int so = (int)((pos + u * 1024u) & (kWin - 1));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(sb + u * 1024), 16, lane * 16, so, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (MODE == 3) {
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int so = (int)((pos + u * 256u) & (kWin - 1));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(sb + u * 256), 4, lane * 4, so, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            u32x4 v[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int so = (int)((pos + u * 1024u) & (kWin - 1));
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, so, 0);
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                if (MODE == 2) *reinterpret_cast<u32x4*>(sb + u * 1024 + lane * 16) = v[u];
                else acc ^= v[u];
            }
        }
        pos += (MODE == 3 ? 256u : 1024u) * kUnroll;
    }
    if (MODE == 2 || MODE == 0 || MODE == 3) {
        __syncthreads();
        acc ^= *reinterpret_cast<u32x4*>(smem + tid * 16);
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

int main() {
    char* buf;
    unsigned* sink;
    hipMalloc(&buf, 8 * (size_t)kWin);
    hipMalloc(&sink, 64);
    hipMemset(buf, 1, 8 * (size_t)kWin);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    const char* names[4] = {"lds-dma x4 (1 KB)", "vgpr x4", "vgpr x4 + ds_write_b128", "lds-dma dword (256 B)"};
    for (int mode = 0; mode < 4; ++mode)
        for (int wg = 1; wg <= 4; ++wg) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                const dim3 grid(256 * wg);
                switch (mode) {
                    case 0: hipLaunchKernelGGL(pull<0>, grid, dim3(256), 0, 0, buf, iters, sink); break;
                    case 1: hipLaunchKernelGGL(pull<1>, grid, dim3(256), 0, 0, buf, iters, sink); break;
                    case 2: hipLaunchKernelGGL(pull<2>, grid, dim3(256), 0, 0, buf, iters, sink); break;
                    default: hipLaunchKernelGGL(pull<3>, grid, dim3(256), 0, 0, buf, iters, sink); break;
                }
                hipEventRecord(b);
                hipEventSynchronize(b);
                hipEventElapsedTime(&ms, a, b);
            }
            const double bytes = (double)256 * wg * 4 * iters * kUnroll * (mode == 3 ? 256.0 : 1024.0);
            printf("%-26s %d WG/CU (%2d waves): %7.3f ms  %6.2f TB/s  %6.1f GB/s per CU  (%.1f B/clk/CU at 2.1 GHz)\n", names[mode], wg, wg * 4,
                   ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.1);
        }
    return 0;
}
