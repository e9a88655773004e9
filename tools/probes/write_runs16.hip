// Probe: HBM write bandwidth of the correlation build's store pattern, 16-byte stores.
// A workgroup of 4 waves owns 128 source records; wave w, store instruction r writes 2 x 512 B (lanes 0-31 / 32-63).
//   pattern 0 (record-major, csrc/corr_blocked.hip today): the two runs go to records (r', r' + 4) of the wave's 32,
//             `rec` bytes apart; the next patch of the same records is written by another workgroup
//   pattern 1 (slab-major): the level-0 lines of one (block row, block group) of ALL records are contiguous:
//             record i at slab + i * 512 -> a workgroup writes 64 KB contiguous
// Build: hipcc --offload-arch=gfx950 -O3 write_runs16.hip -o write_runs16
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN, int NT>
__global__ __launch_bounds__(256) void wr(char* base, long long rec, int patches, long long nrec) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile = blockIdx.x / patches;          // group of 128 records
    const int patch = blockIdx.x % patches;
    const int khalf = lane >> 5, l31 = lane & 31;
    const u32x4 v = {0x3c003c00u, (unsigned)blockIdx.x, (unsigned)lane, 7u};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long long row = tile * 128 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        char* p = PATTERN == 0 ? base + row * rec + (long long)patch * 512 + l31 * 16
                               : base + ((long long)patch * nrec + row) * 512 + l31 * 16;
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
        else *reinterpret_cast<u32x4*>(p) = v;
    }
}

int main() {
    const long long nrec = 24LL * 7040, rec = 19712;     // 8 clips x 3 pairs x 7040 source pixels, Sintel record
    const int patches = 28;                               // 28 x 512 B = the 14336 level-0 bytes of a record
    char* buf;
    hipMalloc(&buf, nrec * rec + (1 << 20));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const unsigned blocks = (unsigned)(nrec / 128 * patches);
    for (int pat = 0; pat < 2; ++pat)
        for (int nt = 0; nt < 2; ++nt) {
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(a);
                if (pat == 0 && nt == 0) hipLaunchKernelGGL((wr<0, 0>), dim3(blocks), dim3(256), 0, 0, buf, rec, patches, nrec);
                if (pat == 0 && nt == 1) hipLaunchKernelGGL((wr<0, 1>), dim3(blocks), dim3(256), 0, 0, buf, rec, patches, nrec);
                if (pat == 1 && nt == 0) hipLaunchKernelGGL((wr<1, 0>), dim3(blocks), dim3(256), 0, 0, buf, rec, patches, nrec);
                if (pat == 1 && nt == 1) hipLaunchKernelGGL((wr<1, 1>), dim3(blocks), dim3(256), 0, 0, buf, rec, patches, nrec);
                hipEventRecord(b);
                hipEventSynchronize(b);
                hipEventElapsedTime(&ms, a, b);
            }
            const double bytes = (double)blocks * 128 * 512;
            printf("pattern %d (%s) nt=%d: %7.3f ms  %6.2f TB/s\n", pat, pat ? "slab-major, 64 KB contiguous per workgroup" : "record-major, 512 B runs", nt,
                   ms, bytes / ms / 1e9);
        }
    return 0;
}
