// Probe: HBM write bandwidth as a function of the contiguous run length a workgroup writes.
// Models the correlation-volume epilogue: a workgroup owns 128 "source rows" (stride = row_bytes apart, e.g. one
// 14 KB fp16 volume slice each) and writes, for `pieces` patch rows, a run of `run` bytes into each of them.
// Build: hipcc --offload-arch=gfx950 -O3 write_pattern.hip -o write_pattern ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int NT>
__global__ void wr(char* base, long long row_bytes, int run, int pieces, int piece_stride, int patches_per_row, long long total_rows) {
    // block -> (row group of 128, patch); thread lanes cover `run` bytes with 4-byte stores: lanes_per_run = run / 4
    const int patch = blockIdx.x % patches_per_row;
    const long long rg = blockIdx.x / patches_per_row;
    const int lanes_per_run = run / 4;
    const int runs_per_pass = 256 / lanes_per_run;
    const int l = threadIdx.x % lanes_per_run, r0 = threadIdx.x / lanes_per_run;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base + rg * 128 * row_bytes + (long long)patch * run, 0, 0x7ffffff0, 0x00020000);
    for (int piece = 0; piece < pieces; ++piece)
        for (int r = r0; r < 128; r += runs_per_pass) {
            const int off = (int)(r * row_bytes + (long long)piece * piece_stride + l * 4);
            __builtin_amdgcn_raw_buffer_store_b32(0x3c003c00u + r, rs, off, 0, NT);
        }
}

int main() {
    const long long row_bytes = 14080;          // one source pixel's fp16 level-0 slice (55 x 128 x 2 B)
    const long long rows = 24LL * 7040;         // 8 clips x 3 pairs x 7040 source pixels
    char* buf;
    hipMalloc(&buf, rows * row_bytes + (1 << 20));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    struct Cfg { const char* name; int run, pieces, piece_stride; };
    // total bytes per row = row_bytes in every configuration: run * pieces * patches = 14080
    std::vector<Cfg> cfgs = {
        {"64 B runs x 8 rows (8x32 fp16 patch)", 64, 8, 256},
        {"128 B runs x 8 rows (8x32 fp32-like)", 128, 8, 256},
        {"256 B runs x 2 rows (2x128 fp16 patch, rows adjacent = 512 B)", 256, 2, 256},
        {"512 B contiguous x 1", 512, 1, 0},
        {"1024 B contiguous x 1", 1024, 1, 0},
    };
    for (auto& c : cfgs) {
        const int per_patch = c.run * c.pieces;
        const int patches = (int)(row_bytes / per_patch);
        const long long blocks = (rows / 128) * patches;
        for (int nt = 0; nt < 2; ++nt) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                if (nt) hipLaunchKernelGGL(wr<2>, dim3((unsigned)blocks), dim3(256), 0, 0, buf, row_bytes, c.run, c.pieces, c.piece_stride, patches, rows);
                else hipLaunchKernelGGL(wr<0>, dim3((unsigned)blocks), dim3(256), 0, 0, buf, row_bytes, c.run, c.pieces, c.piece_stride, patches, rows);
                hipEventRecord(b);
                hipEventSynchronize(b);
            }
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double bytes = (double)(rows / 128) * 128 * patches * per_patch;
            printf("%-62s nt=%d  %7.3f ms  %6.2f TB/s\n", c.name, nt * 2, ms, bytes / ms / 1e9);
        }
    }
    return 0;
}
