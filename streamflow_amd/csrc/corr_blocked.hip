// Correlation pyramids in BLOCKED fp16 layout: build (K1 + K2) and radius-4 lookup (K3).
//
// Reference: core/corr.py:7-21,46-54 (volume + avg-pool pyramid), :23-44 (lookup), core/utils/utils.py:65-79.
//
// WHY A SECOND LAYOUT.  In the row-major volumes of corr.hip ([source pixel][y][x], the reference's own layout) a
// 10 x 10 bilinear footprint is ten 20-byte rows, each in its own 128-byte line: the lookup fetched 4x its algorithmic
// bytes and was bound by exactly that, and the build wrote the volume in 64-byte runs
// of 2- and 4-byte stores.  Here every pyramid level of every source pixel is stored as 8 x 8-cell BLOCKS of 128 bytes
// (= one cache line), cells inside a block column-major:
//
//     record(source pixel i) = [level 0 blocks | level 1 blocks | level 2 blocks | level 3 blocks]       (rec bytes)
//     level l: ceil(hl / 8) x ceil(wl / 8) blocks, block (by, bx) at off[l] + (by * nbx[l] + bx) * 128
//     cell (ty, tx) of the level at block (ty / 8, tx / 8), byte ((tx % 8) * 8 + ty % 8) * 2
//
//  * a footprint touches 2.125 x 2.125 = 4.5 lines per level on average instead of ~11 (levels 0 / 1);
//  * a 16-byte piece is one block COLUMN = eight vertically adjacent cells: in the build a lane of the MFMA C/D layout
//    holds exactly those (8 patch rows of one target column), so level 0 leaves as ONE 16-byte store per lane and
//    accumulator register -- 1 KB per instruction in 512-byte runs, 32 store instructions per wave instead of 108;
//    in the lookup one lane owns one footprint column: the vertical lerp is in-lane, the horizontal one a DPP shift,
//    and the nine results of a lane are nine CONSECUTIVE output channels (channel = l*81 + a*9 + b, corr.py:31-37);
//  * cells of a block that lie outside the level (padding) have UNSPECIFIED contents; the lookup masks them.
//
// The lookup's product is the fp16 k-octet image of the 324 correlation features (SF_LAYOUT_F16_KOCT), i.e. the LDS
// image of the first GEMM of the correlation encoder; fp32 planes are optional (API parity / tests).
#include "sf_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;

constexpr int kThreads = 256;
constexpr int kDrop = (int)0x80000000u;          // buffer offset past any num_records: the access is dropped / reads 0

struct VolGeom {
    int hl[4], wl[4], nby[4], nbx[4], off[4];    // level sizes, blocks per level, byte offset of the level in a record
    int rec;                                     // bytes per source pixel
};

VolGeom make_geom(int h, int w) {
    VolGeom g;
    int o = 0;
    for (int l = 0; l < 4; ++l) {
        g.hl[l] = h >> l; g.wl[l] = w >> l;
        g.nby[l] = (g.hl[l] + 7) / 8; g.nbx[l] = (g.wl[l] + 7) / 8;
        g.off[l] = o;
        o += g.nby[l] * g.nbx[l] * 128;
    }
    g.rec = o;
    return g;
}

inline int src_rows_padded(int N) { return (N + 127) / 128 * 128; }

__device__ __forceinline__ unsigned pack_h2(float a, float b) {
    f16x2 v;
    v[0] = (_Float16)a;
    v[1] = (_Float16)b;
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float dpp_xor1(float v) {     // lane ^ 1 (quad_perm [1,0,3,2])
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float v) {     // lane ^ 2 (quad_perm [2,3,0,1])
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_shl4(float v) {     // lane + 4 inside a row of 16 (row_shl:4)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x104, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_shl1(float v) {     // lane + 1 inside a row of 16 (row_shl:1)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xF, 0xF, true));
}

// ------------------------------------------------------------------------------------------------
// build
// ------------------------------------------------------------------------------------------------
constexpr int PR = 8, PC = 32;    // target patch: 8 rows x 32 columns = one block row of four blocks
constexpr int BN = PR * PC;       // 256 target cells
constexpr int KD = 256;           // feature depth the resident-patch kernel is built for (the StreamFlow encoders: 256)
#ifndef SF_CORRB_NW
#define SF_CORRB_NW 8
#endif
constexpr int NW = SF_CORRB_NW;   // waves per workgroup: 8 = 2 per SIMD with a 256-register budget, 12 = 3 per SIMD with 168
constexpr int kBuildThreads = NW * 64;
constexpr int kPatchBytes = (KD / 8) * BN * 16;          // 128 KB: the whole patch, all k, resident in LDS
constexpr int kTilesPerItem = 60;                         // 32-source tiles per work item (~5 per wave)

struct BuildArgs {
    const char* ws;               // packed features: image (img, side) = [Dp / 8][Np][8] halves, pixel N.. = zeros
    char* vol;
    int64_t vol_img_stride;       // bytes
    int n_img, h, w, N, Np;
    VolGeom g;
    int pcols, np;                // patch columns, patches per image
    int ntiles, nchunks, tpc;     // 32-source tiles per image, chunks per image, tiles per chunk
    float scale;
#ifdef SF_CORR_TIMERS
    long long* ts;
#endif
};

// features fp32 [D][N] -> fp16 k-octet planes [(k / 8)][Np][8]; pixels N .. Np-1 are zero (the target of every patch
// cell that lies outside the image: its products are exactly 0)
__global__ __launch_bounds__(256) void pack_f16z_kernel(const float* f1, const float* f2, int64_t f_clip_stride,
                                                        int64_t f_pair_stride, char* ws, int pairs, int D, int Dp, int N,
                                                        int Np) {
    const int px = blockIdx.x * 256 + threadIdx.x, kq = blockIdx.y;
    const int side = blockIdx.z & 1, img = blockIdx.z >> 1;             // img = b * pairs + pair
    if (px >= Np) return;
    const float* f = (side ? f2 : f1) + (int64_t)(img / pairs) * f_clip_stride + (int64_t)(img % pairs) * f_pair_stride;
    f16x8 hv;
#pragma unroll
    for (int i = 0; i < 8; ++i) hv[i] = (_Float16)((px < N && kq * 8 + i < D) ? f[(int64_t)(kq * 8 + i) * N + px] : 0.f);
    const int64_t plane = (int64_t)(Dp / 8) * Np * 16;
    *reinterpret_cast<f16x8*>(ws + (int64_t)blockIdx.z * plane + ((int64_t)kq * Np + px) * 16) = hv;
}

#ifndef SF_CORRB_NT
#define SF_CORRB_NT 2            // cache policy of the level-0 stores (2 = non-temporal)
#endif

// BUILD, resident-patch form.  Work item = (image, chunk of source tiles, target patch); one workgroup of 12 waves per
// item and CU.  The patch's features for ALL 256 k (128 KB) are DMA'd into LDS once; after ONE barrier every wave is on
// its own: it draws 32-source tiles from a workgroup counter, streams that tile's A fragments straight from L2 into
// registers (a lane's fragment is one 16-byte k-octet of one pixel: no LDS, no DMA, ring of four in flight), reads B
// fragments from the resident patch, and runs its epilogue while the other two waves of its SIMD keep the matrix pipe
// busy.  No barrier, no DMA and no shared staging buffer inside the loop.
// (The first blocked kernel kept round 2's structure -- 128 x 256 tile per workgroup, both operands staged through a
// 2- or 3-stage LDS ring -- and measured 27k cycles of k-loop per tile of which 4.1k were MFMA issue: every wave sat
// ~190 cycles in each of its 48 LDS-DMA instructions and ~0.5k cycles per stage at the barrier; interleaving the DMA
// instructions with the MFMAs moved that time, it did not remove it.  tools/corrb_bench.py)
__global__ __launch_bounds__(kBuildThreads, NW / 4) void corr_build_blocked_kernel(const BuildArgs g) {
    __shared__ __attribute__((aligned(1024))) char smem[kPatchBytes + 16];
#ifdef SF_CORR_TIMERS
    const long long ts0 = __builtin_readcyclecounter();
    const long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, l31_ = lane & 31;
    // item order [image][chunk][patch]: the workgroups that run side by side on an XCD share the chunk's source features
    const int item = sf::xcd_linear_id(blockIdx.x, gridDim.x);
    const int patch = item % g.np, chunk = (item / g.np) % g.nchunks, img = item / (g.np * g.nchunks);
    const int by0 = patch / g.pcols, pxb = patch % g.pcols;
    const int py0 = by0 * PR, px0 = pxb * PC;
    const int plane = (KD / 8) * g.Np * 16;                       // bytes of one packed image (< 2 GiB, host-checked)
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(g.ws) + (int64_t)(img * 2 + 0) * plane, 0, plane, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(g.ws) + (int64_t)(img * 2 + 1) * plane, 0, plane, 0x00020000);
    const int kq_step = g.Np * 16;
    int* const counter = reinterpret_cast<int*>(smem + kPatchBytes);
    if (tid == 0) *counter = 0;
    // ---- the patch: LDS slot = k-octet * 256 + cell (cell = patch row * 32 + patch column), 128 pieces of 1 KB; piece p
    // covers k-octet p / 4, cells (p % 4) * 64 + lane.  Wave w takes pieces w, w + 12, ...: always cell group w % 4.
    // Cells outside the image read the zero pixel. ----
    {
        const int cell = (wave & 3) * 64 + lane;
        const int ty = py0 + cell / PC, tx = px0 + cell % PC;
        const int vob = ((ty < g.h && tx < g.w) ? ty * g.w + tx : g.N) * 16;
        static_assert(NW % 4 == 0, "a wave keeps its cell group");
        for (int p = wave; p < (KD / 8) * 4; p += NW)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(smem + p * 1024), 16, vob, (p >> 2) * kq_step, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef SF_CORR_TIMERS
    const long long ts1 = __builtin_readcyclecounter();
    long long t_k = 0, t_e = 0;
    int n_t = 0;
#endif

    const int rec = g.g.rec;
    const int t_begin = chunk * g.tpc, t_count = min(g.tpc, g.ntiles - t_begin);
    char* const img_base = g.vol + (int64_t)img * g.vol_img_stride;
    const char* const sbB = smem + (khalf * BN + l31_) * 16;       // + (2 ks * 256 + t * 32) * 16

    for (;;) {
        int tile = 0;
        if (lane == 0) tile = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        tile = __builtin_amdgcn_readfirstlane(tile);
        if (tile >= t_count) break;
#ifdef SF_CORR_TIMERS
        const long long tt0 = __builtin_readcyclecounter();
#endif
        const int i0 = (t_begin + tile) * 32;                     // first source pixel of the tile
        // A fragment of k-step ks: k-octet 2 ks + khalf of source pixel i0 + l31 (pixels past N clamped: padding records)
        const int voa = (khalf * g.Np + min(i0 + l31_, g.N - 1)) * 16;
        f32x16 acc[PR];
#pragma unroll
        for (int t = 0; t < PR; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        constexpr int kSteps = KD / 16, kRing = (NW <= 8) ? 4 : 2;
        u32x4 af[kRing];
#pragma unroll
        for (int s = 0; s < kRing; ++s) af[s] = __builtin_amdgcn_raw_buffer_load_b128(ra, voa, s * 2 * kq_step, 0);
#if defined(SF_CORRB_ABLATE) && SF_CORRB_ABLATE == 2      // timing ablation: four k-steps only
        constexpr int kRun = 4;
#else
        constexpr int kRun = kSteps;
#endif
#pragma unroll
        for (int ks = 0; ks < kRun; ++ks) {
            const f16x8 a = __builtin_bit_cast(f16x8, af[ks % kRing]);
#pragma unroll
            for (int t = 0; t < PR; ++t) {
                const f16x8 bv = *reinterpret_cast<const f16x8*>(sbB + (ks * 2 * BN + t * PC) * 16);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bv, acc[t], 0, 0, 0);
            }
            if (ks + kRing < kRun) af[ks % kRing] = __builtin_amdgcn_raw_buffer_load_b128(ra, voa, (ks + kRing) * 2 * kq_step, 0);
        }
        // Issue order of the whole (fully unrolled) k-loop, pinned: the four A loads of the ring first, then B fragment
        // reads running kAhead MFMAs ahead of their use, one A refill behind every eighth MFMA.  Left to itself hipcc sinks
        // every load next to its use (vmcnt(0) / lgkmcnt(0) in front of each MFMA: 28k cycles per tile for 4.1k of MFMA).
        {
            constexpr int kAhead = (NW <= 8) ? 8 : 4, kMfma = kRun * PR;
            __builtin_amdgcn_sched_group_barrier(0x020, kRing, 0);               // VMEM reads
            __builtin_amdgcn_sched_group_barrier(0x100, kAhead, 0);              // DS reads
#pragma unroll
            for (int i = 0; i < kMfma; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               // one MFMA
                if (i + kAhead < kMfma) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (i % PR == PR - 1 && i / PR + kRing < kRun) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
#ifdef SF_CORR_TIMERS
        const long long tt1 = __builtin_readcyclecounter();
#endif
        // ---- epilogue ----
        // ---- per-lane store offsets, recomputed per tile from a laundered lane id: as loop invariants they would sit in
            // ~10 VGPRs through the k-loop, which needs every register it can get for B fragments in flight ----
            int l31 = l31_;
            asm volatile("" : "+v"(l31));
            const int khalf = lane >> 5;
        // C/D layout: lane = (target column l31, k-half), register r = source row (r&3) + 8(r>>2) + 4 khalf, acc[t] = patch row
        // t.  A lane's eight values of one register are one block column of level 0.
        const int rowh = 4 * khalf * rec;
        // level 0: four consecutive blocks of block row by0 -> 512 contiguous bytes per source row
        const int vo0_ = ((px0 >> 3) + (l31 >> 3) < g.g.nbx[0])
                             ? rowh + g.g.off[0] + (by0 * g.g.nbx[0] + (px0 >> 3)) * 128 + l31 * 16 : kDrop;
        // level 1: lane pair (2j, 2j+1) holds cell column j of the patch's 4 x 16 level-1 cells: rows 4 (by0 & 1) .. +3 of
        // block row by0 >> 1, an 8-byte piece; lane parity k1 stores source row (2 jp + k1) of a register pair
        const int j1 = l31 >> 1, k1 = l31 & 1;
        const int by1 = by0 >> 1, bx1 = 2 * pxb + (j1 >> 3);
        const int vo1_ = (by1 < g.g.nby[1] && bx1 < g.g.nbx[1])
                             ? rowh + k1 * rec + g.g.off[1] + (by1 * g.g.nbx[1] + bx1) * 128 + (j1 & 7) * 16 + (by0 & 1) * 8 : kDrop;
        // level 2: lane quad = cell column l31 >> 2 of 2 x 8 cells: rows 2 (by0 & 3) .. +1 of block row by0 >> 2 (4 bytes);
        // lane k2 of the quad stores source row k2 of a register group
        const int j2c = l31 >> 2, k2 = l31 & 3;
        const int by2 = by0 >> 2;
        const int vo2_ = (by2 < g.g.nby[2] && pxb < g.g.nbx[2])
                             ? rowh + k2 * rec + g.g.off[2] + (by2 * g.g.nbx[2] + pxb) * 128 + j2c * 16 + (by0 & 3) * 4 : kDrop;
        // level 3: lanes 0..3 of an octet hold cell column 4 pxb + (l31 >> 3), row by0 & 7 of block row by0 >> 3 (2 bytes)
        const int k3 = l31 & 7, tx3 = 4 * pxb + (l31 >> 3);
        const int by3 = by0 >> 3;
        const int vo3_ = (k3 < 4 && by3 < g.g.nby[3] && (tx3 >> 3) < g.g.nbx[3])
                             ? rowh + k3 * rec + g.g.off[3] + (by3 * g.g.nbx[3] + (tx3 >> 3)) * 128 + (tx3 & 7) * 16 + (by0 & 7) * 2 : kDrop;
        constexpr int kNt = SF_CORRB_NT;
#if defined(SF_CORRB_ABLATE) && SF_CORRB_ABLATE == 1      // timing ablation: no stores leave the CU
        const int vo0 = kDrop | (vo0_ & 0), vo1 = kDrop | (vo1_ & 0), vo2 = kDrop | (vo2_ & 0), vo3 = kDrop | (vo3_ & 0);
#else
        const int vo0 = vo0_, vo1 = vo1_, vo2 = vo2_, vo3 = vo3_;
#endif
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(img_base + (int64_t)i0 * rec, 0, 32 * rec, 0x00020000);
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {                       // register group: source rows 8 rq + 4 khalf + (0..3)
            float sel2[2] = {0.f, 0.f}, sel3 = 0.f;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {                   // register pair (2 jp, 2 jp + 1) of the group
                float sel1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int ri = 2 * jp + u, r = 4 * rq + ri;
                    float v0[PR];
#pragma unroll
                    for (int t = 0; t < PR; ++t) v0[t] = acc[t][r] * g.scale;
                    u32x4 o;
#pragma unroll
                    for (int t = 0; t < 4; ++t) o[t] = pack_h2(v0[2 * t], v0[2 * t + 1]);
                    __builtin_amdgcn_raw_buffer_store_b128(o, rv, vo0, (ri + 8 * rq) * rec, kNt);
                    float v1[4], v2[2];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float sm = v0[2 * t] + v0[2 * t + 1];
                        v1[t] = 0.25f * (sm + dpp_xor1(sm));
                        sel1[t] = (k1 == u) ? v1[t] : sel1[t];
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const float sm = v1[2 * t] + v1[2 * t + 1];
                        v2[t] = 0.25f * (sm + dpp_xor2(sm));
                        sel2[t] = (k2 == ri) ? v2[t] : sel2[t];
                    }
                    const float sm = v2[0] + v2[1];
                    const float v3 = 0.25f * (sm + dpp_shl4(sm));
                    sel3 = (k3 == ri) ? v3 : sel3;
                }
                u32x2 o1;
                o1[0] = pack_h2(sel1[0], sel1[1]);
                o1[1] = pack_h2(sel1[2], sel1[3]);
                __builtin_amdgcn_raw_buffer_store_b64(o1, rv, vo1, (2 * jp + 8 * rq) * rec, 0);
            }
            __builtin_amdgcn_raw_buffer_store_b32(pack_h2(sel2[0], sel2[1]), rv, vo2, (8 * rq) * rec, 0);
            const _Float16 h3 = (_Float16)sel3;
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h3), rv, vo3, (8 * rq) * rec, 0);
        }
#ifdef SF_CORR_TIMERS
        t_k += tt1 - tt0; t_e += __builtin_readcyclecounter() - tt1; ++n_t;
#endif
    }
#ifdef SF_CORR_TIMERS
    if (g.ts && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long* d = g.ts + ((int64_t)blockIdx.x * NW + wave) * 8;
        d[0] = ts0; d[1] = ts1; d[2] = __builtin_readcyclecounter(); d[3] = rt0; d[4] = __builtin_amdgcn_s_memrealtime();
        d[5] = t_k; d[6] = t_e; d[7] = n_t;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// lookup
// ------------------------------------------------------------------------------------------------
constexpr int LP = 32;                          // source pixels per workgroup
constexpr int NCH = 324, NOCT = 41;
constexpr int TROW = NOCT * 8;                  // halves per pixel in the transpose buffer (656 bytes: 16-byte aligned rows,
                                                // 164 dwords = 36 mod 64: the 16-byte read-back of 16 pixels is conflict-free)
#ifndef SF_LOOKB_WAVES
#define SF_LOOKB_WAVES 4                        // waves per SIMD the register budget is sized for
#endif

struct LookArgs {
    const char* vol;
    int64_t vol_img_stride;       // bytes
    const float* coords;
    float* out;                   // optional fp32 planes [324][N] per image
    int64_t out_img_stride;
    _Float16* out16;              // optional k-octet planes [41][N][8] per image
    int64_t out16_img_stride;     // halves
    int h, w, N;
    VolGeom g;
};

struct Foot {                     // one lane's share of one footprint: a column of 18 cells (two block pieces + 2 cells)
    u32x4 w0, w1;
    unsigned w2;
};

__global__ __launch_bounds__(kThreads, SF_LOOKB_WAVES) void corr_lookup_blocked_kernel(const LookArgs a) {
    __shared__ __attribute__((aligned(16))) _Float16 T[LP * TROW];
    const int tid = threadIdx.x;
    const int grp = tid >> 4, c = tid & 15;               // 16 lanes per footprint: lane c = footprint column c (10 used)
    const int img = blockIdx.y, p0 = blockIdx.x * LP;
    const int rec = a.g.rec;
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(a.vol) + (int64_t)img * a.vol_img_stride + (int64_t)p0 * rec, 0, LP * rec, 0x00020000);
    // item (iteration it) = (level it >> 1, pixel (it & 1) * 16 + grp): a thread only ever sees two pixels
    float cxs[2], cys[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int p = min(p0 + e * 16 + grp, a.N - 1);
        cxs[e] = a.coords[((int64_t)img * 2 + 0) * a.N + p];
        cys[e] = a.coords[((int64_t)img * 2 + 1) * a.N + p];
    }
    if (tid < LP) *reinterpret_cast<u32x2*>(T + tid * TROW + NCH) = u32x2{0u, 0u};       // rows 324..327 of the last octet

    // level geometry in SGPRs up front (left to itself hipcc sinks each kernarg load into a branch around the
    // address select of the piece that uses it)
    int g_wl[4], g_hl[4], g_nby[4], g_rowb[4], g_off[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        g_wl[l] = a.g.wl[l]; g_hl[l] = a.g.hl[l]; g_nby[l] = a.g.nby[l]; g_rowb[l] = a.g.nbx[l] * 128; g_off[l] = a.g.off[l];
        asm volatile("" : "+s"(g_wl[l]), "+s"(g_hl[l]), "+s"(g_nby[l]), "+s"(g_rowb[l]), "+s"(g_off[l]));
    }
    struct Item { int x0, y0; float fx, fy; };
    auto locate = [&](int it) {
        const int l = it >> 1, e = it & 1;
        const float inv = 1.0f / (float)(1 << l);
        float cx = cxs[e] * inv, cy = cys[e] * inv;
        // anything this far out samples only zero padding; also swallows NaN/inf
        if (!(cx > -1.0e6f && cx < 1.0e6f)) cx = -1.0e6f;
        if (!(cy > -1.0e6f && cy < 1.0e6f)) cy = -1.0e6f;
        const float fx0 = floorf(cx), fy0 = floorf(cy);
        Item q;
        q.x0 = (int)fx0; q.y0 = (int)fy0; q.fx = cx - fx0; q.fy = cy - fy0;
        return q;
    };
    auto fetch = [&](int it, const Item& q) {
        const int l = it >> 1, pix = (it & 1) * 16 + grp;
        const int tx = q.x0 - 4 + c, ys = q.y0 - 4;
        const bool col_ok = (c < 10) & ((unsigned)tx < (unsigned)g_wl[l]);        // (bitwise: no short-circuit branches)
        const int byf = ys >> 3;
        const int col = pix * rec + g_off[l] + tx * 16;          // block bx = tx / 8, column tx % 8: (bx * 8 + tx % 8) * 16
        auto piece = [&](int k) {
            const int by = byf + k;
            return (col_ok & ((unsigned)by < (unsigned)g_nby[l])) ? col + by * g_rowb[l] : kDrop;
        };
        Foot f;
        f.w0 = __builtin_amdgcn_raw_buffer_load_b128(rv, piece(0), 0, 0);
        f.w1 = __builtin_amdgcn_raw_buffer_load_b128(rv, piece(1), 0, 0);
        f.w2 = __builtin_amdgcn_raw_buffer_load_b32(rv, piece(2), 0, 0);      // rows 16, 17: only needed when ys % 8 == 7
        return f;
    };

    Item qn = locate(0);
    Foot fn = fetch(0, qn);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const Item q = qn;
        const Foot f = fn;
        if (it + 1 < 8) {
            qn = locate(it + 1);
            fn = fetch(it + 1, qn);
        }
        const int l = it >> 1, pix = (it & 1) * 16 + grp;
        const int ys = q.y0 - 4, s = ys & 7;
        // ---- rows ys .. ys+9 out of the 18 loaded ones (first loaded row = 8 (ys >> 3)): shift by s halves ----
        unsigned W[9];
#pragma unroll
        for (int e = 0; e < 4; ++e) { W[e] = f.w0[e]; W[4 + e] = f.w1[e]; }
        W[8] = f.w2;
        // (bit-select masks, v_bfi_b32: written as `cond ? W[i + 2] : W[i]` hipcc turns the chain into a dynamically
        // indexed array in SCRATCH memory)
        unsigned W1[7], W2[6], d[5];
        const unsigned m2 = 0u - ((unsigned)(s >> 2) & 1u), m1 = 0u - ((unsigned)(s >> 1) & 1u);
#pragma unroll
        for (int i = 0; i < 7; ++i) W1[i] = (W[i + 2] & m2) | (W[i] & ~m2);
#pragma unroll
        for (int i = 0; i < 6; ++i) W2[i] = (W1[i + 1] & m1) | (W1[i] & ~m1);
        const unsigned sh = (s & 1) * 16;
#pragma unroll
        for (int i = 0; i < 5; ++i) d[i] = __builtin_amdgcn_alignbit(W2[i + 1], W2[i], sh);
        // ---- padding rows inside the last block row (hl % 8 != 0) hold unspecified data: clear rows >= hl ----
        const int nvalid = g_hl[l] - ys;                 // rows b < nvalid are inside the level (b >= -ys handled by the dropped loads)
        if (__builtin_amdgcn_ballot_w64(nvalid < 10) != 0) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const unsigned m = (2 * i + 1 < nvalid) ? 0xFFFFFFFFu : ((2 * i < nvalid) ? 0x0000FFFFu : 0u);
                d[i] &= m;
            }
        }
        float F[10];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const f16x2 hv = __builtin_bit_cast(f16x2, d[i]);
            F[2 * i] = (float)hv[0];
            F[2 * i + 1] = (float)hv[1];
        }
        // ---- vertical lerp in-lane, horizontal lerp against the next column (lane + 1) ----
        const float wy1 = q.fy, wy0 = 1.f - q.fy, wx1 = q.fx, wx0 = 1.f - q.fx;
        float R[9];
#pragma unroll
        for (int b = 0; b < 9; ++b) {
            const float v = F[b] * wy0 + F[b + 1] * wy1;
            R[b] = v * wx0 + dpp_shl1(v) * wx1;
        }
        const int ch0 = l * 81 + c * 9;                    // channel of R[0]: l*81 + a*9 + b with a = c (corr.py:31-37)
        const bool live = (c < 9) & (p0 + pix < a.N);
        if (a.out != nullptr && live) {                    // optional fp32 planes (API parity / tests; scattered stores)
            float* o = a.out + (int64_t)img * a.out_img_stride + (int64_t)ch0 * a.N + p0 + pix;
#pragma unroll
            for (int b = 0; b < 9; ++b) o[(int64_t)b * a.N] = R[b];
        }
        // ---- nine consecutive channels of one pixel -> T[pix][ch0 ..]: four dword writes + one half, by parity ----
        if (c < 9) {
            const bool odd = (ch0 & 1) != 0;
            _Float16* t0 = T + pix * TROW + ch0;
            const float single = odd ? R[0] : R[8];
            *(t0 + (odd ? 0 : 8)) = (_Float16)single;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float lo = odd ? R[2 * k + 1] : R[2 * k], hi = odd ? R[2 * k + 2] : R[2 * k + 1];
                *reinterpret_cast<unsigned*>(t0 + (odd ? 1 : 0) + 2 * k) = pack_h2(lo, hi);
            }
        }
    }
    if (a.out16 == nullptr) return;                        // workgroup-uniform
    __syncthreads();
    // ---- (octet, pixel) = 16 bytes: 32 consecutive pixels of an octet = 512 contiguous bytes ----
    _Float16* o16 = a.out16 + (int64_t)img * a.out16_img_stride;
    for (int i = tid; i < NOCT * LP; i += kThreads) {
        const int pix = i % LP, oc = i / LP;
        if (p0 + pix >= a.N) continue;
        const u32x4 v = *reinterpret_cast<const u32x4*>(T + pix * TROW + oc * 8);
        *reinterpret_cast<u32x4*>(o16 + ((int64_t)oc * a.N + p0 + pix) * 8) = v;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" int sf_corr_blocked_geometry(int h, int w, int64_t* rec_bytes, int64_t* lvl_off, int32_t* nby, int32_t* nbx,
                                        int64_t* src_rows) {
    SF_REQUIRE(h > 0 && w > 0, "sf_corr_blocked_geometry: bad dims");
    SF_REQUIRE((h >> 3) >= 1 && (w >> 3) >= 1, "sf_corr_blocked_geometry: feature grid %dx%d too small for 4 levels", h, w);
    const VolGeom g = make_geom(h, w);
    if (rec_bytes) *rec_bytes = g.rec;
    for (int l = 0; l < 4; ++l) {
        if (lvl_off) lvl_off[l] = g.off[l];
        if (nby) nby[l] = g.nby[l];
        if (nbx) nbx[l] = g.nbx[l];
    }
    if (src_rows) *src_rows = src_rows_padded(h * w);
    return SF_OK;
}

extern "C" int64_t sf_corr_blocked_bytes(int n_img, int h, int w) {
    if (n_img <= 0 || h < 8 || w < 8) return 0;
    return (int64_t)n_img * src_rows_padded(h * w) * make_geom(h, w).rec;
}

extern "C" int64_t sf_corr_build_blocked_ws_bytes(int n_img, int D, int h, int w) {
    if (n_img <= 0 || D <= 0 || h <= 0 || w <= 0) return 0;
    const int Np = (h * w + 8) / 8 * 8;
    return (int64_t)2 * n_img * (KD / 8) * Np * 16;
}

extern "C" int sf_corr_build_blocked(const float* f1, const float* f2, int64_t f_clip_stride, int64_t f_pair_stride,
                                     void* vol, int64_t vol_img_stride_bytes, int B, int pairs, int D, int h, int w,
                                     void* ws, int64_t ws_bytes, void* stream) {
    SF_REQUIRE(f1 && f2 && vol && ws, "sf_corr_build_blocked: null pointer");
    SF_REQUIRE(B > 0 && pairs > 0 && D > 0 && h > 0 && w > 0, "sf_corr_build_blocked: bad dims");
    SF_REQUIRE(D <= KD, "sf_corr_build_blocked: feature depth %d > %d (the resident-patch kernel holds all k of a patch in LDS)", D, KD);
    SF_REQUIRE((h >> 3) >= 1 && (w >> 3) >= 1, "sf_corr_build_blocked: feature grid %dx%d too small for 4 levels", h, w);
    const int n_img = B * pairs;
    BuildArgs g;
    g.g = make_geom(h, w);
    g.N = h * w;
    g.Np = (g.N + 8) / 8 * 8;
    SF_REQUIRE((int64_t)(KD / 8) * g.Np * 16 < ((int64_t)1 << 31), "sf_corr_build_blocked: feature image larger than 2 GiB");
    SF_REQUIRE((int64_t)32 * g.g.rec < ((int64_t)1 << 31), "sf_corr_build_blocked: feature grid %dx%d too large", h, w);
    SF_REQUIRE(ws_bytes >= sf_corr_build_blocked_ws_bytes(n_img, D, h, w) && (reinterpret_cast<uintptr_t>(ws) & 15) == 0,
               "sf_corr_build_blocked: needs a 16-byte aligned workspace of sf_corr_build_blocked_ws_bytes() bytes");
    SF_REQUIRE((reinterpret_cast<uintptr_t>(vol) & 127) == 0 && (vol_img_stride_bytes & 127) == 0 &&
                   vol_img_stride_bytes >= (int64_t)src_rows_padded(g.N) * g.g.rec,
               "sf_corr_build_blocked: vol must be 128-byte aligned, image stride a multiple of 128 and at least "
               "sf_corr_blocked_bytes(1, h, w)");
    SF_REQUIRE(n_img <= 32767, "sf_corr_build_blocked: B*pairs too large");
    g.ws = static_cast<const char*>(ws);
    g.vol = static_cast<char*>(vol);
    g.vol_img_stride = vol_img_stride_bytes;
    g.n_img = n_img; g.h = h; g.w = w;
    g.pcols = sf::ceil_div(w, PC);
    g.np = g.pcols * sf::ceil_div(h, PR);
    g.ntiles = sf::ceil_div(g.N, 32);
    g.nchunks = sf::ceil_div(g.ntiles, kTilesPerItem);
    g.tpc = sf::ceil_div(g.ntiles, g.nchunks);
    g.nchunks = sf::ceil_div(g.ntiles, g.tpc);
    g.scale = 1.0f / sqrtf((float)D);
#ifdef SF_CORR_TIMERS
    g.ts = getenv("SF_CORR_TS_BUF") ? (long long*)strtoull(getenv("SF_CORR_TS_BUF"), nullptr, 0) : nullptr;
#endif
    const int64_t n_wg = (int64_t)g.np * g.nchunks * n_img;
    SF_REQUIRE(n_wg < ((int64_t)1 << 31), "sf_corr_build_blocked: grid too large");
    hipLaunchKernelGGL(pack_f16z_kernel, dim3(sf::ceil_div(g.Np, 256), KD / 8, 2 * n_img), dim3(256), 0,
                       (hipStream_t)stream, f1, f2, f_clip_stride, f_pair_stride, (char*)ws, pairs, D, KD, g.N, g.Np);
    hipLaunchKernelGGL(corr_build_blocked_kernel, dim3((unsigned)n_wg), dim3(kBuildThreads), 0, (hipStream_t)stream, g);
    return sf::check_launch("sf_corr_build_blocked");
}

extern "C" int sf_corr_lookup_blocked(const void* vol, int64_t vol_img_stride_bytes, const float* coords, float* out,
                                      int64_t out_img_stride, void* out_koct, int64_t out_koct_img_stride, int B, int pairs,
                                      int h, int w, void* stream) {
    SF_REQUIRE(vol && coords && (out || out_koct), "sf_corr_lookup_blocked: null pointer");
    SF_REQUIRE(B > 0 && pairs > 0 && h > 0 && w > 0, "sf_corr_lookup_blocked: bad dims");
    SF_REQUIRE((h >> 3) >= 1 && (w >> 3) >= 1, "sf_corr_lookup_blocked: feature grid %dx%d too small for 4 levels", h, w);
    SF_REQUIRE((int64_t)B * pairs <= 65535, "sf_corr_lookup_blocked: B*pairs too large");
    LookArgs a;
    a.g = make_geom(h, w);
    SF_REQUIRE((int64_t)LP * a.g.rec < ((int64_t)1 << 31), "sf_corr_lookup_blocked: feature grid %dx%d too large", h, w);
    SF_REQUIRE((reinterpret_cast<uintptr_t>(vol) & 15) == 0 && (vol_img_stride_bytes & 15) == 0,
               "sf_corr_lookup_blocked: vol and its image stride must be 16-byte aligned");
    SF_REQUIRE(!out_koct || ((reinterpret_cast<uintptr_t>(out_koct) & 15) == 0 && (out_koct_img_stride & 7) == 0),
               "sf_corr_lookup_blocked: out_koct must be 16-byte aligned, its stride a multiple of 8 halves");
    a.vol = static_cast<const char*>(vol);
    a.vol_img_stride = vol_img_stride_bytes;
    a.coords = coords;
    a.out = out; a.out_img_stride = out_img_stride;
    a.out16 = static_cast<_Float16*>(out_koct); a.out16_img_stride = out_koct_img_stride;
    a.h = h; a.w = w; a.N = h * w;
    hipLaunchKernelGGL(corr_lookup_blocked_kernel, dim3(sf::ceil_div(a.N, LP), B * pairs), dim3(kThreads), 0,
                       (hipStream_t)stream, a);
    return sf::check_launch("sf_corr_lookup_blocked");
}
