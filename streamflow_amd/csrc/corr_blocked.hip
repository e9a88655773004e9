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
    const char* ws;               // packed features: image (img, side) = [KD / 8][Np][8] halves, pixel N.. = zeros
    char* vol;
    int64_t vol_img_stride;       // bytes
    int n_img, h, w, N, Np;
    int pairs, shared;            // shared: packed planes are per frame (see pack_f16z_kernel)
    VolGeom g;
    // target patches: np16 patches of 16 rows x 16 columns (two block rows x two blocks) cover block rows 0 .. 2 * rows16 - 1;
    // an odd last block row is covered by np32 patches of 8 rows x 32 columns
    int pcols16, np16, pcols32, np32, rows16;
    int ntiles, nchunks, tpc;     // 32-source tiles per image, chunks per image, tiles per chunk
    float scale;                  // applied in the epilogue (1 when folded into the packed source features)
#ifdef SF_CORR_TIMERS
    long long* ts;
#endif
};

// features fp32 [D][N] -> fp16 k-octet planes [(k / 8)][Np][8]; pixels N .. Np-1 are zero (the target of every patch
// cell that lies outside the image: its products are exactly 0).
// shared = 0: plane 2 i = the source side (f1) of image i = b * pairs + t, multiplied by `pre`; plane 2 i + 1 = its f2.
// shared = 1 (f2 == f1 + f_pair_stride: consecutive frames of a clip, the engine's call): every FRAME is packed once,
//             plane b * (pairs + 1) + j = frame j of clip b times `pre`; image (b, t) uses planes (b, t) and (b, t + 1).
__global__ __launch_bounds__(256) void pack_f16z_kernel(const float* f1, const float* f2, int64_t f_clip_stride,
                                                        int64_t f_pair_stride, char* ws, int pairs, int D, int Dp, int N,
                                                        int Np, float pre, int shared) {
    const int px = blockIdx.x * 256 + threadIdx.x, kq = blockIdx.y;
    if (px >= Np) return;
    const float* f;
    float m = pre;
    if (shared) {
        f = f1 + (int64_t)(blockIdx.z / (pairs + 1)) * f_clip_stride + (int64_t)(blockIdx.z % (pairs + 1)) * f_pair_stride;
    } else {
        const int side = blockIdx.z & 1, img = blockIdx.z >> 1;         // img = b * pairs + pair
        f = (side ? f2 : f1) + (int64_t)(img / pairs) * f_clip_stride + (int64_t)(img % pairs) * f_pair_stride;
        if (side) m = 1.0f;
    }
    f16x8 hv;
#pragma unroll
    for (int i = 0; i < 8; ++i) hv[i] = (_Float16)((px < N && kq * 8 + i < D) ? m * f[(int64_t)(kq * 8 + i) * N + px] : 0.f);
    const int64_t plane = (int64_t)(Dp / 8) * Np * 16;
    *reinterpret_cast<f16x8*>(ws + (int64_t)blockIdx.z * plane + ((int64_t)kq * Np + px) * 16) = hv;
}

// The same for four consecutive pixels per thread (N % 4 == 0, 16-byte aligned planes): eight 16-byte loads and four 16-byte
// stores in flight per thread instead of eight dwords and one -- the one-pixel form moved 3.7 TB/s.
__global__ __launch_bounds__(256) void pack_f16z4_kernel(const float* f1, const float* f2, int64_t f_clip_stride,
                                                         int64_t f_pair_stride, char* ws, int pairs, int D, int Dp, int N,
                                                         int Np, float pre, int shared) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int px = (blockIdx.x * 256 + threadIdx.x) * 4, kq = blockIdx.y;
    if (px >= Np) return;
    const float* f;
    float m = pre;
    if (shared) {
        f = f1 + (int64_t)(blockIdx.z / (pairs + 1)) * f_clip_stride + (int64_t)(blockIdx.z % (pairs + 1)) * f_pair_stride;
    } else {
        const int side = blockIdx.z & 1, img = blockIdx.z >> 1;
        f = (side ? f2 : f1) + (int64_t)(img / pairs) * f_clip_stride + (int64_t)(img % pairs) * f_pair_stride;
        if (side) m = 1.0f;
    }
    f32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        v[i] = (px < N && kq * 8 + i < D) ? *reinterpret_cast<const f32x4*>(f + (int64_t)(kq * 8 + i) * N + px) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t plane = (int64_t)(Dp / 8) * Np * 16;
    char* o = ws + (int64_t)blockIdx.z * plane + ((int64_t)kq * Np + px) * 16;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f16x8 hv;
#pragma unroll
        for (int i = 0; i < 8; ++i) hv[i] = (_Float16)(m * v[i][e]);
        *reinterpret_cast<f16x8*>(o + e * 16) = hv;
    }
}

#ifndef SF_CORRB_NT
#define SF_CORRB_NT 2            // cache policy of the level-0 stores (2 = non-temporal)
#endif

// BUILD, resident-patch form.  Work item = (image, chunk of source tiles, target patch); one workgroup of 8 waves per
// item and CU.  The patch's features for ALL 256 k (128 KB) are DMA'd into LDS once; after ONE barrier every wave is on
// its own: it draws 32-source tiles from a workgroup counter, gets that tile's sixteen A fragments straight from L2 into
// registers (a lane's fragment is one 16-byte k-octet of one pixel: no LDS, no DMA), reads B fragments from the resident
// patch eight MFMAs ahead of their use, and runs its epilogue while the other wave of its SIMD keeps the matrix pipe busy.
// No barrier, no DMA and no staging buffer inside the loop.
//  * The A fragments of the NEXT tile are requested before the epilogue's stores: vmcnt retires in issue order, so a load
//    issued behind the 32 stores of a tile would wait for their HBM acknowledgement (~16k cycles when the chip writes at
//    full rate) -- the first version of this kernel did, and its waves spent as long draining as computing.
//  * kShape16: the patch is 16 rows x 16 columns, lane = (block row l31 >> 4, column l31 & 15), so that a tile produces a
//    COMPLETE 8 x 8 level-1 block (a full 128-byte line).  With 8 x 32 patches the level-1 cells of a block come from two
//    work items at different times: partial-line writes, which HBM3 (no data mask) serves as read-modify-write -- the
//    store-only rate of that pattern measured 3.9 TB/s against 5.5 for full lines.  8 x 32 is used for an odd last block
//    row only (no wasted MFMA rows).
// (History: the first blocked kernel kept round 2's structure -- 128 x 256 tile per workgroup, both operands staged
// through a 2- or 3-stage LDS ring -- and measured 27k cycles of k-loop per tile of which 4.1k were MFMA issue: every
// wave sat ~190 cycles in each of its 48 LDS-DMA instructions and ~0.5k cycles per stage at the barrier; interleaving the
// DMA instructions with the MFMAs moved that time, it did not remove it.  DESIGN.md section 10.)
#ifdef SF_CORR_TIMERS
#define SF_TIMER_ARGS , long long& t_load, long long& t_k, long long& t_e, int& n_t
#define SF_TIMER_PASS , t_load, t_k, t_e, n_t
#else
#define SF_TIMER_ARGS
#define SF_TIMER_PASS
#endif
template <bool kShape16, bool kScale>
__device__ __forceinline__ void build_item(const BuildArgs& g, char* smem, int item SF_TIMER_ARGS) {
#ifdef SF_CORR_TIMERS
    const long long ts0 = __builtin_readcyclecounter();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, l31_ = lane & 31;
    // item order [image][chunk][patch]: the workgroups that run side by side on an XCD share the chunk's source features
    const int np = kShape16 ? g.np16 : g.np32, pcols = kShape16 ? g.pcols16 : g.pcols32;
    const int patch = item % np, chunk = (item / np) % g.nchunks, img = item / (np * g.nchunks);
    // patch origin in blocks: kShape16: block rows 2 pyb, 2 pyb + 1, blocks 2 pxb, 2 pxb + 1; else block row pyb, blocks 4 pxb ..
    const int pyb = kShape16 ? patch / pcols : 2 * g.rows16 + patch / pcols, pxb = patch % pcols;
    const int plane = (KD / 8) * g.Np * 16;                       // bytes of one packed image (< 2 GiB, host-checked)
    const int pa = g.shared ? (img / g.pairs) * (g.pairs + 1) + img % g.pairs : 2 * img;    // packed plane of the source side
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(g.ws) + (int64_t)pa * plane, 0, plane, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(g.ws) + (int64_t)(pa + 1) * plane, 0, plane, 0x00020000);
    const int kq_step = g.Np * 16;
    int* const counter = reinterpret_cast<int*>(smem + kPatchBytes);
    if (tid == 0) *counter = 0;
    // ---- the patch: LDS slot = k-octet * 256 + cell, cell = t * 32 + l (t = accumulator / row inside a block row, l = the
    // MFMA column lane), 128 pieces of 1 KB; piece p covers k-octet p / 4, cells (p % 4) * 64 + lane.  Wave w takes pieces
    // w, w + 8, ...: always cell group w % 4.  Cells outside the image read the zero pixel. ----
    {
        const int cell = (wave & 3) * 64 + lane, t = cell >> 5, l = cell & 31;
        const int ty = kShape16 ? 16 * pyb + 8 * (l >> 4) + t : 8 * pyb + t;
        const int tx = kShape16 ? 16 * pxb + (l & 15) : 32 * pxb + l;
        const int vob = ((ty < g.h && tx < g.w) ? ty * g.w + tx : g.N) * 16;
        static_assert(NW % 4 == 0, "a wave keeps its cell group");
        for (int p = wave; p < (KD / 8) * 4; p += NW)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(smem + p * 1024), 16, vob, (p >> 2) * kq_step, 0, 0);
    }
    const int t_begin = chunk * g.tpc, t_count = min(g.tpc, g.ntiles - t_begin);
    // (vmcnt is in order: this also drains the wave's stores.  The BUILTIN, not inline asm: hipcc tracks LDS-DMA as a pending
    // write to LDS and, unless it sees this wait itself, puts its own vmcnt(0) in front of the first LDS access of every
    // trip of the tile loop -- i.e. a full drain of the previous tile's stores at the top of each tile.)
    __builtin_amdgcn_s_waitcnt(0x0070);                           // vmcnt(0) expcnt(7) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
#ifdef SF_CORR_TIMERS
    const long long ts1 = __builtin_readcyclecounter();
    t_load += ts1 - ts0;
#endif
    const int rec = g.g.rec;
    char* const img_base = g.vol + (int64_t)img * g.vol_img_stride;
    const char* const sbB = smem + (khalf * BN + l31_) * 16;      // + (2 ks * 256 + t * 32) * 16
    constexpr int kSteps = KD / 16;
#if defined(SF_CORRB_ABLATE) && SF_CORRB_ABLATE == 2      // timing ablation: four k-steps only
    constexpr int kRun = 4;
#else
    constexpr int kRun = kSteps;
#endif
    auto grab = [&]() {
        int t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return __builtin_amdgcn_readfirstlane(t);
    };
    int tile = grab();
    if (tile >= t_count) return;
    for (;;) {
#ifdef SF_CORR_TIMERS
        const long long tt0 = __builtin_readcyclecounter();
#endif
        const int next = grab();                                  // (LDS atomic: its latency hides behind the k-loop)
        const int i0 = (t_begin + tile) * 32;                     // first source pixel of this tile
        // A fragment of k-step ks: k-octet 2 ks + khalf of source pixel i0 + l31 (pixels past N clamped: padding records);
        // a ring of four loads in flight.  (vmcnt retires in issue order, so the first of them also waits for the HBM
        // acknowledgement of the previous tile's stores; requesting the fragments before those stores changes nothing --
        // measured -- and costs 64 registers.)
        const int voa = (khalf * g.Np + min(i0 + l31_, g.N - 1)) * 16;
        constexpr int kRing = (NW <= 8) ? 4 : 2;
        u32x4 af[kRing];
#pragma unroll
        for (int s = 0; s < kRing; ++s) af[s] = __builtin_amdgcn_raw_buffer_load_b128(ra, voa, s * 2 * kq_step, 0);
        f32x16 acc[PR];
#pragma unroll
        for (int t = 0; t < PR; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
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
        // Issue order of the whole (fully unrolled) k-loop, pinned: the ring's loads first, then B fragment reads running
        // kAhead MFMAs ahead of their use, one A refill behind every eighth MFMA.  Left to itself hipcc sinks every load
        // next to its use (vmcnt(0) / lgkmcnt(0) in front of each MFMA: 28k cycles per tile for 4.1k of MFMA issue).
        {
            constexpr int kAhead = (NW <= 8) ? 8 : 3, kMfma = kRun * PR;
            __builtin_amdgcn_sched_group_barrier(0x020, kRing, 0);               // VMEM reads
            __builtin_amdgcn_sched_group_barrier(0x100, kAhead, 0);              // DS reads
#pragma unroll
            for (int i = 0; i < kMfma; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               // one MFMA
                if (i + kAhead < kMfma) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (i % PR == PR - 1 && i / PR + kRing < kRun) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef SF_CORR_TIMERS
        const long long tt1 = __builtin_readcyclecounter();
#endif
        // ---- epilogue.  C/D layout: lane = (MFMA column l31, k-half), register r = source row (r&3) + 8(r>>2) + 4 khalf,
        // acc[t] = row t of the lane's block row: a lane's eight values of one register are one block column of level 0.
        // Per-lane store offsets are recomputed per tile from a laundered lane id: as loop invariants they would occupy
        // ~10 VGPRs through the k-loop. ----
        int l31 = l31_;
        asm volatile("" : "+v"(l31));
        const int rowh = 4 * khalf * rec;
        const int br = kShape16 ? l31 >> 4 : 0, c = kShape16 ? l31 & 15 : l31;     // block row of the lane, column in the patch
        const int by0 = kShape16 ? 2 * pyb + br : pyb, bxp = kShape16 ? 2 * pxb : 4 * pxb;   // level-0 block row / first block
        // level 0: lanes of one block row = consecutive 16-byte block columns (256 or 512 contiguous bytes per source row)
        const int vo0_ = (by0 < g.g.nby[0] && bxp + (c >> 3) < g.g.nbx[0])
                             ? rowh + g.g.off[0] + (by0 * g.g.nbx[0] + bxp) * 128 + c * 16 : kDrop;
        // level 1: lane pair (2j, 2j+1) holds level-1 column j, four rows (block row by0: rows 4 (by0 & 1) .. +3 of level-1
        // block row by0 >> 1): an 8-byte piece; lane parity k1 stores source row (2 jp + k1) of a register pair.
        // kShape16: the two block rows of the patch supply both halves of every block column: 128 full bytes per source row.
        const int j1 = c >> 1, k1 = c & 1;
        const int by1 = by0 >> 1, bx1 = (bxp >> 1) + (j1 >> 3);
        const int vo1_ = (by1 < g.g.nby[1] && bx1 < g.g.nbx[1])
                             ? rowh + k1 * rec + g.g.off[1] + (by1 * g.g.nbx[1] + bx1) * 128 + (j1 & 7) * 16 + (by0 & 1) * 8 : kDrop;
        // level 2: lane quad = level-2 column (bxp * 2 + (c >> 2)), rows 2 (by0 & 3) .. +1 of block row by0 >> 2 (4 bytes);
        // lane k2 of the quad stores source row k2 of a register group
        const int k2 = c & 3, tx2 = bxp * 2 + (c >> 2);
        const int by2 = by0 >> 2;
        const int vo2_ = (by2 < g.g.nby[2] && (tx2 >> 3) < g.g.nbx[2])
                             ? rowh + k2 * rec + g.g.off[2] + (by2 * g.g.nbx[2] + (tx2 >> 3)) * 128 + (tx2 & 7) * 16 + (by0 & 3) * 4 : kDrop;
        // level 3: lanes 0..3 of an octet hold level-3 column bxp + (c >> 3), row by0 & 7 of block row by0 >> 3 (2 bytes)
        const int k3 = c & 7, tx3 = bxp + (c >> 3);
        const int by3 = by0 >> 3;
        const int vo3_ = (k3 < 4 && by3 < g.g.nby[3] && (tx3 >> 3) < g.g.nbx[3])
                             ? rowh + k3 * rec + g.g.off[3] + (by3 * g.g.nbx[3] + (tx3 >> 3)) * 128 + (tx3 & 7) * 16 + (by0 & 7) * 2 : kDrop;
        constexpr int kNt = SF_CORRB_NT;
#if defined(SF_CORRB_ABLATE) && SF_CORRB_ABLATE == 1      // timing ablation: no stores leave the CU
        const int vo0 = kDrop | (vo0_ & 0), vo1 = kDrop | (vo1_ & 0), vo2 = kDrop | (vo2_ & 0), vo3 = kDrop | (vo3_ & 0);
#elif defined(SF_CORRB_ABLATE) && SF_CORRB_ABLATE == 3    // only the level-0 stores leave the CU
        const int vo0 = vo0_, vo1 = kDrop | (vo1_ & 0), vo2 = kDrop | (vo2_ & 0), vo3 = kDrop | (vo3_ & 0);
#elif defined(SF_CORRB_ABLATE) && SF_CORRB_ABLATE == 4    // only the pooled levels' stores leave the CU
        const int vo0 = kDrop | (vo0_ & 0), vo1 = vo1_, vo2 = vo2_, vo3 = vo3_;
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
                    for (int t = 0; t < PR; ++t) v0[t] = kScale ? acc[t][r] * g.scale : acc[t][r];
                    u32x4 o;
#pragma unroll
                    for (int t = 0; t < 4; ++t) o[t] = pack_h2(v0[2 * t], v0[2 * t + 1]);
                    __builtin_amdgcn_raw_buffer_store_b128(o, rv, vo0, (ri + 8 * rq) * rec, kNt);
                    // HARDWARE HAZARD (gfx950, measured): a VALU write to the data registers of a >64-bit buffer store in the
                    // very next issue slots corrupts the stored data for the last lanes of each 16-lane group (sporadic wrong
                    // level-0 cells in columns 12..15 / 28..31).  hipcc only pads this hazard when soffset is an immediate
                    // (GCNHazardRecognizer::createsVALUHazard); here it is an SGPR, so pad by hand.
                    asm volatile("s_nop 1" : "+v"(o) : : "memory");      // (tied to the data registers: they stay live until here)
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
        if (next >= t_count) break;
        tile = next;
    }
}

// One workgroup per work item.  (A persistent form -- one workgroup per CU looping over items -- measured the same
// 1.13 ms and, inlined into that loop, hipcc no longer honours the pinned issue order of the k-loop: DESIGN.md section 10.)
template <bool kScale, bool kMixed>
__global__ __launch_bounds__(kBuildThreads, NW / 4) void corr_build_blocked_kernel(const BuildArgs g) {
    __shared__ __attribute__((aligned(1024))) char smem[kPatchBytes + 16];
#ifdef SF_CORR_TIMERS
    const long long ts0 = __builtin_readcyclecounter();
    const long long rt0 = __builtin_amdgcn_s_memrealtime();
    long long t_load = 0, t_k = 0, t_e = 0;
    int n_t = 0;
#endif
    // 16 x 16 patches first (if any), then the 8 x 32 patches
    const int n16 = g.np16 * g.nchunks * g.n_img;                 // workgroup-uniform
    const int id = sf::xcd_linear_id(blockIdx.x, gridDim.x);
    if (kMixed && id < n16) build_item<true, kScale>(g, smem, id SF_TIMER_PASS);
    else build_item<false, kScale>(g, smem, id - n16 SF_TIMER_PASS);
#ifdef SF_CORR_TIMERS
    if (g.ts && (threadIdx.x & 63) == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long* d = g.ts + ((int64_t)blockIdx.x * NW + (threadIdx.x >> 6)) * 8;
        d[0] = ts0; d[1] = t_load; d[2] = __builtin_readcyclecounter(); d[3] = rt0; d[4] = __builtin_amdgcn_s_memrealtime();
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
#ifndef SF_LOOKB_W2_ALWAYS
#define SF_LOOKB_W2_ALWAYS 0
#endif
#ifndef SF_LOOKB_STREAM_BYTES
// Volumes of at least this size are read with the streaming cache policy (slc): every line of a footprint is used once per lookup.
// Alone (bench.py --corr-only: 15 lookups back to back) that pays from ~2.5 GB on -- Sintel, 24 images = 3.3 GB: 78-81 -> 66-69 us per
// lookup; KITTI, 8 images = 1.2 GB: 27.0 -> 29.7 us, part of one lookup's lines survives to the next in the 256-MB infinity cache.
// INSIDE the step thirty other kernels run between two lookups and nothing survives anyway: streaming is then faster at every size
// (roofline_corr.frac KITTI 0.44-0.48 -> 0.49-0.50, 4 Sintel clips 0.49-0.51 -> 0.54-0.55, one clip 0.37-0.38 -> 0.39) -- the step is
// what ships, so: always.
#define SF_LOOKB_STREAM_BYTES 0
#endif
#ifndef SF_LOOKB_PF
#define SF_LOOKB_PF 1                           // items (footprints) prefetched ahead of the one being processed (1 .. 8)
#endif
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

template <int AUX>
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
        f.w0 = __builtin_amdgcn_raw_buffer_load_b128(rv, piece(0), 0, AUX);
        f.w1 = __builtin_amdgcn_raw_buffer_load_b128(rv, piece(1), 0, AUX);
        // rows 16, 17 (a third block row) are part of the footprint only when ys % 8 == 7: requested only then -- an unconditional
        // 4-byte request pulled two more 128-byte lines per level into L2 for seven footprints of eight (round 5)
#if SF_LOOKB_W2_ALWAYS
        f.w2 = __builtin_amdgcn_raw_buffer_load_b32(rv, piece(2), 0, AUX);
#else
        f.w2 = __builtin_amdgcn_raw_buffer_load_b32(rv, ((ys & 7) == 7) ? piece(2) : kDrop, 0, AUX);
#endif
        return f;
    };

    // footprints of the next SF_LOOKB_PF items are in flight while one is processed
    constexpr int PF = SF_LOOKB_PF;
    Item qs[8];
    Foot fs[8];
#pragma unroll
    for (int it = 0; it < PF; ++it) {
        qs[it] = locate(it);
        fs[it] = fetch(it, qs[it]);
    }
    if (PF > 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const Item q = qs[it];
        const Foot f = fs[it];
        if (it + PF < 8) {
            qs[it + PF] = locate(it + PF);
            fs[it + PF] = fetch(it + PF, qs[it + PF]);
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
#ifndef SF_CORRB_SHAPE16
#define SF_CORRB_SHAPE16 0       // 1: 16 x 16 patches for the even block rows (full-line level-1 writes, 256-byte level-0 runs).
#endif                           // Measured slower than 8 x 32 everywhere (Sintel 1177 vs 1131 us): off.
    g.rows16 = SF_CORRB_SHAPE16 ? g.g.nby[0] / 2 : 0;             // full pairs of block rows
    g.pcols16 = sf::ceil_div(w, 16);
    g.np16 = g.rows16 * g.pcols16;
    g.pcols32 = sf::ceil_div(w, 32);
    g.np32 = (g.g.nby[0] - 2 * g.rows16) * g.pcols32;
    g.ntiles = sf::ceil_div(g.N, 32);
    g.nchunks = sf::ceil_div(g.ntiles, kTilesPerItem);
    g.tpc = sf::ceil_div(g.ntiles, g.nchunks);
    g.nchunks = sf::ceil_div(g.ntiles, g.tpc);
    // 1 / sqrt(D) is folded into the packed features where that is exact (a power of two): into the source side, or -- when
    // f2 is f1 one frame on, so that every frame is packed ONCE and serves both sides -- as sqrt(scale) into every frame
    // (D = 256: 1/4 each).  Otherwise the epilogue multiplies.
    const float scale = 1.0f / sqrtf((float)D), root = sqrtf(scale);
    int e2 = 0;
    const bool pow2 = frexpf(scale, &e2) == 0.5f, root_pow2 = frexpf(root, &e2) == 0.5f;
    g.pairs = pairs;
    g.shared = (pairs > 1 && f2 == f1 + f_pair_stride && (root_pow2 || !pow2)) ? 1 : 0;
    const bool fold = g.shared ? root_pow2 : pow2;
    const float pre = !fold ? 1.0f : (g.shared ? root : scale);
    g.scale = fold ? 1.0f : scale;
#ifdef SF_CORR_TIMERS
    g.ts = getenv("SF_CORR_TS_BUF") ? (long long*)strtoull(getenv("SF_CORR_TS_BUF"), nullptr, 0) : nullptr;
#endif
    const int64_t n_items = (int64_t)(g.np16 + g.np32) * g.nchunks * n_img;
    SF_REQUIRE(n_items < ((int64_t)1 << 31), "sf_corr_build_blocked: too many work items");
    const int64_t n_wg = n_items;
    const bool vec4 = (g.N & 3) == 0 && (g.Np & 3) == 0 && (f_clip_stride & 3) == 0 && (f_pair_stride & 3) == 0 &&
                      ((reinterpret_cast<uintptr_t>(f1) | reinterpret_cast<uintptr_t>(f2)) & 15) == 0;
    if (vec4)
        hipLaunchKernelGGL(pack_f16z4_kernel, dim3(sf::ceil_div(g.Np, 1024), KD / 8, g.shared ? B * (pairs + 1) : 2 * n_img), dim3(256), 0,
                           (hipStream_t)stream, f1, f2, f_clip_stride, f_pair_stride, (char*)ws, pairs, D, KD, g.N, g.Np, pre, g.shared);
    else
        hipLaunchKernelGGL(pack_f16z_kernel, dim3(sf::ceil_div(g.Np, 256), KD / 8, g.shared ? B * (pairs + 1) : 2 * n_img), dim3(256), 0,
                           (hipStream_t)stream, f1, f2, f_clip_stride, f_pair_stride, (char*)ws, pairs, D, KD, g.N, g.Np, pre, g.shared);
    const dim3 grid((unsigned)n_wg), block(kBuildThreads);
    hipStream_t st = (hipStream_t)stream;
    if (g.np16 > 0) {
        if (fold) hipLaunchKernelGGL((corr_build_blocked_kernel<false, true>), grid, block, 0, st, g);
        else hipLaunchKernelGGL((corr_build_blocked_kernel<true, true>), grid, block, 0, st, g);
    } else {
        if (fold) hipLaunchKernelGGL((corr_build_blocked_kernel<false, false>), grid, block, 0, st, g);
        else hipLaunchKernelGGL((corr_build_blocked_kernel<true, false>), grid, block, 0, st, g);
    }
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
    const dim3 grid(sf::ceil_div(a.N, LP), B * pairs);
    if ((int64_t)B * pairs * vol_img_stride_bytes >= SF_LOOKB_STREAM_BYTES)
        hipLaunchKernelGGL(corr_lookup_blocked_kernel<2>, grid, dim3(kThreads), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(corr_lookup_blocked_kernel<0>, grid, dim3(kThreads), 0, (hipStream_t)stream, a);
    return sf::check_launch("sf_corr_lookup_blocked");
}
