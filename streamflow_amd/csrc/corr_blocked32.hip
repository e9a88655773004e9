// Correlation pyramids in BLOCKED fp32 layout: build (K1 + K2) and radius-4 lookup (K3) for the fp32-class presets.
//
// Reference: core/corr.py:7-21,46-54 (volume + avg-pool pyramid, kept in fp32: streamflow.py:107,110), :23-44 (lookup),
// core/utils/utils.py:65-79.  BASELINE.json configuration 3 ("KITTI-shape full-res 4D volume") as worded.
//
// WHY (round 6).  In the reference's own layout -- one [h_l][w_l] fp32 map per source pixel, rows pitched
// to cache lines by csrc/corr.hip -- a 10 x 10 bilinear footprint is ten 40-byte rows in ten different 128-byte lines (~11.6 lines
// per level with the straddles): the lookup fetched 5.9 KB per pixel for 1.6 KB of footprints, and the build wrote 128-byte runs
// with one 4-byte store per lane.  Here, as in csrc/corr_blocked.hip for fp16 cells, every pyramid level of every source pixel is
// stored as BLOCKS of one cache line, 4 rows x 8 columns of fp32 cells, column-major inside the block:
//
//     record(source pixel i) = [level 0 blocks | level 1 | level 2 | level 3]                                     (rec bytes)
//     level l: ceil(hl / 4) x ceil(wl / 8) blocks, block (by, bx) at off[l] + (by * nbx[l] + bx) * 128
//     cell (ty, tx) of the level at block (ty / 4, tx / 8), byte ((tx % 8) * 4 + ty % 4) * 4
//
//  * a footprint touches (1 + 9/4) x (1 + 9/8) = 6.9 lines per level on average instead of ~11.6;
//  * a 16-byte piece is one block COLUMN = four vertically adjacent cells: in the build a lane of the MFMA C/D layout holds two of
//    those (8 patch rows of one target column), so level 0 leaves as two 16-byte stores per lane and accumulator register, each
//    512 contiguous bytes per source row and instruction (8 rows x 4 columns, the first form, left every store instruction with
//    half-written 32-byte sectors: 1.08 ms per KITTI build against 0.95 ms of the row-major kernel); in the lookup one lane owns
//    one footprint column (vertical lerp in-lane, horizontal by DPP);
//  * cells of a block that lie outside the level (padding) have UNSPECIFIED contents; the lookup masks them.
//
// Arithmetic of the build: split fp16 operands (f = hi + lo, three MFMA products, fp32 accumulation: ~2^-20 relative), the main loop
// of corr.hip's corr_build_dma_kernel (operand stages HBM/L2 -> LDS by buffer_load ... lds, two 24-KB stages).  The lookup hands the
// 324 correlation features over as fp32 planes [324][N] (the fp32-class presets' operand format).
#include "sf_common.h"
#include "split_operand.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;
using sf_split::f16x8;

constexpr int kThreads = 256;
constexpr int kDrop = (int)0x80000000u;          // buffer offset past any num_records: the access is dropped / reads 0

struct Geom32 {
    int hl[4], wl[4], nby[4], nbx[4], off[4];    // level sizes, blocks per level, byte offset of the level in a record
    int rec;                                     // bytes per source pixel
};

Geom32 make_geom32(int h, int w) {
    Geom32 g;
    int o = 0;
    for (int l = 0; l < 4; ++l) {
        g.hl[l] = h >> l; g.wl[l] = w >> l;
        g.nby[l] = (g.hl[l] + 3) / 4; g.nbx[l] = (g.wl[l] + 7) / 8;
        g.off[l] = o;
        o += g.nby[l] * g.nbx[l] * 128;
    }
    g.rec = o;
    return g;
}

inline int src_rows_padded(int N) { return (N + 127) / 128 * 128; }

__device__ __forceinline__ float dpp_xor1(float v) {     // lane ^ 1 (quad_perm [1,0,3,2])
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float v) {     // lane ^ 2 (quad_perm [2,3,0,1])
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_shl4(float v) {     // lane + 4 inside a row of 16 (row_shl:4)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x104, 0xF, 0xF, true));
}
__device__ __forceinline__ float swz_xor16(float v) {    // lane ^ 16 (ds_swizzle bit mode: and 0x1F, or 0, xor 0x10)
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
}
__device__ __forceinline__ float dpp_shl1(float v) {     // lane + 1 inside a row of 16 (row_shl:1)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xF, 0xF, true));
}

// ------------------------------------------------------------------------------------------------
// build
// ------------------------------------------------------------------------------------------------
constexpr int BM = 128;          // source pixels per workgroup (2 x 2 waves of 64 sources x 128 targets)
constexpr int PR = 8, PC = 32;   // target patch: 8 rows x 32 columns = two block rows of four blocks
constexpr int BN = PR * PC;
constexpr int DK = 16;                              // k per stage
constexpr int ST_A = (DK / 8) * BM * 16;            // bytes of the A_hi (= A_lo) part of a stage
constexpr int ST_B = (DK / 8) * BN * 16;            // bytes of B_hi (= B_lo)
constexpr int STAGE = 2 * ST_A + 2 * ST_B;          // 24576
#ifndef SF_CORRB32_NSTAGE
#define SF_CORRB32_NSTAGE 2
#endif
#ifndef SF_CORRB32_WGS
#define SF_CORRB32_WGS 3                            // workgroups per CU the register / LDS budget is sized for
#endif
constexpr int NSTAGE = SF_CORRB32_NSTAGE;

struct Build32Args {
    char* vol;
    int64_t vol_img_stride;       // bytes
    int n_img, h, w, N, Dp;
    int pcols, np, mt, pblk;      // patch columns, patches per image, m-tiles per image, patches per L2-resident block
    float scale;
    Geom32 g;
};

// features fp32 [D][N] -> (hi, lo) fp16 k-octet planes: ws image (img, side) = [hi | lo], plane[(k / 8) * N + pixel][k % 8]
__global__ __launch_bounds__(256) void split_pack32_kernel(const float* f1, const float* f2, int64_t f_clip_stride, int64_t f_pair_stride,
                                                           char* ws, int pairs, int D, int Dp, int N) {
    using namespace sf_split;
    const int px = blockIdx.x * 256 + threadIdx.x, kq = blockIdx.y;
    const int side = blockIdx.z & 1, img = blockIdx.z >> 1;             // img = b * pairs + pair
    if (px >= N) return;
    const float* f = (side ? f2 : f1) + (int64_t)(img / pairs) * f_clip_stride + (int64_t)(img % pairs) * f_pair_stride;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (kq * 8 + i < D) ? f[(int64_t)(kq * 8 + i) * N + px] : 0.f;
    const Split8 s8 = split8(v);
    const int64_t half = (int64_t)(Dp / 8) * N * 16;
    char* dst = ws + (int64_t)blockIdx.z * 2 * half + ((int64_t)kq * N + px) * 16;
    *reinterpret_cast<f16x8*>(dst) = s8.hi;
    *reinterpret_cast<f16x8*>(dst + half) = s8.lo;
}

struct TileId { int img, m_tile, patch; };
// XCD-aware order [image][patch block][m-tile][patch in block]: the `pblk` target patches of a block stay in the XCD's L2 while all
// m-tiles stream past them (csrc/corr.hip)
__device__ __forceinline__ TileId build_tile(const Build32Args& g, int id) {
    const int per_img = g.np * g.mt;
    TileId t;
    t.img = id / per_img;
    int r = id % per_img;
    const int full = g.np / g.pblk;
    if (r < full * g.pblk * g.mt) {
        const int blk = r / (g.pblk * g.mt), r2 = r % (g.pblk * g.mt);
        t.m_tile = r2 / g.pblk;
        t.patch = blk * g.pblk + r2 % g.pblk;
    } else {
        r -= full * g.pblk * g.mt;
        const int rem = g.np - full * g.pblk;
        t.m_tile = r / rem;
        t.patch = full * g.pblk + r % rem;
    }
    return t;
}

// One workgroup = 128 source pixels x one target patch of 8 rows x 32 columns; wave (wm, wn) = 64 sources x 128 targets (columns
// 16 wn .. 16 wn + 15 of the patch): 2 x 4 accumulator tiles of 32 x 32.  Per k-step a wave reads 4 A + 8 B fragments for 24 MFMAs;
// the first form (wave = 32 sources x the whole patch: 2 + 16 fragments) asked 96 of the LDS's 128 bytes per clock at full MFMA rate
// on top of the DMA's writes -- its main loop alone ran 626 us per KITTI build for 334 us of MFMA issue.
// Measured (KITTI, 8 pairs): 870-900 us per build; epilogue alone 500 us (2.5 GB written: 4.7 TB/s), k-loops alone 610 us, of
// which operand DMA alone 310 us and fragment reads + MFMA alone 440 us -- at 1.64 GHz: a loop of builds holds the package at its 1400 W cap
// (1.97 GHz), the MFMA-only variant is clock-limited further.  Three stages / two workgroups per CU, and a pseudo-random start stagger of the
// first round of workgroups (against lockstep of k-loops and epilogues) measured the same or worse.
// MFMA column j of N-tile nt = target (row 4 (j >> 4) + nt, column 16 wn + (j & 15)): a lane's four tiles of one register are the four
// rows of one 16-byte block column of level 0.
__global__ __launch_bounds__(kThreads, SF_CORRB32_WGS) void corr_build_blocked32_kernel(const Build32Args g, const char* ws) {
    __shared__ __attribute__((aligned(1024))) char smem[NSTAGE * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int khalf = lane >> 5, l31_ = lane & 31;
    const TileId tile = build_tile(g, sf::xcd_linear_id(blockIdx.x, gridDim.x));
    const int m0 = tile.m_tile * BM;
    const int pyb = tile.patch / g.pcols, pxb = tile.patch % g.pcols;          // patch origin: rows 8 pyb .., columns 32 pxb ..
    const int py0 = pyb * PR, px0 = pxb * PC;
    const int half = (g.Dp / 8) * g.N * 16;                     // bytes of one hi (or lo) plane (< 2 GiB, host-checked)
    const char* imgA = ws + (int64_t)(tile.img * 2 + 0) * 2 * half;
    const char* imgB = ws + (int64_t)(tile.img * 2 + 1) * 2 * half;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(imgA), 0, 2 * half, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(imgB), 0, 2 * half, 0x00020000);
    // per-thread source offsets (pixels / patch cells past the image are clamped: their products land in padding cells / records)
    const int voa = ((tid >> 7) * g.N + min(m0 + (tid & 127), g.N - 1)) * 16;          // slot = kq * 128 + px = tid
    // B slot tid = wn * 128 + nt * 32 + j  ->  target (4 (j >> 4) + nt, 16 wn + (j & 15)) of the patch
    const int bty = 4 * ((tid >> 4) & 1) + ((tid >> 5) & 3), btx = 16 * (tid >> 7) + (tid & 15);
    const int vob = (min(py0 + bty, g.h - 1) * g.w + min(px0 + btx, g.w - 1)) * 16;
    const int kq_step = g.N * 16;
    auto issue = [&](int kt, int buf) {
        char* sb = smem + buf * STAGE + wave * 1024;
        const int so = kt * (DK / 8) * kq_step;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(sb), 16, voa, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(sb + ST_A), 16, voa + half, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(sb + 2 * ST_A), 16, vob, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(sb + 2 * ST_A + 4096), 16, vob, so + kq_step, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(sb + 2 * ST_A + ST_B), 16, vob + half, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(sb + 2 * ST_A + ST_B + 4096), 16, vob + half, so + kq_step, 0, 0);
    };
    constexpr int MT = 2, NT = 4;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][t][r] = 0.f;
#if defined(SF_CORRB32_ABLATE) && SF_CORRB32_ABLATE == 2     // timing ablation: one k-stage only (the epilogue alone)
    const int nk = 1;
#else
    const int nk = g.Dp / DK;
#endif
    const int offa = (khalf * BM + wm * 64 + l31_) * 16;
    const int offb = 2 * ST_A + (khalf * BN + wn * 128 + l31_) * 16;
#pragma unroll
    for (int s0 = 0; s0 < NSTAGE - 1; ++s0)
        if (s0 < nk) issue(s0, s0);
    int cur = 0, nxt = NSTAGE - 1;
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's 6 pieces of stage kt have landed (the NSTAGE - 2 newer stages may still be in flight) ...
        if (NSTAGE == 2 || kt + NSTAGE - 2 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (NSTAGE == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // ... everyone's; and slot nxt (stage kt - 1) is no longer read
#if !(defined(SF_CORRB32_ABLATE) && SF_CORRB32_ABLATE == 4)   // timing ablation 4: no operand loads after the prologue (+ no stores)
        if (kt + NSTAGE - 1 < nk) issue(kt + NSTAGE - 1, nxt);
#endif
        const char* sb = smem + cur * STAGE;
        cur = (cur == NSTAGE - 1) ? 0 : cur + 1;
        nxt = (nxt == NSTAGE - 1) ? 0 : nxt + 1;
#if defined(SF_CORRB32_ABLATE) && SF_CORRB32_ABLATE == 5      // timing ablation 5: operand loads only, no fragment reads / MFMA (+ no stores)
        continue;
#endif
        f16x8 ah[MT], al[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            ah[mt] = *reinterpret_cast<const f16x8*>(sb + offa + mt * 512);
            al[mt] = *reinterpret_cast<const f16x8*>(sb + offa + mt * 512 + ST_A);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f16x8 bh = *reinterpret_cast<const f16x8*>(sb + offb + t * 512);
            const f16x8 bl = *reinterpret_cast<const f16x8*>(sb + offb + ST_B + t * 512);
            // (per accumulator the order of the three products is the one of csrc/corr.hip's f16x3 build: al bh, ah bl, ah bh)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh, acc[mt][t], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl, acc[mt][t], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh, acc[mt][t], 0, 0, 0);
        }
    }

    // ---- epilogue.  C/D layout: lane = (MFMA column l31 = (row half th, patch column c), k-half), register r of tile (mt, nt) = source row
    // 32 mt + (r & 3) + 8 (r >> 2) + 4 khalf, target row 4 th + nt ----
    int l31 = l31_;
    asm volatile("" : "+v"(l31));                                // (per-lane store offsets are not loop invariants of the k-loop)
    const int c = l31 & 15, th = l31 >> 4;
    const int rec = g.g.rec;
    const int i0 = m0 + wm * 64;                                 // first source pixel of the wave (records are padded to 128 sources)
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
        g.vol + (int64_t)tile.img * g.vol_img_stride + (int64_t)i0 * rec, 0, 64 * rec, 0x00020000);
    const int rowh = 4 * khalf * rec;
    const int tx0 = px0 + 16 * wn + c;
    // level 0: the lane's four rows = one 16-byte block column of block row 2 pyb + th; 16 lanes = 256 contiguous bytes
    const int by0 = 2 * pyb + th;
    const int vo0 = (by0 < g.g.nby[0] && (tx0 >> 3) < g.g.nbx[0]) ? rowh + g.g.off[0] + by0 * g.g.nbx[0] * 128 + tx0 * 16 : kDrop;
    // level 1: lane pair (2 j, 2 j + 1) holds level-1 column tx1 after the horizontal step, rows 2 th, 2 th + 1 of level-1 block row pyb:
    // an 8-byte piece (lanes l and l + 16 of one instruction complete the 16-byte column); lane parity k1 stores source row 2 jp + k1
    const int k1 = c & 1, tx1 = tx0 >> 1;
    const int vo1 = (pyb < g.g.nby[1] && (tx1 >> 3) < g.g.nbx[1])
                        ? rowh + k1 * rec + g.g.off[1] + pyb * g.g.nbx[1] * 128 + tx1 * 16 + th * 8 : kDrop;
    // level 2: lane quad = level-2 column tx2, row 2 (pyb & 1) + th of block row pyb >> 1 (4 bytes); lane k2 stores source row k2
    const int k2 = c & 3, tx2 = tx0 >> 2, by2 = pyb >> 1;
    const int vo2 = (by2 < g.g.nby[2] && (tx2 >> 3) < g.g.nbx[2])
                        ? rowh + k2 * rec + g.g.off[2] + by2 * g.g.nbx[2] * 128 + tx2 * 16 + (2 * (pyb & 1) + th) * 4 : kDrop;
    // level 3: lanes 0..3 of an octet of the upper row half hold level-3 column tx3, row pyb & 3 of block row pyb >> 2 (4 bytes)
    const int k3 = c & 7, tx3 = tx0 >> 3, by3 = pyb >> 2;
    const int vo3 = (th == 0 && k3 < 4 && by3 < g.g.nby[3] && (tx3 >> 3) < g.g.nbx[3])
                        ? rowh + k3 * rec + g.g.off[3] + by3 * g.g.nbx[3] * 128 + tx3 * 16 + (pyb & 3) * 4 : kDrop;
#if defined(SF_CORRB32_ABLATE) && (SF_CORRB32_ABLATE == 1 || SF_CORRB32_ABLATE >= 3)   // timing ablations: no pooled-level stores
#define SF_VO(x) (kDrop | ((x) & 0))
#else
#define SF_VO(x) (x)
#endif
#if defined(SF_CORRB32_ABLATE) && (SF_CORRB32_ABLATE == 1 || SF_CORRB32_ABLATE >= 4)     // ... no store at all leaves the CU (the main loop alone)
#define SF_VO0(x) (kDrop | ((x) & 0))
#else
#define SF_VO0(x) (x)
#endif
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {                   // register group: source rows 32 mt + 8 rq + 4 khalf + (0..3)
            float sel2 = 0.f, sel3 = 0.f;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {               // register pair (2 jp, 2 jp + 1) of the group
                float sel1[2] = {0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int ri = 2 * jp + u, r = 4 * rq + ri;
                    float v0[NT];
                    u32x4 o0;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        v0[t] = acc[mt][t][r] * g.scale;
                        o0[t] = __builtin_bit_cast(unsigned, v0[t]);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(o0, rv, SF_VO0(vo0), (32 * mt + ri + 8 * rq) * rec, 2);
                    // (gfx950: a VALU write to the data registers of a > 64-bit buffer store with an SGPR soffset in the very next issue
                    // slots corrupts the stored data -- csrc/corr_blocked.hip; pad by hand, tied to the data registers)
                    asm volatile("s_nop 1" : "+v"(o0) : : "memory");
                    float v1[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const float sm = v0[2 * t] + v0[2 * t + 1];
                        v1[t] = 0.25f * (sm + dpp_xor1(sm));
                        sel1[t] = (k1 == u) ? v1[t] : sel1[t];
                    }
                    const float s2 = v1[0] + v1[1];
                    const float v2 = 0.25f * (s2 + dpp_xor2(s2));
                    sel2 = (k2 == ri) ? v2 : sel2;
                    const float s3 = v2 + swz_xor16(v2);               // the other row half: level-2 rows 2 j, 2 j + 1
                    const float v3 = 0.25f * (s3 + dpp_shl4(s3));
                    sel3 = (k3 == ri) ? v3 : sel3;
                }
                u32x2 o1;
                o1[0] = __builtin_bit_cast(unsigned, sel1[0]); o1[1] = __builtin_bit_cast(unsigned, sel1[1]);
                __builtin_amdgcn_raw_buffer_store_b64(o1, rv, SF_VO(vo1), (32 * mt + 2 * jp + 8 * rq) * rec, 0);
            }
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sel2), rv, SF_VO(vo2), (32 * mt + 8 * rq) * rec, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sel3), rv, SF_VO(vo3), (32 * mt + 8 * rq) * rec, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// lookup
// ------------------------------------------------------------------------------------------------
#ifndef SF_LOOK32_STREAM_BYTES
// volumes past this size are read with the streaming cache policy (slc; csrc/corr_blocked.hip): Sintel, 24 images = 6.7 GB: 126 -> 110 us per
// lookup; KITTI, 8 images = 2.4 GB: 51-53 us either way
#define SF_LOOK32_STREAM_BYTES (5ll << 29)
#endif
#ifndef SF_LOOK32_LP
#define SF_LOOK32_LP 32
#endif
constexpr int LP = SF_LOOK32_LP;                // source pixels per workgroup (8 threads each)
constexpr int kLookThreads = LP * 8;
#ifndef SF_LOOK32_PASSES
#define SF_LOOK32_PASSES 1                      // 2: levels 0-1 and 2-3 leave through a half-size transpose buffer (4 workgroups per CU)
#endif
constexpr int NCH = 324;
constexpr int NP = SF_LOOK32_PASSES;
constexpr int CH_PASS = NCH / NP;               // channels per pass through the transpose buffer
constexpr int TROW = CH_PASS + 1;               // floats per pixel in the transpose buffer (odd: the read-back of 32 pixels is conflict-free)
#ifndef SF_LOOK32_PF0
#define SF_LOOK32_PF0 (SF_LOOK32_PASSES == 1 ? 8 : 5)
#endif
constexpr int PF0 = SF_LOOK32_PF0;              // items whose loads are issued up front

struct Look32Args {
    const char* vol;
    int64_t vol_img_stride;       // bytes
    const float* coords;
    float* out;                   // fp32 planes [324][N] per image
    int64_t out_img_stride;
    int h, w, N;
    Geom32 g;
};

template <int AUX>
__global__ __launch_bounds__(kLookThreads, (NP == 1 ? 768 : 1024) / kLookThreads) void corr_lookup_blocked32_kernel(const Look32Args a) {
    __shared__ float T[LP * TROW];
    const int tid = threadIdx.x;
    const int grp = tid >> 4, c = tid & 15;               // 16 lanes per footprint: lane c = footprint column c (10 used)
    const int img = blockIdx.y, p0 = blockIdx.x * LP;
    const int rec = a.g.rec;
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(a.vol) + (int64_t)img * a.vol_img_stride + (int64_t)p0 * rec, 0, LP * rec, 0x00020000);
    float cxs[2], cys[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int p = min(p0 + e * (LP / 2) + grp, a.N - 1);
        cxs[e] = a.coords[((int64_t)img * 2 + 0) * a.N + p];
        cys[e] = a.coords[((int64_t)img * 2 + 1) * a.N + p];
    }
    int g_wl[4], g_hl[4], g_nby[4], g_rowb[4], g_off[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        g_wl[l] = a.g.wl[l]; g_hl[l] = a.g.hl[l]; g_nby[l] = a.g.nby[l]; g_rowb[l] = a.g.nbx[l] * 128; g_off[l] = a.g.off[l];
        asm volatile("" : "+s"(g_wl[l]), "+s"(g_hl[l]), "+s"(g_nby[l]), "+s"(g_rowb[l]), "+s"(g_off[l]));
    }
    // item it = (level it >> 1, pixel (it & 1) * 16 + grp).  The loads of PF0 items are issued before the first footprint is touched
    // (one pass: all 32 loads of a thread, 104 VGPRs of data in flight) -- memory-level parallelism is what a gather of cache lines
    // that nobody else reads lives on
    struct Item { int ys; float fx, fy; };
    struct Foot { u32x4 w0, w1, w2; unsigned w3; };
    Item q[8];
    Foot f[8];
    auto issue = [&](int it) {
        const int l = it >> 1, e = it & 1, pix = e * (LP / 2) + grp;
        const float inv = 1.0f / (float)(1 << l);
        float cx = cxs[e] * inv, cy = cys[e] * inv;
        if (!(cx > -1.0e6f && cx < 1.0e6f)) cx = -1.0e6f;     // (far out: zero padding only; also swallows NaN / inf)
        if (!(cy > -1.0e6f && cy < 1.0e6f)) cy = -1.0e6f;
        const float fx0 = floorf(cx), fy0 = floorf(cy);
        const int x0 = (int)fx0, y0 = (int)fy0;
        q[it].fx = cx - fx0; q[it].fy = cy - fy0;
        const int tx = x0 - 4 + c, ys = y0 - 4;
        q[it].ys = ys;
        const bool col_ok = (c < 10) & ((unsigned)tx < (unsigned)g_wl[l]);
        const int byf = ys >> 2;
        const int col = pix * rec + g_off[l] + tx * 16;            // block bx = tx / 8, column tx % 8: (bx * 8 + tx % 8) * 16
        auto piece = [&](int k) {
            const int by = byf + k;
            return (col_ok & ((unsigned)by < (unsigned)g_nby[l])) ? col + by * g_rowb[l] : kDrop;
        };
        // rows 4 byf .. 4 byf + 12 of this column: three whole block columns + (when ys % 4 == 3) the first row of a fourth
        f[it].w0 = __builtin_amdgcn_raw_buffer_load_b128(rv, piece(0), 0, AUX);
        f[it].w1 = __builtin_amdgcn_raw_buffer_load_b128(rv, piece(1), 0, AUX);
        f[it].w2 = __builtin_amdgcn_raw_buffer_load_b128(rv, piece(2), 0, AUX);
        f[it].w3 = __builtin_amdgcn_raw_buffer_load_b32(rv, ((ys & 3) == 3) ? piece(3) : kDrop, 0, AUX);
    };
    float* o = a.out + (int64_t)img * a.out_img_stride;
    auto write_out = [&](int ch0) {                        // (channel, pixel): 32 consecutive pixels of a channel = 128 contiguous bytes
        __syncthreads();
        for (int i = tid; i < CH_PASS * LP; i += kLookThreads) {
            const int pix = i % LP, ch = i / LP;
            if (p0 + pix < a.N) o[(int64_t)(ch0 + ch) * a.N + p0 + pix] = T[pix * TROW + ch];
        }
        if (ch0 + CH_PASS < NCH) __syncthreads();
    };
#pragma unroll
    for (int it = 0; it < PF0; ++it) issue(it);
    __builtin_amdgcn_sched_barrier(0);                     // (hipcc would sink every load next to its use)
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int l = it >> 1, pix = (it & 1) * (LP / 2) + grp;
        const int ys = q[it].ys, s = ys & 3;
        unsigned W[13];
#pragma unroll
        for (int i = 0; i < 4; ++i) { W[i] = f[it].w0[i]; W[4 + i] = f[it].w1[i]; W[8 + i] = f[it].w2[i]; }
        W[12] = f[it].w3;
        if (it + PF0 < 8) issue(it + PF0);
        // rows ys .. ys + 9 out of the 13: shift by s in two select steps (bit-select masks, v_bfi_b32: written as
        // `cond ? W[i + 2] : W[i]` hipcc turns the chain into a dynamically indexed array in SCRATCH memory -- csrc/corr_blocked.hip)
        const unsigned m2 = 0u - ((unsigned)(s >> 1) & 1u), m1 = 0u - ((unsigned)s & 1u);
        unsigned W1[11];
        float F[10];
#pragma unroll
        for (int i = 0; i < 11; ++i) W1[i] = (W[i + 2] & m2) | (W[i] & ~m2);
        // padding rows inside the last block row (hl % 4 != 0) hold unspecified data: clear rows >= hl
        const int nvalid = g_hl[l] - ys;                  // rows b < nvalid are inside the level (b < -ys: dropped loads read 0)
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const unsigned d = (W1[i + 1] & m1) | (W1[i] & ~m1);
            F[i] = __builtin_bit_cast(float, (i < nvalid) ? d : 0u);
        }
        const float wy1 = q[it].fy, wy0 = 1.f - q[it].fy, wx1 = q[it].fx, wx0 = 1.f - q[it].fx;
        float R[9];
#pragma unroll
        for (int b = 0; b < 9; ++b) {                      // (all lanes: lane 9 supplies column 9 to lane 8 through the DPP shift)
            const float v = F[b] * wy0 + F[b + 1] * wy1;
            R[b] = v * wx0 + dpp_shl1(v) * wx1;
        }
        if (c < 9) {
            float* t0 = T + pix * TROW + (l * 81 - (NP == 2 && l >= 2 ? CH_PASS : 0)) + c * 9;   // channel l * 81 + a * 9 + b with a = c (corr.py:31-37)
#pragma unroll
            for (int b = 0; b < 9; ++b) t0[b] = R[b];
        }
        if (NP == 2 && it == 3) write_out(0);
    }
    write_out(NCH - CH_PASS);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" int sf_corr_blocked32_geometry(int h, int w, int64_t* rec_bytes, int64_t* lvl_off, int32_t* nby, int32_t* nbx, int64_t* src_rows) {
    SF_REQUIRE(h > 0 && w > 0, "sf_corr_blocked32_geometry: bad dims");
    SF_REQUIRE((h >> 3) >= 1 && (w >> 3) >= 1, "sf_corr_blocked32_geometry: feature grid %dx%d too small for 4 levels", h, w);
    const Geom32 g = make_geom32(h, w);
    if (rec_bytes) *rec_bytes = g.rec;
    for (int l = 0; l < 4; ++l) {
        if (lvl_off) lvl_off[l] = g.off[l];
        if (nby) nby[l] = g.nby[l];
        if (nbx) nbx[l] = g.nbx[l];
    }
    if (src_rows) *src_rows = src_rows_padded(h * w);
    return SF_OK;
}

extern "C" int64_t sf_corr_blocked32_bytes(int n_img, int h, int w) {
    if (n_img <= 0 || h < 8 || w < 8) return 0;
    return (int64_t)n_img * src_rows_padded(h * w) * make_geom32(h, w).rec;
}

extern "C" int64_t sf_corr_build_blocked32_ws_bytes(int n_img, int D, int h, int w) {
    if (n_img <= 0 || D <= 0 || h <= 0 || w <= 0) return 0;
    const int Dp = sf::ceil_div(D, 32) * 32;
    return (int64_t)2 * n_img * 2 * (Dp / 8) * h * w * 16;       // (image, side) x (hi, lo) planes
}

extern "C" int sf_corr_build_blocked32(const float* f1, const float* f2, int64_t f_clip_stride, int64_t f_pair_stride, void* vol,
                                       int64_t vol_img_stride_bytes, int B, int pairs, int D, int h, int w, void* ws, int64_t ws_bytes,
                                       void* stream) {
    SF_REQUIRE(f1 && f2 && vol && ws, "sf_corr_build_blocked32: null pointer");
    SF_REQUIRE(B > 0 && pairs > 0 && D > 0 && h > 0 && w > 0, "sf_corr_build_blocked32: bad dims");
    SF_REQUIRE((h >> 3) >= 1 && (w >> 3) >= 1, "sf_corr_build_blocked32: feature grid %dx%d too small for 4 levels", h, w);
    const int n_img = B * pairs;
    SF_REQUIRE(n_img <= 32767, "sf_corr_build_blocked32: B*pairs too large");
    Build32Args g;
    g.g = make_geom32(h, w);
    g.N = h * w; g.h = h; g.w = w; g.n_img = n_img;
    g.Dp = sf::ceil_div(D, 32) * 32;
    SF_REQUIRE((int64_t)g.Dp * g.N * 4 < ((int64_t)1 << 31), "sf_corr_build_blocked32: feature image larger than 2 GiB");
    SF_REQUIRE((int64_t)64 * g.g.rec < ((int64_t)1 << 31), "sf_corr_build_blocked32: feature grid %dx%d too large", h, w);
    SF_REQUIRE(ws_bytes >= sf_corr_build_blocked32_ws_bytes(n_img, D, h, w) && (reinterpret_cast<uintptr_t>(ws) & 15) == 0,
               "sf_corr_build_blocked32: needs a 16-byte aligned workspace of sf_corr_build_blocked32_ws_bytes() bytes");
    SF_REQUIRE((reinterpret_cast<uintptr_t>(vol) & 127) == 0 && (vol_img_stride_bytes & 127) == 0 &&
                   vol_img_stride_bytes >= (int64_t)src_rows_padded(g.N) * g.g.rec,
               "sf_corr_build_blocked32: vol must be 128-byte aligned, image stride a multiple of 128 and at least sf_corr_blocked32_bytes(1, h, w)");
    g.vol = static_cast<char*>(vol);
    g.vol_img_stride = vol_img_stride_bytes;
    g.pcols = sf::ceil_div(w, PC);
    g.np = g.pcols * sf::ceil_div(h, PR);
    g.mt = sf::ceil_div(g.N, BM);
    g.scale = 1.0f / sqrtf((float)D);
    const int patch_bytes = BN * g.Dp * 4;                       // patches per L2-resident block: ~1.75 MB of packed target features
    g.pblk = (7 << 18) / patch_bytes;
    g.pblk = g.pblk < 1 ? 1 : (g.pblk > g.np ? g.np : g.pblk);
    const int64_t n_wg = (int64_t)g.np * g.mt * n_img;
    SF_REQUIRE(n_wg < ((int64_t)1 << 31), "sf_corr_build_blocked32: grid too large");
    hipLaunchKernelGGL(split_pack32_kernel, dim3(sf::ceil_div(g.N, 256), g.Dp / 8, 2 * n_img), dim3(256), 0, (hipStream_t)stream, f1, f2,
                       f_clip_stride, f_pair_stride, (char*)ws, pairs, D, g.Dp, g.N);
    hipLaunchKernelGGL(corr_build_blocked32_kernel, dim3((unsigned)n_wg), dim3(kThreads), 0, (hipStream_t)stream, g, (const char*)ws);
    return sf::check_launch("sf_corr_build_blocked32");
}

extern "C" int sf_corr_lookup_blocked32(const void* vol, int64_t vol_img_stride_bytes, const float* coords, float* out, int64_t out_img_stride,
                                        int B, int pairs, int h, int w, void* stream) {
    SF_REQUIRE(vol && coords && out, "sf_corr_lookup_blocked32: null pointer");
    SF_REQUIRE(B > 0 && pairs > 0 && h > 0 && w > 0, "sf_corr_lookup_blocked32: bad dims");
    SF_REQUIRE((h >> 3) >= 1 && (w >> 3) >= 1, "sf_corr_lookup_blocked32: feature grid %dx%d too small for 4 levels", h, w);
    SF_REQUIRE((int64_t)B * pairs <= 65535, "sf_corr_lookup_blocked32: B*pairs too large");
    Look32Args a;
    a.g = make_geom32(h, w);
    SF_REQUIRE((int64_t)LP * a.g.rec < ((int64_t)1 << 31), "sf_corr_lookup_blocked32: feature grid %dx%d too large", h, w);
    SF_REQUIRE((reinterpret_cast<uintptr_t>(vol) & 15) == 0 && (vol_img_stride_bytes & 15) == 0,
               "sf_corr_lookup_blocked32: vol and its image stride must be 16-byte aligned");
    a.vol = static_cast<const char*>(vol);
    a.vol_img_stride = vol_img_stride_bytes;
    a.coords = coords;
    a.out = out; a.out_img_stride = out_img_stride;
    a.h = h; a.w = w; a.N = h * w;
    const dim3 grid(sf::ceil_div(a.N, LP), B * pairs);
    if ((int64_t)B * pairs * vol_img_stride_bytes >= SF_LOOK32_STREAM_BYTES)
        hipLaunchKernelGGL(corr_lookup_blocked32_kernel<2>, grid, dim3(kLookThreads), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(corr_lookup_blocked32_kernel<0>, grid, dim3(kLookThreads), 0, (hipStream_t)stream, a);
    return sf::check_launch("sf_corr_lookup_blocked32");
}
