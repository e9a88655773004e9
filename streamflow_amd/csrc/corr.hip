// All-pairs correlation volume + 4-level pyramid (one pass) and the radius-4 pyramid lookup.
//
// Reference: core/corr.py:7-21,46-54 (build), :23-44 (lookup), core/utils/utils.py:65-79 (sampler).
//
// BUILD.  C[i][j] = <f1[:,i], f2[:,j]> / sqrt(D) is a dense contraction, so it runs on the matrix
// cores (exact-fp32 v_mfma_f32_32x32x2_f32, or split fp16 operands fed by LDS-DMA: see corr_build_dma_kernel).  The GEMM "N" tile is not a run of 256 consecutive
// targets but an 8-row x 32-column PATCH of the target image: every 2x2, 4x4 and 8x8 pooling block of
// levels 1..3 then lies inside one wave's accumulators, so the three avg-pool levels are produced in
// the epilogue (vertical pairs = different accumulators, horizontal pairs = DPP lane shifts) and every pyramid cell is written exactly once and never re-read
// (algorithmic bytes: 2*N*D*4 feature reads + N*cells*4 writes).
// Stores are 128-byte runs (32 lanes x consecutive x) for level 0.
//
// LOOKUP.  HBM-bound gather.  A workgroup owns 32 consecutive source pixels of one pyramid level:
// it gathers their 10x10 bilinear footprints into LDS (lanes walk the 40-byte footprint rows), then
// emits the 81 taps per pixel with lanes running over PIXELS, so every output store is a 128-byte
// run inside one of the 324 channel planes (NCHW output, no transposing copy as in the reference).
#include "sf_common.h"
#include "split_operand.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kThreads = 256;

// ------------------------------------------------------------------------------------------------
// build
// ------------------------------------------------------------------------------------------------
constexpr int BM = 128;          // source pixels per workgroup (4 waves x 32)
constexpr int PR = 8, PC = 32;   // target patch rows / cols
constexpr int BN = PR * PC;      // 256
constexpr int BK = 16;
constexpr int SA = BM + 4, SB = BN + 4;

struct BuildArgs {
    const float* f1; const float* f2;
    float* lvl[4];
    int64_t f_clip_stride, f_pair_stride;
    int64_t lvl_pair_stride[4];
    int B, pairs, D, h, w, N;
    int hl[4], wl[4];
    int pcols;
    float scale;
    int vec_a, vec_b;
    // DMA kernels: 1-D grid, XCD-aware order [image][patch block][m-tile][patch in block]: the `pblk` target patches of
    // a block stay in the XCD's L2 while all m-tiles stream past them (np = patches per image, mt = m-tiles)
    int np, mt, pblk;
    int vec_store;               // w % 4 == 0 and 16-byte aligned level bases: transposed 16-byte epilogue stores
#ifdef SF_CORR_TIMERS
    long long* ts;               // SF_CORR_TS_BUF: per-workgroup phase timestamps (tools/corr_one.py)
#endif
};

struct TileId { int img, m_tile, patch; };
__device__ __forceinline__ TileId build_tile(const BuildArgs& g, int id) {
    const int per_img = g.np * g.mt;
    TileId t;
    t.img = id / per_img;
    int r = id % per_img;
    const int full = g.np / g.pblk;                        // full patch blocks
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

// Store side of the epilogue.  One v_mul + one buffer_store per level-0 cell: the wave's 32 source rows are addressed
// from a per-wave buffer resource (32-bit offsets; the row / patch-row part rides in the scalar offset), horizontal
// pooling partners come over DPP instead of ds_bpermute, and each pooled level is stored under one exec-mask region
// per 4-register group.  kGuard = false: interior tile, nothing to check.  kGuard = true (tile touches the image
// border or the end of the source pixels): a cell that must not be written gets an out-of-range offset and the
// buffer unit drops the store -- one v_cndmask per store instead of a branch.  (A branchy, 64-bit-addressed version
// of this epilogue cost ~33k cycles per tile, more than the tile's whole MFMA loop.)
__device__ __forceinline__ float dpp_xor1(float v) {     // lane ^ 1 (quad_perm [1,0,3,2])
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float v) {     // lane ^ 2 (quad_perm [2,3,0,1])
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_shl4(float v) {     // lane + 4 inside a row of 16 (row_shl:4)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x104, 0xF, 0xF, true));
}

// one volume cell through a buffer store: fp32 (dword) or IEEE fp16 (round to nearest, short store)
template <typename OutT, int kAux>
__device__ __forceinline__ void store_cell(float v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    if constexpr (sizeof(OutT) == 4) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, kAux);
    } else {
        const _Float16 hv = (_Float16)v;
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hv), r, voff, soff, kAux);
    }
}

template <bool kGuard, typename OutT>
__device__ __forceinline__ void pyramid_store(const BuildArgs& g, f32x16 (&acc)[PR], int b, int pair, int m0, int wave,
                                              int py0, int px0, int lane) {
    constexpr int ES = sizeof(OutT);
    OutT* const lv0 = reinterpret_cast<OutT*>(g.lvl[0]);
    OutT* const lv1 = reinterpret_cast<OutT*>(g.lvl[1]);
    OutT* const lv2 = reinterpret_cast<OutT*>(g.lvl[2]);
    OutT* const lv3 = reinterpret_cast<OutT*>(g.lvl[3]);
    const int khalf = lane >> 5, l31 = lane & 31;
    const int P0 = g.hl[0] * g.wl[0], P1 = g.hl[1] * g.wl[1], P2 = g.hl[2] * g.wl[2], P3 = g.hl[3] * g.wl[3];
    const int i0 = m0 + wave * 32;                          // first source pixel of the wave
    const int64_t row0 = (int64_t)b * g.N + i0;
    constexpr int kSpan = 0x7ffffff0, kDrop = (int)0x80000000u;
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(
        lv0 + pair * g.lvl_pair_stride[0] + row0 * P0 + py0 * g.wl[0] + px0, 0, kSpan, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
        lv1 + pair * g.lvl_pair_stride[1] + row0 * P1 + (py0 >> 1) * g.wl[1] + (px0 >> 1), 0, kSpan, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(
        lv2 + pair * g.lvl_pair_stride[2] + row0 * P2 + (py0 >> 2) * g.wl[2] + (px0 >> 2), 0, kSpan, 0x00020000);
    const __amdgpu_buffer_rsrc_t r3 = __builtin_amdgcn_make_buffer_rsrc(
        lv3 + pair * g.lvl_pair_stride[3] + row0 * P3 + (py0 >> 3) * g.wl[3] + (px0 >> 3), 0, kSpan, 0x00020000);
    const int x = px0 + l31;
    // per-lane offsets; a lane whose column lies outside the level is dropped for good
    const int vo0 = (!kGuard || x < g.wl[0]) ? (4 * khalf * P0 + l31) * ES : kDrop;
    const int vo1 = (!kGuard || (x >> 1) < g.wl[1]) ? (4 * khalf * P1 + (l31 >> 1)) * ES : kDrop;
    const int vo2 = (!kGuard || (x >> 2) < g.wl[2]) ? (4 * khalf * P2 + (l31 >> 2)) * ES : kDrop;
    const int vo3 = (!kGuard || (x >> 3) < g.wl[3]) ? (4 * khalf * P3 + (l31 >> 3)) * ES : kDrop;
    // level 0 (3/4 of the bytes) is streamed past the caches: it would only displace the operand tiles in L2 and the
    // pooled levels, which are small enough (a quarter of level 0 together) to stay in the 256 MB Infinity Cache
    // for the lookups that follow
#ifndef SF_CORR_NT
#define SF_CORR_NT 2
#endif
#ifndef SF_CORR_PACK
#define SF_CORR_PACK 1      // fp16 level-0 cells leave as packed pairs (A/B knob)
#endif
    constexpr int kNt = SF_CORR_NT;
    // Store-instruction diet (the epilogue is bound by store ISSUE, not bytes: 240 instructions per wave in the plain
    // form).  (a) After the DPP pooling the 2 / 4 / 4 lanes of a pooling group hold the SAME level-1 / 2 / 3 value, so
    // lane k of a group stores source row ri + k: one instruction covers 2 / 4 / 4 source rows (112 -> 44).  (b) fp16
    // cells, even width: lanes 2j and 2j+1 exchange one value over DPP so that the even lane owns columns (2j, 2j+1) of
    // source row ri and the odd lane the same two columns of row ri + 1: one 4-byte store per lane and register PAIR
    // (128 -> 64), same 64-byte runs.  Values and roundings are unchanged.
    const bool pack0 = (ES == 2) && (g.wl[0] & 1) == 0 && SF_CORR_PACK;
    const int odd = l31 & 1;
    const int vo0p = (!kGuard || x < g.wl[0]) ? ((4 * khalf + odd) * P0 + (l31 & ~1)) * ES : kDrop;
    const int k1 = l31 & 1, k2 = l31 & 3, k3 = l31 & 7;       // which source row of a group this lane stores (levels 1..3)
    const int vo1r = (vo1 == kDrop) ? kDrop : vo1 + k1 * P1 * ES;
    const int vo2r = (vo2 == kDrop) ? kDrop : vo2 + k2 * P2 * ES;
    const int vo3r = (vo3 == kDrop || k3 > 3) ? kDrop : vo3 + k3 * P3 * ES;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {                        // register group: source pixels 8*rq + 4*khalf + (0..3)
        // the level-2 / level-3 value (and row guard) THIS lane will store: picked up as the rows go by (a select per
        // value; an indexed pick at the end would put the per-row arrays into scratch memory)
        float sel2[PR / 4], sel3 = 0.f;
        bool rok2 = false, rok3 = false;
#pragma unroll
        for (int t = 0; t < PR / 4; ++t) sel2[t] = 0.f;
        bool iok[4];
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) iok[ri] = !kGuard || (i0 + ri + 8 * rq + 4 * khalf < g.N);
#pragma unroll
        for (int j = 0; j < 2; ++j) {                       // source-row pair (2j, 2j + 1) of the group
            const int ri = 2 * j;
            float v0[2][PR], sel1[PR / 2];
            bool rok1 = false;
#pragma unroll
            for (int t = 0; t < PR / 2; ++t) sel1[t] = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < PR; ++t) v0[u][t] = acc[t][rq * 4 + ri + u] * g.scale;
            if (pack0) {                                    // wave-uniform
                if constexpr (ES == 2) {
                    const bool rok = odd ? iok[ri + 1] : iok[ri];
#pragma unroll
                    for (int t = 0; t < PR; ++t) {
                        const float a = v0[0][t], c = v0[1][t];
                        const float y = dpp_xor1(odd ? a : c);      // even lane: a of lane + 1; odd lane: c of lane - 1
                        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                        h2 hv;
                        hv[0] = (_Float16)(odd ? y : a);
                        hv[1] = (_Float16)(odd ? c : y);
                        const bool ok = !kGuard || (rok && py0 + t < g.hl[0]);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hv), r0, ok ? vo0p : kDrop,
                                                              ((ri + 8 * rq) * P0 + t * g.wl[0]) * ES, kNt);
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int t = 0; t < PR; ++t) {
                        const bool ok = !kGuard || (iok[ri + u] && py0 + t < g.hl[0]);
                        store_cell<OutT, kNt>(v0[u][t], r0, ok ? vo0 : kDrop, ((ri + u + 8 * rq) * P0 + t * g.wl[0]) * ES);
                    }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float v1[PR / 2];
#pragma unroll
                for (int t = 0; t < PR / 2; ++t) {
                    const float s = v0[u][2 * t] + v0[u][2 * t + 1];
                    v1[t] = 0.25f * (s + dpp_xor1(s));
                    sel1[t] = (k1 == u) ? v1[t] : sel1[t];
                }
                rok1 = (k1 == u) ? iok[ri + u] : rok1;
                float v2[PR / 4];
#pragma unroll
                for (int t = 0; t < PR / 4; ++t) {
                    const float s = v1[2 * t] + v1[2 * t + 1];
                    v2[t] = 0.25f * (s + dpp_xor2(s));
                    sel2[t] = (k2 == ri + u) ? v2[t] : sel2[t];
                }
                const float s = v2[0] + v2[1];
                const float v3 = 0.25f * (s + dpp_shl4(s));
                sel3 = (k3 == ri + u) ? v3 : sel3;
                rok2 = (k2 == ri + u) ? iok[ri + u] : rok2;
                rok3 = (k3 == ri + u) ? iok[ri + u] : rok3;
            }
            // level 1: lane parity k1 stores source row 2j + k1
#pragma unroll
            for (int t = 0; t < PR / 2; ++t) {
                const bool ok = !kGuard || (rok1 && (py0 >> 1) + t < g.hl[1]);
                store_cell<OutT, 0>(sel1[t], r1, ok ? vo1r : kDrop, ((ri + 8 * rq) * P1 + t * g.wl[1]) * ES);
            }
        }
        // level 2 / 3: lane k of a group of four stores source row k
#pragma unroll
        for (int t = 0; t < PR / 4; ++t) {
            const bool ok = !kGuard || (rok2 && (py0 >> 2) + t < g.hl[2]);
            store_cell<OutT, 0>(sel2[t], r2, ok ? vo2r : kDrop, ((8 * rq) * P2 + t * g.wl[2]) * ES);
        }
        {
            const bool ok = !kGuard || (rok3 && (py0 >> 3) < g.hl[3]);
            store_cell<OutT, 0>(sel3, r3, ok ? vo3r : kDrop, ((8 * rq) * P3) * ES);
        }
    }
}

// ---- vector store side (w % 4 == 0): 16-byte stores after a transpose through LDS -------------------------------------
// The dword path above issues 170 store instructions per wave (256 bytes each) and the build is bound by exactly that:
// store ISSUE, not bandwidth (2.8 TB/s of writes at best).  Here every pair of patch rows (two 32x32 accumulator
// tiles) goes through a per-wave LDS scratch (row stride 36 floats: conflict-free dword writes, 16-byte aligned rows)
// and comes back with a lane owning FOUR consecutive target columns of one source row (row = lane/8 + 8q, columns
// 4*(lane%8)..+3): level 0 leaves as 16-byte (fp32) / 8-byte (fp16) stores, 1 KB per instruction, and the horizontal
// pooling partners of levels 1 and 2 sit in the same lane (no DPP); only level 3 needs one lane exchange.
// 60 store instructions per wave instead of 170.  Needs w % 4 == 0 (every configuration of BASELINE.json); odd widths
// take the dword path.
constexpr int kTrStride = 36;
constexpr int kTrTile = 32 * kTrStride;                   // floats per transposed tile
typedef unsigned int st_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int st_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_h2(float a, float b) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 v;
    v[0] = (_Float16)a;
    v[1] = (_Float16)b;
    return __builtin_bit_cast(unsigned, v);
}

template <bool kGuard, typename OutT>
__device__ __forceinline__ void pyramid_store_vec(const BuildArgs& g, f32x16 (&acc)[PR], int b, int pair, int m0, int wave,
                                                  int py0, int px0, int lane, float* scratch) {
    constexpr int ES = sizeof(OutT);
    OutT* const lv0 = reinterpret_cast<OutT*>(g.lvl[0]);
    OutT* const lv1 = reinterpret_cast<OutT*>(g.lvl[1]);
    OutT* const lv2 = reinterpret_cast<OutT*>(g.lvl[2]);
    OutT* const lv3 = reinterpret_cast<OutT*>(g.lvl[3]);
    const int khalf = lane >> 5, l31 = lane & 31;
    const int rrow = lane >> 3, xq = (lane & 7) * 4;        // read-back coordinates: source row rrow + 8q, columns xq..xq+3
    const int P0 = g.hl[0] * g.wl[0], P1 = g.hl[1] * g.wl[1], P2 = g.hl[2] * g.wl[2], P3 = g.hl[3] * g.wl[3];
    const int i0 = m0 + wave * 32;
    const int64_t row0 = (int64_t)b * g.N + i0;
    constexpr int kSpan = 0x7ffffff0, kDrop = (int)0x80000000u;
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(
        lv0 + pair * g.lvl_pair_stride[0] + row0 * P0 + py0 * g.wl[0] + px0, 0, kSpan, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
        lv1 + pair * g.lvl_pair_stride[1] + row0 * P1 + (py0 >> 1) * g.wl[1] + (px0 >> 1), 0, kSpan, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(
        lv2 + pair * g.lvl_pair_stride[2] + row0 * P2 + (py0 >> 2) * g.wl[2] + (px0 >> 2), 0, kSpan, 0x00020000);
    const __amdgpu_buffer_rsrc_t r3 = __builtin_amdgcn_make_buffer_rsrc(
        lv3 + pair * g.lvl_pair_stride[3] + row0 * P3 + (py0 >> 3) * g.wl[3] + (px0 >> 3), 0, kSpan, 0x00020000);
    const int x = px0 + xq;                                 // first of this lane's four columns (w % 4 == 0: all in or all out)
    const int co0 = (!kGuard || x < g.wl[0]) ? xq * ES : kDrop;
    const int co1 = (!kGuard || (x >> 1) < g.wl[1]) ? (xq >> 1) * ES : kDrop;
    const int co2 = (!kGuard || (x >> 2) < g.wl[2]) ? (xq >> 2) * ES : kDrop;
    const int co3 = (!kGuard || (x >> 3) < g.wl[3]) ? (xq >> 3) * ES : kDrop;
#ifndef SF_CORR_NT
#define SF_CORR_NT 2
#endif
#ifndef SF_CORR_PACK
#define SF_CORR_PACK 1      // fp16 level-0 cells leave as packed pairs (A/B knob)
#endif
    constexpr int kNt = SF_CORR_NT;
    float l2s[PR / 4][4];                                       // level-2 values (one per lane and source row)
#pragma unroll
    for (int t2 = 0; t2 < PR / 4; ++t2) {
        float l1s[4];                                           // sum of the two level-1 values of the first row pair
#pragma unroll
        for (int u2 = 0; u2 < 2; ++u2) {
            const int tp = 2 * t2 + u2;
            // ---- transpose patch rows 2tp, 2tp+1 (a wave's DS operations execute in order: no barrier) ----
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    scratch[u * kTrTile + ((r & 3) + 8 * (r >> 2) + 4 * khalf) * kTrStride + l31] = acc[2 * tp + u][r] * g.scale;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rr = rrow + 8 * q;
                const bool iok = !kGuard || (i0 + rr < g.N);
                const float4 a = *reinterpret_cast<const float4*>(scratch + rr * kTrStride + xq);
                const float4 c = *reinterpret_cast<const float4*>(scratch + kTrTile + rr * kTrStride + xq);
                const bool ok0 = !kGuard || (iok && py0 + 2 * tp < g.hl[0]);
                const bool ok1 = !kGuard || (iok && py0 + 2 * tp + 1 < g.hl[0]);
                const int so = (rr * P0 + 2 * tp * g.wl[0]) * ES;
                if constexpr (ES == 4) {
                    st_u32x4 va, vc;
                    va[0] = __builtin_bit_cast(unsigned, a.x); va[1] = __builtin_bit_cast(unsigned, a.y);
                    va[2] = __builtin_bit_cast(unsigned, a.z); va[3] = __builtin_bit_cast(unsigned, a.w);
                    vc[0] = __builtin_bit_cast(unsigned, c.x); vc[1] = __builtin_bit_cast(unsigned, c.y);
                    vc[2] = __builtin_bit_cast(unsigned, c.z); vc[3] = __builtin_bit_cast(unsigned, c.w);
                    __builtin_amdgcn_raw_buffer_store_b128(va, r0, ok0 ? co0 : kDrop, so, kNt);
                    __builtin_amdgcn_raw_buffer_store_b128(vc, r0, ok1 ? co0 : kDrop, so + g.wl[0] * ES, kNt);
                } else {
                    st_u32x2 va, vc;
                    va[0] = pack_h2(a.x, a.y); va[1] = pack_h2(a.z, a.w);
                    vc[0] = pack_h2(c.x, c.y); vc[1] = pack_h2(c.z, c.w);
                    __builtin_amdgcn_raw_buffer_store_b64(va, r0, ok0 ? co0 : kDrop, so, kNt);
                    __builtin_amdgcn_raw_buffer_store_b64(vc, r0, ok1 ? co0 : kDrop, so + g.wl[0] * ES, kNt);
                }
                // level 1: 2x2 means of (patch rows 2tp, 2tp+1) x (columns 0-1, 2-3)
                const float p0 = 0.25f * ((a.x + c.x) + (a.y + c.y)), p1 = 0.25f * ((a.z + c.z) + (a.w + c.w));
                const bool ok = !kGuard || (iok && (py0 >> 1) + tp < g.hl[1]);
                const int s1 = (rr * P1 + tp * g.wl[1]) * ES;
                if constexpr (ES == 4) {
                    st_u32x2 v;
                    v[0] = __builtin_bit_cast(unsigned, p0);
                    v[1] = __builtin_bit_cast(unsigned, p1);
                    __builtin_amdgcn_raw_buffer_store_b64(v, r1, ok ? co1 : kDrop, s1, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32(pack_h2(p0, p1), r1, ok ? co1 : kDrop, s1, 0);
                }
                // level 2: mean of the four level-1 cells of patch rows 4 t2 .. 4 t2 + 3
                if (u2 == 0) l1s[q] = p0 + p1;
                else {
                    l2s[t2][q] = 0.25f * (l1s[q] + (p0 + p1));
                    const bool ok2 = !kGuard || (iok && (py0 >> 2) + t2 < g.hl[2]);
                    store_cell<OutT, 0>(l2s[t2][q], r2, ok2 ? co2 : kDrop, (rr * P2 + t2 * g.wl[2]) * ES);
                }
            }
        }
    }
    // ---- level 3: the two level-2 rows of this lane and of the neighbouring column quad ----
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rr = rrow + 8 * q;
        const float s3 = l2s[0][q] + l2s[1][q];
        const float l3 = 0.25f * (s3 + dpp_xor1(s3));
        const bool ok = (lane & 1) == 0 && (!kGuard || (i0 + rr < g.N && (py0 >> 3) < g.hl[3]));
        store_cell<OutT, 0>(l3, r3, ok ? co3 : kDrop, (rr * P3) * ES);
    }
}

// scratch: per-wave LDS region of 2 * kTrTile floats (the main-loop buffers, after a workgroup barrier), or nullptr.
template <typename OutT = float, bool kVec = false>
__device__ __forceinline__ void pyramid_epilogue(const BuildArgs& g, f32x16 (&acc)[PR], int b, int pair, int m0, int wave,
                                                 int py0, int px0, int lane, float* scratch = nullptr) {
    const bool interior = m0 + BM <= g.N && py0 + PR <= g.hl[0] && px0 + PC <= g.wl[0] &&
                          (py0 >> 1) + PR / 2 <= g.hl[1] && (py0 >> 2) + PR / 4 <= g.hl[2] && (py0 >> 3) < g.hl[3] &&
                          ((px0 + PC) >> 1) <= g.wl[1] && ((px0 + PC) >> 2) <= g.wl[2] && ((px0 + PC) >> 3) <= g.wl[3];
    if constexpr (kVec) {                    // one flavour per kernel instantiation: both inlined cost registers (spills)
        if (interior) pyramid_store_vec<false, OutT>(g, acc, b, pair, m0, wave, py0, px0, lane, scratch);
        else pyramid_store_vec<true, OutT>(g, acc, b, pair, m0, wave, py0, px0, lane, scratch);
    } else {
        if (interior) pyramid_store<false, OutT>(g, acc, b, pair, m0, wave, py0, px0, lane);
        else pyramid_store<true, OutT>(g, acc, b, pair, m0, wave, py0, px0, lane);
    }
}

__global__ __launch_bounds__(kThreads, 2) void corr_build_kernel(const BuildArgs g) {
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (SA + SB)];
    float* sA = smem;
    float* sB = smem + 2 * BK * SA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int khalf = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z / g.pairs, pair = blockIdx.z % g.pairs;
    const int m0 = blockIdx.y * BM;
    const int py0 = (blockIdx.x / g.pcols) * PR, px0 = (blockIdx.x % g.pcols) * PC;
    const float* A = g.f1 + (int64_t)b * g.f_clip_stride + (int64_t)pair * g.f_pair_stride;
    const float* Bm = g.f2 + (int64_t)b * g.f_clip_stride + (int64_t)pair * g.f_pair_stride;

    float4 ra[2], rb[4];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {                       // A: [BK][128] -> 512 float4
            const int idx = tid + j * kThreads;
            const int k = k0 + idx / 32, m = m0 + (idx % 32) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < g.D) {
                const float* p = A + (int64_t)k * g.N + m;
                if (g.vec_a && m + 3 < g.N) v = *reinterpret_cast<const float4*>(p);
                else {
                    if (m + 0 < g.N) v.x = p[0];
                    if (m + 1 < g.N) v.y = p[1];
                    if (m + 2 < g.N) v.z = p[2];
                    if (m + 3 < g.N) v.w = p[3];
                }
            }
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {                       // B: [BK][8 rows x 32 cols] -> 1024 float4
            const int idx = tid + j * kThreads;
            const int k = k0 + idx / 64, n = (idx % 64) * 4;
            const int y = py0 + n / PC, x = px0 + n % PC;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < g.D && y < g.h) {
                const float* p = Bm + (int64_t)k * g.N + y * g.w + x;
                if (g.vec_b && x + 3 < g.w) v = *reinterpret_cast<const float4*>(p);
                else {
                    if (x + 0 < g.w) v.x = p[0];
                    if (x + 1 < g.w) v.y = p[1];
                    if (x + 2 < g.w) v.z = p[2];
                    if (x + 3 < g.w) v.w = p[3];
                }
            }
            rb[j] = v;
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + j * kThreads;
            *reinterpret_cast<float4*>(sA + buf * BK * SA + (idx / 32) * SA + (idx % 32) * 4) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = tid + j * kThreads;
            *reinterpret_cast<float4*>(sB + buf * BK * SB + (idx / 64) * SB + (idx % 64) * 4) = rb[j];
        }
    };

    f32x16 acc[PR];
#pragma unroll
    for (int t = 0; t < PR; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nk = (g.D + BK - 1) / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tiles((kt + 1) * BK);
        const float* pa = sA + cur * BK * SA + wave * 32 + l31;
        const float* pb = sB + cur * BK * SB + l31;
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const int kk = 2 * ks + khalf;
            const float a = pa[kk * SA];
            float bv[PR];
#pragma unroll
            for (int t = 0; t < PR; ++t) bv[t] = pb[kk * SB + t * PC];
#pragma unroll
            for (int t = 0; t < PR; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[t], acc[t], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
    }

    pyramid_epilogue(g, acc, b, pair, m0, wave, py0, px0, lane);
}

// ---- split-precision (f16x3) build --------------------------------------------------------------------------
// Operands: f = hi + lo, two IEEE fp16 planes, products ah*bh + ah*bl + al*bh on v_mfma_f32_32x32x16_f16 with
// fp32 accumulation (~2^-22 relative, see gemm_split.hip).  Splitting costs ~5 VALU per element, so it is done
// ONCE per feature image by split_pack_kernel into a workspace laid out for the matrix cores:
//     ws image = [hi | lo],   plane[(k/8)*N + pixel][k%8]      (16 bytes = one MFMA operand k-octet of one pixel)
// The build kernel then moves operand tiles HBM/L2 -> LDS with buffer_load_dwordx4 ... lds (no VGPR staging, no
// conversion, no ds_write): 6 DMA instructions per thread per 16-deep k-step against 24 MFMAs, and the LDS image is
// lane-linear in exactly the order ds_read_b128 wants it (consecutive lanes = consecutive pixels = consecutive
// 16-byte slots, conflict-free).  Two 24 KB stages; two workgroups per CU so that one workgroup's pyramid
// epilogue (170 stores per wave) overlaps the other's MFMA loop.
constexpr int DK = 16;                              // k per stage
constexpr int ST_A = (DK / 8) * BM * 16;            // bytes of the A_hi (= A_lo) part of a stage
constexpr int ST_B = (DK / 8) * BN * 16;            // bytes of B_hi (= B_lo)
constexpr int STAGE = 2 * ST_A + 2 * ST_B;          // 24576
#ifndef SF_CORR_NSTAGE
#define SF_CORR_NSTAGE 2
#endif
constexpr int NSTAGE = SF_CORR_NSTAGE;                           // two stages in flight behind the one being consumed

__global__ __launch_bounds__(256) void split_pack_kernel(const float* f1, const float* f2, int64_t f_clip_stride,
                                                         int64_t f_pair_stride, char* ws, int pairs, int D, int Dp,
                                                         int N) {
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

typedef __attribute__((address_space(3))) void* lds_ptr;

#ifndef SF_CORR_VEC_WAVES
#define SF_CORR_VEC_WAVES 3      // waves per SIMD the kVec = true forms are compiled for (3: 168 registers, 6-7 of them spilled; 2: none)
#endif
template <bool kVec>
__global__ __launch_bounds__(kThreads, NSTAGE == 2 ? (kVec ? SF_CORR_VEC_WAVES : 3) : 2) void corr_build_dma_kernel(const BuildArgs g, const char* ws, int Dp) {
    using namespace sf_split;
    __shared__ __attribute__((aligned(1024))) char smem[NSTAGE * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, l31 = lane & 31;
    const TileId tile = build_tile(g, sf::xcd_linear_id(blockIdx.x, gridDim.x));
    const int b = tile.img / g.pairs, pair = tile.img % g.pairs;
    const int m0 = tile.m_tile * BM;
    const int py0 = (tile.patch / g.pcols) * PR, px0 = (tile.patch % g.pcols) * PC;
    const int half = (Dp / 8) * g.N * 16;                       // bytes of one hi (or lo) plane (< 2 GiB, host-checked)
    const char* imgA = ws + (int64_t)(tile.img * 2 + 0) * 2 * half;
    const char* imgB = ws + (int64_t)(tile.img * 2 + 1) * 2 * half;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(imgA), 0, 2 * half, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(imgB), 0, 2 * half, 0x00020000);
    // per-thread source offsets (pixels / patch cells past the image are clamped: their products are never stored)
    const int voa = ((tid >> 7) * g.N + min(m0 + (tid & 127), g.N - 1)) * 16;          // slot = kq*128 + px = tid
    const int vob = (min(py0 + tid / PC, g.h - 1) * g.w + min(px0 + tid % PC, g.w - 1)) * 16;   // slot = kq*256 + cell
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

    f32x16 acc[PR];
#pragma unroll
    for (int t = 0; t < PR; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nk = Dp / DK;
    const int offa = (khalf * BM + wave * 32 + l31) * 16;
    const int offb = 2 * ST_A + (khalf * BN + l31) * 16;
    issue(0, 0);
    if (NSTAGE > 2 && nk > 1) issue(1, 1);
    int cur = 0, nxt = NSTAGE - 1;                                        // ring slots of stage kt and stage kt + 2
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's 6 pieces of stage kt have landed (the 6 of stage kt+1 may still be in flight) ...
        if (NSTAGE > 2 && kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // ... everyone's; and slot nxt (stage kt-1) is no longer read
        if (kt + NSTAGE - 1 < nk) issue(kt + NSTAGE - 1, nxt);
        const char* sb = smem + cur * STAGE;
        cur = (cur == NSTAGE - 1) ? 0 : cur + 1;
        nxt = (nxt == NSTAGE - 1) ? 0 : nxt + 1;
        const f16x8 ah = *reinterpret_cast<const f16x8*>(sb + offa);
        const f16x8 al = *reinterpret_cast<const f16x8*>(sb + offa + ST_A);
#pragma unroll
        for (int t = 0; t < PR; ++t) {
            const f16x8 bh = *reinterpret_cast<const f16x8*>(sb + offb + t * PC * 16);
            const f16x8 bl = *reinterpret_cast<const f16x8*>(sb + offb + ST_B + t * PC * 16);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
        }
    }
    if (kVec) __syncthreads();                                   // the stage buffers become the transpose scratch
    pyramid_epilogue<float, kVec>(g, acc, b, pair, m0, wave, py0, px0, lane,
                                  reinterpret_cast<float*>(smem) + wave * 2 * kTrTile);
}

// ---- fp16 build (SF_PRECISION_F16): single f16 product, fp32 accumulation, volume stored as IEEE fp16 ------------
// The "bf16/fp16 volume" configurations of BASELINE.json (configs 2 and 5): features are rounded ONCE to fp16 (round to
// nearest) by pack_f16_kernel into k-octet planes [(k/8)*N + pixel][k%8], the contraction is one
// v_mfma_f32_32x32x16_f16 per 32x32x16 block (a third of the split build's matrix work) and every pyramid cell is
// written as fp16 (half the bytes of the fp32 volume, which is what bounds this kernel).  Stage = 32-deep:
// A 4 k-octets x 128 px x 16 B = 8 KB, B 4 x 256 x 16 B = 16 KB; the same 24 KB / two-stage ring as the split build.
constexpr int FK = 32;                               // k per stage
constexpr int FST_A = (FK / 8) * BM * 16;            // 8192
constexpr int FST_B = (FK / 8) * BN * 16;            // 16384
constexpr int FSTAGE = FST_A + FST_B;

__global__ __launch_bounds__(256) void pack_f16_kernel(const float* f1, const float* f2, int64_t f_clip_stride,
                                                       int64_t f_pair_stride, char* ws, int pairs, int D, int Dp, int N) {
    using namespace sf_split;
    const int px = blockIdx.x * 256 + threadIdx.x, kq = blockIdx.y;
    const int side = blockIdx.z & 1, img = blockIdx.z >> 1;             // img = b * pairs + pair
    if (px >= N) return;
    const float* f = (side ? f2 : f1) + (int64_t)(img / pairs) * f_clip_stride + (int64_t)(img % pairs) * f_pair_stride;
    f16x8 h;
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (_Float16)((kq * 8 + i < D) ? f[(int64_t)(kq * 8 + i) * N + px] : 0.f);
    const int64_t plane = (int64_t)(Dp / 8) * N * 16;
    *reinterpret_cast<f16x8*>(ws + (int64_t)blockIdx.z * plane + ((int64_t)kq * N + px) * 16) = h;
}

template <bool kVec>
__global__ __launch_bounds__(kThreads, kVec ? SF_CORR_VEC_WAVES : 3) void corr_build_f16_kernel(const BuildArgs g, const char* ws, int Dp) {
    using namespace sf_split;
    __shared__ __attribute__((aligned(1024))) char smem[2 * FSTAGE];
#ifdef SF_CORR_TIMERS
    const long long ts0 = __builtin_readcyclecounter();
    const long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, l31 = lane & 31;
    const TileId tile = build_tile(g, sf::xcd_linear_id(blockIdx.x, gridDim.x));
    const int b = tile.img / g.pairs, pair = tile.img % g.pairs;
    const int m0 = tile.m_tile * BM;
    const int py0 = (tile.patch / g.pcols) * PR, px0 = (tile.patch % g.pcols) * PC;
    const int plane = (Dp / 8) * g.N * 16;                      // bytes of one packed image (< 2 GiB, host-checked)
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(ws) + (int64_t)(tile.img * 2 + 0) * plane, 0, plane, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(ws) + (int64_t)(tile.img * 2 + 1) * plane, 0, plane, 0x00020000);
    // A slot = kq*128 + px: a piece of 256 threads covers two k-octets; B slot = kq*256 + cell: one k-octet per piece.
    // Pixels / patch cells past the image are clamped: their products are never stored.
    const int voa = ((tid >> 7) * g.N + min(m0 + (tid & 127), g.N - 1)) * 16;
    const int vob = (min(py0 + tid / PC, g.h - 1) * g.w + min(px0 + tid % PC, g.w - 1)) * 16;
    const int kq_step = g.N * 16;

    auto issue = [&](int kt, int buf) {
        char* sb = smem + buf * FSTAGE + wave * 1024;
        const int so = kt * (FK / 8) * kq_step;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(sb), 16, voa, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(sb + 4096), 16, voa, so + 2 * kq_step, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(sb + FST_A + j * 4096), 16, vob, so + j * kq_step, 0, 0);
    };

    f32x16 acc[PR];
#pragma unroll
    for (int t = 0; t < PR; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nk = Dp / FK;
    const int offa = (khalf * BM + wave * 32 + l31) * 16;
    const int offb = FST_A + (khalf * BN + l31) * 16;
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's pieces of stage kt have landed ...
        __builtin_amdgcn_s_barrier();                            // ... everyone's; the other slot is no longer read
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const char* sb = smem + (kt & 1) * FSTAGE;
#pragma unroll
        for (int ks = 0; ks < FK / 16; ++ks) {
            const f16x8 a = *reinterpret_cast<const f16x8*>(sb + offa + ks * 2 * BM * 16);
#pragma unroll
            for (int t = 0; t < PR; ++t) {
                const f16x8 bv = *reinterpret_cast<const f16x8*>(sb + offb + ks * 2 * BN * 16 + t * PC * 16);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bv, acc[t], 0, 0, 0);
            }
        }
    }
#ifdef SF_CORR_TIMERS
    const long long ts1 = __builtin_readcyclecounter();
#endif
    if (kVec) __syncthreads();                                   // the stage buffers become the transpose scratch
    pyramid_epilogue<_Float16, kVec>(g, acc, b, pair, m0, wave, py0, px0, lane,
                                     reinterpret_cast<float*>(smem) + wave * 2 * kTrTile);
#ifdef SF_CORR_TIMERS
    if (g.ts && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long* d = g.ts + (int64_t)blockIdx.x * 8;
        d[0] = ts0; d[1] = ts1; d[2] = __builtin_readcyclecounter(); d[3] = rt0; d[4] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// lookup
// ------------------------------------------------------------------------------------------------
constexpr int LP = 32;             // source pixels per workgroup (2640 workgroups at Sintel shape: one full round at 11 per CU)
constexpr int RAD = 4, WIN = 2 * RAD + 1, FP = WIN + 1;   // 9 taps, 10-cell footprint
constexpr int WSTRIDE = FP * FP + 1;                       // 101: consecutive pixels -> consecutive banks

struct LookupArgs {
    const float* lvl[4];
    const float* coords;
    float* out;
    int64_t out_img_stride;
    int64_t lvl_pair_stride[4];
    int B, pairs, h, w, N;
    int hl[4], wl[4];
    int pl[4];                   // row pitch of the maps in cells (== wl for the reference's dense layout)
    _Float16* out16;             // optional fp16 k-octet copy of `out` ([41 octets][N][8] per image), fp16-volume kernel only
    int64_t out16_img_stride;    // halves
};

__global__ __launch_bounds__(kThreads) void corr_lookup_kernel(const LookupArgs g) {
    __shared__ float win[LP * WSTRIDE];
    __shared__ float sfx[LP], sfy[LP];
    __shared__ int sx0[LP], sy0[LP];
    const int tid = threadIdx.x;
    const int l = blockIdx.y, img = blockIdx.z;
    const int b = img / g.pairs, pair = img % g.pairs;
    const int p0 = blockIdx.x * LP;
    const int hl = g.hl[l], wl = g.wl[l], pl = g.pl[l];
    const float inv = 1.0f / (float)(1 << l);
    if (tid < LP) {
        const int p = p0 + tid;
        float cx = 0.f, cy = 0.f;
        if (p < g.N) {
            cx = g.coords[((int64_t)img * 2 + 0) * g.N + p] * inv;
            cy = g.coords[((int64_t)img * 2 + 1) * g.N + p] * inv;
        }
        // anything this far out samples only zero padding; also swallows NaN/inf
        if (!(cx > -1.0e6f && cx < 1.0e6f)) cx = -1.0e6f;
        if (!(cy > -1.0e6f && cy < 1.0e6f)) cy = -1.0e6f;
        const float fx0 = floorf(cx), fy0 = floorf(cy);
        sx0[tid] = (int)fx0;
        sy0[tid] = (int)fy0;
        sfx[tid] = cx - fx0;
        sfy[tid] = cy - fy0;
    }
    __syncthreads();
    const float* vol = g.lvl[l] + pair * g.lvl_pair_stride[l] + ((int64_t)b * g.N + p0) * hl * pl;
    for (int idx = tid; idx < LP * FP * FP; idx += kThreads) {
        const int pix = idx / (FP * FP), cell = idx % (FP * FP);
        const int yy = sy0[pix] - RAD + cell / FP, xx = sx0[pix] - RAD + cell % FP;
        float v = 0.f;
        if (p0 + pix < g.N && yy >= 0 && yy < hl && xx >= 0 && xx < wl)
            v = vol[(int64_t)pix * hl * pl + yy * pl + xx];
        win[pix * WSTRIDE + cell] = v;
    }
    __syncthreads();
    float* out = g.out + (int64_t)img * g.out_img_stride + (int64_t)l * WIN * WIN * g.N + p0;
    for (int idx = tid; idx < WIN * WIN * LP; idx += kThreads) {
        const int pix = idx % LP, ab = idx / LP;
        if (p0 + pix >= g.N) continue;
        const int a = ab / WIN, bb = ab % WIN;           // a moves x, bb moves y  (corr.py:31-37)
        const float fx = sfx[pix], fy = sfy[pix];
        const float* c = win + pix * WSTRIDE + bb * FP + a;
        const float v = c[0] * ((1.f - fx) * (1.f - fy)) + c[1] * (fx * (1.f - fy)) + c[FP] * ((1.f - fx) * fy) +
                        c[FP + 1] * (fx * fy);
        out[(int64_t)ab * g.N + pix] = v;
    }
}

// ---- lookup in fp16 volumes (SF_PRECISION_F16 builds) -----------------------------------------------------------
// One workgroup owns 32 consecutive source pixels for ALL four levels (coordinates read once).
//  gather: the 4 x 32 footprints (10 x 10 fp16 cells) go to LDS through a flat item index whose fastest part is the
//          cell, so the 64 lanes of a load instruction walk consecutive cells of ONE footprint (6-7 cache lines per
//          instruction; lanes over pixels would touch 64 lines); loads are clamped + selected, not branched;
//  taps:   item = (channel group of three `a`, tap row, level, pixel QUAD): a lane blends the same three taps for four
//          consecutive pixels and stores 16 bytes per channel -- 1 KB per store instruction instead of 256 B (the
//          dword version of this kernel is bound by store issue, like the build epilogue was).  Needs N % 4 == 0 and
//          16-byte aligned output planes (every BASELINE configuration); otherwise dword stores.
// Taps are blended in fp32 from the stored fp16 cells.
constexpr int FSTR = 102;          // halves per footprint in LDS (even: rows of 10 halves stay 4-byte aligned)
constexpr int LQ = LP / 4;         // pixel quads per workgroup

template <bool kVec>
__global__ __launch_bounds__(kThreads) void corr_lookup_f16_kernel(const LookupArgs g) {
    __shared__ __attribute__((aligned(16))) _Float16 win[4 * LP * FSTR];
    __shared__ float sfx[4 * LP], sfy[4 * LP];
    __shared__ int sx0[4 * LP], sy0[4 * LP];
    const int tid = threadIdx.x;
    const int img = blockIdx.y;
    const int b = img / g.pairs, pair = img % g.pairs;
    const int p0 = blockIdx.x * LP;
    if (tid < 4 * LP) {
        const int pix = tid % LP, l = tid / LP;
        const int p = min(p0 + pix, g.N - 1);
        const float inv = 1.0f / (float)(1 << l);
        float cx = g.coords[((int64_t)img * 2 + 0) * g.N + p] * inv;
        float cy = g.coords[((int64_t)img * 2 + 1) * g.N + p] * inv;
        // anything this far out samples only zero padding; also swallows NaN/inf
        if (!(cx > -1.0e6f && cx < 1.0e6f)) cx = -1.0e6f;
        if (!(cy > -1.0e6f && cy < 1.0e6f)) cy = -1.0e6f;
        const float fx0 = floorf(cx), fy0 = floorf(cy);
        sx0[tid] = (int)fx0;
        sy0[tid] = (int)fy0;
        sfx[tid] = cx - fx0;
        sfy[tid] = cy - fy0;
    }
    __syncthreads();
    // ---- gather: flat item = ((level * 32 + pixel) * 100 + cell) ----
#pragma unroll 1
    for (int l = 0; l < 4; ++l) {
        const int hl = g.hl[l], wl = g.wl[l];
        const _Float16* vol = reinterpret_cast<const _Float16*>(g.lvl[l]) + pair * g.lvl_pair_stride[l] +
                              ((int64_t)b * g.N + p0) * hl * wl;                        // workgroup-uniform
        constexpr int kIt = (LP * FP * FP + kThreads - 1) / kThreads;                     // 13 (the last one partial)
        _Float16 v[kIt];
        bool ok[kIt];
#pragma unroll
        for (int j = 0; j < kIt; ++j) {
            const int it = min(tid + j * kThreads, LP * FP * FP - 1);
            const int pix = it / (FP * FP), cell = it % (FP * FP);
            const int yy = sy0[l * LP + pix] - RAD + cell / FP, xx = sx0[l * LP + pix] - RAD + cell % FP;
            ok[j] = (p0 + pix < g.N) && yy >= 0 && yy < hl && xx >= 0 && xx < wl;
            const int pc = min(pix, g.N - 1 - p0);
            v[j] = vol[pc * hl * wl + min(max(yy, 0), hl - 1) * wl + min(max(xx, 0), wl - 1)];
        }
#pragma unroll
        for (int j = 0; j < kIt; ++j) {
            const int it = tid + j * kThreads;
            if (it < LP * FP * FP) win[(l * LP + it / (FP * FP)) * FSTR + it % (FP * FP)] = ok[j] ? v[j] : (_Float16)0.f;
        }
    }
    __syncthreads();
    // ---- taps: channel = l*81 + a*9 + bb (a moves x, corr.py:31-37) ----
    float* out = g.out + (int64_t)img * g.out_img_stride + p0;
    constexpr int kTapIt = (4 * WIN * 3 * LQ + kThreads - 1) / kThreads;          // 4 (the last one partial)
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    h4 keep[kTapIt][3];                                                            // fp16 copies for the k-octet output
#pragma unroll
    for (int ti = 0; ti < kTapIt; ++ti) {
        const int it = tid + ti * kThreads;
        if (it >= 4 * WIN * 3 * LQ) continue;
        const int pq = it % LQ, r1 = it / LQ, ag = r1 % 3, r2 = r1 / 3, bb = r2 % WIN, l = r2 / WIN;
        float res[3][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int pix = pq * 4 + e;
            const float fx = sfx[l * LP + pix], fy = sfy[l * LP + pix];
            const float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy), w10 = (1.f - fx) * fy, w11 = fx * fy;
            const _Float16* c = win + (l * LP + pix) * FSTR + bb * FP + ag * 3;
            float c0[4], c1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { c0[k] = (float)c[k]; c1[k] = (float)c[FP + k]; }
#pragma unroll
            for (int k = 0; k < 3; ++k) res[k][e] = c0[k] * w00 + c0[k + 1] * w01 + c1[k] * w10 + c1[k + 1] * w11;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float* o = out + (int64_t)(l * WIN * WIN + (ag * 3 + k) * WIN + bb) * g.N + pq * 4;
            if (kVec) {
                if (p0 + pq * 4 < g.N) *reinterpret_cast<float4*>(o) = make_float4(res[k][0], res[k][1], res[k][2], res[k][3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (p0 + pq * 4 + e < g.N) o[e] = res[k][e];
            }
            keep[ti][k] = h4{(_Float16)res[k][0], (_Float16)res[k][1], (_Float16)res[k][2], (_Float16)res[k][3]};
        }
    }
    // ---- optional second output: the same 324 x 32 values as fp16 k-octets (SF_LAYOUT_F16_KOCT), the DMA-able B operand
    // of the first GEMM that reads the correlation features.  Transposed through LDS (the footprints are dead by now):
    // [channel][32 pixels] halves in, one (octet, pixel) = 8 channels = 16 bytes out, 512 contiguous bytes per octet.
    if (g.out16 == nullptr) return;                                                // workgroup-uniform
    constexpr int NCH = 4 * WIN * WIN, NOCT = (NCH + 7) / 8;                       // 324 channels, 41 octets
    static_assert(NOCT * 8 * LP <= 4 * LP * FSTR, "the transpose buffer aliases the footprints");
    __syncthreads();
    _Float16* tb = win;
#pragma unroll
    for (int ti = 0; ti < kTapIt; ++ti) {
        const int it = tid + ti * kThreads;
        if (it >= 4 * WIN * 3 * LQ) continue;
        const int pq = it % LQ, r1 = it / LQ, ag = r1 % 3, r2 = r1 / 3, bb = r2 % WIN, l = r2 / WIN;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            *reinterpret_cast<h4*>(tb + (l * WIN * WIN + (ag * 3 + k) * WIN + bb) * LP + pq * 4) = keep[ti][k];
    }
    for (int i = tid; i < (NOCT * 8 - NCH) * LP; i += kThreads) tb[NCH * LP + i] = (_Float16)0.f;   // rows 324..327
    __syncthreads();
    _Float16* o16 = g.out16 + (int64_t)img * g.out16_img_stride;
    for (int i = tid; i < NOCT * LP; i += kThreads) {
        const int pix = i % LP, oc = i / LP;
        if (p0 + pix >= g.N) continue;
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        h8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tb[(oc * 8 + e) * LP + pix];
        *reinterpret_cast<h8*>(o16 + ((int64_t)oc * g.N + p0 + pix) * 8) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// generic bilinear sampler (API parity for utils.bilinear_sampler; not on the fused path)
// ------------------------------------------------------------------------------------------------
__global__ void bilinear_sampler_kernel(const float* img, const float* coords, float* out, float* mask_out, int M,
                                        int C, int Hi, int Wi, int Ho, int Wo) {
    const int64_t total = (int64_t)M * C * Ho * Wo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(i % ((int64_t)Ho * Wo));
        const int c = (int)((i / ((int64_t)Ho * Wo)) % C);
        const int m = (int)(i / ((int64_t)Ho * Wo * C));
        const float* cp = coords + ((int64_t)m * Ho * Wo + o) * 2;
        float x = cp[0], y = cp[1];
        if (mask_out && c == 0) {
            // reference: normalised coords strictly inside (-1, 1)
            const float gx = 2.f * x / (float)(Wi - 1) - 1.f, gy = 2.f * y / (float)(Hi - 1) - 1.f;
            mask_out[(int64_t)m * Ho * Wo + o] = (gx > -1.f && gy > -1.f && gx < 1.f && gy < 1.f) ? 1.f : 0.f;
        }
        if (!(x > -1.0e6f && x < 1.0e6f)) x = -1.0e6f;
        if (!(y > -1.0e6f && y < 1.0e6f)) y = -1.0e6f;
        const float fx0 = floorf(x), fy0 = floorf(y);
        const int x0 = (int)fx0, y0 = (int)fy0;
        const float fx = x - fx0, fy = y - fy0;
        const float* ip = img + ((int64_t)m * C + c) * Hi * Wi;
        auto tap = [&](int yy, int xx) -> float {
            return (yy >= 0 && yy < Hi && xx >= 0 && xx < Wi) ? ip[yy * Wi + xx] : 0.f;
        };
        out[i] = tap(y0, x0) * ((1.f - fx) * (1.f - fy)) + tap(y0, x0 + 1) * (fx * (1.f - fy)) +
                 tap(y0 + 1, x0) * ((1.f - fx) * fy) + tap(y0 + 1, x0 + 1) * (fx * fy);
    }
}

}  // namespace

extern "C" int64_t sf_corr_build_ws_bytes(int B, int pairs, int D, int h, int w) {
    if (B <= 0 || pairs <= 0 || D <= 0 || h <= 0 || w <= 0) return 0;
    return (int64_t)2 * B * pairs * (sf::ceil_div(D, FK) * FK) * h * w * 4;      // (f1, f2) x images x (hi + lo) planes
}

extern "C" int sf_corr_build_pyramid(const float* f1, const float* f2, int64_t f_clip_stride, int64_t f_pair_stride,
                                     float* lvl0, float* lvl1, float* lvl2, float* lvl3,
                                     const int64_t* lvl_pair_stride, int B, int pairs, int D, int h, int w,
                                     int num_levels, int precision, void* split_ws, int64_t split_ws_bytes,
                                     void* stream) {
    return sf_corr_build_pyramid_pitched(f1, f2, f_clip_stride, f_pair_stride, lvl0, lvl1, lvl2, lvl3, lvl_pair_stride, nullptr,
                                         B, pairs, D, h, w, num_levels, precision, split_ws, split_ws_bytes, stream);
}

// Row pitch of the volume maps (cells): lvl_pitch[l] >= w >> l, map of one source pixel = (h >> l) rows of lvl_pitch[l] cells.
// The reference's dense [N, h_l, w_l] layout (lvl_pitch = NULL) puts a KITTI row (156 fp32 cells = 624 bytes) across cache-line
// boundaries with a different phase in every row and map: the build's 128-byte store runs straddle two lines each (1.56 TB/s
// against 2.6 at Sintel's 512-byte rows).  With a pitch of a multiple of 32 cells every row starts on a line;
// the pad cells of a row are WRITTEN (their target pixels lie outside the image: zero features, the stored value is 0 at level
// 0 and a don't-care partial mean above it) so that every line is written whole; lookups never read them.
extern "C" int sf_corr_build_pyramid_pitched(const float* f1, const float* f2, int64_t f_clip_stride, int64_t f_pair_stride,
                                             float* lvl0, float* lvl1, float* lvl2, float* lvl3,
                                             const int64_t* lvl_pair_stride, const int32_t* lvl_pitch, int B, int pairs, int D,
                                             int h, int w, int num_levels, int precision, void* split_ws,
                                             int64_t split_ws_bytes, void* stream) {
    SF_REQUIRE(f1 && f2 && lvl0 && lvl1 && lvl2 && lvl3, "sf_corr_build_pyramid: null pointer");
    if (lvl_pitch) {
        for (int l = 0; l < 4; ++l)
            SF_REQUIRE(lvl_pitch[l] >= (w >> l) && lvl_pitch[l] <= sf::ceil_div(w, 32) * 32,
                       "sf_corr_build_pyramid_pitched: lvl_pitch[%d] = %d must lie in [w >> l, w rounded up to 32]", l, lvl_pitch[l]);
    }
    SF_REQUIRE(B > 0 && pairs > 0 && D > 0 && h > 0 && w > 0, "sf_corr_build_pyramid: bad dims");
    SF_REQUIRE(pairs == 1 || lvl_pair_stride, "sf_corr_build_pyramid: pairs > 1 needs lvl_pair_stride");
    SF_REQUIRE(num_levels == 4, "sf_corr_build_pyramid: num_levels must be 4 (got %d)", num_levels);
    SF_REQUIRE(precision == SF_PRECISION_FP32 || precision == SF_PRECISION_F16X3 || precision == SF_PRECISION_F16,
               "sf_corr_build_pyramid: precision %d not supported", precision);
    const int Dp = sf::ceil_div(D, FK) * FK;                 // multiple of both stage depths (16 and 32)
    SF_REQUIRE(precision == SF_PRECISION_FP32 || (int64_t)Dp * h * w * 4 < ((int64_t)1 << 31),
               "sf_corr_build_pyramid: feature image larger than 2 GiB");
    SF_REQUIRE(precision == SF_PRECISION_FP32 ||
                   (split_ws && split_ws_bytes >= sf_corr_build_ws_bytes(B, pairs, D, h, w) &&
                    (reinterpret_cast<uintptr_t>(split_ws) & 15) == 0),
               "sf_corr_build_pyramid: SF_PRECISION_F16X3 needs a 16-byte aligned workspace of "
               "sf_corr_build_ws_bytes() bytes");
    SF_REQUIRE((int64_t)48 * h * w * 4 < ((int64_t)1 << 31), "sf_corr_build_pyramid: feature grid %dx%d too large", h, w);
    SF_REQUIRE((h >> 3) >= 1 && (w >> 3) >= 1, "sf_corr_build_pyramid: feature grid %dx%d too small for 4 levels", h, w);
    SF_REQUIRE((int64_t)B * pairs <= 65535, "sf_corr_build_pyramid: B*pairs too large");
    BuildArgs g;
    g.f1 = f1; g.f2 = f2;
    g.lvl[0] = lvl0; g.lvl[1] = lvl1; g.lvl[2] = lvl2; g.lvl[3] = lvl3;
    g.f_clip_stride = f_clip_stride; g.f_pair_stride = f_pair_stride;
    g.B = B; g.pairs = pairs; g.D = D; g.h = h; g.w = w; g.N = h * w;
    for (int l = 0; l < 4; ++l) {
        // (the store side uses wl as row pitch AND as column guard: with a pitch the pad columns are stored too, see above)
        g.hl[l] = h >> l; g.wl[l] = lvl_pitch ? lvl_pitch[l] : (w >> l);
        g.lvl_pair_stride[l] = (pairs > 1) ? lvl_pair_stride[l] : 0;
    }
    g.pcols = sf::ceil_div(w, PC);
    g.scale = 1.0f / sqrtf((float)D);
    g.vec_a = ((g.N & 3) == 0) && ((f_clip_stride & 3) == 0) && ((f_pair_stride & 3) == 0) &&
              ((reinterpret_cast<uintptr_t>(f1) & 15) == 0);
    g.vec_b = ((w & 3) == 0) && ((f_clip_stride & 3) == 0) && ((f_pair_stride & 3) == 0) &&
              ((reinterpret_cast<uintptr_t>(f2) & 15) == 0);
    dim3 grid(g.pcols * sf::ceil_div(h, PR), sf::ceil_div(g.N, BM), B * pairs);
    g.np = (int)grid.x; g.mt = (int)grid.y;
#ifdef SF_CORR_TIMERS
    g.ts = getenv("SF_CORR_TS_BUF") ? (long long*)strtoull(getenv("SF_CORR_TS_BUF"), nullptr, 0) : nullptr;
#endif
    {
        bool al = (w % 4 == 0);
        for (int l = 0; l < 4; ++l) {
            al = al && (reinterpret_cast<uintptr_t>(g.lvl[l]) & 15) == 0;
            if (pairs > 1) al = al && ((g.lvl_pair_stride[l] * (precision == SF_PRECISION_F16 ? 2 : 4)) & 15) == 0;
        }
        // measured on MI355X (r02): the transposed 16-byte epilogue is SLOWER than the dword one (fp16 build 2.20 vs
        // 1.68 ms, f16x3 build 3.03 vs 2.34 ms per 24 images) -- the build is not bound by store-instruction count.
        // Kept as a compile-time experiment (-DSF_CORR_VEC=1 through tools/build_variant.sh); see DESIGN.md.
#ifndef SF_CORR_VEC
#define SF_CORR_VEC 0
#endif
        g.vec_store = (al && SF_CORR_VEC == 1) ? 1 : 0;
    }
    // patches per L2-resident block: ~1.75 MB of packed target features (of the 4 MiB L2 of an XCD)
    const int patch_bytes = BN * Dp * (precision == SF_PRECISION_F16 ? 2 : 4);
    g.pblk = (7 << 18) / patch_bytes;
    g.pblk = g.pblk < 1 ? 1 : (g.pblk > g.np ? g.np : g.pblk);
    const int64_t n_wg = (int64_t)g.np * g.mt * B * pairs;
    SF_REQUIRE(precision == SF_PRECISION_FP32 || n_wg < ((int64_t)1 << 31), "sf_corr_build_pyramid: grid too large");
    if (precision == SF_PRECISION_F16X3) {
        hipLaunchKernelGGL(split_pack_kernel, dim3(sf::ceil_div(g.N, 256), Dp / 8, 2 * B * pairs), dim3(256), 0,
                           (hipStream_t)stream, f1, f2, f_clip_stride, f_pair_stride, (char*)split_ws, pairs, D, Dp, g.N);
        if (g.vec_store)
            hipLaunchKernelGGL(corr_build_dma_kernel<true>, dim3((unsigned)n_wg), dim3(kThreads), 0, (hipStream_t)stream, g,
                               (const char*)split_ws, Dp);
        else
            hipLaunchKernelGGL(corr_build_dma_kernel<false>, dim3((unsigned)n_wg), dim3(kThreads), 0, (hipStream_t)stream, g,
                               (const char*)split_ws, Dp);
    } else if (precision == SF_PRECISION_F16) {
        hipLaunchKernelGGL(pack_f16_kernel, dim3(sf::ceil_div(g.N, 256), Dp / 8, 2 * B * pairs), dim3(256), 0,
                           (hipStream_t)stream, f1, f2, f_clip_stride, f_pair_stride, (char*)split_ws, pairs, D, Dp, g.N);
        if (g.vec_store)
            hipLaunchKernelGGL(corr_build_f16_kernel<true>, dim3((unsigned)n_wg), dim3(kThreads), 0, (hipStream_t)stream, g,
                               (const char*)split_ws, Dp);
        else
            hipLaunchKernelGGL(corr_build_f16_kernel<false>, dim3((unsigned)n_wg), dim3(kThreads), 0, (hipStream_t)stream, g,
                               (const char*)split_ws, Dp);
    } else
        hipLaunchKernelGGL(corr_build_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, g);
    return sf::check_launch("sf_corr_build_pyramid");
}

extern "C" int sf_corr_lookup(const float* lvl0, const float* lvl1, const float* lvl2, const float* lvl3,
                              const int64_t* lvl_pair_stride, const float* coords, float* out,
                              int64_t out_img_stride, void* out_koct, int64_t out_koct_img_stride, int B, int pairs,
                              int h, int w, int num_levels, int radius, int vol_precision, void* stream) {
    return sf_corr_lookup_pitched(lvl0, lvl1, lvl2, lvl3, lvl_pair_stride, nullptr, coords, out, out_img_stride, out_koct,
                                  out_koct_img_stride, B, pairs, h, w, num_levels, radius, vol_precision, stream);
}

// lvl_pitch: row pitch (cells) of the maps written by sf_corr_build_pyramid_pitched; NULL = dense (fp32 volumes only)
extern "C" int sf_corr_lookup_pitched(const float* lvl0, const float* lvl1, const float* lvl2, const float* lvl3,
                                      const int64_t* lvl_pair_stride, const int32_t* lvl_pitch, const float* coords, float* out,
                                      int64_t out_img_stride, void* out_koct, int64_t out_koct_img_stride, int B, int pairs,
                                      int h, int w, int num_levels, int radius, int vol_precision, void* stream) {
    SF_REQUIRE(lvl0 && lvl1 && lvl2 && lvl3 && coords && out, "sf_corr_lookup: null pointer");
    SF_REQUIRE(!lvl_pitch || vol_precision != SF_PRECISION_F16, "sf_corr_lookup_pitched: pitched maps are fp32 volumes only");
    SF_REQUIRE(B > 0 && pairs > 0 && h > 0 && w > 0, "sf_corr_lookup: bad dims");
    SF_REQUIRE(pairs == 1 || lvl_pair_stride, "sf_corr_lookup: pairs > 1 needs lvl_pair_stride");
    SF_REQUIRE(num_levels == 4 && radius == RAD, "sf_corr_lookup: only num_levels=4, radius=4 (got %d, %d)",
               num_levels, radius);
    SF_REQUIRE((int64_t)B * pairs <= 65535, "sf_corr_lookup: B*pairs too large");
    LookupArgs g;
    g.lvl[0] = lvl0; g.lvl[1] = lvl1; g.lvl[2] = lvl2; g.lvl[3] = lvl3;
    g.coords = coords; g.out = out; g.out_img_stride = out_img_stride;
    g.out16 = static_cast<_Float16*>(out_koct); g.out16_img_stride = out_koct_img_stride;
    SF_REQUIRE(!out_koct || (vol_precision == SF_PRECISION_F16 && (reinterpret_cast<uintptr_t>(out_koct) & 15) == 0 &&
                             (out_koct_img_stride & 7) == 0),
               "sf_corr_lookup: the k-octet copy needs fp16 volumes, a 16-byte aligned out_koct and stride %% 8 == 0");
    g.B = B; g.pairs = pairs; g.h = h; g.w = w; g.N = h * w;
    for (int l = 0; l < 4; ++l) {
        g.hl[l] = h >> l; g.wl[l] = w >> l;
        g.pl[l] = lvl_pitch ? lvl_pitch[l] : g.wl[l];
        SF_REQUIRE(g.pl[l] >= g.wl[l], "sf_corr_lookup_pitched: lvl_pitch[%d] < level width", l);
        g.lvl_pair_stride[l] = (pairs > 1) ? lvl_pair_stride[l] : 0;
    }
    if (vol_precision == SF_PRECISION_F16) {
        SF_REQUIRE((int64_t)LP * h * w * 2 < ((int64_t)1 << 31), "sf_corr_lookup: feature grid too large");
        const bool vec = (g.N % 4 == 0) && (out_img_stride % 4 == 0) && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
        if (vec)
            hipLaunchKernelGGL(corr_lookup_f16_kernel<true>, dim3(sf::ceil_div(g.N, LP), B * pairs), dim3(kThreads), 0,
                               (hipStream_t)stream, g);
        else
            hipLaunchKernelGGL(corr_lookup_f16_kernel<false>, dim3(sf::ceil_div(g.N, LP), B * pairs), dim3(kThreads), 0,
                               (hipStream_t)stream, g);
        return sf::check_launch("sf_corr_lookup(f16)");
    }
    dim3 grid(sf::ceil_div(g.N, LP), 4, B * pairs);
    hipLaunchKernelGGL(corr_lookup_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, g);
    return sf::check_launch("sf_corr_lookup");
}

extern "C" int sf_bilinear_sampler(const float* img, const float* coords, float* out, float* mask_out, int M, int C,
                                   int Hi, int Wi, int Ho, int Wo, void* stream) {
    SF_REQUIRE(img && coords && out, "sf_bilinear_sampler: null pointer");
    SF_REQUIRE(M > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "sf_bilinear_sampler: bad dims");
    const int64_t total = (int64_t)M * C * Ho * Wo;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bilinear_sampler_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, img, coords,
                       out, mask_out, M, C, Hi, Wi, Ho, Wo);
    return sf::check_launch("sf_bilinear_sampler");
}
