// Shared GEMM epilogue: C = epilogue(alpha * (acc + bias)) for 32x32 MFMA accumulator tiles.
// C/D fragment layout of every gfx950 32x32 MFMA (dtype independent):
//   col = lane & 31,  row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5),  reg in [0,16).
//
// Written for memory-level parallelism and minimal address arithmetic:
//  * per 32-row block, all per-row parameters (bias, depthwise-1x1 scale/shift) and all residual values are
//    fetched with UNCONDITIONAL loads from clamped addresses before any arithmetic, so the 16*(1+TN) loads
//    of a block are in flight together;
//  * residual loads and result stores are BUFFER ops with 32-bit offsets: offset = rowterm[r] + colterm[j],
//    one integer add per element.  Rows m >= M / columns n >= N get a term of 2^30, which puts the offset
//    past num_records, so the hardware drops the store: no branches, no exec-mask juggling.
//    (A first version guarded every row with `if (m >= M) continue` and every store with an `if`: each row's
//    loads sat in their own basic block -- ~32 dependent memory round trips per workgroup.)
#pragma once
#include "sf_common.h"

#ifndef SF_EPI_STORE_AUX
#define SF_EPI_STORE_AUX 0       // cache policy of the result stores.  2 = non-temporal measured +1 % at 8 clips per
                                 // step (outputs larger than the Infinity Cache) and -1 % at one clip: left at default
#endif

namespace sf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// XCD-aware workgroup -> tile mapping.  MI355X dispatches workgroup b to XCD b % 8 (8 XCDs, private 4 MiB L2s).
// Each XCD gets one CONTIGUOUS range of tile ids (bijective also when the grid is not a multiple of 8), and
// tile ids run m-tile fastest: the m-tiles that share one B (activation) tile are consecutive ids on the
// same XCD, so the tile is fetched into that L2 once; and because the id -> (image, pixel tile) mapping is
// the same for every GEMM over the same pixels, the XCD that wrote a pixel range in one GEMM's epilogue is
// the one that reads it as the next GEMM's B operand.  Placement only affects speed, never results.
struct TileCoord { int m_tile, n_tile, z; };
__device__ __forceinline__ TileCoord xcd_tile(int b, int nwg, int mt, int nt) {
    constexpr int kXcd = 8;
    const int xcd = b % kXcd, local = b / kXcd;
    const int q = nwg / kXcd, r = nwg % kXcd;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    TileCoord t;
    t.m_tile = id % mt;
    t.n_tile = (id / mt) % nt;
    t.z = id / (mt * nt);
    return t;
}

constexpr int kOobTerm = 1 << 30;      // > any legal byte offset (host checks spans < 2^30)

template <int EPI>
__device__ __forceinline__ float epi_apply(float v, float r, float dww, float dwb, float gam) {
    if (EPI == SF_EPI_GELU) return gelu_erf(v);
    if (EPI == SF_EPI_RELU) return fmaxf(v, 0.f);
    if (EPI == SF_EPI_RES) return r + v;
    if (EPI == SF_EPI_RES_GELU) return gelu_erf(r + v);
    if (EPI == SF_EPI_RES_GELU_DW1) {
        const float t = gelu_erf(r + v);
        return gelu_erf(t + (dww * t + dwb));
    }
    if (EPI == SF_EPI_AXPY) return r + gam * v;
    return v;
}

// the same for two values at once (packed fp32 GELU)
// kFast: the polynomial GELU of the two-product arithmetic modes (sf_common.h gelu_poly2)
template <int EPI, bool kFast = false>
__device__ __forceinline__ f32x2 epi_apply2(f32x2 v, f32x2 r, float dww, float dwb, float gam) {
    if (EPI == SF_EPI_GELU) return gelu2<kFast>(v);
    if (EPI == SF_EPI_RES_GELU) return gelu2<kFast>(r + v);
    if (EPI == SF_EPI_RES_GELU_DW1) {
        const f32x2 t = gelu2<kFast>(r + v);
        return gelu2<kFast>(t + (splat2(dww) * t + splat2(dwb)));
    }
    f32x2 o;
    o[0] = epi_apply<EPI>(v[0], r[0], dww, dwb, gam);
    o[1] = epi_apply<EPI>(v[1], r[1], dww, dwb, gam);
    return o;
}

template <int EPI, int WM, int WN, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_impl(const SfGemm& g, f32x16 (&acc)[TM][TN], int m0, int n0, int z,
                                                   int wm, int wn, int lane) {
    constexpr bool kNeedsR = (EPI == SF_EPI_RES || EPI == SF_EPI_RES_GELU || EPI == SF_EPI_RES_GELU_DW1 ||
                              EPI == SF_EPI_AXPY);
    const int khalf = lane >> 5, l31 = lane & 31;
    const int c_bytes = (int)(((int64_t)(g.M - 1) * g.ldc + g.N) * 4);
    __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(g.C + (int64_t)z * g.strideC, 0, c_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rr = rc;
    if (kNeedsR) {
        const int mr = g.M - 1;
        const int64_t last = (g.r_group > 0) ? (int64_t)(mr / g.r_group) * g.r_group_stride + (int64_t)(mr % g.r_group) * g.ldr
                                             : (int64_t)mr * g.ldr;
        rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.R) + (int64_t)z * g.strideR, 0,
                                               (int)((last + g.N) * 4), 0x00020000);
    }
    const float gam = (EPI == SF_EPI_AXPY) ? g.gamma[0] : 0.f;
    const bool has_bias = g.bias != nullptr;
    int ccol[TN], rcol[TN];          // column byte terms: store (OOB-poisoned) and load (clamped)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + l31;
        ccol[j] = n < g.N ? n * 4 : kOobTerm;
        rcol[j] = (n < g.N ? n : g.N - 1) * 4;
    }
    constexpr int RC = 8;            // rows per chunk: bounds the live registers (RC*(4+TN) values)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mb = m0 + (wm * TM + i) * 32 + 4 * khalf;
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += RC) {
            float bias[RC], dww[RC], dwb[RC], rv[TN][RC];
            int crow[RC];
            // ---- all loads of the chunk, unconditional, clamped ----
#pragma unroll
            for (int q = 0; q < RC; ++q) {
                const int r = r0 + q;
                const int m = mb + (r & 3) + 8 * (r >> 2);
                const int mc = m < g.M ? m : g.M - 1;
                crow[q] = m < g.M ? m * (int)g.ldc * 4 : kOobTerm;
                bias[q] = has_bias ? g.bias[mc] : 0.f;
                if (EPI == SF_EPI_RES_GELU_DW1) {
                    dww[q] = g.dw_w[mc];
                    dwb[q] = g.dw_b[mc];
                } else {
                    dww[q] = dwb[q] = 0.f;
                }
                if (kNeedsR) {
                    const int rrow = (g.r_group > 0)
                        ? (int)(((int64_t)(mc / g.r_group) * g.r_group_stride + (int64_t)(mc % g.r_group) * g.ldr) * 4)
                        : mc * (int)g.ldr * 4;
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        rv[j][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, rrow + rcol[j], 0, 0));
                }
            }
            // ---- arithmetic + range-checked stores ----
#pragma unroll
            for (int q = 0; q < RC; ++q) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float v = g.alpha * (acc[i][j][r0 + q] + bias[q]);
                    const float o = epi_apply<EPI>(v, kNeedsR ? rv[j][q] : 0.f, dww[q], dwb[q], gam);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rc, crow[q] + ccol[j], 0, SF_EPI_STORE_AUX);
                }
            }
        }
    }
}

template <int WM, int WN, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const SfGemm& g, f32x16 (&acc)[TM][TN], int m0, int n0, int z, int wm,
                                              int wn, int lane) {
    switch (g.epilogue) {     // wave-uniform
        case SF_EPI_GELU: gemm_epilogue_impl<SF_EPI_GELU, WM, WN, TM, TN>(g, acc, m0, n0, z, wm, wn, lane); break;
        case SF_EPI_RELU: gemm_epilogue_impl<SF_EPI_RELU, WM, WN, TM, TN>(g, acc, m0, n0, z, wm, wn, lane); break;
        case SF_EPI_RES: gemm_epilogue_impl<SF_EPI_RES, WM, WN, TM, TN>(g, acc, m0, n0, z, wm, wn, lane); break;
        case SF_EPI_RES_GELU: gemm_epilogue_impl<SF_EPI_RES_GELU, WM, WN, TM, TN>(g, acc, m0, n0, z, wm, wn, lane); break;
        case SF_EPI_RES_GELU_DW1:
            gemm_epilogue_impl<SF_EPI_RES_GELU_DW1, WM, WN, TM, TN>(g, acc, m0, n0, z, wm, wn, lane); break;
        case SF_EPI_AXPY: gemm_epilogue_impl<SF_EPI_AXPY, WM, WN, TM, TN>(g, acc, m0, n0, z, wm, wn, lane); break;
        default: gemm_epilogue_impl<SF_EPI_NONE, WM, WN, TM, TN>(g, acc, m0, n0, z, wm, wn, lane); break;
    }
}


// ---- vector epilogue: 16-byte residual loads and result stores ------------------------------------------------
// The MFMA C/D layout gives a lane ONE column of 16 rows, so the natural store is 64 dword stores per thread (and as
// many dword residual loads): the epilogue is bound by memory-instruction issue, not bandwidth.  Here every 32x32
// accumulator tile goes through a per-wave LDS scratch (row stride 36 floats: conflict-free dword writes, 16-byte
// aligned rows) and comes back as 4 x float4 per lane (row = lane/8 + 8q, 4 consecutive columns), i.e. 4x fewer
// memory instructions with the same 128-byte segments.  Needs N % 4 == 0, ld % 4 == 0 and 16-byte aligned bases
// (checked by the caller, which falls back to the dword epilogue otherwise).
constexpr int kEpiStride = 36;
constexpr int kEpiScratchFloats = 32 * kEpiStride;          // per wave

typedef unsigned int epi_u32x4 __attribute__((ext_vector_type(4)));

// kRK (SfGemm.r_f16 = 2): the residual is an fp16 k-octet image.  In the accumulator layout a lane holds rows
// 8o + 4 khalf + (0..3) of pixel l31 for o = 0..3 -- half an octet of one pixel, 8 contiguous bytes -- so the residual is
// fetched with four 8-byte loads per 32x32 tile (lanes l and l + 32 share a 16-byte piece) and added to the accumulators
// BEFORE the transpose, scaled by 1 / alpha (v = alpha * (acc + R / alpha + bias) = alpha * (acc + bias) + R).
template <int EPI, int WM, int WN, int TM, int TN, bool kFast, bool kRK = false>
__device__ __forceinline__ void gemm_epilogue_vec_impl(const SfGemm& g, f32x16 (&acc)[TM][TN], int m0, int n0, int z,
                                                       int wm, int wn, int lane, float* scratch) {
    constexpr bool kNeedsR = !kRK && (EPI == SF_EPI_RES || EPI == SF_EPI_RES_GELU || EPI == SF_EPI_RES_GELU_DW1 ||
                                      EPI == SF_EPI_AXPY);
    const int khalf = lane >> 5, l31 = lane & 31;
    const int rrow = lane >> 3, rcol = (lane & 7) * 4;       // read-back coordinates inside a 32x32 tile
    // c_f16: C is IEEE fp16 storage (ldc / strideC in halves): four rounded values leave as one 8-byte store
    const int ces = g.c_f16 == 1 ? 2 : 4;
    const int c_bytes = (int)(((int64_t)(g.M - 1) * g.ldc + g.N) * ces);
    __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(g.C) + (int64_t)z * g.strideC * ces, 0, c_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rr = rc;
    if (kNeedsR) {
        const int mr = g.M - 1;
        const int64_t last = (g.r_group > 0) ? (int64_t)(mr / g.r_group) * g.r_group_stride + (int64_t)(mr % g.r_group) * g.ldr
                                             : (int64_t)mr * g.ldr;
        rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.R) + (int64_t)z * g.strideR, 0,
                                               (int)((last + g.N) * 4), 0x00020000);
    }
    // c_f16 = 3: second, k-octet fp16 copy of C at C16 (same ldc, strideC16 in halves)
    __amdgpu_buffer_rsrc_t rc16 = rc;
    if (g.c_f16 == 3)
        rc16 = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(g.C16) + (int64_t)z * g.strideC16 * 2, 0,
                                                 (int)((int64_t)((g.M + 7) / 8) * g.ldc * 16), 0x00020000);
    const float gam = (EPI == SF_EPI_AXPY) ? g.gamma[0] : 0.f;
    const bool has_bias = g.bias != nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mt0 = m0 + (wm * TM + i) * 32;
        if constexpr (kRK) {
            typedef unsigned int epi_u32x2 __attribute__((ext_vector_type(2)));
            typedef _Float16 epi_h2 __attribute__((ext_vector_type(2)));
            const int moct = (g.M + 7) / 8;
            const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(reinterpret_cast<const char*>(g.R)) + (int64_t)z * g.strideR * 2, 0,
                (int)((int64_t)moct * g.ldr * 16), 0x00020000);
            const float inv_alpha = 1.0f / g.alpha;
            epi_u32x2 rh[TN][4];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n1 = n0 + (wn * TN + j) * 32 + l31;
                const int nc = n1 < g.N ? n1 : g.N - 1;
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int oc = min((mt0 >> 3) + o, moct - 1);        // octets past M: clamped, their rows are never stored
                    rh[j][o] = __builtin_amdgcn_raw_buffer_load_b64(rk, (oc * (int)g.ldr + nc) * 16 + khalf * 8, 0, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int o = 0; o < 4; ++o)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const unsigned u = rh[j][o][e];
                        const epi_h2 hv = __builtin_bit_cast(epi_h2, u);
                        acc[i][j][4 * o + 2 * e] += (float)hv[0] * inv_alpha;
                        acc[i][j][4 * o + 2 * e + 1] += (float)hv[1] * inv_alpha;
                    }
        }
        // per-row parameters and residuals of this row of tiles (all loads issued before any arithmetic)
        float bias[4], dww[4], dwb[4];
        int crow[4];
        epi_u32x4 rv[TN][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = mt0 + rrow + 8 * q;
            const int mc = m < g.M ? m : g.M - 1;
            crow[q] = m < g.M ? m * (int)g.ldc * ces : kOobTerm;
            bias[q] = has_bias ? g.bias[mc] : 0.f;
            if (EPI == SF_EPI_RES_GELU_DW1) { dww[q] = g.dw_w[mc]; dwb[q] = g.dw_b[mc]; }
            else { dww[q] = dwb[q] = 0.f; }
            if (kNeedsR) {
                const int rro = (g.r_group > 0)
                    ? (int)(((int64_t)(mc / g.r_group) * g.r_group_stride + (int64_t)(mc % g.r_group) * g.ldr) * 4)
                    : mc * (int)g.ldr * 4;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + (wn * TN + j) * 32 + rcol;
                    const int nc = n < g.N ? n : g.N - 4;         // N % 4 == 0: a column quad is all in or all out
                    rv[j][q] = __builtin_amdgcn_raw_buffer_load_b128(rr, rro + nc * 4, 0, 0);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            // transpose the tile through LDS (a wave's DS operations execute in order, no barrier needed)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                scratch[((r & 3) + 8 * (r >> 2) + 4 * khalf) * kEpiStride + l31] = acc[i][j][r];
            const int n = n0 + (wn * TN + j) * 32 + rcol;
            const int ccol = n < g.N ? n * ces : kOobTerm;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 a = *reinterpret_cast<const float4*>(scratch + (rrow + 8 * q) * kEpiStride + rcol);
                const float av[4] = {a.x, a.y, a.z, a.w};
                epi_u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    f32x2 v, r;
                    v[0] = g.alpha * (av[e] + bias[q]);
                    v[1] = g.alpha * (av[e + 1] + bias[q]);
                    // copy the element to a scalar first: __builtin_bit_cast applied directly to an ext-vector element
                    // lvalue reads the vector's FIRST element (clang quirk seen with ROCm 7.2)
                    const unsigned ru0 = kNeedsR ? rv[j][q][e] : 0u, ru1 = kNeedsR ? rv[j][q][e + 1] : 0u;
                    r[0] = __builtin_bit_cast(float, ru0);
                    r[1] = __builtin_bit_cast(float, ru1);
                    const f32x2 res = epi_apply2<EPI, kFast>(v, r, dww[q], dwb[q], gam);
                    const float r0 = res[0], r1 = res[1];
                    o[e] = __builtin_bit_cast(unsigned, r0);
                    o[e + 1] = __builtin_bit_cast(unsigned, r1);
                }
                if (g.c_f16 == 3)                                // wave-uniform: final values back into the scratch
                    *reinterpret_cast<epi_u32x4*>(scratch + (rrow + 8 * q) * kEpiStride + rcol) = o;
                if (g.c_f16 == 1) {                              // wave-uniform
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                    u32x2 oh;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const unsigned u0 = o[2 * e], u1 = o[2 * e + 1];
                        h2 hv;
                        hv[0] = (_Float16)__builtin_bit_cast(float, u0);
                        hv[1] = (_Float16)__builtin_bit_cast(float, u1);
                        oh[e] = __builtin_bit_cast(unsigned, hv);
                    }
                    __builtin_amdgcn_raw_buffer_store_b64(oh, rc, crow[q] + ccol, 0, SF_EPI_STORE_AUX);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(o, rc, crow[q] + ccol, 0, SF_EPI_STORE_AUX);
                }
            }
            if (g.c_f16 == 3) {
                // dual output: the finished tile (now in the scratch) leaves a second time as fp16 k-octets -- lane =
                // (octet half, pixel) as in the k-octet epilogue below.  Rows >= M of a last, partial octet are NOT
                // written (another producer may own them: the flow rows of StreamFlow's motion features).
                const int n1 = n0 + (wn * TN + j) * 32 + l31;
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const int mo = mt0 + (2 * q2 + khalf) * 8;
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                    epi_u32x4 oh;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        h2 hv;
                        hv[0] = (_Float16)scratch[((2 * q2 + khalf) * 8 + e) * kEpiStride + l31];
                        hv[1] = (_Float16)scratch[((2 * q2 + khalf) * 8 + e + 1) * kEpiStride + l31];
                        oh[e >> 1] = __builtin_bit_cast(unsigned, hv);
                    }
                    const int o16 = ((mo >> 3) * (int)g.ldc + n1) * 16;
                    if (mo + 8 <= g.M) {                                           // wave-uniform per half
                        __builtin_amdgcn_raw_buffer_store_b128(oh, rc16, n1 < g.N ? o16 : kOobTerm, 0, SF_EPI_STORE_AUX);
                    } else if (mo < g.M) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const unsigned w = oh[e >> 1];
                            const unsigned short hs = (unsigned short)((e & 1) ? (w >> 16) : (w & 0xffffu));
                            __builtin_amdgcn_raw_buffer_store_b16(hs, rc16, (n1 < g.N && mo + e < g.M) ? o16 + e * 2 : kOobTerm,
                                                                  0, SF_EPI_STORE_AUX);
                        }
                    }
                }
            }
        }
    }
}

// ---- k-octet epilogue (c_f16 = 2): the result leaves as the NEXT GEMM's LDS image ------------------------------------------
// Output element (m, n) -> IEEE fp16 at ((m / 8) * ldc + n) * 8 + m % 8 (SF_LAYOUT_F16_KOCT).  Each 32x32 accumulator
// tile goes through the per-wave scratch like the vector epilogue, but is read back with a lane owning ONE pixel and the
// 8 channels of one k-octet (lane = (octet half, pixel)): eight ds_read_b32, the epilogue arithmetic on packed pairs,
// one 16-byte store; 32 consecutive lanes = 32 consecutive pixels = 512 contiguous bytes.  Rows past M inside the last
// octet are written too (finite values from clamped parameters; the consumer's weights are zero there).
// kSplit (c_f16 = 4): the value leaves as the (hi, lo) fp16 pair sf_split::split8 would make of it -- hi = the fp32 value with its
// mantissa truncated to 10 bits, lo = the rest, both narrowed with round-to-zero -- in two k-octet images, lo behind hi.
template <int EPI, int WM, int WN, int TM, int TN, bool kFast, bool kSplit = false>
__device__ __forceinline__ void gemm_epilogue_koct_impl(const SfGemm& g, f32x16 (&acc)[TM][TN], int m0, int n0, int z,
                                                        int wm, int wn, int lane, float* scratch) {
    constexpr bool kNeedsR = (EPI == SF_EPI_RES || EPI == SF_EPI_RES_GELU || EPI == SF_EPI_RES_GELU_DW1 ||
                              EPI == SF_EPI_AXPY);
    const int khalf = lane >> 5, l31 = lane & 31;
    const int moct = (g.M + 7) / 8;
    const int c_plane = (int)((int64_t)moct * g.ldc * 16);      // bytes of the hi (= lo) image
    const int c_bytes = kSplit ? 2 * c_plane : c_plane;
    __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(g.C) + (int64_t)z * g.strideC * 2, 0, c_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rr = rc;
    if (kNeedsR) {
        const int mr = g.M - 1;
        const int64_t last = (g.r_group > 0) ? (int64_t)(mr / g.r_group) * g.r_group_stride + (int64_t)(mr % g.r_group) * g.ldr
                                             : (int64_t)mr * g.ldr;
        rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.R) + (int64_t)z * g.strideR, 0,
                                               (int)((last + g.N) * 4), 0x00020000);
    }
    const float gam = (EPI == SF_EPI_AXPY) ? g.gamma[0] : 0.f;
    const bool has_bias = g.bias != nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mt0 = m0 + (wm * TM + i) * 32;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                scratch[((r & 3) + 8 * (r >> 2) + 4 * khalf) * kEpiStride + l31] = acc[i][j][r];
            const int n = n0 + (wn * TN + j) * 32 + l31;
            const int nc = n < g.N ? n : g.N - 1;
#pragma unroll
            for (int q = 0; q < 2; ++q) {                         // octets (2q + khalf) of the 32-row tile
                const int mo = mt0 + (2 * q + khalf) * 8;         // first row of this lane's octet
                epi_u32x4 o, ol;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    f32x2 v, r;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int m = mo + e + u, mc = m < g.M ? m : g.M - 1;
                        const float a = scratch[((2 * q + khalf) * 8 + e + u) * kEpiStride + l31];
                        v[u] = g.alpha * (a + (has_bias ? g.bias[mc] : 0.f));
                        r[u] = 0.f;
                        if (kNeedsR) {
                            const int rro = (g.r_group > 0)
                                ? (int)(((int64_t)(mc / g.r_group) * g.r_group_stride + (int64_t)(mc % g.r_group) * g.ldr) * 4)
                                : mc * (int)g.ldr * 4;
                            r[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, rro + nc * 4, 0, 0));
                        }
                    }
                    const int mc0 = (mo + e < g.M) ? mo + e : g.M - 1;
                    const f32x2 res = epi_apply2<EPI, kFast>(v, r, EPI == SF_EPI_RES_GELU_DW1 ? g.dw_w[mc0] : 0.f,
                                                      EPI == SF_EPI_RES_GELU_DW1 ? g.dw_b[mc0] : 0.f, gam);
                    if (kSplit) {
                        const float r0 = res[0], r1 = res[1];    // (scalars first: a bit_cast of a vector ELEMENT reads element 0)
                        const float ah = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r0) & 0xFFFFE000u);
                        const float bh = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1) & 0xFFFFE000u);
                        o[e >> 1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ah, bh));
                        ol[e >> 1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0 - ah, r1 - bh));
                    } else {
                        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                        h2 hv;
                        hv[0] = (_Float16)res[0];
                        hv[1] = (_Float16)res[1];
                        o[e >> 1] = __builtin_bit_cast(unsigned, hv);
                    }
                }
                const bool ok = n < g.N && mo < g.M;
                const int vo = ok ? ((mo >> 3) * (int)g.ldc + n) * 16 : kOobTerm;
                __builtin_amdgcn_raw_buffer_store_b128(o, rc, vo, 0, SF_EPI_STORE_AUX);
                if (kSplit) __builtin_amdgcn_raw_buffer_store_b128(ol, rc, ok ? vo + c_plane : kOobTerm, 0, SF_EPI_STORE_AUX);
            }
        }
    }
}

template <int WM, int WN, int TM, int TN, bool kFast = false, bool kSplit = false>
__device__ __forceinline__ void gemm_epilogue_koct(const SfGemm& g, f32x16 (&acc)[TM][TN], int m0, int n0, int z, int wm,
                                                   int wn, int lane, float* scratch) {
    switch (g.epilogue) {     // wave-uniform; the hand-over tensors are produced with these three epilogues only
        case SF_EPI_GELU: gemm_epilogue_koct_impl<SF_EPI_GELU, WM, WN, TM, TN, kFast, kSplit>(g, acc, m0, n0, z, wm, wn, lane, scratch); break;
        case SF_EPI_RES_GELU: gemm_epilogue_koct_impl<SF_EPI_RES_GELU, WM, WN, TM, TN, kFast, kSplit>(g, acc, m0, n0, z, wm, wn, lane, scratch); break;
        default: gemm_epilogue_koct_impl<SF_EPI_NONE, WM, WN, TM, TN, kFast, kSplit>(g, acc, m0, n0, z, wm, wn, lane, scratch); break;
    }
}

template <int WM, int WN, int TM, int TN, bool kFast = false>
__device__ __forceinline__ void gemm_epilogue_vec(const SfGemm& g, f32x16 (&acc)[TM][TN], int m0, int n0, int z, int wm,
                                                  int wn, int lane, float* scratch) {
    switch (g.epilogue) {     // wave-uniform
        case SF_EPI_GELU: gemm_epilogue_vec_impl<SF_EPI_GELU, WM, WN, TM, TN, kFast>(g, acc, m0, n0, z, wm, wn, lane, scratch); break;
        case SF_EPI_RELU: gemm_epilogue_vec_impl<SF_EPI_RELU, WM, WN, TM, TN, kFast>(g, acc, m0, n0, z, wm, wn, lane, scratch); break;
        case SF_EPI_RES: gemm_epilogue_vec_impl<SF_EPI_RES, WM, WN, TM, TN, kFast>(g, acc, m0, n0, z, wm, wn, lane, scratch); break;
        case SF_EPI_RES_GELU: gemm_epilogue_vec_impl<SF_EPI_RES_GELU, WM, WN, TM, TN, kFast>(g, acc, m0, n0, z, wm, wn, lane, scratch); break;
        case SF_EPI_RES_GELU_DW1:
            if (g.r_f16 == 2)   // wave-uniform; the one epilogue built for a k-octet residual (host-checked)
                gemm_epilogue_vec_impl<SF_EPI_RES_GELU_DW1, WM, WN, TM, TN, kFast, true>(g, acc, m0, n0, z, wm, wn, lane, scratch);
            else
                gemm_epilogue_vec_impl<SF_EPI_RES_GELU_DW1, WM, WN, TM, TN, kFast>(g, acc, m0, n0, z, wm, wn, lane, scratch);
            break;
        case SF_EPI_AXPY: gemm_epilogue_vec_impl<SF_EPI_AXPY, WM, WN, TM, TN, kFast>(g, acc, m0, n0, z, wm, wn, lane, scratch); break;
        default: gemm_epilogue_vec_impl<SF_EPI_NONE, WM, WN, TM, TN, kFast>(g, acc, m0, n0, z, wm, wn, lane, scratch); break;
    }
}

// wave-uniform: may the vector epilogue be used for this problem?
__device__ __forceinline__ bool epilogue_vec_ok(const SfGemm& g, int z) {
    bool ok = (g.N & 3) == 0 && (g.ldc & 3) == 0 && (g.strideC & 3) == 0 &&
              ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0);
    if (g.R && g.r_f16 == 2) ok = ok && (g.strideR & 7) == 0 && ((reinterpret_cast<uintptr_t>(g.R) & 15) == 0);
    else if (g.R) ok = ok && (g.ldr & 3) == 0 && (g.strideR & 3) == 0 && (g.r_group_stride & 3) == 0 &&
                       ((reinterpret_cast<uintptr_t>(g.R) & 15) == 0);
    return ok;
}

// host-side guard for the 32-bit buffer offsets used above
inline bool epilogue_spans_ok(const SfGemm& g) {
    const int64_t c = (g.c_f16 == 2 || g.c_f16 == 4) ? (int64_t)((g.M + 7) / 8) * g.ldc * 16 * (g.c_f16 == 4 ? 2 : 1)
                                     : ((int64_t)(g.M - 1) * g.ldc + g.N) * (g.c_f16 == 1 ? 2 : 4);
    if (g.c_f16 == 3 && (int64_t)((g.M + 7) / 8) * g.ldc * 16 >= kOobTerm) return false;
    int64_t r = 0;
    if (g.R && g.r_f16 == 2) {
        r = (int64_t)((g.M + 7) / 8) * g.ldr * 16;
    } else if (g.R) {
        const int mr = g.M - 1;
        r = (g.r_group > 0) ? (int64_t)(mr / g.r_group) * g.r_group_stride + (int64_t)(mr % g.r_group) * g.ldr
                            : (int64_t)mr * g.ldr;
        r = (r + g.N) * 4;
    }
    return c < kOobTerm && r < kOobTerm;
}

}  // namespace sf
