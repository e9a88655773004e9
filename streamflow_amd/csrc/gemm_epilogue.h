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
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rc, crow[q] + ccol[j], 0, 0);
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

// host-side guard for the 32-bit buffer offsets used above
inline bool epilogue_spans_ok(const SfGemm& g) {
    const int64_t c = ((int64_t)(g.M - 1) * g.ldc + g.N) * 4;
    int64_t r = 0;
    if (g.R) {
        const int mr = g.M - 1;
        r = (g.r_group > 0) ? (int64_t)(mr / g.r_group) * g.r_group_stride + (int64_t)(mr % g.r_group) * g.ldr
                            : (int64_t)mr * g.ldr;
        r = (r + g.N) * 4;
    }
    return c < kOobTerm && r < kOobTerm;
}

}  // namespace sf
