// Fused fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain).
//
//   C[z][m][n] = epilogue( alpha * ( sum_k A[z][m][k] * B[z][k][n] + bias[m] ) )
//
// Every 1x1 convolution / nn.Linear / einsum of the StreamFlow update block goes through this one
// kernel family (reference call sites are listed in include/streamflow_hip.h at sf_gemm).
// Activations are channel-major planes [C][P] (P = h*w contiguous), so a 1x1 conv is
// W[Cout][Cin] x X[Cin][P]: "M" = output channels, "N" = pixels, "K" = input channels.
//
// Tiling (wave64, 4 waves / workgroup): the workgroup owns a BM x 128 output tile, each wave a
// (TM*32) x (TN*32) sub-tile held as TM*TN accumulators of 16 VGPRs.  A and B tiles of depth BK are
// staged through LDS in k-major rows ([k][m] / [k][n], row stride = tile + 4 floats, so the MFMA
// operand reads `lds[k][lane&31]` hit 32 distinct banks per half-wave) with register double
// buffering: global loads for tile t+1 are issued before the MFMAs of tile t and written to the
// other LDS buffer afterwards (one barrier per k-tile).
#include "sf_common.h"
#include "gemm_epilogue.h"

namespace {

using sf::f32x16;
using sf::gemm_epilogue;

constexpr int kThreads = 256;

template <int V>
struct Regs {
    float4 v[V > 0 ? V : 1];
};

// ---------------------------------------------------------------------------------------------
// tile loaders: global -> registers.  X = M (for A) or N (for B).
// K-major source: element (k, x) at base + row(k) + x.   K-minor source: base + x*ld + k.
// ---------------------------------------------------------------------------------------------
struct SrcDesc {
    const float* base;
    int64_t ld;
    int X, K;
    int group;            // grouped rows (K-major only), 0 = none
    int64_t group_stride;
    int conv3x3, h, w, cin;
    bool vec_ok;          // 16-byte loads allowed (base and ld aligned)
};

__device__ __forceinline__ int64_t krow_offset(const SrcDesc& s, int k) {
    if (s.group > 0) return (int64_t)(k / s.group) * s.group_stride + (int64_t)(k % s.group) * s.ld;
    return (int64_t)k * s.ld;
}

// `interior` is workgroup-uniform: the whole tile lies inside the operand and 16-byte loads are legal, so the
// loads are unconditional (all in flight together); edge tiles take the guarded path.
template <int BX, int BK, int NV>
__device__ __forceinline__ void load_kmajor(const SrcDesc& s, int k0, int x0, int tid, bool interior, Regs<NV>& r) {
    constexpr int C4 = BX / 4;
    constexpr bool kAllLive = (BK * C4) % kThreads == 0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int idx = tid + j * kThreads;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (interior && !s.conv3x3 && (kAllLive || idx < BK * C4)) {
            const int k = k0 + idx / C4, x = x0 + (idx % C4) * 4;
            v = *reinterpret_cast<const float4*>(s.base + krow_offset(s, k) + x);
        } else if (idx < BK * C4) {
            const int k = k0 + idx / C4;
            const int x = x0 + (idx % C4) * 4;
            if (k < s.K) {
                if (s.conv3x3) {
                    const int tap = k / s.cin, c = k - tap * s.cin;
                    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                    const float* p = s.base + (int64_t)c * s.ld;
                    float t[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int n = x + i;
                        float val = 0.f;
                        if (n < s.X) {
                            const int yy = n / s.w + dy, xx = n % s.w + dx;
                            if (yy >= 0 && yy < s.h && xx >= 0 && xx < s.w) val = p[yy * s.w + xx];
                        }
                        t[i] = val;
                    }
                    v = make_float4(t[0], t[1], t[2], t[3]);
                } else {
                    const float* p = s.base + krow_offset(s, k) + x;
                    if (s.vec_ok && x + 3 < s.X) {
                        v = *reinterpret_cast<const float4*>(p);
                    } else {
                        if (x + 0 < s.X) v.x = p[0];
                        if (x + 1 < s.X) v.y = p[1];
                        if (x + 2 < s.X) v.z = p[2];
                        if (x + 3 < s.X) v.w = p[3];
                    }
                }
            }
        }
        r.v[j] = v;
    }
}

template <int BX, int BK, int NV, int STRIDE>
__device__ __forceinline__ void store_kmajor(float* lds, int tid, const Regs<NV>& r) {
    constexpr int C4 = BX / 4;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int idx = tid + j * kThreads;
        if (idx < BK * C4) {
            const int k = idx / C4, x = (idx % C4) * 4;
            *reinterpret_cast<float4*>(lds + k * STRIDE + x) = r.v[j];
        }
    }
}

template <int BX, int BK, int NV>
__device__ __forceinline__ void load_kminor(const SrcDesc& s, int k0, int x0, int tid, bool interior, Regs<NV>& r) {
    constexpr int K4 = BK / 4;
    constexpr bool kAllLive = (BX * K4) % kThreads == 0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int idx = tid + j * kThreads;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (interior && (kAllLive || idx < BX * K4)) {
            const int x = x0 + idx / K4, k = k0 + (idx % K4) * 4;
            v = *reinterpret_cast<const float4*>(s.base + (int64_t)x * s.ld + k);
        } else if (idx < BX * K4) {
            const int x = x0 + idx / K4;
            const int k = k0 + (idx % K4) * 4;
            if (x < s.X) {
                const float* p = s.base + (int64_t)x * s.ld + k;
                if (s.vec_ok && k + 3 < s.K) {
                    v = *reinterpret_cast<const float4*>(p);
                } else {
                    if (k + 0 < s.K) v.x = p[0];
                    if (k + 1 < s.K) v.y = p[1];
                    if (k + 2 < s.K) v.z = p[2];
                    if (k + 3 < s.K) v.w = p[3];
                }
            }
        }
        r.v[j] = v;
    }
}

template <int BX, int BK, int NV, int STRIDE>
__device__ __forceinline__ void store_kminor(float* lds, int tid, const Regs<NV>& r) {
    constexpr int K4 = BK / 4;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int idx = tid + j * kThreads;
        if (idx < BX * K4) {
            const int x = idx / K4, k = (idx % K4) * 4;
            lds[(k + 0) * STRIDE + x] = r.v[j].x;
            lds[(k + 1) * STRIDE + x] = r.v[j].y;
            lds[(k + 2) * STRIDE + x] = r.v[j].z;
            lds[(k + 3) * STRIDE + x] = r.v[j].w;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN, int BK, int ALAY, int BLAY>
__global__ __launch_bounds__(kThreads) void gemm_f32_mfma(const SfGemm g) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int SA = BM + 4, SB = BN + 4;
    constexpr int NVA = (BM * BK / 4 + kThreads - 1) / kThreads;
    constexpr int NVB = (BN * BK / 4 + kThreads - 1) / kThreads;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (SA + SB)];
    float* sA = smem;
    float* sB = smem + 2 * BK * SA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int mt = (g.M + BM - 1) / BM, nt = (g.N + BN - 1) / BN;
    const sf::TileCoord tc = sf::xcd_tile(blockIdx.x, gridDim.x, mt, nt);
    const int n0 = tc.n_tile * BN, m0 = tc.m_tile * BM, z = tc.z;

    SrcDesc da, db;
    da.base = g.A + (int64_t)z * g.strideA;
    da.ld = g.lda; da.X = g.M; da.K = g.K; da.group = 0; da.group_stride = 0;
    da.conv3x3 = 0; da.h = da.w = da.cin = 0;
    da.vec_ok = ((g.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(da.base) & 15) == 0);
    db.base = g.B + (int64_t)z * g.strideB;
    db.ld = g.ldb; db.X = g.N; db.K = g.K; db.group = g.b_group; db.group_stride = g.b_group_stride;
    db.conv3x3 = g.conv3x3; db.h = g.h; db.w = g.w; db.cin = g.conv3x3 ? g.K / 9 : 0;
    db.vec_ok = ((g.ldb & 3) == 0) && ((g.b_group_stride & 3) == 0) &&
                ((reinterpret_cast<uintptr_t>(db.base) & 15) == 0);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    Regs<NVA> ra;
    Regs<NVB> rb;
    const int nk = (g.K + BK - 1) / BK;

    const bool a_in = da.vec_ok && (g.a_padded || m0 + BM <= g.M), b_in = db.vec_ok && n0 + BN <= g.N;
    auto load_tiles = [&](int kt) {
        const bool k_in = (kt + 1) * BK <= g.K;
        const bool ai = a_in && (k_in || (g.a_padded && ALAY == SF_LAYOUT_K_MAJOR)), bi = b_in && k_in;
        if (ALAY == SF_LAYOUT_K_MAJOR) load_kmajor<BM, BK, NVA>(da, kt * BK, m0, tid, ai, ra);
        else load_kminor<BM, BK, NVA>(da, kt * BK, m0, tid, ai, ra);
        if (BLAY == SF_LAYOUT_K_MAJOR) load_kmajor<BN, BK, NVB>(db, kt * BK, n0, tid, bi, rb);
        else load_kminor<BN, BK, NVB>(db, kt * BK, n0, tid, bi, rb);
    };
    auto store_tiles = [&](int buf) {
        if (ALAY == SF_LAYOUT_K_MAJOR) store_kmajor<BM, BK, NVA, SA>(sA + buf * BK * SA, tid, ra);
        else store_kminor<BM, BK, NVA, SA>(sA + buf * BK * SA, tid, ra);
        if (BLAY == SF_LAYOUT_K_MAJOR) store_kmajor<BN, BK, NVB, SB>(sB + buf * BK * SB, tid, rb);
        else store_kminor<BN, BK, NVB, SB>(sB + buf * BK * SB, tid, rb);
    };

    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    const int khalf = lane >> 5, l31 = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tiles(kt + 1);
        const float* pa = sA + cur * BK * SA + wm * TM * 32 + l31;
        const float* pb = sB + cur * BK * SB + wn * TN * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const int kk = 2 * ks + khalf;
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = pa[kk * SA + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = pb[kk * SB + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
    }

    if (sizeof(smem) >= 4 * sf::kEpiScratchFloats * sizeof(float) && sf::epilogue_vec_ok(g, z)) {
        __syncthreads();                                         // the main-loop LDS becomes the transpose scratch
        sf::gemm_epilogue_vec<WM, WN, TM, TN>(g, acc, m0, n0, z, wm, wn, lane, smem + wave * sf::kEpiScratchFloats);
    } else {
        gemm_epilogue<WM, WN, TM, TN>(g, acc, m0, n0, z, wm, wn, lane);
    }
}

template <int WM, int WN, int TM, int TN, int BK>
int launch_cfg(const SfGemm& g, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    dim3 grid(sf::ceil_div(g.N, BN) * sf::ceil_div(g.M, BM) * g.batch);      // 1-D: see sf::xcd_tile
    const int lay = g.a_layout * 2 + g.b_layout;
    switch (lay) {
        case 0: hipLaunchKernelGGL((gemm_f32_mfma<WM, WN, TM, TN, BK, 0, 0>), grid, dim3(kThreads), 0, st, g); break;
        case 1: hipLaunchKernelGGL((gemm_f32_mfma<WM, WN, TM, TN, BK, 0, 1>), grid, dim3(kThreads), 0, st, g); break;
        case 2: hipLaunchKernelGGL((gemm_f32_mfma<WM, WN, TM, TN, BK, 1, 0>), grid, dim3(kThreads), 0, st, g); break;
        default: hipLaunchKernelGGL((gemm_f32_mfma<WM, WN, TM, TN, BK, 1, 1>), grid, dim3(kThreads), 0, st, g); break;
    }
    return sf::check_launch("sf_gemm");
}

// Workgroup tile choice: minimise the estimated makespan in units of (32x32xK wave tiles per SIMD),
// assuming 256 CUs with every workgroup resident; ties go to the larger tile (less L2 traffic).
int pick_bm(const SfGemm& g) {
    const int cands[3] = {128, 64, 32};
    long best = -1;
    int best_bm = 128;
    for (int c = 0; c < 3; ++c) {
        const int bm = cands[c];
        const long wgs = (long)sf::ceil_div(g.M, bm) * sf::ceil_div(g.N, 128) * g.batch;
        const long span = ((wgs + 255) / 256) * (bm / 32);
        if (best < 0 || span < best) { best = span; best_bm = bm; }
    }
    return best_bm;
}

}  // namespace

namespace sf {
int gemm_split_dispatch(const SfGemm& g, hipStream_t st);   // gemm_split.hip
int64_t gemm_split_ws_floats(int M, int N, int K, int batch);
}

extern "C" int64_t sf_gemm_split_ws_floats(int M, int N, int K, int batch) {
    return sf::gemm_split_ws_floats(M, N, K, batch);
}

extern "C" int sf_gemm(const SfGemm* gp, void* stream) {
    SF_REQUIRE(gp != nullptr, "sf_gemm: null descriptor");
    const SfGemm& g = *gp;
    SF_REQUIRE(g.B && g.C, "sf_gemm: null B/C");
    SF_REQUIRE(g.A || (g.a_layout == SF_LAYOUT_SPLIT_F16 && g.A_hi && g.A_lo), "sf_gemm: null A");
    SF_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && g.batch > 0, "sf_gemm: bad dims M=%d N=%d K=%d batch=%d", g.M, g.N,
               g.K, g.batch);
    SF_REQUIRE(g.a_layout >= 0 && g.a_layout <= 2 && g.b_layout >= 0 && g.b_layout <= 6 && g.b_layout != 2, "sf_gemm: bad layout");
    SF_REQUIRE((g.b_layout != SF_LAYOUT_SPLIT_KOCT && g.c_f16 != 4) || g.precision == SF_PRECISION_F16X3,
               "sf_gemm: a split k-octet operand / result (SF_LAYOUT_SPLIT_KOCT, c_f16 = 4) needs SF_PRECISION_F16X3");
    SF_REQUIRE((g.b_layout != SF_LAYOUT_F16_K_MAJOR && g.b_layout != SF_LAYOUT_F16_KOCT) ||
                   g.precision == SF_PRECISION_F16X2 || g.precision == SF_PRECISION_F16,
               "sf_gemm: a stored-fp16 activation operand needs SF_PRECISION_F16X2 or SF_PRECISION_F16");
    SF_REQUIRE(g.c_f16 >= 0 && g.c_f16 <= 4, "sf_gemm: c_f16 must be 0 .. 4");
    SF_REQUIRE(g.c_f16 != 3 || g.C16, "sf_gemm: c_f16 = 3 needs C16");
    SF_REQUIRE(!g.c_f16 || g.precision != SF_PRECISION_FP32, "sf_gemm: c_f16 needs a split precision");
    SF_REQUIRE(!g.r_f16 || g.precision != SF_PRECISION_FP32, "sf_gemm: r_f16 needs a split precision");
    SF_REQUIRE(g.b_layout != SF_LAYOUT_F16_K_MINOR || g.precision != SF_PRECISION_FP32,
               "sf_gemm: a stored-fp16 B operand needs a split precision");
    SF_REQUIRE(g.a_layout != SF_LAYOUT_SPLIT_F16 || g.precision != SF_PRECISION_FP32,
               "sf_gemm: SPLIT_F16 weights need precision F16X3");
    SF_REQUIRE(g.epilogue >= SF_EPI_NONE && g.epilogue <= SF_EPI_AXPY, "sf_gemm: bad epilogue %d", g.epilogue);
    SF_REQUIRE(g.precision >= SF_PRECISION_FP32 && g.precision <= SF_PRECISION_F16,
               "sf_gemm: precision %d not supported", g.precision);
    if (g.epilogue == SF_EPI_RES || g.epilogue == SF_EPI_RES_GELU || g.epilogue == SF_EPI_RES_GELU_DW1 ||
        g.epilogue == SF_EPI_AXPY)
        SF_REQUIRE(g.R != nullptr, "sf_gemm: epilogue %d needs R", g.epilogue);
    if (g.epilogue == SF_EPI_RES_GELU_DW1) SF_REQUIRE(g.dw_w && g.dw_b, "sf_gemm: DW1 epilogue needs dw_w/dw_b");
    if (g.epilogue == SF_EPI_AXPY) SF_REQUIRE(g.gamma != nullptr, "sf_gemm: AXPY epilogue needs gamma");
    if (g.conv3x3) {
        SF_REQUIRE(g.b_layout == SF_LAYOUT_K_MAJOR && g.b_group == 0, "sf_gemm: conv3x3 needs a plain K-major B");
        SF_REQUIRE(g.K % 9 == 0 && g.h > 0 && g.w > 0 && g.h * g.w == g.N, "sf_gemm: conv3x3 needs K=9*Cin, h*w=N");
    }
    if (g.b_group) SF_REQUIRE(g.b_layout == SF_LAYOUT_K_MAJOR || g.b_layout == SF_LAYOUT_F16_KOCT,
                              "sf_gemm: b_group needs a K-major or k-octet B");
    SF_REQUIRE(g.k_splits <= 1 || g.precision != SF_PRECISION_FP32, "sf_gemm: split-K is only built for the split-precision modes");
    SF_REQUIRE(sf::epilogue_spans_ok(g), "sf_gemm: C / R image larger than 1 GiB (32-bit buffer offsets in the epilogue)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (g.precision != SF_PRECISION_FP32) return sf::gemm_split_dispatch(g, st);
    const bool kminor = (g.a_layout == SF_LAYOUT_K_MINOR) || (g.b_layout == SF_LAYOUT_K_MINOR);
    const int bm = pick_bm(g);
    if (kminor) {   // k-contiguous operands: deeper k-tile so each row contributes a full 128-byte line
        if (bm == 128) return launch_cfg<2, 2, 2, 2, 32>(g, st);
        if (bm == 64) return launch_cfg<1, 4, 2, 1, 32>(g, st);
        return launch_cfg<1, 4, 1, 1, 32>(g, st);
    }
    if (bm == 128) return launch_cfg<2, 2, 2, 2, 16>(g, st);
    if (bm == 64) return launch_cfg<1, 4, 2, 1, 16>(g, st);
    return launch_cfg<1, 4, 1, 1, 16>(g, st);
}
