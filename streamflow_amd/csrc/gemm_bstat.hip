// "B-stationary" GEMM for the update block's 1x1 convolutions:  C[M x N] = epi(alpha * (W[M x K] X[K x N] + bias)),  N = pixels >> M, K.
//
// Why this kernel exists (round 4, measured with the phase timers of the tiled kernels): the 128 x 256
// tile kernel spends half of every k-stage in `s_waitcnt vmcnt` (1300 of 2500 cycles per stage at M960 K640).  All M / 128 row
// tiles of one pixel tile run at the same time on different CUs and request the SAME activation bytes: 7 of 8 requests in
// flight are duplicates that merge in the L2 and all of them wait one HBM round trip (~2 us) per stage -- the activation
// stream ran at 0.5 TB/s.  An update-block GEMM is a tall-skinny product: few weights (<= 1.2 MB, L2 resident) against
// hundreds of MB of activations that are touched once.  So the roles are swapped:
//   * a WAVE owns 32 pixels and keeps their K activation values IN REGISTERS for its whole life, already in the byte image
//     of the MFMA B operand (K / 16 fragments of 4 VGPRs: K = 640 -> 160 VGPRs).  They are loaded once, straight from HBM,
//     with every load of the wave in flight together (40 KB per wave, no duplicates anywhere on the chip);
//   * the WEIGHTS stream past them: the workgroup (4 waves = 128 pixels) walks all M rows in steps of 64, pulling 16-KB
//     weight stages (64 rows x 64 k of hi and lo, or x 128 k of hi alone) L2 -> LDS by `buffer_load ... lds` into a ring of
//     three, one barrier per stage = per 16 MFMAs of a wave; every wave reads every
//     weight fragment from LDS (1 KB per MFMA and wave = half the LDS read rate at full MFMA rate);
//   * two workgroups per CU (<= 256 VGPRs): the load burst of one runs under the MFMA loop of the other.
// The activation operand may be fp16 k-octet planes (one 16-byte load per fragment), fp32 planes or fp16 rows (eight loads
// per fragment, converted / packed once) -- it is read exactly once per pixel tile, so its format hardly matters any more.
// Epilogues as sf_gemm's (bias, GELU, residual, depthwise 1 x 1, AXPY), straight from the accumulator layout: a lane holds 4
// consecutive rows of one pixel = 8 bytes of a k-octet, or four dwords of four 128-byte row segments.
#include "sf_common.h"
#include <type_traits>
#include <cstdlib>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;
using sf::f32x2;

constexpr int kThreads = 256;          // 4 waves x 32 pixels
constexpr int BN = 128;
constexpr int TM = 2;                  // 32-row tiles per m-step (64 rows)
// 16-deep k-steps per weight stage: 64 k for two products (hi + lo planes), 128 k for one -- 16 KB and 16 MFMAs per wave either way
template <int PM> struct StageK { static constexpr int value = (PM == 1) ? 8 : 4; };
#ifndef SF_BSTAT_RING
#define SF_BSTAT_RING 3
#endif
constexpr int RING = SF_BSTAT_RING;    // weight stages in LDS
constexpr int kParamRows = 1024;       // rows of the LDS copy of bias / depthwise scale / shift
constexpr int kOob = 1 << 30;          // byte offset beyond every buffer range (host-checked spans < 2^30)

struct BsArgs {
    SfGemm g;
    int a_bytes;          // bytes of one weight plane (hi = lo): [K padded to 64 / 8][lda_h][8] halves
    int nst;              // stages per m-step = ceil(K / (16 SKS))
    int msteps;           // ceil(M / 64)
    int msplit;           // grid.y: ranges of m-steps per pixel tile (small grids only)
    int ntile;            // pixel tiles per image
    int e_ops;            // vector memory operations of one epilogue: stores + the residual loads of the next m-step
#ifdef SF_BSTAT_TIMERS
    long long* ts;        // SF_GEMM_TS_BUF: per-wave phase cycle sums (tools/gemm_bs_timers.py; -DSF_BSTAT_TIMERS builds only)
#endif
};
#ifdef SF_BSTAT_TIMERS
#define SF_BS_STAMP(acc_) { const long long t_ = __builtin_readcyclecounter(); acc_ += t_ - tprev; tprev = t_; }
#else
#define SF_BS_STAMP(acc_)
#endif

template <int N>
__device__ __forceinline__ void wait_vm() {                 // s_waitcnt vmcnt(N) only (expcnt / lgkmcnt untouched)
    __builtin_amdgcn_s_waitcnt((N & 15) | 0x0F70 | ((N >> 4) << 14));
}
// vmcnt(P + e) for the wave-uniform epilogue size e: the first two stages of an m-step have the previous epilogue's stores
// (and the next residual loads) BEHIND the weight pieces they wait for; a smaller count would wait for those stores too
template <int P>
__device__ __forceinline__ void wait_vm_epi(int e) {
    switch (e) {
        case 0: wait_vm<P>(); break;
        case 8: wait_vm<P + 8>(); break;
        case 16: wait_vm<P + 16>(); break;
        case 32: wait_vm<P + 32>(); break;
        case 40: wait_vm<P + 40>(); break;
        case 48: wait_vm<P + 48>(); break;
        default: wait_vm<(P + 56 > 63 ? 63 : P + 56)>(); break;      // e >= 56
    }
}

template <int EPI, bool kFast>
__device__ __forceinline__ f32x2 epi2(f32x2 v, f32x2 r, f32x2 dww, f32x2 dwb, float gam) {
    if (EPI == SF_EPI_GELU) return sf::gelu2<kFast>(v);
    if (EPI == SF_EPI_RELU) return __builtin_elementwise_max(v, sf::splat2(0.f));
    if (EPI == SF_EPI_RES) return r + v;
    if (EPI == SF_EPI_RES_GELU) return sf::gelu2<kFast>(r + v);
    if (EPI == SF_EPI_RES_GELU_DW1) {
        const f32x2 t = sf::gelu2<kFast>(r + v);
        return sf::gelu2<kFast>(t + (dww * t + dwb));
    }
    if (EPI == SF_EPI_AXPY) return r + sf::splat2(gam) * v;
    return v;
}

// NKS: k-steps of 16 whose activation fragments the wave holds (K <= 16 NKS).  PM: MFMA products per element (2: weights hi +
// lo, 1: hi only).  RES: 0 = no residual, 1 = fp32 planes, 2 = fp16 k-octet image (SfGemm.r_f16 = 2).
template <int NKS, int PM, int RES>
__global__ __launch_bounds__(kThreads, (PM == 1 && NKS <= 16 && RES == 0) ? 4 : 2) void gemm_bstat_kernel(const BsArgs a) {
    const SfGemm& g = a.g;
    constexpr int SKS = StageK<PM>::value;
    constexpr int NST = NKS / SKS;
    constexpr int OCT = SKS * 2;                          // k-octets of a stage
    constexpr int kPlane = OCT * 64 * 16;                 // bytes of one plane of a stage: OCT octets x 64 rows x 16 B
    constexpr int kStage = PM * kPlane;                   // 16 KB
    constexpr int kPieces = kStage / 1024 / 4;            // 1-KB DMA pieces per wave and stage (4)
    __shared__ __attribute__((aligned(1024))) char smem[RING * kStage + 3 * kParamRows * 4];
    float* sbias = reinterpret_cast<float*>(smem + RING * kStage);
    float* sdww = sbias + kParamRows;
    float* sdwb = sdww + kParamRows;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, l31 = lane & 31;
    const int tile = blockIdx.x % a.ntile, z = blockIdx.x / a.ntile;
    const int n = tile * BN + wave * 32 + l31, nc = min(n, g.N - 1);
    const int ms_beg = (int)((int64_t)a.msteps * blockIdx.y / a.msplit), ms_end = (int)((int64_t)a.msteps * (blockIdx.y + 1) / a.msplit);
    const int nst = a.nst;

#ifdef SF_BSTAT_TIMERS
    const long long ts0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
    long long tw = 0, tb = 0, ti = 0, tm = 0, te = 0, tprev = ts0;
#endif
    // ---- the wave's activations: B fragments of its 32 pixels for every k-step, loaded once -------------------------------
    f16x8 b[NKS];
    if (g.b_layout == SF_LAYOUT_F16_KOCT) {
        const int noct = (g.K + 7) / 8;
        const int goct = g.b_group > 0 ? g.b_group / 8 : 0;
        const int64_t span = goct ? ((int64_t)((noct - 1) / goct) * g.b_group_stride * 2 + (int64_t)((noct - 1) % goct + 1) * g.ldb * 16)
                                  : (int64_t)noct * g.ldb * 16;
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(g.B)) + (int64_t)z * g.strideB * 2, 0, (int)span, 0x00020000);
        const int vo = (khalf * (int)g.ldb + nc) * 16;
        // scalar offset of octet 2 ks, advanced incrementally (grouped rows: groups of goct octets, b_group_stride halves apart)
        int so = 0, oin = 0, gbase = 0;
        const int ostep = 2 * (int)g.ldb * 16, gstep = (int)(g.b_group_stride * 2);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks < nst * SKS) {                                        // (wave-uniform)
                // octets past K: out of range through the checked (vector) offset -> zeros
                b[ks] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rb, (2 * ks + khalf < noct) ? vo : kOob, so, 0));
                so += ostep; oin += 2;
                if (goct && oin == goct) { oin = 0; gbase += gstep; so = gbase; }
            }
        }
    } else if (g.b_layout == SF_LAYOUT_K_MAJOR) {
        const int kl = g.K - 1;
        const int64_t last = (g.b_group > 0) ? (int64_t)(kl / g.b_group) * g.b_group_stride + (int64_t)(kl % g.b_group) * g.ldb
                                             : (int64_t)kl * g.ldb;
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(g.B) + (int64_t)z * g.strideB, 0, (int)((last + g.N) * 4), 0x00020000);
        const int vo = (khalf * 8 * (int)g.ldb + nc) * 4;
        int so = 0, kin = 0, gbase = 0;                                  // offset of row 16 ks (khalf = 1 lanes: + 8 rows, same group)
        const int rstep = (int)g.ldb * 4, gstep = (int)(g.b_group_stride * 4);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks < nst * SKS) {
                f16x8 f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned u = __builtin_amdgcn_raw_buffer_load_b32(rb, (ks * 16 + i + 8 * khalf < g.K) ? vo : kOob, so + i * rstep, 0);
                    f[i] = (_Float16)__builtin_bit_cast(float, u);       // round to nearest: what a producer storing fp16 would hand over
                }
                b[ks] = f;
                so += 16 * rstep; kin += 16;
                if (g.b_group > 0 && kin == g.b_group) { kin = 0; gbase += gstep; so = gbase; }
            }
        }
    } else {                                                             // SF_LAYOUT_F16_K_MAJOR: fp16 rows [K][ldb]
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(g.B)) + (int64_t)z * g.strideB * 2, 0,
            (int)(((int64_t)(g.K - 1) * g.ldb + g.N) * 2), 0x00020000);
        const int vo = (khalf * 8 * (int)g.ldb + nc) * 2;
        const int rstep = (int)g.ldb * 2;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks < nst * SKS) {
                f16x8 f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned short u = __builtin_amdgcn_raw_buffer_load_b16(rb, (ks * 16 + i + 8 * khalf < g.K) ? vo : kOob, (ks * 16 + i) * rstep, 0);
                    f[i] = __builtin_bit_cast(_Float16, u);
                }
                b[ks] = f;
            }
        }
    }

    // ---- weights by LDS-DMA: stage (m-step m, stage s) = rows 64 m .. 64 m + 63 x octets 8 s .. 8 s + 7 of each plane -------
#ifdef SF_BSTAT_REPL      // experiment: SF_BSTAT_REPL copies of the weight planes back to back, workgroups spread over them
    const int64_t repl_off = (int64_t)((blockIdx.x >> 3) % SF_BSTAT_REPL) * a.a_bytes;
#else
    const int64_t repl_off = 0;
#endif
    const __amdgpu_buffer_rsrc_t rah = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(g.A_hi)) + repl_off, 0, a.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ral = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(g.A_lo)) + repl_off, 0, a.a_bytes, 0x00020000);
    auto issue_a = [&](int m, int s, int slot) {
#pragma unroll
        for (int i = 0; i < kPieces; ++i) {
            const int idx = wave + 4 * i, p = idx / OCT, o = idx % OCT;  // (plane, octet of the stage): wave-uniform
            const int so = ((s * OCT + o) * (int)g.lda_h + m * 64) * 16;
            char* dst = smem + slot * kStage + p * kPlane + o * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds((PM == 2 && p) ? ral : rah, (lds_ptr)dst, 16, lane * 16, so, 0, 0);
        }
    };
    // position of the stage that is requested next (two ahead of the one being multiplied); past the end it stays on the
    // last valid stage: the request count behind every wait must be the same on every trip
    int ma = ms_beg, sa = 0;
    auto advance = [&]() {
        const bool wrap = (sa + 1 == nst);
        sa = wrap ? 0 : sa + 1;
        ma = (wrap && ma + 1 < ms_end) ? ma + 1 : ma;
    };
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) {
        issue_a(ma, sa, i);
        advance();
    }

    // ---- epilogue parameters into LDS (rows >= M: zeros) ----
    for (int i = tid; i < a.msteps * 64; i += kThreads) {
        const bool in = i < g.M;
        sbias[i] = (in && g.bias) ? g.bias[i] : 0.f;
        if (g.epilogue == SF_EPI_RES_GELU_DW1) {
            sdww[i] = in ? g.dw_w[i] : 0.f;
            sdwb[i] = in ? g.dw_b[i] : 0.f;
        }
    }
    const float gam = (g.epilogue == SF_EPI_AXPY) ? g.gamma[0] : 0.f;

    // ---- residual: fetched at the top of an m-step, used by its epilogue ----
    __amdgpu_buffer_rsrc_t rr = rah;
    if (RES == 1) {
        const int mr = g.M - 1;
        const int64_t last = (g.r_group > 0) ? (int64_t)(mr / g.r_group) * g.r_group_stride + (int64_t)(mr % g.r_group) * g.ldr
                                             : (int64_t)mr * g.ldr;
        rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.R) + (int64_t)z * g.strideR, 0, (int)((last + g.N) * 4), 0x00020000);
    } else if (RES == 2) {
        rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(g.R)) + (int64_t)z * g.strideR * 2, 0,
                                               (int)((int64_t)((g.M + 7) / 8) * g.ldr * 16), 0x00020000);
    }
    float rf[(RES == 1) ? TM : 1][16];
    u32x2 rk[(RES == 2) ? TM : 1][4];
    auto load_res = [&](int m) {
        if (RES == 1) {
            const int vo = (4 * khalf * (int)g.ldr + nc) * 4;
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                // the 32 rows of a tile lie in one group (r_group % 32 == 0): one division per tile; rows >= M are dropped by the
                // range check of the vector offset, whatever their scalar offset points at
                const int row0 = m * 64 + t * 32;
                const int base = (g.r_group > 0) ? (int)(((int64_t)(row0 / g.r_group) * g.r_group_stride + (int64_t)(row0 % g.r_group) * g.ldr) * 4)
                                                 : row0 * (int)g.ldr * 4;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);                                // (+ 4 khalf per lane)
                    rf[t][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        rr, (4 * khalf < __builtin_amdgcn_readfirstlane(g.M - row0 - dr)) ? vo : kOob, base + dr * (int)g.ldr * 4, 0));
                }
            }
        } else if (RES == 2) {
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int mr = m * 64 + t * 32 + 8 * j;
                    rk[t][j] = __builtin_amdgcn_raw_buffer_load_b64(rr, (4 * khalf < __builtin_amdgcn_readfirstlane(g.M - mr)) ? (nc * 16 + khalf * 8) : kOob,
                                                                    (mr >> 3) * (int)g.ldr * 16, 0);
                }
        }
    };

    // ---- output descriptors ----
    const int ces = 4;
    const __amdgpu_buffer_rsrc_t rc32 = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(g.C) + (int64_t)z * g.strideC * ces, 0, (g.c_f16 == 2) ? 0 : (int)(((int64_t)(g.M - 1) * g.ldc + g.N) * 4), 0x00020000);
    _Float16* c16 = (g.c_f16 == 2) ? reinterpret_cast<_Float16*>(g.C) + (int64_t)z * g.strideC
                                   : reinterpret_cast<_Float16*>(g.C16) + (int64_t)z * g.strideC16;
    const __amdgpu_buffer_rsrc_t rc16 = __builtin_amdgcn_make_buffer_rsrc(
        c16, 0, (g.c_f16 >= 2) ? (int)((int64_t)((g.M + 7) / 8) * g.ldc * 16) : 0, 0x00020000);

    f32x16 acc[TM];
    // per-lane parts of the store offsets (pixel, k-half); the row part is wave-uniform and travels in the scalar offset
    const int lane_c32 = (n < g.N) ? (4 * khalf * (int)g.ldc + n) * 4 : kOob;             // fp32 planes: + row * ldc * 4
    const int lane_k16 = (n < g.N) ? n * 16 + khalf * 8 : kOob;                           // k-octets: + (row / 8) * ldc * 16
    const int ldc4 = (int)g.ldc * 4;
    const int k4 = 4 * khalf;
    // "row rb + e + 4 khalf < M" as  k4 < (M - rb - e)  with the right side forced into a scalar register: written the other
    // way round, hipcc hoists one per-lane constant PER (tile, group, element) out of the m-loop -- 40 VGPRs of them
    auto rows_left = [&](int r) { return __builtin_amdgcn_readfirstlane(g.M - r); };
    // CF = SfGemm.c_f16 (0: fp32 planes, 2: k-octets, 3: both); results that leave as fp16 ONLY take the polynomial GELU of the
    // two-product modes, like the tiled kernels.  The bias is already in the accumulators (m-step start).
    auto epilogue = [&](int m, auto epi_tag, auto cf_tag) {
        constexpr int EPI = decltype(epi_tag)::value;
        constexpr int CF = decltype(cf_tag)::value;
        constexpr bool kFast = (CF == 2);
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rb = m * 64 + t * 32 + 8 * j;                                   // (wave-uniform) this lane's rows: rb + 4 khalf + 0..3
                f32x4 dw4 = {0.f, 0.f, 0.f, 0.f}, db4 = {0.f, 0.f, 0.f, 0.f};
                if (EPI == SF_EPI_RES_GELU_DW1) {
                    dw4 = *reinterpret_cast<const f32x4*>(sdww + rb + 4 * khalf);
                    db4 = *reinterpret_cast<const f32x4*>(sdwb + rb + 4 * khalf);
                }
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    f32x2 v, r = {0.f, 0.f}, dww, dwb;
                    v[0] = g.alpha * acc[t][4 * j + e];
                    v[1] = g.alpha * acc[t][4 * j + e + 1];
                    if (RES == 1) { r[0] = rf[t][4 * j + e]; r[1] = rf[t][4 * j + e + 1]; }
                    if (RES == 2) {
                        const unsigned u = rk[t][j][e >> 1];
                        const f16x2 h = __builtin_bit_cast(f16x2, u);
                        r[0] = (float)h[0]; r[1] = (float)h[1];
                    }
                    dww[0] = dw4[e]; dww[1] = dw4[e + 1]; dwb[0] = db4[e]; dwb[1] = db4[e + 1];
                    const f32x2 res = epi2<EPI, kFast>(v, r, dww, dwb, gam);
                    o[e] = res[0]; o[e + 1] = res[1];
                }
                if (CF != 2) {                                                            // fp32 planes: 128-byte row segments
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o[e]), rc32,
                                                              (k4 < rows_left(rb + e)) ? lane_c32 : kOob, (rb + e) * ldc4, 0);
                }
                if (CF >= 2) {                                                            // k-octets: 8 bytes = 4 rows of pixel n
                    f16x4 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = (_Float16)o[e];
                    const int so = (rb >> 3) * ldc4 * 4;
                    // CF = 3: rows >= M of a last octet belong to someone else (flow rows of the motion features)
                    const bool full = (CF == 2) ? (k4 < rows_left(rb)) : (k4 < rows_left(rb + 3));
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h), rc16, full ? lane_k16 : kOob, so, 0);
                    if (CF == 3 && (g.M & 3) && rb + 8 > g.M && rb < g.M) {               // (wave-uniform) partial last group: row by row
#pragma unroll
                        for (int e = 0; e < 3; ++e) {
                            const _Float16 he = h[e];                    // (bit_cast of a vector element lvalue reads element 0)
                            const bool part = k4 < rows_left(rb) && !(k4 < rows_left(rb + 3)) && k4 < rows_left(rb + e);
                            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, he), rc16,
                                                                  part ? lane_k16 + e * 2 : kOob, so, 0);
                        }
                    }
                }
                // one group at a time: scheduled together, the eight groups of an m-step keep ~50 more registers alive than
                // the activation fragments leave room for (K = 640: spills)
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    auto run_cf = [&](int m, auto epi_tag) {
        using std::integral_constant;
        constexpr int EPI = decltype(epi_tag)::value;
        constexpr bool kKoctOnly = (EPI == SF_EPI_NONE || EPI == SF_EPI_GELU || EPI == SF_EPI_RES_GELU);   // (host-checked)
        if (kKoctOnly && g.c_f16 == 2) epilogue(m, epi_tag, integral_constant<int, 2>{});
        else if (g.c_f16 == 3) epilogue(m, epi_tag, integral_constant<int, 3>{});
        else epilogue(m, epi_tag, integral_constant<int, 0>{});
    };
    auto run_epilogue = [&](int m) {
        using std::integral_constant;
        switch (g.epilogue) {                                                             // wave-uniform
            case SF_EPI_GELU: run_cf(m, integral_constant<int, SF_EPI_GELU>{}); break;
            case SF_EPI_RELU: run_cf(m, integral_constant<int, SF_EPI_RELU>{}); break;
            case SF_EPI_RES: if (RES) run_cf(m, integral_constant<int, SF_EPI_RES>{}); break;
            case SF_EPI_RES_GELU: if (RES) run_cf(m, integral_constant<int, SF_EPI_RES_GELU>{}); break;
            case SF_EPI_RES_GELU_DW1: if (RES) run_cf(m, integral_constant<int, SF_EPI_RES_GELU_DW1>{}); break;
            case SF_EPI_AXPY: if (RES) run_cf(m, integral_constant<int, SF_EPI_AXPY>{}); break;
            default: run_cf(m, integral_constant<int, SF_EPI_NONE>{}); break;
        }
    };

    // everything requested so far has landed (activations, the first two weight stages) and the parameters are visible
    wait_vm<0>();
    __syncthreads();
#ifdef SF_BSTAT_TIMERS
    const long long ts1 = __builtin_readcyclecounter();
    tprev = ts1;
#endif

    int slot = 0;
    const char* sa_base = smem + (khalf * 64 + l31) * 16;
    for (int m = ms_beg; m < ms_end; ++m) {
        // accumulators start at the bias (LDS copy; rows >= M: zero): v = alpha * (acc + bias) needs no parameter in the epilogue
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + m * 64 + t * 32 + 8 * j + 4 * khalf);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t][4 * j + e] = b4[e];
            }
        load_res(m);
        __builtin_amdgcn_sched_barrier(0);                         // (the counted waits below assume this issue order)
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            if (s < nst) {                                                                // (wave-uniform)
                // this wave's pieces of the stage have landed: behind them in the queue are the pieces of the next stage and,
                // in the first two stages of an m-step, the previous epilogue's stores / this m-step's residual loads
                SF_BS_STAMP(tm)
                if (s < RING - 1) wait_vm_epi<kPieces * (RING - 2)>(a.e_ops);
                else wait_vm<kPieces * (RING - 2)>();
                SF_BS_STAMP(tw)
                __builtin_amdgcn_s_barrier();                      // ... everyone's; and nobody reads the previous slot any more
                SF_BS_STAMP(tb)
                issue_a(ma, sa, slot == 0 ? RING - 1 : slot - 1);  // RING - 1 stages ahead, into the slot just released
                advance();
                SF_BS_STAMP(ti)
                __builtin_amdgcn_sched_barrier(0);
                const char* sp = sa_base + slot * kStage;
                // the stage's MFMAs: fragment f = (k-step, product, tile) is read once and used by ONE MFMA
#pragma unroll
                for (int ks = 0; ks < SKS; ++ks) {
                    f16x8 ah[TM], al[TM];
                    if (PM == 2) {
#pragma unroll
                        for (int t = 0; t < TM; ++t) al[t] = *reinterpret_cast<const f16x8*>(sp + kPlane + ks * 2048 + t * 512);
                    }
#pragma unroll
                    for (int t = 0; t < TM; ++t) ah[t] = *reinterpret_cast<const f16x8*>(sp + ks * 2048 + t * 512);
#pragma unroll
                    for (int t = 0; t < TM; ++t) {
                        if (PM == 2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], b[s * SKS + ks], acc[t], 0, 0, 0);
                    }
#pragma unroll
                    for (int t = 0; t < TM; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], b[s * SKS + ks], acc[t], 0, 0, 0);
                }
                // issue order, pinned: fragment reads run kAhead MFMAs ahead of their use (left to itself hipcc keeps ONE set of
                // fragment registers and waits out the LDS latency in front of every k-step: 31 % MFMA utilisation measured)
                {
                    constexpr int kMfma = SKS * PM * TM, kAhead = (NKS >= 40 ? 1 : 2) * PM * TM;
                    __builtin_amdgcn_sched_group_barrier(0x100, kAhead, 0);
#pragma unroll
                    for (int i = 0; i < kMfma; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (i + kAhead < kMfma) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                slot = (slot == RING - 1) ? 0 : slot + 1;
            }
        }
        SF_BS_STAMP(tm)
        run_epilogue(m);
        SF_BS_STAMP(te)
    }
    wait_vm<0>();                                                  // (pieces requested past the end must land before the LDS is released)
#ifdef SF_BSTAT_TIMERS
    if (a.ts && lane == 0 && blockIdx.x < 4096) {
        long long* d = a.ts + ((int64_t)blockIdx.x * 4 + wave) * 8;
        d[0] = ts1 - ts0; d[1] = tw; d[2] = tb; d[3] = ti; d[4] = tm; d[5] = te; d[6] = __builtin_readcyclecounter() - ts0;
        d[7] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
#endif
}

template <int NKS, int PM>
int launch_res(const BsArgs& a, dim3 grid, hipStream_t st) {
    const SfGemm& g = a.g;
    const bool needs_r = g.epilogue == SF_EPI_RES || g.epilogue == SF_EPI_RES_GELU || g.epilogue == SF_EPI_RES_GELU_DW1 ||
                         g.epilogue == SF_EPI_AXPY;
    if (!needs_r) hipLaunchKernelGGL((gemm_bstat_kernel<NKS, PM, 0>), grid, dim3(kThreads), 0, st, a);
    else if (g.r_f16 == 2) hipLaunchKernelGGL((gemm_bstat_kernel<NKS, PM, 2>), grid, dim3(kThreads), 0, st, a);
    else if constexpr (NKS <= 32) hipLaunchKernelGGL((gemm_bstat_kernel<NKS, PM, 1>), grid, dim3(kThreads), 0, st, a);
    else return sf::fail(SF_ERR_UNSUPPORTED, "sf_gemm(B-stationary): an fp32 residual needs K <= 512 (register budget)");
    return sf::check_launch("sf_gemm(B-stationary)");
}

template <int PM>
int launch_nks(const BsArgs& a, dim3 grid, hipStream_t st) {
    const int nks = a.nst * StageK<PM>::value;
    if (nks <= 8) return launch_res<8, PM>(a, grid, st);
    if (nks <= 16) return launch_res<16, PM>(a, grid, st);
    if (nks <= 24) return launch_res<24, PM>(a, grid, st);
    if (nks <= 32) return launch_res<32, PM>(a, grid, st);
    return launch_res<40, PM>(a, grid, st);
}

}  // namespace

namespace sf {

// Can this problem run on the B-stationary kernel?  (K <= 640 held in registers; pre-split weights padded to 64 in K;
// outputs as fp32 planes and / or k-octets; every epilogue; no implicit 3x3, no split-K.)
bool gemm_bstat_ok(const SfGemm& g) {
    if (g.precision != SF_PRECISION_F16X2 && g.precision != SF_PRECISION_F16) return false;
    if (g.a_layout != SF_LAYOUT_SPLIT_F16 || g.conv3x3 || g.k_splits > 1) return false;
    if (g.b_layout != SF_LAYOUT_F16_KOCT && g.b_layout != SF_LAYOUT_K_MAJOR && g.b_layout != SF_LAYOUT_F16_K_MAJOR) return false;
    if (g.K <= 64 || g.K > 640 || g.M > kParamRows || g.c_f16 == 1) return false;
    if (g.c_f16 == 2 && g.epilogue != SF_EPI_NONE && g.epilogue != SF_EPI_GELU && g.epilogue != SF_EPI_RES_GELU) return false;
    if (g.b_layout == SF_LAYOUT_F16_K_MAJOR && g.b_group) return false;
    if (g.b_group % 32 || g.r_group % 32) return false;
    if (g.a_k_pad < 128 || g.a_k_pad % 128) return false;               // weight planes must reach K rounded up to 128
    const bool needs_r = g.epilogue == SF_EPI_RES || g.epilogue == SF_EPI_RES_GELU || g.epilogue == SF_EPI_RES_GELU_DW1 ||
                         g.epilogue == SF_EPI_AXPY;
    if (g.r_f16 == 2 && (!needs_r || g.r_group)) return false;
    // register budget: 4 NKS for the activations + 32 accumulators + residual (32 / 8) + fragments must stay under 256
    if (needs_r && g.r_f16 != 2 && g.K > 512) return false;
    if (g.c_f16 >= 2 && ((reinterpret_cast<uintptr_t>(g.c_f16 == 2 ? (void*)g.C : g.C16) & 15) || ((g.c_f16 == 2 ? g.strideC : g.strideC16) & 7) || g.ldc < g.N))
        return false;
    if (g.b_layout == SF_LAYOUT_F16_KOCT && ((reinterpret_cast<uintptr_t>(g.B) & 15) || (g.strideB & 7) || (g.b_group_stride & 7) || g.ldb < g.N))
        return false;
    if ((int64_t)g.ldb * 16 * 8 >= ((int64_t)1 << 30)) return false;    // 32-bit offsets of the operand loads
    return true;
}

int gemm_bstat_launch(const SfGemm& g, hipStream_t st) {
    BsArgs a;
    a.g = g;
    const int sk = (g.precision == SF_PRECISION_F16) ? 128 : 64;        // k extent of a weight stage
    const int kp = (g.K + 127) / 128 * 128;                             // extent of the planes (a_k_pad = 128)
    a.a_bytes = (int)((int64_t)kp * g.lda_h * 2);
    a.nst = (g.K + sk - 1) / sk;
    a.msteps = (g.M + 63) / 64;
    a.ntile = ceil_div(g.N, BN);
    // small grids (a single clip): cut the rows into ranges so that ~4 workgroups per CU exist; the activations are then read
    // once per range
    const int64_t wgs = (int64_t)a.ntile * g.batch;
    int msplit = 1;
    if (wgs < 768) msplit = (int)((1024 + wgs - 1) / wgs);
    if (msplit > a.msteps) msplit = a.msteps;
    a.msplit = msplit;
    const bool needs_r = g.epilogue == SF_EPI_RES || g.epilogue == SF_EPI_RES_GELU || g.epilogue == SF_EPI_RES_GELU_DW1 ||
                         g.epilogue == SF_EPI_AXPY;
    a.e_ops = (g.c_f16 == 2 ? 0 : 16 * TM) + (g.c_f16 >= 2 ? 4 * TM : 0) + (!needs_r ? 0 : (g.r_f16 == 2 ? 4 * TM : 16 * TM));
    if (g.c_f16 == 3 && (g.M & 3)) a.e_ops = 0;                          // (row-by-row tail stores: count unknown -> wait for everything)
#ifdef SF_BSTAT_TIMERS
    a.ts = getenv("SF_GEMM_TS_BUF") ? (long long*)strtoull(getenv("SF_GEMM_TS_BUF"), nullptr, 0) : nullptr;
#endif
    dim3 grid((unsigned)(a.ntile * g.batch), (unsigned)msplit);
    return (g.precision == SF_PRECISION_F16) ? launch_nks<1>(a, grid, st) : launch_nks<2>(a, grid, st);
}

}  // namespace sf
