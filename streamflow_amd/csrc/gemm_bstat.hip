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
#ifndef SF_BSTAT_ALT_ACC
#define SF_BSTAT_ALT_ACC 0      // (two accumulation chains per tile: measured 0-4 % SLOWER; kept as a switch)
#endif
constexpr int RING = SF_BSTAT_RING;    // weight stages in LDS
#ifndef SF_BSTAT_STORE_SLACK
#define SF_BSTAT_STORE_SLACK 1
#endif
constexpr int kParamRows = 1024;       // rows of the LDS copy of bias / depthwise scale / shift
constexpr int kOob = 1 << 30;          // byte offset beyond every buffer range (host-checked spans < 2^30)

struct BsArgs {
    SfGemm g;
    int a_bytes;          // bytes of one weight plane (hi = lo): [K padded to 64 / 8][lda_h][8] halves
    int nst;              // stages per m-step = ceil(K / (16 SKS))
    int msteps;           // ceil(M / 64)
    int msplit;           // row ranges per pixel tile of the SPLIT tiles (below)
    int n_main;           // pixel tiles [0, n_main) run whole in one workgroup; tiles >= n_main are cut into msplit row ranges:
                          // the partial last round of a grid that is not a multiple of the resident workgroups (and every
                          // tile of a grid too small to fill the chip)
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

// workgroup -> (pixel tile over all images, row range `part` of `nparts`)
struct WorkItem { int tile, part, nparts; };
__device__ __forceinline__ WorkItem work_item(const BsArgs& a) {
    const int w = blockIdx.x;
    WorkItem wi;
    if (w < a.n_main) { wi.tile = w; wi.part = 0; wi.nparts = 1; }
    else { wi.tile = a.n_main + (w - a.n_main) / a.msplit; wi.part = (w - a.n_main) % a.msplit; wi.nparts = a.msplit; }
    return wi;
}

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
        default:                                                     // e >= 56 (a smaller count than P + e only waits longer);
            if (e >= 56) wait_vm<(P + 56 > 63 ? 63 : P + 56)>();     // anything else is not an epilogue size this file produces:
            else wait_vm<P>();                                       // wait for the pieces AND whatever is behind them
            break;
    }
}

// GELU of NP pairs at a time with the Horner chains INTERLEAVED (coefficient loop outside, pair loop inside).  Evaluated
// pair by pair the epilogue is bound by the latency of one dependent chain -- 14 packed instructions at ~11 cycles each, 160
// cycles per pair measured with the phase timers, 2550 of an m-step's 3670 epilogue cycles -- because hipcc keeps the pairs
// apart to save registers; four chains in flight hide each other's latency.  Same operations per value as sf::gelu_poly2 /
// sf::gelu_erf2 (bit-identical results).
template <int NP>
__device__ __forceinline__ void gelu_poly_n(f32x2 (&x)[NP]) {
    f32x2 xc[NP], t[NP], p[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        xc[i][0] = __builtin_amdgcn_fmed3f(x[i][0], -4.2426405f, 4.2426405f);
        xc[i][1] = __builtin_amdgcn_fmed3f(x[i][1], -4.2426405f, 4.2426405f);
        t[i] = xc[i] * xc[i];
        p[i] = sf::splat2(1.12535e-10f);
    }
    constexpr float c[8] = {-1.074371e-08f, 4.5365834e-07f, -1.12924145e-05f, 0.0001871811f, -0.0022188f, 0.019636236f,
                            -0.13269384f, 0.79780626f};
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int i = 0; i < NP; ++i) p[i] = __builtin_elementwise_fma(p[i], t[i], sf::splat2(c[k]));
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const f32x2 h = sf::splat2(0.5f) * __builtin_elementwise_max(x[i], sf::splat2(-4.2426405f));
        x[i] = __builtin_elementwise_fma(h, xc[i] * p[i], h);
    }
}
template <int NP>
__device__ __forceinline__ void gelu_erf_n(f32x2 (&x)[NP]) {
    f32x2 z[NP], z2[NP], p[NP], q[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        z[i] = x[i] * sf::splat2(0.70710678118654752440f);
        z[i] = __builtin_elementwise_min(__builtin_elementwise_max(z[i], sf::splat2(-4.0f)), sf::splat2(4.0f));
        z2[i] = z[i] * z[i];
        p[i] = sf::splat2(-2.72614225801306e-10f);
        q[i] = sf::splat2(-1.45660718464996e-05f);
    }
    constexpr float cp[6] = {2.77068142495902e-08f, -2.10102402082508e-06f, -5.69250639462346e-05f, -7.34990630326855e-04f,
                             -2.95459980854025e-03f, -1.60960333262415e-02f};
    constexpr float cq[4] = {-2.13374055278905e-04f, -1.68282697438203e-03f, -7.37332916720468e-03f, -1.42647390514189e-02f};
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            p[i] = __builtin_elementwise_fma(p[i], z2[i], sf::splat2(cp[k]));
            if (k < 4) q[i] = __builtin_elementwise_fma(q[i], z2[i], sf::splat2(cq[k]));
        }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        f32x2 rq;
        rq[0] = __builtin_amdgcn_rcpf(q[i][0]);
        rq[1] = __builtin_amdgcn_rcpf(q[i][1]);
        const f32x2 e = (z[i] * p[i]) * rq;
        x[i] = sf::splat2(0.5f) * __builtin_elementwise_max(x[i], sf::splat2(-5.6568542f)) * (sf::splat2(1.0f) + e);
    }
}
template <bool kFast, int NP>
__device__ __forceinline__ void gelu_n(f32x2 (&x)[NP]) {
    if constexpr (kFast) gelu_poly_n<NP>(x);
    else gelu_erf_n<NP>(x);
}

// epilogue arithmetic on NP pairs (v in / result out; r = residual, dww / dwb = depthwise 1 x 1 scale / shift)
template <int EPI, bool kFast, int NP>
__device__ __forceinline__ void epi_n(f32x2 (&v)[NP], const f32x2 (&r)[NP], const f32x2 (&dww)[NP], const f32x2 (&dwb)[NP], float gam) {
    if (EPI == SF_EPI_GELU) { gelu_n<kFast, NP>(v); return; }
    if (EPI == SF_EPI_RELU) {
#pragma unroll
        for (int i = 0; i < NP; ++i) v[i] = __builtin_elementwise_max(v[i], sf::splat2(0.f));
        return;
    }
    if (EPI == SF_EPI_RES || EPI == SF_EPI_RES_GELU || EPI == SF_EPI_RES_GELU_DW1) {
#pragma unroll
        for (int i = 0; i < NP; ++i) v[i] = r[i] + v[i];
        if (EPI == SF_EPI_RES) return;
        gelu_n<kFast, NP>(v);
        if (EPI == SF_EPI_RES_GELU) return;
#pragma unroll
        for (int i = 0; i < NP; ++i) v[i] = v[i] + (dww[i] * v[i] + dwb[i]);
        gelu_n<kFast, NP>(v);
        return;
    }
    if (EPI == SF_EPI_AXPY) {
#pragma unroll
        for (int i = 0; i < NP; ++i) v[i] = r[i] + sf::splat2(gam) * v[i];
    }
}

// The wave's activations: B fragments (8 consecutive k of one pixel per lane and k-half) of its 32 pixels for k-steps
// 0 .. nks_rt - 1, loaded once.  Values past K read as zero (range check of the vector offset).
template <int NKS>
__device__ __forceinline__ void load_b_frags(const SfGemm& g, int z, int nc, int khalf, int nks_rt, f16x8 (&b)[NKS]) {
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
            if (ks < nks_rt) {                                        // (wave-uniform)
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
            if (ks < nks_rt) {
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
            // (four k-steps = 32 loads in flight at a time: left alone, hipcc requests every row of the tile before the first
            // conversion and needs a register per row -- at K = 640 that alone exhausts the budget of the whole kernel)
            if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    } else {                                                             // SF_LAYOUT_F16_K_MAJOR: fp16 rows [K][ldb]
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(g.B)) + (int64_t)z * g.strideB * 2, 0,
            (int)(((int64_t)(g.K - 1) * g.ldb + g.N) * 2), 0x00020000);
        const int vo = (khalf * 8 * (int)g.ldb + nc) * 2;
        const int rstep = (int)g.ldb * 2;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks < nks_rt) {
                f16x8 f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned short u = __builtin_amdgcn_raw_buffer_load_b16(rb, (ks * 16 + i + 8 * khalf < g.K) ? vo : kOob, (ks * 16 + i) * rstep, 0);
                    f[i] = __builtin_bit_cast(_Float16, u);
                }
                b[ks] = f;
            }
            if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }

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
    const WorkItem wi = work_item(a);
    const int tile = wi.tile % a.ntile, z = wi.tile / a.ntile;
    const int n = tile * BN + wave * 32 + l31, nc = min(n, g.N - 1);
    const int ms_beg = (int)((int64_t)a.msteps * wi.part / wi.nparts), ms_end = (int)((int64_t)a.msteps * (wi.part + 1) / wi.nparts);
    const int nst = a.nst;

#ifdef SF_BSTAT_TIMERS
    const long long ts0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
    long long tw = 0, tb = 0, ti = 0, tm = 0, te = 0, tprev = ts0;
#endif
    // ---- the wave's activations: B fragments of its 32 pixels for every k-step, loaded once -------------------------------
    f16x8 b[NKS];
    load_b_frags<NKS>(g, z, nc, khalf, nst * SKS, b);

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
    auto rows_left8 = [&](int r) { return __builtin_amdgcn_readfirstlane((g.M + 7) / 8 * 8 - r); };      // up to the end of the last octet
    // CF = SfGemm.c_f16 (0: fp32 planes, 2: k-octets, 3: both); results that leave as fp16 ONLY take the polynomial GELU of the
    // two-product modes, like the tiled kernels.  The bias is already in the accumulators (m-step start).
    auto epilogue = [&](int m, auto epi_tag, auto cf_tag) {
        constexpr int EPI = decltype(epi_tag)::value;
        constexpr int CF = decltype(cf_tag)::value;
        constexpr bool kFast = (CF == 2);
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int j0 = 0; j0 < 4; j0 += 2) {
                // two row groups = 8 values = 4 pairs per pass (interleaved GELU chains); group j: rows rb + 4 khalf + 0..3
                f32x2 v[4], r[4], dww[4], dwb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = j0 + (q >> 1), e = (q & 1) * 2, rb = m * 64 + t * 32 + 8 * j;
                    v[q][0] = g.alpha * acc[t][4 * j + e];
                    v[q][1] = g.alpha * acc[t][4 * j + e + 1];
                    r[q] = sf::splat2(0.f); dww[q] = sf::splat2(0.f); dwb[q] = sf::splat2(0.f);
                    if (RES == 1) { r[q][0] = rf[t][4 * j + e]; r[q][1] = rf[t][4 * j + e + 1]; }
                    if (RES == 2) {
                        const unsigned u = rk[t][j][e >> 1];
                        const f16x2 h = __builtin_bit_cast(f16x2, u);
                        r[q][0] = (float)h[0]; r[q][1] = (float)h[1];
                    }
                    if (EPI == SF_EPI_RES_GELU_DW1) {
                        dww[q] = *reinterpret_cast<const f32x2*>(sdww + rb + 4 * khalf + e);
                        dwb[q] = *reinterpret_cast<const f32x2*>(sdwb + rb + 4 * khalf + e);
                    }
                }
                epi_n<EPI, kFast, 4>(v, r, dww, dwb, gam);
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = j0 + jj, rb = m * 64 + t * 32 + 8 * j;                  // (wave-uniform)
                    const float o[4] = {v[2 * jj][0], v[2 * jj][1], v[2 * jj + 1][0], v[2 * jj + 1][1]};
                    if (CF != 2) {                                                        // fp32 planes: 128-byte row segments
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o[e]), rc32,
                                                                  (k4 < rows_left(rb + e)) ? lane_c32 : kOob, (rb + e) * ldc4, 0);
                    }
                    if (CF >= 2) {                                                        // k-octets: 8 bytes = 4 rows of pixel n
                        f16x4 h;
#pragma unroll
                        for (int e = 0; e < 4; ++e) h[e] = (_Float16)o[e];
                        const int so = (rb >> 3) * ldc4 * 4;
                        // CF = 3: rows >= M of a last octet belong to someone else (flow rows of the motion features)
                        // CF = 2: every row of the last octet is written (finite: zero weight rows give gelu(bias = 0)) -- the consumer
                        // multiplies rows M .. 8 ceil(M / 8) - 1 by zero weights, and 0 x (whatever the buffer held) must not be NaN
                        const bool full = (CF == 2) ? (k4 < rows_left8(rb)) : (k4 < rows_left(rb + 3));
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h), rc16, full ? lane_k16 : kOob, so, 0);
                        if (CF == 3 && (g.M & 3) && rb + 8 > g.M && rb < g.M) {           // (wave-uniform) partial last group: row by row
#pragma unroll
                            for (int e = 0; e < 3; ++e) {
                                const _Float16 he = h[e];                // (bit_cast of a vector element lvalue reads element 0)
                                const bool part = k4 < rows_left(rb) && !(k4 < rows_left(rb + 3)) && k4 < rows_left(rb + e);
                                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, he), rc16,
                                                                      part ? lane_k16 + e * 2 : kOob, so, 0);
                            }
                        }
                    }
                }
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

#ifdef SF_BSTAT_STAGGER
    // experiment: the second workgroup of a CU (dispatch order: 32 CUs per XCD are filled once before any gets its second
    // workgroup) starts its weight loop half an m-step late, so that its epilogues fall into the other one's MFMA stages
    if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_sleep(SF_BSTAT_STAGGER);
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

// ---- software-pipelined form for the GELU -> k-octet layers (ffn*.0, pw, fc1: 60 % of the family's time) ----------------------
// The phase timers of the kernel above (tools/gemm_bs_timers.py, M960 K640, one product) put 31 % of a wave's life into its
// epilogues (3300 of 9400 cycles per 64 rows: GELU is ~18 VALU instructions per pair of values), during which the wave
// issues no MFMA; with only two waves per SIMD (K = 640 costs 160 registers) nothing else fills the matrix pipe.  Here the
// wave pipelines itself: it walks the rows in tiles of 32 with TWO accumulators, and the epilogue of tile i is cut into
// groups that are scheduled BETWEEN the MFMAs of tile i + 1 (sched_group_barrier: one MFMA, one fragment read, a few VALU).
// The weight stream is fragment-granular: one 1-KB DMA piece = the (32 rows x 16 k) A fragment of one MFMA, laid down in LDS
// in the order the MFMAs consume it, in stages of S fragments (S divides the fragments of a tile: no stage straddles a tile).
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int NF> struct StageF { static constexpr int value = (NF % 16 == 0) ? 16 : (NF % 20 == 0) ? 20 : (NF % 12 == 0) ? 12 : 8; };

template <int NKS, int PM>
__global__ __launch_bounds__(kThreads, 2) void gemm_bstat_gk_kernel(const BsArgs a) {
    const SfGemm& g = a.g;
    constexpr int NF = NKS * PM, S = StageF<NF>::value, NS = NF / S, P = S / 4;
    static_assert(NF % S == 0 && S % 4 == 0, "stage size must divide the fragments of a tile");
    constexpr int kStage = S * 1024;
    __shared__ __attribute__((aligned(1024))) char smem[RING * kStage + kParamRows * 4];
    float* sbias = reinterpret_cast<float*>(smem + RING * kStage);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, l31 = lane & 31;
    const WorkItem wi = work_item(a);
    const int tile = wi.tile % a.ntile, z = wi.tile / a.ntile;
    const int n = tile * BN + wave * 32 + l31, nc = min(n, g.N - 1);
    const int NT = (g.M + 31) / 32;                                                       // row tiles
    const int t_beg = (int)((int64_t)NT * wi.part / wi.nparts), t_end = (int)((int64_t)NT * (wi.part + 1) / wi.nparts);

#ifdef SF_BSTAT_TIMERS
    const long long ts0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
    long long tw = 0, tb = 0, ti = 0, tm = 0, te = 0, tprev = ts0;
#endif
    f16x8 b[NKS];
    load_b_frags<NKS>(g, z, nc, khalf, NKS, b);

    // ---- weight fragments by LDS-DMA: fragment f of tile tau = (k-step f / PM, plane f % PM: lo before hi) ----
    const __amdgpu_buffer_rsrc_t rah = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A_hi), 0, a.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ral = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A_lo), 0, a.a_bytes, 0x00020000);
    const int voa = ((lane >> 5) * (int)g.lda_h + (lane & 31)) * 16;                      // (k-half octet, row) of this lane
    auto issue_stage = [&](int tau, int s, int slot) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int fi = wave + 4 * i;                                                  // fragment of the stage: wave-uniform
            const int f = s * S + fi, ks = f / PM, pl = f % PM;
            const int so = (2 * ks * (int)g.lda_h + tau * 32) * 16;
            // (the k term travels in the checked vector offset: k-steps past the planes' extent read as zeros)
            __builtin_amdgcn_raw_ptr_buffer_load_lds((PM == 2 && pl == 0) ? ral : rah, (lds_ptr)(smem + (slot * S + fi) * 1024), 16,
                                                     voa + so, 0, 0, 0);
        }
    };
    int ta = t_beg, sa = 0;
    auto advance = [&]() {
        const bool wrap = (sa + 1 == NS);
        sa = wrap ? 0 : sa + 1;
        ta = (wrap && ta + 1 < t_end) ? ta + 1 : ta;
    };
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) {
        issue_stage(ta, sa, i);
        advance();
    }
    for (int i = tid; i < NT * 32; i += kThreads) sbias[i] = (i < g.M && g.bias) ? g.bias[i] : 0.f;

    _Float16* c16 = reinterpret_cast<_Float16*>(g.C) + (int64_t)z * g.strideC;
    const __amdgpu_buffer_rsrc_t rc16 = __builtin_amdgcn_make_buffer_rsrc(c16, 0, (int)((int64_t)((g.M + 7) / 8) * g.ldc * 16), 0x00020000);
    const int lane_k16 = (n < g.N) ? n * 16 + khalf * 8 : kOob;
    const int k4 = 4 * khalf, ldc16 = (int)g.ldc * 16;
    auto rows_left8 = [&](int r) { return __builtin_amdgcn_readfirstlane((g.M + 7) / 8 * 8 - r); };      // up to the end of the last octet

    // rows tau * 32 + 8 j + 4 khalf + 0..3 of `acc` for NG consecutive groups j0 ..: gelu(alpha * acc) -> fp16 -> 8-byte stores.
    // CH = pairs of values whose GELU chains run interleaved (K = 640: one group = two pairs at a time, for the registers)
    auto epi_groups = [&](const f32x16& acc, int tau, auto j0_tag, auto ng_tag) {
        constexpr int J0 = decltype(j0_tag)::value, NG = decltype(ng_tag)::value;
        constexpr int CH = (NKS >= 40) ? 2 : 2 * NG;
        if constexpr (NG > 0) {
            unsigned hw[2 * NG];                                                            // packed fp16 pairs
#pragma unroll
            for (int c = 0; c < 2 * NG; c += CH) {
                f32x2 v[CH];
#pragma unroll
                for (int q = 0; q < CH; ++q) {
                    const int j = J0 + ((c + q) >> 1), e = ((c + q) & 1) * 2;
                    v[q][0] = g.alpha * acc[4 * j + e];
                    v[q][1] = g.alpha * acc[4 * j + e + 1];
                }
                gelu_poly_n<CH>(v);
#pragma unroll
                for (int q = 0; q < CH; ++q) {
                    f16x2 h;
                    h[0] = (_Float16)v[q][0]; h[1] = (_Float16)v[q][1];
                    hw[c + q] = __builtin_bit_cast(unsigned, h);
                }
            }
#pragma unroll
            for (int jj = 0; jj < NG; ++jj) {
                const int rb = tau * 32 + 8 * (J0 + jj);
                u32x2 o;
                o[0] = hw[2 * jj]; o[1] = hw[2 * jj + 1];
                // (every row of the last octet is written: see the kernel above)
                __builtin_amdgcn_raw_buffer_store_b64(o, rc16, (k4 < rows_left8(rb)) ? lane_k16 : kOob, (rb >> 3) * ldc16, 0);
            }
        }
    };

    wait_vm<0>();
    __syncthreads();
#ifdef SF_BSTAT_TIMERS
    const long long ts1 = __builtin_readcyclecounter();
    tprev = ts1;
#endif

    int slot = 0;
    const char* sp_base = smem + lane * 16;
    f32x16 acc0, acc1 = {};
    // one tile: MFMAs into `cur`, the epilogue of the previous tile (`prev`, tile tau - 1) between them
    auto phase = [&](f32x16& cur, const f32x16& prev, int tau, auto prev_tag, auto pp_tag) {
        constexpr bool kPrev = decltype(prev_tag)::value;
        constexpr bool kPrevPrev = decltype(pp_tag)::value;        // the phase before this one had an epilogue (stores) too
        using std::integral_constant;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + tau * 32 + 8 * j + 4 * khalf);
#pragma unroll
            for (int e = 0; e < 4; ++e) cur[4 * j + e] = b4[e];
        }
#if SF_BSTAT_ALT_ACC
        // Two accumulation chains per tile (fragments alternate between `cur` and `alt`, summed at the end of the tile): an MFMA that
        // depends on the one before it AND has other instructions issued in between (here: a fragment read and a share of the previous
        // tile's epilogue behind every MFMA) waits for that MFMA's write-back -- ~40 cycles on top of the 32 it occupies the pipe
        // (MI355X_MICROARCH.md, "one EXTRA issue slot between two MFMAs on the SAME accumulator"); with two chains the neighbour is
        // independent.  With two products `cur` collects the lo products and `alt` the hi ones.
        f32x16 alt = {};
#endif
        static_for<0, NS>([&](auto s_tag) {
            constexpr int s = decltype(s_tag)::value;
            SF_BS_STAMP(tm)
            // this wave's pieces of the stage have landed.  Younger than them in the in-order queue: the refill issued at the end of the
            // previous stage (P pieces) AND the epilogue stores of the previous stage (one 8-byte store per group) -- counting those too
            // leaves them in flight for a whole stage instead of draining them here (SF_BSTAT_STORE_SLACK)
            {
                constexpr int kB = 4 / NS, kR = 4 % NS;
                constexpr int sp = (s > 0) ? s - 1 : NS - 1;
                constexpr bool had = (s > 0) ? kPrev : kPrevPrev;
                constexpr int kSt = (SF_BSTAT_STORE_SLACK && had) ? kB + (sp < kR ? 1 : 0) : 0;
                wait_vm<P * (RING - 2) + kSt>();
            }
            SF_BS_STAMP(tw)
            __builtin_amdgcn_s_barrier();                          // ... everyone's; nobody reads the previous slot any more
            SF_BS_STAMP(tb)
            __builtin_amdgcn_sched_barrier(0);
            SF_BS_STAMP(ti)
            const char* sp = sp_base + slot * kStage;
#pragma unroll
            for (int i = 0; i < S; ++i) {
                const f16x8 fr = *reinterpret_cast<const f16x8*>(sp + i * 1024);
#if SF_BSTAT_ALT_ACC
                if ((s * S + i) & 1) alt = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr, b[(s * S + i) / PM], alt, 0, 0, 0);
                else
#endif
                cur = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr, b[(s * S + i) / PM], cur, 0, 0, 0);
            }
            // groups of the previous tile's epilogue that belong to this stage: 4 groups spread over the NS stages
            constexpr int kBase = 4 / NS, kRem = 4 % NS;
            constexpr int kJ0 = s * kBase + (s < kRem ? s : kRem), kGroups = kBase + (s < kRem ? 1 : 0);
            if constexpr (kPrev) epi_groups(prev, tau - 1, integral_constant<int, kJ0>{}, integral_constant<int, kGroups>{});
            // the refill of the slot released by this stage's barrier.  In program order BEHIND the fragment reads (hipcc orders an
            // LDS-DMA against every LDS read of its region) and pinned beside the last MFMAs: issued in front of the MFMAs it
            // cost 350 cycles per stage (phase timers)
            issue_stage(ta, sa, slot == 0 ? RING - 1 : slot - 1);
            advance();
            // issue order, pinned: fragment reads run kAhead MFMAs ahead; behind every MFMA one fragment read and a share of the
            // epilogue's VALU work (~36 instructions per group)
            {
                constexpr int kAhead = 3;
                constexpr int kValu = kPrev ? (40 * kGroups + S - 1) / S : 0;
                // (the fragment reads depend on ONE VALU instruction, the slot's LDS address: it needs a VALU slot in front of
                // them, or the pipeline below has no valid order and is dropped altogether)
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, kAhead, 0);
#pragma unroll
                for (int i = 0; i < S; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i + kAhead < S) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (kValu > 0) __builtin_amdgcn_sched_group_barrier(0x002, kValu, 0);
                    if (i + kAhead >= S) {                         // (no fragment read left: the refill's pieces, offset adds first)
                        constexpr int kPer = (P + kAhead - 1) / kAhead;
                        __builtin_amdgcn_sched_group_barrier(0x002, kPer, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, kPer, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            slot = (slot == RING - 1) ? 0 : slot + 1;
        });
#if SF_BSTAT_ALT_ACC
#pragma unroll
        for (int e = 0; e < 16; ++e) cur[e] += alt[e];
#endif
    };

    using std::integral_constant;
    int tau = t_beg;
    if (tau < t_end) {
        using T_ = integral_constant<bool, true>;
        using F_ = integral_constant<bool, false>;
        phase(acc0, acc1, tau, F_{}, F_{});
        ++tau;
        if (tau + 1 < t_end) {                                     // (the first pair: the phase before it stored nothing)
            phase(acc1, acc0, tau, T_{}, F_{});
            phase(acc0, acc1, tau + 1, T_{}, T_{});
            tau += 2;
        }
        while (tau + 1 < t_end) {                                  // two tiles per trip: the accumulators swap roles
            phase(acc1, acc0, tau, T_{}, T_{});
            phase(acc0, acc1, tau + 1, T_{}, T_{});
            tau += 2;
        }
        if (tau < t_end) {
            phase(acc1, acc0, tau, T_{}, F_{});                    // (conservative: this may be the second phase)
            epi_groups(acc1, tau, integral_constant<int, 0>{}, integral_constant<int, 2>{});
            epi_groups(acc1, tau, integral_constant<int, 2>{}, integral_constant<int, 2>{});
        } else {
            epi_groups(acc0, tau - 1, integral_constant<int, 0>{}, integral_constant<int, 2>{});
            epi_groups(acc0, tau - 1, integral_constant<int, 2>{}, integral_constant<int, 2>{});
        }
    }
    SF_BS_STAMP(tm)
    wait_vm<0>();                                                  // (pieces requested past the end must land before the LDS is released)
#ifdef SF_BSTAT_TIMERS
    if (a.ts && lane == 0 && blockIdx.x < 4096) {
        long long* d = a.ts + ((int64_t)blockIdx.x * 4 + wave) * 8;
        d[0] = ts1 - ts0; d[1] = tw; d[2] = tb; d[3] = ti; d[4] = tm; d[5] = rt0;   /* (gk: start time, 100 MHz) */ d[6] = __builtin_readcyclecounter() - ts0;
        d[7] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
#endif
}

template <int PM>
int launch_gk(const BsArgs& a, dim3 grid, hipStream_t st) {
    const int nks = (a.g.K + 15) / 16;
    if (nks <= 8) hipLaunchKernelGGL((gemm_bstat_gk_kernel<8, PM>), grid, dim3(kThreads), 0, st, a);
    else if (nks <= 16) hipLaunchKernelGGL((gemm_bstat_gk_kernel<16, PM>), grid, dim3(kThreads), 0, st, a);
    else if (nks <= 24) hipLaunchKernelGGL((gemm_bstat_gk_kernel<24, PM>), grid, dim3(kThreads), 0, st, a);
    else if (nks <= 32) hipLaunchKernelGGL((gemm_bstat_gk_kernel<32, PM>), grid, dim3(kThreads), 0, st, a);   // (K = 486: 31 k-steps)
    else hipLaunchKernelGGL((gemm_bstat_gk_kernel<40, PM>), grid, dim3(kThreads), 0, st, a);
    return sf::check_launch("sf_gemm(B-stationary, pipelined GELU)");
}

template <int NKS, int PM>
int launch_res(const BsArgs& a, dim3 grid, hipStream_t st) {
    const SfGemm& g = a.g;
    const bool needs_r = g.epilogue == SF_EPI_RES || g.epilogue == SF_EPI_RES_GELU || g.epilogue == SF_EPI_RES_GELU_DW1 ||
                         g.epilogue == SF_EPI_AXPY;
    if (!needs_r) hipLaunchKernelGGL((gemm_bstat_kernel<NKS, PM, 0>), grid, dim3(kThreads), 0, st, a);
    else if (g.r_f16 == 2) {
        if constexpr (NKS <= 32) hipLaunchKernelGGL((gemm_bstat_kernel<NKS, PM, 2>), grid, dim3(kThreads), 0, st, a);
        else return sf::fail(SF_ERR_UNSUPPORTED, "sf_gemm(B-stationary): a k-octet residual needs K <= 512 (register budget)");
    } else {
        if constexpr (NKS <= 24) hipLaunchKernelGGL((gemm_bstat_kernel<NKS, PM, 1>), grid, dim3(kThreads), 0, st, a);
        else return sf::fail(SF_ERR_UNSUPPORTED, "sf_gemm(B-stationary): an fp32 residual needs K <= 384 (register budget)");
    }
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
    if (needs_r && g.K > (g.r_f16 == 2 ? 512 : 384)) return false;
    if (g.c_f16 >= 2 && ((reinterpret_cast<uintptr_t>(g.c_f16 == 2 ? (void*)g.C : g.C16) & 15) || ((g.c_f16 == 2 ? g.strideC : g.strideC16) & 7) || g.ldc < g.N))
        return false;
    if (g.b_layout == SF_LAYOUT_F16_KOCT && ((reinterpret_cast<uintptr_t>(g.B) & 15) || (g.strideB & 7) || (g.b_group_stride & 7) || g.ldb < g.N))
        return false;
    // argument rules the tiled family checks in check_output_formats(): a problem that breaks them must not become a GPU fault
    // here -- return false, the tiled path then rejects it with its error message
    if (g.r_f16 != 0 && g.r_f16 != 2) return false;
    if (needs_r && (!g.R || g.ldr < g.N)) return false;
    if (g.epilogue == SF_EPI_RES_GELU_DW1 && (!g.dw_w || !g.dw_b)) return false;
    if (g.epilogue == SF_EPI_AXPY && !g.gamma) return false;
    if (g.r_f16 == 2 && ((reinterpret_cast<uintptr_t>(g.R) & 15) || (g.strideR & 7))) return false;
    if (g.c_f16 != 2 && (!g.C || g.ldc < g.N)) return false;
    if (g.c_f16 == 3 && !g.C16) return false;
    // 32-bit buffer ranges and offsets: every span the kernel turns into a descriptor range or a scalar offset stays under kOob = 2^30
    // (an out-of-range lane is given the offset 2^30 and must fall outside EVERY range), computed in 64 bits
    {
        const int64_t lim = (int64_t)1 << 30;
        const int64_t noct = (g.K + 7) / 8, moct = (g.M + 7) / 8;
        int64_t bspan;
        if (g.b_layout == SF_LAYOUT_F16_KOCT)
            bspan = (g.b_group > 0 ? (int64_t)((g.K - 1) / g.b_group) * g.b_group_stride * 2 + (int64_t)((g.b_group + 7) / 8) * g.ldb * 16
                                   : noct * (int64_t)g.ldb * 16);
        else
            bspan = (g.b_group > 0 ? (int64_t)((g.K - 1) / g.b_group) * g.b_group_stride + (int64_t)(g.b_group - 1) * g.ldb + g.N
                                   : (int64_t)(g.K - 1) * g.ldb + g.N) * (g.b_layout == SF_LAYOUT_F16_K_MAJOR ? 2 : 4);
        if (bspan >= lim) return false;
        if (g.c_f16 != 2 && ((int64_t)(g.M - 1) * g.ldc + g.N) * 4 >= lim) return false;
        if (g.c_f16 >= 2 && moct * (int64_t)g.ldc * 16 >= lim) return false;
        if (needs_r) {
            const int64_t mr = g.M - 1;
            const int64_t rspan = (g.r_f16 == 2) ? moct * (int64_t)g.ldr * 16
                                : ((g.r_group > 0 ? (mr / g.r_group) * g.r_group_stride + (mr % g.r_group) * (int64_t)g.ldr : mr * (int64_t)g.ldr) + g.N) * 4;
            if (rspan >= lim) return false;
        }
    }
    return true;
}

int gemm_bstat_launch(const SfGemm& g, hipStream_t st) {
    BsArgs a{};
    a.g = g;
    const int sk = (g.precision == SF_PRECISION_F16) ? 128 : 64;        // k extent of a weight stage
    const int kp = (g.K + 127) / 128 * 128;                             // extent of the planes (a_k_pad = 128)
    a.a_bytes = (int)((int64_t)kp * g.lda_h * 2);
    a.nst = (g.K + sk - 1) / sk;
    a.msteps = (g.M + 63) / 64;
    a.ntile = ceil_div(g.N, BN);
    // Grid shaping.  A grid smaller than the resident workgroup slots (2 per CU; a single clip: 165 pixel tiles) is cut into f
    // row ranges per tile, whose activations are then read f times.  (Cutting only the tiles of a partial LAST round of a large
    // grid -- 1320 tiles = 2 rounds + 296 -- was measured too: every new round starts with all its workgroups loading their
    // activations at once, ~20 us of HBM time that the shorter row ranges cannot amortise: 292 -> 310 us at M960 K640.)
    const int64_t wgs = (int64_t)a.ntile * g.batch;
    const int units = (g.epilogue == SF_EPI_GELU && g.c_f16 == 2) ? (g.M + 31) / 32 : a.msteps;   // row units that can be split
    const int slots = 512;
    int n_main = (int)wgs, f = 1;
    if (wgs < slots) {
        double best = 1.0;
        for (int c = 2; c <= 4 && c <= units; ++c) {
            const double t = (double)((wgs * c + slots - 1) / slots) / c + 0.04 * (c - 1);      // (+ the repeated activation loads)
            if (t < best - 1e-9) { best = t; f = c; }
        }
        if (f > 1) n_main = 0;
    }
    a.n_main = n_main;
    a.msplit = f;
    // vector memory operations one epilogue of gemm_bstat_kernel leaves BEHIND the next m-step's first weight pieces (per wave):
    // 16 TM fp32 row stores and / or 4 TM k-octet stores, + the residual loads of the next m-step (16 TM fp32 / 4 TM k-octets).
    // The row-by-row tail stores of an `M % 4 != 0` fp32 + k-octet output are not counted: wait for everything there.
    {
        const bool needs_r = g.epilogue == SF_EPI_RES || g.epilogue == SF_EPI_RES_GELU || g.epilogue == SF_EPI_RES_GELU_DW1 ||
                             g.epilogue == SF_EPI_AXPY;
        a.e_ops = (g.c_f16 == 2 ? 0 : 16 * TM) + (g.c_f16 >= 2 ? 4 * TM : 0) + (!needs_r ? 0 : (g.r_f16 == 2 ? 4 * TM : 16 * TM));
        if (g.c_f16 == 3 && (g.M & 3)) a.e_ops = 0;
    }
    const int64_t grid_x = n_main + (wgs - n_main) * f;
#ifdef SF_BSTAT_TIMERS
    a.ts = getenv("SF_GEMM_TS_BUF") ? (long long*)strtoull(getenv("SF_GEMM_TS_BUF"), nullptr, 0) : nullptr;
#endif
    dim3 grid((unsigned)grid_x);
#ifndef SF_BSTAT_PIPELINED
#define SF_BSTAT_PIPELINED 1
#endif
    // GELU -> k-octets (the FFN hiddens, x4, the temporal MLP hidden): the software-pipelined kernel
    if (SF_BSTAT_PIPELINED && g.epilogue == SF_EPI_GELU && g.c_f16 == 2) {
        return (g.precision == SF_PRECISION_F16) ? launch_gk<1>(a, grid, st) : launch_gk<2>(a, grid, st);
    }
    return (g.precision == SF_PRECISION_F16) ? launch_nks<1>(a, grid, st) : launch_nks<2>(a, grid, st);
}

}  // namespace sf
