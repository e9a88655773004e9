// The back half of an SK block in ONE launch:  y = ffn2.2( gelu( ffn2.0( gelu( (pw + I) x3 ) ) ) )
//
// Reference: core/update.py:35-36 (PCBlock4_Deep_nopool_res.forward: `x = F.gelu(x + self.pw(x)); x = self.ffn2(x)`, ffn2 =
// nn.Sequential(conv1x1, GELU, conv1x1), update.py:14-16).  VERDICT r5 next #1(a).  Until round 5 these were three launches per block
// (pw -> x4 as k-octets, ffn2.0 -> the 1.5 C hidden as k-octets, ffn2.2), each with its own load burst, launch ramp and tail, the
// small-K ones (C = 128 ... 384) at 40-45 % matrix-pipe occupancy and 2.4 TB/s of traffic for tensors that only the next launch reads.
//
// A per-pixel chain: a wave owns 32 pixels (32 x 32 x 16 tiles, csrc/gemm_bstat.hip's activation-stationary layout) and keeps
//   * x3 of its pixels as B fragments (K = C: C / 4 registers), loaded once from the depthwise kernel's fp16 rows;
//   * x4 = gelu((pw + I) x3) as B fragments again: in the C/D layout a lane holds rows (r & 3) + 8 (r >> 2) + 4 khalf of a 32-row
//     tile for ITS pixel; registers 8 s .. 8 s + 7 (after bias, GELU, fp16 rounding) are the lane's eight k-values of k-step s of
//     the next layer, provided that layer's weight columns are packed in that order (ops.PackedTail) -- no LDS round trip;
//   * the 1.5 C hidden of ffn2 32 rows at a time (one accumulator tile), turned into two B fragments the same way and consumed at
//     once by ffn2.2's accumulators (M2 / 32 tiles, resident for the whole kernel).
// All three layers' weights are ONE host-packed stream of 1-KB MFMA fragments in consumption order (units padded to 16-fragment
// stages), pulled L2 -> LDS through the 3-stage ring of csrc/ffn_pair.hip, one barrier per stage, shared by the 4 waves (128 pixels).
// The rounding points are those of the three launches (x4 and the hidden leave as fp16; accumulation fp32; weights hi + lo or hi).
#include "sf_common.h"
#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;
using sf::f32x2;

constexpr int kThreads = 256, kWaves = 4;
constexpr int BN = 128;                      // pixels per workgroup (4 waves x 32)
#ifndef SF_TAIL_V2
#define SF_TAIL_V2 0      // 1: 32-fragment stages in a ring of TWO (half the barriers), pw as one unit, G hidden tiles per unit (A/B)
#endif
constexpr int S = SF_TAIL_V2 ? 32 : 16;      // fragments per stage
constexpr int kStage = S * 1024;
constexpr int RING = SF_TAIL_V2 ? 2 : 3;
constexpr int PCS = S / kWaves;              // DMA pieces per wave and stage
constexpr int kOob = 1 << 30;
constexpr int kMaxC = 384, kMaxH = 576, kMaxM = 192;

struct TailArgs {
    SfSkTail g;
    int ntile;            // pixel tiles per image
    int nh;               // hidden tiles of 32 rows
    int64_t w_bytes;      // bytes of the packed weight stream
    int x_span;           // bytes of one image of X the kernel may address
};

template <int N>
__device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt((N & 15) | 0x0F70 | ((N >> 4) << 14)); }

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// polynomial GELU of NP pairs with the Horner chains interleaved (csrc/gemm_bstat.hip: same operations per value as sf::gelu_poly2)
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

// gelu(alpha * acc) of a 32-row tile -> the two B fragments (k-steps 0, 1 of the tile) of the next layer.  NP pairs of values per
// GELU pass (4: four interleaved Horner chains; 2: half the temporaries -- the C >= 256 shapes have no registers to spare)
template <int NP>
__device__ __forceinline__ void tile_to_frags(const f32x16& acc, float alpha, f16x8& f0, f16x8& f1) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        f16x8 h;
#pragma unroll
        for (int c = 0; c < 4; c += NP) {
            f32x2 v[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                v[q][0] = alpha * acc[8 * s + 2 * (c + q)];
                v[q][1] = alpha * acc[8 * s + 2 * (c + q) + 1];
            }
            gelu_poly_n<NP>(v);
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                h[2 * (c + q)] = (_Float16)v[q][0];
                h[2 * (c + q) + 1] = (_Float16)v[q][1];
            }
        }
        if (s == 0) f0 = h; else f1 = h;
    }
}

// hidden tiles per phase-2 unit (v2; 1 in v1): G * (2 NC + 2 NM) PM fragments should fill whole 32-fragment stages, G must divide NH
// (12 at H = 384, 18 at 576, 6 at 192: host-checked)
constexpr int tail_group(int NC, int NM, int PM) {
    if (!SF_TAIL_V2) return 1;
    if (NC == 8 && NM == 6) return PM == 2 ? 4 : 4;             // TF = 56 / 28: 224 / 112 -> 7 / 3.5 stages
    if (NC == 8 && NM == 4) return PM == 2 ? 2 : 4;             // TF = 48 / 24: 96 / 96
    return 2;                                                   // (12, 1): 104 / 52; (4, 2): 48 / 24
}

// NC = C / 32 (tiles of x4 = k-step pairs of pw and ffn2.0), NM = ceil(M2 / 32), PM = MFMA products per weight (2: lo + hi, 1: hi)
template <int NC, int NM, int PM>
__global__ __launch_bounds__(kThreads, NC >= 12 ? 1 : 2) void sk_tail_kernel(const TailArgs a) {
    constexpr int GNP = (NC >= 8) ? 2 : 4;                       // pairs per GELU pass
    const SfSkTail& g = a.g;
    constexpr int KS = 2 * NC;                                   // k-steps of 16 over the C channels
    constexpr int NA1 = KS * PM;                                 // fragments of one pw tile
    constexpr int NA2 = KS * PM, NB2 = 2 * NM * PM;              // one hidden tile: ffn2.0's fragments, then ffn2.2's two k-steps
    constexpr int TF = NA2 + NB2;
    // v1: every pw tile and every hidden tile is a unit of its own, padded to whole stages.  v2: pw is ONE unit (NC tiles), the hidden
    // tiles come G to a unit (tail_group(): chosen so that G * TF fills whole 32-fragment stages where NH allows)
    constexpr int G = tail_group(NC, NM, PM);
    constexpr int U1 = SF_TAIL_V2 ? (NC * NA1 + S - 1) / S : (NA1 + S - 1) / S;
    constexpr int U2 = (G * TF + S - 1) / S;
    __shared__ __attribute__((aligned(1024))) char smem[RING * kStage + (kMaxC + kMaxH + kMaxM) * 4];
    float* sb1 = reinterpret_cast<float*>(smem + RING * kStage);
    float* sb2 = sb1 + kMaxC;
    float* sb3 = sb2 + kMaxH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, l31 = lane & 31;
    const int tile = blockIdx.x % a.ntile, z = blockIdx.x / a.ntile;
    const int n = tile * BN + wave * 32 + l31, nc = min(n, g.N - 1);

    // ---- the weight stream: stage s = bytes [16 KB s, 16 KB (s + 1)); wave w moves pieces w, w + 4, ...; requests past the end
    // re-read the last stage (the stage offset travels in the scalar offset, which the raw-buffer range check does not cover) ----
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.wstream), 0, (int)a.w_bytes, 0x00020000);
    const int last_stage = (int)(a.w_bytes / kStage) - 1;
    auto issue_stage = [&](int s, int slot) {
        const int sc = min(s, last_stage);
#pragma unroll
        for (int i = 0; i < PCS; ++i) {
            const int piece = wave + kWaves * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(smem + slot * kStage + piece * 1024), 16, lane * 16,
                                                     sc * kStage + piece * 1024, 0, 0);
        }
    };
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) issue_stage(i, i);

    for (int i = tid; i < NC * 32; i += kThreads) sb1[i] = (i < g.C && g.bias1) ? g.bias1[i] : 0.f;
    for (int i = tid; i < a.nh * 32; i += kThreads) sb2[i] = (i < g.H && g.bias2) ? g.bias2[i] : 0.f;
    for (int i = tid; i < NM * 32; i += kThreads) sb3[i] = (i < g.M2 && g.bias3) ? g.bias3[i] : 0.f;

    // ---- x3: B fragments of this lane's pixel (k = 16 ks + 8 khalf + i) from fp16 rows [C][ldx]; rows >= C read as zero ----
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(g.X)) + (int64_t)z * g.strideX * 2, 0, a.x_span, 0x00020000);
    f16x8 x[KS];
    {
        const int vo = (khalf * 8 * (int)g.ldx + nc) * 2, rstep = (int)g.ldx * 2;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            f16x8 f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned short u = __builtin_amdgcn_raw_buffer_load_b16(rx, (ks * 16 + i + 8 * khalf < g.C) ? vo : kOob, (ks * 16 + i) * rstep, 0);
                f[i] = __builtin_bit_cast(_Float16, u);
            }
            x[ks] = f;
            if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // (four k-steps of loads in flight at a time: csrc/gemm_bstat.hip)
        }
    }
    wait_vm<0>();
    __syncthreads();

    int gs = 0, slot = 0;                                         // global stage index, its ring slot
    // U stages of the stream: body(fragment index in the unit, LDS address of this lane's 16 bytes of the fragment).  NF = fragments of
    // the unit that feed an MFMA (the rest is padding); a GELU block runs in front of fragments GP0, GP0 + GPS, ... (GPS = 0: none) and a
    // tile's accumulator start values (four LDS reads) are fetched in front of fragments TS, 2 TS, ... (TS = 0: none)
    auto run_unit = [&](auto u_tag, auto nf_tag, auto gp0_tag, auto gps_tag, auto ts_tag, auto&& body) {
        constexpr int U = decltype(u_tag)::value, NF = decltype(nf_tag)::value, GP0 = decltype(gp0_tag)::value,
                      GPS = decltype(gps_tag)::value, TS = decltype(ts_tag)::value;
        static_for<0, U>([&](auto st_tag) {
            constexpr int st = decltype(st_tag)::value;
            issue_stage(gs + RING - 1, (RING == 2) ? (slot ^ 1) : (slot == 0 ? RING - 1 : slot - 1));
            const char* sp = smem + slot * kStage + lane * 16;
            static_for<0, S>([&](auto i_tag) {
                constexpr int i = decltype(i_tag)::value;
                body(std::integral_constant<int, st * S + i>{}, sp + i * 1024);
            });
            // issue order, pinned: fragment reads run kAhead MFMAs ahead of their use -- left alone hipcc requests a whole stage's
            // fragments up front (64 registers: spills at C = 256 / 384) or waits out the LDS latency in front of every MFMA
            {
                constexpr int nm = (NF - st * S) < 0 ? 0 : ((NF - st * S) > S ? S : (NF - st * S));
                constexpr int kAhead = 3;
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                     // (the slot's LDS address)
                __builtin_amdgcn_sched_group_barrier(0x100, nm < kAhead ? nm : kAhead, 0);
#pragma unroll
                for (int i = 0; i < nm; ++i) {
                    const int f = st * S + i;
                    if (GPS > 0 && f >= GP0 && (f - GP0) % GPS == 0) __builtin_amdgcn_sched_group_barrier(0x002, 200, 0);   // a GELU block
                    if (TS > 0 && f > 0 && f % TS == 0) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);               // the next tile's bias
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i + kAhead < nm) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            // every fragment read of the stage has EXECUTED before the barrier (the refill race of csrc/ffn_pair.hip)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wait_vm<PCS * (RING - 2)>();                           // this wave's pieces of the NEXT stage have landed ...
            __builtin_amdgcn_s_barrier();                          // ... everyone's; nobody reads this stage's slot any more
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            ++gs;
            slot = (slot == RING - 1) ? 0 : slot + 1;
        });
    };
    auto bias_tile = [&](const float* sb, int t32) {              // accumulator start values of rows 32 t32 ..: the pre-scaled bias
        f32x16 acc;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sb + t32 * 32 + 8 * j + 4 * khalf);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * j + e] = b4[e];
        }
        return acc;
    };

    // ---- phase 1: x4 = gelu((pw + I) x3), tile by tile, as the B fragments of ffn2.0 ----
    using std::integral_constant;
    f16x8 x4[KS];
#if SF_TAIL_V2
    {
        f32x16 acc = bias_tile(sb1, 0);
        run_unit(integral_constant<int, U1>{}, integral_constant<int, NC * NA1>{}, integral_constant<int, NA1>{}, integral_constant<int, NA1>{},
                 integral_constant<int, NA1>{}, [&](auto f_tag, const char* p) {
            constexpr int f = decltype(f_tag)::value;
            if constexpr (f < NC * NA1) {
                if constexpr (f > 0 && f % NA1 == 0) {              // the previous tile is complete: its fragments, the next tile's start
                    tile_to_frags<GNP>(acc, g.alpha1, x4[2 * (f / NA1 - 1)], x4[2 * (f / NA1 - 1) + 1]);
                    acc = bias_tile(sb1, f / NA1);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(p), x[(f % NA1) / PM], acc, 0, 0, 0);
            }
        });
        tile_to_frags<GNP>(acc, g.alpha1, x4[2 * (NC - 1)], x4[2 * (NC - 1) + 1]);
    }
#else
    static_for<0, NC>([&](auto t_tag) {
        constexpr int t = decltype(t_tag)::value;
        f32x16 acc = bias_tile(sb1, t);
        run_unit(integral_constant<int, U1>{}, integral_constant<int, NA1>{}, integral_constant<int, 0>{}, integral_constant<int, 0>{},
                 integral_constant<int, 0>{}, [&](auto f_tag, const char* p) {
            constexpr int f = decltype(f_tag)::value;
            if constexpr (f < NA1)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(p), x[f / PM], acc, 0, 0, 0);
        });
        tile_to_frags<GNP>(acc, g.alpha1, x4[2 * t], x4[2 * t + 1]);
    });
#endif

    // ---- phase 2: per 32 hidden rows: ffn2.0 -> gelu -> two k-steps of ffn2.2 (G hidden tiles per unit) ----
    f32x16 acc2[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) acc2[m] = bias_tile(sb3, m);
    for (int th = 0; th < a.nh; th += G) {
        f32x16 acch = bias_tile(sb2, th);
        f16x8 hf[2] = {};
        run_unit(integral_constant<int, U2>{}, integral_constant<int, G * TF>{}, integral_constant<int, NA2>{}, integral_constant<int, TF>{},
                 integral_constant<int, (G > 1) ? TF : 0>{}, [&](auto f_tag, const char* p) {
            constexpr int f = decltype(f_tag)::value;
            if constexpr (f < G * TF) {
                constexpr int j = f / TF, r = f % TF;
                if constexpr (r == 0 && j > 0) acch = bias_tile(sb2, th + j);
                if constexpr (r == NA2) tile_to_frags<GNP>(acch, g.alpha2, hf[0], hf[1]);
                if constexpr (r < NA2) {
                    acch = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(p), x4[r / PM], acch, 0, 0, 0);
                } else {
                    constexpr int q = (r - NA2) / PM, s_ = q / NM, m = q % NM;
                    acc2[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(p), hf[s_], acc2[m], 0, 0, 0);
                }
            }
        });
    }
    wait_vm<0>();                                                  // (pieces requested past the end must land before the LDS is released)

    // ---- y = alpha3 * acc2 (+ GELU): fp32 planes and / or fp16 k-octets (csrc/gemm_bstat.hip's stores) ----
    const __amdgpu_buffer_rsrc_t rc32 = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(g.Y) + (int64_t)z * g.strideY * 4, 0, g.Y ? (int)(((int64_t)(g.M2 - 1) * g.ldy + g.N) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rc16 = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(g.Y16) + (int64_t)z * g.strideY16 * 2, 0, g.Y16 ? (int)((int64_t)((g.M2 + 7) / 8) * g.ldy16 * 16) : 0, 0x00020000);
    const int lane_c32 = (n < g.N) ? (4 * khalf * (int)g.ldy + n) * 4 : kOob;
    const int lane_k16 = (n < g.N) ? n * 16 + khalf * 8 : kOob;
    const int k4 = 4 * khalf;
    auto rows_left = [&](int r) { return __builtin_amdgcn_readfirstlane(g.M2 - r); };
    auto rows_left8 = [&](int r) { return __builtin_amdgcn_readfirstlane((g.M2 + 7) / 8 * 8 - r); };
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int j0 = 0; j0 < 4; j0 += 2) {
            f32x2 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = j0 + (q >> 1), e = (q & 1) * 2;
                v[q][0] = g.alpha3 * acc2[m][4 * j + e];
                v[q][1] = g.alpha3 * acc2[m][4 * j + e + 1];
            }
            if (g.gelu_out) {                                         // (kernel-uniform)
                if (g.Y) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = sf::gelu_erf2(v[q]);
                } else {
                    gelu_poly_n<4>(v);
                }
            }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = j0 + jj, rb = m * 32 + 8 * j;
                const float o[4] = {v[2 * jj][0], v[2 * jj][1], v[2 * jj + 1][0], v[2 * jj + 1][1]};
                if (g.Y) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o[e]), rc32,
                                                              (k4 < rows_left(rb + e)) ? lane_c32 : kOob, (rb + e) * (int)g.ldy * 4, 0);
                }
                if (g.Y16) {
                    f16x4 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = (_Float16)o[e];
                    const int so = (rb >> 3) * (int)g.ldy16 * 16;
                    // y16_partial: rows >= M2 of the last octet belong to someone else (the flow rows of the motion features); else every
                    // row of the last octet is written (finite: zero weight rows -> the bias padding 0)
                    const bool full = g.y16_partial ? (k4 < rows_left(rb + 3)) : (k4 < rows_left8(rb));
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h), rc16, full ? lane_k16 : kOob, so, 0);
                    if (g.y16_partial && (g.M2 & 3) && rb + 8 > g.M2 && rb < g.M2) {   // (wave-uniform) partial last group: row by row
#pragma unroll
                        for (int e = 0; e < 3; ++e) {
                            const _Float16 he = h[e];
                            const bool part = k4 < rows_left(rb) && !(k4 < rows_left(rb + 3)) && k4 < rows_left(rb + e);
                            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, he), rc16,
                                                                  part ? lane_k16 + e * 2 : kOob, so, 0);
                        }
                    }
                }
            }
        }
}

struct Shape { int C, M2, NC, NM; };
constexpr Shape kShapes[] = {{256, 192, 8, 6}, {256, 126, 8, 4}, {384, 6, 12, 1}, {128, 64, 4, 2}};

inline const Shape* find_shape(int C, int M2) {
    for (const Shape& s : kShapes)
        if (s.C == C && s.M2 == M2) return &s;
    return nullptr;
}

inline int unit_frags(int n) { return (n + S - 1) / S * S; }

}  // namespace

// layout of the packed weight stream (the host packs exactly this: ops.PackedTail): total 1-KB fragments (0: shape / product count not
// built); *stage = fragments per stage (units are padded to multiples of it), *group = hidden tiles per phase-2 unit, *pw_one_unit = 1
// when the NC pw tiles form ONE unit (padded once) instead of a unit each
extern "C" int sf_sk_tail_layout(int C, int H, int M2, int pm, int* stage, int* group, int* pw_one_unit) {
    const Shape* s = find_shape(C, M2);
    if (!s || (pm != 1 && pm != 2) || H <= 0 || H % 32 || H > kMaxH) return 0;
    if (s->NC == 8 && s->NM == 6 && pm == 1) return 0;           // (256 -> H -> 192 with single-product weights spills 8 registers: not built)
    const int ks = 2 * s->NC, G = tail_group(s->NC, s->NM, pm), nh = H / 32;
    if (nh % G) return 0;
    if (stage) *stage = S;
    if (group) *group = G;
    if (pw_one_unit) *pw_one_unit = SF_TAIL_V2 ? 1 : 0;
    const int f1 = SF_TAIL_V2 ? unit_frags(s->NC * ks * pm) : s->NC * unit_frags(ks * pm);
    return f1 + (nh / G) * unit_frags(G * (ks + 2 * s->NM) * pm);
}

extern "C" int sf_sk_tail_frags(int C, int H, int M2, int pm) { return sf_sk_tail_layout(C, H, M2, pm, nullptr, nullptr, nullptr); }

extern "C" int sf_sk_tail(const SfSkTail* p, void* stream) {
    SF_REQUIRE(p && p->X && p->wstream && (p->Y || p->Y16), "sf_sk_tail: null pointer");
    const SfSkTail& g = *p;
    SF_REQUIRE(g.N > 0 && g.batch > 0 && g.ldx >= g.N, "sf_sk_tail: bad dims");
    SF_REQUIRE(g.pm == 1 || g.pm == 2, "sf_sk_tail: pm must be 1 (hi) or 2 (lo + hi)");
    const Shape* s = find_shape(g.C, g.M2);
    const int frags = sf_sk_tail_frags(g.C, g.H, g.M2, g.pm);
    SF_REQUIRE(s && frags > 0, "sf_sk_tail: shape C=%d H=%d M2=%d not built (C -> H -> M2 with (C, M2) in (256,192) (256,126) (384,6) (128,64), "
                               "H a multiple of 32 up to %d)", g.C, g.H, g.M2, kMaxH);
    TailArgs a;
    a.g = g;
    a.ntile = sf::ceil_div(g.N, BN);
    a.nh = g.H / 32;
    a.w_bytes = (int64_t)frags * 1024;
    const int64_t lim = (int64_t)1 << 30;
    SF_REQUIRE(g.wstream_bytes >= a.w_bytes && a.w_bytes < lim && (reinterpret_cast<uintptr_t>(g.wstream) & 15) == 0,
               "sf_sk_tail: weight stream too small (need %lld bytes), > 1 GiB or misaligned", (long long)a.w_bytes);
    const int64_t xspan = ((int64_t)(g.C - 1) * g.ldx + g.N) * 2;
    SF_REQUIRE(xspan < lim, "sf_sk_tail: image of X larger than 1 GiB");
    a.x_span = (int)xspan;
    SF_REQUIRE(!g.Y || (g.ldy >= g.N && ((int64_t)(g.M2 - 1) * g.ldy + g.N) * 4 < lim), "sf_sk_tail: bad fp32 output planes");
    SF_REQUIRE(!g.Y16 || (g.ldy16 >= g.N && (reinterpret_cast<uintptr_t>(g.Y16) & 15) == 0 && (g.strideY16 & 7) == 0 &&
                          (int64_t)((g.M2 + 7) / 8) * g.ldy16 * 16 < lim),
               "sf_sk_tail: k-octet output must be 16-byte aligned (image stride %% 8 halves), < 1 GiB per image");
    SF_REQUIRE((int64_t)a.ntile * g.batch < ((int64_t)1 << 31), "sf_sk_tail: grid too large");
    const dim3 grid((unsigned)((int64_t)a.ntile * g.batch)), block(kThreads);
    hipStream_t st = (hipStream_t)stream;
#define SF_TAIL(NC_, NM_, PM_) hipLaunchKernelGGL((sk_tail_kernel<NC_, NM_, PM_>), grid, block, 0, st, a)
    if (s->NC == 8 && s->NM == 6) SF_TAIL(8, 6, 2);              // (pm = 1 is not built for this shape: see sf_sk_tail_frags)
    else if (s->NC == 8 && s->NM == 4) { if (g.pm == 2) SF_TAIL(8, 4, 2); else SF_TAIL(8, 4, 1); }
    else if (s->NC == 12) { if (g.pm == 2) SF_TAIL(12, 1, 2); else SF_TAIL(12, 1, 1); }
    else { if (g.pm == 2) SF_TAIL(4, 2, 2); else SF_TAIL(4, 2, 1); }
#undef SF_TAIL
    return sf::check_launch("sf_sk_tail");
}
