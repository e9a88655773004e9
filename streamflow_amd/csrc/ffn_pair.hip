// FFN pair of an SK block in ONE launch:   y = W2 gelu(W1 x + b1) + b2   (+ the block's epilogue)
//
// Reference: core/update.py:14-16,30-36 -- PCBlock4_Deep_nopool_res.ffn1 / .ffn2 are nn.Sequential(conv1x1, GELU, conv1x1): ONE
// module call each.  The two-launch form (sf_gemm with a GELU epilogue -> sf_gemm) hands the 1.5 C hidden tensor from launch to
// launch through memory: written once, read once, 3 C bytes per pixel of the 4 - 5 C bytes a pair moves at all, and the second
// launch of an ffn1 pair (residual + GELU + depthwise 1x1 + GELU, fp16 rows out) ran at 13 - 25 % of the matrix-core peak, bound by
// that traffic.  Here the hidden never leaves the registers.
//
// Activation-stationary like csrc/gemm_bstat.hip, but on 16 x 16 x 32 tiles so that BOTH layers' state fits a wave:
//   * a wave owns 16 pixels.  Their K1 input channels are the B operand of layer 1, held in registers for the whole kernel
//     (a k-octet IS the register image: lane (pixel, kq) holds channels 32 s + 8 kq .. + 7 of k-step s: one 16-byte load);
//   * the hidden is produced 32 rows at a time (two 16-row tiles): C/D layout of v_mfma_f32_16x16x32_f16 = (column = pixel,
//     rows 4 kq .. 4 kq + 3 of the tile), so after bias + GELU + fp16 rounding the lane's 2 x 4 values ARE its share of a layer-2
//     B fragment (32 k) -- with the hidden channels of that k-step in the order  k = 8 kq + i  <->  row 4 kq + i (i < 4),
//     16 + 4 kq + i - 4 (i >= 4); the layer-2 weights are packed with their columns in exactly that order (ops.PackedPair);
//   * layer 2 accumulates ALL M2 output rows of the 16 pixels (M2 / 16 tiles x 4 registers) while the hidden streams past;
//   * both layers' weights stream through LDS in ONE linear sequence of 1-KB fragments, packed on the host in the order the
//     MFMAs consume them (per 32 hidden rows: 2 NK1 PM1 fragments of W1, then NT2 PM2 of W2', padded to whole 16-KB stages):
//     the DMA addressing is "next 16 KB", two stages ping-pong, one barrier per stage.
// PM1 / PM2 = MFMA products per layer (1: the round-to-nearest fp16 weight, 2: hi + lo); activations enter as fp16, fp32
// accumulation -- the config-2 arithmetic class (SF_PRECISION_F16X2 / SF_PRECISION_F16 per layer).
// MODE 0 (an ffn2 pair, update.py:36):  out = y  or gelu(y), as fp16 k-octet planes and / or fp32 planes.
// MODE 1 (an ffn1 pair, update.py:31-33 first two lines): x1 = gelu(x + y); x2 = gelu(x1 + dw1x1(x1)); out = x2 as fp16 ROWS (the
//         depthwise K x K kernel's input); the residual x is re-read from the k-octet input (the operand itself).
#include "sf_common.h"

#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;
using sf::f32x2;

constexpr int kPxWave = 16;                                // pixels per wave; a workgroup = NW waves = 16 NW pixels (NW = 4 or 8)
constexpr int S = 16;                                      // fragments per stage
constexpr int kStage = S * 1024;
#ifndef SF_PAIR_RING
#define SF_PAIR_RING 3                                     // stages in flight + 1: the ring of the weight stream (A/B knob)
#endif
constexpr int RING = SF_PAIR_RING;
#ifndef SF_PAIR_ONE_LOADER
#define SF_PAIR_ONE_LOADER 0
#endif

constexpr int kOob = 1 << 30;                              // byte offset beyond every buffer range (host-checked spans < 2^30)
constexpr int kMaxH = 608, kMaxM2 = 384;                   // hidden rows whose bias is kept in LDS; output rows (24 tiles)

#ifdef SF_PAIR_TIMERS
#define SF_PT_STAMP(acc_) { const long long t_ = __builtin_readcyclecounter(); acc_ += t_ - tprev; tprev = t_; }
#else
#define SF_PT_STAMP(acc_)
#endif

struct PairArgs {
#ifdef SF_PAIR_TIMERS
    long long* ts;        // SF_PAIR_TS_BUF: per-wave phase cycle sums (tools/ffn_pair_timers.py; -DSF_PAIR_TIMERS builds only)
#endif
    SfFfnPair p;
    int ntile;            // pixel tiles per image
    int hp;               // hidden row pairs of 16 = ceil(H / 32)
    int x_span;           // bytes of one image (all its groups) of X as the kernel addresses it
    int r32_span;         // ... of the fp32 residual planes (mode 1 with R32)
    int64_t w_bytes;      // bytes of the packed weight stream
};

template <int N>
__device__ __forceinline__ void wait_vm() {
    __builtin_amdgcn_s_waitcnt((N & 15) | 0x0F70 | ((N >> 4) << 14));
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int NK1, int NT2, int PM1, int PM2, int MODE, int NW>
__global__ __launch_bounds__(NW * 64, (NW == 8) ? 4 : ((NK1 > 8 && MODE >= 1) ? 2 : 3)) void ffn_pair_kernel(const PairArgs a) {
    const SfFfnPair& g = a.p;
    constexpr int kWaves = NW, kThreads = NW * 64, kPxWg = NW * kPxWave;
    constexpr int NA = 2 * NK1 * PM1, NB = NT2 * PM2, F = NA + NB, NSTG = (F + S - 1) / S;
    constexpr int PCS = S / kWaves;                               // 1-KB pieces a wave moves per stage
    // LDS: the weight ring + the layer-1 bias (read at the top of every 32-row step); the other parameters are read from global
    // memory outside the streaming loop (where a plain load cannot disturb the counted waits of the DMA queue)
    __shared__ __attribute__((aligned(1024))) char smem[RING * kStage + (kMaxH + kMaxM2) * 4];
    float* sb1 = reinterpret_cast<float*>(smem + RING * kStage);
    float* sb2 = sb1 + kMaxH;
    float* sdw = reinterpret_cast<float*>(smem);                   // (mode 1 epilogue: in the ring, after the stream has drained)
    float* sdb = sdw + kMaxM2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, l15 = lane & 15;
    const int tile = blockIdx.x % a.ntile, z = blockIdx.x / a.ntile;
    const int px = tile * kPxWg + wave * kPxWave + l15;
    const bool pin = px < g.N;

    // ---- the weight stream: stage s = bytes [s * 16 KB, (s + 1) * 16 KB) of the packed buffer; wave w moves pieces w, w + 4, ... ----
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.wstream), 0, (int)a.w_bytes, 0x00020000);
    // (stages requested past the end of the stream -- the loop keeps the request count per trip constant -- re-read the LAST stage
    // into a slot nobody reads any more: the stage offset travels in the scalar offset, which the raw-buffer range check of gfx9
    // does not cover, so "out of range: zeros" must not be relied on: ADVICE r5)
    const int last_stage = (int)(a.w_bytes / kStage) - 1;
    auto issue_stage = [&](int s, int slot) {
        const int sc = min(s, last_stage);
#pragma unroll
        for (int i = 0; i < S / kWaves; ++i) {
            const int piece = wave + kWaves * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(smem + slot * kStage + piece * 1024), 16, lane * 16,
                                                     sc * kStage + piece * 1024, 0, 0);
        }
    };
#ifdef SF_PAIR_TIMERS
    const long long ts0 = __builtin_readcyclecounter();
    long long t_issue = 0, t_mma = 0, t_drain = 0, t_bar = 0, tprev = ts0;
#endif
    const int nstage = a.hp * NSTG;
#pragma unroll
    for (int i = 0; i < RING - 1; ++i)
        if (i < nstage) issue_stage(i, i);

    // ---- epilogue parameters ----
    for (int i = tid; i < a.hp * 32; i += kThreads) sb1[i] = (i < g.H && g.bias1) ? g.bias1[i] : 0.f;
    for (int i = tid; i < NT2 * 16; i += kThreads) sb2[i] = (i < g.M2 && g.bias2) ? g.bias2[i] : 0.f;

    // ---- layer-1 B operand: the K1 channels of this lane's pixel (k-octet 4 s + kq of k-step s) ----
    // (x_group > 0: the K1 rows are x_group-row slices of consecutive images, '(B T) C -> B (T C)': octet oc lives in group
    // oc / goct at octet oc % goct; a k-step never straddles a group: x_group % 32 == 0, host-checked)
    const int noct = (g.K1 + 7) / 8;
    const int goct = g.x_group > 0 ? g.x_group / 8 : noct;         // octets per group (wave-uniform)
    const int gbytes = g.x_group > 0 ? (int)(g.x_group_stride * 2) : 0;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(g.X)) + (int64_t)z * g.strideX * 2, 0, a.x_span, 0x00020000);
    f16x8 b[NK1];
#pragma unroll
    for (int s = 0; s < NK1; ++s) {
        const int oc = 4 * s + kq;
        const int og = __builtin_amdgcn_readfirstlane((4 * s) / goct);          // group of the k-step
        b[s] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(
            rx, (pin && oc < noct) ? ((oc - og * goct) * (int)g.ldx + px) * 16 : kOob, og * gbytes, 0));
    }
    wait_vm<0>();
    __syncthreads();

    f32x4 acc2[NT2];                                              // layer-2 accumulators start at the (pre-scaled) bias
#pragma unroll
    for (int t = 0; t < NT2; ++t) acc2[t] = *reinterpret_cast<const f32x4*>(sb2 + 16 * t + 4 * kq);

#ifdef SF_PAIR_TIMERS
    const long long ts1 = __builtin_readcyclecounter();
    tprev = ts1;
#endif
    int gs = 0, slot = 0;                                         // global stage index, its ring slot
    for (int m = 0; m < a.hp; ++m) {
        f32x4 a1[2];
        a1[0] = *reinterpret_cast<const f32x4*>(sb1 + 32 * m + 4 * kq);
        a1[1] = *reinterpret_cast<const f32x4*>(sb1 + 32 * m + 16 + 4 * kq);
        f16x8 hf = {};
        static_for<0, NSTG>([&](auto st_tag) {
            constexpr int st = decltype(st_tag)::value;
            // stage gs + RING - 1 goes into the slot every wave finished reading before the barrier that ended the previous stage
            // (issued past the end too -- clamped to the last stage -- so that the counted wait below sees the same queue every trip)
#if SF_PAIR_ONE_LOADER
            // (experiment: ONE wave per stage issues all its pieces -- fewer waves queueing on the CU's address path at a time)
            if (wave == gs % kWaves) {
#pragma unroll
                for (int i = 0; i < S; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(smem + (slot == 0 ? RING - 1 : slot - 1) * kStage + i * 1024), 16,
                                                             lane * 16, min(gs + RING - 1, last_stage) * kStage + i * 1024, 0, 0);
            }
#else
            issue_stage(gs + RING - 1, slot == 0 ? RING - 1 : slot - 1);
#endif
            SF_PT_STAMP(t_issue)
            const char* sp = smem + slot * kStage + lane * 16;
            static_for<0, S>([&](auto i_tag) {
                constexpr int i = decltype(i_tag)::value, f = st * S + i;
                if constexpr (f == NA) {
                    // hidden rows 32 m .. 32 m + 31 of this lane's pixel: gelu(alpha1 * acc) -> fp16: the lane's 8 k-values of the
                    // layer-2 k-step m (k = 8 kq + i <-> row 4 kq + i | 16 + 4 kq + i - 4: the packed order of W2's columns)
                    f32x2 v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[q][0] = g.alpha1 * a1[q >> 1][2 * (q & 1)];
                        v[q][1] = g.alpha1 * a1[q >> 1][2 * (q & 1) + 1];
                        v[q] = sf::gelu_poly2(v[q]);
                        hf[2 * q] = (_Float16)v[q][0];
                        hf[2 * q + 1] = (_Float16)v[q][1];
                    }
                }
                if constexpr (f < NA) {
                    constexpr int u = f / (NK1 * PM1), s = (f % (NK1 * PM1)) / PM1;
                    a1[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(sp + i * 1024), b[s], a1[u], 0, 0, 0);
                } else if constexpr (f < F) {
                    constexpr int t = (f - NA) / PM2;
                    acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(sp + i * 1024), hf, acc2[t], 0, 0, 0);
                }
            });
            // Every fragment read of this stage must have EXECUTED before the barrier: the slot is refilled by whichever wave passes
            // the barrier first, and a read that was only issued (hipcc sinks the last MFMAs and their lgkmcnt waits below the
            // s_barrier -- the builtin is no memory barrier to it) then races with the refill's DMA.  Seen as run-to-run differences
            // (tests/test_gpu_ffn_pair.py::test_ffn_pair_is_deterministic) with both ring depths; the explicit drain removed them.
            SF_PT_STAMP(t_mma)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if SF_PAIR_ONE_LOADER
            static_assert(!SF_PAIR_ONE_LOADER || RING == 3, "one-loader experiment: ring of 3");
            if (wave == (gs + kWaves - 1) % kWaves) wait_vm<0>();  // the wave that issued stage gs + 1 (at the top of stage gs - 1)
#else
            wait_vm<PCS * (RING - 2)>();                           // this wave's pieces of the NEXT stage have landed (later ones fly on) ...
#endif
            SF_PT_STAMP(t_drain)
            __builtin_amdgcn_s_barrier();                          // ... everyone's; nobody reads this stage's slot any more
            SF_PT_STAMP(t_bar)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            ++gs;
            slot = (slot == RING - 1) ? 0 : slot + 1;
        });
    }

#ifdef SF_PAIR_TIMERS
    const long long ts2 = __builtin_readcyclecounter();
#endif
    wait_vm<0>();                                                  // (pieces requested past the end must land before the LDS is released)
    constexpr bool kR32 = MODE == 2;                               // MODE 2 = mode 1 with the residual from fp32 planes (SfFfnPair.R32)
    if constexpr (MODE >= 1) {                                     // the depthwise 1x1 parameters into the (now idle) ring
        __syncthreads();
        for (int i = tid; i < NT2 * 16; i += kThreads) {
            sdw[i] = (i < g.M2) ? g.dw_w[i] : 0.f;
            sdb[i] = (i < g.M2) ? g.dw_b[i] : 0.f;
        }
        __syncthreads();
    }
    // ---- epilogue: rows 16 t + 4 kq + e of pixel px.  Lane-dependent address parts (pixel, kq) travel in the VECTOR offset, the
    // tile / element part in the scalar offset (a lane-dependent scalar offset makes hipcc serialise the access lane group by
    // lane group: a "waterfall" loop around every load and store) ----
    if constexpr (MODE == 0) {
        char* c16 = reinterpret_cast<char*>(g.C16) + (int64_t)z * g.strideC16 * 2;
        const int moct = (g.M2 + 7) / 8;
        const __amdgpu_buffer_rsrc_t rc16 = __builtin_amdgcn_make_buffer_rsrc(c16, 0, g.C16 ? (int)((int64_t)moct * g.ldc16 * 16) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rc32 = __builtin_amdgcn_make_buffer_rsrc(
            g.C ? reinterpret_cast<char*>(g.C) + (int64_t)z * g.strideC * 4 : nullptr, 0,
            g.C ? (int)(((int64_t)(g.M2 - 1) * g.ldc + g.N) * 4) : 0, 0x00020000);
        // k-octet planes: octet 2 t + (kq >> 1), halves 4 (kq & 1) .. + 3 of pixel px
        const int v16 = ((kq >> 1) * (int)g.ldc16 + px) * 16 + 8 * (kq & 1);
        const int v32 = (4 * kq * (int)g.ldc + px) * 4;                                   // fp32 planes: + (16 t + e) rows
#pragma unroll
        for (int t = 0; t < NT2; ++t) {
            f32x2 v[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                v[q][0] = g.alpha2 * acc2[t][2 * q];
                v[q][1] = g.alpha2 * acc2[t][2 * q + 1];
                if (g.gelu_out) v[q] = g.C ? sf::gelu_erf2(v[q]) : sf::gelu_poly2(v[q]);      // (wave-uniform)
            }
            const int r0 = 16 * t + 4 * kq;
            if (g.C) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ve = v[e >> 1][e & 1];                // (bit_cast of a vector element lvalue reads element 0)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ve), rc32,
                                                          (pin && r0 + e < g.M2) ? v32 : kOob, (16 * t + e) * (int)g.ldc * 4, 0);
                }
            }
            if (g.C16) {
                // every row of an octet that starts below M2 is written (rows >= M2: gelu / identity of 0: finite -- the consumer
                // multiplies them by zero weights); c16_partial: rows >= M2 of the last octet belong to someone else
                f16x4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (_Float16)v[e >> 1][e & 1];
                const bool oct_in = pin && (2 * t + (kq >> 1)) < moct;
                const bool whole = !g.c16_partial || r0 + 3 < g.M2;                        // (per lane)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h), rc16, (oct_in && whole) ? v16 : kOob,
                                                      2 * t * (int)g.ldc16 * 16, 0);
                if (g.c16_partial && 16 * t + 16 > g.M2 && 16 * t < g.M2) {               // (wave-uniform) the tile that holds row M2 - 1
#pragma unroll
                    for (int e = 0; e < 3; ++e) {
                        const _Float16 he = h[e];
                        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, he), rc16,
                                                              (oct_in && !whole && r0 + e < g.M2) ? v16 + 2 * e : kOob,
                                                              2 * t * (int)g.ldc16 * 16, 0);
                    }
                }
            }
        }
    } else {
        char* c16 = reinterpret_cast<char*>(g.C16) + (int64_t)z * g.strideC16 * 2;
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(c16, 0, (int)(((int64_t)(g.M2 - 1) * g.ldc16 + g.N) * 2), 0x00020000);
        // residual x (update.py:31: x + ffn1(x)): rows 16 t + 4 kq .. + 3 = half an octet of the input planes, all tiles requested first
        const __amdgpu_buffer_rsrc_t rr32 = __builtin_amdgcn_make_buffer_rsrc(
            kR32 ? const_cast<char*>(reinterpret_cast<const char*>(g.R32)) + (int64_t)z * g.strideR32 * 4 : nullptr, 0, kR32 ? a.r32_span : 0,
            0x00020000);
        const int vr32 = (4 * kq * (int)g.ldr32 + px) * 4;
        const int g32bytes = (int)(g.r32_group_stride * 4);
        const int vr = ((kq >> 1) * (int)g.ldx + px) * 16 + 8 * (kq & 1);
        const int vrow = (4 * kq * (int)g.ldc16 + px) * 2;                                // fp16 rows: + (16 t + e) rows
        // (tiles in chunks of CH: the residual of a chunk is requested before its first GELU, not all NT2 tiles at once: registers)
        constexpr int CH = (NT2 % 7 == 0) ? 7 : (NT2 % 8 == 0 ? 8 : (NT2 % 4 == 0 ? 4 : 1));
#pragma unroll
        for (int t0 = 0; t0 < NT2; t0 += CH) {
        u32x2 rk[CH];
        f32x4 rf[CH];                                             // (R32: the residual from the fp32 planes instead of the fp16 operand)
        if constexpr (kR32) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int og = __builtin_amdgcn_readfirstlane((2 * (t0 + c)) / goct);
                const int rowg = 16 * (t0 + c) - og * goct * 8;                            // first row of the tile inside its group
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    rf[c][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        rr32, (pin && 16 * (t0 + c) + 4 * kq + e < g.M2) ? vr32 : kOob, og * g32bytes + (rowg + e) * (int)g.ldr32 * 4, 0));
            }
        } else {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int og = __builtin_amdgcn_readfirstlane((2 * (t0 + c)) / goct);          // (goct is even: the pair of octets shares a group)
            rk[c] = __builtin_amdgcn_raw_buffer_load_b64(rx, (pin && 2 * (t0 + c) + (kq >> 1) < noct) ? vr : kOob,
                                                         og * gbytes + (2 * (t0 + c) - og * goct) * (int)g.ldx * 16, 0);
        }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int t = t0 + c;
            const int r0 = 16 * t + 4 * kq;
            const f32x4 dww = *reinterpret_cast<const f32x4*>(sdw + r0), dwb = *reinterpret_cast<const f32x4*>(sdb + r0);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned ru = rk[c][q];                         // (bit_cast of a vector element lvalue reads element 0)
                const f16x2 rh = __builtin_bit_cast(f16x2, ru);
                f32x2 v, r, w2, b2;
                v[0] = g.alpha2 * acc2[t][2 * q]; v[1] = g.alpha2 * acc2[t][2 * q + 1];
                r[0] = kR32 ? rf[c][2 * q] : (float)rh[0];
                r[1] = kR32 ? rf[c][2 * q + 1] : (float)rh[1];
                w2[0] = dww[2 * q]; w2[1] = dww[2 * q + 1];
                b2[0] = dwb[2 * q]; b2[1] = dwb[2 * q + 1];
                const f32x2 x1 = sf::gelu_poly2(r + v);
                const f32x2 x2 = sf::gelu_poly2(x1 + (w2 * x1 + b2));
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const _Float16 he = (_Float16)x2[e];
                    const int row = r0 + 2 * q + e;
                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, he), rc,
                                                          (pin && row < g.M2) ? vrow : kOob, (16 * t + 2 * q + e) * (int)g.ldc16 * 2, 0);
                }
            }
        }
        }
    }
#ifdef SF_PAIR_TIMERS
    if (a.ts && lane == 0 && blockIdx.x < 8192) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long* d = a.ts + ((int64_t)blockIdx.x * NW + wave) * 8;
        d[0] = ts1 - ts0; d[1] = t_issue; d[2] = t_mma; d[3] = t_drain; d[4] = t_bar; d[5] = ts2 - ts1; d[6] = __builtin_readcyclecounter() - ts2;
        d[7] = nstage;
    }
#endif
}

template <int NK1, int NT2, int MODE, int NW>
int launch_pm(PairArgs& a, hipStream_t st) {
    a.ntile = sf::ceil_div(a.p.N, NW * kPxWave);
    const int64_t nwg = (int64_t)a.ntile * a.p.batch;
    if (nwg >= ((int64_t)1 << 31)) return sf::fail(SF_ERR_UNSUPPORTED, "sf_ffn_pair: grid too large");
    const dim3 grid((unsigned)nwg);
    const int pm = a.p.pm1 * 10 + a.p.pm2;
    switch (pm) {
        case 11: hipLaunchKernelGGL((ffn_pair_kernel<NK1, NT2, 1, 1, MODE, NW>), grid, dim3(NW * 64), 0, st, a); break;
        case 21: hipLaunchKernelGGL((ffn_pair_kernel<NK1, NT2, 2, 1, MODE, NW>), grid, dim3(NW * 64), 0, st, a); break;
        case 22: hipLaunchKernelGGL((ffn_pair_kernel<NK1, NT2, 2, 2, MODE, NW>), grid, dim3(NW * 64), 0, st, a); break;
        default: return sf::fail(SF_ERR_UNSUPPORTED, "sf_ffn_pair: products (%d, %d) not built (1,1 / 2,1 / 2,2)", a.p.pm1, a.p.pm2);
    }
    return sf::check_launch("sf_ffn_pair");
}

}  // namespace

// fragments per 32 hidden rows of the packed stream, padded to whole stages
extern "C" int sf_ffn_pair_frags(int K1, int M2, int pm1, int pm2) {
    if (K1 <= 0 || M2 <= 0 || pm1 < 1 || pm1 > 2 || pm2 < 1 || pm2 > 2) return 0;
    const int nk1 = (K1 + 31) / 32, nt2 = (M2 + 15) / 16;
    const int f = 2 * nk1 * pm1 + nt2 * pm2;
    return (f + S - 1) / S * S;
}

extern "C" int sf_ffn_pair(const SfFfnPair* p, void* stream) {
    SF_REQUIRE(p && p->X && p->wstream, "sf_ffn_pair: null pointer");
    const SfFfnPair& g = *p;
    SF_REQUIRE(g.K1 > 0 && g.H > 0 && g.M2 > 0 && g.N > 0 && g.batch > 0, "sf_ffn_pair: bad dims");
    SF_REQUIRE(g.H <= kMaxH && g.M2 <= kMaxM2, "sf_ffn_pair: H <= %d and M2 <= %d", kMaxH, kMaxM2);
    SF_REQUIRE(g.mode == 0 || g.mode == 1, "sf_ffn_pair: mode must be 0 (ffn2 pair) or 1 (ffn1 pair)");
    SF_REQUIRE((reinterpret_cast<uintptr_t>(g.X) & 15) == 0 && (g.strideX & 7) == 0 && g.ldx >= g.N &&
                   (reinterpret_cast<uintptr_t>(g.wstream) & 15) == 0,
               "sf_ffn_pair: X (k-octet planes) and the weight stream must be 16-byte aligned, strideX %% 8 == 0, ldx >= N");
    const int64_t lim = (int64_t)1 << 30;
    SF_REQUIRE(g.x_group >= 0 && (g.x_group == 0 || (g.x_group % 32 == 0 && g.K1 % g.x_group == 0 && (g.x_group_stride & 7) == 0 && g.x_group_stride >= 0)),
               "sf_ffn_pair: x_group must be 0 or a multiple of 32 that divides K1, x_group_stride %% 8 == 0");
    const int64_t x_span = g.x_group > 0 ? (int64_t)(g.K1 / g.x_group - 1) * g.x_group_stride * 2 + (int64_t)(g.x_group / 8) * g.ldx * 16
                                         : (int64_t)((g.K1 + 7) / 8) * g.ldx * 16;
    SF_REQUIRE(x_span < lim, "sf_ffn_pair: input image larger than 1 GiB");
    if (g.mode == 0) {
        SF_REQUIRE(g.C || g.C16, "sf_ffn_pair: mode 0 needs C (fp32 planes) and / or C16 (k-octet planes)");
        SF_REQUIRE(!g.C16 || ((reinterpret_cast<uintptr_t>(g.C16) & 15) == 0 && (g.strideC16 & 7) == 0 && g.ldc16 >= g.N &&
                              (int64_t)((g.M2 + 7) / 8) * g.ldc16 * 16 < lim),
                   "sf_ffn_pair: C16 (k-octet planes) must be 16-byte aligned, strideC16 %% 8 == 0, ldc16 >= N, image < 1 GiB");
        SF_REQUIRE(!g.C || (g.ldc >= g.N && ((int64_t)(g.M2 - 1) * g.ldc + g.N) * 4 < lim), "sf_ffn_pair: C: ldc >= N, image < 1 GiB");
    } else {
        SF_REQUIRE(g.C16 && g.dw_w && g.dw_b && g.M2 == g.K1, "sf_ffn_pair: mode 1 needs C16 (fp16 rows), dw_w, dw_b and M2 == K1");
        SF_REQUIRE(g.ldc16 >= g.N && ((int64_t)(g.M2 - 1) * g.ldc16 + g.N) * 2 < lim, "sf_ffn_pair: C16 rows: ldc16 >= N, image < 1 GiB");
    }
    PairArgs a;
#ifdef SF_PAIR_TIMERS
    a.ts = getenv("SF_PAIR_TS_BUF") ? (long long*)strtoull(getenv("SF_PAIR_TS_BUF"), nullptr, 0) : nullptr;
#endif
    a.p = g;
    a.ntile = 0;
    a.hp = (g.H + 31) / 32;
    a.x_span = (int)x_span;
    a.r32_span = 0;
    if (g.R32) {
        SF_REQUIRE(g.mode == 1 && g.ldr32 >= g.N && (reinterpret_cast<uintptr_t>(g.R32) & 3) == 0 && g.r32_group_stride >= 0,
                   "sf_ffn_pair: R32 (fp32 residual planes) is a mode-1 option, ldr32 >= N");
        const int64_t rows_g = g.x_group > 0 ? g.x_group : g.M2;
        const int64_t sp = (g.x_group > 0 ? (int64_t)(g.M2 / g.x_group - 1) * g.r32_group_stride : 0) * 4 + ((rows_g - 1) * g.ldr32 + g.N) * 4;
        SF_REQUIRE(sp < lim, "sf_ffn_pair: residual image larger than 1 GiB");
        a.r32_span = (int)sp;
    }
    const int fpad = sf_ffn_pair_frags(g.K1, g.M2, g.pm1, g.pm2);
    SF_REQUIRE(fpad > 0, "sf_ffn_pair: products must be 1 or 2");
    a.w_bytes = (int64_t)a.hp * fpad * 1024;
    SF_REQUIRE(g.wstream_bytes >= a.w_bytes && a.w_bytes < lim, "sf_ffn_pair: weight stream too small (need %lld bytes) or > 1 GiB",
               (long long)a.w_bytes);
    hipStream_t st = (hipStream_t)stream;
    const int nk1 = (g.K1 + 31) / 32, nt2 = (g.M2 + 15) / 16;
    const int kmode = (g.mode == 1 && g.R32) ? 2 : g.mode;           // (kernel MODE 2: mode 1 with the fp32 residual)
#define SF_PAIR_CASE(NK1_, NT2_, MODE_, NW_) \
    if (nk1 == NK1_ && nt2 == NT2_ && kmode == MODE_) return launch_pm<NK1_, NT2_, MODE_, NW_>(a, st)
    // the SK blocks of the update block (update.py:313-339, 739-782): C = 128 / 256 / 324.  NW = waves per workgroup: 8 (128 pixels
    // share every weight stage: half the L2 -> LDS traffic per pixel) where the kernel fits 128 registers (4 waves per SIMD)
    SF_PAIR_CASE(4, 8, 1, 8);      // convf2.ffn1   128 -> 192 -> 128
    SF_PAIR_CASE(8, 16, 1, 4);     // convc2.ffn1, conv.ffn1   256 -> 384 -> 256  (128 accumulator + operand registers: 3 waves per SIMD)
    SF_PAIR_CASE(11, 21, 1, 4);    // convc1.ffn1   324 -> 486 -> 324
    SF_PAIR_CASE(4, 4, 0, 8);      // convf2.ffn2   128 -> 192 -> 64
    SF_PAIR_CASE(8, 12, 0, 8);     // convc2.ffn2   256 -> 384 -> 192
    SF_PAIR_CASE(8, 8, 0, 8);      // conv.ffn2     256 -> 384 -> 126
    SF_PAIR_CASE(11, 16, 0, 4);    // convc1.ffn2   324 -> 486 -> 256
    // the flow head (update.py:744, 775): its input is the '(B T) C -> B (T C)' view of the hidden state (x_group = 128), T - 1 = 3 / 2 / 1 frames
    SF_PAIR_CASE(12, 24, 1, 4);    // flow_head.ffn1   384 -> 576 -> 384   (T = 4)
    SF_PAIR_CASE(12, 24, 2, 4);    //   ... with the residual from the fp32 planes of the hidden state (the engine's form)
    SF_PAIR_CASE(8, 16, 2, 4);     //   (T = 3)
    SF_PAIR_CASE(4, 8, 2, 8);      //   (T = 2)
    SF_PAIR_CASE(12, 1, 0, 8);     // flow_head.ffn2   384 -> 576 -> 6
    SF_PAIR_CASE(8, 1, 0, 8);      //                  256 -> 384 -> 4     (T = 3; its ffn1 is the 256 -> 384 -> 256 case above)
    SF_PAIR_CASE(4, 1, 0, 8);      //                  128 -> 192 -> 2     (T = 2)
#undef SF_PAIR_CASE
    return sf::fail(SF_ERR_UNSUPPORTED, "sf_ffn_pair: shape K1 = %d, M2 = %d, mode %d%s not built", g.K1, g.M2, g.mode, g.R32 ? " with an fp32 residual" : "");
}
