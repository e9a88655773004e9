// The temporal transformer block of the update step in ONE launch.
//
// Reference: core/update.py:459-484,502-513 (TemporalLayer2 -> timm Block: x += proj(attn(LN1 x)); x += fc2(gelu(fc1(LN2 x)))), called
// once per iteration at update.py:770 on the tokens (b, pixel) x (T - 1 frames) x 128 channels.  The tokens of ONE pixel are independent
// of every other pixel, so the whole block is a per-pixel chain: the unfused form runs it as 7 launches (LayerNorm, qkv GEMM,
// attention core, proj GEMM, LayerNorm, fc1 GEMM, fc2 GEMM) that hand 5 intermediate tensors through memory.  Here a wave owns 16
// pixels x TT frames and keeps everything between the block's input and output in registers:
//   * 16 x 16 x 32 MFMA tiles as in csrc/ffn_pair.hip: the C/D layout of a tile (column = pixel, rows 4 kq .. 4 kq + 3) of TWO
//     consecutive row tiles IS a B fragment (32 k) of the next layer, given that the next layer's weight columns are packed in that
//     order (ops.PackedTemporal): LN1 -> qkv -> attention -> proj -> LN2 -> fc1 -> GELU -> fc2 chain without an LDS round trip;
//   * one weight fragment (1 KB, one ds_read_b128) feeds TT MFMAs -- the frames of a pixel share the weights;
//   * the attention over the TT frames of a pixel (1 head, 128 channels: scores TT x TT) runs on the accumulator layout: per-lane
//     partial dot products over its 32 channels, two cross-lane steps (lane ^ 16, lane ^ 32) for the other three quarters;
//   * LayerNorm statistics the same way (in-lane sums over 32 channels + the two cross-lane steps);
//   * all four layers' weights are ONE host-packed stream of 1-KB fragments in consumption order (256 x PM fragments: 256 / 512 KB),
//     moved L2 -> LDS by DMA through a ring of 16-KB stages shared by the workgroup's 4 waves (64 pixels), as in ffn_pair.hip.
// Input: the fp16 k-octet copy of the motion features (the B operand format; also the residual -- the same rounding every other
// consumer of that tensor sees).  Output: fp32 planes (+ their k-octet copy) into the GRU's input buffer.
// Arithmetic: the config-2 class (activations enter every product as fp16, PM = 1: fp16 weights, PM = 2: hi + lo; fp32 accumulation,
// fp32 LayerNorm / softmax / GELU).
#include "sf_common.h"

#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;
using sf::f32x2;

constexpr int kC = 128, kH = 256;                          // channels, MLP hidden rows (timm Block, mlp_ratio 2)
constexpr int kWaves = 4, kThreads = 256, kPxWave = 16, kPxWg = kWaves * kPxWave;
constexpr int S = 16, kStage = S * 1024, RING = 3, PCS = S / kWaves;
constexpr int kOob = 1 << 30;

struct TbArgs {
    SfTemporalBlock p;
    int ntile;            // pixel tiles (64 pixels) per clip
    int64_t w_bytes;
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

// sum over the four lanes (kq = 0 .. 3) that hold the same pixel: lanes l, l ^ 16, l ^ 32, l ^ 48
__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

__device__ __forceinline__ u32x2 pack4(const f32x4 v) {
    f16x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = (_Float16)v[e];
    return __builtin_bit_cast(u32x2, h);
}

__device__ __forceinline__ f16x8 frag_of(const u32x2 lo, const u32x2 hi) {   // rows 4 kq .. + 3 of two consecutive row tiles
    u32x4 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = hi[0]; r[3] = hi[1];
    return __builtin_bit_cast(f16x8, r);
}

template <int TT, int PM>
__global__ __launch_bounds__(kThreads, 2) void temporal_block_kernel(const TbArgs a) {
    const SfTemporalBlock& g = a.p;
    constexpr int FT = 4 * PM;                                    // fragments of one 16-row tile over K = 128
    __shared__ __attribute__((aligned(1024))) char smem[RING * kStage + 1024 * 4];
    float* sp_ln1w = reinterpret_cast<float*>(smem + RING * kStage);
    float* sp_ln1b = sp_ln1w + 128;
    float* sp_ln2w = sp_ln1w + 256;
    float* sp_ln2b = sp_ln1w + 384;
    float* sp_bp = sp_ln1w + 512;                                 // proj bias (pre-scaled), 128
    float* sp_b2 = sp_ln1w + 640;                                 // fc2 bias (pre-scaled), 128
    float* sp_b1 = sp_ln1w + 768;                                 // fc1 bias (pre-scaled), 256
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, l15 = lane & 15;
    const int tile = blockIdx.x % a.ntile, z = blockIdx.x / a.ntile;
    const int px = tile * kPxWg + wave * kPxWave + l15;
    const bool pin = px < g.N;

    // ---- the weight stream (ffn_pair.hip's ring): stage s = bytes [16 KB s, 16 KB (s + 1)); wave w moves pieces w, w + 4, ... ----
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.wstream), 0, (int)a.w_bytes, 0x00020000);
    // (stages requested past the end of the stream -- the loop keeps the request count per trip constant -- re-read the LAST stage
    // into a slot nobody reads any more: the stage offset travels in the scalar offset, which the raw-buffer range check of gfx9
    // does not cover, so "out of range: zeros" must not be relied on: ADVICE r5)
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

    if (tid < 128) {
        sp_ln1w[tid] = g.ln1_w[tid]; sp_ln1b[tid] = g.ln1_b[tid];
        sp_ln2w[tid] = g.ln2_w[tid]; sp_ln2b[tid] = g.ln2_b[tid];
        sp_bp[tid] = g.bias_proj ? g.bias_proj[tid] : 0.f;
        sp_b2[tid] = g.bias_fc2 ? g.bias_fc2[tid] : 0.f;
    }
    sp_b1[tid] = g.bias_fc1 ? g.bias_fc1[tid] : 0.f;

    // ---- tokens of this lane's pixel: k-octet 4 s + kq of k-step s, frame t = image z TT + t ----
    __amdgpu_buffer_rsrc_t rx[TT];
    f16x8 hb[TT][4];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        rx[t] = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(g.X16)) + ((int64_t)z * TT + t) * g.strideX * 2, 0, (int)((int64_t)(kC / 8) * g.ldx * 16),
            0x00020000);
#pragma unroll
        for (int s = 0; s < 4; ++s)
            hb[t][s] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rx[t], pin ? ((4 * s + kq) * (int)g.ldx + px) * 16 : kOob, 0, 0));
    }
    wait_vm<0>();
    __syncthreads();

    // ---- LayerNorm 1 over the 128 channels (B layout: the lane holds channels 32 s + 8 kq .. + 7) -> fp16 fragments in place ----
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        float xs[32], sum = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                xs[8 * s + i] = (float)hb[t][s][i];
                sum += xs[8 * s + i];
            }
        const float mean = quad_sum(sum) * (1.0f / kC);
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            xs[i] -= mean;
            sq = fmaf(xs[i], xs[i], sq);
        }
        const float rstd = 1.0f / sqrtf(quad_sum(sq) * (1.0f / kC) + g.eps);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(sp_ln1w + 32 * s + 8 * kq), w1 = *reinterpret_cast<const f32x4*>(sp_ln1w + 32 * s + 8 * kq + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sp_ln1b + 32 * s + 8 * kq), b1 = *reinterpret_cast<const f32x4*>(sp_ln1b + 32 * s + 8 * kq + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                hb[t][s][i] = (_Float16)fmaf(xs[8 * s + i] * rstd, w0[i], b0[i]);
                hb[t][s][4 + i] = (_Float16)fmaf(xs[8 * s + 4 + i] * rstd, w1[i], b1[i]);
            }
        }
    }

    int gs = 0, slot = 0;                                         // global stage index, its ring slot
    auto stage_begin = [&]() -> const char* {
        issue_stage(gs + RING - 1, slot == 0 ? RING - 1 : slot - 1);      // (past the end: the last stage again -- the counted wait sees the same queue)
        return smem + slot * kStage + lane * 16;
    };
    auto stage_end = [&]() {
        // every fragment read of the stage has EXECUTED before the barrier (ffn_pair.hip: the refill race)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait_vm<PCS * (RING - 2)>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        ++gs;
        slot = (slot == RING - 1) ? 0 : slot + 1;
    };

    // ---- phase 1: q (row tiles 0 .. 7) and k (8 .. 15) of every frame, kept as fp16 in the accumulator layout ----
    u32x2 qk[16][TT];
    {
        constexpr int TPS = S / FT;                               // tiles per stage
        static_for<0, 16 / TPS>([&](auto j_tag) {
            constexpr int j = decltype(j_tag)::value;
            const char* sp = stage_begin();
            f32x4 acc[TPS][TT];
#pragma unroll
            for (int u = 0; u < TPS; ++u)
#pragma unroll
                for (int t = 0; t < TT; ++t) acc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            static_for<0, S>([&](auto i_tag) {
                constexpr int i = decltype(i_tag)::value, u = i / FT, ks = (i % FT) / PM;
                const f16x8 fr = *reinterpret_cast<const f16x8*>(sp + i * 1024);
#pragma unroll
                for (int t = 0; t < TT; ++t) acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr, hb[t][ks], acc[u][t], 0, 0, 0);
            });
#pragma unroll
            for (int u = 0; u < TPS; ++u)
#pragma unroll
                for (int t = 0; t < TT; ++t) qk[j * TPS + u][t] = pack4(g.alpha_qkv * acc[u][t]);
            stage_end();
        });
    }

    // ---- attention weights of the pixel: softmax_u(scale q_t . k_u) over the TT frames (fp32) ----
    float pw[TT][TT];
    {
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int u = 0; u < TT; ++u) {
                float s = 0.f;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const u32x2 qv = qk[m][t], kv = qk[8 + m][u];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const unsigned qe = qv[e], ke = kv[e];
                        const f16x2 qh = __builtin_bit_cast(f16x2, qe), kh = __builtin_bit_cast(f16x2, ke);
                        s = fmaf((float)qh[0], (float)kh[0], s);
                        s = fmaf((float)qh[1], (float)kh[1], s);
                    }
                }
                pw[t][u] = quad_sum(s) * g.scale;
            }
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            float mx = pw[t][0];
#pragma unroll
            for (int u = 1; u < TT; ++u) mx = fmaxf(mx, pw[t][u]);
            float sum = 0.f;
#pragma unroll
            for (int u = 0; u < TT; ++u) {
                pw[t][u] = __expf(pw[t][u] - mx);
                sum += pw[t][u];
            }
            const float inv = 1.0f / sum;
#pragma unroll
            for (int u = 0; u < TT; ++u) pw[t][u] *= inv;
        }
    }

    // ---- phase 2: v two row tiles at a time -> attention output (one proj k-step) -> proj accumulators.  They start at
    // ss_proj * residual + the pre-scaled bias, so that alpha_proj * acc IS x = tokens + proj(...) ----
    f32x4 xa[8][TT];
    {
        const int vr = ((kq >> 1) * (int)g.ldx + px) * 16 + 8 * (kq & 1);          // rows 16 m + 4 kq .. + 3 = half an octet of the input planes
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const u32x2 rk = __builtin_amdgcn_raw_buffer_load_b64(rx[t], pin ? vr : kOob, 2 * m * (int)g.ldx * 16, 0);
                const f32x4 bp = *reinterpret_cast<const f32x4*>(sp_bp + 16 * m + 4 * kq);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const unsigned ru = rk[q];
                    const f16x2 rh = __builtin_bit_cast(f16x2, ru);
                    xa[m][t][2 * q] = fmaf(g.ss_proj, (float)rh[0], bp[2 * q]);
                    xa[m][t][2 * q + 1] = fmaf(g.ss_proj, (float)rh[1], bp[2 * q + 1]);
                }
            }
    }
    for (int pp = 0; pp < 4; ++pp) {
        f32x4 av[2][TT];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t = 0; t < TT; ++t) av[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        f16x8 of[TT];
        static_for<0, PM>([&](auto st_tag) {
            constexpr int st = decltype(st_tag)::value;
            const char* sp = stage_begin();
            static_for<0, S>([&](auto i_tag) {
                constexpr int i = decltype(i_tag)::value, f = st * S + i;
                if constexpr (f == 2 * FT) {
                    // o_t = sum_u p[t][u] v_u on the accumulator layout; the two tiles' rows of this lane = its 8 k-values of proj's k-step pp
#pragma unroll
                    for (int t = 0; t < TT; ++t) {
                        f32x4 o0 = pw[t][0] * av[0][0], o1 = pw[t][0] * av[1][0];
#pragma unroll
                        for (int u = 1; u < TT; ++u) {
                            o0 += pw[t][u] * av[0][u];
                            o1 += pw[t][u] * av[1][u];
                        }
                        of[t] = frag_of(pack4(g.alpha_qkv * o0), pack4(g.alpha_qkv * o1));
                    }
                }
                const f16x8 fr = *reinterpret_cast<const f16x8*>(sp + i * 1024);
                if constexpr (f < 2 * FT) {
                    constexpr int u = f / FT, ks = (f % FT) / PM;
#pragma unroll
                    for (int t = 0; t < TT; ++t) av[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr, hb[t][ks], av[u][t], 0, 0, 0);
                } else {
                    constexpr int m = (f - 2 * FT) / PM;
#pragma unroll
                    for (int t = 0; t < TT; ++t) xa[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr, of[t], xa[m][t], 0, 0, 0);
                }
            });
            stage_end();
        });
    }

    // ---- x = tokens + attention branch (fp32, accumulator layout); LayerNorm 2 -> fc1's B fragments; the fc2 accumulators start at
    // ss_fc2 * x + the pre-scaled bias ----
    f16x8 h2[TT][4];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        float sum = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            xa[m][t] = g.alpha_proj * xa[m][t];
            sum += (xa[m][t][0] + xa[m][t][1]) + (xa[m][t][2] + xa[m][t][3]);
        }
        const float mean = quad_sum(sum) * (1.0f / kC);
        float sq = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = xa[m][t][e] - mean;
                sq = fmaf(d, d, sq);
            }
        const float rstd = 1.0f / sqrtf(quad_sum(sq) * (1.0f / kC) + g.eps);
        u32x2 yk[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(sp_ln2w + 16 * m + 4 * kq), b = *reinterpret_cast<const f32x4*>(sp_ln2b + 16 * m + 4 * kq);
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(sp_b2 + 16 * m + 4 * kq);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = fmaf((xa[m][t][e] - mean) * rstd, w[e], b[e]);
                xa[m][t][e] = fmaf(g.ss_fc2, xa[m][t][e], b2[e]);
            }
            yk[m] = pack4(y);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) h2[t][s] = frag_of(yk[2 * s], yk[2 * s + 1]);
    }

    // ---- phase 3: fc1 32 hidden rows at a time -> GELU -> one fc2 k-step -> the fc2 accumulators ----
    for (int hg = 0; hg < 8; ++hg) {
        f32x4 a1[2][TT];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(sp_b1 + 32 * hg + 16 * u + 4 * kq);
#pragma unroll
            for (int t = 0; t < TT; ++t) a1[u][t] = b1;
        }
        f16x8 hf[TT];
        static_for<0, PM>([&](auto st_tag) {
            constexpr int st = decltype(st_tag)::value;
            const char* sp = stage_begin();
            static_for<0, S>([&](auto i_tag) {
                constexpr int i = decltype(i_tag)::value, f = st * S + i;
                if constexpr (f == 2 * FT) {
#pragma unroll
                    for (int t = 0; t < TT; ++t) {
                        u32x2 hk[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            f32x2 v0, v1;
                            v0[0] = g.alpha_fc1 * a1[u][t][0]; v0[1] = g.alpha_fc1 * a1[u][t][1];
                            v1[0] = g.alpha_fc1 * a1[u][t][2]; v1[1] = g.alpha_fc1 * a1[u][t][3];
                            v0 = sf::gelu_poly2(v0);
                            v1 = sf::gelu_poly2(v1);
                            f32x4 v;
                            v[0] = v0[0]; v[1] = v0[1]; v[2] = v1[0]; v[3] = v1[1];
                            hk[u] = pack4(v);
                        }
                        hf[t] = frag_of(hk[0], hk[1]);
                    }
                }
                const f16x8 fr = *reinterpret_cast<const f16x8*>(sp + i * 1024);
                if constexpr (f < 2 * FT) {
                    constexpr int u = f / FT, ks = (f % FT) / PM;
#pragma unroll
                    for (int t = 0; t < TT; ++t) a1[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr, h2[t][ks], a1[u][t], 0, 0, 0);
                } else {
                    constexpr int m = (f - 2 * FT) / PM;
#pragma unroll
                    for (int t = 0; t < TT; ++t) xa[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr, hf[t], xa[m][t], 0, 0, 0);
                }
            });
            stage_end();
        });
    }
    wait_vm<0>();                                                  // (pieces requested past the end must land before the LDS is released)

    // ---- out = alpha_fc2 * acc: fp32 planes and / or their k-octet copy; lane parts of the address in the vector offset ----
    const int v16 = ((kq >> 1) * (int)g.ldy16 + px) * 16 + 8 * (kq & 1);
    const int v32 = (4 * kq * (int)g.ldy + px) * 4;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const int64_t img = (int64_t)z * TT + t;
        const __amdgpu_buffer_rsrc_t ry16 = __builtin_amdgcn_make_buffer_rsrc(
            g.Y16 ? reinterpret_cast<char*>(g.Y16) + img * g.strideY16 * 2 : nullptr, 0, g.Y16 ? (int)((int64_t)(kC / 8) * g.ldy16 * 16) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t ry32 = __builtin_amdgcn_make_buffer_rsrc(
            g.Y ? reinterpret_cast<char*>(g.Y) + img * g.strideY * 4 : nullptr, 0, g.Y ? (int)(((int64_t)(kC - 1) * g.ldy + g.N) * 4) : 0, 0x00020000);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const f32x4 v = g.alpha_fc2 * xa[m][t];
            if (g.Y) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ve = v[e];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ve), ry32, pin ? v32 : kOob, (16 * m + e) * (int)g.ldy * 4, 0);
                }
            }
            if (g.Y16) __builtin_amdgcn_raw_buffer_store_b64(pack4(v), ry16, pin ? v16 : kOob, 2 * m * (int)g.ldy16 * 16, 0);
        }
    }
}

template <int TT>
int launch_pm(const TbArgs& a, dim3 grid, hipStream_t st) {
    if (a.p.pm == 1) hipLaunchKernelGGL((temporal_block_kernel<TT, 1>), grid, dim3(kThreads), 0, st, a);
    else hipLaunchKernelGGL((temporal_block_kernel<TT, 2>), grid, dim3(kThreads), 0, st, a);
    return sf::check_launch("sf_temporal_block");
}

}  // namespace

extern "C" int sf_temporal_block_frags(int pm) { return (pm == 1 || pm == 2) ? 256 * pm : 0; }

extern "C" int sf_temporal_block(const SfTemporalBlock* p, void* stream) {
    SF_REQUIRE(p, "sf_temporal_block: NULL description");
    const SfTemporalBlock& g = *p;
    SF_REQUIRE(g.X16 && g.wstream && (g.Y || g.Y16) && g.ln1_w && g.ln1_b && g.ln2_w && g.ln2_b, "sf_temporal_block: NULL operand");
    SF_REQUIRE(g.N > 0 && g.B > 0 && g.TT >= 1, "sf_temporal_block: bad sizes");
    if (g.C != kC || g.H != kH || g.TT > 3 || (g.pm != 1 && g.pm != 2))
        return sf::fail(SF_ERR_UNSUPPORTED, "sf_temporal_block: built for C = 128, hidden 256, 1 .. 3 tokens per pixel, 1 or 2 products "
                                            "(got C %d, hidden %d, %d tokens, %d products)", g.C, g.H, g.TT, g.pm);
    SF_REQUIRE(g.wstream_bytes == (int64_t)sf_temporal_block_frags(g.pm) * 1024, "sf_temporal_block: weight stream size does not match sf_temporal_block_frags");
    SF_REQUIRE((reinterpret_cast<uintptr_t>(g.X16) & 15) == 0 && (reinterpret_cast<uintptr_t>(g.wstream) & 15) == 0 && (g.strideX & 7) == 0 &&
               g.ldx >= g.N, "sf_temporal_block: X16 / wstream must be 16-byte aligned k-octet planes with ldx >= N");
    SF_REQUIRE(!g.Y16 || ((reinterpret_cast<uintptr_t>(g.Y16) & 15) == 0 && (g.strideY16 & 7) == 0 && g.ldy16 >= g.N),
               "sf_temporal_block: Y16 must be 16-byte aligned k-octet planes with ldy16 >= N");
    SF_REQUIRE(!g.Y || ((reinterpret_cast<uintptr_t>(g.Y) & 3) == 0 && g.ldy >= g.N), "sf_temporal_block: ldy < N");
    // 32-bit buffer ranges: every span the kernel turns into a descriptor range stays under 2^30 (= the out-of-range offset)
    const int64_t lim = (int64_t)1 << 30;
    SF_REQUIRE((int64_t)(kC / 8) * g.ldx * 16 < lim && (!g.Y16 || (int64_t)(kC / 8) * g.ldy16 * 16 < lim) &&
               (!g.Y || ((int64_t)(kC - 1) * g.ldy + g.N) * 4 < lim), "sf_temporal_block: plane too large for 32-bit buffer offsets");
    TbArgs a{};
    a.p = g;
    a.ntile = (g.N + kPxWg - 1) / kPxWg;
    a.w_bytes = g.wstream_bytes;
    const int64_t nwg = (int64_t)a.ntile * g.B;
    SF_REQUIRE(nwg < (int64_t)1 << 31, "sf_temporal_block: grid too large");
    const dim3 grid((unsigned)nwg);
    hipStream_t st = (hipStream_t)stream;
    switch (g.TT) {
        case 1: return launch_pm<1>(a, grid, st);
        case 2: return launch_pm<2>(a, grid, st);
        default: return launch_pm<3>(a, grid, st);
    }
}
