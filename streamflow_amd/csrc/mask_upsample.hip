// Mask head, second layer + convex upsampling in ONE launch.
//
// Reference: core/update.py:756-759,777 (mask = 0.25 * Conv1x1(256 -> 576)(relu(Conv3x3(net)))) and core/models/streamflow.py:82-93
// (upsample_flow: softmax over the 9 logits of every 8 x 8 sub-pixel, convex combination of the 3 x 3 neighbours of 8 * flow).  The
// unfused form writes the 576-channel mask (2.3 KB per pixel in fp32) and reads it back in sf_upsample_flow.  Per pixel the chain is
// independent, and in the accumulator layout of 16-row MFMA tiles the nine logits of a sub-pixel are ONE lane's registers: mask row
// 64 k + s (k = neighbour, s = 8 i + j the sub-pixel) lives in tile 4 k + s / 16 at (kq, register) = ((s % 16) / 4, s % 4), i.e.
// the same lane and register for every k.  So: a wave owns 16 pixels, accumulates all 36 row tiles (144 registers), and its epilogue
// is the softmax + combination + the 16-byte stores of four horizontally adjacent sub-pixels per lane (a wave's 16 pixels of a row
// make 512-byte runs).  Weights stream as in csrc/ffn_pair.hip (one fragment stream through a 3-stage LDS ring).
// Input: the k-octet fp16 copy of relu(mask.0(net)) (sf_gemm's c_f16 = 3 output); arithmetic: activations fp16, weights fp16 (pm = 1)
// or hi + lo (pm = 2), fp32 accumulation and softmax.
#include "sf_common.h"

#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;

constexpr int kK = 256, kM = 576, kTiles = kM / 16, kKs = kK / 32;
constexpr int kWaves = 4, kThreads = 256, kPxWave = 16, kPxWg = kWaves * kPxWave;
constexpr int S = 16, kStage = S * 1024, RING = 3, PCS = S / kWaves;
constexpr int kOob = 1 << 30;

struct MuArgs {
    SfMaskUpsample p;
    int ntile;
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

template <int PM>
__global__ __launch_bounds__(kThreads, 2) void mask_upsample_kernel(const MuArgs a) {
    const SfMaskUpsample& g = a.p;
    constexpr int FT = kKs * PM;                                  // fragments of one row tile
    constexpr int TPS = S / FT;                                   // row tiles per stage (2 or 1)
    static_assert(S % FT == 0 && kTiles % TPS == 0, "a stage holds whole row tiles");
    __shared__ __attribute__((aligned(1024))) char smem[RING * kStage + kM * 4];
    float* sbias = reinterpret_cast<float*>(smem + RING * kStage);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, l15 = lane & 15;
    const int tile = blockIdx.x % a.ntile, img = blockIdx.x / a.ntile;
    const int P = g.h * g.w;
    const int px = tile * kPxWg + wave * kPxWave + l15;
    const bool pin = px < P;

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
    for (int i = tid; i < kM; i += kThreads) sbias[i] = g.bias ? g.bias[i] : 0.f;

    // ---- operand: the 256 channels of this lane's pixel (k-octet 4 s + kq of k-step s) ----
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(g.X16)) + (int64_t)img * g.strideX * 2, 0, (int)((int64_t)(kK / 8) * g.ldx * 16), 0x00020000);
    f16x8 b[kKs];
#pragma unroll
    for (int s = 0; s < kKs; ++s)
        b[s] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rx, pin ? ((4 * s + kq) * (int)g.ldx + px) * 16 : kOob, 0, 0));
    // ---- 8 * flow at the 3 x 3 neighbours of the pixel (zero outside the grid: F.unfold's padding) ----
    const int y = pin ? px / g.w : 0, x = pin ? px - (px / g.w) * g.w : 0;
    const float* fl = g.flow + (int64_t)img * 2 * P;
    float nb[2][9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        const bool ok = pin && yy >= 0 && yy < g.h && xx >= 0 && xx < g.w;
        const int o = ok ? yy * g.w + xx : 0;
        nb[0][k] = ok ? 8.0f * fl[o] : 0.f;
        nb[1][k] = ok ? 8.0f * fl[P + o] : 0.f;
    }
    wait_vm<0>();
    __syncthreads();

    f32x4 acc[kTiles];
#pragma unroll
    for (int t = 0; t < kTiles; ++t) acc[t] = *reinterpret_cast<const f32x4*>(sbias + 16 * t + 4 * kq);

    static_for<0, kTiles / TPS>([&](auto j_tag) {
        constexpr int j = decltype(j_tag)::value, slot = j % RING;
        issue_stage(j + RING - 1, (slot + RING - 1) % RING);      // (past the end: the last stage again)
        const char* sp = smem + slot * kStage + lane * 16;
        static_for<0, S>([&](auto i_tag) {
            constexpr int i = decltype(i_tag)::value, t = j * TPS + i / FT, ks = (i % FT) / PM;
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(sp + i * 1024), b[ks], acc[t], 0, 0, 0);
        });
        // every fragment read of the stage has EXECUTED before the barrier (ffn_pair.hip: the refill race)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait_vm<PCS * (RING - 2)>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    });
    wait_vm<0>();

    // ---- softmax over the nine neighbours per sub-pixel, convex combination, 16-byte stores ----
    // sub-pixel s = 16 gq + 4 kq + e  ->  (i, j) = (s >> 3, s & 7) = (2 gq + (kq >> 1), 4 (kq & 1) + e): out[c][8 y + i][8 x + j]
    const int W8 = 8 * g.w, H8 = 8 * g.h;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(g.out) + (int64_t)img * 2 * H8 * W8 * 4, 0,
                                                                         2 * H8 * W8 * 4, 0x00020000);
    const int vo = ((8 * y + (kq >> 1)) * W8 + 8 * x + 4 * (kq & 1)) * 4;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        f32x4 ox = {0.f, 0.f, 0.f, 0.f}, oy = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float z[9], m = -INFINITY;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                z[k] = g.alpha * acc[4 * k + gq][e];
                m = fmaxf(m, z[k]);
            }
            float sum = 0.f, ax = 0.f, ay = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float ex = expf(z[k] - m);
                sum += ex;
                ax = fmaf(ex, nb[0][k], ax);
                ay = fmaf(ex, nb[1][k], ay);
            }
            const float inv = 1.0f / sum;
            ox[e] = ax * inv;
            oy[e] = ay * inv;
        }
        // (8-byte stores: with 16-byte stores a few cells of the LAST workgroups came out as values of the next sub-pixel group, run to
        // run -- the store's data registers rewritten by the following VALU work before the store had read them)
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            u32x2 dx, dy;
            const float x0 = ox[2 * hlf], x1 = ox[2 * hlf + 1], y0 = oy[2 * hlf], y1 = oy[2 * hlf + 1];
            dx[0] = __builtin_bit_cast(unsigned, x0); dx[1] = __builtin_bit_cast(unsigned, x1);
            dy[0] = __builtin_bit_cast(unsigned, y0); dy[1] = __builtin_bit_cast(unsigned, y1);
            __builtin_amdgcn_raw_buffer_store_b64(dx, ro, pin ? vo + 8 * hlf : kOob, (2 * gq) * W8 * 4, 0);
            __builtin_amdgcn_raw_buffer_store_b64(dy, ro, pin ? vo + 8 * hlf : kOob, (H8 + 2 * gq) * W8 * 4, 0);
        }
    }
}

}  // namespace

extern "C" int sf_mask_upsample_frags(int pm) { return (pm == 1 || pm == 2) ? kTiles * kKs * pm : 0; }

extern "C" int sf_mask_upsample(const SfMaskUpsample* p, void* stream) {
    SF_REQUIRE(p, "sf_mask_upsample: NULL description");
    const SfMaskUpsample& g = *p;
    SF_REQUIRE(g.X16 && g.wstream && g.flow && g.out, "sf_mask_upsample: NULL operand");
    SF_REQUIRE(g.n_img > 0 && g.h > 0 && g.w > 0, "sf_mask_upsample: bad sizes");
    if (g.K != kK || g.M != kM || (g.pm != 1 && g.pm != 2))
        return sf::fail(SF_ERR_UNSUPPORTED, "sf_mask_upsample: built for 256 -> 576 (9 x 8 x 8) with 1 or 2 products (got %d -> %d, %d)", g.K, g.M, g.pm);
    const int64_t P = (int64_t)g.h * g.w;
    SF_REQUIRE(g.wstream_bytes == (int64_t)sf_mask_upsample_frags(g.pm) * 1024, "sf_mask_upsample: weight stream size does not match sf_mask_upsample_frags");
    SF_REQUIRE((reinterpret_cast<uintptr_t>(g.X16) & 15) == 0 && (reinterpret_cast<uintptr_t>(g.wstream) & 15) == 0 && (g.strideX & 7) == 0 &&
               g.ldx >= P, "sf_mask_upsample: X16 / wstream must be 16-byte aligned k-octet planes with ldx >= h * w");
    SF_REQUIRE((reinterpret_cast<uintptr_t>(g.out) & 15) == 0, "sf_mask_upsample: out must be 16-byte aligned");
    const int64_t lim = (int64_t)1 << 30;
    SF_REQUIRE((int64_t)(kK / 8) * g.ldx * 16 < lim && 2 * 64 * P * 4 < lim, "sf_mask_upsample: image too large for 32-bit buffer offsets");
    MuArgs a{};
    a.p = g;
    a.ntile = (int)((P + kPxWg - 1) / kPxWg);
    a.w_bytes = g.wstream_bytes;
    const int64_t nwg = (int64_t)a.ntile * g.n_img;
    SF_REQUIRE(nwg < (int64_t)1 << 31, "sf_mask_upsample: grid too large");
    if (g.pm == 1) hipLaunchKernelGGL((mask_upsample_kernel<1>), dim3((unsigned)nwg), dim3(kThreads), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((mask_upsample_kernel<2>), dim3((unsigned)nwg), dim3(kThreads), 0, (hipStream_t)stream, a);
    return sf::check_launch("sf_mask_upsample");
}
