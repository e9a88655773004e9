// All-pairs correlation volume + 4-level pyramid (one pass) and the radius-4 pyramid lookup.
//
// Reference: core/corr.py:7-21,46-54 (build), :23-44 (lookup), core/utils/utils.py:65-79 (sampler).
//
// BUILD.  C[i][j] = <f1[:,i], f2[:,j]> / sqrt(D) is a dense contraction, so it runs on the matrix
// cores (exact-fp32 v_mfma_f32_32x32x2_f32).  The GEMM "N" tile is not a run of 256 consecutive
// targets but an 8-row x 32-column PATCH of the target image: every 2x2, 4x4 and 8x8 pooling block of
// levels 1..3 then lies inside one wave's accumulators, so the three avg-pool levels are produced in
// the epilogue (vertical pairs = different accumulators, horizontal pairs = lane^1, lane^2, lane^4
// shuffles) and every pyramid cell is written exactly once and never re-read
// (algorithmic bytes: 2*N*D*4 feature reads + N*cells*4 writes).
// Stores are 128-byte runs (32 lanes x consecutive x) for level 0.
//
// LOOKUP.  HBM-bound gather.  A workgroup owns 64 consecutive source pixels of one pyramid level:
// it gathers their 10x10 bilinear footprints into LDS (lanes walk the 40-byte footprint rows), then
// emits the 81 taps per pixel with lanes running over PIXELS, so every output store is a 256-byte
// run inside one of the 324 channel planes (NCHW output, no transposing copy as in the reference).
#include "sf_common.h"
#include "split_operand.h"

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
};

// Epilogue shared by both arithmetic modes: scale, write level 0, pool levels 1..3 in registers.
// acc[t][r]: target patch row t (0..7), MFMA C/D register r -> source pixel (r&3)+8*(r>>2)+4*(lane>>5) of the
// wave's 32-pixel block, lane&31 = target patch column.
__device__ __forceinline__ void pyramid_epilogue(const BuildArgs& g, f32x16 (&acc)[PR], int b, int pair, int m0, int wave,
                                                 int py0, int px0, int lane) {
    const int khalf = lane >> 5, l31 = lane & 31;
    const int x = px0 + l31;
    const int64_t P0 = (int64_t)g.hl[0] * g.wl[0], P1 = (int64_t)g.hl[1] * g.wl[1];
    const int64_t P2 = (int64_t)g.hl[2] * g.wl[2], P3 = (int64_t)g.hl[3] * g.wl[3];
    float* const L0 = g.lvl[0] + pair * g.lvl_pair_stride[0];
    float* const L1 = g.lvl[1] + pair * g.lvl_pair_stride[1];
    float* const L2 = g.lvl[2] + pair * g.lvl_pair_stride[2];
    float* const L3 = g.lvl[3] + pair * g.lvl_pair_stride[3];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;     // source pixel (wave-half uniform)
        const bool iok = i < g.N;
        const int64_t row = (int64_t)b * g.N + i;
        float v0[PR];
#pragma unroll
        for (int t = 0; t < PR; ++t) v0[t] = acc[t][r] * g.scale;
        if (iok && x < g.wl[0]) {
#pragma unroll
            for (int t = 0; t < PR; ++t)
                if (py0 + t < g.hl[0]) L0[row * P0 + (int64_t)(py0 + t) * g.wl[0] + x] = v0[t];
        }
        float v1[PR / 2];
#pragma unroll
        for (int t = 0; t < PR / 2; ++t) {
            float s = v0[2 * t] + v0[2 * t + 1];
            s += __shfl_xor(s, 1);
            v1[t] = 0.25f * s;
        }
        if (iok && (l31 & 1) == 0 && (x >> 1) < g.wl[1]) {
#pragma unroll
            for (int t = 0; t < PR / 2; ++t)
                if ((py0 >> 1) + t < g.hl[1]) L1[row * P1 + (int64_t)((py0 >> 1) + t) * g.wl[1] + (x >> 1)] = v1[t];
        }
        float v2[PR / 4];
#pragma unroll
        for (int t = 0; t < PR / 4; ++t) {
            float s = v1[2 * t] + v1[2 * t + 1];
            s += __shfl_xor(s, 2);
            v2[t] = 0.25f * s;
        }
        if (iok && (l31 & 3) == 0 && (x >> 2) < g.wl[2]) {
#pragma unroll
            for (int t = 0; t < PR / 4; ++t)
                if ((py0 >> 2) + t < g.hl[2]) L2[row * P2 + (int64_t)((py0 >> 2) + t) * g.wl[2] + (x >> 2)] = v2[t];
        }
        float s3 = v2[0] + v2[1];
        s3 += __shfl_xor(s3, 4);
        s3 *= 0.25f;
        if (iok && (l31 & 7) == 0 && (x >> 3) < g.wl[3] && (py0 >> 3) < g.hl[3])
            L3[row * P3 + (int64_t)(py0 >> 3) * g.wl[3] + (x >> 3)] = s3;
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

// ---- split-precision (f16x3) build: same tiling, operands staged as (hi, lo) f16 (see gemm_split.hip) -----
// A = f1 [D][N] (rows of k, source pixels contiguous), B = f2 gathered as an 8x32 target patch.  Both are fp32
// K-major sources split on the fly; rows k >= D are zeroed, source pixels / patch cells outside the image are
// clamped (their products are never stored).  3 MFMAs of 32x32x16 per product instead of 8 of 32x32x2: the
// kernel goes from fp32-MFMA-bound to (nearly) bound by the single write of the pyramid.
__global__ __launch_bounds__(kThreads, 2) void corr_build_split_kernel(const BuildArgs g) {
    using namespace sf_split;
    __shared__ __attribute__((aligned(16))) _Float16 sA[2][BM * LDK];
    __shared__ __attribute__((aligned(16))) _Float16 sB[2][BN * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int khalf = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z / g.pairs, pair = blockIdx.z % g.pairs;
    const int m0 = blockIdx.y * BM;
    const int py0 = (blockIdx.x / g.pcols) * PR, px0 = (blockIdx.x % g.pcols) * PC;
    const float* A = g.f1 + (int64_t)b * g.f_clip_stride + (int64_t)pair * g.f_pair_stride;
    const float* Bm = g.f2 + (int64_t)b * g.f_clip_stride + (int64_t)pair * g.f_pair_stride;
    const int64_t bytes = (int64_t)g.D * g.N * 4;

    Operand<BM, SF_LAYOUT_K_MAJOR> opa;
    Operand<BN, SF_LAYOUT_K_MAJOR> opb;
    typename Operand<BM, SF_LAYOUT_K_MAJOR>::Regs ra;
    typename Operand<BN, SF_LAYOUT_K_MAJOR>::Regs rb;
    opa.init(A, nullptr, bytes, g.N, g.D, g.N, m0, 0, 0, tid);
    opb.init(Bm, nullptr, bytes, g.N, g.D, g.N, 0, 0, 0, tid);
#pragma unroll
    for (int j = 0; j < Operand<BN, SF_LAYOUT_K_MAJOR>::NI; ++j) {      // column n of the B tile = cell (n/32, n%32) of the patch
        const int n = (tid + j * kThreads) % BN;
        const int y = min(py0 + n / PC, g.h - 1), x = min(px0 + n % PC, g.w - 1);
        opb.voff[j] = (y * g.w + x) * 4;
    }

    f32x16 acc[PR];
#pragma unroll
    for (int t = 0; t < PR; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nk = (g.D + BK - 1) / BK;
    opa.load(0, 0, ra);
    opb.load(0, 0, rb);
    opa.store(0, sA[0], sA[1], ra);
    opb.store(0, sB[0], sB[1], rb);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) {
            opa.load((kt + 1) * BK, (kt + 1) * BK * g.N, ra);
            opb.load((kt + 1) * BK, (kt + 1) * BK * g.N, rb);
        }
        __builtin_amdgcn_sched_barrier(0);
        const _Float16* pah = sA[0] + (wave * 32 + l31) * LDK + khalf * 8;
        const _Float16* pal = sA[1] + (wave * 32 + l31) * LDK + khalf * 8;
        const _Float16* pbh = sB[0] + l31 * LDK + khalf * 8;
        const _Float16* pbl = sB[1] + l31 * LDK + khalf * 8;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(pah + ks * 16);
            const f16x8 al = *reinterpret_cast<const f16x8*>(pal + ks * 16);
#pragma unroll
            for (int t = 0; t < PR; ++t) {
                const f16x8 bh = *reinterpret_cast<const f16x8*>(pbh + t * PC * LDK + ks * 16);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(pbl + t * PC * LDK + ks * 16);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) {
            __syncthreads();
            opa.store((kt + 1) * BK, sA[0], sA[1], ra);
            opb.store((kt + 1) * BK, sB[0], sB[1], rb);
            __syncthreads();
        }
    }
    pyramid_epilogue(g, acc, b, pair, m0, wave, py0, px0, lane);
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
};

__global__ __launch_bounds__(kThreads) void corr_lookup_kernel(const LookupArgs g) {
    __shared__ float win[LP * WSTRIDE];
    __shared__ float sfx[LP], sfy[LP];
    __shared__ int sx0[LP], sy0[LP];
    const int tid = threadIdx.x;
    const int l = blockIdx.y, img = blockIdx.z;
    const int b = img / g.pairs, pair = img % g.pairs;
    const int p0 = blockIdx.x * LP;
    const int hl = g.hl[l], wl = g.wl[l];
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
    const float* vol = g.lvl[l] + pair * g.lvl_pair_stride[l] + ((int64_t)b * g.N + p0) * hl * wl;
    for (int idx = tid; idx < LP * FP * FP; idx += kThreads) {
        const int pix = idx / (FP * FP), cell = idx % (FP * FP);
        const int yy = sy0[pix] - RAD + cell / FP, xx = sx0[pix] - RAD + cell % FP;
        float v = 0.f;
        if (p0 + pix < g.N && yy >= 0 && yy < hl && xx >= 0 && xx < wl)
            v = vol[(int64_t)pix * hl * wl + yy * wl + xx];
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

extern "C" int sf_corr_build_pyramid(const float* f1, const float* f2, int64_t f_clip_stride, int64_t f_pair_stride,
                                     float* lvl0, float* lvl1, float* lvl2, float* lvl3,
                                     const int64_t* lvl_pair_stride, int B, int pairs, int D, int h, int w,
                                     int num_levels, int precision, void* stream) {
    SF_REQUIRE(f1 && f2 && lvl0 && lvl1 && lvl2 && lvl3, "sf_corr_build_pyramid: null pointer");
    SF_REQUIRE(B > 0 && pairs > 0 && D > 0 && h > 0 && w > 0, "sf_corr_build_pyramid: bad dims");
    SF_REQUIRE(pairs == 1 || lvl_pair_stride, "sf_corr_build_pyramid: pairs > 1 needs lvl_pair_stride");
    SF_REQUIRE(num_levels == 4, "sf_corr_build_pyramid: num_levels must be 4 (got %d)", num_levels);
    SF_REQUIRE(precision == SF_PRECISION_FP32 || precision == SF_PRECISION_F16X3,
               "sf_corr_build_pyramid: precision %d not supported", precision);
    SF_REQUIRE(precision == SF_PRECISION_FP32 || (int64_t)D * h * w * 4 < ((int64_t)1 << 31),
               "sf_corr_build_pyramid: feature image larger than 2 GiB");
    SF_REQUIRE((h >> 3) >= 1 && (w >> 3) >= 1, "sf_corr_build_pyramid: feature grid %dx%d too small for 4 levels", h, w);
    SF_REQUIRE((int64_t)B * pairs <= 65535, "sf_corr_build_pyramid: B*pairs too large");
    BuildArgs g;
    g.f1 = f1; g.f2 = f2;
    g.lvl[0] = lvl0; g.lvl[1] = lvl1; g.lvl[2] = lvl2; g.lvl[3] = lvl3;
    g.f_clip_stride = f_clip_stride; g.f_pair_stride = f_pair_stride;
    g.B = B; g.pairs = pairs; g.D = D; g.h = h; g.w = w; g.N = h * w;
    for (int l = 0; l < 4; ++l) {
        g.hl[l] = h >> l; g.wl[l] = w >> l;
        g.lvl_pair_stride[l] = (pairs > 1) ? lvl_pair_stride[l] : 0;
    }
    g.pcols = sf::ceil_div(w, PC);
    g.scale = 1.0f / sqrtf((float)D);
    g.vec_a = ((g.N & 3) == 0) && ((f_clip_stride & 3) == 0) && ((f_pair_stride & 3) == 0) &&
              ((reinterpret_cast<uintptr_t>(f1) & 15) == 0);
    g.vec_b = ((w & 3) == 0) && ((f_clip_stride & 3) == 0) && ((f_pair_stride & 3) == 0) &&
              ((reinterpret_cast<uintptr_t>(f2) & 15) == 0);
    dim3 grid(g.pcols * sf::ceil_div(h, PR), sf::ceil_div(g.N, BM), B * pairs);
    if (precision == SF_PRECISION_F16X3)
        hipLaunchKernelGGL(corr_build_split_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, g);
    else
        hipLaunchKernelGGL(corr_build_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, g);
    return sf::check_launch("sf_corr_build_pyramid");
}

extern "C" int sf_corr_lookup(const float* lvl0, const float* lvl1, const float* lvl2, const float* lvl3,
                              const int64_t* lvl_pair_stride, const float* coords, float* out,
                              int64_t out_img_stride, int B, int pairs, int h, int w, int num_levels, int radius,
                              void* stream) {
    SF_REQUIRE(lvl0 && lvl1 && lvl2 && lvl3 && coords && out, "sf_corr_lookup: null pointer");
    SF_REQUIRE(B > 0 && pairs > 0 && h > 0 && w > 0, "sf_corr_lookup: bad dims");
    SF_REQUIRE(pairs == 1 || lvl_pair_stride, "sf_corr_lookup: pairs > 1 needs lvl_pair_stride");
    SF_REQUIRE(num_levels == 4 && radius == RAD, "sf_corr_lookup: only num_levels=4, radius=4 (got %d, %d)",
               num_levels, radius);
    SF_REQUIRE((int64_t)B * pairs <= 65535, "sf_corr_lookup: B*pairs too large");
    LookupArgs g;
    g.lvl[0] = lvl0; g.lvl[1] = lvl1; g.lvl[2] = lvl2; g.lvl[3] = lvl3;
    g.coords = coords; g.out = out; g.out_img_stride = out_img_stride;
    g.B = B; g.pairs = pairs; g.h = h; g.w = w; g.N = h * w;
    for (int l = 0; l < 4; ++l) {
        g.hl[l] = h >> l; g.wl[l] = w >> l;
        g.lvl_pair_stride[l] = (pairs > 1) ? lvl_pair_stride[l] : 0;
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
