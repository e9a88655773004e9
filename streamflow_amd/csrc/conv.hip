// Depthwise KxK convolution fused with bias, residual and exact GELU:
//     y = gelu(x + dwconv_KxK(x) + b)            (reference core/update.py:33-34, kernels 15 and 7)
//
// Two kernels.  (1) fp32 stencil on the VALU, below (exact fp32 mode, and K = 7 in the f16x3 mode).  (2) Banded-Toeplitz
// GEMMs on the matrix cores, dwconv_mfma_kernel further down: K = 15 in every split precision (three products in f16x3,
// two in f16x2 / f16), K = 7 in the two-product modes.  Either kernel can write y as fp16 rows (y_f16).
//
// (1) VALU-bound stencil (225 or 49 FMAs per output).  One workgroup owns one (image, channel) plane
// strip: the strip plus its K/2 halo is staged once in LDS (zero padded, so the inner loop has no
// bounds checks), each thread produces a 4x4 output tile from a sliding window held in registers
// (one ds_read_b128 row segment feeds 4 output columns x K taps).  The K*K weights of the channel sit in LDS
// behind the tile and are read one kernel row at a time with broadcast reads into VGPRs.  (Measured: fetching
// them through the scalar cache as SGPR operands of v_fmac_f32 was 18 % slower -- every weight row costs an
// lgkmcnt(0) drain -- and a register sliding window that cuts the LDS row reads 4x changed nothing: the
// kernel is bound by VALU issue, 47 TFLOP/s of fp32 FMA.)
#include "sf_common.h"
#include <cstdlib>
#include "split_operand.h"

namespace {

constexpr int TX = 4, TY = 4;
constexpr int kMaxThreads = 512;

struct DwArgs {
    const float* x; const float* wgt; const float* bias; float* y;
    int64_t x_img_stride, y_img_stride;
    int C, h, w;
    int strip_h;      // output rows per strip (multiple of TY)
    int tiles_x;      // ceil(w / TX)
    int wp4;          // LDS row stride in float4 units
    int vec_store;    // rows of y are 16-byte (fp32) / 8-byte (fp16) aligned: one store per 4 outputs
};

template <int KS, bool kOutF16>
__global__ __launch_bounds__(kMaxThreads) void dwconv_res_gelu_kernel(const DwArgs g) {
    constexpr int R = KS / 2;
    constexpr int IN_W = TX + KS - 1;               // window columns a thread needs (18 / 10)
    constexpr int IN_V = (IN_W + 3) / 4;            // as float4
    constexpr int KY_UNROLL = (KS % 3 == 0) ? 3 : 1;
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int plane = blockIdx.x;
    const int c = plane % g.C, img = plane / g.C;
    const int ys = blockIdx.y * g.strip_h;
    const int rows = g.strip_h + KS - 1;
    const float* __restrict__ xp = g.x + img * g.x_img_stride + (int64_t)c * g.h * g.w;
    const float* __restrict__ wc = g.wgt + (int64_t)c * KS * KS;

    const int wp = g.wp4 * 4;
    for (int idx = threadIdx.x; idx < rows * wp; idx += blockDim.x) {
        const int ry = idx / wp, cx = idx - ry * wp;
        const int gy = ys - R + ry, gx = cx - R;
        float v = 0.f;
        if (gy >= 0 && gy < g.h && gx >= 0 && gx < g.w) v = xp[gy * g.w + gx];
        tile[idx] = v;
    }
    // the channel's K*K weights go to LDS behind the tile
    float* wl = tile + rows * wp + 8;
    for (int i = threadIdx.x; i < KS * KS; i += blockDim.x) wl[i] = wc[i];
    __syncthreads();

    const int tiles_y = g.strip_h / TY;
    const int tx = threadIdx.x % g.tiles_x, ty = threadIdx.x / g.tiles_x;
    if (ty >= tiles_y) return;

    float acc[TY][TX];
#pragma unroll
    for (int i = 0; i < TY; ++i)
#pragma unroll
        for (int j = 0; j < TX; ++j) acc[i][j] = 0.f;

    const float4* base = reinterpret_cast<const float4*>(tile) + (ty * TY) * g.wp4 + tx;
    // ky is a rolled loop on purpose: only KY_UNROLL weight rows are live at a time.  Each (ky, oy) pair re-reads
    // its window row from LDS (4x redundant ds_read_b128; measured not to matter, see the header).
#pragma unroll KY_UNROLL
    for (int ky = 0; ky < KS; ++ky) {
        float wrow[KS];
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) wrow[kx] = wl[ky * KS + kx];          // uniform address: LDS broadcast
#pragma unroll
        for (int oy = 0; oy < TY; ++oy) {
            float in[IN_V * 4];
#pragma unroll
            for (int v = 0; v < IN_V; ++v) {
                const float4 q = base[(ky + oy) * g.wp4 + v];
                in[v * 4 + 0] = q.x; in[v * 4 + 1] = q.y; in[v * 4 + 2] = q.z; in[v * 4 + 3] = q.w;
            }
#pragma unroll
            for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                for (int ox = 0; ox < TX; ++ox) acc[oy][ox] = fmaf(in[ox + kx], wrow[kx], acc[oy][ox]);
        }
    }

    const float bv = g.bias[c];
    // kOutF16: y holds IEEE fp16 planes (strides in halves) -- the fp16 hand-over to the pw GEMM (streamflow_hip.h)
    float* yp = g.y + img * g.y_img_stride + (int64_t)c * g.h * g.w;
    _Float16* yh = reinterpret_cast<_Float16*>(g.y) + img * g.y_img_stride + (int64_t)c * g.h * g.w;
#pragma unroll
    for (int oy = 0; oy < TY; ++oy) {
        const int gy = ys + ty * TY + oy;
        if (gy >= g.h) continue;
        const float* ctr = tile + (ty * TY + oy + R) * wp + tx * TX + R;
        float o[TX];
        static_assert(TX % 2 == 0, "outputs are activated in pairs (packed fp32 GELU)");
#pragma unroll
        for (int ox = 0; ox < TX; ox += 2) {
            sf::f32x2 t;
            t[0] = ctr[ox] + (acc[oy][ox] + bv);
            t[1] = ctr[ox + 1] + (acc[oy][ox + 1] + bv);
            const sf::f32x2 a = sf::gelu_erf2(t);
            o[ox] = a[0];
            o[ox + 1] = a[1];
        }
        const int gx = tx * TX;
        if constexpr (kOutF16) {
            if (g.vec_store && gx + 3 < g.w) {
                typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                *reinterpret_cast<f16x4*>(yh + gy * g.w + gx) =
                    f16x4{(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
            } else {
#pragma unroll
                for (int ox = 0; ox < TX; ++ox)
                    if (gx + ox < g.w) yh[gy * g.w + gx + ox] = (_Float16)o[ox];
            }
        } else if (g.vec_store && gx + 3 < g.w) {
            *reinterpret_cast<float4*>(yp + gy * g.w + gx) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
            for (int ox = 0; ox < TX; ++ox)
                if (gx + ox < g.w) yp[gy * g.w + gx + ox] = o[ox];
        }
    }
}


// ------------------------------------------------------------------------------------------------------------
// The same convolution on the matrix cores (split precisions).
//
// For one kernel row ky, a run of 16 outputs along x is a small GEMM over the 32 input columns that can reach them:
//     out[y][x0 + n] += sum_k  in[y + ky - R][x0 - 8 + k] * T_ky[k][n],     T_ky[k][n] = w[ky][k - n - 8 + R]  (0 outside)
// i.e. A = a 16-row x 32-column window of the (zero padded) input plane, B = a banded Toeplitz matrix that depends
// only on (channel, ky).  One v_mfma_f32_16x16x32_f16 does the 16x16 outputs of one ky; with x = hi + lo fp16 halves of
// both operands (products al*bh + ah*bl + ah*bh, fp32 accumulation: the GEMMs' split precision) that is 3 MFMAs per
// ky and 16x16 tile = 48 cycles against 15 x 256 FMAs / 32 lanes = 120 cycles on the VALU, and the matrix cores run
// closer to their peak than the stencil did to the VALU's (47-58 TFLOP/s = ~40 %).  K = 15: 15/32 of the multiplied
// entries are structural zeros -- the price of having no fp32 FMA faster than 1 per lane per 2 cycles.
//  * the plane strip is split ONCE while it is staged (hi and lo fp16 planes in LDS, 8-column octets = one A
//    fragment per lane, row stride = 2 mod 4 sixteen-byte units: conflict-free ds_read_b128 for the 16 rows of a
//    fragment);
//  * the 2 x KS Toeplitz fragments of the channel live in REGISTERS for the whole workgroup (8 VGPRs per ky), built
//    once from a zero-padded copy of the channel's weights; a workgroup walks several images of the same channel;
//  * a wave works on 2 horizontally adjacent tiles at a time (consecutive MFMAs alternate accumulators) and requests
//    the A fragments of kernel row ky+1 before the MFMAs of row ky.
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct DwmArgs {
    const float* x; const float* wgt; const float* bias; float* y;
    int64_t x_img_stride, y_img_stride;
    int n_img, C, h, w;
    int imgs_per_wg;
    int strip_h;      // output rows per strip (multiple of 16)
    int w16;          // w rounded up to 16
    int stride16;     // LDS row stride in 16-byte units (= 8 halfs)
    int vec_ok;       // rows are 16-byte aligned: float4 staging loads
    int plane_bytes;  // bytes of one staged LDS plane (kIn = 2: rounded up to whole 4-KB DMA pieces)
#ifdef SF_DW_TIMERS
    long long* ts;    // SF_DW_TS_BUF: per-workgroup phase cycles (tools/dwconv_one.py)
#endif
};

#ifndef SF_DW_FOLD_SINGLE
#define SF_DW_FOLD_SINGLE 0
#endif
#ifndef SF_DW_FOLD
#define SF_DW_FOLD 1
#endif
#ifndef SF_DW_TG
#define SF_DW_TG 4      // phase timers (tools/dwconv_one.py, -DSF_DW_TIMERS): with 2 tiles per wave the tile loop ran at 5.9k
                        // cycles per pair against 1.4k of MFMA issue -- every kernel row waited for its ds_read_b128 round
                        // trip behind 6 MFMAs; 4 tiles put 12 MFMAs behind each round trip
#endif
#ifndef SF_DW_FU
#define SF_DW_FU 6      // octets a thread stages at a time: 6 x 256 >= the 1404 octets of a 55x128 strip, ONE global round trip
#endif
// kProd = 3: x = hi + lo on both sides (f16x3).  kProd = 2 (the GEMMs' f16x2 arithmetic): the ACTIVATION enters the
// products as its hi half only (weights stay hi + lo) -- two MFMAs and one fragment read per kernel row and tile instead
// of three and two; both planes are still staged, the residual stays exact.  kProd = 1 (SF_PRECISION_F16, a single-product
// layer of the mixed preset): the WEIGHTS are one round-to-nearest fp16 too -- one MFMA per kernel row and tile, half the
// Toeplitz registers (60), three workgroups per CU.
#ifndef SF_GEMM_FAST_GELU
#define SF_GEMM_FAST_GELU 1
#endif
#ifndef SF_DW_MINWG
#define SF_DW_MINWG 2
#endif
#ifndef SF_DW_DMA_WGS
#define SF_DW_DMA_WGS 768      // workgroups aimed at by the double-buffered fp16-input form (tools/build_variant.sh to A/B)
#endif
// kIn: format of x.  0: fp32 planes (staged through registers and split into hi + lo fp16 planes).  1 / 2: fp16 ROWS (the
// config-2 hand-over of x2: the value IS its hi half, lo = 0, so conv input and residual are exact in fp16).  1 stages
// through registers like 0 (any width); 2 -- rows of whole octets, 16-byte aligned -- moves the strip HBM/L2 -> LDS with
// buffer_load ... lds, no registers, no ds_write, and DOUBLE-BUFFERED: the strip of image i + 1 lands in the second plane
// (the `lo` plane of the other forms) while the tiles of image i are computed, one barrier per image.
template <int KS, bool kOutF16, int kProd, int kIn = 0>
__global__ __launch_bounds__(256, kProd == 1 ? 3 : ((KS == 15 && kProd == 2) ? SF_DW_MINWG : 2)) void dwconv_mfma_kernel(const DwmArgs g) {
    using namespace sf_split;
    static_assert(kIn == 0 || kProd <= 2, "fp16 input: two- and one-product forms only");
    // Residual folded into the weights (round 5): with fp16 input the value x IS the A operand, so x + dwconv(x) = dwconv'(x) with
    // w'[centre] = w[centre] + 1 -- exact to the split's 2^-22 with hi + lo weights (kProd = 2).  It removes the epilogue's per-output
    // 2-byte LDS reads of x: 16 per tile group against 60 (15 x 15) / 28 (7 x 7) fragment reads, in a kernel that is LDS-bound, and
    // they were the ones with bank conflicts (SQ_LDS_BANK_CONFLICT 9 % / 17 % of the LDS cycles: rows 4 apart are 288 dwords apart).
    // Single-product layers (kProd = 1) would round w + 1 to fp16 (2^-12 |x| on the residual): SF_DW_FOLD_SINGLE, off.
    constexpr bool kFoldRes = SF_DW_FOLD && (kIn == 2) && (kProd == 2 || (kProd == 1 && SF_DW_FOLD_SINGLE));
    typedef __attribute__((address_space(3))) void* lds_ptr;
    constexpr int R = KS / 2;
    constexpr int WZ = 64;                                      // zero-padded weight row: w[ky][j - 24]
    constexpr int TG = SF_DW_TG;                                // adjacent column tiles a wave works on together
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kg = lane >> 4;
    const int c = blockIdx.x, ys = blockIdx.y * g.strip_h;
    const int rows_in = g.strip_h + KS - 1;
    const int noct = (g.w16 + 16) / 8;                          // octets per staged row: columns -8 .. w16 + 7
    const int plane_bytes = g.plane_bytes;
    char* hi = lds;
    char* lo = lds + plane_bytes;
    float* wz = reinterpret_cast<float*>(lds + 2 * plane_bytes);
    const int img0 = blockIdx.z * g.imgs_per_wg;
    const int img_end = min(img0 + g.imgs_per_wg, g.n_img);

    // ---- kIn = 2: the DMA image of a strip.  LDS slot s (16 bytes) = row s / stride16, octet s % stride16 of the staged
    // strip; a slot outside the plane (halo rows / columns, pad octets) gets an out-of-range offset and reads as zero.
    // The offsets do not depend on the image: computed once.
    constexpr int kMaxPieces = 8;                               // 8 x 256 slots x 16 B = 32 KB >= any plane the host allows
    constexpr int kOobOff = 1 << 30;
    int vo[kMaxPieces];
    const int npieces = plane_bytes >> 12;
    if constexpr (kIn == 2) {
#pragma unroll
        for (int p = 0; p < kMaxPieces; ++p) {
            const int sl = p * 256 + tid;
            const int row = sl / g.stride16, co = sl - row * g.stride16;
            const int gy = ys - R + row, gx = co * 8 - 8;
            const bool ok = row < rows_in && co < noct && gy >= 0 && gy < g.h && gx >= 0 && gx + 8 <= g.w;
            vo[p] = ok ? (gy * g.w + gx) * 2 : kOobOff;
        }
    }
    auto issue_strip = [&](int img, char* buf) {                // one resource per (image, channel) plane: the range check zeroes the halo
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<_Float16*>(reinterpret_cast<const _Float16*>(g.x) + img * g.x_img_stride + (int64_t)c * g.h * g.w), 0,
            g.h * g.w * 2, 0x00020000);
#pragma unroll
        for (int p = 0; p < kMaxPieces; ++p)
            if (p < npieces)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(buf + (p * 256 + wave * 64) * 16), 16, vo[p], 0, 0, 0);
    };
    if constexpr (kIn == 2) {
        if (img0 < img_end) issue_strip(img0, hi);
    }

    // ---- the channel's Toeplitz fragments -> registers --------------------------------------------------
    for (int i = tid; i < KS * WZ; i += 256) {
        const int ky = i / WZ, j = i % WZ - 24;
        // kFoldRes: the block's residual x + dwconv(x) as the centre tap w + 1 (below)
        wz[i] = (j >= 0 && j < KS) ? g.wgt[(int64_t)c * KS * KS + ky * KS + j] + ((kFoldRes && ky == R && j == R) ? 1.0f : 0.f) : 0.f;
    }
    __syncthreads();
    f16x8 bh[KS], bl[KS];
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = wz[ky * WZ + (kg * 8 + i) - n - 8 + R + 24];
        const Split8 s8 = (kProd == 1) ? split8_rn(v) : split8(v);
        bh[ky] = s8.hi;
        bl[ky] = s8.lo;                                          // (dead for kProd = 1)
    }
    const float bv = g.bias[c];
#ifdef SF_DW_TIMERS
    long long t_stage = 0, t_comp = 0;
    const long long t_begin = __builtin_readcyclecounter();
#endif
    const int ntx = g.w16 / 16, nty = g.strip_h / 16;
    const int ngroups = nty * ((ntx + TG - 1) / TG);            // groups of up to TG adjacent column tiles

    const int my_groups = (ngroups - wave + 3) / 4;             // groups this wave works on per image
    for (int img = img0; img < img_end; ++img) {
        const float* __restrict__ xp = g.x + img * g.x_img_stride + (int64_t)c * g.h * g.w;
        const _Float16* __restrict__ xh = reinterpret_cast<const _Float16*>(g.x) + img * g.x_img_stride + (int64_t)c * g.h * g.w;
        // kIn = 2: the plane this image was staged into / the one the next image goes to
        const char* hb = (kIn == 2) ? lds + ((img - img0) & 1) * plane_bytes : hi;
        // kOutF16: y holds fp16 planes, its strides count halves
        void* yp = kOutF16 ? (void*)(reinterpret_cast<_Float16*>(g.y) + img * g.y_img_stride + (int64_t)c * g.h * g.w)
                           : (void*)(g.y + img * g.y_img_stride + (int64_t)c * g.h * g.w);
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(yp, 0, g.h * g.w * (kOutF16 ? 2 : 4), 0x00020000);
#ifdef SF_DW_TIMERS
        const long long t0 = __builtin_readcyclecounter();
#endif
        if constexpr (kIn == 2) {
            // this wave's pieces of the strip have landed: they are older than the 16 output stores of its last tile group of
            // the previous image, which stay in flight (vmcnt retires in order); a wave without a group has no stores
            static_assert(TG == 4, "the wait below counts the TG * 4 = 16 stores of one tile group");
            if (img > img0 && my_groups > 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // everyone's pieces; and the other plane is no longer read
            if (img + 1 < img_end) issue_strip(img + 1, lds + (((img - img0) & 1) ^ 1) * plane_bytes);
        } else {
        // LDS-only barriers: the previous image's output stores stay in flight (a __syncthreads would drain them)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // previous image's tiles are done with the LDS planes
        // ---- stage + split the strip (rows ys-R .., columns -8 ..), zero padded: one octet = 8 consecutive columns
        // of one row = one A-fragment slot; FU octets per thread at a time, all their loads issued before any is used
        constexpr int FU = SF_DW_FU;
        for (int base = tid; base < rows_in * noct; base += 256 * FU) {
            float v[FU][8];
            int off[FU];
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const int idx = base + u * 256;
                const int ry_ = idx / noct, co = idx - ry_ * noct;
                const int gy = ys - R + ry_, gx = co * 8 - 8;
                const bool live = idx < rows_in * noct;
                const bool rowok = live && gy >= 0 && gy < g.h;
                off[u] = live ? (ry_ * g.stride16 + co) * 16 : -1;
#if defined(SF_DW_ABLATE) && SF_DW_ABLATE == 2
                if (false) {
#else
                if (kIn == 0 && rowok && g.vec_ok && gx >= 0 && gx + 8 <= g.w) {
#endif
                    const float4 q0 = *reinterpret_cast<const float4*>(xp + gy * g.w + gx);
                    const float4 q1 = *reinterpret_cast<const float4*>(xp + gy * g.w + gx + 4);
                    v[u][0] = q0.x; v[u][1] = q0.y; v[u][2] = q0.z; v[u][3] = q0.w;
                    v[u][4] = q1.x; v[u][5] = q1.y; v[u][6] = q1.z; v[u][7] = q1.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#if defined(SF_DW_ABLATE) && SF_DW_ABLATE == 2
                        v[u][i] = (float)(idx + i) * 1e-4f;
#else
                        v[u][i] = (rowok && gx + i >= 0 && gx + i < g.w) ? (kIn == 0 ? xp[gy * g.w + gx + i] : (float)xh[gy * g.w + gx + i]) : 0.f;
#endif
                }
            }
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                if (off[u] < 0) continue;
                const Split8 s8 = (kProd <= 2) ? split8_rn(v[u]) : split8(v[u]);
                *reinterpret_cast<f16x8*>(hi + off[u]) = s8.hi;
                *reinterpret_cast<f16x8*>(lo + off[u]) = s8.lo;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        }
#ifdef SF_DW_TIMERS
        const long long t1 = __builtin_readcyclecounter();
        t_stage += t1 - t0;
#endif

        for (int grp = wave; grp < ngroups; grp += 4) {
            const int ty = grp % nty, tx0 = (grp / nty) * TG;
            f32x4 acc[TG];
#pragma unroll
            for (int j = 0; j < TG; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // fragment of tile j, kernel row ky: rows ty*16 + ky + n, octets (tx0 + j)*2 + kg.  Tiles past the right
            // edge (ntx not a multiple of TG) read the last valid tile again and are not stored.
            int offj[TG];
#pragma unroll
            for (int j = 0; j < TG; ++j)
                offj[j] = ((ty * 16 + n) * g.stride16 + min(tx0 + j, ntx - 1) * 2 + kg) * 16;
            const int row_step = g.stride16 * 16;
            // output cell (j, r) of this lane: row gy0 + r, column gx0 + 16 j.  Stores are buffer ops with 32-bit offsets;
            // a cell outside the plane gets an out-of-range offset that the buffer unit drops (no branches).
            const int gy0 = ys + ty * 16 + 4 * kg, gx0 = tx0 * 16 + n;
            constexpr int EB = kOutF16 ? 2 : 4;                 // bytes per output element
            const int off0 = (gy0 * g.w + gx0) * EB;
            // software pipeline: the fragments of kernel row ky + 1 are requested before the MFMAs of row ky
            f16x8 ah[2][TG], al[2][TG];
#pragma unroll
            for (int j = 0; j < TG; ++j) {
                ah[0][j] = *reinterpret_cast<const f16x8*>(hb + offj[j]);
                if constexpr (kProd == 3) al[0][j] = *reinterpret_cast<const f16x8*>(lo + offj[j]);
            }
#if defined(SF_DW_ABLATE) && SF_DW_ABLATE == 3
#pragma unroll
            for (int ky = 0; ky < 1; ++ky) {
#else
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
#endif
                const int cur = ky & 1, nxt = cur ^ 1;
                if (ky + 1 < KS) {
#pragma unroll
                    for (int j = 0; j < TG; ++j) {
                        ah[nxt][j] = *reinterpret_cast<const f16x8*>(hb + offj[j] + (ky + 1) * row_step);
                        if constexpr (kProd == 3) al[nxt][j] = *reinterpret_cast<const f16x8*>(lo + offj[j] + (ky + 1) * row_step);
                    }
                }
                if constexpr (kProd == 3) {
#pragma unroll
                    for (int j = 0; j < TG; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[cur][j], bh[ky], acc[j], 0, 0, 0);
                }
                if constexpr (kProd >= 2) {
#pragma unroll
                    for (int j = 0; j < TG; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur][j], bl[ky], acc[j], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < TG; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur][j], bh[ky], acc[j], 0, 0, 0);
            }
            // C/D layout of the 16x16 MFMA: column = lane & 15, row = 4 * (lane >> 4) + reg.  The residual x comes back
            // from the staged planes as hi + lo (|x - (hi + lo)| <= 2^-20 |x|, the split's own precision): no
            // global load in the tile loop.
            const int xoff0 = ((ty * 16 + 4 * kg + R) * g.stride16) * 16 + (gx0 + 8) * 2;
#pragma unroll
            for (int j = 0; j < TG; ++j)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {                  // two outputs at a time: packed fp32 GELU
                    sf::f32x2 t;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if constexpr (kFoldRes) {
                            t[u] = acc[j][r + u] + bv;
                        } else {
                            const int xo = xoff0 + (r + u) * row_step + min(j, ntx - 1 - tx0) * 32;
                            float xv = (float)*reinterpret_cast<const _Float16*>(hb + xo);
                            if constexpr (kIn != 2) xv += (float)*reinterpret_cast<const _Float16*>(lo + xo);
                            t[u] = xv + (acc[j][r + u] + bv);
                        }
                    }
                    const sf::f32x2 a = sf::gelu2<(kProd <= 2) && kOutF16 && SF_GEMM_FAST_GELU>(t);   // polynomial GELU where the result leaves as fp16
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
#if defined(SF_DW_ABLATE) && SF_DW_ABLATE == 1
                        const bool ok = a[u] == 123.456f;              // (never true: the value still has to be computed)
#else
                        const bool ok = (gx0 + j * 16 < g.w) && (gy0 + r + u < g.h);
#endif
                        const int so = ok ? off0 + ((r + u) * g.w + j * 16) * EB : (int)0x80000000u;
                        if constexpr (kOutF16) {
                            const _Float16 o = (_Float16)a[u];
                            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, o), ry, so, 0, 0);
                        } else {
                            const float o = a[u];
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), ry, so, 0, 0);
                        }
                    }
                }
        }
#ifdef SF_DW_TIMERS
        t_comp += __builtin_readcyclecounter() - t1;
#endif
    }
#ifdef SF_DW_TIMERS
    if (g.ts && tid == 0) {
        long long* d = g.ts + ((int64_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4;
        d[0] = __builtin_readcyclecounter() - t_begin; d[1] = t_stage; d[2] = t_comp; d[3] = img_end - img0;
    }
#endif
}

}  // namespace

static int dwconv_dispatch(const void* x_, int x_f16, int64_t x_img_stride, const float* wgt, const float* bias, void* y_,
                    int64_t y_img_stride, int y_f16, int n_img, int C, int h, int w, int ksize, int precision, void* stream) {
    const float* x = static_cast<const float*>(x_);
    float* y = static_cast<float*>(y_);
    SF_REQUIRE(x && wgt && bias && y, "sf_dwconv_res_gelu: null pointer");
    SF_REQUIRE(y_f16 == 0 || y_f16 == 1, "sf_dwconv_res_gelu: y_f16 must be 0 or 1");
    SF_REQUIRE(n_img > 0 && C > 0 && h > 0 && w > 0, "sf_dwconv_res_gelu: bad dims");
    SF_REQUIRE(ksize == 15 || ksize == 7, "sf_dwconv_res_gelu: kernel size %d not built (7, 15)", ksize);
    SF_REQUIRE(precision >= SF_PRECISION_FP32 && precision <= SF_PRECISION_F16, "sf_dwconv_res_gelu: bad precision");
    SF_REQUIRE((int64_t)h * w < (1 << 30), "sf_dwconv_res_gelu: plane too large");
    // K = 7, three products: only 7/32 of the Toeplitz entries are non-zero and the stencil is the faster kernel (65 vs 87 us at
    // 128 channels x 24 images); K = 15 runs 1.45x faster on the matrix cores
    const bool two = precision != SF_PRECISION_FP32 && precision != SF_PRECISION_F16X3;
    const bool one = two && precision == SF_PRECISION_F16;       // weights rounded once to fp16 as well
    SF_REQUIRE(!x_f16 || (two && y_f16), "sf_dwconv_res_gelu_f16in: fp16 input is the f16x2 / f16 hand-over (fp16 output, "
                                         "SF_PRECISION_F16X2 or SF_PRECISION_F16)");
    // K = 7 in the two-product modes also runs on the matrix cores (7 x 2 MFMAs per tile against 49 FMAs per output)
    if (precision != SF_PRECISION_FP32 && (ksize == 15 || two)) {
        DwmArgs m;
        m.x = x; m.wgt = wgt; m.bias = bias; m.y = y; m.x_img_stride = x_img_stride; m.y_img_stride = y_img_stride;
        m.n_img = n_img; m.C = C; m.h = h; m.w = w;
        m.w16 = sf::ceil_div(w, 16) * 16;
        const int base16 = (m.w16 + 16) / 8;
        m.stride16 = (base16 % 4 == 2) ? base16 : base16 + (2 - base16 % 4 + 4) % 4;
        // rows per strip: as many 16-row tiles as fit ~60 KB of LDS (two fp16 planes)
        const int h16 = sf::ceil_div(h, 16) * 16;
        int strip = h16;
        while (strip > 16 && (size_t)2 * (strip + ksize - 1) * m.stride16 * 16 > 60 * 1024) strip -= 16;
        m.strip_h = strip;
        m.plane_bytes = (strip + ksize - 1) * m.stride16 * 16;
        // fp16 rows of whole, 16-byte aligned octets: the DMA-staged, double-buffered form (whole 4-KB pieces per plane)
        const bool dma = x_f16 && (w % 8 == 0) && (x_img_stride % 8 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                         sf::ceil_div(m.plane_bytes, 4096) <= 8;
        const bool dma_fits = (size_t)2 * sf::ceil_div(m.plane_bytes, 4096) * 4096 + (size_t)ksize * 64 * sizeof(float) <= 64 * 1024;
        const bool use_dma = dma && dma_fits;
        if (use_dma) m.plane_bytes = sf::ceil_div(m.plane_bytes, 4096) * 4096;
        const size_t lds = (size_t)2 * m.plane_bytes + (size_t)ksize * 64 * sizeof(float);
        SF_REQUIRE(lds <= 64 * 1024, "sf_dwconv_res_gelu: width %d too large for the matrix-core kernel", w);
        m.vec_ok = ((w & 3) == 0) && ((x_img_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
        // several images of a channel per workgroup (the Toeplitz fragments are built once), but keep >= ~2048 workgroups
        const int strips = sf::ceil_div(h, strip);
        // (4096 / 8192 / 16384 workgroups measured within +-3 % of 2048 on every layer shape of the update block)
        // (the double-buffered form hides an image's staging behind the previous image's tiles: more images per workgroup)
        const int target_wgs = use_dma ? SF_DW_DMA_WGS : 2048;
        int groups = sf::ceil_div(target_wgs, C * strips);
        if (groups > n_img) groups = n_img;
        if (groups < 1) groups = 1;
        m.imgs_per_wg = sf::ceil_div(n_img, groups);
#ifdef SF_DW_TIMERS
        m.ts = getenv("SF_DW_TS_BUF") ? (long long*)strtoull(getenv("SF_DW_TS_BUF"), nullptr, 0) : nullptr;
#endif
        dim3 grid(C, strips, sf::ceil_div(n_img, m.imgs_per_wg));
        SF_REQUIRE(strips <= 65535 && grid.z <= 65535, "sf_dwconv_res_gelu: grid too large");
        if (x_f16) {
#define SF_DW_F16IN(KS_, PR_)                                                                                            \
    do {                                                                                                                 \
        if (use_dma) hipLaunchKernelGGL((dwconv_mfma_kernel<KS_, true, PR_, 2>), grid, dim3(256), lds, (hipStream_t)stream, m); \
        else hipLaunchKernelGGL((dwconv_mfma_kernel<KS_, true, PR_, 1>), grid, dim3(256), lds, (hipStream_t)stream, m);   \
    } while (0)
            if (ksize == 7 && one) SF_DW_F16IN(7, 1);
            else if (ksize == 7) SF_DW_F16IN(7, 2);
            else if (one) SF_DW_F16IN(15, 1);
            else SF_DW_F16IN(15, 2);
#undef SF_DW_F16IN
            return sf::check_launch("sf_dwconv_res_gelu_f16in");
        }
        if (ksize == 7) {
            if (one && y_f16) hipLaunchKernelGGL((dwconv_mfma_kernel<7, true, 1>), grid, dim3(256), lds, (hipStream_t)stream, m);
            else if (one) hipLaunchKernelGGL((dwconv_mfma_kernel<7, false, 1>), grid, dim3(256), lds, (hipStream_t)stream, m);
            else if (y_f16) hipLaunchKernelGGL((dwconv_mfma_kernel<7, true, 2>), grid, dim3(256), lds, (hipStream_t)stream, m);
            else hipLaunchKernelGGL((dwconv_mfma_kernel<7, false, 2>), grid, dim3(256), lds, (hipStream_t)stream, m);
            return sf::check_launch("sf_dwconv_res_gelu(mfma7)");
        }
        if (one && y_f16)
            hipLaunchKernelGGL((dwconv_mfma_kernel<15, true, 1>), grid, dim3(256), lds, (hipStream_t)stream, m);
        else if (one)
            hipLaunchKernelGGL((dwconv_mfma_kernel<15, false, 1>), grid, dim3(256), lds, (hipStream_t)stream, m);
        else if (y_f16 && two)
            hipLaunchKernelGGL((dwconv_mfma_kernel<15, true, 2>), grid, dim3(256), lds, (hipStream_t)stream, m);
        else if (y_f16)
            hipLaunchKernelGGL((dwconv_mfma_kernel<15, true, 3>), grid, dim3(256), lds, (hipStream_t)stream, m);
        else if (two)
            hipLaunchKernelGGL((dwconv_mfma_kernel<15, false, 2>), grid, dim3(256), lds, (hipStream_t)stream, m);
        else
            hipLaunchKernelGGL((dwconv_mfma_kernel<15, false, 3>), grid, dim3(256), lds, (hipStream_t)stream, m);
        return sf::check_launch("sf_dwconv_res_gelu(mfma)");
    }
    DwArgs g;
    g.x = x; g.wgt = wgt; g.bias = bias; g.y = y; g.C = C; g.h = h; g.w = w;
    g.x_img_stride = x_img_stride; g.y_img_stride = y_img_stride;
    g.tiles_x = sf::ceil_div(w, TX);
    SF_REQUIRE(g.tiles_x <= kMaxThreads, "sf_dwconv_res_gelu: width %d too large", w);
    int tiles_y = sf::ceil_div(h, TY);
    if (tiles_y * g.tiles_x > kMaxThreads) tiles_y = kMaxThreads / g.tiles_x;
    g.strip_h = tiles_y * TY;
    g.wp4 = (g.tiles_x * TX + ksize - 1 + 3) / 4;
    g.vec_store = ((w & 3) == 0) && ((y_img_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(y) & (y_f16 ? 7 : 15)) == 0);
    const int rows = g.strip_h + ksize - 1;
    const size_t lds = ((size_t)rows * g.wp4 * 4 + 8 + ksize * ksize + 8) * sizeof(float);
    SF_REQUIRE(lds <= 64 * 1024, "sf_dwconv_res_gelu: strip needs %zu bytes of LDS", lds);
    const int threads = ((tiles_y * g.tiles_x + 63) / 64) * 64;
    dim3 grid(n_img * C, sf::ceil_div(h, g.strip_h));
    if (ksize == 15 && y_f16)
        hipLaunchKernelGGL((dwconv_res_gelu_kernel<15, true>), grid, dim3(threads), lds, (hipStream_t)stream, g);
    else if (ksize == 15)
        hipLaunchKernelGGL((dwconv_res_gelu_kernel<15, false>), grid, dim3(threads), lds, (hipStream_t)stream, g);
    else if (y_f16)
        hipLaunchKernelGGL((dwconv_res_gelu_kernel<7, true>), grid, dim3(threads), lds, (hipStream_t)stream, g);
    else
        hipLaunchKernelGGL((dwconv_res_gelu_kernel<7, false>), grid, dim3(threads), lds, (hipStream_t)stream, g);
    return sf::check_launch("sf_dwconv_res_gelu");
}

extern "C" int sf_dwconv_res_gelu(const float* x, int64_t x_img_stride, const float* wgt, const float* bias, void* y,
                                  int64_t y_img_stride, int y_f16, int n_img, int C, int h, int w, int ksize,
                                  int precision, void* stream) {
    return dwconv_dispatch(x, 0, x_img_stride, wgt, bias, y, y_img_stride, y_f16, n_img, C, h, w, ksize, precision, stream);
}

// x as fp16 ROWS [img][C][h*w] (x_img_stride in halves), y as fp16 rows: the config-2 hand-over of an SK block's x2 and x3
// (precision SF_PRECISION_F16X2 or SF_PRECISION_F16).  The conv input and the residual are the fp16 value itself.
extern "C" int sf_dwconv_res_gelu_f16in(const void* x_f16, int64_t x_img_stride, const float* wgt, const float* bias,
                                        void* y_f16, int64_t y_img_stride, int n_img, int C, int h, int w, int ksize,
                                        int precision, void* stream) {
    return dwconv_dispatch(x_f16, 1, x_img_stride, wgt, bias, y_f16, y_img_stride, 1, n_img, C, h, w, ksize, precision, stream);
}
