// Depthwise KxK convolution fused with bias, residual and exact GELU:
//     y = gelu(x + dwconv_KxK(x) + b)            (reference core/update.py:33-34, kernels 15 and 7)
//
// VALU-bound stencil (225 or 49 FMAs per output).  One workgroup owns one (image, channel) plane
// strip: the strip plus its K/2 halo is staged once in LDS (zero padded, so the inner loop has no
// bounds checks), each thread produces a 4x4 output tile from a sliding window held in registers
// (one ds_read_b128 row segment feeds 4 output columns x K taps).  The K*K weights of the channel sit in LDS
// behind the tile and are read one kernel row at a time with broadcast reads into VGPRs.  (Measured: fetching
// them through the scalar cache as SGPR operands of v_fmac_f32 was 18 % slower -- every weight row costs an
// lgkmcnt(0) drain -- and a register sliding window that cuts the LDS row reads 4x changed nothing: the
// kernel is bound by VALU issue, 47 TFLOP/s of fp32 FMA.)
#include "sf_common.h"

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
    int vec_store;
};

template <int KS>
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
    float* yp = g.y + img * g.y_img_stride + (int64_t)c * g.h * g.w;
#pragma unroll
    for (int oy = 0; oy < TY; ++oy) {
        const int gy = ys + ty * TY + oy;
        if (gy >= g.h) continue;
        const float* ctr = tile + (ty * TY + oy + R) * wp + tx * TX + R;
        float o[TX];
#pragma unroll
        for (int ox = 0; ox < TX; ++ox) o[ox] = sf::gelu_erf(ctr[ox] + (acc[oy][ox] + bv));
        const int gx = tx * TX;
        if (g.vec_store && gx + 3 < g.w) {
            *reinterpret_cast<float4*>(yp + gy * g.w + gx) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
            for (int ox = 0; ox < TX; ++ox)
                if (gx + ox < g.w) yp[gy * g.w + gx + ox] = o[ox];
        }
    }
}

}  // namespace

extern "C" int sf_dwconv_res_gelu(const float* x, int64_t x_img_stride, const float* wgt, const float* bias, float* y,
                                  int64_t y_img_stride, int n_img, int C, int h, int w, int ksize, void* stream) {
    SF_REQUIRE(x && wgt && bias && y, "sf_dwconv_res_gelu: null pointer");
    SF_REQUIRE(n_img > 0 && C > 0 && h > 0 && w > 0, "sf_dwconv_res_gelu: bad dims");
    SF_REQUIRE(ksize == 15 || ksize == 7, "sf_dwconv_res_gelu: kernel size %d not built (7, 15)", ksize);
    DwArgs g;
    g.x = x; g.wgt = wgt; g.bias = bias; g.y = y; g.C = C; g.h = h; g.w = w;
    g.x_img_stride = x_img_stride; g.y_img_stride = y_img_stride;
    g.tiles_x = sf::ceil_div(w, TX);
    SF_REQUIRE(g.tiles_x <= kMaxThreads, "sf_dwconv_res_gelu: width %d too large", w);
    int tiles_y = sf::ceil_div(h, TY);
    if (tiles_y * g.tiles_x > kMaxThreads) tiles_y = kMaxThreads / g.tiles_x;
    g.strip_h = tiles_y * TY;
    g.wp4 = (g.tiles_x * TX + ksize - 1 + 3) / 4;
    g.vec_store = ((w & 3) == 0) && ((y_img_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
    const int rows = g.strip_h + ksize - 1;
    const size_t lds = ((size_t)rows * g.wp4 * 4 + 8 + ksize * ksize + 8) * sizeof(float);
    SF_REQUIRE(lds <= 64 * 1024, "sf_dwconv_res_gelu: strip needs %zu bytes of LDS", lds);
    const int threads = ((tiles_y * g.tiles_x + 63) / 64) * 64;
    dim3 grid(n_img * C, sf::ceil_div(h, g.strip_h));
    if (ksize == 15)
        hipLaunchKernelGGL(dwconv_res_gelu_kernel<15>, grid, dim3(threads), lds, (hipStream_t)stream, g);
    else
        hipLaunchKernelGGL(dwconv_res_gelu_kernel<7>, grid, dim3(threads), lds, (hipStream_t)stream, g);
    return sf::check_launch("sf_dwconv_res_gelu");
}
