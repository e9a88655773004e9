// Kernels of the Twins_CSC encoder that are not GEMM-shaped: the attention cores of timm's
// LocallyGroupedAttn / GlobalSubSampleAttn and the depthwise 3x3 positional conv.  Reference call sites:
// core/encoders/twins_csc.py:68-76 (block / pos_block loop over the first two twins_svt_large stages); arithmetic as
// published in timm/models/twins.py (third-party, restated; see oracle/twins_oracle.py).
// The encoder runs once per clip, outside the refinement loop: these kernels are exact fp32 on the VALU and written
// for clarity; every Linear / strided conv of the encoder goes through sf_gemm, every LayerNorm through sf_layernorm_cm.
//
// Layout: channel-major planes [img][C][N], N = tokens of the (T*h) x w grid, channel = head * 32 + d.
#include "sf_common.h"

namespace {

constexpr int HD = 32;             // head dim of twins_svt_large in every stage (128/4, 256/8)

// ---- LocallyGroupedAttn core: softmax(scale q k^T) v inside each ws x ws window ------------------------------------
// qkv planes [img][3C][H*W] (rows: q | k | v).  The reference zero-pads the (already normalised) token grid to a multiple
// of the window BEFORE the qkv Linear, so a padded token has q = k = v = the qkv bias: it takes part in the softmax of
// the real tokens of its window (its own output is dropped).  One workgroup = one window, wave = head, lane = query.
__global__ __launch_bounds__(256) void window_attn_kernel(const float* qkv, int64_t img_stride, const float* bias, float* out,
                                                          int64_t out_img_stride, int C, int H, int W, int ws, int nww) {
    __shared__ float lds[4 * 2 * 49 * (HD + 1)];   // per head (wave): K [ws*ws][HD+1], V [ws*ws][HD+1]
    const int hl = threadIdx.x >> 6, head = blockIdx.z * 4 + hl, lane = threadIdx.x & 63;
    const int wy = blockIdx.x / nww, wx = blockIdx.x % nww, img = blockIdx.y;
    const int nt = ws * ws, N = H * W;
    const float* base = qkv + (int64_t)img * img_stride;
    float* sk = lds + hl * 2 * nt * (HD + 1);
    float* sv = sk + nt * (HD + 1);
    for (int i = lane; i < nt * HD; i += 64) {
        const int tok = i % nt, d = i / nt;                          // lanes walk tokens: rows of 7 consecutive pixels
        const int y = wy * ws + tok / ws, x = wx * ws + tok % ws;
        const int ch = head * HD + d;
        const bool in = y < H && x < W;
        sk[tok * (HD + 1) + d] = in ? base[(int64_t)(C + ch) * N + y * W + x] : bias[C + ch];
        sv[tok * (HD + 1) + d] = in ? base[(int64_t)(2 * C + ch) * N + y * W + x] : bias[2 * C + ch];
    }
    __syncthreads();
    if (lane >= nt) return;
    const int y = wy * ws + lane / ws, x = wx * ws + lane % ws;
    if (y >= H || x >= W) return;
    const float scale = 0.17677669529663687f;                        // 32^-0.5
    float q[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) q[d] = base[(int64_t)(head * HD + d) * N + y * W + x] * scale;
    float mx = -3.0e38f;
    float s[49];                                                     // ws <= 7 (host-checked)
#pragma unroll
    for (int j = 0; j < 49; ++j) {
        if (j < nt) {
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) a = fmaf(q[d], sk[j * (HD + 1) + d], a);
            s[j] = a;
            mx = fmaxf(mx, a);
        }
    }
    float sum = 0.f, o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
#pragma unroll
    for (int j = 0; j < 49; ++j) {
        if (j < nt) {
            const float p = expf(s[j] - mx);
            sum += p;
#pragma unroll
            for (int d = 0; d < HD; ++d) o[d] = fmaf(p, sv[j * (HD + 1) + d], o[d]);
        }
    }
    const float inv = 1.0f / sum;
    float* op = out + (int64_t)img * out_img_stride + y * W + x;
#pragma unroll
    for (int d = 0; d < HD; ++d) op[(int64_t)(head * HD + d) * N] = o[d] * inv;
}

// ---- GlobalSubSampleAttn core: every token attends to the M sub-sampled tokens ---------------------------------------------
// q [img][C][N], kv [img][2C][M] (rows k | v).  Workgroup = 256 queries of one head; keys/values of the head stream through
// LDS in chunks of 64; online softmax per query (thread), exact fp32.
__global__ __launch_bounds__(256) void subsample_attn_kernel(const float* q, int64_t q_img_stride, const float* kv,
                                                             int64_t kv_img_stride, float* out, int64_t out_img_stride, int C,
                                                             int N, int M) {
    constexpr int KC = 64;
    __shared__ float sk[KC][HD], sv[KC][HD];
    const int head = blockIdx.y, img = blockIdx.z;
    const int n = blockIdx.x * 256 + threadIdx.x, nc = n < N ? n : N - 1;
    const float scale = 0.17677669529663687f;
    const float* qp = q + (int64_t)img * q_img_stride + (int64_t)head * HD * N + nc;
    const float* kp = kv + (int64_t)img * kv_img_stride + (int64_t)head * HD * M;
    const float* vp = kp + (int64_t)C * M;
    float qv[HD], o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { qv[d] = qp[(int64_t)d * N] * scale; o[d] = 0.f; }
    float mx = -3.0e38f, sum = 0.f;
    for (int j0 = 0; j0 < M; j0 += KC) {
        __syncthreads();
        for (int i = threadIdx.x; i < KC * HD; i += 256) {
            const int j = i % KC, d = i / KC;                        // lanes walk keys: coalesced rows
            const bool in = j0 + j < M;
            sk[j][d] = in ? kp[(int64_t)d * M + j0 + j] : 0.f;
            sv[j][d] = in ? vp[(int64_t)d * M + j0 + j] : 0.f;
        }
        __syncthreads();
        const int jn = (M - j0 < KC) ? M - j0 : KC;
        for (int j = 0; j < jn; ++j) {
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) a = fmaf(qv[d], sk[j][d], a);
            if (a > mx) {                                            // rescale the running sums to the new maximum
                const float r = expf(mx - a);
                sum *= r;
#pragma unroll
                for (int d = 0; d < HD; ++d) o[d] *= r;
                mx = a;
            }
            const float p = expf(a - mx);
            sum += p;
#pragma unroll
            for (int d = 0; d < HD; ++d) o[d] = fmaf(p, sv[j][d], o[d]);
        }
    }
    if (n >= N) return;
    const float inv = 1.0f / sum;
    float* op = out + (int64_t)img * out_img_stride + (int64_t)head * HD * N + n;
#pragma unroll
    for (int d = 0; d < HD; ++d) op[(int64_t)d * N] = o[d] * inv;
}

// ---- PosConv: y = x + dwconv3x3(x) + b over the [H][W] token grid ----------------------------------------------------------
__global__ __launch_bounds__(256) void dwconv3x3_res_kernel(const float* x, int64_t x_img_stride, const float* w, const float* b,
                                                            float* y, int64_t y_img_stride, int C, int H, int W) {
    const int p = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, img = blockIdx.z;
    if (p >= H * W) return;
    const int py = p / W, px = p % W;
    const float* xp = x + (int64_t)img * x_img_stride + (int64_t)c * H * W;
    float acc = b[c];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = py + dy, xx = px + dx;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) acc = fmaf(w[c * 9 + (dy + 1) * 3 + dx + 1], xp[yy * W + xx], acc);
        }
    y[(int64_t)img * y_img_stride + (int64_t)c * H * W + p] = xp[p] + acc;
}

}  // namespace

extern "C" int sf_window_attn(const float* qkv, int64_t qkv_img_stride, const float* qkv_bias, float* out,
                              int64_t out_img_stride, int n_img, int C, int heads, int H, int W, int ws, void* stream) {
    SF_REQUIRE(qkv && qkv_bias && out, "sf_window_attn: null pointer");
    SF_REQUIRE(n_img > 0 && H > 0 && W > 0 && n_img <= 65535, "sf_window_attn: bad dims");
    SF_REQUIRE(heads >= 4 && heads % 4 == 0 && C == heads * HD, "sf_window_attn: needs C = heads * 32, heads a multiple of 4 (got C=%d heads=%d)", C, heads);
    SF_REQUIRE(ws >= 2 && ws <= 7, "sf_window_attn: window size must be 2..7 (got %d)", ws);
    const int nwh = sf::ceil_div(H, ws), nww = sf::ceil_div(W, ws);
    hipLaunchKernelGGL(window_attn_kernel, dim3(nwh * nww, n_img, heads / 4), dim3(256), 0, (hipStream_t)stream, qkv, qkv_img_stride,
                       qkv_bias, out, out_img_stride, C, H, W, ws, nww);
    return sf::check_launch("sf_window_attn");
}

extern "C" int sf_subsample_attn(const float* q, int64_t q_img_stride, const float* kv, int64_t kv_img_stride, float* out,
                                 int64_t out_img_stride, int n_img, int C, int heads, int N, int M, void* stream) {
    SF_REQUIRE(q && kv && out, "sf_subsample_attn: null pointer");
    SF_REQUIRE(n_img > 0 && N > 0 && M > 0 && n_img <= 65535 && heads <= 65535, "sf_subsample_attn: bad dims");
    SF_REQUIRE(heads >= 1 && C == heads * HD, "sf_subsample_attn: needs C = heads * 32 (got C=%d heads=%d)", C, heads);
    hipLaunchKernelGGL(subsample_attn_kernel, dim3(sf::ceil_div(N, 256), heads, n_img), dim3(256), 0, (hipStream_t)stream, q,
                       q_img_stride, kv, kv_img_stride, out, out_img_stride, C, N, M);
    return sf::check_launch("sf_subsample_attn");
}

extern "C" int sf_dwconv3x3_res(const float* x, int64_t x_img_stride, const float* w, const float* b, float* y,
                                int64_t y_img_stride, int n_img, int C, int H, int W, void* stream) {
    SF_REQUIRE(x && w && b && y, "sf_dwconv3x3_res: null pointer");
    SF_REQUIRE(n_img > 0 && C > 0 && H > 0 && W > 0 && n_img <= 65535 && C <= 65535, "sf_dwconv3x3_res: bad dims");
    hipLaunchKernelGGL(dwconv3x3_res_kernel, dim3(sf::ceil_div(H * W, 256), C, n_img), dim3(256), 0, (hipStream_t)stream, x,
                       x_img_stride, w, b, y, y_img_stride, C, H, W);
    return sf::check_launch("sf_dwconv3x3_res");
}
