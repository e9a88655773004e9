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

// ---- GlobalSubSampleAttn core on the matrix cores --------------------------------------------------------------------------
// Same mathematics as subsample_attn_kernel, flash-style with the logits TRANSPOSED (csrc/attn.hip has the derivation):
//   S^T[key][query] = K Q^T          v_mfma_f32_32x32x16_f16, A = K fragment, B = Q fragment (registers, whole kernel)
//   O^T[d][query]  += V^T P^T        A = V fragment, B = P^T = the logits' accumulator registers converted in place
// The C/D layout puts ONE query in a lane (col = lane & 31) with 16 keys of a 32-key tile in its registers
// (key = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)), so max / sum of the online softmax are in-lane plus one exchange with
// lane ^ 32, and registers 8m .. 8m+7 are directly the B operand of k-step m once V is packed in the same key order.
// Head dim 32 = two k-steps of 16; M keys (1760 at the Sintel shape: the 55 x 32 grid of the sr conv) = M/32 tiles.
// K and V of one (image, head) are 4 KB per tile and stay in L2; the softmax (16 v_exp_f32 + bookkeeping per query tile
// and key tile) dominates the MFMA work 3:1, so fragments are read straight from the packed image with one lane-linear
// 16-byte load each -- no LDS, no barriers: a wave is a self-contained stream over the tiles for its 64 queries.
// NP = products per contraction: 1 (every operand rounded once to fp16) or 3 (hi + lo split of Q, K, P, V: fp32-class).
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int SQT = 1;             // 32-query tiles per wave
constexpr int SPARTS = 8;          // 1 KB fragments per key tile: K s0, K s1, V m0, V m1 (hi), then the same four (lo)

__device__ __forceinline__ void split8(const float (&x)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const _Float16 h = (_Float16)x[i];
        hi[i] = h;
        lo[i] = (_Float16)(x[i] - (float)h);
    }
}

// One lane's 16 outputs of a 32-query tile (channels head*32 + (r & 3) + 8 (r >> 2) + 4 khalf of token n): fp32 planes
// [C][N] and / or fp16 k-octet planes [C/8][N][8] (SF_LAYOUT_F16_KOCT, the operand image of the proj GEMM): registers
// 4g .. 4g+3 are four consecutive channels of octet head*4 + g -- one 8-byte store, lanes l and l ^ 32 complete the octet.
__device__ __forceinline__ void store_attn_out(float* out, int64_t out_img_stride, _Float16* out16, int64_t out16_img_stride, int img,
                                               int head, int N, int n, int khalf, const f32x16& o, float inv) {
    if (out) {
        float* op = out + (int64_t)img * out_img_stride + (int64_t)head * HD * N + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) op[(int64_t)((r & 3) + 8 * (r >> 2) + 4 * khalf) * N] = o[r] * inv;
    }
    if (out16) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        _Float16* op = out16 + (int64_t)img * out16_img_stride + ((int64_t)head * 4 * N + n) * 8 + khalf * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            h4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (_Float16)(o[4 * g + e] * inv);
            *reinterpret_cast<h4*>(op + (int64_t)g * N * 8) = v;
        }
    }
}

// kv [img][2C][M] fp32 (rows k | v) -> ws[img][head][tile][part][lane][8 halves]
__global__ __launch_bounds__(256) void subsample_pack_kv_kernel(const float* kv, int64_t kv_img_stride, char* ws, int C, int M,
                                                                int tiles) {
    const int tile = blockIdx.x, head = blockIdx.y, img = blockIdx.z, heads = gridDim.y;
    const int part = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, khalf = lane >> 5;
    const float* kp = kv + (int64_t)img * kv_img_stride + (int64_t)head * HD * M;
    const float* vp = kp + (int64_t)C * M;
    float x[8];
    if (part < 2) {                                                   // K, k-step `part`: row = key, k = dims
        const int key = tile * 32 + l31;
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = key < M ? kp[(int64_t)(16 * part + 8 * khalf + i) * M + key] : 0.f;
    } else {                                                          // V, k-step m: row = d, k = keys in accumulator order
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int key = tile * 32 + 16 * (part - 2) + (i & 3) + 8 * (i >> 2) + 4 * khalf;
            x[i] = key < M ? vp[(int64_t)l31 * M + key] : 0.f;
        }
    }
    f16x8 hi, lo;
    split8(x, hi, lo);
    char* dst = ws + ((((int64_t)img * heads + head) * tiles + tile) * SPARTS + part) * 1024 + lane * 16;
    *reinterpret_cast<f16x8*>(dst) = hi;
    *reinterpret_cast<f16x8*>(dst + 4 * 1024) = lo;
}

template <int NP>
__global__ __launch_bounds__(256) void subsample_attn_mfma_kernel(const float* q, int64_t q_img_stride, const char* ws, float* out,
                                                                  int64_t out_img_stride, _Float16* out16, int64_t out16_img_stride,
                                                                  int N, int M, int tiles) {
    constexpr bool kLo = NP == 3;
    const int head = blockIdx.y, img = blockIdx.z, heads = gridDim.y;
    const int lane = threadIdx.x & 63, l31 = lane & 31, khalf = lane >> 5;
    const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (32 * SQT);
    if (n0 >= N) return;                                              // wave-uniform; the kernel has no barrier
    const float qmul = 0.17677669529663687f * 1.44269504088896340736f;   // 32^-0.5 * log2(e): softmax = exp2(s - max)
    f16x8 qh[SQT][2], ql[SQT][2];
    {
        const float* qp = q + (int64_t)img * q_img_stride + (int64_t)head * HD * N;
#pragma unroll
        for (int t = 0; t < SQT; ++t) {
            const int n = n0 + t * 32 + l31, nc = n < N ? n : N - 1;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float x[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = sf::mul_rn(qp[(int64_t)(16 * s + 8 * khalf + i) * N + nc], qmul);
                split8(x, qh[t][s], ql[t][s]);
            }
        }
    }
    f32x16 o[SQT];
    float m_run[SQT], l_run[SQT];
#pragma unroll
    for (int t = 0; t < SQT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
        m_run[t] = -1.0e30f;
        l_run[t] = 0.f;
    }
    const f16x8* wp = reinterpret_cast<const f16x8*>(ws + ((int64_t)img * heads + head) * tiles * (SPARTS * 1024)) + lane;
    constexpr int NF = kLo ? 8 : 4;
    f16x8 cur[NF], nxt[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) cur[f] = wp[f * 64];
    for (int kt = 0; kt < tiles; ++kt) {
        const int kn = kt + 1 < tiles ? kt + 1 : kt;                  // the last trip re-reads its own tile (no branch)
#pragma unroll
        for (int f = 0; f < NF; ++f) nxt[f] = wp[(kn * SPARTS + f) * 64];
        const bool ragged = (kt + 1) * 32 > M;
#pragma unroll
        for (int t = 0; t < SQT; ++t) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (kLo) {
                    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[4 + ks], qh[t][ks], s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[ks], ql[t][ks], s, 0, 0, 0);
                }
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[ks], qh[t][ks], s, 0, 0, 0);
            }
            if (ragged) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf < M) ? s[r] : -1.0e30f;
            }
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[t], mx);
            const bool grew = __any(m_new > m_run[t]);                // rescale the accumulators only when some lane needs it
            const float alpha = grew ? __builtin_amdgcn_exp2f(m_run[t] - m_new) : 1.0f;
            m_run[t] = m_new;
            float psum = 0.f;
            f16x8 ph[2], pl[2];
            if (kLo) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s[r] - m_new);
                    psum += p;
                    const _Float16 h = (_Float16)p;
                    ph[r >> 3][r & 7] = h;
                    pl[r >> 3][r & 7] = (_Float16)(p - (float)h);
                }
            } else {
                // one product: the row is normalised by the ROUNDED weights, the ones that are actually multiplied
                const f16x2 ones = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f16x2 pp;
                    pp[0] = (_Float16)__builtin_amdgcn_exp2f(s[r] - m_new);
                    pp[1] = (_Float16)__builtin_amdgcn_exp2f(s[r + 1] - m_new);
                    psum = __builtin_amdgcn_fdot2(pp, ones, psum, false);
                    ph[r >> 3][r & 7] = pp[0];
                    ph[r >> 3][(r & 7) + 1] = pp[1];
                }
            }
            l_run[t] = l_run[t] * alpha + psum;
            if (grew) {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                if (kLo) {
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[6 + m], ph[m], o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[2 + m], pl[m], o[t], 0, 0, 0);
                }
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[2 + m], ph[m], o[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) cur[f] = nxt[f];
    }
#pragma unroll
    for (int t = 0; t < SQT; ++t) {
        const float inv = 1.0f / (l_run[t] + __shfl_xor(l_run[t], 32, 64));
        const int n = n0 + t * 32 + l31;
        if (n < N) store_attn_out(out, out_img_stride, out16, out16_img_stride, img, head, N, n, khalf, o[t], inv);
    }
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

// ---- LocallyGroupedAttn core on the matrix cores ---------------------------------------------------------------------------
// One wave = one (window, head): ws x ws <= 49 tokens padded to 64 = two 32-token tiles on both sides.  Same transposed
// scheme as the sub-sample kernel, without the online part (all keys of a query are in registers at once):
//   S^T[key][query] = K Q^T : A = K fragment (row = key token, k = dims), B = Q fragment (col = query token), both loaded
//        straight from the planes -- a lane is a token, so a wave instruction reads the window's rows of ws pixels;
//   O^T[d][query] = V^T P^T : the V fragment wants row = d, k = tokens in accumulator order: V is loaded like K (lane =
//        token) and turned through a per-wave LDS image [d][token] (fp16 hi [, lo]).
// Tokens of the window that lie outside the grid are the reference's zero padding: k = v = the qkv bias, they take part in
// the softmax; tokens >= ws*ws (padding to 64) are masked out.  NP as in the sub-sample kernel.
constexpr int WLS = 72;            // halves per d-row of the LDS V image (64 tokens + 8: 144-byte rows)

template <int NP>
__global__ __launch_bounds__(256) void window_attn_mfma_kernel(const float* qkv, int64_t img_stride, const float* bias, float* out,
                                                               int64_t out_img_stride, _Float16* out16, int64_t out16_img_stride,
                                                               int C, int H, int W, int ws, int nww) {
    constexpr bool kLo = NP == 3;
    __shared__ _Float16 lds[4 * 2 * HD * WLS];
    const int wave = threadIdx.x >> 6, head = blockIdx.z * 4 + wave, lane = threadIdx.x & 63, l31 = lane & 31, khalf = lane >> 5;
    // a 128-byte line of a token row is shared by 4.6 windows of a window row: consecutive windows must meet in ONE L2
    // (workgroup b runs on XCD b % 8: with the plain order every line was fetched by ~4.5 XCDs)
    const int win = sf::xcd_linear_id(blockIdx.x, gridDim.x);
    const int wy = win / nww, wx = win % nww, img = blockIdx.y;
    const int nt = ws * ws, N = H * W;
    const float* base = qkv + (int64_t)img * img_stride;
    _Float16* vh = lds + wave * 2 * HD * WLS;
    _Float16* vl = vh + HD * WLS;
    const float qmul = 0.17677669529663687f * 1.44269504088896340736f;

    // this lane's token in each of the two tiles: pixel offset, or -1 (outside the grid / beyond ws*ws)
    int pix[2];
    bool tok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int k = t * 32 + l31, y = wy * ws + k / ws, x = wx * ws + k % ws;
        tok[t] = k < nt;
        pix[t] = (tok[t] && y < H && x < W) ? y * W + x : -1;
    }
    f16x8 qh[2][2], ql[2][2], kh[2][2], kl[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float xq[8], xk[8], xv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ch = head * HD + 16 * ks + 8 * khalf + i;
                const bool in = pix[t] >= 0;
                const int64_t o = (int64_t)ch * N + (in ? pix[t] : 0);
                xq[i] = in ? sf::mul_rn(base[o], qmul) : 0.f;
                xk[i] = in ? base[o + (int64_t)C * N] : (tok[t] ? bias[C + ch] : 0.f);
                xv[i] = in ? base[o + (int64_t)2 * C * N] : (tok[t] ? bias[2 * C + ch] : 0.f);
            }
            split8(xq, qh[t][ks], ql[t][ks]);
            split8(xk, kh[t][ks], kl[t][ks]);
            f16x8 h8, l8;
            split8(xv, h8, l8);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int d = 16 * ks + 8 * khalf + i;
                vh[d * WLS + t * 32 + l31] = h8[i];
                if (kLo) vl[d * WLS + t * 32 + l31] = l8[i];
            }
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        // the image is this wave's own: no workgroup barrier
    __builtin_amdgcn_wave_barrier();
    // V^T fragments: k-step j (keys 16j .. 16j+15): lane (d = l31, khalf) holds keys 16j + (i & 3) + 8 (i >> 2) + 4 khalf
    f16x8 vfh[4], vfl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const h4 a = *reinterpret_cast<const h4*>(vh + l31 * WLS + 16 * j + 4 * khalf);
        const h4 b = *reinterpret_cast<const h4*>(vh + l31 * WLS + 16 * j + 4 * khalf + 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) { vfh[j][i] = a[i]; vfh[j][4 + i] = b[i]; }
        if (kLo) {
            const h4 c = *reinterpret_cast<const h4*>(vl + l31 * WLS + 16 * j + 4 * khalf);
            const h4 e = *reinterpret_cast<const h4*>(vl + l31 * WLS + 16 * j + 4 * khalf + 8);
#pragma unroll
            for (int i = 0; i < 4; ++i) { vfl[j][i] = c[i]; vfl[j][4 + i] = e[i]; }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {                                     // query tile
        if (t * 32 >= nt) break;
        f32x16 sc[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[kt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (kLo) {
                    sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[kt][ks], qh[t][ks], sc[kt], 0, 0, 0);
                    sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[kt][ks], ql[t][ks], sc[kt], 0, 0, 0);
                }
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[kt][ks], qh[t][ks], sc[kt], 0, 0, 0);
            }
        }
        float mx = -1.0e30f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool valid = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf < nt;
                sc[kt][r] = valid ? sc[kt][r] : -1.0e30f;
                mx = fmaxf(mx, sc[kt][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float psum = 0.f;
        f16x8 ph[4], pl[4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            if (kLo) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(sc[kt][r] - mx);
                    psum += p;
                    const _Float16 h = (_Float16)p;
                    ph[kt * 2 + (r >> 3)][r & 7] = h;
                    pl[kt * 2 + (r >> 3)][r & 7] = (_Float16)(p - (float)h);
                }
            } else {
                const f16x2 ones = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f16x2 pp;
                    pp[0] = (_Float16)__builtin_amdgcn_exp2f(sc[kt][r] - mx);
                    pp[1] = (_Float16)__builtin_amdgcn_exp2f(sc[kt][r + 1] - mx);
                    psum = __builtin_amdgcn_fdot2(pp, ones, psum, false);
                    ph[kt * 2 + (r >> 3)][r & 7] = pp[0];
                    ph[kt * 2 + (r >> 3)][(r & 7) + 1] = pp[1];
                }
            }
        }
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (kLo) {
                o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfl[j], ph[j], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfh[j], pl[j], o, 0, 0, 0);
            }
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfh[j], ph[j], o, 0, 0, 0);
        }
        const float inv = 1.0f / (psum + __shfl_xor(psum, 32, 64));
        if (pix[t] >= 0) store_attn_out(out, out_img_stride, out16, out16_img_stride, img, head, N, pix[t], khalf, o, inv);
    }
}

// The same with q, k, v read from fp16 k-octet planes [3C/8][N][8] (the qkv GEMM's c_f16 = 2 output): a lane's 8 dims of a
// token are ONE 16-byte load (12 loads per wave instead of 96 dwords, half the bytes, 112-byte runs per window row instead
// of 28).  One-product arithmetic (the operands ARE fp16); the 32^-0.5 log2(e) factor moves from q into the exponent.
__global__ __launch_bounds__(256) void window_attn_mfma16_kernel(const _Float16* qkv, int64_t img_stride, const float* bias, float* out,
                                                                 int64_t out_img_stride, _Float16* out16, int64_t out16_img_stride,
                                                                 int C, int H, int W, int ws, int nww) {
    __shared__ _Float16 lds[4 * HD * WLS];
    const int wave = threadIdx.x >> 6, head = blockIdx.z * 4 + wave, lane = threadIdx.x & 63, l31 = lane & 31, khalf = lane >> 5;
    const int win = sf::xcd_linear_id(blockIdx.x, gridDim.x);
    const int wy = win / nww, wx = win % nww, img = blockIdx.y;
    const int nt = ws * ws, N = H * W;
    const _Float16* base = qkv + (int64_t)img * img_stride;
    _Float16* vh = lds + wave * HD * WLS;
    const float qmul = 0.17677669529663687f * 1.44269504088896340736f;
    int pix[2];
    bool tok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int k = t * 32 + l31, y = wy * ws + k / ws, x = wx * ws + k % ws;
        tok[t] = k < nt;
        pix[t] = (tok[t] && y < H && x < W) ? y * W + x : -1;
    }
    f16x8 qh[2][2], kh[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int oct = head * 4 + 2 * ks + khalf;                   // channels head*32 + 16 ks + 8 khalf .. + 7
            f16x8 q8, k8, v8;
            if (pix[t] >= 0) {
                const _Float16* p = base + ((int64_t)oct * N + pix[t]) * 8;
                q8 = *reinterpret_cast<const f16x8*>(p);
                k8 = *reinterpret_cast<const f16x8*>(p + (int64_t)(C / 8) * N * 8);
                v8 = *reinterpret_cast<const f16x8*>(p + (int64_t)(C / 4) * N * 8);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    q8[i] = (_Float16)0.f;
                    k8[i] = tok[t] ? (_Float16)bias[C + oct * 8 + i] : (_Float16)0.f;
                    v8[i] = tok[t] ? (_Float16)bias[2 * C + oct * 8 + i] : (_Float16)0.f;
                }
            }
            qh[t][ks] = q8;
            kh[t][ks] = k8;
#pragma unroll
            for (int i = 0; i < 8; ++i) vh[(16 * ks + 8 * khalf + i) * WLS + t * 32 + l31] = v8[i];
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f16x8 vfh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const h4 a = *reinterpret_cast<const h4*>(vh + l31 * WLS + 16 * j + 4 * khalf);
        const h4 b = *reinterpret_cast<const h4*>(vh + l31 * WLS + 16 * j + 4 * khalf + 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) { vfh[j][i] = a[i]; vfh[j][4 + i] = b[i]; }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (t * 32 >= nt) break;
        f32x16 sc[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[kt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[kt][ks], qh[t][ks], sc[kt], 0, 0, 0);
        }
        float mx = -1.0e30f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool valid = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf < nt;
                sc[kt][r] = valid ? sc[kt][r] : -1.0e30f;
                mx = fmaxf(mx, sc[kt][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float off = -mx * qmul;
        float psum = 0.f;
        f16x8 ph[4];
        const f16x2 ones = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f16x2 pp;
                pp[0] = (_Float16)__builtin_amdgcn_exp2f(fmaf(sc[kt][r], qmul, off));
                pp[1] = (_Float16)__builtin_amdgcn_exp2f(fmaf(sc[kt][r + 1], qmul, off));
                psum = __builtin_amdgcn_fdot2(pp, ones, psum, false);
                ph[kt * 2 + (r >> 3)][r & 7] = pp[0];
                ph[kt * 2 + (r >> 3)][(r & 7) + 1] = pp[1];
            }
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfh[j], ph[j], o, 0, 0, 0);
        const float inv = 1.0f / (psum + __shfl_xor(psum, 32, 64));
        if (pix[t] >= 0) store_attn_out(out, out_img_stride, out16, out16_img_stride, img, head, N, pix[t], khalf, o, inv);
    }
}

// Four consecutive pixels per thread (W % 4 == 0): one 16-byte load per row plus the two neighbours; same taps in the same
// order as the scalar kernel (a tap outside the grid contributes fmaf(w, 0, acc) = acc).
__global__ __launch_bounds__(256) void dwconv3x3_res_vec4_kernel(const float* x, int64_t x_img_stride, const float* w, const float* b,
                                                                 float* y, int64_t y_img_stride, int C, int H, int W) {
    const int p = (blockIdx.x * 256 + threadIdx.x) * 4, c = blockIdx.y, img = blockIdx.z;
    if (p >= H * W) return;
    const int py = p / W, px = p % W;
    const float* xp = x + (int64_t)img * x_img_stride + (int64_t)c * H * W;
    float wt[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wt[i] = w[c * 9 + i];
    const float bias = b[c];
    float acc[4] = {bias, bias, bias, bias}, mid[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = py + dy;
        float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (yy >= 0 && yy < H) {
            const float4 m = *reinterpret_cast<const float4*>(xp + yy * W + px);
            v[1] = m.x; v[2] = m.y; v[3] = m.z; v[4] = m.w;
            if (px > 0) v[0] = xp[yy * W + px - 1];
            if (px + 4 < W) v[5] = xp[yy * W + px + 4];
        }
        if (dy == 0) { mid[0] = v[1]; mid[1] = v[2]; mid[2] = v[3]; mid[3] = v[4]; }
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(wt[(dy + 1) * 3 + dx], v[j + dx], acc[j]);
    }
    float4 o;
    o.x = mid[0] + acc[0]; o.y = mid[1] + acc[1]; o.z = mid[2] + acc[2]; o.w = mid[3] + acc[3];
    *reinterpret_cast<float4*>(y + (int64_t)img * y_img_stride + (int64_t)c * H * W + p) = o;
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

extern "C" int sf_window_attn_mfma(const void* qkv, int64_t qkv_img_stride, int qkv_koct, const float* qkv_bias, float* out,
                                   int64_t out_img_stride, void* out_koct, int64_t out_koct_img_stride, int n_img, int C,
                                   int heads, int H, int W, int ws, int precision, void* stream) {
    SF_REQUIRE(qkv && qkv_bias && (out || out_koct), "sf_window_attn_mfma: null pointer");
    SF_REQUIRE(!out_koct || ((reinterpret_cast<uintptr_t>(out_koct) & 15) == 0 && (out_koct_img_stride & 7) == 0),
               "sf_window_attn_mfma: out_koct must be 16-byte aligned with an image stride that is a multiple of 8 halves");
    SF_REQUIRE(n_img > 0 && H > 0 && W > 0 && n_img <= 65535, "sf_window_attn_mfma: bad dims");
    SF_REQUIRE(heads >= 4 && heads % 4 == 0 && C == heads * HD, "sf_window_attn_mfma: needs C = heads * 32, heads a multiple of 4 (got C=%d heads=%d)", C, heads);
    SF_REQUIRE(ws >= 2 && ws <= 7, "sf_window_attn_mfma: window size must be 2..7 (got %d)", ws);
    SF_REQUIRE(precision == SF_PRECISION_F16X3 || precision == SF_PRECISION_F16X2 || precision == SF_PRECISION_F16,
               "sf_window_attn_mfma: precision must be one of the split / fp16 classes (the exact fp32 core is sf_window_attn)");
    const int nwh = sf::ceil_div(H, ws), nww = sf::ceil_div(W, ws);
    const dim3 grid(nwh * nww, n_img, heads / 4);
    if (qkv_koct) {
        SF_REQUIRE(precision != SF_PRECISION_F16X3, "sf_window_attn_mfma: fp16 k-octet q / k / v are an input of the one-product classes only");
        SF_REQUIRE((reinterpret_cast<uintptr_t>(qkv) & 15) == 0 && (qkv_img_stride & 7) == 0,
                   "sf_window_attn_mfma: k-octet qkv must be 16-byte aligned with an image stride that is a multiple of 8 halves");
        hipLaunchKernelGGL(window_attn_mfma16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)qkv, qkv_img_stride,
                           qkv_bias, out, out_img_stride, (_Float16*)out_koct, out_koct_img_stride, C, H, W, ws, nww);
    } else if (precision == SF_PRECISION_F16X3) {
        hipLaunchKernelGGL(window_attn_mfma_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)qkv, qkv_img_stride,
                           qkv_bias, out, out_img_stride, (_Float16*)out_koct, out_koct_img_stride, C, H, W, ws, nww);
    } else {
        hipLaunchKernelGGL(window_attn_mfma_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)qkv, qkv_img_stride,
                           qkv_bias, out, out_img_stride, (_Float16*)out_koct, out_koct_img_stride, C, H, W, ws, nww);
    }
    return sf::check_launch("sf_window_attn_mfma");
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

extern "C" int64_t sf_subsample_attn_ws_bytes(int n_img, int heads, int M) {
    if (n_img <= 0 || heads <= 0 || M <= 0) return 0;
    return (int64_t)n_img * heads * sf::ceil_div(M, 32) * (SPARTS * 1024);
}

extern "C" int sf_subsample_attn_mfma(const float* q, int64_t q_img_stride, const float* kv, int64_t kv_img_stride, float* out,
                                      int64_t out_img_stride, void* out_koct, int64_t out_koct_img_stride, int n_img, int C,
                                      int heads, int N, int M, void* ws, int64_t ws_bytes, int precision, void* stream) {
    SF_REQUIRE(q && kv && (out || out_koct) && ws, "sf_subsample_attn_mfma: null pointer");
    SF_REQUIRE(!out_koct || ((reinterpret_cast<uintptr_t>(out_koct) & 15) == 0 && (out_koct_img_stride & 7) == 0),
               "sf_subsample_attn_mfma: out_koct must be 16-byte aligned with an image stride that is a multiple of 8 halves");
    SF_REQUIRE(n_img > 0 && N > 0 && M > 0 && n_img <= 65535 && heads <= 65535, "sf_subsample_attn_mfma: bad dims");
    SF_REQUIRE(heads >= 1 && C == heads * HD, "sf_subsample_attn_mfma: needs C = heads * 32 (got C=%d heads=%d)", C, heads);
    SF_REQUIRE(precision == SF_PRECISION_F16X3 || precision == SF_PRECISION_F16X2 || precision == SF_PRECISION_F16,
               "sf_subsample_attn_mfma: precision must be one of the split / fp16 classes (the exact fp32 core is sf_subsample_attn)");
    SF_REQUIRE(ws_bytes >= sf_subsample_attn_ws_bytes(n_img, heads, M) && (reinterpret_cast<uintptr_t>(ws) & 15) == 0,
               "sf_subsample_attn_mfma: workspace too small or misaligned");
    const int tiles = sf::ceil_div(M, 32);
    hipLaunchKernelGGL(subsample_pack_kv_kernel, dim3(tiles, heads, n_img), dim3(256), 0, (hipStream_t)stream, kv, kv_img_stride,
                       (char*)ws, C, M, tiles);
    const dim3 grid(sf::ceil_div(N, 4 * 32 * SQT), heads, n_img);
    if (precision == SF_PRECISION_F16X3)
        hipLaunchKernelGGL(subsample_attn_mfma_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, q, q_img_stride, (const char*)ws,
                           out, out_img_stride, (_Float16*)out_koct, out_koct_img_stride, N, M, tiles);
    else
        hipLaunchKernelGGL(subsample_attn_mfma_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, q, q_img_stride, (const char*)ws,
                           out, out_img_stride, (_Float16*)out_koct, out_koct_img_stride, N, M, tiles);
    return sf::check_launch("sf_subsample_attn_mfma");
}

extern "C" int sf_dwconv3x3_res(const float* x, int64_t x_img_stride, const float* w, const float* b, float* y,
                                int64_t y_img_stride, int n_img, int C, int H, int W, void* stream) {
    SF_REQUIRE(x && w && b && y, "sf_dwconv3x3_res: null pointer");
    SF_REQUIRE(n_img > 0 && C > 0 && H > 0 && W > 0 && n_img <= 65535 && C <= 65535, "sf_dwconv3x3_res: bad dims");
    const bool vec4 = W % 4 == 0 && (x_img_stride & 3) == 0 && (y_img_stride & 3) == 0 &&
                      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    if (vec4)
        hipLaunchKernelGGL(dwconv3x3_res_vec4_kernel, dim3(sf::ceil_div(H * W / 4, 256), C, n_img), dim3(256), 0, (hipStream_t)stream,
                           x, x_img_stride, w, b, y, y_img_stride, C, H, W);
    else
        hipLaunchKernelGGL(dwconv3x3_res_kernel, dim3(sf::ceil_div(H * W, 256), C, n_img), dim3(256), 0, (hipStream_t)stream, x,
                           x_img_stride, w, b, y, y_img_stride, C, H, W);
    return sf::check_launch("sf_dwconv3x3_res");
}
