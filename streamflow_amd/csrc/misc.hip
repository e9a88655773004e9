// Error plumbing + the small HBM-bound kernels of the StreamFlow update loop:
// coords grid, context split, flow bookkeeping, row softmax, channel LayerNorm, per-pixel temporal
// attention, convex upsampling.  All are coalesced along the pixel axis (P contiguous).
#include "sf_common.h"

namespace sf {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace sf

extern "C" int sf_version(void) { return SF_VERSION; }
extern "C" const char* sf_last_error(void) { return sf::err_buf(); }

namespace {

constexpr int kBlock = 256;

// ---- coords_grid ----------------------------------------------------------------------------------
__global__ void coords_grid_kernel(float* out, int batch, int ht, int wd) {
    const int P = ht * wd;
    const int64_t total = (int64_t)batch * 2 * P;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % P);
        const int c = (int)((i / P) & 1);
        out[i] = (float)(c == 0 ? p % wd : p / wd);
    }
}

// ---- context split: nets = tanh(cnets[:, :hdim]), inps = relu(cnets[:, hdim:]) ------------------------
__global__ void context_split_kernel(const float* cnets, float* nets, int64_t nets_stride, float* inps,
                                     int64_t inps_stride, int n_img, int hdim, int P) {
    const int64_t per_img = (int64_t)2 * hdim * P;
    const int64_t total = per_img * n_img;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int img = (int)(i / per_img);
        const int64_t rem = i - (int64_t)img * per_img;
        const int c = (int)(rem / P), p = (int)(rem % P);
        const float v = cnets[i];
        if (c < hdim) nets[img * nets_stride + (int64_t)c * P + p] = tanhf(v);
        else inps[img * inps_stride + (int64_t)(c - hdim) * P + p] = fmaxf(v, 0.f);
    }
}

// ---- coords1 += delta; flow = coords1 - grid ---------------------------------------------------------
__global__ void flow_update_kernel(float* coords1, const float* delta, float* fa, int64_t fa_stride, float* fb,
                                   int64_t fb_stride, _Float16* fk, int64_t fk_stride, int fk_row, int n_img, int h, int w) {
    const int P = h * w;
    const int64_t total = (int64_t)n_img * 2 * P;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % P);
        const int c = (int)((i / P) & 1);
        const int img = (int)(i / (2 * (int64_t)P));
        float v = coords1[i];
        if (delta) {
            v += delta[i];
            coords1[i] = v;
        }
        const float f = v - (float)(c == 0 ? p % w : p / w);
        if (fa) fa[img * fa_stride + (int64_t)c * P + p] = f;
        if (fb) fb[img * fb_stride + (int64_t)c * P + p] = f;
        // k-octet planes: rows fk_row (x) and fk_row + 1 (y) of the fp16 copy that shadows flow_b's tensor
        if (fk) fk[img * fk_stride + ((int64_t)((fk_row + c) >> 3) * P + p) * 8 + ((fk_row + c) & 7)] = (_Float16)f;
    }
}

// ---- row softmax (one workgroup per row) -------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One workgroup per row; the row lives in registers (up to kSmMax values per thread), so it is read once and
// written once (the first version made three passes over memory: 1.4x the reads and 2x the writes).
constexpr int kSmMax = 32;              // values per thread: 256 threads cover rows up to 8192 columns, 1024 up to 32768
template <int TPB>
__global__ __launch_bounds__(TPB) void softmax_rows_kernel(float* x, int cols, _Float16* y16) {
    __shared__ float red[TPB / 64];
    float* row = x + (int64_t)blockIdx.x * cols;
    _Float16* row16 = y16 ? y16 + (int64_t)blockIdx.x * cols : nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float v[kSmMax];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < kSmMax; ++j) {
        const int c = tid + j * TPB;
        v[j] = (c < cols) ? row[c] : -INFINITY;
        m = fmaxf(m, v[j]);
    }
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int i = 1; i < TPB / 64; ++i) m = fmaxf(m, red[i]);
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < kSmMax; ++j) {
        v[j] = expf(v[j] - m);          // exp(-inf) = 0 for the padding
        s += v[j];
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    // fixed summation tree over the waves (pairs, then pairs of pairs ...): the same value in every thread
    float part[TPB / 64];
#pragma unroll
    for (int i = 0; i < TPB / 64; ++i) part[i] = red[i];
#pragma unroll
    for (int w = 1; w < TPB / 64; w *= 2)
#pragma unroll
        for (int i = 0; i + w < TPB / 64; i += 2 * w) part[i] += part[i + w];
    const float inv = 1.0f / part[0];
#pragma unroll
    for (int j = 0; j < kSmMax; ++j) {
        const int c = tid + j * TPB;
        if (c < cols) {
            if (row16) row16[c] = (_Float16)(v[j] * inv);
            else row[c] = v[j] * inv;
        }
    }
}

// fallback for very long rows: three passes over memory
__global__ __launch_bounds__(kBlock) void softmax_rows_long_kernel(float* x, int cols, _Float16* y16) {
    __shared__ float red[kBlock / 64];
    float* row = x + (int64_t)blockIdx.x * cols;
    _Float16* row16 = y16 ? y16 + (int64_t)blockIdx.x * cols : nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m = -INFINITY;
    for (int c = tid; c < cols; c += kBlock) m = fmaxf(m, row[c]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = tid; c < cols; c += kBlock) {
        const float e = expf(row[c] - m);
        row[c] = e;
        s += e;
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    const float inv = 1.0f / s;
    for (int c = tid; c < cols; c += kBlock) {
        if (row16) row16[c] = (_Float16)(row[c] * inv);
        else row[c] *= inv;
    }
}

// ---- LayerNorm over channels (channel-major planes) -------------------------------------------------
// generic fallback: one thread per pixel, three passes over the channels
__global__ __launch_bounds__(kBlock) void layernorm_cm_kernel(const float* x, int64_t xs, const float* gamma,
                                                               const float* beta, float* y, int64_t ys, int C, int P,
                                                               float eps) {
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= P) return;
    const float* xp = x + (int64_t)blockIdx.y * xs + p;
    float* yp = y + (int64_t)blockIdx.y * ys + p;
    float mean = 0.f;
    for (int c = 0; c < C; ++c) mean += xp[(int64_t)c * P];
    mean /= (float)C;
    float var = 0.f;
    for (int c = 0; c < C; ++c) {
        const float d = xp[(int64_t)c * P] - mean;
        var += d * d;
    }
    const float rstd = 1.0f / sqrtf(var / (float)C + eps);
    for (int c = 0; c < C; ++c) yp[(int64_t)c * P] = (xp[(int64_t)c * P] - mean) * rstd * gamma[c] + beta[c];
}

// fast path: a workgroup owns 64 consecutive pixels; the C channels are split over the 4 waves, each thread
// keeps its CPT = C/4 values in registers (x is read exactly once, 256-byte coalesced rows), the two
// reductions (mean, then centred variance: same two-pass arithmetic as nn.LayerNorm) go through LDS.
constexpr int kLnPix = 64;
template <int CPT>
__global__ __launch_bounds__(kBlock) void layernorm_cm_split_kernel(const float* x, int64_t xs, const float* gamma,
                                                                     const float* beta, float* y, int64_t ys,
                                                                     _Float16* y16, int64_t y16s, int P, float eps) {
    __shared__ float red[4][kLnPix];
    constexpr int C = 4 * CPT;
    const int px = threadIdx.x & (kLnPix - 1), cg = threadIdx.x >> 6;
    const int p = blockIdx.x * kLnPix + px;
    const bool ok = p < P;
    const int pc = ok ? p : P - 1;
    const float* xp = x + (int64_t)blockIdx.y * xs + (int64_t)cg * CPT * P + pc;
    float v[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) v[c] = xp[(int64_t)c * P];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CPT; ++c) s += v[c];
    red[cg][px] = s;
    __syncthreads();
    const float mean = ((red[0][px] + red[1][px]) + (red[2][px] + red[3][px])) * (1.0f / (float)C);
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const float d = v[c] - mean;
        q = fmaf(d, d, q);
    }
    red[cg][px] = q;
    __syncthreads();
    const float var = ((red[0][px] + red[1][px]) + (red[2][px] + red[3][px])) * (1.0f / (float)C);
    const float rstd = 1.0f / sqrtf(var + eps);
    if (!ok) return;
    const float* gp = gamma + cg * CPT;
    const float* bp = beta + cg * CPT;
#pragma unroll
    for (int c = 0; c < CPT; ++c) v[c] = (v[c] - mean) * rstd * gp[c] + bp[c];
    if (y) {
        float* yp = y + (int64_t)blockIdx.y * ys + (int64_t)cg * CPT * P + p;
#pragma unroll
        for (int c = 0; c < CPT; ++c) yp[(int64_t)c * P] = v[c];
    }
    if (y16) {      // fp16 k-octet planes (SF_LAYOUT_F16_KOCT): a thread owns CPT / 8 whole octets of its pixel, 16 bytes each
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        _Float16* yp = y16 + (int64_t)blockIdx.y * y16s + ((int64_t)cg * (CPT / 8) * P + p) * 8;
#pragma unroll
        for (int o = 0; o < CPT / 8; ++o) {
            h8 hv;
#pragma unroll
            for (int i = 0; i < 8; ++i) hv[i] = (_Float16)v[o * 8 + i];
            *reinterpret_cast<h8*>(yp + (int64_t)o * P * 8) = hv;
        }
    }
}

// ---- attention over the TT tokens of one pixel ---------------------------------------------------------
// A workgroup owns 64 consecutive pixels of one clip; the C channels are split over the 4 waves.  Each thread
// accumulates the TT x TT partial scores of its channel slice, the slices are summed through LDS, every
// thread then holds the full softmax weights of its pixel and produces its slice of the output channels.
// All global accesses are 256-byte rows (64 lanes x consecutive pixels of one channel plane).
constexpr int kTaPix = 64;
// TIn = float (fp32 planes) or _Float16 (fp16 rows: the qkv GEMM's c_f16 = 1 hand-over of the config-2 presets).
template <int TT, typename TIn>
__global__ __launch_bounds__(kBlock) void temporal_attn_kernel(const TIn* qkv, float* out, _Float16* out16, int C, int P) {
    __shared__ float red[4][TT * TT][kTaPix];
    const int px = threadIdx.x & (kTaPix - 1), cg = threadIdx.x >> 6;
    const int p = blockIdx.x * kTaPix + px;
    const bool ok = p < P;
    const int pc = ok ? p : P - 1;
    const int b = blockIdx.y;
    const int cpt = C / 4, c0 = cg * cpt;
    const int64_t img = (int64_t)3 * C * P;      // one token image of qkv
    const TIn* base = qkv + (int64_t)b * TT * img + pc;
    float s[TT][TT];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int u = 0; u < TT; ++u) s[t][u] = 0.f;
#pragma unroll 4
    for (int c = c0; c < c0 + cpt; ++c) {
        float q[TT], k[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            q[t] = (float)base[t * img + (int64_t)c * P];
            k[t] = (float)base[t * img + (int64_t)(C + c) * P];
        }
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int u = 0; u < TT; ++u) s[t][u] = fmaf(q[t], k[u], s[t][u]);
    }
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int u = 0; u < TT; ++u) red[cg][t * TT + u][px] = s[t][u];
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)C);
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        float m = -INFINITY;
#pragma unroll
        for (int u = 0; u < TT; ++u) {
            const int e = t * TT + u;
            s[t][u] = ((red[0][e][px] + red[1][e][px]) + (red[2][e][px] + red[3][e][px])) * scale;
            m = fmaxf(m, s[t][u]);
        }
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < TT; ++u) {
            s[t][u] = expf(s[t][u] - m);
            sum += s[t][u];
        }
        const float inv = 1.0f / sum;
#pragma unroll
        for (int u = 0; u < TT; ++u) s[t][u] *= inv;
    }
    if (!ok) return;
    if (out16) {    // fp16 k-octet planes (SF_LAYOUT_F16_KOCT, C % 32 == 0): 8 channels of a token = one 16-byte store
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        _Float16* ob = out16 + (int64_t)b * TT * C * P + (int64_t)p * 8;
        for (int c8 = c0; c8 < c0 + cpt; c8 += 8) {
            h8 hv[TT];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v[TT];
#pragma unroll
                for (int u = 0; u < TT; ++u) v[u] = (float)base[u * img + (int64_t)(2 * C + c8 + i) * P];
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    float o = 0.f;
#pragma unroll
                    for (int u = 0; u < TT; ++u) o = fmaf(s[t][u], v[u], o);
                    hv[t][i] = (_Float16)o;
                }
            }
#pragma unroll
            for (int t = 0; t < TT; ++t) *reinterpret_cast<h8*>(ob + (int64_t)t * C * P + (int64_t)(c8 / 8) * P * 8) = hv[t];
        }
        if (!out) return;
    }
    float* obase = out + (int64_t)b * TT * C * P + p;
#pragma unroll 4
    for (int c = c0; c < c0 + cpt; ++c) {
        float v[TT];
#pragma unroll
        for (int u = 0; u < TT; ++u) v[u] = (float)base[u * img + (int64_t)(2 * C + c) * P];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            float o = 0.f;
#pragma unroll
            for (int u = 0; u < TT; ++u) o = fmaf(s[t][u], v[u], o);
            obase[(int64_t)t * C * P + (int64_t)c * P] = o;
        }
    }
}

// ---- convex 8x upsampling ---------------------------------------------------------------------------
// One workgroup per (image, low-res row y, 32-pixel x segment).  Thread = (sub-pixel s = i*8+j, 4 of the
// 32 pixels): mask reads are coalesced along x inside one channel plane; the 8x8 outputs of the segment
// are staged in LDS and written as full 256-float rows.
constexpr int kUpSeg = 32;
__global__ __launch_bounds__(kBlock) void upsample_kernel(const float* flow, const float* mask, float* out, int h,
                                                          int w) {
    __shared__ float nb[2][9][kUpSeg];                 // 8*flow at the 3x3 neighbours
    __shared__ float stage[2][8][kUpSeg * 8 + 1];
    const int P = h * w;
    const int x0 = blockIdx.x * kUpSeg, y = blockIdx.y, n = blockIdx.z;
    const int tid = threadIdx.x;
    const float* fl = flow + (int64_t)n * 2 * P;
    for (int i = tid; i < 2 * 9 * kUpSeg; i += kBlock) {
        const int xl = i % kUpSeg, k = (i / kUpSeg) % 9, c = i / (9 * kUpSeg);
        const int yy = y + k / 3 - 1, xx = x0 + xl + k % 3 - 1;
        float v = 0.f;
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) v = 8.0f * fl[(int64_t)c * P + yy * w + xx];
        nb[c][k][xl] = v;
    }
    __syncthreads();
    const float* mk = mask + (int64_t)n * 576 * P + (int64_t)y * w;
    // 64 sub-pixels x 32 pixels = 2048 items, 8 per thread; consecutive threads -> consecutive x
    for (int it = tid; it < 64 * kUpSeg; it += kBlock) {
        const int xl = it % kUpSeg, s = it / kUpSeg;
        const int x = x0 + xl;
        if (x < w) {
            float lg[9], m = -INFINITY;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                lg[k] = mk[(int64_t)(k * 64 + s) * P + x];
                m = fmaxf(m, lg[k]);
            }
            float sum = 0.f, ax = 0.f, ay = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float e = expf(lg[k] - m);
                sum += e;
                ax = fmaf(e, nb[0][k][xl], ax);
                ay = fmaf(e, nb[1][k][xl], ay);
            }
            const float inv = 1.0f / sum;
            const int i = s >> 3, j = s & 7;
            stage[0][i][xl * 8 + j] = ax * inv;
            stage[1][i][xl * 8 + j] = ay * inv;
        }
    }
    __syncthreads();
    const int W8 = w * 8;
    const int valid = min(kUpSeg, w - x0) * 8;
    for (int it = tid; it < 2 * 8 * kUpSeg * 8; it += kBlock) {
        const int col = it % (kUpSeg * 8), i = (it / (kUpSeg * 8)) % 8, c = it / (kUpSeg * 64);
        if (col < valid)
            out[((int64_t)n * 2 + c) * (int64_t)(h * 8) * W8 + (int64_t)(y * 8 + i) * W8 + x0 * 8 + col] =
                stage[c][i][col];
    }
}

// ---- split-K combine: out = R + gamma * sum_s partial_s ------------------------------------------------
__global__ __launch_bounds__(kBlock) void splitk_combine_kernel(const float* partial, int64_t split_stride, int ks,
                                                                 int64_t part_img_stride, const float* R,
                                                                 int64_t r_img_stride, const float* gamma, float* out,
                                                                 int64_t out_img_stride, int64_t per_img4) {
    const int img = blockIdx.y;
    const float g = gamma[0];
    const float4* p4 = reinterpret_cast<const float4*>(partial + img * part_img_stride);
    const float4* r4 = reinterpret_cast<const float4*>(R + img * r_img_stride);
    float4* o4 = reinterpret_cast<float4*>(out + img * out_img_stride);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < per_img4; i += (int64_t)gridDim.x * kBlock) {
        float4 a = p4[i];
        for (int s = 1; s < ks; ++s) {
            const float4 b = p4[i + s * (split_stride / 4)];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        const float4 r = r4[i];
        o4[i] = make_float4(r.x + g * a.x, r.y + g * a.y, r.z + g * a.z, r.w + g * a.w);
    }
}

inline int grid_for(int64_t total) {
    int64_t b = (total + kBlock - 1) / kBlock;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// ---- forward_interpolate (core/utils/utils.py:34-62): nearest valid forward-warped source point per grid pixel ------
// Exact nearest neighbour = argmin over all N source points of the squared float64 distance (what scipy's
// griddata(method='nearest') returns through a k-d tree).  N^2 = 5e7 distance evaluations per Sintel-size flow:
// brute force is ~100 us on the GPU and removes the reference's D2H -> scipy -> H2D round trip between clips.
// Source points are staged 256 at a time in LDS (broadcast reads); products and the sum are rounded separately
// (no FMA contraction) so that the comparison sees the same float64 values as the CPU restatement; ties go to the
// lowest source index.
__global__ __launch_bounds__(kBlock) void forward_interp_kernel(const float* __restrict__ flow, float* __restrict__ out,
                                                                int h, int w) {
    __shared__ double sx[kBlock], sy[kBlock];
    const int N = h * w, tid = threadIdx.x;
    const float* f = flow + (int64_t)blockIdx.y * 2 * N;
    const int t = blockIdx.x * kBlock + tid;
    const double gx = (double)(t % w), gy = (double)(t / w);
    double best = INFINITY;
    int bi = -1;
    for (int base = 0; base < N; base += kBlock) {
        const int s = base + tid;
        double x1 = INFINITY, y1 = INFINITY;                 // dropped point: distance inf never beats `best`
        if (s < N) {
            const double px = (double)(s % w) + (double)f[s], py = (double)(s / w) + (double)f[N + s];
            if (px > 0.0 && px < (double)w && py > 0.0 && py < (double)h) { x1 = px; y1 = py; }
        }
        __syncthreads();
        sx[tid] = x1;
        sy[tid] = y1;
        __syncthreads();
        const int cnt = min(kBlock, N - base);
        for (int j = 0; j < cnt; ++j) {
            const double d = sx[j] - gx, e = sy[j] - gy;
            const double d2 = __dadd_rn(__dmul_rn(d, d), __dmul_rn(e, e));
            if (d2 < best) { best = d2; bi = base + j; }
        }
    }
    if (t < N) {
        float* o = out + (int64_t)blockIdx.y * 2 * N;
        o[t] = bi >= 0 ? f[bi] : 0.f;
        o[N + t] = bi >= 0 ? f[N + bi] : 0.f;
    }
}

}  // namespace

extern "C" int sf_coords_grid(float* out, int batch, int ht, int wd, void* stream) {
    SF_REQUIRE(out && batch > 0 && ht > 0 && wd > 0, "sf_coords_grid: bad args");
    hipLaunchKernelGGL(coords_grid_kernel, dim3(grid_for((int64_t)batch * 2 * ht * wd)), dim3(kBlock), 0,
                       (hipStream_t)stream, out, batch, ht, wd);
    return sf::check_launch("sf_coords_grid");
}


namespace {
// fp32 channel-major planes -> fp16 k-octet planes (SF_LAYOUT_F16_KOCT): one thread = one (octet, pixel) = eight
// coalesced dword loads (one per row) and one 16-byte store.  Rows past `rows` inside a last, partial octet are NOT
// written (they may belong to another producer; the caller zero-initialises the planes once).
__global__ __launch_bounds__(256) void pack_koct_kernel(const float* __restrict__ x, int64_t x_img_stride, int rows, int P,
                                                         _Float16* __restrict__ y, int64_t y_img_stride) {
    const int p = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y, img = blockIdx.z;
    if (p >= P) return;
    const float* xp = x + img * x_img_stride + (int64_t)o * 8 * P + p;
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    h8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (o * 8 + i < rows) ? (_Float16)xp[(int64_t)i * P] : (_Float16)0.f;
    _Float16* yp = y + img * y_img_stride + ((int64_t)o * P + p) * 8;
    if (o * 8 + 8 <= rows) {
        *reinterpret_cast<h8*>(yp) = v;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (o * 8 + i < rows) yp[i] = v[i];
    }
}
}  // namespace

extern "C" int sf_pack_koct(const float* x, int64_t x_img_stride, int n_img, int rows, int P, void* y,
                            int64_t y_img_stride, void* stream) {
    SF_REQUIRE(x && y && n_img > 0 && rows > 0 && P > 0, "sf_pack_koct: bad args");
    SF_REQUIRE((reinterpret_cast<uintptr_t>(y) & 15) == 0 && (y_img_stride & 7) == 0, "sf_pack_koct: y must be 16-byte aligned");
    const int noct = (rows + 7) / 8;
    SF_REQUIRE(noct <= 65535 && n_img <= 65535, "sf_pack_koct: grid too large");
    hipLaunchKernelGGL(pack_koct_kernel, dim3((P + 255) / 256, noct, n_img), dim3(256), 0, (hipStream_t)stream, x,
                       x_img_stride, rows, P, (_Float16*)y, y_img_stride);
    return sf::check_launch("sf_pack_koct");
}

namespace {
// One wave that watches the clock: lane 0 reads the shader-cycle counter (s_memtime) and the constant 100 MHz counter
// (s_memrealtime) until `spin_us` have passed, sleeping in between.
__global__ __launch_bounds__(64) void clock_probe_kernel(long long* out, long long ticks) {
    if (threadIdx.x != 0) return;
    const long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    long long r = r0;
    while (r - r0 < ticks) {
        __builtin_amdgcn_s_sleep(64);
        r = __builtin_amdgcn_s_memrealtime();
    }
    out[0] = __builtin_readcyclecounter() - c0;
    out[1] = r - r0;
}
}  // namespace

extern "C" int sf_clock_probe(int64_t* out, int spin_us, void* stream) {
    SF_REQUIRE(out && spin_us > 0 && spin_us <= 2000000, "sf_clock_probe: bad args");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<long long*>(out),
                       (long long)spin_us * 100);
    return sf::check_launch("sf_clock_probe");
}

extern "C" int sf_context_split(const float* cnets, float* nets, int64_t nets_img_stride, float* inps,
                                int64_t inps_img_stride, int n_img, int hdim, int P, void* stream) {
    SF_REQUIRE(cnets && nets && inps && n_img > 0 && hdim > 0 && P > 0, "sf_context_split: bad args");
    hipLaunchKernelGGL(context_split_kernel, dim3(grid_for((int64_t)n_img * 2 * hdim * P)), dim3(kBlock), 0,
                       (hipStream_t)stream, cnets, nets, nets_img_stride, inps, inps_img_stride, n_img, hdim, P);
    return sf::check_launch("sf_context_split");
}

extern "C" int sf_flow_update(float* coords1, const float* delta, float* flow_a, int64_t flow_a_img_stride,
                              float* flow_b, int64_t flow_b_img_stride, void* flow_koct, int64_t flow_koct_img_stride,
                              int flow_koct_row, int n_img, int h, int w, void* stream) {
    SF_REQUIRE(coords1 && n_img > 0 && h > 0 && w > 0, "sf_flow_update: bad args");
    SF_REQUIRE(!flow_koct || flow_koct_row >= 0, "sf_flow_update: bad k-octet row");
    hipLaunchKernelGGL(flow_update_kernel, dim3(grid_for((int64_t)n_img * 2 * h * w)), dim3(kBlock), 0,
                       (hipStream_t)stream, coords1, delta, flow_a, flow_a_img_stride, flow_b, flow_b_img_stride,
                       static_cast<_Float16*>(flow_koct), flow_koct_img_stride, flow_koct_row, n_img, h, w);
    return sf::check_launch("sf_flow_update");
}

extern "C" int sf_softmax_rows(float* x, int64_t rows, int cols, void* out_f16, void* stream) {
    SF_REQUIRE(x && rows > 0 && cols > 0, "sf_softmax_rows: bad args");
    SF_REQUIRE(rows <= 0x7fffffffLL, "sf_softmax_rows: too many rows");
    if (cols <= kSmMax * 256)
        hipLaunchKernelGGL(softmax_rows_kernel<256>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, cols, (_Float16*)out_f16);
    else if (cols <= kSmMax * 1024)      // high-resolution rows (1080p: 32640 columns) still make one pass
        hipLaunchKernelGGL(softmax_rows_kernel<1024>, dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream, x, cols, (_Float16*)out_f16);
    else
        hipLaunchKernelGGL(softmax_rows_long_kernel, dim3((unsigned)rows), dim3(kBlock), 0, (hipStream_t)stream, x, cols, (_Float16*)out_f16);
    return sf::check_launch("sf_softmax_rows");
}

extern "C" int sf_layernorm_cm(const float* x, int64_t x_img_stride, const float* gamma, const float* beta, float* y,
                               int64_t y_img_stride, void* y_koct, int64_t y_koct_img_stride, int n_img, int C, int P,
                               float eps, void* stream) {
    SF_REQUIRE(x && gamma && beta && (y || y_koct) && n_img > 0 && C > 0 && P > 0, "sf_layernorm_cm: bad args");
    SF_REQUIRE(!y_koct || ((C == 128 || C == 256) && (reinterpret_cast<uintptr_t>(y_koct) & 15) == 0 &&
                           (y_koct_img_stride & 7) == 0),
               "sf_layernorm_cm: the k-octet output needs C = 128 or 256, a 16-byte aligned y_koct, stride %% 8 == 0");
    _Float16* y16 = static_cast<_Float16*>(y_koct);
    if (C == 128)
        hipLaunchKernelGGL(layernorm_cm_split_kernel<32>, dim3(sf::ceil_div(P, kLnPix), n_img), dim3(kBlock), 0,
                           (hipStream_t)stream, x, x_img_stride, gamma, beta, y, y_img_stride, y16, y_koct_img_stride, P, eps);
    else if (C == 256)                                       // second stage of the Twins_CSC encoder
        hipLaunchKernelGGL(layernorm_cm_split_kernel<64>, dim3(sf::ceil_div(P, kLnPix), n_img), dim3(kBlock), 0,
                           (hipStream_t)stream, x, x_img_stride, gamma, beta, y, y_img_stride, y16, y_koct_img_stride, P, eps);
    else
        hipLaunchKernelGGL(layernorm_cm_kernel, dim3(sf::ceil_div(P, kBlock), n_img), dim3(kBlock), 0,
                           (hipStream_t)stream, x, x_img_stride, gamma, beta, y, y_img_stride, C, P, eps);
    return sf::check_launch("sf_layernorm_cm");
}

template <typename TIn>
static int temporal_attn_launch(const TIn* qkv, float* out, void* out_koct, int B, int TT, int C, int P, void* stream) {
    SF_REQUIRE(qkv && (out || out_koct) && B > 0 && C > 0 && P > 0, "sf_temporal_attn: bad args");
    SF_REQUIRE(C % 4 == 0, "sf_temporal_attn: C must be a multiple of 4");
    SF_REQUIRE(!out_koct || (C % 32 == 0 && (reinterpret_cast<uintptr_t>(out_koct) & 15) == 0),
               "sf_temporal_attn: the k-octet output needs C %% 32 == 0 and a 16-byte aligned out_koct");
    _Float16* out16 = static_cast<_Float16*>(out_koct);
    dim3 grid(sf::ceil_div(P, kTaPix), B), block(kBlock);
    hipStream_t st = (hipStream_t)stream;
    switch (TT) {
        case 1: hipLaunchKernelGGL((temporal_attn_kernel<1, TIn>), grid, block, 0, st, qkv, out, out16, C, P); break;
        case 2: hipLaunchKernelGGL((temporal_attn_kernel<2, TIn>), grid, block, 0, st, qkv, out, out16, C, P); break;
        case 3: hipLaunchKernelGGL((temporal_attn_kernel<3, TIn>), grid, block, 0, st, qkv, out, out16, C, P); break;
        case 4: hipLaunchKernelGGL((temporal_attn_kernel<4, TIn>), grid, block, 0, st, qkv, out, out16, C, P); break;
        case 5: hipLaunchKernelGGL((temporal_attn_kernel<5, TIn>), grid, block, 0, st, qkv, out, out16, C, P); break;
        case 6: hipLaunchKernelGGL((temporal_attn_kernel<6, TIn>), grid, block, 0, st, qkv, out, out16, C, P); break;
        case 7: hipLaunchKernelGGL((temporal_attn_kernel<7, TIn>), grid, block, 0, st, qkv, out, out16, C, P); break;
        default: return sf::fail(SF_ERR_UNSUPPORTED, "sf_temporal_attn: T-1=%d tokens not supported (1..7)", TT);
    }
    return sf::check_launch("sf_temporal_attn");
}

extern "C" int sf_temporal_attn(const float* qkv, float* out, void* out_koct, int B, int TT, int C, int P, void* stream) {
    return temporal_attn_launch<float>(qkv, out, out_koct, B, TT, C, P, stream);
}

// qkv as fp16 ROWS [img][3 C][P] (the qkv GEMM's c_f16 = 1 output in the config-2 presets): q, k, v enter the scores and
// the weighted sum as the fp16 values they are; scores, softmax and accumulation in fp32 as above.
extern "C" int sf_temporal_attn_f16in(const void* qkv_f16, float* out, void* out_koct, int B, int TT, int C, int P, void* stream) {
    return temporal_attn_launch<_Float16>(static_cast<const _Float16*>(qkv_f16), out, out_koct, B, TT, C, P, stream);
}

extern "C" int sf_upsample_flow(const float* flow, const float* mask, float* out, int n, int h, int w, void* stream) {
    SF_REQUIRE(flow && mask && out && n > 0 && h > 0 && w > 0, "sf_upsample_flow: bad args");
    hipLaunchKernelGGL(upsample_kernel, dim3(sf::ceil_div(w, kUpSeg), h, n), dim3(kBlock), 0, (hipStream_t)stream,
                       flow, mask, out, h, w);
    return sf::check_launch("sf_upsample_flow");
}

extern "C" int sf_splitk_combine(const float* partial, int64_t split_stride, int k_splits, int64_t part_img_stride,
                                 const float* R, int64_t r_img_stride, const float* gamma, float* out,
                                 int64_t out_img_stride, int n_img, int64_t floats_per_img, void* stream) {
    SF_REQUIRE(partial && R && gamma && out && k_splits >= 1 && n_img > 0 && floats_per_img > 0, "sf_splitk_combine: bad args");
    SF_REQUIRE(((floats_per_img | split_stride | part_img_stride | r_img_stride | out_img_stride) & 3) == 0 &&
               ((reinterpret_cast<uintptr_t>(partial) | reinterpret_cast<uintptr_t>(R) | reinterpret_cast<uintptr_t>(out)) & 15) == 0,
               "sf_splitk_combine: pointers/strides must be 16-byte aligned");
    const int64_t per4 = floats_per_img / 4;
    int64_t bx = (per4 + kBlock - 1) / kBlock;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(splitk_combine_kernel, dim3((unsigned)bx, n_img), dim3(kBlock), 0, (hipStream_t)stream, partial,
                       split_stride, k_splits, part_img_stride, R, r_img_stride, gamma, out, out_img_stride, per4);
    return sf::check_launch("sf_splitk_combine");
}

extern "C" int sf_forward_interpolate(const float* flow, float* out, int n_img, int h, int w, void* stream) {
    SF_REQUIRE(flow && out && n_img > 0 && h > 0 && w > 0, "sf_forward_interpolate: bad args");
    SF_REQUIRE(flow != out, "sf_forward_interpolate: in-place operation is not supported");
    SF_REQUIRE((int64_t)h * w < (1 << 24) && n_img <= 65535, "sf_forward_interpolate: flow field too large");
    hipLaunchKernelGGL(forward_interp_kernel, dim3(sf::ceil_div(h * w, kBlock), n_img), dim3(kBlock), 0,
                       (hipStream_t)stream, flow, out, h, w);
    return sf::check_launch("sf_forward_interpolate");
}
