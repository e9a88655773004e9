// Fused GMA aggregation: out = mf + gamma * softmax(scale * q k^T) v  without an N x N tensor.
//
// Reference: demo.py:235-258 (the demo's Aggregate recomputes softmax(q k^T) v in every refinement iteration through
// flash_attn_func, or a naive einsum) == core/gma.py:53-65 + 91-104 (attention matrix computed once, attn @ v per
// iteration).  Same mathematics; this is the path for resolutions whose N x N matrix cannot be kept (1080p: 4.2 GB per
// image), selectable for any shape.
//
// Operands are packed ONCE into the exact byte images the matrix cores want, so the main loop has no conversion,
// no transposition and no VGPR staging -- tiles go HBM/L2 -> LDS by buffer_load ... lds:
//   q, k  (constant over the refinement loop: functions of the context features) -> IEEE fp16 (hi [, lo]) k-octet
//         planes [(d/8)][Ppad][8]; q is pre-multiplied by scale * log2(e) so that softmax is exp2(s - max);
//   v     (changes every iteration)  -> fp16 key-octet planes [(key/8)][128 d][8] with the 16 keys of an MFMA k-step
//         permuted into the order in which the logits' accumulator registers hold them (below).
// One workgroup = 128 queries (4 waves x 32), streaming 64-key tiles:
//   S^T[key][query] = K Q^T  (v_mfma_f32_32x32x16_f16, A = K fragment from LDS, B = Q fragment held in registers):
//        the C/D layout puts ONE query in a lane (col = lane & 31) and 16 keys in its registers, so the running max /
//        sum of the online softmax are in-lane reductions plus one exchange with lane ^ 32;
//   P^T -> fp16 in registers is directly the B operand of  O^T[d][query] += V^T P^T : lane (query, khalf) holds keys
//        (r & 3) + 8 (r >> 2) + 4 khalf of a 32-key block in registers r = 0..15, i.e. for the k-step m (registers
//        8m..8m+7) the keys 16 m + (i & 3) + 8 (i >> 2) + 4 khalf -- the v pack stores exactly that order, so the V
//        fragment is one lane-linear ds_read_b128;
//   epilogue: O^T / rowsum, gamma, residual; lanes run over queries = consecutive pixels: 128-byte stores.
// QKP = MFMA products per logit block: 1 (q_hi k_hi), 2 (+ q_lo k_hi), 3 (+ q_hi k_lo: the f16x3 split, fp32-class
// logits).  P and V are single fp16 (the materialised path already stores the attention matrix in fp16).
#include "sf_common.h"
#include <cstdlib>

#ifndef SF_FLASH_XCD_MAP
#define SF_FLASH_XCD_MAP 1
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int sf_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

constexpr int HD = 128;              // head dim (GMA: heads = 1, dim_head = 128)
constexpr int BQ = 128, BJ = 64;     // queries per workgroup, keys per tile
constexpr int KPLANE = (HD / 8) * BJ * 16;      // bytes of a K tile (hi or lo): 16 d-octets x 64 keys x 16 B = 16 KB
constexpr int VTILE = (BJ / 8) * HD * 16;       // bytes of a V tile: 8 key-octets x 128 d x 16 B = 16 KB

struct FlashArgs {
    const char* ws;                 // per image: [Qh | Ql | Kh | Kl | Vp], each 256 * Ppad bytes
    const float* mf; const float* gamma; float* out;
    _Float16* out16; int64_t out16_img_stride;      // optional fp16 k-octet copy of out (SF_LAYOUT_F16_KOCT), halves
    int64_t mf_img_stride, out_img_stride;
    int P, Ppad;
    float* part;                    // nsplit > 1 (statistics mode only): partial results [split][img][128][Ppad], summed by
    int nsplit;                     // flash_combine_kernel -- key ranges in parallel when the query tiles cannot fill the chip
    char* pbuf;                     // stored softmax weights (MODE 3 writes, gma_pv_kernel reads): per image [Ppad / 32 query tiles]
    int64_t p_img_stride;           // [Ppad / 16 key groups][64 lanes][8 halves] = the B fragments of O^T += V^T P^T; bytes per image
};

__host__ __device__ inline int64_t plane_bytes(int Ppad) { return (int64_t)256 * Ppad; }
// per image: [Qh | Ql | Kh | Kl | Vp] planes, then the softmax statistics of every query: (row maximum, 1 / row sum) in the
// log2 domain of the scaled logits -- q and k are constant over the refinement loop, so they are computed ONCE per clip
// (sf_gma_flash_pack_qk) and the per-iteration kernel starts its logit accumulators at -max: no running maximum, no
// accumulator rescale, no row sum in the loop that runs 15 times.
// ... then a 16-byte header {magic, products per logit the statistics were computed with (0 = none stored)}: the per-iteration
// kernel checks it, so statistics of a different (or no) pack call are never used silently.
constexpr int kHdrMagic = 0x53464b51;
__host__ __device__ inline int64_t img_ws_bytes(int Ppad) { return 5 * plane_bytes(Ppad) + (int64_t)Ppad * 8 + 16; }
__host__ __device__ inline int64_t hdr_offset(int Ppad) { return 5 * plane_bytes(Ppad) + (int64_t)Ppad * 8; }
constexpr int kMaxSplit = 2;        // key-range splits of the statistics-mode kernel (partial buffers follow the images in ws)
// key ranges are split exactly when the query tiles alone cannot fill the chip (a single Sintel clip: 165 workgroups for 256
// CUs, each a serial chain over 110 key tiles); the workspace carries the partial buffers only then
__host__ inline bool use_key_split(int n_img, int Ppad);
__host__ __device__ inline int64_t part_bytes(int n_img, int Ppad) { return (int64_t)kMaxSplit * n_img * HD * Ppad * 4; }

// ---- pack q, k: qk planes [img][2*HD][P] fp32 (rows 0..127 = q, 128..255 = k) -------------------------------------------
__global__ __launch_bounds__(256) void flash_pack_qk_kernel(const float* qk, int64_t qk_img_stride, char* ws, int P, int Ppad,
                                                            float qscale, int stats_products) {
    const int p = blockIdx.x * 256 + threadIdx.x, dq = blockIdx.y & 15, side = blockIdx.y >> 4, img = blockIdx.z;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        int* hdr = reinterpret_cast<int*>(ws + (int64_t)img * img_ws_bytes(Ppad) + hdr_offset(Ppad));
        hdr[0] = kHdrMagic; hdr[1] = stats_products; hdr[2] = P; hdr[3] = 0;
    }
    if (p >= Ppad) return;
    const float* src = qk + (int64_t)img * qk_img_stride + (int64_t)(side * HD + dq * 8) * P + p;
    const float mul = side == 0 ? qscale : 1.0f;
    f16x8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float x = (p < P) ? sf::mul_rn(src[(int64_t)i * P], mul) : 0.f;
        const _Float16 h = (_Float16)x;                    // round to nearest: x = hi + lo to ~22 bits
        hi[i] = h;
        lo[i] = (_Float16)(x - (float)h);
    }
    char* img_ws = ws + (int64_t)img * img_ws_bytes(Ppad);
    char* dst = img_ws + (int64_t)(side * 2) * plane_bytes(Ppad) + ((int64_t)dq * Ppad + p) * 16;
    *reinterpret_cast<f16x8*>(dst) = hi;
    *reinterpret_cast<f16x8*>(dst + plane_bytes(Ppad)) = lo;
}

// ---- pack v: planes [img][HD][P] fp32 -> fp16 [(key/8)][HD][8], keys of each 16-group in accumulator-register order ----
template <typename TIn>       // float planes, or _Float16 rows (the to_v GEMM's c_f16 = 1 output: half the bytes in)
__global__ __launch_bounds__(256) void flash_pack_v_kernel(const TIn* v, int64_t v_img_stride, char* ws, int P, int Ppad) {
    const int o = blockIdx.x * 32 + (threadIdx.x & 31);            // key octet
    const int d = blockIdx.y * 8 + (threadIdx.x >> 5), img = blockIdx.z;
    if (o * 8 >= Ppad) return;
    const int khalf = o & 1, base = (o >> 1) * 16 + 4 * khalf;     // keys base + {0,1,2,3, 8,9,10,11}
    const TIn* row = v + (int64_t)img * v_img_stride + (int64_t)d * P;
    f16x8 h;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int key = base + (i & 3) + 8 * (i >> 2);
        h[i] = (key < P) ? (_Float16)row[key] : (_Float16)0.f;
    }
    char* dst = ws + (int64_t)img * img_ws_bytes(Ppad) + 4 * plane_bytes(Ppad) + ((int64_t)o * HD + d) * 16;
    *reinterpret_cast<f16x8*>(dst) = h;
}

// ---- to_v and the v pack in one kernel (gma.py:93 + the pack above) --------------------------------------------------------
// V^T[key][d] = sum_c x[c][key] W_v[d][c] with the KEYS as MFMA rows: the C/D layout then leaves a lane (column d, k-half) with
// the 16 keys (r & 3) + 8 (r >> 2) + 4 khalf of its 32-key block -- registers 0..7 / 8..15 ARE the two packed key octets
// (keys 16 g + 4 khalf + {0,1,2,3, 8,9,10,11}) of the v planes: no transposition, one 16-byte store per 8 registers, 512
// contiguous bytes per half wave.  Operands in the formats they already have: x = the k-octet fp16 copy of the motion features
// (A fragment = one 16-byte load), W_v = the split weight planes [c / 8][128][8] (B fragment: staged once per workgroup in LDS).
// Replaces a GEMM launch (fp16 rows out) + the pack launch (rows in, octets out) on the critical path of every iteration.
template <int PM>
__global__ __launch_bounds__(256) void flash_project_v_kernel(const char* x_koct, int64_t x_img_stride_bytes, int ldx,
                                                              const char* w_hi, const char* w_lo, float alpha, char* ws, int P,
                                                              int Ppad) {
    constexpr int WPLANE = (HD / 8) * HD * 16;                       // 32 KB: [16 c-octets][128 d][8]
    __shared__ __attribute__((aligned(1024))) char smem[PM * WPLANE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, l31 = lane & 31;
    const int img = blockIdx.y, p0 = blockIdx.x * BQ + wave * 32;
    const __amdgpu_buffer_rsrc_t rwh = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(w_hi), 0, WPLANE, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(PM == 2 ? w_lo : w_hi), 0, WPLANE, 0x00020000);
#pragma unroll
    for (int i = 0; i < WPLANE / 4096; ++i) {                        // 1-KB pieces, wave w takes pieces w, w + 4, ...
        const int piece = wave + 4 * i;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rwh, (lds_ptr)(smem + piece * 1024), 16, lane * 16, piece * 1024, 0, 0);
        if (PM == 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rwl, (lds_ptr)(smem + WPLANE + piece * 1024), 16, lane * 16, piece * 1024, 0, 0);
    }
    // A fragments: c-octet 2 ks + khalf of key p0 + l31 (keys >= P: zeros -- the padded keys of the v planes must be 0)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(x_koct) + (int64_t)img * x_img_stride_bytes, 0, (HD / 8) * ldx * 16, 0x00020000);
    const int p = p0 + l31;
    f16x8 a[HD / 16];
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks)
        a[ks] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(
            rx, (p < P) ? ((2 * ks + khalf) * ldx + p) * 16 : (int)0x80000000u, 0, 0));
    f32x16 acc[HD / 32];
#pragma unroll
    for (int td = 0; td < HD / 32; ++td)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[td][r] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks)
#pragma unroll
        for (int td = 0; td < HD / 32; ++td) {
            const int off = ((2 * ks + khalf) * HD + td * 32 + l31) * 16;
            if (PM == 2) acc[td] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], *reinterpret_cast<const f16x8*>(smem + WPLANE + off), acc[td], 0, 0, 0);
            acc[td] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], *reinterpret_cast<const f16x8*>(smem + off), acc[td], 0, 0, 0);
        }
    if (p0 >= Ppad) return;                                          // (wave-uniform; grid covers Ppad exactly: never taken)
    char* dst = ws + (int64_t)img * img_ws_bytes(Ppad) + 4 * plane_bytes(Ppad);
#pragma unroll
    for (int td = 0; td < HD / 32; ++td)
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
            f16x8 h;
#pragma unroll
            for (int i = 0; i < 8; ++i) h[i] = (_Float16)(alpha * acc[td][8 * grp + i]);
            const int o = 2 * (p0 / 16 + grp) + khalf;               // key octet of the v planes
            *reinterpret_cast<f16x8*>(dst + ((int64_t)o * HD + td * 32 + l31) * 16) = h;
        }
}

__device__ __forceinline__ float xor32(float v) {          // value of lane ^ 32
    return __shfl_xor(v, 32, 64);
}

#ifndef SF_FLASH_PRIO
#define SF_FLASH_PRIO 0   // 1: s_setprio(1) around the two MFMA clusters of a tile (A/B knob, tools/build_variant.sh)
#endif
#ifndef SF_FLASH_V1
#define SF_FLASH_V1 1     // one-product kernel: ONE V stage (48 KB of LDS, 3 workgroups per CU) instead of two (64 KB, 2 per CU)
#endif
// MODE 0: self-contained online softmax (running maximum, rescale, row sum).  MODE 1: the statistics of the workspace are
// used (accumulators start at -max, weights are exp2 of the accumulator, the result is scaled by the stored 1 / row sum).
// MODE 2: the statistics pass -- logits and the online maximum / sum only (no V tile, no P V), writes (max, 1 / sum).
// MODE 3: the store pass -- MODE 1's softmax weights (exp2 of the logit minus the stored maximum, rounded to fp16: bit for bit what
// MODE 1 multiplies) written to g.pbuf as the B fragments of the second contraction, no V tile, no P V (gma_pv_kernel below).
template <int QKP, int MODE>
__global__ __launch_bounds__(256, (QKP == 3) ? 1 : ((QKP == 1 && SF_FLASH_V1) ? 3 : 2)) void gma_flash_kernel(const FlashArgs g) {
    constexpr bool kKlo = (QKP == 3);
    constexpr bool kUseStats = MODE == 1 || MODE == 3, kStatsPass = MODE == 2, kStoreP = MODE == 3;
    constexpr bool kNoPV = kStatsPass || kStoreP;
    constexpr int KSTAGE = KPLANE * (kKlo ? 2 : 1);
    // kV1: K tiles double-buffered, V single-buffered.  V(t) is requested at the top of tile t (every wave has finished
    // P V of tile t-1 by then) and has the logits + softmax of tile t to land; a second barrier precedes P V.
    constexpr bool kV1 = (QKP == 1) && SF_FLASH_V1;
    constexpr int NV = kV1 ? 1 : 2;
    __shared__ __attribute__((aligned(1024))) char smem[2 * KSTAGE + NV * VTILE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, l31 = lane & 31;
    // (query tile, image) from the linear workgroup id such that every XCD walks whole images: the K / V planes of an image (3.6 MB at
    // the Sintel grid) are read by all its query tiles and live in ONE XCD's L2 instead of eight (round 5; FETCH_SIZE of the kernel
    // 889 -> see profiles: the requests went to the Infinity Cache before)
#if SF_FLASH_XCD_MAP
    const int wg_lin = sf::xcd_linear_id((int)(blockIdx.x + gridDim.x * blockIdx.y), (int)(gridDim.x * gridDim.y));
    const int img = wg_lin / (int)gridDim.x, q0 = (wg_lin % (int)gridDim.x) * BQ;
#else
    const int img = blockIdx.y, q0 = blockIdx.x * BQ;
#endif
    const int P = g.P, Ppad = g.Ppad;
    const int plane = (int)plane_bytes(Ppad);                       // < 2 GiB (host-checked)
    const char* ws = g.ws + (int64_t)img * img_ws_bytes(Ppad);
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ws), 0, 2 * plane, 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ws) + 2 * (int64_t)plane, 0, 2 * plane, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ws) + 4 * (int64_t)plane, 0, plane, 0x00020000);

    // ---- Q fragments of this lane's query (B operand: 8 consecutive d per k-half), held for the whole kernel ----
    const int q = q0 + wave * 32 + l31;                              // < Ppad always (Ppad is a multiple of 128)
    f16x8 qh[HD / 16], ql[(QKP >= 2) ? HD / 16 : 1];
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) {
        const int off = ((ks * 2 + khalf) * Ppad + q) * 16;
        qh[ks] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rq, off, 0, 0));
        if (QKP >= 2) ql[ks] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rq, off, plane, 0));
    }

    float2* stats = reinterpret_cast<float2*>(const_cast<char*>(ws) + 5 * (int64_t)plane);
    float st_m = 0.f, st_inv = 0.f;
    if (kUseStats) {
        const float2 st = stats[q]; st_m = st.x; st_inv = st.y;
        // statistics stored by another product count (or none at all): poison the result instead of using them
        const int* hdr = reinterpret_cast<const int*>(ws + hdr_offset(Ppad));
        if (hdr[0] != kHdrMagic || hdr[1] != QKP || hdr[2] != P) { st_inv = __builtin_nanf(""); if (kStoreP) st_m = st_inv; }
    }
    // MODE 3: the image's fragments, [key tile of 64][query tile of 32][4 key groups of 16] x 1 KB (lane-linear): the workgroups of an
    // image walk the key tiles roughly together, so what the chip reads at any time is a few contiguous regions (DRAM rows, TLB
    // entries) instead of one private stream per wave
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
        kStoreP ? g.pbuf + (int64_t)img * g.p_img_stride : const_cast<char*>(ws), 0, kStoreP ? (int)(unsigned)g.p_img_stride : 0, 0x00020000);
    const unsigned p_lane = (unsigned)(((q0 >> 5) + wave) * 4096 + lane * 16), p_tile = (unsigned)(Ppad / 32) * 4096u;
    // ---- tile DMA: K tile = 16 d-octet rows of 64 keys x 16 B (1 KB pieces), V tile = 16 KB contiguous ----
    auto issue_v = [&](int t, int buf) {
        const int j0 = t * BJ;
        char* vb = smem + 2 * KSTAGE + buf * VTILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave * 4 + i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr)(vb + piece * 1024), 16, lane * 16, j0 * (HD * 2) + piece * 1024, 0, 0);
        }
    };
    auto issue_k = [&](int t, int buf) {
        const int j0 = t * BJ;
        char* kb = smem + buf * KSTAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int dq = wave * 4 + i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_ptr)(kb + dq * 1024), 16, (j0 + lane) * 16, dq * Ppad * 16, 0, 0);
            if (kKlo)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_ptr)(kb + KPLANE + dq * 1024), 16, (j0 + lane) * 16,
                                                         plane + dq * Ppad * 16, 0, 0);
        }
    };

    f32x16 o[HD / 32];
#pragma unroll
    for (int t = 0; t < HD / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -1.0e30f, l_run = 0.f;

    const int nt_all = Ppad / BJ;
    const int nsp = (kUseStats && g.nsplit > 1) ? g.nsplit : 1, sp = (nsp > 1) ? (int)blockIdx.z : 0;
    const int tb = nt_all * sp / nsp, nt = nt_all * (sp + 1) / nsp;            // this workgroup's key tiles [tb, nt)
    issue_k(tb, tb & 1);
    if (!kV1 && !kNoPV) issue_v(tb, tb & 1);
    for (int t = tb; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of tile t have landed ...
        __builtin_amdgcn_s_barrier();                             // ... everyone's; the other stage is no longer read
        if (kV1 && !kNoPV) issue_v(t, 0);                         // (V first: its wait below leaves the K pieces in flight)
        if (t + 1 < nt) {
            issue_k(t + 1, (t + 1) & 1);
            if (!kV1 && !kNoPV) issue_v(t + 1, (t + 1) & 1);
        }
        const char* kb = smem + (t & 1) * KSTAGE;
        const char* vb = smem + 2 * KSTAGE + (kV1 ? 0 : (t & 1)) * VTILE;

        // ---- logits (transposed): s[sub][r] = <k_key, q_query>, key = j0 + sub*32 + (r&3) + 8(r>>2) + 4 khalf ----
        f32x16 s[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[sub][r] = kUseStats ? -st_m : 0.f;     // (a lane holds ONE query: its -max is the start value)
        }
#if SF_FLASH_PRIO
        __builtin_amdgcn_s_setprio(1);                            // (experiment: MFMA clusters at raised priority, guide T5)
#endif
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int off = ((ks * 2 + khalf) * BJ + sub * 32 + l31) * 16;
                const f16x8 kh = *reinterpret_cast<const f16x8*>(kb + off);
                if (QKP >= 2) s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], s[sub], 0, 0, 0);
                if (kKlo) {
                    const f16x8 kl = *reinterpret_cast<const f16x8*>(kb + KPLANE + off);
                    s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], s[sub], 0, 0, 0);
                }
                s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s[sub], 0, 0, 0);
            }
        }
#if SF_FLASH_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        // ---- online softmax for this lane's query ----
        if ((t + 1) * BJ > P) {                                   // last tile with padded keys (workgroup-uniform)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * BJ + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    s[sub][r] = (key < P) ? s[sub][r] : -1.0e30f;
                }
        }
        float m_new = 0.f, alpha = 1.0f;
        bool grew = false;
        if constexpr (!kUseStats) {
            float mx = s[0][0];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[sub][r]);
            mx = fmaxf(mx, xor32(mx));
            m_new = fmaxf(m_run, mx);
            // the running maximum of a query stops growing after a few tiles: rescale the 64 accumulator registers only when
            // some lane of the wave needs it (wave-uniform branch)
            grew = __any(m_new > m_run);
            alpha = grew ? __builtin_amdgcn_exp2f(m_run - m_new) : 1.0f;
            m_run = m_new;
        }
        float psum = 0.f;
        f16x8 pf[4];                                              // P^T as B operands: k-step kk = sub*2 + m
        // two weights at a time: one v_cvt_pk_f16_f32 (round to nearest) and one v_dot2 that adds the two ROUNDED values
        // to the row sum -- the row is normalised by what is actually multiplied
        typedef _Float16 hp2 __attribute__((ext_vector_type(2)));
        const hp2 ones = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                // (kUseStats: the accumulator started at -max, the logit IS the exponent)
                const float p0 = __builtin_amdgcn_exp2f(kUseStats ? s[sub][r] : s[sub][r] - m_new);
                const float p1 = __builtin_amdgcn_exp2f(kUseStats ? s[sub][r + 1] : s[sub][r + 1] - m_new);
                hp2 ph;
                ph[0] = (_Float16)p0;
                ph[1] = (_Float16)p1;
                if constexpr (!kUseStats) psum = __builtin_amdgcn_fdot2(ph, ones, psum, false);
                const unsigned bits = __builtin_bit_cast(unsigned, ph);
                pf[sub * 2 + (r >> 3)][r & 7] = __builtin_bit_cast(_Float16, (unsigned short)(bits & 0xffffu));
                pf[sub * 2 + (r >> 3)][(r & 7) + 1] = __builtin_bit_cast(_Float16, (unsigned short)(bits >> 16));
            }
        if constexpr (!kUseStats) {
            l_run = l_run * alpha + psum;
            if (grew && !kNoPV) {
#pragma unroll
                for (int td = 0; td < HD / 32; ++td)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[td][r] *= alpha;
            }
        }
        if constexpr (kStoreP) {                                  // the four B fragments of this key tile: 1 KB per instruction, lane-linear
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sf_u32x4, pf[kk]), rp, (int)(p_lane + (unsigned)t * p_tile + kk * 1024), 0, 0);
        }
        if constexpr (kNoPV) continue;                            // (no V tile was requested: nothing to wait for, no P V)
        if (kV1) {                                                // V(t): 4 pieces per wave, requested before the K(t+1) pieces
            if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        // ---- O^T += V^T P^T ----
#if SF_FLASH_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int td = 0; td < HD / 32; ++td) {
                const f16x8 vf = *reinterpret_cast<const f16x8*>(vb + ((2 * kk + khalf) * HD + td * 32 + l31) * 16);
                o[td] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kk], o[td], 0, 0, 0);
            }
        }
#if SF_FLASH_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    }

    // ---- epilogue: out[d][q] = mf[d][q] + gamma * O^T[d][q] / rowsum ----
    const float l_tot = l_run + xor32(l_run);
    if constexpr (kStoreP) {                                  // the header remembers which logits the stored weights belong to
        if (q0 == 0 && tid == 0) reinterpret_cast<int*>(const_cast<char*>(ws) + hdr_offset(Ppad))[3] = 0x100 | QKP;
        return;
    }
    if constexpr (kStatsPass) {
        if (khalf == 0) stats[q] = make_float2(m_run, 1.0f / l_tot);
        return;
    }
    const float w = kUseStats ? g.gamma[0] * st_inv : g.gamma[0] / l_tot;
    if (kUseStats && nsp > 1) {                              // partial sums over this key range: [split][img][d][Ppad]
        float* pp = g.part + (((int64_t)sp * gridDim.y + img) * HD) * Ppad + q;
#pragma unroll
        for (int td = 0; td < HD / 32; ++td)
#pragma unroll
            for (int r = 0; r < 16; ++r) pp[(int64_t)(td * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf) * Ppad] = w * o[td][r];
        return;
    }
    if (q < P) {
        const float* mf = g.mf + (int64_t)img * g.mf_img_stride + q;
        float* out = g.out + (int64_t)img * g.out_img_stride + q;
        // k-octet copy: registers 4m..4m+3 of a lane are channels 8m + 4 khalf .. + 3 of octet td*4 + m -- one 8-byte
        // store per octet; lanes l and l + 32 complete an octet, a wave instruction covers 512 contiguous bytes
        _Float16* o16 = g.out16 ? g.out16 + (int64_t)img * g.out16_img_stride + (int64_t)q * 8 + khalf * 4 : nullptr;
#pragma unroll
        for (int td = 0; td < HD / 32; ++td)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                h4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * m + e;
                    const int d = td * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    const float val = mf[(int64_t)d * P] + w * o[td][r];
                    out[(int64_t)d * P] = val;
                    hv[e] = (_Float16)val;
                }
                if (o16) *reinterpret_cast<h4*>(o16 + (int64_t)(td * 4 + m) * P * 8) = hv;
            }
    }
}

// ---- the recompute kernel, software-pipelined inside the wave (round 6; MODE 1 of the kernel above, same arithmetic) -----------
// PMC / timers of gma_flash_kernel<1, 1>: matrix pipe 54 % busy with three waves per SIMD.  A wave there alternates phases that use
// ONE unit each -- 16 logit MFMAs, then 32 exp2 + 16 conversions on the VALU, a barrier, then 16 P V MFMAs -- and whether another
// wave fills the idle unit is left to chance (the workgroups of a CU drift into the same phase behind their barriers).  Here the
// wave overlaps itself: while the softmax weights of key tile t are computed on the VALU, the matrix pipe runs the logits of tile
// t + 1 (two logit register sets, the instruction streams interleaved with sched_group_barrier: one MFMA, one fragment read, two
// exp2, one conversion), then P V of tile t.  K and V tiles both double-buffered, requested a whole iteration before their use,
// ONE barrier per key tile.  Two workgroups per CU (64 KB of LDS, <= 256 registers).  Statistics mode only (accumulators start at
// -max): the logits, the fp16 weights and the MFMA order of P V are those of gma_flash_kernel<QKP, 1> -- bit-identical results.
#ifndef SF_FLASH_PIPE
#define SF_FLASH_PIPE 1
#endif
template <int QKP>
__global__ __launch_bounds__(256, (QKP == 1) ? 2 : 1) void gma_flash_pipe_kernel(const FlashArgs g) {
    constexpr bool kKlo = (QKP == 3);
    constexpr int KSTAGE = KPLANE * (kKlo ? 2 : 1);
    constexpr int KP = kKlo ? 8 : 4;                                // DMA pieces of a K tile per wave
    __shared__ __attribute__((aligned(1024))) char smem[2 * KSTAGE + 2 * VTILE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, l31 = lane & 31;
#if SF_FLASH_XCD_MAP
    const int wg_lin = sf::xcd_linear_id((int)(blockIdx.x + gridDim.x * blockIdx.y), (int)(gridDim.x * gridDim.y));
    const int img = wg_lin / (int)gridDim.x, q0 = (wg_lin % (int)gridDim.x) * BQ;
#else
    const int img = blockIdx.y, q0 = blockIdx.x * BQ;
#endif
    const int P = g.P, Ppad = g.Ppad;
    const int plane = (int)plane_bytes(Ppad);
    const char* ws = g.ws + (int64_t)img * img_ws_bytes(Ppad);
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ws), 0, 2 * plane, 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ws) + 2 * (int64_t)plane, 0, 2 * plane, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ws) + 4 * (int64_t)plane, 0, plane, 0x00020000);
    const int q = q0 + wave * 32 + l31;
    f16x8 qh[HD / 16], ql[(QKP >= 2) ? HD / 16 : 1];
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) {
        const int off = ((ks * 2 + khalf) * Ppad + q) * 16;
        qh[ks] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rq, off, 0, 0));
        if (QKP >= 2) ql[ks] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rq, off, plane, 0));
    }
    const float2* stats = reinterpret_cast<const float2*>(ws + 5 * (int64_t)plane);
    float st_m, st_inv;
    {
        const float2 st = stats[q]; st_m = st.x; st_inv = st.y;
        const int* hdr = reinterpret_cast<const int*>(ws + hdr_offset(Ppad));
        if (hdr[0] != kHdrMagic || hdr[1] != QKP || hdr[2] != P) st_inv = __builtin_nanf("");
    }
    auto issue_v = [&](int t) {
        char* vb = smem + 2 * KSTAGE + (t & 1) * VTILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave * 4 + i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr)(vb + piece * 1024), 16, lane * 16, t * (BJ * HD * 2) + piece * 1024, 0, 0);
        }
    };
    auto issue_k = [&](int t) {
        char* kb = smem + (t & 1) * KSTAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int dq = wave * 4 + i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_ptr)(kb + dq * 1024), 16, (t * BJ + lane) * 16, dq * Ppad * 16, 0, 0);
            if (kKlo)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_ptr)(kb + KPLANE + dq * 1024), 16, (t * BJ + lane) * 16,
                                                         plane + dq * Ppad * 16, 0, 0);
        }
    };
    f32x16 o[HD / 32];
#pragma unroll
    for (int t = 0; t < HD / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    const int nt_all = Ppad / BJ;
    const int nsp = (g.nsplit > 1) ? g.nsplit : 1, sp = (nsp > 1) ? (int)blockIdx.z : 0;
    const int tb = nt_all * sp / nsp, nt = nt_all * (sp + 1) / nsp;

    // logits of one key tile (transposed: a lane holds ONE query, 32 of the tile's keys in its registers), accumulators start at -max
    auto qk_tile = [&](const char* kb, f32x16 (&s)[2]) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[sub][r] = -st_m;
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int off = ((ks * 2 + khalf) * BJ + sub * 32 + l31) * 16;
                const f16x8 kh = *reinterpret_cast<const f16x8*>(kb + off);
                if (QKP >= 2) s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], s[sub], 0, 0, 0);
                if (kKlo) {
                    const f16x8 kl = *reinterpret_cast<const f16x8*>(kb + KPLANE + off);
                    s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], s[sub], 0, 0, 0);
                }
                s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s[sub], 0, 0, 0);
            }
        }
    };
    // fp16 softmax weights of a tile as the B fragments of the second contraction (k-step kk = sub * 2 + m)
    typedef _Float16 hp2 __attribute__((ext_vector_type(2)));
    auto weights = [&](const f32x16 (&s)[2], f16x8 (&pf)[4]) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                hp2 ph;
                ph[0] = (_Float16)__builtin_amdgcn_exp2f(s[sub][r]);
                ph[1] = (_Float16)__builtin_amdgcn_exp2f(s[sub][r + 1]);
                const unsigned bits = __builtin_bit_cast(unsigned, ph);
                pf[sub * 2 + (r >> 3)][r & 7] = __builtin_bit_cast(_Float16, (unsigned short)(bits & 0xffffu));
                pf[sub * 2 + (r >> 3)][(r & 7) + 1] = __builtin_bit_cast(_Float16, (unsigned short)(bits >> 16));
            }
    };

    // ---- prologue: K(tb), V(tb), K(tb + 1) requested; logits of tile tb ----
    issue_k(tb);
    issue_v(tb);
    if (tb + 1 < nt) {
        issue_k(tb + 1);
        if (kKlo) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    static_assert(KP == 4 || KP == 8, "the prologue wait counts the pieces of K(tb + 1)");
    __builtin_amdgcn_s_barrier();
    f32x16 sa[2], sb[2];
    qk_tile(smem + (tb & 1) * KSTAGE, sa);

    // one key tile: `cur` = its logits (ready), `nxt` receives the logits of tile t + 1
    auto iter = [&](f32x16 (&cur)[2], f32x16 (&nxt)[2], int t) {
        // everyone has finished tile t - 1 (its fragment reads EXECUTED: the refill race of DESIGN 12.3); K(t + 1) and V(t), requested
        // an iteration ago, have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 2 < nt) issue_k(t + 2);                           // into K(t)'s buffer
        if (t + 1 < nt) issue_v(t + 1);                           // into V(t - 1)'s buffer
        if ((t + 1) * BJ > P) {                                   // last tile with padded keys (workgroup-uniform)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * BJ + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    cur[sub][r] = (key < P) ? cur[sub][r] : -1.0e30f;
                }
        }
        f16x8 pf[4];
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase A: softmax weights of tile t on the VALU beside the logits of tile t + 1 on the matrix pipe.  Unconditional: behind
        // the last tile the logits of a stale K buffer are computed and dropped (1 tile in Ppad / 64) -- a branch here lets hipcc hoist
        // the exponentials out of the MFMA region ----
        qk_tile(smem + ((t + 1) & 1) * KSTAGE, nxt);
        weights(cur, pf);
        {
            constexpr int kMfma = 16 * QKP, kReads = kKlo ? 32 : 16, kAhead = 4;
            // (the accumulator start values and the fragment address are VALU work the reads / MFMAs depend on: they need slots in front,
            // or the pipeline has no valid order and hipcc drops it altogether)
            __builtin_amdgcn_sched_group_barrier(0x002, 36, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, kAhead, 0);
#pragma unroll
            for (int i = 0; i < kMfma; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                // fragment reads: one per MFMA (QKP = 3: 32 reads for 48 MFMAs -- one behind two of every three)
                if (kKlo) { if (i % 3 != 2 && (i / 3) * 2 + (i % 3) + kAhead < kReads) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                else if (i / QKP + kAhead < kReads && i % QKP == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                // 32 exp2 + 16 conversions spread over the first MFMAs
                if (i < 16) __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
                if (i < 16) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase B: O^T += V^T P^T ----
        const char* vb = smem + 2 * KSTAGE + (t & 1) * VTILE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int td = 0; td < HD / 32; ++td) {
                const f16x8 vf = *reinterpret_cast<const f16x8*>(vb + ((2 * kk + khalf) * HD + td * 32 + l31) * 16);
                o[td] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kk], o[td], 0, 0, 0);
            }
        }
        {
            constexpr int kAhead = 4;
            __builtin_amdgcn_sched_group_barrier(0x100, kAhead, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i + kAhead < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    {
        int t = tb;
        for (;;) {
            iter(sa, sb, t);
            if (++t >= nt) break;
            iter(sb, sa, t);
            if (++t >= nt) break;
        }
    }

    // ---- epilogue: out[d][q] = mf[d][q] + gamma / rowsum * O^T[d][q] ----
    const float w = g.gamma[0] * st_inv;
    if (nsp > 1) {
        float* pp = g.part + (((int64_t)sp * gridDim.y + img) * HD) * Ppad + q;
#pragma unroll
        for (int td = 0; td < HD / 32; ++td)
#pragma unroll
            for (int r = 0; r < 16; ++r) pp[(int64_t)(td * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf) * Ppad] = w * o[td][r];
        return;
    }
    if (q < P) {
        const float* mf = g.mf + (int64_t)img * g.mf_img_stride + q;
        float* out = g.out + (int64_t)img * g.out_img_stride + q;
        _Float16* o16 = g.out16 ? g.out16 + (int64_t)img * g.out16_img_stride + (int64_t)q * 8 + khalf * 4 : nullptr;
#pragma unroll
        for (int td = 0; td < HD / 32; ++td)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                h4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * m + e;
                    const int d = td * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    const float val = mf[(int64_t)d * P] + w * o[td][r];
                    out[(int64_t)d * P] = val;
                    hv[e] = (_Float16)val;
                }
                if (o16) *reinterpret_cast<h4*>(o16 + (int64_t)(td * 4 + m) * P * 8) = hv;
            }
    }
}

// ---- stored softmax weights: out = mf + gamma / rowsum * V^T P^T with P^T streamed from HBM ------------------------------------
// q and k are constant over the refinement loop, so are the softmax weights: MODE 3 above stores them ONCE per clip (fp16, n Ppad^2
// 2 bytes: 2.4 GB for 24 Sintel images) in the register image of the second contraction's B operand, and every iteration only
// streams them past V: HALF the matrix-core work of the recompute kernel, no exp2, no conversions -- an HBM-bound kernel (99 MB per
// image-iteration) on a chip whose step is power-bound.  core/gma.py:53-65 materialises the same matrix
// once ("attn") and multiplies it every iteration (gma.py:99-102); this is that path with the matrix in fragment order.
// A wave owns 32 queries: its fragments of a key tile are 4 KB, loaded straight into registers (16 bytes per lane and fragment, a
// ring of four key tiles = 16 KB in flight per wave; layout [key tile][query tile][4] x 1 KB: see MODE 3); V tiles as in the recompute kernel (16 KB by LDS-DMA, shared by the
// four waves), a ring of three.  Per key tile one barrier and, per wave, vmcnt(12): everything but the newest three request groups
// (two P tiles and one V tile) has landed -- P tiles get two iterations, V tiles one, to arrive.  The MFMA sequence per tile is the
// recompute kernel's: with the same statistics the results are bit-identical to MODE 1.
#ifndef SF_PV_WAVES
#define SF_PV_WAVES 2
#endif
#ifndef SF_PV_NT
#define SF_PV_NT 2              // cache policy of the weight stream's loads (2 = non-temporal): A/B knob, tools/gma_stored_bench.py
#endif
#ifndef SF_PV_ABLATE
#define SF_PV_ABLATE 0          // timing ablations: 1 = no V tiles requested, 2 = no weight loads, 3 = no MFMAs
#endif
__global__ __launch_bounds__(256, SF_PV_WAVES) void gma_pv_kernel(const FlashArgs g) {
    constexpr int PD = 4, NVB = 3;                                 // P tiles in registers, V tiles in LDS
    __shared__ __attribute__((aligned(1024))) char smem[NVB * VTILE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, l31 = lane & 31;
#if SF_FLASH_XCD_MAP
    const int wg_lin = sf::xcd_linear_id((int)(blockIdx.x + gridDim.x * blockIdx.y), (int)(gridDim.x * gridDim.y));
    const int img = wg_lin / (int)gridDim.x, q0 = (wg_lin % (int)gridDim.x) * BQ;
#else
    const int img = blockIdx.y, q0 = blockIdx.x * BQ;
#endif
    const int P = g.P, Ppad = g.Ppad;
    const int plane = (int)plane_bytes(Ppad);
    const char* ws = g.ws + (int64_t)img * img_ws_bytes(Ppad);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ws) + 4 * (int64_t)plane, 0, plane, 0x00020000);
    // the image's fragments: [key tile][query tile of 32][4 key groups] x 1 KB (< 4 GiB per image, host-checked)
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(g.pbuf + (int64_t)img * g.p_img_stride, 0, (int)(unsigned)g.p_img_stride, 0x00020000);
    const unsigned p_lane = (unsigned)(((q0 >> 5) + wave) * 4096 + lane * 16), p_tile = (unsigned)(Ppad / 32) * 4096u;
    const int q = q0 + wave * 32 + l31;
    const float2* stats = reinterpret_cast<const float2*>(ws + 5 * (int64_t)plane);
    float st_inv = stats[q].y;
    {   // the stored weights must belong to the statistics in ws (same pack call, same products per logit): else poison the result
        const int* hdr = reinterpret_cast<const int*>(ws + hdr_offset(Ppad));
        if (hdr[0] != kHdrMagic || hdr[1] == 0 || hdr[2] != P || hdr[3] != (0x100 | hdr[1])) st_inv = __builtin_nanf("");
    }
    // every offset travels in the CHECKED vector offset: tiles past the end of the planes / the stream read as zeros
    auto issue_v = [&](int t, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave * 4 + i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr)(smem + buf * VTILE + piece * 1024), 16,
                                                     (SF_PV_ABLATE == 1) ? (int)0x7ffffff0 : lane * 16 + t * (BJ * HD * 2) + piece * 1024, 0, 0, 0);
        }
    };
    f16x8 pf[PD][4];
    auto load_p = [&](int t, f16x8 (&dst)[4]) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            dst[kk] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(
                rp, (SF_PV_ABLATE == 2) ? (int)0xfffffff0u : (int)(p_lane + (unsigned)t * p_tile + kk * 1024), 0, SF_PV_NT));
    };
    f32x16 o[HD / 32];
#pragma unroll
    for (int t = 0; t < HD / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;

    const int nt_all = Ppad / BJ;
    const int nsp = (g.nsplit > 1) ? g.nsplit : 1, sp = (nsp > 1) ? (int)blockIdx.z : 0;
    const int tb = nt_all * sp / nsp, nt = nt_all * (sp + 1) / nsp;
    issue_v(tb, 0);
    issue_v(tb + 1, 1);
    load_p(tb, pf[0]);
    load_p(tb + 1, pf[1]);
    load_p(tb + 2, pf[2]);
    for (int t0 = tb; t0 < nt; t0 += 12) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int t = t0 + u;
            if (t < nt) {                                              // (wave-uniform)
                load_p(t + 3, pf[(u + 3) % PD]);
                // every V fragment read of the previous tile has EXECUTED before the barrier (the refill race of DESIGN 12.3)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");     // V(t), P(t) [and P(t+1)] have landed ...
                __builtin_amdgcn_s_barrier();                         // ... everyone's V pieces; buffer (t+2) % 3 is no longer read
                issue_v(t + 2, (u + 2) % NVB);
                const char* vb = smem + (u % NVB) * VTILE;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int td = 0; td < HD / 32; ++td) {
                        const f16x8 vf = *reinterpret_cast<const f16x8*>(vb + ((2 * kk + khalf) * HD + td * 32 + l31) * 16);
#if SF_PV_ABLATE == 3
                        o[td][kk] += (float)vf[0] * (float)pf[u % PD][kk][0];
#else
                        o[td] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[u % PD][kk], o[td], 0, 0, 0);
#endif
                    }
                }
                // issue order, pinned: the V fragment reads run kAhead MFMAs ahead of their use (left alone hipcc reads two fragments,
                // waits for them, multiplies: the LDS latency in front of every pair)
                {
                    constexpr int kAhead = 4;
                    __builtin_amdgcn_sched_group_barrier(0x100, kAhead, 0);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (i + kAhead < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // (V tiles requested past the end must land before the LDS is released)
    const float w = g.gamma[0] * st_inv;
    if (nsp > 1) {
        float* pp = g.part + (((int64_t)sp * gridDim.y + img) * HD) * Ppad + q;
#pragma unroll
        for (int td = 0; td < HD / 32; ++td)
#pragma unroll
            for (int r = 0; r < 16; ++r) pp[(int64_t)(td * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf) * Ppad] = w * o[td][r];
        return;
    }
    if (q < P) {
        const float* mf = g.mf + (int64_t)img * g.mf_img_stride + q;
        float* out = g.out + (int64_t)img * g.out_img_stride + q;
        _Float16* o16 = g.out16 ? g.out16 + (int64_t)img * g.out16_img_stride + (int64_t)q * 8 + khalf * 4 : nullptr;
#pragma unroll
        for (int td = 0; td < HD / 32; ++td)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                h4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * m + e;
                    const int d = td * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    const float val = mf[(int64_t)d * P] + w * o[td][r];
                    out[(int64_t)d * P] = val;
                    hv[e] = (_Float16)val;
                }
                if (o16) *reinterpret_cast<h4*>(o16 + (int64_t)(td * 4 + m) * P * 8) = hv;
            }
    }
}

// out = mf + sum of the key-range partials (+ the fp16 k-octet copy): thread = (pixel, channel octet)
__global__ __launch_bounds__(256) void flash_combine_kernel(const float* part, int nsplit, const float* mf, int64_t mf_img_stride,
                                                            float* out, int64_t out_img_stride, _Float16* out16,
                                                            int64_t out16_img_stride, int P, int Ppad) {
    const int q = blockIdx.x * 256 + threadIdx.x, oct = blockIdx.y, img = blockIdx.z, n_img = gridDim.z;
    if (q >= P) return;
    f16x8 hv;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int d = oct * 8 + e;
        float v = mf[(int64_t)img * mf_img_stride + (int64_t)d * P + q];
        for (int s = 0; s < nsplit; ++s) v += part[(((int64_t)s * n_img + img) * HD + d) * Ppad + q];
        out[(int64_t)img * out_img_stride + (int64_t)d * P + q] = v;
        hv[e] = (_Float16)v;
    }
    if (out16) *reinterpret_cast<f16x8*>(out16 + (int64_t)img * out16_img_stride + ((int64_t)oct * P + q) * 8) = hv;
}

}  // namespace

namespace {
__host__ inline bool use_key_split(int n_img, int Ppad) { return Ppad / BJ >= 8 && (int64_t)(Ppad / BQ) * n_img < 384; }
}

extern "C" int64_t sf_gma_flash_ws_bytes(int n_img, int P) {
    if (n_img <= 0 || P <= 0) return 0;
    const int Ppad = sf::ceil_div(P, BQ) * BQ;
    return (int64_t)n_img * img_ws_bytes(Ppad) + (use_key_split(n_img, Ppad) ? part_bytes(n_img, Ppad) : 0);
}

extern "C" int sf_gma_flash_pack_qk(const float* qk, int64_t qk_img_stride, void* ws, int64_t ws_bytes, int n_img, int P,
                                    float scale, int stats_qk_products, void* stream) {
    SF_REQUIRE(stats_qk_products >= 0 && stats_qk_products <= 3, "sf_gma_flash_pack_qk: stats_qk_products must be 0 (none), 1, 2 or 3");
    SF_REQUIRE(qk && ws, "sf_gma_flash_pack_qk: null pointer");
    SF_REQUIRE(n_img > 0 && P > 0 && n_img <= 65535, "sf_gma_flash_pack_qk: bad dims");
    SF_REQUIRE(ws_bytes >= sf_gma_flash_ws_bytes(n_img, P) && (reinterpret_cast<uintptr_t>(ws) & 15) == 0,
               "sf_gma_flash_pack_qk: workspace too small or misaligned");
    const int Ppad = sf::ceil_div(P, BQ) * BQ;
    SF_REQUIRE(2 * plane_bytes(Ppad) < ((int64_t)1 << 31), "sf_gma_flash_pack_qk: image too large");
    hipLaunchKernelGGL(flash_pack_qk_kernel, dim3(sf::ceil_div(Ppad, 256), 32, n_img), dim3(256), 0, (hipStream_t)stream, qk,
                       qk_img_stride, (char*)ws, P, Ppad, scale * 1.44269504088896340736f, stats_qk_products);
    if (stats_qk_products) {                                 // the softmax statistics of every query, once per clip
        FlashArgs g = {};
        g.ws = (const char*)ws; g.P = P; g.Ppad = Ppad;
        dim3 grid(Ppad / BQ, n_img);
        switch (stats_qk_products) {
            case 1: hipLaunchKernelGGL((gma_flash_kernel<1, 2>), grid, dim3(256), 0, (hipStream_t)stream, g); break;
            case 2: hipLaunchKernelGGL((gma_flash_kernel<2, 2>), grid, dim3(256), 0, (hipStream_t)stream, g); break;
            default: hipLaunchKernelGGL((gma_flash_kernel<3, 2>), grid, dim3(256), 0, (hipStream_t)stream, g); break;
        }
    }
    return sf::check_launch("sf_gma_flash_pack_qk");
}

static int flash_aggregate(void* ws, int64_t ws_bytes, const void* v_, int v_f16, int64_t v_img_stride, const float* mf,
                                      int64_t mf_img_stride, const float* gamma, float* out, int64_t out_img_stride,
                                      void* out_koct, int64_t out_koct_img_stride, int n_img, int P, int qk_products,
                                      int use_stats, void* stream) {
    SF_REQUIRE(!out_koct || ((reinterpret_cast<uintptr_t>(out_koct) & 15) == 0 && (out_koct_img_stride & 7) == 0),
               "sf_gma_flash_aggregate: out_koct must be 16-byte aligned, its image stride a multiple of 8 halves");
    const float* v = static_cast<const float*>(v_);
    SF_REQUIRE(ws && mf && gamma && out, "sf_gma_flash_aggregate: null pointer");       // (v == NULL: sf_gma_flash_project_v packed it)
    SF_REQUIRE(n_img > 0 && P > 0 && n_img <= 65535, "sf_gma_flash_aggregate: bad dims");
    SF_REQUIRE(qk_products >= 1 && qk_products <= 3, "sf_gma_flash_aggregate: qk_products must be 1, 2 or 3");
    SF_REQUIRE(ws_bytes >= sf_gma_flash_ws_bytes(n_img, P) && (reinterpret_cast<uintptr_t>(ws) & 15) == 0,
               "sf_gma_flash_aggregate: workspace too small or misaligned");
    const int Ppad = sf::ceil_div(P, BQ) * BQ;
    SF_REQUIRE(2 * plane_bytes(Ppad) < ((int64_t)1 << 31), "sf_gma_flash_aggregate: image too large");
    if (!v_) {}                                               // the v planes of ws are current (sf_gma_flash_project_v)
    else if (v_f16)
        hipLaunchKernelGGL(flash_pack_v_kernel<_Float16>, dim3(sf::ceil_div(Ppad / 8, 32), HD / 8, n_img), dim3(256), 0,
                           (hipStream_t)stream, static_cast<const _Float16*>(v_), v_img_stride, (char*)ws, P, Ppad);
    else
        hipLaunchKernelGGL(flash_pack_v_kernel<float>, dim3(sf::ceil_div(Ppad / 8, 32), HD / 8, n_img), dim3(256), 0,
                           (hipStream_t)stream, v, v_img_stride, (char*)ws, P, Ppad);
    FlashArgs g = {};
    g.ws = (const char*)ws; g.mf = mf; g.gamma = gamma; g.out = out;
    g.out16 = static_cast<_Float16*>(out_koct); g.out16_img_stride = out_koct_img_stride;
    g.mf_img_stride = mf_img_stride; g.out_img_stride = out_img_stride; g.P = P; g.Ppad = Ppad;
    // Too few query tiles to fill the chip: with stored statistics the key range splits without any rescaling -- partial sums,
    // then one add pass (use_key_split: the same predicate sizes the workspace)
    g.nsplit = (use_stats && use_key_split(n_img, Ppad)) ? kMaxSplit : 1;
    g.part = reinterpret_cast<float*>(static_cast<char*>(ws) + (int64_t)n_img * img_ws_bytes(Ppad));
    dim3 grid(Ppad / BQ, n_img, g.nsplit);
    const char* pe = getenv("SF_FLASH_PIPE");                 // (A/B and the bit-identity test: SF_FLASH_PIPE=0 selects the round-5 kernel)
    // measured (profiles/r06_flash_pipe_ab.txt): with ONE product per logit the pipelined form is 2-3 % SLOWER than the round-5 kernel
    // (621 vs 605 us inside the step: the kernel sits at the chip's power budget, overlapping the units inside a wave buys nothing),
    // with three products it is 4 % faster (1481 vs 1549 us): used for the split-precision logits only.  SF_FLASH_PIPE=2 forces it.
    const bool pipe = SF_FLASH_PIPE && !(pe && atoi(pe) == 0) && (qk_products >= 2 || (pe && atoi(pe) == 2));
    if (use_stats && pipe) {                                  // the software-pipelined form of the statistics mode (same results)
        switch (qk_products) {
            case 1: hipLaunchKernelGGL((gma_flash_pipe_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, g); break;
            case 2: hipLaunchKernelGGL((gma_flash_pipe_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, g); break;
            default: hipLaunchKernelGGL((gma_flash_pipe_kernel<3>), grid, dim3(256), 0, (hipStream_t)stream, g); break;
        }
    } else
    switch (qk_products * 2 + (use_stats ? 1 : 0)) {
        case 2: hipLaunchKernelGGL((gma_flash_kernel<1, 0>), grid, dim3(256), 0, (hipStream_t)stream, g); break;
        case 3: hipLaunchKernelGGL((gma_flash_kernel<1, 1>), grid, dim3(256), 0, (hipStream_t)stream, g); break;
        case 4: hipLaunchKernelGGL((gma_flash_kernel<2, 0>), grid, dim3(256), 0, (hipStream_t)stream, g); break;
        case 5: hipLaunchKernelGGL((gma_flash_kernel<2, 1>), grid, dim3(256), 0, (hipStream_t)stream, g); break;
        case 6: hipLaunchKernelGGL((gma_flash_kernel<3, 0>), grid, dim3(256), 0, (hipStream_t)stream, g); break;
        default: hipLaunchKernelGGL((gma_flash_kernel<3, 1>), grid, dim3(256), 0, (hipStream_t)stream, g); break;
    }
    if (g.nsplit > 1)
        hipLaunchKernelGGL(flash_combine_kernel, dim3(sf::ceil_div(P, 256), HD / 8, n_img), dim3(256), 0, (hipStream_t)stream, g.part,
                           g.nsplit, mf, mf_img_stride, out, out_img_stride, g.out16, g.out16_img_stride, P, Ppad);
    return sf::check_launch("sf_gma_flash_aggregate");
}

extern "C" int sf_gma_flash_project_v(void* ws, int64_t ws_bytes, const void* x_koct, int64_t x_koct_img_stride, int64_t ldx,
                                      const void* w_hi, const void* w_lo, int lda_h, float alpha, int products, int n_img, int P,
                                      void* stream) {
    SF_REQUIRE(ws && x_koct && w_hi && (products == 1 || w_lo), "sf_gma_flash_project_v: null pointer");
    SF_REQUIRE(products == 1 || products == 2, "sf_gma_flash_project_v: products must be 1 (w_hi) or 2 (w_hi + w_lo)");
    SF_REQUIRE(n_img > 0 && P > 0 && n_img <= 65535 && ldx >= P && lda_h == HD, "sf_gma_flash_project_v: bad dims (to_v is 128 x 128)");
    SF_REQUIRE(ws_bytes >= sf_gma_flash_ws_bytes(n_img, P) && (reinterpret_cast<uintptr_t>(ws) & 15) == 0,
               "sf_gma_flash_project_v: workspace too small or misaligned");
    SF_REQUIRE(((reinterpret_cast<uintptr_t>(x_koct) | reinterpret_cast<uintptr_t>(w_hi) | reinterpret_cast<uintptr_t>(w_lo)) & 15) == 0 &&
                   (x_koct_img_stride & 7) == 0 && (int64_t)(HD / 8) * ldx * 16 < ((int64_t)1 << 30),
               "sf_gma_flash_project_v: operands must be 16-byte aligned (image stride %% 8 halves), image < 1 GiB");
    const int Ppad = sf::ceil_div(P, BQ) * BQ;
    dim3 grid(Ppad / BQ, n_img);
    if (products == 2)
        hipLaunchKernelGGL(flash_project_v_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, (const char*)x_koct,
                           x_koct_img_stride * 2, (int)ldx, (const char*)w_hi, (const char*)w_lo, alpha, (char*)ws, P, Ppad);
    else
        hipLaunchKernelGGL(flash_project_v_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, (const char*)x_koct,
                           x_koct_img_stride * 2, (int)ldx, (const char*)w_hi, (const char*)w_hi, alpha, (char*)ws, P, Ppad);
    return sf::check_launch("sf_gma_flash_project_v");
}

extern "C" int sf_gma_flash_aggregate(void* ws, int64_t ws_bytes, const float* v, int64_t v_img_stride, const float* mf,
                                      int64_t mf_img_stride, const float* gamma, float* out, int64_t out_img_stride,
                                      void* out_koct, int64_t out_koct_img_stride, int n_img, int P, int qk_products,
                                      int use_stats, void* stream) {
    return flash_aggregate(ws, ws_bytes, v, 0, v_img_stride, mf, mf_img_stride, gamma, out, out_img_stride, out_koct,
                           out_koct_img_stride, n_img, P, qk_products, use_stats, stream);
}

// v as fp16 ROWS [n_img][128][P] (v_img_stride in halves): what sf_gemm writes with c_f16 = 1 -- the values enter the
// second contraction as fp16 either way.
extern "C" int sf_gma_flash_aggregate_f16v(void* ws, int64_t ws_bytes, const void* v_f16, int64_t v_img_stride,
                                           const float* mf, int64_t mf_img_stride, const float* gamma, float* out,
                                           int64_t out_img_stride, void* out_koct, int64_t out_koct_img_stride, int n_img,
                                           int P, int qk_products, int use_stats, void* stream) {
    return flash_aggregate(ws, ws_bytes, v_f16, 1, v_img_stride, mf, mf_img_stride, gamma, out, out_img_stride, out_koct,
                           out_koct_img_stride, n_img, P, qk_products, use_stats, stream);
}

// ---- stored softmax weights (gma.py:53-65 "attn" kept, gma.py:99-102 per iteration) ----------------------------------------------
extern "C" int64_t sf_gma_stored_p_bytes(int n_img, int P) {
    if (n_img <= 0 || P <= 0) return 0;
    const int64_t Ppad = sf::ceil_div(P, BQ) * BQ;
    return (int64_t)n_img * Ppad * Ppad * 2;
}

extern "C" int sf_gma_flash_store_p(void* ws, int64_t ws_bytes, void* pbuf, int64_t pbuf_bytes, int n_img, int P, int qk_products,
                                    void* stream) {
    SF_REQUIRE(ws && pbuf, "sf_gma_flash_store_p: null pointer");
    SF_REQUIRE(n_img > 0 && P > 0 && n_img <= 65535, "sf_gma_flash_store_p: bad dims");
    SF_REQUIRE(qk_products >= 1 && qk_products <= 3, "sf_gma_flash_store_p: qk_products must be 1, 2 or 3");
    SF_REQUIRE(ws_bytes >= sf_gma_flash_ws_bytes(n_img, P) && (reinterpret_cast<uintptr_t>(ws) & 15) == 0,
               "sf_gma_flash_store_p: workspace too small or misaligned");
    SF_REQUIRE(pbuf_bytes >= sf_gma_stored_p_bytes(n_img, P) && (reinterpret_cast<uintptr_t>(pbuf) & 15) == 0,
               "sf_gma_flash_store_p: pbuf must hold sf_gma_stored_p_bytes(n_img, P) bytes, 16-byte aligned");
    const int Ppad = sf::ceil_div(P, BQ) * BQ;
    SF_REQUIRE(2 * plane_bytes(Ppad) < ((int64_t)1 << 31) && (int64_t)Ppad * Ppad * 2 + 3 * (int64_t)(Ppad / 32) * 4096 < ((int64_t)1 << 32),
               "sf_gma_flash_store_p: image too large (the stored weights of one image must stay under 4 GiB)");
    FlashArgs g = {};
    g.ws = (const char*)ws; g.P = P; g.Ppad = Ppad; g.nsplit = 1;
    g.pbuf = static_cast<char*>(pbuf); g.p_img_stride = (int64_t)Ppad * Ppad * 2;
    dim3 grid(Ppad / BQ, n_img);
    switch (qk_products) {
        case 1: hipLaunchKernelGGL((gma_flash_kernel<1, 3>), grid, dim3(256), 0, (hipStream_t)stream, g); break;
        case 2: hipLaunchKernelGGL((gma_flash_kernel<2, 3>), grid, dim3(256), 0, (hipStream_t)stream, g); break;
        default: hipLaunchKernelGGL((gma_flash_kernel<3, 3>), grid, dim3(256), 0, (hipStream_t)stream, g); break;
    }
    return sf::check_launch("sf_gma_flash_store_p");
}

extern "C" int sf_gma_stored_aggregate(void* ws, int64_t ws_bytes, const void* pbuf, int64_t pbuf_bytes, const void* v, int v_f16,
                                       int64_t v_img_stride, const float* mf, int64_t mf_img_stride, const float* gamma, float* out,
                                       int64_t out_img_stride, void* out_koct, int64_t out_koct_img_stride, int n_img, int P,
                                       void* stream) {
    SF_REQUIRE(!out_koct || ((reinterpret_cast<uintptr_t>(out_koct) & 15) == 0 && (out_koct_img_stride & 7) == 0),
               "sf_gma_stored_aggregate: out_koct must be 16-byte aligned, its image stride a multiple of 8 halves");
    SF_REQUIRE(ws && pbuf && mf && gamma && out, "sf_gma_stored_aggregate: null pointer");   // (v == NULL: sf_gma_flash_project_v packed it)
    SF_REQUIRE(n_img > 0 && P > 0 && n_img <= 65535, "sf_gma_stored_aggregate: bad dims");
    SF_REQUIRE(v_f16 == 0 || v_f16 == 1, "sf_gma_stored_aggregate: v_f16 must be 0 (fp32 planes) or 1 (fp16 rows)");
    SF_REQUIRE(ws_bytes >= sf_gma_flash_ws_bytes(n_img, P) && (reinterpret_cast<uintptr_t>(ws) & 15) == 0,
               "sf_gma_stored_aggregate: workspace too small or misaligned");
    SF_REQUIRE(pbuf_bytes >= sf_gma_stored_p_bytes(n_img, P) && (reinterpret_cast<uintptr_t>(pbuf) & 15) == 0,
               "sf_gma_stored_aggregate: pbuf must hold sf_gma_stored_p_bytes(n_img, P) bytes, 16-byte aligned");
    const int Ppad = sf::ceil_div(P, BQ) * BQ;
    SF_REQUIRE(2 * plane_bytes(Ppad) < ((int64_t)1 << 31) && (int64_t)Ppad * Ppad * 2 + 3 * (int64_t)(Ppad / 32) * 4096 < ((int64_t)1 << 32),
               "sf_gma_stored_aggregate: image too large (the stored weights of one image must stay under 4 GiB)");
    if (!v) {}                                                // the v planes of ws are current (sf_gma_flash_project_v)
    else if (v_f16)
        hipLaunchKernelGGL(flash_pack_v_kernel<_Float16>, dim3(sf::ceil_div(Ppad / 8, 32), HD / 8, n_img), dim3(256), 0,
                           (hipStream_t)stream, static_cast<const _Float16*>(v), v_img_stride, (char*)ws, P, Ppad);
    else
        hipLaunchKernelGGL(flash_pack_v_kernel<float>, dim3(sf::ceil_div(Ppad / 8, 32), HD / 8, n_img), dim3(256), 0,
                           (hipStream_t)stream, static_cast<const float*>(v), v_img_stride, (char*)ws, P, Ppad);
    FlashArgs g = {};
    g.ws = (const char*)ws; g.mf = mf; g.gamma = gamma; g.out = out;
    g.out16 = static_cast<_Float16*>(out_koct); g.out16_img_stride = out_koct_img_stride;
    g.mf_img_stride = mf_img_stride; g.out_img_stride = out_img_stride; g.P = P; g.Ppad = Ppad;
    g.pbuf = const_cast<char*>(static_cast<const char*>(pbuf)); g.p_img_stride = (int64_t)Ppad * Ppad * 2;
    g.nsplit = use_key_split(n_img, Ppad) ? kMaxSplit : 1;    // (the same predicate sizes the workspace's partial buffers)
    g.part = reinterpret_cast<float*>(static_cast<char*>(ws) + (int64_t)n_img * img_ws_bytes(Ppad));
    dim3 grid(Ppad / BQ, n_img, g.nsplit);
    hipLaunchKernelGGL(gma_pv_kernel, grid, dim3(256), 0, (hipStream_t)stream, g);
    if (g.nsplit > 1)
        hipLaunchKernelGGL(flash_combine_kernel, dim3(sf::ceil_div(P, 256), HD / 8, n_img), dim3(256), 0, (hipStream_t)stream, g.part,
                           g.nsplit, mf, mf_img_stride, out, out_img_stride, g.out16, g.out16_img_stride, P, Ppad);
    return sf::check_launch("sf_gma_stored_aggregate");
}
