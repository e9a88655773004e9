/* streamflow_hip.h -- C ABI of libstreamflow_hip.so (MI355X / gfx950 only).
 *
 * The reference (littlespray/StreamFlow) is pure Python/PyTorch and has no FFI or plugin registry
 * (SURVEY.md section 8b): its "interface" for the hot path is the Python call surface of
 * core/corr.py, core/utils/utils.py, core/gma.py, core/update.py and core/models/streamflow.py.
 * Each entry point below replaces the library kernels one of those call sites dispatches to; the
 * reference file:line it stands in for is cited per function.  The Python classes in
 * streamflow_amd/ keep the reference signatures and state-dict keys and call these through ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 data unless stated; the caller owns all memory
 *     (inputs, outputs, workspaces); the library never allocates or frees device memory and keeps
 *     no global state besides a thread-local error string.
 *   - `stream` is a hipStream_t passed as void*; every call only ENQUEUES work on it (no host sync,
 *     no default-stream use), so calls may be captured into a HIP graph.
 *   - return value: 0 on success, negative SF_ERR_* otherwise; sf_last_error() gives the message.
 *   - feature planes are "channel-major": tensor [n_img][C][P] with P = h*w contiguous (NCHW).
 */
#ifndef STREAMFLOW_HIP_H
#define STREAMFLOW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SF_VERSION 121

enum {
    SF_OK = 0,
    SF_ERR_BAD_ARG = -1,     /* null pointer, non-positive dim, unsupported combination */
    SF_ERR_UNSUPPORTED = -2, /* shape outside what the kernels are built for            */
    SF_ERR_HIP = -3          /* a HIP runtime call failed (message holds hipGetErrorString) */
};

int sf_version(void);
const char* sf_last_error(void);

/* ---- a5: coords_grid  (core/utils/utils.py:82-85) --------------------------------------------
 * out [batch][2][ht][wd]; out[b][0][y][x] = x, out[b][1][y][x] = y (exact small integers). */
int sf_coords_grid(float* out, int batch, int ht, int wd, void* stream);

/* ---- a4: bilinear_sampler  (core/utils/utils.py:65-79; F.grid_sample bilinear/zeros/align_corners)
 * img [M][C][Hi][Wi], coords [M][Ho][Wo][2] pixel (x,y)  ->  out [M][C][Ho][Wo].
 * mask_out (optional, may be NULL): [M][Ho][Wo][1] in-bounds indicator as the reference's mask=True. */
int sf_bilinear_sampler(const float* img, const float* coords, float* out, float* mask_out,
                        int M, int C, int Hi, int Wi, int Ho, int Wo, void* stream);

/* ---- a1+a2: CorrBlock.__init__  (core/corr.py:7-21,46-54) -------------------------------------
 * One launch builds the 4-level pyramids of `pairs` frame pairs of `B` clips.
 * Features: clip b, pair t reads f1 + b*f_clip_stride + t*f_pair_stride and the same offset from f2,
 * each a [D][h*w] plane set (for a single reference-style call: pairs=1, f_clip_stride=D*h*w).
 * Volumes: level l of (pair t, clip b, source pixel i) is the [h>>l][w>>l] map at
 * lvl{l} + t*lvl_pair_stride[l] + (b*h*w + i)*(h>>l)*(w>>l)   (floor pooling over the TARGET dims),
 * i.e. per pair exactly the reference's corr_pyramid[l] of shape [B*h*w, 1, h>>l, w>>l].
 * Level 0 = f1^T f2 / sqrt(D); levels 1..3 are pooled in the GEMM epilogue while the tile is still in
 * registers, so every pyramid cell is written once and level 0 is never re-read.
 * lvl_pair_stride: HOST array of 4 strides in floats (ignored when pairs == 1; may be NULL then).
 * num_levels must be 4 (streamflow.py:38).  precision: SF_PRECISION_FP32 (exact fp32 MFMA, k-ordered fmaf
 * chain), SF_PRECISION_F16X3 (split fp16, fp32 accumulate; see sf_gemm) -- both write fp32 volumes -- or
 * SF_PRECISION_F16: lvl0..lvl3 then point to IEEE fp16 cells (same indexing, lvl_pair_stride in cells; half the
 * bytes), products are single f16 MFMAs with fp32 accumulation. */
int sf_corr_build_pyramid(const float* f1, const float* f2, int64_t f_clip_stride, int64_t f_pair_stride,
                          float* lvl0, float* lvl1, float* lvl2, float* lvl3,
                          const int64_t* lvl_pair_stride, int B, int pairs, int D, int h, int w,
                          int num_levels, int precision, void* split_ws, int64_t split_ws_bytes, void* stream);
/* Workspace of the SF_PRECISION_F16X3 build: the (hi, lo) fp16 split of every f1 / f2 image, packed in k-octets
 * per pixel so that the build kernel can DMA operand tiles straight into LDS.  Not needed (may be NULL / 0) for
 * SF_PRECISION_FP32.  Contents are scratch: dead once the call's kernels have run. */
int64_t sf_corr_build_ws_bytes(int B, int pairs, int D, int h, int w);
/* The same build into PITCHED maps (fp32 and fp16 cells): level l of (pair t, clip b, source pixel i) is (h>>l) rows of
 * lvl_pitch[l] cells at lvl{l} + t*lvl_pair_stride[l] + (b*h*w + i)*(h>>l)*lvl_pitch[l]; cell (y, x), x < w>>l, at y*lvl_pitch[l] + x.
 * lvl_pitch: HOST array of 4 row pitches in cells, w>>l <= lvl_pitch[l] <= w rounded up to 32; NULL = the reference's dense layout
 * (== sf_corr_build_pyramid).  Why: core/corr.py:13-21 keeps [N, h_l, w_l] maps; at KITTI's 156-cell rows (624 bytes) every
 * 128-byte store run of the build straddles two cache lines.  A pitch of a multiple of 32 cells puts every row on a line
 * boundary; the pad cells x >= w>>l of a row are WRITTEN (unspecified values) so that whole lines are written; the reference's
 * [N, 1, h_l, w_l] tensor is the strided view [..., :w_l] of the pitched one (streamflow_amd.corr.CorrBlock.corr_pyramid). */
int sf_corr_build_pyramid_pitched(const float* f1, const float* f2, int64_t f_clip_stride, int64_t f_pair_stride,
                                  float* lvl0, float* lvl1, float* lvl2, float* lvl3,
                                  const int64_t* lvl_pair_stride, const int32_t* lvl_pitch, int B, int pairs, int D, int h, int w,
                                  int num_levels, int precision, void* split_ws, int64_t split_ws_bytes, void* stream);

/* ---- a3: CorrBlock.__call__  (core/corr.py:23-44) ----------------------------------------------
 * Image index img = b*pairs + t.  coords [B*pairs][2][h][w] (ch0 = x, ch1 = y)  ->
 * out[img][l*(2r+1)^2 + a*(2r+1) + b'][p] samples level l of (pair t, clip b) at
 * (x/2^l + a - r, y/2^l + b' - r), bilinear, zeros outside (first window axis moves x, corr.py:31-37).
 * out image img starts at out + img*out_img_stride (floats), channel stride h*w, so the caller can
 * write straight into a wider concatenation buffer.  Volumes are addressed as in
 * sf_corr_build_pyramid.  radius must be 4, num_levels 4.  vol_precision: the precision the volumes were built
 * with -- SF_PRECISION_F16 means lvl0..lvl3 hold fp16 cells (taps are blended in fp32), anything else fp32 cells.
 * out_koct (optional, fp16 volumes only): the same 324 channels a second time, rounded to fp16, as k-octet planes
 * [41][h*w][8] per image (SF_LAYOUT_F16_KOCT, image stride out_koct_img_stride halves, rows 324..327 zero): the
 * B operand of the first sf_gemm of the correlation encoder. */
int sf_corr_lookup(const float* lvl0, const float* lvl1, const float* lvl2, const float* lvl3,
                   const int64_t* lvl_pair_stride, const float* coords, float* out, int64_t out_img_stride,
                   void* out_koct, int64_t out_koct_img_stride, int B, int pairs, int h, int w, int num_levels,
                   int radius, int vol_precision, void* stream);
/* Lookup in the pitched maps of sf_corr_build_pyramid_pitched (fp32 cells only; lvl_pitch as there, NULL = dense). */
int sf_corr_lookup_pitched(const float* lvl0, const float* lvl1, const float* lvl2, const float* lvl3,
                           const int64_t* lvl_pair_stride, const int32_t* lvl_pitch, const float* coords, float* out,
                           int64_t out_img_stride, void* out_koct, int64_t out_koct_img_stride, int B, int pairs, int h, int w,
                           int num_levels, int radius, int vol_precision, void* stream);

/* ---- a1 + a2 + a3 in the BLOCKED fp16 volume layout (core/corr.py:7-54; csrc/corr_blocked.hip) --------------------------
 * Same mathematics as sf_corr_build_pyramid(SF_PRECISION_F16) / sf_corr_lookup, different memory layout: ONE buffer per
 * launch; image img = b*pairs + t starts at vol + img * vol_img_stride_bytes and holds one RECORD of rec_bytes per source
 * pixel (src_rows = h*w rounded up to 128 records; the padding records are written by the build and never read):
 *     record = [level 0 | level 1 | level 2 | level 3], level l = nby[l] x nbx[l] blocks of 8 x 8 cells (nb = ceil(size/8)),
 *     block (by, bx) at lvl_off[l] + (by*nbx[l] + bx)*128, cell (ty, tx) at byte ((tx%8)*8 + ty%8)*2 of block (ty/8, tx/8).
 * A 128-byte block is one cache line: a 10 x 10 lookup footprint touches ~4.5 lines per level instead of ~11 in the
 * row-major volume, and the build stores a level-0 block column (16 bytes) per lane and accumulator register.
 * Cells of a block outside the level (hl, wl not multiples of 8) have UNSPECIFIED contents; the lookup ignores them.
 *   sf_corr_blocked_geometry: the numbers above for an h x w feature grid (any output pointer may be NULL).
 *   sf_corr_blocked_bytes: n_img * src_rows * rec_bytes.
 *   sf_corr_build_blocked: features as in sf_corr_build_pyramid; ws = scratch of sf_corr_build_blocked_ws_bytes() bytes
 *       (fp16 k-octet repack of the features, 16-byte aligned); vol 128-byte aligned, vol_img_stride_bytes % 128 == 0.
 *   sf_corr_lookup_blocked: coords / channel order / sampling rule exactly as sf_corr_lookup.  out_koct: the 324 channels
 *       rounded to fp16 as k-octet planes [41][h*w][8] per image (SF_LAYOUT_F16_KOCT, rows 324..327 zero) -- the operand
 *       (and, through SfGemm.r_f16, the residual) of the correlation encoder's first block: no fp32 copy is written.
 *       out: optional fp32 planes [324][h*w] per image (un-rounded taps; API parity and tests).  At least one of the two. */
int sf_corr_blocked_geometry(int h, int w, int64_t* rec_bytes, int64_t* lvl_off, int32_t* nby, int32_t* nbx,
                             int64_t* src_rows);
int64_t sf_corr_blocked_bytes(int n_img, int h, int w);
int64_t sf_corr_build_blocked_ws_bytes(int n_img, int D, int h, int w);
int sf_corr_build_blocked(const float* f1, const float* f2, int64_t f_clip_stride, int64_t f_pair_stride,
                          void* vol, int64_t vol_img_stride_bytes, int B, int pairs, int D, int h, int w,
                          void* ws, int64_t ws_bytes, void* stream);
int sf_corr_lookup_blocked(const void* vol, int64_t vol_img_stride_bytes, const float* coords, float* out,
                           int64_t out_img_stride, void* out_koct, int64_t out_koct_img_stride, int B, int pairs,
                           int h, int w, void* stream);

/* ---- a1 + a2 + a3 with fp32 cells in a BLOCKED layout (core/corr.py:7-54 in the reference's own fp32: streamflow.py:107,110;
 * csrc/corr_blocked32.hip) -- the volumes of the fp32-class presets and of BASELINE.json configuration 3 as worded ---------
 * Same mathematics as sf_corr_build_pyramid(SF_PRECISION_F16X3) / sf_corr_lookup(SF_PRECISION_FP32); the memory layout is the
 * one of sf_corr_build_blocked with fp32 cells: level l = nby[l] x nbx[l] blocks of 4 ROWS x 8 COLUMNS (nby = ceil(hl/4),
 * nbx = ceil(wl/8)), block (by, bx) at lvl_off[l] + (by*nbx[l] + bx)*128, cell (ty, tx) at byte ((tx%8)*4 + ty%4)*4 of block
 * (ty/4, tx/8).  A footprint touches ~6.9 cache lines per level instead of ~11.6 in the pitched row-major maps.
 * Padding cells / records: unspecified contents, never read.
 *   sf_corr_lookup_blocked32: out = fp32 planes [324][h*w] per image (channel order / sampling rule of sf_corr_lookup). */
int sf_corr_blocked32_geometry(int h, int w, int64_t* rec_bytes, int64_t* lvl_off, int32_t* nby, int32_t* nbx,
                               int64_t* src_rows);
int64_t sf_corr_blocked32_bytes(int n_img, int h, int w);
int64_t sf_corr_build_blocked32_ws_bytes(int n_img, int D, int h, int w);
int sf_corr_build_blocked32(const float* f1, const float* f2, int64_t f_clip_stride, int64_t f_pair_stride,
                            void* vol, int64_t vol_img_stride_bytes, int B, int pairs, int D, int h, int w,
                            void* ws, int64_t ws_bytes, void* stream);
int sf_corr_lookup_blocked32(const void* vol, int64_t vol_img_stride_bytes, const float* coords, float* out,
                             int64_t out_img_stride, int B, int pairs, int h, int w, void* stream);

/* ---- generic fused GEMM: every 1x1 conv / nn.Linear / einsum on the path ------------------------
 * C[z][m][n] = epilogue( alpha * ( sum_k A[z][m][k] * B[z][k][n] + bias[m] ) )
 * replaces: nn.Conv2d 1x1 in PCBlock4_Deep_nopool_res (update.py:18-28), convf1 (update.py:323),
 * to_qk/to_v (gma.py:48,82), einsum QK^T (gma.py:60) and attn@v (gma.py:97), timm Linear layers
 * (update.py:466-479), mask head (update.py:756-759; 3x3 conv as implicit GEMM). */
enum { SF_LAYOUT_K_MAJOR = 0,   /* A[k*lda + m]   /  B[k*ldb + n]   (rows of k)            */
       SF_LAYOUT_K_MINOR = 1,   /* A[m*lda + k]   /  B[n*ldb + k]   (k contiguous)          */
       SF_LAYOUT_SPLIT_F16 = 2,/* A only, split precisions: weights pre-split on the host into two IEEE fp16
                                    images A_hi/A_lo laid out in k-octet planes [K padded to 32 / 8][lda_h][8] with
                                    lda_h = M padded to 128 (zero padded): element (m,k) at ((k/8)*lda_h + m)*8 + k%8,
                                    w = hi + lo to ~22 bits.  16 bytes = one MFMA operand k-octet of one row, so a
                                    tile is moved to LDS by buffer_load ... lds without passing through registers */
       SF_LAYOUT_F16_K_MINOR = 3,  /* B only, split precisions: B points to IEEE fp16 values B[n*ldb + k]
                                    (ldb, strideB in halfs; ldb % 2 == 0); used as they are, no lo part:
                                    a*b = ah*b + al*b.  The stored attention matrix (sf_softmax_rows).   */
       SF_LAYOUT_F16_KOCT = 5,   /* B only, F16X2 / F16 with a SPLIT_F16 A and M > 96 (the 128-row tile): IEEE fp16 in k-octet
                                    planes, element (k, n) at ((k/8)*ldb + n)*8 + k%8 (ldb = pixels per plane, strideB in
                                    halves; 16-byte aligned) -- what a producing sf_gemm stores with c_f16 = 2.  16 bytes =
                                    one MFMA operand octet of one pixel: the consumer moves its B tiles HBM/L2 -> LDS with
                                    buffer_load ... lds like the weights (no registers, no conversion, no ds_write).
                                    Octets past ceil(K/8) read as zero; rows K..8*ceil(K/8)-1 must be finite.          */
       SF_LAYOUT_SPLIT_KOCT = 6,  /* B only, F16X3 with a SPLIT_F16 A and M > 96 (the 128-row tile), no grouping: the activation
                                    ALREADY split, x = hi + lo, as two k-octet images of the F16_KOCT form -- hi planes
                                    [ceil(K/8)][ldb][8] at B, lo planes right behind them (ceil(K/8)*ldb*8 halves further);
                                    strideB in halves, 16-byte aligned -- what a producing sf_gemm stores with c_f16 = 4.  The
                                    same (hi, lo) the consumer would compute from the fp32 value (bit-identical results), at the
                                    same 4 bytes per element, but both operands now move HBM/L2 -> LDS by DMA: no conversion and
                                    no ds_write in the loop (fp32_class: the GEMM-to-GEMM tensors of an SK block).        */
       SF_LAYOUT_F16_K_MAJOR = 4 };/* B only, SF_PRECISION_F16X2 with a SPLIT_F16 A: IEEE fp16 rows B[k*ldb + n]
                                    (ldb, strideB in halfs; N, ldb, strideB even) -- what a producing sf_gemm stores
                                    with c_f16 = 1.  Bit-identical to handing the fp32 values over (F16X2 rounds a
                                    B operand to fp16 on load), at half the bytes.                          */
enum { SF_PRECISION_FP32 = 0,   /* exact fp32: v_mfma_f32_32x32x2_f32, k-ordered fmaf chain          */
       SF_PRECISION_F16X3 = 1,  /* split precision: x = hi+lo (fp16 each); a*b = ah*bh + ah*bl + al*bh
                                    on v_mfma_f32_32x32x16_f16 with fp32 accumulation (~2^-22 relative)  */
       SF_PRECISION_F16X2 = 2,  /* A (weights) split hi+lo, B (activations) rounded once to fp16:
                                    a*b = ah*b + al*b, 2 MFMAs; error = the fp16 rounding of B (2^-11 rel.) */
       SF_PRECISION_F16 = 3 };  /* both operands rounded once to fp16, ONE f16 MFMA per product, fp32 accumulation: the
                                    arithmetic class of the reference's own deployment (fp16 autocast, demo.py:427-456).
                                    sf_corr_build_pyramid / sf_corr_lookup: every pyramid cell is also STORED as fp16
                                    (the "bf16/fp16 volume" configurations of BASELINE.json).  sf_gemm: needs SPLIT_F16
                                    weights (their hi image is the round-to-nearest fp16 of the scaled weight); an fp32
                                    A operand is treated as in F16X2 */
enum { SF_EPI_NONE = 0,         /* C = v                              v = alpha*(acc+bias)   */
       SF_EPI_GELU = 1,         /* C = gelu(v)                        exact erf GELU         */
       SF_EPI_RELU = 2,         /* C = max(v,0)                                              */
       SF_EPI_RES = 3,          /* C = R + v                                                 */
       SF_EPI_RES_GELU = 4,     /* C = gelu(R + v)                                           */
       SF_EPI_RES_GELU_DW1 = 5, /* t = gelu(R + v); C = gelu(t + dw_w[m]*t + dw_b[m])        */
       SF_EPI_AXPY = 6 };       /* C = R + gamma[0]*v     (gma.py:102)                       */

typedef struct SfGemm {
    const float* A; const float* B; float* C;
    const float* bias;            /* [M] or NULL */
    const float* R;               /* residual (same row/col addressing as C, own strides) or NULL */
    const float* dw_w; const float* dw_b;   /* [M] each, for SF_EPI_RES_GELU_DW1 */
    const float* gamma;           /* device scalar for SF_EPI_AXPY */
    int32_t M, N, K, batch;
    int64_t lda, ldb, ldc, ldr;
    int64_t strideA, strideB, strideC, strideR;   /* per batch index z (floats) */
    int32_t a_layout, b_layout;
    /* grouped rows (K_MAJOR B, and R): row k lives at (k / group)*group_stride + (k % group)*ld.
       group = 0 means "no grouping".  Used to read '(B T) C H W -> B (T C) H W' views in place. */
    int32_t b_group; int64_t b_group_stride;
    int32_t r_group; int64_t r_group_stride;
    /* implicit 3x3 conv (pad 1) on a K_MAJOR B: K = 9*Cin, k = tap*Cin + c, tap = ky*3+kx; needs h*w == N */
    int32_t conv3x3, h, w;
    float alpha;
    int32_t epilogue;
    int32_t precision;            /* SF_PRECISION_* */
    const void* A_hi; const void* A_lo;   /* SF_LAYOUT_SPLIT_F16 operand (shared by all batch indices) */
    int64_t lda_h;                /* rows per k-octet plane of A_hi/A_lo (M padded to 128) */
    int32_t a_padded;             /* K_MAJOR fp32 A is zero padded to [K up to 32][M up to 128]: no bounds checks */
    /* split-K (precision F16X3 only): the K range is cut into k_splits slices, slice s writes its partial
       product (epilogue must be SF_EPI_NONE, no bias) to C + s*split_stride; combine with sf_splitk_combine. */
    int32_t k_splits; int64_t split_stride;
    /* optional scratch (caller-owned).  If given (>= auto_split_max * batch * M * N floats... see sf_gemm_split_ws_floats)
       and k_splits == 0, the library may split K on its own for small grids with deep K (F16X3 only): partial
       products go to the scratch and a second kernel applies bias / epilogue.  Results are deterministic. */
    float* split_ws; int64_t split_ws_floats;
    /* c_f16 = 1 (split precisions, 16-byte-aligned C, N % 4 == 0, ldc % 4 == 0): C points to IEEE fp16 storage, results are
       rounded to nearest and stored as halves (ldc, strideC in halves): the K-major fp16 operand of the next sf_gemm.
       c_f16 = 2: the same values in k-octet planes (SF_LAYOUT_F16_KOCT; ldc = pixels per plane, strideC in halves,
       8*ceil(M/8) rows are written: the caller provides room for them).
       c_f16 = 3: C is written in fp32 as usual AND a second time as fp16 k-octet planes at C16 (ldc pixels per plane,
       strideC16 halves between images; same alignment rules as c_f16 = 1 for C and c_f16 = 2 for C16; every epilogue).
       Only rows < M are written to C16: a last, partial octet keeps its other rows.
       c_f16 = 4 (F16X3 only): the result split as the next sf_gemm would split it, x = hi + lo, stored as the two k-octet images
       of SF_LAYOUT_SPLIT_KOCT (hi planes at C, lo planes ceil(M/8)*ldc*8 halves behind; strideC in halves; rules of c_f16 = 2). */
    int32_t c_f16;
    void* C16; int64_t strideC16;
    /* r_f16 = 2 (split precisions, vector epilogue: the alignment rules of c_f16 = 1 for C): the residual R is not fp32
       planes but an fp16 k-octet image (SF_LAYOUT_F16_KOCT: element (m, n) at ((m/8)*ldr + n)*8 + m%8, ldr = pixels per
       plane, strideR in halves, 16-byte aligned, no grouping) -- e.g. the tensor that was this block's GEMM operand, read
       a second time as the residual without an fp32 copy of it ever having been written (correlation features). */
    int32_t r_f16;
    /* SPLIT_F16 weights: the k extent the A_hi / A_lo planes are zero-padded to is K rounded up to a multiple of a_k_pad
       (0 or 32: the minimum the tiled kernels need; 128: what the activation-stationary kernel needs -- streamflow_amd
       packs 128). */
    int32_t a_k_pad;
    /* kernel family: SF_ALGO_AUTO picks per problem; the other two force one (tests, A/B timing) and fail when it cannot run */
    int32_t algo;
} SfGemm;
enum { SF_ALGO_AUTO = 0,
       SF_ALGO_TILED = 1,     /* 128 x 128 / 128 x 256 output tiles, operands staged per k-tile (csrc/gemm_split.hip) */
       SF_ALGO_BSTAT = 2 };   /* activation-stationary: a wave keeps the K values of its 32 pixels in registers and the weights
                                 stream past them through LDS (csrc/gemm_bstat.hip): F16X2 / F16, SPLIT_F16 weights with
                                 a_k_pad = 128, 64 < K <= 640, B as fp32 planes / fp16 rows / k-octets, C as fp32 planes and / or
                                 k-octets (c_f16 0, 2, 3) */

/* floats of scratch that let sf_gemm auto-split a problem of this size (0 if it never would) */
int64_t sf_gemm_split_ws_floats(int M, int N, int K, int batch);

int sf_gemm(const SfGemm* g, void* stream);

/* out[i] = R[i] + gamma[0] * sum_s partial[s*split_stride + i]  for n_img images of rows*P floats each
 * (image strides in floats): the AXPY epilogue of gma.py:102 applied after a split-K attn @ v. */
int sf_splitk_combine(const float* partial, int64_t split_stride, int k_splits, int64_t part_img_stride,
                      const float* R, int64_t r_img_stride, const float* gamma, float* out,
                      int64_t out_img_stride, int n_img, int64_t floats_per_img, void* stream);

/* ---- a6' + a7 fused: GMA aggregation without the N x N matrix (demo.py:235-258; == gma.py:53-65 + 91-104) ----------
 * out[img][d][p] = mf[img][d][p] + gamma[0] * sum_j softmax_j(scale * <q_p, k_j>) * v[img][d][j]     (heads = 1, dim 128)
 * The reference's demo recomputes softmax(q k^T) v in every refinement iteration (flash_attn_func or a naive einsum)
 * instead of keeping the attention matrix; this is that path: online softmax, logits never leave the CU.
 *   sf_gma_flash_pack_qk: once per clip.  qk [n_img][256][P] fp32 (rows 0..127 = q, 128..255 = k: the to_qk output,
 *       gma.py:57) -> fp16 (hi, lo) operand images in `ws` (q pre-scaled by scale * log2 e).  stats_qk_products = 1 / 2 / 3
 *       additionally runs the logits once (with that many products) and stores every query's softmax statistics (row
 *       maximum, 1 / row sum) in `ws`: q and k do not change over the refinement loop, so the per-iteration kernel can
 *       skip its running maximum, accumulator rescale and row sum (use_stats below; -21 % per iteration).  0 = none.
 *   sf_gma_flash_aggregate: every iteration.  v [n_img][128][P] (to_v output), mf and out [n_img][128][P] planes with
 *       image strides in floats; packs v into `ws`, then one fused kernel.  qk_products = MFMA products per logit:
 *       3 = split precision (q_hi k_hi + q_lo k_hi + q_hi k_lo, fp32-class logits), 2 = k rounded to fp16,
 *       1 = q and k rounded to fp16 (the arithmetic of the reference's fp16 flash-attn path).  Softmax weights and v
 *       enter the second contraction as fp16 (like the materialised matrix of sf_softmax_rows), accumulation is fp32.
 *       use_stats = 1: use the statistics stored by pack_qk, which must have been called with stats_qk_products ==
 *       qk_products (same logits); 0: self-contained online softmax.
 *       out_koct (optional): `out` a second time, rounded to fp16, as k-octet planes [16][P][8] per image
 *       (SF_LAYOUT_F16_KOCT, image stride in halves) -- the operand format of the GEMM that reads it next.
 * ws: caller-owned scratch of sf_gma_flash_ws_bytes(n_img, P) bytes, 16-byte aligned; must persist from pack_qk to the
 * last aggregate of the clip. */
int64_t sf_gma_flash_ws_bytes(int n_img, int P);
int sf_gma_flash_pack_qk(const float* qk, int64_t qk_img_stride, void* ws, int64_t ws_bytes, int n_img, int P,
                         float scale, int stats_qk_products, void* stream);
int sf_gma_flash_aggregate(void* ws, int64_t ws_bytes, const float* v, int64_t v_img_stride, const float* mf,
                           int64_t mf_img_stride, const float* gamma, float* out, int64_t out_img_stride,
                           void* out_koct, int64_t out_koct_img_stride, int n_img, int P, int qk_products, int use_stats,
                           void* stream);
/* The same with v as fp16 ROWS [n_img][128][P] (v_img_stride in halves: sf_gemm's c_f16 = 1 output of the to_v layer in the
 * config-2 presets): v enters the second contraction as fp16 either way, the to_v GEMM writes and the pack reads half the bytes. */
int sf_gma_flash_aggregate_f16v(void* ws, int64_t ws_bytes, const void* v_f16, int64_t v_img_stride, const float* mf,
                                int64_t mf_img_stride, const float* gamma, float* out, int64_t out_img_stride,
                                void* out_koct, int64_t out_koct_img_stride, int n_img, int P, int qk_products,
                                int use_stats, void* stream);
/* to_v (core/gma.py:93: v = to_v(fmap), 1x1 conv 128 -> 128, no bias) AND the v pack in one launch: v = fp16(alpha * W_v x) written
 * straight into the packed v planes of ws, from operands in the formats they already have -- x_koct: the k-octet fp16 copy of the
 * motion features [16][ldx][8] per image (SF_LAYOUT_F16_KOCT; x_koct_img_stride in halves, ldx = pixels per octet row >= P);
 * w_hi / w_lo: the split weight planes [128 / 8][lda_h = 128][8] of sf_gemm's SF_LAYOUT_SPLIT_F16 (products = 1: w_hi alone,
 * 2: w_hi + w_lo); alpha: 1 / the weights' power-of-two pre-scale.  A following sf_gma_flash_aggregate* call takes v == NULL
 * ("the v planes of ws are current").  The activation enters as fp16 (the config-2 arithmetic class). */
int sf_gma_flash_project_v(void* ws, int64_t ws_bytes, const void* x_koct, int64_t x_koct_img_stride, int64_t ldx,
                           const void* w_hi, const void* w_lo, int lda_h, float alpha, int products, int n_img, int P,
                           void* stream);

/* ---- a6 + a7 with the attention weights KEPT (core/gma.py:53-65 computes `attn` once per clip, gma.py:99-102 multiplies it every
 * iteration): the fused path above recomputes softmax(q k^T) in every iteration; q and k do not change over the refinement loop, so
 * the weights can be stored once -- as fp16, unnormalised (exp2 of the logit minus the stored row maximum: bit for bit what the fused
 * kernel multiplies), in the register image of the second contraction's B operand ([key tile of 64][query tile of 32][4] x 1 KB per
 * image, an opaque format between the two calls below) -- and every iteration only streams them past v:
 * half the matrix-core work, no exponentials, HBM-bound (n_img * Ppad^2 * 2 bytes per iteration, Ppad = P rounded up to 128).
 *   sf_gma_stored_p_bytes: size of `pbuf` (caller-owned, 16-byte aligned, persists over the clip's iterations).
 *   sf_gma_flash_store_p: once per clip, after sf_gma_flash_pack_qk(stats_qk_products = qk_products): writes pbuf.
 *   sf_gma_stored_aggregate: every iteration; out = mf + gamma / rowsum * v P^T.  v as in sf_gma_flash_aggregate (v_f16 = 1: fp16
 *       rows; v == NULL: the v planes of ws are current, sf_gma_flash_project_v).  With the same ws the result is bit-identical to
 *       sf_gma_flash_aggregate(use_stats = 1).  Weights stored by another pack call / product count poison the result (NaN). */
int64_t sf_gma_stored_p_bytes(int n_img, int P);
int sf_gma_flash_store_p(void* ws, int64_t ws_bytes, void* pbuf, int64_t pbuf_bytes, int n_img, int P, int qk_products,
                         void* stream);
int sf_gma_stored_aggregate(void* ws, int64_t ws_bytes, const void* pbuf, int64_t pbuf_bytes, const void* v, int v_f16,
                            int64_t v_img_stride, const float* mf, int64_t mf_img_stride, const float* gamma, float* out,
                            int64_t out_img_stride, void* out_koct, int64_t out_koct_img_stride, int n_img, int P, void* stream);

/* ---- a8: one FFN pair of an SK block in ONE launch (core/update.py:14-16, 30-36; csrc/ffn_pair.hip) ---------------------------
 * y = W2 gelu(W1 x + b1) + b2 with the 1.5 C hidden tensor kept in registers (the two-launch form writes and re-reads it).
 * X: the input as fp16 k-octet planes [K1/8][ldx][8] per image (SF_LAYOUT_F16_KOCT; strideX in halves), N pixels, `batch` images.
 * wstream: both layers' weights as ONE stream of 1-KB MFMA fragments in consumption order, built by the host
 *   (streamflow_amd.ops.PackedPair): per 32 hidden rows  2 * ceil(K1/32) * pm1 fragments of W1 (16 rows x 32 k each: lane
 *   (row, kq) = 8 halves W1[row][32 s + 8 kq ..]; with pm = 2 the `lo` fragment precedes the `hi` one), then ceil(M2/16) * pm2
 *   fragments of W2 whose 32 columns are ordered  k = 8 kq + i <-> hidden row 4 kq + i (i < 4) | 16 + 4 kq + i - 4 (i >= 4),
 *   zero-padded to sf_ffn_pair_frags(K1, M2, pm1, pm2) fragments.  Weights pre-scaled by a power of two per layer (alpha1 / alpha2
 *   undo it; bias1 / bias2 carry the same scale), rows / columns beyond H, K1, M2 zero.
 * mode 0 (an ffn2 pair): out = y, or gelu(y) with gelu_out, to C (fp32 planes [M2][ldc], optional) and / or C16 (fp16 k-octet
 *   planes [M2/8][ldc16][8], optional; c16_partial: rows >= M2 of the last octet are not written).
 * mode 1 (an ffn1 pair, M2 == K1): x1 = gelu(x + y); x2 = gelu(x1 + dw_w * x1 + dw_b)  (update.py:31-32: residual, then the
 *   depthwise 1x1 layer of conv_list); C16 = x2 as fp16 ROWS [M2][ldc16].  The residual is the fp16 operand itself.
 * Shapes built: the SK blocks of the update block (C = 128 / 256 / 324) and the flow head (384 / 256 / 128 -> 2 (T - 1) rows); see the
 * dispatch table in csrc/ffn_pair.hip. */
typedef struct SfFfnPair {
    const void* X; int64_t strideX; int64_t ldx;
    const void* wstream; int64_t wstream_bytes;
    const float* bias1; const float* bias2; const float* dw_w; const float* dw_b;
    float* C; int64_t strideC; int64_t ldc;
    void* C16; int64_t strideC16; int64_t ldc16;
    int32_t N, batch, K1, H, M2, pm1, pm2, mode, gelu_out, c16_partial;
    float alpha1, alpha2;
    int32_t x_group; int64_t x_group_stride;   /* x_group > 0: X's K1 rows are x_group-row slices (a multiple of 32 that divides K1) of
                                                  consecutive groups x_group_stride halves apart -- the flow head's '(B T) C -> B (T C)'
                                                  view of the hidden state (update.py:775) without a copy; 0: plain planes */
    const float* R32; int64_t strideR32, ldr32, r32_group_stride;   /* mode 1, optional: the residual x from fp32 planes [M2][ldr32]
                                                  (grouped like X: groups r32_group_stride floats apart) instead of the fp16 operand */
} SfFfnPair;
int sf_ffn_pair(const SfFfnPair* p, void* stream);
int sf_ffn_pair_frags(int K1, int M2, int pm1, int pm2);

/* ---- a8, back half: pw -> GELU -> ffn2.0 -> GELU -> ffn2.2 of an SK block in ONE launch (core/update.py:35-36 with ffn2 of
 * update.py:14-16; csrc/sk_tail.hip) ------------------------------------------------------------------------------------------
 * y = W3 gelu(W2 gelu(W1' x3 + b1) + b2) + b3,  W1' = pw + I (the residual of `x + pw(x)` folded into the weights).  x4 and the
 * 1.5 C hidden stay in registers (the three-launch form writes and re-reads both).
 * X: x3 as fp16 ROWS [C][ldx] per image (what sf_dwconv_res_gelu_f16in writes; strideX in halves), N pixels, `batch` images.
 * wstream: ONE stream of 1-KB fragments (32 rows x 16 k: lane (row m, k-half) = 8 halves W[32 t + m][k(8 khalf + i)]; pm = 2:
 *   the `lo` fragment precedes the `hi` one) in consumption order (streamflow_amd.ops.PackedTail):
 *     for t = 0 .. C/32 - 1: W1' row tile t, k-steps 0 .. C/16 - 1 (natural column order), zero-padded to a multiple of 16 fragments;
 *     for th = 0 .. H/32 - 1: W2 row tile th, k-steps 0 .. C/16 - 1; then for s = 0, 1: W3 row tiles 0 .. ceil(M2/32) - 1 at the
 *       k-step (th, s) of the hidden rows; zero-padded to a multiple of 16 fragments;
 *   the columns of a k-step (t, s) of W2 and W3 are ordered  k = 8 khalf + i <-> input row 32 t + 16 s + (i & 3) + 8 (i >> 2) + 4 khalf
 *   (the accumulator layout of the producing tile).  sf_sk_tail_frags(C, H, M2, pm) fragments in all (0: shape not built).
 *   Weights pre-scaled by a power of two per layer (alpha* = 1 / scale, bias* carry the scale).
 * Y (fp32 planes [M2][ldy], optional) and / or Y16 (fp16 k-octet planes, optional; y16_partial = 1: rows >= M2 of the last octet
 * are left alone).  gelu_out: y = gelu(...).  Built for (C, M2) = (256, 192), (256, 126), (384, 6), (128, 64), H % 32 == 0, H <= 576,
 * pm = 1 / 2; anything else: SF_ERR_BAD_ARG (the caller keeps the three launches). */
typedef struct SfSkTail {
    const void* X; int64_t strideX; int64_t ldx;
    const void* wstream; int64_t wstream_bytes;
    const float* bias1; const float* bias2; const float* bias3;
    float* Y; int64_t strideY; int64_t ldy;
    void* Y16; int64_t strideY16; int64_t ldy16;
    int32_t N, batch, C, H, M2, pm, gelu_out, y16_partial;
    float alpha1, alpha2, alpha3;
} SfSkTail;
int sf_sk_tail(const SfSkTail* p, void* stream);
int sf_sk_tail_frags(int C, int H, int M2, int pm);
/* The stream's unit structure (what the host packer needs besides the order above): returns sf_sk_tail_frags(); *stage = fragments per
 * stage (every unit is zero-padded to a multiple of it), *group = hidden tiles th per phase-2 unit, *pw_one_unit = 1 when the C/32 pw row
 * tiles form ONE unit instead of a unit each.  Any pointer may be NULL. */
int sf_sk_tail_layout(int C, int H, int M2, int pm, int* stage, int* group, int* pw_one_unit);

/* ---- a10: the temporal transformer block in ONE launch (core/update.py:459-484,502-513 -> timm Block; called at update.py:770;
 * csrc/temporal.hip) ---------------------------------------------------------------------------------------------------------
 * tokens of one pixel = its TT = T - 1 frames x C = 128 channels:  x += proj(softmax(C^-1/2 q k^T) v), (q, k, v) = qkv(LN1 x);
 * x += fc2(gelu(fc1(LN2 x))) -- the unfused form is sf_layernorm_cm, sf_gemm, sf_temporal_attn, sf_gemm, sf_layernorm_cm, sf_gemm,
 * sf_gemm.  Here every intermediate stays in registers (a wave owns 16 pixels x TT frames).
 * X16: the tokens as fp16 k-octet planes [C/8][ldx][8] per image, image = clip * TT + frame (strideX in halves); they are the
 *   operand AND the residual.  B clips of N pixels.
 * wstream: the four layers' weights as ONE stream of 1-KB fragments (16 rows x 32 k: lane (row, kq) = 8 halves W[row][k(8 kq ..)];
 *   with pm = 2 the `lo` fragment precedes the `hi` one) in consumption order (streamflow_amd.ops.PackedTemporal):
 *     q and k rows of qkv: row tile m = 0 .. 15, k-step s = 0 .. 3 (natural column order);
 *     then for p = 0 .. 3: v row tiles 16 + 2 p, 16 + 2 p + 1 (x 4 k-steps), then proj row tiles 0 .. 7 at k-step p;
 *     then for h = 0 .. 7: fc1 row tiles 2 h, 2 h + 1 (x 4 k-steps), then fc2 row tiles 0 .. 7 at k-step h;
 *   the columns of a k-step of proj, fc1 and fc2 are ordered  k = 8 kq + i <-> input row 4 kq + i (i < 4) | 16 + 4 kq + i - 4
 *   (i >= 4) of that k-step's 32 rows (the accumulator layout of the producing tiles).  sf_temporal_block_frags(pm) fragments.
 *   Weights pre-scaled by a power of two per layer: alpha_* = 1 / scale; bias_* carry the scale; ss_proj / ss_fc2 = the scale of
 *   proj / fc2 (the residual enters their accumulators).  qkv has no bias (timm: qkv_bias = False).
 * Y (fp32 planes [C][ldy], optional) and / or Y16 (their fp16 k-octet copy, optional): image stride strideY floats / strideY16 halves.
 * Built for C = 128, H = 256 (mlp_ratio 2), TT = 1 .. 3, pm = 1 / 2; anything else: SF_ERR_UNSUPPORTED (the caller keeps the
 * seven launches).  Arithmetic: activations enter every product as fp16; fp32 accumulation, LayerNorm, softmax and GELU. */
typedef struct SfTemporalBlock {
    const void* X16; int64_t strideX, ldx;
    const void* wstream; int64_t wstream_bytes;
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *bias_proj, *bias_fc1, *bias_fc2;
    float* Y; int64_t strideY, ldy;
    void* Y16; int64_t strideY16, ldy16;
    int32_t N, B, TT, C, H, pm;
    float alpha_qkv, alpha_proj, alpha_fc1, alpha_fc2, ss_proj, ss_fc2, eps, scale;
} SfTemporalBlock;
int sf_temporal_block(const SfTemporalBlock* p, void* stream);
int sf_temporal_block_frags(int pm);

/* ---- f2: mask head, second layer + convex upsampling in ONE launch (core/update.py:756-759,777 + core/models/streamflow.py:82-93;
 * csrc/mask_upsample.hip) ------------------------------------------------------------------------------------------------------
 * out[n][c][8 y + i][8 x + j] = sum_k softmax_k(mask[n][64 k + 8 i + j][y][x]) * 8 flow[n][c][y + k / 3 - 1][x + k % 3 - 1]  (0 outside),
 * mask = alpha * (W x + bias) with W the 576 x 256 weights of mask.2 and alpha = 0.25 / (the weights' power-of-two pre-scale): the
 * unfused form is sf_gemm (mask.2) + sf_upsample_flow, with the 576-channel mask written and read back in between.
 * X16: relu(mask.0(net)) as fp16 k-octet planes [256/8][ldx][8] per image (strideX in halves) -- sf_gemm's c_f16 = 3 copy;
 * wstream: the weights as 1-KB fragments (16 rows x 32 k, lane (row, kq) = 8 halves W[row][32 s + 8 kq ..]; pm = 2: `lo` before `hi`)
 *   in the order row tile 0 .. 35, k-step 0 .. 7 (streamflow_amd.ops.PackedMask): sf_mask_upsample_frags(pm) fragments;
 * bias: 576 floats carrying the pre-scale (or NULL); flow [n][2][h][w] fp32; out [n][2][8h][8w] fp32, 16-byte aligned. */
typedef struct SfMaskUpsample {
    const void* X16; int64_t strideX, ldx;
    const void* wstream; int64_t wstream_bytes;
    const float* bias; const float* flow; float* out;
    int32_t n_img, h, w, K, M, pm;
    float alpha;
} SfMaskUpsample;
int sf_mask_upsample(const SfMaskUpsample* p, void* stream);
int sf_mask_upsample_frags(int pm);

/* ---- row softmax (gma.py:63): x [rows][cols] ----------------------------------------------------
 * out_f16 == NULL: in place.  Otherwise the weights are written as IEEE fp16 to out_f16 [rows][cols] (x is then
 * scratch): the attention matrix is re-read by every refinement iteration's attn @ v and that read is HBM-bound,
 * so the split-precision modes store it in half the bytes (SF_LAYOUT_F16_K_MINOR below). */
int sf_softmax_rows(float* x, int64_t rows, int cols, void* out_f16, void* stream);

/* ---- depthwise KxK conv + bias + residual + GELU  (update.py:33-34 with kernel in {7,15}) -------
 * y = gelu(x + dwconv(x) + b);  plane (img, c) of x is the [h][w] map at x + img*x_img_stride + c*h*w
 * (same for y with y_img_stride); wgt [C][K][K], bias [C].  x and y must not overlap.
 * precision SF_PRECISION_FP32: fp32 FMA stencil on the VALU.  Split precisions: every kernel row is a banded
 * Toeplitz GEMM on the matrix cores with fp32 accumulation.  SF_PRECISION_F16X3: (hi, lo) fp16 halves of both
 * operands, three products.  SF_PRECISION_F16X2: sf_gemm's f16x2 arithmetic -- weights hi + lo, the activation
 * enters the products rounded to fp16 (two products).  SF_PRECISION_F16: the weights are rounded once to fp16 as well (one
 * product; a single-product layer of the mixed preset).  The residual x stays exact in every mode.
 * y_f16 = 1: y receives IEEE fp16 planes of the same [img][c][h][w] order, y_img_stride counted in halves -- the
 * hand-over to a GEMM that reads them as SF_LAYOUT_F16_K_MAJOR (the engine's pw layer in the f16x2 mode, whose
 * residual is folded into its weights so that x3 has no other reader). */
int sf_dwconv_res_gelu(const float* x, int64_t x_img_stride, const float* wgt, const float* bias, void* y,
                       int64_t y_img_stride, int y_f16, int n_img, int C, int h, int w, int ksize, int precision,
                       void* stream);
/* The same layer with x ALSO as fp16 rows [img][c][h][w] (both strides in halves): the config-2 hand-over of an SK block's
 * x2 (written by sf_gemm with c_f16 = 1) and x3.  precision SF_PRECISION_F16X2 or SF_PRECISION_F16 only.  The convolution
 * input and the residual are the fp16 value itself (what the two-product arithmetic multiplies anyway; the residual loses
 * the 2^-12 relative rounding of x2, which x3's own fp16 rounding already has).  Rows of whole, 16-byte aligned octets
 * (w % 8 == 0) are staged HBM/L2 -> LDS by DMA, double-buffered over the images of a workgroup; other widths through
 * registers. */
int sf_dwconv_res_gelu_f16in(const void* x_f16, int64_t x_img_stride, const float* wgt, const float* bias, void* y_f16,
                             int64_t y_img_stride, int n_img, int C, int h, int w, int ksize, int precision, void* stream);

/* ---- LayerNorm over channels of channel-major planes (update.py:462-463,481-483) --------------
 * x,y [n_img][C][P] (image strides given in floats), normalises each (img,p) column over C.
 * y_koct (optional, C = 128 / 256): the result as fp16 k-octet planes (SF_LAYOUT_F16_KOCT, image stride in halves) for
 * a consumer that is an sf_gemm; y may then be NULL. */
int sf_layernorm_cm(const float* x, int64_t x_img_stride, const float* gamma, const float* beta,
                    float* y, int64_t y_img_stride, void* y_koct, int64_t y_koct_img_stride, int n_img, int C, int P,
                    float eps, void* stream);

/* ---- per-pixel attention over the T-1 tokens (timm Attention core, update.py:466-474) ----------
 * qkv [B*TT][3*C][P] (rows [q|k|v]) -> out [B*TT][C][P]; softmax(q k^T / sqrt(C)) v over t.
 * out_koct (optional, C % 32 == 0): the result as fp16 k-octet planes [B*TT][C/8][P][8] (SF_LAYOUT_F16_KOCT); out may
 * then be NULL. */
int sf_temporal_attn(const float* qkv, float* out, void* out_koct, int B, int TT, int C, int P, void* stream);
/* The same with qkv as fp16 ROWS [B*TT][3C][P] (what sf_gemm writes with c_f16 = 1): the config-2 hand-over -- the qkv GEMM
 * writes and this kernel reads half the bytes; scores, softmax and the weighted sum stay fp32. */
int sf_temporal_attn_f16in(const void* qkv_f16, float* out, void* out_koct, int B, int TT, int C, int P, void* stream);

/* ---- fp32 planes -> fp16 k-octet planes (no reference counterpart: an operand format of sf_gemm) -----------------
 * x [n_img][rows][P] fp32 (x_img_stride in floats) -> y [n_img][ceil(rows/8)][P][8] IEEE fp16 (y_img_stride in halves),
 * element (r, p) at ((r / 8) * P + p) * 8 + r % 8; rows past `rows` in a last, partial octet are left untouched (they
 * must be finite: zero-initialise the planes once): SF_LAYOUT_F16_KOCT, the
 * image sf_gemm moves to LDS by DMA.  Used once per clip for the static context features; tensors produced inside
 * the loop get their k-octet copy from the producing kernel (SfGemm.C16). */
int sf_pack_koct(const float* x, int64_t x_img_stride, int n_img, int rows, int P, void* y, int64_t y_img_stride,
                 void* stream);

/* ---- measurement aid (no reference counterpart) -----------------------------------------------------------------
 * One wave that reads the shader-cycle counter and the constant 100 MHz counter for `spin_us` microseconds (sleeping in
 * between) and writes out[0] = shader cycles, out[1] = 100 MHz ticks that passed: launched on a side stream while the hot path
 * replays, it gives the clock the chip SUSTAINS under that load (MI355X clocks to its power budget; bench.py prices the
 * matrix-core roof at 2.4 GHz as the guide prescribes and quotes this clock next to it).  out: two int64 in device memory. */
int sf_clock_probe(int64_t* out, int spin_us, void* stream);

/* ---- context split (streamflow.py:119-122): cnets [n_img][2*hdim][P] ->
 * nets = tanh(first half) (written with nets_img_stride), inps = relu(second half). */
int sf_context_split(const float* cnets, float* nets, int64_t nets_img_stride, float* inps,
                     int64_t inps_img_stride, int n_img, int hdim, int P, void* stream);

/* ---- flow bookkeeping (streamflow.py:133,138) ----------------------------------------------------
 * coords1 += delta (if delta != NULL); flow = coords1 - grid; flow written to up to two places, and (flow_koct != NULL)
 * as fp16 into rows flow_koct_row (x), flow_koct_row + 1 (y) of k-octet planes (SF_LAYOUT_F16_KOCT, image stride in
 * halves): the flow rows of the k-octet copy of the motion features. */
int sf_flow_update(float* coords1, const float* delta, float* flow_a, int64_t flow_a_img_stride,
                   float* flow_b, int64_t flow_b_img_stride, void* flow_koct, int64_t flow_koct_img_stride,
                   int flow_koct_row, int n_img, int h, int w, void* stream);

/* ---- f1: Twins_CSC encoder (core/encoders/twins_csc.py:59-85 over timm's twins_svt_large stages 1-2) -----------------
 * Token planes [n_img][C][N]: N = H*W tokens of the (T*h) x w grid of a clip, channel = head*32 + d (head dim 32).
 * Linear layers / strided convs of the encoder use sf_gemm, LayerNorms sf_layernorm_cm; these three are the rest.
 * sf_window_attn: timm LocallyGroupedAttn core.  qkv [n_img][3C][H*W] (rows q | k | v, the qkv Linear output) ->
 *     out [n_img][C][H*W] = softmax(q k^T / sqrt(32)) v inside non-overlapping ws x ws windows.  Windows reaching past
 *     the grid are completed with zero tokens, whose k and v are qkv_bias (timm pads after the norm, before the Linear).
 * sf_window_attn_mfma: the same on the matrix cores, one wave per (window, head); precision as sf_subsample_attn_mfma.
 * sf_subsample_attn: timm GlobalSubSampleAttn core.  q [n_img][C][N], kv [n_img][2C][M] (rows k | v) -> out [n_img][C][N].
 * sf_subsample_attn_mfma: the same contraction on the matrix cores (flash-style, transposed logits, no N x M tensor):
 *     precision SF_PRECISION_F16X3 = hi + lo fp16 split of q, k, the softmax weights and v (3 products per contraction,
 *     fp32-class), SF_PRECISION_F16X2 / SF_PRECISION_F16 = every operand rounded once to fp16 (1 product); fp32
 *     accumulation, softmax statistics in fp32.  ws: sf_subsample_attn_ws_bytes(n_img, heads, M) bytes, 16-byte aligned
 *     (the packed k / v operand images; scratch, dead when the call's kernels have run).
 *     sf_window_attn_mfma, qkv_koct = 1 (one-product classes only): qkv points to fp16 k-octet planes [3C/8][H*W][8] (the
 *     qkv sf_gemm's c_f16 = 2 output; image stride in halves) instead of fp32 planes.
 *     Both *_mfma cores: out_koct (optional) = the result as fp16 k-octet planes [C/8][N][8] (SF_LAYOUT_F16_KOCT, image
 *     stride in halves): the B operand image of the proj sf_gemm that consumes it; out may then be NULL.
 * sf_dwconv3x3_res: timm PosConv: y = x + depthwise3x3(x) + b on [n_img][C][H][W]; w [C][9]. */
int sf_window_attn(const float* qkv, int64_t qkv_img_stride, const float* qkv_bias, float* out, int64_t out_img_stride,
                   int n_img, int C, int heads, int H, int W, int ws, void* stream);
int sf_window_attn_mfma(const void* qkv, int64_t qkv_img_stride, int qkv_koct, const float* qkv_bias, float* out,
                        int64_t out_img_stride, void* out_koct, int64_t out_koct_img_stride, int n_img, int C, int heads,
                        int H, int W, int ws, int precision, void* stream);
int sf_subsample_attn(const float* q, int64_t q_img_stride, const float* kv, int64_t kv_img_stride, float* out,
                      int64_t out_img_stride, int n_img, int C, int heads, int N, int M, void* stream);
int64_t sf_subsample_attn_ws_bytes(int n_img, int heads, int M);
int sf_subsample_attn_mfma(const float* q, int64_t q_img_stride, const float* kv, int64_t kv_img_stride, float* out,
                           int64_t out_img_stride, void* out_koct, int64_t out_koct_img_stride, int n_img, int C, int heads,
                           int N, int M, void* ws, int64_t ws_bytes, int precision, void* stream);
int sf_dwconv3x3_res(const float* x, int64_t x_img_stride, const float* w, const float* b, float* y,
                     int64_t y_img_stride, int n_img, int C, int H, int W, void* stream);

/* ---- K12 convex upsampling (streamflow.py:82-93) -------------------------------------------------
 * flow [n][2][h][w], mask [n][9*64][h][w] -> out [n][2][8h][8w]. */
int sf_upsample_flow(const float* flow, const float* mask, float* out, int n, int h, int w, void* stream);

/* ---- f4: forward_interpolate  (core/utils/utils.py:34-62; warm start, evaluate_mf.py:305,362) ----
 * flow, out [n_img][2][h][w] (ch0 = dx, ch1 = dy).  Each source pixel is pushed along its flow; points landing
 * outside the open rectangle (0, w) x (0, h) are dropped; every grid pixel receives the flow of the nearest
 * remaining point (exact Euclidean nearest neighbour in float64, lowest source index on ties), zero if none is left.
 * Replaces scipy.interpolate.griddata(method='nearest') on the host. */
int sf_forward_interpolate(const float* flow, float* out, int n_img, int h, int w, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STREAMFLOW_HIP_H */
