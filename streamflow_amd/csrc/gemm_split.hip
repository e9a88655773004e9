// Split-precision GEMM ("f16x3"): fp32-class accuracy at bf16/f16 matrix-core rate.
//
// Every fp32 operand x is split as x = hi + lo with hi = f16(x), lo = f16(x - hi)  (22 mantissa bits
// together), and a*b is accumulated in fp32 as  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  with three
// v_mfma_f32_32x32x16_f16 instead of eight v_mfma_f32_32x32x2_f32: 3*32 cycles instead of 8*64 per
// 32x32x16 block, i.e. 5.3x the exact-fp32 MFMA rate.  The dropped lo*lo term and the lo roundings are
// ~2^-22 relative; measured end-to-end effect on the flow fields is ~3e-6 px EPE, three
// orders inside the 1e-3 parity budget.  The reference itself deploys this network under fp16 autocast
// (demo.py:427-456), so every activation on this path is known to fit the f16 range.
//
// Same interface, tiling and epilogues as gemm.hip.  Differences:
//  * LDS holds the tiles ALREADY split, as f16 rows [x][k] (k contiguous, row stride BK+8 halfs = 80 B so
//    the 16-byte fragment reads of 16 consecutive rows cover all 64 banks once);
//  * weights are split once on the host into k-octet planes [K/8][M][8] (SF_LAYOUT_SPLIT_F16); the 128-row tile moves
//    them HBM/L2 -> LDS with buffer_load ... lds (no registers, no ds_write: the VGPR -> LDS store path, ~79 B/clk per
//    CU, is the scarcest resource of the loop), two LDS stages; fp32 activations are split
//    on the fly while being staged (K-major source: 8 row loads per thread = one k-octet of one pixel;
//    K-minor source: two 16-byte loads);
//  * every global read is a buffer load: the per-thread byte offset (voffset) is computed once before the
//    k-loop, the k-tile / row offset is a wave-uniform SGPR (soffset).  There is no branch and no 64-bit
//    address arithmetic in the loop.  Columns n >= N (rows m >= M) of a tile may load anything: an output
//    element only depends on its own row of A and column of B, and those outputs are never stored.  Rows
//    k >= K are forced to zero (clamped row + select on a wave-uniform or per-lane compare).
//  * template parameter PM = products per element: 3 (f16x3, above), 2 (f16x2: the activation is ONE fp16 -- rounded while
//    staged, or already stored as fp16 rows SF_LAYOUT_F16_K_MAJOR / k-octets SF_LAYOUT_F16_KOCT by its producer; the
//    k-octet form goes HBM/L2 -> LDS by DMA like the weights, so the loop has no VALU staging at all), 1 (f16);
//  * c_f16: the epilogue can hand the result to the next GEMM as fp16 rows (1), k-octets (2) or fp32 planes plus a k-octet
//    copy (3) -- see include/streamflow_hip.h and DESIGN.md section 5.
#include "sf_common.h"
#include "gemm_epilogue.h"
#include "split_operand.h"
#include <cstdlib>

namespace {

using sf::f32x16;
using sf::gemm_epilogue;

using namespace sf_split;

struct SplitArgs {
    SfGemm g;
    int64_t a_bytes, b_bytes;      // bytes spanned by one batch image of A / B (buffer range check)
#ifdef SF_GEMM_TIMERS
    long long* ts;                 // SF_GEMM_TS_BUF: per-workgroup phase timestamps (tools/gemm_one.py; built with
                                   // tools/build_variant.sh timers gemm_split.hip -DSF_GEMM_TIMERS)
#endif
};

#ifndef SF_GEMM_FAST_GELU
#define SF_GEMM_FAST_GELU 1      // A/B knob: 0 = the rational erf GELU in every mode
#endif
#ifndef SF_GEMM_FRAG_PREFETCH
#define SF_GEMM_FRAG_PREFETCH 0
#endif
#ifndef SF_GEMM_WAVES
#define SF_GEMM_WAVES 3          // workgroups per CU the register budget is sized for (128x128 tile)
#endif
// PM = MFMA products per element product: 3 = f16x3 (A and B hi + lo), 2 = f16x2 (A hi + lo, B one fp16), 1 = f16 (both one
// fp16: the arithmetic of an fp16-autocast deployment, with fp32 accumulation)
template <int WM, int WN, int TM, int TN, int ALAY, int BLAY, int PM>
// (the split k-octet form holds 64 KB of stages: two workgroups per CU)
__global__ __launch_bounds__(kThreads, (TM * TN >= 8 || BLAY == SF_LAYOUT_SPLIT_KOCT) ? 2 : SF_GEMM_WAVES) void gemm_f16x3_mfma(const SplitArgs args) {
    constexpr bool SB = (PM == 3), SA = (PM >= 2);
    const SfGemm& g = args.g;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    // B: [hi|lo][rows][LDK], ONE LDS stage (the next tile waits in registers).  A: the same for register-staged A; for
    // pre-split weights on the 128-row tile (kDmaA) two DMA-filled stages [hi|lo][4 k-octets][128 rows][8].  52 KB at
    // most, so 3 workgroups fit per CU.
    constexpr bool kDmaA = (ALAY == 2) && (BM == 128);
    constexpr int kAStage = (SA ? 2 : 1) * (BK / 8) * BM * 8;                    // halfs of one DMA stage (hi [+ lo])
    constexpr int kAHalfs = kDmaA ? 2 * kAStage : (SA ? 2 : 1) * BM * LDK;
#ifndef SF_GEMM_B2
#define SF_GEMM_B2 1     // two B stages where they fit (no lo plane + DMA-fed A): ONE barrier per k-tile instead of two
#endif
    // kB2: B double-buffered in LDS.  The staged registers of tile kt+1 are written to the OTHER stage right after the
    // MFMAs of tile kt (nobody reads that stage: it was last read in tile kt-1, a barrier ago), so the barrier that
    // separated "everyone is done reading" from the store disappears.
    // kDmaB: B arrives as fp16 k-octet planes (SF_LAYOUT_F16_KOCT) and is DMA-fed like A: two stages [4 k-octets][BN][8]
    // kDmaB2 (SF_LAYOUT_SPLIT_KOCT, f16x3): the activation arrives ALREADY split, hi and lo k-octet images: a stage is [hi | lo]
    constexpr bool kDmaB2 = (BLAY == SF_LAYOUT_SPLIT_KOCT);
    constexpr bool kDmaB = (BLAY == SF_LAYOUT_F16_KOCT) || kDmaB2;
    static_assert(!kDmaB || (kDmaA && SB == kDmaB2), "k-octet B needs the DMA-fed 128-row tile; a lo plane exactly in the split form");
    constexpr int kBPlane = (BK / 8) * BN * 8;                                   // halfs of one plane of a DMA stage of B
    constexpr int kBStage = (kDmaB2 ? 2 : 1) * kBPlane;                          // halfs of one DMA stage of B (hi [+ lo])
    constexpr bool kB2 = SF_GEMM_B2 && !SB && kDmaA && !kDmaB;
    constexpr int kMainHalfs = kAHalfs + (kDmaB ? 2 * kBStage : (SB ? 2 : (kB2 ? 2 : 1)) * BN * LDK);   // SB = false: no lo part
    constexpr int kEpiHalfs = 4 * sf::kEpiScratchFloats * 2;
    __shared__ __attribute__((aligned(1024))) _Float16 smem[kMainHalfs > kEpiHalfs ? kMainHalfs : kEpiHalfs];
    _Float16 (*sA)[BM * LDK] = reinterpret_cast<_Float16 (*)[BM * LDK]>(smem);
    _Float16 (*sB)[BN * LDK] = reinterpret_cast<_Float16 (*)[BN * LDK]>(smem + kAHalfs);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int mt = (g.M + BM - 1) / BM, nt = (g.N + BN - 1) / BN;
    const sf::TileCoord tc = sf::xcd_tile(blockIdx.x, gridDim.x, mt, nt);
    const int ksp = g.k_splits > 1 ? g.k_splits : 1;
    const int n0 = tc.n_tile * BN, m0 = tc.m_tile * BM, z = tc.z / ksp, split = tc.z % ksp;

#ifdef SF_GEMM_TIMERS
    const long long ts0 = __builtin_readcyclecounter();
    const long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    typedef typename OperandSel<BN, BLAY>::type OperandB;
    Operand<BM, ALAY> opa;
    OperandB opb;
    typename Operand<BM, ALAY>::Regs ra;
    typename OperandB::Regs rb;
    if (ALAY == 2) opa.init(g.A_hi, g.A_lo, args.a_bytes, (int)g.lda_h, g.K, g.M, m0, 0, 0, tid);
    else opa.init(g.A + (int64_t)z * g.strideA, nullptr, args.a_bytes, (int)g.lda, g.K, g.M, m0, 0, 0, tid);
    opb.init(reinterpret_cast<const char*>(g.B) + (int64_t)z * g.strideB * ((BLAY == SF_LAYOUT_F16_K_MINOR || BLAY == SF_LAYOUT_F16_K_MAJOR) ? 2 : 4), nullptr,
             args.b_bytes, (int)g.ldb, g.K, g.N, n0, g.b_group, g.b_group_stride, tid);
    const bool conv = (BLAY == SF_LAYOUT_K_MAJOR) && g.conv3x3;
    const int cin = conv ? g.K / 9 : 1;
    if (BLAY == SF_LAYOUT_K_MAJOR && conv) opb.set_conv3x3(n0, g.h, g.w, tid);
    // per k-tile of an implicit 3x3 conv: tap, row offset of channel c0 = k0 % cin, byte shift of the tap
    auto conv_tap = [&](int k0) { return k0 / cin; };
    auto conv_off = [&](int k0) { return (k0 % cin) * (int)g.ldb; };
    auto conv_shift = [&](int k0) { const int t = k0 / cin; return ((t / 3 - 1) * g.w + (t % 3 - 1)) * 4; };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk_all = (g.K + BK - 1) / BK;
    const int kt_beg = (int)((int64_t)nk_all * split / ksp), kt_end = (int)((int64_t)nk_all * (split + 1) / ksp);
    RowCursor ca, cb;
    ca.init(0, (int)g.lda, 0);
    cb.init(g.b_group, (int)g.ldb, g.b_group_stride);
    for (int t = 0; t < kt_beg; ++t) { ca.advance(); cb.advance(); }       // split-K: start of this slice
    // ---- A by LDS-DMA (kDmaA): slot = k-octet * 128 + row = j * 256 + tid for the j-th of two 4 KB pieces per plane ----
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int a_plane = (int)(args.a_bytes);                                       // bytes of the hi (= lo) plane
    const __amdgpu_buffer_rsrc_t rah = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A_hi), 0, a_plane, 0x00020000);
    const __amdgpu_buffer_rsrc_t ral = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A_lo), 0, a_plane, 0x00020000);
    const int voa0 = ((tid >> 7) * (int)g.lda_h + m0 + (tid & 127)) * 16, voa1 = voa0 + 2 * (int)g.lda_h * 16;
    auto issue_a = [&](int kt, int buf) {
        char* dst = reinterpret_cast<char*>(smem) + buf * kAStage * 2 + wave_u * 1024;
        const int so = kt * (BK / 8) * (int)g.lda_h * 16;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rah, (lds_ptr)(dst), 16, voa0, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rah, (lds_ptr)(dst + 4096), 16, voa1, so, 0, 0);
        if (SA) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ral, (lds_ptr)(dst + kAStage), 16, voa0, so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ral, (lds_ptr)(dst + kAStage + 4096), 16, voa1, so, 0, 0);
        }
    };

    // ---- B by LDS-DMA (kDmaB): slot = k-octet * BN + pixel = j * 256 + tid; pixels past N are clamped (never stored) ----
    const int b_plane = (int)(args.b_bytes);
    const __amdgpu_buffer_rsrc_t rbd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(g.B)) + (int64_t)z * g.strideB * 2, 0, b_plane, 0x00020000);
    static_assert(!kDmaB || BN == 128 || BN == 256, "KOCT B: 128 or 256 pixel columns per workgroup");
    constexpr int kBPieces = BN / 64;                      // 4 KB DMA pieces per stage: piece j = slots j*256 .. j*256+255
    const int bpx = min(n0 + (tid & (BN - 1)), g.N - 1);
    const int vob0 = ((tid / BN) * (int)g.ldb + bpx) * 16, vobs = (kThreads / BN) * (int)g.ldb * 16;
    // b_group > 0 ('(B T) C -> B (T C)' views): rows come in groups of b_group (a multiple of 32, so a k-tile never
    // straddles two), group gi starts b_group_stride halves after group gi - 1
    const int b_goct = g.b_group > 0 ? g.b_group / 8 : 0;
    const int b_lo_off = ((g.K + 7) / 8) * (int)g.ldb * 16;                        // (kDmaB2) bytes from the hi image to the lo image
    auto issue_b = [&](int kt, int buf) {
        char* dst = reinterpret_cast<char*>(smem + kAHalfs) + buf * kBStage * 2 + wave_u * 1024;
        const int o = kt * (BK / 8), gi = b_goct ? o / b_goct : 0;
        const int so = (o - gi * b_goct) * (int)g.ldb * 16 + gi * (int)(g.b_group_stride * 2);
#pragma unroll
        for (int j = 0; j < kBPieces; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rbd, (lds_ptr)(dst + j * 4096), 16, vob0 + j * vobs, so, 0, 0);
        if (kDmaB2) {
#pragma unroll
            for (int j = 0; j < kBPieces; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rbd, (lds_ptr)(dst + kBPlane * 2 + j * 4096), 16, vob0 + b_lo_off + j * vobs, so, 0, 0);
        }
    };

    // the A piece is requested BEFORE the B loads of the same k-tile: vmcnt retires in order, so the wait that the B
    // registers need also covers the DMA
    if (kDmaA) issue_a(kt_beg, 0);
    else opa.load(kt_beg * BK, ca.off, ra);
    if (kDmaB) issue_b(kt_beg, 0);
    if (conv) opb.load(kt_beg * BK, conv_off(kt_beg * BK), rb, conv_shift(kt_beg * BK));
    else opb.load(kt_beg * BK, cb.off, rb);
    if (!kDmaA) opa.template store<SA>(kt_beg * BK, sA[0], sA[SA ? 1 : 0], ra);
    opb.template store<SB>(kt_beg * BK, sB[0], sB[SB ? 1 : 0], rb, conv ? conv_tap(kt_beg * BK) : -1);
    if (kDmaA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef SF_GEMM_TIMERS
    const long long ts1 = __builtin_readcyclecounter();
#endif

    const int khalf = lane >> 5, l31 = lane & 31;
    for (int kt = kt_beg; kt < kt_end; ++kt) {
        const int abuf = (kt - kt_beg) & 1;
        if (kt + 1 < kt_end) {
            ca.advance();
            cb.advance();
            if (kDmaA) issue_a(kt + 1, abuf ^ 1);          // the other stage was last read one k-tile (two barriers) ago
            else opa.load((kt + 1) * BK, ca.off, ra);
            if (kDmaB) issue_b(kt + 1, abuf ^ 1);
#ifndef SF_ABLATE_B          // timing ablation only (wrong results): B tile staged once, never refreshed
            if (conv) opb.load((kt + 1) * BK, conv_off((kt + 1) * BK), rb, conv_shift((kt + 1) * BK));
            else opb.load((kt + 1) * BK, cb.off, rb);
#endif
        }
        // keep the staged loads in flight across the MFMA block: nothing below may be hoisted above it
        __builtin_amdgcn_sched_barrier(0);
        // A fragment of lane (row l31, k-half): register-staged image rows [x][LDK]; DMA image [k-octet][128 rows][8]
        const _Float16* pah = kDmaA ? smem + abuf * kAStage + (khalf * BM + wm * TM * 32 + l31) * 8
                                    : sA[0] + (wm * TM * 32 + l31) * LDK + khalf * 8;
        const _Float16* pal = kDmaA ? pah + kAStage / 2 : sA[SA ? 1 : 0] + (wm * TM * 32 + l31) * LDK + khalf * 8;   // (SA only)
        constexpr int kATile = kDmaA ? 32 * 8 : 32 * LDK;        // halfs between the 32-row tiles of a wave
        constexpr int kAStep = kDmaA ? 2 * BM * 8 : 16;          // halfs per 16-deep k-step
        const int bbuf = kB2 ? ((kt - kt_beg) & 1) : 0;
        constexpr int kBTile = kDmaB ? 32 * 8 : 32 * LDK;        // halfs between the 32-column tiles of a wave
        constexpr int kBStep = kDmaB ? 2 * BN * 8 : 16;          // halfs per 16-deep k-step
        const _Float16* pbh = kDmaB ? smem + kAHalfs + abuf * kBStage + (khalf * BN + wn * TN * 32 + l31) * 8
                                    : sB[bbuf] + (wn * TN * 32 + l31) * LDK + khalf * 8;
        const _Float16* pbl = kDmaB2 ? pbh + kBPlane : sB[SB ? 1 : 0] + (wn * TN * 32 + l31) * LDK + khalf * 8;
#if SF_GEMM_FRAG_PREFETCH
        // all fragment reads of the k-tile are issued before its first MFMA (2 k-steps x 8 x ds_read_b128 = 64 VGPRs):
        // the LDS latency is exposed once per k-tile instead of once per register reuse (hipcc otherwise recycles 24
        // fragment registers and waits lgkmcnt(0) three times per k-step)
        f16x8 ah[BK / 16][TM], al[BK / 16][TM], bh[BK / 16][TN], bl[BK / 16][TN];
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[ks][i] = *reinterpret_cast<const f16x8*>(pah + i * kATile + ks * kAStep);
                if (SA) al[ks][i] = *reinterpret_cast<const f16x8*>(pal + i * kATile + ks * kAStep);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[ks][j] = *reinterpret_cast<const f16x8*>(pbh + j * kBTile + ks * kBStep);
                if (SB) bl[ks][j] = *reinterpret_cast<const f16x8*>(pbl + j * kBTile + ks * kBStep);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            // product-major order: the three MFMAs that hit one accumulator are TM*TN instructions apart
            if (SA) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
            }
            if (SB) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][i], bl[ks][j], acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
        }
#else
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const f16x8*>(pah + i * kATile + ks * kAStep);
                if (SA) al[i] = *reinterpret_cast<const f16x8*>(pal + i * kATile + ks * kAStep);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const f16x8*>(pbh + j * kBTile + ks * kBStep);
                if (SB) bl[j] = *reinterpret_cast<const f16x8*>(pbl + j * kBTile + ks * kBStep);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    // small terms first, so the dominant hi*hi product is added to an already-formed correction
                    if (SA) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    if (SB) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < kt_end) {
            if (!kB2 && !kDmaB) __syncthreads();   // every wave is done reading tile kt
            if (!kDmaA) opa.template store<SA>((kt + 1) * BK, sA[0], sA[SA ? 1 : 0], ra);
#ifndef SF_ABLATE_B
            opb.template store<SB>((kt + 1) * BK, sB[kB2 ? (bbuf ^ 1) : 0], sB[SB ? 1 : 0], rb, conv ? conv_tap((kt + 1) * BK) : -1);
#endif
            if (kDmaA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (already retired by the wait for rb)
            __syncthreads();
        }
    }
#ifdef SF_GEMM_TIMERS
    const long long ts2 = __builtin_readcyclecounter();
#endif
    // the two-product modes evaluate GELU as a polynomial (sf_common.h) WHERE THE RESULT LEAVES AS fp16 (the k-octet
    // epilogue): a tenth of the rounding it receives there; results kept in fp32 use the erf-rational form in every mode
    constexpr bool kFastGelu = (PM <= 2) && SF_GEMM_FAST_GELU;
    SfGemm gs = g;
    if (ksp > 1) gs.C = g.C + (int64_t)split * g.split_stride;   // partial product of this K slice: its own slab
    if (gs.c_f16 == 2) {
        __syncthreads();                                         // the main-loop LDS becomes the transpose scratch
        sf::gemm_epilogue_koct<WM, WN, TM, TN, kFastGelu>(gs, acc, m0, n0, z, wm, wn, lane,
                                               reinterpret_cast<float*>(smem) + wave * sf::kEpiScratchFloats);
    } else if (PM == 3 && gs.c_f16 == 4) {                       // the (hi, lo) pair the next f16x3 GEMM takes by DMA
        __syncthreads();
        sf::gemm_epilogue_koct<WM, WN, TM, TN, false, PM == 3>(gs, acc, m0, n0, z, wm, wn, lane,
                                                               reinterpret_cast<float*>(smem) + wave * sf::kEpiScratchFloats);
    } else if (sf::epilogue_vec_ok(gs, z)) {
        __syncthreads();                                         // the main-loop LDS becomes the transpose scratch
        sf::gemm_epilogue_vec<WM, WN, TM, TN>(gs, acc, m0, n0, z, wm, wn, lane,
                                              reinterpret_cast<float*>(smem) + wave * sf::kEpiScratchFloats);
    } else {
        gemm_epilogue<WM, WN, TM, TN>(gs, acc, m0, n0, z, wm, wn, lane);
    }
#ifdef SF_GEMM_TIMERS
    if (args.ts && tid == 0) {
        long long* d = args.ts + (int64_t)blockIdx.x * 8;
        d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = __builtin_readcyclecounter();
        d[4] = rt0; d[5] = __builtin_amdgcn_s_memrealtime();
        d[6] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));      // XCC_ID[3:0]
        d[7] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_ID
    }
#endif
}


// ---- "B-direct" kernel: split weights x fp16 k-octet activations (SF_LAYOUT_SPLIT_F16 x SF_LAYOUT_F16_KOCT) ----------------
// 128 (rows) x 256 (pixels) tile, 8 waves as 2 (rows) x 4 (pixels), each 64 x 64 = 2 x 2 MFMA tiles.
//  * WEIGHTS (shared by all waves) go L2 -> LDS by DMA into a ring of three 32-deep stages: ONE 1-KB piece per plane
//    (hi, lo) per wave and stage, requested two stages ahead, waited for with a counted vmcnt;
//  * ACTIVATIONS are private to a wave column, and in the k-octet planes a lane's MFMA B fragment (8 consecutive k of one
//    pixel) is one aligned 16-byte piece: every wave loads its own fragments straight from L2 into REGISTERS with plain
//    buffer loads, one stage ahead.  No LDS, no DMA, no ds_read for them.
// Why: the 128 x 128 kernel above moves BOTH operands by LDS-DMA -- 6 pieces per wave per 16 MFMAs -- and a wave sits
// ~190 cycles in each such instruction in a busy CU (measured in the correlation build): more
// than the MFMAs take.  Here it is 2 pieces and 4 plain loads per 16 MFMAs, a third of the LDS reads, one barrier per
// 16 MFMAs of EIGHT waves, and 16 waves per CU (128 registers) instead of 12.
// Same epilogues (gemm_epilogue.h) as the 128 x 128 kernel.
constexpr int kBdThreads = 512;
// wave grid WM x (8 / WM): 2 x 4 (64 x 64 per wave: every activation fragment is loaded by two waves) or 1 x 8 (each wave all
// 128 rows x 32 pixels: no duplicate activation loads, but every wave re-reads all weight fragments from LDS)
// BM = 256 (single-product weights only: three 16-KB stages): 2 x 4 waves of 128 x 64 -- every activation fragment feeds FOUR
// MFMAs of a wave instead of two and is pulled from L2 once per 256 output rows instead of once per 128: the activations
// (N = images x pixels columns) are the large operand of every GEMM of the update block, the weights are not.
template <int PM, int WM, int BM>
__device__ __forceinline__ void gemm_bdirect_body(const SplitArgs& args) {
    const SfGemm& g = args.g;
    static_assert(BM == 128 || (BM == 256 && PM == 1 && WM == 2), "the 256-row tile is built for one product, 2 x 4 waves");
    constexpr int BN = 256, WN = 8 / WM, TM = BM / 32 / WM, TN = WM, NST = 3;
    constexpr int NPC = BM / 128;                                    // DMA pieces per wave, plane and stage
    constexpr int kPlane = (BK / 8) * BM * 16;                       // bytes of one plane (hi or lo) of a stage: 8 KB
    constexpr int kStage = PM * kPlane;
    constexpr int kEpiBytes = 8 * sf::kEpiScratchFloats * 4;
    __shared__ __attribute__((aligned(1024))) char smem[NST * kStage > kEpiBytes ? NST * kStage : kEpiBytes];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int khalf = lane >> 5, l31 = lane & 31;
    const int mt = (g.M + BM - 1) / BM, nt = (g.N + BN - 1) / BN;
    const sf::TileCoord tc = sf::xcd_tile(blockIdx.x, gridDim.x, mt, nt);
    const int n0 = tc.n_tile * BN, m0 = tc.m_tile * BM, z = tc.z;
    const int nk = (g.K + BK - 1) / BK;

    // ---- weights by DMA: slot = k-octet * 128 + row = tid (512 slots = one 8-KB plane of a stage) ----
    const int a_plane = (int)(args.a_bytes);
    const __amdgpu_buffer_rsrc_t rah = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A_hi), 0, a_plane, 0x00020000);
    const __amdgpu_buffer_rsrc_t ral = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A_lo), 0, a_plane, 0x00020000);
    int voa[NPC];
#pragma unroll
    for (int p_ = 0; p_ < NPC; ++p_) {
        const int sl = tid + p_ * kBdThreads;                        // slot = k-octet * BM + row
        voa[p_] = ((sl / BM) * (int)g.lda_h + m0 + (sl % BM)) * 16;
    }
    auto issue_a = [&](int kt, int slot) {
        const int so = kt * (BK / 8) * (int)g.lda_h * 16;
#pragma unroll
        for (int p_ = 0; p_ < NPC; ++p_) {
            char* dst = smem + slot * kStage + (p_ * 8 + wave) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rah, (lds_ptr)(dst), 16, voa[p_], so, 0, 0);
            if (PM == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(ral, (lds_ptr)(dst + kPlane), 16, voa[p_], so, 0, 0);
        }
    };
    // ---- activations straight into registers: fragment (j, k-step ks) of stage kt = k-octet 4 kt + 2 ks + khalf of pixel
    // n0 + (wn * 2 + j) * 32 + l31 (pixels past N clamped: never stored) ----
    const __amdgpu_buffer_rsrc_t rbd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(g.B)) + (int64_t)z * g.strideB * 2, 0, (int)args.b_bytes, 0x00020000);
    int vob[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) vob[j] = (khalf * (int)g.ldb + min(n0 + (wn * TN + j) * 32 + l31, g.N - 1)) * 16;
    const int b_goct = g.b_group > 0 ? g.b_group / 8 : 0;           // grouped rows: see the 128 x 128 kernel
    // (two explicit register sets, the k-loop unrolled by two stages: indexed by kt & 1 the set becomes a dynamically indexed
    // register array -- s_set_gpr_idx + v_mov behind a vmcnt(0) after every single load)
    u32x4 bq0[BK / 16][TN], bq1[BK / 16][TN];
    auto load_b = [&](int kt, u32x4 (&bq)[BK / 16][TN]) {
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int o = kt * (BK / 8) + ks * 2, gi = b_goct ? o / b_goct : 0;
            const int so = (o - gi * b_goct) * (int)g.ldb * 16 + gi * (int)(g.b_group_stride * 2);
#pragma unroll
            for (int j = 0; j < TN; ++j) bq[ks][j] = __builtin_amdgcn_raw_buffer_load_b128(rbd, vob[j], so, 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // issue order matters (vmcnt retires in order): B(0), DMA(0), DMA(1); then per stage kt: B(kt + 1), DMA(kt + 2)
    load_b(0, bq0);
    issue_a(0, 0);
    issue_a(1, 1);
    const int offa = (khalf * BM + wm * TM * 32 + l31) * 16;
    int slot = 0, slot2 = 2;
#ifdef SF_GEMM_TIMERS
    const long long ts0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
    long long tw = 0, tb = 0, ti = 0, tm = 0, tprev = ts0;
#define SF_BD_STAMP(acc_) { const long long t_ = __builtin_readcyclecounter(); acc_ += t_ - tprev; tprev = t_; }
#else
#define SF_BD_STAMP(acc_)
#endif
    auto stage = [&](int kt, u32x4 (&bcur)[BK / 16][TN], u32x4 (&bnext)[BK / 16][TN]) {
        SF_BD_STAMP(tm)
        // stage kt of the weights has landed (this wave's pieces; the barrier makes it everyone's): behind it in the queue
        // are only B(kt) -- needed now anyway -- and the PM pieces of DMA(kt + 1)
        __builtin_amdgcn_s_waitcnt(0x0F70 | (PM * NPC));                             // vmcnt(PM * NPC)
        SF_BD_STAMP(tw)
        __builtin_amdgcn_s_barrier();                       // ... and slot2 (stage kt - 1) is no longer read by anyone
        SF_BD_STAMP(tb)
        // (unconditional, also past the last stage -- out-of-range buffer reads return zeros into registers / a slot nobody
        // reads: with the same number of memory operations behind every load on every path, hipcc's vmcnt waits are exact;
        // with `if (kt + 1 < nk)` around them it sizes them for the shortest path, i.e. vmcnt(1) and (0))
        load_b(kt + 1, bnext);
        issue_a(kt + 2, slot2);
        SF_BD_STAMP(ti)
        const char* sa = smem + slot * kStage + offa;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            f16x8 ah[TM], al[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const f16x8*>(sa + ks * 2 * BM * 16 + i * 32 * 16);
                if (PM == 2) al[i] = *reinterpret_cast<const f16x8*>(sa + kPlane + ks * 2 * BM * 16 + i * 32 * 16);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const f16x8 b = __builtin_bit_cast(f16x8, bcur[ks][j]);
                    if (PM == 2) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], b, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], b, acc[i][j], 0, 0, 0);
                }
        }
        slot = (slot == NST - 1) ? 0 : slot + 1;
        slot2 = (slot2 == NST - 1) ? 0 : slot2 + 1;
    };
    for (int kt = 0; kt < nk; kt += 2) {                     // (an odd last stage multiplies zeros: nk is rounded up to even)
        stage(kt, bq0, bq1);
        stage(kt + 1, bq1, bq0);
    }
    constexpr bool kFastGelu = SF_GEMM_FAST_GELU;
    // the stages issue DMA pieces past the end of K unconditionally (zero fills): drain them explicitly before the stage ring
    // becomes the epilogues' transpose scratch -- a late piece would otherwise land in another wave's scratch
    __builtin_amdgcn_s_waitcnt(0x0070);                      // vmcnt(0) lgkmcnt(0)
    __syncthreads();
#ifdef SF_GEMM_TIMERS
    const long long ts2 = __builtin_readcyclecounter();
#endif
    float* scratch = reinterpret_cast<float*>(smem) + wave * sf::kEpiScratchFloats;
    if (g.c_f16 == 2) sf::gemm_epilogue_koct<WM, WN, TM, TN, kFastGelu>(g, acc, m0, n0, z, wm, wn, lane, scratch);
    else if (sf::epilogue_vec_ok(g, z)) sf::gemm_epilogue_vec<WM, WN, TM, TN>(g, acc, m0, n0, z, wm, wn, lane, scratch);
    else gemm_epilogue<WM, WN, TM, TN>(g, acc, m0, n0, z, wm, wn, lane);
#ifdef SF_GEMM_TIMERS
    if (args.ts && lane == 0 && blockIdx.x < 8192) {
        long long* d = args.ts + ((int64_t)blockIdx.x * 8 + wave) * 8;
        d[0] = ts0; d[1] = tw; d[2] = tb; d[3] = ti; d[4] = tm; d[5] = ts2 - ts0; d[6] = __builtin_readcyclecounter() - ts2;
        d[7] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
#endif
}
// (two __global__ wrappers: a __launch_bounds__ that depends on a template parameter left hipcc without the host stub of
// some instantiations -- "undefined symbol __device_stub__gemm_bdirect_kernel<1, 2, 256>" at load time)
// Wave grid: 1 x 8 waves of 128 x 32 (every activation fragment loaded once, used for four MFMAs) is what ships.  The 2 x 4
// grid of 64 x 64 measured equal within the noise for two-product layers and slower for single-product ones (DESIGN.md
// section 11), and at its 128-register cap its epilogue spilled 140 VGPRs; the 256-row tile was no better
// than 1 x 8 either.  Both stay reachable for experiments: tools/build_variant.sh ... -DSF_GEMM_BD_WM=2 / -DSF_GEMM_BD256_MIN_M=..
#ifndef SF_GEMM_BD_WM
#define SF_GEMM_BD_WM 1
#endif
#ifndef SF_GEMM_BD256_MIN_M
#define SF_GEMM_BD256_MIN_M 0            // 0 = never
#endif
template <int PM>
__global__ __launch_bounds__(kBdThreads, 4) void gemm_bdirect_kernel(const SplitArgs args) { gemm_bdirect_body<PM, SF_GEMM_BD_WM, 128>(args); }
#if SF_GEMM_BD256_MIN_M
__global__ __launch_bounds__(kBdThreads, 2) void gemm_bdirect256_kernel(const SplitArgs args) { gemm_bdirect_body<1, 2, 256>(args); }
#endif

template <int WM, int WN, int TM, int TN, int PM>
int launch_cfg(const SplitArgs& a, hipStream_t st) {
    constexpr bool SB = (PM == 3);
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const SfGemm& g = a.g;
    dim3 grid(sf::ceil_div(g.N, BN) * sf::ceil_div(g.M, BM) * g.batch * (g.k_splits > 1 ? g.k_splits : 1));   // 1-D: see sf::xcd_tile
    const int lay = g.a_layout * 4 + g.b_layout;
    if (lay == 7) {
        if constexpr (!SB && TM * TN == 4) {             // only the 128x128 tile is built for the stored-fp16 B
            hipLaunchKernelGGL((gemm_f16x3_mfma<WM, WN, TM, TN, 1, 3, 2>), grid, dim3(kThreads), 0, st, a);
            return sf::check_launch("sf_gemm(f16x3)");
        }
        return sf::fail(SF_ERR_UNSUPPORTED, "sf_gemm(f16x3): SF_LAYOUT_F16_K_MINOR B not built for this tile");
    }
    if (lay == 13) {                                     // split weights x fp16 k-octet activations: both operands by LDS-DMA
        if constexpr (!SB && WM * TM * 32 == 128) {
            // B-direct kernel (128 x 256 tile, activations straight into registers).  Dispatch rule, measured per shape
            // (tools/gemm_koct_bench.py, 24 x 7040 pixels) and fixed at compile time (experiments:
            // tools/build_variant.sh with -DSF_GEMM_BD_MIN_M / -DSF_GEMM_BD_MIN_WG / -DSF_GEMM_BDIRECT=0):
            //  * M >= 192 (+0..15 %), or one row tile (96 < M <= 128) with a short K <= 192 (M128 K192 33.1 -> 29.7 us; K = 960: no);
            //  * only when the 128 x 256 tiles fill the chip: a single clip gives M = 384 just 249 of them for 256 CUs and runs
            //    better on the 128 x 128 kernel (threshold 0: 186.7 / 189.2 ff/s, 384: 192.7 / 192.6); batched steps always pass.
#ifndef SF_GEMM_BDIRECT
#define SF_GEMM_BDIRECT 1
#endif
#ifndef SF_GEMM_BD_MIN_M
#define SF_GEMM_BD_MIN_M 192
#endif
#ifndef SF_GEMM_BD_MIN_WG
#define SF_GEMM_BD_MIN_WG 384
#endif
            const int64_t n_wg2 = (int64_t)sf::ceil_div(g.N, 256) * sf::ceil_div(g.M, 128) * g.batch;
            const bool bd_m = g.M >= SF_GEMM_BD_MIN_M || (g.M > 96 && g.K <= 192);
            if (SF_GEMM_BDIRECT && bd_m && n_wg2 >= SF_GEMM_BD_MIN_WG && g.k_splits <= 1 && (int64_t)g.ldb * 16 * 2 < ((int64_t)1 << 31)) {
                dim3 grid2(sf::ceil_div(g.N, 256) * sf::ceil_div(g.M, 128) * g.batch);
#if SF_GEMM_BD256_MIN_M
                if (PM == 1 && sf::ceil_div(g.M, 256) * 256 == sf::ceil_div(g.M, 128) * 128 && g.M >= SF_GEMM_BD256_MIN_M) {
                    dim3 grid3(sf::ceil_div(g.N, 256) * sf::ceil_div(g.M, 256) * g.batch);
                    hipLaunchKernelGGL(gemm_bdirect256_kernel, grid3, dim3(kBdThreads), 0, st, a);
                    return sf::check_launch("sf_gemm(B-direct 256)");
                }
#endif
                hipLaunchKernelGGL((gemm_bdirect_kernel<PM>), grid2, dim3(kBdThreads), 0, st, a);
                return sf::check_launch("sf_gemm(B-direct)");
            }
            hipLaunchKernelGGL((gemm_f16x3_mfma<WM, WN, TM, TN, 2, 5, PM>), grid, dim3(kThreads), 0, st, a);
            return sf::check_launch("sf_gemm(f16x2, k-octet B)");
        }
        return sf::fail(SF_ERR_UNSUPPORTED, "sf_gemm: SF_LAYOUT_F16_KOCT B needs the 128-row tile (M > 96) and F16X2 / F16");
    }
    if (lay == 14) {                                     // split weights x split k-octet activations (F16X3): both operands by LDS-DMA
        if constexpr (SB && WM * TM * 32 == 128 && TM * TN == 4) {
            hipLaunchKernelGGL((gemm_f16x3_mfma<WM, WN, TM, TN, 2, 6, 3>), grid, dim3(kThreads), 0, st, a);
            return sf::check_launch("sf_gemm(f16x3, split k-octet B)");
        }
        return sf::fail(SF_ERR_UNSUPPORTED, "sf_gemm: SF_LAYOUT_SPLIT_KOCT B needs the 128-row tile (M > 96) and F16X3");
    }
    if (lay == 12) {                                     // split weights x stored-fp16 K-major activations (F16X2 only)
        if constexpr (!SB) {
            hipLaunchKernelGGL((gemm_f16x3_mfma<WM, WN, TM, TN, 2, 4, PM>), grid, dim3(kThreads), 0, st, a);
            return sf::check_launch("sf_gemm(f16x2, fp16 B)");
        }
        return sf::fail(SF_ERR_UNSUPPORTED, "sf_gemm: SF_LAYOUT_F16_K_MAJOR B needs SF_PRECISION_F16X2");
    }
    switch (lay) {
        case 0: hipLaunchKernelGGL((gemm_f16x3_mfma<WM, WN, TM, TN, 0, 0, PM>), grid, dim3(kThreads), 0, st, a); break;
        case 5: hipLaunchKernelGGL((gemm_f16x3_mfma<WM, WN, TM, TN, 1, 1, PM>), grid, dim3(kThreads), 0, st, a); break;
        case 8: hipLaunchKernelGGL((gemm_f16x3_mfma<WM, WN, TM, TN, 2, 0, PM>), grid, dim3(kThreads), 0, st, a); break;
        default: return sf::fail(SF_ERR_UNSUPPORTED, "sf_gemm(f16x3): layout combination a=%d b=%d not built",
                                 g.a_layout, g.b_layout);
    }
    return sf::check_launch("sf_gemm(f16x3)");
}


template <int PM>
int pick_tile(const SplitArgs& a, hipStream_t st) {
    const SfGemm& g = a.g;
    // tile choice: the 128-row tile moves the fewest bytes per MAC; drop to 64/32 rows when padding M would
    // waste more than a quarter of the MFMAs
    const int M = g.M;
    auto padded = [&](int bm) { return (M + bm - 1) / bm * bm; };
#ifdef SF_GEMM_BM                                        // experiment builds only (tools/build_variant.sh ... -DSF_GEMM_BM=64)
    if (SF_GEMM_BM == 128) return launch_cfg<2, 2, 2, 2, PM>(a, st);
    if (SF_GEMM_BM == 64) return launch_cfg<1, 4, 2, 1, PM>(a, st);
    if (SF_GEMM_BM == 32) return launch_cfg<1, 4, 1, 1, PM>(a, st);
#endif
    // (a wave-specialised producer/consumer variant of the 128x128 kernel was faster for K >= 768 early in the round;
    // after the cheaper split and epilogue it measured 2-25 % slower at every batch size and was removed)
    if (padded(128) * 4 <= M * 5) return launch_cfg<2, 2, 2, 2, PM>(a, st);
    if (padded(64) * 4 <= M * 5 || M > 32) return launch_cfg<1, 4, 2, 1, PM>(a, st);
    return launch_cfg<1, 4, 1, 1, PM>(a, st);
}

int64_t span_bytes(int layout, int X, int K, int64_t ld, int group, int64_t group_stride) {
    if (layout == SF_LAYOUT_F16_K_MINOR) return ((int64_t)(X - 1) * ld + K) * 2;
    if (layout == SF_LAYOUT_F16_K_MAJOR) return ((int64_t)(K - 1) * ld + X) * 2;
    if (layout == SF_LAYOUT_SPLIT_KOCT) return (int64_t)((K + 7) / 8) * ld * 16 * 2;
    if (layout == SF_LAYOUT_F16_KOCT)
        return (group > 0) ? ((int64_t)((K - 1) / group) * group_stride * 2 + (int64_t)(((K - 1) % group) / 8 + 1) * ld * 16)
                           : (int64_t)((K + 7) / 8) * ld * 16;
    if (layout == SF_LAYOUT_K_MINOR) return ((int64_t)(X - 1) * ld + K) * 4;
    if (group > 0) return ((int64_t)((K - 1) / group) * group_stride + (int64_t)((K - 1) % group) * ld + X) * 4;
    return ((int64_t)(K - 1) * ld + X) * 4;
}


// ---- split-K combine with the full epilogue: C = epi(alpha * (sum_s partial_s + bias)) -----------------------------
// partial slabs: [split][batch][M][N] (N contiguous).  One thread = 4 consecutive columns of one row.
template <int EPI>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const SfGemm g, const float* part, int ks, int64_t slab) {
    constexpr bool kNeedsR = (EPI == SF_EPI_RES || EPI == SF_EPI_RES_GELU || EPI == SF_EPI_RES_GELU_DW1 ||
                              EPI == SF_EPI_AXPY);
    const int m = blockIdx.y, z = blockIdx.z;
    const int n = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (n >= g.N) return;
    const float* p = part + ((int64_t)z * g.M + m) * g.N + n;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < ks; ++s) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (n + i < g.N) a[i] += p[(int64_t)s * slab + i];
    }
    const float bias = g.bias ? g.bias[m] : 0.f;
    const float gam = (EPI == SF_EPI_AXPY) ? g.gamma[0] : 0.f;
    const float dww = (EPI == SF_EPI_RES_GELU_DW1) ? g.dw_w[m] : 0.f, dwb = (EPI == SF_EPI_RES_GELU_DW1) ? g.dw_b[m] : 0.f;
    const float* R = nullptr;
    if (kNeedsR) {
        const int64_t roff = (g.r_group > 0) ? (int64_t)(m / g.r_group) * g.r_group_stride + (int64_t)(m % g.r_group) * g.ldr
                                             : (int64_t)m * g.ldr;
        R = g.R + (int64_t)z * g.strideR + roff + n;
    }
    float* C = g.C + (int64_t)z * g.strideC + (int64_t)m * g.ldc + n;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (n + i < g.N) {
            const float v = g.alpha * (a[i] + bias);
            C[i] = sf::epi_apply<EPI>(v, kNeedsR ? R[i] : 0.f, dww, dwb, gam);
        }
    }
}

int launch_splitk_epilogue(const SfGemm& g, const float* part, int ks, int64_t slab, hipStream_t st) {
    dim3 grid(sf::ceil_div(sf::ceil_div(g.N, 4), 256), g.M, g.batch), block(256);
    switch (g.epilogue) {
        case SF_EPI_GELU: hipLaunchKernelGGL(splitk_epilogue_kernel<SF_EPI_GELU>, grid, block, 0, st, g, part, ks, slab); break;
        case SF_EPI_RELU: hipLaunchKernelGGL(splitk_epilogue_kernel<SF_EPI_RELU>, grid, block, 0, st, g, part, ks, slab); break;
        case SF_EPI_RES: hipLaunchKernelGGL(splitk_epilogue_kernel<SF_EPI_RES>, grid, block, 0, st, g, part, ks, slab); break;
        case SF_EPI_RES_GELU: hipLaunchKernelGGL(splitk_epilogue_kernel<SF_EPI_RES_GELU>, grid, block, 0, st, g, part, ks, slab); break;
        case SF_EPI_RES_GELU_DW1: hipLaunchKernelGGL(splitk_epilogue_kernel<SF_EPI_RES_GELU_DW1>, grid, block, 0, st, g, part, ks, slab); break;
        case SF_EPI_AXPY: hipLaunchKernelGGL(splitk_epilogue_kernel<SF_EPI_AXPY>, grid, block, 0, st, g, part, ks, slab); break;
        default: hipLaunchKernelGGL(splitk_epilogue_kernel<SF_EPI_NONE>, grid, block, 0, st, g, part, ks, slab); break;
    }
    return sf::check_launch("sf_gemm(split-K epilogue)");
}

// automatic split-K policy: grids that cannot fill the 256 CUs while each workgroup walks a long K chain
int auto_splits(int M, int N, int K, int batch) {
    const int bm = (M > 64) ? 128 : (M > 32 ? 64 : 32);
    const long wgs = (long)((M + bm - 1) / bm) * ((N + 127) / 128) * batch;
    const int nk = (K + BK - 1) / BK;
    if (wgs >= 96 || nk < 8) return 1;       // measured: for 165-workgroup grids the slab round trip eats the gain
    int ks = (int)((512 + wgs - 1) / wgs);
    if (ks > nk / 3) ks = nk / 3;
    const int cap = (wgs <= 16 && nk >= 64) ? 16 : 8;   // skinny outputs over a deep K (the encoder's sr convs: 14 workgroups, K = 8192)
    if (ks > cap) ks = cap;
    return ks < 2 ? 1 : ks;
}

// c_f16 / r_f16 rules of the vector and k-octet epilogues.  The kernel picks the vector epilogue only when
// epilogue_vec_ok() holds and would otherwise fall back to the scalar one, which knows neither format: reject here.
int check_output_formats(const SfGemm& g) {
    using sf::fail;
    const bool vec_c = (g.N & 3) == 0 && (g.ldc & 3) == 0 && (g.strideC & 3) == 0 && (reinterpret_cast<uintptr_t>(g.C) & 15) == 0;
    bool vec_r = true;
    if (g.R && g.r_f16 != 2)
        vec_r = (g.ldr & 3) == 0 && (g.strideR & 3) == 0 && (g.r_group_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(g.R) & 15) == 0;
    if (g.c_f16 == 1 && (!vec_c || !vec_r || g.k_splits > 1))
        return fail(SF_ERR_UNSUPPORTED, "sf_gemm: c_f16 = 1 needs the vector epilogue (N, ldc, strideC, ldr, strideR %% 4 == 0, "
                                        "16-byte aligned C / R), no split-K");
    if (g.c_f16 == 3) {
        const bool ok = vec_c && vec_r && g.k_splits <= 1 && (g.strideC16 & 7) == 0 &&
                        (reinterpret_cast<uintptr_t>(g.C16) & 15) == 0 && g.ldc >= g.N;
        if (!ok)
            return fail(SF_ERR_UNSUPPORTED, "sf_gemm: c_f16 = 3 (fp32 + k-octet output) needs the vector epilogue (N, ldc, strideC, "
                                            "ldr, strideR %% 4 == 0, 16-byte aligned C / R), 16-byte aligned C16, strideC16 %% 8 == 0, "
                                            "no split-K");
    }
    if ((g.c_f16 == 2 || g.c_f16 == 4) && (g.ldc < g.N || (g.strideC & 7) || (reinterpret_cast<uintptr_t>(g.C) & 15) || g.k_splits > 1 ||
                         (g.epilogue != SF_EPI_NONE && g.epilogue != SF_EPI_GELU && g.epilogue != SF_EPI_RES_GELU)))
        return fail(SF_ERR_UNSUPPORTED, "sf_gemm: c_f16 = 2 / 4 (k-octet output) needs ldc >= N, strideC %% 8 == 0, 16-byte aligned C, "
                                        "no split-K, epilogue NONE / GELU / RES_GELU");
    if (g.c_f16 == 4 && (g.precision != SF_PRECISION_F16X3 || g.r_f16))
        return fail(SF_ERR_UNSUPPORTED, "sf_gemm: c_f16 = 4 (split k-octet output) needs SF_PRECISION_F16X3 and an fp32 residual");
    if (g.r_f16 != 0 && g.r_f16 != 2) return fail(SF_ERR_BAD_ARG, "sf_gemm: r_f16 must be 0 or 2");
    if (g.r_f16 == 2) {
        const bool ok = g.R && g.epilogue == SF_EPI_RES_GELU_DW1 && g.c_f16 != 2 && vec_c && g.r_group == 0 && g.ldr >= g.N &&
                        (g.strideR & 7) == 0 && (reinterpret_cast<uintptr_t>(g.R) & 15) == 0 && g.k_splits <= 1 && g.alpha != 0.f;
        if (!ok)
            return fail(SF_ERR_UNSUPPORTED, "sf_gemm: r_f16 = 2 (k-octet residual) needs epilogue RES_GELU_DW1, the vector epilogue "
                                            "(N, ldc, strideC %% 4 == 0, 16-byte aligned C), c_f16 != 2, no grouping, ldr >= N, "
                                            "16-byte aligned R, strideR %% 8 == 0, no split-K");
    }
    return SF_OK;
}

}  // namespace

namespace sf {

// called from sf_gemm (gemm.hip) when precision == SF_PRECISION_F16X3 (never for conv3x3)
int64_t gemm_split_ws_floats(int M, int N, int K, int batch) {
    const int ks = auto_splits(M, N, K, batch);
    return ks > 1 ? (int64_t)ks * batch * M * N : 0;
}

int gemm_split_dispatch_inner(const SfGemm& g, hipStream_t st);
bool gemm_bstat_ok(const SfGemm& g);                       // gemm_bstat.hip
int gemm_bstat_launch(const SfGemm& g, hipStream_t st);

int gemm_split_dispatch(const SfGemm& g, hipStream_t st) {
    // automatic split-K through caller-provided scratch
    // (not for problems the activation-stationary kernel takes: it cuts small grids into row ranges instead, and a k-octet
    // operand with M <= 96 has no tiled kernel to split)
    const bool to_bstat = g.algo != SF_ALGO_TILED && gemm_bstat_ok(g) &&
                          (g.algo == SF_ALGO_BSTAT || g.c_f16 == 2 || (g.b_layout == SF_LAYOUT_F16_KOCT && (g.M + 127) / 128 * 128 * 4 > g.M * 5));
    if (g.k_splits == 0 && g.split_ws && !g.conv3x3 && !g.c_f16 && !g.r_f16 && !to_bstat) {  // (no split-K form for the 3x3 conv / fp16 output)
        const int ks = auto_splits(g.M, g.N, g.K, g.batch);
        const int64_t slab = (int64_t)g.batch * g.M * g.N;
        if (ks > 1 && g.split_ws_floats >= ks * slab) {
            SfGemm p = g;
            p.C = g.split_ws; p.ldc = g.N; p.strideC = (int64_t)g.M * g.N;
            p.bias = nullptr; p.R = nullptr; p.epilogue = SF_EPI_NONE; p.alpha = 1.0f;
            p.k_splits = ks; p.split_stride = slab;
            const int rc = gemm_split_dispatch_inner(p, st);
            if (rc != SF_OK) return rc;
            return launch_splitk_epilogue(g, g.split_ws, ks, slab, st);
        }
    }
    return gemm_split_dispatch_inner(g, st);
}

int gemm_split_dispatch_inner(const SfGemm& g, hipStream_t st) {
    if (g.k_splits > 1 && (g.epilogue != SF_EPI_NONE || g.bias || g.k_splits > 16))
        return fail(SF_ERR_BAD_ARG, "sf_gemm(f16x3): split-K needs SF_EPI_NONE, no bias, k_splits <= 16");
    if (g.conv3x3 && (g.b_layout != SF_LAYOUT_K_MAJOR || (g.K / 9) % 32 || g.b_group || g.k_splits > 1))
        return fail(SF_ERR_UNSUPPORTED, "sf_gemm(split): conv3x3 needs a plain K-major B with Cin %% 32 == 0");
    if (g.b_group % 32) return fail(SF_ERR_BAD_ARG, "sf_gemm(f16x3): b_group must be a multiple of 32");
    SplitArgs a;
    a.g = g;
#ifdef SF_GEMM_TIMERS
    a.ts = getenv("SF_GEMM_TS_BUF") ? (long long*)strtoull(getenv("SF_GEMM_TS_BUF"), nullptr, 0) : nullptr;
#endif
    if (g.a_layout == SF_LAYOUT_SPLIT_F16) {
        if (!g.A_hi || !g.A_lo || g.lda_h < (g.M + 127) / 128 * 128 || (g.lda_h & 127) ||
            ((reinterpret_cast<uintptr_t>(g.A_hi) | reinterpret_cast<uintptr_t>(g.A_lo)) & 15))
            return fail(SF_ERR_BAD_ARG, "sf_gemm(f16x3): SPLIT_F16 A needs 16-byte aligned A_hi/A_lo and lda_h = M padded to 128");
        a.a_bytes = (int64_t)((g.K + 31) / 32 * 32) * g.lda_h * 2;          // [K up to 32 / 8][lda_h rows][8] halfs
    } else {
        a.a_bytes = span_bytes(g.a_layout, g.M, g.K, g.lda, 0, 0);
    }
    a.b_bytes = span_bytes(g.b_layout, g.N, g.K, g.ldb, g.b_group, g.b_group_stride);
    if (a.a_bytes >= ((int64_t)1 << 31) || a.b_bytes >= ((int64_t)1 << 31))
        return fail(SF_ERR_UNSUPPORTED, "sf_gemm(f16x3): operand image larger than 2 GiB (32-bit buffer offsets)");
    // tile choice: the 128-row tile moves the fewest bytes per MAC; drop to 64/32 rows when padding M would
    // waste more than a quarter of the MFMAs
    // output / residual format rules: checked for EVERY operand layout (the fp16-in -> fp16-out hand-over is the common case)
    // activation-stationary kernel (gemm_bstat.hip) wherever it applies: K <= 640 held in registers, weights streamed
    // (SF_ALGO_AUTO: where the result leaves as k-octets only -- the FFN hiddens, x4, the temporal MLP hidden: its software-
    // pipelined GELU form runs 1.15-1.3x the tiled kernels; with fp32 planes to write, the tiled kernels' transposed 16-byte
    // stores still win -- and where the tiled family has no kernel at all: a k-octet operand with fewer than 97 rows)
    const bool bstat_auto = g.c_f16 == 2 || (g.b_layout == SF_LAYOUT_F16_KOCT && (g.M + 127) / 128 * 128 * 4 > g.M * 5);
    if ((g.algo == SF_ALGO_BSTAT || (g.algo == SF_ALGO_AUTO && bstat_auto)) && gemm_bstat_ok(g)) return gemm_bstat_launch(g, st);
    if (g.algo == SF_ALGO_BSTAT) return fail(SF_ERR_UNSUPPORTED, "sf_gemm: SF_ALGO_BSTAT cannot run this problem (see include/streamflow_hip.h)");
    if (const int rc = check_output_formats(g); rc != SF_OK) return rc;
    if (g.b_layout == SF_LAYOUT_F16_KOCT) {
        if (g.a_layout != SF_LAYOUT_SPLIT_F16 || (g.precision != SF_PRECISION_F16X2 && g.precision != SF_PRECISION_F16) ||
            (g.b_group & 31) || (g.b_group_stride & 7) || g.conv3x3 || (g.strideB & 7) ||
            (reinterpret_cast<uintptr_t>(g.B) & 15) || g.ldb < g.N)
            return fail(SF_ERR_UNSUPPORTED, "sf_gemm: SF_LAYOUT_F16_KOCT B needs F16X2 / F16, a SPLIT_F16 A, groups of a multiple of "
                                            "32 rows, 16-byte aligned B, strideB and group stride, ldb >= N");
        return (g.precision == SF_PRECISION_F16) ? pick_tile<1>(a, st) : pick_tile<2>(a, st);
    }
    if (g.b_layout == SF_LAYOUT_SPLIT_KOCT) {
        if (g.a_layout != SF_LAYOUT_SPLIT_F16 || g.precision != SF_PRECISION_F16X3 || g.b_group || g.conv3x3 || (g.strideB & 7) ||
            (reinterpret_cast<uintptr_t>(g.B) & 15) || g.ldb < g.N || (g.M + 127) / 128 * 128 * 4 > g.M * 5)
            return fail(SF_ERR_UNSUPPORTED, "sf_gemm: SF_LAYOUT_SPLIT_KOCT B needs F16X3, a SPLIT_F16 A on the 128-row tile (M > 96), no "
                                            "grouping, 16-byte aligned B, strideB %% 8 == 0, ldb >= N");
        return pick_tile<3>(a, st);
    }
    if (g.b_layout == SF_LAYOUT_F16_K_MAJOR) {
        if (g.a_layout != SF_LAYOUT_SPLIT_F16 || (g.precision != SF_PRECISION_F16X2 && g.precision != SF_PRECISION_F16) || (g.N & 1) || (g.ldb & 1) || (g.strideB & 1) ||
            g.b_group || g.conv3x3 || (reinterpret_cast<uintptr_t>(g.B) & 3))
            return fail(SF_ERR_UNSUPPORTED, "sf_gemm: SF_LAYOUT_F16_K_MAJOR B needs SF_PRECISION_F16X2, a SPLIT_F16 A, even N / ldb / "
                                            "strideB, no grouping, 4-byte aligned B");
        return (g.precision == SF_PRECISION_F16) ? pick_tile<1>(a, st) : pick_tile<2>(a, st);
    }
    if (g.b_layout == SF_LAYOUT_F16_K_MINOR) {
        if (g.a_layout != SF_LAYOUT_K_MINOR || (g.ldb & 1) || g.b_group || g.conv3x3 || (reinterpret_cast<uintptr_t>(g.B) & 3))
            return fail(SF_ERR_UNSUPPORTED, "sf_gemm(f16x3): SF_LAYOUT_F16_K_MINOR B needs a K-minor A, even ldb, 4-byte aligned B");
        return pick_tile<2>(a, st);
    }
    // SF_PRECISION_F16 (one product) is built for pre-packed weights; an fp32 A operand (the attention / correlation
    // contractions called through sf_gemm) is split on the fly as in F16X2
    if (g.precision == SF_PRECISION_F16 && g.a_layout == SF_LAYOUT_SPLIT_F16) return pick_tile<1>(a, st);
    return (g.precision != SF_PRECISION_F16X3) ? pick_tile<2>(a, st) : pick_tile<3>(a, st);
}

}  // namespace sf
