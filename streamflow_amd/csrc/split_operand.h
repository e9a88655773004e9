// Operand staging for the split-precision (f16x3) matrix-core kernels: buffer-load descriptors, the
// fp32 -> (hi, lo) f16 split, and the LDS image layout.  Shared by gemm_split.hip and corr.hip.
#pragma once
#include "sf_common.h"

namespace sf_split {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 256;
constexpr int BK = 32;                 // k-tile depth
constexpr int LDK = BK + 8;            // LDS row stride in halfs (80 bytes)

__device__ __forceinline__ float as_f(unsigned u) { return __builtin_bit_cast(float, u); }

struct Split8 {
    f16x8 hi, lo;
};

// x = hi + lo with 3 VALU instructions per element instead of 5+: hi is x with its mantissa truncated to the 10
// explicit bits of fp16 (one v_and, exactly representable in fp16 for normal-range values), lo = x - hi is exact in
// fp32, and both are narrowed two at a time with v_cvt_pkrtz_f16_f32.  |x - (hi + lo)| <= 2^-20 |x| (round-to-zero
// twice; the round-to-nearest variant reaches 2^-22 but costs a convert, a convert back, a subtract, a convert and a
// pack per element -- the GEMM is bound by instruction issue, not by MFMA, so the cheaper split is the faster one).
typedef __fp16 pk_half2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ Split8 split8(const float (&x)[8]) {
    union { f16x8 v; pk_half2 h[4]; } hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        const float ah = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a) & 0xFFFFE000u);
        const float bh = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, b) & 0xFFFFE000u);
        hi.h[i] = __builtin_amdgcn_cvt_pkrtz(ah, bh);
        lo.h[i] = __builtin_amdgcn_cvt_pkrtz(a - ah, b - bh);
    }
    Split8 s;
    s.hi = hi.v;
    s.lo = lo.v;
    return s;
}

// The same split with hi = round-to-nearest(x) (lo = x - hi, exact, either sign): for kernels that use hi ALONE in some
// products (conv.hip's two-product mode) -- a truncated hi would bias every such product towards zero by up to 2^-10.
__device__ __forceinline__ Split8 split8_rn(const float (&x)[8]) {
    Split8 s;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const _Float16 h = (_Float16)x[i];
        s.hi[i] = h;
        s.lo[i] = (_Float16)(x[i] - (float)h);
    }
    return s;
}

// One operand (A or B) of the GEMM as seen by one thread.
template <int BX, int LAY>
struct Operand {
    static constexpr int NI = (BX * (BK / 8) + kThreads - 1) / kThreads;   // (row, k-octet) items per thread
    static constexpr bool kUniformKo = (LAY == SF_LAYOUT_K_MAJOR) && (BX % 64 == 0);
    __amdgpu_buffer_rsrc_t rsrc, rsrc_lo;
    int voff[NI];        // per-thread byte offset (constant over the k-loop)
    int ko[NI];          // k-octet of the item
    int lds_off[NI];     // destination offset in halfs
    bool live[NI];
    int K, ld, group;
    int64_t group_stride;
    int rowc[NI][8];     // K-major: byte offset of row (ko*8 + i) inside a k-tile -- loop invariant, wave uniform
    unsigned tapmask[NI];   // implicit 3x3 conv: bit t set = tap t of this thread's pixel lies inside the image
    struct Regs {                                  // one staged k-tile of this thread
        float v[(LAY >= 2) ? 1 : NI][8];
        u32x4 ph[(LAY >= 2) ? NI : 1], pl[(LAY == 2) ? NI : 1];
    };

    __device__ __forceinline__ void init(const void* base, const void* base_lo, int64_t bytes, int ld_, int K_, int X,
                                         int x0, int group_, int64_t group_stride_, int tid) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
        rsrc_lo = rsrc;
        if (LAY == 2) rsrc_lo = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base_lo), 0, (int)bytes, 0x00020000);
        K = K_; ld = ld_; group = group_; group_stride = group_stride_;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int idx = tid + j * kThreads;
            int xl, kq;
            if (LAY == SF_LAYOUT_K_MAJOR || LAY == 2) { xl = idx % BX; kq = idx / BX; }   // lanes walk x: coalesced rows
            else { kq = idx % (BK / 8); xl = idx / (BK / 8); }                   // lanes walk k
            live[j] = (BX * (BK / 8)) % kThreads == 0 || idx < BX * (BK / 8);
            ko[j] = kUniformKo ? __builtin_amdgcn_readfirstlane(kq) : kq;
            lds_off[j] = xl * LDK + kq * 8;
            // rows/columns past the operand are clamped: they are loaded (harmlessly) but never stored
            const int xc = (x0 + xl < X) ? x0 + xl : X - 1;
            tapmask[j] = 0x1ffu;
            if (LAY == SF_LAYOUT_K_MAJOR) {
                voff[j] = xc * 4;
#pragma unroll
                for (int i = 0; i < 8; ++i) rowc[j][i] = (ko[j] * 8 + i) * ld * 4;
            }
            else if (LAY == SF_LAYOUT_K_MINOR) voff[j] = (xc * ld + kq * 8) * 4;
            else if (LAY == SF_LAYOUT_F16_K_MINOR) voff[j] = (xc * ld + kq * 8) * 2;
            else voff[j] = (kq * ld + x0 + xl) * 16;            // k-octet planes [K/8][ld rows][8], host-padded: always in range
        }
    }

    // implicit 3x3 convolution (pad 1) over an h x w image: K = 9 * cin, k = tap * cin + c.  A k-tile never straddles
    // taps (cin % 32 == 0, checked on the host), so a tile is a plain K-major tile of the input shifted by the tap's
    // (dy, dx); pixels whose tap falls outside the image are zeroed in store() through tapmask.
    __device__ __forceinline__ void set_conv3x3(int x0, int h, int w, int tid) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int idx = tid + j * kThreads;
            const int x = x0 + idx % BX;                       // K-major item mapping: lanes walk x
            const int yy = x / w, xx = x % w;
            unsigned m = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int y2 = yy + t / 3 - 1, x2 = xx + t % 3 - 1;
                if (y2 >= 0 && y2 < h && x2 >= 0 && x2 < w) m |= 1u << t;
            }
            tapmask[j] = m;
            voff[j] = x * 4;                                    // no clamping: out-of-range taps are masked / range-checked
        }
    }

    // Issue the raw loads of k-tile k0 (nothing is consumed here, so no wait is needed before the MFMAs).
    // K-major rows: tile_off = element offset of row k0 (tracked incrementally by the caller for grouped
    // operands); rows past K are clamped to the last valid row and zeroed in store().
    __device__ __forceinline__ void load(int k0, int tile_off, Regs& rg, int shift_bytes = 0) const {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            if (LAY == SF_LAYOUT_F16_K_MINOR) {
                // already fp16: k0 rides in the VGPR offset so that the range check sees it (reads past the end give 0;
                // the tail k >= K inside a row reads the next row's finite values against zeroed A columns)
                rg.ph[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[j] + k0 * 2, 0, 0);
            } else if (LAY == 2) {
                rg.ph[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[j], (k0 / 8) * ld * 16, 0);
                rg.pl[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_lo, voff[j], (k0 / 8) * ld * 16, 0);
            } else if (LAY == SF_LAYOUT_K_MAJOR) {
                if (k0 + BK <= K) {
                    // interior k-tile (workgroup-uniform): one v_add per item, the row offsets are SGPR constants
                    const int vo = voff[j] + tile_off * 4 + shift_bytes;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (kUniformKo) rg.v[j][i] = as_f(__builtin_amdgcn_raw_buffer_load_b32(rsrc, vo, rowc[j][i], 0));
                        else rg.v[j][i] = as_f(__builtin_amdgcn_raw_buffer_load_b32(rsrc, vo + rowc[j][i], 0, 0));
                    }
                } else {
                    const int kb = ko[j] * 8;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int over = k0 + kb + i - (K - 1);                 // > 0: row past the end
                        const int r = (tile_off + (kb + i - (over > 0 ? over : 0)) * ld) * 4;
                        if (kUniformKo) rg.v[j][i] = as_f(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff[j], r, 0));
                        else rg.v[j][i] = as_f(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff[j] + r, 0, 0));
                    }
                }
            } else {
                // k0 goes into the VGPR offset: soffset is excluded from the hardware range check, and the last
                // k-octet of the last row may reach past the end of the buffer (then it reads as zero)
                const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[j] + k0 * 4, 0, 0);
                const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[j] + k0 * 4 + 16, 0, 0);
                rg.v[j][0] = as_f(a[0]); rg.v[j][1] = as_f(a[1]); rg.v[j][2] = as_f(a[2]); rg.v[j][3] = as_f(a[3]);
                rg.v[j][4] = as_f(b[0]); rg.v[j][5] = as_f(b[1]); rg.v[j][6] = as_f(b[2]); rg.v[j][7] = as_f(b[3]);
            }
        }
    }

    // Consume the staged tile (loaded for k-tile k0): zero rows k >= K, split into hi/lo f16, write to LDS.
    // kLo = false: keep only the fp16 rounding of the values (SF_PRECISION_F16X2's B operand)
    template <bool kLo = true>
    __device__ __forceinline__ void store(int k0, _Float16* lds_hi, _Float16* lds_lo, Regs& rg, int tap = -1) const {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            if (!live[j]) continue;
            if (LAY == SF_LAYOUT_F16_K_MINOR) {
                *reinterpret_cast<u32x4*>(lds_hi + lds_off[j]) = rg.ph[j];
            } else if (LAY == 2) {
                *reinterpret_cast<u32x4*>(lds_hi + lds_off[j]) = rg.ph[j];
                if (kLo) *reinterpret_cast<u32x4*>(lds_lo + lds_off[j]) = rg.pl[j];
            } else {
                if (k0 + BK > K) {          // last, partial k-tile (workgroup-uniform)
                    const int k = k0 + ko[j] * 8;
#pragma unroll
                    for (int i = 0; i < 8; ++i) rg.v[j][i] = (k + i < K) ? rg.v[j][i] : 0.f;
                }
                if (tap >= 0) {             // implicit 3x3 conv: zero padding
                    const bool in = (tapmask[j] >> tap) & 1u;
#pragma unroll
                    for (int i = 0; i < 8; ++i) rg.v[j][i] = in ? rg.v[j][i] : 0.f;
                }
                if (kLo) {
                    const Split8 s8 = split8(rg.v[j]);
                    *reinterpret_cast<f16x8*>(lds_hi + lds_off[j]) = s8.hi;
                    *reinterpret_cast<f16x8*>(lds_lo + lds_off[j]) = s8.lo;
                } else {
                    f16x8 h;
#pragma unroll
                    for (int i = 0; i < 8; ++i) h[i] = (_Float16)rg.v[j][i];
                    *reinterpret_cast<f16x8*>(lds_hi + lds_off[j]) = h;
                }
            }
        }
    }
};

// B operand stored as IEEE fp16 rows, K-major: B[k * ld + n] halves (SF_LAYOUT_F16_K_MAJOR) -- what a producing GEMM
// writes with SfGemm.c_f16 when its only consumer is another contraction of the f16x2 mode (the FFN hidden
// activations): the consumer would round the fp32 value to fp16 anyway, so storing the rounded value is bit-identical,
// halves both HBM passes and removes the conversion from the loop.  One item = (pixel PAIR, k-octet): 8 dword loads
// (two adjacent pixels of one row each), 8 v_perm to un-interleave them into two k-octets, two ds_write_b128.
template <int BX>
struct OperandF16KMajor {
    static_assert(BX % 128 == 0, "pairs per k-octet must fill whole waves");
    static constexpr int NI = (BX / 2 * (BK / 8) + kThreads - 1) / kThreads;
    __amdgpu_buffer_rsrc_t rsrc;
    int voff[NI], ko[NI], lds_off[NI];
    int rowc[NI][8];
    int K, ld;
    struct Regs { unsigned w[NI][8]; };

    __device__ __forceinline__ void init(const void* base, const void*, int64_t bytes, int ld_, int K_, int X, int x0, int,
                                         int64_t, int tid) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
        K = K_; ld = ld_;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int idx = tid + j * kThreads;
            const int pi = idx % (BX / 2);
            ko[j] = __builtin_amdgcn_readfirstlane(idx / (BX / 2));        // 64 pairs = one wave per k-octet
            lds_off[j] = (2 * pi) * LDK + ko[j] * 8;
            // pairs past the operand are clamped (X is even, host-checked): loaded, never stored by the epilogue
            const int xc = (x0 + 2 * pi < X) ? x0 + 2 * pi : X - 2;
            voff[j] = xc * 2;
#pragma unroll
            for (int i = 0; i < 8; ++i) rowc[j][i] = (ko[j] * 8 + i) * ld * 2;
        }
    }
    __device__ __forceinline__ void set_conv3x3(int, int, int, int) {}

    __device__ __forceinline__ void load(int k0, int tile_off, Regs& rg, int = 0) const {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            if (k0 + BK <= K) {
                const int vo = voff[j] + tile_off * 2;
#pragma unroll
                for (int i = 0; i < 8; ++i) rg.w[j][i] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, vo, rowc[j][i], 0);
            } else {                                        // last, partial k-tile: rows past K are clamped (zeroed in store)
                const int kb = ko[j] * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int over = k0 + kb + i - (K - 1);
                    const int r = (tile_off + (kb + i - (over > 0 ? over : 0)) * ld) * 2;
                    rg.w[j][i] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff[j], r, 0);
                }
            }
        }
    }

    template <bool kLo = false>
    __device__ __forceinline__ void store(int k0, _Float16* lds_hi, _Float16*, Regs& rg, int = -1) const {
        static_assert(!kLo, "a stored-fp16 operand has no lo part");
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            if (k0 + BK > K) {
                const int k = k0 + ko[j] * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i) rg.w[j][i] = (k + i < K) ? rg.w[j][i] : 0u;
            }
            u32x4 p0, p1;                                   // pixel 2 pi (low halves) and 2 pi + 1 (high halves)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                p0[i] = __builtin_amdgcn_perm(rg.w[j][2 * i + 1], rg.w[j][2 * i], 0x05040100u);
                p1[i] = __builtin_amdgcn_perm(rg.w[j][2 * i + 1], rg.w[j][2 * i], 0x07060302u);
            }
            *reinterpret_cast<u32x4*>(lds_hi + lds_off[j]) = p0;
            *reinterpret_cast<u32x4*>(lds_hi + lds_off[j] + LDK) = p1;
        }
    }
};

// SF_LAYOUT_F16_KOCT: the producer already wrote the consumer's LDS image (k-octet planes of fp16), tiles go by
// buffer_load ... lds inside the kernel: nothing is staged through registers, this stand-in only keeps the code shape.
struct OperandDma {
    struct Regs {};
    __device__ __forceinline__ void init(const void*, const void*, int64_t, int, int, int, int, int, int64_t, int) {}
    __device__ __forceinline__ void set_conv3x3(int, int, int, int) {}
    __device__ __forceinline__ void load(int, int, Regs&, int = 0) const {}
    template <bool kLo = false>
    __device__ __forceinline__ void store(int, _Float16*, _Float16*, Regs&, int = -1) const {}
};

template <int BX, int LAY>
struct OperandSel { typedef Operand<BX, LAY> type; };
template <int BX>
struct OperandSel<BX, SF_LAYOUT_F16_KOCT> { typedef OperandDma type; };
template <int BX>
struct OperandSel<BX, SF_LAYOUT_SPLIT_KOCT> { typedef OperandDma type; };
template <int BX>
struct OperandSel<BX, SF_LAYOUT_F16_K_MAJOR> { typedef OperandF16KMajor<BX> type; };

// element offset of K-major row k0 (start of a k-tile) for a possibly grouped operand; tiles never straddle
// groups (group % 32 == 0 is checked on the host)
struct RowCursor {
    int group, ld, within, off;
    int64_t group_stride;
    __device__ __forceinline__ void init(int group_, int ld_, int64_t gs) { group = group_; ld = ld_; group_stride = gs; within = 0; off = 0; }
    __device__ __forceinline__ void advance() {
        if (group > 0) {
            within += BK;
            if (within >= group) { within = 0; off += (int)group_stride - (group - BK) * ld; }
            else off += BK * ld;
        } else off += BK * ld;
    }
};

}  // namespace sf_split
