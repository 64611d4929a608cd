// kernels.hpp - device kernels of the PAFUSE hot path for gfx950 (MI355X, CDNA4).  Wave = 64 lanes.
//
// Everything in memory, LayerNorm, softmax, attention and the DDIM arithmetic are fp32 (the parity contract is 1e-4 mm
// MPJPE, SURVEY.md section 7 "Hard parts" 1), with the reference's fp64 islands kept in fp64.  The linear layers'
// products have three modes (GemmParams::bf16): the inference default splits every fp32 operand into three bf16 slices and
// keeps the six products above 2^-24 relative on the bf16 matrix cores with fp32 accumulation (v_mfma_f32_32x32x16_bf16;
// the qkv layers v_mfma_f32_16x16x32_bf16) - fp32-equivalent results; training and the 'f32' mode run the f32-input
// matrix cores (v_mfma_f32_32x32x2_f32), attention v_mfma_f32_16x16x4_f32 in every mode.
// The library is compiled WITHOUT packed-fp32 VALU instructions (pafuse_amd/build_flags.py): beside waves of another
// queue that issue v_mfma_f32_32x32x16_bf16 a v_pk_*_f32 with a high-register src1 select returns wrong lanes on MI355X.
//
// Token matrix layout: one row per (r, f, j) with r = flip*B*P + b*P + p, row-major [M, C].  The reference's
// "(b f) n c <-> (b n) f c" rearranges (common/mixste.py:244,270,274,288) never materialise here: linear layers
// and norms are per-row, and attention addresses the rows of a sequence through strides.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

namespace pafuse {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;   // K-chunk of the linear-layer kernels
constexpr int LDK = 36;  // padded LDS row (floats): 144 B stride makes the ds_read_b128 fragment reads conflict-free

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 to_bf16x8(const f32x4 lo, const f32x4 hi) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (__bf16)lo[i], v[4 + i] = (__bf16)hi[i];
    return v;
}

// One 16-deep bf16 MFMA step on a 32x32 accumulator.  -DPAFUSE_MFMA_K8 (diagnostic build) issues it as two 8-deep
// v_mfma_f32_32x32x8_bf16_1k on the lower / upper four elements of both fragments (same k pairing for A and B).
#ifdef PAFUSE_MFMA_K8
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mfma_bf16_k16(const bf16x8 a, const bf16x8 b, f32x16 c) {
    const bf16x4_t a0 = {a[0], a[1], a[2], a[3]}, a1 = {a[4], a[5], a[6], a[7]};
    const bf16x4_t b0 = {b[0], b[1], b[2], b[3]}, b1 = {b[4], b[5], b[6], b[7]};
    c = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4_t, a0), __builtin_bit_cast(s16x4_t, b0), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4_t, a1), __builtin_bit_cast(s16x4_t, b1), c, 0, 0, 0);
}
#else
__device__ __forceinline__ f32x16 mfma_bf16_k16(const bf16x8 a, const bf16x8 b, f32x16 c) {
#ifdef PAFUSE_MFMA_NOP
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop %0" ::"n"(PAFUSE_MFMA_NOP));
    __builtin_amdgcn_sched_barrier(0);
#endif
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
#endif

// ---- split precision ("bf16x3"): an fp32 value as the sum of three bf16 numbers, x = s0 + s1 + s2 (exact up to
// 2^-24 |x|: each slice takes the next 8 significand bits of what is left, round-to-nearest).  A product of two split
// operands keeps the six terms of order <= 2^-16 (s0*t0; s0*t1, s1*t0; s1*t1, s0*t2, s2*t0) on the bf16 matrix cores
// with fp32 accumulation: 6 v_mfma_f32_32x32x16_bf16 (192 cycles per 16-deep step) replace 8 v_mfma_f32_32x32x2_f32
// (512 cycles); the dropped terms are below 2^-24 |x t|, i.e. below the rounding of a single fp32 product.
struct bf16x8x3 {
    bf16x8 s0, s1, s2;
};
__device__ __forceinline__ bf16x8x3 split3(const f32x4 lo, const f32x4 hi) {
    bf16x8x3 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float x = i < 4 ? lo[i] : hi[i - 4];
        const __bf16 b0 = (__bf16)x;
        const float r1 = x - (float)b0;  // exact
        const __bf16 b1 = (__bf16)r1;
        const float r2 = r1 - (float)b1;  // exact
        o.s0[i] = b0, o.s1[i] = b1, o.s2[i] = (__bf16)r2;
    }
    return o;
}

// The same split in stages, for hand-placed software pipelines: two fp32 values -> three packed bf16 pairs (low half = the
// first value).  Stage k consumes stage k-1; the stages are spread over the gaps of an MFMA chain by the caller.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
struct SplitPair {
    float x0, x1;                    // what is left of the two values
    uint32_t s0 = 0, s1 = 0, s2 = 0;  // the three slices, packed {bf16(x0), bf16(x1)}
    __device__ __forceinline__ static uint32_t pack(float a, float b) {
        const bf16x2 v = {(__bf16)a, (__bf16)b};  // one v_cvt_pk_bf16_f32 (RNE)
        return __builtin_bit_cast(uint32_t, v);
    }
    __device__ __forceinline__ void sub(uint32_t sl) {
        x0 -= __builtin_bit_cast(float, sl << 16);          // exact
        x1 -= __builtin_bit_cast(float, sl & 0xffff0000u);  // exact
    }
    template <int STAGE>
    __device__ __forceinline__ void stage() {
        if constexpr (STAGE == 0) s0 = pack(x0, x1);
        if constexpr (STAGE == 1) sub(s0);
        if constexpr (STAGE == 2) s1 = pack(x0, x1);
        if constexpr (STAGE == 3) sub(s1);
        if constexpr (STAGE == 4) s2 = pack(x0, x1);
    }
};
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ---- split precision, second scheme ("f16x2", round 4; the kernels are in hgemm.hpp): fp16 has 11 significand bits, so two
// slices carry 22-23 bits of an fp32 value and THREE matrix products replace bf16x3's six (ceiling 2500 / 3 = 833 TFLOP/s of
// fp32-equivalent work).
//   activation  a = hi + 2^-11 lo,   hi = f16(a),  lo = f16((a - hi) 2^11)      (the subtraction and the scaling are exact;
//               the scaled lo stays a normal fp16 number for every |a| down to 2^-25, no per-tensor scale is needed;
//               |a| must stay below 65504 - beyond it hi is inf and the output row shows it)
//   weight      Ws = 2^k W with the tensor's largest |Ws| in [2^14, 2^15) (k chosen when the image is made), as fp16 slices
//               w0 = f16(Ws),  w1 = f16(Ws - w0)  (unscaled: at this magnitude it is normal, or a subnormal whose rounding
//               is 2^-39 of the tensor's largest weight),  and  w2 = f16(w0 2^-11), formed in registers
//   product     a Ws ~ hi w0 + hi w1 + lo w2:  three MFMAs into ONE accumulator, nothing rescaled inside the K loop; the
//               epilogue multiplies the accumulator by 2^-k (exact).  Dropped: lo (Ws - w0) 2^-11 ~ 2^-22 |a Ws|; the slices
//               represent a to 2^-23 |a| and Ws likewise.  Each kept product is exact in the fp32 accumulator (22 bits).
// Against exact arithmetic the scheme is as close as bf16x3 and closer than the fp32 FMA chain (one rounding per MFMA
// instead of one per k; tools/f16x2_emulation.py, tests/test_hip_parity.py::test_linear_split).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
struct f16x8x2 {
    f16x8 hi, lo;
};
__device__ __forceinline__ f16x8x2 split2h(const f32x4 lo4, const f32x4 hi4) {
    f16x8x2 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float x = i < 4 ? lo4[i] : hi4[i - 4];
        // ONE value of x for both slices: under register pressure the compiler re-evaluates the expression x came from (a GELU, a
        // LayerNorm) for its second use, contracted differently - two results one fp32 ulp apart round to DIFFERENT fp16 values
        // when they straddle a tie, and the low slice then belongs to another high slice than the one stored: an error of one
        // fp16 ulp in a few elements per million (found in the register-resident form of the fused MLP, round 4)
        asm volatile("" : "+v"(x));
        const _Float16 h = (_Float16)x;                  // RNE
        o.hi[i] = h;
        o.lo[i] = (_Float16)((x - (float)h) * 2048.0f);  // exact difference, exact scaling, RNE
    }
    return o;
}
// "H image" of a matrix [rows][K]: rows of 4 K bytes, per sub-block of 8 k the 16 bytes of the first slice, then the 16 of the
// second - [rows][K/8][hi 8 x f16 | lo 8 x f16] (weights: [w0 | w1]).  Producers that own four consecutive values of a row
// store their halves of a sub-block: 8 bytes of hi, 8 bytes of lo.
__device__ __forceinline__ void hsplit_store4(uint8_t* sub_block_base, int first /* 0 or 4: position inside the sub-block */,
                                              f32x4 v) {
    asm volatile("" : "+v"(v));   // one value for both slices (see split2h)
    f16x2_t h0 = {(_Float16)v[0], (_Float16)v[1]}, h1 = {(_Float16)v[2], (_Float16)v[3]};   // v_cvt_pk_f16_f32 (RNE)
    f16x2_t l0 = {(_Float16)((v[0] - (float)h0[0]) * 2048.0f), (_Float16)((v[1] - (float)h0[1]) * 2048.0f)};
    f16x2_t l1 = {(_Float16)((v[2] - (float)h1[0]) * 2048.0f), (_Float16)((v[3] - (float)h1[1]) * 2048.0f)};
    *reinterpret_cast<u32x2*>(sub_block_base + 2 * first) = u32x2{__builtin_bit_cast(uint32_t, h0), __builtin_bit_cast(uint32_t, h1)};
    *reinterpret_cast<u32x2*>(sub_block_base + 16 + 2 * first) = u32x2{__builtin_bit_cast(uint32_t, l0), __builtin_bit_cast(uint32_t, l1)};
}
// "X image" of a matrix [rows][K] (K % 32 == 0; the bf16x3 scheme on the LDS-DMA pipeline, xgemm.hpp): rows of 6 K bytes, per
// chunk of 32 k the 32 bf16 of slice 0, then of slice 1, then of slice 2 - [rows][K/32][3][32 x bf16].  The slices of an fp32
// number are exact (x = s0 + s1 + s2, split3), so an X image holds the fp32 tensor itself, bit for bit, in 6 bytes per element.
__host__ __device__ inline size_t ximage_bytes(int64_t R, int64_t K) { return (size_t)R * (size_t)K * 6; }
// byte offset of element k's slice-0 bf16 inside its row (slices 1, 2 at + 64, + 128)
__device__ __forceinline__ size_t xoff(int k) { return (size_t)(k >> 5) * 192 + (size_t)(((k >> 3) & 3) * 16 + (k & 7) * 2); }
// the three packed bf16 pairs of two fp32 values (low half = the first value): every step exact but the roundings to bf16
__device__ __forceinline__ void xsplit_pair(float x0, float x1, uint32_t& s0, uint32_t& s1, uint32_t& s2) {
    asm volatile("" : "+v"(x0), "+v"(x1));   // ONE value for all slices (see split2h: a re-evaluated input would mix two roundings)
    SplitPair sp{x0, x1};
    sp.stage<0>(), sp.stage<1>(), sp.stage<2>(), sp.stage<3>(), sp.stage<4>();
    s0 = sp.s0, s1 = sp.s1, s2 = sp.s2;
}
// four consecutive values of a row at element k (k % 4 == 0): 8 bytes in each slice plane
__device__ __forceinline__ void xsplit_store4(uint8_t* row_base, int k, const f32x4 v) {
    uint32_t a0, a1, a2, b0, b1, b2;
    xsplit_pair(v[0], v[1], a0, a1, a2);
    xsplit_pair(v[2], v[3], b0, b1, b2);
    uint8_t* const d = row_base + xoff(k);
    *reinterpret_cast<u32x2*>(d) = u32x2{a0, b0};
    *reinterpret_cast<u32x2*>(d + 64) = u32x2{a1, b1};
    *reinterpret_cast<u32x2*>(d + 128) = u32x2{a2, b2};
}
// a whole sub-block of eight values at element k (k % 8 == 0): 16 bytes in each slice plane
__device__ __forceinline__ void xsplit_store8(uint8_t* row_base, int k, const f32x4 lo, const f32x4 hi) {
    uint32_t s0[4], s1[4], s2[4];
    xsplit_pair(lo[0], lo[1], s0[0], s1[0], s2[0]);
    xsplit_pair(lo[2], lo[3], s0[1], s1[1], s2[1]);
    xsplit_pair(hi[0], hi[1], s0[2], s1[2], s2[2]);
    xsplit_pair(hi[2], hi[3], s0[3], s1[3], s2[3]);
    uint8_t* const d = row_base + xoff(k);
    *reinterpret_cast<u32x4*>(d) = u32x4{s0[0], s0[1], s0[2], s0[3]};
    *reinterpret_cast<u32x4*>(d + 64) = u32x4{s1[0], s1[1], s1[2], s1[3]};
    *reinterpret_cast<u32x4*>(d + 128) = u32x4{s2[0], s2[1], s2[2], s2[3]};
}
// the fp32 values of four consecutive elements from their three packed slices (two bf16 pairs per slice): s0 + (s1 + s2),
// both sums exact (s1 + s2 has at most 16 significant bits, the total is the number that was split)
__device__ __forceinline__ f32x4 xjoin4(const u32x2 s0, const u32x2 s1, const u32x2 s2) {
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float l0 = __builtin_bit_cast(float, s0[i] << 16), h0 = __builtin_bit_cast(float, s0[i] & 0xffff0000u);
        const float l1 = __builtin_bit_cast(float, s1[i] << 16), h1 = __builtin_bit_cast(float, s1[i] & 0xffff0000u);
        const float l2 = __builtin_bit_cast(float, s2[i] << 16), h2 = __builtin_bit_cast(float, s2[i] & 0xffff0000u);
        v[2 * i] = l0 + (l1 + l2), v[2 * i + 1] = h0 + (h1 + h2);
    }
    return v;
}
__global__ void __launch_bounds__(256) xsplit_weights_kernel(const float* W, uint8_t* out, int N, int K) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one (n, sub-block of 8 k) per thread
    const int spr = K / 8;
    if (idx >= (int64_t)N * spr) return;
    const int64_t n = idx / spr;
    const int k = (int)(idx % spr) * 8;
    const float* src = W + n * K + k;
    xsplit_store8(out + (size_t)n * K * 6, k, *reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4));
}
constexpr int HSPLIT_TAIL_BYTES = 256;   // behind a WEIGHT H image: [0] float 2^-k (what the epilogue multiplies by), [1] scratch
__host__ __device__ inline size_t hsplit_bytes(int64_t R, int64_t K) { return (size_t)R * (size_t)K * 4 + HSPLIT_TAIL_BYTES; }
__device__ __forceinline__ f32x16 mfma_f16_k16(const f16x8 a, const f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// the largest |W| of a tensor as fp32 bits (monotonic for non-negative values) into *out (zeroed by the caller)
__global__ void __launch_bounds__(256) absmax_kernel(const float* W, int64_t n, uint32_t* out) {
    uint32_t m = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        m = max(m, __builtin_bit_cast(uint32_t, W[i]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// Pre-split weight image W' of a [N,K] fp32 matrix, in chunks of BKC = 32 or 16 along K:
// [K/BKC chunks][N rows][6*BKC bytes].  A row of a chunk holds BKC/8 sub-blocks of 8 consecutive k (sub-block sb covers
// k = BKC*chunk + 8*sb + 0..7; sb = 2*s2 + h is exactly what MFMA lane half h feeds to 16-deep step s2) x three slices,
// 16 bytes each, slices at +0, +16, +32 of the sub-block.  Sub-blocks are rotated inside the row so that the
// ds_read_b128 fragment reads of 16 consecutive rows hit 16 different bank quads with UNPADDED rows:
//   BKC = 32 (192-byte rows): sub-block sb at position (sb + (n >> 2)) & 3;   BKC = 16 (96-byte rows): (sb + (n >> 3)) & 1.
// The image is read as it lies: a tile's chunk is one contiguous run of BN * 6 * BKC bytes in HBM and in LDS.
// BKC = 32 serves the plain linear layers (qkv, fc1), BKC = 16 the whole-row ones (proj, fc2), whose W' stage
// (N = C rows) would not fit a multi-stage LDS ring at 32.
// Third layout (M16 = 1, BKC = 32): the qkv layer runs on v_mfma_f32_16x16x32_bf16 (gemm16_tile), whose fragment read takes
// 16 rows x 4 sub-blocks per wave instruction; that pattern is conflict-free with sub-block sb at (sb + (n >> 1)) & 3.
constexpr int WSPLIT_ROW_BYTES = 192;  // BKC = 32
__host__ __device__ inline size_t wsplit_bytes(int64_t N, int64_t K) { return (size_t)N * (size_t)K * 6; }
template <int BKC = 32, int M16 = 0>
__device__ __forceinline__ int wsplit_sub_offset(int n, int sb) {
    static_assert(BKC == 32 || (BKC == 16 && !M16), "chunk depth");
    if (M16) return ((sb + (n >> 1)) & 3) * 48;
    return BKC == 32 ? ((sb + (n >> 2)) & 3) * 48 : ((sb + (n >> 3)) & 1) * 48;
}

template <int BKC, int M16 = 0>
__global__ void __launch_bounds__(256) split_weights_kernel(const float* W, uint8_t* out, int N, int K) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one (n, group of 8 k) per thread
    const int groups = K / 8;
    if (idx >= (int64_t)N * groups) return;
    constexpr int SUBS = BKC / 8, ROW = 6 * BKC;
    const int n = (int)(idx / groups), g8 = (int)(idx % groups);
    const int chunk = g8 / SUBS, sb = g8 % SUBS;
    const float* src = W + (int64_t)n * K + g8 * 8;
    const bf16x8x3 s = split3(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4));
    uint8_t* dst = out + ((int64_t)chunk * N + n) * ROW + wsplit_sub_offset<BKC, M16>(n, sb);
    *reinterpret_cast<bf16x8*>(dst) = s.s0;
    *reinterpret_cast<bf16x8*>(dst + 16) = s.s1;
    *reinterpret_cast<bf16x8*>(dst + 32) = s.s2;
}

// The same image of the TRANSPOSE of a [R,Nrows] fp32 matrix (training: dX = dY W runs the plain GEMM on W^T; the image is
// made from W as it lies, no transposed copy in between): image row n = column n of `W`, contraction index k = row k of `W`.
// One (group of 8 k, n) per thread, n fastest: the eight loads of a wave are coalesced along n.
template <int BKC, int M16 = 0>
__global__ void __launch_bounds__(256) split_weights_transposed_kernel(const float* W, uint8_t* out, int Nrows, int R) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int groups = R / 8;
    if (idx >= (int64_t)Nrows * groups) return;
    constexpr int SUBS = BKC / 8, ROW = 6 * BKC;
    const int n = (int)(idx % Nrows), g8 = (int)(idx / Nrows);
    const int chunk = g8 / SUBS, sb = g8 % SUBS;
    const float* src = W + (int64_t)g8 * 8 * Nrows + n;
    f32x4 lo, hi;
#pragma unroll
    for (int i = 0; i < 4; ++i) lo[i] = src[(int64_t)i * Nrows], hi[i] = src[(int64_t)(4 + i) * Nrows];
    const bf16x8x3 sp = split3(lo, hi);
    uint8_t* dst = out + ((int64_t)chunk * Nrows + n) * ROW + wsplit_sub_offset<BKC, M16>(n, sb);
    *reinterpret_cast<bf16x8*>(dst) = sp.s0;
    *reinterpret_cast<bf16x8*>(dst + 16) = sp.s1;
    *reinterpret_cast<bf16x8*>(dst + 32) = sp.s2;
}

// torch.clamp(x, -lim, lim): a NaN stays a NaN (common/diffusionpose.py:193,216-217) - fminf(fmaxf(x, -lim), lim) would turn it into
// -lim, a finite wrong pose.  Both comparisons are false for a NaN.
__device__ __forceinline__ float clamp_keep_nan(float x, float lim) { return x < -lim ? -lim : (x > lim ? lim : x); }

__device__ __forceinline__ float gelu_erf(float x) {
    // nn.GELU() default (approximate='none'): x * 0.5 * (1 + erf(x / sqrt(2)))   (common/mixste.py:25,32)
    // erf(z) = sign(z) (1 - 2^(a Q(a))), a = min(|z|, 4): ONE branch-free form for every z instead of libm's two
    // (|z| < 1: odd polynomial; else 1 - exp(..)), which a wave with mixed lanes executes both of - 18 instead of 38
    // VALU instructions per element, and the fc1 epilogue spent as long on GELU as its wave on MFMAs.  a Q(a) is the
    // degree-9 weighted least-squares fit of log2(erfc(a)) on [0, 4] (weight erfc: what matters is the ABSOLUTE error of
    // erf, it is added to 1).  Against an fp64 evaluation of the same GELU over x ~ N(0, 1.5) and a sweep of [-8, 8]:
    // |error| max 4.5e-7 / mean 2.3e-8, 0.51 ulp mean - torch's own CPU fp32 GELU: 1.2e-6 / 4.4e-8, 0.54 ulp
    // (tests/test_hip_parity.py::test_gelu_against_fp64).
    const float z = x * 0.70710678118654752440f;
    const float a = fminf(fabsf(z), 4.0f);
    float q = 1.1622890269791242e-05f;
    q = q * a + -0.00015313830226659775f;
    q = q * a + 0.000848921830765903f;
    q = q * a + -0.0022762208245694637f;
    q = q * a + 8.650000381749123e-05f;
    q = q * a + 0.02772335335612297f;
    q = q * a + -0.14830751717090607f;
    q = q * a + -0.918442964553833f;
    q = q * a + -1.6279072761535645f;
    const float e = 1.0f - __builtin_amdgcn_exp2f(q * a);
    return x * 0.5f * (1.0f + copysignf(e, z));
}
// d/dx [x Phi(x)] = Phi(x) + x phi(x), Phi from the same erf form (absolute error 4.5e-7), phi = exp(-x^2/2) / sqrt(2 pi)
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float z = x * 0.70710678118654752440f;
    const float a = fminf(fabsf(z), 4.0f);
    float q = 1.1622890269791242e-05f;
    q = q * a + -0.00015313830226659775f;
    q = q * a + 0.000848921830765903f;
    q = q * a + -0.0022762208245694637f;
    q = q * a + 8.650000381749123e-05f;
    q = q * a + 0.02772335335612297f;
    q = q * a + -0.14830751717090607f;
    q = q * a + -0.918442964553833f;
    q = q * a + -1.6279072761535645f;
    const float e = 1.0f - __builtin_amdgcn_exp2f(q * a);
    const float cdf = 0.5f * (1.0f + copysignf(e, z));
    const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);  // exp(-x^2/2) = 2^(-x^2 / (2 ln 2))
    return cdf + x * pdf;
}

// sum over the 32 lanes that share (lane >> 5); every lane of the half ends with the total
__device__ __forceinline__ float half_wave_sum(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v += __shfl_xor(v, 32);
    return half_wave_sum(v);
}

// ----------------------------------------------------------------------------------------------------------------
// Linear layer:  out = epilogue(A[M,K] @ W[N,K]^T + bias)
// ----------------------------------------------------------------------------------------------------------------
enum { EPI_BIAS = 0, EPI_ROWLN = 1, EPI_ROWLN_TRAIN = 2 };  // _TRAIN: + DropPath row factor, + pre-norm sum output

struct GemmParams {
    const float* A;     // [M,K]
    const float* W;     // [N,K]  (torch nn.Linear.weight)
    const float* bias;  // [N]
    float* out;         // EPI_BIAS: [M,N]
    int64_t M;
    int N, K;
    int bf16;  // host-side only: matrix-product mode - 0 fp32 MFMA, 1 bf16-rounded operands, 2 split bf16x3 (needs Wsplit)
    const uint8_t* Wsplit;  // mode 2: the pre-split image of W (split_weights_kernel), wsplit_bytes(N, K) bytes
    int wlayout;            // host-side only: 0 = the 32x32x16 kernels' image, 2 = the gemm16_tile (qkv) image
    int act;   // EPI_BIAS: 0 none, 1 GELU
    // training, EPI_BIAS through the coalesced (slab) epilogue of gemm_tile or gemm16_tile's (gemm_bias checks): with out_act, `out` receives the
    // pre-activation u and out_act gelu(u) (fc1 forward: the backward needs both); with dact_u, out = (acc + bias) * gelu'(dact_u)
    // (the dX GEMM of fc2: the gradient reaches the pre-activation in the same pass)
    float* out_act;
    const float* dact_u;
    // EPI_ROWLN (the workgroup owns whole rows, N == BN; row-per-lane form only):  y = A W^T + bias + resid
    //   z  = post_w ? LN(y; post) : y ;  z += pos[(m / posJ) % posF] if pos ;  out_x = z
    //   n  = next_w ? LN(z; next) : -  ;  out_n = n   |  out_head = n @ head_w^T + head_b
    const float* resid;
    float* out_x;
    float* out_n;
    const float* post_w;
    const float* post_b;
    float post_eps;
    const float* pos;
    int posJ, posF;
    const float* next_w;
    const float* next_b;
    float next_eps;
    const float* head_w;
    const float* head_b;
    float* out_head;
    // EPI_ROWLN_TRAIN only (training forward): y = resid + rowscale[seq(m)] * (A W^T + bias), out_pre = y
    // LayerNorm folded into the consumer GEMM (split-precision inference only; run_blocks sets these up):
    //   producer (EPI_ROWLN with ln_stats): the NEXT LayerNorm of the chain is not applied; the row's (mean, rstd) go to
    //     ln_stats[m][2] and out_n is not written - one [M,C] store and one normalise pass less per whole-row launch;
    //   consumer (EPI_BIAS with ln_in): A is the row CENTRED on its mean, x - mean(x) (what the producer stores since round 5), Wsplit the
    //     image of W (.) g (g = the LayerNorm's weight, scaled along k), bias[n] = sum_k beta_k W_nk + b_n (formed in fp64), and
    //     out = act(rstd_m * acc + bias[n])  ==  act(LN(x) W^T + b) up to rounding.  (ln_s: round 3's uncentred term, never read.)
    float* ln_stats;
    const float* ln_in;
    const float* ln_s;
    // f16x2 "H" pipeline (hgemm.hpp): operands and outputs as H images (see hsplit_store4)
    const uint8_t* Ah;   // A [M,K] split by its producer (hi = f16(a), lo = f16((a - hi) 2^11)); row stride 4 K bytes
    const uint8_t* Wh;   // W [N,K] as the H image of 2^k W (w0 = f16(Ws), w1 = f16(Ws - w0)), 2^-k in the tail behind it
    uint8_t* out_h;      // EPI_BIAS: the output as an H image [M,N] instead of fp32 `out` (its consumer is another hgemm)
    uint8_t* out_xh;     // EPI_ROWLN: the H image of out_x - CENTRED: of out_x - mean(row), the mean being the one written to ln_stats -
    //                      (A operand of the next qkv / fc1 with the LayerNorm folded, and the residual stream itself)
    uint8_t* out_nh;     // EPI_ROWLN: the H image of the next LayerNorm's output instead of fp32 out_n (LayerNorm not folded)
    const uint8_t* resid_h;  // EPI_ROWLN: the residual as the (centred) H image of x (then `resid` is unused): with the LayerNorm
    //                          folded the residual stream lives in memory in this form ONLY - x - mean(row) as hi + 2^-11 lo, 22-23
    //                          significant bits; the row mean is dropped, no reader of the stream sees it (hgemm.hpp); out_x null
    const float* rowscale;  // [nseq] DropPath factor of the row's sequence, or null (= 1)
    int rs_temporal, rs_J, rs_FJ;  // seq(m) = rs_temporal ? (m / rs_FJ) * rs_J + m % rs_J : m / rs_J
    float* out_pre;
    unsigned long long* stamps;  // diagnostic builds (-DPAFUSE_STAMPS) only: per wave {start, loop end, end}
};

// Diagnostic builds only (-DPAFUSE_XQ_FENCE=1 acquire, 2 release, 3 both, 4 acquire + wait + barrier, 6 = 4 + release; tests/cabi/queue_concurrency.c): every wave of
// every kernel of the denoiser chain opens with an agent-scope acquire and / or closes with an agent-scope release of its
// own, on top of what the command processor does at kernel boundaries - the experiment that tells a cache-maintenance
// problem between dependent kernels under multi-queue concurrency from a problem inside the kernels.  (Round 3: the
// fences moved the failure rate - release 10x down, acquire up - but did not remove it; the cause was inside a kernel,
// the packed-fp32 VALU forms of profiles/r03_bf16_mfma_concurrency.md.)
#ifdef PAFUSE_XQ_FENCE
struct XqFence {
    __device__ __forceinline__ XqFence() {
        if (PAFUSE_XQ_FENCE & 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (PAFUSE_XQ_FENCE & 4) {   // the consumer recipe of cdna_hip_programming.md G16: acquire, wait for it, workgroup barrier
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    __device__ __forceinline__ ~XqFence() {
        if (PAFUSE_XQ_FENCE & 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
};
#define PAFUSE_XQ_GUARD() XqFence xq_fence_guard
#else
#define PAFUSE_XQ_GUARD()
#endif

#ifdef PAFUSE_STAMPS
__device__ __forceinline__ unsigned long long pafuse_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#ifndef PAFUSE_STAMP_SLOTS
#define PAFUSE_STAMP_SLOTS 4
#endif
#define PAFUSE_STAMP(i)                                                                         \
    if (p.stamps && (threadIdx.x & 63) == 0)                                                    \
    p.stamps[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * PAFUSE_STAMP_SLOTS + (i)] = pafuse_stamp()
#else
#define PAFUSE_STAMP(i)
#endif

template <int WM, int WN, int NT>
struct GemmTile {
    static constexpr int NTHR = WM * WN * 64;
    static constexpr int BM = WM * 32;
    static constexpr int BN = WN * NT * 32;
    static constexpr int STAGE_FLOATS = (BM + BN) * LDK;
    static constexpr int W_ROW_SPLIT = WSPLIT_ROW_BYTES / 4;                 // floats per W' row of a chunk
    static constexpr int STAGE_FLOATS_SPLIT = BM * LDK + BN * W_ROW_SPLIT;  // mode 2: A fp32 padded + W' image
};

// ----------------------------------------------------------------------------------------------------------------
// Epilogues on row-per-lane accumulators (shared by gemm_kernel<.., TR = 1> and gemm_dma_kernel): lane (r, h) of wave
// (wm, wn) owns output row m0 + 32*wm + r; acc[nt][4q..4q+3] are columns n0 + 32*(wn*NT + nt) + 8q + 4h + {0,1,2,3}.
// EPI_BIAS: out = act(acc + bias).  EPI_ROWLN / _TRAIN: the whole-row residual + LayerNorm chain of GemmParams; `smem` is
// scratch for the cross-wave row sums (WN > 1; the caller guarantees every wave is done with its staging contents).
// ----------------------------------------------------------------------------------------------------------------
// VEC > 0 (LDS-DMA tiles): the per-column vectors of the chain - bias, post_w, post_b, next_w, next_b, BN floats each - were
// staged by the caller at smem + VEC (stage_epilogue_vectors) and are read from LDS; with VEC = 0 every lane loads its
// 16-byte pieces of them from global memory between the dependent steps of the chain, ~120 small loads per lane that the
// full register file cannot keep in flight (the epilogue was 52 000 cycles of a 64 x 384 tile's 122 000, tools/gemm_life.hip).
template <int WN, int NT, int BM, int EPI, int VEC = 0>
__device__ __forceinline__ void epilogue_row_per_lane(f32x16 (&acc)[NT], const GemmParams& p, const int64_t m0, const int n0,
                                                      const int wm, const int wn, const int r, const int h, float* smem,
                                                      const float ws = 1.0f) {   // ws: 2^-k of an f16x2 weight image (exact), else 1
    constexpr int BNV = WN * NT * 32;
    auto vec4 = [&](const float* gptr, int slot, int n) -> f32x4 {
        if constexpr (VEC > 0) return *reinterpret_cast<const f32x4*>(smem + VEC + slot * BNV + (n - n0));
        else return *reinterpret_cast<const f32x4*>(gptr + n);
    };
    // lane (r, h) owns row m = m0 + 32*wm + r; acc[nt][4q..4q+3] are columns 32*(wn*NT+nt) + 8q + 4h + {0,1,2,3}
    const int64_t m = m0 + wm * 32 + r;
    const bool live = m < p.M;
    const int64_t mo = (live ? m : p.M - 1) * p.N;
    const int nb = n0 + wn * NT * 32 + 4 * h;  // + 32*nt + 8*q
    if constexpr (EPI == EPI_BIAS) {
        // folded LayerNorm (ln_in): A is the CENTRED row x - mean(x), so out = rstd * acc + T[n]; the lane owns the row
        float rstd = 1.0f;
        if (p.ln_in) rstd = p.ln_in[2 * (live ? m : p.M - 1) + 1];
        rstd *= ws;   // (exact: a power of two)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = nb + 32 * nt + 8 * q;
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + n);
                f32x4 v;
                if (p.ln_in) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(rstd, acc[nt][4 * q + e], b4[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[nt][4 * q + e], ws, b4[e]);   // ws = 1: acc + b, same bits
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (p.act) v[e] = gelu_erf(v[e]);
                if (live) *reinterpret_cast<f32x4*>(p.out + mo + n) = v;
            }
        PAFUSE_STAMP(2);
        return;
    } else {
        float* red = smem;  // [slot][BM][WN] cross-wave partial sums (staging LDS is dead after the last barrier)
        const float invC = 1.0f / (float)p.N;
        auto row_total = [&](float s, int slot) {
            s += __shfl_xor(s, 32);
            if (WN > 1) {
                float* rs = red + slot * BM * WN + (wm * 32 + r) * WN;
                if (h == 0) rs[wn] = s;
                __syncthreads();
                s = rs[0];
#pragma unroll
                for (int w = 1; w < WN; ++w) s += rs[w];
            }
            return s;
        };
        auto layer_norm = [&](const float* gw, const float* gb, float eps, int slot, int vslot) {
            float s = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) s += acc[nt][i];
            float mean = row_total(s, slot) * invC;
                asm volatile("" : "+v"(mean));   // ONE rounded value: `x - mean` below must not contract into an fma on the unrounded product (HIP's __fmul_rn is a plain multiply)
            float qv = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float d = acc[nt][i] - mean;
                    qv = fmaf(d, d, qv);   // (explicit fma: both whole-row epilogues contract alike - the same bits on every launch route)
                }
            const float rstd = 1.0f / sqrtf(fmaf(row_total(qv, slot + 1), invC, eps));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = nb + 32 * nt + 8 * q;
                    const f32x4 g4 = vec4(gw, vslot, n);
                    const f32x4 b4 = vec4(gb, vslot + 1, n);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[nt][4 * q + e] = fmaf((acc[nt][4 * q + e] - mean) * rstd, g4[e], b4[e]);
                }
        };
        float rs = 1.0f;
        if constexpr (EPI == EPI_ROWLN_TRAIN) {
            if (p.rowscale) {
                const int64_t mm = live ? m : p.M - 1;
                rs = p.rowscale[p.rs_temporal ? (mm / p.rs_FJ) * p.rs_J + mm % p.rs_J : mm / p.rs_J];
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = nb + 32 * nt + 8 * q;
                const f32x4 b4 = vec4(p.bias, 0, n);
                const f32x4 r4 = *reinterpret_cast<const f32x4*>(p.resid + mo + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (EPI == EPI_ROWLN_TRAIN)
                        acc[nt][4 * q + e] = r4[e] + rs * (acc[nt][4 * q + e] + b4[e]);
                    else
                        acc[nt][4 * q + e] = fmaf(acc[nt][4 * q + e], ws, b4[e]) + r4[e];   // ws = 1: (acc + b) + r, same bits
                }
                if constexpr (EPI == EPI_ROWLN_TRAIN) {
                    if (p.out_pre && live) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[nt][4 * q + e];
                        *reinterpret_cast<f32x4*>(p.out_pre + mo + n) = v;
                    }
                }
            }
        if (p.post_w) layer_norm(p.post_w, p.post_b, p.post_eps, 0, 1);
        if (p.pos) {  // only the first spatial block of a pass
            const int f = (int)(((live ? m : p.M - 1) / p.posJ) % p.posF);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 pe = *reinterpret_cast<const f32x4*>(p.pos + (int64_t)f * p.N + nb + 32 * nt + 8 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[nt][4 * q + e] += pe[e];
                }
        }
        const bool folded = p.next_w && p.ln_stats;
        auto store_x = [&](const float mean) {   // out_x = z - mean (mean = 0: z itself)
            if (p.out_x && live) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[nt][4 * q + e] - mean;
                        *reinterpret_cast<f32x4*>(p.out_x + mo + nb + 32 * nt + 8 * q) = v;
                        // the same values as the H image the next qkv / fc1 reads: columns nb + 32 nt + 8 q .. + 3 are the half
                        // 4 h of the sub-block that starts at column (nb - 4 h) + 32 nt + 8 q
                        if (p.out_xh) hsplit_store4(p.out_xh + 4 * (mo + (nb - 4 * h) + 32 * nt + 8 * q), 4 * h, v);
                    }
            }
        };
        if (!folded) store_x(0.0f);
        if (folded) {
            // the next LayerNorm is folded into the GEMM that consumes it: emit the row's statistics only (the same two
            // fixed-order reductions layer_norm makes), no normalise pass, no out_n
            float s = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) s += acc[nt][i];
            float mean = row_total(s, 2) * invC;
                asm volatile("" : "+v"(mean));   // ONE rounded value: `x - mean` below must not contract into an fma on the unrounded product (HIP's __fmul_rn is a plain multiply)
            float qv = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float d = acc[nt][i] - mean;
                    qv = fmaf(d, d, qv);   // (explicit fma: both whole-row epilogues contract alike - the same bits on every launch route)
                }
            const float rstd = 1.0f / sqrtf(fmaf(row_total(qv, 3), invC, p.next_eps));
            if (live && h == 0 && wn == 0) {
                p.ln_stats[2 * m] = mean;
                p.ln_stats[2 * m + 1] = rstd;
            }
            // The residual stream is stored CENTRED on the row mean (round 5; the image pipelines of hgemm.hpp / xgemm.hpp do the
            // same): the consumer GEMM then multiplies x - mean and its folded LayerNorm is rstd acc + lt - no mean (W g) term, so
            // nothing cancels however large a row's mean is.  The mean itself is carried nowhere: every reader of the stream is
            // a LayerNorm or the residual add that feeds one, and LayerNorm does not see a row's mean.
            store_x(mean);
        } else if (p.next_w) {
            layer_norm(p.next_w, p.next_b, p.next_eps, 2, 3);
            if (p.out_nh && live) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[nt][4 * q + e];
                        hsplit_store4(p.out_nh + 4 * (mo + (nb - 4 * h) + 32 * nt + 8 * q), 4 * h, v);
                    }
            }
            if (p.out_n && live) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[nt][4 * q + e];
                        *reinterpret_cast<f32x4*>(p.out_n + mo + nb + 32 * nt + 8 * q) = v;
                    }
            }
            if (p.out_head) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    float s = 0.f;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 hw = *reinterpret_cast<const f32x4*>(p.head_w + k * p.N + nb + 32 * nt + 8 * q);
#pragma unroll
                            for (int e = 0; e < 4; ++e) s = fmaf(acc[nt][4 * q + e], hw[e], s);
                        }
                    s = row_total(s, 4 + k);
                    if (live && h == 0 && wn == 0) p.out_head[m * 3 + k] = s + p.head_b[k];
                }
            }
        }
        PAFUSE_STAMP(2);
        return;
    }
}

// Workgroup = WM x WN waves, each wave a 32 x (32*NT) strip of the BM x BN tile, K streamed in 32-wide chunks:
// global -> registers (issued before the MFMAs of the current chunk) -> LDS (after them), NSTAGE LDS buffers.
// Fragment reads are ds_read_b128: lane (r = lane&31, h = lane>>5) takes k = 8g+4h..8g+4h+3 of row r, and MFMA
// step (g, j) multiplies k = 8g+j (h = 0) and 8g+4+j (h = 1): a permutation of k shared by A and W.
// TR = 1 ("row per lane"): the MFMA operand roles are swapped (W fragment as the A operand), so the accumulator of
// lane (r, h) holds output ROW r of the strip and, per 32-column block, the 16 columns {8q+4h .. 8q+4h+3}: four
// consecutive columns per register quad = one dwordx4, a whole row's statistics = in-lane adds + one
// xor-32 shuffle.  Epilogues then need no LDS transposition.
// BF16 = 1 (opt-in reduced precision, BASELINE configs[1]): the same tiles, staging and epilogues, but the fp32
// fragments are rounded to bf16 in registers (v_cvt_pk_bf16_f32, RNE) and multiplied by v_mfma_f32_32x32x16_bf16 with
// fp32 accumulation: per 16-wide K step lane (r, h) contributes k = {16s + 4h + 0..3} u {16s + 8 + 4h + 0..3} - the two
// fragments it already holds - for A and W alike, so no data moves differently; only the products are bf16 x bf16.
// BF16 = 2 (split precision, "bf16x3"): fp32-equivalent products on the bf16 matrix cores.  A stays fp32 in memory and
// in LDS and is split into three bf16 slices in registers when a fragment is read (each wave owns its rows, so every A
// element is split once per workgroup); W comes pre-split (p.Wsplit, the W' image above) and is staged as it lies.  Per
// 16-deep step lane (r, h) multiplies k = 16*s2 + 8*h + 0..7 - six MFMAs per (A, W) fragment pair.
// (the body is a device function: gemm_kernel runs it for workgroup blockIdx.x of gridDim.x, the shared-grid kernels for
// workgroup b of the nb of one slot of a grouped launch; nb may exceed the tile count, surplus workgroups return)
template <int WM, int WN, int NT, int EPI, int NSTAGE, int TR = 0, int BF16 = 0>
__device__ __forceinline__ void gemm_tile(const GemmParams& p, const int b, const int nb, float* smem) {
    using T = GemmTile<WM, WN, NT>;
    constexpr int NTHR = T::NTHR, BM = T::BM, BN = T::BN;
    constexpr int A_LD = (BM * 8 + NTHR - 1) / NTHR, W_LD = (BN * 8 + NTHR - 1) / NTHR;  // float4 per thread per chunk
    constexpr bool SPLIT = BF16 == 2;
    constexpr int W_ROW = SPLIT ? T::W_ROW_SPLIT : LDK;  // floats per W row of a stage
    float* As = smem;
    float* Ws = smem + NSTAGE * BM * LDK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_n = p.N / BN;
    PAFUSE_STAMP(0);
    // XCD-aware tile order: workgroups b and b+8 share an XCD (round-robin dispatch), so hand each XCD a
    // contiguous run of tiles - the N-tiles of one M-tile then hit the same L2 for their A rows (speed only).
    int tile;
    {
        const int xcd = b & 7, q = nb >> 3, rem = nb & 7;
        tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (b >> 3);
    }
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int64_t m0 = (int64_t)tile_m * BM;
    if (m0 >= p.M) return;  // padded slot of a grouped launch (workgroup-uniform)
    const int n0 = tile_n * BN;
    const int K = p.K;

    // Staging addresses: a wave-uniform base (SGPR pair) plus one 32-bit per-thread offset; thread t stages the
    // float4 at row (t>>3) + i*(NTHR/8), column chunk t&7.  Tail rows of A read a valid row (never stored);
    // surplus threads (BM*8 < NTHR) duplicate row BM-1 (same value, same slot).
    constexpr int RSTEP = NTHR / 8;
    const int trow = tid >> 3, tc4 = tid & 7;
    const float* Abase = p.A + m0 * K;
    const float* Wbase = p.W + (int64_t)n0 * K;
    int a_off[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        int row = trow + i * RSTEP;
        row = row < BM ? row : BM - 1;
        const int64_t lim = p.M - 1 - m0;  // >= 0
        const int grow = row < lim ? row : (int)lim;
        a_off[i] = grow * K + tc4 * 4;
    }
    int w_row0 = trow;  // BN*8 is a multiple of NTHR for every instantiated tile except when clamped below
    const int a_dst0 = (trow < BM ? trow : BM - 1) * LDK + tc4 * 4;
    const int w_dst0 = trow * LDK + tc4 * 4;
    auto a_dst = [&](int i) {
        if ((i + 1) * RSTEP <= BM) return a_dst0 + i * RSTEP * LDK;  // compile-time after unrolling
        int row = trow + i * RSTEP;
        row = row < BM ? row : BM - 1;
        return row * LDK + tc4 * 4;
    };
    static_assert(SPLIT || (BN * 8) % NTHR == 0, "W staging must divide evenly");
    const int w_off = w_row0 * K + tc4 * 4;
    // split mode: the tile's chunk of the W' image is BN * 192 contiguous bytes, copied as it lies (16-byte pieces)
    constexpr int WS_PIECES = BN * (WSPLIT_ROW_BYTES / 16), WS_LD = (WS_PIECES + NTHR - 1) / NTHR;
    const uint8_t* Wsbase = SPLIT ? p.Wsplit + (int64_t)n0 * WSPLIT_ROW_BYTES + tid * 16 : nullptr;
    const int64_t ws_chunk = (int64_t)p.N * WSPLIT_ROW_BYTES;  // bytes per K chunk of the image
    f32x4 a_reg[A_LD], w_reg[SPLIT ? WS_LD : W_LD];
    auto load_chunk = [&](int kc) {
        const float* Ak = Abase + kc * BK;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) a_reg[i] = *reinterpret_cast<const f32x4*>(Ak + a_off[i]);
        if constexpr (SPLIT) {
            const uint8_t* Wk = Wsbase + kc * ws_chunk;
#pragma unroll
            for (int i = 0; i < WS_LD; ++i)
                if ((i + 1) * NTHR <= WS_PIECES || tid + i * NTHR < WS_PIECES)
                    w_reg[i] = *reinterpret_cast<const f32x4*>(Wk + i * NTHR * 16);
        } else {
            const float* Wk = Wbase + kc * BK;
#pragma unroll
            for (int i = 0; i < W_LD; ++i) w_reg[i] = *reinterpret_cast<const f32x4*>(Wk + (int64_t)i * RSTEP * K + w_off);
        }
    };
    auto store_chunk = [&](float* Ad, float* Wd) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) *reinterpret_cast<f32x4*>(Ad + a_dst(i)) = a_reg[i];
        if constexpr (SPLIT) {
#pragma unroll
            for (int i = 0; i < WS_LD; ++i)
                if ((i + 1) * NTHR <= WS_PIECES || tid + i * NTHR < WS_PIECES)
                    *reinterpret_cast<f32x4*>(Wd + (tid + i * NTHR) * 4) = w_reg[i];
        } else {
#pragma unroll
            for (int i = 0; i < W_LD; ++i) *reinterpret_cast<f32x4*>(Wd + w_dst0 + i * RSTEP * LDK) = w_reg[i];
        }
    };
    load_chunk(0);
    store_chunk(As, Ws);
    __syncthreads();

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;

    const int a_frag = (wm * 32 + r) * LDK + (SPLIT ? 8 : 4) * h;
    const int w_frag = (wn * NT * 32 + r) * W_ROW + (SPLIT ? 0 : 4 * h);
    // split mode: float offset of this lane's sub-block (s2, h) inside a W' row (rotation by row, see split_weights_kernel)
    const int ws_sub[2] = {wsplit_sub_offset<32>(r, h) / 4, wsplit_sub_offset<32>(r, 2 + h) / 4};
    const int nk = K / BK;
    for (int kc = 0; kc < nk; ++kc) {
        const int cur = (NSTAGE == 2) ? (kc & 1) : 0;
        const bool more = kc + 1 < nk;
        if (more) load_chunk(kc + 1);
        const float* Ac = As + cur * BM * LDK + a_frag;
        const float* Wc = Ws + cur * BN * W_ROW + w_frag;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (SPLIT) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8x3 a = split3(*reinterpret_cast<const f32x4*>(Ac + 16 * s2),
                                          *reinterpret_cast<const f32x4*>(Ac + 16 * s2 + 4));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float* wp = Wc + nt * 32 * W_ROW + ws_sub[s2];
                    const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wp);
                    const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wp + 4);
                    const bf16x8 w2 = *reinterpret_cast<const bf16x8*>(wp + 8);
                    // small terms first, the leading product last
                    if constexpr (TR) {
                        acc[nt] = mfma_bf16_k16(w0, a.s2, acc[nt]);
                        acc[nt] = mfma_bf16_k16(w2, a.s0, acc[nt]);
                        acc[nt] = mfma_bf16_k16(w1, a.s1, acc[nt]);
                        acc[nt] = mfma_bf16_k16(w0, a.s1, acc[nt]);
                        acc[nt] = mfma_bf16_k16(w1, a.s0, acc[nt]);
                        acc[nt] = mfma_bf16_k16(w0, a.s0, acc[nt]);
                    } else {
                        acc[nt] = mfma_bf16_k16(a.s2, w0, acc[nt]);
                        acc[nt] = mfma_bf16_k16(a.s0, w2, acc[nt]);
                        acc[nt] = mfma_bf16_k16(a.s1, w1, acc[nt]);
                        acc[nt] = mfma_bf16_k16(a.s1, w0, acc[nt]);
                        acc[nt] = mfma_bf16_k16(a.s0, w1, acc[nt]);
                        acc[nt] = mfma_bf16_k16(a.s0, w0, acc[nt]);
                    }
                }
            }
        } else if constexpr (BF16 != 0) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 a8 = to_bf16x8(*reinterpret_cast<const f32x4*>(Ac + 16 * s2),
                                            *reinterpret_cast<const f32x4*>(Ac + 16 * s2 + 8));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const bf16x8 w8 = to_bf16x8(*reinterpret_cast<const f32x4*>(Wc + nt * 32 * LDK + 16 * s2),
                                                *reinterpret_cast<const f32x4*>(Wc + nt * 32 * LDK + 16 * s2 + 8));
                    acc[nt] = TR ? mfma_bf16_k16(w8, a8, acc[nt])
                                 : mfma_bf16_k16(a8, w8, acc[nt]);
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 af = *reinterpret_cast<const f32x4*>(Ac + 8 * g);
                f32x4 wf[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) wf[nt] = *reinterpret_cast<const f32x4*>(Wc + nt * 32 * LDK + 8 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt] = TR ? __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt][j], af[j], acc[nt], 0, 0, 0)
                                     : __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], wf[nt][j], acc[nt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (NSTAGE == 1) __syncthreads();  // everyone done reading before the single buffer is refilled
        if (more) {
            const int nxt = (NSTAGE == 2) ? (cur ^ 1) : 0;
            store_chunk(As + nxt * BM * LDK, Ws + nxt * BN * W_ROW);
        }
        __syncthreads();
    }

    PAFUSE_STAMP(1);
    if constexpr (TR) {
        epilogue_row_per_lane<WN, NT, BM, EPI>(acc, p, m0, n0, wm, wn, r, h, smem);
        return;
    }
    // accumulator element (nt, reg) of this lane is  row = (reg&3) + 8*(reg>>2) + 4*h,  col = 32*nt + r  of the strip
    if constexpr (EPI == EPI_BIAS) {
        // Coalesced store: each wave transposes its strip through its own LDS slab, NTH 32-column blocks at a
        // time, and writes whole 128/256-byte row segments with dwordx4 stores (16 instead of 64 store
        // instructions per lane for a 32x128 strip; the scalar-store tail cost 17 % of the kernel).
        constexpr int WAVES = WM * WN;
        constexpr int FIT = (NSTAGE * T::STAGE_FLOATS / (WAVES * 32) - 4) / 32;
        constexpr int NTH = FIT >= 4 ? 4 : (FIT >= 2 ? 2 : 1);
        static_assert(FIT >= 1, "epilogue slab does not fit the staging LDS");
        constexpr int ST = 32 * NTH + 4;
        float* slab = smem + wave * 32 * ST;
        const int64_t mw = m0 + wm * 32;
        // folded LayerNorm (ln_in; A is the centred row): lane i < 32 fetches rstd of the strip's row i now and parks it in the
        // slab's pad columns behind the first barrier below, where phase 2 reads it from LDS
        float st_a = 1.0f;
        if (p.ln_in && lane < 32) st_a = p.ln_in[2 * (mw + lane < p.M ? mw + lane : p.M - 1) + 1];
#pragma unroll
        for (int nt0 = 0; nt0 < NT; nt0 += NTH) {
            const int nth = (NT - nt0) < NTH ? (NT - nt0) : NTH;  // compile-time after unrolling
            const int c4n = 8 * (nth == 3 ? 4 : nth);  // float4 per row segment (nth is 1, 2 or 4 by construction)
            const int rpi = 64 / c4n;
            const int row_in = lane / c4n, c4 = lane % c4n;
            const int ncol = n0 + (wn * NT + nt0) * 32 + c4 * 4;
            f32x4 t4 = {0.f, 0.f, 0.f, 0.f};
            if (p.ln_in && c4 < 8 * nth) t4 = *reinterpret_cast<const f32x4*>(p.bias + ncol);   // this lane's column vector of phase 2, in flight during phase 1
#pragma unroll
            for (int q = 0; q < NTH; ++q) {
                if (q < nth) {
                    if (p.ln_in) {   // the row factors are applied after the transpose, where a lane owns row pieces
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg)
                            slab[((reg & 3) + 8 * (reg >> 2) + 4 * h) * ST + q * 32 + r] = acc[nt0 + q][reg];
                    } else {
                        const float bv = p.bias[n0 + (wn * NT + nt0 + q) * 32 + r];
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            float v = acc[nt0 + q][reg] + bv;
                            if (p.act) v = gelu_erf(v);
                            slab[((reg & 3) + 8 * (reg >> 2) + 4 * h) * ST + q * 32 + r] = v;
                        }
                    }
                }
            }
            if (nt0 == 0 && p.ln_in && lane < 32) slab[lane * ST + 32 * NTH] = st_a;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 32 * 8 * NTH / 64; ++i) {
                const int row = i * rpi + row_in;
                if (row < 32 && c4 < 8 * nth && mw + row < p.M) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(slab + row * ST + c4 * 4);
                    if (p.ln_in) {
                        const float rstd = slab[row * ST + 32 * NTH];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = fmaf(rstd, v[e], t4[e]);
                            if (p.act) v[e] = gelu_erf(v[e]);
                        }
                    }
                    if (p.dact_u) {
                        const f32x4 u = *reinterpret_cast<const f32x4*>(p.dact_u + (mw + row) * p.N + ncol);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= gelu_erf_grad(u[e]);
                    }
                    *reinterpret_cast<f32x4*>(p.out + (mw + row) * p.N + ncol) = v;
                    if (p.out_act) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                        *reinterpret_cast<f32x4*>(p.out_act + (mw + row) * p.N + ncol) = v;
                    }
                }
            }
            if (nt0 + NTH < NT) __syncthreads();
        }
        PAFUSE_STAMP(2);
        return;
    } else {
        static_assert(TR, "the whole-row epilogue exists in row-per-lane (TR) form only");
    }
}

template <int WM, int WN, int NT, int EPI, int NSTAGE, int MINW = 1, int TR = 0, int BF16 = 0>
__global__ void __launch_bounds__(WM* WN * 64, MINW) gemm_kernel(const GemmParams p) {
    PAFUSE_XQ_GUARD();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_tile<WM, WN, NT, EPI, NSTAGE, TR, BF16>(p, blockIdx.x, gridDim.x, smem);
}

// ----------------------------------------------------------------------------------------------------------------
// Plain split-precision linear layer on v_mfma_f32_16x16x32_bf16 (round 3; the qkv layers):
//   out = act(A[M,K] @ W'[N,K]^T + bias),  or with the LayerNorm folded in (p.ln_in)  act(rstd (acc - mean ls) + lt).
//
// The chip is power-limited under the bf16 matrix instructions, and the 16 x 16 x 32 shape costs less energy per FLOP than
// 32 x 32 x 16 (half the accumulator traffic per MAC): the same six-product loop holds 2.1 - 2.3 GHz instead of 1.75 - 1.9
// (tools/mfma_bf16_peak.hip, profiles/r03_mfma_shape_ab.log).  At the qkv shapes that is 3 - 10 % per launch against
// gemm_tile<.., BF16 = 2> (profiles/r03_mfma16_plain_kernel.log); at fc1 the two are equal and fc1 stays where it was.
// Same arithmetic per product as gemm_tile<.., BF16 = 2> (bf16 x bf16 exact in the fp32 accumulator, small terms first),
// one rounding per 32-deep step instead of one per 16.
// Workgroup = 4 waves, tile 128 tokens x BN columns, one 32-deep chunk per stage (register-staged, as gemm_tile):
//   A stage : fp32 in FRAGMENT order - for every group of 16 tokens two 1 KiB blocks (o = 0, 1), block o holds for lane
//             (c = l & 15, qd = l >> 4) the four floats k = 8 qd + 4 o .. + 3 of token c: fragment reads and staging writes
//             are lane-linear 16-byte accesses (conflict-free without padding).
//   W stage : the 32-deep W' image in its M16 layout as it lies ([n][192 B], sub-block sb at (sb + (n >> 1)) & 3).
// The W' fragment is the MFMA's A operand, the token fragment its B operand: lane (c, qd) ends with token c's outputs
// n = 16 nb + 4 qd + {0,1,2,3} of every 16-column block - one dwordx4 store each, 64 contiguous bytes per token.
// ----------------------------------------------------------------------------------------------------------------
template <int NB>   // NB = BN / 16 column blocks per wave (every wave spans the tile's columns)
struct Tile16 {
    static constexpr int NTHR = 256, BM = 128, BN = NB * 16;
    static constexpr int A_BYTES = BM * 128, W_BYTES = BN * WSPLIT_ROW_BYTES, STAGE_BYTES = A_BYTES + W_BYTES;
    static constexpr int A_LD = BM * 8 / NTHR;                        // float4 per thread per chunk (4)
    static constexpr int W_PIECES = W_BYTES / 16, W_LD = (W_PIECES + NTHR - 1) / NTHR;
};

__device__ __forceinline__ f32x4 mfma16_bf16_k32(const bf16x8 a, const bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <int NB>
__device__ __forceinline__ void gemm16_tile(const GemmParams& p, const int b, const int nb, float* smem) {
    using T = Tile16<NB>;
    constexpr int BM = T::BM, BN = T::BN, A_LD = T::A_LD, W_LD = T::W_LD, NTHR = T::NTHR;
    uint8_t* const As = reinterpret_cast<uint8_t*>(smem);
    uint8_t* const Ws = As + T::A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, qd = lane >> 4;
    const int tiles_n = p.N / BN;
    int tile;
    {
        const int xcd = b & 7, q = nb >> 3, rem = nb & 7;
        tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (b >> 3);
    }
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int64_t m0 = (int64_t)tile_m * BM;
    if (m0 >= p.M) return;
    const int n0 = tile_n * BN;
    const int K = p.K, nk = K / 32;

    // ---- staging.  A: load i of thread (wave, lane) covers token group g = 2 wave + (i >> 1), token c, 16-byte piece
    // j = 4 (i & 1) + qd of the 128-byte row (lanes with equal c read 64 contiguous bytes); it lands at
    // g * 2048 + (j & 1) * 1024 + ((j >> 1) * 16 + c) * 16.  W': the tile's chunk of the image, copied as it lies.
    const float* const Abase = p.A + m0 * K;
    int a_src[A_LD], a_dst[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int g = 2 * wave + (i >> 1), j = 4 * (i & 1) + qd;
        const int row = 16 * g + c;
        const int64_t lim = p.M - 1 - m0;  // >= 0: rows past the last token read a valid row (never stored)
        a_src[i] = (row < lim ? row : (int)lim) * K + 4 * j;
        a_dst[i] = g * 2048 + (j & 1) * 1024 + ((j >> 1) * 16 + c) * 16;
    }
    const uint8_t* const Wsbase = p.Wsplit + (int64_t)n0 * WSPLIT_ROW_BYTES + tid * 16;
    const int64_t ws_chunk = (int64_t)p.N * WSPLIT_ROW_BYTES;
    f32x4 a_reg[A_LD], w_reg[W_LD];
    auto load_chunk = [&](int kc) {
        const float* Ak = Abase + kc * 32;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) a_reg[i] = *reinterpret_cast<const f32x4*>(Ak + a_src[i]);
        const uint8_t* Wk = Wsbase + kc * ws_chunk;
#pragma unroll
        for (int i = 0; i < W_LD; ++i)
            if ((i + 1) * NTHR <= T::W_PIECES || tid + i * NTHR < T::W_PIECES)
                w_reg[i] = *reinterpret_cast<const f32x4*>(Wk + i * NTHR * 16);
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) *reinterpret_cast<f32x4*>(As + a_dst[i]) = a_reg[i];
#pragma unroll
        for (int i = 0; i < W_LD; ++i)
            if ((i + 1) * NTHR <= T::W_PIECES || tid + i * NTHR < T::W_PIECES)
                *reinterpret_cast<f32x4*>(Ws + (tid + i * NTHR) * 16) = w_reg[i];
    };
    load_chunk(0);
    store_chunk();
    __syncthreads();

    f32x4 acc[2][NB];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[g][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint8_t* const a_frag = As + (2 * wave) * 2048 + lane * 16;                          // + g * 2048 + o * 1024
    const uint8_t* const w_frag = Ws + c * WSPLIT_ROW_BYTES + wsplit_sub_offset<32, 1>(c, qd);  // + nb * 16 rows (same rotation)
    for (int kc = 0; kc < nk; ++kc) {
        const bool more = kc + 1 < nk;
        if (more) load_chunk(kc + 1);
        __builtin_amdgcn_s_setprio(1);
        bf16x8x3 a[2];
#pragma unroll
        for (int g = 0; g < 2; ++g)
            a[g] = split3(*reinterpret_cast<const f32x4*>(a_frag + g * 2048), *reinterpret_cast<const f32x4*>(a_frag + g * 2048 + 1024));
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const uint8_t* wp = w_frag + n * 16 * WSPLIT_ROW_BYTES;
            const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wp);
            const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wp + 16);
            const bf16x8 w2 = *reinterpret_cast<const bf16x8*>(wp + 32);
#pragma unroll
            for (int g = 0; g < 2; ++g) {   // small terms first, the leading product last
                acc[g][n] = mfma16_bf16_k32(w0, a[g].s2, acc[g][n]);
                acc[g][n] = mfma16_bf16_k32(w2, a[g].s0, acc[g][n]);
                acc[g][n] = mfma16_bf16_k32(w1, a[g].s1, acc[g][n]);
                acc[g][n] = mfma16_bf16_k32(w0, a[g].s1, acc[g][n]);
                acc[g][n] = mfma16_bf16_k32(w1, a[g].s0, acc[g][n]);
                acc[g][n] = mfma16_bf16_k32(w0, a[g].s0, acc[g][n]);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();  // everyone done reading before the single buffer is refilled
        if (more) store_chunk();
        __syncthreads();
    }
    // ---- epilogue: lane (c, qd) owns token 32 wave + 16 g + c, columns n0 + 16 n + 4 qd + {0,1,2,3}
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int64_t m = m0 + 32 * wave + 16 * g + c;
        if (m >= p.M) continue;
        float rstd = 1.0f;
        if (p.ln_in) rstd = p.ln_in[2 * m + 1];   // folded LayerNorm (A is the centred row): the lane owns the token
        float* const orow = p.out + m * p.N + n0 + 4 * qd;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + n0 + 16 * n + 4 * qd);
            f32x4 v;
            if (p.ln_in) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(rstd, acc[g][n][e], b4[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[g][n][e] + b4[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (p.act) v[e] = gelu_erf(v[e]);
            if (p.dact_u) {   // training: the gradient reaches the pre-activation in the same pass (GemmParams.dact_u)
                const f32x4 u = *reinterpret_cast<const f32x4*>(p.dact_u + m * p.N + n0 + 4 * qd + 16 * n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= gelu_erf_grad(u[e]);
            }
            *reinterpret_cast<f32x4*>(orow + 16 * n) = v;
            if (p.out_act) {  // training: `out` keeps the pre-activation, out_act its GELU
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                *reinterpret_cast<f32x4*>(p.out_act + m * p.N + n0 + 4 * qd + 16 * n) = v;
            }
        }
    }
}

template <int NB, int MINW>
__global__ void __launch_bounds__(256, MINW) gemm16_kernel(const GemmParams p) {
    PAFUSE_XQ_GUARD();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm16_tile<NB>(p, blockIdx.x, gridDim.x, smem);
}

// ----------------------------------------------------------------------------------------------------------------
// Split-precision linear layer, LDS-DMA pipelined form:  out = epilogue(A[M,K] @ W[N,K]^T + bias)  with bf16x3 products.
//
// Same arithmetic and the same W' image as gemm_kernel<.., BF16 = 2>; what changes is how operands reach the matrix
// cores.  With the MFMA time cut to 3/8 the kernel is bound by the bytes a CU can pull in per clock, so (i) tiles are
// as tall as the register file allows (WM waves x 32 rows share one W' stream; 256 x 128: 37 FLOP per staged byte
// against 26 for 128 x 128), (ii) nothing is staged through registers: both operands go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4, 1 KiB per wave instruction) into an NSTAGE ring that runs ahead of the MFMAs behind a
// counted vmcnt, one raw s_barrier per 32-deep chunk, one workgroup per CU.
//   A stage : [BM rows][128 B] unpadded; 16-byte chunk c of row r sits at position c ^ ((r >> 1) & 7), applied on the
//             SOURCE address of the DMA (its LDS side is lane-linear) and again on the fragment reads: with 128-byte
//             rows even and odd rows own the two halves of the 64 banks, and (r >> 1) & 7 spreads the 8 rows of a
//             half over its 8 bank quads - conflict-free ds_read_b128.
//   W stage : [BN rows][192 B], the W' image as it lies (rotated sub-blocks, see split_weights_kernel).
// Accumulators are row-per-lane (operand roles swapped) and the epilogues are epilogue_row_per_lane.
// ----------------------------------------------------------------------------------------------------------------
template <int WM, int WN, int NT, int BKC = 32>
struct DmaTile {
    static_assert(BKC == 32 || BKC == 16, "chunk depth");
    static constexpr int NW = WM * WN, NTHR = NW * 64;
    static constexpr int BM = WM * 32, BN = WN * NT * 32;
    static constexpr int A_ROW = BKC * 4, W_ROW = BKC * 6;  // bytes per row of a stage
    static constexpr int A_BYTES = BM * A_ROW, W_BYTES = BN * W_ROW, STAGE_BYTES = A_BYTES + W_BYTES;
    static constexpr int IA = A_BYTES / 1024, IW = W_BYTES / 1024;  // DMA wave-instructions per chunk
    static constexpr int CNT = (IA + IW + NW - 1) / NW;              // per wave (uniform: surplus slots re-issue the last)
    static_assert(W_BYTES % 1024 == 0 && A_BYTES % 1024 == 0, "whole DMA pieces");
    static_assert(BKC == 16 || NW % 2 == 0, "the A swizzle of a wave's DMA lanes must not depend on the instruction index");
};

template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// LDS read hipcc does not know about: no compiler wait is ever inserted for it - the caller waits (counted) with an
// asm s_waitcnt that names the destination as "+v" before the first use (cdna_hip_programming.md section 5.7, form ii).
template <int OFF>
__device__ __forceinline__ u32x4 lds_read128(uint32_t addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// the whole-row epilogue through per-wave LDS slabs (defined in hgemm.hpp, round 4: every global access of a tensor row covers
// whole row segments; a direct access of row-per-lane accumulators touches 32 rows per instruction) - since round 5 also the
// inference epilogue of the LDS-DMA whole-row tiles below (same arithmetic in the same order as epilogue_row_per_lane)
template <int WN, int NT, int BM, int NW, int NTH, bool HRES, int NSLICE = 2>
__device__ __forceinline__ void epilogue_rows_h(f32x16 (&acc)[NT], const GemmParams& p, const int64_t m0, const int n0, const int wm,
                                                const int wn, const int r, const int h, const int wave, const int lane, float* smem,
                                                const float ws);

// ABL (diagnostic builds of tools/gemm_bench.hip only; results wrong by design): 1 = no fragment reads / split / MFMAs
// (the operand stream alone), 2 = no DMA (the compute side alone, on whatever the LDS holds), 3 = 2 without the split
// arithmetic (raw fragment bits as slices: LDS reads + MFMAs only), 4 = 3 without the epilogue.
// One tile of the kernel as a device function: `b` of `nb` is the workgroup's index in its launch (or in its slot of a
// grouped launch, see grouped_rowln_kernel); nb may exceed the tile count (slots are padded to multiples of 8 so that
// b & 7 stays the XCD of the workgroup): surplus workgroups return at once.
// SLAB: the inference whole-row epilogue through per-wave LDS slabs (the per-part launches); false = the direct row-per-lane form (the
// grouped grids, whose one kernel holds three tile shapes: with the slab form in all three it spilled 1 320 bytes per lane) - same bits
template <int WM, int WN, int NT, int EPI, int NSTAGE, int ABL = 0, int BKC = 32, bool SLAB = true>
__device__ __forceinline__ void gemm_dma_tile(const GemmParams& p, const int b, const int nb, float* smem) {
    using T = DmaTile<WM, WN, NT, BKC>;
    constexpr int NW = T::NW, BM = T::BM, BN = T::BN, CNT = T::CNT, IA = T::IA, IW = T::IW;
    constexpr int NS2 = BKC / 16;                 // 16-deep MFMA steps per chunk
    constexpr int RPI = 1024 / T::A_ROW;          // A rows per DMA instruction (8 at BKC = 32, 16 at BKC = 16)
    constexpr int CPR = T::A_ROW / 16;            // 16-byte chunks per A row
    static_assert(NSTAGE >= 2 && NSTAGE <= 4, "ring depth");
    static_assert(CNT * (NSTAGE - 1) < 64, "vmcnt range");
    uint8_t* const lds = reinterpret_cast<uint8_t*>(smem);
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)smem;  // LDS byte address of the ring

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_n = p.N / BN;
    int tile;
    {
        const int xcd = b & 7, q = nb >> 3, rem = nb & 7;
        tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (b >> 3);
    }
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int64_t m0 = (int64_t)tile_m * BM;
    if (m0 >= p.M) return;  // padded slot of a grouped launch (workgroup-uniform)
    PAFUSE_STAMP(0);
    const int n0 = tile_n * BN;
    const int K = p.K, nk = K / BKC;

    // ---- DMA sources.  Instruction i of a chunk (0 .. IA + IW - 1) belongs to wave i % NW; A instruction ia covers
    // rows RPI ia .. RPI ia + RPI - 1 (lane l: row RPI ia + l / CPR, LDS position l % CPR), W instruction iw the iw-th KiB
    // of the tile's chunk of the image.  A-stage swizzle: chunk c of row r at position c ^ ((r >> 1) & 7) (128-byte
    // rows) / c ^ ((r >> 2) & 3) (64-byte rows): both depend on the lane only, not on the instruction index.
    const float* Abase = p.A + m0 * K;
    const uint8_t* Wbase = p.Wsplit + (int64_t)n0 * T::W_ROW + lane * 16;
    const int64_t ws_chunk = (int64_t)p.N * T::W_ROW;
    const int sw_src = BKC == 32 ? (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)   // source chunk of this lane's
                                 : (lane & 3) ^ ((lane >> 4) & 3);                     // LDS position
    int a_off[CNT];   // float offset of this lane's source in instruction slot j (if that slot is an A instruction)
#pragma unroll
    for (int j = 0; j < CNT; ++j) {
        int i = wave + j * NW;
        i = i < IA + IW ? i : IA + IW - 1;
        const int row = RPI * (i < IA ? i : 0) + lane / CPR;
        const int64_t lim = p.M - 1 - m0;  // >= 0: tail rows read a valid row (never stored)
        const int grow = row < lim ? row : (int)lim;
        a_off[j] = grow * K + sw_src * 4;
    }
    auto issue_piece = [&](int kc, int st, int j) {  // DMA instruction slot j (0 .. CNT - 1) of chunk kc into stage st
        if constexpr (ABL >= 2) return;
        uint8_t* const sa = lds + st * T::STAGE_BYTES;
        int i = wave + j * NW;  // wave-uniform
        i = i < IA + IW ? i : IA + IW - 1;
        if (i < IA)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Abase + kc * BKC + a_off[j]),
                                             (__attribute__((address_space(3))) void*)(sa + i * 1024), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(Wbase + kc * ws_chunk + (i - IA) * 1024),
                (__attribute__((address_space(3))) void*)(sa + T::A_BYTES + (i - IA) * 1024), 16, 0, 0);
    };
    auto issue = [&](int kc, int st) {
#pragma unroll
        for (int j = 0; j < CNT; ++j) issue_piece(kc, st, j);
    };

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;

    // fragment addresses (bytes inside a stage)
    const int sw = BKC == 32 ? (r >> 1) & 7 : (r >> 2) & 3;
    const int a_row = (wm * 32 + r) * T::A_ROW;
    int a_pos[4];  // step s2, half e: position of logical chunk 4 s2 + 2 h + e (entries 2, 3 unused at BKC = 16)
#pragma unroll
    for (int x = 0; x < 4; ++x) a_pos[x] = a_row + (((4 * (x >> 1) + 2 * h + (x & 1)) ^ sw) & (CPR - 1)) * 16;
    const int w_row = T::A_BYTES + (wn * NT * 32 + r) * T::W_ROW;
    const int w_sub[2] = {w_row + wsplit_sub_offset<BKC>(r, h), w_row + wsplit_sub_offset<BKC>(r, (2 + h) % (BKC / 8))};

    if constexpr (NS2 == 1 && NSTAGE == 3 && ABL != 1) {
        // ---- cross-chunk software pipeline (16-deep chunks, three stages: chunk kc being multiplied, chunk kc + 1 landed
        // and visible, chunk kc + 2 in flight).  Everything chunk kc + 1 needs from LDS before its first MFMA - its A
        // fragment, split into slices, and the W' fragment of its group 0 - is fetched and computed inside the MFMA
        // gaps of chunk kc, so a wave leaves the barrier with its operands in registers and the matrix pipe never
        // waits for a split (one split per tile is exposed, in the prologue).
        static_assert(NT >= 1, "groups");
#define PAFUSE_PIN_ACC(A) asm volatile("" : "+v"(A))
#define PAFUSE_PIN_PAIR(P) asm volatile("" : "+v"((P).x0), "+v"((P).x1), "+v"((P).s0), "+v"((P).s1), "+v"((P).s2))
        issue(0, 0);
        if (1 < nk) issue(1, 1);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        u32x4 wf[2][3], cur[3], nxt[3];
        SplitPair sp[4];
        auto load_w = [&](auto SLOT, auto NTI, uint32_t stage_addr) {  // W' fragment of column block NTI into wf[SLOT]
            constexpr int slot = decltype(SLOT)::value, nti = decltype(NTI)::value;
            constexpr int full = nti * 32 * T::W_ROW;
            constexpr int off = full + 32 < 65536 ? full : 0;
            const uint32_t addr = stage_addr + (uint32_t)w_sub[0] + (uint32_t)(full - off);
            wf[slot][0] = lds_read128<off>(addr);
            wf[slot][1] = lds_read128<off + 16>(addr);
            wf[slot][2] = lds_read128<off + 32>(addr);
        };
        {   // chunk 0: the one exposed fragment read + split of the tile
            u32x4 a_lo = lds_read128<0>(lds0 + (uint32_t)a_pos[0]);
            u32x4 a_hi = lds_read128<0>(lds0 + (uint32_t)a_pos[1]);
            load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, lds0);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(a_lo), "+v"(a_hi), "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(wf[0][2]));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sp[q].x0 = __builtin_bit_cast(float, q < 2 ? a_lo[2 * q] : a_hi[2 * q - 4]);
                sp[q].x1 = __builtin_bit_cast(float, q < 2 ? a_lo[2 * q + 1] : a_hi[2 * q - 3]);
                sp[q].template stage<0>(), sp[q].template stage<1>(), sp[q].template stage<2>();
                sp[q].template stage<3>(), sp[q].template stage<4>();
                cur[0][q] = sp[q].s0, cur[1][q] = sp[q].s1, cur[2][q] = sp[q].s2;
            }
        }
        for (int kc = 0; kc < nk; ++kc) {
            if (kc > 0) {
                wait_vmcnt<0>();               // chunk kc + 1 (issued during chunk kc - 1) has landed
                __builtin_amdgcn_s_barrier();  // ... for every wave; every wave is done with chunk kc - 1
            }
            const bool refill = kc + 2 < nk;
            const int kn = kc + 2, stn = kn % 3;
            const uint32_t s_cur = lds0 + (uint32_t)((kc % 3) * T::STAGE_BYTES);
            const uint32_t s_nxt = lds0 + (uint32_t)(((kc + 1) % 3) * T::STAGE_BYTES);  // stale on the last chunk: unused
            __builtin_amdgcn_s_setprio(1);
            u32x4 n_lo = lds_read128<0>(s_nxt + (uint32_t)a_pos[0]);
            u32x4 n_hi = lds_read128<0>(s_nxt + (uint32_t)a_pos[1]);
            static_for<NT>([&](auto G) {
                constexpr int g = decltype(G)::value;
                constexpr int q0 = ABL < 3 ? (4 * g) / NT : 0, q1 = ABL < 3 ? (4 * (g + 1)) / NT : 0;
                // the fragment the NEXT group needs: column block g + 1 of this chunk, or block 0 of the next chunk
                if constexpr (g + 1 < NT)
                    load_w(std::integral_constant<int, (g + 1) & 1>{}, std::integral_constant<int, g + 1>{}, s_cur);
                else
                    load_w(std::integral_constant<int, (g + 1) & 1>{}, std::integral_constant<int, 0>{}, s_nxt);
                if constexpr (g == 0) {
                    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(n_lo), "+v"(n_hi));  // only the three reads just issued fly
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        sp[q].x0 = __builtin_bit_cast(float, q < 2 ? n_lo[2 * q] : n_hi[2 * q - 4]);
                        sp[q].x1 = __builtin_bit_cast(float, q < 2 ? n_lo[2 * q + 1] : n_hi[2 * q - 3]);
                    }
                } else {
                    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(wf[g & 1][0]), "+v"(wf[g & 1][1]), "+v"(wf[g & 1][2]));
                }
                const u32x4(&w)[3] = wf[g & 1];
                auto mm = [&](int wi, int ai) {
                    PAFUSE_PIN_ACC(acc[g]);
                    acc[g] = mfma_bf16_k16(__builtin_bit_cast(bf16x8, w[wi]), __builtin_bit_cast(bf16x8, cur[ai]), acc[g]);
                };
                auto pins = [&]() {
#pragma unroll
                    for (int q = q0; q < q1; ++q) PAFUSE_PIN_PAIR(sp[q]);
                };
                pins();
                mm(0, 2);
                {   // this group's share of the refill DMA, in the shadow of the MFMA just issued
                    constexpr int PER = (CNT + NT - 1) / NT, j0 = g * PER, j1 = (g + 1) * PER < CNT ? (g + 1) * PER : CNT;
                    if constexpr (j0 < j1 && ABL < 2) {
                        asm volatile("" ::: "memory");
                        if (refill) {
#pragma unroll
                            for (int j = j0; j < j1; ++j) issue_piece(kn, stn, j);
                        }
                        asm volatile("" ::: "memory");
                    }
                }
#pragma unroll
                for (int q = q0; q < q1; ++q) sp[q].template stage<0>();
                pins();
                mm(2, 0);
#pragma unroll
                for (int q = q0; q < q1; ++q) sp[q].template stage<1>();
                pins();
                mm(1, 1);
#pragma unroll
                for (int q = q0; q < q1; ++q) sp[q].template stage<2>();
                pins();
                mm(0, 1);
#pragma unroll
                for (int q = q0; q < q1; ++q) sp[q].template stage<3>();
                pins();
                mm(1, 0);
#pragma unroll
                for (int q = q0; q < q1; ++q) {
                    sp[q].template stage<4>();
                    nxt[0][q] = sp[q].s0, nxt[1][q] = sp[q].s1, nxt[2][q] = sp[q].s2;
                }
                pins();
                mm(0, 0);
            });
            // the last group left the next chunk's group-0 fragment in wf[NT & 1]: wait for it, hand everything over
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[NT & 1][0]), "+v"(wf[NT & 1][1]), "+v"(wf[NT & 1][2]));
            if constexpr ((NT & 1) != 0) wf[0][0] = wf[1][0], wf[0][1] = wf[1][1], wf[0][2] = wf[1][2];
            if constexpr (ABL < 3) cur[0] = nxt[0], cur[1] = nxt[1], cur[2] = nxt[2];
            __builtin_amdgcn_s_setprio(0);
        }
#undef PAFUSE_PIN_ACC
#undef PAFUSE_PIN_PAIR
        wait_vmcnt<0>();
    } else {
    #pragma unroll
        for (int s = 0; s < NSTAGE - 1; ++s)
            if (s < nk) issue(s, s);

        for (int kc = 0; kc < nk; ++kc) {
            // chunk kc has landed once at most the NSTAGE - 2 younger chunks of this wave are still in flight
            if (kc + NSTAGE - 2 < nk)
                wait_vmcnt<CNT*(NSTAGE - 2)>();
            else
                wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();  // chunk kc visible to every wave; every wave is done reading chunk kc - 1
            if (kc == 0) { PAFUSE_STAMP(3); }
            // the refill of the stage chunk kc - 1 occupied (chunk kc + NSTAGE - 1) is issued piece by piece inside the MFMA
            // groups below: a DMA instruction costs its wave 60-180 cycles of issue, which a burst here would take from
            // the matrix pipe of every SIMD at once (all waves leave the barrier together)
            const bool refill = kc + NSTAGE - 1 < nk;
            const int kn = kc + NSTAGE - 1, stn = kn % NSTAGE;
            const uint8_t* st = lds + (kc % NSTAGE) * T::STAGE_BYTES;
            if constexpr (ABL == 1) continue;
            // ---- one 32-deep chunk = NG = 2 NT groups (s2, nt) of six MFMAs on one accumulator.  Hand-placed pipeline
            // (sched_barrier fences pin the order): the W' fragments of group g + 1 are read while group g's MFMAs run; the
            // A fragment of step s2 = 1 is split in the gaps of the s2 = 0 MFMA chains, one pair-stage per gap; only the
            // split of step 0 (44 VALU instructions) runs with the matrix pipe idle, once per chunk.
            constexpr int NG = NS2 * NT;
            __builtin_amdgcn_s_setprio(1);
            // Order pins.  hipcc moves loads and register-only instructions freely across __builtin_amdgcn_sched_barrier and
            // waits lgkmcnt(0) where a counted wait would do, so (i) every LDS read of the loop is an asm ds_read_b128 and
            // every wait an asm s_waitcnt that names the registers it makes valid (nothing of hipcc's own is in flight on
            // lgkmcnt inside the loop), (ii) the order is pinned through data: an empty volatile asm that "rewrites" the
            // accumulator sits between consecutive MFMAs of a chain, one that rewrites a SplitPair's registers between
            // consecutive split stages.  (The accumulator pins need it in VGPRs: MINW >= 2, no AGPR allocation.)
    #define PAFUSE_PIN_ACC(A) asm volatile("" : "+v"(A))
    #define PAFUSE_PIN_PAIR(P) asm volatile("" : "+v"((P).x0), "+v"((P).x1), "+v"((P).s0), "+v"((P).s1), "+v"((P).s2))
            const uint32_t sbase = lds0 + (uint32_t)((kc % NSTAGE) * T::STAGE_BYTES);
            u32x4 wf[2][3];
            auto load_w = [&](auto G) {  // the three slices of group G's W' fragment
                constexpr int g = decltype(G)::value;
                constexpr int full = (g % NT) * 32 * T::W_ROW;
                constexpr int off = full + 32 < 65536 ? full : 0;  // ds_read immediates are 16 bits: fold the rest into the address
                const uint32_t addr = sbase + (uint32_t)w_sub[g / NT] + (uint32_t)(full - off);
                wf[g & 1][0] = lds_read128<off>(addr);
                wf[g & 1][1] = lds_read128<off + 16>(addr);
                wf[g & 1][2] = lds_read128<off + 32>(addr);
            };
            u32x4 a_lo = lds_read128<0>(sbase + (uint32_t)a_pos[0]);
            u32x4 a_hi = lds_read128<0>(sbase + (uint32_t)a_pos[1]);
            load_w(std::integral_constant<int, 0>{});
            u32x4 b_lo, b_hi;
            if constexpr (NS2 == 2) {
                b_lo = lds_read128<0>(sbase + (uint32_t)a_pos[2]);
                b_hi = lds_read128<0>(sbase + (uint32_t)a_pos[3]);
                asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(a_lo), "+v"(a_hi));  // the five younger reads stay in flight
            } else {
                asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a_lo), "+v"(a_hi));
            }
            SplitPair sp[4];
            u32x4 cur[3], nxt[3];
    #pragma unroll
            for (int q = 0; q < 4; ++q) {
                sp[q].x0 = __builtin_bit_cast(float, q < 2 ? a_lo[2 * q] : a_hi[2 * q - 4]);
                sp[q].x1 = __builtin_bit_cast(float, q < 2 ? a_lo[2 * q + 1] : a_hi[2 * q - 3]);
                if constexpr (ABL >= 3) {
                    cur[0][q] = __builtin_bit_cast(uint32_t, sp[q].x0), cur[1][q] = __builtin_bit_cast(uint32_t, sp[q].x1);
                    cur[2][q] = cur[0][q];
                    continue;
                }
                sp[q].template stage<0>(), sp[q].template stage<1>(), sp[q].template stage<2>(), sp[q].template stage<3>();
                sp[q].template stage<4>();
                cur[0][q] = sp[q].s0, cur[1][q] = sp[q].s1, cur[2][q] = sp[q].s2;
            }
            static_for<NG>([&](auto G) {
                constexpr int g = decltype(G)::value;
                constexpr int nt = g % NT;
                // pairs of the s2 = 1 fragment split inside this group (all of them are done when group NT - 1 ends)
                constexpr bool SPL = g < NT && ABL < 3 && NS2 == 2;
                constexpr int q0 = SPL ? (4 * nt) / NT : 0, q1 = SPL ? (4 * (nt + 1)) / NT : 0;
                if constexpr (g + 1 < NG) {
                    load_w(std::integral_constant<int, g + 1>{});  // in flight during this group's MFMAs
                    if constexpr (g == 0 && NS2 == 2) {
                        asm volatile("s_waitcnt lgkmcnt(3)"
                                     : "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(wf[0][2]), "+v"(b_lo), "+v"(b_hi));
    #pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            sp[q].x0 = __builtin_bit_cast(float, q < 2 ? b_lo[2 * q] : b_hi[2 * q - 4]);
                            sp[q].x1 = __builtin_bit_cast(float, q < 2 ? b_lo[2 * q + 1] : b_hi[2 * q - 3]);
                        }
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(wf[g & 1][0]), "+v"(wf[g & 1][1]), "+v"(wf[g & 1][2]));
                    }
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[g & 1][0]), "+v"(wf[g & 1][1]), "+v"(wf[g & 1][2]));
                }
                const u32x4(&w)[3] = wf[g & 1];
                auto mm = [&](int wi, int ai) {
                    PAFUSE_PIN_ACC(acc[nt]);
                    acc[nt] = mfma_bf16_k16(__builtin_bit_cast(bf16x8, w[wi]), __builtin_bit_cast(bf16x8, cur[ai]), acc[nt]);
                };
                auto pins = [&]() {
    #pragma unroll
                    for (int q = q0; q < q1; ++q) PAFUSE_PIN_PAIR(sp[q]);
                };
                // small terms first, the leading product last; one split stage per gap
                pins();
                mm(0, 2);
                {   // this group's share of the refill DMA, in the shadow of the MFMA just issued
                    constexpr int PER = (CNT + NG - 1) / NG, j0 = g * PER, j1 = (g + 1) * PER < CNT ? (g + 1) * PER : CNT;
                    if constexpr (j0 < j1 && ABL < 2) {
                        asm volatile("" ::: "memory");
                        if (refill) {
    #pragma unroll
                            for (int j = j0; j < j1; ++j) issue_piece(kn, stn, j);
                        }
                        asm volatile("" ::: "memory");
                    }
                }
    #pragma unroll
                for (int q = q0; q < q1; ++q) sp[q].template stage<0>();
                pins();
                mm(2, 0);
    #pragma unroll
                for (int q = q0; q < q1; ++q) sp[q].template stage<1>();
                pins();
                mm(1, 1);
    #pragma unroll
                for (int q = q0; q < q1; ++q) sp[q].template stage<2>();
                pins();
                mm(0, 1);
    #pragma unroll
                for (int q = q0; q < q1; ++q) sp[q].template stage<3>();
                pins();
                mm(1, 0);
    #pragma unroll
                for (int q = q0; q < q1; ++q) {
                    sp[q].template stage<4>();
                    nxt[0][q] = sp[q].s0, nxt[1][q] = sp[q].s1, nxt[2][q] = sp[q].s2;
                }
                pins();
                mm(0, 0);
                if constexpr (g == NT - 1 && ABL < 3 && NS2 == 2) cur[0] = nxt[0], cur[1] = nxt[1], cur[2] = nxt[2];
                if constexpr (g == NT - 1 && ABL >= 3 && NS2 == 2) {
    #pragma unroll
                    for (int q = 0; q < 4; ++q)
                        cur[0][q] = cur[2][q] = __builtin_bit_cast(uint32_t, sp[q].x0), cur[1][q] = __builtin_bit_cast(uint32_t, sp[q].x1);
                }
            });
    #undef PAFUSE_PIN_ACC
    #undef PAFUSE_PIN_PAIR
            __builtin_amdgcn_s_setprio(0);
        }
    }
    PAFUSE_STAMP(1);
    __syncthreads();  // the staging LDS becomes the epilogue's scratch
    constexpr int VEC = EPI == EPI_BIAS ? 0 : (7 * BM * WN + 3) / 4 * 4;  // behind the cross-wave reduction slots
    if constexpr (VEC > 0) {
        static_assert((size_t)(VEC + 5 * BN) * sizeof(float) <= (size_t)NSTAGE * T::STAGE_BYTES, "epilogue vectors must fit the ring");
        const float* const src[5] = {p.bias, p.post_w, p.post_b, p.next_w, p.next_b};
#pragma unroll
        for (int v = 0; v < 5; ++v)
            if (src[v])  // workgroup-uniform
                for (int i = tid; i < BN / 4; i += T::NTHR)
                    *reinterpret_cast<f32x4*>(smem + VEC + v * BN + 4 * i) = *reinterpret_cast<const f32x4*>(src[v] + n0 + 4 * i);
        __syncthreads();
    }
    if constexpr (ABL == 4) {
        float sacc = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc += acc[nt][i];
        if (sacc == 123.456f) p.out[0] = sacc;  // keeps the accumulators alive, stores nothing
        return;
    }
    if constexpr (EPI == EPI_ROWLN && SLAB) {   // inference: rows in and out through per-wave slabs (fp32 residual, fp32 x, (mean, rstd))
        constexpr size_t RINGF = (size_t)NSTAGE * T::STAGE_BYTES / sizeof(float);
        constexpr auto need = [](int nth) { return (size_t)VEC + 5 * BN + (size_t)NW * 32 * (32 * nth + 4); };
        constexpr int NTH = (NT % 2 == 0 && need(2) <= RINGF) ? 2 : 1;
        static_assert(need(NTH) <= RINGF, "epilogue scratch must fit the ring");
        epilogue_rows_h<WN, NT, BM, NW, NTH, false, 2>(acc, p, m0, n0, wm, wn, r, h, wave, lane, smem, 1.0f);
    } else {
        epilogue_row_per_lane<WN, NT, BM, EPI, VEC>(acc, p, m0, n0, wm, wn, r, h, smem);
    }
}

template <int WM, int WN, int NT, int EPI, int NSTAGE, int MINW, int ABL = 0, int BKC = 32>
__global__ void __launch_bounds__(WM* WN * 64, MINW) gemm_dma_kernel(const GemmParams p) {
    PAFUSE_XQ_GUARD();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_dma_tile<WM, WN, NT, EPI, NSTAGE, ABL, BKC>(p, blockIdx.x, gridDim.x, smem);
}

// ----------------------------------------------------------------------------------------------------------------
// Grouped whole-row launch: the same layer (proj, or fc2) of up to GROUP_MAX independent body-part denoisers in ONE
// grid.  Alone, each part's launch fills the 256 CUs badly - a part's 405 / 574 / 709 tiles on 512 workgroup slots run as
// 0.79 / 1.12 / 1.38 rounds; inside one grid the hardware dispatcher hands the next tile of whichever part to each CU
// as it frees (the single-stream schedule; with side streams the parts overlap as kernels of three queues instead).  Slot s owns workgroups first[s] .. first[s+1]-1 (multiples of 8, so b & 7 is still the XCD
// and each slot keeps its XCD-contiguous tile order; surplus workgroups return at once), most expensive tiles first.
// Every variant is a 4-wave LDS-DMA tile on the 16-deep image with a two-stage ring, two workgroups per CU (the LN
// epilogue of one overlaps the K loop of the other): 384 -> 64 x 384, 256 -> 64 x 256, 224 -> 128 x 224 (7 column blocks
// do not split over two waves).  A tile's arithmetic does not depend on the grid it runs in: results are bit-identical
// to the per-part launches of the same variants.
// ----------------------------------------------------------------------------------------------------------------
constexpr int GROUP_MAX = 4;
struct GroupedGemmParams {
    GemmParams p[GROUP_MAX];
    int first[GROUP_MAX + 1];
    int n;
};

template <int EPI>
__global__ void __launch_bounds__(256, 2) grouped_rowln_kernel(const GroupedGemmParams g) {
    PAFUSE_XQ_GUARD();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    int s = 0;
#pragma unroll
    for (int i = 1; i < GROUP_MAX; ++i)
        if (i < g.n && b >= g.first[i]) s = i;
    const GemmParams& p = g.p[s];
    const int lb = b - g.first[s], nb = g.first[s + 1] - g.first[s];
    switch (p.N) {  // workgroup-uniform
        case 384: gemm_dma_tile<2, 2, 6, EPI, 2, 0, 16, false>(p, lb, nb, smem); break;
        case 256: gemm_dma_tile<2, 2, 4, EPI, 2, 0, 16, false>(p, lb, nb, smem); break;
        case 224: gemm_dma_tile<4, 1, 7, EPI, 2, 0, 16, false>(p, lb, nb, smem); break;
        default: break;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Attention between qkv and proj (common/mixste.py:65-79): one (sequence, head) per group of LP/16 waves, each
// wave one tile of 16 queries.  S^T = K Q^T on 16x16x4 MFMAs puts the query on the lane and the keys in the
// accumulator registers, so softmax reduces in registers + two xor-shuffles and the probabilities are already
// the A operand of P V.
// ----------------------------------------------------------------------------------------------------------------
struct AttnParams {
    const float* qkv;  // [M, 3C]
    float* o;          // [M, C]
    uint8_t* o_h;      // or (f16x2 H pipeline): the output as an H image [M, C] (A operand of the proj hgemm); then o is unused
    int64_t nseq;
    int L, C, heads, d;
    int64_t group, group_stride, seq_stride, tok_stride;
    float scale;
};

// Item -> (sequence, head).  The 8 heads of a sequence read interleaved 112..192-byte slices of the same qkv rows;
// workgroups b and b+8 share an XCD (round-robin dispatch), so the heads of one sequence are given to workgroups of
// one XCD and the partially used cache lines are L2 hits for 7 of them (speed only; plain order when nseq % 8 != 0).
template <int ITEMS>
__device__ __forceinline__ void attn_item(int64_t i, int64_t nseq, int heads, int64_t& seq, int& head) {
    if (nseq % 8 == 0) {
        const int64_t blk = i / ITEMS;
        const int lo = (int)(blk % 8);
        const int64_t u = (blk / 8) * ITEMS + (i % ITEMS);
        head = (int)(u % heads);
        seq = (u / heads) * 8 + lo;
    } else {
        seq = i / heads;
        head = (int)(i % heads);
    }
}

template <int LP, int DP, int NW>
__global__ void __launch_bounds__(NW * 64) attn_kernel(const AttnParams p) {
    PAFUSE_XQ_GUARD();
    constexpr int QT = LP / 16, KT = LP / 16, CT = DP / 16, SD = DP / 16, ITEMS = NW / QT, LDV = DP + 4;
    constexpr int NTHR = NW * 64, C4 = DP / 4;
    static_assert(NW % QT == 0, "waves must be a multiple of the query tiles");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                     // [ITEMS][LP][LDV]
    float* Vs = smem + ITEMS * LP * LDV;  // [ITEMS][LP][LDV]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t nitems = p.nseq * p.heads;
    const int C3 = 3 * p.C;

    // this wave's item and Q fragments first (straight from global: query = 16*qt + l15, dk = 16*s + 4*g .. +3), so
    // their latency overlaps the K/V staging below
    const int il = wave / QT, qt = wave % QT;
    const int64_t item = (int64_t)blockIdx.x * ITEMS + il;
    const bool mine = item < nitems;
    int64_t seq = 0;
    int head = 0;
    if (mine) attn_item<ITEMS>(item, p.nseq, p.heads, seq, head);
    const int64_t base = (seq / p.group) * p.group_stride + (seq % p.group) * p.seq_stride;
    const int l15 = lane & 15, g = lane >> 4;
    f32x4 qf[SD];
    {
        const int q = qt * 16 + l15;
#pragma unroll
        for (int s = 0; s < SD; ++s) {
            qf[s] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (mine && q < p.L && 16 * s + 4 * g < p.d)
                qf[s] = *reinterpret_cast<const f32x4*>(p.qkv + (base + q * p.tok_stride) * C3 + head * p.d + 16 * s +
                                                        4 * g);
        }
    }
    // stage K and V of this workgroup's items (zero-padded to LP x DP)
    for (int sl = 0; sl < ITEMS; ++sl) {
        const int64_t it2 = (int64_t)blockIdx.x * ITEMS + sl;
        const bool ok = it2 < nitems;
        int64_t seq2 = 0;
        int head2 = 0;
        if (ok) attn_item<ITEMS>(it2, p.nseq, p.heads, seq2, head2);
        const int64_t base2 = (seq2 / p.group) * p.group_stride + (seq2 % p.group) * p.seq_stride;
        for (int idx = tid; idx < LP * C4; idx += NTHR) {
            const int t = idx / C4, c4 = idx % C4;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (ok && t < p.L && c4 * 4 < p.d) {
                const float* src = p.qkv + (base2 + t * p.tok_stride) * C3 + head2 * p.d + c4 * 4;
                kv = *reinterpret_cast<const f32x4*>(src + p.C);
                vv = *reinterpret_cast<const f32x4*>(src + 2 * p.C);
            }
            *reinterpret_cast<f32x4*>(Ks + (sl * LP + t) * LDV + c4 * 4) = kv;
            *reinterpret_cast<f32x4*>(Vs + (sl * LP + t) * LDV + c4 * 4) = vv;
        }
    }
    __syncthreads();
    if (!mine) return;
    const float* Kb = Ks + il * LP * LDV;
    const float* Vb = Vs + il * LP * LDV;
    // key tiles innermost: KT independent accumulators back to back (a 16x16x4 MFMA has 40 cycles of dependent
    // latency against 32 of issue)
    f32x4 sc[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < SD; ++s) {
        f32x4 kf[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
            kf[kt] = *reinterpret_cast<const f32x4*>(Kb + (kt * 16 + l15) * LDV + 16 * s + 4 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
                sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][j], qf[s][j], sc[kt], 0, 0, 0);
    }
    // sc[kt][reg] = <q, k> for query 16*qt + l15 and key 16*kt + 4*g + reg
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int key = kt * 16 + 4 * g + reg;
            const float v = key < p.L ? sc[kt][reg] * p.scale : -INFINITY;
            sc[kt][reg] = v;
            mx = fmaxf(mx, v);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    // exp((s - max)) as exp2 of a pre-scaled argument: |s - max| stays below ~30 here, so the fp32 product with
    // log2(e) costs at most a few 1e-6 of RELATIVE error on terms that are themselves exponentially small, and the
    // hardware v_exp_f32 is accurate to ~1 ulp; one reciprocal of the row sum replaces KT*4 divisions.
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float e = __builtin_amdgcn_exp2f((sc[kt][reg] - mx) * 1.44269504088896340736f);
            sc[kt][reg] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) sc[kt][reg] *= inv;

    // O^T = V^T P^T: the V element is the A operand (row = channel), the probability the B operand (column =
    // query), so each lane ends with 4 consecutive channels of ONE query per 16-channel block: a dwordx4 store
    f32x4 oc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) oc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float* vrow = Vb + (kt * 16 + 4 * g + reg) * LDV + l15;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                oc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[ct * 16], sc[kt][reg], oc[ct], 0, 0, 0);
        }
    // oc[ct][reg] = O[query 16*qt + l15][channel 16*ct + 4*g + reg]
    const int q = qt * 16 + l15;
    if (q < p.L) {
        if (p.o_h) {   // channel c = head d + 16 ct + 4 g (a multiple of 4): sub-block c / 8, its second half when c & 4
            uint8_t* const hrow = p.o_h + (base + q * p.tok_stride) * p.C * 4;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                if (ct * 16 + 4 * g < p.d) {
                    const int c = head * p.d + ct * 16 + 4 * g;
                    hsplit_store4(hrow + (c >> 3) * 32, c & 4, oc[ct]);
                }
            return;
        }
        float* orow = p.o + (base + q * p.tok_stride) * p.C + head * p.d + 4 * g;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
            if (ct * 16 + 4 * g < p.d) *reinterpret_cast<f32x4*>(orow + ct * 16) = oc[ct];
    }
}

// ----------------------------------------------------------------------------------------------------------------
// qkv projection + attention of one head in ONE kernel (round 3; split-precision inference):
//   o[rows of the tile, head] = softmax(q k^T * scale) v,   (q | k | v) = A[rows] @ W'_head^T + b_head   (+ folded LayerNorm)
// replaces a qkv GEMM launch + an attn_kernel launch (common/mixste.py:65-79) and the [M,3C] buffer between them: q, k, v
// of the tile's tokens never leave the CU.
// Workgroup = 4 waves = one tile of whole sequences x one head.  The tile holds NSEQ = 128 / L sequences, token i of
// sequence s at tile row s L + i (rows past NSEQ L are computed on a valid row and dropped).
//   phase 1  gemm16_tile's K loop (v_mfma_f32_16x16x32_bf16, six products) on the gathered rows against the head's slice
//            of the HEAD-MAJOR image: per head 3 DP rows - q, k, v of the head, each zero-padded from d to DP rows - so a
//            tile's W' stage is one contiguous run; bias / ls / lt are in the same order.
//   phase 2  the accumulators (token on the lane, 4 consecutive columns per register quad) go to LDS as three
//            [rows][DP + 4] fp32 tiles (over the dead staging buffers);
//   phase 3  attn_kernel's arithmetic per (sequence, 16-query tile) - S^T = K Q^T and O^T = V^T P^T on 16x16x4 fp32 MFMAs,
//            softmax in registers - with Q, K, V read from those tiles; o is written once.
// Blocks: b -> XCD x = b & 7, then (tile, head) = ((b >> 3) / heads * 8 + x, (b >> 3) % heads): the eight heads of a tile
// run on one XCD (its A rows are L2 hits for seven of them) and every XCD keeps the whole image (2.6 MB) in its L2.
// ----------------------------------------------------------------------------------------------------------------
struct FqaParams {
    GemmParams g;      // A [M,K = C], Wsplit = head-major M16 image, bias / ln_in / ln_s (head-major order); N = heads * 3 * DP
    float* o;          // [M, C]
    int64_t nseq;
    int L, C, heads, d, nseq_tile;
    int64_t group, group_stride, seq_stride, tok_stride;   // sequence addressing, as AttnParams
    float scale;
};

template <int LP, int DP>
struct FqaTile {
    static constexpr int NB = 3 * DP / 16, LDV = DP + 4, ROWS = 128 + (LP == 48 ? 4 : 0);
    static constexpr int STAGE_BYTES = Tile16<NB>::STAGE_BYTES, QKV_BYTES = 3 * ROWS * LDV * 4;
    static constexpr int LDS_BYTES = STAGE_BYTES > QKV_BYTES ? STAGE_BYTES : QKV_BYTES;
};

template <int LP, int DP>
__global__ void __launch_bounds__(256, 2) fqa_kernel(const FqaParams fp) {
    PAFUSE_XQ_GUARD();
    using FT = FqaTile<LP, DP>;
    constexpr int NB = FT::NB, LDV = FT::LDV, ROWS = FT::ROWS;
    using T = Tile16<NB>;
    constexpr int A_LD = T::A_LD, W_LD = T::W_LD, NTHR = T::NTHR;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const GemmParams& p = fp.g;
    uint8_t* const As = reinterpret_cast<uint8_t*>(smem);
    uint8_t* const Ws = As + T::A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, qd = lane >> 4;
    const int L = fp.L, NSEQ = fp.nseq_tile;
    const int64_t ntiles = (fp.nseq + NSEQ - 1) / NSEQ;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int64_t tile = (int64_t)(idx / fp.heads) * 8 + xcd;
    const int head = idx % fp.heads;
    if (tile >= ntiles) return;   // (workgroup-uniform: the grid is padded to a multiple of 8 tiles)
    const int K = p.K, nk = K / 32;
    const int64_t seq0 = tile * NSEQ;
    const int64_t last_seq = fp.nseq - 1;
    // token (row of A / o) of tile row r: sequence seq0 + r / L, position r % L; rows of absent sequences alias the last one
    auto token_of = [&](int r) -> int64_t {
        int sl = r / L, t = r - sl * L;
        if (sl >= NSEQ) sl = NSEQ - 1, t = L - 1;
        int64_t sq = seq0 + sl;
        if (sq > last_seq) sq = last_seq;
        return (sq / fp.group) * fp.group_stride + (sq % fp.group) * fp.seq_stride + t * fp.tok_stride;
    };

    // ---- phase 1: the K loop of gemm16_tile on gathered rows
    const float* a_ptr[A_LD];
    int a_dst[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int g = 2 * wave + (i >> 1), j = 4 * (i & 1) + qd;
        a_ptr[i] = p.A + token_of(16 * g + c) * K + 4 * j;
        a_dst[i] = g * 2048 + (j & 1) * 1024 + ((j >> 1) * 16 + c) * 16;
    }
    const int n0 = head * 3 * DP;
    const uint8_t* const Wsbase = p.Wsplit + (int64_t)n0 * WSPLIT_ROW_BYTES + tid * 16;
    const int64_t ws_chunk = (int64_t)p.N * WSPLIT_ROW_BYTES;
    f32x4 a_reg[A_LD], w_reg[W_LD];
    auto load_chunk = [&](int kc) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) a_reg[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + kc * 32);
        const uint8_t* Wk = Wsbase + kc * ws_chunk;
#pragma unroll
        for (int i = 0; i < W_LD; ++i)
            if ((i + 1) * NTHR <= T::W_PIECES || tid + i * NTHR < T::W_PIECES)
                w_reg[i] = *reinterpret_cast<const f32x4*>(Wk + i * NTHR * 16);
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) *reinterpret_cast<f32x4*>(As + a_dst[i]) = a_reg[i];
#pragma unroll
        for (int i = 0; i < W_LD; ++i)
            if ((i + 1) * NTHR <= T::W_PIECES || tid + i * NTHR < T::W_PIECES)
                *reinterpret_cast<f32x4*>(Ws + (tid + i * NTHR) * 16) = w_reg[i];
    };
    load_chunk(0);
    store_chunk();
    __syncthreads();
    f32x4 acc[2][NB];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[g][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint8_t* const a_frag = As + (2 * wave) * 2048 + lane * 16;
    const uint8_t* const w_frag = Ws + c * WSPLIT_ROW_BYTES + wsplit_sub_offset<32, 1>(c, qd);
    for (int kc = 0; kc < nk; ++kc) {
        const bool more = kc + 1 < nk;
        if (more) load_chunk(kc + 1);
        __builtin_amdgcn_s_setprio(1);
        bf16x8x3 a[2];
#pragma unroll
        for (int g = 0; g < 2; ++g)
            a[g] = split3(*reinterpret_cast<const f32x4*>(a_frag + g * 2048), *reinterpret_cast<const f32x4*>(a_frag + g * 2048 + 1024));
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const uint8_t* wp = w_frag + n * 16 * WSPLIT_ROW_BYTES;
            const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wp);
            const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wp + 16);
            const bf16x8 w2 = *reinterpret_cast<const bf16x8*>(wp + 32);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                acc[g][n] = mfma16_bf16_k32(w0, a[g].s2, acc[g][n]);
                acc[g][n] = mfma16_bf16_k32(w2, a[g].s0, acc[g][n]);
                acc[g][n] = mfma16_bf16_k32(w1, a[g].s1, acc[g][n]);
                acc[g][n] = mfma16_bf16_k32(w0, a[g].s1, acc[g][n]);
                acc[g][n] = mfma16_bf16_k32(w1, a[g].s0, acc[g][n]);
                acc[g][n] = mfma16_bf16_k32(w0, a[g].s0, acc[g][n]);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();  // everyone done reading before the single buffer is refilled (after the last chunk: before phase 2)
        if (more) store_chunk();
        if (more) __syncthreads();
    }

    // ---- phase 2: q | k | v of the tile's tokens to LDS (bias or the folded LayerNorm applied), over the staging buffers
    float* const Qs = smem;                    // [ROWS][LDV] each
    float* const Ks = Qs + ROWS * LDV;
    float* const Vs = Ks + ROWS * LDV;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int r = 32 * wave + 16 * g + c;
        float rstd = 1.0f;
        if (p.ln_in) rstd = p.ln_in[2 * token_of(r) + 1];   // (A is the centred row)
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int col = 16 * n + 4 * qd;   // 0 .. 3 DP - 1: part = col / DP (a 16-column block never straddles parts)
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + n0 + col);
            f32x4 v;
            if (p.ln_in) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(rstd, acc[g][n][e], b4[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[g][n][e] + b4[e];
            }
            const int part = (16 * n) / DP, cc = col - part * DP;
            *reinterpret_cast<f32x4*>(Qs + part * ROWS * LDV + r * LDV + cc) = v;
        }
    }
    if (ROWS > 128)   // LP = 48: the key / value tiles of the last sequence reach 4 rows past the tile - keep them finite
        for (int i = tid; i < 3 * (ROWS - 128) * LDV; i += NTHR) {
            const int part = i / ((ROWS - 128) * LDV), rem = i % ((ROWS - 128) * LDV);
            Qs[part * ROWS * LDV + 128 * LDV + rem] = 0.f;
        }
    __syncthreads();

#ifdef PAFUSE_FQA_ABL   // timing ablation (tools/fqa_ablation.sh): K loop + phase 2, no attention
    if (Qs[tid] == 12345.678f) fp.o[0] = Ks[tid];
    return;
#endif
    // ---- phase 3: attention per (sequence of the tile, 16-query tile), one item per wave at a time (attn_kernel's arithmetic)
    constexpr int QT = LP / 16, KT = LP / 16, CT = DP / 16, SD = DP / 16;
    const int l15 = c, g4 = qd;
    const int nseq_here = (int)((fp.nseq - seq0) < NSEQ ? (fp.nseq - seq0) : NSEQ);
    for (int item = wave; item < nseq_here * QT; item += 4) {
        const int sl = item / QT, qt = item % QT;
        const int rb = sl * L;
        if (qt * 16 >= L) continue;
        f32x4 qf[SD];
#pragma unroll
        for (int sd = 0; sd < SD; ++sd) qf[sd] = *reinterpret_cast<const f32x4*>(Qs + (rb + qt * 16 + l15) * LDV + 16 * sd + 4 * g4);
        const float* Kb = Ks + rb * LDV;
        const float* Vb = Vs + rb * LDV;
        f32x4 sc[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sd = 0; sd < SD; ++sd) {
            f32x4 kf[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) kf[kt] = *reinterpret_cast<const f32x4*>(Kb + (kt * 16 + l15) * LDV + 16 * sd + 4 * g4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
                    sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][j], qf[sd][j], sc[kt], 0, 0, 0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int key = kt * 16 + 4 * g4 + reg;
                const float v = key < L ? sc[kt][reg] * fp.scale : -INFINITY;
                sc[kt][reg] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float e = __builtin_amdgcn_exp2f((sc[kt][reg] - mx) * 1.44269504088896340736f);
                sc[kt][reg] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) sc[kt][reg] *= inv;
        f32x4 oc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) oc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int key = kt * 16 + 4 * g4 + reg;      // (masked keys read the sequence's own last row: see hgemm.hpp)
                const float* vrow = Vb + (key < L ? key : L - 1) * LDV + l15;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    oc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[ct * 16], sc[kt][reg], oc[ct], 0, 0, 0);
            }
        const int q = qt * 16 + l15;
        if (q < L) {
            float* orow = fp.o + token_of(rb + q) * fp.C + head * fp.d + 4 * g4;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                if (ct * 16 + 4 * g4 < fp.d) *reinterpret_cast<f32x4*>(orow + ct * 16) = oc[ct];
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Row-wise LayerNorm, one wave per row (stand-alone form; inside the loop the norms live in GEMM epilogues)
// ----------------------------------------------------------------------------------------------------------------
constexpr int LN_MAX_PER_LANE = 12;  // C <= 768

__device__ __forceinline__ void wave_layer_norm(float (&v)[LN_MAX_PER_LANE], int C, int lane, const float* w,
                                                const float* b, float eps) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i)
        if (lane + 64 * i < C) s += v[i];
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i)
        if (lane + 64 * i < C) {
            const float d = v[i] - mean;
            q += d * d;
        }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i)
        if (lane + 64 * i < C) v[i] = (v[i] - mean) * rstd * w[lane + 64 * i] + b[lane + 64 * i];
}

__global__ void __launch_bounds__(256) layernorm_kernel(const float* x, const float* w, const float* b, float* out,
                                                        int64_t M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float v[LN_MAX_PER_LANE];
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) v[i] = (lane + 64 * i < C) ? x[row * C + lane + 64 * i] : 0.f;
    wave_layer_norm(v, C, lane, w, b, eps);
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i)
        if (lane + 64 * i < C) out[row * C + lane + 64 * i] = v[i];
}

// ----------------------------------------------------------------------------------------------------------------
// Timestep embedding: sinusoid -> Linear(C,2C) -> GELU -> Linear(2C,C)   (common/mixste.py:127-139,179-184)
// `freqs` is the host-computed omega table so that t*omega is the same fp32 product the reference forms; sin/cos
// are the accurate device functions (arguments reach 999 rad).
// ----------------------------------------------------------------------------------------------------------------
struct TimeEmbedParams {
    const int64_t* t;  // [B] or null -> t_scalar
    int64_t t_scalar;
    const float* freqs;
    const float *w1, *b1, *w3, *b3;
    float* hid;  // [B,2C] scratch
    float* out;  // [B,C]
    int C;
};

// phase 0: hid = GELU(W1 sinusoid(t) + b1)   phase 1: out = W3 hid + b3.   grid = (ceil(rows/8), B), 8 waves per
// workgroup, one output row per wave (coalesced weight-row reads + wave reduction).
template <int PHASE>
__global__ void __launch_bounds__(512) time_embed_kernel(const TimeEmbedParams p) {
    PAFUSE_XQ_GUARD();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y, C = p.C;
    const int K = PHASE == 0 ? C : 2 * C, NOUT = PHASE == 0 ? 2 * C : C;
    if (PHASE == 0) {
        const int half = C / 2;
        const float tf = (float)(p.t ? p.t[b] : p.t_scalar);
        for (int i = tid; i < half; i += 512) {
            const float a = tf * p.freqs[i];
            smem[i] = sinf(a);
            smem[half + i] = cosf(a);
        }
    } else {
        for (int i = tid; i < K; i += 512) smem[i] = p.hid[(int64_t)b * K + i];
    }
    __syncthreads();
    const int o = blockIdx.x * 8 + wave;
    if (o >= NOUT) return;
    const float* w = (PHASE == 0 ? p.w1 : p.w3) + (int64_t)o * K;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += w[k] * smem[k];
    s = wave_sum(s);
    if (lane == 0) {
        if (PHASE == 0)
            p.hid[(int64_t)b * NOUT + o] = gelu_erf(s + p.b1[o]);
        else
            p.out[(int64_t)b * NOUT + o] = s + p.b3[o];
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Embedding of one part: clamp / scale / flip / joint gather of the noised 3-D pose, 2-D broadcast over P,
// Linear(5 -> C) + spatial pos-embed + time embed (common/diffusionpose.py:193-198, 328-335;
// common/mixste.py:227-235), then LayerNorm (norm1 of STEblocks[0]) for the first QKV GEMM.  One wave per token.
// ----------------------------------------------------------------------------------------------------------------
struct EmbedParams {
    const float* x3d;       // [B,P,F,J3,3]  (J3 = joints of the x3d tensor: num_kps in the loop, J in mixste2_forward)
    const float* x2d;       // [B,F,J2,2]
    const float* x2d_flip;  // same, or null
    const int32_t* joints;  // [J] index of this part's joints in the J3/J2 axes, or null (identity)
    const int32_t* perm;    // [J3] flip permutation, or null
    const float *pw, *pb, *pos, *temb;  // [C,5], [C], [J,C], [B,C]
    const float *n_w, *n_b;             // next LayerNorm
    float n_eps;
    float *x, *xn;  // [M,C]  (x may be null when only the H image of x is wanted)
    uint8_t* xh;    // f16x2 H pipeline: the H image [M,C] the first qkv hgemm reads - of x - mean(row) when `stats` is set
    //                 (LayerNorm folded), else of the normalised row (then xn is not written)
    int x_image;    // xh is an X image (bf16x3 on the LDS-DMA pipeline, xgemm.hpp: 6 bytes per element) instead of an H image
    int centre_x;   // with `stats`: the fp32 rows `x` are stored centred, x - mean(row) (mode 2's folded LayerNorm, as the images are)
    float* stats;   // folded LayerNorm (GemmParams::ln_in of the first qkv GEMM): (mean, rstd) of row row0 + i at stats[2 i]
    //                 instead of xn; null = write xn
    int B, P, F, J, J3, C, nflip;
    int do_clamp;
    float scale;  // (float)args.ft2d.scale: the divisor, as torch demotes the Python scalar
    float lim;    // (float)(1.1 * scale) formed in fp64 on the host, as torch.clamp demotes its Python-float bound
    int64_t row0, nrows;  // this launch embeds rows [row0, row0 + nrows) of the part's token matrix
};

// one half-wave per token (8 tokens per 256-thread workgroup): lane li owns the channel quads li, li+32, li+64
constexpr int EMBED_ROWS_PER_BLOCK = 8;
constexpr int EMBED_NV = 3;  // C <= 384

__global__ void __launch_bounds__(256) embed_kernel(const EmbedParams p) {
    PAFUSE_XQ_GUARD();
    const int lane = threadIdx.x & 63, li = lane & 31, hh = lane >> 5;
    const int64_t local = (int64_t)blockIdx.x * EMBED_ROWS_PER_BLOCK + (threadIdx.x >> 6) * 2 + hh;
    const bool live = local < p.nrows;  // uniform per half-wave; dead halves still take part in the shuffles
    const int64_t row = p.row0 + (live ? local : p.nrows - 1);
    const int j = (int)(row % p.J);
    const int f = (int)((row / p.J) % p.F);
    const int64_t rr = row / ((int64_t)p.J * p.F);
    const int pp = (int)(rr % p.P);
    const int b = (int)((rr / p.P) % p.B);
    const int fl = (int)(rr / ((int64_t)p.P * p.B));
    const int jj = p.joints ? p.joints[j] : j;
    const int j3 = fl ? p.perm[jj] : jj;
    const float* s2 = (fl ? p.x2d_flip : p.x2d) + (((int64_t)b * p.F + f) * p.J3 + jj) * 2;
    const float* s3 = p.x3d + ((((int64_t)b * p.P + pp) * p.F + f) * p.J3 + j3) * 3;
    float in[5] = {s2[0], s2[1], s3[0], s3[1], s3[2]};
    if (p.do_clamp) {
        const float lim = p.lim;
#pragma unroll
        for (int i = 2; i < 5; ++i) in[i] = clamp_keep_nan(in[i], lim) / p.scale;
    }
    if (fl) in[2] = -in[2];
    const int NQ = p.C / 4;
    f32x4 v[EMBED_NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < EMBED_NV; ++i) {
        const int c4 = li + 32 * i;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c4 < NQ) {
            const float* wr = p.pw + c4 * 20;  // 4 channels x 5 inputs, contiguous
            f32x4 w4[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) w4[q] = *reinterpret_cast<const f32x4*>(wr + 4 * q);
            const float* wf = reinterpret_cast<const float*>(w4);
            const f32x4 pb = *reinterpret_cast<const f32x4*>(p.pb + 4 * c4);
            const f32x4 ps = *reinterpret_cast<const f32x4*>(p.pos + j * p.C + 4 * c4);
            const f32x4 te = *reinterpret_cast<const f32x4*>(p.temb + (int64_t)b * p.C + 4 * c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = in[0] * wf[5 * e + 0];
                a += in[1] * wf[5 * e + 1];
                a += in[2] * wf[5 * e + 2];
                a += in[3] * wf[5 * e + 3];
                a += in[4] * wf[5 * e + 4];
                a += pb[e];
                a += ps[e];
                a += te[e];
                v[i][e] = a;
                s += a;
            }
            if (live && p.x && !(p.stats && p.centre_x)) *reinterpret_cast<f32x4*>(p.x + row * p.C + 4 * c4) = v[i];

        }
    }
    const float invC = 1.0f / (float)p.C;
    const float mean = half_wave_sum(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < EMBED_NV; ++i)
        if (li + 32 * i < NQ) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[i][e] - mean;
                q += d * d;
            }
        }
    const float rstd = 1.0f / sqrtf(half_wave_sum(q) * invC + p.n_eps);
    if (p.stats) {
        if (live && li == 0) {
            p.stats[2 * local] = mean;
            p.stats[2 * local + 1] = rstd;
        }
        if ((p.xh || (p.centre_x && p.x)) && live) {   // the image of x (and, mode 2, its fp32 rows), CENTRED on the row mean
#pragma unroll
            for (int i = 0; i < EMBED_NV; ++i) {
                const int c4 = li + 32 * i;
                if (c4 < NQ) {
                    f32x4 cv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) cv[e] = v[i][e] - mean;
                    if (p.centre_x && p.x) *reinterpret_cast<f32x4*>(p.x + row * p.C + 4 * c4) = cv;
                    if (!p.xh) continue;
                    if (p.x_image) xsplit_store4(p.xh + (size_t)row * p.C * 6, 4 * c4, cv);
                    else hsplit_store4(p.xh + (row * p.C + 8 * (c4 >> 1)) * 4, 4 * (c4 & 1), cv);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < EMBED_NV; ++i) {
        const int c4 = li + 32 * i;
        if (c4 < NQ) {
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.n_w + 4 * c4);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.n_b + 4 * c4);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g4[e] + b4[e];
            if (live && p.xh && p.x_image) xsplit_store4(p.xh + (size_t)row * p.C * 6, 4 * c4, o);
            else if (live && p.xh) hsplit_store4(p.xh + (row * p.C + 8 * (c4 >> 1)) * 4, 4 * (c4 & 1), o);
            else if (live) *reinterpret_cast<f32x4*>(p.xn + row * p.C + 4 * c4) = o;
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// End of a DDIM step (common/diffusionpose.py:211-223 and :298-312): part concat, un-flip, TTA average, scale,
// clamp -> x_start (written into preds_all[:, step]); epsilon in fp64; img update with the step's noise draw.
// One thread per (b, p, f, joint).  Every fp32 op is individually rounded (no FMA contraction) like the
// reference's separate ATen ops.
// ----------------------------------------------------------------------------------------------------------------
struct FinalizeParams {
    const float* pred[4];  // per part [nflip*B*P*F*Jp, 3]
    int Jp[4];
    const int32_t *joint_part, *joint_local, *perm;
    float* img;          // [B,P,F,J,3] in/out
    const float* noise;  // [B,P,F,J,3] or null when last
    float* out;          // [B,T,P,F,J,3]
    int B, P, F, J, T, step, flip, last;
    float scale, lim;  // (float)scale and (float)(1.1 * scale) (fp64 product), see EmbedParams
    double sr, srm1, c;
    float an_f, c_f, sigma_f;
    int32_t* range_flag;   // device word (or null): set to 1 when a denoiser output of this step is not finite - in 'f16x2' the loud
    //                        form of an activation beyond the fp16 range (|a| >= 65504 -> inf -> NaN); the values themselves flow on as
    //                        NaN, as the reference's would (no clamp hides them)
};

__global__ void __launch_bounds__(256) finalize_kernel(const FinalizeParams p) {
    PAFUSE_XQ_GUARD();
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)p.B * p.P * p.F * p.J;
    if (e >= total) return;
    const int j = (int)(e % p.J);
    const int f = (int)((e / p.J) % p.F);
    const int64_t bp = e / ((int64_t)p.J * p.F);
    const int pp = (int)(bp % p.P);
    const int b = (int)(bp / p.P);
    const int part = p.joint_part[j], lj = p.joint_local[j];
    const float* a = p.pred[part] + ((bp * p.F + f) * p.Jp[part] + lj) * 3;
    float x0[3] = {a[0], a[1], a[2]};
    bool finite = isfinite(x0[0]) && isfinite(x0[1]) && isfinite(x0[2]);
    if (p.flip) {
        const int js = p.perm[j];
        const int part2 = p.joint_part[js], lj2 = p.joint_local[js];
        const float* u = p.pred[part2] + ((((int64_t)p.B * p.P + bp) * p.F + f) * p.Jp[part2] + lj2) * 3;
        finite = finite && isfinite(u[0]) && isfinite(u[1]) && isfinite(u[2]);
        x0[0] = __fdiv_rn(__fadd_rn(x0[0], -u[0]), 2.0f);
        x0[1] = __fdiv_rn(__fadd_rn(x0[1], u[1]), 2.0f);
        x0[2] = __fdiv_rn(__fadd_rn(x0[2], u[2]), 2.0f);
    }
    if (!finite && p.range_flag) *p.range_flag = 1;   // (every writer stores the same word: no atomic needed)
    const float lim = p.lim;
    float* o = p.out + ((((int64_t)b * p.T + p.step) * p.P + pp) * p.F + f) * p.J * 3 + j * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        x0[k] = clamp_keep_nan(__fmul_rn(x0[k], p.scale), lim);
        o[k] = x0[k];
    }
    float* im = p.img + e * 3;
    if (p.last) {
#pragma unroll
        for (int k = 0; k < 3; ++k) im[k] = x0[k];
        return;
    }
    const float* nz = p.noise + e * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double eps64 = __ddiv_rn(__dsub_rn(__dmul_rn(p.sr, (double)im[k]), (double)x0[k]), p.srm1);
        if (p.flip) {
            // pred_noise.float(); then fp32 tensor ops with fp64 0-dim scalars demoted to fp32
            const float eps = (float)eps64;
            im[k] = __fadd_rn(__fadd_rn(__fmul_rn(x0[k], p.an_f), __fmul_rn(p.c_f, eps)), __fmul_rn(p.sigma_f, nz[k]));
        } else {
            // ddim_sample keeps eps in fp64 and casts img at the end (common/diffusionpose.py:264-267)
            const double t1 = (double)__fmul_rn(x0[k], p.an_f);
            const double t2 = __dmul_rn(p.c, eps64);
            const double t3 = (double)__fmul_rn(p.sigma_f, nz[k]);
            im[k] = (float)__dadd_rn(__dadd_rn(t1, t2), t3);
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Hypothesis aggregation behind the all-gather (caller side of the path: main_h3wb.py:327-362 with
// common/utils.py:113-126, common/camera.py:30-60, common/loss.py:36-168).  One thread per (b, t, f, joint) walks
// the P hypotheses once: whole-body pose from parts, per-joint errors, J-Best minimum, P-Agg mean pose, the 2-D
// reprojection argmin of J-Agg, and the part-re-centred variants.  The per-hypothesis errors go to scratch so
// that the (tiny) P-Best means/argmin over (b, f, j) are formed afterwards in a fixed order.
// ----------------------------------------------------------------------------------------------------------------
struct MetricsParams {
    const float* pred;  // [B,T,P,F,J,3] part-centred predictions
    const float* gt;    // [B,F,J,3] part-centred ground truth
    const float* x2d;   // [B,F,J,2]
    const float* traj;  // [B,F,3] root trajectory
    const float* cam;   // [9] f(2) c(2) k(3) p(2)
    const int32_t *conn, *pbroot;  // [J]: connection joint (wb_pose_from_parts), part root (center_pose_parts)
    float *e3, *epb;                       // [B,T,P,F,J] per-hypothesis errors (whole-body / part-centred)
    float *jbest, *pagg, *jagg, *paggpb;   // [B,T,F,J]
    int B, T, P, F, J;
};

__device__ __forceinline__ void wb_joint(const float* pose /*[J,3]*/, const int32_t* conn, int j, float (&o)[3]) {
    // out[j] = pose[j] + pose[conn[j]], joint 0 forced to 0 (the net effect of wb_pose_from_parts' in-place pass)
    if (j == 0) {
        o[0] = o[1] = o[2] = 0.f;
        return;
    }
    const float* a = pose + j * 3;
    const float* c = pose + conn[j] * 3;
    o[0] = a[0] + c[0], o[1] = a[1] + c[1], o[2] = a[2] + c[2];
}

__global__ void __launch_bounds__(256) metrics_kernel(const MetricsParams p) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)p.B * p.T * p.F * p.J;
    if (e >= total) return;
    const int j = (int)(e % p.J);
    const int f = (int)((e / p.J) % p.F);
    const int t = (int)((e / ((int64_t)p.J * p.F)) % p.T);
    const int b = (int)(e / ((int64_t)p.J * p.F * p.T));
    const int jr = p.pbroot[j];
    const float* gpose = p.gt + ((int64_t)b * p.F + f) * p.J * 3;
    float g[3], gr[3];
    wb_joint(gpose, p.conn, j, g);
    wb_joint(gpose, p.conn, jr, gr);
    const float gc[3] = {g[0] - gr[0], g[1] - gr[1], g[2] - gr[2]};
    const float* tr = p.traj + ((int64_t)b * p.F + f) * 3;
    const float* x2 = p.x2d + (((int64_t)b * p.F + f) * p.J + j) * 2;
    const float fx = p.cam[0], fy = p.cam[1], cx = p.cam[2], cy = p.cam[3];
    const float k1 = p.cam[4], k2 = p.cam[5], k3 = p.cam[6], p1 = p.cam[7], p2 = p.cam[8];
    float best3 = INFINITY, best2 = INFINITY, sel3 = 0.f;
    float sm[3] = {0.f, 0.f, 0.f}, smr[3] = {0.f, 0.f, 0.f};
    for (int h = 0; h < p.P; ++h) {
        const float* pose = p.pred + ((((int64_t)b * p.T + t) * p.P + h) * p.F + f) * p.J * 3;
        float w[3], wr[3];
        wb_joint(pose, p.conn, j, w);
        wb_joint(pose, p.conn, jr, wr);
        const float d0 = w[0] - g[0], d1 = w[1] - g[1], d2 = w[2] - g[2];
        const float e3 = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
        const float c0 = (w[0] - wr[0]) - gc[0], c1 = (w[1] - wr[1]) - gc[1], c2 = (w[2] - wr[2]) - gc[2];
        const float epb = sqrtf(c0 * c0 + c1 * c1 + c2 * c2);
        const int64_t o = ((((int64_t)b * p.T + t) * p.P + h) * p.F + f) * p.J + j;
        p.e3[o] = e3;
        p.epb[o] = epb;
        best3 = fminf(best3, e3);
#pragma unroll
        for (int k = 0; k < 3; ++k) sm[k] += w[k], smr[k] += wr[k];
        // 2-D reprojection of the absolute pose (common/camera.py:30-60)
        const float X = w[0] + tr[0], Y = w[1] + tr[1], Z = w[2] + tr[2];
        const float xx = fminf(fmaxf(X / Z, -1.f), 1.f), yy = fminf(fmaxf(Y / Z, -1.f), 1.f);
        const float r2 = xx * xx + yy * yy;
        const float radial = 1.f + (k1 * r2 + k2 * (r2 * r2) + k3 * (r2 * r2 * r2));
        const float tan = p1 * xx + p2 * yy;
        const float u = fx * (xx * (radial + tan) + p1 * r2) + cx, v = fy * (yy * (radial + tan) + p2 * r2) + cy;
        const float q0 = u - x2[0], q1 = v - x2[1];
        const float e2 = sqrtf(q0 * q0 + q1 * q1);
        if (e2 < best2) best2 = e2, sel3 = e3;  // first minimum wins, like torch.min(...).indices
    }
    const float invP = 1.0f / (float)p.P;
    const float m0 = sm[0] * invP - g[0], m1 = sm[1] * invP - g[1], m2 = sm[2] * invP - g[2];
    const float n0 = (sm[0] - smr[0]) * invP - gc[0], n1 = (sm[1] - smr[1]) * invP - gc[1],
                n2 = (sm[2] - smr[2]) * invP - gc[2];
    p.jbest[e] = best3;
    p.jagg[e] = sel3;
    p.pagg[e] = sqrtf(m0 * m0 + m1 * m1 + m2 * m2);
    p.paggpb[e] = sqrtf(n0 * n0 + n1 * n1 + n2 * n2);
}

}  // namespace pafuse
