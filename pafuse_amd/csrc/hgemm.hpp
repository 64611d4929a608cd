// hgemm.hpp - the linear layers of the f16x2 product scheme as ONE LDS-DMA pipelined kernel family ("H pipeline", round 4).
//
// With the matrix work cut to three fp16 MFMAs per fp32-equivalent product (kernels.hpp, "f16x2") the bf16x3 tile set stops
// being matrix-bound: run on it, f16x2 gains 18 %, not 100 % - the kernels then wait for the operand stream (a whole-row
// launch pulls 24.6 B/clk/CU through L2 -> LDS of the ~29 the path sustains), for LDS reads of three weight slices per three
// MFMAs and for the in-register split of A, repeated by every column tile (profiles/r04_f16x2_v1_bench_line.json).  This
// family removes all three:
//   * both operands are "H images": rows of 4 K bytes, per sub-block of 8 k the 16 bytes of the first slice then the 16 of
//     the second - [rows][K/8][hi 8 x f16 | lo 8 x f16].  An ACTIVATION is split ONCE, by the kernel that produces it
//     (whole-row epilogue -> x, attention -> o, fc1 epilogue -> the MLP hidden, embed -> x), into hi = f16(a) and
//     lo = f16((a - hi) 2^11); the bytes are those of the fp32 tensor it replaces.  No VALU work is left in a K loop.
//   * a WEIGHT image holds w0 = f16(Ws), w1 = f16(Ws - w0) of Ws = 2^k W (4 bytes per element instead of 6); the third slice
//     w2 = f16(w0 2^-11) is four v_pk_mul_f16 per fragment, in registers (RNE into the fp16 subnormals exactly as the stored
//     slice of the first implementation was).  Two LDS reads per weight fragment instead of three, 2/3 of the L2 stream.
//   * 128-row tiles (the hands' whole rows: 64) on eight or four waves, one to three workgroups per CU, picked per shape
//     with tools/hgemm_bench.hip (pafuse_hip.hip: hgemm_bias, hgemm_rowln).
// Three kernels live here: hgemm_kernel (one linear layer: plain, or whole-row with residual + LayerNorms), hfqa_kernel (qkv
// projection + attention of whole sequences x 1 - 2 heads) and hmlp_kernel (fc1 -> GELU -> fc2 -> whole-row epilogue with the
// hidden activations in registers).
// Arithmetic per product and its order (lo w2, hi w1, hi w0 into one accumulator, fp32, one rounding per MFMA; the
// accumulator times 2^-k in the epilogue) are those of the first f16x2 kernels - same bits as gemm_tile<.., BF16 = 3> on the
// same operands up to the order of the K sum inside a 16-deep step (identical: both feed k = 16 s + 8 h + j to lane half h).
//
// Stage layout in LDS (one per ring slot): A [BM rows][4 BKC bytes] then W [BN rows][4 BKC bytes], unpadded; the 16-byte
// slot s of row r (s = 2 * sub-block + slice) sits at position s ^ ((r >> 1) & 7) (BKC = 32, 128-byte rows: even and odd
// rows own the two halves of the 64 banks, and (r >> 1) & 7 spreads the 8 rows of a half over its 8 bank quads) or
// s ^ ((r >> 2) & 3) (BKC = 16, 64-byte rows) - applied on the SOURCE address of the LDS-DMA (its LDS side is lane-linear)
// and again on the fragment reads: conflict-free ds_read_b128 for both operands.
// Accumulators are row-per-lane (the weight fragment is the MFMA's A operand): lane (r, h) of wave (wm, wn) owns token
// m0 + 32 wm + r and, per 32-column block, the columns 8 q + 4 h + {0..3}.  Every row a tile reads or writes in its
// epilogue goes through a per-wave LDS slab so that a wave instruction covers whole row segments (epilogue_rows_h for the
// whole-row layers; the plain layers store fp32 rows or the H image of their output, split on the way out).
#pragma once
#include "kernels.hpp"

namespace pafuse {

// ---- H images --------------------------------------------------------------------------------------------------------
// activation rows (unit tests and bring-up; in the loop every producer writes its output in this form itself)
__global__ void __launch_bounds__(256) hsplit_rows_kernel(const float* X, uint8_t* out, int64_t R, int K) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one sub-block of 8 k per thread
    if (idx >= R * (K / 8)) return;
    const float* src = X + idx * 8;
    const f16x8x2 s = split2h(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4));
    *reinterpret_cast<f16x8*>(out + idx * 32) = s.hi;
    *reinterpret_cast<f16x8*>(out + idx * 32 + 16) = s.lo;
}

// weight image: the tensor's largest |W| is in the tail's scratch word (absmax_kernel); Ws = 2^k W, k = 14 - floor(log2(max))
__global__ void __launch_bounds__(256) hsplit_weights_kernel(const float* W, uint8_t* out, int N, int K) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)N * (K / 8)) return;
    uint8_t* const tail = out + (size_t)N * K * 4;   // HSPLIT_TAIL_BYTES behind the image
    const uint32_t mbits = reinterpret_cast<const uint32_t*>(tail)[1];
    int e = (int)(mbits >> 23) - 127;
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    const float mult = __builtin_bit_cast(float, (uint32_t)(127 + 14 - e) << 23);
    if (idx == 0) *reinterpret_cast<float*>(tail) = __builtin_bit_cast(float, (uint32_t)(127 - 14 + e) << 23);
    const float* src = W + idx * 8;
    f16x8 w0, w1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float x = src[i] * mult;
        const _Float16 h = (_Float16)x;
        w0[i] = h;
        w1[i] = (_Float16)(x - (float)h);
    }
    *reinterpret_cast<f16x8*>(out + idx * 32) = w0;
    *reinterpret_cast<f16x8*>(out + idx * 32 + 16) = w1;
}

// ---- whole-row epilogue of the H pipeline (inference): the chain of kernels.hpp's epilogue_row_per_lane<.., EPI_ROWLN> -
//   y = ws acc + bias + resid;  z = post_w ? LN(y; post) : y;  z += pos;  out_x = z (fp32) and its H image;
//   next: (mean, rstd) of z -> ln_stats (LayerNorm folded into the consumer), or LN(z; next) -> its H image, or the head -
// same arithmetic in the same order, but every global access of a tensor row goes THROUGH a per-wave LDS slab so that a
// wave instruction covers whole row segments.  With row-per-lane accumulators a direct dwordx4 access of lane (r, h) touches
// 32 different rows - 32 cache lines for 1 KiB - and the residual read, the x store and the H-image store of a 128 x 384 tile
// cost 30 000 line transactions: 62 000 cycles of epilogue against 35 000 of K loop (tools/hgemm_bench.hip stamps).  Through
// the slab a wave instruction moves 1 KiB in 8 lines (4 rows x 256 B).
// LDS (the dead ring): [7 BM WN floats: cross-wave partial sums][5 BN floats: per-column vectors][NW slabs of 32 x (32 NTH + 4)].
// NSLICE = 3: the X pipeline (xgemm.hpp) - images are X images (three bf16 slices, 6 bytes per element, exact), ws = 1
template <int WN, int NT, int BM, int NW, int NTH, bool HRES, int NSLICE>   // HRES: the residual is the centred image of x (p.resid_h); NSLICE defaults to 2 (kernels.hpp)
__device__ __forceinline__ void epilogue_rows_h(f32x16 (&acc)[NT], const GemmParams& p, const int64_t m0, const int n0, const int wm,
                                                const int wn, const int r, const int h, const int wave, const int lane, float* smem,
                                                const float ws) {
    static_assert(NSLICE == 2 || NSLICE == 3, "H image (two fp16 slices) or X image (three bf16 slices)");
    constexpr int SEG = (HRES && NSLICE == 3) ? 48 : 32;   // floats of a slab row per 32-column block: a row segment of the X image is 192 bytes
    constexpr int EB = (HRES && NSLICE == 3) ? 6 : 4;      // bytes per element of the residual's storage
    constexpr int BNV = WN * NT * 32, VEC = (7 * BM * WN + 3) / 4 * 4, SLAB0 = VEC + 5 * BNV, ST = SEG * NTH + 4;
    constexpr int NPASS = (NT + NTH - 1) / NTH;
    float* const red = smem;
    float* const slab = smem + SLAB0 + wave * 32 * ST;
    auto vec4 = [&](int slot, int n) -> f32x4 { return *reinterpret_cast<const f32x4*>(smem + VEC + slot * BNV + (n - n0)); };
    const int64_t mw = m0 + wm * 32;             // first row of this wave's strip
    const int64_t m = mw + r;
    const bool live = m < p.M;
    const int nb = n0 + wn * NT * 32 + 4 * h;     // + 32 nt + 8 q: this lane's columns
    const float invC = 1.0f / (float)p.N;
    // pass geometry: pass ps covers the column blocks nt0 = ps NTH .. of this wave's strip; a slab row segment holds 32 nth floats
    auto pass_cols = [&](int ps) { return NT - ps * NTH < NTH ? NT - ps * NTH : NTH; };
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
    auto layer_norm = [&](float eps, int slot, int vslot) {
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
                const f32x4 g4 = vec4(vslot, n), b4 = vec4(vslot + 1, n);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[nt][4 * q + e] = fmaf((acc[nt][4 * q + e] - mean) * rstd, g4[e], b4[e]);
            }
    };

    // ---- y = ws acc + bias + resid: the residual rows of a pass come in as whole row segments (lane l: float4 l % (8 nth) of
    // row l / (8 nth) of each group of 64 / (8 nth) rows), the next pass' loads fly while this pass is consumed
    constexpr int Q4 = SEG / 4;                  // float4 per row and 32-column block
    constexpr int LD = 32 * Q4 * NTH / 64;       // float4 per lane per (full) pass
    constexpr int NBUF = NT * 16 + 8 * LD <= 160 ? 2 : 1;   // the next pass in flight only where the registers allow it
    f32x4 rin[NBUF][LD];
    // (the residual is either fp32 rows or - p.resid_h - the H image of x: the same 4 bytes per element, so the same
    // row-segment loads; what differs is how a lane picks its four values out of the slab)
    const uint8_t* const resid_rows = HRES ? p.resid_h : reinterpret_cast<const uint8_t*>(p.resid);
    // The H image of x is CENTRED: it holds x - mean(row), so what the next GEMM multiplies has no common mode.  The mean
    // itself is carried nowhere: every reader of the residual stream is a LayerNorm (norm1, norm2, the block's post-norm, the
    // head's) or the residual add that feeds them, and LayerNorm does not see a row's mean - x and x - mean(x) give the same
    // network output.  So the residual here is the image as it is, the sum is re-centred on its own mean before it is stored,
    // and no number of the stream is ever rounded at the magnitude of an outlier mean (the fp32 reference rounds there).
    auto load_resid = [&](int ps, f32x4 (&dst)[LD]) {
        const int nth = pass_cols(ps), qpr = Q4 * nth;
        int psv = ps;
        asm volatile("" : "+s"(psv));   // (addresses formed in program order, as in store_rows below)
        const int ncol0 = n0 + (wn * NT + psv * NTH) * 32;
#pragma unroll
        for (int it = 0; it < LD; ++it) {
            const int idx = it * 64 + lane, row = idx / qpr, c4 = idx % qpr;
            const int64_t mr = mw + row < p.M ? mw + row : p.M - 1;
            if (idx < 32 * qpr) dst[it] = *reinterpret_cast<const f32x4*>(resid_rows + ((size_t)mr * p.N + ncol0) * EB + 16 * c4);
        }
    };
    load_resid(0, rin[0]);
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int nth = pass_cols(ps), qpr = Q4 * nth;
        if (NBUF == 2 && ps + 1 < NPASS) load_resid(ps + 1, rin[(ps + 1) % NBUF]);
#pragma unroll
        for (int it = 0; it < LD; ++it) {
            const int idx = it * 64 + lane, row = idx / qpr, c4 = idx % qpr;
            if (idx < 32 * qpr) *reinterpret_cast<f32x4*>(slab + row * ST + 4 * c4) = rin[ps % NBUF][it];
        }
        if (NBUF == 1 && ps + 1 < NPASS) load_resid(ps + 1, rin[0]);   // (in flight during this pass' arithmetic)
        __builtin_amdgcn_wave_barrier();   // the slab is this wave's own (DS operations of a wave complete in order)
#pragma unroll
        for (int j = 0; j < NTH; ++j)
            if (j < nth) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nt = ps * NTH + j;
                    const f32x4 b4 = vec4(0, nb + 32 * nt + 8 * q);
                    f32x4 r4;
                    if constexpr (HRES && NSLICE == 3) {   // chunk j of the slab row: slice planes at + 0, + 64, + 128 bytes; sub-block q, this lane's half 4 h
                        const uint8_t* sbk = reinterpret_cast<const uint8_t*>(slab + r * ST + SEG * j) + 16 * q + 8 * h;
                        r4 = xjoin4(*reinterpret_cast<const u32x2*>(sbk), *reinterpret_cast<const u32x2*>(sbk + 64),
                                    *reinterpret_cast<const u32x2*>(sbk + 128));   // exact: the fp32 number that was split
                    } else if constexpr (HRES) {   // sub-block (32 j + 8 q) / 8 of the slab row: hi at + 0, lo at + 16 bytes; this lane's half 4 h
                        const uint8_t* sbk = reinterpret_cast<const uint8_t*>(slab + r * ST + 32 * j + 8 * q);
                        typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
                        const f16x4_t hi = *reinterpret_cast<const f16x4_t*>(sbk + 8 * h), lo = *reinterpret_cast<const f16x4_t*>(sbk + 16 + 8 * h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) r4[e] = fmaf((float)lo[e], 0.00048828125f, (float)hi[e]);   // hi + 2^-11 lo: exact
                    } else {
                        r4 = *reinterpret_cast<const f32x4*>(slab + r * ST + 32 * j + 8 * q + 4 * h);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[nt][4 * q + e] = fmaf(acc[nt][4 * q + e], ws, b4[e]) + r4[e];
                }
            }
        __builtin_amdgcn_wave_barrier();
    }
    if (p.post_w) layer_norm(p.post_eps, 0, 1);
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
    // a pass of the accumulators out through the slab: fp32 rows to `dst32` and / or their H image to `dsth` (whole sub-blocks
    // of 8 columns per lane: 32 contiguous bytes of either)
    float* const rowmean = smem + 6 * BM * WN + wave * 32;     // (slot 6 of the reduction scratch is unused by the chain: per-wave row means)
    auto store_rows = [&](float* dst32, uint8_t* dsth, bool centred) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int nth = pass_cols(ps), spr = 4 * nth;
            int psv = ps;
            asm volatile("" : "+s"(psv));   // the pass' addresses are formed HERE, in program order: hoisted out of the unrolled loop
                                            // (seven passes at NT = 7) they cost 116 bytes of scratch per lane
            const int ncol0 = n0 + (wn * NT + psv * NTH) * 32;
#pragma unroll
            for (int j = 0; j < NTH; ++j)
                if (j < nth) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[ps * NTH + j][4 * q + e];
                        *reinterpret_cast<f32x4*>(slab + r * ST + 32 * j + 8 * q + 4 * h) = v;
                    }
                }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < (32 * 4 * NTH + 63) / 64; ++it) {
                const int idx = it * 64 + lane, row = idx / spr, sb = idx % spr;
                if (idx < 32 * spr && mw + row < p.M) {
                    const f32x4 lo4 = *reinterpret_cast<const f32x4*>(slab + row * ST + 8 * sb);
                    const f32x4 hi4 = *reinterpret_cast<const f32x4*>(slab + row * ST + 8 * sb + 4);
                    const size_t at = (size_t)(mw + row) * p.N + ncol0 + 8 * sb;
                    f32x4 c0 = lo4, c1 = hi4;
                    if (centred) {   // the stream is stored centred on the row mean, in whichever form it is kept (fp32 rows, image, both)
                        const float mu = rowmean[row];
#pragma unroll
                        for (int e = 0; e < 4; ++e) c0[e] -= mu, c1[e] -= mu;
                    }
                    if (dst32) {
                        *reinterpret_cast<f32x4*>(dst32 + at) = c0;
                        *reinterpret_cast<f32x4*>(dst32 + at + 4) = c1;
                    }
                    if (dsth) {
                        if constexpr (NSLICE == 3) {
                            xsplit_store8(dsth + (size_t)(mw + row) * p.N * 6, ncol0 + 8 * sb, c0, c1);
                        } else {
                            const f16x8x2 sp = split2h(c0, c1);
                            *reinterpret_cast<f16x8*>(dsth + at * 4) = sp.hi;
                            *reinterpret_cast<f16x8*>(dsth + at * 4 + 16) = sp.lo;
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    };
    // Up to two row stores: (0) x (fp32 and / or its H image - centred on the row mean when the next LayerNorm is folded into
    // its consumer, whose statistics are formed first), (1) without the fold, the next LayerNorm's output.  ONE copy of the
    // store code (a loop the compiler may not unroll): three inlined copies cost 170 - 380 bytes of scratch per lane.
    const bool folded = p.next_w && p.ln_stats;
#pragma clang loop unroll(disable)
    for (int ph = 0; ph < 2; ++ph) {
        float* d32;
        uint8_t* dh;
        bool centred = false;
        if (ph == 0) {
            if (folded) {   // the row's statistics only (the same two fixed-order reductions layer_norm makes)
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
                if (h == 0) rowmean[r] = mean;      // (every wave of the row knows the mean: its own copy, wave-local)
                __builtin_amdgcn_wave_barrier();
                centred = true;
            }
            d32 = p.out_x, dh = p.out_xh;
        } else {
            if (!p.next_w || folded) break;
            layer_norm(p.next_eps, 2, 3);
            d32 = p.out_n, dh = p.out_nh;
        }
        if (d32 || dh) store_rows(d32, dh, centred);
    }
    if (p.next_w && !folded) {
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
}

// ---- the tile -----------------------------------------------------------------------------------------------------------
template <int WM, int WN, int NT, int BKC>
struct HTile {
    static_assert(BKC == 32 || BKC == 16, "chunk depth");
    static constexpr int NW = WM * WN, NTHR = NW * 64;
    static constexpr int BM = WM * 32, BN = WN * NT * 32;
    static constexpr int ROWB = BKC * 4;                      // bytes per row of a stage (both operands)
    static constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + W_BYTES;
    static constexpr int IA = A_BYTES / 1024, IW = W_BYTES / 1024;    // DMA wave-instructions per chunk
    static constexpr int CNT = (IA + IW + NW - 1) / NW;               // per wave (surplus slots re-issue the last piece)
    static_assert(A_BYTES % 1024 == 0 && W_BYTES % 1024 == 0, "whole DMA pieces");
    // the source swizzle of a lane must not depend on the instruction index: with 128-byte rows an instruction covers 8 rows
    // and (row >> 1) & 7 takes the instruction's parity, which is the wave's when NW and IA are even
    static_assert(BKC == 16 || (NW % 2 == 0 && IA % 2 == 0), "instruction parity = wave parity");
};

template <int WM, int WN, int NT, int EPI, int NSTAGE, int BKC, bool HRES = false>
__device__ __forceinline__ void hgemm_tile(const GemmParams& p, const int b, const int nb, float* smem) {
    using T = HTile<WM, WN, NT, BKC>;
    constexpr int NW = T::NW, BM = T::BM, BN = T::BN, CNT = T::CNT, IA = T::IA, IW = T::IW, ROWB = T::ROWB;
    constexpr int NS2 = BKC / 16;            // 16-deep MFMA steps per chunk
    constexpr int RPI = 1024 / ROWB;         // rows per DMA instruction (8 / 16)
    constexpr int CPR = ROWB / 16;           // 16-byte slots per row (8 / 4)
    static_assert(NSTAGE >= 2 && NSTAGE <= 4 && CNT * (NSTAGE - 1) < 64, "ring depth / vmcnt range");
    uint8_t* const lds = reinterpret_cast<uint8_t*>(smem);
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_n = p.N / BN;
    int tile;
    {   // XCD-aware tile order (speed only): workgroups b and b + 8 share an XCD, each XCD gets a contiguous run of tiles
        const int xcd = b & 7, q = nb >> 3, rem = nb & 7;
        tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (b >> 3);
    }
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int64_t m0 = (int64_t)tile_m * BM;
    if (m0 >= p.M) return;
    PAFUSE_STAMP(0);
    const int n0 = tile_n * BN;
    const int K = p.K, nk = K / BKC;

    // ---- DMA sources.  Instruction i of a chunk (0 .. IA + IW - 1) belongs to wave i % NW; A instruction ia covers rows
    // RPI ia .. + RPI - 1 (lane l: row RPI ia + l / CPR, LDS position l % CPR), W instruction iw rows RPI iw .. of the tile's
    // weight rows.  The lane reads the slot that belongs at its position: position ^ swizzle(row).
    const uint8_t* const Abase = p.Ah + (size_t)m0 * K * 4;
    const uint8_t* const Wbase = p.Wh + (size_t)n0 * K * 4;
    int src_off[CNT];   // byte offset of this lane's source inside the tile's rows of its operand, chunk 0
#pragma unroll
    for (int j = 0; j < CNT; ++j) {
        int i = wave + j * NW;
        i = i < IA + IW ? i : IA + IW - 1;
        const bool is_a = i < IA;
        const int row = RPI * (is_a ? i : i - IA) + lane / CPR;       // row inside the tile's operand rows
        const int sw = BKC == 32 ? (row >> 1) & 7 : (row >> 2) & 3;
        const int slot = (lane % CPR) ^ sw;
        int64_t grow = row;
        if (is_a) {   // tail rows of A read a valid row (never stored)
            const int64_t lim = p.M - 1 - m0;
            grow = row < lim ? row : lim;
        }
        src_off[j] = (int)(grow * K * 4) + slot * 16;
    }
    auto issue_piece = [&](int kc, int st, int j) {
        uint8_t* const sa = lds + st * T::STAGE_BYTES;
        int i = wave + j * NW;  // wave-uniform
        i = i < IA + IW ? i : IA + IW - 1;
        const uint8_t* src = (i < IA ? Abase : Wbase) + (size_t)kc * ROWB + src_off[j];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(sa + i * 1024), 16, 0, 0);
    };

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;

    // fragment addresses inside a stage: row base + the swizzled position of slot 2 (sub-block) + slice
    const int sw = BKC == 32 ? (r >> 1) & 7 : (r >> 2) & 3;
    uint32_t pos[NS2][2];
#pragma unroll
    for (int s2 = 0; s2 < NS2; ++s2)
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) pos[s2][sl] = (uint32_t)(((2 * (2 * s2 + h) + sl) ^ sw) & (CPR - 1)) * 16;
    const uint32_t a_row = (uint32_t)((wm * 32 + r) * ROWB);
    const uint32_t w_row = (uint32_t)(T::A_BYTES + (wn * NT * 32 + r) * ROWB);

#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < nk) {
#pragma unroll
            for (int j = 0; j < CNT; ++j) issue_piece(s, s, j);
        }

#define PAFUSE_PIN_ACC(A) asm volatile("" : "+v"(A))
    for (int kc = 0; kc < nk; ++kc) {
        if (kc + NSTAGE - 2 < nk)
            wait_vmcnt<CNT*(NSTAGE - 2)>();   // chunk kc of this wave has landed (the younger chunks may still fly)
        else
            wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();         // ... of every wave; and every wave is done reading chunk kc - 1
        if (kc == 0) { PAFUSE_STAMP(3); }
        const bool refill = kc + NSTAGE - 1 < nk;
        const int kn = kc + NSTAGE - 1, stn = kn % NSTAGE;
        const uint32_t sbase = lds0 + (uint32_t)((kc % NSTAGE) * T::STAGE_BYTES);
#ifdef PAFUSE_HGEMM_BURST   // A/B build (tools/hgemm_bench.hip): the whole refill at the top of the chunk instead of piece by piece
        if (refill) {
#pragma unroll
            for (int j = 0; j < CNT; ++j) issue_piece(kn, stn, j);
        }
#endif
        __builtin_amdgcn_s_setprio(1);
        constexpr int NG = NS2 * NT;          // groups (s2, nt) of three MFMAs on one accumulator
        u32x4 af[2][2], wf[2][2];             // [buffer][slice]: A fragment per 16-deep step, W' fragment per group
        auto read_a = [&](auto S2) {
            constexpr int s2 = decltype(S2)::value;
            af[s2 & 1][0] = lds_read128<0>(sbase + a_row + pos[s2][0]);
            af[s2 & 1][1] = lds_read128<0>(sbase + a_row + pos[s2][1]);
        };
        auto read_w = [&](auto G) {
            constexpr int g = decltype(G)::value, s2 = g / NT, off = (g % NT) * 32 * ROWB;
            wf[g & 1][0] = lds_read128<off>(sbase + w_row + pos[s2][0]);
            wf[g & 1][1] = lds_read128<off>(sbase + w_row + pos[s2][1]);
        };
        read_a(std::integral_constant<int, 0>{});
        read_w(std::integral_constant<int, 0>{});
        static_for<NG>([&](auto G) {
            constexpr int g = decltype(G)::value, s2 = g / NT, nt = g % NT;
            // what the NEXT group needs is issued first and stays in flight during this group's MFMAs
            constexpr bool next_a = g + 1 < NG && (g + 1) % NT == 0;
            constexpr int flying = g + 1 < NG ? (next_a ? 4 : 2) : 0;
            if constexpr (next_a) read_a(std::integral_constant<int, (g + 1) / NT>{});
            if constexpr (g + 1 < NG) read_w(std::integral_constant<int, g + 1>{});
            if constexpr (flying == 4)
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af[s2 & 1][0]), "+v"(af[s2 & 1][1]), "+v"(wf[g & 1][0]), "+v"(wf[g & 1][1]));
            else if constexpr (flying == 2)
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(af[s2 & 1][0]), "+v"(af[s2 & 1][1]), "+v"(wf[g & 1][0]), "+v"(wf[g & 1][1]));
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[s2 & 1][0]), "+v"(af[s2 & 1][1]), "+v"(wf[g & 1][0]), "+v"(wf[g & 1][1]));
            const f16x8 w0 = __builtin_bit_cast(f16x8, wf[g & 1][0]), w1 = __builtin_bit_cast(f16x8, wf[g & 1][1]);
            const f16x8 w2 = w0 * (_Float16)0.00048828125f;   // 2^-11: four v_pk_mul_f16 (RNE into the subnormals)
            const f16x8 a_hi = __builtin_bit_cast(f16x8, af[s2 & 1][0]), a_lo = __builtin_bit_cast(f16x8, af[s2 & 1][1]);
            PAFUSE_PIN_ACC(acc[nt]);
            acc[nt] = mfma_f16_k16(w2, a_lo, acc[nt]);   // small terms first, the leading product last
            {   // this group's share of the refill DMA, in the shadow of the MFMA just issued
                constexpr int PER = (CNT + NG - 1) / NG, j0 = g * PER, j1 = (g + 1) * PER < CNT ? (g + 1) * PER : CNT;
#ifndef PAFUSE_HGEMM_BURST
                if constexpr (j0 < j1) {
                    asm volatile("" ::: "memory");
                    if (refill) {
#pragma unroll
                        for (int j = j0; j < j1; ++j) issue_piece(kn, stn, j);
                    }
                    asm volatile("" ::: "memory");
                }
#endif
            }
            PAFUSE_PIN_ACC(acc[nt]);
            acc[nt] = mfma_f16_k16(w1, a_hi, acc[nt]);
            PAFUSE_PIN_ACC(acc[nt]);
            acc[nt] = mfma_f16_k16(w0, a_hi, acc[nt]);
        });
        __builtin_amdgcn_s_setprio(0);
    }
#undef PAFUSE_PIN_ACC
    PAFUSE_STAMP(1);
    __syncthreads();  // the ring becomes the epilogue's scratch
    const float ws = *reinterpret_cast<const float*>(p.Wh + (size_t)p.N * K * 4);   // 2^-k of the weight image (exact)

    if constexpr (EPI != EPI_BIAS) {
        static_assert(EPI == EPI_ROWLN, "the H pipeline is inference only");
        constexpr int VEC = (7 * BM * WN + 3) / 4 * 4;  // behind the cross-wave reduction slots
        constexpr size_t RINGF = (size_t)NSTAGE * T::STAGE_BYTES / sizeof(float);
        constexpr auto need = [](int nth) { return (size_t)VEC + 5 * BN + (size_t)NW * 32 * (32 * nth + 4); };
        // column blocks per pass of the row I/O: a 32-column segment is a whole 128-byte line already, two halve the passes
        constexpr int NTH = (NT % 2 == 0 && need(2) <= RINGF) ? 2 : 1;
        static_assert(need(NTH) <= RINGF, "epilogue scratch must fit the ring");
        const float* const src[5] = {p.bias, p.post_w, p.post_b, p.next_w, p.next_b};
#pragma unroll
        for (int v = 0; v < 5; ++v)
            if (src[v])  // workgroup-uniform
                for (int i = tid; i < BN / 4; i += T::NTHR)
                    *reinterpret_cast<f32x4*>(smem + VEC + v * BN + 4 * i) = *reinterpret_cast<const f32x4*>(src[v] + n0 + 4 * i);
        __syncthreads();
        epilogue_rows_h<WN, NT, BM, NW, NTH, HRES>(acc, p, m0, n0, wm, wn, r, h, wave, lane, smem, ws);
        return;
    } else {
        // ---- plain layers: out = act(ws acc + bias), or the folded LayerNorm  act(rstd (ws acc - mean ls) + lt); every wave
        // transposes its 32 x (32 NT) strip through its own LDS slab and stores whole row segments - fp32, or the H image of
        // the output (p.out_h: the split is done here, once, for every consumer tile)
        // (NTH column blocks of the strip per pass: the slabs of all waves share the ring's bytes)
        constexpr size_t RING = (size_t)NSTAGE * T::STAGE_BYTES;
        constexpr auto slab_bytes = [](int nth) { return (size_t)NW * 32 * (32 * nth + 4) * sizeof(float); };
        constexpr int NTH = slab_bytes(NT) <= RING ? NT : (NT > 4 && slab_bytes(4) <= RING ? 4 : (slab_bytes(2) <= RING ? 2 : 1));
        static_assert(slab_bytes(NTH) <= RING, "epilogue slabs must fit the ring");
        constexpr int ST = 32 * NTH + 4;    // slab row stride (floats): + 4 keeps 16-byte alignment and shifts the banks per row
        float* const slab = smem + wave * 32 * ST;
        const int64_t mw = m0 + wm * 32;
        const int64_t m = mw + r;
        const int64_t mm = m < p.M ? m : p.M - 1;
        float rstd = 1.0f, nmr = 0.0f;
        if (p.ln_in) {
            const float mean = p.ln_in[2 * mm];
            rstd = p.ln_in[2 * mm + 1];
            nmr = -mean * rstd;
        }
        rstd *= ws;
#pragma unroll
        for (int nt0 = 0; nt0 < NT; nt0 += NTH) {
            const int nth = NT - nt0 < NTH ? NT - nt0 : NTH;          // compile-time after unrolling
            const int ncol0 = n0 + (wn * NT + nt0) * 32;              // first column of this pass
#pragma unroll
            for (int j = 0; j < NTH; ++j) {
                if (j < nth) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = ncol0 + 32 * j + 8 * q + 4 * h;
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + n);
                        f32x4 v;
                        if (p.ln_in && p.ln_s) {
                            const f32x4 s4 = *reinterpret_cast<const f32x4*>(p.ln_s + n);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaf(rstd, acc[nt0 + j][4 * q + e], fmaf(nmr, s4[e], b4[e]));
                        } else if (p.ln_in) {   // A is the CENTRED image (x - mean): LN(x) W^T + b = rstd acc + lt, no mean term
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaf(rstd, acc[nt0 + j][4 * q + e], b4[e]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[nt0 + j][4 * q + e], ws, b4[e]);
                        }
                        if (p.act) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                        }
                        *reinterpret_cast<f32x4*>(slab + r * ST + 32 * j + 8 * q + 4 * h) = v;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();   // the slab is this wave's own (DS operations of a wave complete in order): no barrier
            if (p.out_h) {
                // one sub-block of 8 columns per lane: 32 contiguous bytes (hi | lo) of the H image
                const int spr = 4 * nth;                            // sub-blocks per row of this pass
#pragma unroll
                for (int it = 0; it < (32 * 4 * NTH + 63) / 64; ++it) {
                    const int idx = it * 64 + lane;
                    const int row = idx / spr, sb = idx % spr;
                    if (idx < 32 * spr && mw + row < p.M) {
                        const f32x4 lo4 = *reinterpret_cast<const f32x4*>(slab + row * ST + 8 * sb);
                        const f32x4 hi4 = *reinterpret_cast<const f32x4*>(slab + row * ST + 8 * sb + 4);
                        const f16x8x2 sp = split2h(lo4, hi4);
                        uint8_t* dst = p.out_h + ((size_t)(mw + row) * p.N + ncol0 + 8 * sb) * 4;
                        *reinterpret_cast<f16x8*>(dst) = sp.hi;
                        *reinterpret_cast<f16x8*>(dst + 16) = sp.lo;
                    }
                }
            } else {
                const int qpr = 8 * nth;                            // float4 per row of this pass
#pragma unroll
                for (int it = 0; it < (32 * 8 * NTH + 63) / 64; ++it) {
                    const int idx = it * 64 + lane;
                    const int row = idx / qpr, c4 = idx % qpr;
                    if (idx < 32 * qpr && mw + row < p.M)
                        *reinterpret_cast<f32x4*>(p.out + (size_t)(mw + row) * p.N + ncol0 + 4 * c4) =
                            *reinterpret_cast<const f32x4*>(slab + row * ST + 4 * c4);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        PAFUSE_STAMP(2);
    }
}

template <int WM, int WN, int NT, int EPI, int NSTAGE, int BKC, int MINW, bool HRES = false>
__global__ void __launch_bounds__(WM* WN * 64, MINW) hgemm_kernel(const GemmParams p) {
    PAFUSE_XQ_GUARD();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    hgemm_tile<WM, WN, NT, EPI, NSTAGE, BKC, HRES>(p, blockIdx.x, gridDim.x, smem);
}


// ----------------------------------------------------------------------------------------------------------------
// qkv projection + attention of one head in ONE kernel, H pipeline (round 4):
//   o[rows of the tile, head] = softmax(q k^T * scale) v,   (q | k | v) = A[rows] @ W_head^T + b_head   (+ folded LayerNorm)
// replaces a qkv hgemm launch + an attn_kernel launch (common/mixste.py:65-79) and the [M,3C] fp32 tensor between them.
// In the f16x2 regime the loop is bound by the bytes it moves (the split-precision GEMMs run at three MFMAs per product):
// writing q, k, v (3 C floats per token) and reading them back is 6 of the 20 row-passes a block makes over HBM.  Here the
// tile's q, k, v never leave the CU.
// Workgroup = 4 waves = one tile of whole sequences x one head (kernels.hpp fqa_kernel's decomposition): the tile holds
// nseq_tile sequences, token i of sequence s at tile row s L + i (rows past the last sequence alias a valid token and are
// dropped).
//   phase 1  the projection on v_mfma_f32_16x16x32_f16 (three products per k): LDS-DMA ring of 32-deep chunks, A = the
//            gathered rows of the H image of x (temporal blocks: rows J apart), W = the head's 3 DP rows of the HEAD-MAJOR H
//            image (q, k, v of the head, each zero-padded from d to DP rows; bias / ls / lt in that order).  16-byte slot s
//            of stage row r sits at s ^ (2 ((r >> 1) & 3) ^ ((r >> 3) & 1)): conflict-free ds_read_b128 for the 16-row fragments
//            (lane (c, qd): row c, sub-block qd).  The token sits on the lane: lane (c, qd) ends with token c's outputs
//            n = 16 nb + 4 qd + {0..3}.
//   phase 2  accumulators (x 2^-k, bias or folded LayerNorm) -> three [rows][DP + 4] fp32 LDS tiles over the dead ring;
//   phase 3  attn_kernel's arithmetic per (sequence, 16-query tile) from those tiles; o is written once, as the H image the
//            proj hgemm reads.
// Blocks: b -> XCD x = b & 7, (tile, head) = ((b >> 3) / heads * 8 + x, (b >> 3) % heads): the eight heads of a tile run on
// one XCD (its A rows are L2 hits for seven of them).
// ----------------------------------------------------------------------------------------------------------------
template <int LP, int DP, int HPW = 1>   // HPW: heads per workgroup (their projections share the A stream, their attention phases run in turn)
struct HfqaTile {
    // waves = 32-row strips of the tile: 4 (128 rows); the 80-token form (68 joints of the face: two sequences = 136 rows) takes 5
    static constexpr int NWV = LP == 80 ? 5 : 4, TROWS = 32 * NWV, NTHR = 64 * NWV;
    static constexpr int NBH = 3 * DP / 16, NB = HPW * NBH, LDV = DP + 4, ROWS = TROWS + (LP == 48 ? 4 : 0), NSTAGE = 2;
    static constexpr int A_BYTES = TROWS * 128, W_BYTES = HPW * 3 * DP * 128, STAGE_BYTES = A_BYTES + W_BYTES;
    static constexpr int IA = A_BYTES / 1024, IW = W_BYTES / 1024, CNT = (IA + IW + NWV - 1) / NWV;
    static constexpr int QKV_BYTES = 3 * ROWS * LDV * 4;
    static constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES > QKV_BYTES ? NSTAGE * STAGE_BYTES : QKV_BYTES;
    static_assert(W_BYTES % 1024 == 0, "whole DMA pieces");
};
__device__ __forceinline__ int hfqa_swizzle(int row) { return (((row >> 1) & 3) << 1) ^ ((row >> 3) & 1); }

// phase 3 of the fused qkv + attention kernels (hfqa_kernel here, xfqa_kernel in xgemm.hpp): attention per (sequence of the tile,
// 16-query tile) from the q | k | v tiles in LDS ([ROWS][DP + 4] fp32 each, token i of tile sequence s at row s L + i) -
// attn_kernel's arithmetic per item on v_mfma_f32_16x16x4_f32 - and o written once, as the image the proj GEMM reads
// (NSLICE = 2: H image, 3: X image).  Lane (c, qd) = (lane & 15, lane >> 4).
template <int LP, int DP, int NWV, int ROWS, int NSLICE, class TokenOf>
__device__ __forceinline__ void fqa_attention_from_lds(const FqaParams& fp, float* smem, const int64_t seq0, const int head, const int wave,
                                                       const int c, const int qd, TokenOf token_of) {
    constexpr int LDV = DP + 4;
    const int L = fp.L, NSEQ = fp.nseq_tile;
    float* const Qs = smem;                    // [ROWS][LDV] each
    float* const Ks = Qs + ROWS * LDV;
    float* const Vs = Ks + ROWS * LDV;
    // attn_kernel's arithmetic per item, TWO items of a wave in
    // flight together: an item is one dependent chain (scores -> row maximum across the lane groups -> exponentials -> row sum ->
    // weights -> P V) and a wave that walks it alone waits for every cross-lane exchange and every MFMA result; the second
    // item's instructions fill those waits (same results: the items do not interact).
#ifndef PAFUSE_HFQA_ITEMS
#define PAFUSE_HFQA_ITEMS 1
#endif
    constexpr int QT = LP / 16, KT = LP / 16, CT = DP / 16, SD = DP / 16, U = PAFUSE_HFQA_ITEMS;
    const int l15 = c, g4 = qd;
    const int nseq_here = (int)((fp.nseq - seq0) < NSEQ ? (fp.nseq - seq0) : NSEQ);
    const int n_items = nseq_here * QT;
    for (int base = wave; base < n_items; base += U * NWV) {
        int rb[U], qt[U];
        bool valid[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int item = base + u * NWV;
            const int it = item < n_items ? item : base;       // an absent second item repeats the first (never stored)
            const int sl = it / QT;
            qt[u] = it - sl * QT, rb[u] = sl * L;
            valid[u] = item < n_items && qt[u] * 16 < L;
        }
        if (!valid[0] && !valid[U - 1]) continue;              // (wave-uniform)
        f32x4 qf[U][SD];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int sd = 0; sd < SD; ++sd) qf[u][sd] = *reinterpret_cast<const f32x4*>(Qs + (rb[u] + qt[u] * 16 + l15) * LDV + 16 * sd + 4 * g4);
        f32x4 sc[U][KT];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) sc[u][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sd = 0; sd < SD; ++sd) {
            f32x4 kf[U][KT];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) kf[u][kt] = *reinterpret_cast<const f32x4*>(Ks + (rb[u] + kt * 16 + l15) * LDV + 16 * sd + 4 * g4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        sc[u][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[u][kt][j], qf[u][sd][j], sc[u][kt], 0, 0, 0);
        }
        float mx[U], sum[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            mx[u] = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int key = kt * 16 + 4 * g4 + reg;
                    const float v = key < L ? sc[u][kt][reg] * fp.scale : -INFINITY;
                    sc[u][kt][reg] = v;
                    mx[u] = fmaxf(mx[u], v);
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) mx[u] = fmaxf(mx[u], __shfl_xor(mx[u], 16));
#pragma unroll
        for (int u = 0; u < U; ++u) mx[u] = fmaxf(mx[u], __shfl_xor(mx[u], 32));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            sum[u] = 0.f;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const float e = __builtin_amdgcn_exp2f((sc[u][kt][reg] - mx[u]) * 1.44269504088896340736f);
                    sc[u][kt][reg] = e;
                    sum[u] += e;
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) sum[u] += __shfl_xor(sum[u], 16);
#pragma unroll
        for (int u = 0; u < U; ++u) sum[u] += __shfl_xor(sum[u], 32);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float inv = 1.0f / sum[u];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) sc[u][kt][reg] *= inv;
        }
        f32x4 oc[U][CT];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) oc[u][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    // (a masked key - beyond the sequence's L tokens - carries weight 0, but its row of the tile belongs to the NEXT
                    // sequence: 0 x NaN would hand a neighbour's NaN to this sequence, so masked keys read the sequence's own last row)
                    const int key = kt * 16 + 4 * g4 + reg;
                    const float* vrow = Vs + (rb[u] + (key < L ? key : L - 1)) * LDV + l15;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        oc[u][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[ct * 16], sc[u][kt][reg], oc[u][ct], 0, 0, 0);
                }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = qt[u] * 16 + l15;
            if (valid[u] && q < L) {   // channel ch = head d + 16 ct + 4 g4 (a multiple of 4): sub-block ch / 8, its second half when ch & 4
                uint8_t* const hrow = reinterpret_cast<uint8_t*>(fp.o) + (size_t)token_of(rb[u] + q) * fp.C * (NSLICE == 3 ? 6 : 4);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    if (ct * 16 + 4 * g4 < fp.d) {
                        const int ch = head * fp.d + ct * 16 + 4 * g4;
                        if constexpr (NSLICE == 3) xsplit_store4(hrow, ch, oc[u][ct]);
                        else hsplit_store4(hrow + (ch >> 3) * 32, ch & 4, oc[u][ct]);
                    }
            }
        }
    }
}

// phases 2 and 3 of the fused kernel, shared by its two projection forms: the accumulators (token on the lane: lane (c, qd) of
// wave w holds, for row blocks g = 0, 1, token 32 w + 16 g + c's outputs n = 16 nb + 4 qd + {0..3}) -> q | k | v tiles in LDS ->
// attention per (sequence, 16-query tile) -> o as the H image.  The caller has passed a workgroup barrier behind its last use
// of the LDS.
template <int LP, int DP, class TokenOf>
__device__ __forceinline__ void hfqa_attention_phases(const FqaParams& fp, f32x4 (&acc)[2][3 * DP / 16], float* smem, const int64_t seq0,
                                                      const int head, const int n0, const int wave, const int c, const int qd,
                                                      const int tid, TokenOf token_of, const f32x4 (&bias4)[3 * DP / 16], const float (&row_rstd)[2],
                                                      const float (&row_nmr)[2], const float ws, const int hh = 0) {
    using FT = HfqaTile<LP, DP>;
    constexpr int NB = FT::NBH, LDV = FT::LDV, ROWS = FT::ROWS, NWV = FT::NWV, TROWS = FT::TROWS;
    const GemmParams& p = fp.g;
    const int L = fp.L, NSEQ = fp.nseq_tile, K = p.K;
    // ---- phase 2: q | k | v of the tile's tokens to LDS (2^-k, bias or the folded LayerNorm applied)
    float* const Qs = smem;                    // [ROWS][LDV] each
    float* const Ks = Qs + ROWS * LDV;
    float* const Vs = Ks + ROWS * LDV;
    // (2^-k of the image, the rows' LayerNorm factors and this head's bias were fetched before / during the projection: a
    // global load issued here costs the workgroup its whole latency, 2 - 4 thousand cycles under the other workgroups' streams)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int r = 32 * wave + 16 * g + c;
        const float rstd = row_rstd[g] * ws, nmr = row_nmr[g];
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int col = 16 * n + 4 * qd;   // 0 .. 3 DP - 1: part = col / DP (a 16-column block never straddles parts)
            const f32x4 b4 = bias4[n];
            f32x4 v;
            if (p.ln_in && p.ln_s) {
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(p.ln_s + n0 + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(rstd, acc[g][n][e], fmaf(nmr, s4[e], b4[e]));
            } else if (p.ln_in) {   // centred A: no mean term
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(rstd, acc[g][n][e], b4[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[g][n][e], ws, b4[e]);
            }
            const int part = (16 * n) / DP, cc = col - part * DP;
            *reinterpret_cast<f32x4*>(Qs + part * ROWS * LDV + r * LDV + cc) = v;
        }
    }
    if (ROWS > TROWS)   // LP = 48: the key / value tiles of the last sequence reach 4 rows past the tile - keep them finite
        for (int i = tid; i < 3 * (ROWS - TROWS) * LDV; i += FT::NTHR) {
            const int part = i / ((ROWS - TROWS) * LDV), rem = i % ((ROWS - TROWS) * LDV);
            Qs[part * ROWS * LDV + TROWS * LDV + rem] = 0.f;
        }
    __syncthreads();
#if PAFUSE_STAMP_SLOTS >= 8
    PAFUSE_STAMP(4 + 2 * hh);   // diagnostic builds: q | k | v of this head are in LDS
#endif

    // ---- phase 3
    fqa_attention_from_lds<LP, DP, NWV, ROWS, 2>(fp, smem, seq0, head, wave, c, qd, token_of);
#if PAFUSE_STAMP_SLOTS >= 8
    PAFUSE_STAMP(5 + 2 * hh);   // diagnostic builds: this head's items are done
#endif
}

template <int LP, int DP, int HPW>
// (two workgroups per CU; the five-wave form asks for the register budget of three)
__global__ void __launch_bounds__((HfqaTile<LP, DP, HPW>::NTHR), (HfqaTile<LP, DP, HPW>::NWV == 5 ? 3 : 2)) hfqa_kernel(const FqaParams fp) {
    PAFUSE_XQ_GUARD();
    using FT = HfqaTile<LP, DP, HPW>;
    constexpr int NB = FT::NB, NBH = FT::NBH, LDV = FT::LDV, ROWS = FT::ROWS, IA = FT::IA, IW = FT::IW, CNT = FT::CNT, NSTAGE = FT::NSTAGE;
    constexpr int NWV = FT::NWV;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const GemmParams& p = fp.g;
    uint8_t* const lds = reinterpret_cast<uint8_t*>(smem);
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, qd = lane >> 4;
    const int L = fp.L, NSEQ = fp.nseq_tile;
    const int64_t ntiles = (fp.nseq + NSEQ - 1) / NSEQ;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int hgroups = fp.heads / HPW;                      // workgroups per tile
    const int64_t tile = (int64_t)(idx / hgroups) * 8 + xcd;
    const int head0 = (idx % hgroups) * HPW;                 // this workgroup's heads: head0 .. head0 + HPW - 1
    if (tile >= ntiles) return;   // (workgroup-uniform: the grid is padded to a multiple of 8 tiles)
    PAFUSE_STAMP(0);
    const int K = p.K, nk = K / 32;
    const int64_t seq0 = tile * NSEQ;
    const int64_t last_seq = fp.nseq - 1;
    // token (row of A / o) of tile row r: sequence seq0 + r / L, position r % L; rows of absent sequences alias the last one
    // (32-bit arithmetic: sequence counts and token indices of the hot path fit - the launcher checks - and a 64-bit division
    // costs a wave hundreds of instructions; a lane calls this eight times)
    const uint32_t grp = (uint32_t)fp.group, grp_stride = (uint32_t)fp.group_stride, sq_stride = (uint32_t)fp.seq_stride, tk_stride = (uint32_t)fp.tok_stride;
    auto token_of = [&](int r) -> int64_t {
        int sl = r / L, t = r - sl * L;
        if (sl >= NSEQ) sl = NSEQ - 1, t = L - 1;
        int64_t sq64 = seq0 + sl;
        if (sq64 > last_seq) sq64 = last_seq;
        const uint32_t sq = (uint32_t)sq64, gi = sq / grp;
        return (int64_t)(gi * grp_stride + (sq - gi * grp) * sq_stride + (uint32_t)t * tk_stride);
    };

    // ---- phase 1: the projection.  DMA instruction i of a chunk (0 .. IA + IW - 1) belongs to wave i % 4; A instruction ia
    // covers tile rows 8 ia .. + 7 (lane l: row 8 ia + l / 8, LDS position l % 8), W instruction iw rows 8 iw .. of the head.
    const int n0 = head0 * 3 * DP;
    const uint8_t* src[CNT];
#pragma unroll
    for (int j = 0; j < CNT; ++j) {
        int i = wave + NWV * j;
        i = i < IA + IW ? i : IA + IW - 1;
        const bool is_a = i < IA;
        const int row = 8 * (is_a ? i : i - IA) + (lane >> 3);
        const int slot = (lane & 7) ^ hfqa_swizzle(row);
        src[j] = (is_a ? p.Ah + (size_t)token_of(row) * K * 4 : p.Wh + (size_t)(n0 + row) * K * 4) + slot * 16;
    }
    auto issue_piece = [&](int kc, int st, int j) {
        uint8_t* const sa = lds + st * FT::STAGE_BYTES;
        int i = wave + NWV * j;  // wave-uniform
        i = i < IA + IW ? i : IA + IW - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (size_t)kc * 128),
                                         (__attribute__((address_space(3))) void*)(sa + i * 1024), 16, 0, 0);
    };
    auto issue = [&](int kc, int st) {
#pragma unroll
        for (int j = 0; j < CNT; ++j) issue_piece(kc, st, j);
    };
    // what phase 2 needs from memory, asked for now: 2^-k of the image, (mean, rstd) of this lane's two rows, the first head's bias
    const float ws = *reinterpret_cast<const float*>(p.Wh + (size_t)p.N * K * 4);
    float row_rstd[2] = {1.0f, 1.0f}, row_nmr[2] = {0.0f, 0.0f};
    if (p.ln_in) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int64_t m = token_of(32 * wave + 16 * g + c);
            row_rstd[g] = p.ln_in[2 * m + 1];
            if (p.ln_s) row_nmr[g] = -p.ln_in[2 * m] * row_rstd[g];
        }
    }
    f32x4 bias_cur[NBH], bias_nxt[NBH];
    auto fetch_bias = [&](f32x4 (&dst)[NBH], int first_col) {
#pragma unroll
        for (int n = 0; n < NBH; ++n) dst[n] = *reinterpret_cast<const f32x4*>(p.bias + first_col + 16 * n + 4 * qd);
    };
    // (the five-wave form cannot afford the registers across the loop: two of its workgroups share a CU only while four waves
    // fit one SIMD - 128 registers each - since both may start on the same SIMD)
    constexpr bool EARLY_BIAS = NWV != 5;
    if (EARLY_BIAS) fetch_bias(bias_cur, n0);
    f32x4 acc[HPW][2][NBH];
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int n = 0; n < NBH; ++n) acc[hh][g][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int sw = hfqa_swizzle(c);
    const uint32_t pos0 = (uint32_t)(((2 * qd) ^ sw) & 7) * 16, pos1 = (uint32_t)(((2 * qd + 1) ^ sw) & 7) * 16;
    const uint32_t a_row = (uint32_t)((32 * wave + c) * 128);         // + rb * 2048
    const uint32_t w_row = (uint32_t)(FT::A_BYTES + c * 128);          // + nb * 2048
    issue(0, 0);
    for (int kc = 0; kc < nk; ++kc) {
        wait_vmcnt<0>();                  // chunk kc of this wave has landed
        __builtin_amdgcn_s_barrier();     // ... of every wave; every wave is done reading chunk kc - 1
        if (kc == 0) { PAFUSE_STAMP(3); }
        // the refill of the other stage, all of it at once: this phase is bound by the operand stream (one chunk in flight),
        // and every cycle a piece waits for its issue slot behind MFMAs is a cycle the stream idles (spreading the pieces over
        // the MFMA groups, as hgemm_tile does, cost 9 % here)
        if (kc + 1 < nk) issue(kc + 1, (kc + 1) & 1);
        const uint32_t sbase = lds0 + (uint32_t)((kc & 1) * FT::STAGE_BYTES);
        __builtin_amdgcn_s_setprio(1);
        u32x4 af[2][2], wf[2][2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            af[g][0] = lds_read128<0>(sbase + a_row + g * 2048 + pos0);
            af[g][1] = lds_read128<0>(sbase + a_row + g * 2048 + pos1);
        }
        wf[0][0] = lds_read128<0>(sbase + w_row + pos0);
        wf[0][1] = lds_read128<0>(sbase + w_row + pos1);
        static_for<NB>([&](auto N) {
            constexpr int n = decltype(N)::value;
            if constexpr (n + 1 < NB) {
                wf[(n + 1) & 1][0] = lds_read128<(n + 1) * 2048>(sbase + w_row + pos0);
                wf[(n + 1) & 1][1] = lds_read128<(n + 1) * 2048>(sbase + w_row + pos1);
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(wf[n & 1][0]), "+v"(wf[n & 1][1]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(wf[n & 1][0]), "+v"(wf[n & 1][1]));
            }
            const f16x8 w0 = __builtin_bit_cast(f16x8, wf[n & 1][0]), w1 = __builtin_bit_cast(f16x8, wf[n & 1][1]);
            const f16x8 w2 = w0 * (_Float16)0.00048828125f;   // 2^-11
#pragma unroll
            for (int g = 0; g < 2; ++g) {   // small terms first, the leading product last
                const f16x8 a_hi = __builtin_bit_cast(f16x8, af[g][0]), a_lo = __builtin_bit_cast(f16x8, af[g][1]);
                f32x4& d = acc[n / NBH][g][n % NBH];
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2, a_lo, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, a_hi, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, a_hi, d, 0, 0, 0);
            }
        });
        __builtin_amdgcn_s_setprio(0);
    }
    PAFUSE_STAMP(1);
    if (!EARLY_BIAS) fetch_bias(bias_cur, n0);
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh) {
        if (hh + 1 < HPW) fetch_bias(bias_nxt, n0 + (hh + 1) * 3 * DP);   // (lands during this head's phases)
        __syncthreads();   // every wave is done with the ring / the tiles of the head before: the LDS becomes this head's q | k | v tiles
        hfqa_attention_phases<LP, DP>(fp, acc[hh], smem, seq0, head0 + hh, n0 + hh * 3 * DP, wave, c, qd, tid, token_of, bias_cur, row_rstd,
                                      row_nmr, ws, hh);
        if (hh + 1 < HPW) {
#pragma unroll
            for (int n = 0; n < NBH; ++n) bias_cur[n] = bias_nxt[n];
        }
    }
    PAFUSE_STAMP(2);
}


// ----------------------------------------------------------------------------------------------------------------
// The MLP of a block in ONE kernel, H pipeline (round 4):
//   x <- epilogue(x + GELU(LNfold(x) W1^T + b1) W2^T + b2)        (common/mixste.py:37-43,115: fc1 -> GELU -> fc2 -> + residual)
// replaces the fc1 hgemm launch + the fc2 whole-row launch and the [M, 2C] hidden H image between them (8 C bytes per token
// written and read back: 28 % of the bytes a block still moves once qkv + attention are fused).
// Workgroup = 4 waves = 128 tokens x all C channels; wave w owns tokens 32 w .. 32 w + 31 through BOTH layers.  The hidden
// activations never leave the registers: with the weight fragment as the MFMA's first operand the accumulator of
//   phase 1   acc1[nt1] = W1[slab rows 32 nt1 ..] . x^T     lane (r, h): token r, hidden units 8 q + 4 h + {0..3}, q = 0 .. 3
// is, after bias / folded LayerNorm, GELU and the hi / lo split, exactly a B operand of the next MFMA
//   phase 2   acc2[nt2] += W2[rows 32 nt2 .., 16 k-slots] . h^T     lane (r, h) supplies 8 k-slots of token r
// if k-slot (h, i) of the 16-deep step s of column block nt1 is taken to be hidden unit 32 nt1 + 16 s + 8 (i >> 2) + 4 h + (i & 3)
// - a permutation inside each group of 16 hidden units that the fc2 weight image carries in its column order
// (pafuse_block_weights.fc2_hp: the H image of W2[:, perm]; the two operands of a product only have to agree on the slot).
// Loop: hidden slabs of 64 units (NT1 = 2 column blocks per wave); per slab  [phase 1: C / 32 chunks of 32 k: A 128 x 128 B +
// W1 64 x 128 B]  [phase 2: 4 stages of 16 k-slots: W2 C x 64 B], all through ONE ring of three 24 KB slots fed by LDS-DMA two
// stages ahead (the A tile is re-streamed from L2 for every slab: 128 x 4 C bytes against 2 x 64 x 4 C of weights).  Two
// workgroups per CU.  acc2 (C / 2 registers) lives through the whole loop; the whole-row epilogue is hgemm's.
// ----------------------------------------------------------------------------------------------------------------
struct MlpParams {
    GemmParams g;            // the fc2 launch's parameters: Ah = fc1's operand (the centred H image of x), Wh = fc2_hp, bias = b2,
    //                          resid_h / out_xh / ln_stats / post / next / head as for hgemm's whole-row epilogue; N = C, K = 2 C
    const uint8_t* W1h;      // H image of fc1.weight (of W1 (.) g with the LayerNorm folded) [2C][C]
    const float* bias1;      // [2C] fc1 bias (folded: lt)
    const float* ln_in;      // folded LayerNorm in front of fc1: (mean, rstd) per row - the image is centred, only rstd is used - or null
};

template <int NT2>
struct MlpTile {
    static constexpr int C = 32 * NT2, HID = 2 * C, NSLAB = HID / 64, NK1 = C / 32;
    static constexpr int SLOT = 24 * 1024, NSLOT = 3, RING = SLOT * NSLOT;
    static constexpr int CNT1 = 6;                       // phase-1 stage: 16 A + 8 W1 DMA instructions over 4 waves
    static constexpr int P2_PIECES = C * 64 / 1024, CNT2 = (P2_PIECES + 3) / 4;
    static constexpr int LDS_BYTES = RING + HID * 4;     // + the fc1 bias vector
    static_assert(C * 64 <= SLOT && CNT2 <= CNT1, "a phase-2 stage must fit a ring slot");
};

template <int NT2, int MINW>
__global__ void __launch_bounds__(256, MINW) hmlp_kernel(const MlpParams mp) {
    PAFUSE_XQ_GUARD();
    using T = MlpTile<NT2>;
    constexpr int C = T::C, HID = T::HID, NSLAB = T::NSLAB, NK1 = T::NK1, CNT1 = T::CNT1, CNT2 = T::CNT2, SLOT = T::SLOT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const GemmParams& p = mp.g;
    uint8_t* const lds = reinterpret_cast<uint8_t*>(smem);
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)smem;
    float* const b1s = smem + T::RING / 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    int tile;
    {   // XCD-aware tile order (speed only), as hgemm_tile
        const int b = blockIdx.x, nb = gridDim.x;
        const int xcd = b & 7, q = nb >> 3, rem = nb & 7;
        tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (b >> 3);
    }
    const int64_t m0 = (int64_t)tile * 128;
    if (m0 >= p.M) return;
    PAFUSE_STAMP(0);

    // ---- DMA sources (lane-constant parts).  Phase-1 stage: instruction i = wave + 4 j; j < 4: A rows 8 i .. (128-byte rows,
    // lane l: row 8 i + l / 8, LDS position l % 8, source slot position ^ ((row >> 1) & 7)); j = 4, 5: W1 rows 8 (i - 16) .. of the
    // slab.  Phase-2 stage: instruction i = wave + 4 j < P2_PIECES (surplus slots re-issue the last one): W2 rows 16 i .. (64-byte
    // rows, lane l: row 16 i + l / 4, position l % 4, source slot position ^ ((row >> 2) & 3)).
    const uint8_t* const Abase = p.Ah + (size_t)m0 * C * 4;
    int a_off[4], w1_off[2], w2_off[CNT2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * (wave + 4 * j) + (lane >> 3);
        const int64_t lim = p.M - 1 - m0;
        const int grow = row < lim ? row : (int)lim;     // tail rows read a valid row (never stored)
        a_off[j] = grow * C * 4 + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 8 * (wave + 4 * j) + (lane >> 3);     // 0 .. 63 inside the slab
        w1_off[j] = row * C * 4 + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < CNT2; ++j) {
        int i = wave + 4 * j;
        i = i < T::P2_PIECES ? i : T::P2_PIECES - 1;
        const int row = 16 * i + (lane >> 2);
        w2_off[j] = row * HID * 4 + (((lane & 3) ^ ((row >> 2) & 3)) << 4);
    }
    // stage (slab, idx): idx < NK1: phase-1 chunk idx; else phase-2 stage idx - NK1.  Piece j of it into ring slot `slot`.
    auto issue_piece = [&](auto J, int slab, int idx, int slot) {
        constexpr int j = decltype(J)::value;
#if defined(PAFUSE_HMLP_ABL) && (PAFUSE_HMLP_ABL & 2)   // diagnostic: no operand stream (the compute side alone, on whatever the LDS holds)
        return;
#endif
        uint8_t* const dst = lds + slot * SLOT;
        if (idx < NK1) {      // workgroup-uniform
            if constexpr (j < 4) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Abase + a_off[j] + idx * 128),
                                                 (__attribute__((address_space(3))) void*)(dst + (wave + 4 * j) * 1024), 16, 0, 0);
            } else if constexpr (j < 6) {
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(mp.W1h + (size_t)slab * 64 * C * 4 + w1_off[j - 4] + idx * 128),
                    (__attribute__((address_space(3))) void*)(dst + (16 + wave + 4 * (j - 4)) * 1024), 16, 0, 0);
            }
        } else {
            if constexpr (j < CNT2) {
                int i = wave + 4 * j;
                i = i < T::P2_PIECES ? i : T::P2_PIECES - 1;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(p.Wh + w2_off[j] + (slab * 64 + (idx - NK1) * 16) * 4),
                    (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
            }
        }
    };
    // the producer's position: the next stage to issue
    int pslab = 0, pidx = 0, pslot = 0;
    auto advance = [&]() {
        pidx = pidx + 1 == NK1 + 4 ? 0 : pidx + 1;
        pslab += pidx == 0;
        pslot = pslot == 2 ? 0 : pslot + 1;
    };
    // fc1 bias -> LDS; this lane's row factor rstd * 2^-k1
    for (int i = tid; i < HID / 4; i += 256) *reinterpret_cast<f32x4*>(b1s + 4 * i) = *reinterpret_cast<const f32x4*>(mp.bias1 + 4 * i);
    float rs1 = *reinterpret_cast<const float*>(mp.W1h + (size_t)HID * C * 4);
    {
        const int64_t m = m0 + 32 * wave + r;
        if (mp.ln_in) rs1 *= mp.ln_in[2 * (m < p.M ? m : p.M - 1) + 1];
    }
    wait_vmcnt<0>();   // (nothing but LDS-DMA may be in flight inside the loop: its waits count instructions)
    asm volatile("" : "+v"(rs1));
    __syncthreads();   // the bias vector is in place for every wave
    static_for<CNT1>([&](auto J) { issue_piece(J, 0, 0, 0); });
    static_for<CNT1>([&](auto J) { issue_piece(J, 0, 1, 1); });
    pidx = 2, pslot = 2;
    static_assert(NK1 >= 2, "two phase-1 chunks open the ring");

    f32x16 acc2[NT2];
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc2[nt][i] = 0.f;

    // fragment positions: phase 1, 128-byte rows (slot 2 (2 s2 + h) + slice, swizzle (r >> 1) & 7); phase 2, 64-byte rows
    // (slot 2 h + slice, swizzle (r >> 2) & 3)
    const int sw7 = (r >> 1) & 7, sw3 = (r >> 2) & 3;
    uint32_t pos1[2][2], pos2[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) pos1[s2][sl] = (uint32_t)(((2 * (2 * s2 + h) + sl) ^ sw7) & 7) * 16;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) pos2[sl] = (uint32_t)(((2 * h + sl) ^ sw3) & 3) * 16;
    const uint32_t a_row = (uint32_t)((wave * 32 + r) * 128);
    const uint32_t w1_row = (uint32_t)(16 * 1024 + r * 128);     // + nt1 * 32 * 128
    const uint32_t w2_row = (uint32_t)(r * 64);                  // + nt2 * 32 * 64

    int cslot = 0;                                  // the consumer's ring slot
    int left = NSLAB * (NK1 + 4);                   // stages not yet consumed
    // top of a stage: its pieces have landed (the next stage's - CNT1 or CNT2 per wave, whichever it is - may still fly), every
    // wave is past the stage before; then the refill two stages ahead goes out piece by piece behind the first MFMAs
    auto open_stage = [&](bool next_is_p1) {
        if (left == 1) wait_vmcnt<0>();
        else if (next_is_p1) wait_vmcnt<CNT1>();
        else wait_vmcnt<CNT2>();
        __builtin_amdgcn_s_barrier();
    };
    for (int slab = 0; slab < NSLAB; ++slab) {
        f32x16 acc1[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc1[nt][i] = 0.f;
        // ---- phase 1: acc1 = W1[slab] . x^T over the C input channels
        for (int kc = 0; kc < NK1; ++kc) {
            open_stage(kc + 1 < NK1);
            if (slab == 0 && kc == 0) { PAFUSE_STAMP(3); }
            const bool refill = left > 2;
            const uint32_t sbase = lds0 + (uint32_t)(cslot * SLOT);
            __builtin_amdgcn_s_setprio(1);
            u32x4 af[2][2], wf[2][2];
            af[0][0] = lds_read128<0>(sbase + a_row + pos1[0][0]);
            af[0][1] = lds_read128<0>(sbase + a_row + pos1[0][1]);
            wf[0][0] = lds_read128<0>(sbase + w1_row + pos1[0][0]);
            wf[0][1] = lds_read128<0>(sbase + w1_row + pos1[0][1]);
            static_for<4>([&](auto G) {
                constexpr int g = decltype(G)::value, s2 = g >> 1, nt = g & 1;
                constexpr bool next_a = g == 1;
                if constexpr (next_a) {
                    af[1][0] = lds_read128<0>(sbase + a_row + pos1[1][0]);
                    af[1][1] = lds_read128<0>(sbase + a_row + pos1[1][1]);
                }
                if constexpr (g + 1 < 4) {
                    constexpr int s2n = (g + 1) >> 1, off = ((g + 1) & 1) * 32 * 128;
                    wf[(g + 1) & 1][0] = lds_read128<off>(sbase + w1_row + pos1[s2n][0]);
                    wf[(g + 1) & 1][1] = lds_read128<off>(sbase + w1_row + pos1[s2n][1]);
                }
                if constexpr (next_a)
                    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af[s2][0]), "+v"(af[s2][1]), "+v"(wf[g & 1][0]), "+v"(wf[g & 1][1]));
                else if constexpr (g + 1 < 4)
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(af[s2][0]), "+v"(af[s2][1]), "+v"(wf[g & 1][0]), "+v"(wf[g & 1][1]));
                else
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[s2][0]), "+v"(af[s2][1]), "+v"(wf[g & 1][0]), "+v"(wf[g & 1][1]));
                const f16x8 w0 = __builtin_bit_cast(f16x8, wf[g & 1][0]), w1 = __builtin_bit_cast(f16x8, wf[g & 1][1]);
                const f16x8 w2 = w0 * (_Float16)0.00048828125f;   // 2^-11
                const f16x8 a_hi = __builtin_bit_cast(f16x8, af[s2][0]), a_lo = __builtin_bit_cast(f16x8, af[s2][1]);
                acc1[nt] = mfma_f16_k16(w2, a_lo, acc1[nt]);   // small terms first, the leading product last
                if (refill) {   // two pieces of the refill per group (CNT1, CNT2 <= 6 < 8)
                    issue_piece(std::integral_constant<int, 2 * g>{}, pslab, pidx, pslot);
                    issue_piece(std::integral_constant<int, 2 * g + 1>{}, pslab, pidx, pslot);
                }
                acc1[nt] = mfma_f16_k16(w1, a_hi, acc1[nt]);
                acc1[nt] = mfma_f16_k16(w0, a_hi, acc1[nt]);
            });
            __builtin_amdgcn_s_setprio(0);
            if (refill) advance();
            cslot = cslot == 2 ? 0 : cslot + 1;
            --left;
        }
        // ---- phase 2: the slab's 64 hidden units as four 16-deep steps of fc2
        static_for<4>([&](auto JJ) {
            constexpr int jj = decltype(JJ)::value, nt1 = jj >> 1, s = jj & 1;
            open_stage(jj == 3);   // (the stage behind the slab's last one is the next slab's first phase-1 chunk)
            const bool refill = left > 2;
            const uint32_t sbase = lds0 + (uint32_t)(cslot * SLOT);
            u32x4 wf[2][2];
            wf[0][0] = lds_read128<0>(sbase + w2_row + pos2[0]);
            wf[0][1] = lds_read128<0>(sbase + w2_row + pos2[1]);
            // this lane's 8 hidden values of the step: registers 4 q + e, q = 2 s, 2 s + 1 -> k-slots 4 (q - 2 s) + e
            f32x4 v[2];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int q = 2 * s + qq;
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(b1s + slab * 64 + 32 * nt1 + 8 * q + 4 * h);
#pragma unroll
#if defined(PAFUSE_HMLP_ABL) && (PAFUSE_HMLP_ABL & 1)   // diagnostic builds of tools/hgemm_bench.hip: no GELU (results wrong by design)
                for (int e = 0; e < 4; ++e) v[qq][e] = fmaf(rs1, acc1[nt1][4 * q + e], b4[e]);
#else
                for (int e = 0; e < 4; ++e) v[qq][e] = gelu_erf(fmaf(rs1, acc1[nt1][4 * q + e], b4[e]));
#endif
            }
            const f16x8x2 hf = split2h(v[0], v[1]);
            __builtin_amdgcn_s_setprio(1);
            static_for<NT2>([&](auto N) {
                constexpr int nt = decltype(N)::value;
                if constexpr (nt + 1 < NT2) {
                    wf[(nt + 1) & 1][0] = lds_read128<(nt + 1) * 32 * 64>(sbase + w2_row + pos2[0]);
                    wf[(nt + 1) & 1][1] = lds_read128<(nt + 1) * 32 * 64>(sbase + w2_row + pos2[1]);
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(wf[nt & 1][0]), "+v"(wf[nt & 1][1]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[nt & 1][0]), "+v"(wf[nt & 1][1]));
                }
                const f16x8 w0 = __builtin_bit_cast(f16x8, wf[nt & 1][0]), w1 = __builtin_bit_cast(f16x8, wf[nt & 1][1]);
                const f16x8 w2 = w0 * (_Float16)0.00048828125f;
                acc2[nt] = mfma_f16_k16(w2, hf.lo, acc2[nt]);
                if constexpr (nt < CNT1) {
                    if (refill) issue_piece(std::integral_constant<int, nt>{}, pslab, pidx, pslot);
                }
                acc2[nt] = mfma_f16_k16(w1, hf.hi, acc2[nt]);
                acc2[nt] = mfma_f16_k16(w0, hf.hi, acc2[nt]);
            });
            __builtin_amdgcn_s_setprio(0);
            if (refill) advance();
            cslot = cslot == 2 ? 0 : cslot + 1;
            --left;
        });
    }
    PAFUSE_STAMP(1);
    __syncthreads();   // the ring becomes the epilogue's scratch
    const float ws2 = *reinterpret_cast<const float*>(p.Wh + (size_t)C * HID * 4);
    {
        constexpr int VEC = (7 * 128 + 3) / 4 * 4;
        constexpr size_t RINGF = T::RING / sizeof(float);
        constexpr auto need = [](int nth) { return (size_t)VEC + 5 * C + (size_t)4 * 32 * (32 * nth + 4); };
        constexpr int NTH = (NT2 % 2 == 0 && need(2) <= RINGF) ? 2 : 1;
        static_assert(need(NTH) <= RINGF, "epilogue scratch must fit the ring");
        const float* const src[5] = {p.bias, p.post_w, p.post_b, p.next_w, p.next_b};
#pragma unroll
        for (int v = 0; v < 5; ++v)
            if (src[v])
                for (int i = tid; i < C / 4; i += 256)
                    *reinterpret_cast<f32x4*>(smem + VEC + v * C + 4 * i) = *reinterpret_cast<const f32x4*>(src[v] + 4 * i);
        __syncthreads();
        epilogue_rows_h<1, NT2, 128, 4, NTH, true>(acc2, p, m0, 0, wave, 0, r, h, wave, lane, smem, ws2);
    }
}

}  // namespace pafuse
