// xgemm.hpp - the linear layers of the bf16x3 product scheme on the LDS-DMA pipeline of hgemm.hpp ("X pipeline", round 5).
//
// bf16x3 is the scheme whose operands carry every bit of the fp32 numbers they stand for: x = s0 + s1 + s2 exactly, three bf16
// slices of 8 significand bits each (kernels.hpp split3), six MFMA products per k (the terms above 2^-24 relative), fp32
// accumulation.  Rounds 2 - 3 ran it on kernels that stage A as fp32 and split it in registers, per column tile; round 4 built
// a better structure - operands pre-split ONCE by their producer, both streamed global -> LDS by DMA, qkv + attention in one
// kernel, the residual stream centred and kept only as the image the GEMMs read - but for the 22-bit f16x2 scheme only.  This
// file is that structure at the reference's operand width:
//   * "X image" of a matrix [rows][K], K % 32 == 0: rows of 6 K bytes, per chunk of 32 k the 32 bf16 of slice 0, then of
//     slice 1, then of slice 2 - [rows][K/32][3][32 x bf16], 192 bytes per row and chunk.  An ACTIVATION is split once, by
//     the kernel that produces it (whole-row epilogue -> x, attention -> o, fc1 epilogue -> the MLP hidden, embed -> x); a
//     WEIGHT once per weight version (xsplit_weights_kernel; no scaling: bf16 has fp32's exponent range).  The three slices of
//     an fp32 number are exact (split3), so the image of the centred residual stream IS the fp32 row x - mean(x), bit for bit.
//   * LDS stage (one per ring slot): per operand three slice PLANES of [rows][2 BKC bytes] (64-byte rows at 32-deep chunks,
//     32-byte rows at 16-deep ones), A planes then W planes.  The 16-byte sub-block sb of plane row r sits at position
//     sb ^ ((r >> 2) & 3) (64-byte rows) or sb ^ ((r >> 3) & 1) (32-byte rows) - applied on the SOURCE address of the DMA
//     (its LDS side is lane-linear: one 1 KiB wave instruction fills 16 or 32 whole plane rows) and again on the fragment
//     reads: every ds_read_b128 of the K loop is conflict-free (tools/lds_layout_check.py walks the lane groups).
//   * K loop: per 16-deep step and 32-column block SIX v_mfma_f32_32x32x16_bf16 on one accumulator, small terms first
//     (w0 a2, w2 a0, w1 a1, w0 a1, w1 a0, w0 a0), the next fragments and the refill DMA pieces issued in their shadow; the ring,
//     its counted vmcnt waits and the one raw s_barrier per chunk are hgemm_tile's.
// Kernels: xgemm_kernel (one linear layer: plain with bias / folded LayerNorm / GELU, or whole-row with residual + LayerNorms,
// the epilogue shared with the H pipeline: hgemm.hpp epilogue_rows_h<.., NSLICE = 3>) and xfqa_kernel (qkv projection + attention
// of whole sequences x 1 - 2 heads; the projection is the same K loop on gathered rows).
#pragma once
#include "hgemm.hpp"

namespace pafuse {

// ---- X images ---------------------------------------------------------------------------------------------------------
// (xoff / xsplit_store4 / xsplit_store8 / xjoin4 live in kernels.hpp beside the H-image helpers: embed_kernel uses them)
__global__ void __launch_bounds__(256) xsplit_rows_kernel(const float* X, uint8_t* out, int64_t R, int K) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one sub-block of 8 k per thread
    const int spr = K / 8;
    if (idx >= R * spr) return;
    const int64_t row = idx / spr;
    const int k = (int)(idx % spr) * 8;
    const float* src = X + row * K + k;
    xsplit_store8(out + (size_t)row * K * 6, k, *reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4));
}

// ---- the tile -----------------------------------------------------------------------------------------------------------
template <int WM, int WN, int NT, int BKC>
struct XTile {
    static_assert(BKC == 32 || BKC == 16, "chunk depth");
    static constexpr int NW = WM * WN, NTHR = NW * 64;
    static constexpr int BM = WM * 32, BN = WN * NT * 32;
    static constexpr int PROWB = BKC * 2;                     // bytes per row of one slice plane
    static constexpr int RPI = 1024 / PROWB, CPR = PROWB / 16;   // plane rows per DMA instruction (16 / 32), 16-byte sub-blocks per row (4 / 2)
    static constexpr int A_PLANE = BM * PROWB, W_PLANE = BN * PROWB;
    static constexpr int A_BYTES = 3 * A_PLANE, W_BYTES = 3 * W_PLANE, STAGE_BYTES = A_BYTES + W_BYTES;
    static constexpr int IAP = A_PLANE / 1024, IWP = W_PLANE / 1024;   // DMA wave-instructions per plane
    static constexpr int IA = 3 * IAP, IW = 3 * IWP;                   // ... per chunk
    static constexpr int CNT = (IA + IW + NW - 1) / NW;                // per wave (surplus slots re-issue the last piece)
    static_assert(A_PLANE % 1024 == 0 && W_PLANE % 1024 == 0, "whole DMA pieces");
    static_assert(2 * W_PLANE + (NT - 1) * 32 * PROWB < 65536 && 2 * A_PLANE < 65536, "ds_read offset field");
};
template <int BKC>
__device__ __forceinline__ int x_swizzle(int row) { return BKC == 32 ? (row >> 2) & 3 : (row >> 3) & 1; }

// One tile's K loop: acc[nt] (row-per-lane: lane (r, h) of wave (wm, wn) owns tile row 32 wm + r and, per 32-column block, the
// columns 8 q + 4 h + {0..3} - the weight fragment is the MFMA's first operand) += A[tile rows] . W[tile columns]^T.
// a_row(i): pointer to the first byte of tile row i's image row (plain layers: consecutive rows; the fused qkv + attention
// kernel: gathered tokens); Wrows: the image row of the tile's first column.  smem: NSTAGE * STAGE_BYTES.
template <int WM, int WN, int NT, int NSTAGE, int BKC, class ARow>
__device__ __forceinline__ void xgemm_mainloop(f32x16 (&acc)[NT], ARow a_row, const uint8_t* const Wrows, const int K, float* smem,
                                               const int wave, const int lane, const GemmParams& p) {
    using T = XTile<WM, WN, NT, BKC>;
    constexpr int NW = T::NW, CNT = T::CNT, IA = T::IA, IW = T::IW, IAP = T::IAP, IWP = T::IWP, PROWB = T::PROWB;
    constexpr int RPI = T::RPI, CPR = T::CPR, A_PLANE = T::A_PLANE, W_PLANE = T::W_PLANE;
    constexpr int NS2 = BKC / 16;            // 16-deep MFMA steps per chunk
    static_assert(NSTAGE >= 2 && NSTAGE <= 4 && CNT * (NSTAGE - 1) < 64, "ring depth / vmcnt range");
    uint8_t* const lds = reinterpret_cast<uint8_t*>(smem);
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)smem;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int nk = K / BKC;

    // ---- DMA sources.  Instruction i of a chunk (0 .. IA + IW - 1) belongs to wave i % NW and fills the 1 KiB at i * 1024 of
    // the stage: plane i / IAP (A) resp. (i - IA) / IWP (W), plane rows RPI (i % I.P) .. + RPI - 1.  Lane l fills position
    // l % CPR of row l / CPR with the sub-block that belongs there: position ^ swizzle(row).
    const uint8_t* src[CNT];   // this lane's source of piece j at chunk 0
#pragma unroll
    for (int j = 0; j < CNT; ++j) {
        int i = wave + j * NW;
        i = i < IA + IW ? i : IA + IW - 1;
        const bool is_a = i < IA;
        const int plane = is_a ? i / IAP : (i - IA) / IWP;
        const int row = RPI * (is_a ? i % IAP : (i - IA) % IWP) + lane / CPR;
        const int sb = (lane % CPR) ^ x_swizzle<BKC>(row);
        src[j] = (is_a ? a_row(row) : Wrows + (size_t)row * K * 6) + plane * 64 + sb * 16;
    }
    auto issue_piece = [&](int kc, int st, int j) {
#if defined(PAFUSE_X_ABL) && (PAFUSE_X_ABL & 1)   // diagnostic builds of tools/xgemm_bench.hip: no operand stream (results wrong by design)
        return;
#endif
        int i = wave + j * NW;  // wave-uniform
        i = i < IA + IW ? i : IA + IW - 1;
        // chunk kc of a row: BKC = 32: the 192-byte chunk kc; BKC = 16: the sub-blocks 2 (kc & 1), + 1 of chunk kc >> 1
        const size_t off = BKC == 32 ? (size_t)kc * 192 : (size_t)(kc >> 1) * 192 + (kc & 1) * 32;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + off),
                                         (__attribute__((address_space(3))) void*)(lds + st * T::STAGE_BYTES + i * 1024), 16, 0, 0);
    };

#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;

    // fragment addresses inside a stage: plane row base + the swizzled position of sub-block 2 s2 + h
    uint32_t pos[NS2];
#pragma unroll
    for (int s2 = 0; s2 < NS2; ++s2) pos[s2] = (uint32_t)(((2 * s2 + h) ^ x_swizzle<BKC>(r)) & (CPR - 1)) * 16;
    const uint32_t a_row0 = (uint32_t)((wm * 32 + r) * PROWB);
    const uint32_t w_row0 = (uint32_t)(T::A_BYTES + (wn * NT * 32 + r) * PROWB);

#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < nk) {
#pragma unroll
            for (int j = 0; j < CNT; ++j) issue_piece(s, s, j);
        }

#define PAFUSE_PIN_ACC(A) asm volatile("" : "+v"(A))
#if defined(PAFUSE_X_ABL) && (PAFUSE_X_ABL & 2)   // diagnostic: no MFMAs (the stream, the fragment reads and the barriers alone)
#define PAFUSE_X_MFMA(W, A, ACC) asm volatile("" : "+v"(ACC) : "v"(W), "v"(A))
#else
#define PAFUSE_X_MFMA(W, A, ACC) ACC = mfma_bf16_k16(W, A, ACC)
#endif
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
        __builtin_amdgcn_s_setprio(1);
        constexpr int NG = NS2 * NT;          // groups (s2, nt) of six MFMAs on one accumulator
        u32x4 af[2][3], wf[2][3];             // [buffer][slice]: A fragment per 16-deep step, W fragment per group
        auto read_a = [&](auto S2) {
            constexpr int s2 = decltype(S2)::value;
            af[s2 & 1][0] = lds_read128<0>(sbase + a_row0 + pos[s2]);
            af[s2 & 1][1] = lds_read128<A_PLANE>(sbase + a_row0 + pos[s2]);
            af[s2 & 1][2] = lds_read128<2 * A_PLANE>(sbase + a_row0 + pos[s2]);
        };
        auto read_w = [&](auto G) {
            constexpr int g = decltype(G)::value, s2 = g / NT, off = (g % NT) * 32 * PROWB;
            wf[g & 1][0] = lds_read128<off>(sbase + w_row0 + pos[s2]);
            wf[g & 1][1] = lds_read128<off + W_PLANE>(sbase + w_row0 + pos[s2]);
            wf[g & 1][2] = lds_read128<off + 2 * W_PLANE>(sbase + w_row0 + pos[s2]);
        };
        read_a(std::integral_constant<int, 0>{});
        read_w(std::integral_constant<int, 0>{});
        static_for<NG>([&](auto G) {
            constexpr int g = decltype(G)::value, s2 = g / NT, nt = g % NT;
            // what the NEXT group needs is issued first and stays in flight during this group's MFMAs
            constexpr bool next_a = g + 1 < NG && (g + 1) % NT == 0;
            constexpr int flying = g + 1 < NG ? (next_a ? 6 : 3) : 0;
            if constexpr (next_a) read_a(std::integral_constant<int, (g + 1) / NT>{});
            if constexpr (g + 1 < NG) read_w(std::integral_constant<int, g + 1>{});
#define PAFUSE_X_WAIT(N)                                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                                         \
                 : "+v"(af[s2 & 1][0]), "+v"(af[s2 & 1][1]), "+v"(af[s2 & 1][2]), "+v"(wf[g & 1][0]), "+v"(wf[g & 1][1]), "+v"(wf[g & 1][2]))
            if constexpr (flying == 6) PAFUSE_X_WAIT(6);
            else if constexpr (flying == 3) PAFUSE_X_WAIT(3);
            else PAFUSE_X_WAIT(0);
#undef PAFUSE_X_WAIT
            const bf16x8 w0 = __builtin_bit_cast(bf16x8, wf[g & 1][0]), w1 = __builtin_bit_cast(bf16x8, wf[g & 1][1]),
                         w2 = __builtin_bit_cast(bf16x8, wf[g & 1][2]);
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, af[s2 & 1][0]), a1 = __builtin_bit_cast(bf16x8, af[s2 & 1][1]),
                         a2 = __builtin_bit_cast(bf16x8, af[s2 & 1][2]);
            PAFUSE_PIN_ACC(acc[nt]);
            PAFUSE_X_MFMA(w0, a2, acc[nt]);   // small terms first, the leading product last
            {   // this group's share of the refill DMA, in the shadow of the MFMA just issued
                constexpr int PER = (CNT + NG - 1) / NG, j0 = g * PER, j1 = (g + 1) * PER < CNT ? (g + 1) * PER : CNT;
                if constexpr (j0 < j1) {
                    asm volatile("" ::: "memory");
                    if (refill) {
#pragma unroll
                        for (int j = j0; j < j1; ++j) issue_piece(kn, stn, j);
                    }
                    asm volatile("" ::: "memory");
                }
            }
            PAFUSE_PIN_ACC(acc[nt]);
            PAFUSE_X_MFMA(w2, a0, acc[nt]);
            PAFUSE_PIN_ACC(acc[nt]);
            PAFUSE_X_MFMA(w1, a1, acc[nt]);
            PAFUSE_PIN_ACC(acc[nt]);
            PAFUSE_X_MFMA(w0, a1, acc[nt]);
            PAFUSE_PIN_ACC(acc[nt]);
            PAFUSE_X_MFMA(w1, a0, acc[nt]);
            PAFUSE_PIN_ACC(acc[nt]);
            PAFUSE_X_MFMA(w0, a0, acc[nt]);
        });
        __builtin_amdgcn_s_setprio(0);
    }
#undef PAFUSE_PIN_ACC
#undef PAFUSE_X_MFMA
}

template <int WM, int WN, int NT, int EPI, int NSTAGE, int BKC, bool HRES = false>
__device__ __forceinline__ void xgemm_tile(const GemmParams& p, const int b, const int nb, float* smem) {
    using T = XTile<WM, WN, NT, BKC>;
    constexpr int NW = T::NW, BM = T::BM, BN = T::BN;
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
    const int K = p.K;
    const uint8_t* const Abase = p.Ah + (size_t)m0 * K * 6;
    const int64_t lim = p.M - 1 - m0;     // tail rows of A read a valid row (never stored)
    f32x16 acc[NT];
    xgemm_mainloop<WM, WN, NT, NSTAGE, BKC>(
        acc, [&](int row) { return Abase + (size_t)(row < lim ? row : lim) * K * 6; }, p.Wh + (size_t)n0 * K * 6, K, smem, wave, lane, p);
    PAFUSE_STAMP(1);
    __syncthreads();  // the ring becomes the epilogue's scratch

    if constexpr (EPI != EPI_BIAS) {
        static_assert(EPI == EPI_ROWLN, "the X pipeline is inference only");
        constexpr int VEC = (7 * BM * WN + 3) / 4 * 4;  // behind the cross-wave reduction slots
        constexpr size_t RINGF = (size_t)NSTAGE * T::STAGE_BYTES / sizeof(float);
        constexpr int SEG = HRES ? 48 : 32;             // floats of a slab row per 32-column block (an image row segment is 192 bytes)
        constexpr auto need = [](int nth) { return (size_t)VEC + 5 * BN + (size_t)NW * 32 * (SEG * nth + 4); };
        constexpr int NTH = (NT % 2 == 0 && need(2) <= RINGF) ? 2 : 1;
        static_assert(need(NTH) <= RINGF, "epilogue scratch must fit the ring");
        const float* const src[5] = {p.bias, p.post_w, p.post_b, p.next_w, p.next_b};
#pragma unroll
        for (int v = 0; v < 5; ++v)
            if (src[v])  // workgroup-uniform
                for (int i = tid; i < BN / 4; i += T::NTHR)
                    *reinterpret_cast<f32x4*>(smem + VEC + v * BN + 4 * i) = *reinterpret_cast<const f32x4*>(src[v] + n0 + 4 * i);
        __syncthreads();
        epilogue_rows_h<WN, NT, BM, NW, NTH, HRES, 3>(acc, p, m0, n0, wm, wn, r, h, wave, lane, smem, 1.0f);
        return;
    } else {
        // ---- plain layers: out = act(acc + bias), or the folded LayerNorm  act(rstd acc + lt)  (A is the CENTRED image of x);
        // every wave transposes its 32 x (32 NT) strip through its own LDS slab and stores whole row segments - fp32, or the X
        // image of the output (p.out_h: the split is done here, once, for every consumer tile)
        constexpr size_t RING = (size_t)NSTAGE * T::STAGE_BYTES;
        constexpr auto slab_bytes = [](int nth) { return (size_t)NW * 32 * (32 * nth + 4) * sizeof(float); };
        constexpr int NTH = slab_bytes(NT) <= RING ? NT : (NT > 4 && slab_bytes(4) <= RING ? 4 : (slab_bytes(2) <= RING ? 2 : 1));
        static_assert(slab_bytes(NTH) <= RING, "epilogue slabs must fit the ring");
        constexpr int ST = 32 * NTH + 4;    // slab row stride (floats): + 4 keeps 16-byte alignment and shifts the banks per row
        float* const slab = smem + wave * 32 * ST;
        const int64_t mw = m0 + wm * 32;
        const int64_t m = mw + r;
        const int64_t mm = m < p.M ? m : p.M - 1;
        const float rstd = p.ln_in ? p.ln_in[2 * mm + 1] : 1.0f;
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
                        if (p.ln_in) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaf(rstd, acc[nt0 + j][4 * q + e], b4[e]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[nt0 + j][4 * q + e] + b4[e];
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
                // one sub-block of 8 columns per lane: 16 bytes in each of the three slice planes of its chunk
                const int spr = 4 * nth;                            // sub-blocks per row of this pass
#pragma unroll
                for (int it = 0; it < (32 * 4 * NTH + 63) / 64; ++it) {
                    const int idx = it * 64 + lane;
                    const int row = idx / spr, sb = idx % spr;
                    if (idx < 32 * spr && mw + row < p.M) {
                        const f32x4 lo4 = *reinterpret_cast<const f32x4*>(slab + row * ST + 8 * sb);
                        const f32x4 hi4 = *reinterpret_cast<const f32x4*>(slab + row * ST + 8 * sb + 4);
                        xsplit_store8(p.out_h + (size_t)(mw + row) * p.N * 6, ncol0 + 8 * sb, lo4, hi4);
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
__global__ void __launch_bounds__(WM* WN * 64, MINW) xgemm_kernel(const GemmParams p) {
    PAFUSE_XQ_GUARD();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    xgemm_tile<WM, WN, NT, EPI, NSTAGE, BKC, HRES>(p, blockIdx.x, gridDim.x, smem);
}


// ----------------------------------------------------------------------------------------------------------------
// qkv projection + attention in ONE kernel, X pipeline:
//   o[rows of the tile, heads of the workgroup] = softmax(q k^T * scale) v,   (q | k | v) = A[rows] @ W_head^T + b_head
// (common/mixste.py:65-79; hgemm.hpp hfqa_kernel's decomposition: a tile of whole sequences x HPW heads, q | k | v of the tile
// as fp32 LDS tiles, attention from LDS on v_mfma_f32_16x16x4_f32, o written once as the image the proj GEMM reads).
// The projection is xgemm_mainloop on the GATHERED rows of the tile (temporal blocks: rows J apart) against the 3 DP rows
// per head of the HEAD-MAJOR image (q_h, k_h, v_h, each zero-padded from d to DP rows): 32x32x16 MFMAs on 16-deep chunks, a
// two-stage ring.  3 DP must be a multiple of 32 columns per workgroup: one head at DP = 32 (96 columns) or two (192), two
// heads at DP = 48 (288 columns: nine column blocks, 144 accumulator registers - two workgroups per CU still fit).
// Row-per-lane accumulators (lane (r, h): token r of the wave's strip, columns 8 q + 4 h + {0..3} per 32-column block) go to
// the LDS tiles head by head; an 8-column group never straddles q | k | v or two heads (DP % 8 == 0).
// LDS: max(ring, three [ROWS][DP + 4] tiles) + the workgroup's bias vector behind them.
// ----------------------------------------------------------------------------------------------------------------
template <int LP, int DP, int HPW>
struct XfqaTile {
    static constexpr int NWV = LP == 80 ? 5 : 4, TROWS = 32 * NWV, NTHR = 64 * NWV;
    static_assert((HPW * 3 * DP) % 32 == 0, "whole 32-column blocks per workgroup");
    static constexpr int NT = HPW * 3 * DP / 32, NCOL = HPW * 3 * DP;
    using T = XTile<NWV, 1, NT, 16>;
    static constexpr int NSTAGE = 2, RING = NSTAGE * T::STAGE_BYTES;
    static constexpr int LDV = DP + 4, ROWS = TROWS + (LP == 48 ? 4 : 0);
    static constexpr int QKV_BYTES = 3 * ROWS * LDV * 4;
    static constexpr int BIAS_OFF = RING > QKV_BYTES ? RING : QKV_BYTES;    // (bytes; 16-aligned)
    static constexpr int LDS_BYTES = BIAS_OFF + NCOL * 4;
    static_assert(BIAS_OFF % 16 == 0, "alignment of the bias vector");
};

// phase 3 of the fused kernels (the attention from the LDS tiles) lives in hgemm.hpp: fqa_attention_from_lds<.., NSLICE>

template <int LP, int DP, int HPW>
__global__ void __launch_bounds__((XfqaTile<LP, DP, HPW>::NTHR), 2) xfqa_kernel(const FqaParams fp) {
    PAFUSE_XQ_GUARD();
    using FT = XfqaTile<LP, DP, HPW>;
    constexpr int NT = FT::NT, LDV = FT::LDV, ROWS = FT::ROWS, NWV = FT::NWV, TROWS = FT::TROWS, NCOL = FT::NCOL;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const GemmParams& p = fp.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int L = fp.L, NSEQ = fp.nseq_tile;
    const int64_t ntiles = (fp.nseq + NSEQ - 1) / NSEQ;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int hgroups = fp.heads / HPW;                      // workgroups per tile
    const int64_t tile = (int64_t)(idx / hgroups) * 8 + xcd;
    const int head0 = (idx % hgroups) * HPW;                 // this workgroup's heads: head0 .. head0 + HPW - 1
    if (tile >= ntiles) return;   // (workgroup-uniform: the grid is padded to a multiple of 8 tiles)
    PAFUSE_STAMP(0);
    const int K = p.K;
    const int64_t seq0 = tile * NSEQ;
    const int64_t last_seq = fp.nseq - 1;
    // token (row of A / o) of tile row i: sequence seq0 + i / L, position i % L; rows of absent sequences alias the last one
    const uint32_t grp = (uint32_t)fp.group, grp_stride = (uint32_t)fp.group_stride, sq_stride = (uint32_t)fp.seq_stride, tk_stride = (uint32_t)fp.tok_stride;
    auto token_of = [&](int i) -> int64_t {
        int sl = i / L, t = i - sl * L;
        if (sl >= NSEQ) sl = NSEQ - 1, t = L - 1;
        int64_t sq64 = seq0 + sl;
        if (sq64 > last_seq) sq64 = last_seq;
        const uint32_t sq = (uint32_t)sq64, gi = sq / grp;
        return (int64_t)(gi * grp_stride + (sq - gi * grp) * sq_stride + (uint32_t)t * tk_stride);
    };
    const int n0 = head0 * 3 * DP;
    // what the phases behind the projection need from memory, asked for now (the loads land under the first DMA wait):
    // this thread's quad of the workgroup's bias vector, the LayerNorm factor of this lane's row
    f32x4 bias_q = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < NCOL / 4) bias_q = *reinterpret_cast<const f32x4*>(p.bias + n0 + 4 * tid);
    float rstd = 1.0f;
    if (p.ln_in) rstd = p.ln_in[2 * token_of(32 * wave + r) + 1];

    // ---- phase 1: the projection
    f32x16 acc[NT];
    xgemm_mainloop<NWV, 1, NT, FT::NSTAGE, 16>(
        acc, [&](int row) { return p.Ah + (size_t)token_of(row) * K * 6; }, p.Wh + (size_t)n0 * K * 6, K, smem, wave, lane, p);
    PAFUSE_STAMP(1);
    float* const bias_s = smem + FT::BIAS_OFF / 4;     // behind the ring and the tiles: written once, read by every head's phase 2
    if (tid < NCOL / 4) *reinterpret_cast<f32x4*>(bias_s + 4 * tid) = bias_q;

    float* const Qs = smem;                    // [ROWS][LDV] each: q | k | v of one head
    const int row = 32 * wave + r;
    static_for<HPW>([&](auto HH) {
        constexpr int hh = decltype(HH)::value;
        __syncthreads();   // every wave is done with the ring / the tiles of the head before (and the bias vector is in place)
        // ---- phase 2: this head's columns of the accumulators (bias or the folded LayerNorm applied) -> its q | k | v tiles
        static_for<NT * 4>([&](auto NQ) {
            constexpr int nt = decltype(NQ)::value / 4, q = decltype(NQ)::value % 4;
            constexpr int col8 = 32 * nt + 8 * q;                   // the group's first column (of the workgroup's NCOL)
            if constexpr (col8 / (3 * DP) == hh) {
                constexpr int c3 = col8 - hh * 3 * DP, part = c3 / DP, cc0 = c3 - part * DP;
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_s + col8 + 4 * h);
                f32x4 v;
                if (p.ln_in) {   // centred A: no mean term
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(rstd, acc[nt][4 * q + e], b4[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[nt][4 * q + e] + b4[e];
                }
                *reinterpret_cast<f32x4*>(Qs + part * ROWS * LDV + row * LDV + cc0 + 4 * h) = v;
            }
        });
        if constexpr (ROWS > TROWS)   // LP = 48: the key / value tiles of the last sequence reach 4 rows past the tile - keep them finite
            for (int i = tid; i < 3 * (ROWS - TROWS) * LDV; i += FT::NTHR) {
                const int part = i / ((ROWS - TROWS) * LDV), rem = i % ((ROWS - TROWS) * LDV);
                Qs[part * ROWS * LDV + TROWS * LDV + rem] = 0.f;
            }
        __syncthreads();
        // ---- phase 3: attention per (sequence of the tile, 16-query tile) from the tiles; o as the X image
        fqa_attention_from_lds<LP, DP, NWV, ROWS, 3>(fp, smem, seq0, head0 + hh, wave, lane & 15, lane >> 4, token_of);
    });
    PAFUSE_STAMP(2);
}

}  // namespace pafuse
