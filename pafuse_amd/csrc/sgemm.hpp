// sgemm.hpp - "strip" GEMM of the split-precision (bf16x3) mode, round 6: one kernel structure for all four linear layers of a
// transformer block (common/mixste.py:37-43 Mlp.forward, :63-82 Attention.forward's qkv / proj, :113-116 Block.forward's
// residual adds and LayerNorms), on v_mfma_f32_16x16x32_bf16.
//
//   out = epilogue(A[M,K] @ W'[N,K]^T)      A fp32 rows (split into three bf16 slices in registers), W' the pre-split M16 image
//
// What is different from gemm16_tile / gemm_dma_tile (kernels.hpp):
//   * Every wave owns RG groups of 16 token rows and ALL BN = 16 NB columns of the tile ("strip"): an A fragment is read and
//     split exactly once per workgroup, a token's whole output row of the tile sits in four lanes of one wave (LayerNorm
//     statistics = in-lane sums + two xor shuffles, no cross-wave exchange), and the A rows are PRIVATE to the wave - each wave
//     DMAs its own rows, so only the W' stage needs the workgroup barrier.
//   * Both operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4) in full 128-byte lines into an NSTAGE ring behind a
//     counted vmcnt, ONE raw s_barrier per 32-deep chunk.  The A stage is XOR-swizzled on the DMA's source address so that the
//     16x16x32 fragment reads (lane (c, qd): row c, 16-byte chunks 2 qd, 2 qd + 1) are conflict-free ds_read_b128:
//       chunk ch of row r sits at position ch ^ f((r >> 1) & 7),  f(x) = x ^ 2 for x in 2..5, else x
//     (the four 16-lane groups a ds_read_b128 is served in each hold 8 rows with qd = q and 8 with qd = q + 1: f maps the rows
//     of either kind onto {0, 1, 6, 7}, so chunks 2q and 2q + 2 land on disjoint position sets; tools/lds_layout_check.py replays it).
//   * PERSISTENT tile stream: a workgroup runs tiles b, b + G, b + 2G, ... (G = gridDim.x) as ONE stream of K chunks through the
//     ring - the first chunks of the next tile are in flight while the last chunks of this one are multiplied and while its
//     epilogue runs, so a tile pays no prologue latency of its own.
// Arithmetic per product is gemm16_tile's (six bf16 products per 32-deep step into one fp32 accumulator, small terms first).
#pragma once
#include "kernels.hpp"

namespace pafuse {

template <int NB, int RG, int NW, int NSTAGE>
struct StripTile {
    static_assert(NSTAGE >= 2 && NSTAGE <= 4, "ring depth");
    static constexpr int NTHR = NW * 64, BM = NW * RG * 16, BN = NB * 16;
    static constexpr int A_WAVE = RG * 2048;                                  // bytes of one wave's A rows in a stage
    static constexpr int A_BYTES = NW * A_WAVE, W_BYTES = BN * WSPLIT_ROW_BYTES, STAGE_BYTES = A_BYTES + W_BYTES;
    static constexpr int IA = RG * 2;                                         // A DMA instructions per wave and chunk (its own rows)
    static constexpr int IWT = W_BYTES / 1024, IW = (IWT + NW - 1) / NW;      // W' DMA instructions: total, per wave
    static constexpr int CNT = IA + IW;                                       // per wave and chunk (uniform: surplus slots re-issue the last)
    static constexpr size_t LDS_BYTES = (size_t)NSTAGE * STAGE_BYTES;
    static_assert(W_BYTES % 1024 == 0, "whole DMA pieces");
    // epilogue stores counted as "younger than the DMAs" by the first waits behind an epilogue (any lower bound is safe)
    static constexpr int NSTC = RG * NB < 63 - CNT * (NSTAGE - 2) ? RG * NB : 63 - CNT * (NSTAGE - 2);
    static_assert(CNT * (NSTAGE - 2) >= 0 && CNT * (NSTAGE - 2) < 56, "vmcnt range");
};

// position swizzle of the A stage (see the header): x = (row >> 1) & 7
__device__ __forceinline__ int strip_f(int x) { return x ^ ((((x + 2) >> 2) & 1) << 1); }

enum { SEPI_BIAS = 0 };   // (a whole-row form, SEPI_ROWLN, was built and measured in round 6 - it ties the LDS-DMA whole-row kernels and is
//                           not part of the library: profiles/r06_whole_row_strip_experiment.patch, profiles/r06_sgemm_whole_row_strip_experiment.log)

// -DSGEMM_STAMPS (tools/sgemm_bench.hip only): per wave, shader cycles summed over its stream in four phases - [0] waiting at the top of
// a chunk (counted vmcnt + barrier), [1] A fragment reads + split, [2] the MFMA groups with the refill DMA in their shadow, [3] epilogues -
// then [4] the wave's lifetime in cycles and [5] in 100 MHz ticks (s_memrealtime): the clock the chip held.  GemmParams.stamps[wave][8].
#ifdef SGEMM_STAMPS
__device__ __forceinline__ unsigned long long sgemm_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define SGEMM_T(v) const unsigned long long v = sgemm_stamp()
#define SGEMM_ADD(i, a, b) st_acc[i] += (b) - (a)
#else
#define SGEMM_T(v)
#define SGEMM_ADD(i, a, b)
#endif

// ----------------------------------------------------------------------------------------------------------------
// sgemm2_kernel - the strip GEMM with the two per-tile costs that the stamps of its plain first form show (sgemm_kernel, kept for
// the A/B in tools/sgemm_v1.hpp; tools/sgemm_bench.hip, profiles/r06_sgemm_v1_stamps_and_ablations.log: the in-register split of the A
// fragment 23 % of a chunk, the epilogue 18 - 27 % of a tile) taken off a wave's critical path:
//   * A runs ONE CHUNK AHEAD of W' through the same two-stage ring.  A rows are private to their wave, so the fragment of chunk
//     g + 1 is read (own vmcnt, no barrier) at the top of chunk g and split into its three bf16 slices in the gaps of chunk g's
//     MFMAs, one SplitPair stage per gap (hand-placed, order pinned through data as in gemm_dma_tile); its LDS slot is then free and
//     the DMA of chunk g + 3 goes into it mid-chunk.  W' of chunk g + 1 is issued in the first gaps of chunk g.  In flight at the top of a
//     chunk: only the A rows two chunks ahead - `s_waitcnt vmcnt(IA)`.
//   * DEFERRED STORES: at the end of a tile the epilogue arithmetic (bias / folded LayerNorm / GELU) runs at once into a second
//     register set `fin`, the accumulators restart at zero, and the 1 KiB store instructions of `fin` are issued four per chunk inside
//     the gaps of the NEXT tile's first chunks (the last tile of the stream flushes at the end).  A wave never sits in a store queue.
// Same products in the same order as sgemm_kernel / gemm16_tile: equal bits.
// ----------------------------------------------------------------------------------------------------------------
#define SGEMM_PIN_ACC(A) asm volatile("" : "+v"(A))
#define SGEMM_PIN_PAIR(P) asm volatile("" : "+v"((P).x0), "+v"((P).x1), "+v"((P).s0), "+v"((P).s1), "+v"((P).s2))

// FLAGS (compile-time, so that the epilogue is straight-line code with every load issued up front): bit 0 = folded LayerNorm
// (p.ln_in), bit 1 = GELU (p.act); training's two epilogue options (GemmParams): bit 2 = p.dact_u (out = (acc + bias) * gelu'(dact_u): the
// dX GEMM of fc2), bit 3 = p.out_act (`out` receives the pre-activation u - stored at the end of the tile -, `out_act` gelu(u) - the
// deferred stores: fc1 forward)
template <int NB, int RG, int NW, int EPI, int MINW, int FLAGS>
__global__ void __launch_bounds__(NW * 64, MINW) sgemm2_kernel(const GemmParams p) {
    constexpr bool LNIN = (FLAGS & 1) != 0, ACT = (FLAGS & 2) != 0, DACT = (FLAGS & 4) != 0, OUTACT = (FLAGS & 8) != 0;
    float* const fin_base = OUTACT ? p.out_act : p.out;   // where the deferred stores go
    PAFUSE_XQ_GUARD();
    using T = StripTile<NB, RG, NW, 2>;
    constexpr int BM = T::BM, BN = T::BN, IA = T::IA, IW = T::IW, IWT = T::IWT;
    constexpr int TOT = NB * RG * 6;          // MFMAs (= filler slots) of a chunk
    constexpr int NSPLIT = RG * 4 * 5;        // split stage-steps of one chunk's A fragments (4 pairs x 5 stages per row group)
    constexpr int SPC = 4;                    // deferred stores per chunk
    constexpr int NSTORE = RG * NB, KST = (NSTORE + SPC - 1) / SPC;   // chunks of the next tile that carry them (host: nk >= KST)
    static_assert(TOT >= 48, "filler schedule needs room");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint8_t* const lds = reinterpret_cast<uint8_t*>(smem);
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, qd = lane >> 4;

    const int tiles_n = p.N / BN;
    const int tiles_m = (int)((p.M + BM - 1) / BM);
    const int ntiles = tiles_m * tiles_n;
    const int b = blockIdx.x, G = gridDim.x;
    if (b >= ntiles) return;
    const int my_tiles = (ntiles - b + G - 1) / G;
    const int K = p.K, nk = K / 32;
    const int total = my_tiles * nk;
    auto tile_of = [&](int j, int& tm, int& tn) {
        const int v = b + j * G;
        const int xcd = v & 7, q = ntiles >> 3, rem = ntiles & 7;
        const int tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (v >> 3);
        tm = tile / tiles_n, tn = tile % tiles_n;
    };

    // ---- the two DMA cursors
    const int64_t ws_chunk = (int64_t)p.N * WSPLIT_ROW_BYTES;
    int wj = 0, wkc = 0, wg = 0;     // W': next chunk to issue (tile wj of mine, chunk wkc, stream index wg)
    int aj = 0, akc = 0, ag = 0;     // A : likewise
    int a_off[IA];
    const uint8_t* w_src = nullptr;
    auto w_setup = [&]() {
        int tm, tn;
        tile_of(wj, tm, tn);
        w_src = p.Wsplit + (int64_t)tn * BN * WSPLIT_ROW_BYTES + lane * 16;
    };
    auto a_setup = [&]() {
        int tm, tn;
        tile_of(aj, tm, tn);
        const int64_t m0 = (int64_t)tm * BM;
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const int64_t row = m0 + wave * (RG * 16) + 8 * i + (lane >> 3);
            const int x = (4 * (i & 1) + (lane >> 4)) & 7;
            const int ch = (lane & 7) ^ strip_f(x);
            a_off[i] = (int)((row < p.M ? row : p.M - 1) * K) + 4 * ch;
        }
    };
    auto issue_w = [&](int j) {   // W' DMA instruction slot j (0 .. IW - 1) of the chunk under the W' cursor
        uint8_t* const sa = lds + (wg & 1) * T::STAGE_BYTES + T::A_BYTES;
        int i = wave + j * NW;
        i = i < IWT ? i : IWT - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src + wkc * ws_chunk + i * 1024),
                                         (__attribute__((address_space(3))) void*)(sa + i * 1024), 16, 0, 0);
    };
    auto issue_a = [&](int j) {   // A DMA instruction j (0 .. IA - 1) of the chunk under the A cursor
        uint8_t* const sa = lds + (ag & 1) * T::STAGE_BYTES + wave * T::A_WAVE;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.A + akc * 32 + a_off[j]),
                                         (__attribute__((address_space(3))) void*)(sa + j * 1024), 16, 0, 0);
    };
    auto w_advance = [&]() {
        ++wg;
        if (++wkc == nk) {
            wkc = 0, ++wj;
            if (wj < my_tiles) w_setup();
        }
    };
    auto a_advance = [&]() {
        ++ag;
        if (++akc == nk) {
            akc = 0, ++aj;
            if (aj < my_tiles) a_setup();
        }
    };

    const int fc = strip_f((c >> 1) & 7);
    const uint32_t a_frag = (uint32_t)(wave * T::A_WAVE + c * 128 + (((2 * qd) ^ fc) * 16));
    const uint32_t w_frag = (uint32_t)(T::A_BYTES + c * WSPLIT_ROW_BYTES + wsplit_sub_offset<32, 1>(c, qd));

    f32x4 acc[RG][NB], fin[RG][NB];
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[g][n] = f32x4{0.f, 0.f, 0.f, 0.f}, fin[g][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    int fin_off[RG];        // float offset in p.out of the pending tile's row stores (+ 16 n); negative: the lane's row is past M (no store)
#pragma unroll
    for (int g = 0; g < RG; ++g) fin_off[g] = -1;
    bool pend = false;

    u32x4 cur[RG][3];       // the three slices of the current chunk's A fragments
    SplitPair sp[RG][4];    // the next chunk's, being split

    // ---- prologue: A(0), A(1), W'(0); split A(0); then A(2) into the slot A(0) left
    w_setup();
    a_setup();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int j = 0; j < IA; ++j) issue_a(j);
        a_advance();
    }
#pragma unroll
    for (int j = 0; j < IW; ++j) issue_w(j);
    w_advance();
    wait_vmcnt<0>();
    {
        u32x4 lo[RG], hi[RG];
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            lo[g] = lds_read128<0>(lds0 + a_frag + g * 2048);
            hi[g] = lds_read128<0>(lds0 + (a_frag ^ 16u) + g * 2048);
        }
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[g]), "+v"(hi[g]));
            const bf16x8x3 s3 = split3(__builtin_bit_cast(f32x4, lo[g]), __builtin_bit_cast(f32x4, hi[g]));
            cur[g][0] = __builtin_bit_cast(u32x4, s3.s0), cur[g][1] = __builtin_bit_cast(u32x4, s3.s1), cur[g][2] = __builtin_bit_cast(u32x4, s3.s2);
        }
    }
    if (ag < total) {
#pragma unroll
        for (int j = 0; j < IA; ++j) issue_a(j);
        a_advance();
    }

#ifdef SGEMM_STAMPS
    unsigned long long st_acc[4] = {0, 0, 0, 0};
    const unsigned long long st_begin = sgemm_stamp();
    const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime();
#endif
    int g_idx = 0;
    for (int tj = 0; tj < my_tiles; ++tj) {
        int tm, tn;
        tile_of(tj, tm, tn);
        const int64_t m0 = (int64_t)tm * BM;
        const int n0 = tn * BN;
        for (int kc = 0; kc < nk; ++kc, ++g_idx) {
            // in flight at most: the A rows of chunk g_idx + 2 (issued mid-chunk g_idx - 1, behind W' of this chunk).  Deferred stores are
            // not counted: an operation assumed absent only makes the wait stricter.
            SGEMM_T(t0);
            wait_vmcnt<IA>();               // (past the end of the stream the cursors keep issuing: re-fetches of valid rows into free slots)
            __builtin_amdgcn_s_barrier();   // W' of this chunk visible to every wave; every wave is done reading W' of the last one
            SGEMM_T(t1);
            SGEMM_ADD(0, t0, t1);
            const bool st_now = pend && kc < KST;       // this chunk carries SPC of the pending tile's stores
            const uint32_t sbase = lds0 + (uint32_t)((g_idx & 1) * T::STAGE_BYTES);
            const uint32_t snext = lds0 + (uint32_t)(((g_idx + 1) & 1) * T::STAGE_BYTES);

            __builtin_amdgcn_s_setprio(1);
            u32x4 n_lo[RG], n_hi[RG];
#pragma unroll
            for (int g = 0; g < RG; ++g) {   // (stale bytes on the last chunk of the stream: never used)
                n_lo[g] = lds_read128<0>(snext + a_frag + g * 2048);
                n_hi[g] = lds_read128<0>(snext + (a_frag ^ 16u) + g * 2048);
            }
            u32x4 wf[2][3];
            auto load_w = [&](auto N_) {
                constexpr int n = decltype(N_)::value;
                constexpr int off = n * 16 * WSPLIT_ROW_BYTES;
                static_assert(off + 32 < 65536, "ds_read immediate");
                const uint32_t addr = sbase + w_frag;
                wf[n & 1][0] = lds_read128<off>(addr);
                wf[n & 1][1] = lds_read128<off + 16>(addr);
                wf[n & 1][2] = lds_read128<off + 32>(addr);
            };
            load_w(std::integral_constant<int, 0>{});

            static_for<NB>([&](auto N_) {
                constexpr int n = decltype(N_)::value;
                if constexpr (n + 1 < NB) {
                    load_w(std::integral_constant<int, n + 1>{});
                    if constexpr (n == 0) {   // the A reads and W' fragment 0 are older than the three reads just issued
                        if constexpr (RG == 2)
                            asm volatile("s_waitcnt lgkmcnt(3)"
                                         : "+v"(n_lo[0]), "+v"(n_hi[0]), "+v"(n_lo[RG - 1]), "+v"(n_hi[RG - 1]), "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(wf[0][2]));
                        else
                            asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(n_lo[0]), "+v"(n_hi[0]), "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(wf[0][2]));
#pragma unroll
                        for (int g = 0; g < RG; ++g)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                sp[g][q].x0 = __builtin_bit_cast(float, q < 2 ? n_lo[g][2 * q] : n_hi[g][2 * q - 4]);
                                sp[g][q].x1 = __builtin_bit_cast(float, q < 2 ? n_lo[g][2 * q + 1] : n_hi[g][2 * q - 3]);
                            }
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(wf[n & 1][0]), "+v"(wf[n & 1][1]), "+v"(wf[n & 1][2]));
                    }
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[n & 1][0]), "+v"(wf[n & 1][1]), "+v"(wf[n & 1][2]));
                }
                const u32x4(&w)[3] = wf[n & 1];
                static_for<RG * 6>([&](auto S_) {
                    constexpr int s = decltype(S_)::value, g = s / 6, pi = s % 6;
                    constexpr int t = n * RG * 6 + s;   // this MFMA's index in the chunk = the filler slot behind it
                    // small terms first, the leading product last: (w0,a2) (w2,a0) (w1,a1) (w0,a1) (w1,a0) (w0,a0)
                    constexpr int wi = pi == 0 ? 0 : pi == 1 ? 2 : pi == 2 ? 1 : pi == 3 ? 0 : pi == 4 ? 1 : 0;
                    constexpr int ai = pi == 0 ? 2 : pi == 1 ? 0 : pi == 2 ? 1 : pi == 3 ? 1 : pi == 4 ? 0 : 0;
                    SGEMM_PIN_ACC(acc[g][n]);
                    acc[g][n] = mfma16_bf16_k32(__builtin_bit_cast(bf16x8, w[wi]), __builtin_bit_cast(bf16x8, cur[g][ai]), acc[g][n]);
                    // ---- the filler of slot t
                    // (i) W' of the next chunk: piece j behind MFMA 1 + 4 j
                    if constexpr (t % 4 == 1 && t / 4 < IW) {
                        asm volatile("" ::: "memory");
                        issue_w(t / 4);
                        asm volatile("" ::: "memory");
                    }
                    // (ii) one split stage-step of the next chunk's A per slot from slot RG * 6 on (the fragment reads were waited for in
                    //      front of group 0's MFMAs), spread over the rest of the chunk
                    constexpr int T0 = RG * 6, SPAN = TOT - T0 - 2;
                    if constexpr (t >= T0) {
                        // step k runs in slot T0 + floor(k * SPAN / NSPLIT): the steps of this slot
                        constexpr int k_lo = ((t - T0) * NSPLIT + SPAN - 1) / SPAN;          // first k with slot(k) >= t
                        constexpr int k_hi = ((t - T0 + 1) * NSPLIT + SPAN - 1) / SPAN;      // first k with slot(k) >= t + 1
                        static_for<(k_hi < NSPLIT ? k_hi : NSPLIT) - (k_lo < NSPLIT ? k_lo : NSPLIT)>([&](auto K_) {
                            constexpr int k = k_lo + decltype(K_)::value;
                            constexpr int stage = k / (RG * 4), pr = k % (RG * 4), sg = pr / 4, sq = pr % 4;   // stage-major: independent pairs back to back
                            SGEMM_PIN_PAIR(sp[sg][sq]);
                            sp[sg][sq].template stage<stage>();
                            SGEMM_PIN_PAIR(sp[sg][sq]);
                        });
                    }
                    // (iii) the A rows three chunks ahead, into the slot whose fragment reads completed in front of group 0
                    if constexpr (t >= TOT / 2 && t < TOT / 2 + 3 * IA && (t - TOT / 2) % 3 == 0) {
                        asm volatile("" ::: "memory");
                        issue_a((t - TOT / 2) / 3);
                        asm volatile("" ::: "memory");
                    }
                    // (iv) SPC deferred stores of the previous tile, spread over the chunk
                    if constexpr (t % (TOT / SPC) == TOT / SPC - 3) {
                        constexpr int i = t / (TOT / SPC);
                        if (st_now) {
                            static_for<KST>([&](auto C_) {
                                constexpr int cs = decltype(C_)::value, si = cs * SPC + i;
                                if constexpr (si < NSTORE) {
                                    constexpr int sg = si / NB, sn = si % NB;
                                    if (kc == cs && fin_off[sg] >= 0) *reinterpret_cast<f32x4*>(fin_base + fin_off[sg] + 16 * sn) = fin[sg][sn];
                                }
                            });
                        }
                    }
                });
            });
            __builtin_amdgcn_s_setprio(0);
            w_advance();
            a_advance();
#pragma unroll
            for (int g = 0; g < RG; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) cur[g][0][q] = sp[g][q].s0, cur[g][1][q] = sp[g][q].s1, cur[g][2][q] = sp[g][q].s2;
            SGEMM_T(t3);
            SGEMM_ADD(2, t1, t3);
        }
        SGEMM_T(t4);

        // ---- end of tile tj: the epilogue arithmetic into `fin` (lane (c, qd): token m0 + wave RG 16 + 16 g + c, columns n0 + 16 n + 4 qd + ..)
        static_assert(EPI == SEPI_BIAS, "the plain form is the one in the library");
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            const int64_t m = m0 + wave * (RG * 16) + 16 * g + c;
            const bool live = m < p.M;
            const int64_t mm = live ? m : p.M - 1;
            float rstd = 1.0f;
            if constexpr (LNIN) rstd = p.ln_in[2 * mm + 1];
            fin_off[g] = live ? (int)(m * p.N) + n0 + 4 * qd : -1;   // (M N < 2^31 floats: checked by the host)
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + n0 + 16 * n + 4 * qd);
                f32x4 v;
                if constexpr (LNIN) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(rstd, acc[g][n][e], b4[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[g][n][e] + b4[e];
                }
                if constexpr (ACT) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                }
                if constexpr (DACT) {   // the gradient reaches the pre-activation in the same pass
                    const f32x4 u = *reinterpret_cast<const f32x4*>(p.dact_u + mm * p.N + n0 + 16 * n + 4 * qd);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= gelu_erf_grad(u[e]);
                }
                if constexpr (OUTACT) {   // `out` keeps the pre-activation (the backward needs both), `out_act` its GELU
                    if (live) *reinterpret_cast<f32x4*>(p.out + mm * p.N + n0 + 16 * n + 4 * qd) = v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                }
                fin[g][n] = v;
                acc[g][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        pend = true;
        SGEMM_T(t5);
        SGEMM_ADD(3, t4, t5);
    }
    // ---- the last tile's stores (and the cursors' surplus DMAs must have landed before the workgroup's LDS is released)
    wait_vmcnt<0>();
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int n = 0; n < NB; ++n)
            if (fin_off[g] >= 0) *reinterpret_cast<f32x4*>(fin_base + fin_off[g] + 16 * n) = fin[g][n];
#ifdef SGEMM_STAMPS
    if (p.stamps && lane == 0) {
        unsigned long long* o = p.stamps + ((size_t)blockIdx.x * NW + wave) * 8;
        o[0] = st_acc[0], o[1] = st_acc[1], o[2] = st_acc[2], o[3] = st_acc[3];
        o[4] = sgemm_stamp() - st_begin, o[5] = __builtin_amdgcn_s_memrealtime() - rt_begin;
    }
#endif
}

}  // namespace pafuse
