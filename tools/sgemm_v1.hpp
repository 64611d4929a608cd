// sgemm_v1.hpp (tools only) - the FIRST form of the round-6 strip GEMM: the same tile geometry, LDS-DMA ring and persistent tile stream as
// pafuse_amd/csrc/sgemm.hpp's sgemm2_kernel, but the A fragment is read and split at the top of its own chunk and the epilogue runs
// (and stores) at the end of its tile.  Kept for tools/sgemm_bench.hip: its stamps and ablation builds (-DSGEMM_ABL=1 no split, 2 no
// epilogue, 3 neither) are the measurement behind the software pipeline of sgemm2_kernel (profiles/r06_sgemm_v1_stamps_and_ablations.log).
#pragma once
#include "../pafuse_amd/csrc/sgemm.hpp"

namespace pafuse {

// EPI = SEPI_BIAS:  out = act(acc + bias)  |  act(rstd * acc + lt)  with the LayerNorm folded (p.ln_in; A is the centred row)
// EPI = SEPI_ROWLN (BN == p.N): the whole-row chain of GemmParams (inference forms): y = acc + bias + resid; post LayerNorm; + pos;
//       then statistics + centred store (folded), or the next LayerNorm -> out_n, or the head.
template <int NB, int RG, int NW, int NSTAGE, int EPI, int MINW, int FLAGS>
__global__ void __launch_bounds__(NW * 64, MINW) sgemm_kernel(const GemmParams p) {
    constexpr bool LNIN = (FLAGS & 1) != 0, ACT = (FLAGS & 2) != 0;   // compile-time: the epilogue is straight-line code
    PAFUSE_XQ_GUARD();
    using T = StripTile<NB, RG, NW, NSTAGE>;
    constexpr int BM = T::BM, BN = T::BN, IA = T::IA, IW = T::IW, IWT = T::IWT, CNT = T::CNT;
    constexpr int NST = T::NSTC;   // dwordx4 stores per lane that the epilogue of a FULL tile issues at least, in every wave
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
    const int total = my_tiles * nk;   // chunks of this workgroup's stream
    // XCD-aware order (kernels.hpp): virtual workgroup v = b + j G of `ntiles` (G is a multiple of 8 or >= ntiles, so v & 7 == b & 7)
    auto tile_of = [&](int j, int& tm, int& tn) {
        const int v = b + j * G;
        const int xcd = v & 7, q = ntiles >> 3, rem = ntiles & 7;
        const int tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (v >> 3);
        tm = tile / tiles_n, tn = tile % tiles_n;
    };

    // ---- DMA side: the issue cursor (tile index ij, chunk ikc) runs NSTAGE - 1 chunks ahead of the compute cursor
    const int64_t ws_chunk = (int64_t)p.N * WSPLIT_ROW_BYTES;
    int ij = 0, ikc = 0, ig = 0;     // next chunk to issue: tile ij of mine, chunk ikc; ig = its index in the stream
    int a_off[IA];                   // float offset of this lane's source in A instruction i (row base + swizzled chunk), current issue tile
    const uint8_t* w_src = nullptr;  // this lane's source of W' instruction 0, chunk 0, current issue tile
    auto issue_tile_setup = [&]() {
        int tm, tn;
        tile_of(ij, tm, tn);
        const int64_t m0 = (int64_t)tm * BM;
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const int64_t row = m0 + wave * (RG * 16) + 8 * i + (lane >> 3);
            const int x = (4 * (i & 1) + (lane >> 4)) & 7;
            const int ch = (lane & 7) ^ strip_f(x);
            a_off[i] = (int)((row < p.M ? row : p.M - 1) * K) + 4 * ch;   // rows past the last token read a valid row (never stored)
        }
        w_src = p.Wsplit + (int64_t)tn * BN * WSPLIT_ROW_BYTES + lane * 16;
    };
    auto issue_piece = [&](int j) {   // DMA instruction slot j (0 .. CNT - 1) of the chunk under the issue cursor
        uint8_t* const sa = lds + (ig % NSTAGE) * T::STAGE_BYTES;
        if (j < IA) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.A + ikc * 32 + a_off[j]),
                                             (__attribute__((address_space(3))) void*)(sa + wave * T::A_WAVE + j * 1024), 16, 0, 0);
        } else {
            int i = wave + (j - IA) * NW;   // wave-uniform
            i = i < IWT ? i : IWT - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src + ikc * ws_chunk + i * 1024),
                                             (__attribute__((address_space(3))) void*)(sa + T::A_BYTES + i * 1024), 16, 0, 0);
        }
    };
    auto issue_advance = [&]() {
        ++ig;
        if (++ikc == nk) {
            ikc = 0, ++ij;
            if (ij < my_tiles) issue_tile_setup();
        }
    };

    // ---- fragment addresses (bytes inside a stage)
    const int fc = strip_f((c >> 1) & 7);
    const uint32_t a_frag = (uint32_t)(wave * T::A_WAVE + c * 128 + (((2 * qd) ^ fc) * 16));   // + g * 2048; second half at ^ 16
    const uint32_t w_frag = (uint32_t)(T::A_BYTES + c * WSPLIT_ROW_BYTES + wsplit_sub_offset<32, 1>(c, qd));   // + nb * 3072

    f32x4 acc[RG][NB];
    auto zero_acc = [&]() {
#pragma unroll
        for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[g][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    // ---- prologue: the first NSTAGE - 1 chunks of the stream
    issue_tile_setup();
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (ig < total) {
#pragma unroll
            for (int j = 0; j < CNT; ++j) issue_piece(j);
            issue_advance();
        }

#ifdef SGEMM_STAMPS
    unsigned long long st_acc[4] = {0, 0, 0, 0};
    const unsigned long long st_begin = sgemm_stamp();
    const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime();
#endif
    int g_idx = 0;   // compute cursor in the stream
    bool prev_counted = false;   // the previous tile was a full tile: every wave issued its NST epilogue stores (a wave whose rows
    //                              are all past M branches around them, so a ragged tile's stores cannot be counted on)
    for (int tj = 0; tj < my_tiles; ++tj) {
        int tm, tn;
        tile_of(tj, tm, tn);
        const int64_t m0 = (int64_t)tm * BM;
        const int n0 = tn * BN;
        for (int kc = 0; kc < nk; ++kc, ++g_idx) {
            // chunk g_idx has landed once at most the younger operations of this wave are still in flight: the DMAs of the chunks
            // g_idx + 1 .. g_idx + NSTAGE - 2 and - in the first NSTAGE - 1 chunks behind an epilogue - that epilogue's stores
            SGEMM_T(t0);
            const int younger = total - 1 - g_idx;   // chunks of the stream behind this one
            if (younger >= NSTAGE - 2) {
                if (prev_counted && kc < NSTAGE - 1) wait_vmcnt<CNT*(NSTAGE - 2) + NST>();
                else wait_vmcnt<CNT*(NSTAGE - 2)>();
            } else {
                wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();   // W' of chunk g_idx visible to every wave; every wave is done reading chunk g_idx - 1
            SGEMM_T(t1);
            SGEMM_ADD(0, t0, t1);
            const bool refill = ig < total;  // (ig == g_idx + NSTAGE - 1 while there is work left)
            const uint32_t sbase = lds0 + (uint32_t)((g_idx % NSTAGE) * T::STAGE_BYTES);

            __builtin_amdgcn_s_setprio(1);
            u32x4 a_lo[RG], a_hi[RG];
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                a_lo[g] = lds_read128<0>(sbase + a_frag + g * 2048);
                a_hi[g] = lds_read128<0>(sbase + (a_frag ^ 16u) + g * 2048);
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
            bf16x8x3 a[RG];
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                // the reads behind this fragment's pair stay in flight: 2 (RG - 1 - g) of A + 3 of W'
                if (g == 0 && RG == 2) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(a_lo[0]), "+v"(a_hi[0]));
                else asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a_lo[g]), "+v"(a_hi[g]));
#if defined(SGEMM_ABL) && (SGEMM_ABL & 1)   // ablation (results wrong): the raw fragment bits as slices - what the K loop costs without the split
                a[g].s0 = __builtin_bit_cast(bf16x8, a_lo[g]), a[g].s1 = __builtin_bit_cast(bf16x8, a_hi[g]), a[g].s2 = a[g].s0;
#else
                a[g] = split3(__builtin_bit_cast(f32x4, a_lo[g]), __builtin_bit_cast(f32x4, a_hi[g]));
#endif
            }
            SGEMM_T(t2);
            SGEMM_ADD(1, t1, t2);
            static_for<NB>([&](auto N_) {
                constexpr int n = decltype(N_)::value;
                if constexpr (n + 1 < NB) {
                    load_w(std::integral_constant<int, n + 1>{});
                    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(wf[n & 1][0]), "+v"(wf[n & 1][1]), "+v"(wf[n & 1][2]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[n & 1][0]), "+v"(wf[n & 1][1]), "+v"(wf[n & 1][2]));
                }
                const bf16x8 w0 = __builtin_bit_cast(bf16x8, wf[n & 1][0]);
                const bf16x8 w1 = __builtin_bit_cast(bf16x8, wf[n & 1][1]);
                const bf16x8 w2 = __builtin_bit_cast(bf16x8, wf[n & 1][2]);
#pragma unroll
                for (int g = 0; g < RG; ++g) {   // small terms first, the leading product last
                    acc[g][n] = mfma16_bf16_k32(w0, a[g].s2, acc[g][n]);
                    acc[g][n] = mfma16_bf16_k32(w2, a[g].s0, acc[g][n]);
                    acc[g][n] = mfma16_bf16_k32(w1, a[g].s1, acc[g][n]);
                    acc[g][n] = mfma16_bf16_k32(w0, a[g].s1, acc[g][n]);
                    acc[g][n] = mfma16_bf16_k32(w1, a[g].s0, acc[g][n]);
                    acc[g][n] = mfma16_bf16_k32(w0, a[g].s0, acc[g][n]);
                }
                {   // this column block's share of the refill DMA, in the shadow of the MFMAs just issued
                    constexpr int PER = (CNT + NB - 1) / NB, j0 = n * PER, j1 = (n + 1) * PER < CNT ? (n + 1) * PER : CNT;
                    if constexpr (j0 < j1) {
                        asm volatile("" ::: "memory");
                        if (refill) {
#pragma unroll
                            for (int j = j0; j < j1; ++j) issue_piece(j);
                        }
                        asm volatile("" ::: "memory");
                    }
                }
            });
            __builtin_amdgcn_s_setprio(0);
            if (refill) issue_advance();
            SGEMM_T(t3);
            SGEMM_ADD(2, t2, t3);
        }
        SGEMM_T(t4);

        // ---- epilogue of tile tj: lane (c, qd) owns token m0 + wave RG 16 + 16 g + c, columns n0 + 16 n + 4 qd + {0,1,2,3}
#if defined(SGEMM_ABL) && (SGEMM_ABL & 2)   // ablation (results wrong): no epilogue - the accumulators are kept alive, nothing is stored
        {
            float sacc = 0.f;
#pragma unroll
            for (int g = 0; g < RG; ++g)
#pragma unroll
                for (int n = 0; n < NB; ++n) sacc += acc[g][n][0] + acc[g][n][1] + acc[g][n][2] + acc[g][n][3];
            if (sacc == 123.456f) p.out[0] = sacc;
        }
        if constexpr (false) {
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                const int64_t m = m0 + wave * (RG * 16) + 16 * g + c;
#else
        if constexpr (EPI == SEPI_BIAS) {
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                const int64_t m = m0 + wave * (RG * 16) + 16 * g + c;
#endif
                const bool live = m < p.M;
                const int64_t mm = live ? m : p.M - 1;
                float rstd = 1.0f;
                if constexpr (LNIN) rstd = p.ln_in[2 * mm + 1];   // folded LayerNorm (A is the centred row): the lane owns the token
                float* const orow = p.out + mm * p.N + n0 + 4 * qd;
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
                    if (live) *reinterpret_cast<f32x4*>(orow + 16 * n) = v;
                }
            }
        }
        prev_counted = m0 + BM <= p.M;
        zero_acc();
        SGEMM_T(t5);
        SGEMM_ADD(3, t4, t5);
    }
#ifdef SGEMM_STAMPS
    if (p.stamps && lane == 0) {
        unsigned long long* o = p.stamps + ((size_t)blockIdx.x * NW + wave) * 8;
        o[0] = st_acc[0], o[1] = st_acc[1], o[2] = st_acc[2], o[3] = st_acc[3];
        o[4] = sgemm_stamp() - st_begin, o[5] = __builtin_amdgcn_s_memrealtime() - rt_begin;
    }
#endif
}



}  // namespace pafuse
