// train_kernels.hpp - device code of the training path: the backward of every layer of MixSTE2
// (the train-mode forward runs on the inference kernels; its DropPath factor and saved pre-norm sums are the
// EPI_ROWLN_TRAIN form of the whole-row GEMM epilogue in kernels.hpp).
// gfx950 only.  All reductions over rows are two-stage (per-workgroup partials in a fixed order, then one summing
// pass), so gradients are bit-reproducible run to run - no float atomics anywhere.
//
// Reference layers (common/mixste.py): Mlp :24-43, Attention :46-82, Block :84-125 (timm DropPath on both residual
// branches), MixSTE2.forward train branch :215-225,260-298; nn.LayerNorm / nn.GELU / nn.Linear backward as autograd
// derives them.
#pragma once
#include "kernels.hpp"

namespace pafuse {

// row of the [(b,f,j),C] token matrix -> index of the sequence DropPath draws its mask for
//   spatial block: sequences are (b,f): row / J          temporal block: sequences are (b,j)
struct SeqMap {
    int temporal, J, FJ;
};
__device__ __forceinline__ int64_t seq_of(int64_t row, const SeqMap m) {
    return m.temporal ? (row / m.FJ) * m.J + row % m.J : row / m.J;
}

// ----------------------------------------------------------------------------------------------------------------
// LayerNorm backward.  y = (x - mean) * rstd * w + b  =>  with g = dy * w, xh = (x - mean) * rstd:
//   dx = rstd * (g - mean(g) - xh * mean(g * xh)),  dw = sum_rows dy * xh,  db = sum_rows dy.
// dx_out = add + dx (add may be null: the residual gradient that bypasses the norm); out_scaled = drop[seq] * dx_out
// (the gradient entering the DropPath-scaled branch below).  A wave walks LNB_ROWS_PER_WAVE rows and keeps its slice of
// dw/db (and the column sums of out_scaled) in registers; the four waves of a workgroup combine through LDS into
// partial[block][3][C].
// ----------------------------------------------------------------------------------------------------------------
constexpr int LNB_ROWS_PER_WAVE = 16;
constexpr int LNB_ROWS_PER_BLOCK = 4 * LNB_ROWS_PER_WAVE;

struct LnBackwardParams {
    const float *dy, *x, *w;  // [M,C], [M,C], [C]
    float eps;
    const float* add;  // [M,C] or null
    float* dx;         // [M,C]
    const float* drop;  // [nseq] or null
    SeqMap map;
    float* out_scaled;  // [M,C] or null
    float* partial;     // [blocks][3][C]: dw, db, column sums of out_scaled (= bias gradient of the branch's last Linear)
    int64_t M;
    int C;
};

__global__ void __launch_bounds__(256) ln_backward_kernel(const LnBackwardParams p) {
    __shared__ float red[4][3][64 * LN_MAX_PER_LANE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, C = p.C;
    const float invC = 1.0f / (float)C;
    float dw[LN_MAX_PER_LANE], db[LN_MAX_PER_LANE], ds[LN_MAX_PER_LANE], wv[LN_MAX_PER_LANE];
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        dw[i] = db[i] = ds[i] = 0.f;
        wv[i] = (lane + 64 * i < C) ? p.w[lane + 64 * i] : 0.f;
    }
    const int64_t row0 = (int64_t)blockIdx.x * LNB_ROWS_PER_BLOCK + wave * LNB_ROWS_PER_WAVE;
    for (int rr = 0; rr < LNB_ROWS_PER_WAVE; ++rr) {
        const int64_t row = row0 + rr;
        if (row >= p.M) break;  // wave-uniform
        float x[LN_MAX_PER_LANE], dy[LN_MAX_PER_LANE];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
            const bool in = lane + 64 * i < C;
            x[i] = in ? p.x[row * C + lane + 64 * i] : 0.f;
            dy[i] = in ? p.dy[row * C + lane + 64 * i] : 0.f;
            s += x[i];
        }
        const float mean = wave_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAX_PER_LANE; ++i)
            if (lane + 64 * i < C) {
                const float dlt = x[i] - mean;
                q += dlt * dlt;
            }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * invC + p.eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAX_PER_LANE; ++i)
            if (lane + 64 * i < C) {
                x[i] = (x[i] - mean) * rstd;  // xh
                const float g = dy[i] * wv[i];
                sg += g;
                sgx += g * x[i];
                dw[i] += dy[i] * x[i];
                db[i] += dy[i];
            }
        const float c1 = wave_sum(sg) * invC, c2 = wave_sum(sgx) * invC;
        const float d = p.drop ? p.drop[seq_of(row, p.map)] : 1.0f;
#pragma unroll
        for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            if (c < C) {
                float v = rstd * (dy[i] * wv[i] - c1 - x[i] * c2);
                if (p.add) v += p.add[row * C + c];
                p.dx[row * C + c] = v;
                if (p.out_scaled) {
                    p.out_scaled[row * C + c] = d * v;
                    ds[i] += d * v;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        red[wave][0][lane + 64 * i] = dw[i];
        red[wave][1][lane + 64 * i] = db[i];
        red[wave][2][lane + 64 * i] = ds[i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * C; i += 256) {
        const int which = i / C, c = i % C;
        p.partial[((int64_t)blockIdx.x * 3 + which) * C + c] =
            ((red[0][which][c] + red[1][which][c]) + red[2][which][c]) + red[3][which][c];
    }
}

// The same for the PAFUSE widths (C % 4 == 0, C <= 384): a HALF-wave per row - lane li owns the channel quads li, li + 32,
// li + 64 as 16-byte loads / stores, the two rows of a wave run side by side and every row reduction is a 32-lane
// shuffle chain (the one-wave-per-row form above issues 6 scalar loads per tensor and row and walks its 16 rows one after
// the other behind five 64-lane reductions each: 86 us per launch against a 37 us byte floor).  Eight half-waves per
// workgroup, 8 rows each: the same 64 rows per workgroup and the same partial[block][3][C] layout.
constexpr int LNB_NV = 3;  // channel quads per lane (C <= 384)

__global__ void __launch_bounds__(256) ln_backward_quad_kernel(const LnBackwardParams p) {
    __shared__ __attribute__((aligned(16))) float red[8][3][4 * 32 * LNB_NV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, hw = wave * 2 + (lane >> 5), C = p.C;
    const int NQ = C / 4;
    const float invC = 1.0f / (float)C;
    f32x4 dw[LNB_NV], db[LNB_NV], ds[LNB_NV], wv[LNB_NV];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < LNB_NV; ++i) {
        dw[i] = db[i] = ds[i] = zero;
        wv[i] = (li + 32 * i < NQ) ? *reinterpret_cast<const f32x4*>(p.w + 4 * (li + 32 * i)) : zero;
    }
    constexpr int ROWS = LNB_ROWS_PER_BLOCK / 8;
    const int64_t row0 = (int64_t)blockIdx.x * LNB_ROWS_PER_BLOCK + hw * ROWS;
    for (int rr = 0; rr < ROWS; ++rr) {
        const int64_t row = row0 + rr;
        const bool live = row < p.M;              // uniform per half-wave; a dead half still takes part in the shuffles
        const int64_t ro = (live ? row : p.M - 1) * C;
        f32x4 x[LNB_NV], dy[LNB_NV], ad[LNB_NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LNB_NV; ++i) {
            const bool in = li + 32 * i < NQ;
            x[i] = in ? *reinterpret_cast<const f32x4*>(p.x + ro + 4 * (li + 32 * i)) : zero;
            dy[i] = (in && live) ? *reinterpret_cast<const f32x4*>(p.dy + ro + 4 * (li + 32 * i)) : zero;
            ad[i] = (in && p.add) ? *reinterpret_cast<const f32x4*>(p.add + ro + 4 * (li + 32 * i)) : zero;
#pragma unroll
            for (int e = 0; e < 4; ++e) s += x[i][e];
        }
        const float mean = half_wave_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LNB_NV; ++i)
            if (li + 32 * i < NQ) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dlt = x[i][e] - mean;
                    q += dlt * dlt;
                }
            }
        const float rstd = 1.0f / sqrtf(half_wave_sum(q) * invC + p.eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < LNB_NV; ++i)
            if (li + 32 * i < NQ) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x[i][e] = (x[i][e] - mean) * rstd;  // xh
                    const float g = dy[i][e] * wv[i][e];
                    sg += g;
                    sgx += g * x[i][e];
                    dw[i][e] += dy[i][e] * x[i][e];
                    db[i][e] += dy[i][e];
                }
            }
        const float c1 = half_wave_sum(sg) * invC, c2 = half_wave_sum(sgx) * invC;
        const float d = (p.drop && live) ? p.drop[seq_of(row, p.map)] : 1.0f;
#pragma unroll
        for (int i = 0; i < LNB_NV; ++i) {
            const int c4 = li + 32 * i;
            if (c4 < NQ && live) {
                f32x4 v, sv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = rstd * (dy[i][e] * wv[i][e] - c1 - x[i][e] * c2);
                    if (p.add) v[e] += ad[i][e];
                    sv[e] = d * v[e];
                }
                *reinterpret_cast<f32x4*>(p.dx + ro + 4 * c4) = v;
                if (p.out_scaled) {
                    *reinterpret_cast<f32x4*>(p.out_scaled + ro + 4 * c4) = sv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ds[i][e] += sv[e];
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < LNB_NV; ++i) {
        *reinterpret_cast<f32x4*>(&red[hw][0][4 * (li + 32 * i)]) = dw[i];
        *reinterpret_cast<f32x4*>(&red[hw][1][4 * (li + 32 * i)]) = db[i];
        *reinterpret_cast<f32x4*>(&red[hw][2][4 * (li + 32 * i)]) = ds[i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * C; i += 256) {
        const int which = i / C, c = i % C;
        float t = red[0][which][c];
#pragma unroll
        for (int h = 1; h < 8; ++h) t += red[h][which][c];   // fixed order
        p.partial[((int64_t)blockIdx.x * 3 + which) * C + c] = t;
    }
}

// out[i] = (accumulate ? out[i] : 0) + sum_s partial[s * stride + i], in a fixed order: a workgroup owns 256 / G
// columns; its G lane groups each sum the parts s = g, g + G, ... in ascending order, then the G group sums are added
// in ascending g.  G = 1 for wide outputs (plenty of columns to fill the chip), 16 for narrow ones with many parts
// (LayerNorm / bias gradients: a few hundred columns, up to ~1000 parts).  Columns >= split go to out2[i - split]
// (LayerNorm: dw and db partials lie side by side but their gradients are separate tensors), columns >= split2 to out3 when
// it is given (the third vector of a LayerNorm backward: the bias gradient of the branch below).
template <int G>
__global__ void __launch_bounds__(256) reduce_partials_kernel(const float* partial, float* out, float* out2, int split,
                                                              int n, int nparts, int64_t stride, int accumulate,
                                                              float* out3 = nullptr, int split2 = 0) {
    constexpr int COLS = 256 / G;
    __shared__ float red[G][COLS];
    const int col = threadIdx.x % COLS, g = threadIdx.x / COLS;
    const int i = blockIdx.x * COLS + col;
    float s = 0.f;
    if (i < n) {  // four independent chains (parts k = g + G (4 q + c), c = 0..3), joined in a fixed order: the loads of a
                  // chain of ~60 parts were a serial latency chain, 14 us for 3 MB
        float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
        const float* src = partial + i;
        int k = g;
        for (; k + 3 * G < nparts; k += 4 * G) {
            const float v0 = src[(int64_t)k * stride], v1 = src[(int64_t)(k + G) * stride];
            const float v2 = src[(int64_t)(k + 2 * G) * stride], v3 = src[(int64_t)(k + 3 * G) * stride];
            c0 += v0, c1 += v1, c2 += v2, c3 += v3;
        }
        if (k < nparts) c0 += src[(int64_t)k * stride];
        if (k + G < nparts) c1 += src[(int64_t)(k + G) * stride];
        if (k + 2 * G < nparts) c2 += src[(int64_t)(k + 2 * G) * stride];
        s = (c0 + c1) + (c2 + c3);
    }
    // columns [0, split) -> out, [split, split2) -> out2, [split2, n) -> out3 (out3 null: everything from split on -> out2)
    float* dst = i < split ? out + i : ((out3 && i >= split2) ? out3 + (i - split2) : out2 + (i - split));
    if (G == 1) {
        if (i < n) *dst = accumulate ? *dst + s : s;
        return;
    }
    red[g][col] = s;
    __syncthreads();
    if (g == 0 && i < n) {
        float t = red[0][col];
#pragma unroll
        for (int k = 1; k < G; ++k) t += red[k][col];
        *dst = accumulate ? *dst + t : t;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Every weight image of a pass in ONE launch.  The weights change once per step, their images are needed once in the forward
// (W: layout of the plain / the whole-row kernels) and once in the backward (W^T for the dX GEMMs): made one by one in front
// of their GEMMs they were 96 five-microsecond launches per part and pass in the middle of the dependent chain.
// kind 0: split_weights_kernel<32> (plain GEMMs)   1: <16> (whole-row GEMMs)   2: split_weights_transposed_kernel<32>
//      3, 4: as 0, 2 in the layout of the 16x16x32 kernel (gemm16_tile)
// ----------------------------------------------------------------------------------------------------------------
constexpr int SPLIT_BATCH_MAX = 4 * 2 * PAFUSE_MAX_DEPTH;
struct SplitBatchItem {
    const float* W;
    uint8_t* out;
    int N, K;          // the weight as it lies: [N, K]
    int kind;
    int first_block;   // index of this item's first workgroup in the launch
};
struct SplitBatchParams {
    SplitBatchItem item[SPLIT_BATCH_MAX];
    int count;
};

template <int BKC, int M16>
__device__ __forceinline__ void split_group_of_8(const float* W, uint8_t* out, int N, int K, int64_t idx) {
    const int groups = K / 8;
    if (idx >= (int64_t)N * groups) return;
    constexpr int SUBS = BKC / 8, ROW = 6 * BKC;
    const int n = (int)(idx / groups), g8 = (int)(idx % groups);
    const int chunk = g8 / SUBS, sb = g8 % SUBS;
    const float* src = W + (int64_t)n * K + g8 * 8;
    const bf16x8x3 sp = split3(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4));
    uint8_t* dst = out + ((int64_t)chunk * N + n) * ROW + wsplit_sub_offset<BKC, M16>(n, sb);
    *reinterpret_cast<bf16x8*>(dst) = sp.s0;
    *reinterpret_cast<bf16x8*>(dst + 16) = sp.s1;
    *reinterpret_cast<bf16x8*>(dst + 32) = sp.s2;
}

__global__ void __launch_bounds__(256) split_weights_batch_kernel(const SplitBatchParams p) {
    int lo = 0, hi = p.count - 1;   // the last item whose first workgroup is <= blockIdx.x (uniform: scalar loads)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (p.item[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const SplitBatchItem it = p.item[lo];
    const int64_t idx = (int64_t)((int)blockIdx.x - it.first_block) * 256 + threadIdx.x;
    if (it.kind == 0) {
        split_group_of_8<32, 0>(it.W, it.out, it.N, it.K, idx);
    } else if (it.kind == 1) {
        split_group_of_8<16, 0>(it.W, it.out, it.N, it.K, idx);
    } else if (it.kind == 3) {
        split_group_of_8<32, 1>(it.W, it.out, it.N, it.K, idx);
    } else {   // image rows = the K columns of W, contraction = its N rows (split_weights_transposed_kernel<32>)
        const int Nrows = it.K, R = it.N, groups = R / 8;
        if (idx >= (int64_t)Nrows * groups) return;
        const int n = (int)(idx % Nrows), g8 = (int)(idx / Nrows);
        const int chunk = g8 / 4, sb = g8 % 4;
        const float* src = it.W + (int64_t)g8 * 8 * Nrows + n;
        f32x4 l4, h4;
#pragma unroll
        for (int i = 0; i < 4; ++i) l4[i] = src[(int64_t)i * Nrows], h4[i] = src[(int64_t)(4 + i) * Nrows];
        const bf16x8x3 sp = split3(l4, h4);
        uint8_t* dst = it.out + ((int64_t)chunk * Nrows + n) * 192 + (it.kind == 4 ? wsplit_sub_offset<32, 1>(n, sb) : wsplit_sub_offset<32, 0>(n, sb));
        *reinterpret_cast<bf16x8*>(dst) = sp.s0;
        *reinterpret_cast<bf16x8*>(dst + 16) = sp.s1;
        *reinterpret_cast<bf16x8*>(dst + 32) = sp.s2;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Exact GELU forward / backward (nn.GELU(), approximate='none'):  d/du [u Phi(u)] = Phi(u) + u phi(u)
// ----------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gelu_forward_kernel(const float* u, float* h, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 v = reinterpret_cast<const f32x4*>(u)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
    reinterpret_cast<f32x4*>(h)[i] = v;
}

// du = dh * gelu'(u) for the rows of one chunk, and the chunk's column sums of du (= partial bias gradient of fc1):
// thread q owns the column quad q of every row of the chunk, so the sums stay in registers
__global__ void __launch_bounds__(256) gelu_backward_kernel(const float* u, const float* dh, float* du, float* partial,
                                                            int64_t M, int N, int64_t rows_per_chunk) {
    const int64_t lo = (int64_t)blockIdx.x * rows_per_chunk;
    const int64_t hi = lo + rows_per_chunk < M ? lo + rows_per_chunk : M;
    for (int q = threadIdx.x; q < N / 4; q += 256) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int64_t m = lo; m < hi; ++m) {
            const f32x4 uv = *reinterpret_cast<const f32x4*>(u + m * N + 4 * q);
            const f32x4 g = *reinterpret_cast<const f32x4*>(dh + m * N + 4 * q);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = uv[e];
                const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
                const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
                o[e] = g[e] * (cdf + x * pdf);
                acc[e] += o[e];
            }
            *reinterpret_cast<f32x4*>(du + m * N + 4 * q) = o;
        }
        *reinterpret_cast<f32x4*>(partial + (int64_t)blockIdx.x * N + 4 * q) = acc;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Weight-gradient GEMM: dW[n][k] = sum_m dY[m][n] * X[m][k]  (nn.Linear: dY [M,N] output grad, X [M,K] its input).
// Both operands are read as they lie in memory (rows = the contracted index m), staged in LDS in that natural [m][.]
// layout, and fed to v_mfma_f32_32x32x2_f32 with scalar LDS reads: lane (r,h) needs Y[m=2s+h][n=r'] - 32 consecutive
// floats per half-wave, and the +32 row padding puts the two halves on disjoint banks.  128(n) x 128(k) output tiles
// (64-wide ones re-read Y twice as often and ran into the per-CU load path),
// wave w owns the 32-row strip w.  The contraction is M = 25-70 k rows and the output is small, so M is cut into
// `splits` ranges (grid.y) whose partial tiles are summed afterwards in a fixed order.
// ----------------------------------------------------------------------------------------------------------------
struct TnParams {
    const float *Y, *X;  // [M,N], [M,K]
    float* partial;      // [splits][N][K]
    int64_t M, rows_per_split;
    int N, K;
};

constexpr int TN_BN = 128, TN_BK = 128, TN_LDY = TN_BN + 32, TN_LDX = TN_BK + 32;
constexpr int TN_KB = TN_BK / 32;  // 32-column blocks per wave strip

// (fp32 products; the split-precision weight gradients are tn_split_gemm_kernel / tn_split_big_kernel below)
__global__ void __launch_bounds__(256) tn_gemm_kernel(const TnParams p) {
    __shared__ __attribute__((aligned(16))) float Ys[32 * TN_LDY];
    __shared__ __attribute__((aligned(16))) float Xs[32 * TN_LDX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int tiles_k = (p.K + TN_BK - 1) / TN_BK;
    const int n0 = (blockIdx.x / tiles_k) * TN_BN, k0 = (blockIdx.x % tiles_k) * TN_BK;
    const int64_t m_lo = (int64_t)blockIdx.y * p.rows_per_split;
    const int64_t m_hi = m_lo + p.rows_per_split < p.M ? m_lo + p.rows_per_split : p.M;
    // staging: a 32-row chunk of Y (TN_BN columns) and of X (TN_BK columns), one float4 per thread and row group
    constexpr int YC4 = TN_BN / 4, XC4 = TN_BK / 4, YR = 256 / YC4, XR = 256 / XC4, YL = 32 / YR, XL = 32 / XR;
    const int yrow = tid / YC4, yc4 = tid % YC4, xrow = tid / XC4, xc4 = tid % XC4;
    const bool y_in = n0 + 4 * yc4 < p.N, x_in = k0 + 4 * xc4 < p.K;  // N, K are multiples of 4
    f32x4 yreg[YL], xreg[XL];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    auto load = [&](int64_t m0) {
#pragma unroll
        for (int i = 0; i < YL; ++i) {
            const int64_t m = m0 + yrow + YR * i;
            yreg[i] = (y_in && m < m_hi) ? *reinterpret_cast<const f32x4*>(p.Y + m * p.N + n0 + 4 * yc4) : zero;
        }
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int64_t m = m0 + xrow + XR * i;
            xreg[i] = (x_in && m < m_hi) ? *reinterpret_cast<const f32x4*>(p.X + m * p.K + k0 + 4 * xc4) : zero;
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int i = 0; i < YL; ++i) *reinterpret_cast<f32x4*>(Ys + (yrow + YR * i) * TN_LDY + 4 * yc4) = yreg[i];
#pragma unroll
        for (int i = 0; i < XL; ++i) *reinterpret_cast<f32x4*>(Xs + (xrow + XR * i) * TN_LDX + 4 * xc4) = xreg[i];
    };
    f32x16 acc[TN_KB];
#pragma unroll
    for (int b = 0; b < TN_KB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
    if (m_lo < m_hi) {
        load(m_lo);
        store();
        __syncthreads();
        for (int64_t m0 = m_lo; m0 < m_hi; m0 += 32) {
            const bool more = m0 + 32 < m_hi;
            if (more) load(m0 + 32);
            const float* ya = Ys + h * TN_LDY + wave * 32 + r;
            const float* xb = Xs + h * TN_LDX + r;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float a = ya[2 * s * TN_LDY];
#pragma unroll
                for (int b = 0; b < TN_KB; ++b)
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xb[2 * s * TN_LDX + 32 * b], acc[b], 0, 0, 0);
            }
            __syncthreads();
            if (more) store();
            __syncthreads();
        }
    }
    float* out = p.partial + (int64_t)blockIdx.y * p.N * p.K;
#pragma unroll
    for (int b = 0; b < TN_KB; ++b) {
        const int k = k0 + 32 * b + r;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int n = n0 + wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (n < p.N && k < p.K) out[(int64_t)n * p.K + k] = acc[b][reg];
        }
    }
}

// The split-precision weight gradient, round 5: both operands are split ONCE per workgroup, where they are staged.
// Round 4's kernel split every fragment in registers right before its MFMAs: the X fragments are the same for the four
// waves, so 5 splits of 8 values (~ 280 VALU instructions of 4 cycles) sat beside 24 MFMAs (768 cycles), at 0.3 of the
// scheme's ceiling.  Here the thread that stages 8 consecutive m of a column pair splits them
// (xsplit_pair, 5.5 instructions per value, once) and writes the three bf16 slices TRANSPOSED: LDS holds per operand
// [slice][m group of 8][column][8 x bf16], so a fragment (8 consecutive m of one column) is one ds_read_b128 per slice -
// 30 b128 reads per wave and chunk instead of 80 scalar ones, and 2.5x fewer VALU instructions.
//   column position inside its block of 16: (col & 15) ^ ((col >> 4) & 1) - a staging wave writes columns 2 lane + j, i.e.
//   every other 16-byte slot; the flip of odd blocks puts the 16 lanes of a write group on 16 different bank quads, and
//   the fragment reads (32 consecutive columns) stay a permutation inside each block.
// colsum (may be null): the column sums of Y over the split's rows, partial[split][n] - the bias gradient of the same
// Linear, taken from the values the staging threads hold anyway (written by the workgroups of the first k tile).
constexpr int TNS_PLANE = 4 * 128 * 16;  // bytes of one slice plane of one operand: 4 m groups x 128 columns x 16 B
__device__ __forceinline__ int tns_slot(int mg, int col) {
    return (mg * 128 + (col & ~15) + ((col & 15) ^ ((col >> 4) & 1))) * 16;
}

struct TnSplitParams {
    const float *Y, *X;  // [M,N], [M,K]
    float* partial;      // [splits][N][K]
    float* colsum;       // [splits][N] or null
    int64_t M, rows_per_split;
    int N, K;
};

#ifndef PAFUSE_TNS_MINW
#define PAFUSE_TNS_MINW 2
#endif
__global__ void __launch_bounds__(256, PAFUSE_TNS_MINW) tn_split_gemm_kernel(const TnSplitParams p) {
    __shared__ __attribute__((aligned(16))) uint8_t Ys[3 * TNS_PLANE];
    __shared__ __attribute__((aligned(16))) uint8_t Xs[3 * TNS_PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int tiles_k = (p.K + TN_BK - 1) / TN_BK;
    const int n0 = (blockIdx.x / tiles_k) * TN_BN, k0 = (blockIdx.x % tiles_k) * TN_BK;
    const int64_t m_lo = (int64_t)blockIdx.y * p.rows_per_split;
    const int64_t m_hi = m_lo + p.rows_per_split < p.M ? m_lo + p.rows_per_split : p.M;
    // staging: wave w holds the m group w (8 rows) of the 32-row chunk, lane c the column pair 2c, 2c + 1
    const int mg = wave, c2 = 2 * lane;
    const bool y_in = n0 + c2 < p.N, x_in = k0 + c2 < p.K;  // N, K are even
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    f32x2_t yreg[8], xreg[8];
    const f32x2_t zero2 = {0.f, 0.f};
    auto load = [&](int64_t m0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t m = m0 + 8 * mg + i, mc = m < m_hi ? m : m_hi - 1;   // clamped, unconditional loads: no branch per row
            const f32x2_t yv = *reinterpret_cast<const f32x2_t*>(p.Y + mc * p.N + (y_in ? n0 + c2 : 0));
            const f32x2_t xv = *reinterpret_cast<const f32x2_t*>(p.X + mc * p.K + (x_in ? k0 + c2 : 0));
            yreg[i] = (y_in && m < m_hi) ? yv : zero2;
            xreg[i] = (x_in && m < m_hi) ? xv : zero2;
        }
    };
    float csum[2] = {0.f, 0.f};
    auto store_operand = [&](const f32x2_t* v, uint8_t* base) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            uint32_t s0[4], s1[4], s2[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) xsplit_pair(v[2 * q][j], v[2 * q + 1][j], s0[q], s1[q], s2[q]);
            uint8_t* dst = base + tns_slot(mg, c2 + j);
            *reinterpret_cast<u32x4*>(dst) = u32x4{s0[0], s0[1], s0[2], s0[3]};
            *reinterpret_cast<u32x4*>(dst + TNS_PLANE) = u32x4{s1[0], s1[1], s1[2], s1[3]};
            *reinterpret_cast<u32x4*>(dst + 2 * TNS_PLANE) = u32x4{s2[0], s2[1], s2[2], s2[3]};
        }
    };
    auto store = [&]() {
        if (p.colsum) {
#pragma unroll
            for (int i = 0; i < 8; ++i) csum[0] += yreg[i][0], csum[1] += yreg[i][1];  // ascending m
        }
        store_operand(yreg, Ys);
        store_operand(xreg, Xs);
    };
    f32x16 acc[TN_KB];
#pragma unroll
    for (int b = 0; b < TN_KB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
    if (m_lo < m_hi) {
        load(m_lo);
        store();
        __syncthreads();
        for (int64_t m0 = m_lo; m0 < m_hi; m0 += 32) {
            const bool more = m0 + 32 < m_hi;
            if (more) load(m0 + 32);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                // lane (r, h) of a 16-deep step feeds m = 16 s2 + 8 h + 0..7 of its column = m group 2 s2 + h
                const uint8_t* ya = Ys + tns_slot(2 * s2 + h, wave * 32 + r);
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(ya);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(ya + TNS_PLANE);
                const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(ya + 2 * TNS_PLANE);
#pragma unroll
                for (int b = 0; b < TN_KB; ++b) {
                    const uint8_t* xb = Xs + tns_slot(2 * s2 + h, 32 * b + r);
                    const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(xb);
                    const bf16x8 x1 = *reinterpret_cast<const bf16x8*>(xb + TNS_PLANE);
                    const bf16x8 x2 = *reinterpret_cast<const bf16x8*>(xb + 2 * TNS_PLANE);
                    acc[b] = mfma_bf16_k16(a0, x2, acc[b]);   // small terms first, the leading product last
                    acc[b] = mfma_bf16_k16(a2, x0, acc[b]);
                    acc[b] = mfma_bf16_k16(a1, x1, acc[b]);
                    acc[b] = mfma_bf16_k16(a0, x1, acc[b]);
                    acc[b] = mfma_bf16_k16(a1, x0, acc[b]);
                    acc[b] = mfma_bf16_k16(a0, x0, acc[b]);
                }
            }
            __syncthreads();
            if (more) store();
            __syncthreads();
        }
    }
    float* out = p.partial + (int64_t)blockIdx.y * p.N * p.K;
#pragma unroll
    for (int b = 0; b < TN_KB; ++b) {
        const int k = k0 + 32 * b + r;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int n = n0 + wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (n < p.N && k < p.K) out[(int64_t)n * p.K + k] = acc[b][reg];
        }
    }
    if (p.colsum && k0 == 0) {  // the four m groups of every column, added in ascending group order
        float* red = reinterpret_cast<float*>(Ys);  // the K loop is over: every wave passed its last barrier
        red[wave * 128 + c2] = csum[0], red[wave * 128 + c2 + 1] = csum[1];
        __syncthreads();
        if (tid < 128 && n0 + tid < p.N)
            p.colsum[(int64_t)blockIdx.y * p.N + n0 + tid] = ((red[tid] + red[128 + tid]) + red[256 + tid]) + red[384 + tid];
    }
}

// The same product on LARGE tiles.  The 128 x 128 tile above streams 32 KB per 32 rows of the contraction for 1 MFLOP: two
// workgroups per CU ask the CU's memory pipe for 42 B per clock - the kernel is bound there (it measured 112 us per launch
// with the staging split against 118 without: the VALU work was not the limit).  The output of a weight gradient is small and
// the contraction (M = 24 - 68 k rows) is cut into splits anyway, so the tile can be as large as the accumulators allow:
// (32 NB) x (32 KB) outputs on WN x WK waves, every wave NB/WN x KB/WK blocks of 32 x 32 (96 - 128 accumulator registers),
// picked so that the three PAFUSE widths divide without padding - 256 x 256 on eight waves (hands: 256, 512, 768),
// 192 x 192 on six (body: 384, 768, 1152), 224 x 224 on seven (face: 224, 448, 672).  Bytes per FLOP halve, one workgroup
// per CU, 11 B per clock.
// Chunks of 16 rows of the contraction, two LDS stages ([stage][operand][slice][m group of 8][column][8 x bf16], the layout
// and column flip of tn_split_gemm_kernel): iteration i splits and stores the registers of chunk i + 1 into the other stage,
// loads chunk i + 2 into registers and multiplies chunk i - one barrier per chunk, the split instructions sit in the
// shadow of the MFMAs.  Thread t stages 8 rows x 2 columns: the first 32 NB threads Y (16 NB column pairs x 2 m groups),
// the rest X - 32 (NB + KB) = the workgroup's thread count for all three shapes.
template <int NB, int KB, int WN, int WK>
__global__ void __launch_bounds__(64 * WN * WK, 2) tn_split_big_kernel(const TnSplitParams p) {
    constexpr int NTHR = 64 * WN * WK, BNW = NB / WN, BKW = KB / WK, YC = 32 * NB, XC = 32 * KB;
    static_assert(NB % WN == 0 && KB % WK == 0 && 32 * (NB + KB) == NTHR, "tile / wave grid");
    constexpr int Y_PLANE = 2 * YC * 16, X_PLANE = 2 * XC * 16;       // bytes of one slice plane (two m groups)
    constexpr int X_BASE = 3 * Y_PLANE, STAGE = 3 * (Y_PLANE + X_PLANE);
    extern __shared__ __attribute__((aligned(16))) uint8_t tn_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wn = wave / WK, wk = wave % WK;
    const int tiles_k = p.K / XC;
    const int n0 = (blockIdx.x / tiles_k) * YC, k0 = (blockIdx.x % tiles_k) * XC;
    const int64_t m_lo = (int64_t)blockIdx.y * p.rows_per_split;
    const int64_t m_hi = m_lo + p.rows_per_split < p.M ? m_lo + p.rows_per_split : p.M;
    auto slot = [](int pitch, int mg, int col) { return (mg * pitch + (col & ~15) + ((col & 15) ^ ((col >> 4) & 1))) * 16; };
    // staging role of this thread
    const bool is_y = tid < 32 * NB;
    const int st = is_y ? tid : tid - 32 * NB;
    const int pairs = is_y ? 16 * NB : 16 * KB;
    const int smg = st / pairs, sc2 = 2 * (st % pairs);
    const float* sbase = is_y ? p.Y + n0 + sc2 : p.X + k0 + sc2;
    const int sld = is_y ? p.N : p.K;
    const int s_lds = is_y ? slot(YC, smg, sc2) : X_BASE + slot(XC, smg, sc2);
    const int s_lds1 = is_y ? slot(YC, smg, sc2 + 1) : X_BASE + slot(XC, smg, sc2 + 1);
    const int s_plane = is_y ? Y_PLANE : X_PLANE;
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    f32x2_t sreg[8];
    const f32x2_t zero2 = {0.f, 0.f};
    auto load = [&](int64_t m0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // unconditional load from a clamped row, zeroed afterwards: a guarded load compiles to a branch per row, and the
            // eight branches cut the loop body into blocks the scheduler cannot overlap
            const int64_t m = m0 + 8 * smg + i;
            const f32x2_t v = *reinterpret_cast<const f32x2_t*>(sbase + (m < m_hi ? m : m_hi - 1) * sld);
            sreg[i] = m < m_hi ? v : zero2;
        }
    };
    float csum[2] = {0.f, 0.f};
    auto store = [&](int stage) {
        if (p.colsum && is_y) {
#pragma unroll
            for (int i = 0; i < 8; ++i) csum[0] += sreg[i][0], csum[1] += sreg[i][1];  // ascending m
        }
        uint8_t* base = tn_lds + stage * STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            uint32_t s0[4], s1[4], s2[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) xsplit_pair(sreg[2 * q][j], sreg[2 * q + 1][j], s0[q], s1[q], s2[q]);
            uint8_t* dst = base + (j ? s_lds1 : s_lds);
            *reinterpret_cast<u32x4*>(dst) = u32x4{s0[0], s0[1], s0[2], s0[3]};
            *reinterpret_cast<u32x4*>(dst + s_plane) = u32x4{s1[0], s1[1], s1[2], s1[3]};
            *reinterpret_cast<u32x4*>(dst + 2 * s_plane) = u32x4{s2[0], s2[1], s2[2], s2[3]};
        }
    };
    f32x16 acc[BNW][BKW];
#pragma unroll
    for (int a = 0; a < BNW; ++a)
#pragma unroll
        for (int b = 0; b < BKW; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    // fragment addresses of this lane inside a stage: m group h, column (block) * 32 + r
    int ya[BNW], xa[BKW];
#pragma unroll
    for (int a = 0; a < BNW; ++a) ya[a] = slot(YC, h, (wn * BNW + a) * 32 + r);
#pragma unroll
    for (int b = 0; b < BKW; ++b) xa[b] = X_BASE + slot(XC, h, (wk * BKW + b) * 32 + r);
    const int64_t nchunks = m_lo < m_hi ? (m_hi - m_lo + 15) / 16 : 0;
    if (nchunks > 0) {
        load(m_lo);
        store(0);
        if (nchunks > 1) load(m_lo + 16);
        __syncthreads();
        // One iteration = the 6 BNW BKW MFMAs of chunk i, and BETWEEN them, one small piece per MFMA in program order (a
        // scheduling fence after every slot): the split of the registers of chunk i + 1 (16 half splits of a value pair, 5 - 6
        // VALU instructions each), their six stage writes, then the eight row loads of chunk i + 2.  A wave issues in order: an
        // MFMA holds the matrix pipe for 32 cycles, the 5 - 6 instructions behind it issue in its shadow.  (The compiler's own
        // order put all ~115 split instructions in front of the first MFMA.)  Past the last chunk the loads return zeros
        // (rows >= m_hi) and the writes fill a stage nobody reads: the body has no branch.
        constexpr int NSLOT = 6 * BNW * BKW;
        const float* sbase_lo = sbase + m_lo * sld;
        const int nrows = (int)(m_hi - m_lo);
        static_assert(NSLOT >= 30, "not enough MFMA slots for the staging pieces");
        for (int64_t i = 0; i < nchunks; ++i) {
            const uint8_t* st_base = tn_lds + (int)(i & 1) * STAGE;
            uint8_t* wr_base = tn_lds + (int)((i + 1) & 1) * STAGE;
            const int r_next = 16 * ((int)i + 2) + 8 * smg;
            const int r_cur = 16 * ((int)i + 1) + 8 * smg;   // first of this thread's rows of the chunk being split, from m_lo
            const bool want_csum = p.colsum && is_y;
            bf16x8 a0[BNW], a1[BNW], a2[BNW], x0[2], x1[2], x2[2];
#pragma unroll
            for (int a = 0; a < BNW; ++a) {
                a0[a] = *reinterpret_cast<const bf16x8*>(st_base + ya[a]);
                a1[a] = *reinterpret_cast<const bf16x8*>(st_base + ya[a] + Y_PLANE);
                a2[a] = *reinterpret_cast<const bf16x8*>(st_base + ya[a] + 2 * Y_PLANE);
            }
            x0[0] = *reinterpret_cast<const bf16x8*>(st_base + xa[0]);
            x1[0] = *reinterpret_cast<const bf16x8*>(st_base + xa[0] + X_PLANE);
            x2[0] = *reinterpret_cast<const bf16x8*>(st_base + xa[0] + 2 * X_PLANE);
            SplitPair sp[8];   // pair 4 j + q: rows 2 q, 2 q + 1 of column j
            static_for<NSLOT>([&](auto S) {
                constexpr int sl = decltype(S)::value;
                constexpr int b = sl / (6 * BNW), term = (sl / BNW) % 6, a = sl % BNW, cur = b & 1;
                if constexpr (term == 0 && a == 0 && b + 1 < BKW) {   // the next block column's fragments, one column ahead
                    x0[cur ^ 1] = *reinterpret_cast<const bf16x8*>(st_base + xa[b + 1]);
                    x1[cur ^ 1] = *reinterpret_cast<const bf16x8*>(st_base + xa[b + 1] + X_PLANE);
                    x2[cur ^ 1] = *reinterpret_cast<const bf16x8*>(st_base + xa[b + 1] + 2 * X_PLANE);
                }
                // the six products of a block small terms first, the leading one last; term-major over the wave's row blocks
                if constexpr (term == 0) acc[a][b] = mfma_bf16_k16(a0[a], x2[cur], acc[a][b]);
                if constexpr (term == 1) acc[a][b] = mfma_bf16_k16(a2[a], x0[cur], acc[a][b]);
                if constexpr (term == 2) acc[a][b] = mfma_bf16_k16(a1[a], x1[cur], acc[a][b]);
                if constexpr (term == 3) acc[a][b] = mfma_bf16_k16(a0[a], x1[cur], acc[a][b]);
                if constexpr (term == 4) acc[a][b] = mfma_bf16_k16(a1[a], x0[cur], acc[a][b]);
                if constexpr (term == 5) acc[a][b] = mfma_bf16_k16(a0[a], x0[cur], acc[a][b]);
                if constexpr (sl < 16) {
                    constexpr int pi = sl / 2, jj = pi / 4, q = pi % 4;
                    if constexpr (sl % 2 == 0) {
                        // rows past the split's end are zeroed HERE, an iteration after their load was issued: a select right
                        // behind the load makes the wave wait for it on the spot
                        float v0 = r_cur + 2 * q < nrows ? sreg[2 * q][jj] : 0.f, v1 = r_cur + 2 * q + 1 < nrows ? sreg[2 * q + 1][jj] : 0.f;
                        if (want_csum) csum[jj] += v0, csum[jj] += v1;   // ascending m
                        asm volatile("" : "+v"(v0), "+v"(v1));   // ONE value for all slices (see split2h)
                        sp[pi].x0 = v0, sp[pi].x1 = v1;
                        sp[pi].template stage<0>(), sp[pi].template stage<1>(), sp[pi].template stage<2>();
                    } else {
                        sp[pi].template stage<3>(), sp[pi].template stage<4>();
                    }
                } else if constexpr (sl < 22) {
                    constexpr int w = sl - 16, jj = w / 3, slice = w % 3;
                    uint8_t* dst = wr_base + (jj ? s_lds1 : s_lds) + slice * s_plane;
                    if constexpr (slice == 0) *reinterpret_cast<u32x4*>(dst) = u32x4{sp[4 * jj].s0, sp[4 * jj + 1].s0, sp[4 * jj + 2].s0, sp[4 * jj + 3].s0};
                    if constexpr (slice == 1) *reinterpret_cast<u32x4*>(dst) = u32x4{sp[4 * jj].s1, sp[4 * jj + 1].s1, sp[4 * jj + 2].s1, sp[4 * jj + 3].s1};
                    if constexpr (slice == 2) *reinterpret_cast<u32x4*>(dst) = u32x4{sp[4 * jj].s2, sp[4 * jj + 1].s2, sp[4 * jj + 2].s2, sp[4 * jj + 3].s2};
                } else if constexpr (sl < 30) {
                    constexpr int q = sl - 22;
                    const int rel = r_next + q;   // row from m_lo, clamped into the split (32-bit: a split spans a few MB)
                    sreg[q] = *reinterpret_cast<const f32x2_t*>(sbase_lo + (uint32_t)((rel < nrows ? rel : nrows - 1) * sld));
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            __syncthreads();
        }
    }
    float* out = p.partial + (int64_t)blockIdx.y * p.N * p.K;
#pragma unroll
    for (int a = 0; a < BNW; ++a)
#pragma unroll
        for (int b = 0; b < BKW; ++b) {
            const int k = k0 + (wk * BKW + b) * 32 + r;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int n = n0 + (wn * BNW + a) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                out[(int64_t)n * p.K + k] = acc[a][b][reg];
            }
        }
    if (p.colsum && k0 == 0) {  // the two m groups of every column (the last barrier of the loop is behind every wave)
        float* red = reinterpret_cast<float*>(tn_lds);
        if (is_y) red[smg * YC + sc2] = csum[0], red[smg * YC + sc2 + 1] = csum[1];
        __syncthreads();
        if (tid < YC) p.colsum[(int64_t)blockIdx.y * p.N + n0 + tid] = red[tid] + red[YC + tid];
    }
}

// partial[chunk][n] = sum over the rows of the chunk of X[row][n]   (bias gradients)
__global__ void __launch_bounds__(256) colsum_kernel(const float* X, float* partial, int64_t M, int N,
                                                     int64_t rows_per_chunk) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int64_t lo = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t hi = lo + rows_per_chunk < M ? lo + rows_per_chunk : M;
    float s = 0.f;
    for (int64_t m = lo; m < hi; ++m) s += X[m * N + n];
    partial[(int64_t)blockIdx.y * N + n] = s;
}

// out[k][n] = in[n][k]  (weights, once per backward pass: the dX GEMMs reuse the inference kernel, which wants W^T rows)
__global__ void __launch_bounds__(256) transpose_kernel(const float* in, float* out, int N, int K) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int n0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    for (int i = ty; i < 32; i += 8)
        if (n0 + i < N && k0 + tx < K) tile[i][tx] = in[(int64_t)(n0 + i) * K + k0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (k0 + i < K && n0 + tx < N) out[(int64_t)(k0 + i) * N + n0 + tx] = tile[tx][i];
}

// ----------------------------------------------------------------------------------------------------------------
// Attention backward for one (sequence, head) per workgroup.  With S = scale q k^T, P = softmax(S), O = P v:
//   dV = P^T dO,  dP = dO v^T,  dS = P * (dP - rowsum(P * dP)),  dq = scale dS k,  dk = scale dS^T q.
// P is recomputed from the saved qkv.  L <= 80, d <= 48: everything of one item lives in LDS; ~3 % of the FLOPs of a
// training step, so plain FMA loops.
// ----------------------------------------------------------------------------------------------------------------
struct AttnBackwardParams {
    const float* qkv;  // [M,3C]
    const float* d_o;  // [M,C]
    float* dqkv;       // [M,3C]
    int64_t nseq, group, group_stride, seq_stride, tok_stride;  // addressing as in AttnParams
    int L, C, heads, d;
    float scale;
};

// LDS row pitch of the [L][d] operands: a multiple of 4 floats (float4 reads) whose count of 16-byte granules is odd,
// so that lanes reading different rows at the same column spread over the banks (36 for d = 28/32, 52 for d = 48)
__host__ __device__ inline int attn_bwd_pitch(int d) {
    int ld = d + 4;
    if (((ld / 4) & 1) == 0) ld += 4;
    return ld;
}

__global__ void __launch_bounds__(256) attn_backward_kernel(const AttnBackwardParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = p.L, d = p.d, d4 = d / 4, ld = attn_bwd_pitch(d), L4 = (L + 3) / 4 * 4, lp = L4 + 1;
    float* Q = smem;  // [L4][ld] each, rows >= L zero
    float* Kk = Q + L4 * ld;
    float* V = Kk + L4 * ld;
    float* G = V + L4 * ld;   // dO
    float* P = G + L4 * ld;   // [L][lp]
    float* D = P + L * lp;    // dP, then dS
    const int tid = threadIdx.x;
    const int64_t item = blockIdx.x;
    const int64_t seq = item / p.heads;
    const int head = (int)(item % p.heads);
    const int64_t base = (seq / p.group) * p.group_stride + (seq % p.group) * p.seq_stride;
    const int C3 = 3 * p.C;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < L4 * d4; i += 256) {
        const int t = i / d4, c = 4 * (i % d4);
        f32x4 q = zero, k = zero, v = zero, g = zero;
        if (t < L) {
            const int64_t row = base + (int64_t)t * p.tok_stride;
            const float* src = p.qkv + row * C3 + head * d + c;
            q = *reinterpret_cast<const f32x4*>(src);
            k = *reinterpret_cast<const f32x4*>(src + p.C);
            v = *reinterpret_cast<const f32x4*>(src + 2 * p.C);
            g = *reinterpret_cast<const f32x4*>(p.d_o + row * p.C + head * d + c);
        }
        *reinterpret_cast<f32x4*>(Q + t * ld + c) = q;
        *reinterpret_cast<f32x4*>(Kk + t * ld + c) = k;
        *reinterpret_cast<f32x4*>(V + t * ld + c) = v;
        *reinterpret_cast<f32x4*>(G + t * ld + c) = g;
    }
    __syncthreads();
    // S = scale q k^T and dP = dO v^T: one query row x four keys per thread
    for (int i = tid; i < L * (L4 / 4); i += 256) {
        const int a = i / (L4 / 4), b0 = 4 * (i % (L4 / 4));
        float sv[4] = {0.f, 0.f, 0.f, 0.f}, gv[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < d; c += 4) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(Q + a * ld + c);
            const f32x4 g = *reinterpret_cast<const f32x4*>(G + a * ld + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 k = *reinterpret_cast<const f32x4*>(Kk + (b0 + j) * ld + c);
                const f32x4 v = *reinterpret_cast<const f32x4*>(V + (b0 + j) * ld + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) sv[j] += q[e] * k[e], gv[j] += g[e] * v[e];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (b0 + j < L) P[a * lp + b0 + j] = sv[j] * p.scale, D[a * lp + b0 + j] = gv[j];
    }
    __syncthreads();
    if (tid < L) {  // softmax of row tid, then dS in place
        float mx = -INFINITY;
        for (int b = 0; b < L; ++b) mx = fmaxf(mx, P[tid * lp + b]);
        float sum = 0.f;
        for (int b = 0; b < L; ++b) {
            const float e = expf(P[tid * lp + b] - mx);
            P[tid * lp + b] = e;
            sum += e;
        }
        const float inv = 1.0f / sum;
        float dot = 0.f;
        for (int b = 0; b < L; ++b) {
            const float pr = P[tid * lp + b] * inv;
            P[tid * lp + b] = pr;
            dot += pr * D[tid * lp + b];
        }
        for (int b = 0; b < L; ++b) D[tid * lp + b] = P[tid * lp + b] * (D[tid * lp + b] - dot) * p.scale;
    }
    __syncthreads();
    // dq = dS k, dk = dS^T q, dv = P^T dO: one token x four channels per thread
    for (int i = tid; i < L * d4; i += 256) {
        const int t = i / d4, c = 4 * (i % d4);
        f32x4 dq = zero, dk = zero, dv = zero;
        for (int b = 0; b < L; ++b) {
            const float s_tb = D[t * lp + b], s_bt = D[b * lp + t], p_bt = P[b * lp + t];
            const f32x4 k = *reinterpret_cast<const f32x4*>(Kk + b * ld + c);
            const f32x4 q = *reinterpret_cast<const f32x4*>(Q + b * ld + c);
            const f32x4 g = *reinterpret_cast<const f32x4*>(G + b * ld + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) dq[e] += s_tb * k[e], dk[e] += s_bt * q[e], dv[e] += p_bt * g[e];
        }
        const int64_t row = base + (int64_t)t * p.tok_stride;
        float* dst = p.dqkv + row * C3 + head * d + c;
        *reinterpret_cast<f32x4*>(dst) = dq;
        *reinterpret_cast<f32x4*>(dst + p.C) = dk;
        *reinterpret_cast<f32x4*>(dst + 2 * p.C) = dv;
    }
}

// The same on the matrix cores (v_mfma_f32_16x16x4_f32, fp32 products), nothing but Q, K, V, dO and three row vectors in LDS.
// One (sequence, head) per workgroup of T = LP/16 waves; wave w owns QUERY tile w in pass 1 and KEY tile w in pass 2:
//   pass 1  S^T = K Q^T and dP^T = V dO^T (the forward kernel's form: the query on the lane, the keys in the accumulator
//           registers): softmax statistics of the wave's 16 queries in registers + two xor-shuffles, dS in registers, and
//           dQ^T = K^T dS^T takes dS straight from the accumulators as its B operand.  max, 1/sum and rowsum(P dP) of
//           every query go to LDS.
//   pass 2  S = Q K^T and dP = dO V^T for the wave's 16 KEYS against all query tiles (the key on the lane, the queries in
//           the accumulator registers); P and dS are rebuilt from the statistics of pass 1 and feed dK^T = Q^T dS and
//           dV^T = dO^T P as B operands.
// S is computed twice (the products are 20 - 35 % of the MFMAs of an item) so that neither P nor dS ever has to be
// transposed through LDS.  Contraction indices are permuted the same way for both operands of every product (lane group g
// holds k = 16 s + 4 g + j, or the key / query 16 t + 4 g + reg), which is all an inner product needs.
template <int LP, int DP>
__global__ void __launch_bounds__((LP / 16) * 64) attn_backward_mfma_kernel(const AttnBackwardParams p) {
    constexpr int T = LP / 16, CT = DP / 16, SD = DP / 16, LDV = DP + 4, NTHR = T * 64, C4 = DP / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;               // [LP][LDV] each, zero-padded
    float* Ks = Qs + LP * LDV;
    float* Vs = Ks + LP * LDV;
    float* Gs = Vs + LP * LDV;      // dO
    float* St = Gs + LP * LDV;      // [3][LP]: row max, 1 / row sum, rowsum(P dP)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, g = lane >> 4;
    int64_t seq;
    int head;
    attn_item<1>((int64_t)blockIdx.x, p.nseq, p.heads, seq, head);
    const int64_t base = (seq / p.group) * p.group_stride + (seq % p.group) * p.seq_stride;
    const int C3 = 3 * p.C;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int idx = tid; idx < LP * C4; idx += NTHR) {
        const int t = idx / C4, c4 = idx % C4;
        f32x4 q = zero, k = zero, v = zero, gg = zero;
        if (t < p.L && 4 * c4 < p.d) {
            const int64_t row = base + (int64_t)t * p.tok_stride;
            const float* src = p.qkv + row * C3 + head * p.d + 4 * c4;
            q = *reinterpret_cast<const f32x4*>(src);
            k = *reinterpret_cast<const f32x4*>(src + p.C);
            v = *reinterpret_cast<const f32x4*>(src + 2 * p.C);
            gg = *reinterpret_cast<const f32x4*>(p.d_o + row * p.C + head * p.d + 4 * c4);
        }
        *reinterpret_cast<f32x4*>(Qs + t * LDV + 4 * c4) = q;
        *reinterpret_cast<f32x4*>(Ks + t * LDV + 4 * c4) = k;
        *reinterpret_cast<f32x4*>(Vs + t * LDV + 4 * c4) = v;
        *reinterpret_cast<f32x4*>(Gs + t * LDV + 4 * c4) = gg;
    }
    __syncthreads();
    constexpr float LOG2E = 1.44269504088896340736f;
    // ------------------------------------------------------------------ pass 1: query 16 w + l15 on the lane
    {
        f32x4 sc[T], dc[T];
#pragma unroll
        for (int kt = 0; kt < T; ++kt) sc[kt] = zero, dc[kt] = zero;
#pragma unroll
        for (int s = 0; s < SD; ++s) {
            const f32x4 qf = *reinterpret_cast<const f32x4*>(Qs + (16 * w + l15) * LDV + 16 * s + 4 * g);
            const f32x4 gf = *reinterpret_cast<const f32x4*>(Gs + (16 * w + l15) * LDV + 16 * s + 4 * g);
            f32x4 kf[T], vf[T];
#pragma unroll
            for (int kt = 0; kt < T; ++kt) {
                kf[kt] = *reinterpret_cast<const f32x4*>(Ks + (16 * kt + l15) * LDV + 16 * s + 4 * g);
                vf[kt] = *reinterpret_cast<const f32x4*>(Vs + (16 * kt + l15) * LDV + 16 * s + 4 * g);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int kt = 0; kt < T; ++kt) {
                    sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][j], qf[j], sc[kt], 0, 0, 0);
                    dc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kt][j], gf[j], dc[kt], 0, 0, 0);
                }
        }
        // sc[kt][reg] = <q, k>, dc[kt][reg] = <dO, v> for query 16 w + l15 and key 16 kt + 4 g + reg
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < T; ++kt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float v = 16 * kt + 4 * g + reg < p.L ? sc[kt][reg] * p.scale : -INFINITY;
                sc[kt][reg] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < T; ++kt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float e = __builtin_amdgcn_exp2f((sc[kt][reg] - mx) * LOG2E);   // 0 for the masked keys
                sc[kt][reg] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        float dot = 0.f;
#pragma unroll
        for (int kt = 0; kt < T; ++kt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                sc[kt][reg] *= inv;                       // P
                dot += sc[kt][reg] * dc[kt][reg];
            }
        dot += __shfl_xor(dot, 16);
        dot += __shfl_xor(dot, 32);
        if (g == 0) St[16 * w + l15] = mx, St[LP + 16 * w + l15] = inv, St[2 * LP + 16 * w + l15] = dot;
#pragma unroll
        for (int kt = 0; kt < T; ++kt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) sc[kt][reg] = sc[kt][reg] * (dc[kt][reg] - dot) * p.scale;   // dS (0 where P is 0)
        // dQ^T = K^T dS^T: the K element is the A operand (row = channel), dS the B operand (column = query)
        f32x4 dq[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) dq[ct] = zero;
#pragma unroll
        for (int kt = 0; kt < T; ++kt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float* krow = Ks + (16 * kt + 4 * g + reg) * LDV + l15;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    dq[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(krow[16 * ct], sc[kt][reg], dq[ct], 0, 0, 0);
            }
        // dq[ct][r] = dQ[query 16 w + l15][channel 16 ct + 4 g + r]
        const int q1 = 16 * w + l15;
        if (q1 < p.L) {
            float* dst = p.dqkv + (base + (int64_t)q1 * p.tok_stride) * C3 + head * p.d + 4 * g;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                if (16 * ct + 4 * g < p.d) *reinterpret_cast<f32x4*>(dst + 16 * ct) = dq[ct];
        }
    }
    __syncthreads();
    // ------------------------------------------------------------------ pass 2: key 16 w + l15 on the lane
    {
        f32x4 s2[T], d2[T];
#pragma unroll
        for (int qt = 0; qt < T; ++qt) s2[qt] = zero, d2[qt] = zero;
#pragma unroll
        for (int s = 0; s < SD; ++s) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + (16 * w + l15) * LDV + 16 * s + 4 * g);
            const f32x4 vf = *reinterpret_cast<const f32x4*>(Vs + (16 * w + l15) * LDV + 16 * s + 4 * g);
            f32x4 qf[T], gf[T];
#pragma unroll
            for (int qt = 0; qt < T; ++qt) {
                qf[qt] = *reinterpret_cast<const f32x4*>(Qs + (16 * qt + l15) * LDV + 16 * s + 4 * g);
                gf[qt] = *reinterpret_cast<const f32x4*>(Gs + (16 * qt + l15) * LDV + 16 * s + 4 * g);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int qt = 0; qt < T; ++qt) {
                    s2[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[qt][j], kf[j], s2[qt], 0, 0, 0);
                    d2[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[qt][j], vf[j], d2[qt], 0, 0, 0);
                }
        }
        // s2[qt][reg] = <q, k>, d2[qt][reg] = <dO, v> for query 16 qt + 4 g + reg and key 16 w + l15
        const int k1 = 16 * w + l15;
#pragma unroll
        for (int qt = 0; qt < T; ++qt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int q = 16 * qt + 4 * g + reg;
                const float pr = (q < p.L && k1 < p.L)
                                     ? __builtin_amdgcn_exp2f((s2[qt][reg] * p.scale - St[q]) * LOG2E) * St[LP + q] : 0.f;
                s2[qt][reg] = pr;                                           // P
                d2[qt][reg] = pr * (d2[qt][reg] - St[2 * LP + q]) * p.scale;   // dS
            }
        // dK^T = Q^T dS, dV^T = dO^T P: the Q / dO element is the A operand (row = channel), dS / P the B operand (column = key)
        f32x4 dk[CT], dv[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) dk[ct] = zero, dv[ct] = zero;
#pragma unroll
        for (int qt = 0; qt < T; ++qt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float* qrow = Qs + (16 * qt + 4 * g + reg) * LDV + l15;
                const float* grow = Gs + (16 * qt + 4 * g + reg) * LDV + l15;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    dk[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(qrow[16 * ct], d2[qt][reg], dk[ct], 0, 0, 0);
                    dv[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(grow[16 * ct], s2[qt][reg], dv[ct], 0, 0, 0);
                }
            }
        if (k1 < p.L) {
            float* dst = p.dqkv + (base + (int64_t)k1 * p.tok_stride) * C3 + head * p.d + 4 * g;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                if (16 * ct + 4 * g < p.d) {
                    *reinterpret_cast<f32x4*>(dst + p.C + 16 * ct) = dk[ct];
                    *reinterpret_cast<f32x4*>(dst + 2 * p.C + 16 * ct) = dv[ct];
                }
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Small reductions of the first / last layers
// ----------------------------------------------------------------------------------------------------------------
// partial[chunk][g][c] = sum over the chunk's share of the rows of group g of x[row][c].  Groups over the (b,f,j) rows:
//   mode 0: g = j (Spatial_pos_embed grad)   mode 1: g = f (Temporal_pos_embed grad)   mode 2: g = b (time embedding)
__global__ void __launch_bounds__(256) group_sum_kernel(const float* x, float* partial, int mode, int B, int F, int J,
                                                        int C, int G, int per_chunk) {
    const int g = blockIdx.x, chunk = blockIdx.y;
    const int count = mode == 0 ? B * F : (mode == 1 ? B * J : F * J);
    const int lo = chunk * per_chunk, hi = lo + per_chunk < count ? lo + per_chunk : count;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int i = lo; i < hi; ++i) {
            int64_t row;
            if (mode == 0)
                row = (int64_t)i * J + g;
            else if (mode == 1)
                row = ((int64_t)(i / J) * F + g) * J + i % J;
            else
                row = (int64_t)g * F * J + i;
            s += x[row * C + c];
        }
        partial[((int64_t)chunk * G + g) * C + c] = s;
    }
}

// partial[chunk][i][c] = sum over the chunk's rows of s[row][i] * x[row][c]   (i < NI <= 8):
// head.1 weight grad (s = d_out [M,3], x = head-norm output) and the patch-embedding weight grad (s = the 5 inputs).
__global__ void __launch_bounds__(256) outer_sum_kernel(const float* s, const float* x, float* partial, int64_t M, int NI,
                                                        int C, int64_t rows_per_chunk) {
    const int64_t lo = (int64_t)blockIdx.x * rows_per_chunk;
    const int64_t hi = lo + rows_per_chunk < M ? lo + rows_per_chunk : M;
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int64_t m = lo; m < hi; ++m) {
            const float xv = x[m * C + c];
            for (int i = 0; i < NI; ++i) acc[i] += s[m * NI + i] * xv;
        }
        for (int i = 0; i < NI; ++i) partial[((int64_t)blockIdx.x * NI + i) * C + c] = acc[i];
    }
}

// head.1 backward: dhn[row][c] = sum_i dout[row][i] w[i][c]   (the forward lives in the GEMM epilogue)
__global__ void __launch_bounds__(256) head_backward_kernel(const float* dout, const float* w, float* dhn, int64_t M,
                                                            int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * C) return;
    const int64_t row = i / C;
    const int c = (int)(i % C);
    dhn[i] = (dout[row * 3] * w[c] + dout[row * 3 + 1] * w[C + c]) + dout[row * 3 + 2] * w[2 * C + c];
}

// in5[row] = (x2d[b,f,j,:], x3d[b,f,j,:]) - the input of Spatial_patch_to_embedding, kept for its weight gradient
__global__ void __launch_bounds__(256) concat_inputs_kernel(const float* x2d, const float* x3d, float* in5, int64_t M) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= M) return;
    in5[row * 5 + 0] = x2d[row * 2], in5[row * 5 + 1] = x2d[row * 2 + 1];
    in5[row * 5 + 2] = x3d[row * 3], in5[row * 5 + 3] = x3d[row * 3 + 1], in5[row * 5 + 4] = x3d[row * 3 + 2];
}

// sinusoid[b] = [sin(t_b w), cos(t_b w)]   (common/mixste.py:132-138)
__global__ void __launch_bounds__(256) sinusoid_kernel(const int64_t* t, const float* freqs, float* out, int B, int C) {
    const int i = blockIdx.x * 256 + threadIdx.x, half = C / 2;
    if (i >= B * half) return;
    const int b = i / half, k = i % half;
    const float a = (float)t[b] * freqs[k];
    out[b * C + k] = sinf(a);
    out[b * C + half + k] = cosf(a);
}

// adds a small [NI][C] sum, transposed, into a [C][NI] weight gradient (Spatial_patch_to_embedding.weight is [C,5])
__global__ void __launch_bounds__(256) transpose_small_kernel(const float* in, float* out, int NI, int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NI * C) return;
    out[(i % C) * NI + i / C] += in[i];
}

// D3DP.prepare_diffusion_concat + q_sample (common/diffusionpose.py:319-326,358-374): per sample b
//   x = clamp(sqrt_acp[t_b] * (x0 * scale) + sqrt_1m_acp[t_b] * noise, +-1.1 scale) / scale, fp64 buffers, cast to fp32
__global__ void __launch_bounds__(256) qsample_kernel(const float* x0, const float* noise, const int64_t* t,
                                                      const double* sqrt_acp, const double* sqrt_1m_acp, double scale,
                                                      float* out, int64_t per_sample, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t tb = t[i / per_sample];
    const float xs = __fmul_rn(x0[i], (float)scale);  // fp32 tensor * python scalar stays fp32
    double v = __dadd_rn(__dmul_rn(sqrt_acp[tb], (double)xs), __dmul_rn(sqrt_1m_acp[tb], (double)noise[i]));
    const double lim = 1.1 * scale;
    v = v < -lim ? -lim : (v > lim ? lim : v);
    out[i] = (float)(v / scale);
}

}  // namespace pafuse
