// cm_mlp_wide.h -- the LAYERED schedule: MLPs the fused kernel (cm_mlp_kernel.h) does not cover, i.e. hidden widths 65..256
// (the reference's COMA critic defaults to 128, cleanmarl/coma_multienvs.py:35) or more than two hidden->hidden layers.
//
// The fused kernel keeps a 64-wide network resident in LDS; a 128..256-wide one does not fit next to its activations, so this
// schedule runs layer by layer with activations in HBM (caller's workspace; 288 GB makes rows x 256 floats per layer cheap):
//   forward   act_0 = relu(X W0^T + b0), act_l = relu(act_{l-1} Wl^T + bl), out = act_L Wout^T + bout      k_wide_gemm
//   loss      per-row head math of the four training modes (same formulas as the fused epilogue)           k_wide_loss<MODE>
//   backward  dW = dZ^T A (k_dw0_stream, the streaming MFMA GEMM of the split schedule, one 64-unit slab at a time),
//             dZ_{l} = (dZ_{l+1} W) .* relu'(act_l) (k_wide_gemm on a transposed weight copy; its epilogue also accumulates the
//             column sums of dZ_l = the bias gradient of the layer below), db_out = k_wide_colsum
// Same MFMA (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation) and the same deterministic
// per-workgroup-partials + ordered-reduce pattern as the fused path.  HBM-bound by design (every activation makes a round trip; the
// GEMM epilogue is staged through LDS so that those round trips are whole rows): it exists so that every width the reference's CLI
// accepts up to 256 runs, not to be the fast path.
#pragma once
#include "cm_mlp_split.h"

namespace {

constexpr int WT_M = 128;   // rows per workgroup tile (4 waves x 32 rows)
#ifndef CM_WT_K
#define CM_WT_K 32
#endif
constexpr int WT_K = CM_WT_K;      // contraction chunk (32 or 64)
constexpr int WT_LD = WT_K + 4;    // LDS row stride (36 / 68): 16-lane groups of ds_read_b128 down a column of rows hit 64 distinct banks
constexpr int WT_KQ = WT_K / 2;    // operands per lane and chunk (lane (lc, h) takes k = WT_KQ h + kk)
constexpr int WT_TPR = WT_K / 4;   // loader threads per row (one float4 each)
constexpr int WT_RPP = NTHREADS / WT_TPR;  // rows per loader pass
constexpr int WT_XP = WT_M / WT_RPP;       // loader passes over the X tile
constexpr int WIDE_HMAX = 256;
enum WideEpi { EPI_BIAS_RELU = 0, EPI_BIAS = 1, EPI_GATE = 2, EPI_NONE = 3, EPI_ADD = 4, EPI_COMA = 5, EPI_COMA16 = 6 };  // BIAS_RELU adds `gate` (if set) as a pre-activation addend; ADD: product + gate, nothing else
// EPI_COMA (cm_coma.hip, the factored critic input): the product's rows are S[e,t] = state W0s^T; the epilogue writes the A rows
// z0[(e,a,t)] = S[e,t] + sum_{j != a} W0a[:, slot(j,a) K + u_j] straight from the staged tile -- S never reaches HBM and the separate
// k_coma_z0_add pass (one more 2 GB stream at config-3 shapes and 128 units) is gone.  The sums run in k_coma_z0_add's order (bit-identical).
struct WideComa { const int* action; const float* W0; int A, T, Kact, Dc, col0, H; };
constexpr int COMA_EPI_MAXA = 16;  // EPI_COMA serves up to 8 agents, EPI_COMA16 up to 16 (twice the registers and unrolled guards per row)

#define WIDE_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

constexpr int KWMAX = 64;  // widest head of the layered schedule (the fused kernels' heads stop at KMAX = 32: SMAC maps with more than 26 enemies)
inline int wide_kw(int dout) { return dout <= KMAX ? KMAX : KWMAX; }  // row stride of the logits / dlogits plane
inline bool wide_shape(int H, int L, int dout = 1) { return H > HP || L > LMAX || dout > KMAX; }
inline int wide_hs(int H) { return (H + 63) / 64 * 64; }  // activation row stride (zero padded; whole 64-unit slabs)

// Y[r][n] = epi( sum_k X[r][k] * W[n][k] ),  n < N <= 32 * NJ, k < K.  Columns N..ncols-1 of Y are written as zeros.
// Any k permutation is a valid contraction order as long as both operands use it: lane (lc, h) takes k = WT_KQ h + kk at step
// kk, so its WT_KQ operands of a chunk are contiguous in LDS (ds_read_b128).
template <int NJ, int EPI>
__global__ __launch_bounds__(NTHREADS, (NJ > 6 ? 1 : 2)) void k_wide_gemm(const float* __restrict__ X, long ldx, long rows, int K,
                                                           const float* __restrict__ W, int ldw, int N,
                                                           const float* __restrict__ bias, const uint8_t* __restrict__ avail, long lda,
                                                           const float* __restrict__ gate, long ldg,
                                                           float* __restrict__ Y, long ldy, int ncols, int vecx, int vecw,
                                                           int vecy, int vecg, float* __restrict__ colsum_part, const WideComa cx) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;
    float* Ws = smem + WT_M * WT_LD;
    constexpr int TLD = 32 * NJ;  // EPI_COMA: row stride of the transposed action block of W0, parked behind the operand / staging area
    float* tabs = smem + (((WT_M + 32 * NJ) * WT_LD > 64 * (32 * NJ + 8)) ? (WT_M + 32 * NJ) * WT_LD : 64 * (32 * NJ + 8));
    constexpr bool COMA = EPI == EPI_COMA || EPI == EPI_COMA16;
    constexpr int CMAXA = EPI == EPI_COMA16 ? 16 : 8;
    if constexpr (COMA) {
        const int Da = (cx.A - 1) * cx.Kact;
        for (int i = threadIdx.x; i < Da * TLD; i += NTHREADS) {
            const int c = i / TLD, hh = i - c * TLD;
            tabs[i] = hh < cx.H ? cx.W0[(long)hh * cx.Dc + cx.col0 + c] : 0.0f;
        }
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lc = lane & 31, h = lane >> 5;
    const int lr = tid / WT_TPR, lk = 4 * (tid % WT_TPR);  // loader: row lr + WT_RPP i, floats lk..lk+3 of the chunk
    constexpr int WP = 32 * NJ / WT_RPP;                   // loader passes over the W tile
    const int nk = (K + WT_K - 1) / WT_K;
    const long ntiles = (rows + WT_M - 1) / WT_M;
    // epilogue geometry: a lane owns the float4 column group c4 of the rows rsub, rsub + RPI, ... of a 16-row half tile
    constexpr int CG = 8 * NJ, CGP = CG <= 8 ? 8 : (CG <= 16 ? 16 : (CG <= 32 ? 32 : 64)), RPI = 64 / CGP, SLD = 32 * NJ + 8;
    const int c4 = lane % CGP, rsub = lane / CGP;
    float bv4[4], cs4[4];  // bias of the lane's columns; EPI_GATE: their column sums (= the bias gradient of the layer below)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        cs4[q] = 0.0f;
        bv4[q] = ((EPI == EPI_BIAS_RELU || EPI == EPI_BIAS) && c4 < CG && 4 * c4 + q < N) ? bias[4 * c4 + q] : 0.0f;
    }
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = tile * WT_M;
        f32x16 acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[j][g] = 0.0f;
        float4 xr[WT_XP], wr[WP];
        auto load = [&](int c) {
            const int k0 = c * WT_K + lk;
#pragma unroll
            for (int i = 0; i < WT_XP; ++i) {
                const long row = row0 + lr + WT_RPP * i;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < rows) {
                    const float* p = X + row * ldx + k0;
                    if (vecx) { if (k0 < K) { v = *reinterpret_cast<const float4*>(p); if (k0 + 3 >= K) { if (k0 + 1 >= K) v.y = 0.f; if (k0 + 2 >= K) v.z = 0.f; v.w = 0.f; } } }
                    else {
                        if (k0 < K) v.x = p[0];
                        if (k0 + 1 < K) v.y = p[1];
                        if (k0 + 2 < K) v.z = p[2];
                        if (k0 + 3 < K) v.w = p[3];
                    }
                }
                xr[i] = v;
            }
#pragma unroll
            for (int i = 0; i < WP; ++i) {
                const int n = lr + WT_RPP * i;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < N) {
                    const float* p = W + (long)n * ldw + k0;
                    if (vecw) { if (k0 < K) { v = *reinterpret_cast<const float4*>(p); if (k0 + 3 >= K) { if (k0 + 1 >= K) v.y = 0.f; if (k0 + 2 >= K) v.z = 0.f; v.w = 0.f; } } }
                    else {
                        if (k0 < K) v.x = p[0];
                        if (k0 + 1 < K) v.y = p[1];
                        if (k0 + 2 < K) v.z = p[2];
                        if (k0 + 3 < K) v.w = p[3];
                    }
                }
                wr[i] = v;
            }
        };
        load(0);
        for (int c = 0; c < nk; ++c) {
            __syncthreads();  // the previous chunk's (or tile's) readers are done
#pragma unroll
            for (int i = 0; i < WT_XP; ++i) *reinterpret_cast<float4*>(Xs + (lr + WT_RPP * i) * WT_LD + lk) = xr[i];
#pragma unroll
            for (int i = 0; i < WP; ++i) *reinterpret_cast<float4*>(Ws + (lr + WT_RPP * i) * WT_LD + lk) = wr[i];
            __syncthreads();
            if (c + 1 < nk) load(c + 1);  // in flight under this chunk's MFMAs
            float a[WT_KQ];
#pragma unroll
            for (int q = 0; q < WT_KQ / 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(Xs + (32 * wave + lc) * WT_LD + WT_KQ * h + 4 * q);
                a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float b[WT_KQ];
#pragma unroll
                for (int q = 0; q < WT_KQ / 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(Ws + (32 * j + lc) * WT_LD + WT_KQ * h + 4 * q);
                    b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
                }
#pragma unroll
                for (int kk = 0; kk < WT_KQ; ++kk) acc[j] = mfma32(a[kk], b[kk], acc[j]);
            }
        }
        // epilogue through LDS so that HBM sees whole rows: acc[j][g] = Y[row0 + 32 wave + (g&3) + 8 (g>>2) + 4h][32 j + lc] would
        // store 128-byte row segments; instead each wave parks 16 rows at a time in its own slice of the (now dead) operand
        // buffers and reads them back row-contiguous, one float4 per lane (addend / gate / output: 16-byte accesses, full rows).
        __syncthreads();  // every wave is done with Xs / Ws
        {
            float* stg = smem + wave * (16 * SLD);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int gg = 0; gg < 8; ++gg) stg[((gg & 3) + 8 * (gg >> 2) + 4 * h) * SLD + 32 * j + lc] = acc[j][8 * p + gg];
                WIDE_WAVE_SYNC();
                if (c4 < CG) {
#pragma unroll
                    for (int it = 0; it < 16 / RPI; ++it) {
                        const int rl = it * RPI + rsub;
                        const long row = row0 + 32 * wave + 16 * p + rl;
                        if (row < rows) {
                            const float4 t = *reinterpret_cast<const float4*>(stg + rl * SLD + 4 * c4);
                            if constexpr (COMA) {
                                const int A_ = cx.A, T_ = cx.T, Ka = cx.Kact;
                                const long e_ = row / T_;
                                const int t_ = (int)(row - e_ * T_);
                                int u[CMAXA];
#pragma unroll
                                for (int j = 0; j < CMAXA; ++j) u[j] = j < A_ ? cx.action[(e_ * A_ + j) * T_ + t_] : 0;
                                const float* tb = tabs + 4 * c4;
                                float4 Q[CMAXA - 1];  // Q[j-1]: agent j's action in the column block the agents before it see
#pragma unroll
                                for (int j = 1; j < CMAXA; ++j)
                                    Q[j - 1] = j < A_ ? *reinterpret_cast<const float4*>(tb + ((j - 1) * Ka + u[j]) * TLD) : make_float4(0.f, 0.f, 0.f, 0.f);
                                float4 pre = t;  // S + the blocks of the agents 0 .. a-1 (as the agents behind them see them)
#pragma unroll
                                for (int ag = 0; ag < CMAXA; ++ag) {
                                    if (ag < A_) {
                                        float4 z = pre;
#pragma unroll
                                        for (int j = 1; j < CMAXA; ++j)
                                            if (j > ag && j < A_) { z.x += Q[j - 1].x; z.y += Q[j - 1].y; z.z += Q[j - 1].z; z.w += Q[j - 1].w; }
                                        float* yp = Y + ((e_ * A_ + ag) * T_ + t_) * ldy + 4 * c4;
                                        if (vecy && 4 * c4 + 3 < ncols) *reinterpret_cast<float4*>(yp) = z;
                                        else {
                                            if (4 * c4 < ncols) yp[0] = z.x;
                                            if (4 * c4 + 1 < ncols) yp[1] = z.y;
                                            if (4 * c4 + 2 < ncols) yp[2] = z.z;
                                            if (4 * c4 + 3 < ncols) yp[3] = z.w;
                                        }
                                        if (ag < A_ - 1) {
                                            const float4 pa = *reinterpret_cast<const float4*>(tb + (ag * Ka + u[ag]) * TLD);
                                            pre.x += pa.x; pre.y += pa.y; pre.z += pa.z; pre.w += pa.w;
                                        }
                                    }
                                }
                                continue;
                            }
                            float v[4] = {t.x, t.y, t.z, t.w};
                            float gt[4] = {0.f, 0.f, 0.f, 0.f};
                            const bool has_g = (EPI == EPI_GATE) || ((EPI == EPI_BIAS_RELU || EPI == EPI_ADD) && gate != nullptr);
                            if (has_g) {
                                const float* gp = gate + row * ldg + 4 * c4;
                                if (vecg && 4 * c4 + 3 < N) { const float4 q4 = *reinterpret_cast<const float4*>(gp); gt[0] = q4.x; gt[1] = q4.y; gt[2] = q4.z; gt[3] = q4.w; }
                                else {
#pragma unroll
                                    for (int q = 0; q < 4; ++q) if (4 * c4 + q < N) gt[q] = gp[q];
                                }
                            }
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int col = 4 * c4 + q;
                                const bool cv = col < N;
                                float x = v[q] + bv4[q];
                                if (EPI == EPI_BIAS_RELU) x = fmaxf(x + gt[q], 0.0f);  // gt = COMA's factored layer-0 addend (or 0)
                                if (EPI == EPI_ADD) x += gt[q];
                                if (EPI == EPI_BIAS && avail && cv && !avail[row * lda + col]) x = -1e9f;  // masked_fill(~avail, -1e9)
                                if (EPI == EPI_GATE) { x = gt[q] > 0.0f ? x : 0.0f; cs4[q] += cv ? x : 0.0f; }
                                v[q] = cv ? x : 0.0f;
                            }
                            float* yp = Y + row * ldy + 4 * c4;
                            if (vecy && 4 * c4 + 3 < ncols) *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
                            else {
#pragma unroll
                                for (int q = 0; q < 4; ++q) if (4 * c4 + q < ncols) yp[q] = v[q];
                            }
                        }
                    }
                }
                WIDE_WAVE_SYNC();
            }
        }
        // the next tile's first barrier (top of its chunk loop) orders these LDS reads before its operand stores
    }
    if (EPI == EPI_GATE && colsum_part) {  // per-workgroup partial [32 NJ]: (4 waves x RPI row groups) summed in a fixed order
        __syncthreads();
        if (c4 < CG) {
#pragma unroll
            for (int q = 0; q < 4; ++q) smem[(wave * RPI + rsub) * 32 * NJ + 4 * c4 + q] = cs4[q];
        }
        __syncthreads();
        for (int i = tid; i < 32 * NJ; i += NTHREADS) {
            float t = 0.0f;
            for (int k = 0; k < 4 * RPI; ++k) t += smem[k * 32 * NJ + i];
            colsum_part[(long)blockIdx.x * 32 * NJ + i] = t;
        }
    }
}

// operand tiles (X: WT_M rows, W: 32 NJ rows, stride WT_LD) or the epilogue's four 16-row staging slices (stride 32 NJ + 8)
inline size_t wide_gemm_lds(int nj) {
    const size_t ops = (size_t)(WT_M + 32 * nj) * WT_LD, stage = (size_t)64 * (32 * nj + 8);
    return (ops > stage ? ops : stage) * sizeof(float);
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <int EPI>
inline void wide_gemm(const float* X, long ldx, long rows, int K, const float* W, int ldw, int N, const float* bias,
                      const uint8_t* avail, long lda, const float* gate, long ldg, float* Y, long ldy, int ncols, hipStream_t s,
                      float* colsum_part = nullptr, float* colsum_out = nullptr, const WideComa* coma = nullptr) {
    const int nj = (max(N, ncols) + 31) / 32;
    const WideComa cx = coma ? *coma : WideComa{};
    const size_t tab_bytes = coma ? (size_t)(cx.A - 1) * cx.Kact * 32 * nj * sizeof(float) : 0;
    // 16-byte loads: aligned rows whose last float4 of the contraction stays inside the row -- K a multiple of 4, or a padded leading
    // dimension.  The lanes of the tail quad beyond K are zeroed AFTER the load (a few v_cndmask in the last chunk only), so the padding
    // columns of a caller's buffer may hold anything -- torch.empty garbage, Inf, NaN -- without reaching a product
    const long k4 = (K + 3) & ~3;
    const int vecx = (ldx % 4 == 0 && (K % 4 == 0 || ldx >= k4) && al16(X)) ? 1 : 0;
    const int vecw = (ldw % 4 == 0 && (K % 4 == 0 || (long)ldw >= k4) && al16(W)) ? 1 : 0;
    const int vecy = (ldy % 4 == 0 && al16(Y)) ? 1 : 0;
    const int vecg = (gate && ldg % 4 == 0 && al16(gate)) ? 1 : 0;
    const long ntiles = (rows + WT_M - 1) / WT_M;
    const int grid = (int)min(ntiles, 512L);
#define CM_WIDE_CASE(NJ)                                                                                                        \
    case NJ: {                                                                                                                  \
        const size_t lds = wide_gemm_lds(NJ) + tab_bytes;                                                                       \
        if (lds > 64 * 1024)                                                                                                    \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wide_gemm<NJ, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_wide_gemm<NJ, EPI>), dim3(grid), dim3(NTHREADS), lds, s, X, ldx, rows, K, W, ldw, N, bias, avail, \
                           lda, gate, ldg, Y, ldy, ncols, vecx, vecw, vecy, vecg, colsum_part, cx);                                  \
    } break;
    switch (nj) {
        CM_WIDE_CASE(1) CM_WIDE_CASE(2) CM_WIDE_CASE(3) CM_WIDE_CASE(4) CM_WIDE_CASE(5) CM_WIDE_CASE(6) CM_WIDE_CASE(7)
        default: { constexpr int NJ8 = 8;
            const size_t lds = wide_gemm_lds(NJ8) + tab_bytes;
            if (lds > 64 * 1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wide_gemm<NJ8, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((k_wide_gemm<NJ8, EPI>), dim3(grid), dim3(NTHREADS), lds, s, X, ldx, rows, K, W, ldw, N, bias, avail,
                               lda, gate, ldg, Y, ldy, ncols, vecx, vecw, vecy, vecg, colsum_part, cx);
        } break;
    }
#undef CM_WIDE_CASE
    if (colsum_part)  // EPI_GATE only: fold the per-workgroup column sums (fixed order)
        hipLaunchKernelGGL(k_reduce_partials, dim3((N + RED_COLS - 1) / RED_COLS), dim3(RED_COLS * RED_GROUPS), 0, s, colsum_part, grid, 32 * nj, 0, N,
                           colsum_out);
}

// whether the one-launch S + z0 GEMM serves a COMA critic of A agents, Kact actions and row stride hs (LDS: operand tiles + the action table)
inline bool coma_epi_fits(int A, int Kact, int hs) {
    const int nj = (hs + 31) / 32;
    return A <= COMA_EPI_MAXA && nj <= 8 && wide_gemm_lds(nj) + (size_t)(A - 1) * Kact * 32 * nj * sizeof(float) <= 160 * 1024;
}
template <int DUMMY = 0>
inline void wide_gemm_coma(const float* X, long ldx, long rows, int K, const float* W, int ldw, int N, float* Y, long ldy, int ncols, hipStream_t s,
                           const WideComa& cx) {
    if (cx.A <= 8) wide_gemm<EPI_COMA>(X, ldx, rows, K, W, ldw, N, nullptr, nullptr, 0, nullptr, 0, Y, ldy, ncols, s, nullptr, nullptr, &cx);
    else wide_gemm<EPI_COMA16>(X, ldx, rows, K, W, ldw, N, nullptr, nullptr, 0, nullptr, 0, Y, ldy, ncols, s, nullptr, nullptr, &cx);
}

// Wt[k][n] = W[n][k] (n < N, k < K), row stride ldt >= N, columns N..ldt-1 zeroed
__global__ void k_wide_transpose(const float* __restrict__ W, int N, int K, float* __restrict__ Wt, int ldt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * ldt) return;
    const int k = i / ldt, n = i - k * ldt;
    Wt[i] = n < N ? W[n * K + k] : 0.0f;
}

// partial[wg][c] = sum over the workgroup's rows of Z[row][c], c < N (N <= 256); 256 threads = NC columns x 256/NC row groups
constexpr int CS_GRID = 1024;
__global__ __launch_bounds__(NTHREADS) void k_wide_colsum(const float* __restrict__ Z, long ldz, long rows, int N, int NC,
                                                          long rows_per_wg, float* __restrict__ partial) {
    __shared__ float sh[NTHREADS];
    const int c = threadIdx.x % NC, rg = threadIdx.x / NC, RG = NTHREADS / NC;
    const long lo = (long)blockIdx.x * rows_per_wg, hi = min(rows, lo + rows_per_wg);
    float s = 0.0f;
    if (c < N)
        for (long r = lo + rg; r < hi; r += RG) s += Z[r * ldz + c];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (rg == 0) {
        for (int g = 1; g < RG; ++g) s += sh[g * NC + c];
        partial[(long)blockIdx.x * NC + c] = s;
    }
}

inline int wide_colsum(const float* Z, long ldz, long rows, int N, float* partial, float* out, hipStream_t s) {
    int NC = 32;
    while (NC < N) NC *= 2;
    long rpw = (rows + CS_GRID - 1) / CS_GRID;
    if (rpw < 64) rpw = 64;
    const int grid = (int)((rows + rpw - 1) / rpw);
    hipLaunchKernelGGL(k_wide_colsum, dim3(grid), dim3(NTHREADS), 0, s, Z, ldz, rows, N, NC, rpw, partial);
    hipLaunchKernelGGL(k_reduce_partials, dim3((N + RED_COLS - 1) / RED_COLS), dim3(RED_COLS * RED_GROUPS), 0, s, partial, grid, NC, 0, N, out);
    return 0;
}

// ---- per-row loss heads: one thread per row, logits in out[row][32] -> d(loss)/d(logits) written in place; statistics as
// per-workgroup partials [grid][8] = {pg, entropy, kl, clipfrac, value loss, count, 0, 0}.  Formulas: the fused epilogue's
// (cm_mlp_kernel.h; cleanmarl/mappo_multienvs.py:527-576, cleanmarl/coma_multienvs.py:620-631, :649-676).
constexpr int LOSS_GRID = 2048;
// one row's head: z = the row's (masked) logits / values, overwritten with d(loss)/dz; st = {pg, entropy, kl, clipfrac, value loss, count}
// KCAP: compile-time bound of the head width (8 keeps a K <= 8 head in a few registers)
template <int MODE, int KCAP = KMAX>
__device__ __forceinline__ void wide_loss_row(const MlpArgs& a, long row, float* z, float (&st)[6]) {
    const int dout = a.dout;
    const int Aseq = (MODE == M_CRITIC && !a.per_agent) ? 1 : a.A;
    const float invA = 1.0f / (float)a.A;
    float &st_pg = st[0], &st_ent = st[1], &st_kl = st[2], &st_clip = st[3], &st_vl = st[4], &st_cnt = st[5];
    {
        const long seq = row / a.T;
        const int t = (int)(row - seq * a.T);
        const int e = (int)(seq / Aseq), ag = (int)(seq - (long)e * Aseq);
        const bool valid = t < a.ep_len[e];
        if (MODE == M_CRITIC) {
            float d = 0.0f;
            if (valid) {
                const float v = z[0];
                if (a.per_agent) {
                    const float df = v - a.ret[row];
                    st_vl += invA * df * df;
                    d = 2.0f * invA * df;
                    if (ag == 0) st_cnt += 1.0f;
                } else {
                    float sd = 0.0f, sq = 0.0f;
                    for (int q = 0; q < a.A; ++q) {
                        const float df = v - a.ret[((long)e * a.A + q) * a.T + t];
                        sd += df; sq += df * df;
                    }
                    st_vl += invA * sq;
                    d = 2.0f * invA * sd;
                    st_cnt += 1.0f;
                }
            }
            z[0] = d;
            return;
        }
        const int act = a.action[row];
        if (MODE == M_QCRITIC) {
            const float df = z[act] - a.ret[row];
            if (valid) {
                st_vl += invA * df * df;
                if (ag == 0) st_cnt += 1.0f;
            }
            for (int k = 0; k < dout; ++k) z[k] = (valid && k == act) ? 2.0f * invA * df : 0.0f;
            return;
        }
        // actors: logits were masked (-1e9) by the head GEMM's epilogue
        float zr[KCAP], p[KCAP];
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < KCAP; ++k) { zr[k] = k < dout ? z[k] : -INFINITY; m = fmaxf(m, zr[k]); }
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < KCAP; ++k) { p[k] = k < dout ? expf(zr[k] - m) : 0.0f; s += p[k]; }
        const float rs = 1.0f / s;
        const float advv = a.adv[row];
        if (MODE == M_ACTOR) {
            const float lse = m + logf(s);
            float ent = 0.0f, lpa = 0.0f;
#pragma unroll
            for (int k = 0; k < KCAP; ++k)
                if (k < dout) {
                    const float lp = zr[k] - lse;
                    p[k] *= rs;
                    ent -= p[k] * lp;
                    if (k == act) lpa = lp;
                }
            const float log_ratio = lpa - a.logp_old[row];
            const float ratio = expf(log_ratio);
            const float pg1 = advv * ratio;
            const float pg2 = advv * fminf(fmaxf(ratio, a.clip_lo), a.clip_hi);
            const bool inr = (ratio >= a.clip_lo) && (ratio <= a.clip_hi);
            float g;  // d min(pg1, pg2) / d ratio with torch's tie rule (half the gradient to each operand)
            if (pg1 < pg2) g = advv;
            else if (pg1 > pg2) g = inr ? advv : 0.0f;
            else g = 0.5f * advv + (inr ? 0.5f * advv : 0.0f);
            if (valid) {
                st_pg += invA * fminf(pg1, pg2);
                st_ent += invA * ent;
                st_kl += invA * ((ratio - 1.0f) - log_ratio);
                st_clip += (fabsf(ratio - 1.0f) > a.clip_eps) ? invA : 0.0f;
                if (ag == 0) st_cnt += 1.0f;
            }
            const float gr = g * ratio;
#pragma unroll
            for (int k = 0; k < KCAP; ++k)
                if (k < dout) {
                    const float lp = zr[k] - lse;
                    float d = invA * (-gr * ((k == act ? 1.0f : 0.0f) - p[k]) + a.ent_coef * p[k] * (lp + ent));
                    if (!valid || zr[k] <= -5e8f) d = 0.0f;  // padded rows; masked_fill blocks the gradient
                    z[k] = d;
                }
        } else {  // M_COMA_ACTOR
            const float invK = 1.0f / (float)dout;
            float lq[KCAP];
            float ent = 0.0f, lpa = 0.0f, pa = 0.0f;
#pragma unroll
            for (int k = 0; k < KCAP; ++k) {
                lq[k] = 0.0f;
                if (k < dout) {
                    p[k] *= rs;
                    lq[k] = logf(p[k] + 1e-8f);
                    ent -= p[k] * lq[k];
                    if (k == act) { lpa = lq[k]; pa = p[k]; }
                }
            }
            ent *= invK;
            if (valid) {
                st_pg += lpa * advv;
                st_ent += ent;
                if (ag == 0) st_cnt += 1.0f;
            }
            float gbar = 0.0f;
#pragma unroll
            for (int k = 0; k < KCAP; ++k)
                if (k < dout) {
                    float gk = a.ent_coef * invK * (lq[k] + p[k] / (p[k] + 1e-8f));
                    if (k == act) gk -= advv / (pa + 1e-8f);
                    lq[k] = gk;
                    gbar += gk * p[k];
                }
#pragma unroll
            for (int k = 0; k < KCAP; ++k)
                if (k < dout) {
                    float d = p[k] * (lq[k] - gbar);
                    if (!valid || zr[k] <= -5e8f) d = 0.0f;
                    z[k] = d;
                }
        }
    }
}

template <int MODE, int KCAP = KMAX>
__global__ __launch_bounds__(NTHREADS) void k_wide_loss(const MlpArgs a, float* __restrict__ out, float* __restrict__ partial) {
    __shared__ float red[6][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float sv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long row = (long)blockIdx.x * NTHREADS + tid; row < a.rows; row += (long)gridDim.x * NTHREADS)
        wide_loss_row<MODE, KCAP>(a, row, out + row * KCAP, sv);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const float v = cm_wave_sum(sv[i]);
        if (lane == 0) red[i][wave] = v;
    }
    __syncthreads();
    if (tid < CM_NUM_STATS) partial[(long)blockIdx.x * CM_NUM_STATS + tid] = tid < 6 ? red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3] : 0.0f;
}

}  // namespace
#include "cm_mlp_fused128.h"
namespace {

// ---- workspace carve (floats)
struct WideWs { size_t act, dz, out, wt, part, total; int Hs; };
inline WideWs wide_ws(long rows, int din, int H, int L, int dout, bool train, bool with_out = false) {
    WideWs w; size_t p = 0;
    w.Hs = wide_hs(H);
    const size_t plane = ((size_t)rows * w.Hs + 63) / 64 * 64;
    w.act = p; p += (train ? (size_t)(L + 1) : 2) * plane;  // forward only: two ping-pong planes
    w.dz = p; if (train) p += 2 * plane;
    w.out = p; if (train || with_out) p += ((size_t)rows * wide_kw(dout) + 63) / 64 * 64;
    w.wt = p; if (train) p += (size_t)WIDE_HMAX * WIDE_HMAX;
    const int kmax = max(din, w.Hs);
    w.part = p;
    if (train) p += max(max((size_t)DW0_GRID * 64 * kmax, (H > HP && H <= F_HP && L == 1) ? fused128_part_floats(rows, din, H, dout) : (size_t)0),
                        max((size_t)CS_GRID * WIDE_HMAX, (size_t)LOSS_GRID * CM_NUM_STATS));
    w.total = p;
    return w;
}
inline size_t wide_ws_bytes(long rows, int din, int H, int L, int dout, bool train) { return wide_ws(rows, din, H, L, dout, train).total * sizeof(float); }

inline int wide_check(const char* who, int din, int H, int L, int dout) {
    CM_REQUIRE(din > 0 && H > 0 && L >= 0 && dout > 0, "%s: bad dims din=%d H=%d L=%d dout=%d", who, din, H, L, dout);
    CM_REQUIRE(H <= WIDE_HMAX, "%s: hidden_dim=%d > %d is not supported by this build", who, H, WIDE_HMAX);
    CM_REQUIRE(dout <= KWMAX, "%s: output width %d > %d is not supported by this build", who, dout, KWMAX);
    return 0;
}

// forward into the activation planes (act(l), l = 0..L; forward-only: two ping-pong planes); y (row stride ldy, ncols columns) receives the head
inline int wide_forward_layers(const MlpArgs& a, float* wsf, const WideWs& w, bool train, float* y, long ldy, int ncols, hipStream_t s,
                               const char* who) {
    const Offsets off = make_offsets(a.din, a.H, a.L, a.dout);
    const size_t plane = ((size_t)a.rows * w.Hs + 63) / 64 * 64;
    auto act = [&](int l) { return wsf + w.act + (size_t)(train ? l : (l & 1)) * plane; };
    if (fused128_shape(a.H, a.L) && a.dout <= KMAX) {  // one launch (cm_mlp_fused128.h); inputs wider than 64 columns: layer 0's product first
        F128X e = {};
        e.z0 = a.z0_add; e.ldz0 = w.Hs; e.y = y; e.ldy = ldy; e.ncols = ncols; e.vecx = x_rows_vec(a) ? 1 : 0;
        if (a.din > KC) {
            if (a.z0_add) wide_gemm<EPI_ADD>(a.x, a.x_stride, a.rows, a.din, a.params + off.W0, a.din, a.H, nullptr, nullptr, 0, a.z0_add, w.Hs, act(0), w.Hs, w.Hs, s);
            else wide_gemm<EPI_NONE>(a.x, a.x_stride, a.rows, a.din, a.params + off.W0, a.din, a.H, nullptr, nullptr, 0, nullptr, 0, act(0), w.Hs, w.Hs, s);
            e.z0 = act(0);
            fused128_launch<M_FWD, true>(a, e, s);
        } else fused128_launch<M_FWD, false>(a, e, s);
        CM_CHECK_LAUNCH(who);
        return 0;
    }
    wide_gemm<EPI_BIAS_RELU>(a.x, a.x_stride, a.rows, a.din, a.params + off.W0, a.din, a.H, a.params + off.b0, nullptr, 0, a.z0_add, w.Hs,
                             act(0), w.Hs, w.Hs, s);
    for (int l = 1; l <= a.L; ++l)
        wide_gemm<EPI_BIAS_RELU>(act(l - 1), w.Hs, a.rows, a.H, a.params + off.Wl(l - 1), a.H, a.H, a.params + off.bl(l - 1), nullptr, 0,
                                 nullptr, 0, act(l), w.Hs, w.Hs, s);
    wide_gemm<EPI_BIAS>(act(a.L), w.Hs, a.rows, a.H, a.params + off.Wout, a.H, a.dout, a.params + off.bout, a.avail, a.avail_stride, nullptr, 0, y,
                        ldy, ncols, s);
    CM_CHECK_LAUNCH(who);
    return 0;
}

// cm_mlp_forward for wide shapes: y[rows][dout]
inline int wide_forward(const MlpArgs& a, void* ws, size_t ws_bytes, hipStream_t s, const char* who) {
    if (int rc = wide_check(who, a.din, a.H, a.L, a.dout)) return rc;
    const WideWs w = wide_ws(a.rows, a.din, a.H, a.L, a.dout, false);
    CM_REQUIRE(ws && ws_bytes >= w.total * sizeof(float), "%s: workspace too small (%zu < %zu)", who, ws_bytes, w.total * sizeof(float));
    return wide_forward_layers(a, (float*)ws, w, false, a.y, a.dout, a.dout, s, who);
}

// Actor.act for wide shapes: layered forward to masked logits [rows][wide_kw(K)] in the workspace, then one thread per row draws with the
// SAME Philox keying and samplers as the fused M_ACT kernel (eps > 0: COMA's mixed sampling, eps < 0: greedy)
__global__ void k_wide_sample(const float* __restrict__ logits, long rows, int K, unsigned long long seed, long row_offset, int t,
                              float eps, int* __restrict__ action, float* __restrict__ logp, long out_stride, int ldl) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const unsigned long long gr = (unsigned long long)(row_offset + row);
    const cm_u4 rnd = cm_philox4x32((uint32_t)gr, (uint32_t)(gr >> 32), (uint32_t)t, CM_STREAM_ACT, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float* z = logits + row * ldl;
    int chosen; float lp;
    if (eps < 0.0f) cm_categorical_greedy(z, K, &chosen, &lp);
    else if (eps > 0.0f) cm_categorical_sample_eps(z, K, cm_u01(rnd.x), eps, &chosen, &lp);
    else cm_categorical_sample(z, K, cm_u01(rnd.x), &chosen, &lp);
    action[row * out_stride] = chosen;
    logp[row * out_stride] = lp;
}

inline int wide_act(const MlpArgs& a, void* ws, size_t ws_bytes, hipStream_t s, const char* who) {
    if (int rc = wide_check(who, a.din, a.H, a.L, a.dout)) return rc;
    const WideWs w = wide_ws(a.rows, a.din, a.H, a.L, a.dout, false, true);
    CM_REQUIRE(ws && ws_bytes >= w.total * sizeof(float), "%s: workspace too small (%zu < %zu)", who, ws_bytes, w.total * sizeof(float));
    float* out = (float*)ws + w.out;
    const int kw = wide_kw(a.dout);
    if (int rc = wide_forward_layers(a, (float*)ws, w, false, out, kw, kw, s, who)) return rc;
    hipLaunchKernelGGL(k_wide_sample, dim3((unsigned)((a.rows + 255) / 256)), dim3(256), 0, s, out, a.rows, a.dout, a.seed, a.row_offset, a.t,
                       a.act_eps, a.action_out, a.logp_out, a.out_stride, kw);
    CM_CHECK_LAUNCH(who);
    return 0;
}

// training pass of MODE over a wide MLP: grad_and_stats[P + 8]
// a.z0_add (optional): [rows][Hs] addend of the layer-0 pre-activation; dz0_out (optional) receives the pointer to dZ_0 [rows][Hs]
// inside the workspace (COMA's factored critic input needs both).
template <int MODE>
inline int wide_train(const MlpArgs& a, float* grad_and_stats, void* ws, size_t ws_bytes, hipStream_t s, const char* who,
                      float** dz0_out = nullptr) {
    if (int rc = wide_check(who, a.din, a.H, a.L, a.dout)) return rc;
    const WideWs w = wide_ws(a.rows, a.din, a.H, a.L, a.dout, true);
    CM_REQUIRE(ws && ws_bytes >= w.total * sizeof(float), "%s: workspace too small (%zu < %zu)", who, ws_bytes, w.total * sizeof(float));
    float* wsf = (float*)ws;
    const Offsets off = make_offsets(a.din, a.H, a.L, a.dout);
    const size_t plane = ((size_t)a.rows * w.Hs + 63) / 64 * 64;
    auto act = [&](int l) { return wsf + w.act + (size_t)l * plane; };
    float* out = wsf + w.out;
    float* part = wsf + w.part;
    float* wt = wsf + w.wt;
    const int H = a.H, Hs = w.Hs;
    if (fused128_shape(a.H, a.L) && a.dout <= KMAX) {  // forward + loss + backward in one launch (cm_mlp_fused128.h)
        MlpArgs b = a;
        const int64_t P = off.P;
        b.partial = part; b.PS = (int)((P + CM_NUM_STATS + 63) / 64 * 64);
#ifdef CM_PHASE_PROF
        b.prof = g_prof;
#endif
        const bool ext0 = a.din > KC;
        float* z0 = act(0);
        float* dz = wsf + w.dz;
        F128X e = {};
        e.z0 = a.z0_add; e.ldz0 = Hs; e.vecx = x_rows_vec(a) ? 1 : 0;
        e.dz0 = (ext0 || dz0_out) ? dz : nullptr; e.lddz0 = Hs;
        const int grid = fused128_grid(a.rows);
        if (ext0) {
            if (a.z0_add) wide_gemm<EPI_ADD>(a.x, a.x_stride, a.rows, a.din, a.params + off.W0, a.din, H, nullptr, nullptr, 0, a.z0_add, Hs, z0, Hs, Hs, s);
            else wide_gemm<EPI_NONE>(a.x, a.x_stride, a.rows, a.din, a.params + off.W0, a.din, H, nullptr, nullptr, 0, nullptr, 0, z0, Hs, Hs, s);
            e.z0 = z0;
            fused128_launch<MODE, true>(b, e, s);
        } else fused128_launch<MODE, false>(b, e, s);
        CM_CHECK_LAUNCH(who);
        const int n = (int)(P + CM_NUM_STATS), i0 = ext0 ? H * a.din : 0;
        hipLaunchKernelGGL(k_reduce_partials, dim3((n - i0 + RED_COLS - 1) / RED_COLS), dim3(RED_COLS * RED_GROUPS), 0, s, part, grid, b.PS, i0, n, grad_and_stats);
        CM_CHECK_LAUNCH(who);
        if (ext0)  // layer 0's weight gradient: dW0 = dZ0^T X, one 64-unit slab at a time (the partial rows above are folded by now)
            for (int n0 = 0; n0 < H; n0 += 64)
                if (int rc = stream_dw<true>(dz + n0, a.x, a.rows, a.din, min(64, H - n0), part, grad_and_stats + off.W0 + (size_t)n0 * a.din, s, who,
                                             Hs, a.x_stride)) return rc;
        if (dz0_out) *dz0_out = dz;
        return 0;
    }
    const int kw = wide_kw(a.dout);
    if (int rc = wide_forward_layers(a, wsf, w, true, out, kw, kw, s, who)) return rc;
    // ---- loss heads: logits -> dlogits in place, statistics
    {
        const int grid = (int)min((a.rows + NTHREADS - 1) / NTHREADS, (long)LOSS_GRID);
        if (kw == KMAX) hipLaunchKernelGGL((k_wide_loss<MODE, KMAX>), dim3(grid), dim3(NTHREADS), 0, s, a, out, part);
        else hipLaunchKernelGGL((k_wide_loss<MODE, KWMAX>), dim3(grid), dim3(NTHREADS), 0, s, a, out, part);
        hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(RED_COLS * RED_GROUPS), 0, s, part, grid, CM_NUM_STATS, 0, CM_NUM_STATS,
                           grad_and_stats + off.P);
        CM_CHECK_LAUNCH(who);
    }
    // ---- head: dWout = dOut^T act_L, dbout, dZ_L = (dOut Wout) .* relu'(act_L)
    if (int rc = stream_dw<true>(out, act(a.L), a.rows, H, a.dout, part, grad_and_stats + off.Wout, s, who, kw, Hs)) return rc;
    wide_colsum(out, kw, a.rows, a.dout, part, grad_and_stats + off.bout, s);
    float* dz = wsf + w.dz;
    float* dz2 = dz + plane;
    {
        const int ldt = (a.dout + 3) / 4 * 4;
        hipLaunchKernelGGL(k_wide_transpose, dim3((H * ldt + 255) / 256), dim3(256), 0, s, a.params + off.Wout, a.dout, H, wt, ldt);
        // Wt[h][k] = Wout[k][h]: N = H output columns, contraction over the ldt (zero padded) head outputs
        // the bias gradient of the layer that produced act(L) = column sums of dZ_L, accumulated in the GEMM's epilogue
        wide_gemm<EPI_GATE>(out, kw, a.rows, ldt, wt, ldt, H, nullptr, nullptr, 0, act(a.L), Hs, dz, Hs, Hs, s, part,
                            grad_and_stats + (a.L >= 1 ? off.bl(a.L - 1) : off.b0));
        CM_CHECK_LAUNCH(who);
    }
    // ---- hidden layers, top down: dz = dZ_{l+1} (pre-activation gradient of act(l+1))
    for (int l = a.L - 1; l >= 0; --l) {
        for (int n0 = 0; n0 < H; n0 += 64)
            if (int rc = stream_dw<true>(dz + n0, act(l), a.rows, H, min(64, H - n0), part, grad_and_stats + off.Wl(l) + (size_t)n0 * H, s, who,
                                         Hs, Hs)) return rc;
        const int ldt = (H + 3) / 4 * 4;
        hipLaunchKernelGGL(k_wide_transpose, dim3((H * ldt + 255) / 256), dim3(256), 0, s, a.params + off.Wl(l), H, H, wt, ldt);
        wide_gemm<EPI_GATE>(dz, Hs, a.rows, ldt, wt, ldt, H, nullptr, nullptr, 0, act(l), Hs, dz2, Hs, Hs, s, part,
                            grad_and_stats + (l >= 1 ? off.bl(l - 1) : off.b0));
        CM_CHECK_LAUNCH(who);
        float* tmp = dz; dz = dz2; dz2 = tmp;
    }
    // ---- layer 0: dW0 = dZ_0^T X, db0
    for (int n0 = 0; n0 < H; n0 += 64)
        if (int rc = stream_dw<true>(dz + n0, a.x, a.rows, a.din, min(64, H - n0), part, grad_and_stats + off.W0 + (size_t)n0 * a.din, s, who,
                                     Hs, a.x_stride)) return rc;
    CM_CHECK_LAUNCH(who);
    if (dz0_out) *dz0_out = dz;
    return 0;
}

}  // namespace
