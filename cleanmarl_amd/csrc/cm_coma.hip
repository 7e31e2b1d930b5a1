// cm_coma.hip -- COMA's target / critic / actor path (cleanmarl/coma_multienvs.py:553-684, coma.py)
//
// The MLP passes reuse the fused kernel template (k_mlp, modes M_QCRITIC / M_COMA_ACTOR) and the two schedules of
// cm_mlp_split.h; this file adds the HBM-bound glue kernels around them:
//   k_coma_build_inputs   Critic.coma_inputs (:222-240): [state | own obs | one-hot actions of the other agents]
//   k_gather_taken        torch.gather(q, -1, action) (:571-575, :596-600, :625-627)
//   k_nstep_returns       n-step targets (:581-613)
//   k_coma_adv / k_tstats counterfactual advantage q_a - sum_k pi_k q_k (:657-663) + per-time-step sums for the
//                         normalisation of :664-667 (raw double sums, so ranks can all-reduce them)
//   k_coma_normalize_adv  (adv - mean_t) / (std_t + 1e-8) where the reference's (sic) condition holds
//   k_polyak              soft_update (:266-270)
#include "cm_mlp_wide.h"

namespace {

__global__ __launch_bounds__(256) void k_coma_build_inputs(const float* __restrict__ state, const float* __restrict__ obs,
                                                           const int* __restrict__ action, int E, int A, int T, int Ds, int Do,
                                                           int K, float* __restrict__ out) {
    const int Dc = Ds + Do + (A - 1) * K;
    const long total = (long)E * A * T * Dc;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long row = i / Dc;
        const int col = (int)(i - row * Dc);
        const int t = (int)(row % T);
        const long ea = row / T;
        const int a = (int)(ea % A);
        const long e = ea / A;
        float v;
        if (col < Ds) v = state[(e * T + t) * Ds + col];
        else if (col < Ds + Do) v = obs[row * Do + (col - Ds)];
        else {
            const int c = col - Ds - Do, slot = c / K, k = c - slot * K;
            const int j = slot < a ? slot : slot + 1;  // the OTHER agents, in agent order
            v = (action[(e * A + j) * T + t] == k) ? 1.0f : 0.0f;
        }
        out[i] = v;
    }
}

__global__ __launch_bounds__(256) void k_gather_taken(const float* __restrict__ q, const int* __restrict__ action, long rows, int K,
                                                      float* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows; i += (long)gridDim.x * 256) out[i] = q[i * K + action[i]];
}

__global__ __launch_bounds__(256) void k_nstep_returns(const float* __restrict__ reward, const float* __restrict__ qtaken,
                                                       const int* __restrict__ ep_len, int E, int A, int T, double gamma, int n,
                                                       float* __restrict__ ret) {
    const long total = (long)E * A * T;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int t = (int)(i % T);
        const long e = i / ((long)A * T);
        const int L = min(max(ep_len[e], 0), T);
        float r = 0.0f;
        if (t < L) {
            const bool boot = t < L - n;
            const int m = boot ? n : L - t;
            const float* rp = reward + e * T + t;
            double d = 1.0;
            for (int k = 0; k < m; ++k) { r += rp[k] * (float)d; d *= gamma; }  // fp32 discounts of double powers, fp32 sum
            if (boot) r += (float)d * qtaken[i + n];
        }
        ret[i] = r;
    }
}

// one thread per row: pi = softmax(masked logits), adv = q[a] - sum_k pi_k q_k
__global__ __launch_bounds__(256) void k_coma_adv(const float* __restrict__ logits, const float* __restrict__ q,
                                                  const int* __restrict__ action, long rows, int K, float* __restrict__ adv) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows; i += (long)gridDim.x * 256) {
        const float* z = logits + i * K;
        const float* qq = q + i * K;
        float m = -INFINITY;
        for (int k = 0; k < K; ++k) m = fmaxf(m, z[k]);
        float s = 0.0f, b = 0.0f;
        for (int k = 0; k < K; ++k) { const float e = expf(z[k] - m); s += e; b += e * qq[k]; }
        adv[i] = qq[action[i]] - b / s;
    }
}

// per-time-step raw sums over this rank's rows: tstats[t] = {n valid, sum adv, sum adv^2, sum of action indices over ALL rows}
constexpr int TS_CHUNKS = 64;
__global__ __launch_bounds__(256) void k_tstats_partial(const float* __restrict__ adv, const int* __restrict__ action,
                                                        const int* __restrict__ ep_len, int E, int A, int T, double* __restrict__ part) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + lane;
    const long nseq = (long)E * A;
    const long per = (nseq + gridDim.y - 1) / gridDim.y;
    const long s0 = (long)blockIdx.y * per, s1 = min(nseq, s0 + per);
    double n = 0, s = 0, s2 = 0, as = 0;
    if (t < T) {
        for (long seq = s0 + w; seq < s1; seq += 4) {
            const int e = (int)(seq / A);
            as += (double)action[seq * T + t];
            if (t < ep_len[e]) { const double v = adv[seq * T + t]; n += 1.0; s += v; s2 += v * v; }
        }
    }
    __shared__ double sh[4][64][4];
    sh[w][lane][0] = n; sh[w][lane][1] = s; sh[w][lane][2] = s2; sh[w][lane][3] = as;
    __syncthreads();
    if (w == 0 && t < T) {
        double* o = part + ((size_t)blockIdx.y * T + t) * 4;
        for (int c = 0; c < 4; ++c) o[c] = sh[0][lane][c] + sh[1][lane][c] + sh[2][lane][c] + sh[3][lane][c];
    }
}
__global__ __launch_bounds__(256) void k_tstats_final(const double* __restrict__ part, int nchunks, int T, double* __restrict__ tstats) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= T * 4) return;
    double s = 0;
    for (int c = 0; c < nchunks; ++c) s += part[(size_t)c * T * 4 + i];
    tstats[i] = s;
}

__global__ __launch_bounds__(256) void k_coma_normalize_adv(float* __restrict__ adv, const double* __restrict__ tstats, int E, int A,
                                                            int T) {
    const long total = (long)E * A * T;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int t = (int)(i % T);
        const double n = tstats[4 * t], s = tstats[4 * t + 1], s2 = tstats[4 * t + 2], as = tstats[4 * t + 3];
        if (as > (double)A && n >= 2.0) {  // coma_multienvs.py:664 (sic): sum of the step's action indices > n_agents
            const double mean = s / n;
            const double var = fmax(0.0, (s2 - n * mean * mean) / (n - 1.0));  // unbiased, like torch.std
            adv[i] = (adv[i] - (float)mean) / ((float)sqrt(var) + 1e-8f);
        }
    }
}

__global__ __launch_bounds__(256) void k_polyak(float* __restrict__ target, const float* __restrict__ src, long n, float tau, float keep) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) target[i] = tau * src[i] + keep * target[i];
}

// ------------------------------------------------------------------------------------------------------------------
// Factored critic input.  The reference materialises x = [state | obs | one-hot(other agents' actions)] per (e, a, t)
// (coma_multienvs.py:222-240; 475 floats per row at config-3 shapes, 8 GB per batch) and multiplies it by W0.  The state
// block is the same for the A agents of a step and the action block is one-hot, so
//     W0 x + b0 = W0o obs[e,a,t] + ( W0s state[e,t] + sum_{j != a} W0a[:, slot(j) K + u_j] ) + b0
// : the fused MLP kernel runs on the OBS block only (din = Do) with a per-row addend z0_add[rows][64] =
// S[e,t] + gathered columns, where S = state W0s^T is one [E T x Ds] x [Ds x 64] GEMM (A times fewer rows).
// Backward: dW0o falls out of the fused kernel; one pass over dZ0[rows][64] yields dS = sum_a dZ0 (then dW0s = dS^T state,
// a streaming MFMA GEMM, k_dw0_stream) and the action block dW0a by prefix-sum gather-adds (k_coma_bwd_gather).  ~7x fewer layer-0 FLOPs and no 8 GB tensor.
// ------------------------------------------------------------------------------------------------------------------

// pc = [W0o (H x Do) | everything after W0 (b0, hidden layers, head)] from the torch-order buffer [W0 (H x Dc) | ...]
__global__ __launch_bounds__(256) void k_coma_compact_params(const float* __restrict__ full, int H, int Dc, int Ds, int Do, int rest,
                                                             float* __restrict__ pc) {
    const int n = H * Do + rest;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        if (i < H * Do) { const int h = i / Do, c = i - h * Do; pc[i] = full[h * Dc + Ds + c]; }
        else pc[i] = full[H * Dc + (i - H * Do)];
    }
}

// ws[h][c] = W0[h][c] for the state block (c < Ds), row stride ldp (a multiple of 4 floats): the torch-order rows (stride Dc, 475 floats at
// config-3 shapes) are not 16-byte aligned, and k_wide_gemm then loads its weight tile with four predicated scalar loads per float4
__global__ __launch_bounds__(256) void k_coma_pack_w0s(const float* __restrict__ full, int H, int Dc, int Ds, int ldp, float* __restrict__ ws) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H * ldp; i += gridDim.x * 256) {
        const int h = i / ldp, c = i - h * ldp;
        ws[i] = c < Ds ? full[(long)h * Dc + c] : 0.0f;
    }
}

// out[rows][HP] = X[rows][ncols] * W[H][ncols]^T (W row stride w_stride); columns >= H are zero
__global__ __launch_bounds__(NTHREADS, 2) void k_linear_nt(const float* __restrict__ x, long rows, long x_stride, int ncols,
                                                           const float* __restrict__ W, long w_stride, int H, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float Xs[TM * LDT];
    __shared__ __attribute__((aligned(16))) float Wsm[HP * LDT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1, h = lane >> 5, lc = lane & 31;
    const long ntiles = (rows + TM - 1) / TM;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = tile * TM;
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
        for (int c0 = 0; c0 < ncols; c0 += KC) {
            __syncthreads();
            stage_rows(Xs, x, row0, rows, x_stride, c0, min(KC, ncols - c0));
            stage_rows(Wsm, W, 0, H, w_stride, c0, min(KC, ncols - c0));
            __syncthreads();
            rowpar_nt(acc, Xs + 32 * wm * LDT, Wsm + 32 * wn * LDT, KC / 8);
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const long row = row0 + 32 * wm + (g & 3) + 8 * (g >> 2) + 4 * h;
            if (row < rows) out[row * HP + 32 * wn + lc] = acc[g];
        }
    }
}

// z0_add[(e,a,t)][h] = S[e,t][h] + sum_{j != a} W0a[h][slot(j,a) K + u_j].  One wave per (e,t), lane = hidden unit: the A
// actions of the step are fetched once (lane j holds u_j, broadcast by readlane), S once, then the A rows are produced from
// the transposed action block of W0 in LDS ([(A-1)K][64], conflict-free across lanes).
// VW = hidden units per lane (1: one 64-unit slab per launch; 2: 128 units in ONE pass, float2 per lane -- the fused 128-wide critic)
template <int VW>
__global__ __launch_bounds__(256) void k_coma_z0_add(const float* __restrict__ S, const int* __restrict__ action,
                                                     const float* __restrict__ W0, int E, int A, int T, int H, int Dc, int Ds, int Do,
                                                     int K, float* __restrict__ out, int h0, long ldS, long ldo) {
    // h0 / ldS / ldo: this launch covers hidden units h0..h0+64 VW-1 of S and out rows with strides ldS / ldo (fused path: 0, HP, HP;
    // the layered schedule of wide critics runs one launch per 64-unit slab)
    constexpr int HW = 64 * VW;
    extern __shared__ __attribute__((aligned(16))) float tab[];
    const int Da = (A - 1) * K;
    for (int i = threadIdx.x; i < Da * HW; i += 256) {
        const int c = i / HW, hh = i - c * HW;
        tab[i] = h0 + hh < H ? W0[(long)(h0 + hh) * Dc + Ds + Do + c] : 0.0f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long net = (long)E * T;
    // a wave takes runs of ZR consecutive (e,t): the ZR action / S requests of a run are in flight together, and its stores to one agent's
    // rows are ZR adjacent segments (the four waves of a block: 4 ZR adjacent segments per agent -- longer bursts per DRAM page)
    constexpr int ZR = 4;
    const long nrun = (net + ZR - 1) / ZR;
    for (long run = (long)blockIdx.x * 4 + w; run < nrun; run += (long)gridDim.x * 4) {
        int u[ZR];
        float sv[ZR][VW];
#pragma unroll
        for (int i = 0; i < ZR; ++i) {
            const long et = min(run * ZR + i, net - 1);
            const long e = et / T;
            const int t = (int)(et - e * T);
            u[i] = action[(e * A + (lane < A ? lane : 0)) * T + t];
            if constexpr (VW == 2) { const float2 q = *reinterpret_cast<const float2*>(S + et * ldS + h0 + 2 * lane); sv[i][0] = q.x; sv[i][1] = q.y; }
            else sv[i][0] = S[et * ldS + h0 + lane];
        }
#pragma unroll
        for (int i = 0; i < ZR; ++i) {
            const long et = run * ZR + i;
            if (et >= net) break;
            const long e = et / T;
            const int t = (int)(et - e * T);
            for (int a = 0; a < A; ++a) {
                float v[VW];
#pragma unroll
                for (int q = 0; q < VW; ++q) v[q] = sv[i][q];
                for (int slot = 0; slot < A - 1; ++slot) {
                    const int j = slot < a ? slot : slot + 1;
                    const float* tp = tab + (slot * K + __shfl(u[i], j, 64)) * HW + VW * lane;
#pragma unroll
                    for (int q = 0; q < VW; ++q) v[q] += tp[q];
                }
                float* op = out + ((e * A + a) * T + t) * ldo + h0 + VW * lane;
                if constexpr (VW == 2) *reinterpret_cast<float2*>(op) = make_float2(v[0], v[1]);
                else op[0] = v[0];
            }
        }
    }
}

// Backward of the factored input, one wave per (e,t), lane = hidden unit:
//   dS[e,t][h]  = sum_a dZ0[e,a,t][h]                                   (feeds dW0s = dS^T state)
//   dW0a[h][c] += dZ0[e,a,t][h] for every (a, j != a) with c = slot(j,a) K + u_j.  Agent j's action lands in column block
//   j-1 for the agents a < j and in block j for the agents a > j, so with prefix sums over a this is 2 adds per j:
//       tab[(j-1) K + u_j] += sum_{a<j} dZ0[e,a,t]     tab[j K + u_j] += sum_{a>j} dZ0[e,a,t]
//   accumulated in a PRIVATE per-wave LDS table (fixed (e,t) -> wave assignment, sequential adds: deterministic), written as
//   per-wave partials [nwaves][Da][HP] and reduced in a fixed order by k_reduce_partials.
constexpr int OH_MAXA = 32;
constexpr int OH_NWAVES = 8192;  // upper bound of gather waves (= partial tables)
constexpr int OH_LDS = 80 * 1024;  // LDS of one gather block: two blocks per CU
// Round 4: the pass is a stream over dZ0 (2.1 GB at config-3 shapes and 128 units) and ran at 1.25 TB/s -- every (e,t) paid a full HBM
// round trip (one request batch in flight per wave, six waves per CU) and a chain of dependent LDS read-modify-writes.  Now THREE (e,t)
// records rotate through registers (two in flight under the one being folded; AMAX = 8 keeps a record at 8 VW registers), the table
// reads of one record are all issued before its adds and writes (see fold; ds_add_f32 was tried and is 2.7x SLOWER: LDS float atomics
// retire a lane at a time), the table is [column][VW][64] (each plane contiguous across the lanes: no bank conflicts), and four waves per
// block / two blocks per CU fit 80 KB.
template <int VW, int AMAX>
__global__ __launch_bounds__(256) void k_coma_bwd_gather(const float* __restrict__ dz0, const int* __restrict__ action, int E, int A, int T,
                                                         int K, int waves_per_block, float* __restrict__ dS, float* __restrict__ part,
                                                         int h0, long ldz, long ldS) {
    constexpr int HW = 64 * VW;  // hidden units per pass (VW per lane, see k_coma_z0_add)
    extern __shared__ __attribute__((aligned(16))) float tab[];
    const int Da = (A - 1) * K;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (w >= waves_per_block) return;  // no workgroup barrier below: the tables are wave-private
    float* mine = tab + (size_t)w * Da * HW;
    for (int i = lane; i < Da * HW; i += 64) mine[i] = 0.0f;
    const long net = (long)E * T;
    const long nw = (long)gridDim.x * waves_per_block;
    struct Rec { float z[AMAX][VW]; int u; };
    auto fetch = [&](Rec& r, long et) {
        if (et >= net) return;
        const long e = et / T;
        const int t = (int)(et - e * T);
        r.u = action[(e * A + (lane < A ? lane : 0)) * T + t];
#pragma unroll
        for (int a = 0; a < AMAX; ++a) {
            if (a < A) {
                const float* zp = dz0 + ((e * A + a) * T + t) * ldz + h0 + VW * lane;
                if constexpr (VW == 2) { const float2 q2 = *reinterpret_cast<const float2*>(zp); r.z[a][0] = q2.x; r.z[a][1] = q2.y; }
                else r.z[a][0] = zp[0];
            }
        }
    };
    auto fold = [&](const Rec& r, long et) {
        float tot[VW];
#pragma unroll
        for (int q = 0; q < VW; ++q) tot[q] = 0.0f;
#pragma unroll
        for (int a = 0; a < AMAX; ++a)
            if (a < A) {
#pragma unroll
                for (int q = 0; q < VW; ++q) tot[q] += r.z[a][q];
            }
        float* sp = dS + et * ldS + h0 + VW * lane;
        if constexpr (VW == 2) *reinterpret_cast<float2*>(sp) = make_float2(tot[0], tot[1]);
        else sp[0] = tot[0];
        // Column block s (s = 0 .. A-2) receives two sums per (e,t): sum_{a>s} dZ0 at column s K + u_s (agent s seen by the agents behind it)
        // and sum_{a<=s} dZ0 at column s K + u_{s+1} (agent s+1 seen by the agents before it).  Different blocks never alias, the two
        // updates of one block alias exactly when u_s == u_{s+1} (wave-uniform) -- so ALL table reads are issued first, then the adds (in
        // the order "behind" then "before", the order of the round-3 kernel), then the writes: no chain of dependent LDS round trips.
        float rb[AMAX - 1][VW], rf[AMAX - 1][VW];
#pragma unroll
        for (int sl = 0; sl < AMAX - 1; ++sl) {
            if (sl < A - 1) {
                const int us = __builtin_amdgcn_readlane(r.u, sl), un = __builtin_amdgcn_readlane(r.u, sl + 1);
#pragma unroll
                for (int q = 0; q < VW; ++q) {
                    rb[sl][q] = mine[((sl * K + us) * VW + q) * 64 + lane];
                    rf[sl][q] = mine[((sl * K + un) * VW + q) * 64 + lane];
                }
            }
        }
        float pre[VW];  // sum_{a<=sl}
#pragma unroll
        for (int q = 0; q < VW; ++q) pre[q] = 0.0f;
#pragma unroll
        for (int sl = 0; sl < AMAX - 1; ++sl) {
            if (sl < A - 1) {
                const int us = __builtin_amdgcn_readlane(r.u, sl), un = __builtin_amdgcn_readlane(r.u, sl + 1);
#pragma unroll
                for (int q = 0; q < VW; ++q) {
                    const float suf = tot[q] - pre[q] - r.z[sl][q];  // sum_{a>sl}
                    pre[q] += r.z[sl][q];
                    float* pb = mine + ((sl * K + us) * VW + q) * 64 + lane;
                    if (us == un) pb[0] = (rb[sl][q] + suf) + pre[q];
                    else { pb[0] = rb[sl][q] + suf; mine[((sl * K + un) * VW + q) * 64 + lane] = rf[sl][q] + pre[q]; }
                }
            }
        }
    };
    Rec r0, r1, r2;
    long et = (long)blockIdx.x * waves_per_block + w;
    fetch(r0, et);
    fetch(r1, et + nw);
    while (true) {
        if (et >= net) break;
        fetch(r2, et + 2 * nw); fold(r0, et); et += nw;
        if (et >= net) break;
        fetch(r0, et + 2 * nw); fold(r1, et); et += nw;
        if (et >= net) break;
        fetch(r1, et + 2 * nw); fold(r2, et); et += nw;
    }
    float* o = part + ((size_t)blockIdx.x * waves_per_block + w) * Da * HW;  // [column][HW], unit VW lane + q
    for (int c = 0; c < Da; ++c)
#pragma unroll
        for (int q = 0; q < VW; ++q) o[c * HW + VW * lane + q] = mine[(c * VW + q) * 64 + lane];
}

// launch geometry of the gather pass: waves per block within OH_LDS, blocks capped at the resident set (more would only add partial tables)
struct GatherGeom { int wpb; long blocks; size_t lds; };
inline GatherGeom gather_geom(long et, size_t per_wave, int vw) {
    GatherGeom g;
    g.wpb = (int)(OH_LDS / per_wave);
    g.wpb = g.wpb > 4 ? 4 : (g.wpb < 1 ? 1 : g.wpb);
    g.lds = per_wave * g.wpb;
    long per_cu = (160 * 1024) / (long)g.lds;
    per_cu = per_cu > 4 ? 4 : (per_cu < 1 ? 1 : per_cu);
    g.blocks = (et + g.wpb - 1) / g.wpb;
    const long resident = 256 * per_cu, cap = (OH_NWAVES / vw) / g.wpb;
    if (g.blocks > resident) g.blocks = resident;
    if (g.blocks > cap) g.blocks = cap;
    return g;
}
template <int VW>
inline void gather_launch(const GatherGeom& g, int A, hipStream_t s, const float* dz0, const int* action, int E, int T, int K, float* dS, float* part,
                          int h0, long ldz, long ldS) {
    if (A <= 8) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_coma_bwd_gather<VW, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, OH_LDS);
        hipLaunchKernelGGL((k_coma_bwd_gather<VW, 8>), dim3((int)g.blocks), dim3(256), g.lds, s, dz0, action, E, A, T, K, g.wpb, dS, part, h0, ldz, ldS);
    } else if (A <= 16) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_coma_bwd_gather<VW, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, OH_LDS);
        hipLaunchKernelGGL((k_coma_bwd_gather<VW, 16>), dim3((int)g.blocks), dim3(256), g.lds, s, dz0, action, E, A, T, K, g.wpb, dS, part, h0, ldz, ldS);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_coma_bwd_gather<VW, OH_MAXA>), hipFuncAttributeMaxDynamicSharedMemorySize, OH_LDS);
        hipLaunchKernelGGL((k_coma_bwd_gather<VW, OH_MAXA>), dim3((int)g.blocks), dim3(256), g.lds, s, dz0, action, E, A, T, K, g.wpb, dS, part, h0, ldz, ldS);
    }
}

// ga[h][c] = reduced[c][h] (H x Da, torch order) from the [Da][HP] layout of the gather kernel
__global__ __launch_bounds__(256) void k_transpose_ga(const float* __restrict__ red, int H, int Da, float* __restrict__ ga, int ldr) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H * Da; i += gridDim.x * 256) {
        const int hh = i / Da, c = i - hh * Da;
        ga[i] = red[c * ldr + hh];
    }
}

// full-layout gradient + stats from the pieces: W0 = [gs (H x Ds) | gc's W0o (H x Do) | ga (H x Da)], rest (+ 8 stats) from gc
__global__ __launch_bounds__(256) void k_coma_scatter_grads(const float* __restrict__ gc, const float* __restrict__ gs,
                                                            const float* __restrict__ ga, int H, int Dc, int Ds, int Do, int rest,
                                                            float* __restrict__ out) {
    const int Da = Dc - Ds - Do;
    const int n = H * Dc + rest;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float v;
        if (i < H * Dc) {
            const int hh = i / Dc, c = i - hh * Dc;
            v = c < Ds ? gs[hh * Ds + c] : (c < Ds + Do ? gc[hh * Do + (c - Ds)] : ga[hh * Da + (c - Ds - Do)]);
        } else v = gc[H * Do + (i - H * Dc)];
        out[i] = v;
    }
}

struct ComaWs {  // float offsets into the caller's workspace
    size_t pc, S, z0, dz0, dS, gc, gs, ga, part, train, total;
};
inline size_t al64(size_t x) { return (x + 63) & ~(size_t)63; }
inline ComaWs coma_ws(int E, int A, int T, int Ds, int Do, int K, int H, int L, bool train) {
    const long rows = (long)E * A * T, et = (long)E * T;
    const int Da = (A - 1) * K;
    const size_t Pc = (size_t)cm_mlp_param_count(Do, H, L, K);
    ComaWs w; size_t p = 0;
    w.pc = p; p += al64(Pc);
    // S = state W0s^T of the separate-launch path; the one-launch path (EPI_COMA) parks its aligned copy of W0's state block here instead
    { const size_t s1 = (size_t)et * HP, s2 = (size_t)H * ((Ds + 3) & ~3); w.S = p; p += al64(s1 > s2 ? s1 : s2); }
    w.z0 = p; p += al64((size_t)rows * HP);
    w.dz0 = w.dS = w.gc = w.gs = w.ga = w.part = w.train = p;
    if (train) {
        w.dz0 = p; p += al64((size_t)rows * HP);
        w.dS = p; p += al64((size_t)et * HP);
        w.gc = p; p += al64(Pc + CM_NUM_STATS);
        w.gs = p; p += al64((size_t)H * Ds);
        w.ga = p; p += al64((size_t)H * (Da > 0 ? Da : 1));
        w.part = p; p += al64(stream_dw_ws_floats(H, Ds) > (size_t)(OH_NWAVES + 1) * (Da > 0 ? Da : 1) * HP ? stream_dw_ws_floats(H, Ds)
                                                                                              : (size_t)(OH_NWAVES + 1) * (Da > 0 ? Da : 1) * HP);
        w.train = p; p += al64(split_ws_bytes(rows, Do, H, L, K) / sizeof(float) + 1);
    }
    w.total = p;
    return w;
}

// steps a-c shared by the forward and the training entry points: compact params, S, z0_add
inline int coma_prepare(const float* state, const float* obs, const int32_t* action, int E, int A, int T, int Ds, int Do, int K, int H,
                        int L, const float* params, float* wsf, const ComaWs& w, hipStream_t s, const char* who, long state_ld = 0) {
    (void)obs;
    if (state_ld <= 0) state_ld = Ds;  // row stride of `state` (the "_ld" entry points; padding columns hold zeros)
    const int Dc = Ds + Do + (A - 1) * K;
    const int rest = (int)(cm_mlp_param_count(Dc, H, L, K) - (int64_t)H * Dc);
    hipLaunchKernelGGL(k_coma_compact_params, dim3(64), dim3(256), 0, s, params, H, Dc, Ds, Do, rest, wsf + w.pc);
    CM_CHECK_LAUNCH(who);
    const long et = (long)E * T;
    if (coma_epi_fits(A, K, HP) && cm_option(CM_OPTION_WIDE_SCHEDULE) != 3) {  // one launch: S in the GEMM's tile, z0's A rows from its epilogue (cm_mlp_wide.h, EPI_COMA)
        const WideComa cx = {action, params, A, T, K, Dc, Ds + Do, H};
        const int ldp = (Ds + 3) & ~3;  // aligned copy of the state block of W0 in the (unused) S region
        hipLaunchKernelGGL(k_coma_pack_w0s, dim3(64), dim3(256), 0, s, params, H, Dc, Ds, ldp, wsf + w.S);
        wide_gemm_coma(state, state_ld, et, Ds, wsf + w.S, ldp, H, wsf + w.z0, HP, HP, s, cx);
        CM_CHECK_LAUNCH(who);
        return 0;
    }
    const long nt = (et + TM - 1) / TM;
    hipLaunchKernelGGL(k_linear_nt, dim3((int)(nt < 512 ? nt : 512)), dim3(NTHREADS), 0, s, state, et, state_ld, Ds, params, (long)Dc, H, wsf + w.S);
    CM_CHECK_LAUNCH(who);
    const size_t tab_bytes = (size_t)(A - 1) * K * HP * sizeof(float);
    CM_REQUIRE(tab_bytes <= 64 * 1024, "%s: (n_agents - 1) * n_actions = %d exceeds the 256-column LDS table", who, (A - 1) * K);
    CM_REQUIRE(A <= 64, "%s: n_agents=%d > 64 is not supported by the factored critic input", who, A);
    const long g = (et + 15) / 16;  // runs of four (e,t) per wave, four waves per block
    hipLaunchKernelGGL(k_coma_z0_add<1>, dim3((int)(g < 4096 ? g : 4096)), dim3(256), tab_bytes > 0 ? tab_bytes : 4, s, wsf + w.S, action, params, E, A, T, H,
                       Dc, Ds, Do, K, wsf + w.z0, 0, (long)HP, (long)HP);
    CM_CHECK_LAUNCH(who);
    return 0;
}

// ---- the same factoring on the layered schedule (cm_mlp_wide.h) for critics wider than 64 units / deeper than 2 hidden layers
// (the reference default is 128): S, z0_add, dZ0 and dS have row stride Hs and the two lane-per-unit kernels run once per 64-unit slab.
struct ComaWideWs { size_t pc, S, z0, dS, gc, gs, ga, part, train, total; int Hs; };
inline ComaWideWs coma_wide_ws(int E, int A, int T, int Ds, int Do, int K, int H, int L, bool train) {
    const long rows = (long)E * A * T, et = (long)E * T;
    const int Da = (A - 1) * K, DaP = Da > 0 ? Da : 1;
    const size_t Pc = (size_t)cm_mlp_param_count(Do, H, L, K);
    ComaWideWs w; size_t p = 0;
    w.Hs = wide_hs(H);
    w.pc = p; p += al64(Pc);
    { const size_t s1 = (size_t)et * w.Hs, s2 = (size_t)H * ((Ds + 3) & ~3); w.S = p; p += al64(s1 > s2 ? s1 : s2); }  // see coma_ws
    w.z0 = p; p += al64((size_t)rows * w.Hs);
    w.dS = w.gc = w.gs = w.ga = w.part = p;
    if (train) {
        w.dS = p; p += al64((size_t)et * w.Hs);
        w.gc = p; p += al64(Pc + CM_NUM_STATS);
        w.gs = p; p += al64((size_t)H * Ds);
        w.ga = p; p += al64((size_t)H * DaP);
        const size_t a1 = (size_t)DW0_GRID * 64 * Ds, a2 = (size_t)(OH_NWAVES + 2) * DaP * HP;
        w.part = p; p += al64(a1 > a2 ? a1 : a2);
    }
    w.train = p; p += al64(wide_ws_bytes(rows, Do, H, L, K, train) / sizeof(float) + 1);
    w.total = p;
    return w;
}

inline int coma_wide_prepare(const float* state, const int32_t* action, int E, int A, int T, int Ds, int Do, int K, int H, int L,
                             const float* params, float* wsf, const ComaWideWs& w, hipStream_t s, const char* who, long state_ld = 0) {
    if (state_ld <= 0) state_ld = Ds;
    const int Dc = Ds + Do + (A - 1) * K;
    const int rest = (int)(cm_mlp_param_count(Dc, H, L, K) - (int64_t)H * Dc);
    hipLaunchKernelGGL(k_coma_compact_params, dim3(64), dim3(256), 0, s, params, H, Dc, Ds, Do, rest, wsf + w.pc);
    CM_CHECK_LAUNCH(who);
    const long et = (long)E * T;
    if (coma_epi_fits(A, K, w.Hs) && cm_option(CM_OPTION_WIDE_SCHEDULE) != 3) {
        // one launch: S = state W0s^T stays in the GEMM's tile, its epilogue writes the A rows of z0 per (e,t) (cm_mlp_wide.h, EPI_COMA)
        const WideComa cx = {action, params, A, T, K, Dc, Ds + Do, H};
        const int ldp = (Ds + 3) & ~3;  // aligned copy of the state block of W0 in the (unused) S region
        hipLaunchKernelGGL(k_coma_pack_w0s, dim3(64), dim3(256), 0, s, params, H, Dc, Ds, ldp, wsf + w.S);
        wide_gemm_coma(state, state_ld, et, Ds, wsf + w.S, ldp, H, wsf + w.z0, w.Hs, w.Hs, s, cx);
        CM_CHECK_LAUNCH(who);
        return 0;
    }
    wide_gemm<EPI_NONE>(state, state_ld, et, Ds, params, Dc, H, nullptr, nullptr, 0, nullptr, 0, wsf + w.S, w.Hs, w.Hs, s);  // S = state W0s^T
    CM_CHECK_LAUNCH(who);
    const size_t tab_bytes = (size_t)(A - 1) * K * HP * sizeof(float);
    CM_REQUIRE(tab_bytes <= 64 * 1024, "%s: (n_agents - 1) * n_actions = %d exceeds the 256-column LDS table", who, (A - 1) * K);
    CM_REQUIRE(A <= OH_MAXA, "%s: n_agents=%d > %d is not supported by the factored critic input", who, A, OH_MAXA);
    const long g = (et + 15) / 16;  // runs of four (e,t) per wave, four waves per block
    if (w.Hs == 2 * HP && 2 * tab_bytes <= 64 * 1024) {  // 65..128 units: both slabs in one pass, float2 per lane
        hipLaunchKernelGGL(k_coma_z0_add<2>, dim3((int)(g < 4096 ? g : 4096)), dim3(256), tab_bytes > 0 ? 2 * tab_bytes : 8, s, wsf + w.S, action, params, E,
                           A, T, H, Dc, Ds, Do, K, wsf + w.z0, 0, (long)w.Hs, (long)w.Hs);
        CM_CHECK_LAUNCH(who);
        return 0;
    }
    for (int h0 = 0; h0 < H; h0 += 64) {
        hipLaunchKernelGGL(k_coma_z0_add<1>, dim3((int)(g < 4096 ? g : 4096)), dim3(256), tab_bytes > 0 ? tab_bytes : 4, s, wsf + w.S, action, params, E,
                           A, T, H, Dc, Ds, Do, K, wsf + w.z0, h0, (long)w.Hs, (long)w.Hs);
        CM_CHECK_LAUNCH(who);
    }
    return 0;
}

inline int coma_wide_q_forward(const float* state, const float* obs, const int32_t* action, const uint8_t* avail, int E, int A, int T,
                               int Ds, int Do, int K, int H, int L, const float* params, float* q, void* ws, size_t ws_bytes,
                               hipStream_t s, const char* who, long obs_ld = 0, long state_ld = 0) {
    if (int rc = wide_check(who, Do, H, L, K)) return rc;
    if (obs_ld <= 0) obs_ld = Do;
    const ComaWideWs w = coma_wide_ws(E, A, T, Ds, Do, K, H, L, false);
    CM_REQUIRE(ws && ws_bytes >= w.total * sizeof(float), "%s: workspace too small (%zu < %zu)", who, ws_bytes, w.total * sizeof(float));
    float* wsf = (float*)ws;
    if (int rc = coma_wide_prepare(state, action, E, A, T, Ds, Do, K, H, L, params, wsf, w, s, who, state_ld)) return rc;
    MlpArgs a = {};
    a.x = obs; a.x_stride = obs_ld; a.rows = (long)E * A * T; a.din = Do; a.H = H; a.L = L; a.dout = K;
    a.params = wsf + w.pc; a.avail = avail; a.avail_stride = K; a.y = q; a.z0_add = wsf + w.z0;
    return wide_forward(a, wsf + w.train, (w.total - w.train) * sizeof(float), s, who);
}

inline int coma_wide_critic_fwd_bwd(const float* state, const float* obs, const int32_t* action, const float* target,
                                    const int32_t* ep_len, int E, int A, int T, int Ds, int Do, int K, int H, int L, const float* params,
                                    float* grad_and_stats, void* ws, size_t ws_bytes, hipStream_t s, const char* who, long obs_ld = 0,
                                    long state_ld = 0) {
    if (int rc = wide_check(who, Do, H, L, K)) return rc;
    if (obs_ld <= 0) obs_ld = Do;
    if (state_ld <= 0) state_ld = Ds;
    const long rows = (long)E * A * T, et = (long)E * T;
    if (int rc = check_rows(who, rows)) return rc;
    const ComaWideWs w = coma_wide_ws(E, A, T, Ds, Do, K, H, L, true);
    CM_REQUIRE(ws && ws_bytes >= w.total * sizeof(float), "%s: workspace too small (%zu < %zu)", who, ws_bytes, w.total * sizeof(float));
    float* wsf = (float*)ws;
    const int Da = (A - 1) * K, Dc = Ds + Do + Da, DaP = Da > 0 ? Da : 1;
    if (int rc = coma_wide_prepare(state, action, E, A, T, Ds, Do, K, H, L, params, wsf, w, s, who, state_ld)) return rc;
    MlpArgs a = {};
    a.x = obs; a.x_stride = obs_ld; a.rows = rows; a.din = Do; a.H = H; a.L = L; a.dout = K;
    a.params = wsf + w.pc; a.action = action; a.ret = target; a.ep_len = ep_len; a.A = A; a.T = T; a.per_agent = 1;
    a.z0_add = wsf + w.z0;
    float* dz0 = nullptr;
    if (int rc = wide_train<M_QCRITIC>(a, wsf + w.gc, wsf + w.train, (w.total - w.train) * sizeof(float), s, who, &dz0)) return rc;
    const size_t per_wave = (size_t)DaP * HP * sizeof(float);
    CM_REQUIRE(per_wave <= 64 * 1024, "%s: (n_agents - 1) * n_actions = %d exceeds the LDS table", who, Da);
    const GatherGeom g1 = gather_geom(et, per_wave, 1);
    float* gpart = wsf + w.part;
    float* gred = gpart + (size_t)g1.blocks * g1.wpb * DaP * HP;
    const bool one_pass = w.Hs == 2 * HP && 2 * per_wave <= 64 * 1024;  // 65..128 units: both slabs per lane (float2)
    if (one_pass) {
        const GatherGeom g2 = gather_geom(et, 2 * per_wave, 2);
        float* gred2 = gpart + (size_t)g2.blocks * g2.wpb * DaP * 2 * HP;
        gather_launch<2>(g2, A, s, dz0, action, E, T, K, wsf + w.dS, gpart, 0, (long)w.Hs, (long)w.Hs);
        CM_CHECK_LAUNCH(who);
        if (Da > 0) {
            const int n = Da * 2 * HP;
            hipLaunchKernelGGL(k_reduce_partials, dim3((n + RED_COLS - 1) / RED_COLS), dim3(RED_COLS * RED_GROUPS), 0, s, gpart, (int)(g2.blocks * g2.wpb), n, 0, n, gred2);
            hipLaunchKernelGGL(k_transpose_ga, dim3(16), dim3(256), 0, s, gred2, H, Da, wsf + w.ga, 2 * HP);
            CM_CHECK_LAUNCH(who);
        }
    }
    for (int h0 = 0; h0 < H && !one_pass; h0 += 64) {
        const int hn = min(64, H - h0);
        gather_launch<1>(g1, A, s, dz0, action, E, T, K, wsf + w.dS, gpart, h0, (long)w.Hs, (long)w.Hs);
        CM_CHECK_LAUNCH(who);
        if (Da > 0) {
            const int n = Da * HP;
            hipLaunchKernelGGL(k_reduce_partials, dim3((n + RED_COLS - 1) / RED_COLS), dim3(RED_COLS * RED_GROUPS), 0, s, gpart, (int)(g1.blocks * g1.wpb), n, 0, n, gred);
            hipLaunchKernelGGL(k_transpose_ga, dim3(16), dim3(256), 0, s, gred, hn, Da, wsf + w.ga + (size_t)h0 * Da, HP);
            CM_CHECK_LAUNCH(who);
        }
    }
    for (int h0 = 0; h0 < H; h0 += 64)  // state block: dW0s = dS^T state
        if (int rc = stream_dw<true>(wsf + w.dS + h0, state, et, Ds, min(64, H - h0), wsf + w.part, wsf + w.gs + (size_t)h0 * Ds, s, who, w.Hs, state_ld))
            return rc;
    const int rest = (int)(cm_mlp_param_count(Dc, H, L, K) - (int64_t)H * Dc) + CM_NUM_STATS;
    hipLaunchKernelGGL(k_coma_scatter_grads, dim3(128), dim3(256), 0, s, wsf + w.gc, wsf + w.gs, wsf + w.ga, H, Dc, Ds, Do, rest, grad_and_stats);
    CM_CHECK_LAUNCH(who);
    return 0;
}

inline int ew_grid(long n) { long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }

}  // namespace

extern "C" int cm_coma_build_inputs(const float* state, const float* obs, const int32_t* action, int E, int A, int T, int Ds, int Do,
                                    int n_actions, float* out, cm_stream_t stream) {
    CM_REQUIRE(E > 0 && A > 0 && T > 0 && Ds > 0 && Do > 0 && n_actions > 0, "cm_coma_build_inputs: bad dims");
    const long total = (long)E * A * T * (Ds + Do + (A - 1) * n_actions);
    hipLaunchKernelGGL(k_coma_build_inputs, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, state, obs, action, E, A, T, Ds, Do,
                       n_actions, out);
    CM_CHECK_LAUNCH("cm_coma_build_inputs");
    return 0;
}

extern "C" int cm_gather_taken(const float* q, const int32_t* action, int64_t rows, int n_actions, float* out, cm_stream_t stream) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(k_gather_taken, dim3(ew_grid(rows)), dim3(256), 0, (hipStream_t)stream, q, action, (long)rows, n_actions, out);
    CM_CHECK_LAUNCH("cm_gather_taken");
    return 0;
}

extern "C" int cm_nstep_returns(const float* reward, const float* qtaken, const int32_t* ep_len, int E, int A, int T, double gamma,
                                int nsteps, float* ret, cm_stream_t stream) {
    CM_REQUIRE(E > 0 && A > 0 && T > 0 && nsteps >= 1, "cm_nstep_returns: bad dims E=%d A=%d T=%d n=%d", E, A, T, nsteps);
    hipLaunchKernelGGL(k_nstep_returns, dim3(ew_grid((long)E * A * T)), dim3(256), 0, (hipStream_t)stream, reward, qtaken, ep_len, E, A, T,
                       gamma, nsteps, ret);
    CM_CHECK_LAUNCH("cm_nstep_returns");
    return 0;
}

extern "C" size_t cm_coma_advantage_workspace_bytes(int E, int A, int T) { (void)E; (void)A; return (size_t)TS_CHUNKS * T * 4 * sizeof(double); }

extern "C" int cm_coma_advantage(const float* logits, const float* q, const int32_t* action, const int32_t* ep_len, int E, int A, int T,
                                 int n_actions, float* adv, double* tstats, void* ws, size_t ws_bytes, cm_stream_t stream) {
    CM_REQUIRE(E > 0 && A > 0 && T > 0 && n_actions > 0, "cm_coma_advantage: bad dims");
    CM_REQUIRE(ws && ws_bytes >= cm_coma_advantage_workspace_bytes(E, A, T), "cm_coma_advantage: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const long rows = (long)E * A * T;
    hipLaunchKernelGGL(k_coma_adv, dim3(ew_grid(rows)), dim3(256), 0, s, logits, q, action, rows, n_actions, adv);
    CM_CHECK_LAUNCH("cm_coma_advantage/adv");
    const long nseq = (long)E * A;
    const int chunks = (int)(nseq < TS_CHUNKS ? nseq : TS_CHUNKS);
    hipLaunchKernelGGL(k_tstats_partial, dim3((T + 63) / 64, chunks), dim3(256), 0, s, adv, action, ep_len, E, A, T, (double*)ws);
    CM_CHECK_LAUNCH("cm_coma_advantage/tstats");
    hipLaunchKernelGGL(k_tstats_final, dim3((4 * T + 255) / 256), dim3(256), 0, s, (const double*)ws, chunks, T, tstats);
    CM_CHECK_LAUNCH("cm_coma_advantage/final");
    return 0;
}

extern "C" int cm_coma_normalize_adv(float* adv, const double* tstats, int E, int A, int T, cm_stream_t stream) {
    hipLaunchKernelGGL(k_coma_normalize_adv, dim3(ew_grid((long)E * A * T)), dim3(256), 0, (hipStream_t)stream, adv, tstats, E, A, T);
    CM_CHECK_LAUNCH("cm_coma_normalize_adv");
    return 0;
}

extern "C" int cm_polyak_update(float* target, const float* src, int64_t n, double polyak, cm_stream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_polyak, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, target, src, (long)n, (float)polyak,
                       (float)(1.0 - polyak));
    CM_CHECK_LAUNCH("cm_polyak_update");
    return 0;
}

extern "C" size_t cm_mlp_split_workspace_bytes(int64_t rows, int din, int hidden, int n_hidden_layers, int dout) {
    if (wide_shape(hidden, n_hidden_layers)) return wide_ws_bytes((long)rows, din, hidden, n_hidden_layers, dout, true);
    return split_ws_bytes((long)rows, din, hidden, n_hidden_layers, dout);
}

extern "C" int cm_qcritic_fwd_bwd(const float* x, const int32_t* action, const float* target, const int32_t* ep_len, int E, int A, int T,
                                  int din, int hidden, int n_hidden_layers, int n_actions, const float* params, float* grad_and_stats,
                                  void* ws, size_t ws_bytes, cm_stream_t stream) {
    const bool wide = wide_shape(hidden, n_hidden_layers);  // layered schedule (cm_mlp_wide.h)
    if (!wide) if (int rc = check_shapes("cm_qcritic_fwd_bwd", din, hidden, n_hidden_layers, n_actions)) return rc;
    CM_REQUIRE(E > 0 && A > 0 && T > 0, "cm_qcritic_fwd_bwd: bad dims E=%d A=%d T=%d", E, A, T);
    const long rows = (long)E * A * T;
    if (int rc = check_rows("cm_qcritic_fwd_bwd", rows)) return rc;
    MlpArgs a = {};
    a.x = x; a.x_stride = din; a.rows = rows; a.din = din; a.H = hidden; a.L = n_hidden_layers; a.dout = n_actions;
    a.params = params; a.action = action; a.ret = target; a.ep_len = ep_len; a.A = A; a.T = T; a.per_agent = 1;
    if (wide) return wide_train<M_QCRITIC>(a, grad_and_stats, ws, ws_bytes, (hipStream_t)stream, "cm_qcritic_fwd_bwd");
    return run_train<M_QCRITIC>(a, grad_and_stats, ws, ws_bytes, (hipStream_t)stream, "cm_qcritic_fwd_bwd");
}

extern "C" int cm_coma_actor_fwd_bwd_ld(const float* obs, int64_t obs_ld, const uint8_t* avail, const int32_t* action, const float* adv,
                                        const int32_t* ep_len, int E, int A, int T, int din, int hidden, int n_hidden_layers, int n_actions,
                                        const float* params, double entropy_coef, float* grad_and_stats, void* ws, size_t ws_bytes,
                                        cm_stream_t stream);
extern "C" int cm_coma_actor_fwd_bwd(const float* obs, const uint8_t* avail, const int32_t* action, const float* adv,
                                     const int32_t* ep_len, int E, int A, int T, int din, int hidden, int n_hidden_layers, int n_actions,
                                     const float* params, double entropy_coef, float* grad_and_stats, void* ws, size_t ws_bytes,
                                     cm_stream_t stream) {
    return cm_coma_actor_fwd_bwd_ld(obs, din, avail, action, adv, ep_len, E, A, T, din, hidden, n_hidden_layers, n_actions, params, entropy_coef,
                                    grad_and_stats, ws, ws_bytes, stream);
}
extern "C" int cm_coma_actor_fwd_bwd_ld(const float* obs, int64_t obs_ld, const uint8_t* avail, const int32_t* action, const float* adv,
                                        const int32_t* ep_len, int E, int A, int T, int din, int hidden, int n_hidden_layers, int n_actions,
                                        const float* params, double entropy_coef, float* grad_and_stats, void* ws, size_t ws_bytes,
                                        cm_stream_t stream) {
    CM_REQUIRE(obs_ld >= din, "cm_coma_actor_fwd_bwd_ld: obs_ld=%ld below the width %d", (long)obs_ld, din);
    const bool wide = wide_shape(hidden, n_hidden_layers);  // layered schedule (cm_mlp_wide.h)
    if (!wide) if (int rc = check_shapes("cm_coma_actor_fwd_bwd", din, hidden, n_hidden_layers, n_actions)) return rc;
    CM_REQUIRE(E > 0 && A > 0 && T > 0, "cm_coma_actor_fwd_bwd: bad dims E=%d A=%d T=%d", E, A, T);
    const long rows = (long)E * A * T;
    if (int rc = check_rows("cm_coma_actor_fwd_bwd", rows)) return rc;
    MlpArgs a = {};
    a.x = obs; a.x_stride = (long)obs_ld; a.rows = rows; a.din = din; a.H = hidden; a.L = n_hidden_layers; a.dout = n_actions;
    a.params = params; a.avail = avail; a.avail_stride = n_actions; a.action = action; a.adv = adv; a.ep_len = ep_len;
    a.A = A; a.T = T; a.per_agent = 1; a.ent_coef = (float)entropy_coef;
    if (wide) return wide_train<M_COMA_ACTOR>(a, grad_and_stats, ws, ws_bytes, (hipStream_t)stream, "cm_coma_actor_fwd_bwd");
    return run_train<M_COMA_ACTOR>(a, grad_and_stats, ws, ws_bytes, (hipStream_t)stream, "cm_coma_actor_fwd_bwd");
}

extern "C" size_t cm_coma_critic_workspace_bytes(int E, int A, int T, int Ds, int Do, int n_actions, int hidden, int n_hidden_layers,
                                                 int train) {
    if (wide_shape(hidden, n_hidden_layers)) return coma_wide_ws(E, A, T, Ds, Do, n_actions, hidden, n_hidden_layers, train != 0).total * sizeof(float);
    return coma_ws(E, A, T, Ds, Do, n_actions, hidden, n_hidden_layers, train != 0).total * sizeof(float);
}

extern "C" int cm_coma_q_forward_ld(const float* state, int64_t state_ld, const float* obs, int64_t obs_ld, const int32_t* action,
                                    const uint8_t* avail, int E, int A, int T, int Ds, int Do, int n_actions, int hidden, int n_hidden_layers,
                                    const float* params, float* q, void* ws, size_t ws_bytes, cm_stream_t stream);
extern "C" int cm_coma_q_forward(const float* state, const float* obs, const int32_t* action, const uint8_t* avail, int E, int A, int T,
                                 int Ds, int Do, int n_actions, int hidden, int n_hidden_layers, const float* params, float* q, void* ws,
                                 size_t ws_bytes, cm_stream_t stream) {
    return cm_coma_q_forward_ld(state, Ds, obs, Do, action, avail, E, A, T, Ds, Do, n_actions, hidden, n_hidden_layers, params, q, ws, ws_bytes, stream);
}
extern "C" int cm_coma_q_forward_ld(const float* state, int64_t state_ld, const float* obs, int64_t obs_ld, const int32_t* action,
                                    const uint8_t* avail, int E, int A, int T, int Ds, int Do, int n_actions, int hidden, int n_hidden_layers,
                                    const float* params, float* q, void* ws, size_t ws_bytes, cm_stream_t stream) {
    CM_REQUIRE(E > 0 && A > 0 && T > 0 && Ds > 0, "cm_coma_q_forward: bad dims E=%d A=%d T=%d Ds=%d", E, A, T, Ds);
    CM_REQUIRE(state_ld >= Ds && obs_ld >= Do, "cm_coma_q_forward_ld: leading dimensions %ld / %ld below the widths %d / %d", (long)state_ld, (long)obs_ld, Ds, Do);
    if (wide_shape(hidden, n_hidden_layers))
        return coma_wide_q_forward(state, obs, action, avail, E, A, T, Ds, Do, n_actions, hidden, n_hidden_layers, params, q, ws, ws_bytes,
                                   (hipStream_t)stream, "cm_coma_q_forward", (long)obs_ld, (long)state_ld);
    if (int rc = check_shapes("cm_coma_q_forward", Do, hidden, n_hidden_layers, n_actions)) return rc;
    const ComaWs w = coma_ws(E, A, T, Ds, Do, n_actions, hidden, n_hidden_layers, false);
    CM_REQUIRE(ws && ws_bytes >= w.total * sizeof(float), "cm_coma_q_forward: workspace too small (%zu < %zu)", ws_bytes, w.total * sizeof(float));
    hipStream_t s = (hipStream_t)stream;
    float* wsf = (float*)ws;
    if (int rc = coma_prepare(state, obs, action, E, A, T, Ds, Do, n_actions, hidden, n_hidden_layers, params, wsf, w, s, "cm_coma_q_forward", (long)state_ld)) return rc;
    MlpArgs a = {};
    a.x = obs; a.x_stride = (long)obs_ld; a.rows = (long)E * A * T; a.din = Do; a.H = hidden; a.L = n_hidden_layers; a.dout = n_actions;
    a.params = wsf + w.pc; a.avail = avail; a.avail_stride = n_actions; a.y = q; a.z0_add = wsf + w.z0;
    const size_t lds_bytes = (size_t)make_lds(a.L, a.dout, (a.din + KC - 1) / KC).total * sizeof(float);
    launch_infer<M_FWD>(a, grid_for(a.rows), lds_bytes, s);
    CM_CHECK_LAUNCH("cm_coma_q_forward");
    return 0;
}

extern "C" int cm_coma_critic_fwd_bwd_ld(const float* state, int64_t state_ld, const float* obs, int64_t obs_ld, const int32_t* action,
                                         const float* target, const int32_t* ep_len, int E, int A, int T, int Ds, int Do, int n_actions, int hidden,
                                         int n_hidden_layers, const float* params, float* grad_and_stats, void* ws, size_t ws_bytes,
                                         cm_stream_t stream);
extern "C" int cm_coma_critic_fwd_bwd(const float* state, const float* obs, const int32_t* action, const float* target,
                                      const int32_t* ep_len, int E, int A, int T, int Ds, int Do, int n_actions, int hidden,
                                      int n_hidden_layers, const float* params, float* grad_and_stats, void* ws, size_t ws_bytes,
                                      cm_stream_t stream) {
    return cm_coma_critic_fwd_bwd_ld(state, Ds, obs, Do, action, target, ep_len, E, A, T, Ds, Do, n_actions, hidden, n_hidden_layers, params,
                                     grad_and_stats, ws, ws_bytes, stream);
}
extern "C" int cm_coma_critic_fwd_bwd_ld(const float* state, int64_t state_ld, const float* obs, int64_t obs_ld, const int32_t* action,
                                         const float* target, const int32_t* ep_len, int E, int A, int T, int Ds, int Do, int n_actions, int hidden,
                                         int n_hidden_layers, const float* params, float* grad_and_stats, void* ws, size_t ws_bytes,
                                         cm_stream_t stream) {
    const char* who = "cm_coma_critic_fwd_bwd";
    CM_REQUIRE(E > 0 && A > 0 && T > 0 && Ds > 0, "cm_coma_critic_fwd_bwd: bad dims E=%d A=%d T=%d Ds=%d", E, A, T, Ds);
    CM_REQUIRE(state_ld >= Ds && obs_ld >= Do, "cm_coma_critic_fwd_bwd_ld: leading dimensions %ld / %ld below the widths %d / %d", (long)state_ld, (long)obs_ld, Ds, Do);
    if (wide_shape(hidden, n_hidden_layers))
        return coma_wide_critic_fwd_bwd(state, obs, action, target, ep_len, E, A, T, Ds, Do, n_actions, hidden, n_hidden_layers, params,
                                        grad_and_stats, ws, ws_bytes, (hipStream_t)stream, who, (long)obs_ld, (long)state_ld);
    if (int rc = check_shapes(who, Do, hidden, n_hidden_layers, n_actions)) return rc;
    const long rows = (long)E * A * T;
    if (int rc = check_rows(who, rows)) return rc;
    const ComaWs w = coma_ws(E, A, T, Ds, Do, n_actions, hidden, n_hidden_layers, true);
    CM_REQUIRE(ws && ws_bytes >= w.total * sizeof(float), "cm_coma_critic_fwd_bwd: workspace too small (%zu < %zu)", ws_bytes, w.total * sizeof(float));
    hipStream_t s = (hipStream_t)stream;
    float* wsf = (float*)ws;
    const int K = n_actions, H = hidden, Da = (A - 1) * K, Dc = Ds + Do + Da;
    if (int rc = coma_prepare(state, obs, action, E, A, T, Ds, Do, K, H, n_hidden_layers, params, wsf, w, s, who, (long)state_ld)) return rc;
    MlpArgs a = {};
    a.x = obs; a.x_stride = (long)obs_ld; a.rows = rows; a.din = Do; a.H = H; a.L = n_hidden_layers; a.dout = K;
    a.params = wsf + w.pc; a.action = action; a.ret = target; a.ep_len = ep_len; a.A = A; a.T = T; a.per_agent = 1;
    a.z0_add = wsf + w.z0; a.dz0 = wsf + w.dz0;
    if (int rc = run_train<M_QCRITIC>(a, wsf + w.gc, wsf + w.train, (w.total - w.train) * sizeof(float), s, who)) return rc;
    // one pass over dZ0: dS = sum_a dZ0 and the action block dW0a (per-wave LDS tables -> partials -> fixed-order reduce)
    const long et = (long)E * T;
    CM_REQUIRE(A <= OH_MAXA, "%s: n_agents=%d > %d is not supported by the factored critic input", who, A, OH_MAXA);
    {
        const int DaP = Da > 0 ? Da : 1;
        const size_t per_wave = (size_t)DaP * HP * sizeof(float);
        CM_REQUIRE(per_wave <= 64 * 1024, "%s: (n_agents - 1) * n_actions = %d exceeds the LDS table", who, Da);
        const GatherGeom g1 = gather_geom(et, per_wave, 1);
        float* gpart = wsf + w.part;                                    // [blocks * wpb][Da][HP]
        float* gred = gpart + (size_t)g1.blocks * g1.wpb * DaP * HP;    // [Da][HP]
        const long blocks = g1.blocks; const int wpb = g1.wpb;
        gather_launch<1>(g1, A, s, wsf + w.dz0, action, E, T, K, wsf + w.dS, gpart, 0, (long)HP, (long)HP);
        CM_CHECK_LAUNCH(who);
        if (Da > 0) {
            const int n = Da * HP;
            hipLaunchKernelGGL(k_reduce_partials, dim3((n + RED_COLS - 1) / RED_COLS), dim3(RED_COLS * RED_GROUPS), 0, s, gpart, (int)(blocks * wpb), n, 0, n, gred);
            CM_CHECK_LAUNCH(who);
            hipLaunchKernelGGL(k_transpose_ga, dim3(16), dim3(256), 0, s, gred, H, Da, wsf + w.ga, HP);
            CM_CHECK_LAUNCH(who);
        }
    }
    // state block: dW0s = dS^T state
    if (int rc = stream_dw(wsf + w.dS, state, et, Ds, H, wsf + w.part, wsf + w.gs, s, who, HP, (long)state_ld)) return rc;
    const int rest = (int)(cm_mlp_param_count(Dc, H, n_hidden_layers, K) - (int64_t)H * Dc) + CM_NUM_STATS;
    hipLaunchKernelGGL(k_coma_scatter_grads, dim3(128), dim3(256), 0, s, wsf + w.gc, wsf + w.gs, wsf + w.ga, H, Dc, Ds, Do, rest, grad_and_stats);
    CM_CHECK_LAUNCH(who);
    return 0;
}
