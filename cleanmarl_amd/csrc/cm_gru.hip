// cm_gru.hip -- GRU actor: rollout step and fused TBPTT chunk forward + backward-through-time.
//
// Reference: Actor of cleanmarl/mappo_lstm_multienvs.py:162-184
//     x1 = relu(fc1(obs));  h' = GRUCell(x1, h);  logits = Linear(relu(h'));  masked_fill(~avail, -1e9)
// and the truncated-BPTT actor update of :562-620: for every chunk of `tbptt` steps the loss
//     sum_{t in chunk} (-pg_t - c_ent * ent_t) / (N_chunk * T_chunk)
// is back-propagated through the chunk only (h detached at the chunk boundary) and followed by an optimiser step.
//
// One workgroup owns a tile of 64 (env,agent) sequences; the chunk is walked by two launches with the same tiling
// (k_gru_chunk_fwd keeps no accumulators, k_gru_chunk_bwd keeps eight MFMA accumulator tiles -- as one kernel the
// register allocator spilled ~500 VGPRs): forward t0..t1-1 (saving
// x1, r, z, n, W_hn h + b_hn and h' per step to a workspace in HBM and the per-step dlogits of the PPO head),
// then backward t1-1..t0 with dh carried in LDS.  Every GEMM is a 64x64x64 block on v_mfma_f32_32x32x2_f32
// (same three forms as cm_mlp_kernel.h); gate weight blocks are streamed through LDS, weight-gradient
// accumulators (W_ih, W_hh: 6 blocks, fc1, fc2) stay in registers for the whole chunk and are written once
// per workgroup, then folded by k_reduce_partials.  Time is inherently sequential here; parallelism is over
// sequences only (SURVEY.md §7 "hard parts" (e)).
#include <atomic>
#include "cm_mlp_train.h"
#ifdef CM_PHASE_PROF
extern unsigned long long* g_prof;
#endif

namespace {

// register-staged 64x64 block prefetch (Tile16 of cm_mlp_kernel.h).  The six H x H gate blocks use 16-byte loads when
// the kernel is instantiated with WV (H % 4 == 0 and 16-byte aligned parameters); the obs tile and W1 (row stride
// din, arbitrary) always use 4-byte loads -- they are one block out of seven per step.
template <bool WV>
__device__ __forceinline__ void gate_load(Tile16& t, const float* src, int H) { tile_load<WV>(t, src, 0, H, H, 0, H); }
__device__ __forceinline__ void x_load(Tile16& t, const float* src, long row0, long nrows, long stride, int ncols) {
    tile_load<false>(t, src, row0, nrows, stride, 0, ncols);
}
inline bool gru_wvec(const float* params, int H) { return (H % 4 == 0) && ((reinterpret_cast<uintptr_t>(params) & 15) == 0); }

struct GruOff { int W1, b1, Wih, Whh, bih, bhh, W2, b2, P; };
__host__ __device__ inline GruOff gru_offsets(int din, int H, int K) {
    GruOff o;
    o.W1 = 0; o.b1 = H * din; o.Wih = o.b1 + H; o.Whh = o.Wih + 3 * H * H; o.bih = o.Whh + 3 * H * H;
    o.bhh = o.bih + 3 * H; o.W2 = o.bhh + 3 * H; o.b2 = o.W2 + K * H; o.P = o.b2 + K;
    return o;
}

constexpr int WS_ACT = 6 * HP;  // per (step,row): x1 | r | z | n | ghn | hnew, each padded to HP floats
constexpr int WS_DL = KMAX;     // per (step,row): dlogits padded to KMAX

struct GruArgs {
    const float* obs; const uint8_t* avail; const int* action; const float* logp_old; const float* adv; const int* ep_len;
    int E, A, T, t0, t1, din, H, K;
    const float* params; const float* h_in; float* h_out;
    float clip_lo, clip_hi, clip_eps, ent_coef;
    float* ws_act; float* ws_dl; float* partial; int PS;
    // act mode
    const float* x; long x_stride; const uint8_t* av; long av_stride; long rows; float* h;
    unsigned long long seed; long row_offset; int t; int* action_out; float* logp_out; long out_stride;
    unsigned long long* prof;  // CM_PHASE_PROF builds only
};

// gate non-linearities on the hardware exp (v_exp_f32, ~1 ulp): 48 transcendental evaluations per lane per step make
// the libm versions (~30 instructions each) a visible part of the sequential per-step latency
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }  // v_rcp_f32: 1 ulp, one instruction
__device__ __forceinline__ float tanhf_(float x) {
    const float e = __expf(-2.0f * fabsf(x));           // in (0, 1]: no overflow
    const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    return copysignf(t, x);
}

#include "cm_gru_step2.h"

// LDS carve: 8 activation/weight buffers + head weights + per-row scratch + biases
struct GruLds { float *b[8], *wouts, *ls, *b1, *bih, *bhh, *b2, *red; };
__device__ __forceinline__ GruLds gru_lds(float* smem, int KP) {
    GruLds s; float* p = smem;
    for (int i = 0; i < 8; ++i) { s.b[i] = p; p += TM * LDT; }
    s.wouts = p; p += KP * HP;
    s.ls = p; p += TM * LSP;
    s.b1 = p; p += HP; s.bih = p; p += 3 * HP; s.bhh = p; p += 3 * HP; s.b2 = p; p += KMAX;
    s.red = p;
    return s;
}
inline size_t gru_lds_bytes(int K) {
    const int KP = (K <= 8) ? 8 : KMAX;
    return (size_t)(8 * TM * LDT + KP * HP + TM * LSP + HP + 6 * HP + KMAX + 4 * HP) * sizeof(float);
}

__device__ __forceinline__ void gru_stage_consts(const GruLds& L, const GruArgs& a, const GruOff& off, int KP) {
    const int tid = threadIdx.x, H = a.H, K = a.K;
    for (int i = tid; i < KP * HP; i += NTHREADS) {
        const int k = i / HP, c = i % HP;
        L.wouts[i] = (c < H && k < K) ? a.params[off.W2 + k * H + c] : 0.0f;
    }
    for (int i = tid; i < HP; i += NTHREADS) L.b1[i] = (i < H) ? a.params[off.b1 + i] : 0.0f;
    for (int i = tid; i < 3 * HP; i += NTHREADS) {
        const int g = i / HP, c = i % HP;
        L.bih[i] = (c < H) ? a.params[off.bih + g * H + c] : 0.0f;
        L.bhh[i] = (c < H) ? a.params[off.bhh + g * H + c] : 0.0f;
    }
    for (int i = tid; i < KMAX; i += NTHREADS) L.b2[i] = (i < K) ? a.params[off.b2 + i] : 0.0f;
    for (int i = tid; i < TM * LSP; i += NTHREADS) L.ls[i] = 0.0f;
}

// One GRU forward step for the tile whose obs rows are already addressable through (xbase, xstride).
// X=b0 W=b1 A1=b2 HP=hp HN=hn GR=b5 GZ=b6 W2=b7.  SAVE: also write the activations the backward pass needs.
// tA / tB arrive holding this step's obs tile and W1 (issued by the caller or by the previous step); every weight block
// of the step is requested one phase ahead of the ds_write that needs it, and the next step's obs tile + W1 are
// requested under the last MFMA phase, so no staging load is ever waited for right after it was issued.
template <bool SAVE, bool WV>
__device__ __forceinline__ void gru_fwd_step(const GruLds& L, const GruArgs& a, const GruOff& off, Tile16& tA, Tile16& tB,
                                             const float* xnext, long xstride, long row0, long nrows,
                                             float* hp, float* hn, float* wsrow /* ws_act + (s*R + row0)*WS_ACT */) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, lc = lane & 31;
    const int H = a.H, din = a.din;
    float *X = L.b[0], *W = L.b[1], *A1 = L.b[2], *GR = L.b[5], *GZ = L.b[6], *W2 = L.b[7];
    const int col = 32 * wn + lc;
    f32x16 acc;
    // ---- F1: x1 = relu(fc1(obs))
    __syncthreads();
    tile_store<false>(X, tA);
    tile_store<false>(W, tB);
    gate_load<WV>(tA, a.params + off.Wih, H);
    gate_load<WV>(tB, a.params + off.Whh, H);
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
    rowpar_nt(acc, X + 32 * wm * LDT, W + 32 * wn * LDT, (din + 7) >> 3);
    {
        const float bias = L.b1[col];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int row = 32 * wm + (g & 3) + 8 * (g >> 2) + 4 * h;
            const float v = fmaxf(acc[g] + bias, 0.0f);
            A1[row * LDT + col] = v;
            if (SAVE && row0 + row < nrows) wsrow[(long)row * WS_ACT + 0 * HP + col] = v;
        }
    }
    // ---- F2: r, z gates
#pragma unroll
    for (int gate = 0; gate < 2; ++gate) {
        __syncthreads();  // A1 complete / previous readers of W, W2 done
        tile_store<WV>(W, tA);
        tile_store<WV>(W2, tB);
        gate_load<WV>(tA, a.params + off.Wih + (gate + 1) * H * H, H);
        gate_load<WV>(tB, a.params + off.Whh + (gate + 1) * H * H, H);
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
        rowpar_nt(acc, A1 + 32 * wm * LDT, W + 32 * wn * LDT, HP / 8);
        rowpar_nt(acc, hp + 32 * wm * LDT, W2 + 32 * wn * LDT, HP / 8);
        float* G = gate == 0 ? GR : GZ;
        const float bias = L.bih[gate * HP + col] + L.bhh[gate * HP + col];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int row = 32 * wm + (g & 3) + 8 * (g >> 2) + 4 * h;
            const float v = sigmoidf_(acc[g] + bias);
            G[row * LDT + col] = v;
            if (SAVE && row0 + row < nrows) wsrow[(long)row * WS_ACT + (1 + gate) * HP + col] = v;
        }
    }
    // ---- F3: candidate n and the new hidden state
    __syncthreads();
    tile_store<WV>(W, tA);
    tile_store<WV>(W2, tB);
    if (xnext) {  // next step's obs tile + W1 land under this phase's MFMAs
        x_load(tA, xnext, row0, nrows, xstride, din);
        x_load(tB, a.params + off.W1, 0, H, din, din);
    }
    __syncthreads();
    f32x16 acch;
#pragma unroll
    for (int g = 0; g < 16; ++g) { acc[g] = 0.0f; acch[g] = 0.0f; }
    rowpar_nt(acc, A1 + 32 * wm * LDT, W + 32 * wn * LDT, HP / 8);
    rowpar_nt(acch, hp + 32 * wm * LDT, W2 + 32 * wn * LDT, HP / 8);
    {
        const float bi = L.bih[2 * HP + col], bh = L.bhh[2 * HP + col];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int row = 32 * wm + (g & 3) + 8 * (g >> 2) + 4 * h;
            const float ghn = acch[g] + bh;
            const float r = GR[row * LDT + col], z = GZ[row * LDT + col], hprev = hp[row * LDT + col];
            const float n = tanhf_(acc[g] + bi + r * ghn);
            const float hv = (col < H) ? (1.0f - z) * n + z * hprev : 0.0f;
            hn[row * LDT + col] = hv;
            if (SAVE && row0 + row < nrows) {
                float* w = wsrow + (long)row * WS_ACT;
                w[3 * HP + col] = n; w[4 * HP + col] = ghn; w[5 * HP + col] = hv;
            }
        }
    }
    __syncthreads();
}

// logits of the head for this lane's row (4 lanes per row): zreg[j] holds k = 4j + hq
template <int KJ>
__device__ __forceinline__ void gru_head_logits(const GruLds& L, const float* hn, int K, const unsigned char* avb, float (&zreg)[KJ]) {
    const int tid = threadIdx.x, hrow = tid >> 2, hq = tid & 3;
    float hreg[16];
    const float4* hp4 = reinterpret_cast<const float4*>(hn + hrow * LDT + 16 * hq);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 v = hp4[i];
        hreg[4 * i] = fmaxf(v.x, 0.f); hreg[4 * i + 1] = fmaxf(v.y, 0.f); hreg[4 * i + 2] = fmaxf(v.z, 0.f); hreg[4 * i + 3] = fmaxf(v.w, 0.f);
    }
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
        zreg[j] = -1e9f;
        if (4 * j < K) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = 4 * j + q;
                if (k < K) {
                    const float4* wp4 = reinterpret_cast<const float4*>(L.wouts + k * HP + 16 * hq);
                    float p = 0.0f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float4 w4 = wp4[i];
                        p = fmaf(hreg[4 * i], w4.x, p); p = fmaf(hreg[4 * i + 1], w4.y, p);
                        p = fmaf(hreg[4 * i + 2], w4.z, p); p = fmaf(hreg[4 * i + 3], w4.w, p);
                    }
                    p = quad_sum(p);
                    if (hq == q) zreg[j] = avb[j] ? p + L.b2[k] : -1e9f;
                }
            }
        }
    }
}

// ============================================================================================ chunk fwd + bwd
template <int KJ, bool WV>
__global__ __launch_bounds__(NTHREADS) void k_gru_chunk_fwd(const GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KP = KJ * 4;
    const GruOff off = gru_offsets(a.din, a.H, a.K);
    const GruLds L = gru_lds(smem, KP);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, lc = lane & 31;
    const int hrow = tid >> 2, hq = tid & 3;
    const int H = a.H, K = a.K, din = a.din, T = a.T, CL = a.t1 - a.t0;
    const long R = (long)a.E * a.A;
    const int col = 32 * wn + lc;
    gru_stage_consts(L, a, off, KP);

    float st_pg = 0.f, st_ent = 0.f, st_kl = 0.f, st_clip = 0.f, st_cnt = 0.f;
    const long ntiles = (R + TM - 1) / TM;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = tile * TM;
        const int grow = (int)row0 + hrow;
        const bool rvalid = grow < R;
        const int e_row = rvalid ? grow / a.A : 0;
        const int ag = grow - e_row * a.A;
        const int eplen = rvalid ? a.ep_len[e_row] : 0;
        float* hp = L.b[3];
        float* hn = L.b[4];
        // h_in -> hp
        __syncthreads();
        for (int i = tid; i < TM * HP; i += NTHREADS) {
            const int r = i >> 6, c = i & 63;
            hp[r * LDT + c] = (row0 + r < R && c < H && a.h_in) ? a.h_in[(row0 + r) * H + c] : 0.0f;
        }
        // ================================ forward over the chunk
        Tile16 tA, tB;
        x_load(tA, a.obs + (long)a.t0 * din, row0, R, (long)T * din, din);
        x_load(tB, a.params + off.W1, 0, H, din, din);
        for (int s = 0; s < CL; ++s) {
            const int t = a.t0 + s;
            gru_fwd_step<true, WV>(L, a, off, tA, tB, (s + 1 < CL) ? a.obs + (long)(t + 1) * din : nullptr, (long)T * din, row0, R,
                               hp, hn, a.ws_act + (s * R + row0) * WS_ACT);
            // ---- PPO head on relu(h'): statistics + dlogits (saved for the backward sweep)
            unsigned char avb[KJ];
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                avb[j] = 1;
                if (rvalid && 4 * j + hq < K) avb[j] = a.avail[((long)grow * T + t) * K + 4 * j + hq];
            }
            float zreg[KJ];
            gru_head_logits<KJ>(L, hn, K, avb, zreg);
            {
                const bool valid = rvalid && t < eplen;
                const float invA = 1.0f / (float)a.A;
                const long o = (long)grow * T + t;
                const int act = rvalid ? a.action[o] : 0;
                const float lpo = rvalid ? a.logp_old[o] : 0.f, advv = rvalid ? a.adv[o] : 0.f;
                float m = -INFINITY;
#pragma unroll
                for (int j = 0; j < KJ; ++j) if (4 * j + hq < K) m = fmaxf(m, zreg[j]);
                m = quad_max(m);
                float ssum = 0.0f, pj[KJ];
#pragma unroll
                for (int j = 0; j < KJ; ++j) { pj[j] = 0.f; if (4 * j + hq < K) { pj[j] = expf(zreg[j] - m); ssum += pj[j]; } }
                ssum = quad_sum(ssum);
                const float lse = m + logf(ssum), rs = 1.0f / ssum;
                float ent = 0.f, lpa = 0.f;
#pragma unroll
                for (int j = 0; j < KJ; ++j) if (4 * j + hq < K) {
                    const float lp = zreg[j] - lse;
                    pj[j] *= rs; ent -= pj[j] * lp;
                    if (4 * j + hq == act) lpa = lp;
                }
                ent = quad_sum(ent); lpa = quad_sum(lpa);
                const float log_ratio = lpa - lpo, ratio = expf(log_ratio);
                const float pg1 = advv * ratio, pg2 = advv * fminf(fmaxf(ratio, a.clip_lo), a.clip_hi);
                const bool inr = (ratio >= a.clip_lo) && (ratio <= a.clip_hi);
                float gsel;
                if (pg1 < pg2) gsel = advv; else if (pg1 > pg2) gsel = inr ? advv : 0.f; else gsel = 0.5f * advv + (inr ? 0.5f * advv : 0.f);
                if (valid && hq == 0) {
                    st_pg += invA * fminf(pg1, pg2); st_ent += invA * ent; st_kl += invA * ((ratio - 1.f) - log_ratio);
                    st_clip += (fabsf(ratio - 1.f) > a.clip_eps) ? invA : 0.f;
                    if (ag == 0) st_cnt += 1.f;
                }
                const float gr = gsel * ratio;
#pragma unroll
                for (int j = 0; j < KJ; ++j) {
                    const int k = 4 * j + hq;
                    if (k < K && rvalid) {
                        const float lp = zreg[j] - lse;
                        float d = invA * (-gr * ((k == act ? 1.f : 0.f) - pj[j]) + a.ent_coef * pj[j] * (lp + ent));
                        if (!valid || zreg[j] <= -5e8f) d = 0.f;
                        a.ws_dl[(s * R + grow) * WS_DL + k] = d;
                    }
                }
            }
            float* tmp = hp; hp = hn; hn = tmp;
        }
        // h at the end of the chunk (detached carry for the next chunk)
        __syncthreads();
        if (a.h_out)
            for (int i = tid; i < TM * HP; i += NTHREADS) {
                const int r = i >> 6, c = i & 63;
                if (row0 + r < R && c < H) a.h_out[(row0 + r) * H + c] = hp[r * LDT + c];
            }
    }
    // statistics of this workgroup (the gradient part of the partial row is written by k_gru_chunk_bwd)
    float* out = a.partial + (size_t)blockIdx.x * a.PS;
    float sv6[6] = {st_pg, st_ent, st_kl, st_clip, 0.f, st_cnt};
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const float v = cm_wave_sum(sv6[q]);
        if (lane == 0) L.red[q * 4 + wave] = v;
    }
    __syncthreads();
    if (tid < CM_NUM_STATS) {
        float v = 0.f;
        if (tid < 6) v = L.red[tid * 4] + L.red[tid * 4 + 1] + L.red[tid * 4 + 2] + L.red[tid * 4 + 3];
        out[off.P + tid] = v;
    }
}

// 32-row tiles (cm_gru_v2.h, the fused rollout below): tile height and the register-staged obs tile
constexpr int T32 = 32;

// obs tile of 32 rows x din (<= 64) columns, register-staged one step ahead by the four recurrence waves: wave w takes rows
// 8w .. 8w+7, one row per load with lane = column, so an instruction touches the two or three cache lines of ONE row.  (The first
// mapping -- 8 lanes per row, 8 columns each -- spread every instruction over ~24 lines of 8 rows and cost ~1.3 k cycles per step at
// issue: profiles/r03_phase_gru.txt.)
struct X32 { float v[8]; };
__device__ __forceinline__ void x32_load(X32& x, const float* src, long row0, long nrows, long stride, int ncols) {
    const int lane = threadIdx.x & 63, r0 = (threadIdx.x >> 6) * 8;
    const float* p = src + (row0 + r0) * stride + lane;
#pragma unroll
    for (int i = 0; i < 8; ++i) x.v[i] = (row0 + r0 + i < nrows && lane < ncols) ? p[i * stride] : 0.0f;
}
__device__ __forceinline__ void x32_store(float* XA, const X32& x) {
    const int lane = threadIdx.x & 63, r0 = (threadIdx.x >> 6) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) XA[(r0 + i) * LDT + lane] = x.v[i];
}

// ============================================================================================ fused GRU rollout
// The whole episode of the synthetic MPE-like env with the GRU actor in ONE persistent launch (replaces T x (cm_gru_policy_act +
// cm_synth_env_step), cleanmarl/mappo_lstm_multienvs.py:392-479 on the synthetic configs): a workgroup keeps floor(32 / A) envs
// (positions, velocities, landmarks) and their agents' hidden states in LDS for all T steps, the seven weight blocks stay
// resident, observations / states / actions / log-probs / rewards are written straight into the [E,A,T,F] buffers.
// Same Philox keys and the same arithmetic as the per-step kernels, so both paths produce the same rollout.
struct GruRollArgs {
    float* env_state; int E, A, T, agent_ids;
    unsigned long long seed, act_seed; long env_offset, episode;
    const float* params; int din, H, K;
    float* obs; float* state; int* action; float* logp; float* reward;
    long state_ld;  // row stride of `state` (>= 6 A A; the learner's critic reads 16-byte aligned rows when it is a multiple of 4)
    unsigned long long* prof;  // CM_PHASE_PROF builds only
};
constexpr float GR_DAMP = 0.25f, GR_DT = 0.1f, GR_ACCEL = 5.0f, GR_COLLIDE = 0.3f;  // cm_env.hip / cm_rollout.hip constants
constexpr int G32R_LDS_FLOATS = 4 * T32 * LDT + 8 * HP + KMAX + 64 + T32 * 8 + 3 * T32 * 2 + 2 * T32 + 4 * T32 + 16;  // 4 tiles, head weights, ls, pos/vel/landmarks, reward partials, 2 x long[32]

__global__ __launch_bounds__(NTHREADS, 1) void k_gru32_rollout(const GruRollArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KJ = 2, KP = 8;
    const GruOff off = gru_offsets(a.din, a.H, a.K);
    float* p = smem;
    float* XA = p; p += T32 * LDT;   // obs tile
    float* X1 = p; p += T32 * LDT;   // x1
    float* hp = p; p += T32 * LDT;
    float* hn = p; p += T32 * LDT;
    GruLds L = {};
    L.wouts = p; p += KP * HP;
    L.b2 = p; p += KMAX;
    L.red = p; p += 64;
    float* ls = p; p += T32 * 8;
    float* epos = p; p += T32 * 2;
    float* evel = p; p += T32 * 2;
    float* elm = p; p += T32 * 2;
    float* rscr = p; p += 2 * T32;
    long* obase = reinterpret_cast<long*>(p);  // 8-byte aligned: every carve above is a multiple of 2 floats
    long* sbase = obase + T32;
    const int tid = threadIdx.x;
    const int hrow = tid >> 2, hq = tid & 3;
    const int A = a.A, T = a.T, K = a.K, H = a.H, din = a.din;
    const int EPT = T32 / A, RT = EPT * A; const long Ds = a.state_ld;  // row stride of the state buffer
    G2W w;  // the seven weight blocks: this wave's 16 hidden columns in registers for the whole episode (cm_gru_step2.h)
    g2_load_weights<false>(w, a.params, off, din, H);
    for (int i = tid; i < KP * HP; i += NTHREADS) {
        const int k = i / HP, c = i % HP;
        L.wouts[i] = (c < H && k < K) ? a.params[off.W2 + k * H + c] : 0.0f;
    }
    for (int i = tid; i < KMAX; i += NTHREADS) L.b2[i] = (i < K) ? a.params[off.b2 + i] : 0.0f;

    PH_DECL
    const int ntiles = (a.E + EPT - 1) / EPT;
    const float inv_din = 1.0f / (float)din, inv_sw = 1.0f / (float)(6 * A);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int e0 = tile * EPT;
        __syncthreads();
        // ---- reset (cm_env.hip k_env_reset): thread per (env, agent) row; h = 0 at the start of the episode
        if (tid < T32) {
            const int el = tid / A, i = tid - el * A;
            const long e = e0 + el;
            const bool live = tid < RT && e < a.E;
            obase[tid] = live ? (e * A + i) * (long)T * din : -1;
            sbase[tid] = live ? e * (long)T * Ds + (long)i * 6 * A : -1;
            if (live) {
                const unsigned long long ge = (unsigned long long)(a.env_offset + e);
                const cm_u4 ra = cm_philox4x32((uint32_t)ge, (uint32_t)a.episode, (uint32_t)i, CM_STREAM_ENV_RESET,
                                               (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
                epos[2 * tid] = 2.0f * cm_u01(ra.x) - 1.0f; epos[2 * tid + 1] = 2.0f * cm_u01(ra.y) - 1.0f;
                elm[2 * tid] = 2.0f * cm_u01(ra.z) - 1.0f; elm[2 * tid + 1] = 2.0f * cm_u01(ra.w) - 1.0f;
            } else {
                epos[2 * tid] = epos[2 * tid + 1] = 0.0f; elm[2 * tid] = elm[2 * tid + 1] = 0.0f;
            }
            evel[2 * tid] = 0.0f; evel[2 * tid + 1] = 0.0f;
        }
        for (int i = tid; i < T32 * LDT; i += NTHREADS) hp[i] = 0.0f;
        auto reward_partials = [&]() {  // nearest-agent distance per landmark, collisions per agent (current positions)
            if (tid < RT) {
                const int el = tid / A, l = tid - el * A;
                const float* pos = epos + el * 2 * A;
                const float lx = elm[2 * tid], ly = elm[2 * tid + 1];
                const float qx = pos[2 * l], qy = pos[2 * l + 1];
                float best = 3.0e38f, col = 0.0f;
                for (int j = 0; j < A; ++j) {
                    const float dx = pos[2 * j] - lx, dy = pos[2 * j + 1] - ly;
                    best = fminf(best, __builtin_amdgcn_sqrtf(dx * dx + dy * dy));
                    if (j > l) {
                        const float cx = qx - pos[2 * j], cy = qy - pos[2 * j + 1];
                        if (__builtin_amdgcn_sqrtf(cx * cx + cy * cy) < GR_COLLIDE) col += 1.0f;
                    }
                }
                rscr[tid] = best; rscr[T32 + tid] = col;
            }
        };
        for (int t = 0; t < T; ++t) {
            __syncthreads();
            PH(0);
            if (t > 0) reward_partials();  // reward of step t-1 from the positions after its physics update
            // ---- observations of step t -> XA: 4 lanes per row (threads 0..127), lane hq handles entities j = hq, hq+4, ...
            if (hrow < T32) {
                const int el = hrow / A, i = hrow - el * A;
                const bool live = hrow < RT && (e0 + el) < a.E;
                float* xr = XA + hrow * LDT;
                if (live) {
                    const float* pos = epos + el * 2 * A; const float* vel = evel + el * 2 * A; const float* lm = elm + el * 2 * A;
                    const float px = pos[2 * i], py = pos[2 * i + 1];
                    if (hq == 0) { xr[0] = vel[2 * i]; xr[1] = vel[2 * i + 1]; xr[2] = px; xr[3] = py; }
                    for (int j = hq; j < A; j += 4) {
                        xr[4 + 2 * j] = lm[2 * j] - px; xr[5 + 2 * j] = lm[2 * j + 1] - py;
                        if (j != i) {
                            const int jj = j < i ? j : j - 1;
                            xr[4 + 2 * A + 2 * jj] = pos[2 * j] - px; xr[5 + 2 * A + 2 * jj] = pos[2 * j + 1] - py;
                            xr[2 + 4 * A + 2 * jj] = 0.0f; xr[3 + 4 * A + 2 * jj] = 0.0f;  // comm channel
                        }
                        if (a.agent_ids) xr[6 * A + j] = (j == i) ? 1.0f : 0.0f;
                    }
                    for (int c = din + hq; c < KC; c += 4) xr[c] = 0.0f;
                } else {
#pragma unroll
                    for (int j = 0; j < KC / 4; ++j) xr[4 * j + hq] = 0.0f;
                }
            }
            PH(1);
            __syncthreads();
            PH(2);
            // ---- rollout-buffer writes: flat (row, column) enumeration, consecutive threads -> consecutive addresses
            for (int idx = tid; idx < RT * din; idx += NTHREADS) {
                const int r = (int)(((float)idx + 0.5f) * inv_din), c = idx - r * din;
                const long ob = obase[r];
                if (ob >= 0) a.obs[ob + (long)t * din + c] = XA[r * LDT + c];
            }
            for (int idx = tid; idx < RT * 6 * A; idx += NTHREADS) {
                const int r = (int)(((float)idx + 0.5f) * inv_sw), c = idx - r * 6 * A;
                const long sb = sbase[r];
                if (sb >= 0) a.state[sb + (long)t * Ds + c] = XA[r * LDT + c];
            }
            if (t > 0 && tid < EPT && e0 + tid < a.E) {
                float r = 0.0f;
                for (int l = 0; l < A; ++l) r -= rscr[tid * A + l];
                for (int l = 0; l < A; ++l) r -= rscr[T32 + tid * A + l];
                a.reward[(long)(e0 + tid) * T + (t - 1)] = r;
            }
            float u_row = 0.0f;  // the step's uniform does not depend on the logits: Philox rounds overlap the GRU step
            if (tid < RT) {
                const int el = tid / A, i = tid - el * A;
                const unsigned long long gr = (unsigned long long)((a.env_offset + e0 + el) * A + i);
                const cm_u4 rnd = cm_philox4x32((uint32_t)gr, (uint32_t)(gr >> 32), (uint32_t)t, CM_STREAM_ACT,
                                                (uint32_t)a.act_seed, (uint32_t)(a.act_seed >> 32));
                u_row = cm_u01(rnd.x);
            }
            PH(3);
            // ---- GRU step: the obs tile is complete since the barrier above, h_{t-1} since the end of the previous step
            gru2_step<false>(w, XA, X1, hp, hn, nullptr, nullptr, nullptr, nullptr, din, H);
            PH(4);
            if (hrow < T32) {
                unsigned char avb[KJ] = {1, 1};
                float zreg[KJ];
                gru_head_logits<KJ>(L, hn, K, avb, zreg);
#pragma unroll
                for (int j = 0; j < KJ; ++j)
                    if (4 * j + hq < K) ls[hrow * 8 + 4 * j + hq] = zreg[j];
            }
            PH(5);
            __syncthreads();
            PH(6);
            if (tid < RT) {
                const int el = tid / A, i = tid - el * A;
                const long e = e0 + el;
                if (e < a.E) {
                    int chosen; float lpv;
                    cm_categorical_sample(ls + tid * 8, K, u_row, &chosen, &lpv);
                    const long o = (e * A + i) * (long)T + t;
                    a.action[o] = chosen;
                    a.logp[o] = lpv;
                    const float ux = (chosen == 1) ? -GR_ACCEL : (chosen == 2 ? GR_ACCEL : 0.0f);
                    const float uy = (chosen == 3) ? -GR_ACCEL : (chosen == 4 ? GR_ACCEL : 0.0f);
                    const float vx = evel[2 * tid] * (1.0f - GR_DAMP) + ux * GR_DT;
                    const float vy = evel[2 * tid + 1] * (1.0f - GR_DAMP) + uy * GR_DT;
                    evel[2 * tid] = vx; evel[2 * tid + 1] = vy;
                    epos[2 * tid] += vx * GR_DT; epos[2 * tid + 1] += vy * GR_DT;
                }
            }
            float* tmp = hp; hp = hn; hn = tmp;
            PH(7);
        }
        __syncthreads();
        reward_partials();
        __syncthreads();
        if (tid < EPT && e0 + tid < a.E) {
            float r = 0.0f;
            for (int l = 0; l < A; ++l) r -= rscr[tid * A + l];
            for (int l = 0; l < A; ++l) r -= rscr[T32 + tid * A + l];
            a.reward[(long)(e0 + tid) * T + (T - 1)] = r;
        }
        if (tid < RT) {
            const int el = tid / A, i = tid - el * A;
            const long e = e0 + el;
            if (e < a.E) {
                float* es = a.env_state + e * 6 * A;
                es[2 * i] = epos[2 * tid]; es[2 * i + 1] = epos[2 * tid + 1];
                es[2 * A + 2 * i] = evel[2 * tid]; es[2 * A + 2 * i + 1] = evel[2 * tid + 1];
                es[4 * A + 2 * i] = elm[2 * tid]; es[4 * A + 2 * i + 1] = elm[2 * tid + 1];
            }
        }
    }
    PH_FLUSH;
}

// Six-wave form of the fused GRU rollout (round 3; the role split of cm_rollout.hip's six-wave kernels): waves 0-3 run the chain
// obs tile -> gru2_step -> head -> sample + physics exactly as k_gru32_rollout does; everything that only CONSUMES a step moves off it:
//   * wave 4 (writer) copies the step's obs tile to the rollout buffer (obs rows while the chain is in its first products, state rows
//     during the x-products; the tile is stable from the barrier after the obs build to the top of the next step);
//   * wave 5 (scorer) takes the reward partials + reward store of step t - 1 and the Philox uniforms of step t (LDS: uscr).
// In the four-wave kernel those were 31 % of the step (phase profile: reward partials + obs build 2.7 k, buffer writes + Philox 4.3 k of
// 22.5 k cycles).  Every wave passes the same four barriers per step (two of them inside gru2_step); the helpers' barriers order LDS
// traffic only (lds_barrier), so their global stores stay in flight across steps.  Same arithmetic in the same order: bit-identical
// buffers to k_gru32_rollout (option rollout_tile = 16 / 64 keeps the four-wave kernel; tests/test_hip_parity.py compares the two).
constexpr int NT6R = 6 * 64;
constexpr int G32R6_LDS_FLOATS = G32R_LDS_FLOATS + T32;  // + uscr

__global__ __launch_bounds__(NT6R, 1) void k_gru32_rollout6(const GruRollArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KJ = 2, KP = 8;
    const GruOff off = gru_offsets(a.din, a.H, a.K);
    float* p = smem;
    float* XA = p; p += T32 * LDT;   // obs tile
    float* X1 = p; p += T32 * LDT;   // x1
    float* hp = p; p += T32 * LDT;
    float* hn = p; p += T32 * LDT;
    GruLds L = {};
    L.wouts = p; p += KP * HP;
    L.b2 = p; p += KMAX;
    L.red = p; p += 64;
    float* ls = p; p += T32 * 8;
    float* epos = p; p += T32 * 2;
    float* evel = p; p += T32 * 2;
    float* elm = p; p += T32 * 2;
    float* rscr = p; p += 2 * T32;
    long* obase = reinterpret_cast<long*>(p);  // 8-byte aligned: every carve above is a multiple of 2 floats
    long* sbase = obase + T32;
    float* uscr = reinterpret_cast<float*>(sbase + T32);  // [T32] the step's uniforms (scorer -> sampler)
    const int tid = threadIdx.x;
    // the role is wave-uniform and the compiler must know it (scalar branches: no helper registers live through the chain)
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool compute = role < 4;
    const int hl = tid & 63;                   // lane of a helper wave
    const int hrow = tid >> 2, hq = tid & 3;   // compute waves: four lanes per row
    const int A = a.A, T = a.T, K = a.K, H = a.H, din = a.din;
    const int EPT = T32 / A, RT = EPT * A; const long Ds = a.state_ld;  // row stride of the state buffer
    G2W w;  // the seven weight blocks: this wave's 16 hidden columns in registers for the whole episode (cm_gru_step2.h)
    if (compute) {
        g2_load_weights<false>(w, a.params, off, din, H);
        for (int i = tid; i < KP * HP; i += NTHREADS) {
            const int k = i / HP, c = i % HP;
            L.wouts[i] = (c < H && k < K) ? a.params[off.W2 + k * H + c] : 0.0f;
        }
        for (int i = tid; i < KMAX; i += NTHREADS) L.b2[i] = (i < K) ? a.params[off.b2 + i] : 0.0f;
    }
    PH_DECL
    const int ntiles = (a.E + EPT - 1) / EPT;
    const float inv_din = 1.0f / (float)din, inv_sw = 1.0f / (float)(6 * A);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int e0 = tile * EPT;
        __syncthreads();
        // ---- reset (cm_env.hip k_env_reset): thread per (env, agent) row; h = 0 at the start of the episode
        if (tid < T32) {
            const int el = tid / A, i = tid - el * A;
            const long e = e0 + el;
            const bool live = tid < RT && e < a.E;
            obase[tid] = live ? (e * A + i) * (long)T * din : -1;
            sbase[tid] = live ? e * (long)T * Ds + (long)i * 6 * A : -1;
            if (live) {
                const unsigned long long ge = (unsigned long long)(a.env_offset + e);
                const cm_u4 ra = cm_philox4x32((uint32_t)ge, (uint32_t)a.episode, (uint32_t)i, CM_STREAM_ENV_RESET,
                                               (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
                epos[2 * tid] = 2.0f * cm_u01(ra.x) - 1.0f; epos[2 * tid + 1] = 2.0f * cm_u01(ra.y) - 1.0f;
                elm[2 * tid] = 2.0f * cm_u01(ra.z) - 1.0f; elm[2 * tid + 1] = 2.0f * cm_u01(ra.w) - 1.0f;
            } else {
                epos[2 * tid] = epos[2 * tid + 1] = 0.0f; elm[2 * tid] = elm[2 * tid + 1] = 0.0f;
            }
            evel[2 * tid] = 0.0f; evel[2 * tid + 1] = 0.0f;
        }
        if (compute)  // the chain's waves only: hp / hn swap every step in THEIR loop, a helper's copy of the pointers does not
            for (int i = tid; i < T32 * LDT; i += NTHREADS) hp[i] = 0.0f;
        // scorer: nearest-agent distance per landmark, collisions per agent (current positions), then the env's reward of step ts.
        // One wave: its LDS writes are performed in order before its later reads (no barrier needed, the wave runs in lockstep)
        auto score = [&](int ts) {
            if (hl < RT) {
                const int el = hl / A, l = hl - el * A;
                const float* pos = epos + el * 2 * A;
                const float lx = elm[2 * hl], ly = elm[2 * hl + 1];
                const float qx = pos[2 * l], qy = pos[2 * l + 1];
                float best = 3.0e38f, col = 0.0f;
                for (int j = 0; j < A; ++j) {
                    const float dx = pos[2 * j] - lx, dy = pos[2 * j + 1] - ly;
                    best = fminf(best, __builtin_amdgcn_sqrtf(dx * dx + dy * dy));
                    if (j > l) {
                        const float cx = qx - pos[2 * j], cy = qy - pos[2 * j + 1];
                        if (__builtin_amdgcn_sqrtf(cx * cx + cy * cy) < GR_COLLIDE) col += 1.0f;
                    }
                }
                rscr[hl] = best; rscr[T32 + hl] = col;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // compiler fence + the partials are in LDS
            __builtin_amdgcn_wave_barrier();
            if (hl < EPT && e0 + hl < a.E) {
                float r = 0.0f;
                for (int l = 0; l < A; ++l) r -= rscr[hl * A + l];
                for (int l = 0; l < A; ++l) r -= rscr[T32 + hl * A + l];
                a.reward[(long)(e0 + hl) * T + ts] = r;
            }
        };
        // three step loops, one per role (separate live ranges: nothing of a helper's state occupies registers of the chain);
        // every wave passes the same four barriers per step
        if (compute) {
            for (int t = 0; t < T; ++t) {
                lds_barrier();  // B0: positions of step t (sampler of t - 1), h_{t-1}; the writer is done reading the previous tile
                PH(0);
                // ---- observations of step t -> XA: 8 lanes per row on all four chain waves, lane oq handles entities j = oq, oq+8, ...
                // (the four-wave kernel builds them with 4 lanes per row on two waves; pure data movement: same values)
                {
                    const int orow = tid >> 3, oq = tid & 7;
                    const int el = orow / A, i = orow - el * A;
                    const bool live = orow < RT && (e0 + el) < a.E;
                    float* xr = XA + orow * LDT;
                    if (live) {
                        const float* pos = epos + el * 2 * A; const float* vel = evel + el * 2 * A; const float* lm = elm + el * 2 * A;
                        const float px = pos[2 * i], py = pos[2 * i + 1];
                        if (oq == 7) { xr[0] = vel[2 * i]; xr[1] = vel[2 * i + 1]; xr[2] = px; xr[3] = py; }  // the lane with the fewest entities
                        for (int j = oq; j < A; j += 8) {
                            xr[4 + 2 * j] = lm[2 * j] - px; xr[5 + 2 * j] = lm[2 * j + 1] - py;
                            if (j != i) {
                                const int jj = j < i ? j : j - 1;
                                xr[4 + 2 * A + 2 * jj] = pos[2 * j] - px; xr[5 + 2 * A + 2 * jj] = pos[2 * j + 1] - py;
                                xr[2 + 4 * A + 2 * jj] = 0.0f; xr[3 + 4 * A + 2 * jj] = 0.0f;  // comm channel
                            }
                            if (a.agent_ids) xr[6 * A + j] = (j == i) ? 1.0f : 0.0f;
                        }
                        for (int c = din + oq; c < KC; c += 8) xr[c] = 0.0f;
                    } else {
#pragma unroll
                        for (int j = 0; j < KC / 8; ++j) xr[8 * j + oq] = 0.0f;
                    }
                }
                PH(1);
                lds_barrier();  // B1: obs tile complete
                PH(2);
                gru2_step<false>(w, XA, X1, hp, hn, nullptr, nullptr, nullptr, nullptr, din, H);  // two barriers inside
                PH(4);
                if (hrow < T32) {
                    // head + sampler on the four lanes of a row, straight from registers (quad_categorical_sample: the operation order of
                    // cm_categorical_sample, two exponentials per lane instead of eight on one) -- no logits tile, no barrier before the sampler
                    unsigned char avb[KJ] = {1, 1};
                    float zreg[KJ];
                    gru_head_logits<KJ>(L, hn, K, avb, zreg);
                    PH(5);
                    int chosen; float lpv;
                    quad_categorical_sample<KJ>(zreg, K, hq, uscr[hrow], &chosen, &lpv);
                    const int el = hrow / A, i = hrow - el * A;
                    const long e = e0 + el;
                    if (hq == 0 && hrow < RT && e < a.E) {
                        const long o = (e * A + i) * (long)T + t;
                        a.action[o] = chosen;
                        a.logp[o] = lpv;
                        const float ux = (chosen == 1) ? -GR_ACCEL : (chosen == 2 ? GR_ACCEL : 0.0f);
                        const float uy = (chosen == 3) ? -GR_ACCEL : (chosen == 4 ? GR_ACCEL : 0.0f);
                        const float vx = evel[2 * hrow] * (1.0f - GR_DAMP) + ux * GR_DT;
                        const float vy = evel[2 * hrow + 1] * (1.0f - GR_DAMP) + uy * GR_DT;
                        evel[2 * hrow] = vx; evel[2 * hrow + 1] = vy;
                        epos[2 * hrow] += vx * GR_DT; epos[2 * hrow + 1] += vy * GR_DT;
                    }
                }
                PH(7);
                float* tmp = hp; hp = hn; hn = tmp;
            }
        } else if (role == 4) {
            // writer geometry of this tile: row 16 * pass + (lane >> 2), columns (lane & 3) + 4k
            long wob[2], wsb[2];
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int r = 16 * ps + (hl >> 2), el = r / A, i = r - el * A;
                const long e = e0 + el;
                const bool live = r < RT && e < a.E;
                wob[ps] = live ? (e * A + i) * (long)T * din : -1;
                wsb[ps] = live ? e * (long)T * Ds + (long)i * 6 * A : -1;
            }
            auto write_rows = [&](int ps, int t) {
                if (wob[ps] < 0) return;
                const int q = hl & 3;
                const float* xs = XA + (16 * ps + (hl >> 2)) * LDT + q;
                float* od = a.obs + wob[ps] + (long)t * din + q;
                float* sd = a.state + wsb[ps] + (long)t * Ds + q;
                float v[KC / 4];
#pragma unroll
                for (int k = 0; k < KC / 4; ++k) v[k] = xs[4 * k];
#pragma unroll
                for (int k = 0; k < KC / 4; ++k) {
                    if (q + 4 * k < din) od[4 * k] = v[k];
                    if (q + 4 * k < 6 * A) sd[4 * k] = v[k];
                }
            };
            for (int t = 0; t < T; ++t) {
                lds_barrier();  // B0: positions of step t (sampler of t - 1), h_{t-1}; the writer is done reading the previous tile
                // ---- writer: four lanes per row (lane q of a row: columns q, q + 4, ...), 16 rows per pass -- static LDS / global offsets,
                // no index math and no dependent LDS read on the way (the flat enumeration of the four-wave kernel cost this single wave
                // ~5 k cycles per step and made it the last arrival at the chain's barriers).  The first 6A columns of an obs row are
                // also the agent's block of the env's state row: one LDS read feeds both stores.
                lds_barrier();  // B1
                lds_barrier();  // x1 complete (inside the chain's gru2_step)
                write_rows(0, t);  // under the x-products
                lds_barrier();  // h' complete
                write_rows(1, t);  // under the head + sampler (two active waves, no products): the tile is rebuilt only after the next B0
            }
        } else {
            for (int t = 0; t < T; ++t) {
                lds_barrier();  // B0: positions of step t (sampler of t - 1), h_{t-1}; the writer is done reading the previous tile
                // ---- scorer: the step's uniforms under the chain's first products (the sampler reads them after gru2_step), the reward of step
                // t - 1 (positions after its physics update: stable until this step's sampler) under the x-products
                lds_barrier();  // B1 (nothing before it: the obs build is the shortest stretch of the chain)
                if (hl < RT) {
                    const int el = hl / A, i = hl - el * A;
                    const unsigned long long gr = (unsigned long long)((a.env_offset + e0 + el) * A + i);
                    const cm_u4 rnd = cm_philox4x32((uint32_t)gr, (uint32_t)(gr >> 32), (uint32_t)t, CM_STREAM_ACT,
                                                    (uint32_t)a.act_seed, (uint32_t)(a.act_seed >> 32));
                    uscr[hl] = cm_u01(rnd.x);
                }
                lds_barrier();  // x1 complete
                if (t > 0) score(t - 1);
                lds_barrier();  // h' complete
            }
        }
        __syncthreads();
        if (role == 5) score(T - 1);
        if (tid < RT) {
            const int el = tid / A, i = tid - el * A;
            const long e = e0 + el;
            if (e < a.E) {
                float* es = a.env_state + e * 6 * A;
                es[2 * i] = epos[2 * tid]; es[2 * i + 1] = epos[2 * tid + 1];
                es[2 * A + 2 * i] = evel[2 * tid]; es[2 * A + 2 * i + 1] = evel[2 * tid + 1];
                es[4 * A + 2 * i] = elm[2 * tid]; es[4 * A + 2 * i + 1] = elm[2 * tid + 1];
            }
        }
    }
    PH_FLUSH;
}

template <int KJ, bool WV>
__global__ __launch_bounds__(NTHREADS) void k_gru_chunk_bwd(const GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KP = KJ * 4;
    const GruOff off = gru_offsets(a.din, a.H, a.K);
    const GruLds L = gru_lds(smem, KP);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, lc = lane & 31;
    const int hrow = tid >> 2, hq = tid & 3;
    const int H = a.H, K = a.K, din = a.din, T = a.T, CL = a.t1 - a.t0;
    const long R = (long)a.E * a.A;
    const int col = 32 * wn + lc;
    gru_stage_consts(L, a, off, KP);

    f32x16 accW1, accWih[3], accWhh[3], accWo;
    float db1 = 0.f, dbg[4] = {0.f, 0.f, 0.f, 0.f}, dbo = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        accW1[g] = 0.f; accWo[g] = 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q) { accWih[q][g] = 0.f; accWhh[q][g] = 0.f; }
    }
    const long ntiles = (R + TM - 1) / TM;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = tile * TM;
        // ================================ backward through the chunk
        float *DH = L.b[0], *W = L.b[1], *A1 = L.b[2], *HPV = L.b[3], *G0 = L.b[4], *G1 = L.b[5], *G2 = L.b[6], *G3 = L.b[7];
        __syncthreads();
        for (int i = tid; i < TM * LDT; i += NTHREADS) DH[i] = 0.0f;
        for (int s = CL - 1; s >= 0; --s) {
            const int t = a.t0 + s;
            const float* wsS = a.ws_act + (s * R + row0) * WS_ACT;
            // ---- B1: head backward. ls <- dlogits[s], G3 <- relu(h'_s)
            __syncthreads();
#pragma unroll 1
            for (int i = tid; i < TM * KP; i += NTHREADS) {
                const int r = i / KP, k = i - r * KP;
                L.ls[r * LSP + k] = (row0 + r < R && k < K) ? a.ws_dl[(s * R + row0 + r) * WS_DL + k] : 0.0f;
            }
#pragma unroll 4
            for (int i = tid; i < TM * HP; i += NTHREADS) {
                const int r = i >> 6, c = i & 63;
                G3[r * LDT + c] = (row0 + r < R) ? fmaxf(wsS[(long)r * WS_ACT + 5 * HP + c], 0.0f) : 0.0f;
            }
            __syncthreads();
            colred_head(accWo, L.ls + 32 * wm * LSP, G3 + 32 * wm * LDT + 32 * wn);
            {
                const int k = tid & 31, part = tid >> 5;
                float sb = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) sb += L.ls[(part * 8 + r) * LSP + k];
                dbo += sb;
            }
            {   // DH += (dlogits * W2) .* (h' > 0)
                float dz[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) dz[i] = 0.f;
                for (int k0 = 0; k0 < K; k0 += 4) {
                    const float4 d4 = *reinterpret_cast<const float4*>(L.ls + hrow * LSP + k0);
                    const float dk[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4* wp4 = reinterpret_cast<const float4*>(L.wouts + (k0 + q) * HP + 16 * hq);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float4 w4 = wp4[i];
                            dz[4 * i] = fmaf(dk[q], w4.x, dz[4 * i]); dz[4 * i + 1] = fmaf(dk[q], w4.y, dz[4 * i + 1]);
                            dz[4 * i + 2] = fmaf(dk[q], w4.z, dz[4 * i + 2]); dz[4 * i + 3] = fmaf(dk[q], w4.w, dz[4 * i + 3]);
                        }
                    }
                }
                float* dp = DH + hrow * LDT + 16 * hq;
                const float* gp = G3 + hrow * LDT + 16 * hq;
#pragma unroll
                for (int i = 0; i < 16; ++i) if (gp[i] > 0.0f) dp[i] += dz[i];
            }
            __syncthreads();
            // ---- B2: gate derivatives (elementwise, flat mapping), h_prev -> HPV, x1 -> A1
#pragma unroll 2
            for (int i = tid; i < TM * HP; i += NTHREADS) {
                const int r = i >> 6, c = i & 63;
                float rr = 0.f, zz = 0.f, nn = 0.f, ghn = 0.f, hprev = 0.f, x1 = 0.f;
                if (row0 + r < R && c < H) {
                    const float* w = wsS + (long)r * WS_ACT;
                    x1 = w[c]; rr = w[HP + c]; zz = w[2 * HP + c]; nn = w[3 * HP + c]; ghn = w[4 * HP + c];
                    if (s > 0) hprev = a.ws_act[((s - 1) * R + row0 + r) * WS_ACT + 5 * HP + c];
                    else if (a.h_in) hprev = a.h_in[(row0 + r) * H + c];
                }
                const float dh = DH[r * LDT + c];
                const float dn = dh * (1.0f - zz), dzg = dh * (hprev - nn);
                const float dn_pre = dn * (1.0f - nn * nn);
                const float dr_pre = dn_pre * ghn * rr * (1.0f - rr);
                const float dz_pre = dzg * zz * (1.0f - zz);
                G0[r * LDT + c] = dr_pre; G1[r * LDT + c] = dz_pre; G2[r * LDT + c] = dn_pre; G3[r * LDT + c] = dn_pre * rr;
                HPV[r * LDT + c] = hprev; A1[r * LDT + c] = x1;
                DH[r * LDT + c] = dh * zz;
            }
            __syncthreads();
            // ---- B3: weight gradients of the gates (the first weight block of B4 is requested now, under these MFMAs)
            Tile16 tw;
            gate_load<WV>(tw, a.params + off.Wih, H);
            colred(accWih[0], G0 + 32 * wm, A1 + 32 * wn);
            colred(accWih[1], G1 + 32 * wm, A1 + 32 * wn);
            colred(accWih[2], G2 + 32 * wm, A1 + 32 * wn);
            colred(accWhh[0], G0 + 32 * wm, HPV + 32 * wn);
            colred(accWhh[1], G1 + 32 * wm, HPV + 32 * wn);
            colred(accWhh[2], G3 + 32 * wm, HPV + 32 * wn);
            {
                const int c = tid & 63, part = tid >> 6;
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
                for (int r = 0; r < TM / 4; ++r) {
                    const int o = (part * (TM / 4) + r) * LDT + c;
                    s0 += G0[o]; s1 += G1[o]; s2 += G2[o]; s3 += G3[o];
                }
                dbg[0] += s0; dbg[1] += s1; dbg[2] += s2; dbg[3] += s3;
            }
            // ---- B4: dx1 = sum_g dgi_g * W_ih[g]   (.* relu'(x1), in place over A1)
            f32x16 acc;
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[g] = 0.f;
#pragma unroll
            for (int gate = 0; gate < 3; ++gate) {
                __syncthreads();
                tile_store<WV>(W, tw);
                gate_load<WV>(tw, a.params + (gate < 2 ? off.Wih + (gate + 1) * H * H : off.Whh), H);
                __syncthreads();
                rowpar_tn(acc, (gate == 0 ? G0 : gate == 1 ? G1 : G2) + 32 * wm * LDT, W + 32 * wn);
            }
            // ---- B5: dh_prev += sum_g dgh_g * W_hh[g]
            f32x16 acch;
#pragma unroll
            for (int g = 0; g < 16; ++g) acch[g] = 0.f;
#pragma unroll
            for (int gate = 0; gate < 3; ++gate) {
                __syncthreads();
                tile_store<WV>(W, tw);
                if (gate < 2) gate_load<WV>(tw, a.params + off.Whh + (gate + 1) * H * H, H);
                else x_load(tw, a.obs + (long)t * din, row0, R, (long)T * din, din);  // obs tile for B6
                __syncthreads();
                rowpar_tn(acch, (gate == 0 ? G0 : gate == 1 ? G1 : G3) + 32 * wm * LDT, W + 32 * wn);
            }
            __syncthreads();  // every wave is done with A1 (B3) and G* (B4/B5)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int row = 32 * wm + (g & 3) + 8 * (g >> 2) + 4 * h;
                float* p = A1 + row * LDT + col;
                *p = (*p > 0.0f) ? acc[g] : 0.0f;
                DH[row * LDT + col] += acch[g];
            }
            // ---- B6: fc1 weight gradient: obs tile -> G0
            tile_store<false>(G0, tw);
            __syncthreads();
            colred(accW1, A1 + 32 * wm, G0 + 32 * wn);
            {
                const int c = tid & 63, part = tid >> 6;
                float s0 = 0.f;
#pragma unroll
                for (int r = 0; r < TM / 4; ++r) s0 += A1[(part * (TM / 4) + r) * LDT + c];
                db1 += s0;
            }
        }
    }
    // ================================ partial gradient + stats of this workgroup
    float* out = a.partial + (size_t)blockIdx.x * a.PS;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const int n = 32 * wm + (g & 3) + 8 * (g >> 2) + 4 * h;
        if (n < H && col < din) out[off.W1 + n * din + col] = accW1[g];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (n < H && col < H) {
                out[off.Wih + (q * H + n) * H + col] = accWih[q][g];
                out[off.Whh + (q * H + n) * H + col] = accWhh[q][g];
            }
        }
    }
    {   // fc2 weight: combine the two wave-rows through LDS (activations are dead)
        float* scr = L.b[0];
        __syncthreads();
        if (wm == 1) {
#pragma unroll
            for (int g = 0; g < 16; ++g) scr[(wn * 32 + (g & 3) + 8 * (g >> 2) + 4 * h) * 33 + lc] = accWo[g];
        }
        __syncthreads();
        if (wm == 0) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int k = (g & 3) + 8 * (g >> 2) + 4 * h;
                if (k < K && col < H) out[off.W2 + k * H + col] = accWo[g] + scr[(wn * 32 + k) * 33 + lc];
            }
        }
        __syncthreads();
        L.red[tid] = dbo;
        __syncthreads();
        if (tid < K) {
            float sb = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) sb += L.red[q * 32 + tid];
            out[off.b2 + tid] = sb;
        }
    }
    {   // column-sum biases: 4 row parts per column
        float vals[5] = {db1, dbg[0], dbg[1], dbg[2], dbg[3]};
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            __syncthreads();
            L.red[(tid >> 6) * HP + (tid & 63)] = vals[q];
            __syncthreads();
            if (tid < H) {
                const float sv = L.red[tid] + L.red[HP + tid] + L.red[2 * HP + tid] + L.red[3 * HP + tid];
                if (q == 0) out[off.b1 + tid] = sv;
                else if (q == 1) { out[off.bih + tid] = sv; out[off.bhh + tid] = sv; }                // r gate: d b_ir == d b_hr
                else if (q == 2) { out[off.bih + H + tid] = sv; out[off.bhh + H + tid] = sv; }        // z gate
                else if (q == 3) out[off.bih + 2 * H + tid] = sv;                                      // b_in
                else out[off.bhh + 2 * H + tid] = sv;                                                  // b_hn (scaled by r)
            }
        }
    }
}

// ============================================================================================ rollout step
template <int KJ, bool WV>
__global__ __launch_bounds__(NTHREADS) void k_gru_act(const GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KP = KJ * 4;
    const GruOff off = gru_offsets(a.din, a.H, a.K);
    const GruLds L = gru_lds(smem, KP);
    const int tid = threadIdx.x, hrow = tid >> 2, hq = tid & 3;
    const int H = a.H, K = a.K;
    gru_stage_consts(L, a, off, KP);
    const long ntiles = (a.rows + TM - 1) / TM;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = tile * TM;
        float* hp = L.b[3];
        float* hn = L.b[4];
        __syncthreads();
        for (int i = tid; i < TM * HP; i += NTHREADS) {
            const int r = i >> 6, c = i & 63;
            hp[r * LDT + c] = (row0 + r < a.rows && c < H) ? a.h[(row0 + r) * H + c] : 0.0f;
        }
        Tile16 tA, tB;
        x_load(tA, a.x, row0, a.rows, a.x_stride, a.din);
        x_load(tB, a.params + off.W1, 0, H, a.din, a.din);
        gru_fwd_step<false, WV>(L, a, off, tA, tB, nullptr, a.x_stride, row0, a.rows, hp, hn, nullptr);
        for (int i = tid; i < TM * HP; i += NTHREADS) {
            const int r = i >> 6, c = i & 63;
            if (row0 + r < a.rows && c < H) a.h[(row0 + r) * H + c] = hn[r * LDT + c];
        }
        const long grow = row0 + hrow;
        const bool rvalid = grow < a.rows;
        unsigned char avb[KJ];
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            avb[j] = 1;
            if (rvalid && a.av && 4 * j + hq < K) avb[j] = a.av[grow * a.av_stride + 4 * j + hq];
        }
        float zreg[KJ];
        gru_head_logits<KJ>(L, hn, K, avb, zreg);
#pragma unroll
        for (int j = 0; j < KJ; ++j)
            if (4 * j + hq < K) L.ls[hrow * LSP + 4 * j + hq] = zreg[j];
        __syncthreads();
        if (hq == 0 && rvalid) {  // same sampler as k_mlp<M_ACT>
            const unsigned long long gr = (unsigned long long)(a.row_offset + grow);
            const cm_u4 rnd = cm_philox4x32((uint32_t)gr, (uint32_t)(gr >> 32), (uint32_t)a.t, CM_STREAM_ACT,
                                            (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
            int chosen; float lp;
            cm_categorical_sample(L.ls + hrow * LSP, K, cm_u01(rnd.x), &chosen, &lp);
            a.action_out[grow * a.out_stride] = chosen;
            a.logp_out[grow * a.out_stride] = lp;
        }
    }
}

int gru_check(const char* who, int din, int H, int K) {
    CM_REQUIRE(din > 0 && H > 0 && K > 0, "%s: bad dims din=%d H=%d K=%d", who, din, H, K);
    CM_REQUIRE(H <= HP, "%s: hidden_dim=%d > %d needs the layered schedule (workspace-taking entry points)", who, H, HP);
    CM_REQUIRE(din <= KC, "%s: obs width %d > %d needs the layered schedule (workspace-taking entry points)", who, din, KC);
    CM_REQUIRE(K <= KMAX, "%s: n_actions=%d > %d is not supported by this build", who, K, KMAX);
    return 0;
}

#include "cm_gru_v2.h"

}  // namespace

static size_t gru_ps(int din, int hidden, int K) {
    return (size_t)((cm_gru_param_count(din, hidden, K) + CM_NUM_STATS + 63) / 64 * 64);
}

static bool gru_wide(int din, int hidden, int K = 1) { return din > KC || hidden > HP || K > KMAX; }  // layered schedule (cm_gru_wide.hip)

extern "C" size_t cm_gru_workspace_bytes(int E, int A, int din, int hidden, int n_actions, int chunk_len) {
    const size_t R = (size_t)E * A;
    if (gru_wide(din, hidden, n_actions)) return cm_gru_wide_ws_bytes((int64_t)R, chunk_len, din, hidden, n_actions, 1);
    // per (step, row): the larger of the two activation formats (v2: 7 slots, cm_gru_v2.h) + the first generation's dlogits
    // + the pipelined sweeps' hand-over areas: step flags of k_gru2_fwdx (four 8-byte words per 32-row tile), {dh_t, tag} words of k_gru2_bwd<true>
    // + the W_ih accumulators of the split forward sweep (k_gru2_pre -> k_gru2_fwdx<.., PRE>: GI_UNIT floats per 32-row tile and step)
    return ((size_t)chunk_len * R * (WS2 + WS_DL) + (size_t)MAX_GRID * gru_ps(din, hidden, n_actions)) * sizeof(float) + 8 + (size_t)MAX_GRID * 4 * 8 + (size_t)chunk_len * R * HP * 8
           + 16 + (size_t)chunk_len * ((R + T32 - 1) / T32) * GI_UNIT * sizeof(float);
}

static int gru_device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
        else cus = 1;
    }
    return cus;
}
static std::atomic<unsigned> g_gru_tag{1};

static int gru_chunk_pass(const float* obs, const uint8_t* avail, const int32_t* action,
                          const float* logp_old, const float* adv, const int32_t* ep_len,
                          int E, int A, int T, int t0, int t1, int din, int hidden, int n_actions,
                          const float* params, const float* h_in, float* h_out,
                          double ppo_clip, double entropy_coef,
                          float* grad_and_stats, void* ws, size_t ws_bytes, cm_stream_t stream, const cm_opt_step_t* opt) {
    CM_REQUIRE(E > 0 && A > 0 && T > 0 && 0 <= t0 && t0 < t1 && t1 <= T, "cm_gru_actor_chunk_fwd_bwd: bad dims E=%d A=%d T=%d t0=%d t1=%d", E, A, T, t0, t1);
    if (gru_wide(din, hidden, n_actions))
        return cm_gru_wide_chunk(obs, avail, action, logp_old, adv, ep_len, E, A, T, t0, t1, din, hidden, n_actions, params, h_in, h_out, ppo_clip,
                                 entropy_coef, grad_and_stats, ws, ws_bytes, (hipStream_t)stream, opt);
    if (int rc = gru_check("cm_gru_actor_chunk_fwd_bwd", din, hidden, n_actions)) return rc;
    const size_t need = cm_gru_workspace_bytes(E, A, din, hidden, n_actions, t1 - t0);
    CM_REQUIRE(ws && ws_bytes >= need, "cm_gru_actor_chunk_fwd_bwd: workspace too small (%zu < %zu)", ws_bytes, need);
    const size_t R = (size_t)E * A;
    const int CL = t1 - t0;
    GruArgs a = {};
    a.obs = obs; a.avail = avail; a.action = action; a.logp_old = logp_old; a.adv = adv; a.ep_len = ep_len;
    a.E = E; a.A = A; a.T = T; a.t0 = t0; a.t1 = t1; a.din = din; a.H = hidden; a.K = n_actions;
    a.params = params; a.h_in = h_in; a.h_out = h_out;
    a.clip_lo = (float)(1.0 - ppo_clip); a.clip_hi = (float)(1.0 + ppo_clip); a.clip_eps = (float)ppo_clip; a.ent_coef = (float)entropy_coef;
    a.ws_act = (float*)ws; a.ws_dl = a.ws_act + (size_t)CL * R * WS2; a.partial = a.ws_dl + (size_t)CL * R * WS_DL;
    a.PS = (int)gru_ps(din, hidden, n_actions);
#ifdef CM_PHASE_PROF
    a.prof = g_prof;
#endif
    const int grid = grid_for((long)R);
    const size_t lds = gru_lds_bytes(n_actions);
    const bool wv = gru_wvec(params, hidden);
#define CM_GRU_LAUNCH2(KJ_, WV_) do { \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_chunk_fwd<KJ_, WV_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_chunk_bwd<KJ_, WV_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_gru_chunk_fwd<KJ_, WV_>), dim3(grid), dim3(NTHREADS), lds, (hipStream_t)stream, a); \
        hipLaunchKernelGGL((k_gru_chunk_bwd<KJ_, WV_>), dim3(grid), dim3(NTHREADS), lds, (hipStream_t)stream, a); } while (0)
    // small / medium batch (<= 512 64-row tiles): the 32-row sweeps of cm_gru_v2.h (weights in registers, head outside the recurrence,
    // one partial row per 32-row tile) -- they pay off while the 64-row tiling leaves CUs idle or barely filled (measured: 5k and 20k
    // sequences faster, 82k sequences slower); above, or with cm_set_option("gru_tile", "64"), the 64-row streaming kernels (same
    // workspace format).  Both are pinned to the reference goldens (tests/test_hip_parity.py).
    const bool force64 = cm_option(CM_OPTION_GRU_TILE) == 64;
    const int64_t P = cm_gru_param_count(din, hidden, n_actions);
    MlpArgs m = {};
    m.partial = a.partial; m.PS = a.PS;
    if (!force64 && din <= KC && (long)R <= 512L * TM) {
        const long nt32 = ((long)R + T32 - 1) / T32;
        const int grid32 = (int)(nt32 < MAX_GRID ? nt32 : MAX_GRID);
        const int KP = n_actions <= 16 ? 16 : 32;
        const size_t lf = gru2_fwd_lds_bytes(KP), lb = gru2_bwd_lds_bytes();
#define CM_GRU2_F(WV_, KP_) do { \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru2_fwd<WV_, KP_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lf); \
        hipLaunchKernelGGL((k_gru2_fwd<WV_, KP_>), dim3(grid32), dim3(NTHREADS), lf, (hipStream_t)stream, a); } while (0)
        // forward: eight waves per tile (stores + head on helper waves, behind the chain); gru_tile = 32 keeps the four-wave kernel
        const size_t lf8 = gru2_fwd8_lds_bytes(KP);
#define CM_GRU2_F8(WV_, KP_) do { \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru2_fwd8<WV_, KP_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lf8); \
        hipLaunchKernelGGL((k_gru2_fwd8<WV_, KP_>), dim3(grid32), dim3(NT8), lf8, (hipStream_t)stream, a); } while (0)
        // ... and, while the tiles leave a third of the CUs idle, the head on those CUs (k_gru2_fwdx: one head workgroup per two tiles)
        const int tile_opt = cm_option(CM_OPTION_GRU_TILE);
        const int nh = (int)((nt32 + 1) / 2);
        // gru_tile = "split" (opt-in, round 6): the pipelined forward sweep split at its dependence on h -- fc1 and the W_ih products of the
        // whole chunk as one throughput launch (k_gru2_pre), the chain on W_hh h + gates with one barrier per step.  Bit-identical, and
        // SLOWER at config 5 (profiles/r06_gru_split_ab.txt: k_gru2_pre 23.6 us + chain launch 51 us against 66 us unsplit -- the chain's
        // ten steps take 26 us instead of 45, but the launch ends with the head workgroups, 80 of them serving 160 tiles, and the
        // throughput launch costs more than the chain saves): not the default
        // gru_tile = "nosplit": the unsplit pipelined sweeps of round 5 (A/B runs); auto: split inside the workgroup (k_gru2_fwdx<.., 2>)
        const bool pipelined = (tile_opt == 0 || tile_opt == 1 || tile_opt == 2) && nt32 <= MAX_GRID && nt32 + nh <= gru_device_cus();
        const bool split_fwd = pipelined && tile_opt == 1, split_in_wg = pipelined && tile_opt == 0;
        GruXArgs xa = {};
        xa.nt = (int)nt32; xa.nh = nh;
        {
            uintptr_t fp = reinterpret_cast<uintptr_t>(a.partial + (size_t)MAX_GRID * a.PS);
            xa.flags = reinterpret_cast<unsigned long long*>((fp + 7) & ~(uintptr_t)7);
        }
        const size_t lfx = gru2_fwdx_lds_bytes(KP);
#define CM_GRU2_FX(WV_, KP_) do { \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru2_fwdx<WV_, KP_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lfx); \
        hipLaunchKernelGGL((k_gru2_fwdx<WV_, KP_>), dim3(grid32 + nh), dim3(NT8), lfx, (hipStream_t)stream, a, xa); } while (0)
        // the split forward sweep: the h-independent products of all (tile, step) units on every CU first, then the chain on W_hh h + gates
#define CM_GRU2_FXP(WV_, KP_) do { \
        const long units_ = nt32 * CL; const int cus2_ = 2 * gru_device_cus(); \
        hipLaunchKernelGGL((k_gru2_pre<WV_>), dim3((unsigned)(units_ < cus2_ ? units_ : cus2_)), dim3(NTHREADS), 0, (hipStream_t)stream, a, xa); \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru2_fwdx<WV_, KP_, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lfx); \
        hipLaunchKernelGGL((k_gru2_fwdx<WV_, KP_, 1>), dim3(grid32 + nh), dim3(NT8), lfx, (hipStream_t)stream, a, xa); } while (0)
        // the same split inside the workgroup (helper waves run the h-independent half a step ahead): no extra launch
#define CM_GRU2_FXI(WV_, KP_) do { \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru2_fwdx<WV_, KP_, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lfx); \
        hipLaunchKernelGGL((k_gru2_fwdx<WV_, KP_, 2>), dim3(grid32 + nh), dim3(NT8), lfx, (hipStream_t)stream, a, xa); } while (0)
        if (tile_opt == 32) {
            if (KP == 16) { if (wv) CM_GRU2_F(true, 16); else CM_GRU2_F(false, 16); }
            else          { if (wv) CM_GRU2_F(true, 32); else CM_GRU2_F(false, 32); }
        } else if (split_fwd) {
            xa.tag = g_gru_tag.fetch_add(1, std::memory_order_relaxed);
            {   // behind the {dh_t, tag} words of the backward sweep, 16-byte aligned
                uintptr_t gp = reinterpret_cast<uintptr_t>(xa.flags + (size_t)MAX_GRID * 4 + (size_t)CL * R * HP);
                xa.gi = reinterpret_cast<float*>((gp + 15) & ~(uintptr_t)15);
            }
            if (KP == 16) { if (wv) CM_GRU2_FXP(true, 16); else CM_GRU2_FXP(false, 16); }
            else          { if (wv) CM_GRU2_FXP(true, 32); else CM_GRU2_FXP(false, 32); }
        } else if (split_in_wg) {
            xa.tag = g_gru_tag.fetch_add(1, std::memory_order_relaxed);
            if (KP == 16) { if (wv) CM_GRU2_FXI(true, 16); else CM_GRU2_FXI(false, 16); }
            else          { if (wv) CM_GRU2_FXI(true, 32); else CM_GRU2_FXI(false, 32); }
        } else if (pipelined) {
            xa.tag = g_gru_tag.fetch_add(1, std::memory_order_relaxed);
            if (KP == 16) { if (wv) CM_GRU2_FX(true, 16); else CM_GRU2_FX(false, 16); }
            else          { if (wv) CM_GRU2_FX(true, 32); else CM_GRU2_FX(false, 32); }
        } else {
            if (KP == 16) { if (wv) CM_GRU2_F8(true, 16); else CM_GRU2_F8(false, 16); }
            else          { if (wv) CM_GRU2_F8(true, 32); else CM_GRU2_F8(false, 32); }
        }
#undef CM_GRU2_F8
#undef CM_GRU2_FX
#undef CM_GRU2_FXP
#undef CM_GRU2_FXI
#undef CM_GRU2_F
        // backward: likewise, five of the seven weight-gradient products of a step on the idle CUs (k_gru2_bwd<true> / gru2_grad_wg: as many
        // workgroups as CUs are left, at most one per tile; their partial rows follow those of the tiles)
        int ng = 0;
        if (pipelined) {
            ng = gru_device_cus() - (int)nt32;
            if (ng > (int)nt32) ng = (int)nt32;
            if ((long)nt32 + ng > MAX_GRID) ng = MAX_GRID - (int)nt32;
        }
        if (ng > 0) {
            GruXArgs xb = xa;
            xb.dhq = xa.flags + (size_t)MAX_GRID * 4;
            xb.tag = g_gru_tag.fetch_add(1, std::memory_order_relaxed);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru2_bwd<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
            hipLaunchKernelGGL(k_gru2_bwd<true>, dim3(grid32 + ng), dim3(NTHREADS), lb, (hipStream_t)stream, a, xb);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru2_bwd<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
            hipLaunchKernelGGL(k_gru2_bwd<false>, dim3(grid32), dim3(NTHREADS), lb, (hipStream_t)stream, a, xa);
        }
        CM_CHECK_LAUNCH("cm_gru_actor_chunk_fwd_bwd");
        return finish_train(m, grid32 + ng, P, grad_and_stats, (hipStream_t)stream, "cm_gru_actor_chunk_fwd_bwd", 0, opt);
    }
    if (n_actions <= 8) { if (wv) CM_GRU_LAUNCH2(2, true); else CM_GRU_LAUNCH2(2, false); }
    else { if (wv) CM_GRU_LAUNCH2(8, true); else CM_GRU_LAUNCH2(8, false); }
#undef CM_GRU_LAUNCH2
    CM_CHECK_LAUNCH("cm_gru_actor_chunk_fwd_bwd");
    return finish_train(m, grid, P, grad_and_stats, (hipStream_t)stream, "cm_gru_actor_chunk_fwd_bwd", 0, opt);
}

extern "C" int cm_gru_actor_chunk_fwd_bwd(const float* obs, const uint8_t* avail, const int32_t* action,
                                          const float* logp_old, const float* adv, const int32_t* ep_len,
                                          int E, int A, int T, int t0, int t1, int din, int hidden, int n_actions,
                                          const float* params, const float* h_in, float* h_out,
                                          double ppo_clip, double entropy_coef,
                                          float* grad_and_stats, void* ws, size_t ws_bytes, cm_stream_t stream) {
    return gru_chunk_pass(obs, avail, action, logp_old, adv, ep_len, E, A, T, t0, t1, din, hidden, n_actions, params, h_in, h_out, ppo_clip,
                          entropy_coef, grad_and_stats, ws, ws_bytes, stream, nullptr);
}
extern "C" int cm_gru_actor_chunk_train_step(const float* obs, const uint8_t* avail, const int32_t* action,
                                             const float* logp_old, const float* adv, const int32_t* ep_len,
                                             int E, int A, int T, int t0, int t1, int din, int hidden, int n_actions,
                                             const float* h_in, float* h_out, double ppo_clip, double entropy_coef,
                                             float* grad_and_stats, void* ws, size_t ws_bytes, const cm_opt_step_t* opt, cm_stream_t stream) {
    CM_REQUIRE(opt && opt->params, "cm_gru_actor_chunk_train_step: cm_opt_step_t / params is NULL");
    return gru_chunk_pass(obs, avail, action, logp_old, adv, ep_len, E, A, T, t0, t1, din, hidden, n_actions, opt->params, h_in, h_out, ppo_clip,
                          entropy_coef, grad_and_stats, ws, ws_bytes, stream, opt);
}

extern "C" int cm_gru_policy_act(const float* x, int64_t x_row_stride, const uint8_t* avail, int64_t avail_row_stride,
                                 int64_t rows, int din, int hidden, int n_actions, const float* params, float* h,
                                 uint64_t seed, int64_t row_offset, int t,
                                 int32_t* action, float* logp, int64_t out_stride, cm_stream_t stream) {
    if (int rc = gru_check("cm_gru_policy_act", din, hidden, n_actions)) return rc;
    if (rows <= 0) return 0;
    CM_REQUIRE(h != nullptr, "cm_gru_policy_act: hidden state pointer is NULL");
    GruArgs a = {};
    a.x = x; a.x_stride = x_row_stride; a.av = avail; a.av_stride = avail_row_stride; a.rows = rows; a.din = din; a.H = hidden;
    a.K = n_actions; a.params = params; a.h = h; a.seed = seed; a.row_offset = row_offset; a.t = t;
    a.action_out = action; a.logp_out = logp; a.out_stride = out_stride;
    const size_t lds = gru_lds_bytes(n_actions);
    const int grid = grid_for(rows);
    const bool wv = gru_wvec(params, hidden);
#define CM_GRU_LAUNCH1(KJ_, WV_) do { \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_act<KJ_, WV_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_gru_act<KJ_, WV_>), dim3(grid), dim3(NTHREADS), lds, (hipStream_t)stream, a); } while (0)
    if (n_actions <= 8) { if (wv) CM_GRU_LAUNCH1(2, true); else CM_GRU_LAUNCH1(2, false); }
    else { if (wv) CM_GRU_LAUNCH1(8, true); else CM_GRU_LAUNCH1(8, false); }
#undef CM_GRU_LAUNCH1
    CM_CHECK_LAUNCH("cm_gru_policy_act");
    return 0;
}

/* the same step with a caller workspace: also serves the layered shapes, and eps < 0 takes the argmax (greedy evaluation) */
extern "C" size_t cm_gru_policy_act_workspace_bytes(int64_t rows, int din, int hidden, int n_actions) {
    return cm_gru_wide_ws_bytes(rows, 1, din, hidden, n_actions, 0);
}
extern "C" int cm_gru_policy_act_ws(const float* x, int64_t x_row_stride, const uint8_t* avail, int64_t avail_row_stride,
                                    int64_t rows, int din, int hidden, int n_actions, const float* params, float* h, double eps,
                                    uint64_t seed, int64_t row_offset, int t, int32_t* action, float* logp, int64_t out_stride,
                                    void* ws, size_t ws_bytes, cm_stream_t stream) {
    CM_REQUIRE(h != nullptr, "cm_gru_policy_act_ws: hidden state pointer is NULL");
    CM_REQUIRE(eps <= 0.0, "cm_gru_policy_act_ws: eps=%g (the recurrent scripts have no epsilon-mixed exploration; eps < 0 = greedy)", eps);
    if (!gru_wide(din, hidden, n_actions) && eps == 0.0)  // the fused step kernel
        return cm_gru_policy_act(x, x_row_stride, avail, avail_row_stride, rows, din, hidden, n_actions, params, h, seed, row_offset, t, action, logp,
                                 out_stride, stream);
    return cm_gru_wide_act(x, x_row_stride, avail, avail_row_stride, rows, din, hidden, n_actions, params, h, seed, row_offset, t, (float)eps, action,
                           logp, out_stride, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int cm_gru_rollout_spread_supported(int A, int agent_ids, int hidden) {
    const int din = 6 * A + (agent_ids ? A : 0);
    return (A >= 1 && A <= T32 && din <= KC && hidden <= HP) ? 1 : 0;
}

extern "C" int cm_gru_rollout_spread_ld(float* env_state, int E, int A, int T, int agent_ids, uint64_t seed, uint64_t act_seed,
                                        int64_t env_offset, int64_t episode, const float* params, int hidden,
                                        float* obs, float* state, int64_t state_ld, int32_t* action, float* logp, float* reward, cm_stream_t stream);
extern "C" int cm_gru_rollout_spread(float* env_state, int E, int A, int T, int agent_ids, uint64_t seed, uint64_t act_seed,
                                     int64_t env_offset, int64_t episode, const float* params, int hidden,
                                     float* obs, float* state, int32_t* action, float* logp, float* reward, cm_stream_t stream) {
    return cm_gru_rollout_spread_ld(env_state, E, A, T, agent_ids, seed, act_seed, env_offset, episode, params, hidden, obs, state, 6L * A * A, action,
                                    logp, reward, stream);
}
extern "C" int cm_gru_rollout_spread_ld(float* env_state, int E, int A, int T, int agent_ids, uint64_t seed, uint64_t act_seed,
                                        int64_t env_offset, int64_t episode, const float* params, int hidden,
                                        float* obs, float* state, int64_t state_ld, int32_t* action, float* logp, float* reward, cm_stream_t stream) {
    CM_REQUIRE(E > 0 && T > 0, "cm_gru_rollout_spread: bad dims E=%d T=%d", E, T);
    CM_REQUIRE(state_ld >= 6L * A * A, "cm_gru_rollout_spread_ld: state_ld=%ld below the state width %d", (long)state_ld, 6 * A * A);
    CM_REQUIRE(cm_gru_rollout_spread_supported(A, agent_ids, hidden),
               "cm_gru_rollout_spread: unsupported shape A=%d hidden=%d (use cm_gru_policy_act + cm_synth_env_step)", A, hidden);
    GruRollArgs a = {};
    a.env_state = env_state; a.E = E; a.A = A; a.T = T; a.agent_ids = agent_ids; a.seed = seed; a.act_seed = act_seed;
    a.env_offset = env_offset; a.episode = episode; a.params = params; a.din = 6 * A + (agent_ids ? A : 0); a.H = hidden; a.K = 5;
    a.obs = obs; a.state = state; a.action = action; a.logp = logp; a.reward = reward; a.state_ld = (long)state_ld;
#ifdef CM_PHASE_PROF
    a.prof = g_prof;
#endif
    const int EPT = T32 / A;
    const int ntiles = (E + EPT - 1) / EPT;
    const size_t lds = (size_t)G32R_LDS_FLOATS * sizeof(float);
    const int tiling = cm_option(CM_OPTION_ROLLOUT_TILE);  // 16 / 64: the four-wave kernel; auto / 16s / 64s: writer + scorer waves
    if (tiling == 16 || tiling == 64) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru32_rollout), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_gru32_rollout, dim3(ntiles < 256 ? ntiles : 256), dim3(NTHREADS), lds, (hipStream_t)stream, a);
    } else {
        const size_t lds6 = (size_t)G32R6_LDS_FLOATS * sizeof(float);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru32_rollout6), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds6);
        hipLaunchKernelGGL(k_gru32_rollout6, dim3(ntiles < 256 ? ntiles : 256), dim3(NT6R), lds6, (hipStream_t)stream, a);
    }
    CM_CHECK_LAUNCH("cm_gru_rollout_spread");
    return 0;
}
