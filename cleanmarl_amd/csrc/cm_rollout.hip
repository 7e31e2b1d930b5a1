// cm_rollout.hip -- fused, persistent rollout kernel for the on-device synthetic MPE-like env.
//
// Replaces the whole inner rollout loop of cleanmarl/mappo_multienvs.py:393-453 (reset, then per step:
// Actor.act :409-414 -> env.step over the pipes :415-424 -> append to the episode lists :425-434) and the
// collate copy of RolloutBuffer.get_batch (:109-157) by ONE launch: environments are independent, so a
// workgroup keeps a tile of floor(64/A) envs (<= 64 (env,agent) rows) resident in LDS and walks all T steps
//     obs from env state -> [write obs/state to the rollout buffer] -> actor MLP on the MFMA units ->
//     Categorical sample (Philox keyed by seed, global row, t) -> [write action/logp] -> point-mass physics
//     -> team reward -> next step
// with the actor weights LDS-stationary for the whole episode.  No inter-workgroup dependency, no host round
// trip, no per-step launch.  The per-step kernels (cm_policy_act + cm_synth_env_step) remain the C-ABI for
// real / host-side environments and are the parity reference for this kernel (tests/test_hip_parity.py).
#include "cm_mlp_kernel.h"
#ifdef CM_PHASE_PROF
extern unsigned long long* g_prof;
#endif

namespace {

// The rollout is a latency chain (a few waves per CU, each step a sequence of dependent LDS / MFMA / DPP operations) that the learner
// deliberately overlaps with the critic's epochs of the previous iteration (learner.overlap_critic): its waves are rarely ready, but
// when they are, every issue slot lost to a co-resident throughput kernel lengthens the chain (512-env share: 0.475 ms alone, 0.51 -
// 0.68 ms beside the critic's k_mlp).  Highest wave priority makes the SIMD's arbiter pick them first; CM_ROLLOUT_PRIO=0 builds without.
#ifndef CM_ROLLOUT_PRIO
#define CM_ROLLOUT_PRIO 3
#endif
#define ROLLOUT_WAVE_PRIO() __builtin_amdgcn_s_setprio(CM_ROLLOUT_PRIO)
#ifdef CM_PHASE_PROF  // the store wave's own slots 8..15 of the workgroup's profile row (tools/phase_prof.py rollout)
#define PH_FLUSH_SW do { if (a.prof && threadIdx.x == NTHREADS) { _Pragma("unroll") for (int i_ = 8; i_ < 14; ++i_) a.prof[(size_t)blockIdx.x * 16 + i_] = ph_[i_]; } } while (0)
#define PH_FLUSH_SC do { if (a.prof && threadIdx.x == NTHREADS + 64) { a.prof[(size_t)blockIdx.x * 16 + 14] = ph_[14]; a.prof[(size_t)blockIdx.x * 16 + 15] = ph_[15]; } } while (0)
#define PH_FLUSH_C do { if (a.prof && threadIdx.x == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) a.prof[(size_t)blockIdx.x * 16 + i_] = ph_[i_]; } } while (0)
#else
#define PH_FLUSH_SW
#define PH_FLUSH_SC
#define PH_FLUSH_C
#endif

constexpr float DAMP = 0.25f, DT = 0.1f, ACCEL = 5.0f, COLLIDE = 0.3f;

constexpr int SPREAD_K = 5;  // actions of the synthetic spread env (cm_env.hip: no-op + four accelerations); the launcher sets a.K to it
struct RolloutArgs {
    float* env_state;  // [E][6A]: pos(2A) vel(2A) landmarks(2A) -- written at the end (state after step T-1)
    int E, A, T, agent_ids;
    unsigned long long seed, act_seed;
    float act_eps;  // > 0: COMA's epsilon-mixed sampling (cm_rollout_spread_eps); < 0: greedy (evaluation rollouts)
    long env_offset, episode;
    const float* params; int din, H, L, K;
    float* obs; float* state; int* action; float* logp; float* reward;
    long obs_ld, state_ld;  // leading dimensions of obs [E][A][T][obs_ld] / state [E][T][state_ld] (>= din / 6 A A; cm_rollout_spread_ld)
    unsigned long long* prof;
};

__global__ __launch_bounds__(NTHREADS) void k_rollout_spread(const RolloutArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ROLLOUT_WAVE_PRIO();
    const Offsets off = make_offsets(a.din, a.H, a.L, a.K);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, lc = lane & 31;
    constexpr int K = SPREAD_K;  // compile-time: the samplers' k < 8 loops then evaluate 5 exponentials, not 8 selected ones
    const int A = a.A, T = a.T, H = a.H, L = a.L, din = a.din;
    const int EPT = TM / A, RT = EPT * A;  // envs / valid rows per tile
    const int Ds = 6 * A * A;
    // LDS carve
    float* Xs = smem;                    // obs tile; aliased by H1 once layer 0 has consumed it
    float* W0s = Xs + TM * LDT;
    float* H0 = W0s + HP * LDT;
    float* Ws = H0 + TM * LDT;
    float* wouts = Ws + HP * LDT;        // [16][WLD], rows >= K zero (operand of the 16x16x4 MFMA head)
    float* b0s = wouts + 16 * WLD;
    float* b1s = b0s + HP;
    float* bos = b1s + HP;               // [8]
    float* ls = bos + 8;                 // [TM][8] logits
    float* epos = ls + TM * 8;           // [TM][2] per row (env-local agent)
    float* evel = epos + TM * 2;
    float* elm = evel + TM * 2;          // [EPT*A][2] landmarks of env el at elm + el*2A
    long* obase = reinterpret_cast<long*>(elm + TM * 2 + TM);  // [TM] obs row base (elements), -1 = dead row (8-byte aligned: +TM pad)
    long* sbase = obase + TM;                          // [TM] state row base
    float* rscr = reinterpret_cast<float*>(sbase + TM);  // [2][TM] reward partials
    float* ubuf = rscr + 2 * TM;                         // [4][TM] uniforms of four steps

    for (int i = tid; i < 16 * HP; i += NTHREADS) {
        const int k = i / HP, c = i % HP;
        wouts[k * WLD + c] = (c < H && k < K) ? a.params[off.Wout + k * H + c] : 0.0f;
    }
    for (int i = tid; i < HP; i += NTHREADS) {
        b0s[i] = (i < H) ? a.params[off.b0 + i] : 0.0f;
        b1s[i] = (i < H && L > 0) ? a.params[off.bl(0) + i] : 0.0f;
    }
    if (tid < 8) bos[tid] = (tid < K) ? a.params[off.bout + tid] : 0.0f;
    stage_rows(W0s, a.params + off.W0, 0, H, din, 0, din);
    if (L > 0) stage_rows(Ws, a.params + off.Wl(0), 0, H, H, 0, H);

    const int ntiles = (a.E + EPT - 1) / EPT;
    const int hrow = tid >> 2, hq = tid & 3;
    // 16-byte buffer stores need 4-float-aligned obs rows and state segments and <= 1024 / 768 quads per tile
    // (obs rows may be padded to a leading dimension that is a multiple of 4: the last quad then carries the tile's zero padding)
    const int nq = (din + 3) >> 2, ns = (6 * A) >> 2;
    const bool vo = (a.obs_ld % 4 == 0) && (4 * nq <= a.obs_ld) && (TM * nq <= 4 * NTHREADS);
    const bool vs = ((6 * A) % 4 == 0) && (a.state_ld % 4 == 0) && (TM * ns <= 3 * NTHREADS);
    PH_DECL
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int e0 = tile * EPT;
        __syncthreads();
        // ---------------- reset (cm_env.hip k_env_reset): thread per (env, agent) row
        if (tid < TM) {
            const int el = tid / A, i = tid - el * A;
            const long e = e0 + el;
            const bool live = tid < RT && e < a.E;
            obase[tid] = live ? (e * A + i) * (long)T * a.obs_ld : -1;
            sbase[tid] = live ? e * (long)T * a.state_ld + (long)i * 6 * A : -1;
        }
        // 16-byte store slots of this thread: obs rows are din/4 quads wide, state segments 6A/4 quads (<= 4 / 3 slots)
        int oslot[4], sslot[3];
        {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int idx = tid + NTHREADS * it;
                const int r = idx / nq, c4 = idx - r * nq;
                oslot[it] = (vo && r < RT && e0 + r / A < a.E) ? ((r << 8) | c4) : -1;
                const int r2 = vs ? idx / ns : 0, c42 = idx - r2 * ns;
                if (it < 3) sslot[it] = (vs && r2 < RT && e0 + r2 / A < a.E) ? ((r2 << 8) | c42) : -1;
            }
        }
        if (tid < RT) {
            const int el = tid / A, i = tid - el * A;
            const int e = e0 + el;
            if (e < a.E) {
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
        // nearest-agent distance per landmark and collisions per agent from the CURRENT positions (thread per row)
        auto reward_partials = [&]() {
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
                        if (__builtin_amdgcn_sqrtf(cx * cx + cy * cy) < COLLIDE) col += 1.0f;
                    }
                }
                rscr[tid] = best; rscr[TM + tid] = col;
            }
        };
        for (int t = 0; t < T; ++t) {
            __syncthreads();
            PH(0);
            if (t > 0) reward_partials();  // reward of step t-1: positions after its physics update
            // ---------------- observations of step t -> Xs (4 lanes per row, features f = hq, hq+4, ...)
            {   // 4 lanes per row: lane hq handles entities j = hq, hq+4, ... (landmark j, other agent j, id j)
                const int el = hrow / A, i = hrow - el * A;
                const bool live = hrow < RT && (e0 + el) < a.E;
                float* xr = Xs + hrow * LDT;
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
                    for (int c = din + hq; c < KC; c += 4) xr[c] = 0.0f;  // MFMA chunk padding (H1 recycles this buffer)
                } else {
#pragma unroll
                    for (int j = 0; j < KC / 4; ++j) xr[4 * j + hq] = 0.0f;
                }
            }
            __syncthreads();
            PH(1);
            // ---------------- rollout-buffer writes (coalesced along the feature axis)
            // 16-byte stores from thread-private (row, quad-column) slots precomputed per tile; otherwise 4-byte stores in a flat
            // (row, column) enumeration: consecutive threads -> consecutive addresses of a row
            if (vo) {
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    if (oslot[it] >= 0) {
                        const int r = oslot[it] >> 8, c4 = oslot[it] & 255;
                        *reinterpret_cast<float4*>(a.obs + obase[r] + (long)t * a.obs_ld + 4 * c4) =
                            *reinterpret_cast<const float4*>(Xs + r * LDT + 4 * c4);
                    }
            } else {
                const float inv_din = 1.0f / (float)din;
                for (int idx = tid; idx < RT * din; idx += NTHREADS) {
                    const int r = (int)(((float)idx + 0.5f) * inv_din), c = idx - r * din;  // exact for idx < 2^22
                    const long ob = obase[r];
                    if (ob >= 0) a.obs[ob + (long)t * a.obs_ld + c] = Xs[r * LDT + c];
                }
            }
            if (vs) {
#pragma unroll
                for (int it = 0; it < 3; ++it)
                    if (sslot[it] >= 0) {
                        const int r = sslot[it] >> 8, c4 = sslot[it] & 255;
                        *reinterpret_cast<float4*>(a.state + sbase[r] + (long)t * a.state_ld + 4 * c4) =
                            *reinterpret_cast<const float4*>(Xs + r * LDT + 4 * c4);
                    }
            } else {
                const float inv_sw = 1.0f / (float)(6 * A);
                for (int idx = tid; idx < RT * 6 * A; idx += NTHREADS) {
                    const int r = (int)(((float)idx + 0.5f) * inv_sw), c = idx - r * 6 * A;
                    const long sb = sbase[r];
                    if (sb >= 0) a.state[sb + (long)t * a.state_ld + c] = Xs[r * LDT + c];
                }
            }
            // team reward of step t-1 (its partials were produced in the obs phase from the post-physics positions)
            if (t > 0 && tid < EPT && e0 + tid < a.E) {
                float r = 0.0f;
                for (int l = 0; l < A; ++l) r -= rscr[tid * A + l];
                for (int l = 0; l < A; ++l) r -= rscr[TM + tid * A + l];
                a.reward[(long)(e0 + tid) * T + (t - 1)] = r;
            }
            PH(2);
            // the uniforms do not depend on the logits: every 4th step the four waves draw the uniforms of the next four steps (wave w:
            // step t + w, thread = row) -- ten Philox rounds of quarter-rate integer multiplies on ONE wave per step were ~15 % of it
            if ((t & 3) == 0) {
                const int r = tid & 63, tq = tid >> 6;
                if (r < RT) {
                    const int el = r / A, i = r - el * A;
                    const unsigned long long gr = (unsigned long long)((a.env_offset + e0 + el) * A + i);
                    const cm_u4 rnd = cm_philox4x32((uint32_t)gr, (uint32_t)(gr >> 32), (uint32_t)(t + tq), CM_STREAM_ACT,
                                                    (uint32_t)a.act_seed, (uint32_t)(a.act_seed >> 32));
                    ubuf[tq * TM + r] = cm_u01(rnd.x);
                }
            }
            // ---------------- actor forward, layer 0
            f32x16 acc;
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
            rowpar_nt_hand(acc, Xs + 32 * wm * LDT, W0s + 32 * wn * LDT, (din + 7) >> 3);  // hand-ordered LDS reads (cm_mlp_kernel.h): 1.115 -> 1.08 ms at config 3
            {
                const float bias = b0s[32 * wn + lc];
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int row = 32 * wm + (g & 3) + 8 * (g >> 2) + 4 * h;
                    H0[row * LDT + 32 * wn + lc] = fmaxf(acc[g] + bias, 0.0f);
                }
            }
            __syncthreads();
            float* HL = H0;
            if (L > 0) {  // hidden layer; H1 aliases Xs (every wave is past its layer-0 reads)
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
                rowpar_nt_hand(acc, H0 + 32 * wm * LDT, Ws + 32 * wn * LDT, HP / 8);
                const float bias = b1s[32 * wn + lc];
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int row = 32 * wm + (g & 3) + 8 * (g >> 2) + 4 * h;
                    Xs[row * LDT + 32 * wn + lc] = fmaxf(acc[g] + bias, 0.0f);
                }
                HL = Xs;
                __syncthreads();
            }
            PH(3);
            // ---------------- head: logits on the 16x16x4 MFMA (same routine and summation order as k_mlp<M_ACT>)
            {
                const f32x4 lg = head_logits_mfma(HL + 16 * wave * LDT, wouts);
                const int n = lane & 15, g4 = lane >> 4;
                if (n < 8) {
                    const float bias = bos[n];
#pragma unroll
                    for (int q = 0; q < 4; ++q) ls[(16 * wave + 4 * g4 + q) * 8 + n] = lg[q] + bias;
                }
            }
            __syncthreads();
            PH(4);
            // ---------------- Categorical sample + log_prob (same arithmetic as k_mlp<M_ACT>), then physics
            if (tid < RT) {
                const int el = tid / A, i = tid - el * A;
                const long e = e0 + el;
                if (e < a.E) {
                    int chosen; float lpv;
                    const float u_row = ubuf[(t & 3) * TM + tid];
                    if (a.act_eps > 0.0f) cm_categorical_sample_eps(ls + tid * 8, K, u_row, a.act_eps, &chosen, &lpv);
                    else if (a.act_eps < 0.0f) cm_categorical_greedy(ls + tid * 8, K, &chosen, &lpv);  // evaluation rollouts (--greedy_eval)
                    else cm_categorical_sample(ls + tid * 8, K, u_row, &chosen, &lpv);
                    const long o = (e * A + i) * (long)T + t;
                    a.action[o] = chosen;
                    a.logp[o] = lpv;
                    // point-mass physics (cm_env.hip k_env_step)
                    const float ux = (chosen == 1) ? -ACCEL : (chosen == 2 ? ACCEL : 0.0f);
                    const float uy = (chosen == 3) ? -ACCEL : (chosen == 4 ? ACCEL : 0.0f);
                    const float vx = evel[2 * tid] * (1.0f - DAMP) + ux * DT;
                    const float vy = evel[2 * tid + 1] * (1.0f - DAMP) + uy * DT;
                    evel[2 * tid] = vx; evel[2 * tid + 1] = vy;
                    epos[2 * tid] += vx * DT; epos[2 * tid + 1] += vy * DT;
                }
            }
            PH(5);  // the barrier at the top of the next step orders the physics update before its readers
        }
        // reward of the last step
        __syncthreads();
        reward_partials();
        __syncthreads();
        if (tid < EPT && e0 + tid < a.E) {
            float r = 0.0f;
            for (int l = 0; l < A; ++l) r -= rscr[tid * A + l];
            for (int l = 0; l < A; ++l) r -= rscr[TM + tid * A + l];
            a.reward[(long)(e0 + tid) * T + (T - 1)] = r;
        }
        __syncthreads();
        // ---------------- final env state back to global (pos | vel | landmarks)
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


// ---------------------------------------------------------------------------------------------------------------------
// Small-tile form of the same rollout for SMALL env counts (one GPU's share of a sharded batch, config 2): the 64-row kernel
// above puts floor(64/A) envs on a workgroup, so 512 envs x 8 agents occupy 64 of the 256 CUs and every step is a serial
// chain over 64 rows (6 barriers, a 32x32 MFMA tile per wave and layer, a thread-per-row softmax).  Here a workgroup owns
// 16 (env, agent) rows -- 4x the workgroups -- and each step is cut to the work of 16 rows:
//   * obs / reward partials: 16 lanes per row (lane j = entity j), min / count over the lanes with DPP row rotations;
//   * layers on v_mfma_f32_16x16x4_f32, wave w = hidden columns 16w..16w+15 of all 16 rows (16 MFMAs x 32 cycles per layer
//     instead of 32 x 64).  k is fed in the order the 32x32x2 form consumes it ({8j+i, 8j+4+i}, i = 0..3), so that -- the
//     MFMA being an fmaf chain over k -- the logits come out bit-identical to the 64-row kernel and the per-step kernels;
//   * head: every wave computes the 16x16 logit tile (same routine as k_mlp<M_ACT>) and keeps row 4g+w of lane group g:
//     the K logits of a row sit in the 16 lanes of a DPP row, softmax / inverse-CDF sample run there (max, exp, serial-order
//     prefix sums by row shifts: same summation order as cm_categorical_sample), no LDS round trip and no barrier;
//   * 4 barriers per step, ~49 KB of LDS: three workgroups per CU.
constexpr int TS = 16;  // rows per small tile
#ifndef RO16_ABL
#define RO16_ABL 0  // probe builds only (tools/probes/rollout16_ablate.sh): bit mask of step phases compiled out
#endif

template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {  // lanes shifted in from outside the 16-lane row read 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
// all-reduce over the 16 lanes of a DPP row (row_ror:8,4,2,1 = 0x128, 0x124, 0x122, 0x121)
__device__ __forceinline__ float row16_min(float v) {
    v = fminf(v, dpp_f<0x128>(v)); v = fminf(v, dpp_f<0x124>(v)); v = fminf(v, dpp_f<0x122>(v)); return fminf(v, dpp_f<0x121>(v));
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<0x128>(v)); v = fmaxf(v, dpp_f<0x124>(v)); v = fmaxf(v, dpp_f<0x122>(v)); return fmaxf(v, dpp_f<0x121>(v));
}
__device__ __forceinline__ float row16_sum(float v) {  // exact for the small integer counts it is used on
    v += dpp_f<0x128>(v); v += dpp_f<0x124>(v); v += dpp_f<0x122>(v); return v + dpp_f<0x121>(v);
}
__device__ __forceinline__ int row16_imin(int v) {
    v = min(v, dpp_i<0x128>(v)); v = min(v, dpp_i<0x124>(v)); v = min(v, dpp_i<0x122>(v)); return min(v, dpp_i<0x121>(v));
}
__device__ __forceinline__ int row16_imax(int v) {
    v = max(v, dpp_i<0x128>(v)); v = max(v, dpp_i<0x124>(v)); v = max(v, dpp_i<0x122>(v)); return max(v, dpp_i<0x121>(v));
}
// lane n of a row <- x_0 + x_1 + ... + x_n summed LEFT TO RIGHT (the order of the serial sampler), for n < K; row_shr:1 = 0x111
__device__ __forceinline__ float row16_serial_prefix(float x, int n, int K) {
    float cum = x;
    for (int i = 1; i < K; ++i) {
        const float t = dpp_f<0x111>(cum);
        if (n == i) cum = t + x;
    }
    return cum;
}
// value of lane K-1 handed down to lanes 0..K-2 of the row (row_shl:1 = 0x101)
__device__ __forceinline__ float row16_from_lane(float v, int n, int K) {
    for (int i = K - 2; i >= 0; --i) {
        const float t = dpp_f<0x101>(v);
        if (n == i) v = t;
    }
    return v;
}

// acc[16 rows x 16 cols] += A[16 rows][8*kb] * B[16 cols(n)][8*kb]^T on the 16x16x4 MFMA, k in the order of rowpar_nt: lane group g
// supplies k = 8j + {0,4,1,5}[g] to the first MFMA of a block and + 2 to the second (bit-identical to the 64-row kernel's 32x32x2
// products).  The B operand lives in REGISTERS: w[2j] / w[2j + 1] = B[n][8j + {0,4,1,5}[g]] and + 2 (load_nt16_k8) -- the weights of
// the 16-row kernel never touch LDS, which shrinks its LDS footprint from 49 KB to under 10 KB and halves the LDS reads of a step
// (16-row rollout of 512 envs x 8 agents: 0.492 -> 0.466 ms; x 3 agents 0.50 -> 0.43 ms)
__device__ __forceinline__ void tile16_nt_reg(f32x4& acc, const float* As, const float (&w)[16], int kb) {
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const float4* ap = reinterpret_cast<const float4*>(As + n * LDT + 4 * (g & 1));
    const bool lo = g < 2;
    float4 a = ap[0];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (j < kb) {
            float4 an = a;
            if (j + 1 < kb) an = ap[2 * (j + 1)];
            acc = mfma16(lo ? a.x : a.y, w[2 * j], acc);
            acc = mfma16(lo ? a.z : a.w, w[2 * j + 1], acc);
            a = an;
        }
    }
}
// w[2j + i] = W[c0 + n][8j + 4 (g & 1) + (g >> 1) + 2 i], zero outside [nrows x ncols]
__device__ __forceinline__ void load_nt16_k8(float (&w)[16], const float* W, int c0, int nrows, int ld, int ncols) {
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const int c = c0 + n, k0 = 4 * (g & 1) + (g >> 1);
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = 8 * j + k0 + 2 * i;
            w[2 * j + i] = (c < nrows && k < ncols) ? W[(long)c * ld + k] : 0.0f;
        }
}
// head: logits[16 rows][16 outputs] with the zero-padded head weights in registers, k order of head_logits_mfma ({16j + 4g + i})
__device__ __forceinline__ f32x4 head_logits_reg(const float* HL, const float (&w)[16]) {
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const float4* ap = reinterpret_cast<const float4*>(HL + n * LDT + 4 * g);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < HP / 16; ++j) {
        const float4 a = ap[4 * j];
        acc = mfma16(a.x, w[4 * j], acc); acc = mfma16(a.y, w[4 * j + 1], acc);
        acc = mfma16(a.z, w[4 * j + 2], acc); acc = mfma16(a.w, w[4 * j + 3], acc);
    }
    return acc;
}

__global__ __launch_bounds__(NTHREADS, 3) void k_rollout_spread16(const RolloutArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ROLLOUT_WAVE_PRIO();
    const Offsets off = make_offsets(a.din, a.H, a.L, a.K);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n16 = lane & 15, g16 = lane >> 4;
    constexpr int K = SPREAD_K;  // compile-time: the samplers' k < 8 loops then evaluate 5 exponentials, not 8 selected ones
    const int A = a.A, T = a.T, H = a.H, L = a.L, din = a.din;
    const int EPT = TS / A, RT = EPT * A;  // envs / valid rows per tile
    const int Ds = 6 * A * A;
    float* Xs = smem;                     // [TS][LDT] obs tile; aliased by H1 once layer 0 has consumed it
    float* H0 = Xs + TS * LDT;            // [TS][LDT]
    float* epos = H0 + TS * LDT;          // [TS][2]
    float* evel = epos + TS * 2;
    float* elm = evel + TS * 2;
    long* obase = reinterpret_cast<long*>(elm + TS * 2);  // [TS] obs row base (elements), -1 = dead row
    long* sbase = obase + TS;
    float* rscr = reinterpret_cast<float*>(sbase + TS);   // [2][TS] reward partials

    // ---- the policy in registers: wave w owns hidden columns 16w .. 16w+15 of both layers; every wave holds the whole (padded) head
    float w0r[16], w1r[16], wor[16];
    load_nt16_k8(w0r, a.params + off.W0, 16 * wave, H, din, din);
    if (L > 0) load_nt16_k8(w1r, a.params + off.Wl(0), 16 * wave, H, H, H);
    else {
#pragma unroll
        for (int i = 0; i < 16; ++i) w1r[i] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 16 * j + 4 * g16 + i;
            wor[4 * j + i] = (n16 < K && k < H) ? a.params[off.Wout + n16 * H + k] : 0.0f;
        }
    const int hc = 16 * wave + n16;
    const float b0r = hc < H ? a.params[off.b0 + hc] : 0.0f, b1r = (hc < H && L > 0) ? a.params[off.bl(0) + hc] : 0.0f;
    const float bor = n16 < K ? a.params[off.bout + n16] : 0.0f;

    const int ntiles = (a.E + EPT - 1) / EPT;
    const int orow = tid >> 4, oq = tid & 15;  // obs phase: 16 lanes per row
    const int srow = 4 * g16 + wave;           // sampling phase: lane group g of wave w owns row 4g + w
    const int nq = (din + 3) >> 2, ns = (6 * A) >> 2;
    const bool vo = (a.obs_ld % 4 == 0) && (4 * nq <= a.obs_ld);                 // TS x nq <= 256 quads: at most one per thread
    const bool vs = ((6 * A) % 4 == 0) && (a.state_ld % 4 == 0);
    PH_DECL
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int e0 = tile * EPT;
        __syncthreads();
        if (tid < TS) {
            const int el = tid / A, i = tid - el * A;
            const long e = e0 + el;
            const bool live = tid < RT && e < a.E;
            obase[tid] = live ? (e * A + i) * (long)T * a.obs_ld : -1;
            sbase[tid] = live ? e * (long)T * a.state_ld + (long)i * 6 * A : -1;
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
        // geometry of this thread's obs-phase row and sampling-phase row
        const int o_el = orow / A, o_i = orow - o_el * A;
        const bool o_live = orow < RT && (e0 + o_el) < a.E;
        const int s_el = srow / A, s_i = srow - s_el * A;
        const bool s_live = srow < RT && (e0 + s_el) < a.E;
        const unsigned long long s_gr = (unsigned long long)((a.env_offset + e0 + s_el) * A + s_i);
        const long s_out = ((long)(e0 + s_el) * A + s_i) * (long)T;
        // nearest-agent distance of landmark o_i and collisions of agent o_i from the CURRENT positions: lane j = agent j
        auto reward_partials = [&]() {
            const float* pos = epos + o_el * 2 * A;
            float d = 3.0e38f, c = 0.0f;
            if (o_live && oq < A) {
                const float lx = elm[2 * orow], ly = elm[2 * orow + 1];
                const float qx = pos[2 * o_i], qy = pos[2 * o_i + 1];
                const float dx = pos[2 * oq] - lx, dy = pos[2 * oq + 1] - ly;
                d = __builtin_amdgcn_sqrtf(dx * dx + dy * dy);
                if (oq > o_i) {
                    const float cx = qx - pos[2 * oq], cy = qy - pos[2 * oq + 1];
                    if (__builtin_amdgcn_sqrtf(cx * cx + cy * cy) < COLLIDE) c = 1.0f;
                }
            }
            d = row16_min(d); c = row16_sum(c);
            if (oq == 0 && orow < RT) { rscr[orow] = d; rscr[TS + orow] = c; }
        };
        auto reward_write = [&](int t_prev) {
            if (tid < EPT && e0 + tid < a.E) {
                float r = 0.0f;
                for (int l = 0; l < A; ++l) r -= rscr[tid * A + l];
                for (int l = 0; l < A; ++l) r -= rscr[TS + tid * A + l];
                a.reward[(long)(e0 + tid) * T + t_prev] = r;
            }
        };
        float u16 = 0.0f;
        for (int t = 0; t < T; ++t) {
            __syncthreads();
            PH(0);
            // ---- observations of step t -> Xs (lane oq = entity oq: landmark, other agent, id) AND, from the same loads, the reward
            // partials of step t-1 (the positions are the ones after its physics update): nearest-agent distance of landmark o_i,
            // collisions of agent o_i -- as a separate pass over the same LDS words they were 0.84 us of the 3.84 us step
            {
                float* xr = Xs + orow * LDT;
                float dmin = 3.0e38f, ccol = 0.0f;
                if (o_live && (!(RO16_ABL & 8) || t == 0)) {
                    const float* pos = epos + o_el * 2 * A; const float* vel = evel + o_el * 2 * A; const float* lm = elm + o_el * 2 * A;
                    const float px = pos[2 * o_i], py = pos[2 * o_i + 1];
                    if (oq == 0) { xr[0] = vel[2 * o_i]; xr[1] = vel[2 * o_i + 1]; xr[2] = px; xr[3] = py; }
                    const int j = oq;
                    if (j < A) {
                        const float pjx = pos[2 * j], pjy = pos[2 * j + 1];
                        xr[4 + 2 * j] = lm[2 * j] - px; xr[5 + 2 * j] = lm[2 * j + 1] - py;
                        const float ax = pjx - px, ay = pjy - py;
                        if (j != o_i) {
                            const int jj = j < o_i ? j : j - 1;
                            xr[4 + 2 * A + 2 * jj] = ax; xr[5 + 2 * A + 2 * jj] = ay;
                            xr[2 + 4 * A + 2 * jj] = 0.0f; xr[3 + 4 * A + 2 * jj] = 0.0f;  // comm channel
                        }
                        if (a.agent_ids) xr[6 * A + j] = (j == o_i) ? 1.0f : 0.0f;
                        if (!(RO16_ABL & 4)) {
                            const float dx = pjx - elm[2 * orow], dy = pjy - elm[2 * orow + 1];
                            dmin = __builtin_amdgcn_sqrtf(dx * dx + dy * dy);
                            if (j > o_i && __builtin_amdgcn_sqrtf(ax * ax + ay * ay) < COLLIDE) ccol = 1.0f;  // (p_i - p_j)^2 == (p_j - p_i)^2 exactly
                        }
                    }
                    for (int c = din + oq; c < KC; c += 16) xr[c] = 0.0f;  // MFMA padding (H1 recycles this buffer)
                } else if (!o_live) {
#pragma unroll
                    for (int j = 0; j < KC / 16; ++j) xr[16 * j + oq] = 0.0f;
                }
                if (t > 0 && !(RO16_ABL & 4)) {
                    dmin = row16_min(dmin); ccol = row16_sum(ccol);
                    if (oq == 0 && orow < RT) { rscr[orow] = dmin; rscr[TS + orow] = ccol; }
                }
            }
            __syncthreads();
            PH(1);
            // ---------------- rollout-buffer writes (coalesced along the feature axis)
            if (!(RO16_ABL & 1)) {
                if (vo) {  // <= 16 x 16 obs quads and <= 16 x 15 state quads: at most one of each per thread
                    const int r = tid / nq, c4 = tid - r * nq;
                    if (r < RT && obase[r] >= 0)
                        *reinterpret_cast<float4*>(a.obs + obase[r] + (long)t * a.obs_ld + 4 * c4) = *reinterpret_cast<const float4*>(Xs + r * LDT + 4 * c4);
                } else {
                    for (int idx = tid; idx < RT * din; idx += NTHREADS) {
                        const int r = idx / din, c = idx - r * din;
                        const long ob = obase[r];
                        if (ob >= 0) a.obs[ob + (long)t * a.obs_ld + c] = Xs[r * LDT + c];
                    }
                }
                if (vs) {
                    const int r = tid / ns, c4 = tid - r * ns;
                    if (r < RT && sbase[r] >= 0)
                        *reinterpret_cast<float4*>(a.state + sbase[r] + (long)t * a.state_ld + 4 * c4) = *reinterpret_cast<const float4*>(Xs + r * LDT + 4 * c4);
                } else {
                    for (int idx = tid; idx < RT * 6 * A; idx += NTHREADS) {
                        const int r = idx / (6 * A), c = idx - r * 6 * A;
                        const long sb = sbase[r];
                        if (sb >= 0) a.state[sb + (long)t * a.state_ld + c] = Xs[r * LDT + c];
                    }
                }
            }
            if (t > 0 && !(RO16_ABL & 4)) reward_write(t - 1);
            // The uniforms do not depend on the logits and ten Philox rounds are ~2000 cycles of quarter-rate integer multiplies: the 16
            // lanes of a row's group draw the uniforms of 16 consecutive steps at once (lane n: step t + n) every 16th step, and each
            // step fetches its own with one ds_bpermute issued here, far ahead of the sampler that reads it.
            if ((t & 15) == 0) {
                const cm_u4 rnd = cm_philox4x32((uint32_t)s_gr, (uint32_t)(s_gr >> 32), (uint32_t)(t + n16), CM_STREAM_ACT,
                                                (uint32_t)a.act_seed, (uint32_t)(a.act_seed >> 32));
                u16 = cm_u01(rnd.x);
            }
            const float u_row = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * ((lane & 48) | (t & 15)), __builtin_bit_cast(int, u16)));
            PH(2);
            // ---------------- actor forward, layer 0: wave = hidden columns 16w..16w+15
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (!(RO16_ABL & 64)) tile16_nt_reg(acc, Xs, w0r, (din + 7) >> 3);
            {
                const float bias = b0r;
#pragma unroll
                for (int q = 0; q < 4; ++q) H0[(4 * g16 + q) * LDT + 16 * wave + n16] = fmaxf(acc[q] + bias, 0.0f);
            }
            __syncthreads();
            PH(3);
            const float* HL = H0;
            if (L > 0) {  // hidden layer; H1 aliases Xs (every wave is past its layer-0 reads and its buffer writes)
                acc = f32x4{0.f, 0.f, 0.f, 0.f};
                if (!(RO16_ABL & 16)) tile16_nt_reg(acc, H0, w1r, HP / 8);
                const float bias = b1r;
#pragma unroll
                for (int q = 0; q < 4; ++q) Xs[(4 * g16 + q) * LDT + 16 * wave + n16] = fmaxf(acc[q] + bias, 0.0f);
                HL = Xs;
                __syncthreads();
            }
            PH(4);
            // ---------------- head: the whole 16 x 16 logit tile per wave, row 4g + w kept by lane group g
            const f32x4 lg = (RO16_ABL & 32) ? f32x4{HL[n16], HL[n16 + 1], HL[n16 + 2], HL[n16 + 3]} : head_logits_reg(HL, wor);
            const float zraw = wave == 0 ? lg[0] : (wave == 1 ? lg[1] : (wave == 2 ? lg[2] : lg[3]));
            const bool kin = n16 < K;
            const float z = kin ? zraw + bor : -INFINITY;
            PH(5);
            // ---------------- Categorical sample + log_prob in the 16 lanes of the row (arithmetic of cm_categorical_sample[_eps])
            const float m = row16_max(z);
            const float ex = kin ? expf(z - m) : 0.0f;
            const bool av = kin && z > -5e8f;
            const float ssum = row16_from_lane(row16_serial_prefix(ex, n16, K), n16, K);  // s on lanes 0..K-1
            int chosen; float lpv;
            if (RO16_ABL & 2) {
                chosen = (int)(m * 0.0f + u_row * 4.0f); lpv = z;
            } else if (a.act_eps > 0.0f) {
                const float navail = row16_sum(av ? 1.0f : 0.0f);
                const float ca = (1.0f - a.act_eps) / ssum, cb = a.act_eps / fmaxf(navail, 1.0f);
                const float p = av ? ca * ex + cb : 0.0f;
                const float cum = row16_serial_prefix(p, n16, K);
                chosen = row16_imin((av && u_row < cum) ? n16 : 99);
                const int last = row16_imax(av ? n16 : 0);
                if (chosen == 99) chosen = last;
                lpv = logf(row16_max(n16 == chosen ? p : -INFINITY));
            } else {
                const float cum = row16_serial_prefix(ex, n16, K);
                chosen = row16_imin((av && u_row * ssum < cum) ? n16 : 99);
                const int last = row16_imax(av ? n16 : 0);
                if (chosen == 99) chosen = last;
                const float zc = row16_max(n16 == chosen ? z : -INFINITY);
                lpv = zc - (m + logf(ssum));
            }
            if (n16 == 0 && s_live) {
                if (!(RO16_ABL & 1)) { a.action[s_out + t] = chosen; a.logp[s_out + t] = lpv; }
                // point-mass physics (cm_env.hip k_env_step)
                const float ux = (chosen == 1) ? -ACCEL : (chosen == 2 ? ACCEL : 0.0f);
                const float uy = (chosen == 3) ? -ACCEL : (chosen == 4 ? ACCEL : 0.0f);
                const float vx = evel[2 * srow] * (1.0f - DAMP) + ux * DT;
                const float vy = evel[2 * srow + 1] * (1.0f - DAMP) + uy * DT;
                evel[2 * srow] = vx; evel[2 * srow + 1] = vy;
                epos[2 * srow] += vx * DT; epos[2 * srow + 1] += vy * DT;
            }
            PH(6);  // the barrier at the top of the next step orders the physics update before its readers
        }
        __syncthreads();
        reward_partials();
        __syncthreads();
        reward_write(T - 1);
        if (tid < RT) {  // final env state back to global (pos | vel | landmarks)
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

// ---------------------------------------------------------------------------------------------------------------------
// Store-wave form of the 16-row rollout (round 3): the same four compute waves plus a FIFTH wave that owns everything nothing in
// the step chain waits for.  One GPU's share of a sharded batch (512 envs x 8 agents = 256 tiles, one workgroup per CU) is a pure
// latency chain  positions -> obs tile -> layer 0 -> layer 1 -> logits -> sample -> physics -> positions,  and in k_rollout_spread16
// the rollout-buffer stores (0.43 us), the reward partials + reward store (0.6 us), the Philox draws and the action / log-prob stores
// all sat on it (3.64 us per step, profiles/r02_rollout16_ablation.txt).  Here the compute waves only build the obs tile, run the
// MLP, sample and move the agents; the store wave, in the same four barrier intervals of the step,
//   * computes the reward partials of step t-1 from the positions (stable until the physics at the end of step t) and stores the reward,
//   * copies the obs tile to the obs / state rollout buffers (16-byte stores) while layer 0 runs -- the tile is recycled as H1 afterwards,
//   * draws the sampler's uniforms 16 steps ahead (one Philox evaluation per lane on 4 of every 16 steps) into an LDS ring,
//   * flushes actions / log-probs from an LDS ring every 4 steps as 16-byte stores (a row's time axis is contiguous).
// It only has to arrive at each barrier before the compute waves do.  Same arithmetic, same Philox keys, same summation orders:
// bit-identical rollouts to both other tilings (tests/test_hip_parity.py).
// tile16_nt_reg with every A-operand quad requested before the first MFMA (16 registers): the chain of 2 kb dependent MFMAs then never
// waits for LDS (one quad ahead, as in tile16_nt_reg, leaves ~64 cycles between a read and its use).  Same operands, same order.
__device__ __forceinline__ void tile16_nt_reg_pre(f32x4& acc, const float* As, const float (&w)[16], int kb) {
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const float4* ap = reinterpret_cast<const float4*>(As + n * LDT + 4 * (g & 1));
    const bool lo = g < 2;
    float a0[8], a1[8];  // this lane group's two k values of every quad (selected at once: 16 registers, and no lane-indexed array)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 v = ap[2 * (j < kb ? j : 0)];
        a0[j] = lo ? v.x : v.y; a1[j] = lo ? v.z : v.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (j < kb) {
            acc = mfma16(a0[j], w[2 * j], acc);
            acc = mfma16(a1[j], w[2 * j + 1], acc);
        }
    }
}
// cm_categorical_sample / cm_categorical_sample_eps (cm_common.h) on K <= 8 logits held in REGISTERS: the same operations in the same
// order (one exp per action, left-to-right sums, first available action whose cumulative mass exceeds the threshold), with the chosen
// logit / probability tracked in the loop instead of indexed afterwards.
__device__ __forceinline__ void sample_row8(const float (&z)[8], int K, float u, float eps, int* action, float* logp) {
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (k < K) m = fmaxf(m, z[k]);
    float e[8], s = 0.0f;
    int chosen = -1, last = 0;
    if (eps < 0.0f) {  // greedy (cm_categorical_greedy: first maximal logit, log-prob = -log sum exp(z - max))
        int best = 0;
        float mm = -INFINITY;
#pragma unroll
        for (int k = 0; k < 8; ++k) if (k < K && z[k] > mm) { mm = z[k]; best = k; }
#pragma unroll
        for (int k = 0; k < 8; ++k) if (k < K) s += expf(z[k] - mm);
        *action = best;
        *logp = -logf(s);
        return;
    }
    if (eps > 0.0f) {
        int navail = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) if (k < K) navail += (z[k] > -5e8f) ? 1 : 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { e[k] = (k < K) ? expf(z[k] - m) : 0.0f; if (k < K) s += e[k]; }
        const float ca = (1.0f - eps) / s, cb = eps / (float)max(navail, 1);
        float cum = 0.0f, pc = 0.0f, plast = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k < K && z[k] > -5e8f) {
                const float p = ca * e[k] + cb;
                cum += p;
                last = k; plast = p;
                if (chosen < 0 && u < cum) { chosen = k; pc = p; }
            }
        }
        if (chosen < 0) { chosen = last; pc = plast; }
        *action = chosen;
        *logp = logf(pc);
        return;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { e[k] = (k < K) ? expf(z[k] - m) : 0.0f; s += e[k]; }
    const float thr = u * s;
    float cum = 0.0f, zc = 0.0f, zl = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k < K && z[k] > -5e8f) {
            cum += e[k];
            last = k; zl = z[k];
            if (chosen < 0 && thr < cum) { chosen = k; zc = z[k]; }
        }
    }
    if (chosen < 0) { chosen = last; zc = zl; }
    *action = chosen;
    *logp = zc - (m + logf(s));
}

constexpr int NT_SW = NTHREADS + 128;  // four compute waves + the writer wave + the scorer wave

__global__ __launch_bounds__(NT_SW, 2) void k_rollout_spread16s(const RolloutArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Offsets off = make_offsets(a.din, a.H, a.L, a.K);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool writer = wave == 4, scorer = wave == 5;
    if (wave < 4) ROLLOUT_WAVE_PRIO();  // the chain only: writer and scorer have slack and leave their issue slots to co-resident kernels
    const int n16 = lane & 15, g16 = lane >> 4;
    constexpr int K = SPREAD_K;  // compile-time: the samplers' k < 8 loops then evaluate 5 exponentials, not 8 selected ones
    const int A = a.A, T = a.T, H = a.H, L = a.L, din = a.din;
    const int EPT = TS / A, RT = EPT * A;  // envs / valid rows per tile
    float* Xs = smem;                     // [TS][LDT] obs tile; aliased by H1 once layer 0 has consumed it
    float* H0 = Xs + TS * LDT;            // [TS][LDT]
    float* epos = H0 + TS * LDT;          // [TS][2]
    float* evel = epos + TS * 2;
    float* elm = evel + TS * 2;
    float* rscr = elm + TS * 2;           // [2][TS] reward partials
    float* ubuf = rscr + 2 * TS;          // [2][16 steps][TS] uniforms, two blocks of 16 steps
    float* alog = ubuf + 2 * 16 * TS;     // [2][TS][4] action bits / log-prob of four steps per row

    // ---- compute waves: the policy in registers (wave w = hidden columns 16w .. 16w+15 of both layers, the whole padded head)
    float w0r[16], w1r[16], wor[16];
    float b0r = 0.f, b1r = 0.f, bor = 0.f;
    if (wave < 4) {
        load_nt16_k8(w0r, a.params + off.W0, 16 * wave, H, din, din);
        if (L > 0) load_nt16_k8(w1r, a.params + off.Wl(0), 16 * wave, H, H, H);
        else {
#pragma unroll
            for (int i = 0; i < 16; ++i) w1r[i] = 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = 16 * j + 4 * g16 + i;
                wor[4 * j + i] = (n16 < K && k < H) ? a.params[off.Wout + n16 * H + k] : 0.0f;
            }
        const int hc = 16 * wave + n16;
        b0r = hc < H ? a.params[off.b0 + hc] : 0.0f; b1r = (hc < H && L > 0) ? a.params[off.bl(0) + hc] : 0.0f;
        bor = n16 < K ? a.params[off.bout + n16] : 0.0f;
    }
    const int ntiles = (a.E + EPT - 1) / EPT;
    const int orow = (tid >> 4) & 15, oq = tid & 15;  // obs phase: 16 lanes per row (compute waves)
    const int srow = 4 * g16 + (wave & 3);             // sampling phase: lane group g of compute wave w owns row 4g + w
    const int nq = (din + 3) >> 2, ns = (6 * A) >> 2;
    const bool vo = (a.obs_ld % 4 == 0) && (4 * nq <= a.obs_ld);
    const bool vs = ((6 * A) % 4 == 0) && (a.state_ld % 4 == 0);
    const bool v4 = (T % 4) == 0;                      // 16-byte action / log-prob flushes need 4-aligned rows of the time axis
    PH_DECL
    // The two roles run separate tile loops with the SAME barrier sequence (one at the top of a tile, four per step, one after the last
    // step): as one loop the compute waves' 50 weight registers stayed live across the store wave's code (236 registers per lane).
    if (writer) {
        // ======================================= the writer wave: obs / state rollout buffers + the sampler's uniforms
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const int e0 = tile * EPT;
            __syncthreads();
            // 16-byte store slots of this lane: <= 16 x 16 obs quads and <= 16 x 15 state quads over 64 lanes.  Source (LDS) and
            // destination (t = 0) addresses are per-tile constants
            const float* osrc[4]; float* odst[4]; const float* ssrc[4]; float* sdst[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int idx = lane + 64 * it;
                const int r = idx / nq, c4 = idx - r * nq;
                const bool ok = vo && r < RT && e0 + r / A < a.E;
                osrc[it] = Xs + (ok ? r * LDT + 4 * c4 : 0);
                odst[it] = ok ? a.obs + ((long)(e0 + r / A) * A + (r % A)) * (long)T * a.obs_ld + 4 * c4 : nullptr;
                const int r2 = vs ? idx / ns : 0, c42 = idx - r2 * ns;
                const bool ok2 = vs && r2 < RT && e0 + r2 / A < a.E;
                ssrc[it] = Xs + (ok2 ? r2 * LDT + 4 * c42 : 0);
                sdst[it] = ok2 ? a.state + (long)(e0 + r2 / A) * (long)T * a.state_ld + (long)(r2 % A) * 6 * A + 4 * c42 : nullptr;
            }
            // uniforms: lane = (row n16, step offset g16) of a 4-step quarter of a 16-step block
            const int u_el = n16 / A, u_i = n16 - u_el * A;
            const unsigned long long u_gr = (unsigned long long)((a.env_offset + e0 + u_el) * A + u_i);
            auto draw = [&](int step) {
                const cm_u4 rnd = cm_philox4x32((uint32_t)u_gr, (uint32_t)(u_gr >> 32), (uint32_t)step, CM_STREAM_ACT,
                                                (uint32_t)a.act_seed, (uint32_t)(a.act_seed >> 32));
                ubuf[((step >> 4) & 1) * 16 * TS + (step & 15) * TS + n16] = cm_u01(rnd.x);
            };
#pragma unroll
            for (int q = 0; q < 4; ++q) draw(4 * q + g16);  // block 0, before the first step's barrier
            for (int t = 0; t < T; ++t) {
                PH(8);
                __syncthreads();  // B0
                PH(9);
                if ((t & 15) < 4 && ((t >> 4) + 1) * 16 < T) draw(((t >> 4) + 1) * 16 + 4 * (t & 15) + g16);  // next block, a quarter per step
                PH(10);
                __syncthreads();  // B1: obs tile of step t complete
                PH(11);
                if (vo) {
#pragma unroll
                    for (int it = 0; it < 4; ++it)
                        if (odst[it]) *reinterpret_cast<float4*>(odst[it] + (long)t * a.obs_ld) = *reinterpret_cast<const float4*>(osrc[it]);
                } else {
                    for (int idx = lane; idx < RT * din; idx += 64) {
                        const int r = idx / din, c = idx - r * din;
                        if (e0 + r / A < a.E) a.obs[((long)(e0 + r / A) * A + (r % A)) * (long)T * a.obs_ld + (long)t * a.obs_ld + c] = Xs[r * LDT + c];
                    }
                }
                if (vs) {
#pragma unroll
                    for (int it = 0; it < 4; ++it)
                        if (sdst[it]) *reinterpret_cast<float4*>(sdst[it] + (long)t * a.state_ld) = *reinterpret_cast<const float4*>(ssrc[it]);
                } else {
                    for (int idx = lane; idx < RT * 6 * A; idx += 64) {
                        const int r = idx / (6 * A), c = idx - r * 6 * A;
                        if (e0 + r / A < a.E) a.state[(long)(e0 + r / A) * (long)T * a.state_ld + (long)(r % A) * 6 * A + (long)t * a.state_ld + c] = Xs[r * LDT + c];
                    }
                }
                PH(12);
                __syncthreads();  // B2: layer 0 done -- the LDS reads above are complete (the barrier waits for them); the tile becomes H1
                PH(13);
                __syncthreads();  // B3
            }
            __syncthreads();      // after the last physics update
        }
        PH_FLUSH_SW;
        return;
    }
    if (scorer) {
        // ======================================= the scorer wave: team rewards + action / log-prob flushes
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const int e0 = tile * EPT;
            __syncthreads();
            // reward partials of the CURRENT positions (stable from B0 to B3 of a step): pass p covers rows 4p + g16, lane n16 = agent j
            // -- nearest-agent distance of landmark i, collisions of agent i; the expressions of k_rollout_spread16
            auto partials = [&](int p) {
                const int r = 4 * p + g16, el = r / A, i = r - el * A, j = n16;
                const bool live = r < RT && (e0 + el) < a.E;
                float dmin = 3.0e38f, ccol = 0.0f;
                if (live && j < A) {
                    const float* pos = epos + el * 2 * A;
                    const float px = pos[2 * i], py = pos[2 * i + 1];
                    const float pjx = pos[2 * j], pjy = pos[2 * j + 1];
                    const float ax = pjx - px, ay = pjy - py;
                    const float dx = pjx - elm[2 * r], dy = pjy - elm[2 * r + 1];
                    dmin = __builtin_amdgcn_sqrtf(dx * dx + dy * dy);
                    if (j > i && __builtin_amdgcn_sqrtf(ax * ax + ay * ay) < COLLIDE) ccol = 1.0f;
                }
                dmin = row16_min(dmin); ccol = row16_sum(ccol);
                if (n16 == 0 && r < RT) { rscr[r] = dmin; rscr[TS + r] = ccol; }
            };
            // the serial sums of k_rollout_spread16's reward_write (all 2 x 16 LDS words requested before the first subtraction)
            auto reward_store = [&](int t_out) {
                __builtin_amdgcn_wave_barrier();
                if (lane < EPT && e0 + lane < a.E) {
                    float d[TS], c[TS];
#pragma unroll
                    for (int l = 0; l < TS; ++l) { d[l] = l < A ? rscr[lane * A + l] : 0.0f; c[l] = l < A ? rscr[TS + lane * A + l] : 0.0f; }
                    float r = 0.0f;
#pragma unroll
                    for (int l = 0; l < TS; ++l) if (l < A) r -= d[l];
#pragma unroll
                    for (int l = 0; l < TS; ++l) if (l < A) r -= c[l];
                    a.reward[(long)(e0 + lane) * T + t_out] = r;
                }
                __builtin_amdgcn_wave_barrier();
            };
            // actions / log-probs of steps [t4, t4 + n) from the ring: lanes 0..15 the action rows, 16..31 the log-prob rows
            auto flush_alog = [&](int t4, int n) {
                if (lane < 32) {
                    const int r = lane & 15, which = lane >> 4;
                    const int el = r / A, i = r - el * A;
                    if (r < RT && e0 + el < a.E) {
                        const long o = ((long)(e0 + el) * A + i) * (long)T + t4;
                        const float4 v = *reinterpret_cast<const float4*>(alog + (which * TS + r) * 4);
                        float* dst = which ? a.logp + o : reinterpret_cast<float*>(a.action) + o;
                        if (v4 && n == 4) *reinterpret_cast<float4*>(dst) = v;
                        else {
                            if (n > 0) dst[0] = v.x;
                            if (n > 1) dst[1] = v.y;
                            if (n > 2) dst[2] = v.z;
                            if (n > 3) dst[3] = v.w;
                        }
                    }
                }
            };
            for (int t = 0; t < T; ++t) {
                __syncthreads();  // B0: positions after the physics of step t-1
                if (t > 0) {
                    if ((t & 3) == 0) flush_alog(t - 4, 4);  // the ring entries of steps t-4 .. t-1 (written before B0)
                    partials(0); partials(1);
                }
                __syncthreads();  // B1
                if (t > 0) { partials(2); partials(3); }
                __syncthreads();  // B2
                if (t > 0) reward_store(t - 1);
                __syncthreads();  // B3: the compute waves move the agents after this one
            }
            __syncthreads();      // positions after the last physics update
            partials(0); partials(1); partials(2); partials(3);
            reward_store(T - 1);
            { const int t4 = (T - 1) & ~3; flush_alog(t4, T - t4); }
        }
        return;
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int e0 = tile * EPT;
        __syncthreads();
        if (tid < TS) {
            const int el = tid / A, i = tid - el * A;
            const long e = e0 + el;
            const bool live = tid < RT && e < a.E;
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
            // =============================================== the compute waves ===============================================
            const int o_el = orow / A, o_i = orow - o_el * A;
            const bool o_live = orow < RT && (e0 + o_el) < a.E;
            const int s_el = srow / A;
            const bool s_live = srow < RT && (e0 + s_el) < a.E;
            for (int t = 0; t < T; ++t) {
                __syncthreads();  // B0
                PH(0);  // wait at B0 (+ the tail of the previous step after PH(7))
                {   // observations of step t -> Xs (lane oq = entity oq: landmark, other agent, id)
                    float* xr = Xs + orow * LDT;
                    if (o_live) {
                        const float* pos = epos + o_el * 2 * A; const float* vel = evel + o_el * 2 * A; const float* lm = elm + o_el * 2 * A;
                        const float px = pos[2 * o_i], py = pos[2 * o_i + 1];
                        if (oq == 0) { xr[0] = vel[2 * o_i]; xr[1] = vel[2 * o_i + 1]; xr[2] = px; xr[3] = py; }
                        const int j = oq;
                        if (j < A) {
                            const float pjx = pos[2 * j], pjy = pos[2 * j + 1];
                            xr[4 + 2 * j] = lm[2 * j] - px; xr[5 + 2 * j] = lm[2 * j + 1] - py;
                            const float ax = pjx - px, ay = pjy - py;
                            if (j != o_i) {
                                const int jj = j < o_i ? j : j - 1;
                                xr[4 + 2 * A + 2 * jj] = ax; xr[5 + 2 * A + 2 * jj] = ay;
                                xr[2 + 4 * A + 2 * jj] = 0.0f; xr[3 + 4 * A + 2 * jj] = 0.0f;  // comm channel
                            }
                            if (a.agent_ids) xr[6 * A + j] = (j == o_i) ? 1.0f : 0.0f;
                        }
                        for (int c = din + oq; c < KC; c += 16) xr[c] = 0.0f;  // MFMA padding (H1 recycles this buffer)
                    } else {
#pragma unroll
                        for (int j = 0; j < KC / 16; ++j) xr[16 * j + oq] = 0.0f;
                    }
                }
                const float u_row = ubuf[((t >> 4) & 1) * 16 * TS + (t & 15) * TS + srow];  // drawn >= 12 steps ago; consumed after the head
                PH(1);  // obs build
                __syncthreads();  // B1
                PH(2);  // wait at B1
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                tile16_nt_reg_pre(acc, Xs, w0r, (din + 7) >> 3);
#pragma unroll
                for (int q = 0; q < 4; ++q) H0[(4 * g16 + q) * LDT + 16 * wave + n16] = fmaxf(acc[q] + b0r, 0.0f);
                PH(3);  // layer 0
                __syncthreads();  // B2
                PH(4);  // wait at B2
                const float* HL = H0;
                if (L > 0) {  // hidden layer; H1 aliases Xs (every reader of the obs tile is past B2)
                    acc = f32x4{0.f, 0.f, 0.f, 0.f};
                    tile16_nt_reg_pre(acc, H0, w1r, HP / 8);
#pragma unroll
                    for (int q = 0; q < 4; ++q) Xs[(4 * g16 + q) * LDT + 16 * wave + n16] = fmaxf(acc[q] + b1r, 0.0f);
                    HL = Xs;
                }
                PH(5);  // layer 1
                __syncthreads();  // B3
                PH(6);  // wait at B3
                // head: the whole 16 x 16 logit tile per wave, row 4g + w kept by lane group g
                const f32x4 lg = head_logits_reg(HL, wor);
                const float zraw = wave == 0 ? lg[0] : (wave == 1 ? lg[1] : (wave == 2 ? lg[2] : lg[3]));
                const float zl = zraw + bor;
                // the K <= 8 logits of the row from the first lanes of its DPP row into EVERY lane's registers (row_shl:k, independent moves),
                // then the serial sampler of the per-step kernels in registers: ~30 dependent DPP steps (max, two serial prefix sums, hand-
                // down, argmin / argmax) were 1.7 k cycles of the step
                float zr[8];
                zr[0] = zl; zr[1] = dpp_f<0x101>(zl); zr[2] = dpp_f<0x102>(zl); zr[3] = dpp_f<0x103>(zl);
                zr[4] = dpp_f<0x104>(zl); zr[5] = dpp_f<0x105>(zl); zr[6] = dpp_f<0x106>(zl); zr[7] = dpp_f<0x107>(zl);
                int chosen; float lpv;
                sample_row8(zr, K, u_row, a.act_eps, &chosen, &lpv);   // valid in lane n16 == 0 of each row
                if (n16 == 0 && s_live) {
                    // point-mass physics (cm_env.hip k_env_step) first: the next step's chain starts from these positions
                    const float ux = (chosen == 1) ? -ACCEL : (chosen == 2 ? ACCEL : 0.0f);
                    const float uy = (chosen == 3) ? -ACCEL : (chosen == 4 ? ACCEL : 0.0f);
                    const float vx = evel[2 * srow] * (1.0f - DAMP) + ux * DT;
                    const float vy = evel[2 * srow + 1] * (1.0f - DAMP) + uy * DT;
                    evel[2 * srow] = vx; evel[2 * srow + 1] = vy;
                    epos[2 * srow] += vx * DT; epos[2 * srow + 1] += vy * DT;
                    alog[srow * 4 + (t & 3)] = __builtin_bit_cast(float, chosen);   // flushed by the store wave every 4 steps
                    alog[(TS + srow) * 4 + (t & 3)] = lpv;
                }
                PH(7);  // head + sample + physics
            }
            __syncthreads();
            if (tid < RT) {  // final env state back to global (pos | vel | landmarks)
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
    PH_FLUSH_C;
}

// ---------------------------------------------------------------------------------------------------------------------
// Six-wave form of the 64-row rollout (round 3): the role split of k_rollout_spread16s on 64-row tiles, for env counts that fill
// the chip (config 3: 512 tiles, two workgroups per CU).  In k_rollout_spread the non-matrix phases were 56 % of a step
// (profiles/r02_phase_rollout.txt: obs + reward partials 16 %, buffer writes 19 %, sample + physics 21 % -- the sampler and the reward
// on ONE wave while three wait); here
//   * the four compute waves build the obs tile, run the MLP, and EACH samples its own 16 rows straight after its head MFMAs
//     (wave-private logits, no barrier, four samplers in parallel) and moves those agents: four barriers per step instead of six;
//   * the writer wave copies the obs tile to the obs / state buffers while layer 0 runs and draws the uniforms four steps ahead;
//   * the scorer wave computes the reward partials and team rewards and flushes actions / log-probs every four steps (16-byte stores).
// Same arithmetic and keys as every other tiling: bit-identical rollouts (tests/test_hip_parity.py).
__global__ __launch_bounds__(NT_SW, 4) void k_rollout_spread64s(const RolloutArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Offsets off = make_offsets(a.din, a.H, a.L, a.K);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool writer = wave == 4, scorer = wave == 5;
    if (wave < 4) ROLLOUT_WAVE_PRIO();
    const int wm = (wave >> 1) & 1, wn = wave & 1, h = lane >> 5, lc = lane & 31;
    constexpr int K = SPREAD_K;  // compile-time: the samplers' k < 8 loops then evaluate 5 exponentials, not 8 selected ones
    const int A = a.A, T = a.T, H = a.H, L = a.L, din = a.din;
    const int EPT = TM / A, RT = EPT * A;  // envs / valid rows per tile
    // LDS carve (two workgroups per CU: 2 x 81.7 KB)
    float* Xs = smem;                    // obs tile; aliased by H1 once layer 0 has consumed it
    float* W0s = Xs + TM * LDT;
    float* H0 = W0s + HP * LDT;
    float* Ws = H0 + TM * LDT;
    float* wouts = Ws + HP * LDT;        // [16][WLD], rows >= K zero (operand of the 16x16x4 MFMA head)
    float* b0s = wouts + 16 * WLD;
    float* b1s = b0s + HP;
    float* bos = b1s + HP;               // [8]
    float* ls = bos + 8;                 // [TM][8] logits (wave w: rows 16w .. 16w+15); the scorer's reward partials [2][TM] alias it between B0 and B1
    float* epos = ls + TM * 8;           // [TM][2] per row (env-local agent)
    float* evel = epos + TM * 2;
    float* elm = evel + TM * 2;
    float* ubuf = elm + TM * 2;                          // [4][TM] uniforms of steps t .. t+3 (slot t & 3)
    float* alog = ubuf + 4 * TM;                         // [2][TM][4] action bits / log-prob of four steps per row
    float* rscr = ls;

    const int ntiles = (a.E + EPT - 1) / EPT;
    const int nq = (din + 3) >> 2, ns = (6 * A) >> 2;
    const bool vo = (a.obs_ld % 4 == 0) && (4 * nq <= a.obs_ld) && nq <= 16;
    const bool vs = ((6 * A) % 4 == 0) && (a.state_ld % 4 == 0) && ns <= 16;
    const bool v4 = (T % 4) == 0;
    PH_DECL
    if (writer) {
        // ======================================= the writer wave: the obs tile -> obs / state rollout buffers
        // 26 KB of 224- / 1536-byte row segments per step go through one CU's store path in ~5 k cycles whoever issues them (the
        // four-wave kernel spends them on the step chain).  The tile is only valid between B1 and B2 (layer 1 recycles it as H1), so
        // the wave copies its 16 quads to registers right after B1 -- 16 LDS reads -- and spreads the stores over the rest of the step,
        // a few per barrier interval, never the last to arrive.
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const int e0 = tile * EPT;
            __syncthreads();  // tile top
            const int wr = lane >> 4, wc = lane & 15;  // slot it: tile row wr + 4 it, quad column wc
            // element offsets of this lane's 16 (row, quad) slots relative to the tile's first obs row / state row (a tile spans
            // < 64 T ld floats: 32 bits always do).  Obs rows are one tile row apart: offset = o0 + it * ostep, valid below rmax live rows;
            // the state segments are not linear in the row (env-major), their offsets are kept (-1 = nothing to store)
            const int rmax = min(RT, (int)min((long)(a.E - e0) * A, (long)TM));
            const int o0 = wr * T * (int)a.obs_ld + 4 * wc, ostep = 4 * T * (int)a.obs_ld;
            const bool do_o = vo && wc < nq;
            int so[16];
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int r = wr + 4 * it, el = r / A, i = r - el * A;
                so[it] = (r < rmax && vs && wc < ns) ? el * T * (int)a.state_ld + i * 6 * A + 4 * wc : -1;
            }
            float* const obase = a.obs + (long)e0 * A * T * a.obs_ld;
            float* const sbase = a.state + (long)e0 * T * a.state_ld;
            float4 v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15;  // named, not an array: they live across barriers in registers
#define CM_GET(it) v##it = *reinterpret_cast<const float4*>(Xs + (wr + 4 * (it)) * LDT + 4 * wc)
#define CM_PUT(it) do { if (do_o && wr + 4 * (it) < rmax) *reinterpret_cast<float4*>(obase + (o0 + (it) * ostep + t * (int)a.obs_ld)) = v##it; \
                        if (so[it] >= 0) *reinterpret_cast<float4*>(sbase + (so[it] + t * (int)a.state_ld)) = v##it; } while (0)
            for (int t = 0; t < T; ++t) {
                PH(8);
                __syncthreads();  // B0
                PH(9);
                __syncthreads();  // B1: obs tile of step t complete
                PH(10);
                CM_GET(0); CM_GET(1); CM_GET(2); CM_GET(3); CM_GET(4); CM_GET(5); CM_GET(6); CM_GET(7);
                CM_GET(8); CM_GET(9); CM_GET(10); CM_GET(11); CM_GET(12); CM_GET(13); CM_GET(14); CM_GET(15);
                // feature widths that are not multiples of four floats: 4-byte stores in a flat (row, column) enumeration -- consecutive
                // lanes -> consecutive addresses of a row (row = idx / width by an exact float reciprocal, idx < 2^22)
                if (!vo) {
                    const float inv = 1.0f / (float)din;
                    for (int idx = lane; idx < RT * din; idx += 64) {
                        const int r = (int)(((float)idx + 0.5f) * inv), c = idx - r * din, el = r / A;
                        if (e0 + el < a.E) obase[(long)r * T * a.obs_ld + (long)t * a.obs_ld + c] = Xs[r * LDT + c];
                    }
                }
                if (!vs) {
                    const float inv = 1.0f / (float)(6 * A);
                    for (int idx = lane; idx < RT * 6 * A; idx += 64) {
                        const int r = (int)(((float)idx + 0.5f) * inv), c = idx - r * 6 * A, el = r / A, i = r - el * A;
                        if (e0 + el < a.E) sbase[((long)el * T + t) * a.state_ld + i * 6 * A + c] = Xs[r * LDT + c];
                    }
                }
                CM_PUT(0); CM_PUT(1); CM_PUT(2); CM_PUT(3);
                PH(11);
                __syncthreads();  // B2: the LDS reads above are complete (the barrier waits for them); the tile becomes H1
                PH(12);
                CM_PUT(4); CM_PUT(5); CM_PUT(6); CM_PUT(7); CM_PUT(8);
                PH(13);
                __syncthreads();  // B3
                CM_PUT(9); CM_PUT(10); CM_PUT(11); CM_PUT(12); CM_PUT(13); CM_PUT(14); CM_PUT(15);  // under the compute waves' head + sampling phase
            }
            __syncthreads();      // after the last physics update
#undef CM_PUT
#undef CM_GET
        }
        PH_FLUSH_SW;
        return;
    }
    if (scorer) {
        // ======================================= the scorer wave: team rewards, action / log-prob flushes, the sampler's uniforms
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const int e0 = tile * EPT;
            __syncthreads();  // tile top
            const int s_el = lane / A, s_l = lane - s_el * A;
            const bool s_live = lane < RT && e0 + s_el < a.E;
            const unsigned long long u_gr = (unsigned long long)((a.env_offset + e0 + s_el) * A + s_l);
            auto draw = [&](int step) {  // uniform of (row = lane, step) into slot step & 3
                if (s_live && step < T) {
                    const cm_u4 rnd = cm_philox4x32((uint32_t)u_gr, (uint32_t)(u_gr >> 32), (uint32_t)step, CM_STREAM_ACT,
                                                    (uint32_t)a.act_seed, (uint32_t)(a.act_seed >> 32));
                    ubuf[(step & 3) * TM + lane] = cm_u01(rnd.x);
                }
            };
#pragma unroll
            for (int q = 0; q < 4; ++q) draw(q);  // steps 0..3, before the first step's barrier
            // nearest-agent distance of landmark s_l and collisions of agent s_l from the CURRENT positions (lane = row; stable from B0 to
            // B3 of a step): the expressions of k_rollout_spread, agents [j0, j1) per call so that the work spreads over two barrier intervals
            float best = 3.0e38f, col = 0.0f;
            auto partials = [&](int j0, int j1) {
                if (lane < RT) {
                    const float* pos = epos + s_el * 2 * A;
                    const float lx = elm[2 * lane], ly = elm[2 * lane + 1];
                    const float qx = pos[2 * s_l], qy = pos[2 * s_l + 1];
                    float pxs[5], pys[5];  // requested before the first use
#pragma unroll
                    for (int q = 0; q < 5; ++q) { const int j = j0 + q; pxs[q] = j < j1 ? pos[2 * j] : 0.0f; pys[q] = j < j1 ? pos[2 * j + 1] : 0.0f; }
#pragma unroll
                    for (int q = 0; q < 5; ++q) {
                        const int j = j0 + q;
                        if (j < j1) {
                            const float dx = pxs[q] - lx, dy = pys[q] - ly;
                            best = fminf(best, __builtin_amdgcn_sqrtf(dx * dx + dy * dy));
                            if (j > s_l) {
                                const float cx = qx - pxs[q], cy = qy - pys[q];
                                if (__builtin_amdgcn_sqrtf(cx * cx + cy * cy) < COLLIDE) col += 1.0f;
                            }
                        }
                    }
                }
            };
            // the team reward of every env: the serial sums of k_rollout_spread
            auto reward_store = [&](int t_out) {
                if (lane < RT) { rscr[lane] = best; rscr[TM + lane] = col; }
                best = 3.0e38f; col = 0.0f;
                __builtin_amdgcn_wave_barrier();
                if (lane < EPT && e0 + lane < a.E) {
                    float d[10], c[10];
#pragma unroll
                    for (int l = 0; l < 10; ++l) { d[l] = l < A ? rscr[lane * A + l] : 0.0f; c[l] = l < A ? rscr[TM + lane * A + l] : 0.0f; }
                    float r = 0.0f;
#pragma unroll
                    for (int l = 0; l < 10; ++l) if (l < A) r -= d[l];
#pragma unroll
                    for (int l = 0; l < 10; ++l) if (l < A) r -= c[l];
                    a.reward[(long)(e0 + lane) * T + t_out] = r;
                }
                __builtin_amdgcn_wave_barrier();
            };
            auto flush_alog = [&](int t4, int n) {  // lane = row: actions and log-probs of steps [t4, t4 + n)
                if (s_live) {
                    const long o = ((long)(e0 + s_el) * A + s_l) * (long)T + t4;
                    const float4 va = *reinterpret_cast<const float4*>(alog + lane * 4);
                    const float4 vl = *reinterpret_cast<const float4*>(alog + (TM + lane) * 4);
                    float* da = reinterpret_cast<float*>(a.action) + o;
                    float* dl = a.logp + o;
                    if (v4 && n == 4) { *reinterpret_cast<float4*>(da) = va; *reinterpret_cast<float4*>(dl) = vl; }
                    else {
                        if (n > 0) { da[0] = va.x; dl[0] = vl.x; }
                        if (n > 1) { da[1] = va.y; dl[1] = vl.y; }
                        if (n > 2) { da[2] = va.z; dl[2] = vl.z; }
                        if (n > 3) { da[3] = va.w; dl[3] = vl.w; }
                    }
                }
            };
            // A <= 10: two parts of at most five agents.  The first part runs beside the compute waves' obs build (few issue conflicts), the
            // second beside their layer-0 products, where this wave gets about half the issue rate (profiles/r04_phase_rollout64s.txt: the
            // same four agents took 2.4 k cycles before B1 and ~4.5 k after it): the larger share goes first (A = 8: 5 + 3)
            const int jh = max(A - 5, min(5, (5 * A + 7) / 8));
            for (int t = 0; t < T; ++t) {
                __syncthreads();  // B0: positions after the physics of step t-1; the logit buffer (= rscr) is idle until after B3
                PH(14);
                if (t > 0) partials(0, jh);
                PH(15);
                __syncthreads();  // B1
                if (t > 0) partials(jh, A);
                __syncthreads();  // B2
                // the sums and the flush run in the B2 .. B3 interval (layer 1 of the compute waves), not before B2: with both halves of
                // the partials AND the serial sums before B2 this wave reached that barrier ~2.4 k cycles after the compute waves had
                // finished layer 0 (profiles/r04_phase_rollout64s_before.txt: "c: wait B2" 15.5 % of the step).  rscr (= the logit buffer) and the
                // action / log-prob ring are idle until the head phase after B3
                if (t > 0) {
                    reward_store(t - 1);
                    if ((t & 3) == 0) flush_alog(t - 4, 4);  // ring entries of steps t-4 .. t-1; slot 0 is rewritten after B3
                }
                __syncthreads();  // B3
                draw(t + 4);      // slot t & 3 was consumed before B1 of this step; under the compute waves' head + sampling phase
            }
            __syncthreads();      // positions after the last physics update
            partials(0, jh); partials(jh, A);
            reward_store(T - 1);
            { const int t4 = (T - 1) & ~3; flush_alog(t4, T - t4); }
        }
        PH_FLUSH_SC;
        return;
    }
    // =============================================== the compute waves ===============================================
    for (int i = tid; i < 16 * HP; i += NTHREADS) {
        const int k = i / HP, c = i % HP;
        wouts[k * WLD + c] = (c < H && k < K) ? a.params[off.Wout + k * H + c] : 0.0f;
    }
    for (int i = tid; i < HP; i += NTHREADS) {
        b0s[i] = (i < H) ? a.params[off.b0 + i] : 0.0f;
        b1s[i] = (i < H && L > 0) ? a.params[off.bl(0) + i] : 0.0f;
    }
    if (tid < 8) bos[tid] = (tid < K) ? a.params[off.bout + tid] : 0.0f;
    stage_rows(W0s, a.params + off.W0, 0, H, din, 0, din);
    if (L > 0) stage_rows(Ws, a.params + off.Wl(0), 0, H, H, 0, H);
    const int hrow = tid >> 2, hq = tid & 3;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int e0 = tile * EPT;
        __syncthreads();  // tile top
        if (tid < TM) {  // reset (cm_env.hip k_env_reset): thread per (env, agent) row
            const int el = tid / A, i = tid - el * A;
            const long e = e0 + el;
            const bool live = tid < RT && e < a.E;
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
        const int srow = 16 * wave + lane;  // sampling: lanes 0..15 of wave w own rows 16w .. 16w+15
        const int s_el = srow / A;
        const bool s_live = lane < 16 && srow < RT && (e0 + s_el) < a.E;
        for (int t = 0; t < T; ++t) {
            __syncthreads();  // B0
            PH(0);
            {   // observations of step t -> Xs: 4 lanes per row, lane hq handles entities j = hq, hq+4, ... (landmark j, other agent j, id j)
                const int el = hrow / A, i = hrow - el * A;
                const bool live = hrow < RT && (e0 + el) < a.E;
                float* xr = Xs + hrow * LDT;
                if (live) {
                    const float* pos = epos + el * 2 * A; const float* vel = evel + el * 2 * A; const float* lm = elm + el * 2 * A;
                    const float px = pos[2 * i], py = pos[2 * i + 1];
                    // entities j = hq, hq + 4, hq + 8 (7 A <= 64: at most three per lane): ALL their positions are requested before the first
                    // store -- the runtime-bounded loop was a chain of dependent LDS round trips (read, wait, seven stores, read ...)
                    float lx[3], ly[3], qx[3], qy[3];
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const int j = hq + 4 * u;
                        const bool ok = j < A;
                        lx[u] = ok ? lm[2 * j] : 0.0f; ly[u] = ok ? lm[2 * j + 1] : 0.0f;
                        qx[u] = ok ? pos[2 * j] : 0.0f; qy[u] = ok ? pos[2 * j + 1] : 0.0f;
                    }
                    if (hq == 0) { xr[0] = vel[2 * i]; xr[1] = vel[2 * i + 1]; xr[2] = px; xr[3] = py; }
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const int j = hq + 4 * u;
                        if (j < A) {
                            xr[4 + 2 * j] = lx[u] - px; xr[5 + 2 * j] = ly[u] - py;
                            if (j != i) {
                                const int jj = j < i ? j : j - 1;
                                xr[4 + 2 * A + 2 * jj] = qx[u] - px; xr[5 + 2 * A + 2 * jj] = qy[u] - py;
                                xr[2 + 4 * A + 2 * jj] = 0.0f; xr[3 + 4 * A + 2 * jj] = 0.0f;  // comm channel
                            }
                            if (a.agent_ids) xr[6 * A + j] = (j == i) ? 1.0f : 0.0f;
                        }
                    }
                    for (int c = din + hq; c < KC; c += 4) xr[c] = 0.0f;  // MFMA chunk padding (H1 recycles this buffer)
                } else {
#pragma unroll
                    for (int j = 0; j < KC / 4; ++j) xr[4 * j + hq] = 0.0f;
                }
            }
            const float u_row = s_live ? ubuf[(t & 3) * TM + srow] : 0.0f;  // drawn four steps ago
            PH(1);
            __syncthreads();  // B1
            PH(2);
            f32x16 acc;
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
            rowpar_nt_hand(acc, Xs + 32 * wm * LDT, W0s + 32 * wn * LDT, (din + 7) >> 3);
            {
                const float bias = b0s[32 * wn + lc];
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int row = 32 * wm + (g & 3) + 8 * (g >> 2) + 4 * h;
                    H0[row * LDT + 32 * wn + lc] = fmaxf(acc[g] + bias, 0.0f);
                }
            }
            PH(3);
            __syncthreads();  // B2
            PH(4);
            float* HL = H0;
            if (L > 0) {  // hidden layer; H1 aliases Xs (every reader of the obs tile is past B2)
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
                rowpar_nt_hand(acc, H0 + 32 * wm * LDT, Ws + 32 * wn * LDT, HP / 8);
                const float bias = b1s[32 * wn + lc];
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int row = 32 * wm + (g & 3) + 8 * (g >> 2) + 4 * h;
                    Xs[row * LDT + 32 * wn + lc] = fmaxf(acc[g] + bias, 0.0f);
                }
                HL = Xs;
            }
            PH(5);
            __syncthreads();  // B3
            PH(6);
            {   // head: logits of this wave's 16 rows on the 16x16x4 MFMA (same routine and summation order as k_mlp<M_ACT>), wave-private
                const f32x4 lg = head_logits_mfma(HL + 16 * wave * LDT, wouts);
                const int n = lane & 15, g4 = lane >> 4;
                if (n < 8) {
                    const float bias = bos[n];
#pragma unroll
                    for (int q = 0; q < 4; ++q) ls[(16 * wave + 4 * g4 + q) * 8 + n] = lg[q] + bias;
                }
            }
            __builtin_amdgcn_wave_barrier();  // same-wave LDS hand-off (DS operations of one wave execute in order)
            if (s_live) {  // Categorical sample + log_prob (the per-step kernels' routines), then this agent's physics
                const int i = srow - s_el * A;
                int chosen; float lpv;
                if (a.act_eps > 0.0f) cm_categorical_sample_eps(ls + srow * 8, K, u_row, a.act_eps, &chosen, &lpv);
                else if (a.act_eps < 0.0f) cm_categorical_greedy(ls + srow * 8, K, &chosen, &lpv);  // evaluation rollouts (--greedy_eval)
                else cm_categorical_sample(ls + srow * 8, K, u_row, &chosen, &lpv);
                (void)i;
                const float ux = (chosen == 1) ? -ACCEL : (chosen == 2 ? ACCEL : 0.0f);
                const float uy = (chosen == 3) ? -ACCEL : (chosen == 4 ? ACCEL : 0.0f);
                const float vx = evel[2 * srow] * (1.0f - DAMP) + ux * DT;
                const float vy = evel[2 * srow + 1] * (1.0f - DAMP) + uy * DT;
                evel[2 * srow] = vx; evel[2 * srow + 1] = vy;
                epos[2 * srow] += vx * DT; epos[2 * srow + 1] += vy * DT;
                alog[srow * 4 + (t & 3)] = __builtin_bit_cast(float, chosen);   // flushed by the scorer wave every 4 steps
                alog[(TM + srow) * 4 + (t & 3)] = lpv;
            }
            PH(7);
        }
        __syncthreads();  // after the last physics update (the scorer wave reads the final positions)
        if (tid < RT) {   // final env state back to global (pos | vel | landmarks): nothing writes these LDS words before the next tile's reset
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
    PH_FLUSH_C;
}

}  // namespace

extern "C" int cm_rollout_spread_supported(int A, int agent_ids, int hidden, int n_hidden_layers) {
    const int din = 6 * A + (agent_ids ? A : 0);
    return (A >= 1 && din <= KC && hidden <= HP && n_hidden_layers <= 1) ? 1 : 0;
}

static int rollout_spread(float* env_state, int E, int A, int T, int agent_ids, uint64_t seed, uint64_t act_seed,
                          int64_t env_offset, int64_t episode, const float* params, int hidden, int n_hidden_layers, float eps,
                          float* obs, float* state, int32_t* action, float* logp, float* reward, cm_stream_t stream,
                          int64_t obs_ld = 0, int64_t state_ld = 0) {
    CM_REQUIRE(E > 0 && T > 0, "cm_rollout_spread: bad dims E=%d T=%d", E, T);
    CM_REQUIRE(cm_rollout_spread_supported(A, agent_ids, hidden, n_hidden_layers),
               "cm_rollout_spread: unsupported shape A=%d hidden=%d layers=%d (use cm_policy_act + cm_synth_env_step)", A, hidden, n_hidden_layers);
    RolloutArgs a = {};
    a.env_state = env_state; a.E = E; a.A = A; a.T = T; a.agent_ids = agent_ids; a.seed = seed; a.act_seed = act_seed; a.act_eps = eps;
    a.env_offset = env_offset; a.episode = episode; a.params = params; a.din = 6 * A + (agent_ids ? A : 0);
    a.H = hidden; a.L = n_hidden_layers; a.K = SPREAD_K;
    a.obs = obs; a.state = state; a.action = action; a.logp = logp; a.reward = reward;
    a.obs_ld = obs_ld ? obs_ld : a.din; a.state_ld = state_ld ? state_ld : 6 * A * A;
    CM_REQUIRE(a.obs_ld >= a.din && a.state_ld >= 6 * A * A, "cm_rollout_spread: leading dimensions %ld / %ld below the widths %d / %d", a.obs_ld, a.state_ld, a.din, 6 * A * A);
#ifdef CM_PHASE_PROF
    a.prof = g_prof;
#endif
    const int EPT = TM / A;
    const int ntiles = (E + EPT - 1) / EPT;
    // Few 64-row tiles (a GPU's share of a sharded batch, config 2) leave most CUs idle and every step a 64-row serial chain: while
    // all 16-row tiles are resident at once (768 workgroups; a second pass would double the time) the 16-row form runs instead (same arithmetic; cm_set_option("rollout_tile", "16s" / "16" / "64") forces one for A/B runs and tests)
    const int forced = cm_option(CM_OPTION_ROLLOUT_TILE);  // 0 auto, 64, 16 (four-wave form), 17 ("16s": store-wave form)
    const bool can16 = A <= TS;
    const int EPT16 = can16 ? TS / A : 1;
    const int nt16 = (E + EPT16 - 1) / EPT16;
    const bool use16 = can16 && (forced == 16 || forced == 17 || (forced == 0 && nt16 <= 768));  // forced: 64 = four-wave 64-row, 65 = "64s"  // one resident wave of 16-row workgroups (3 per CU)
    // at most one 16-row workgroup per CU: the store-wave form (four compute waves + writer + scorer)
    const bool use16s = use16 && (forced == 17 || (forced == 0 && nt16 <= 256));  // one six-wave workgroup per CU
    if (use16s) {
        const size_t lds = ((size_t)TS * LDT * 2 + TS * 2 * 3 + 2 * TS + 2 * 16 * TS + 2 * TS * 4) * sizeof(float);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rollout_spread16s), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_rollout_spread16s, dim3(nt16 < 256 ? nt16 : 256), dim3(NT_SW), lds, (hipStream_t)stream, a);
        CM_CHECK_LAUNCH("cm_rollout_spread");
        return 0;
    }
    if (use16 && !(eps < 0.0f)) {  // greedy evaluation rollouts of 257 .. 768 16-row tiles take the 64-row forms below
        const size_t lds16 = ((size_t)TS * LDT * 2 + TS * 2 * 3 + 4 * TS + 2 * TS) * sizeof(float);  // tiles + env scratch: the weights are in registers
        const int grid16 = nt16 < 768 ? nt16 : 768;  // three workgroups per CU (launch bounds)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rollout_spread16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
        hipLaunchKernelGGL(k_rollout_spread16, dim3(grid16), dim3(NTHREADS), lds16, (hipStream_t)stream, a);
        CM_CHECK_LAUNCH("cm_rollout_spread");
        return 0;
    }
    // the six-wave form ("64s"; default for 64-row tiles): writer + scorer waves, per-wave samplers.  Buffers whose rows are not
    // 16-byte aligned (unpadded odd widths) stay on the four-wave kernel, whose 256 threads share the 4-byte stores
    const bool vec64 = (a.obs_ld % 4 == 0) && (4 * ((a.din + 3) >> 2) <= a.obs_ld) && ((6 * A) % 4 == 0) && (a.state_ld % 4 == 0);
    if (forced == 65 || (forced != 64 && vec64)) {
        const size_t lds64s = ((size_t)TM * LDT * 2 + (size_t)HP * LDT * 2 + 16 * WLD + 2 * HP + 8 + TM * 8 + TM * 2 * 3 + 4 * TM + 2 * TM * 4) * sizeof(float);  // 81 184 B: two per CU
        const int grid64s = ntiles < 512 ? ntiles : 512;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rollout_spread64s), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64s);
        hipLaunchKernelGGL(k_rollout_spread64s, dim3(grid64s), dim3(NT_SW), lds64s, (hipStream_t)stream, a);
        CM_CHECK_LAUNCH("cm_rollout_spread");
        return 0;
    }
    const size_t lds_floats = (size_t)TM * LDT * 2 + (size_t)HP * LDT * 2 + 16 * WLD + 2 * HP + 8 + TM * 8 + TM * 2 * 3 + TM + 4 * TM + 2 * TM + 4 * TM;
    const size_t lds_bytes = lds_floats * sizeof(float);
    const int grid = ntiles < 512 ? ntiles : 512;  // <= 80 KB of LDS: two workgroups per CU overlap each other's latencies
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rollout_spread), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(k_rollout_spread, dim3(grid), dim3(NTHREADS), lds_bytes, (hipStream_t)stream, a);
    CM_CHECK_LAUNCH("cm_rollout_spread");
    return 0;
}

extern "C" int cm_rollout_spread(float* env_state, int E, int A, int T, int agent_ids, uint64_t seed, uint64_t act_seed,
                                 int64_t env_offset, int64_t episode, const float* params, int hidden, int n_hidden_layers,
                                 float* obs, float* state, int32_t* action, float* logp, float* reward, cm_stream_t stream) {
    return rollout_spread(env_state, E, A, T, agent_ids, seed, act_seed, env_offset, episode, params, hidden, n_hidden_layers, 0.0f, obs,
                          state, action, logp, reward, stream);
}

/* the same fused rollout with COMA's epsilon-mixed policy (coma_multienvs.py:177-186, 477-484) */
extern "C" int cm_rollout_spread_eps(float* env_state, int E, int A, int T, int agent_ids, uint64_t seed, uint64_t act_seed,
                                     int64_t env_offset, int64_t episode, const float* params, int hidden, int n_hidden_layers,
                                     double eps, float* obs, float* state, int32_t* action, float* logp, float* reward,
                                     cm_stream_t stream) {
    CM_REQUIRE(eps >= 0.0 && eps <= 1.0, "cm_rollout_spread_eps: eps=%g outside [0, 1]", eps);
    return rollout_spread(env_state, E, A, T, agent_ids, seed, act_seed, env_offset, episode, params, hidden, n_hidden_layers, (float)eps,
                          obs, state, action, logp, reward, stream);
}

/* both of the above with explicit leading dimensions of the obs / state buffers (include/cleanmarl_hip.h, "_ld" variants) */
extern "C" int cm_rollout_spread_ld(float* env_state, int E, int A, int T, int agent_ids, uint64_t seed, uint64_t act_seed,
                                    int64_t env_offset, int64_t episode, const float* params, int hidden, int n_hidden_layers, double eps,
                                    float* obs, int64_t obs_ld, float* state, int64_t state_ld, int32_t* action, float* logp,
                                    float* reward, cm_stream_t stream) {
    CM_REQUIRE(eps <= 1.0, "cm_rollout_spread_ld: eps=%g above 1 (eps < 0: greedy, the argmax of the masked logits)", eps);
    return rollout_spread(env_state, E, A, T, agent_ids, seed, act_seed, env_offset, episode, params, hidden, n_hidden_layers, (float)eps,
                          obs, state, action, logp, reward, stream, obs_ld, state_ld);
}
