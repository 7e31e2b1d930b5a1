// cm_env.hip -- on-device synthetic MPE-like environment ("simple_spread" shapes and dynamics).
//
// Stands in for the per-env OS process + Pipe round trip of cleanmarl/mappo_multienvs.py:246-285, 393-453
// in the synthetic benchmark configs.  Semantics follow the reference's CommonInterface
// (cleanmarl/env/common_interface.py:5-23) as wrapped by cleanmarl/env/pettingzoo_wrapper.py:
//   obs[a]  = [vel_a(2), pos_a(2), landmark_j - pos_a (2A), pos_j - pos_a for j != a (2(A-1)),
//              comm zeros (2(A-1))]  (+ one-hot agent id when agent_ids)      -> Do = 6A (+A)   (:68-73, :93-98)
//   state   = concat_a raw obs[a]                                             -> Ds = 6A*A      (:95)
//   reward  = one team scalar per env step                                     (:66)
//   actions = {0: no-op, 1: -x, 2: +x, 3: -y, 4: +y}; every action always available (:79-90)
// Physics are our own re-statement of MPE-style point-mass dynamics (damping 0.25, dt 0.1, accel 5):
// pettingzoo itself is not installed here, so this is "MPE-like", not a bit-copy of simple_spread_v3.
// Randomness: Philox4x32-10 keyed by the run seed and counted by (global env index, episode, entity), so
// the same env gets the same episode no matter which GPU owns it.
// cleanmarl_amd/env/synthetic.py is the numpy twin (CommonInterface env) the parity tests compare against.
#include "cm_common.h"

namespace {

constexpr float DAMP = 0.25f, DT = 0.1f, ACCEL = 5.0f, COLLIDE = 0.3f;

__device__ __forceinline__ void write_obs(const float* pos, const float* vel, const float* lm, int A, int i,
                                          int agent_ids, float* orow, float* srow) {
    // pos/vel/lm: this env's [A][2] arrays (LDS).  orow: obs[e][i][t][:], srow: state[e][t][i*6A : (i+1)*6A]
    const float px = pos[2 * i], py = pos[2 * i + 1];
    int o = 0;
    float v;
#define PUT(val) do { v = (val); orow[o] = v; srow[o] = v; ++o; } while (0)
    PUT(vel[2 * i]); PUT(vel[2 * i + 1]);
    PUT(px); PUT(py);
    for (int j = 0; j < A; ++j) { PUT(lm[2 * j] - px); PUT(lm[2 * j + 1] - py); }
    for (int j = 0; j < A; ++j) if (j != i) { PUT(pos[2 * j] - px); PUT(pos[2 * j + 1] - py); }
    for (int j = 0; j < 2 * (A - 1); ++j) PUT(0.0f);
#undef PUT
    if (agent_ids) for (int j = 0; j < A; ++j) orow[o + j] = (j == i) ? 1.0f : 0.0f;
}

__global__ __launch_bounds__(256) void k_env_reset(float* __restrict__ env_state, int E, int A, int agent_ids,
                                                   unsigned long long seed, long env_offset, long episode,
                                                   float* __restrict__ obs, float* __restrict__ state, int T) {
    extern __shared__ float sm[];
    const int epb = 256 / A;
    const int el = threadIdx.x / A, i = threadIdx.x % A;
    const long e = (long)blockIdx.x * epb + el;
    const bool on = (el < epb) && (e < E);
    float* pos = sm + el * 6 * A; float* vel = pos + 2 * A; float* lm = vel + 2 * A;
    if (on) {
        const unsigned long long ge = (unsigned long long)(env_offset + e);
        const cm_u4 ra = cm_philox4x32((uint32_t)ge, (uint32_t)episode, (uint32_t)i, CM_STREAM_ENV_RESET,
                                       (uint32_t)seed, (uint32_t)(seed >> 32));
        pos[2 * i] = 2.0f * cm_u01(ra.x) - 1.0f; pos[2 * i + 1] = 2.0f * cm_u01(ra.y) - 1.0f;
        lm[2 * i] = 2.0f * cm_u01(ra.z) - 1.0f; lm[2 * i + 1] = 2.0f * cm_u01(ra.w) - 1.0f;
        vel[2 * i] = 0.0f; vel[2 * i + 1] = 0.0f;
    }
    __syncthreads();
    if (on) {
        float* es = env_state + e * 6 * A;
        es[2 * i] = pos[2 * i]; es[2 * i + 1] = pos[2 * i + 1];
        es[2 * A + 2 * i] = 0.0f; es[2 * A + 2 * i + 1] = 0.0f;
        es[4 * A + 2 * i] = lm[2 * i]; es[4 * A + 2 * i + 1] = lm[2 * i + 1];
        const int Do = 6 * A + (agent_ids ? A : 0);
        write_obs(pos, vel, lm, A, i, agent_ids, obs + ((e * A + i) * (long)T + 0) * Do,
                  state + (e * (long)T + 0) * (6L * A * A) + (long)i * 6 * A);
    }
}

__global__ __launch_bounds__(256) void k_env_step(float* __restrict__ env_state, const int* __restrict__ action,
                                                  int E, int A, int agent_ids, int t, int T,
                                                  float* __restrict__ reward, float* __restrict__ obs,
                                                  float* __restrict__ state) {
    extern __shared__ float sm[];
    const int epb = 256 / A;
    const int el = threadIdx.x / A, i = threadIdx.x % A;
    const long e = (long)blockIdx.x * epb + el;
    const bool on = (el < epb) && (e < E);
    float* pos = sm + el * 6 * A; float* vel = pos + 2 * A; float* lm = vel + 2 * A;
    if (on) {
        float* es = env_state + e * 6 * A;
        const int k = action[(e * A + i) * (long)T + t];
        const float ux = (k == 1) ? -ACCEL : (k == 2 ? ACCEL : 0.0f);
        const float uy = (k == 3) ? -ACCEL : (k == 4 ? ACCEL : 0.0f);
        float vx = es[2 * A + 2 * i] * (1.0f - DAMP) + ux * DT;
        float vy = es[2 * A + 2 * i + 1] * (1.0f - DAMP) + uy * DT;
        float px = es[2 * i] + vx * DT;
        float py = es[2 * i + 1] + vy * DT;
        es[2 * i] = px; es[2 * i + 1] = py; es[2 * A + 2 * i] = vx; es[2 * A + 2 * i + 1] = vy;
        pos[2 * i] = px; pos[2 * i + 1] = py; vel[2 * i] = vx; vel[2 * i + 1] = vy;
        lm[2 * i] = es[4 * A + 2 * i]; lm[2 * i + 1] = es[4 * A + 2 * i + 1];
    }
    __syncthreads();
    if (on) {
        if (i == 0) {
            float r = 0.0f;
            for (int l = 0; l < A; ++l) {
                float best = 3.0e38f;
                for (int j = 0; j < A; ++j) {
                    const float dx = pos[2 * j] - lm[2 * l], dy = pos[2 * j + 1] - lm[2 * l + 1];
                    best = fminf(best, __builtin_amdgcn_sqrtf(dx * dx + dy * dy));
                }
                r -= best;
            }
            for (int j = 0; j < A; ++j)
                for (int q = j + 1; q < A; ++q) {
                    const float dx = pos[2 * j] - pos[2 * q], dy = pos[2 * j + 1] - pos[2 * q + 1];
                    if (__builtin_amdgcn_sqrtf(dx * dx + dy * dy) < COLLIDE) r -= 1.0f;
                }
            reward[e * (long)T + t] = r;
        }
        if (t + 1 < T) {
            const int Do = 6 * A + (agent_ids ? A : 0);
            write_obs(pos, vel, lm, A, i, agent_ids, obs + ((e * A + i) * (long)T + (t + 1)) * Do,
                      state + (e * (long)T + (t + 1)) * (6L * A * A) + (long)i * 6 * A);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// "shape" env: a fixed-shape stand-in for envs we cannot ship (SMAClite-like: wide obs, separate global state, 17
// actions with availability masks).  Observations / states / masks are a pure function of (seed, env, episode, t),
// so the whole episode's inputs are generated by ONE launch; the team reward depends on the sampled actions.
__device__ __forceinline__ float normal01(uint32_t a, uint32_t b) {  // Box-Muller on two Philox words
    const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f), u2 = cm_u01(b);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

// FOUR normals per Philox call (both Box-Muller outputs of both word pairs): features 4g .. 4g+3 of a row share the counter
// (.., 0x100 + g); the generator was ten Philox rounds + log + sqrt + cos PER ELEMENT and ran at 0.8 TB/s of buffer writes
struct N4 { float v[4]; };
__device__ __forceinline__ N4 normal4(const cm_u4& w) {
    N4 o;
    const float u1 = ((float)(w.x >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = ((float)(w.z >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float r1 = sqrtf(-2.0f * logf(u1)), r2 = sqrtf(-2.0f * logf(u3));
    float s1, c1, s2, c2;
    sincosf(6.283185307179586f * cm_u01(w.y), &s1, &c1);
    sincosf(6.283185307179586f * cm_u01(w.w), &s2, &c2);
    o.v[0] = r1 * c1; o.v[1] = r1 * s1; o.v[2] = r2 * c2; o.v[3] = r2 * s2;
    return o;
}

// Index walker of one segment of k_shape_fill: element j of [0, n) decomposes as j = (((e * A + ag) * T + t) * G + g) (A = 1 for the state
// segment).  The grid-stride loop used to take the six 64-bit runtime divisions / remainders of that decomposition for EVERY group of four
// outputs -- several hundred instructions beside the ~150 of the Philox block and the Box-Muller pair it feeds; round 3 measured the launch at
// 1.37 ms for 3 GB written (2.2 TB/s).  The walker divides ONCE per thread and then advances by the constant stride with carries.
struct ShapeWalk {
    int g, t, ag; long e, r;         // current element: group in the row, time step, agent, env; r = flat row index
    int dg, dt, dag; long de, dr;    // the stride's digits
    int G, T, A;
    __device__ ShapeWalk(long j, long stride, int G_, int T_, int A_) : G(G_), T(T_), A(A_) {
        g = (int)(j % G); r = j / G; t = (int)(r % T); long ea = r / T; ag = (int)(ea % A); e = ea / A;
        dg = (int)(stride % G); dr = stride / G; dt = (int)(dr % T); long dea = dr / T; dag = (int)(dea % A); de = dea / A;
    }
    __device__ __forceinline__ void next() {
        g += dg; r += dr;
        int c = 0;
        if (g >= G) { g -= G; r += 1; c = 1; }
        t += dt + c; c = 0;
        if (t >= T) { t -= T; c = 1; }
        ag += dag + c; c = 0;
        if (ag >= A) { ag -= A; c = 1; }
        e += de + c;
    }
};

__global__ __launch_bounds__(256) void k_shape_fill(int E, int A, int T, int obs_raw, int agent_ids, int Ds, int K, float avail_p,
                                                    unsigned long long seed, long env_offset, long episode,
                                                    float* __restrict__ obs, float* __restrict__ state, uint8_t* __restrict__ avail,
                                                    long obs_ld, long state_ld) {
    const int Do = obs_raw + (agent_ids ? A : 0);
    const int go = (Do + 3) >> 2, gs = (Ds + 3) >> 2, gk = (K + 3) >> 2;  // groups of four per row
    const long n_obs = (long)E * A * T * go, n_state = (long)E * T * gs, n_av = (long)E * A * T * gk;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const bool vo = (obs_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(obs) & 15) == 0) && (long)4 * go <= obs_ld;
    const bool vs = (state_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(state) & 15) == 0) && (long)4 * gs <= state_ld;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x, stride = (long)gridDim.x * 256;
    // the three segments keep the element -> thread assignment of the single grid-stride loop over [obs | state | avail]
    auto first_in = [&](long seg_lo) { return seg_lo <= gid ? gid - seg_lo : ((seg_lo - gid + stride - 1) / stride) * stride + gid - seg_lo; };
    {   // ---- observations: four normals per Philox block, one-hot agent ids behind them
        long j = first_in(0);
        if (j < n_obs) {
            ShapeWalk w(j, stride, go, T, A);
            for (; j < n_obs; j += stride, w.next()) {
                const int g = w.g, t = w.t, ag = w.ag;
                const unsigned long long ge = (unsigned long long)(env_offset + w.e);
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (4 * g < obs_raw) {
                    const N4 nn = normal4(cm_philox4x32((uint32_t)ge, (uint32_t)episode, (uint32_t)(t * A + ag), 0x100u + (uint32_t)g, k0, k1));
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = nn.v[q];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f = 4 * g + q;
                    if (f >= obs_raw) v[q] = (f < Do && f - obs_raw == ag) ? 1.0f : 0.0f;  // one-hot id; zero in the padding columns
                }
                float* o = obs + w.r * obs_ld + 4 * g;
                if (vo) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (4 * g + q < Do) o[q] = v[q];
                }
            }
        }
    }
    {   // ---- global state
        long j = first_in(n_obs);
        if (j < n_state) {
            ShapeWalk w(j, stride, gs, T, 1);
            for (; j < n_state; j += stride, w.next()) {
                const int g = w.g, t = w.t;
                const unsigned long long ge = (unsigned long long)(env_offset + w.e);
                const N4 nn = normal4(cm_philox4x32((uint32_t)ge, (uint32_t)episode, (uint32_t)t, 0x40000000u + (uint32_t)g, k0, k1));
                float* o = state + w.r * state_ld + 4 * g;
                if (vs) *reinterpret_cast<float4*>(o) = make_float4(nn.v[0], 4 * g + 1 < Ds ? nn.v[1] : 0.f, 4 * g + 2 < Ds ? nn.v[2] : 0.f, 4 * g + 3 < Ds ? nn.v[3] : 0.f);
                else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (4 * g + q < Ds) o[q] = nn.v[q];
                }
            }
        }
    }
    {   // ---- availability masks
        long j = first_in(n_obs + n_state);
        if (j < n_av) {
            ShapeWalk w(j, stride, gk, T, A);
            for (; j < n_av; j += stride, w.next()) {
                const int g = w.g, t = w.t, ag = w.ag;
                const unsigned long long ge = (unsigned long long)(env_offset + w.e);
                const cm_u4 ww4 = cm_philox4x32((uint32_t)ge, (uint32_t)episode, (uint32_t)(t * A + ag), 0x80000000u + (uint32_t)g, k0, k1);
                const uint32_t ww[4] = {ww4.x, ww4.y, ww4.z, ww4.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k = 4 * g + q;
                    if (k < K) avail[w.r * K + k] = (k == 0 || cm_u01(ww[q]) < avail_p) ? 1 : 0;  // action 0 is always legal
                }
            }
        }
    }
}

// reward[e][t] = N(0,1) noise + fraction of agents that picked action (t mod K)
__global__ __launch_bounds__(256) void k_shape_reward(int E, int A, int T, int K, unsigned long long seed, long env_offset,
                                                      long episode, const int* __restrict__ action, float* __restrict__ reward) {
    const long n = (long)E * T;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int t = (int)(i % T); const long e = i / T;
        const unsigned long long ge = (unsigned long long)(env_offset + e);
        const cm_u4 w = cm_philox4x32((uint32_t)ge, (uint32_t)episode, (uint32_t)t, 0xC0000000u, (uint32_t)seed, (uint32_t)(seed >> 32));
        int hits = 0;
        for (int a = 0; a < A; ++a) hits += (action[(e * A + a) * (long)T + t] == t % K) ? 1 : 0;
        reward[i] = normal01(w.x, w.y) + (float)hits / (float)A;
    }
}

}  // namespace

extern "C" int cm_shape_env_fill_ld(int E, int A, int T, int obs_raw, int agent_ids, int state_dim, int n_actions, double avail_p,
                                    uint64_t seed, int64_t env_offset, int64_t episode, float* obs, int64_t obs_ld, float* state,
                                    int64_t state_ld, uint8_t* avail, cm_stream_t stream) {
    CM_REQUIRE(E > 0 && A > 0 && T > 0 && obs_raw > 0 && state_dim > 0 && n_actions > 0, "cm_shape_env_fill: bad dims");
    CM_REQUIRE(obs_ld >= obs_raw + (agent_ids ? A : 0) && state_ld >= state_dim, "cm_shape_env_fill: leading dimensions below the widths");
    hipLaunchKernelGGL(k_shape_fill, dim3(2048), dim3(256), 0, (hipStream_t)stream, E, A, T, obs_raw, agent_ids, state_dim, n_actions,
                       (float)avail_p, (unsigned long long)seed, (long)env_offset, (long)episode, obs, state, avail, (long)obs_ld, (long)state_ld);
    CM_CHECK_LAUNCH("cm_shape_env_fill");
    return 0;
}
extern "C" int cm_shape_env_fill(int E, int A, int T, int obs_raw, int agent_ids, int state_dim, int n_actions, double avail_p,
                                 uint64_t seed, int64_t env_offset, int64_t episode, float* obs, float* state, uint8_t* avail,
                                 cm_stream_t stream) {
    return cm_shape_env_fill_ld(E, A, T, obs_raw, agent_ids, state_dim, n_actions, avail_p, seed, env_offset, episode, obs,
                                obs_raw + (agent_ids ? A : 0), state, state_dim, avail, stream);
}

extern "C" int cm_shape_env_reward(int E, int A, int T, int n_actions, uint64_t seed, int64_t env_offset, int64_t episode,
                                   const int32_t* action, float* reward, cm_stream_t stream) {
    CM_REQUIRE(E > 0 && A > 0 && T > 0 && n_actions > 0, "cm_shape_env_reward: bad dims");
    const long n = (long)E * T;
    hipLaunchKernelGGL(k_shape_reward, dim3((int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       E, A, T, n_actions, (unsigned long long)seed, (long)env_offset, (long)episode, action, reward);
    CM_CHECK_LAUNCH("cm_shape_env_reward");
    return 0;
}

extern "C" int cm_synth_env_reset(float* env_state, int E, int A, int agent_ids, uint64_t seed, int64_t env_offset,
                                  int64_t episode, float* obs, float* state, int T, cm_stream_t stream) {
    CM_REQUIRE(E > 0 && A > 0 && A <= 256 && T > 0, "cm_synth_env_reset: bad dims E=%d A=%d T=%d", E, A, T);
    const int epb = 256 / A;
    hipLaunchKernelGGL(k_env_reset, dim3((E + epb - 1) / epb), dim3(256), (size_t)epb * 6 * A * sizeof(float), (hipStream_t)stream,
                       env_state, E, A, agent_ids, (unsigned long long)seed, (long)env_offset, (long)episode, obs, state, T);
    CM_CHECK_LAUNCH("cm_synth_env_reset");
    return 0;
}

extern "C" int cm_synth_env_step(float* env_state, const int32_t* action, int E, int A, int agent_ids, int t, int T,
                                 float* reward, float* obs, float* state, cm_stream_t stream) {
    CM_REQUIRE(E > 0 && A > 0 && A <= 256 && T > 0 && t >= 0 && t < T, "cm_synth_env_step: bad dims E=%d A=%d T=%d t=%d", E, A, T, t);
    const int epb = 256 / A;
    hipLaunchKernelGGL(k_env_step, dim3((E + epb - 1) / epb), dim3(256), (size_t)epb * 6 * A * sizeof(float), (hipStream_t)stream,
                       env_state, action, E, A, agent_ids, t, T, reward, obs, state);
    CM_CHECK_LAUNCH("cm_synth_env_step");
    return 0;
}
