// cm_common.h -- shared device/host helpers for libcleanmarl_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/cleanmarl_hip.h"

#define CM_WAVE 64

// ---------------------------------------------------------------- error plumbing
void cm_set_error(const char* fmt, ...);
#define CM_FAIL(code, ...) do { cm_set_error(__VA_ARGS__); return (code); } while (0)
#define CM_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) CM_FAIL(-2, "%s: launch failed: %s", name, hipGetErrorString(e_)); } while (0)
#define CM_REQUIRE(cond, ...) do { if (!(cond)) CM_FAIL(-1, __VA_ARGS__); } while (0)

// options set through cm_set_option (cm_api.hip); values are the ints listed there (0 = the default / "auto")
enum { CM_OPTION_MLP_FORMS = 0, CM_OPTION_CRITIC_SCHEDULE, CM_OPTION_GRU_TILE, CM_OPTION_ROLLOUT_TILE, CM_OPTION_MFMA, CM_OPTION_WIDE_SCHEDULE, CM_OPTION_DW0_BATCH, CM_OPTION_DW0_GRID, CM_OPTION_TRAIN_GRID, CM_OPTION_TILE_SPLIT, CM_OPTION_COUNT };
int cm_option(int which);

// fold per-workgroup partial gradient rows and apply the optimiser step in one launch (cm_optim.hip); part2 / isplit: a second partial
// set holding columns [0, isplit) (the split critic's streamed dW0), NULL if none
// peer_tags != NULL (cm_peer.hip): the np1 partial rows are mailbox slots written by peer GPUs; the launch first waits (at most
// peer_timeout_s of wall time; <= 0: 30 s) until the np1 tag words equal peer_seq and reads the rows with system-scope loads; a wait that
// runs out skips the step and writes peer_seq to *peer_status (optional, host-visible)
int cm_launch_reduce_step(const float* part1, int np1, int PS1, const float* part2, int np2, int PS2, int isplit, int64_t n_params,
                          float* grad_and_stats, const cm_opt_step_t* opt, hipStream_t s, const char* who,
                          const unsigned long long* peer_tags = nullptr, unsigned peer_seq = 0, double peer_timeout_s = 0.0,
                          unsigned* peer_status = nullptr);

// launches whose optimiser step is the stand-alone cm_grad_norm_clip_adam (layered schedules): cm_opt_step_t::stats_out is served by a copy
inline void cm_copy_stats_out(const cm_opt_step_t* o, const float* grad_and_stats, int64_t n_params, hipStream_t s) {
    if (o && o->stats_out) (void)hipMemcpyAsync(o->stats_out, grad_and_stats + n_params, CM_NUM_STATS * sizeof(float), hipMemcpyDeviceToDevice, s);
}
unsigned cm_next_step_tag();  // tags of the step's hand-off words: unique per launch within the process, never 0 (cm_optim.hip)
int cm_opt_check(const char* who, int64_t n_params, const cm_opt_step_t* o);

// layered GRU schedule (cm_gru_wide.hip): obs wider than 64 columns and / or 65..256 hidden units
size_t cm_gru_wide_ws_bytes(int64_t R, int chunk_len, int din, int H, int K, int train);
int cm_gru_wide_chunk(const float* obs, const uint8_t* avail, const int32_t* action, const float* logp_old, const float* adv,
                      const int32_t* ep_len, int E, int A, int T, int t0, int t1, int din, int H, int K, const float* params,
                      const float* h_in, float* h_out, double ppo_clip, double entropy_coef, float* grad_and_stats, void* ws,
                      size_t ws_bytes, hipStream_t s, const cm_opt_step_t* opt);
int cm_gru_wide_act(const float* x, int64_t x_stride, const uint8_t* avail, int64_t avail_stride, int64_t rows, int din, int H, int K,
                    const float* params, float* h, uint64_t seed, int64_t row_offset, int t, float eps, int32_t* action, float* logp,
                    int64_t out_stride, void* ws, size_t ws_bytes, hipStream_t s);

// ---------------------------------------------------------------- Philox4x32-10 (counter RNG)
// Keyed by the run seed, counted by (global row, time step, stream id): the draw for a given
// (env, agent, t) is the same no matter how envs are sharded over GPUs.
struct cm_u4 { uint32_t x, y, z, w; };

__host__ __device__ inline cm_u4 cm_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    return cm_u4{c0, c1, c2, c3};
}
// uniform in [0,1) with 24 random bits (exactly representable in fp32)
__host__ __device__ inline float cm_u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

enum { CM_STREAM_ACT = 1, CM_STREAM_ENV_RESET = 2 };  // Philox stream ids (4th counter word)

// Categorical(logits).sample() + log_prob (cleanmarl/mappo_multienvs.py:172-176) by inverse CDF on one uniform u.
// z: K logits already masked with -1e9 (entries <= -5e8 count as unavailable).  One exp per action:
// e_k = exp(z_k - max), s = sum e_k, pick the first available k with u*s < e_0 + ... + e_k.  Shared by every
// act kernel (and mirrored by oracle/sampling.py) so the fused rollout and the per-step path agree.
__device__ __forceinline__ void cm_categorical_sample(const float* z, int K, float u, int* action, float* logp) {
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, z[k]);
    float s = 0.0f, cum = 0.0f;
    int chosen = -1, last = 0;
    if (K <= 8) {  // common case: keep the K exponentials in registers (one expf per action)
        float e[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { e[k] = (k < K) ? expf(z[k] - m) : 0.0f; s += e[k]; }
        const float thr = u * s;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k < K && z[k] > -5e8f) {
                cum += e[k];
                last = k;
                if (chosen < 0 && thr < cum) chosen = k;
            }
        }
    } else {
        for (int k = 0; k < K; ++k) s += expf(z[k] - m);
        const float thr = u * s;
        for (int k = 0; k < K; ++k) {
            if (z[k] > -5e8f) {
                cum += expf(z[k] - m);
                last = k;
                if (chosen < 0 && thr < cum) chosen = k;
            }
        }
    }
    if (chosen < 0) chosen = last;
    *action = chosen;
    *logp = z[chosen] - (m + logf(s));
}

// greedy evaluation (build option --greedy_eval; the reference always samples): first maximal available logit
__device__ __forceinline__ void cm_categorical_greedy(const float* z, int K, int* action, float* logp) {
    float m = -INFINITY;
    int best = 0;
    for (int k = 0; k < K; ++k) if (z[k] > m) { m = z[k]; best = k; }
    float s = 0.0f;
    for (int k = 0; k < K; ++k) s += expf(z[k] - m);
    *action = best;
    *logp = -logf(s);
}

// COMA exploration (cleanmarl/coma_multienvs.py:177-186): probs = (1 - eps) * softmax(z) + eps * avail / n_avail, inverse CDF
// over the available actions in index order; *logp = log(probs[action]).
__device__ __forceinline__ void cm_categorical_sample_eps(const float* z, int K, float u, float eps, int* action, float* logp) {
    float m = -INFINITY;
    int navail = 0;
    for (int k = 0; k < K; ++k) { m = fmaxf(m, z[k]); navail += (z[k] > -5e8f) ? 1 : 0; }
    float s = 0.0f;
    for (int k = 0; k < K; ++k) s += expf(z[k] - m);
    const float a = (1.0f - eps) / s, b = eps / (float)max(navail, 1);
    float cum = 0.0f, pc = 0.0f, plast = 0.0f;
    int chosen = -1, last = 0;
    for (int k = 0; k < K; ++k) {
        if (z[k] > -5e8f) {
            const float p = a * expf(z[k] - m) + b;
            cum += p;
            last = k; plast = p;
            if (chosen < 0 && u < cum) { chosen = k; pc = p; }
        }
    }
    if (chosen < 0) { chosen = last; pc = plast; }
    *action = chosen;
    *logp = logf(pc);
}

// ---------------------------------------------------------------- wave / block reductions
__device__ __forceinline__ float cm_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double cm_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS reads whose ISSUE ORDER is fixed in the source: for kernels that run one wave per SIMD (cm_critic_fused.h, the GRU sweeps), where nothing hides LDS latency unless a read is
// issued a step or two ahead of the MFMAs that consume it -- and the compiler, at the register limit, schedules every ds_read next to
// its use (read, s_waitcnt lgkmcnt(0), 4 MFMAs: measured 39 cycles per 32-cycle MFMA).  The reads are therefore inline asm (kept in
// program order) and waited for by hand: cf_wait<N>(v) = "at most N younger LDS operations still in flight", tied to the value so
// the consuming MFMAs cannot move above it.  lgkmcnt retires LDS operations in order, so LDS operations the compiler adds in between
// only make these waits (and the compiler's own) more conservative.
template <int OFF> __device__ __forceinline__ f32x4 cf_lds128(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N> __device__ __forceinline__ void cf_wait(f32x4& v) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N)); }
template <int N> __device__ __forceinline__ void cf_wait(f32x4& v, f32x4& w) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(v), "+v"(w) : "n"(N)); }
__device__ __forceinline__ unsigned cf_lds_addr(const float* p) { return (unsigned)reinterpret_cast<uintptr_t>(p); }  // low 32 bits of a flat LDS address = LDS offset
// scalar twin and the waits of the 32x32x2 product forms (cm_mlp_kernel.h): 4 + 4 scalars, or one 16-byte and 4 scalars
template <int OFF> __device__ __forceinline__ float cf_lds32(unsigned addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N> __device__ __forceinline__ void cf_wait(float (&a)[4], float (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N));
}
template <int N> __device__ __forceinline__ void cf_wait(f32x4& a, float (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N));
}
