// cm_step.h -- device code of the gradient fold + norm + optimiser step (launched by cm_optim.hip: k_reduce_step).  See cm_optim.hip for
// the design.  (Round 5 also ran this code at the tail of the training kernels -- last arrivers fold the partial rows published with sc1
// stores -- and dropped it again: the in-launch fold does the same 16 MB of reads over a slower path, and its 150 bytes of launch
// arguments cost the actor kernel 3.5 %; docs/HISTORY.md.)
#pragma once
#include "cm_common.h"
#include <math.h>

// one parameter's update; mi / vi are the two state slots (Adam: exp_avg, exp_avg_sq; RMSprop: vi = square_avg; SGD: unused)
// Every product / sum is an explicitly rounded operation (no fp contraction): the function is inlined into kernels with different
// surrounding code, and the fused and the stand-alone step must round identically.
__device__ __forceinline__ float cm_opt_apply(int kind, float p, float gi, float& mi, float& vi, float lr, float step_size, float beta1,
                                              float beta2, float eps, float weight_decay, float bc2_sqrt) {
    if (kind == CM_OPT_SGD) return __fsub_rn(p, __fmul_rn(lr, gi));
    if (kind == CM_OPT_RMSPROP) {
        vi = __fadd_rn(__fmul_rn(beta2, vi), __fmul_rn(__fmul_rn(1.0f - beta2, gi), gi));
        return __fsub_rn(p, __fmul_rn(lr, gi / __fadd_rn(sqrtf(vi), eps)));
    }
    if (kind == CM_OPT_ADAMW) p = __fmul_rn(p, 1.0f - __fmul_rn(lr, weight_decay));
    mi = __fadd_rn(__fmul_rn(beta1, mi), __fmul_rn(1.0f - beta1, gi));
    vi = __fadd_rn(__fmul_rn(beta2, vi), __fmul_rn(__fmul_rn(1.0f - beta2, gi), gi));
    const float denom = __fadd_rn(sqrtf(vi) / bc2_sqrt, eps);
    return __fsub_rn(p, __fmul_rn(step_size, mi / denom));
}

namespace {


constexpr int STEP_COLS = 64, STEP_GROUPS = 16;       // the tiling of k_reduce_partials (cm_mlp_kernel.h): same summation order
constexpr int STEP_MAX_WG = 1024;                     // sumsq slots in the scratch: n + 8 <= 65536 columns take the fused launch
constexpr size_t STEP_SCRATCH_BYTES = 64 + STEP_MAX_WG * sizeof(unsigned long long);

struct StepArgs {
    const float* part1; int np1, PS1;  // partial rows holding columns [isplit, ntot) -- always the statistics
    const float* part2; int np2, PS2;  // partial rows holding columns [0, isplit)    -- the split critic's streamed dW0; isplit = 0: none
    int isplit, n, ntot;               // n parameters, ntot = n + CM_NUM_STATS columns
    float* g; float* params; float* m; float* v;
    float lr, bc1, beta1, beta2, eps, wd, grad_scale, bc2_sqrt; int kind;
    float* out_norm; unsigned long long* nword; unsigned long long* slots; unsigned tag;
    float* stats_out;   // optional second destination of the CM_NUM_STATS statistic sums (cm_opt_step_t::stats_out)
    const unsigned long long* peer_tags; unsigned peer_seq;
    unsigned long long peer_timeout;  // wall-clock bound of the wait for the peers' tags, in s_memrealtime ticks (100 MHz)
    unsigned* peer_status;            // optional host-visible word: set to peer_seq by a launch whose wait ran out (the step is then SKIPPED)
};

// sum of column i over the np partial rows, in the order of k_reduce_partials: row group g takes rows g, g + 16, g + 32, ... into two
// alternating accumulators.  All loads of a batch of rows are issued before the first add (the loop of k_reduce_partials keeps two in flight).
template <bool PEER = false>
__device__ __forceinline__ float step_colsum(const float* __restrict__ p, int np, int PS, int i, int g) {
    // STEP_BATCH loads in flight per thread (16 waves per workgroup keep the memory pipe busy); a larger batch only costs registers, and a
    // 1024-thread workgroup at > 64 registers per lane no longer fits beside the critic's persistent workgroups on the other stream
    // (32 in flight = 108 registers: the launch then waited for whole free CUs, 512-env share 1.54 -> 1.72 ms)
    constexpr int STEP_BATCH = 8;
    float s0 = 0.f, s1 = 0.f;
    for (int w0 = g; w0 < np; w0 += STEP_BATCH * STEP_GROUPS) {
        float v[STEP_BATCH];
#pragma unroll
        for (int k = 0; k < STEP_BATCH; ++k) {
            const int w = w0 + k * STEP_GROUPS;
            v[k] = 0.0f;
            if (w < np) v[k] = PEER ? __hip_atomic_load(p + (size_t)w * PS + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : p[(size_t)w * PS + i];
        }
#pragma unroll
        for (int k = 0; k < STEP_BATCH; k += 2) { s0 += v[k]; s1 += v[k + 1]; }
    }
    return s0 + s1;
}

// Cross-workgroup hand-offs WITHOUT fences.  The eight XCDs of the chip have private L2s, so an agent-scope release / acquire
// (__threadfence, acquire loads) compiles to buffer_wbl2 / buffer_inv of a whole L2 -- per workgroup, under a concurrently running
// critic kernel: the first version of this launch was SLOWER than the three it replaced (512-env share 1.54 -> 1.77 ms).  Every
// hand-off here is therefore ONE relaxed 64-bit atomic word {launch tag, fp32 payload} (sc1: performed at the device coherence
// point, no cache maintenance); a word is valid when its tag is this launch's -- nothing is reset, stale words carry older tags.
//   * N = b_mask.sum() is the sum of one column (n + CM_STAT_COUNT) that every workgroup needs before it can scale.  Re-reading
//     that column in every workgroup put np x grid requests on the same few L2 channels; instead workgroup 0 takes the slab that
//     holds it and publishes {tag, N}; the others fold their own columns meanwhile and then wait for that word -- only ever for
//     workgroup 0, which is dispatched before them (the dependence direction of a decoupled look-back scan).
//   * the norm: every workgroup publishes {tag, sum of squares of its slab}; workgroup 0, done with its own slab, polls the slots
//     and adds them in slot order (deterministic).  It waits for workgroups that wait for nothing but its own earlier N word.
__device__ __forceinline__ unsigned long long step_word(unsigned tag, float v) {
    return ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
}
__device__ __forceinline__ float step_wait(const unsigned long long* p, unsigned tag) {
    unsigned long long w;
    while ((unsigned)((w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != tag) __builtin_amdgcn_s_sleep(2);
    return __uint_as_float((unsigned)w);
}

// The fused launch runs on 256-thread workgroups: wave w folds row groups 4w .. 4w+3 (same rows, same alternating accumulators, same final
// order over the 16 groups as step_colsum / k_reduce_partials => bit-identical sums), 4 x 8 loads in flight per thread.  A 1024-thread
// workgroup needs four wave slots of 40 registers on EVERY SIMD at once: beside the persistent workgroups of the other stream's k_mlp
// (2 x 200 registers per SIMD, 112 free) it could not be placed until a whole workgroup retired -- in the two-stream schedule the critic's
// step sat 218 us behind the actor's pass and the actor's 30 - 56 us behind the critic's (kernel trace of the 512-env share, round 3);
// one 64-register wave per SIMD fits beside every kernel of the path.
constexpr int STEP_WAVES = 4, STEP_GPW = STEP_GROUPS / STEP_WAVES;
// host side: the launch arguments of one fold + step (o checked by cm_opt_check)
inline void step_args_fill(StepArgs& a, const float* part1, int np1, int PS1, const float* part2, int np2, int PS2, int isplit, int64_t n_params,
                           float* grad_and_stats, const cm_opt_step_t* o) {
    const double bc1 = 1.0 - pow(o->beta1, (double)o->step), bc2 = 1.0 - pow(o->beta2, (double)o->step);
    a.part1 = part1; a.np1 = np1; a.PS1 = PS1; a.part2 = part2; a.np2 = np2; a.PS2 = PS2; a.isplit = part2 ? isplit : 0;
    a.n = (int)n_params; a.ntot = (int)(n_params + CM_NUM_STATS);
    a.g = grad_and_stats; a.params = o->params; a.m = o->exp_avg; a.v = o->exp_avg_sq;
    a.lr = (float)o->lr; a.bc1 = (float)bc1; a.beta1 = (float)o->beta1; a.beta2 = (float)o->beta2; a.eps = (float)o->eps;
    a.wd = (float)o->weight_decay; a.grad_scale = (float)o->grad_scale; a.bc2_sqrt = (float)sqrt(bc2); a.kind = o->opt_kind;
    a.out_norm = o->out_norm; a.nword = (unsigned long long*)o->scratch; a.slots = a.nword + 8; a.stats_out = o->stats_out;
    a.tag = cm_next_step_tag();
}
// LD: how the partial rows are read -- 0 plain loads (rows written by an earlier launch), 1 system-scope loads (peer mailbox slots),
// 2 agent-scope (sc1) loads (rows published write-through by workgroups of the SAME launch)
template <int LD = 0>
__device__ __forceinline__ void step_colsum_gpw(const float* __restrict__ p, int np, int PS, int i, int w, float (&out)[STEP_GPW]) {
    constexpr int STEP_BATCH = 8;
    float s0[STEP_GPW], s1[STEP_GPW];
#pragma unroll
    for (int j = 0; j < STEP_GPW; ++j) s0[j] = s1[j] = 0.f;
    const unsigned last = (unsigned)(np - 1), col = (unsigned)i;
    for (int r0 = 0; r0 < np; r0 += STEP_BATCH * STEP_GROUPS) {  // a group whose rows are exhausted adds +0.0f: no change
        // unconditional loads of clamped rows (no branch per load: 32 requests leave back to back), masked after they arrived;
        // 32-bit element offsets: one address register per load in flight (cm_launch_reduce_step checks np * PS < 2^30)
        float v[STEP_GPW][STEP_BATCH];
#pragma unroll
        for (int j = 0; j < STEP_GPW; ++j)
#pragma unroll
            for (int k = 0; k < STEP_BATCH; ++k) {
                const unsigned row = (unsigned)(r0 + w * STEP_GPW + j + k * STEP_GROUPS);
                const unsigned off = min(row, last) * (unsigned)PS + col;
                v[j][k] = LD == 1 ? __hip_atomic_load(p + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                          : (LD == 2 ? __hip_atomic_load(p + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p[off]);
            }
#pragma unroll
        for (int j = 0; j < STEP_GPW; ++j)
#pragma unroll
            for (int k = 0; k < STEP_BATCH; ++k)
                if ((unsigned)(r0 + w * STEP_GPW + j + k * STEP_GROUPS) > last) v[j][k] = 0.0f;
#pragma unroll
        for (int j = 0; j < STEP_GPW; ++j)
#pragma unroll
            for (int k = 0; k < STEP_BATCH; k += 2) { s0[j] += v[j][k]; s1[j] += v[j][k + 1]; }
    }
#pragma unroll
    for (int j = 0; j < STEP_GPW; ++j) out[j] = s0[j] + s1[j];
}

// One 64-column slab of the fold + step: slab order index `bid` of `nb` (bid 0 takes the slab holding N and finishes the norm), run by one
// 256-thread workgroup; sh = 16 x 64 floats of LDS.  The body of k_reduce_step (cm_optim.hip).
// the pre-clip norm from the slabs' sums of squares, in slot order (wave 0 of the workgroup that took slab 0; it waits for workgroups that
// wait for nothing but its own earlier N word)
__device__ __forceinline__ void step_norm(const StepArgs& a, int nb, bool skip) {
    const int c = threadIdx.x & (STEP_COLS - 1);
    float tot = 0.f;
    for (int j = c; j < nb; j += STEP_COLS) tot += step_wait(a.slots + j, a.tag);
    tot = cm_wave_sum(tot);
    if (c == 0) {
        a.out_norm[0] = skip ? __builtin_nanf("") : sqrtf(tot);  // the logged norm of a skipped step is NaN (the status word is the error channel)
        // the launch's verdict for the clipping update that follows it (k_clip_adam_update): {tag, 1} = skipped, {tag, 0} = apply
        __hip_atomic_store(a.nword + 1, ((unsigned long long)a.tag << 32) | (skip ? 1ull : 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// collect: workgroup 0 finishes the norm right behind its slab (one slab per workgroup); false: the caller does it after its last slab
template <bool UPDATE, int LD>
__device__ __forceinline__ void step_fold_slab(const StepArgs& a, int bid, int nb, bool peer_lost, float (*sh)[STEP_COLS], bool collect = true) {
    const int c = threadIdx.x & (STEP_COLS - 1), g = threadIdx.x / STEP_COLS;  // g: wave of the workgroup
    const int icnt = a.n + CM_STAT_COUNT, slab_n = icnt / STEP_COLS;
    const int slab = bid == 0 ? slab_n : (bid <= slab_n ? bid - 1 : bid);
    const int i = slab * STEP_COLS + c;
    float s[STEP_GPW];
#pragma unroll
    for (int j = 0; j < STEP_GPW; ++j) s[j] = 0.f;
    if (i < a.ntot) {
        if (i < a.isplit) step_colsum_gpw<LD == 1 ? 0 : LD>(a.part2, a.np2, a.PS2, i, g, s);
        else step_colsum_gpw<LD>(a.part1, a.np1, a.PS1, i, g, s);
    }
#pragma unroll
    for (int j = 0; j < STEP_GPW; ++j) sh[g * STEP_GPW + j][c] = s[j];
    __syncthreads();
    if (g != 0) return;  // wave 0 finishes its 64 columns (a caller that folds several slabs re-joins the waves with its own barrier)
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < STEP_GROUPS; ++q) t += sh[q][c];
    float N;
    if (bid == 0) {
        N = __shfl(t, icnt - slab_n * STEP_COLS, 64);
        // a lost step is skipped by EVERY workgroup: workgroup 0 hands its verdict on in the N word it publishes anyway (NaN = skip)
        if ((LD == 1) && peer_lost) N = __builtin_nanf("");
        if (c == 0) __hip_atomic_store(a.nword, step_word(a.tag, N), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        N = 0.f;
        if (c == 0) N = step_wait(a.nword, a.tag);
        N = __shfl(N, 0, 64);
    }
    const bool skip = (LD == 1) && (peer_lost || !(N == N));  // the hand-off words below are still published: nobody may wait for a workgroup that gave up
    const float scale = (N > 0.0f) ? a.grad_scale / N : 0.0f;
    const float step_size = a.lr / a.bc1;
    float ss = 0.f;
    if (skip) {
        ss = 0.f;  // nothing is read back, nothing is written: parameters, moments and the gradient buffer keep their values
    } else if (i < a.n) {
        const float gi = __fmul_rn(t, scale);
        ss = __fmul_rn(gi, gi);
        if (UPDATE) {
            a.g[i] = gi;  // what optimizer.step() consumed stays readable
            float mi = a.m ? a.m[i] : 0.0f, vi = a.v ? a.v[i] : 0.0f;
            a.params[i] = cm_opt_apply(a.kind, a.params[i], gi, mi, vi, a.lr, step_size, a.beta1, a.beta2, a.eps, a.wd, a.bc2_sqrt);
            if (a.m) a.m[i] = mi;
            if (a.v) a.v[i] = vi;
        } else {
            a.g[i] = t;   // k_clip_adam_update scales and clips
        }
    } else if (i < a.ntot) {
        a.g[i] = t;       // statistics: un-normalised sums
        if (a.stats_out) a.stats_out[i - a.n] = t;
    }
    ss = cm_wave_sum(ss);
    if (c == 0) __hip_atomic_store(a.slots + bid, step_word(a.tag, ss), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (bid != 0 || !collect) return;
    step_norm(a, nb, skip);
}

}  // namespace
