// cm_optim.hip -- fused gradient scaling + norm_d + clip_grad_norm_ + Adam / AdamW step.
//
// Replaces cleanmarl/mappo_multienvs.py:572-594 after the backward pass:
//   loss /= b_mask.sum()            -> gradient sums are scaled by grad_scale / N (N read from the stats tail)
//   norm_d([p.grad ...], 2)         -> pre-clip global L2 norm, written to out_norm[0]
//   clip_grad_norm_(max_norm)       -> g *= min(1, max_norm / (norm + 1e-6)) when max_norm > 0
//   optimizer.step()                -> torch.optim.Adam / AdamW update rule; the script looks the class up
//                                      with getattr(optim, args.optimizer), so SGD and RMSprop (torch defaults: no momentum,
//                                      alpha = 0.99 passed as beta2, eps = 1e-8, not centred) are here as well
// The parameter vectors are tiny (8k - 30k floats): a one-workgroup norm pass followed by an element-wise update over a handful of
// workgroups (larger vectors: ONE workgroup of 1024 threads does both passes in a single launch).  No host sync, no atomics, deterministic.
#include "cm_common.h"
#include <atomic>

#define OPT_THREADS 1024

// ---- the optimiser step fused behind a training pass: ONE launch folds the per-workgroup partial gradients, scales by
// grad_scale / N, takes the pre-clip norm and applies the update (a9 tail + a10 + a12; cleanmarl/mappo_multienvs.py:572-594).
//
// Before: k_reduce_partials -> k_grad_norm_small -> k_clip_adam_update, three dependent launches per network and optimiser step
// (a 512-env share of config 3 spent ~75 us per actor epoch in them; the GRU learner takes 39 steps per iteration).  The update of a
// column needs only that column's sum and N, so every workgroup of the reduction applies it at once; the norm (logged, :584-585) is
// finished by workgroup 0 from per-workgroup sums of squares handed over through caller-owned scratch, in a fixed order.
// Only clipping (max_norm > 0, off by default, :62-63) makes the update depend on the global norm: then this launch stops after
// the norm and k_clip_adam_update follows (two launches instead of three).
// The column sums are formed in exactly the order of k_reduce_partials and the update uses the same two roundings as
// k_clip_adam_update, so parameters and moments are bit-identical to the three-launch path (tests/test_hip_parity.py).

#include "cm_step.h"

namespace {

template <bool UPDATE, bool PEER = false>
__global__ __launch_bounds__(STEP_COLS * STEP_WAVES) void k_reduce_step(const StepArgs a) {
    __shared__ float sh[STEP_GROUPS][STEP_COLS];
    bool peer_lost = false;
    if (PEER) {  // the partial rows are mailbox slots: wait until every rank has published this step's (cm_peer.hip)
        // BOUNDED BY WALL TIME (s_memrealtime: the constant 100 MHz reference clock, whatever the shader clock does): a peer that never
        // publishes (a mapping that silently does not reach this GPU, a dead rank) must not leave a kernel spinning for ever on a box
        // nobody can reset -- but a SLOW peer (rank 0 writing a checkpoint, a host-env stall, a first-launch module load, eight
        // processes time-slicing one GPU in the tests) is healthy, so the bound is generous (default 30 s, cm_optimizer_step_peer's
        // timeout_s) and the deadline is checked only every 256 polls.  When it does run out the step is SKIPPED -- parameters, moments and
        // the gradient buffer stay untouched (round 4 poisoned them with NaN, which the next push then spread to every peer) --
        // and peer_seq is written to the caller's status word, which dist.PeerAllReduce reads on the host and raises on.
        bool lost = false;
        if (threadIdx.x < a.np1) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            unsigned polls = 0;
            while (__hip_atomic_load(a.peer_tags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (unsigned long long)a.peer_seq) {
                __builtin_amdgcn_s_sleep(8);
                if ((++polls & 255u) == 0u && __builtin_amdgcn_s_memrealtime() - t0 > a.peer_timeout) { lost = true; break; }
            }
        }
        peer_lost = __syncthreads_or(lost ? 1 : 0) != 0;
        // ONE verdict per launch, workgroup 0's: it travels in the N word that every other workgroup waits for anyway.  A
        // workgroup whose OWN wait ran out while workgroup 0's did not (the last tag arrived between the two deadlines) must not skip its
        // slab while the others update theirs: it waits for that word first -- a valid N means workgroup 0 saw every tag, so the slots are
        // published and this workgroup folds them after all; a NaN N means the whole launch skips.  Only workgroup 0 raises the status word.
        if (peer_lost && blockIdx.x != 0) {
            float N0 = 0.f;
            if (threadIdx.x == 0) N0 = step_wait(a.nword, a.tag);
            sh[0][0] = N0;  // (sh is rewritten by the fold below, behind the barrier that follows)
            __syncthreads();
            N0 = sh[0][0];
            __syncthreads();
            peer_lost = !(N0 == N0);
        }
        if (peer_lost && blockIdx.x == 0 && threadIdx.x == 0 && a.peer_status)
            __hip_atomic_store(a.peer_status, a.peer_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // acquire at system scope, pairing with the pusher's release store of the tag (cm_peer.hip): the slot loads below are relaxed
        // system-scope loads and must not be satisfied from anything older than the tag that certified them.  One fence per
        // workgroup of the OPT-IN peer path only; the default launches (PEER = false) contain no fence (why)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        __syncthreads();
    }
    step_fold_slab<UPDATE, PEER ? 1 : 0>(a, (int)blockIdx.x, (int)gridDim.x, peer_lost, sh);
}

// plain fold of the same two partial sets (vectors beyond the fused launch's scratch)
__global__ __launch_bounds__(STEP_COLS * STEP_GROUPS) void k_reduce_cols(const float* part1, int np1, int PS1, const float* part2, int np2, int PS2,
                                                                         int isplit, int ntot, float* __restrict__ out) {
    __shared__ float sh[STEP_GROUPS][STEP_COLS];
    const int c = threadIdx.x & (STEP_COLS - 1), g = threadIdx.x / STEP_COLS;
    const int i = blockIdx.x * STEP_COLS + c;
    float s = 0.f;
    if (i < ntot) s = (part2 && i < isplit) ? step_colsum(part2, np2, PS2, i, g) : step_colsum(part1, np1, PS1, i, g);
    sh[g][c] = s;
    __syncthreads();
    if (g == 0 && i < ntot) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < STEP_GROUPS; ++q) t += sh[q][c];
        out[i] = t;
    }
}

}  // namespace


__global__ __launch_bounds__(OPT_THREADS) void k_grad_norm_clip_adam(
    float* __restrict__ params, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
    float lr, float beta1, float beta2, float eps, float weight_decay, int opt_kind, float max_norm,
    float grad_scale, float bc1, float bc2_sqrt, float* __restrict__ out_norm) {
    __shared__ float sh[OPT_THREADS / 64];
    __shared__ float s_coef;
    const float N = g[n + CM_STAT_COUNT];
    const float scale = (N > 0.0f) ? grad_scale / N : 0.0f;
    float ss = 0.0f;
    for (long i = threadIdx.x; i < n; i += OPT_THREADS) {
        const float gi = g[i] * scale;
        g[i] = gi;
        ss = fmaf(gi, gi, ss);
    }
    ss = cm_wave_sum(ss);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.0f;
        for (int w = 0; w < OPT_THREADS / 64; ++w) tot += sh[w];
        const float norm = sqrtf(tot);
        out_norm[0] = norm;
        float coef = 1.0f;
        if (max_norm > 0.0f) coef = fminf(max_norm / (norm + 1e-6f), 1.0f);
        s_coef = coef;
    }
    __syncthreads();
    const float coef = s_coef;
    const float step_size = lr / bc1;
    for (long i = threadIdx.x; i < n; i += OPT_THREADS) {
        const float gi = g[i] * coef;
        g[i] = gi;  // post-clip gradient stays readable (what optimizer.step() consumed)
        float mi = m[i], vi = v[i];
        params[i] = cm_opt_apply(opt_kind, params[i], gi, mi, vi, lr, step_size, beta1, beta2, eps, weight_decay, bc2_sqrt);
        m[i] = mi; v[i] = vi;
    }
}

#define OPT_PT 32
// n <= OPT_THREADS * OPT_PT parameters (every network of the hot path): two launches.  ONE compute unit's load / store pipeline bounds
// a single-workgroup kernel (7 accesses per parameter through one L1: 20.7 us per step for 8 k - 29 k parameters), so the norm is a
// one-workgroup read-only pass (each thread keeps its <= OPT_PT values in flight at once, 4.4 us) and the element-wise update is
// spread over n / 1024 workgroups (4.5 us).  The norm travels through out_norm[0]; no atomics, no grid barrier, deterministic.
__global__ __launch_bounds__(OPT_THREADS) void k_grad_norm_small(const float* __restrict__ g, int n, float grad_scale, float* __restrict__ out_norm) {
    __shared__ float sh[OPT_THREADS / 64];
    const float N = g[n + CM_STAT_COUNT];
    const float scale = (N > 0.0f) ? grad_scale / N : 0.0f;
    float gv[OPT_PT];
#pragma unroll
    for (int k = 0; k < OPT_PT; ++k) {
        const int i = threadIdx.x + k * OPT_THREADS;
        gv[k] = (i < n) ? g[i] : 0.0f;
    }
    float ss = 0.0f;
#pragma unroll
    for (int k = 0; k < OPT_PT; ++k) { gv[k] *= scale; ss = fmaf(gv[k], gv[k], ss); }
    ss = cm_wave_sum(ss);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.0f;
        for (int w = 0; w < OPT_THREADS / 64; ++w) tot += sh[w];
        out_norm[0] = sqrtf(tot);
    }
}

#define UPD_THREADS 256
#define UPD_PT 4
__global__ __launch_bounds__(UPD_THREADS) void k_clip_adam_update(
    float* __restrict__ params, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int n,
    float lr, float beta1, float beta2, float eps, float weight_decay, int opt_kind, float max_norm,
    float grad_scale, float bc1, float bc2_sqrt, const float* __restrict__ norm_in, const unsigned long long* __restrict__ skipword, unsigned tag) {
    // the peer step (cm_peer.hip) whose wait for the peers ran out: workgroup 0 of the fold launch in front of this one left {tag, 1} in the
    // scratch's skip word -- nothing may be applied then.  A dedicated word, not "the norm is NaN" (ADVICE r5): a norm that is NaN for real
    // (Inf / NaN gradients, every tag arrived) is NOT a skipped step -- it is applied and poisons the parameters loudly, as on the RCCL path
    if (skipword) { const unsigned long long w = skipword[0]; if ((unsigned)(w >> 32) == tag && (unsigned)w != 0u) return; }
    const float N = g[n + CM_STAT_COUNT];
    const float scale = (N > 0.0f) ? grad_scale / N : 0.0f;
    float coef = 1.0f;
    if (max_norm > 0.0f) coef = fminf(max_norm / (norm_in[0] + 1e-6f), 1.0f);
    const float step_size = lr / bc1;
    const int base = blockIdx.x * (UPD_THREADS * UPD_PT) + threadIdx.x;
    float gv[UPD_PT], pv[UPD_PT], mv[UPD_PT], vv[UPD_PT];
#pragma unroll
    for (int k = 0; k < UPD_PT; ++k) {
        const int i = base + k * UPD_THREADS;
        const bool ok = i < n;
        gv[k] = ok ? g[i] : 0.0f; pv[k] = ok ? params[i] : 0.0f; mv[k] = ok ? m[i] : 0.0f; vv[k] = ok ? v[i] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < UPD_PT; ++k) {
        const int i = base + k * UPD_THREADS;
        if (i < n) {
            const float gi = (gv[k] * scale) * coef;  // same two roundings as the one-launch kernel
            g[i] = gi;
            float mi = mv[k], vi = vv[k];
            params[i] = cm_opt_apply(opt_kind, pv[k], gi, mi, vi, lr, step_size, beta1, beta2, eps, weight_decay, bc2_sqrt);
            m[i] = mi; v[i] = vi;
        }
    }
}

// tags of the hand-off words: unique per launch within the process, never 0 (a zeroed scratch holds tag 0).  A nonce, not state: no
// result depends on its value.
unsigned cm_next_step_tag() {
    static std::atomic<unsigned> tag{0};
    unsigned t;
    do { t = tag.fetch_add(1u, std::memory_order_relaxed) + 1u; } while (t == 0u);
    return t;
}

int cm_opt_check(const char* who, int64_t n_params, const cm_opt_step_t* o) {
    CM_REQUIRE(o && o->params && o->out_norm, "%s: cm_opt_step_t with NULL params / out_norm", who);
    CM_REQUIRE(n_params > 0 && o->step >= 1, "%s: bad n_params=%ld step=%d", who, (long)n_params, (int)o->step);
    CM_REQUIRE(o->opt_kind >= CM_OPT_ADAM && o->opt_kind <= CM_OPT_RMSPROP, "%s: unknown optimiser kind %d", who, (int)o->opt_kind);
    CM_REQUIRE(o->opt_kind == CM_OPT_SGD || (o->exp_avg_sq && (o->opt_kind == CM_OPT_RMSPROP || o->exp_avg)), "%s: optimiser state pointer is NULL", who);
    return 0;
}

extern "C" size_t cm_opt_step_scratch_bytes(void) { return STEP_SCRATCH_BYTES; }

// [part1 | part2] partial rows -> grad_and_stats + optimiser step.  part2 == NULL: every column comes from part1.
int cm_launch_reduce_step(const float* part1, int np1, int PS1, const float* part2, int np2, int PS2, int isplit, int64_t n_params,
                          float* grad_and_stats, const cm_opt_step_t* o, hipStream_t s, const char* who,
                          const unsigned long long* peer_tags, unsigned peer_seq, double peer_timeout_s, unsigned* peer_status) {
    if (int rc = cm_opt_check(who, n_params, o)) return rc;
    const int64_t ntot = n_params + CM_NUM_STATS;
    const int grid = (int)((ntot + STEP_COLS - 1) / STEP_COLS);
    CM_REQUIRE(!peer_tags || (grid <= STEP_MAX_WG && o->scratch), "%s: the peer step needs the fused launch (<= %d parameters, a scratch)", who, STEP_MAX_WG * STEP_COLS);
    const bool small_offsets = (int64_t)np1 * PS1 < (1LL << 30) && (int64_t)np2 * PS2 < (1LL << 30);  // step_colsum_gpw's 32-bit offsets
    CM_REQUIRE(!peer_tags || small_offsets, "%s: mailbox too large for the fused step", who);
    if (grid > STEP_MAX_WG || !o->scratch || !small_offsets) {
        // beyond the scratch's sumsq slots (or no scratch): plain reduction + the stand-alone step
        hipLaunchKernelGGL(k_reduce_cols, dim3(grid), dim3(STEP_COLS * STEP_GROUPS), 0, s, part1, np1, PS1, part2, np2, PS2, isplit, (int)ntot, grad_and_stats);
        CM_CHECK_LAUNCH(who);
        cm_copy_stats_out(o, grad_and_stats, n_params, s);
        return cm_grad_norm_clip_adam(o->params, grad_and_stats, o->exp_avg, o->exp_avg_sq, n_params, o->step, o->lr, o->beta1, o->beta2, o->eps,
                                      o->weight_decay, o->opt_kind, o->max_norm, o->grad_scale, o->out_norm, (cm_stream_t)s);
    }
    CM_REQUIRE((reinterpret_cast<uintptr_t>(o->scratch) & 15) == 0, "%s: cm_opt_step_t scratch must be 16-byte aligned", who);
    StepArgs a = {};
    step_args_fill(a, part1, np1, PS1, part2, np2, PS2, isplit, n_params, grad_and_stats, o);
    a.peer_tags = peer_tags; a.peer_seq = peer_seq; a.peer_status = peer_status;
    a.peer_timeout = (unsigned long long)((peer_timeout_s > 0.0 ? peer_timeout_s : 30.0) * 1e8);  // s_memrealtime ticks at 100 MHz
    if (o->max_norm > 0.0) {
        if (peer_tags) hipLaunchKernelGGL((k_reduce_step<false, true>), dim3(grid), dim3(STEP_COLS * STEP_WAVES), 0, s, a);
        else hipLaunchKernelGGL(k_reduce_step<false>, dim3(grid), dim3(STEP_COLS * STEP_WAVES), 0, s, a);
        const int ugrid = (int)((n_params + UPD_THREADS * UPD_PT - 1) / (UPD_THREADS * UPD_PT));
        hipLaunchKernelGGL(k_clip_adam_update, dim3(ugrid), dim3(UPD_THREADS), 0, s, o->params, grad_and_stats, o->exp_avg, o->exp_avg_sq,
                           (int)n_params, a.lr, a.beta1, a.beta2, a.eps, a.wd, a.kind, (float)o->max_norm, a.grad_scale, a.bc1, a.bc2_sqrt, o->out_norm, peer_tags ? a.nword + 1 : nullptr, a.tag);
    } else {
        if (peer_tags) hipLaunchKernelGGL((k_reduce_step<true, true>), dim3(grid), dim3(STEP_COLS * STEP_WAVES), 0, s, a);
        else hipLaunchKernelGGL(k_reduce_step<true>, dim3(grid), dim3(STEP_COLS * STEP_WAVES), 0, s, a);
    }
    CM_CHECK_LAUNCH(who);
    return 0;
}

extern "C" int cm_optimizer_step(float* grad_and_stats, int64_t n_params, const cm_opt_step_t* opt, cm_stream_t stream) {
    CM_REQUIRE(grad_and_stats, "cm_optimizer_step: grad_and_stats is NULL");
    // the reduced buffer is its own single "partial row" (row stride irrelevant)
    return cm_launch_reduce_step(grad_and_stats, 1, 0, nullptr, 0, 0, 0, n_params, grad_and_stats, opt, (hipStream_t)stream, "cm_optimizer_step");
}

extern "C" int cm_grad_norm_clip_adam(float* params, float* grad_and_stats, float* exp_avg, float* exp_avg_sq,
                                      int64_t n_params, int step, double lr, double beta1, double beta2, double eps,
                                      double weight_decay, int opt_kind, double max_norm, double grad_scale,
                                      float* out_norm, cm_stream_t stream) {
    CM_REQUIRE(n_params > 0 && step >= 1, "cm_grad_norm_clip_adam: bad n_params=%ld step=%d", (long)n_params, step);
    CM_REQUIRE(opt_kind >= CM_OPT_ADAM && opt_kind <= CM_OPT_RMSPROP, "cm_grad_norm_clip_adam: unknown optimiser kind %d", opt_kind);
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    if (n_params <= (int64_t)OPT_THREADS * OPT_PT) {
        hipLaunchKernelGGL(k_grad_norm_small, dim3(1), dim3(OPT_THREADS), 0, (hipStream_t)stream, grad_and_stats, (int)n_params, (float)grad_scale, out_norm);
        const int grid = (int)((n_params + UPD_THREADS * UPD_PT - 1) / (UPD_THREADS * UPD_PT));
        hipLaunchKernelGGL(k_clip_adam_update, dim3(grid), dim3(UPD_THREADS), 0, (hipStream_t)stream, params, grad_and_stats,
                           exp_avg, exp_avg_sq, (int)n_params, (float)lr, (float)beta1, (float)beta2, (float)eps,
                           (float)weight_decay, opt_kind, (float)max_norm, (float)grad_scale, (float)bc1, (float)sqrt(bc2), out_norm, nullptr, 0u);
    }
    else
        hipLaunchKernelGGL(k_grad_norm_clip_adam, dim3(1), dim3(OPT_THREADS), 0, (hipStream_t)stream, params, grad_and_stats,
                           exp_avg, exp_avg_sq, (long)n_params, (float)lr, (float)beta1, (float)beta2, (float)eps,
                           (float)weight_decay, opt_kind, (float)max_norm, (float)grad_scale, (float)bc1, (float)sqrt(bc2), out_norm);
    CM_CHECK_LAUNCH("cm_grad_norm_clip_adam");
    return 0;
}
