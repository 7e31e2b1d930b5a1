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

#define OPT_THREADS 1024

// one parameter's update; mi / vi are the two state slots (Adam: exp_avg, exp_avg_sq; RMSprop: vi = square_avg; SGD: unused)
__device__ __forceinline__ float cm_opt_step(int kind, float p, float gi, float& mi, float& vi, float lr, float step_size, float beta1,
                                             float beta2, float eps, float weight_decay, float bc2_sqrt) {
    if (kind == CM_OPT_SGD) return p - lr * gi;
    if (kind == CM_OPT_RMSPROP) {
        vi = beta2 * vi + (1.0f - beta2) * gi * gi;
        return p - lr * (gi / (sqrtf(vi) + eps));
    }
    if (kind == CM_OPT_ADAMW) p *= (1.0f - lr * weight_decay);
    mi = beta1 * mi + (1.0f - beta1) * gi;
    vi = beta2 * vi + (1.0f - beta2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    return p - step_size * (mi / denom);
}

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
        params[i] = cm_opt_step(opt_kind, params[i], gi, mi, vi, lr, step_size, beta1, beta2, eps, weight_decay, bc2_sqrt);
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
    float grad_scale, float bc1, float bc2_sqrt, const float* __restrict__ norm_in) {
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
            params[i] = cm_opt_step(opt_kind, pv[k], gi, mi, vi, lr, step_size, beta1, beta2, eps, weight_decay, bc2_sqrt);
            m[i] = mi; v[i] = vi;
        }
    }
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
                           (float)weight_decay, opt_kind, (float)max_norm, (float)grad_scale, (float)bc1, (float)sqrt(bc2), out_norm);
    }
    else
        hipLaunchKernelGGL(k_grad_norm_clip_adam, dim3(1), dim3(OPT_THREADS), 0, (hipStream_t)stream, params, grad_and_stats,
                           exp_avg, exp_avg_sq, (long)n_params, (float)lr, (float)beta1, (float)beta2, (float)eps,
                           (float)weight_decay, opt_kind, (float)max_norm, (float)grad_scale, (float)bc1, (float)sqrt(bc2), out_norm);
    CM_CHECK_LAUNCH("cm_grad_norm_clip_adam");
    return 0;
}
