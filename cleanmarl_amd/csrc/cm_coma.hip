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
#include "cm_mlp_split.h"

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
    return split_ws_bytes((long)rows, din, hidden, n_hidden_layers, dout);
}

extern "C" int cm_qcritic_fwd_bwd(const float* x, const int32_t* action, const float* target, const int32_t* ep_len, int E, int A, int T,
                                  int din, int hidden, int n_hidden_layers, int n_actions, const float* params, float* grad_and_stats,
                                  void* ws, size_t ws_bytes, cm_stream_t stream) {
    if (int rc = check_shapes("cm_qcritic_fwd_bwd", din, hidden, n_hidden_layers, n_actions)) return rc;
    CM_REQUIRE(E > 0 && A > 0 && T > 0, "cm_qcritic_fwd_bwd: bad dims E=%d A=%d T=%d", E, A, T);
    const long rows = (long)E * A * T;
    if (int rc = check_rows("cm_qcritic_fwd_bwd", rows)) return rc;
    MlpArgs a = {};
    a.x = x; a.x_stride = din; a.rows = rows; a.din = din; a.H = hidden; a.L = n_hidden_layers; a.dout = n_actions;
    a.params = params; a.action = action; a.ret = target; a.ep_len = ep_len; a.A = A; a.T = T; a.per_agent = 1;
    return run_train<M_QCRITIC>(a, grad_and_stats, ws, ws_bytes, (hipStream_t)stream, "cm_qcritic_fwd_bwd");
}

extern "C" int cm_coma_actor_fwd_bwd(const float* obs, const uint8_t* avail, const int32_t* action, const float* adv,
                                     const int32_t* ep_len, int E, int A, int T, int din, int hidden, int n_hidden_layers, int n_actions,
                                     const float* params, double entropy_coef, float* grad_and_stats, void* ws, size_t ws_bytes,
                                     cm_stream_t stream) {
    if (int rc = check_shapes("cm_coma_actor_fwd_bwd", din, hidden, n_hidden_layers, n_actions)) return rc;
    CM_REQUIRE(E > 0 && A > 0 && T > 0, "cm_coma_actor_fwd_bwd: bad dims E=%d A=%d T=%d", E, A, T);
    const long rows = (long)E * A * T;
    if (int rc = check_rows("cm_coma_actor_fwd_bwd", rows)) return rc;
    MlpArgs a = {};
    a.x = obs; a.x_stride = din; a.rows = rows; a.din = din; a.H = hidden; a.L = n_hidden_layers; a.dout = n_actions;
    a.params = params; a.avail = avail; a.avail_stride = n_actions; a.action = action; a.adv = adv; a.ep_len = ep_len;
    a.A = A; a.T = T; a.per_agent = 1; a.ent_coef = (float)entropy_coef;
    return run_train<M_COMA_ACTOR>(a, grad_and_stats, ws, ws_bytes, (hipStream_t)stream, "cm_coma_actor_fwd_bwd");
}
