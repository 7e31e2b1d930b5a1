// cm_gru_wide.hip -- the LAYERED schedule of the GRU actor: observations wider than 64 columns and / or 65..256 hidden units.
//
// The reference's recurrent Actor takes any input_dim / hidden_dim (cleanmarl/mappo_lstm_multienvs.py:162-184: fc1 -> ReLU -> GRUCell
// -> ReLU -> fc2), and the env families the scripts are written for report observations wider than the 64 columns the fused sweeps
// of cm_gru.hip / cm_gru_v2.h hold in registers and LDS (SMAClite ~100+, a 10-agent MPE 70).  Such shapes run here, layer by layer on
// the kernels of the layered MLP schedule (cm_mlp_wide.h), activations in the caller's workspace:
//   everything that does not depend on h is batched over the whole TBPTT chunk ([chunk steps x sequences] rows, time-major):
//       x1 = relu(fc1(obs)),  gi = W_ih x1 + b_ih,  after the recurrence: logits = fc2(relu(h')), PPO head, dW2 / dh_head,
//       and after the reverse recurrence: dW_hh = dGh^T h_prev, dW_ih = dGi^T x1, dx1 = dGi W_ih, dfc1
//   the recurrence itself is one small GEMM group + one element-wise launch per step, forward and backward:
//       gh = W_hh h + b_hh;  r, z = sigma(gi + gh);  n = tanh(gi_n + r gh_n);  h' = (1 - z) n + z h        (:171-174, nn.GRUCell)
// Exact fp32 MFMA like every other path; the same statistics / gradient layout as the fused sweeps, so GRUPPOLearner does not know
// which schedule ran.  HBM-bound and launch-bound by design: it exists so that no shape the reference accepts is refused.
#include "cm_mlp_wide.h"

namespace {

struct GwOff { int W1, b1, Wih, Whh, bih, bhh, W2, b2, P; };
inline GwOff gw_offsets(int din, int H, int K) {  // torch parameters() order of fc1, GRUCell, fc2 (include/cleanmarl_hip.h)
    GwOff o;
    o.W1 = 0; o.b1 = H * din; o.Wih = o.b1 + H; o.Whh = o.Wih + 3 * H * H; o.bih = o.Whh + 3 * H * H;
    o.bhh = o.bih + 3 * H; o.W2 = o.bhh + 3 * H; o.b2 = o.W2 + K * H; o.P = o.b2 + K;
    return o;
}

__device__ __forceinline__ float gw_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float gw_tanh(float x) {
    const float e = __expf(-2.0f * fabsf(x));
    return copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), x);
}

// XO[(s, r)][0..dl) = obs[r][t0 + s][0..din), zero padded to dl columns (time-major rows: one GEMM per chunk instead of one per step)
__global__ void k_gw_gather_obs(const float* __restrict__ obs, long R, int T, int t0, int CL, int din, int dl, float* __restrict__ xo) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)CL * R * dl) return;
    const long row = i / dl; const int c = (int)(i - row * dl);
    const long s = row / R, r = row - s * R;
    xo[i] = c < din ? obs[(r * T + t0 + s) * din + c] : 0.0f;
}
// h0[r][0..Hs) = h_in[r][0..H) (zeros when h_in == NULL: the reference's h = None), padding columns zero
__global__ void k_gw_init_h(const float* __restrict__ h_in, long R, int H, int Hs, float* __restrict__ h0) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * Hs) return;
    const long r = i / Hs; const int c = (int)(i - r * Hs);
    h0[i] = (h_in && c < H) ? h_in[r * H + c] : 0.0f;
}
// one recurrent step, element-wise: gi / gh [R][3 Hs] (r | z | n blocks), hprev [R][Hs] -> saved r | z | n, gh_n, h', relu(h')
__global__ void k_gw_gates_fwd(const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ hprev, long R, int H, int Hs,
                               float* __restrict__ rzn, float* __restrict__ ghn, float* __restrict__ hnew, float* __restrict__ hrelu,
                               float* __restrict__ h_out /* [R][H] or NULL */) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * Hs) return;
    const long r = i / Hs; const int c = (int)(i - r * Hs);
    float rr = 0.f, zz = 0.f, nn = 0.f, gn = 0.f, hn = 0.f;
    if (c < H) {
        const long b = r * 3 * Hs + c;
        rr = gw_sigmoid(gi[b] + gh[b]);
        zz = gw_sigmoid(gi[b + Hs] + gh[b + Hs]);
        gn = gh[b + 2 * Hs];
        nn = gw_tanh(gi[b + 2 * Hs] + rr * gn);
        hn = (1.0f - zz) * nn + zz * hprev[i];
        if (h_out) h_out[r * H + c] = hn;
    }
    if (rzn) { const long b = r * 3 * Hs + c; rzn[b] = rr; rzn[b + Hs] = zz; rzn[b + 2 * Hs] = nn; ghn[i] = gn; }
    hnew[i] = hn;
    if (hrelu) hrelu[i] = fmaxf(hn, 0.0f);
}
// reverse step: dh = dh_head[s] + (dh_rec + dh z of step s + 1); writes dGi / dGh [R][3 Hs] and this step's dh z
__global__ void k_gw_gates_bwd(const float* __restrict__ dh_head, const float* __restrict__ dh_rec, const float* __restrict__ dhz_in,
                               const float* __restrict__ rzn, const float* __restrict__ ghn, const float* __restrict__ hprev, long R, int H, int Hs,
                               float* __restrict__ dgi, float* __restrict__ dgh, float* __restrict__ dhz_out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * Hs) return;
    const long r = i / Hs; const int c = (int)(i - r * Hs);
    const long b = r * 3 * Hs + c;
    float d_r = 0.f, d_z = 0.f, d_n = 0.f, d_gn = 0.f, dz_out = 0.f;
    if (c < H) {
        float dh = dh_head[i];
        if (dh_rec) dh += dh_rec[i] + dhz_in[i];
        const float rr = rzn[b], zz = rzn[b + Hs], nn = rzn[b + 2 * Hs], gn = ghn[i];
        const float dn = dh * (1.0f - zz);
        const float dz = dh * (hprev[i] - nn);
        const float dan = dn * (1.0f - nn * nn);
        d_n = dan;                     // d gi_n
        d_gn = dan * rr;               // d gh_n
        d_r = dan * gn * rr * (1.0f - rr);
        d_z = dz * zz * (1.0f - zz);
        dz_out = dh * zz;
    }
    dgi[b] = d_r; dgi[b + Hs] = d_z; dgi[b + 2 * Hs] = d_n;
    dgh[b] = d_r; dgh[b + Hs] = d_z; dgh[b + 2 * Hs] = d_gn;
    dhz_out[i] = dz_out;
}
// Wt[n][g Hs + j] = W[(g H + j)][n] for the three gate blocks of a [3H][H] matrix (zero padding): the right-hand side of
// dX = [dG_r | dG_z | dG_n] W as ONE GEMM with a 3 Hs-wide contraction
__global__ void k_gw_transpose3(const float* __restrict__ W, int H, int Hs, float* __restrict__ Wt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * 3 * Hs) return;
    const int n = i / (3 * Hs), k = i - n * 3 * Hs, g = k / Hs, j = k - g * Hs;
    Wt[i] = j < H ? W[(long)(g * H + j) * H + n] : 0.0f;
}

struct GwLossArgs {
    const uint8_t* avail; const int* action; const float* logp_old; const float* adv; const int* ep_len;
    long R; int A, T, t0, CL, K; float clip_lo, clip_hi, clip_eps, ent_coef;
};
// PPO head of the chunk (formulas of k_wide_loss<M_ACTOR> / the fused epilogue; cleanmarl/mappo_lstm_multienvs.py:580-607): one thread
// per (step, sequence); logits [CL R][KCAP] -> d(loss)/d(logits) in place, statistics as per-workgroup partials (KCAP = wide_kw(K): 32 or 64)
template <int KCAP>
__global__ __launch_bounds__(NTHREADS) void k_gw_loss(const GwLossArgs a, float* __restrict__ out, float* __restrict__ partial) {
    __shared__ float red[6][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float invA = 1.0f / (float)a.A;
    float st_pg = 0.f, st_ent = 0.f, st_kl = 0.f, st_clip = 0.f, st_cnt = 0.f;
    const long rows = (long)a.CL * a.R;
    for (long row = (long)blockIdx.x * NTHREADS + tid; row < rows; row += (long)gridDim.x * NTHREADS) {
        const long s = row / a.R, seq = row - s * a.R;
        const int t = a.t0 + (int)s;
        const int e = (int)(seq / a.A), ag = (int)(seq - (long)e * a.A);
        const long idx = seq * a.T + t;
        const bool valid = t < a.ep_len[e];
        float* z = out + row * KCAP;
        const int act = a.action[idx];
        float zr[KCAP], p[KCAP];
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < KCAP; ++k) {
            zr[k] = -INFINITY;
            if (k < a.K) zr[k] = a.avail[idx * a.K + k] ? z[k] : -1e9f;  // masked_fill(~avail, -1e9), :182
            m = fmaxf(m, zr[k]);
        }
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < KCAP; ++k) { p[k] = k < a.K ? expf(zr[k] - m) : 0.0f; sum += p[k]; }
        const float rs = 1.0f / sum, lse = m + logf(sum), advv = a.adv[idx];
        float ent = 0.0f, lpa = 0.0f;
#pragma unroll
        for (int k = 0; k < KCAP; ++k)
            if (k < a.K) {
                const float lp = zr[k] - lse;
                p[k] *= rs;
                ent -= p[k] * lp;
                if (k == act) lpa = lp;
            }
        const float log_ratio = lpa - a.logp_old[idx];
        const float ratio = expf(log_ratio);
        const float pg1 = advv * ratio;
        const float pg2 = advv * fminf(fmaxf(ratio, a.clip_lo), a.clip_hi);
        const bool inr = (ratio >= a.clip_lo) && (ratio <= a.clip_hi);
        float g;  // d min(pg1, pg2) / d ratio with torch's tie rule
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
        for (int k = 0; k < KCAP; ++k) {
            float d = 0.0f;
            if (k < a.K) {
                const float lp = zr[k] - lse;
                d = invA * (-gr * ((k == act ? 1.0f : 0.0f) - p[k]) + a.ent_coef * p[k] * (lp + ent));
                if (!valid || zr[k] <= -5e8f) d = 0.0f;
            }
            z[k] = d;
        }
    }
    float sv[6] = {st_pg, st_ent, st_kl, st_clip, 0.0f, st_cnt};
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const float v = cm_wave_sum(sv[i]);
        if (lane == 0) red[i][wave] = v;
    }
    __syncthreads();
    if (tid < CM_NUM_STATS) partial[(long)blockIdx.x * CM_NUM_STATS + tid] = tid < 6 ? red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3] : 0.0f;
}

inline unsigned gw_blocks(long n) { return (unsigned)((n + 255) / 256); }
inline size_t gw_al(size_t x) { return (x + 63) / 64 * 64; }

// workspace carve of one chunk (floats)
struct GwWs { size_t xo, x1, gi, gh, rzn, ghn, hall, hr, out, dhh, dgi, dgh, rec, dhz, dz1, wthh, wtih, wt2, part, total; int Hs, dl; };
inline GwWs gw_ws(long R, int CL, int din, int H, int K, bool train) {
    GwWs w; size_t p = 0;
    w.Hs = wide_hs(H); w.dl = (din + 3) / 4 * 4;
    const size_t SR = (size_t)CL * R, Hs = w.Hs;
    w.xo = p; p += gw_al(SR * w.dl);
    w.x1 = p; p += gw_al(SR * Hs);
    w.gi = p; p += gw_al(SR * 3 * Hs);
    w.gh = p; p += gw_al((size_t)R * 3 * Hs);
    w.hall = p; p += gw_al((SR + R) * Hs);
    w.hr = p; p += gw_al(SR * Hs);
    w.out = p; p += gw_al(SR * wide_kw(K));
    w.rzn = w.ghn = w.dhh = w.dgi = w.dgh = w.rec = w.dhz = w.dz1 = w.wthh = w.wtih = w.wt2 = w.part = p;
    if (train) {
        w.rzn = p; p += gw_al(SR * 3 * Hs);
        w.ghn = p; p += gw_al(SR * Hs);
        w.dhh = p; p += gw_al(SR * Hs);
        w.dgi = p; p += gw_al(SR * 3 * Hs);
        w.dgh = p; p += gw_al(SR * 3 * Hs);
        w.rec = p; p += gw_al((size_t)R * Hs);
        w.dhz = p; p += 2 * gw_al((size_t)R * Hs);
        w.dz1 = p; p += gw_al(SR * Hs);
        w.wthh = p; p += gw_al((size_t)H * 3 * Hs);
        w.wtih = p; p += gw_al((size_t)H * 3 * Hs);
        w.wt2 = p; p += gw_al((size_t)H * wide_kw(K));
        w.part = p;
        const size_t kmax = (size_t)max(w.dl, (int)Hs);
        p += max((size_t)DW0_GRID * 64 * kmax, max((size_t)CS_GRID * WIDE_HMAX, (size_t)LOSS_GRID * CM_NUM_STATS));
    }
    w.total = p;
    return w;
}

inline int gw_check(const char* who, int din, int H, int K) {
    CM_REQUIRE(din > 0 && H > 0 && K > 0, "%s: bad dims din=%d H=%d K=%d", who, din, H, K);
    CM_REQUIRE(H <= WIDE_HMAX, "%s: hidden_dim=%d > %d is not supported by this build", who, H, WIDE_HMAX);
    CM_REQUIRE(K <= KWMAX, "%s: n_actions=%d > %d is not supported by this build", who, K, KWMAX);
    return 0;
}

// gh[R][3 Hs] = h W_hh^T + b_hh, gate by gate (a GEMM tile is at most 256 columns wide)
inline void gw_gates_gemm(const float* X, long ldx, long rows, int H, int Hs, const float* W, const float* b, float* Y, hipStream_t s) {
    for (int g = 0; g < 3; ++g)
        wide_gemm<EPI_BIAS>(X, ldx, rows, H, W + (size_t)g * H * H, H, H, b + g * H, nullptr, 0, nullptr, 0, Y + g * Hs, 3 * Hs, Hs, s);
}

}  // namespace

size_t cm_gru_wide_ws_bytes(int64_t R, int chunk_len, int din, int H, int K, int train) {
    return gw_ws(R, chunk_len, din, H, K, train != 0).total * sizeof(float);
}

// TBPTT chunk [t0, t1): forward + backward through time + (opt != NULL) the actor's optimiser step
int cm_gru_wide_chunk(const float* obs, const uint8_t* avail, const int32_t* action, const float* logp_old, const float* adv,
                      const int32_t* ep_len, int E, int A, int T, int t0, int t1, int din, int H, int K, const float* params,
                      const float* h_in, float* h_out, double ppo_clip, double entropy_coef, float* grad_and_stats, void* ws,
                      size_t ws_bytes, hipStream_t s, const cm_opt_step_t* opt) {
    const char* who = "cm_gru_actor_chunk_fwd_bwd";
    if (int rc = gw_check(who, din, H, K)) return rc;
    const long R = (long)E * A;
    const int CL = t1 - t0;
    const GwWs w = gw_ws(R, CL, din, H, K, true);
    CM_REQUIRE(ws && ws_bytes >= w.total * sizeof(float), "%s: workspace too small (%zu < %zu)", who, ws_bytes, w.total * sizeof(float));
    CM_REQUIRE((long)CL * R < (1L << 31), "%s: %ld chunk rows exceed the 2^31 row limit of one launch", who, (long)CL * R);
    float* f = (float*)ws;
    const GwOff off = gw_offsets(din, H, K);
    const int Hs = w.Hs, dl = w.dl;
    const long SR = (long)CL * R;
    float *xo = f + w.xo, *x1 = f + w.x1, *gi = f + w.gi, *gh = f + w.gh, *rzn = f + w.rzn, *ghn = f + w.ghn, *hall = f + w.hall, *hr = f + w.hr,
          *out = f + w.out, *dhh = f + w.dhh, *dgi = f + w.dgi, *dgh = f + w.dgh, *rec = f + w.rec, *dz1 = f + w.dz1, *part = f + w.part;
    float* dhz[2] = {f + w.dhz, f + w.dhz + gw_al((size_t)R * Hs)};
    float* g = grad_and_stats;
    // ---------------- forward: what does not depend on h, batched over the chunk
    hipLaunchKernelGGL(k_gw_gather_obs, dim3(gw_blocks(SR * dl)), dim3(256), 0, s, obs, R, T, t0, CL, din, dl, xo);
    hipLaunchKernelGGL(k_gw_init_h, dim3(gw_blocks(R * Hs)), dim3(256), 0, s, h_in, R, H, Hs, hall);
    wide_gemm<EPI_BIAS_RELU>(xo, dl, SR, din, params + off.W1, din, H, params + off.b1, nullptr, 0, nullptr, 0, x1, Hs, Hs, s);
    gw_gates_gemm(x1, Hs, SR, H, Hs, params + off.Wih, params + off.bih, gi, s);
    // ---------------- the recurrence
    for (int st = 0; st < CL; ++st) {
        const float* hp = hall + (size_t)st * R * Hs;
        gw_gates_gemm(hp, Hs, R, H, Hs, params + off.Whh, params + off.bhh, gh, s);
        hipLaunchKernelGGL(k_gw_gates_fwd, dim3(gw_blocks(R * Hs)), dim3(256), 0, s, gi + (size_t)st * R * 3 * Hs, gh, hp, R, H, Hs,
                           rzn + (size_t)st * R * 3 * Hs, ghn + (size_t)st * R * Hs, hall + (size_t)(st + 1) * R * Hs, hr + (size_t)st * R * Hs,
                           st == CL - 1 ? h_out : nullptr);
    }
    CM_CHECK_LAUNCH(who);
    // ---------------- head over the chunk: logits -> PPO loss -> dlogits, statistics
    const int kw = wide_kw(K);
    wide_gemm<EPI_BIAS>(hr, Hs, SR, H, params + off.W2, H, K, params + off.b2, nullptr, 0, nullptr, 0, out, kw, kw, s);
    {
        GwLossArgs la = {avail, action, logp_old, adv, ep_len, R, A, T, t0, CL, K, (float)(1.0 - ppo_clip), (float)(1.0 + ppo_clip), (float)ppo_clip,
                         (float)entropy_coef};
        const int grid = (int)min((SR + NTHREADS - 1) / NTHREADS, (long)LOSS_GRID);
        if (kw == KMAX) hipLaunchKernelGGL(k_gw_loss<KMAX>, dim3(grid), dim3(NTHREADS), 0, s, la, out, part);
        else hipLaunchKernelGGL(k_gw_loss<KWMAX>, dim3(grid), dim3(NTHREADS), 0, s, la, out, part);
        hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(RED_COLS * RED_GROUPS), 0, s, part, grid, CM_NUM_STATS, 0, CM_NUM_STATS, g + off.P);
        CM_CHECK_LAUNCH(who);
    }
    if (int rc = stream_dw<true>(out, hr, SR, H, K, part, g + off.W2, s, who, kw, Hs)) return rc;
    wide_colsum(out, kw, SR, K, part, g + off.b2, s);
    {   // dh_head = (dlogits W2) .* (h' > 0)
        const int ldt = (K + 3) / 4 * 4;
        hipLaunchKernelGGL(k_wide_transpose, dim3((H * ldt + 255) / 256), dim3(256), 0, s, params + off.W2, K, H, f + w.wt2, ldt);
        wide_gemm<EPI_GATE>(out, kw, SR, ldt, f + w.wt2, ldt, H, nullptr, nullptr, 0, hr, Hs, dhh, Hs, Hs, s);
    }
    // ---------------- reverse recurrence
    hipLaunchKernelGGL(k_gw_transpose3, dim3((H * 3 * Hs + 255) / 256), dim3(256), 0, s, params + off.Whh, H, Hs, f + w.wthh);
    hipLaunchKernelGGL(k_gw_transpose3, dim3((H * 3 * Hs + 255) / 256), dim3(256), 0, s, params + off.Wih, H, Hs, f + w.wtih);
    for (int st = CL - 1; st >= 0; --st) {
        const bool last = st == CL - 1;
        hipLaunchKernelGGL(k_gw_gates_bwd, dim3(gw_blocks(R * Hs)), dim3(256), 0, s, dhh + (size_t)st * R * Hs, last ? nullptr : rec,
                           last ? nullptr : dhz[(st + 1) & 1], rzn + (size_t)st * R * 3 * Hs, ghn + (size_t)st * R * Hs, hall + (size_t)st * R * Hs,
                           R, H, Hs, dgi + (size_t)st * R * 3 * Hs, dgh + (size_t)st * R * 3 * Hs, dhz[st & 1]);
        if (st > 0)  // dh_rec = [dGh_r | dGh_z | dGh_n] W_hh: one GEMM with a 3 Hs-wide contraction
            wide_gemm<EPI_NONE>(dgh + (size_t)st * R * 3 * Hs, 3 * Hs, R, 3 * Hs, f + w.wthh, 3 * Hs, H, nullptr, nullptr, 0, nullptr, 0, rec, Hs, Hs, s);
    }
    CM_CHECK_LAUNCH(who);
    // ---------------- weight gradients over the whole chunk
    for (int gt = 0; gt < 3; ++gt) {
        for (int n0 = 0; n0 < H; n0 += 64) {
            const int nh = min(64, H - n0);
            if (int rc = stream_dw<true>(dgh + gt * Hs + n0, hall, SR, H, nh, part, g + off.Whh + ((size_t)gt * H + n0) * H, s, who, 3 * Hs, Hs)) return rc;
            if (int rc = stream_dw<true>(dgi + gt * Hs + n0, x1, SR, H, nh, part, g + off.Wih + ((size_t)gt * H + n0) * H, s, who, 3 * Hs, Hs)) return rc;
        }
        wide_colsum(dgh + gt * Hs, 3 * Hs, SR, H, part, g + off.bhh + gt * H, s);
        wide_colsum(dgi + gt * Hs, 3 * Hs, SR, H, part, g + off.bih + gt * H, s);
    }
    // dz1 = (dGi W_ih) .* (x1 > 0); its column sums are db1 (accumulated in the GEMM's epilogue)
    wide_gemm<EPI_GATE>(dgi, 3 * Hs, SR, 3 * Hs, f + w.wtih, 3 * Hs, H, nullptr, nullptr, 0, x1, Hs, dz1, Hs, Hs, s, part, g + off.b1);
    for (int n0 = 0; n0 < H; n0 += 64)
        if (int rc = stream_dw<true>(dz1 + n0, xo, SR, din, min(64, H - n0), part, g + off.W1 + (size_t)n0 * din, s, who, Hs, dl)) return rc;
    CM_CHECK_LAUNCH(who);
    if (opt)  // the reduced buffer is its own single partial row (cm_optimizer_step)
        return cm_launch_reduce_step(g, 1, 0, nullptr, 0, 0, 0, off.P, g, opt, s, who);
    return 0;
}

// one rollout step: h updated in place, action / log-prob sampled with the Philox keying of every other act kernel (eps < 0: greedy)
int cm_gru_wide_act(const float* x, int64_t x_stride, const uint8_t* avail, int64_t avail_stride, int64_t rows, int din, int H, int K,
                    const float* params, float* h, uint64_t seed, int64_t row_offset, int t, float eps, int32_t* action, float* logp,
                    int64_t out_stride, void* ws, size_t ws_bytes, hipStream_t s) {
    const char* who = "cm_gru_policy_act";
    if (int rc = gw_check(who, din, H, K)) return rc;
    if (rows <= 0) return 0;
    const GwWs w = gw_ws(rows, 1, din, H, K, false);
    CM_REQUIRE(ws && ws_bytes >= w.total * sizeof(float), "%s: workspace too small (%zu < %zu): size it with cm_gru_policy_act_workspace_bytes", who,
               ws_bytes, w.total * sizeof(float));
    float* f = (float*)ws;
    const GwOff off = gw_offsets(din, H, K);
    const int Hs = w.Hs;
    float *x1 = f + w.x1, *gi = f + w.gi, *gh = f + w.gh, *hall = f + w.hall, *hr = f + w.hr, *out = f + w.out;
    hipLaunchKernelGGL(k_gw_init_h, dim3(gw_blocks(rows * Hs)), dim3(256), 0, s, h, rows, H, Hs, hall);
    wide_gemm<EPI_BIAS_RELU>(x, x_stride, rows, din, params + off.W1, din, H, params + off.b1, nullptr, 0, nullptr, 0, x1, Hs, Hs, s);
    gw_gates_gemm(x1, Hs, rows, H, Hs, params + off.Wih, params + off.bih, gi, s);
    gw_gates_gemm(hall, Hs, rows, H, Hs, params + off.Whh, params + off.bhh, gh, s);
    hipLaunchKernelGGL(k_gw_gates_fwd, dim3(gw_blocks(rows * Hs)), dim3(256), 0, s, gi, gh, hall, rows, H, Hs, nullptr, nullptr,
                       hall + (size_t)rows * Hs, hr, h);
    wide_gemm<EPI_BIAS>(hr, Hs, rows, H, params + off.W2, H, K, params + off.b2, avail, avail_stride, nullptr, 0, out, wide_kw(K), wide_kw(K), s);
    hipLaunchKernelGGL(k_wide_sample, dim3(gw_blocks(rows)), dim3(256), 0, s, out, (long)rows, K, (unsigned long long)seed, (long)row_offset, t, eps,
                       action, logp, (long)out_stride, wide_kw(K));
    CM_CHECK_LAUNCH(who);
    return 0;
}
