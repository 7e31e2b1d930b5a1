/* oracle/ppo_loss.c -- plain-C restatement of the reference's per-time-step PPO loss loop (TEST INFRASTRUCTURE).
 *
 * Follows cleanmarl/mappo_multienvs.py:527-576 literally (IPPO: cleanmarl/ippo_multienvs.py:526-575): for every time step t and
 * every alive env, actor MLP -> masked_fill(~avail, -1e9) -> Categorical log-prob / entropy -> ratio, clipped surrogate, approx KL,
 * clip fraction (agent-mean, env-sum), critic MLP -> squared error against the lambda-return (agent-mean, env-sum); the five sums
 * are divided by N = mask.sum() at the end.  Scalar fp32 arithmetic with double accumulation of the outer sums (the quantities
 * compared are 1e-6-level logged scalars).  Layouts are the REFERENCE's: obs [B][T][A][Do], states [B][T][Ds], avail [B][T][A][K]
 * (uint8), actions [B][T][A] (int64), log_probs / adv / ret [B][T][A], mask [B][T] (uint8).  Parameters are flat fp32 buffers in
 * torch parameters() order: W0[H][Din], b0[H], L x (W[H][H], b[H]), Wout[Dout][H], bout[Dout].
 * Pinned against tests/golden/*.npz in tests/test_oracle_golden.py; built by __graft_entry__.build(). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static void mlp(const float* p, const float* x, int din, int H, int L, int dout, float* out, float* h0, float* h1) {
    const float* W = p; const float* b = p + (size_t)H * din;
    for (int n = 0; n < H; ++n) {
        float s = b[n];
        for (int k = 0; k < din; ++k) s += W[(size_t)n * din + k] * x[k];
        h0[n] = s > 0.0f ? s : 0.0f;
    }
    p = b + H;
    for (int l = 0; l < L; ++l) {
        W = p; b = p + (size_t)H * H;
        for (int n = 0; n < H; ++n) {
            float s = b[n];
            for (int k = 0; k < H; ++k) s += W[(size_t)n * H + k] * h0[k];
            h1[n] = s > 0.0f ? s : 0.0f;
        }
        for (int n = 0; n < H; ++n) h0[n] = h1[n];
        p = b + H;
    }
    W = p; b = p + (size_t)dout * H;
    for (int n = 0; n < dout; ++n) {
        float s = b[n];
        for (int k = 0; k < H; ++k) s += W[(size_t)n * H + k] * h0[k];
        out[n] = s;
    }
}

/* out[6] = actor_loss, critic_loss, entropy, kl, clip fraction, N */
void ppo_losses_ref(const float* obs, const float* states, const uint8_t* avail, const int64_t* actions, const float* logp_old,
                    const float* adv, const float* ret, const uint8_t* mask, int B, int T, int A, int Do, int Ds, int K,
                    const float* actor, int Ha, int La, const float* critic, int Hc, int Lc, int per_agent_critic,
                    double ppo_clip, double entropy_coef, double* out) {
    const int Hmax = Ha > Hc ? Ha : Hc;
    float* h0 = (float*)malloc(sizeof(float) * Hmax); float* h1 = (float*)malloc(sizeof(float) * Hmax);
    float* z = (float*)malloc(sizeof(float) * K);
    double s_pg = 0, s_ent = 0, s_kl = 0, s_clip = 0, s_v = 0, N = 0;
    const float lo = (float)(1.0 - ppo_clip), hi = (float)(1.0 + ppo_clip), eps = (float)ppo_clip;
    for (int t = 0; t < T; ++t) {
        for (int b = 0; b < B; ++b) {
            if (!mask[b * T + t]) continue;
            N += 1.0;
            float pg_a = 0, ent_a = 0, kl_a = 0, clip_a = 0, v_a = 0, vshared = 0;
            if (!per_agent_critic) mlp(critic, states + ((size_t)b * T + t) * Ds, Ds, Hc, Lc, 1, &vshared, h0, h1);
            for (int a = 0; a < A; ++a) {
                const size_t o = ((size_t)b * T + t) * A + a;
                mlp(actor, obs + o * Do, Do, Ha, La, K, z, h0, h1);
                float m = -INFINITY;
                for (int k = 0; k < K; ++k) { if (!avail[o * K + k]) z[k] = -1e9f; if (z[k] > m) m = z[k]; }
                float se = 0;
                for (int k = 0; k < K; ++k) se += expf(z[k] - m);
                const float lse = m + logf(se);
                float ent = 0;
                for (int k = 0; k < K; ++k) { const float lp = z[k] - lse; ent -= expf(lp) * lp; }
                const float lpa = z[actions[o]] - lse;
                const float log_ratio = lpa - logp_old[o], ratio = expf(log_ratio);
                float cr = ratio < lo ? lo : (ratio > hi ? hi : ratio);
                const float p1 = adv[o] * ratio, p2 = adv[o] * cr;
                pg_a += p1 < p2 ? p1 : p2;
                ent_a += ent;
                kl_a += (ratio - 1.0f) - log_ratio;
                clip_a += fabsf(ratio - 1.0f) > eps ? 1.0f : 0.0f;
                float v = vshared;
                if (per_agent_critic) mlp(critic, obs + o * Do, Do, Hc, Lc, 1, &v, h0, h1);
                v_a += (v - ret[o]) * (v - ret[o]);
            }
            s_pg += pg_a / A; s_ent += ent_a / A; s_kl += kl_a / A; s_clip += clip_a / A; s_v += v_a / A;
        }
    }
    out[0] = (-s_pg - entropy_coef * s_ent) / N; out[1] = s_v / N; out[2] = s_ent / N; out[3] = s_kl / N; out[4] = s_clip / N; out[5] = N;
    free(h0); free(h1); free(z);
}
