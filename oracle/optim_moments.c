/* oracle/optim_moments.c -- plain-C restatements of the small stateful pieces of the path (TEST INFRASTRUCTURE, never linked
 * into libcleanmarl_hip.so; the CPU twins SURVEY.md 8(b) lists for cm_grad_norm_clip_adam, cm_masked_moments / cm_normalize,
 * cm_gru_policy_act and cm_policy_act).
 *
 *   clip_optim_step_ref   torch.nn.utils.clip_grad_norm_ (cleanmarl/mappo_multienvs.py:587-592, norm_d :221-224) followed by ONE
 *                         torch.optim step with torch's defaults (the script passes only lr, :341-343): Adam / AdamW (weight decay
 *                         0.01, decoupled) / SGD (no momentum) / RMSprop (alpha = beta2, eps, not centred).  fp32 tensor
 *                         arithmetic, Python-float (double) scalars: bias corrections, step size and sqrt(bc2) are formed in
 *                         double and meet the tensors as fp32 -- the single-tensor code path of torch 2.x.
 *   masked_normalize_ref  the advantage / return normalisation of :505-512: mean and UNBIASED std of the agent-mean over valid
 *                         (env, t) pairs, applied to every entry (padded ones included), no epsilon; with eps > 0 and
 *                         valid_only = 1 it is the reward normalisation of RolloutBuffer.get_batch (:143-146).
 *   gru_cell_ref          torch.nn.GRUCell for one row (cleanmarl/mappo_lstm_multienvs.py:162-184), libm exp / tanh.
 *   categorical_sample_ref  Categorical(logits).sample() by inverse CDF on ONE uniform + log_prob: the sampling rule this build
 *                         defines (the reference draws from torch's global generator, which cannot be reproduced on a device);
 *                         same arithmetic and summation order as cm_categorical_sample (csrc/cm_common.h).
 * Pinned in tests/test_oracle_golden.py: the optimiser against the reference's post-step parameters for all four optimisers
 * (goldens mappo_dense / ragged_norm / deep / rmsprop, ippo_sgd ...), the rest against the pinned Python oracle. */
#include <math.h>
#include <stdint.h>

enum { OPT_ADAM = 0, OPT_ADAMW = 1, OPT_SGD = 2, OPT_RMSPROP = 3 };

/* returns the pre-clip global L2 norm; grads are clipped IN PLACE (what optimizer.step() then consumes), m / v updated */
double clip_optim_step_ref(float* params, float* grads, float* m, float* v, long n, int step, double lr, double beta1, double beta2,
                           double eps, double weight_decay, int kind, double max_norm) {
    /* norm_d / clip_grad_norm_: torch stacks the per-tensor norms and takes their norm; over a flat buffer that is one fp32 norm */
    float ss = 0.0f;
    for (long i = 0; i < n; ++i) ss += grads[i] * grads[i];
    const float norm = sqrtf(ss);
    if (max_norm > 0.0) {
        float coef = (float)max_norm / (norm + 1e-6f);
        if (coef > 1.0f) coef = 1.0f;
        for (long i = 0; i < n; ++i) grads[i] *= coef;
    }
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    const float b1 = (float)beta1, b2 = (float)beta2, e = (float)eps, lrf = (float)lr;
    for (long i = 0; i < n; ++i) {
        const float g = grads[i];
        float p = params[i];
        if (kind == OPT_SGD) { params[i] = p - lrf * g; continue; }
        if (kind == OPT_RMSPROP) {
            v[i] = b2 * v[i] + (1.0f - b2) * g * g;
            params[i] = p - lrf * (g / (sqrtf(v[i]) + e));
            continue;
        }
        if (kind == OPT_ADAMW) p *= (float)(1.0 - lr * weight_decay);
        m[i] = m[i] + (g - m[i]) * (1.0f - b1);               /* exp_avg.lerp_(grad, 1 - beta1) */
        v[i] = b2 * v[i] + (1.0f - b2) * g * g;               /* exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2) */
        const float denom = sqrtf(v[i]) / bc2_sqrt + e;
        params[i] = p - step_size * (m[i] / denom);
    }
    return (double)norm;
}

/* x [B][T][A] in place; mask [B][T].  out3 = {count, mean, unbiased std} of the agent-mean over valid (b, t). */
void masked_normalize_ref(float* x, const uint8_t* mask, int B, int T, int A, double eps, int valid_only, double* out3) {
    double n = 0.0, s = 0.0;
    for (int i = 0; i < B * T; ++i) if (mask[i]) {
        float am = 0.0f;
        for (int a = 0; a < A; ++a) am += x[(long)i * A + a];
        am /= (float)A;
        s += am; n += 1.0;
    }
    const double mean = n > 0 ? s / n : 0.0;
    double m2 = 0.0;
    for (int i = 0; i < B * T; ++i) if (mask[i]) {
        float am = 0.0f;
        for (int a = 0; a < A; ++a) am += x[(long)i * A + a];
        am /= (float)A;
        m2 += ((double)am - mean) * ((double)am - mean);
    }
    const double sd = n > 1 ? sqrt(m2 / (n - 1.0)) : 0.0;
    out3[0] = n; out3[1] = mean; out3[2] = sd;
    const float mu = (float)mean, den = (float)(sd + eps);
    for (int i = 0; i < B * T; ++i) {
        if (valid_only && !mask[i]) continue;
        for (int a = 0; a < A; ++a) x[(long)i * A + a] = (x[(long)i * A + a] - mu) / den;
    }
}

/* one GRUCell step for one row: x [I], h [H] -> hn [H]; W_ih [3H][I], W_hh [3H][H], b_ih / b_hh [3H] (gate order r, z, n) */
void gru_cell_ref(const float* x, const float* h, const float* W_ih, const float* W_hh, const float* b_ih, const float* b_hh,
                  int I, int H, float* hn) {
    for (int j = 0; j < H; ++j) {
        float gi[3], gh[3];
        for (int g = 0; g < 3; ++g) {
            float si = b_ih[g * H + j], sh = b_hh[g * H + j];
            for (int k = 0; k < I; ++k) si += W_ih[((long)g * H + j) * I + k] * x[k];
            for (int k = 0; k < H; ++k) sh += W_hh[((long)g * H + j) * H + k] * h[k];
            gi[g] = si; gh[g] = sh;
        }
        const float r = 1.0f / (1.0f + expf(-(gi[0] + gh[0])));
        const float z = 1.0f / (1.0f + expf(-(gi[1] + gh[1])));
        const float nn = tanhf(gi[2] + r * gh[2]);
        hn[j] = (1.0f - z) * nn + z * h[j];
    }
}

/* z: K logits already masked with -1e9; u in [0, 1).  Returns the action, *logp = log-probability of it. */
int categorical_sample_ref(const float* z, int K, float u, float* logp) {
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) if (z[k] > mx) mx = z[k];
    float s = 0.0f;
    for (int k = 0; k < K; ++k) s += expf(z[k] - mx);
    const float thr = u * s;
    float cum = 0.0f;
    int chosen = -1, last = 0;
    for (int k = 0; k < K; ++k) if (z[k] > -5e8f) {
        cum += expf(z[k] - mx);
        last = k;
        if (chosen < 0 && thr < cum) chosen = k;
    }
    if (chosen < 0) chosen = last;
    *logp = z[chosen] - (mx + logf(s));
    return chosen;
}
