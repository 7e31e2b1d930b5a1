/* oracle/td_lambda.c -- plain-C restatement of the reference's TD(lambda) double loop (TEST INFRASTRUCTURE).
 *
 * Follows cleanmarl/mappo_multienvs.py:484-504 literally: per episode, t = L-1 .. 0,
 *     next_value = 0 if t == L-1 else V[t+1]
 *     R[t] = last = r[t] + gamma * (lambda * last + (1 - lambda) * next_value)     (1 - lambda evaluated in double)
 *     A[t] = R[t] - V[t]
 * with fp32 tensors and Python-float (double) scalars: every tensor op rounds to fp32, scalars are cast to fp32
 * when they meet a tensor.  Layout: reward [B][T], values [B][T][A], mask [B][T] (uint8), outputs [B][T][A].
 * Pinned against tests/golden/*.npz in tests/test_oracle_golden.py; built by __graft_entry__.build(). */
#include <stdint.h>

void td_lambda_ref(const float* reward, const float* values, const uint8_t* mask, int B, int T, int A,
                   double gamma, double lam, float* ret, float* adv) {
    const float g = (float)gamma, l = (float)lam, oml = (float)(1.0 - lam);
    for (int b = 0; b < B; ++b) {
        int L = 0;
        for (int t = 0; t < T; ++t) L += mask[b * T + t] ? 1 : 0;
        for (int a = 0; a < A; ++a) {
            float last = 0.0f;
            for (int t = 0; t < T; ++t) { ret[(b * T + t) * A + a] = 0.0f; adv[(b * T + t) * A + a] = 0.0f; }
            for (int t = L - 1; t >= 0; --t) {
                const float nv = (t == L - 1) ? 0.0f : values[(b * T + t + 1) * A + a];
                const float t1 = l * last, t2 = oml * nv;
                const float t3 = t1 + t2;
                last = reward[b * T + t] + g * t3;
                ret[(b * T + t) * A + a] = last;
                adv[(b * T + t) * A + a] = last - values[(b * T + t) * A + a];
            }
        }
    }
}
