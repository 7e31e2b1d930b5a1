"""CPU twin of the on-device action sampler (csrc/cm_mlp.hip, mode M_ACT) -- TEST INFRASTRUCTURE.
Restates Actor.act (cleanmarl/mappo_multienvs.py:172-176) with the build's counter-based RNG:
inverse-CDF on softmax(logits) with one Philox4x32-10 uniform keyed by (seed, global row, t)."""
import numpy as np

from .philox import act_uniforms  # the checker's own Philox4x32-10 (pinned to the Random123 known-answer vectors), not the product's


def act(logits, avail, seed, row_offset, t):
    """logits [R,K] float32 (already masked with -1e9), avail [R,K] bool -> (action[R], logp[R], u[R])"""
    logits = np.asarray(logits, dtype=np.float32)
    R, K = logits.shape
    u = act_uniforms(R, seed, row_offset, t)
    m = logits.max(1, keepdims=True)
    e = np.exp(logits - m).astype(np.float32)
    ssum = np.zeros(R, np.float32)
    for k in range(K):  # serial fp32 sum, like the kernel
        ssum = (ssum + e[:, k]).astype(np.float32)
    thr = (u * ssum).astype(np.float32)
    action = np.zeros(R, np.int64)
    for r in range(R):
        cum, chosen, last = np.float32(0), -1, 0
        for k in range(K):
            if logits[r, k] > -5e8:
                cum = np.float32(cum + e[r, k]); last = k
                if chosen < 0 and thr[r] < cum:
                    chosen = k
        action[r] = chosen if chosen >= 0 else last
    lse = (m[:, 0] + np.log(ssum)).astype(np.float32)
    logp = logits[np.arange(R), action] - lse
    return action, logp.astype(np.float32), u


def act_eps(logits, avail, eps, seed, row_offset, t):
    """COMA's Actor.act (cleanmarl/coma_multienvs.py:177-186) with the build's RNG: probs = (1-eps) softmax + eps avail/n_avail,
    inverse CDF over the available actions.  -> (action[R], log prob of the sampled action [R], u[R])"""
    logits = np.asarray(logits, dtype=np.float32)
    avail = np.asarray(avail, dtype=bool)
    R, K = logits.shape
    u = act_uniforms(R, seed, row_offset, t)
    m = logits.max(1, keepdims=True)
    e = np.exp(logits - m).astype(np.float64)
    probs = (1.0 - eps) * e / e.sum(1, keepdims=True) + eps * avail / np.maximum(avail.sum(1, keepdims=True), 1)
    probs = np.where(avail, probs, 0.0)
    cum = np.cumsum(probs, axis=1)
    action = np.zeros(R, np.int64)
    for r in range(R):
        idx = np.nonzero(avail[r] & (u[r] < cum[r]))[0]
        action[r] = idx[0] if idx.size else np.nonzero(avail[r])[0][-1]
    return action, np.log(probs[np.arange(R), action]).astype(np.float32), u
