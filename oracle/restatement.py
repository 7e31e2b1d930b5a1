"""CPU restatement of the reference's MAPPO / IPPO learner math (TEST INFRASTRUCTURE).

This module is the ORACLE for the hot path: a batched torch-CPU (fp32) restatement of
what cleanmarl/mappo_multienvs.py (and its IPPO / GRU siblings) compute with Python
loops.  It is pinned against golden vectors produced by running the unmodified
reference scripts (tests/golden/make_golden.py -> tests/golden/*.npz; checked by
tests/test_oracle_golden.py).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it; the product path (cleanmarl_amd/) never does.

All tensors use the REFERENCE layout: obs [B,T,A,Do], actions/logp/adv/ret [B,T,A],
reward/mask [B,T], states [B,T,Ds], avail [B,T,A,K] (cleanmarl/mappo_multienvs.py:113-132).
Parameters are flat python lists in torch ``module.parameters()`` order:
  MLP  : [W0,b0, W1,b1, ..., Wout,bout]              (mappo_multienvs.py:160-200)
  GRU  : [fc1.W, fc1.b, W_ih[3H,H], W_hh[3H,H], b_ih, b_hh, fc2.W, fc2.b]
         (mappo_lstm_multienvs.py:162-168)
"""
import math

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------- a2: reward normalisation
def normalize_reward(reward, mask):
    """RolloutBuffer.get_batch, mappo_multienvs.py:143-146 (unbiased std, +1e-6)."""
    reward = reward.clone()
    mu = reward[mask].mean()
    std = reward[mask].std()
    reward[mask] = (reward[mask] - mu) / (std + 1e-6)
    return reward


# ---------------------------------------------------------------- a3 / a5: MLPs
def mlp_forward(params, x):
    """Linear+ReLU, L x (Linear+ReLU), Linear.  mappo_multienvs.py:165-170, 178-183, 197-200."""
    n = len(params) // 2
    for i in range(n):
        x = F.linear(x, params[2 * i], params[2 * i + 1])
        if i < n - 1:
            x = torch.relu(x)
    return x


def actor_logits(params, x, avail=None):
    x = mlp_forward(params, x)
    if avail is not None:
        x = x.masked_fill(~avail, -1e9)  # mappo_multienvs.py:182
    return x


def gru_actor_logits(params, x, h, avail=None):
    """mappo_lstm_multienvs.py:176-184.  x [N,Do], h [N,H] or None."""
    W1, b1, Wih, Whh, bih, bhh, W2, b2 = params
    x = torch.relu(F.linear(x, W1, b1))
    if h is None:
        h = torch.zeros(x.size(0), W1.shape[0], dtype=x.dtype)
    gi = F.linear(x, Wih, bih)
    gh = F.linear(h, Whh, bhh)
    i_r, i_z, i_n = gi.chunk(3, 1)
    h_r, h_z, h_n = gh.chunk(3, 1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    h = (1 - z) * n + z * h
    out = F.linear(torch.relu(h), W2, b2)
    if avail is not None:
        out = out.masked_fill(~avail, -1e9)
    return out, h


def critic_values(critic_params, batch, algo):
    """V[B,T,A].  MAPPO: critic(state) broadcast over agents (mappo_multienvs.py:495,503,554);
    IPPO: critic(obs).squeeze() per agent (ippo_multienvs.py:200,495,503,554)."""
    A = batch["obs"].shape[2]
    if algo == "mappo":
        v = mlp_forward(critic_params, batch["states"])  # [B,T,1]
        return v.expand(-1, -1, A)
    return mlp_forward(critic_params, batch["obs"]).squeeze(-1)  # [B,T,A]


# ---------------------------------------------------------------- a6: TD(lambda)
def td_lambda(reward, values, mask, gamma, lam):
    """Closed form of the double loop at mappo_multienvs.py:484-504.

    R_t = r_t + gamma*(lam*R_{t+1} + (1-lam)*V_{t+1});  R_L = V_L = 0 at the episode's last
    valid step regardless of done/truncated;  A_t = R_t - V_t;  zeros on padded steps.
    reward [B,T], values [B,T,A], mask [B,T] bool -> (returns, advantages) [B,T,A]."""
    B, T, A = values.shape
    m = mask.unsqueeze(-1).to(values.dtype)
    v = values * m
    ret = torch.zeros_like(values)
    nxt_r = torch.zeros(B, A, dtype=values.dtype)
    nxt_v = torch.zeros(B, A, dtype=values.dtype)
    one_minus = 1 - lam  # python float64, as in the reference
    for t in reversed(range(T)):
        r_t = reward[:, t].unsqueeze(-1) + gamma * (lam * nxt_r + one_minus * nxt_v)
        r_t = r_t * m[:, t]
        ret[:, t] = r_t
        nxt_r = r_t
        nxt_v = v[:, t]
    adv = (ret - v) * m
    return ret, adv


# ---------------------------------------------------------------- a7: normalisation
def normalize_masked(x, mask):
    """mappo_multienvs.py:505-512: agent-mean first, unbiased std, no eps, applied everywhere."""
    mu = x.mean(dim=-1)[mask].mean()
    std = x.mean(dim=-1)[mask].std()
    return (x - mu) / std


# ---------------------------------------------------------------- a8 / a9: losses
def categorical_stats(logits, actions):
    """torch.distributions.Categorical(logits=...) log_prob / entropy semantics."""
    logp_all = logits - logits.logsumexp(dim=-1, keepdim=True)
    p = torch.softmax(logp_all, dim=-1)
    logp = logp_all.gather(-1, actions.unsqueeze(-1)).squeeze(-1)
    ent = -(torch.clamp(logp_all, min=torch.finfo(logp_all.dtype).min) * p).sum(-1)
    return logp, ent


def actor_terms(logits, batch, adv, ppo_clip, entropy_coef):
    """Everything the actor graph produces for one [B,T,A,K] logits tensor.
    Returns dict of masked SUMS (not yet divided by N) -- mappo_multienvs.py:533-570."""
    mask = batch["mask"]
    logp, ent = categorical_stats(logits, batch["actions"])
    log_ratio = logp - batch["log_probs"]
    ratio = torch.exp(log_ratio)
    pg1 = adv * ratio
    pg2 = adv * torch.clamp(ratio, 1 - ppo_clip, 1 + ppo_clip)
    pg = torch.min(pg1, pg2)
    mf = mask.to(logits.dtype)
    s_pg = (pg.mean(-1) * mf).sum()
    s_ent = (ent.mean(-1) * mf).sum()
    s_kl = (((ratio - 1) - log_ratio).mean(-1) * mf).sum()
    s_clip = (((ratio - 1.0).abs() > ppo_clip).float().mean(-1) * mf).sum()
    return dict(pg=s_pg, ent=s_ent, kl=s_kl, clip=s_clip,
                loss=-s_pg - entropy_coef * s_ent)


def critic_term(values, ret, mask):
    """sum_{b,t} mask * mean_a (V - R)^2   (mappo_multienvs.py:554-558)."""
    return (((values - ret) ** 2).mean(-1) * mask.to(values.dtype)).sum()


# ---------------------------------------------------------------- a10-a12: optimiser
def grad_norm(grads):
    """norm_d(grads, 2): mappo_multienvs.py:221-224."""
    return torch.linalg.vector_norm(torch.tensor([torch.linalg.vector_norm(g, 2) for g in grads]), 2)


def clip_grads_(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (mappo_multienvs.py:586-592)."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g, 2) for g in grads]), 2)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


class AdamState:
    """Hand-written torch.optim.Adam / AdamW (defaults betas=(0.9,0.999), eps=1e-8;
    Adam wd=0, AdamW decoupled wd=0.01) -- SURVEY.md §8(a) row a12; plus SGD (no momentum) and RMSprop (alpha 0.99,
    eps 1e-8, not centred), the other classes getattr(optim, args.optimizer) is commonly pointed at."""

    def __init__(self, params, lr, kind="Adam", betas=(0.9, 0.999), eps=1e-8, weight_decay=None):
        self.lr, self.kind, self.b1, self.b2, self.eps = lr, kind, betas[0], betas[1], eps
        self.wd = (0.01 if kind == "AdamW" else 0.0) if weight_decay is None else weight_decay
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.t = 0

    def step(self, params, grads):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        for p, g, m, v in zip(params, grads, self.m, self.v):
            if self.kind == "SGD":
                p.add_(g, alpha=-self.lr)
                continue
            if self.kind == "RMSprop":
                v.mul_(0.99).addcmul_(g, g, value=1 - 0.99)
                p.addcdiv_(g, v.sqrt().add_(self.eps), value=-self.lr)
                continue
            if self.kind == "AdamW" and self.wd:
                p.mul_(1 - self.lr * self.wd)
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(m, denom, value=-self.lr / bc1)


# ---------------------------------------------------------------- whole update (MLP scripts)
def prepare_targets(batch, critic_params, hp, algo):
    """TD(lambda) + optional normalisations, mappo_multienvs.py:484-512."""
    with torch.no_grad():
        values = critic_values(critic_params, batch, algo)
        ret, adv = td_lambda(batch["reward"], values, batch["mask"], hp["gamma"], hp["td_lambda"])
        if hp.get("normalize_advantage"):
            adv = normalize_masked(adv, batch["mask"])
        if hp.get("normalize_return"):
            ret = normalize_masked(ret, batch["mask"])
    return ret, adv


def mlp_epoch(actor_params, critic_params, batch, ret, adv, hp, algo):
    """One epoch of mappo_multienvs.py:521-594 up to (not including) the optimiser step.
    Returns (scalars dict, actor grads, critic grads) with grads PRE-clip."""
    ap = [p.detach().clone().requires_grad_(True) for p in actor_params]
    cp = [p.detach().clone().requires_grad_(True) for p in critic_params]
    N = batch["mask"].sum()
    logits = actor_logits(ap, batch["obs"], batch["avail"])
    at = actor_terms(logits, batch, adv, hp["ppo_clip"], hp["entropy_coef"])
    actor_loss = at["loss"] / N
    values = critic_values(cp, batch, algo)
    critic_loss = critic_term(values, ret, batch["mask"]) / N
    ag = torch.autograd.grad(actor_loss, ap)
    cg = torch.autograd.grad(critic_loss, cp)
    scal = dict(actor_loss=actor_loss.item(), critic_loss=critic_loss.item(),
                entropy=(at["ent"] / N).item(), kl=(at["kl"] / N).item(),
                clipfrac=(at["clip"] / N).item())
    return scal, [g.clone() for g in ag], [g.clone() for g in cg]


def mlp_update(actor_params, critic_params, batch, hp, algo, actor_opt=None, critic_opt=None):
    """Full update of one iteration (targets + epochs x [losses, clip, Adam]).
    Mutates the param lists in place; returns per-epoch records."""
    ret, adv = prepare_targets(batch, critic_params, hp, algo)
    kind = hp.get("optimizer", "Adam")
    actor_opt = actor_opt or AdamState(actor_params, hp["learning_rate_actor"], kind)
    critic_opt = critic_opt or AdamState(critic_params, hp["learning_rate_critic"], kind)
    recs = []
    for _ in range(int(hp["epochs"])):
        scal, ag, cg = mlp_epoch(actor_params, critic_params, batch, ret, adv, hp, algo)
        scal["actor_gnorm"] = grad_norm(ag).item()
        scal["critic_gnorm"] = grad_norm(cg).item()
        if hp.get("clip_gradients", -1) > 0:
            clip_grads_(ag, hp["clip_gradients"])
            clip_grads_(cg, hp["clip_gradients"])
        with torch.no_grad():
            actor_opt.step(actor_params, ag)
            critic_opt.step(critic_params, cg)
        recs.append(dict(scal, actor_grads=ag, critic_grads=cg,
                         actor_after=[p.clone() for p in actor_params],
                         critic_after=[p.clone() for p in critic_params]))
    return ret, adv, recs


# ---------------------------------------------------------------- whole update (GRU scripts)
def gru_update(actor_params, critic_params, batch, hp, algo, actor_opt=None, critic_opt=None):
    """mappo_lstm_multienvs.py:551-664: TBPTT actor (optimiser step per chunk, chunk loss divided
    by N_chunk * T_chunk, h detached at chunk boundaries), critic one step per epoch."""
    ret, adv = prepare_targets(batch, critic_params, hp, algo)
    kind = hp.get("optimizer", "Adam")
    actor_opt = actor_opt or AdamState(actor_params, hp["learning_rate_actor"], kind)
    critic_opt = critic_opt or AdamState(critic_params, hp["learning_rate_critic"], kind)
    B, T, A, Do = batch["obs"].shape
    mask = batch["mask"]
    N = mask.sum()
    tb = int(hp["tbptt"])
    recs = []
    for _ in range(int(hp["epochs"])):
        tot = dict(loss=0.0, ent=0.0, kl=0.0, clip=0.0)
        chunk_norms, actor_steps = [], []
        h = None
        t0 = 0
        while t0 < T:
            t1 = min(t0 + tb, T)
            ap = [p.detach().clone().requires_grad_(True) for p in actor_params]
            hh = h
            chunk_loss = 0.0
            denom = 0
            for t in range(t0, t1):
                lg, hh = gru_actor_logits(ap, batch["obs"][:, t].reshape(B * A, Do), hh,
                                          batch["avail"][:, t].reshape(B * A, -1))
                lg = lg.reshape(B, 1, A, -1)
                sub = dict(mask=mask[:, t:t + 1], actions=batch["actions"][:, t:t + 1],
                           log_probs=batch["log_probs"][:, t:t + 1])
                at = actor_terms(lg, sub, adv[:, t:t + 1], hp["ppo_clip"], hp["entropy_coef"])
                chunk_loss = chunk_loss + at["loss"]
                denom += int(mask[:, t].sum())
                for k in tot:
                    tot[k] += float(at[k].detach())
            chunk_loss = chunk_loss / (denom * (t1 - t0))
            ag = [g.clone() for g in torch.autograd.grad(chunk_loss, ap)]
            chunk_norms.append(grad_norm(ag).item())
            if hp.get("clip_gradients", -1) > 0:
                clip_grads_(ag, hp["clip_gradients"])
            with torch.no_grad():
                actor_opt.step(actor_params, ag)
            actor_steps.append(dict(grads=ag, after=[p.clone() for p in actor_params]))
            h = hh.detach()
            t0 = t1
        cp = [p.detach().clone().requires_grad_(True) for p in critic_params]
        critic_loss = critic_term(critic_values(cp, batch, algo), ret, mask) / N
        cg = [g.clone() for g in torch.autograd.grad(critic_loss, cp)]
        cn = grad_norm(cg).item()
        if hp.get("clip_gradients", -1) > 0:
            clip_grads_(cg, hp["clip_gradients"])
        with torch.no_grad():
            critic_opt.step(critic_params, cg)
        Nf = float(N)
        recs.append(dict(actor_loss=tot["loss"] / Nf, critic_loss=critic_loss.item(), entropy=tot["ent"] / Nf,
                         kl=tot["kl"] / Nf, clipfrac=tot["clip"] / Nf,
                         actor_gnorm=sum(chunk_norms) / len(chunk_norms), critic_gnorm=cn,
                         actor_steps=actor_steps, critic_grads=cg,
                         critic_after=[p.clone() for p in critic_params]))
    return ret, adv, recs


def gru_chunk_sums(actor_params, batch, adv, h_in, t0, t1, ppo_clip, entropy_coef):
    """UN-NORMALISED gradient of one TBPTT chunk's loss  sum_{t in [t0, t1)} (-pg_t - c ent_t)  from the detached hidden state
    h_in [B*A, H] (the quantity the chunk kernels accumulate before the division by N_chunk * T_chunk of
    mappo_lstm_multienvs.py:605-607) and the chunk's valid (env, t) count.  batch: obs / actions / log_probs / avail / mask in the
    reference layout [B, T, A, ...]; adv [B, T, A]."""
    B, T, A, Do = batch["obs"].shape
    ap = [p.detach().clone().requires_grad_(True) for p in actor_params]
    hh = h_in
    loss, count = 0.0, 0
    for t in range(t0, t1):
        lg, hh = gru_actor_logits(ap, batch["obs"][:, t].reshape(B * A, Do), hh, batch["avail"][:, t].reshape(B * A, -1))
        sub = dict(mask=batch["mask"][:, t:t + 1], actions=batch["actions"][:, t:t + 1], log_probs=batch["log_probs"][:, t:t + 1])
        loss = loss + actor_terms(lg.reshape(B, 1, A, -1), sub, adv[:, t:t + 1], ppo_clip, entropy_coef)["loss"]
        count += int(batch["mask"][:, t].sum())
    return [g.clone() for g in torch.autograd.grad(loss, ap)], dict(count=count)


# ---------------------------------------------------------------- fixture helpers
def load_golden(path):
    """Read a tests/golden/*.npz into (batch dict, actor params, critic params, hp, golden dict)."""
    import numpy as np
    z = np.load(path, allow_pickle=False)
    batch = dict(obs=torch.from_numpy(z["b_obs"]), actions=torch.from_numpy(z["b_actions"]),
                 log_probs=torch.from_numpy(z["b_log_probs"]), reward=torch.from_numpy(z["b_reward"]),
                 states=torch.from_numpy(z["b_states"]), avail=torch.from_numpy(z["b_avail_actions"]),
                 mask=torch.from_numpy(z["b_mask"]))
    ap = [torch.from_numpy(z[f"actor_init_{i}"]).clone() for i in range(int(z["actor_nparam"]))]
    cp = [torch.from_numpy(z[f"critic_init_{i}"]).clone() for i in range(int(z["critic_nparam"]))]
    hp = {}
    for k in z.files:
        if k.startswith("hp_"):
            v = z[k]
            hp[k[3:]] = float(v) if v.dtype.kind == "f" else str(v)
    return batch, ap, cp, hp, z


def flat(params):
    return torch.cat([p.reshape(-1) for p in params])
