"""CPU restatement of the reference's COMA learner math (TEST INFRASTRUCTURE, like oracle/restatement.py).

Batched torch-CPU (fp32) restatement of cleanmarl/coma_multienvs.py:553-676 (identical in cleanmarl/coma.py): targets from
the TARGET critic (TD(lambda), :556-580, or n-step returns, :581-613), optional return normalisation (:615-618), ONE critic
step on the gathered Q (:620-643), polyak target update (:645-647), ONE actor step with the counterfactual baseline and the
per-time-step advantage normalisation (:648-684).  Pinned against goldens captured from the unmodified reference
(tests/golden/coma_*.npz, tests/test_oracle_golden.py).  Only tests/ may import it; the product path never does.

Reference layout: obs [B,T,A,Do], actions [B,T,A] int64, reward/mask [B,T], states [B,T,Ds], avail [B,T,A,K] bool.
"""
import torch
import torch.nn.functional as F

from . import restatement as R


def coma_inputs(states, obs, actions, K):
    """Critic.coma_inputs (coma_multienvs.py:222-240): [state | own obs | one-hot actions of the OTHER agents, in agent order].
    states [...,Ds], obs [...,A,Do], actions [...,A] -> [...,A, Ds+Do+(A-1)K]."""
    A = obs.shape[-2]
    lead = obs.shape[:-2]
    oh = F.one_hot(actions.long(), K).to(obs.dtype)                      # [...,A,K]
    others = []
    for a in range(A):
        idx = [j for j in range(A) if j != a]
        others.append(oh[..., idx, :].reshape(*lead, (A - 1) * K))
    others = torch.stack(others, dim=-2)                                  # [...,A,(A-1)K]
    st = states.unsqueeze(-2).expand(*lead, A, states.shape[-1])
    return torch.cat([st, obs, others], dim=-1)


def q_values(critic_params, batch, K, avail=None):
    """Critic.forward on the whole batch: [B,T,A,K]; masked_fill(~avail, -1e9) only where the reference passes avail
    (the TARGET critic inside the return loops, :565-570, :590-595)."""
    q = R.mlp_forward(critic_params, coma_inputs(batch["states"], batch["obs"], batch["actions"], K))
    if avail is not None:
        q = q.masked_fill(~avail, -1e9)
    return q


def q_taken(q, actions):
    return q.gather(-1, actions.long().unsqueeze(-1)).squeeze(-1)


def td_lambda_targets(batch, target_params, gamma, lam, K):
    """:556-580.  Same recursion as MAPPO's with V := Q_target(s, u)[taken action]; bootstrap 0 at the last valid step."""
    with torch.no_grad():
        qt = q_taken(q_values(target_params, batch, K, batch["avail"]), batch["actions"])
        ret, _ = R.td_lambda(batch["reward"], qt, batch["mask"], gamma, lam)
    return ret


def nstep_targets(batch, target_params, gamma, n, K):
    """:581-613.  G_t = sum_{i<n} gamma^i r_{t+i} + gamma^n Q_target(t+n)[taken]  if t < L-n, else the discounted
    reward-to-go (no bootstrap).  Discounts are python-float powers cast to fp32, summed left to right like torch.sum."""
    with torch.no_grad():
        qt = q_taken(q_values(target_params, batch, K, batch["avail"]), batch["actions"])
        B, T, A = qt.shape
        ret = torch.zeros(B, T, A)
        lens = batch["mask"].sum(1)
        for b in range(B):
            L = int(lens[b])
            for t in range(L):
                if t < L - n:
                    r = batch["reward"][b, t:t + n]
                    disc = torch.tensor([gamma ** i for i in range(r.numel())])
                    ret[b, t] = (r * disc).sum() + gamma ** n * qt[b, t + n]
                else:
                    r = batch["reward"][b, t:L]
                    disc = torch.tensor([gamma ** i for i in range(r.numel())])
                    ret[b, t] = (r * disc).sum()
    return ret


def critic_loss_sum(critic_params, batch, targets, K):
    """:621-631 -> masked SUM over (b,t) of the agent-mean squared error (divide by N = mask.sum() for cr_loss)."""
    q = q_taken(q_values(critic_params, batch, K), batch["actions"])
    return (((q - targets) ** 2).mean(-1) * batch["mask"]).sum()


def actor_terms(actor_params, critic_params, batch, K, entropy_coef, normalize_advantage):
    """:649-676 with eps = 0 (actor.logits is called without eps in the update).  Returns masked SUMS:
    loss = sum_t[-(log pi_a * adv).sum - c * ent_t], ent = sum of the K-MEAN entropies, plus the advantages used."""
    logits = R.actor_logits(actor_params, batch["obs"], batch["avail"])
    pi = torch.softmax(logits, -1)
    log_pi = torch.log(pi + 1e-8)
    ent = -(pi * log_pi).mean(-1)                                         # [B,T,A]  (mean over K, :652)
    with torch.no_grad():
        q = q_values(critic_params, batch, K)
        adv = q_taken(q, batch["actions"]) - (pi.detach() * q).sum(-1)
        if normalize_advantage:
            A = adv.shape[-1]
            for t in range(adv.shape[1]):
                if batch["actions"][:, t].sum() > A:                      # :664 (sic: sum of action indices over ALL envs)
                    sel = adv[:, t][batch["mask"][:, t]]
                    adv[:, t] = (adv[:, t] - sel.mean()) / (sel.std() + 1e-8)
    lpa = q_taken(log_pi, batch["actions"])
    m = batch["mask"].unsqueeze(-1).to(lpa.dtype)
    ent_sum = (ent * m).sum()
    loss = -((lpa * adv) * m).sum() - entropy_coef * ent_sum
    return dict(loss=loss, ent=ent_sum, adv=adv)


def update(actor_params, critic_params, target_params, batch, hp, actor_opt=None, critic_opt=None, training_step=0):
    """One training iteration of coma_multienvs.py:553-684 (in place on the parameter lists).  Returns a record."""
    K = actor_params[-1].shape[0]
    N = batch["mask"].sum().float()
    if hp.get("use_tdlamda", 1.0):
        ret = td_lambda_targets(batch, target_params, hp["gamma"], hp["td_lambda"], K)
    else:
        ret = nstep_targets(batch, target_params, hp["gamma"], int(hp["nsteps"]), K)
    if hp.get("normalize_return"):
        ret = R.normalize_masked(ret, batch["mask"])
    actor_opt = actor_opt or R.AdamState(actor_params, hp["learning_rate_actor"], hp["optimizer"])
    critic_opt = critic_opt or R.AdamState(critic_params, hp["learning_rate_critic"], hp["optimizer"])
    cpr = [p.detach().clone().requires_grad_(True) for p in critic_params]
    cl = critic_loss_sum(cpr, batch, ret, K) / N
    cg = [g.clone() for g in torch.autograd.grad(cl, cpr)]
    cnorm = R.grad_norm(cg)
    if hp["clip_gradients"] > 0:
        R.clip_grads_(cg, hp["clip_gradients"])
    with torch.no_grad():
        critic_opt.step(critic_params, cg)
    training_step += 1
    if training_step % int(hp["target_network_update_freq"]) == 0:        # soft_update, :266-270
        with torch.no_grad():
            for tp, p in zip(target_params, critic_params):
                tp.copy_(hp["polyak"] * p + (1.0 - hp["polyak"]) * tp)
    apr = [p.detach().clone().requires_grad_(True) for p in actor_params]
    at = actor_terms(apr, critic_params, batch, K, hp["entropy_coef"], bool(hp["normalize_advantage"]))
    al = at["loss"] / N
    ag = [g.clone() for g in torch.autograd.grad(al, apr)]
    anorm = R.grad_norm(ag)
    if hp["clip_gradients"] > 0:
        R.clip_grads_(ag, hp["clip_gradients"])
    with torch.no_grad():
        actor_opt.step(actor_params, ag)
    return dict(ret=ret, adv=at["adv"], critic_loss=float(cl.detach()), actor_loss=float(al.detach()), entropy=float(at["ent"].detach() / N),
                critic_gnorm=float(cnorm), actor_gnorm=float(anorm), critic_grads=R.flat(cg), actor_grads=R.flat(ag),
                training_step=training_step)  # grads as the optimiser consumed them (post-clip), like the goldens


def load_golden(path):
    import numpy as np
    z = np.load(path, allow_pickle=False)
    batch = dict(obs=torch.from_numpy(z["b_obs"]), actions=torch.from_numpy(z["b_actions"]), reward=torch.from_numpy(z["b_reward"]),
                 states=torch.from_numpy(z["b_states"]), avail=torch.from_numpy(z["b_avail_actions"]), mask=torch.from_numpy(z["b_mask"]))
    ap = [torch.from_numpy(z[f"actor_init_{i}"]).clone() for i in range(int(z["actor_nparam"]))]
    cp = [torch.from_numpy(z[f"critic_init_{i}"]).clone() for i in range(int(z["critic_nparam"]))]
    hp = {}
    for k in z.files:
        if k.startswith("hp_"):
            v = z[k]
            hp[k[3:]] = float(v) if v.dtype.kind == "f" else str(v)
    return batch, ap, cp, hp, z
