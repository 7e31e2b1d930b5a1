"""Reference-STRUCTURED CPU driver (TEST / BASELINE INFRASTRUCTURE -- never imported by cleanmarl_amd/).

The reference's own files cannot travel to the GPU box, so the CPU number reported beside the GPU
number (bench.py `cpu_baseline`, kind "port") comes from this re-creation of the reference's control
structure, written from the call stack in SURVEY.md §3.1:
  * one OS process + Pipe per environment, one pickled round trip per env per step
    (cleanmarl/mappo_multienvs.py:246-285, 299-319, 393-453),
  * per-env python lists collated into zero-padded tensors (:103-157),
  * TD(lambda) as a per-episode reversed python loop with two single-row critic calls per step (:484-504),
  * `epochs` x T per-timestep autograd graphs, two Adam optimisers (:521-594).
Its numerics are checked against oracle/restatement.py in tests/test_reference_loop.py.
"""
import time
from multiprocessing import Pipe, Process

import numpy as np
import torch
import torch.nn as nn
from torch.distributions.categorical import Categorical

from cleanmarl_amd.env.synthetic import SyntheticSpreadEnv


def _worker(conn, kw):
    env = SyntheticSpreadEnv(**kw)
    while True:
        task, payload = conn.recv()
        if task == "reset":
            obs, _ = env.reset()
            conn.send(dict(obs=obs, avail_actions=env.get_avail_actions(), state=env.get_state()))
        elif task == "step":
            nobs, r, done, trunc, info = env.step(payload)
            conn.send(dict(next_obs=nobs, reward=r, done=done, truncated=trunc, infos=info,
                           avail_actions=env.get_avail_actions(), next_state=env.get_state()))
        elif task == "close":
            conn.close()
            break


def _mlp(din, hidden, n_layers, dout):
    layers = [nn.Linear(din, hidden), nn.ReLU()]
    for _ in range(n_layers):
        layers += [nn.Linear(hidden, hidden), nn.ReLU()]
    layers.append(nn.Linear(hidden, dout))
    return nn.Sequential(*layers)


def _pick_threads(fn, sweep):
    """Time fn() under each torch thread count of `sweep`; leave the fastest one set and return (best, {threads: seconds})."""
    seen = {}
    for n in sweep:
        torch.set_num_threads(int(n))
        t0 = time.perf_counter()
        fn()
        seen[int(n)] = time.perf_counter() - t0
    best = min(seen, key=seen.get)
    torch.set_num_threads(best)
    return best, seen


def run(E, A, T, hidden=64, n_layers=1, epochs=3, seed=1, gamma=0.99, lam=0.95, clip=0.2, ent_coef=1e-3, lr=8e-4,
        iterations=1, thread_sweep=None):
    """One (or more) full training iteration(s) in the reference's loop structure.  Returns a dict with the
    per-phase wall times, agent-steps/s and the final batch / params (for the numerics check).
    thread_sweep: torch intra-op thread counts to try, e.g. (1, 8, os.cpu_count()).  The TD(lambda) loop (single-row critic
    calls) and the update (per-timestep graphs over E rows) are probed on a slice of the first iteration's batch -- 8 episodes for
    the scan, one throw-away epoch on copies of the networks for the update -- and each phase then runs with ITS fastest count
    (the probes are not part of the reported times).  The reference itself runs with torch's default (= all cores), which is
    the slowest choice for these tiny ops on a many-core host; the baseline reported is the fair one."""
    import copy
    torch.manual_seed(seed)
    thr = {"rollout": torch.get_num_threads(), "gae": torch.get_num_threads(), "update": torch.get_num_threads()}
    probes = {}
    if thread_sweep:
        thr["rollout"] = int(min(thread_sweep))  # [E*A, Do] forward per step: too small to split
    Do, Ds, K = 7 * A, 6 * A * A, 5
    actor = _mlp(Do, hidden, n_layers, K)
    critic = _mlp(Ds, hidden, n_layers, 1)
    opt_a = torch.optim.Adam(actor.parameters(), lr=lr)
    opt_c = torch.optim.Adam(critic.parameters(), lr=lr)
    init = ([p.detach().clone() for p in actor.parameters()], [p.detach().clone() for p in critic.parameters()])
    pipes = [Pipe() for _ in range(E)]
    procs = [Process(target=_worker, args=(pipes[i][1], dict(n_agents=A, agent_ids=True, max_cycles=T, seed=seed, env_index=i)),
                     daemon=True) for i in range(E)]
    for p in procs:
        p.start()
    conns = [p[0] for p in pipes]
    t_roll = t_gae = t_upd = 0.0
    out = {}
    for _ in range(iterations):
        torch.set_num_threads(thr["rollout"])
        t0 = time.perf_counter()
        # ---------------- rollout: one pipe round trip per env per step
        eps = [dict(obs=[], actions=[], log_prob=[], reward=[], states=[], avail=[]) for _ in range(E)]
        for c in conns:
            c.send(("reset", None))
        cont = [c.recv() for c in conns]
        obs = np.stack([c["obs"] for c in cont]); avail = np.stack([c["avail_actions"] for c in cont])
        state = np.stack([c["state"] for c in cont])
        alive = list(range(E))
        done_eps = [None] * E
        while alive:
            with torch.no_grad():
                logits = actor(torch.from_numpy(obs).float()).masked_fill(~torch.from_numpy(avail).bool(), -1e9)
                dist = Categorical(logits=logits)
                actions = dist.sample()
                logp = dist.log_prob(actions)
            for i, j in enumerate(alive):
                conns[j].send(("step", actions[i]))
            cont = [conns[j].recv() for j in alive]
            nobs, nstate, navail, still = [], [], [], []
            for i, j in enumerate(alive):
                e = eps[j]
                e["obs"].append(obs[i]); e["actions"].append(actions[i]); e["log_prob"].append(logp[i])
                e["reward"].append(cont[i]["reward"]); e["states"].append(state[i]); e["avail"].append(avail[i])
                if cont[i]["done"] or cont[i]["truncated"]:
                    done_eps[j] = {k: torch.from_numpy(np.stack(v)).float() for k, v in e.items()}
                else:
                    still.append(j); nobs.append(cont[i]["next_obs"]); nstate.append(cont[i]["next_state"])
                    navail.append(cont[i]["avail_actions"])
            alive = still
            if alive:
                obs, state, avail = np.stack(nobs), np.stack(nstate), np.stack(navail)
        # ---------------- collate (zero pad + mask)
        lens = [len(e["obs"]) for e in done_eps]
        Tm = max(lens)
        b_obs = torch.zeros(E, Tm, A, Do); b_av = torch.zeros(E, Tm, A, K); b_act = torch.zeros(E, Tm, A)
        b_lp = torch.zeros(E, Tm, A); b_rew = torch.zeros(E, Tm); b_st = torch.zeros(E, Tm, Ds)
        b_mask = torch.zeros(E, Tm, dtype=torch.bool)
        for i, e in enumerate(done_eps):
            n = lens[i]
            b_obs[i, :n] = e["obs"]; b_av[i, :n] = e["avail"]; b_act[i, :n] = e["actions"]; b_lp[i, :n] = e["log_prob"]
            b_rew[i, :n] = e["reward"]; b_st[i, :n] = e["states"]; b_mask[i, :n] = True
        b_act = b_act.long(); b_av = b_av.bool()
        t1 = time.perf_counter()
        if thread_sweep and "gae" not in probes:
            def gae_probe():
                with torch.no_grad():
                    for i in range(min(E, 8)):
                        for t in range(lens[i]):
                            critic(b_st[i, t]); critic(b_st[i, t])
            thr["gae"], probes["gae"] = _pick_threads(gae_probe, thread_sweep)
        torch.set_num_threads(thr["gae"])
        t1b = time.perf_counter()  # the probe is not part of any phase
        # ---------------- TD(lambda): per-episode reversed loop, single-row critic calls
        ret = torch.zeros(E, Tm, A); adv = torch.zeros(E, Tm, A)
        with torch.no_grad():
            for i in range(E):
                last = 0
                for t in reversed(range(lens[i])):
                    nv = 0 if t == lens[i] - 1 else critic(b_st[i, t + 1])
                    ret[i, t] = last = b_rew[i, t] + gamma * (lam * last + (1 - lam) * nv)
                    adv[i, t] = ret[i, t] - critic(b_st[i, t])
        t2 = time.perf_counter()
        if thread_sweep and "update" not in probes:
            def upd_probe():
                a2, c2 = copy.deepcopy(actor), copy.deepcopy(critic)
                al = 0; cl = 0
                for t in range(0, Tm, 8):  # every 8th time step: the probe only ranks the thread counts
                    d = Categorical(logits=a2(b_obs[:, t]).masked_fill(~b_av[:, t], -1e9))
                    al = al + (torch.exp(d.log_prob(b_act[:, t]) - b_lp[:, t]) * adv[:, t]).sum() + d.entropy().sum()
                    cl = cl + nn.functional.mse_loss(c2(b_st[:, t]).expand(-1, A), ret[:, t])
                al.backward(); cl.backward()
            thr["update"], probes["update"] = _pick_threads(upd_probe, thread_sweep)
        torch.set_num_threads(thr["update"])
        t2b = time.perf_counter()
        # ---------------- update: epochs x T per-timestep graphs
        logs = []
        for _ep in range(epochs):
            a_loss = 0; c_loss = 0
            for t in range(Tm):
                m = b_mask[:, t]
                dist = Categorical(logits=actor(b_obs[:, t]).masked_fill(~b_av[:, t], -1e9))
                lr_ = dist.log_prob(b_act[:, t]) - b_lp[:, t]
                ratio = torch.exp(lr_)
                pg = torch.min(adv[:, t] * ratio, adv[:, t] * torch.clamp(ratio, 1 - clip, 1 + clip))[m].mean(-1).sum()
                ent = dist.entropy()[m].mean(-1).sum()
                a_loss = a_loss + (-pg - ent_coef * ent)
                v = critic(b_st[:, t]).expand(-1, A)
                c_loss = c_loss + nn.functional.mse_loss(v[m], ret[:, t][m]) * m.sum()
            a_loss = a_loss / b_mask.sum(); c_loss = c_loss / b_mask.sum()
            opt_a.zero_grad(); opt_c.zero_grad()
            a_loss.backward(); c_loss.backward()
            opt_a.step(); opt_c.step()
            logs.append((a_loss.item(), c_loss.item()))
        t3 = time.perf_counter()
        t_roll += t1 - t0; t_gae += t2 - t1b; t_upd += t3 - t2b
        out = dict(batch=dict(obs=b_obs, actions=b_act, log_probs=b_lp, reward=b_rew, states=b_st, avail=b_av, mask=b_mask),
                   ret=ret, adv=adv, logs=logs)
    for c in conns:
        c.send(("close", None))
    for p in procs:
        p.join(timeout=5)
    total = t_roll + t_gae + t_upd
    out.update(init=init, actor=[p.detach().clone() for p in actor.parameters()],
               critic=[p.detach().clone() for p in critic.parameters()],
               rollout_s=t_roll, gae_s=t_gae, update_s=t_upd, total_s=total,
               agent_steps_per_s=iterations * E * A * T / total, threads=thr, thread_probes=probes)
    return out
