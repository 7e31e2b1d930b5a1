"""Shared-memory, batched-step vector env for HOST environments.

The reference spends 84 % of its wall time in the rollout because every env is its own OS process behind a
Pipe and every step costs one pickled round trip PER ENV (cleanmarl/mappo_multienvs.py:299-319, 415-424).  Here
W worker processes each own a contiguous block of envs; observations, states, availability masks, rewards and
done flags live in shared-memory numpy arrays that the workers write in place, actions are read from a shared
array, and one step costs ONE tiny token per WORKER (not per env) in each direction.  Any CommonInterface env
(PettingZoo / SMAClite adapters, the bundled synthetic env) plugs in through the same `environment()` factory;
semantics (alive set shrinking as episodes end, zero-padded collation, team reward, infos at episode end) are
those of the reference loop, so `collect_episode` returns exactly what `driver.host_rollout` builds from the
pipe-per-env protocol.
"""
import os
from multiprocessing import Pipe, Process, shared_memory

import numpy as np

from .vector import EnvWorkerDied, environment, recv_from_worker, send_to_worker  # noqa: F401


def _worker(conn, factory_args, lo, hi, names, shapes, index_offset=0):
    shms = {k: shared_memory.SharedMemory(name=n) for k, n in names.items()}
    arr = {k: np.ndarray(shapes[k][0], dtype=shapes[k][1], buffer=shms[k].buf) for k in names}
    envs = [environment(**dict(factory_args, index=index_offset + i)) for i in range(lo, hi)]
    try:
        while True:
            task = conn.recv()
            if task == "reset":
                for j, env in enumerate(envs):
                    e = lo + j
                    obs, _ = env.reset()
                    arr["obs"][e] = obs; arr["state"][e] = env.get_state(); arr["avail"][e] = env.get_avail_actions()
                    arr["alive"][e] = 1
                conn.send(None)
            elif task == "step":
                infos = {}
                for j, env in enumerate(envs):
                    e = lo + j
                    if not arr["alive"][e]:
                        continue
                    obs, r, done, trunc, info = env.step(arr["actions"][e])
                    arr["reward"][e] = r; arr["done"][e] = done; arr["trunc"][e] = trunc
                    if done or trunc:
                        arr["alive"][e] = 0
                        infos[e] = info
                    else:
                        arr["obs"][e] = obs; arr["state"][e] = env.get_state(); arr["avail"][e] = env.get_avail_actions()
                conn.send(infos or None)
            elif task == "close":
                for env in envs:
                    env.close()
                conn.send(None)
                break
    finally:
        for s in shms.values():
            s.close()


class ShmVectorEnv:
    def __init__(self, n_envs, factory_args, n_workers=None, index_offset=0):
        """index_offset: global index of local env 0 (this rank's first env when the batch is env-sharded over ranks)."""
        probe = environment(**dict(factory_args, index=index_offset))
        self.E, self.A = n_envs, probe.n_agents
        self.Do, self.Ds, self.K = probe.get_obs_size(), probe.get_state_size(), probe.get_action_size()
        probe.close()
        E, A = self.E, self.A
        self.shapes = {"obs": ((E, A, self.Do), np.float32), "state": ((E, self.Ds), np.float32), "avail": ((E, A, self.K), np.uint8),
                       "reward": ((E,), np.float32), "done": ((E,), np.uint8), "trunc": ((E,), np.uint8),
                       "alive": ((E,), np.uint8), "actions": ((E, A), np.int32)}
        self.shms = {k: shared_memory.SharedMemory(create=True, size=max(1, int(np.prod(s)) * np.dtype(d).itemsize))
                     for k, (s, d) in self.shapes.items()}
        self.arr = {k: np.ndarray(s, dtype=d, buffer=self.shms[k].buf) for k, (s, d) in self.shapes.items()}
        W = n_workers or max(1, min(E, (os.cpu_count() or 2) - 1))
        self.conns, self.procs = [], []
        names = {k: s.name for k, s in self.shms.items()}
        for w in range(W):
            lo, hi = w * E // W, (w + 1) * E // W
            if lo == hi:
                continue
            parent, child = Pipe()
            p = Process(target=_worker, args=(child, factory_args, lo, hi, names, self.shapes, index_offset), daemon=True)
            p.start()
            child.close()  # see vector.recv_from_worker: a dead worker must surface as an error, not as a hang
            self.conns.append(parent); self.procs.append(p)

    def info(self):
        return {"obs_size": self.Do, "action_size": self.K, "n_agents": self.A, "state_size": self.Ds}

    def _all(self, task):
        for w, (c, p) in enumerate(zip(self.conns, self.procs)):
            send_to_worker(c, p, task, f"shared-memory env worker {w}")
        return [recv_from_worker(c, p, f"shared-memory env worker {w}") for w, (c, p) in enumerate(zip(self.conns, self.procs))]

    def collect_episode(self, act_fn, recurrent_state=None):
        """One episode per env.  act_fn(obs[n,A,Do], avail[n,A,K], alive_idx) -> (actions[n,A], logp[n,A]).
        Returns reference-layout numpy arrays (b_obs [E,T,A,Do], ... , b_mask [E,T]) + per-episode stats."""
        E, A, a = self.E, self.A, self.arr
        self._all("reset")
        cap = 32
        buf = dict(obs=np.zeros((E, cap, A, self.Do), np.float32), avail=np.zeros((E, cap, A, self.K), bool),
                   act=np.zeros((E, cap, A), np.int64), logp=np.zeros((E, cap, A), np.float32),
                   rew=np.zeros((E, cap), np.float32), state=np.zeros((E, cap, self.Ds), np.float32))
        ep_len = np.zeros(E, np.int64)
        ep_info = [None] * E
        alive = np.arange(E)
        t = 0
        while alive.size:
            if t == cap:  # grow the time axis geometrically
                for k in buf:
                    pad = np.zeros_like(buf[k])
                    buf[k] = np.concatenate([buf[k], pad], axis=1)
                cap *= 2
            obs, avail = a["obs"][alive], a["avail"][alive]
            actions, logp = act_fn(obs, avail, alive)
            a["actions"][alive] = actions
            buf["obs"][alive, t] = obs; buf["avail"][alive, t] = avail.astype(bool); buf["act"][alive, t] = actions
            buf["logp"][alive, t] = logp; buf["state"][alive, t] = a["state"][alive]
            for infos in self._all("step"):
                if infos:
                    for e, info in infos.items():
                        ep_info[e] = info
            buf["rew"][alive, t] = a["reward"][alive]
            ep_len[alive] += 1
            alive = alive[a["alive"][alive] == 1]
            t += 1
        T = int(ep_len.max())
        mask = np.arange(T)[None, :] < ep_len[:, None]
        out = {k: v[:, :T] for k, v in buf.items()}
        stats = dict(ep_reward=(out["rew"] * mask).sum(1).tolist(), ep_len=ep_len.tolist(), infos=ep_info)
        return out, mask, stats

    def close(self):
        try:
            self._all("close")
        except (BrokenPipeError, OSError, EOFError, EnvWorkerDied):
            pass
        for p in self.procs:
            p.join(timeout=5)
        for s in self.shms.values():
            s.close()
            try:
                s.unlink()
            except FileNotFoundError:
                pass
