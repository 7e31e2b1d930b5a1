"""Process-per-environment vectorisation speaking the reference's 5-message pipe protocol
(cleanmarl/mappo_multienvs.py:246-285): requests ("reset", None) ("step", actions) ("get_env_info", None)
("sample", None) ("close", None); replies are dicts with the reference's keys.  Kept for real (host) envs --
PettingZoo / SMAClite adapters written against CommonInterface plug in unchanged.  Actions travel as plain
int lists (the reference pickles a torch tensor per env per step)."""
import importlib
from multiprocessing import Pipe, Process


def environment(env_type, env_name, env_family, agent_ids, kwargs=None, index=0, seed=1, synthetic=None):
    """Factory with the reference's signature (cleanmarl/mappo_multienvs.py:208-218) + the synthetic types."""
    kwargs = dict(kwargs or {})
    if env_type in ("synthetic", "synthetic_cpu"):
        from .synthetic import SyntheticSpreadEnv
        s = synthetic or {}
        steps = s.get("steps", 25)
        if s.get("ragged"):  # episodes of different lengths (exercises the shrinking alive set / zero padding)
            steps = max(1, steps - index % 4)
        return SyntheticSpreadEnv(n_agents=s.get("agents", 3), agent_ids=agent_ids, max_cycles=steps, seed=seed, env_index=index)
    if env_type in ("synthetic_shape", "synthetic_shape_cpu"):
        from .synthetic import SyntheticShapeEnv
        s = synthetic or {}
        return SyntheticShapeEnv(n_agents=s.get("agents", 10), obs_raw=s.get("obs", 105), state_dim=s.get("state", 243),
                                 n_actions=s.get("actions", 17), avail_p=s.get("avail_p", 0.7), agent_ids=agent_ids,
                                 max_cycles=s.get("steps", 25), seed=seed, env_index=index)
    if env_type == "pz":
        mod = importlib.import_module("cleanmarl_amd.env.pettingzoo_wrapper")
        return mod.PettingZooWrapper(family=env_family, env_name=env_name, agent_ids=agent_ids, **kwargs)
    if env_type == "smaclite":
        mod = importlib.import_module("cleanmarl_amd.env.smaclite_wrapper")
        return mod.SMACliteWrapper(map_name=env_name, agent_ids=agent_ids, **kwargs)
    if env_type == "lbf":  # the COMA scripts' third env type (cleanmarl/coma_multienvs.py:248-258): SURVEY.md §2 row 8, out of scope
        raise ValueError("env_type 'lbf' (level-based foraging) is outside this build's scope (SURVEY.md §2); use pz / smaclite / synthetic*")
    raise ValueError(f"unknown env_type {env_type!r} (pz, smaclite, synthetic[_cpu], synthetic_shape[_cpu])")


def env_worker(conn, factory_args):
    env = environment(**factory_args)
    while True:
        task, content = conn.recv()
        if task == "reset":
            obs, _ = env.reset(seed=content)
            conn.send({"obs": obs, "avail_actions": env.get_avail_actions(), "state": env.get_state()})
        elif task == "get_env_info":
            conn.send({"obs_size": env.get_obs_size(), "action_size": env.get_action_size(), "n_agents": env.n_agents,
                       "state_size": env.get_state_size()})
        elif task == "sample":
            conn.send({"actions": env.sample()})
        elif task == "step":
            next_obs, reward, done, truncated, infos = env.step(content)
            conn.send({"next_obs": next_obs, "reward": reward, "done": done, "truncated": truncated, "infos": infos,
                       "avail_actions": env.get_avail_actions(), "next_state": env.get_state()})
        elif task == "close":
            env.close()
            conn.close()
            break


class PipeVectorEnv:
    """B daemon processes, one env each.  index_offset: global index of env 0 (the first env of this rank's shard when the
    batch is env-sharded over ranks), so that index-keyed envs differ between ranks."""

    def __init__(self, n, factory_args, index_offset=0):
        self.n = n
        self.conns, self.procs = [], []
        for i in range(n):
            parent, child = Pipe()
            p = Process(target=env_worker, args=(child, dict(factory_args, index=index_offset + i)), daemon=True)
            p.start()
            self.conns.append(parent)
            self.procs.append(p)

    def info(self):
        self.conns[0].send(("get_env_info", None))
        return self.conns[0].recv()

    def reset_all(self, seeds=None):
        for i, c in enumerate(self.conns):
            c.send(("reset", None if seeds is None else seeds[i]))
        return [c.recv() for c in self.conns]

    def step(self, env_ids, actions):
        for i, a in zip(env_ids, actions):
            self.conns[i].send(("step", a))
        return [self.conns[i].recv() for i in env_ids]

    def close(self):
        for c in self.conns:
            try:
                c.send(("close", None))
            except (BrokenPipeError, OSError):
                pass
        for p in self.procs:
            p.join(timeout=5)
