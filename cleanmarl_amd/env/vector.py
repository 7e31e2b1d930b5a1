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


class EnvWorkerDied(RuntimeError):
    """An env worker process ended while the driver was waiting for its answer."""


def recv_from_worker(conn, proc, what="env worker"):
    """conn.recv() that notices a dead worker.  The reference blocks in a bare recv() (cleanmarl/mappo_multienvs.py:318, 396) and, because the
    parent keeps its copy of the child's pipe end, never even gets an EOFError: an env that crashes (an assertion inside a SMAC map, an OOM
    kill) hangs the run for ever.  Here the parent closes the child's end after the fork (a dead worker then raises EOFError on the
    parent's end) and polls in one-second slices, checking that the process is still alive."""
    while True:
        try:
            if conn.poll(1.0):
                return conn.recv()
        except (EOFError, ConnectionResetError, BrokenPipeError, OSError) as ex:
            raise EnvWorkerDied(f"{what} (pid {proc.pid}) closed its pipe: exit code {proc.exitcode}") from ex
        if not proc.is_alive():
            if conn.poll(0):  # its last message may still be in the pipe
                continue
            raise EnvWorkerDied(f"{what} (pid {proc.pid}) died with exit code {proc.exitcode} while the driver waited for its answer")


def send_to_worker(conn, proc, msg, what="env worker"):
    try:
        conn.send(msg)
    except (BrokenPipeError, ConnectionResetError, OSError) as ex:
        raise EnvWorkerDied(f"{what} (pid {proc.pid}) is gone (exit code {proc.exitcode}): cannot send {msg[0] if isinstance(msg, tuple) else msg!r}") from ex


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
            child.close()  # the parent's copy of the child's end: with it open a dead worker never produces an EOFError here
            self.conns.append(parent)
            self.procs.append(p)

    def _recv(self, i):
        return recv_from_worker(self.conns[i], self.procs[i], f"env worker {i}")

    def _send(self, i, msg):
        send_to_worker(self.conns[i], self.procs[i], msg, f"env worker {i}")

    def info(self):
        self._send(0, ("get_env_info", None))
        return self._recv(0)

    def reset_all(self, seeds=None):
        for i in range(self.n):
            self._send(i, ("reset", None if seeds is None else seeds[i]))
        return [self._recv(i) for i in range(self.n)]

    def step(self, env_ids, actions):
        for i, a in zip(env_ids, actions):
            self._send(i, ("step", a))
        return [self._recv(i) for i in env_ids]

    def close(self):
        for c in self.conns:
            try:
                c.send(("close", None))
            except (BrokenPipeError, OSError):
                pass
        for p in self.procs:
            p.join(timeout=5)
