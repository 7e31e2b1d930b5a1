"""Command-line surface of the hot-path scripts (four *_multienvs.py + their four single-environment siblings).

Every field, default and help string mirrors the reference's ``@dataclass Args`` parsed by ``tyro.cli``
(cleanmarl/mappo_multienvs.py:18-79, cleanmarl/ippo_multienvs.py:18-79, cleanmarl/mappo_lstm_multienvs.py:18-80,
cleanmarl/ippo_lstm_multienvs.py:18-80); tyro is not a dependency here, so an argparse front-end reproduces its
spelling rules: ``--a_b=v``, ``--a_b v``, ``--a-b v`` and ``--flag / --no-flag`` (also ``--flag=True|False``).
Build-only additions never change an existing default: ``--env_type=synthetic`` (on-device MPE-like env) /
``synthetic_cpu`` (same env on the host behind the vector-env protocol), ``synthetic_shape`` / ``synthetic_shape_cpu``
(SMAClite-shaped random-feature env with availability masks) with ``--synthetic_agents`` and
``--synthetic_steps``; ``--device`` defaults to ``cuda`` because this build has no CPU compute path.
"""
import argparse
from dataclasses import dataclass, fields


@dataclass
class Args:
    env_type: str = "smaclite"
    """ Pettingzoo, SMAClite ... (build adds: synthetic, synthetic_cpu) """
    env_name: str = "3m"
    """ Name of the environment"""
    env_family: str = "mpe"
    """ Env family when using pz"""
    agent_ids: bool = True
    """ Include id (one-hot vector) at the agent of the observations"""
    batch_size: int = 3
    """ Number of episodes to collect in each rollout"""
    actor_hidden_dim: int = 32
    """ Hidden dimension of actor network"""
    actor_num_layers: int = 1
    """ Number of hidden layers of actor network"""
    critic_hidden_dim: int = 64
    """ Hidden dimension of critic network"""
    critic_num_layers: int = 1
    """ Number of hidden layers of critic network"""
    optimizer: str = "Adam"
    """ The optimizer (torch.optim class name with its defaults: Adam, AdamW, SGD, RMSprop have a HIP implementation)"""
    learning_rate_actor: float = 0.0008
    """ Learning rate for the actor"""
    learning_rate_critic: float = 0.0008
    """ Learning rate for the critic"""
    total_timesteps: int = 1000000
    """ Total steps in the environment during training"""
    gamma: float = 0.99
    """ Discount factor"""
    td_lambda: float = 0.95
    """ TD(lambda) discount factor"""
    normalize_reward: bool = False
    """ Normalize the rewards if True"""
    normalize_advantage: bool = False
    """ Normalize the advantage if True"""
    normalize_return: bool = False
    """ Normalize the returns if True"""
    epochs: int = 3
    """ Number of training epochs"""
    ppo_clip: float = 0.2
    """ PPO clipping factor """
    entropy_coef: float = 0.001
    """ Entropy coefficient """
    clip_gradients: float = -1
    """ 0< for no clipping and 0> if clipping at clip_gradients"""
    tbptt: int = 10
    """ Chunck size for Truncated Backpropagation Through Time tbptt (recurrent scripts only)"""
    log_every: int = 10
    """ Logging steps """
    eval_steps: int = 50
    """ Evaluate the policy each eval_steps training steps"""
    num_eval_ep: int = 10
    """ Number of evaluation episodes"""
    use_wnb: bool = False
    """ Logging to Weights & Biases if True"""
    wnb_project: str = ""
    """ Weights & Biases project name"""
    wnb_entity: str = ""
    """ Weights & Biases entity name"""
    device: str = "cuda"
    """ Device (this build: cuda only; the reference defaults to cpu)"""
    seed: int = 1
    """ Random seed"""
    # ---- build-only flags
    synthetic_agents: int = 3
    """ [build] number of agents of the synthetic MPE-like env"""
    synthetic_steps: int = 25
    """ [build] fixed episode length (max_cycles) of the synthetic env"""
    synthetic_obs: int = 105
    """ [build] raw obs width of --env_type=synthetic_shape (one-hot agent ids are appended when agent_ids)"""
    synthetic_state: int = 243
    """ [build] global state width of --env_type=synthetic_shape"""
    synthetic_actions: int = 17
    """ [build] number of actions of --env_type=synthetic_shape (action 0 is always available)"""
    synthetic_avail_p: float = 0.7
    """ [build] availability probability of the other actions of --env_type=synthetic_shape"""
    vector_env: str = "pinned"
    """ [build] host-env vectorisation: pinned (shared-memory workers whose blocks are page-locked and copied straight into the device rollout buffer), shm (the same workers, host-side collation) or pipe (the reference's one process + Pipe per env)"""
    env_workers: int = 0
    """ [build] worker processes of the shm vector env (0 = one per host core, at most one per env)"""
    checkpoint: str = ""
    """ [build] path of a checkpoint file: loaded at start if it exists, written at the end (and every checkpoint_every iterations)"""
    checkpoint_every: int = 0
    """ [build] write the checkpoint every N training iterations (0 = only at the end)"""
    greedy_eval: bool = False
    """ [build] evaluate with argmax actions instead of sampling (MLP actors; the reference always samples)"""
    eval_live_envs: int = 0
    """ [build] host-env evaluation: at most this many evaluation envs alive at once per rank, episodes played in waves (0 = all num_eval_ep side by side; 1 = the reference's sequential evaluation, for heavy envs)"""


@dataclass
class ComaArgs:
    """cleanmarl/coma_multienvs.py:19-89 (coma.py differs only in eval_steps / num_eval_ep, see SCRIPT_DEFAULTS)."""
    env_type: str = "smaclite"
    """ Pettingzoo, SMAClite ... (build adds: synthetic, synthetic_cpu, synthetic_shape, synthetic_shape_cpu) """
    env_name: str = "3m"
    """ Name of the environment"""
    env_family: str = "mpe"
    """ Env family when using pz"""
    agent_ids: bool = True
    """ Include id (one-hot vector) at the agent of the observations"""
    batch_size: int = 3
    """ Number of episodes to collect in each rollout"""
    actor_hidden_dim: int = 32
    """ Hidden dimension of actor network"""
    actor_num_layers: int = 1
    """ Number of hidden layers of actor network"""
    critic_hidden_dim: int = 128
    """ Hidden dimension of critic network (<= 64: fused kernels + factored critic input; 65..256: layered schedule)"""
    critic_num_layers: int = 1
    """ Number of hidden layers of critic network"""
    optimizer: str = "Adam"
    """ The optimizer (torch.optim class name with its defaults: Adam, AdamW, SGD, RMSprop have a HIP implementation)"""
    learning_rate_actor: float = 0.0005
    """ Learning rate for the actor"""
    learning_rate_critic: float = 0.0005
    """ Learning rate for the critic"""
    total_timesteps: int = 1000000
    """ Total steps in the environment during training"""
    gamma: float = 0.99
    """ Discount factor"""
    td_lambda: float = 0.8
    """ TD(lambda) discount factor"""
    normalize_reward: bool = False
    """ Normalize the rewards if True"""
    normalize_advantage: bool = True
    """ Normalize the advantage if True"""
    normalize_return: bool = False
    """ Normalize the returns if True"""
    target_network_update_freq: int = 1
    """ Update the target network each target_network_update_freq step in the environment"""
    polyak: float = 0.005
    """ Polyak coefficient when using polyak averaging for target network update"""
    entropy_coef: float = 0.001
    """ Entropy coefficient """
    use_tdlamda: bool = True
    """ Use TD(lambda) as a target for the critic, if False use n-step returns (n=nsteps) """
    nsteps: int = 1
    """ number of stpes when using n-step returns as a target for the critic"""
    start_e: float = 0.5
    """ The starting value of epsilon. See Architecture & Training in COMA's paper Sec. 5"""
    end_e: float = 0.002
    """ The end value of epsilon. See Architecture & Training in COMA's paper Sec. 5"""
    exploration_fraction: float = 750
    """ The number of training steps it takes from to go from start_e to  end_e"""
    clip_gradients: float = -1
    """ 0< for no clipping and 0> if clipping at clip_gradients"""
    log_every: int = 10
    """ Log rollout stats every log_every episode"""
    eval_steps: int = 10
    """ Evaluate the policy each eval_steps training steps"""
    num_eval_ep: int = 10
    """ Number of evaluation episodes"""
    use_wnb: bool = False
    """ Logging to Weights & Biases if True"""
    wnb_project: str = ""
    """ Weights & Biases project name"""
    wnb_entity: str = ""
    """ Weights & Biases entity name"""
    device: str = "cuda"
    """ Device (this build: cuda only; the reference defaults to cpu)"""
    seed: int = 1
    """ Random seed"""
    # ---- build-only flags (same meaning as in Args)
    synthetic_agents: int = 3
    synthetic_steps: int = 25
    synthetic_obs: int = 105
    synthetic_state: int = 243
    synthetic_actions: int = 17
    synthetic_avail_p: float = 0.7
    vector_env: str = "pinned"
    env_workers: int = 0
    checkpoint: str = ""
    checkpoint_every: int = 0
    greedy_eval: bool = False
    """ [build] evaluate with argmax actions instead of sampling with eps = 0 (the reference samples, coma_multienvs.py:703-709)"""
    eval_live_envs: int = 0
    """ [build] host-env evaluation: at most this many evaluation envs alive at once per rank (0 = all side by side; 1 = sequential)"""


# per-script default overrides
SCRIPT_DEFAULTS = {
    "mappo_multienvs": dict(),
    "ippo_multienvs": dict(critic_hidden_dim=32),
    "mappo_lstm_multienvs": dict(num_eval_ep=5, tbptt=10),
    "ippo_lstm_multienvs": dict(critic_hidden_dim=32, optimizer="AdamW", tbptt=5),
    # single-environment front-ends (cleanmarl/mappo.py:18-77, ippo.py, mappo_lstm.py, ippo_lstm.py)
    "mappo": dict(eval_steps=10),
    "ippo": dict(critic_hidden_dim=32),
    "mappo_lstm": dict(num_eval_ep=5, tbptt=10),
    "ippo_lstm": dict(critic_hidden_dim=32, tbptt=5),
    # COMA (cleanmarl/coma_multienvs.py:77-80, coma.py:76-79)
    "coma_multienvs": dict(),
    "coma": dict(eval_steps=50, num_eval_ep=5),
}


def args_class(script):
    return ComaArgs if script.startswith("coma") else Args


def _str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("1", "true", "t", "yes", "y"):
        return True
    if v.lower() in ("0", "false", "f", "no", "n"):
        return False
    raise argparse.ArgumentTypeError(f"boolean expected, got {v!r}")


def build_parser(script):
    defaults = dict(SCRIPT_DEFAULTS[script])
    p = argparse.ArgumentParser(prog=script + ".py", description=f"MI355X-native {script} (cleanmarl CLI surface)")
    for f in fields(args_class(script)):
        default = defaults.get(f.name, f.default)
        names = ["--" + f.name]
        if "_" in f.name:
            names.append("--" + f.name.replace("_", "-"))
        if f.type is bool or isinstance(f.default, bool):
            p.add_argument(*names, dest=f.name, nargs="?", const=True, default=default, type=_str2bool)
            no = ["--no-" + f.name] + (["--no-" + f.name.replace("_", "-")] if "_" in f.name else [])
            p.add_argument(*no, dest=f.name, action="store_false", help=argparse.SUPPRESS)
        else:
            p.add_argument(*names, dest=f.name, type=f.type if isinstance(f.type, type) else type(f.default), default=default)
    return p


def parse_args(script, argv=None):
    ns = build_parser(script).parse_args(argv)
    return args_class(script)(**vars(ns))
