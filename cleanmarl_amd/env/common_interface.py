"""Environment contract of the hot path.

Any multi-agent environment that exposes these members can sit behind the vector envs of this package
(`vector.PipeVectorEnv`, `shm_vector.ShmVectorEnv`) and therefore behind the four training scripts.  The member
names and call signatures are the ones the reference's adapters implement (cleanmarl/env/common_interface.py:5-23
plus the ``n_agents`` attribute its scripts read, cleanmarl/mappo_multienvs.py:369), so an adapter written for the
reference works here unchanged and vice versa.

Shapes (A agents, Do per-agent observation width incl. optional one-hot id, Ds global-state width, K actions):

==========================  =====================================================================================
``n_agents``                int A
``reset(seed=None)``        -> (obs float[A, Do], info dict)
``step(actions[A])``        -> (obs float[A, Do], team_reward float, done bool, truncated bool, info dict)
``get_avail_actions()``     -> int/bool [A, K]; 1 marks a legal action
``get_state()``             -> float[Ds] global state of the CURRENT step (critic input of MAPPO)
``get_obs_size()`` etc.     -> Do, Ds (``get_state_size``), K (``get_action_size``)
``sample()``                -> one random legal action per agent
``close()``                 release resources
==========================  =====================================================================================
"""
import abc


class CommonInterface(abc.ABC):
    n_agents: int = 0

    @abc.abstractmethod
    def reset(self, seed=None):
        ...

    @abc.abstractmethod
    def step(self, actions):
        ...

    @abc.abstractmethod
    def get_avail_actions(self):
        ...

    @abc.abstractmethod
    def get_state(self):
        ...

    @abc.abstractmethod
    def get_obs_size(self):
        ...

    @abc.abstractmethod
    def get_state_size(self):
        ...

    @abc.abstractmethod
    def get_action_size(self):
        ...

    def sample(self):
        raise NotImplementedError(f"{type(self).__name__} does not implement sample()")

    def close(self):
        pass
