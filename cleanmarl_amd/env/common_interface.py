"""Environment API every env adapter implements -- same nine-method surface (+ ``n_agents``) as the
reference's cleanmarl/env/common_interface.py:5-23, so env wrappers written for the reference plug in."""


class CommonInterface(object):
    n_agents = 0

    def step(self, actions):
        """-> (obs[A,Do], team_reward: float, done: bool, truncated: bool, info: dict)"""
        raise NotImplementedError

    def reset(self, seed=None):
        """-> (obs[A,Do], {})"""
        raise NotImplementedError

    def get_avail_actions(self):
        raise NotImplementedError

    def get_action_size(self):
        raise NotImplementedError

    def get_state(self):
        raise NotImplementedError

    def get_state_size(self):
        raise NotImplementedError

    def get_obs_size(self):
        raise NotImplementedError

    def close(self):
        raise NotImplementedError

    def sample(self):
        raise NotImplementedError
