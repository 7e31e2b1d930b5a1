import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cleanmarl_amd.learner import HParams, NetSpec, PPOLearner, init_params_like_torch
from cleanmarl_amd.rollout import SyntheticSpreadRollout
E, A, T = 4096, 8, 128
dev = torch.device("cuda:0")
roll = SyntheticSpreadRollout(E, A, T, seed=1, agent_ids=True, device=dev)
aspec, cspec = NetSpec(roll.Do, 128, 1, roll.K), NetSpec(roll.Ds, 128, 1, 1)
L = PPOLearner("mappo", aspec, cspec, A, HParams(), dev, init_params_like_torch(aspec), init_params_like_torch(cspec))
b = roll.collect(L.actor, NetSpec(roll.Do, 64, 1, roll.K)) if False else None
a64 = NetSpec(roll.Do, 64, 1, roll.K)
import cleanmarl_amd.learner as LL
b = roll.collect(LL.flatten_params(init_params_like_torch(a64), dev), a64)
for _ in range(3):
    L.train_iteration(b)
torch.cuda.synchronize()
