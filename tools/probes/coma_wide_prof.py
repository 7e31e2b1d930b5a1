"""Three COMA iterations with the reference's default 128-wide critic at config-3 shapes (for rocprofv3 --kernel-trace --stats)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cleanmarl_amd.coma_learner import COMAHParams, COMALearner, coma_critic_input_dim
from cleanmarl_amd.learner import NetSpec, init_params_like_torch
from cleanmarl_amd.rollout import SyntheticSpreadRollout
E, A, T = 4096, 8, 128
dev = torch.device("cuda:0")
roll = SyntheticSpreadRollout(E, A, T, seed=1, agent_ids=True, device=dev, pad=False)  # COMA's kernels read contiguous rows
Do, Ds, K = roll.Do, roll.Ds, roll.K
aspec, cspec = NetSpec(Do, 64, 1, K), NetSpec(coma_critic_input_dim(Do, Ds, A, K), 128, 1, K)
L = COMALearner(aspec, cspec, A, COMAHParams(), dev, init_params_like_torch(aspec), init_params_like_torch(cspec))
b = roll.collect(L.actor, aspec, eps=0.3)
for _ in range(3):
    L.train_iteration(b)
torch.cuda.synchronize()
