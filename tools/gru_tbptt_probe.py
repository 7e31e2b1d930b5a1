import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cleanmarl_amd.gru import GRUPPOLearner, GRUSyntheticRollout
from cleanmarl_amd.learner import HParams, NetSpec
tb = int(sys.argv[1])
dev = torch.device("cuda:0")
E, A, T = 1024, 5, 128
roll = GRUSyntheticRollout(E, A, T, seed=1, agent_ids=True, device=dev)
aspec = NetSpec(roll.Do, 64, 0, roll.K, "gru"); cspec = NetSpec(roll.Ds, 64, 1, 1)
L = GRUPPOLearner("mappo", aspec, cspec, A, HParams(tbptt=tb), dev)
b = roll.collect(L.actor, aspec)
for _ in range(3):
    L.train_iteration(b)
torch.cuda.synchronize()
