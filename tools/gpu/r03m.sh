R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -q -x -k "rollout" 2>&1 | tail -3
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from cleanmarl_amd import _native as N
from cleanmarl_amd.learner import NetSpec, flatten_params, init_params_like_torch
from cleanmarl_amd.rollout import SyntheticSpreadRollout
dev = torch.device("cuda:0")
def run(E, A, T, tile, reps=30):
    N.set_option("rollout_tile", tile)
    torch.manual_seed(3)
    r = SyntheticSpreadRollout(E, A, T, seed=7, device=dev, env_offset=5)
    spec = NetSpec(r.Do, 64, 1, 5)
    p = flatten_params(init_params_like_torch(spec), dev)
    for _ in range(5): r.collect(p, spec, fused=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): r.collect(p, spec, fused=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (E, A) in [(4096, 8), (2048, 8), (1024, 3), (8192, 8)]:
    print(f"E={E:5d} A={A} T=128: " + "  ".join(f"{t} {run(E, A, 128, t):.3f} ms" for t in ("64", "64s")))
PY
R=$GRAFT_REPO_ROOT
cd $R
CM_PROF_WARMUP=100 python tools/phase_prof.py rollout 2048 8 2>&1 | grep -v amdgpu.ids
CM_PROF_WARMUP=100 python tools/phase_prof.py rollout 4096 8 2>&1 | grep -v amdgpu.ids
