R=$GRAFT_REPO_ROOT
cd $R
CM_PROF_WARMUP=100 python tools/phase_prof.py rollout 2048 8 2>&1 | grep -v amdgpu.ids
CM_PROF_WARMUP=100 python tools/phase_prof.py rollout 4096 8 2>&1 | grep -v amdgpu.ids
