"""cleanmarl_amd -- MI355X-native MAPPO / IPPO hot path (rollout buffer -> TD(lambda) scan -> PPO update).

Only what the hot path needs lives here: csrc/ (HIP kernels + the C-ABI of include/cleanmarl_hip.h),
_native.py (ctypes binding), learner.py / rollout.py (host-side sequencing) and thin CLI front-ends that keep
the reference's single-file script names and flags.  See DESIGN.md.
"""
__version__ = "0.1.0"
