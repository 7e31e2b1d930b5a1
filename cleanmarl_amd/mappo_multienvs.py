#!/usr/bin/env python3
"""MI355X-native drop-in for the reference script cleanmarl/mappo_multienvs.py (same flags, same TensorBoard tags).

    python cleanmarl_amd/mappo_multienvs.py --env_type=pz --env_family=mpe --env_name=simple_spread_v3 --batch_size=4
    python cleanmarl_amd/mappo_multienvs.py --env_type=synthetic --synthetic_agents=8 --synthetic_steps=128 --batch_size=4096
    torchrun --nproc-per-node 8 cleanmarl_amd/mappo_multienvs.py --env_type=synthetic ...     # env-sharded, RCCL grad all-reduce
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cleanmarl_amd.driver import run  # noqa: E402

if __name__ == "__main__":
    run("mappo_multienvs")
