#!/usr/bin/env python3
"""MI355X-native drop-in for the reference script cleanmarl/ippo_lstm.py (single-environment front-end: --batch_size episodes
are collected one after the other from one in-process env; same flags, defaults and TensorBoard tags; the learner is
the one behind ippo_lstm_multienvs.py).

    python cleanmarl_amd/ippo_lstm.py --env_type=pz --env_family=mpe --env_name=simple_spread_v3 --batch_size=4
    python cleanmarl_amd/ippo_lstm.py --env_type=synthetic_cpu --synthetic_agents=3 --synthetic_steps=25
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cleanmarl_amd.driver import run  # noqa: E402

if __name__ == "__main__":
    run("ippo_lstm")
