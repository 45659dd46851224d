#!/usr/bin/env python3
"""Where a host-env step's time goes (envs.HostVecEnv with worker processes): sweeps the worker count, with and without a
search between the steps.  python tools/hostenv_probe.py [envs] [steps]"""
import os
import sys
import time
from importlib import import_module

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stochastic_muzero_amd  # noqa: E402,F401

envs_mod = import_module("stochastic-muzero_amd.envs")
sp = import_module("stochastic-muzero_amd.selfplay")
mcts_mod = import_module("stochastic-muzero_amd.mcts")
model_mod = import_module("stochastic-muzero_amd.model")

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
model = model_mod.Muzero.from_arrays(os.path.join(ROOT, "tests", "golden", "weights_ckpt421.npz"))
heads = model.heads("cuda:0")
for W in [int(x) for x in os.environ.get("WORKERS", "0,4,8,12,14,16,24").split(",")]:
    env = envs_mod.HostVecEnv([envs_mod.HostCartPole for _ in range(B)], 4, 2, "cuda:0", on_end="reset", workers=W)
    env.reset()
    act = torch.zeros(B, dtype=torch.int32, device="cuda")
    for _ in range(10):
        env.step(act)
    torch.cuda.synchronize()
    env.transfer_seconds = env.host_step_seconds = 0.0
    t0 = time.perf_counter()
    for _ in range(N):
        env.step(act)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    line = f"workers {W:4d}: bare step {1e3 * dt:7.3f} ms (wait actions {1e3 * env.transfer_seconds / N:.3f}, host step {1e3 * env.host_step_seconds / N:.3f})"
    m = mcts_mod.BatchedMCTS(B, num_simulations=50, discount=0.999, root_exploration_fraction=0.1, use_graph=False)
    m.seed(np.arange(B, dtype=np.uint64))
    sp.play_games(env, heads, m, 1.0, 8)
    torch.cuda.synchronize()
    env.transfer_seconds = env.host_step_seconds = 0.0
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        sp.play_games(env, heads, m, 1.0, 32)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 32)
    n = 5 * 32
    print(line + f" | with search: {1e3 * np.median(ts):7.3f} ms/step (min {1e3 * min(ts):.3f} max {1e3 * max(ts):.3f}; wait actions "
          f"{1e3 * env.transfer_seconds / n:.3f}, host step {1e3 * env.host_step_seconds / n:.3f}) = {B * 50 / np.median(ts) / 1e6:.1f} M sims/s", flush=True)
    env.close()
