#!/usr/bin/env python3
"""Phase stamps of k_search_vision (SMZ_DEBUG_SKIP=16 + statistics on: s_memtime accounting per wavefront; the library prints
the sums on read_stats).  python tools/vision_phase_probe.py [envs]  -- prints cycles per wavefront and simulation round."""
import os
import sys
from importlib import import_module

os.environ["SMZ_DEBUG_SKIP"] = "16"
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stochastic_muzero_amd  # noqa: E402,F401

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mcts_mod, model_mod = import_module("stochastic-muzero_amd.mcts"), import_module("stochastic-muzero_amd.model")
model = model_mod.Muzero.from_state_dicts(os.path.join(ROOT, "tests", "golden", "visionnet_L1_seed0.npz"))
heads = model.heads("cuda:0")
sims, reps = 50, 10
m = mcts_mod.BatchedMCTS(B, num_simulations=sims, discount=0.997, root_exploration_fraction=0.25, use_graph=False, single_launch=True)
m.seed(np.arange(B, dtype=np.uint64))
obs = torch.rand(B, 3, 98, 98, generator=torch.Generator().manual_seed(0)).cuda()
e = m.run(obs, heads, train=True, act_temperature=1.0)
e.enable_stats(True)
e.read_stats(reset=True)
for _ in range(reps):
    e = m.run(obs, heads, train=True, act_temperature=1.0)
torch.cuda.synchronize()
print("kernel:", e.last_kernel(), "| waves:", B, "| rounds per wave:", sims * reps, "| divide the sums by", B * sims * reps, flush=True)
e.read_stats(reset=True)
