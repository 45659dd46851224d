#!/usr/bin/env python3
"""Phase stamps inside the production LDS-resident search kernel (variant builds with -DSMZ_BPS_PROBE; tools/bps_probe.sh).
python tools/bps_probe.py [envs] [waves per workgroup]  -- prints cycles per wavefront and simulation round."""
import os
import sys
from importlib import import_module

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stochastic_muzero_amd  # noqa: E402,F401

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
if len(sys.argv) > 2:
    os.environ["SMZ_SEARCH_WAVES"] = sys.argv[2]
mcts_mod, model_mod = import_module("stochastic-muzero_amd.mcts"), import_module("stochastic-muzero_amd.model")
model = model_mod.Muzero.from_arrays(os.path.join(ROOT, "tests", "golden", "weights_ckpt421.npz"))
heads = model.heads("cuda:0")
sims, reps = 50, 20
m = mcts_mod.BatchedMCTS(B, num_simulations=sims, discount=0.999, root_exploration_fraction=0.1, use_graph=False, single_launch=True)
m.seed(np.arange(B, dtype=np.uint64))
obs = torch.from_numpy(np.random.RandomState(0).uniform(-0.05, 0.05, (B, 4)).astype(np.float32)).cuda()
e = m.run(obs, heads, train=True, act_temperature=1.0)
e.read_stats(reset=True)
for _ in range(reps):
    e = m.run(obs, heads, train=True, act_temperature=1.0)
torch.cuda.synchronize()
print("kernel:", e.last_kernel(), "| waves:", B // 2, "| rounds per wave:", sims * reps, flush=True)
e.read_stats(reset=True)          # (the library prints the summed stamps; divide by waves x rounds)
