#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -1
python - <<'PY'
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import stochastic_muzero_amd
from importlib import import_module
mcts_mod = import_module("stochastic-muzero_amd.mcts"); model_mod = import_module("stochastic-muzero_amd.model")
for w in ("weights_lunar_L2", "weights_wide_A11", "weights_ckpt421"):
    model = model_mod.Muzero.from_arrays("tests/golden/%s.npz" % w)
    heads = model.heads("cuda:0", backend="hip")
    B = 4096
    m = mcts_mod.BatchedMCTS(B, num_simulations=50, maxium_action_sample=2, discount=0.999, root_exploration_fraction=0.1)
    m.seed(np.arange(B, dtype=np.uint64))
    obs = torch.randn(B, model.observation_dimension, generator=torch.Generator().manual_seed(0)).mul(0.3).cuda()
    for _ in range(3): m.run(obs, heads)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m.run(obs, heads)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(w, "%.3f ms/search  %.1f M sims/s" % (dt * 1e3, B * 50 / dt / 1e6))
PY
