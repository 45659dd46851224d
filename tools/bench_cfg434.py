"""Search throughput on the reference's other shipped network shape (config 434: S 61, H 126): the wide tile heads
(heads.HipMlpTileHeads) against the torch-GEMM heads, 4096 and 65 536 trees x 50 simulations.  GPU only."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, "."); import stochastic_muzero_amd
from importlib import import_module
mcts_mod = import_module("stochastic-muzero_amd.mcts"); model_mod = import_module("stochastic-muzero_amd.model")
model = model_mod.Muzero.from_arrays("tests/golden/weights_cfg434shape.npz")
for B in (4096, 65536):
    sims = 50
    obs = torch.from_numpy(np.random.RandomState(0).uniform(-0.05, 0.05, (B, 4)).astype(np.float32)).cuda()
    for backend in ("auto", "torch"):
        heads = model.heads("cuda:0", backend=backend)
        m = mcts_mod.BatchedMCTS(B, num_simulations=sims, discount=0.999, root_exploration_fraction=0.1, use_graph=True)
        m.seed(np.arange(B, dtype=np.uint64))
        for _ in range(2): m.run(obs, heads, train=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): m.run(obs, heads, train=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(B, type(heads).__name__, "search:", round(dt * 1e3, 3), "ms ->", round(B * sims / dt / 1e6, 1), "M simulations/s", flush=True)
