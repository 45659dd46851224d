import os, sys, numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import stochastic_muzero_amd
from importlib import import_module
mcts_mod = import_module("stochastic-muzero_amd.mcts"); model_mod = import_module("stochastic-muzero_amd.model")
model = model_mod.Muzero.from_state_dicts(R + "/tests/golden/visionnet_L2_bn.npz")
B = 256
obs = torch.rand(B, 3, 98, 98, generator=torch.Generator().manual_seed(0)).cuda()
res = []
for backend in ("hip", "torch"):
    m = mcts_mod.BatchedMCTS(B, num_simulations=30, discount=0.997, root_exploration_fraction=0.25, use_graph=(backend == "hip"))
    m.seed(np.arange(B, dtype=np.uint64))
    for _ in range(2):
        m.seed(np.arange(B, dtype=np.uint64))
        e = m.run(obs, model.heads("cuda:0", backend=backend), train=True)
    v = e.root_stats()[0]; torch.cuda.synchronize(); res.append(v.cpu().numpy().copy())
same = (res[0] == res[1]).all(1).mean()
print("vision searches identical between HIP kernels and torch modules: %.1f %%" % (100 * same))
