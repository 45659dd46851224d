import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import stochastic_muzero_amd
from importlib import import_module
mcts_mod = import_module("stochastic-muzero_amd.mcts"); model_mod = import_module("stochastic-muzero_amd.model")
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
model = model_mod.Muzero.from_arrays(R + "/tests/golden/weights_ckpt421.npz")
heads = model.heads("cuda:0", backend="hip")
B = 4096
m = mcts_mod.BatchedMCTS(B, num_simulations=50, maxium_action_sample=2, discount=0.999, use_graph=False, fused=True, single_launch=False)
m.seed(np.arange(B, dtype=np.uint64))
obs = torch.rand(B, 4).cuda() * 0.1
eng = m.run(obs, heads)
torch.cuda.synchronize()
hidden, policy = heads.initial(obs)
for mode in ("sleep", "nosleep"):
    eng.root_init(hidden, policy, train=True)
    eng.select(want_parent_hidden=False)
    if mode == "sleep":
        torch.cuda._sleep(20_000_000)
    pairs = []
    for s in range(49):
        o = heads.recurrent(eng)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.expand_backup_select(*o, want_parent_hidden=False); e1.record()
        pairs.append((e0, e1))
    eng.expand_backup(*heads.recurrent(eng))
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in pairs]) * 1e3
    print(mode, "mean", t.mean(), "median", np.median(t), "first", t[:6].round(1), "last", t[-4:].round(1))
