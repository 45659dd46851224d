"""Per-launch fixed cost of the one-launch env step (smz_search_mlp_act_cartpole): the same loop as bench.py's headline
workload at num_simulations = 0, 1, 2, 10, 25, 50 -- prologue (weights into LDS, root networks, root expansion and noise), act,
env step, record and the write-back of the trees, against the per-simulation slope."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stochastic_muzero_amd  # noqa
from importlib import import_module
P = lambda n: import_module("stochastic-muzero_amd." + n)
envs, sp, mcts_mod, model_mod = P("envs"), P("selfplay"), P("mcts"), P("model")
B, T = 4096, 20
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
model = model_mod.Muzero.from_arrays(os.path.join(ROOT, "tests", "golden", "weights_ckpt421.npz"))
heads = model.heads("cuda:0")
for sims in (0, 1, 2, 10, 25, 50):
    env = envs.CartPoleVec(B, "cuda:0", seed=0)
    env.reset()
    m = mcts_mod.BatchedMCTS(B, num_simulations=sims, discount=0.999, root_exploration_fraction=0.1)
    m.seed(np.arange(B, dtype=np.uint64))
    chunk = sp.TrajectoryChunk(T, B, 4, 2, "cuda:0")
    sp.play_games(env, heads, m, 1.0, T, chunk=chunk); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); sp.play_games(env, heads, m, 1.0, T, chunk=chunk); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / T * 1e6)
    print("sims %2d: %.1f us per env step (median of 5 x %d steps), kernel %s" % (sims, sorted(ts)[2], T, m.engine.last_kernel()))
