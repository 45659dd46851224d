#!/usr/bin/env python3
"""Does a HIP graph of K consecutive one-launch env steps close the dispatch gaps between them?  Headline workload, K = 20:
eager enqueue (what play_games does) against one captured graph of the same 20 launches, replayed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import stochastic_muzero_amd as smz
from importlib import import_module
mcts_mod, model_mod, envs_mod, sp = (import_module("stochastic-muzero_amd." + m) for m in ("mcts", "model", "envs", "selfplay"))
B, K = 4096, 20
model = model_mod.Muzero.from_arrays(os.path.join(ROOT, "tests", "golden", "weights_ckpt421.npz"))
heads = model.heads("cuda:0")
env = envs_mod.CartPoleVec(B, "cuda:0", seed=0); env.reset()
m = mcts_mod.BatchedMCTS(B, num_simulations=50, discount=0.999, root_exploration_fraction=0.1, use_graph=False)
m.seed(np.arange(B, dtype=np.uint64))
chunk = sp.TrajectoryChunk(K, B, 4, 2, "cuda:0")
sp.play_games(env, heads, m, 1.0, K, chunk=chunk); torch.cuda.synchronize()
def timed(fn, reps=30):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return np.median(ts)
eager = timed(lambda: sp.play_games(env, heads, m, 1.0, K, chunk=chunk))
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    sp.play_games(env, heads, m, 1.0, K, chunk=chunk)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    sp.play_games(env, heads, m, 1.0, K, chunk=chunk)
graph = timed(lambda: g.replay())
print(f"eager {eager / K * 1e3:.4f} ms/step ({B * 50 * K / eager / 1e6:.1f} M sims/s)   graph {graph / K * 1e3:.4f} ms/step ({B * 50 * K / graph / 1e6:.1f} M sims/s)")
