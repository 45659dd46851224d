"""One rank of the N > 1 functional test (started by torch.distributed.run from tests/test_gpu_multirank.py):
weight broadcast learner -> actors, self-play on this rank's env shard, trajectory gather to the learner.
RCCL ("nccl") when every rank has a GPU of its own, gloo when ranks share one (the 1-GPU box)."""
import argparse
import os
import sys
from importlib import import_module

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--total", type=int, default=96)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--sims", type=int, default=8)
    ap.add_argument("--limit", type=int, default=4)
    ap.add_argument("--overlap", type=int, default=0, help="slices of gather.TrajectoryGather (0: one gather_to_learner call)")
    a = ap.parse_args()
    rank, world, local = (int(os.environ[k]) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"))
    n_dev = torch.cuda.device_count()
    backend = "nccl" if world <= n_dev else "gloo"
    dev = torch.device("cuda", local % n_dev)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    import stochastic_muzero_amd  # noqa: F401
    mcts_mod, model_mod, envs_mod, sp, g = (import_module("stochastic-muzero_amd." + m)
                                            for m in ("mcts", "model", "envs", "selfplay", "gather"))
    wpath = os.path.join(ROOT, "tests", "golden", "weights_ckpt421.npz")
    if rank == 0:
        model = model_mod.Muzero.from_arrays(wpath)                 # the learner holds the trained weights
    else:
        torch.manual_seed(100 + rank)                               # an actor starts from something else entirely
        model = model_mod.Muzero(model_structure="mlp_model", observation_space_dimensions=4, action_space_dimensions=2,
                                 state_space_dimensions=31, hidden_layer_dimensions=64, number_of_hidden_layer=0, random_tag=1)
    stale = model.heads(dev)                                        # packed BEFORE the broadcast: must not survive it
    g.broadcast_model(model, src=0, device=dev if backend == "nccl" else None)
    heads = model.heads(dev)
    assert heads is not stale
    lo, hi = g.shard_range(a.total, rank, world)
    env = envs_mod.CartPoleVec(hi - lo, dev, seed=0, first_env=lo, total_envs=a.total, on_end="reset", limit=a.limit)
    env.reset()
    m = mcts_mod.BatchedMCTS(hi - lo, num_simulations=a.sims, discount=0.999, root_exploration_fraction=0.1, device=dev.index,
                             use_graph=False)
    m.seed(np.arange(lo, hi, dtype=np.uint64))
    if a.overlap:
        # the overlapped exchange: the chunk is played in slices, every finished slice's rows travel (compact wire format) while
        # the next slice is searched; the learner reassembles the chunk
        tg = g.TrajectoryGather(env.obs_dim, env.num_actions, slices=a.overlap)
        k = min(a.overlap, a.steps)
        cuts = [a.steps * i // k for i in range(k + 1)]
        chunk = sp.TrajectoryChunk(a.steps, env.B, env.obs_dim, env.num_actions, env.device)
        for i in range(k):
            sp.play_games(env, heads, m, 1.0, cuts[i + 1] - cuts[i], chunk=chunk, t0=cuts[i])
            tg.start(chunk.data[cuts[i]:cuts[i + 1]])
        got = tg.finish()
        parts = None if got is None else [got[0]]
    else:
        chunk = sp.play_games(env, heads, m, 1.0, a.steps)
        parts = g.gather_to_learner(chunk.data)
    torch.cuda.synchronize(dev)
    ones = torch.ones(1, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(ones)                                           # the rank count as the collective itself sees it
    single = [None] * world
    dist.all_gather_object(single, m._single is True)
    if rank == 0:
        assert parts is not None and len(parts) == (1 if a.overlap else world)
        torch.save(dict(data=torch.cat([p.cpu() for p in parts], dim=1), backend=backend, world=world,
                        ranks_seen_by_collective=int(ones.item()), single_launch=single,
                        weights=heads.weights.cpu()), os.path.join(a.out, "gathered.pt"))
    else:
        assert parts is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
