"""One rank of the multi-rank learning_cycle test (started by torch.distributed.run from tests/test_gpu_loop.py): every
rank plays its env shard, the learner rank stores the gathered games and "trains" (a stub that perturbs the weights in
place), the new weights are broadcast inside the loop and the next iteration searches with them.  world 1 = the
single-process run the result is compared with.  RCCL when every rank has its own GPU, gloo when they share one."""
import argparse
import os
import sys
from importlib import import_module

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def game_arrays(g):
    return dict(observations=torch.cat(list(g.observations)).numpy(), rewards=np.asarray(g.rewards),
                policies=np.stack(g.policies), actions=np.stack(g.action_history),
                root_values=np.asarray(g.root_values, np.float32), child_visits=np.stack(g.child_visits), done=bool(g.done))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--total", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--sims", type=int, default=8)
    ap.add_argument("--limit", type=int, default=6)
    ap.add_argument("--iterations", type=int, default=3)
    ap.add_argument("--training", type=int, default=1, help="number_of_training_before_self_play")
    ap.add_argument("--pipeline", default="auto", choices=["auto", "off", "on"], help="learning_cycle(pipeline=None | False | True)")
    ap.add_argument("--sliced", type=int, default=0, help="slices of a gather.TrajectoryGather (0: gather_to_learner)")
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    n_dev = torch.cuda.device_count()
    dev = torch.device("cuda", local % n_dev)
    torch.cuda.set_device(dev)
    backend = "nccl" if world <= n_dev else "gloo"
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    import stochastic_muzero_amd  # noqa: F401
    import seam_harness as sh
    mcts_mod, model_mod, envs_mod, sp, g = (import_module("stochastic-muzero_amd." + m)
                                            for m in ("mcts", "model", "envs", "selfplay", "gather"))
    if rank == 0:
        model = model_mod.Muzero.from_arrays(os.path.join(ROOT, "tests", "golden", "weights_ckpt421.npz"))
    else:
        torch.manual_seed(100 + rank)                               # an actor starts from something else entirely
        model = model_mod.Muzero(model_structure="mlp_model", observation_space_dimensions=4, action_space_dimensions=2,
                                 state_space_dimensions=31, hidden_layer_dimensions=64, number_of_hidden_layer=0, random_tag=1)
    calls = dict(train=0, save=[])

    def train(batch):                       # only the learner may get here: an optimiser step, as far as the search can tell
        calls["train"] += 1
        with torch.no_grad():
            model.prediction_function.value[-1].bias.add_(torch.linspace(-2, 2, 31) * calls["train"])
            model.dynamics_function.next_state_normalized[0].bias.add_(0.01 * calls["train"])
        model.store_loss = getattr(model, "store_loss", []) + [[1.0 / calls["train"]]]
        return "prio", "pos"
    model.train = train
    per_iteration = []
    # (save_model closes an iteration's games: learning_cycle stores them, saves, then trains -- self_play.py:266-288)
    model.save_model = lambda **k: (calls["save"].append(k.get("model_update_or_backtrack")), per_iteration.append([]))
    lo, hi = g.shard_range(a.total, rank, world)
    env = envs_mod.CartPoleVec(hi - lo, dev, seed=0, first_env=lo, total_envs=a.total, on_end="reset", limit=a.limit)
    m = mcts_mod.BatchedMCTS(hi - lo, num_simulations=a.sims, discount=0.999, root_exploration_fraction=0.1, device=dev.index)
    m.seed(np.arange(lo, hi, dtype=np.uint64))
    buf = sh.FakeBuffer()
    save_game = buf.save_game
    buf.save_game = lambda gm: (save_game(gm), per_iteration[-1].append(game_arrays(gm)))
    per_iteration.append([])
    epoch_pr, loss, reward, conf = sp.learning_cycle(
        number_of_iteration=a.iterations, number_of_self_play_before_training=1, number_of_training_before_self_play=a.training,
        model_tag_number=1, number_of_worker_selfplay="gpu", temperature_type="static_one_temperature", verbose=False,
        muzero_model=model, gameplay=env, monte_carlo_tree_search=m, replay_buffer=buf, steps_per_iteration=a.steps,
        gather=(g.TrajectoryGather(env.obs_dim, env.num_actions, slices=a.sliced, total_envs=a.total) if a.sliced else g.gather_to_learner) if world > 1 else None,
        pipeline={"auto": None, "off": False, "on": True}[a.pipeline])
    torch.cuda.synchronize(dev)
    final = model.heads(dev).weights.cpu()
    if world > 1:
        every = [None] * world
        dist.all_gather_object(every, dict(rank=rank, train=calls["train"], save=len(calls["save"]), games=len(buf.saved),
                                           reward=[float(r) for r in reward], weights=final))
    else:
        every = [dict(rank=0, train=calls["train"], save=len(calls["save"]), games=len(buf.saved),
                      reward=[float(r) for r in reward], weights=final)]
    if rank == 0:
        torch.save(dict(world=world, backend=backend if world > 1 else None, ranks=every, loss=loss,
                        games=[it for it in per_iteration if it]), os.path.join(a.out, f"learning_w{world}{a.tag}.pt"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
