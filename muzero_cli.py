#!/usr/bin/env python3
"""Command line surface of the self-play path, keyed like the reference's CLI.

    python muzero_cli.py play       config/experiment_421_config.json [--envs N]
    python muzero_cli.py benchmark  config/experiment_421_config.json [--envs N]
    python muzero_cli.py train      config/experiment_421_config.json [--iterations I] [--envs N]

Same argv conventions as the reference (mode words and the config path are found by substring, muzero_cli.py:13-25),
same JSON sections and keys (game, random_seed, muzero, replaybuffer, monte_carlo_tree_search, gameplay,
learning_cycle, play_game_from_checkpoint).  What runs is this engine's batched GPU self-play:
  play / benchmark : loads `model_checkpoint/{model_tag}_muzero_*` (reference file layout) and plays `--envs` games
                     in parallel for `game_iter` steps; benchmark plays 100 trials' worth (muzero_cli.py:203).
  train            : the SELF-PLAY half of learning_cycle (self_play.py:245-271): games -> replay buffer records.
                     Optimisation steps are outside this engine's scope: with --require-training the command fails
                     instead of silently skipping them.
Differences kept visible rather than copied: the reference passes `maxium_action_sample` as the number of
simulations to `play` (muzero_cli.py:187,219); here `num_simulations` from the config is used unless
--reference-play-sims is given.
"""
import json
import sys


def parse_argv(argv):
    lower = [a.lower() for a in argv]
    cfg = [a for a in argv if "config" in a and a.endswith(".json")] or [a for a in argv if "config" in a and not a.endswith(".py")]
    modes = {m: any(m in a for a in lower if not a.endswith(".json") and not a.endswith(".py"))
             for m in ("train", "play", "report", "benchmark", "human_buffer")}

    def opt(name, default, cast=int):
        for i, a in enumerate(argv):
            if a == name and i + 1 < len(argv):
                return cast(argv[i + 1])
        return default
    opts = dict(envs=opt("--envs", None), iterations=opt("--iterations", None),
                require_training="--require-training" in argv, reference_play_sims="--reference-play-sims" in argv,
                checkpoint_dir=opt("--checkpoint-dir", "model_checkpoint", str), steps=opt("--steps", None))
    if not modes["human_buffer"]:
        if not cfg:
            raise Exception("Specify a config directory and folder such as: config/config_file.json  "
                            "Example: python muzero_cli.py train config/config_file.json | "
                            "python muzero_cli.py play config/config_file.json")
        if not (modes["train"] or modes["play"] or modes["report"] or modes["benchmark"]):
            raise Exception("Specify a mode such as : train , play , benchmark or any of this combination")
    return modes, (cfg[0] if cfg else None), opts


def mcts_kwargs(config, num_simulations=None):
    m = config["monte_carlo_tree_search"]
    return dict(pb_c_base=m["pb_c_base"], pb_c_init=m["pb_c_init"], discount=m["discount"],
                root_dirichlet_alpha=m["root_dirichlet_alpha"],
                root_exploration_fraction=m["root_exploration_fraction"],
                num_simulations=m["num_simulations"] if num_simulations is None else num_simulations,
                maxium_action_sample=m["maxium_action_sample"], number_of_player=m["number_of_player"],
                custom_loop=m["custom_loop"])


def make_env(name, num_envs, device, seed, limit=0, on_end="continue"):
    """The vectorised environment of a run: the built-in device-resident CartPole, or -- for any other name -- gymnasium
    environments stepped on the host behind the pinned-memory adapter (envs.HostVecEnv; needs gymnasium installed)."""
    from importlib import import_module
    envs = import_module("stochastic-muzero_amd.envs")
    if name.startswith("CartPole"):
        return envs.CartPoleVec(num_envs, device, seed=seed, on_end=on_end, limit=limit)
    try:
        import gymnasium as gym
    except ImportError:
        raise Exception(f"environment {name!r} needs gymnasium, which is not installed here; built-in: CartPole-v1 "
                        "(host environments go through stochastic-muzero_amd.envs.HostVecEnv)")
    import functools
    import os
    probe = gym.make(name)                                 # the spaces; the envs themselves are built inside the worker processes
    obs_dim, n_actions = int(np_prod(probe.observation_space.shape)), int(probe.action_space.n)
    probe.close()
    workers = int(os.environ.get("SMZ_HOST_WORKERS", min(64, 3 * _host_envs().usable_cores()))) if num_envs >= 64 else 0
    return envs.HostVecEnv([functools.partial(gym.make, name) for _ in range(num_envs)], obs_dim, n_actions, device, env_seed=seed,
                           limit=limit, on_end="reset" if on_end == "continue" else on_end, workers=workers)


def _host_envs():
    from importlib import import_module
    return import_module("stochastic-muzero_amd.host_envs")


def np_prod(shape):
    n = 1
    for v in shape:
        n *= int(v)
    return n


def main(argv):
    modes, cfg_path, opts = parse_argv(argv)
    if modes["human_buffer"] or modes["report"] and not modes["train"]:
        raise Exception("human_buffer / report are interactive or plotting utilities of the reference, not part of this engine")
    with open(cfg_path) as fh:
        config = json.load(fh)
    import numpy as np
    import torch
    import stochastic_muzero_amd  # noqa: F401
    from importlib import import_module
    mcts_mod, model_mod, sp = (import_module("stochastic-muzero_amd." + m) for m in ("mcts", "model", "selfplay"))
    np.random.seed(config["random_seed"]["np_random_seed"])
    torch.manual_seed(config["random_seed"]["torch_manual_seed"])
    device = "cuda:0"
    out = {}
    if modes["play"] or modes["benchmark"]:
        p = config["play_game_from_checkpoint"]
        n_env = opts["envs"] or (100 if modes["benchmark"] else 1)
        model = model_mod.Muzero.from_checkpoint(opts["checkpoint_dir"], tag=p["model_tag"])
        sims = config["monte_carlo_tree_search"]["maxium_action_sample"] if opts["reference_play_sims"] else None
        search = mcts_mod.BatchedMCTS(n_env, **mcts_kwargs(config, sims))
        search.seed(np.arange(n_env, dtype=np.uint64) + np.uint64(config["random_seed"]["np_random_seed"]))
        env = make_env(config["game"]["env"], n_env, device, config["random_seed"]["env_seed"])
        env.reset()
        steps = opts["steps"] or int(min(p["game_iter"], config["gameplay"]["limit_of_game_play"]))
        chunk = sp.play_games(env, model.heads(device), search, p["temperature"], steps,
                              train=bool(p["mcts_with_or_without_dirichlet_noise"]))
        # (the chunk itself says where its observations live: wider than TrajectoryChunk.SPLIT_OBS they are in chunk.obs)
        games = sp.chunk_to_records(chunk, None, env.num_actions, search.discount, limit_of_game_play=steps,
                                    observation_shape=getattr(env, "frame", None))
        rewards = [sum(g.rewards) for g in games]
        out["play"] = dict(games=len(games), mean_reward=float(np.mean(rewards)), max_reward=float(np.max(rewards)),
                           min_reward=float(np.min(rewards)), steps=steps)
        print(f"played {len(games)} games x <= {steps} steps | reward mean {np.mean(rewards):.1f} "
              f"min {np.min(rewards):.0f} max {np.max(rewards):.0f}")
    if modes["train"]:
        lc, mz = config["learning_cycle"], config["muzero"]
        if opts["require_training"]:
            raise Exception("training (Muzero.train) is outside this engine's scope; feed the produced games to the "
                            "reference's ReplayBuffer / Muzero.train")
        n_env = opts["envs"] or max(1, lc["number_of_self_play_before_training"])
        iters = opts["iterations"] or 1
        if mz.get("load"):
            model = model_mod.Muzero.from_checkpoint(opts["checkpoint_dir"], tag=lc["model_tag_number"])
        else:
            assert mz["model_structure"] == "mlp_model", "fresh models are built for mlp_model"
            probe = make_env(config["game"]["env"], 1, device, 0)
            model = model_mod.Muzero(model_structure="mlp_model", observation_space_dimensions=probe.obs_dim,
                                     action_space_dimensions=probe.num_actions,
                                     state_space_dimensions=mz["state_space_dimensions"],
                                     hidden_layer_dimensions=mz["hidden_layer_dimensions"],
                                     number_of_hidden_layer=mz["number_of_hidden_layer"], random_tag=lc["model_tag_number"])
        search = mcts_mod.BatchedMCTS(n_env, **mcts_kwargs(config))
        search.seed(np.arange(n_env, dtype=np.uint64) + np.uint64(config["random_seed"]["np_random_seed"]))
        limit = int(config["gameplay"]["limit_of_game_play"])
        steps = opts["steps"] or limit
        # every env plays game after game inside an iteration's chunk (a finished game restarts at once)
        env = make_env(config["game"]["env"], n_env, device, config["random_seed"]["env_seed"], limit=min(limit, steps),
                       on_end="reset")
        buffer = []

        class _Sink:                          # stands where the replay buffer is (self_play.py:267-268)
            def save_game(self, g):
                buffer.append(g)
        model.save_model = lambda **k: None   # best-model gating writes checkpoints only when training is attached
        epoch_pr, loss, reward, conf = sp.learning_cycle(
            number_of_iteration=iters, number_of_self_play_before_training=lc["number_of_self_play_before_training"],
            number_of_training_before_self_play=0, model_tag_number=lc["model_tag_number"], number_of_worker_selfplay="gpu",
            temperature_type=lc["temperature_type"], verbose=bool(lc["verbose"]), muzero_model=model, gameplay=env,
            monte_carlo_tree_search=search, replay_buffer=_Sink(), steps_per_iteration=steps,
            model_directory=opts["checkpoint_dir"])
        out["train"] = dict(iterations=iters, games=len(buffer), rewards=reward[1:])
    return out


if __name__ == "__main__":
    main(sys.argv[:])
