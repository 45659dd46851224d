"""The self-play LOOP around the search (self_play.py:79-94, 236-288) on the device: one code path for grouped and ungrouped
play, captured graphs that follow weight updates, per-game value targets for chunks with several games per env, and
learning_cycle on several ranks (actors neither store nor train; the learner's new weights are broadcast inside the loop)."""
import os
import socket
import subprocess
import sys
from importlib import import_module

import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pkg(name):
    import stochastic_muzero_amd  # noqa: F401
    return import_module("stochastic-muzero_amd." + name)


def _model():
    return _pkg("model").Muzero.from_arrays(os.path.join(gu.GOLDEN, "weights_ckpt421.npz"))


def _mcts(B, sims=8, **kw):
    return _pkg("mcts").BatchedMCTS(B, num_simulations=sims, discount=0.999, root_exploration_fraction=0.1, **kw)


def _tip(env, rows):
    """poles that are falling and cannot be caught: early terminations"""
    env.state[rows, 2] = 0.2
    env.state[rows, 3] = 3.0
    env.obs.copy_(env.state.float())


@pytest.mark.parametrize("on_end", ["continue", "mask", "reset"])
def test_grouped_play_equals_ungrouped_play(on_end):
    """play_games_grouped (G = 2 stream groups) == play_games env by env, for every end-of-game rule: the two share one
    per-step body (selfplay._play_step), so a masked env stops consuming simulations in both and the records agree."""
    envs_mod, sp = _pkg("envs"), _pkg("selfplay")
    B, T, sims, limit = 96, 20, 8, 9
    heads = _model().heads("cuda:0")

    def env_of(lo, n):
        env = envs_mod.CartPoleVec(n, "cuda:0", seed=5, first_env=lo, total_envs=B, on_end=on_end, limit=limit)
        env.reset()
        _tip(env, [i - lo for i in range(lo, lo + n) if i % 4 == 0])
        return env
    env = env_of(0, B)
    m = _mcts(B, sims, use_graph=False)
    m.seed(np.arange(B, dtype=np.uint64))
    whole = sp.play_games(env, heads, m, 1.0, T).data
    groups = []
    for gi, lo in enumerate((0, B // 2)):
        genv = env_of(lo, B // 2)
        gm = _mcts(B // 2, sims, use_graph=False)
        gm.seed(np.arange(lo, lo + B // 2, dtype=np.uint64))
        groups.append(sp.StreamGroup(genv, _model().heads("cuda:0", instance=gi), gm, T))
    parts = sp.play_games_grouped(groups, 1.0, T)
    torch.cuda.synchronize()
    got = torch.cat([p.data for p in parts], dim=1)
    assert torch.equal(got, whole)
    flags = whole[..., 5].cpu().numpy()
    if on_end == "mask":
        assert (flags == 3).any()                                          # switched-off envs: recorded as "no step"
        for g in groups:                                                   # ... and their trees were skipped: no visits added
            off = g.env.active.cpu().numpy() == 0
            assert off.any() and g.mcts.engine._active is g.env.active
    if on_end == "reset":
        assert (flags == 1).any() and (flags == 2).any() and not (flags == 3).any()


@pytest.mark.parametrize("on_end,B", [("continue", 4096), ("mask", 1000), ("reset", 300)])
def test_one_launch_per_env_step_equals_search_launch_plus_env_launch(on_end, B, monkeypatch):
    """smz_search_mlp_act_cartpole (search + action + env step + record in ONE launch; what bench.py's headline runs) ==
    smz_search_mlp_act followed by smz_cartpole_step_pack / _ctl: trajectory chunk, env state, next observations, game
    bookkeeping and the trees' random streams, bit for bit, for every end-of-game rule (self_play.py:79-94)."""
    envs_mod, sp = _pkg("envs"), _pkg("selfplay")
    T, sims, limit = 12, 10, 7
    heads = _model().heads("cuda:0")
    res = []
    for fused in ("1", "0"):
        monkeypatch.setenv("SMZ_FUSED_ENV_STEP", fused)
        env = envs_mod.CartPoleVec(B, "cuda:0", seed=5, on_end=on_end, limit=0 if on_end == "continue" else limit)
        env.reset()
        _tip(env, list(range(0, B, 5)))
        m = _mcts(B, sims, use_graph=False)
        m.seed(np.arange(B, dtype=np.uint64))
        chunk = sp.play_games(env, heads, m, 1.0, T)
        torch.cuda.synchronize()
        assert m._single is True and m.engine.env_stepped == (fused == "1")
        res.append(dict(chunk=chunk.data.cpu(), state=env.state.cpu(), obs=env.obs.cpu(), reward=env.reward.cpu(),
                        flag=env.terminated.cpu(), count=env.step_count.cpu(), episode=env.episode.cpu(),
                        active=None if env.active is None else env.active.cpu(), visits=m.engine.root_stats()[0].cpu(),
                        rng=[m.engine.get_rng_state(i) for i in (0, 1, B // 2, B - 1)]))
    a, b = res
    for k in ("chunk", "state", "obs", "reward", "flag", "count", "episode", "visits"):
        assert torch.equal(a[k], b[k]), k
    assert (a["active"] is None and b["active"] is None) or torch.equal(a["active"], b["active"])
    for (ka, pa), (kb, pb) in zip(a["rng"], b["rng"]):
        assert np.array_equal(ka, kb) and pa == pb
    flags = a["chunk"][..., 5].numpy()
    assert (flags == 1).any()
    if on_end == "mask":
        assert (flags == 3).any() and (a["active"] == 0).all()
    if on_end == "reset":
        assert (flags == 2).any() and int(a["episode"].min()) >= 1


def test_grouped_play_with_host_envs_records_the_post_step_observation():
    """envs.HostVecEnv(on_end="reset") -- what bench.py --host-env python builds -- through the grouped path: the record of a
    step that ends a game holds the post-step observation (env.record_obs), not the next game's reset one."""
    envs_mod, sp = _pkg("envs"), _pkg("selfplay")
    B, T, sims, limit = 24, 14, 6, 5
    heads = _model().heads("cuda:0")

    def make(lo, n):
        env = envs_mod.HostVecEnv([envs_mod.HostCartPole() for _ in range(n)], 4, 2, "cuda:0", env_seed=3, limit=limit,
                                  on_end="reset", first_env=lo)
        env.reset()
        return env
    env = make(0, B)
    m = _mcts(B, sims, use_graph=False)
    m.seed(np.arange(B, dtype=np.uint64))
    whole = sp.play_games(env, heads, m, 1.0, T).data
    groups = []
    for gi, lo in enumerate((0, B // 2)):
        gm = _mcts(B // 2, sims, use_graph=False)
        gm.seed(np.arange(lo, lo + B // 2, dtype=np.uint64))
        groups.append(sp.StreamGroup(make(lo, B // 2), _model().heads("cuda:0", instance=gi), gm, T))
    parts = sp.play_games_grouped(groups, 1.0, T)
    torch.cuda.synchronize()
    got = torch.cat([p.data for p in parts], dim=1).cpu().numpy()
    assert np.array_equal(got, whole.cpu().numpy())
    # at a limit-stop (flag 2) the record is the stepped state, which cannot be a fresh reset state |x| <= 0.05 everywhere
    ends = np.argwhere(got[..., 5] == 2)
    assert len(ends) >= B
    for t, e in ends:
        assert np.abs(got[t, e, :4]).max() > 0.05


def test_a_reused_search_object_rebuilds_its_graph_after_a_weight_update():
    """the captured graph of the step-wise path points into the heads object (weights, output buffers).
    Muzero.heads() builds a new evaluator after an in-place weight update; the long-lived BatchedMCTS must notice (identity,
    not id()) and capture again, otherwise it replays a graph over freed memory."""
    model = _model()
    B = 64
    obs = torch.randn(B, 4, generator=torch.Generator().manual_seed(0)).mul(0.05).cuda()
    m = _mcts(B, 8, use_graph=True, single_launch=False)
    m.seed(np.arange(B, dtype=np.uint64))

    def run():
        m.engine and m.engine.seed(np.arange(B, dtype=np.uint64))
        e = m.run(obs, model.heads("cuda:0"), train=False)
        rv = e.root_stats()[2]
        torch.cuda.synchronize()
        return rv.cpu().numpy().copy()
    before = run()
    g0 = m._graph
    assert g0 is not None and np.array_equal(before, run()) and m._graph is g0
    with torch.no_grad():
        model.prediction_function.value[-1].bias.add_(torch.linspace(-3, 3, 31))
        model.afterstate_prediction_function.value[-1].bias.add_(torch.linspace(-3, 3, 31))
    after = run()
    assert m._graph is not g0 and m._graph_heads is model.heads("cuda:0")
    fresh = _mcts(B, 8, use_graph=False, single_launch=False)
    fresh.seed(np.arange(B, dtype=np.uint64))
    e = fresh.run(obs, model.heads("cuda:0"), train=False)
    want = e.root_stats()[2]
    torch.cuda.synchronize()
    assert not np.array_equal(before, after) and np.array_equal(after, want.cpu().numpy())


@pytest.mark.parametrize("td", [1, 5, 30])
def test_value_targets_of_chunks_with_several_games_per_env(td):
    """on_end="reset" puts several games per env into one chunk.  chunk_targets(after_end="new_game")
    (smz_traj_targets_games) gives every game its own n-step targets and priorities -- equal, bit for bit, to
    chunk_to_games(after_end="new_game") + GameRecord.make_target / make_priority (game.py:291-337) per game; rows of a
    switched-off env (flag 3) get none."""
    sp = _pkg("selfplay")
    T, B, obs, A, disc = 40, 29, 4, 2, 0.997
    g = np.random.RandomState(td)
    F = obs + 3 * A + 3
    d = np.zeros((T, B, F))
    d[..., :obs] = g.randn(T, B, obs).astype(np.float32)
    d[..., obs] = g.randn(T, B).astype(np.float32)
    d[..., obs + 1] = g.choice([0, 1, 2], size=(T, B), p=[0.88, 0.08, 0.04])
    d[:, 3, obs + 1] = 0                                                      # one env: a single unfinished game
    d[T - 1, 4, obs + 1] = 1                                                  # a game that ends exactly with the chunk
    d[10:, 5, obs + 1] = 3; d[9, 5, obs + 1] = 1                              # switched off after its first game
    d[..., obs + 2 + 2 * A] = (10 * g.randn(T, B)).astype(np.float32)
    dev = torch.from_numpy(d).cuda()
    length, target, err, game_end = sp.chunk_targets(dev, obs, A, disc, td, after_end="new_game", return_game_end=True)
    torch.cuda.synchronize()
    length, target, err, game_end = (x.cpu().numpy() for x in (length, target, err, game_end))
    n_games = 0
    for e in range(B):
        games = sp.chunk_to_games(d[:, e:e + 1], obs, A, disc, after_end="new_game", keep_partial=True)
        t0 = 0
        assert length[e] == games[0].game_length
        for game in games:
            n = game.game_length
            assert (game_end[t0:t0 + n, e] == t0 + n).all()
            pos, _ = game.make_priority(td)
            assert np.array_equal(err[t0:t0 + n, e], np.asarray(pos, np.float64)), (e, t0)
            want = [np.float64(game.make_target(t, 1, td)[0][0]) for t in range(n)]
            assert np.array_equal(target[t0:t0 + n, e], np.asarray(want)), (e, t0)
            t0 += n
            n_games += 1
        assert (game_end[t0:, e] == -1).all() and (target[t0:, e] == 0).all()       # only flag-3 rows are left
        assert t0 == T or e == 5
    assert n_games > 3 * B
    # after_end="drop" is the old single-game cut (smz_traj_targets)
    l1, t1, e1 = sp.chunk_targets(dev, obs, A, disc, td)
    lib = _pkg("_lib")
    import ctypes as C
    l0 = torch.empty(B, dtype=torch.int32, device="cuda"); t0_ = torch.empty(T, B, dtype=torch.float64, device="cuda")
    e0 = torch.empty_like(t0_)
    pows = torch.tensor([disc ** i for i in range(td + 1)], dtype=torch.float64).cuda()
    P = lambda x: C.c_void_p(x.data_ptr())
    lib.check(lib.load().smz_traj_targets(P(dev), T, obs, A, B, td, P(pows), 0, P(l0), P(t0_), P(e0), None))
    torch.cuda.synchronize()
    assert torch.equal(l0, l1) and torch.equal(t0_, t1) and torch.equal(e0, e1)


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _clean_env():
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e["OMP_NUM_THREADS"] = "1"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return e


def test_learning_cycle_on_two_ranks_trains_on_the_learner_and_broadcasts_inside_the_loop(tmp_path):
    """VERDICT r2 #1b (self_play.py:236-271, 285-288).  Two rank processes run learning_cycle(gather=...) for three
    iterations with a stub train() that perturbs the learner's weights: the actor rank neither stores games nor saves nor
    trains (and does not divide by zero), every rank ends with the learner's trained weights, and the games of EVERY
    iteration equal those of the same job on one rank -- iteration 2's and 3's only can if the actor searched with the
    weights broadcast after the previous training phase (it starts from different random weights altogether)."""
    worker = os.path.join(ROOT, "tests", "dist_learning_worker.py")
    args = ["--out", str(tmp_path), "--total", "64", "--steps", "10", "--sims", "8", "--limit", "6", "--iterations", "3"]
    r = subprocess.run([sys.executable, worker] + args, env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), worker] + args
    r = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    one = torch.load(os.path.join(tmp_path, "learning_w1.pt"), weights_only=False)
    two = torch.load(os.path.join(tmp_path, "learning_w2.pt"), weights_only=False)
    learner, actor = two["ranks"]
    assert (learner["train"], learner["save"]) == (3, 3) and learner["games"] == one["ranks"][0]["games"] > 0
    assert (actor["train"], actor["save"], actor["games"]) == (0, 0, 0)
    assert all(np.isnan(x) for x in actor["reward"][1:]) and learner["reward"] == one["ranks"][0]["reward"]
    assert torch.equal(actor["weights"], learner["weights"]) and torch.equal(learner["weights"], one["ranks"][0]["weights"])
    assert two["loss"] == one["loss"] == [1.0, 0.5, 1.0 / 3]
    assert len(two["games"]) == len(one["games"]) == 3
    for it, (ga, gb) in enumerate(zip(two["games"], one["games"])):
        assert len(ga) == len(gb) and len(ga) >= 64, it
        for x, y in zip(ga, gb):
            for k in x:
                assert np.array_equal(x[k], y[k]), (it, k)
    print("learning_cycle on 2 ranks over", two["backend"])


def test_pipelined_learning_cycle_equals_the_synchronous_loop_on_one_and_two_ranks(tmp_path):
    """with number_of_training_before_self_play = 0 learning_cycle drives self_play_iterations (the search
    of iteration k + 1 enqueued before iteration k's games are built, stored and the model saved); the games of EVERY
    iteration, the rewards and the model saves equal the synchronous loop's (pipeline=False) -- on one rank and on two ranks
    (sliced exchange), and the two-rank run equals the one-rank run."""
    worker = os.path.join(ROOT, "tests", "dist_learning_worker.py")
    base = ["--out", str(tmp_path), "--total", "64", "--steps", "10", "--sims", "8", "--limit", "6", "--iterations", "4", "--training", "0"]
    runs = {}
    for world in (1, 2):
        for pipe in ("off", "auto"):
            tag = f"_{pipe}"
            args = base + ["--pipeline", pipe, "--tag", tag] + (["--sliced", "2"] if world == 2 else [])
            cmd = [sys.executable, worker] + args if world == 1 else \
                [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                 "--master-port", str(_port()), worker] + args
            r = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
            runs[(world, pipe)] = torch.load(os.path.join(tmp_path, f"learning_w{world}{tag}.pt"), weights_only=False)
    ref = runs[(1, "off")]
    assert len(ref["games"]) == 4 and ref["ranks"][0]["save"] == 4 and ref["ranks"][0]["train"] == 0
    for key, got in runs.items():
        lead = got["ranks"][0]
        assert lead["reward"] == ref["ranks"][0]["reward"] and lead["save"] == 4 and lead["games"] == ref["ranks"][0]["games"], key
        assert len(got["games"]) == 4, key
        for it, (ga, gb) in enumerate(zip(got["games"], ref["games"])):
            assert len(ga) == len(gb) and len(ga) >= 64, (key, it)
            for x, y in zip(ga, gb):
                for k in x:
                    assert np.array_equal(x[k], y[k]), (key, it, k)
        if key[0] == 2:
            actor = got["ranks"][1]
            assert (actor["train"], actor["save"], actor["games"]) == (0, 0, 0)
