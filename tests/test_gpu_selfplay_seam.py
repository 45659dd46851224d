"""The self-play seam over the GPU engine: the reference's call sequences (self_play.py:63-98, 168-306) with this package's
Monte_carlo_tree_search / BatchedMCTS underneath, game termination handled on the device (finished games consume no
simulations), host-resident environments through the pinned-memory adapter, and the stored-game reanalyse replay."""
import os
import random
from importlib import import_module

import numpy as np
import pytest
import torch

import golden_util as gu
import seam_harness as sh

pytestmark = pytest.mark.gpu


def _pkg(name):
    import stochastic_muzero_amd  # noqa: F401
    return import_module("stochastic-muzero_amd." + name)


def _search(cfg):
    return _pkg("mcts").Monte_carlo_tree_search(
        pb_c_base=int(cfg["pb_c_base"]), pb_c_init=float(cfg["pb_c_init"]), discount=float(cfg["discount"]),
        root_dirichlet_alpha=float(cfg["root_dirichlet_alpha"]), root_exploration_fraction=float(cfg["root_exploration_fraction"]),
        num_simulations=int(cfg["num_simulations"]), maxium_action_sample=int(cfg["maxium_action_sample"]))


def _game(env, limit, obs_dim=4, priority_scale=0.5):
    return _pkg("game").Game(gym_env=env, discount=0.999, limit_of_game_play=limit, observation_dimension=obs_dim,
                             action_dimension=2, rgb_observation=False, action_map=[0, 1], priority_scale=priority_scale)


@pytest.mark.parametrize("name", gu.SELFPLAY_FIXTURES + ["game_illegal_moves"])
def test_play_game_on_the_gpu_tree_reproduces_the_references_games(name):
    """play_game + Game + Monte_carlo_tree_search (tree on the GPU, numpy's global stream carried in and out) replay the
    reference's own games bit for bit, including the illegal-move game."""
    sp = _pkg("selfplay")
    cfg, data = gu.load(name)
    illegal = name == "game_illegal_moves"
    env = sh.PickyWalk() if illegal else sh.MathCartPole()
    random.seed(int(data["seed"]))
    np.random.seed(int(data["seed"]))
    g = sp.play_game(environment=_game(env, int(data["limit"]), obs_dim=1 if illegal else 4, priority_scale=1 if illegal else 0.5),
                     model=sh.TapePlayer(data), monte_carlo_tree_search=_search(cfg), temperature=float(data["temperature"]),
                     replay_buffer=sh.FakeBuffer())
    sh.assert_game_equals(g, data)
    assert np.random.random_sample() == data["probe"]


@pytest.mark.parametrize("name", ["reanalyse421_sims10_T1", "reanalyse421_sims10_T0"])
def test_reanalyse_branch_on_the_gpu_tree(name):
    sp = _pkg("selfplay")
    cfg, data = gu.load(name)
    stored = _pkg("game").GameRecord(0.999, 2, 0.5, int(data["limit"]))
    stored.observations = [torch.from_numpy(o[None].copy()) for o in data["src_observations"]]
    stored.rewards = [float(r) for r in data["src_rewards"]]
    buf = sh.FakeBuffer(stored, (data["np_key_before_first_search"], data["np_pos_before_first_search"]))
    g = sp.play_game(environment=_game(sh.MathCartPole(), int(data["limit"])), model=sh.TapePlayer(data),
                     monte_carlo_tree_search=_search(cfg), temperature=float(data["temperature"]), replay_buffer=buf)
    sh.assert_game_equals(g, data)
    assert g.reanalyzed and np.random.random_sample() == data["probe"]


def _model():
    return _pkg("model").Muzero.from_arrays(os.path.join(gu.GOLDEN, "weights_ckpt421.npz"))


def _batched(B, sims=8):
    return _pkg("mcts").BatchedMCTS(B, num_simulations=sims, discount=0.999, root_exploration_fraction=0.1, use_graph=False)


@pytest.mark.parametrize("single_launch", [True, False])
def test_finished_games_stop_consuming_simulations(single_launch):
    """on_end="mask": once an env's game is over (terminated, or limit_of_game_play reached) its tree is switched off on
    the device -- no search, no action draw, no env step: its visit counts, root value and random stream stay exactly
    where its last search left them while the other envs go on (self_play.py:79's loop condition, per env)."""
    envs_mod, sp = _pkg("envs"), _pkg("selfplay")
    B, T, sims, limit = 96, 30, 8, 22
    env = envs_mod.CartPoleVec(B, "cuda:0", seed=2, on_end="mask", limit=limit)
    env.reset()
    env.state[:24, 2] = 0.2                       # these poles are falling and cannot be caught: early terminations
    env.state[:24, 3] = 3.0
    env.obs.copy_(env.state.float())
    m = _batched(B, sims)
    m.single_launch = single_launch
    m.seed(np.arange(B, dtype=np.uint64))
    heads = _model().heads("cuda:0")
    chunk = sp.TrajectoryChunk(T, B, 4, 2, "cuda:0")
    snapshots = []
    for t in range(T):
        sp.play_games(env, heads, m, 1.0, 1, chunk=_OneRow(chunk, t))
        torch.cuda.synchronize()
        snapshots.append((env.active.cpu().numpy().copy(), m.engine.root_stats()[0].cpu().numpy().copy(),
                          [m.engine.get_rng_state(i) for i in (0, 5, B - 1)]))
    flags = chunk.data[..., 5].cpu().numpy()
    assert (flags[:, :24] == 1).any(axis=0).all()                              # the tipped poles terminated
    assert set(np.unique(flags)) <= {0.0, 1.0, 2.0, 3.0} and (flags == 3).any()
    ended = np.argmax(flags != 0, axis=0)                                      # first non-zero flag per env
    assert ((flags != 0).any(axis=0)).all() and ended.max() == limit - 1       # everybody stops by the limit
    for e in range(B):
        assert (flags[ended[e] + 1:, e] == 3).all()                            # and never steps again
        assert flags[ended[e], e] == (2 if ended[e] == limit - 1 else 1)
    for t in range(1, T):
        off = snapshots[t - 1][0] == 0                                         # switched off BEFORE step t's search
        assert np.array_equal(snapshots[t][1][off], snapshots[t - 1][1][off])   # trees untouched
        assert (snapshots[t][1][~off].sum(1) == sims).all()
        for k, i in enumerate((0, 5, B - 1)):                                  # ... and so are their random streams
            if off[i]:
                assert np.array_equal(snapshots[t][2][k][0], snapshots[t - 1][2][k][0]) and snapshots[t][2][k][1] == snapshots[t - 1][2][k][1]
    first_off = int(np.where(snapshots[-1][0] == 0)[0][0])
    # simulations actually spent = sum over steps of the envs still playing
    spent = sum(int((flags[t] != 3).sum()) for t in range(T)) * sims
    assert spent == int((ended + 1).sum()) * sims < B * T * sims
    games = sp.chunk_to_games(chunk.data, 4, 2, 0.999, limit_of_game_play=limit)
    assert [g.game_length for g in games] == list(ended + 1) and first_off >= 0
    assert all(g.done == (g.game_length < limit) for g in games)


class _OneRow:
    """A view of a TrajectoryChunk that makes play_games(…, steps=1) write row t."""
    def __init__(self, chunk, t):
        self.T, self.B, self.data, self.obs = 1, chunk.B, chunk.data[t:t + 1], None


def test_restarting_envs_play_game_after_game_inside_a_chunk():
    """on_end="reset": a finished env starts its next game at once from a counter-based reset state (the same on any
    shard), so every simulation of the chunk belongs to some game; learning_cycle's batched branch hands the finished
    games to the replay buffer."""
    envs_mod, sp = _pkg("envs"), _pkg("selfplay")
    B, T, limit = 64, 40, 16
    env = envs_mod.CartPoleVec(B, "cuda:0", seed=3, on_end="reset", limit=limit, first_env=100, total_envs=200)
    m = _batched(B)
    m.seed(np.arange(B, dtype=np.uint64))
    buf = sh.FakeBuffer()
    model = _model()
    model.save_model = lambda **k: None
    epoch_pr, loss, reward, conf = sp.learning_cycle(
        number_of_iteration=1, number_of_self_play_before_training=1, number_of_training_before_self_play=0,
        model_tag_number=1, number_of_worker_selfplay="gpu", temperature_type="static_one_temperature", verbose=False,
        muzero_model=model, gameplay=env, monte_carlo_tree_search=m, replay_buffer=buf, steps_per_iteration=T)
    games = buf.saved
    assert len(games) >= 2 * B and all(1 <= g.game_length <= limit for g in games)
    assert all(g.done == (g.game_length < limit) for g in games)
    assert reward[-1] == sum(sum(g.rewards) for g in games) / len(games) and np.isnan(loss[0])
    ep = env.episode.cpu().numpy()
    assert ep.min() >= 2 and len(games) == int(ep.sum())
    # the reset states are a function of (seed, global env index, episode): host evaluation == device
    e = 7
    st = env.reset_state_of(100 + e, int(ep[e]))
    assert (np.abs(st) <= 0.05).all() and not np.array_equal(st, env.reset_state_of(100 + e, int(ep[e]) + 1))
    assert 0 <= int(env.step_count[e].item()) < limit


def test_host_resident_envs_through_the_pinned_memory_adapter():
    """envs.HostVecEnv over B host CartPoles == the device env on the same initial states and seeds: same actions, same
    observations, same flags (the physics is the same float64 arithmetic), illegal moves and restarts included."""
    envs_mod, sp = _pkg("envs"), _pkg("selfplay")
    B, T, limit = 48, 24, 12
    heads = _model().heads("cuda:0")
    dev_env = envs_mod.CartPoleVec(B, "cuda:0", seed=5, on_end="mask", limit=limit)
    dev_env.reset()
    m = _batched(B); m.seed(np.arange(B, dtype=np.uint64))
    dchunk = sp.play_games(dev_env, heads, m, 1.0, T)
    torch.cuda.synchronize()

    class FromState(envs_mod.HostCartPole):           # host env started from the device env's initial state
        def __init__(self, st):
            super().__init__(); self._st = st
        def reset(self, seed=None):
            self.state = self._st.copy()
            return self.state.astype(np.float32), {}
    init = np.random.RandomState(5).uniform(-0.05, 0.05, size=(B, 4))
    host = envs_mod.HostVecEnv([FromState(init[i]) for i in range(B)], 4, 2, "cuda:0", limit=limit, on_end="mask")
    host.reset()
    m2 = _batched(B); m2.seed(np.arange(B, dtype=np.uint64))
    hchunk = sp.play_games(host, heads, m2, 1.0, T)
    torch.cuda.synchronize()
    a, b = dchunk.data.cpu().numpy(), hchunk.data.cpu().numpy()
    live = a[..., 5] != 3
    assert np.array_equal(a[..., 5], b[..., 5])
    assert np.array_equal(a[live][:, 4:], b[live][:, 4:])                       # reward, flag, policy, action, value, visits
    np.testing.assert_allclose(a[live][:, :4], b[live][:, :4], rtol=1e-6, atol=1e-7)   # libm cos/sin: host vs device

    # illegal moves + restart on the host: the penalty rule and record_obs (post-step) vs obs (next search's input)
    class Picky(envs_mod.HostCartPole):
        def step(self, action):
            if action == 1:
                raise ValueError("illegal")
            return super().step(action)
    host = envs_mod.HostVecEnv([Picky() for _ in range(8)], 4, 2, "cuda:0", limit=5, on_end="reset", env_seed=11)
    host.reset()
    first = host.obs.cpu().numpy().copy()
    act = torch.tensor([1, 0] * 4, dtype=torch.int32, device="cuda:0")
    for step in range(5):
        obs, rew, flag = host.step(act)
        torch.cuda.synchronize()
    r, f = rew.cpu().numpy(), flag.cpu().numpy()
    assert (r[0::2] == -5).all() and (r[1::2] == 1).all() and (f == 2).all()     # min(-4, -5, -1) at the limit step
    assert np.array_equal(host.record_obs.cpu().numpy()[0::2], first[0::2])      # illegal movers never moved
    assert not np.array_equal(host.obs.cpu().numpy(), host.record_obs.cpu().numpy()) and (host.episode == 1).all()


def test_batched_stored_game_replay_matches_the_single_game_loop():
    """reanalyse_replay_games: all steps of all stored games as one batch.  With the process-global stream replaced by
    per-position streams the numbers differ from the sequential loop, so the check is structural + against a direct
    search of the same positions with the same seeds."""
    sp = _pkg("selfplay")
    cfg, data = gu.load("reanalyse421_sims10_T1")
    GameRecord = _pkg("game").GameRecord
    stored = []
    for k in range(3):
        g = GameRecord(0.999, 2, 0.5, 50)
        n = len(data["src_observations"]) - 2 * k
        g.observations = [torch.from_numpy(o[None].copy()) for o in data["src_observations"][:n]]
        g.rewards = [float(r) + k for r in data["src_rewards"][:n]]
        stored.append(g)
    n_pos = sum(len(g.observations) - 2 for g in stored)
    m = _batched(n_pos, sims=10); m.seed(np.arange(n_pos, dtype=np.uint64))
    model = _model()
    out = sp.reanalyse_replay_games(stored, model, m, "cuda:0", temperature=1.0, train=True)
    assert [g.game_length for g in out] == [len(g.observations) - 2 for g in stored] and all(g.reanalyzed and g.done for g in out)
    d = _batched(n_pos, sims=10); d.seed(np.arange(n_pos, dtype=np.uint64))
    obs = torch.stack([g.observations[i].reshape(-1) for g in stored for i in range(len(g.observations) - 2)]).cuda()
    e = d.run(obs, model.heads("cuda:0"), train=True, act_temperature=1.0)
    action, policy, cv, rv = (t.cpu().numpy() for t in e.act(1.0))
    torch.cuda.synchronize()
    k = 0
    for src, g in zip(stored, out):
        for i in range(g.game_length):
            assert int(np.argmax(g.action_history[i])) == action[k] and np.array_equal(g.policies[i], policy[k])
            assert g.root_values[i] == np.float32(rv[k]) and np.array_equal(g.child_visits[i], cv[k])
            assert g.rewards[i] == src.rewards[action[k] + 1]                  # indexed by the action (game.py:255)
            assert torch.equal(g.observations[i], src.observations[i + 1])
            k += 1


def test_the_search_sees_weights_updated_in_place():
    """Muzero.heads() re-packs after an in-place update of a module (what an optimizer step or load_state_dict does):
    the next search uses the new weights.  (the cache used to serve the stale pack.)"""
    model = _model()
    B = 64
    obs = torch.randn(B, 4, generator=torch.Generator().manual_seed(0)).mul(0.05).cuda()
    def root_values():
        m = _batched(B); m.seed(np.arange(B, dtype=np.uint64))
        e = m.run(obs, model.heads("cuda:0"), train=False)
        rv = e.root_stats()[2]
        torch.cuda.synchronize()
        return rv.cpu().numpy().copy()
    before = root_values()
    assert np.array_equal(before, root_values())
    with torch.no_grad():
        model.prediction_function.value[-1].bias.add_(torch.linspace(-3, 3, 31))
        model.afterstate_prediction_function.value[-1].bias.add_(torch.linspace(-3, 3, 31))
    assert not np.array_equal(before, root_values())
