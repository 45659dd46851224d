"""Array-backed game records (game.ArrayGameRecord, selfplay.records_from_host_copy) against the per-step-list records of
selfplay.chunk_to_games -- the checker, itself pinned to the reference's own games (tests/test_abi_and_host.py) -- field by
field, for every way a chunk is cut into games; and through a buffer that stores and samples them with the access pattern of
the reference's ReplayBuffer (replay_buffer.py:109-137 save_game, :155-214 sample_position / sample_batch).  CPU tests: the
device half (transposes, smz_traj_targets_games) is tests/test_gpu_records.py."""
import copy
import pickle
from importlib import import_module

import numpy as np
import pytest
import torch

import stochastic_muzero_amd  # noqa: F401


def _sp():
    return import_module("stochastic-muzero_amd.selfplay")


def game_ends(flags, new_game):
    """numpy restatement of k_traj_game_ends (csrc/smz_kernels.hip): flags [T][B] -> game_end [T][B]."""
    T, B = flags.shape
    out = np.empty((T, B), np.int32)
    for e in range(B):
        end = first = T
        for t in range(T - 1, -1, -1):
            f = int(flags[t, e])
            if f == 3:
                end = t; out[t, e] = -1
                continue
            if f != 0:
                end = t + 1
            out[t, e] = end
            first = end
        if not new_game:
            out[first:, e] = -1
    return out


def make_chunk(T, B, o, A, seed, p_end=0.15, mask_tail=False, odd_env=None):
    r = np.random.RandomState(seed)
    F = o + 3 * A + 3
    d = np.zeros((T, B, F))
    d[..., :o] = r.randn(T, B, o).astype(np.float32)
    d[..., o] = r.randint(-2, 3, (T, B)) + r.choice([0.0, 0.25], (T, B))
    flags = (r.rand(T, B) < p_end) * r.choice([1, 2], (T, B))
    if mask_tail:                                   # on_end="mask": after its end an env takes no step
        for e in range(B):
            ends = np.nonzero(flags[:, e])[0]
            if len(ends):
                flags[ends[0] + 1:, e] = 3
    if odd_env is not None:                         # switched off and on again inside the chunk
        flags[2:4, odd_env] = 3
        flags[4:6, odd_env] = 0
    d[..., o + 1] = flags
    pol = r.rand(T, B, A); d[..., o + 2:o + 2 + A] = pol / pol.sum(-1, keepdims=True)
    act = r.randint(0, A, (T, B)); d[..., o + 2 + A:o + 2 + 2 * A] = np.eye(A)[act]
    d[..., o + 2 + 2 * A] = r.randn(T, B).astype(np.float32) * 10
    cv = r.rand(T, B, A); d[..., o + 3 + 2 * A:] = cv / cv.sum(-1, keepdims=True)
    return d


def host_targets(games_by_env, B, T, td):
    """target / abs_td [B][T] from the checker's own make_target / make_priority (what the device computes bit for bit:
    tests/test_gpu_loop.py, test_gpu_records.py)."""
    target, err = np.zeros((B, T)), np.zeros((B, T))
    for (e, t0), g in games_by_env.items():
        n = g.game_length
        target[e, t0:t0 + n] = [g.make_target(i, 1, td)[0][0] for i in range(n)]
        err[e, t0:t0 + n] = g.make_priority(td)[0] ** (1.0 / g.priority_scale) if g.priority_scale != 1 else g.make_priority(td)[0]
    return target, err


def build(d, o, A, td=5, priority_scale=1, **kw):
    sp = _sp()
    T, B, F = d.shape
    ignore = kw.get("ignore_termination", False)
    flags = np.zeros((T, B), np.int64) if ignore else d[..., o + 1].astype(np.int64)
    ge = game_ends(flags, kw.get("after_end", "drop") == "new_game")
    want = sp.chunk_to_games(d, o, A, 0.97, priority_scale, **kw)
    # the checker's games of EVERY window (partial ones included) give the host-side stand-in for the device's target arrays
    rec = np.ascontiguousarray(d.transpose(1, 0, 2))
    by_env = {}
    for e in range(B):
        if ((flags[:-1, e] == 3) & (flags[1:, e] != 3)).any():
            continue                                # not windows: records_from_host_copy hands this env to the general loop
        every = sp.chunk_to_games(d[:, e:e + 1], o, A, 0.97, 1, **dict(kw, keep_partial=True))
        t = 0
        for g in every:
            while ge[t, e] < 0:
                t += 1
            by_env[(e, t)] = g
            t = ge[t, e]
    target, err = host_targets(by_env, B, T, td)
    got = sp.records_from_host_copy(rec, np.ascontiguousarray(ge.T), o, A, 0.97, priority_scale, td_steps=td, target=target,
                                    abs_td=err, **kw)
    return want, got


def same_game(a, b, td=5):
    assert a.game_length == b.game_length and a.done == b.done and a.reanalyzed == b.reanalyzed
    assert (a.discount, a.action_space_size, a.priority_scale, a.limit_of_game_play) == \
        (b.discount, b.action_space_size, b.priority_scale, b.limit_of_game_play)
    for name in ("observations", "rewards", "policies", "action_history", "root_values", "child_visits"):
        x, y = getattr(a, name), getattr(b, name)
        assert len(x) == len(y), name
        for u, v in zip(x, y):
            assert type(u) is type(v), (name, type(u), type(v))
            if torch.is_tensor(u):
                assert u.dtype == v.dtype and u.shape == v.shape and torch.equal(u, v), name
            else:
                assert np.array_equal(np.asarray(u), np.asarray(v)) and np.asarray(u).dtype == np.asarray(v).dtype, name
    pa, ta = a.make_priority(td); pb, tb = b.make_priority(td)
    assert np.array_equal(pa, pb) and ta == tb
    n = a.game_length
    for i in (0, n // 2, max(0, n - 2)):
        for x, y in zip(a.make_target(i, 4, td), b.make_target(i, 4, td)):
            assert x[0] == y[0] and type(x[0]) is type(y[0]), (x[0], y[0])
            assert x[1] == y[1] and np.array_equal(x[2], y[2])
    assert a.make_target(1, 3, td + 2)[0][0] == b.make_target(1, 3, td + 2)[0][0]     # another td_steps: the lists' own loop


@pytest.mark.parametrize("kw", [dict(), dict(keep_partial=False), dict(after_end="new_game"),
                                dict(after_end="new_game", keep_partial=False), dict(ignore_termination=True),
                                dict(limit_of_game_play=3, after_end="new_game")])
@pytest.mark.parametrize("mask_tail", [False, True])
def test_array_records_equal_the_list_records_field_by_field(kw, mask_tail):
    d = make_chunk(12, 9, 4, 3, seed=len(kw) + 7 * mask_tail, mask_tail=mask_tail)
    want, got = build(d, 4, 3, **kw)
    assert len(want) == len(got) and len(got) > 0
    for a, b in zip(want, got):
        same_game(a, b)


def test_env_switched_off_and_on_inside_a_chunk_falls_back_to_the_general_loop():
    sp = _sp()
    d = make_chunk(12, 6, 4, 2, seed=3, p_end=0.1, odd_env=2)
    assert (d[2:4, 2, 5] == 3).all()
    for kw in (dict(), dict(after_end="new_game")):
        want, got = build(d, 4, 2, **kw)
        assert len(want) == len(got)
        for a, b in zip(want, got):
            same_game(a, b)
        assert any(not isinstance(g, sp.ArrayGameRecord) for g in got) and any(isinstance(g, sp.ArrayGameRecord) for g in got)


def test_priority_scale_and_mutation_fall_back_to_the_lists():
    d = make_chunk(10, 4, 4, 2, seed=11, p_end=0.0)
    want, got = build(d, 4, 2, priority_scale=0.5)
    for a, b in zip(want, got):
        pa, ta = a.make_priority(5); pb, tb = b.make_priority(5)
        np.testing.assert_allclose(pa, pb, rtol=1e-15)        # (x ** 0.5 of the same x)
        assert ta == pb.max()
    g, w = got[0], want[0]
    # the lists behave like lists: slices are lists, appends and item assignment work and switch the record to its own loops
    tail = g.action_history[3:]
    assert isinstance(tail, list) and len(tail) == 7
    tail += [np.zeros(2)] * 2                                # replay_buffer.py:171-180 fill_gap_empty_action
    g.rewards[2] = 100.0; w.rewards[2] = 100.0
    g.child_visits.append(np.array([0.5, 0.5])); w.child_visits.append(np.array([0.5, 0.5]))
    assert g.rewards[2] == 100.0 and len(g.child_visits) == 11
    assert np.array_equal(g.make_priority(5)[0], w.make_priority(5)[0])
    assert g.make_target(0, 3, 5)[0][0] == w.make_target(0, 3, 5)[0][0]
    assert g.game_length == 10
    g.action_history.append(np.zeros(2))
    assert g.game_length == 11


class Buffer:
    """The access pattern of the reference's ReplayBuffer on stored games (replay_buffer.py:109-137, 155-214), restated:
    save_game -> make_priority(td) + game_length + reanalyzed; sample -> make_extended_image, action_history[pos:] padded with
    zero actions, make_target."""

    def __init__(self, td, unroll, window=1000):
        self.td_steps, self.num_unroll, self.window = td, unroll, window
        self.buffer, self.prio_position, self.prio_game, self.total = [], [], [], 0

    def save_game(self, game):
        if len(self.buffer) > self.window:
            self.total -= self.buffer.pop(0).game_length
            self.prio_position.pop(0); self.prio_game.pop(0)
        pos, top = game.make_priority(self.td_steps)
        self.prio_position.append(pos); self.prio_game.append(top)
        self.buffer.append(game)
        self.total += game.game_length
        assert game.reanalyzed is False

    def sample(self, gi, pos):
        g = self.buffer[gi]
        actions = g.action_history[pos:][:self.num_unroll]
        if self.num_unroll - len(actions) > 0:
            actions += [np.zeros(actions[0].shape)] * (self.num_unroll - len(actions))
        return g.make_extended_image(pos, self.num_unroll), actions, g.make_target(pos, self.num_unroll, self.td_steps)


def test_a_buffer_with_the_references_access_pattern_sees_the_same_games():
    d = make_chunk(16, 12, 4, 2, seed=5, p_end=0.08)
    want, got = build(d, 4, 2, td=4, after_end="new_game")
    bw, bg = Buffer(4, 5), Buffer(4, 5)
    for a, b in zip(want, got):
        bw.save_game(a); bg.save_game(b)
    assert bw.total == bg.total and bw.prio_game == bg.prio_game
    r = np.random.RandomState(0)
    for _ in range(60):
        gi = r.randint(len(bw.buffer)); n = bw.buffer[gi].game_length
        pos = r.randint(0, n)
        (ia, aa, ta), (ib, ab, tb) = bw.sample(gi, pos), bg.sample(gi, pos)
        assert len(ia) == len(ib) and all(torch.equal(x, y) for x, y in zip(ia, ib))
        assert len(aa) == len(ab) and all(np.array_equal(x, y) for x, y in zip(aa, ab))
        for x, y in zip(ta, tb):
            assert x[0] == y[0] and x[1] == y[1] and np.array_equal(x[2], y[2])
    # update_value (replay_buffer.py:216-222) writes new priorities into a game's position array in place: another game's stay
    before = bg.prio_position[1].copy()
    bg.prio_position[0][:] = 7.0
    assert np.array_equal(bg.prio_position[1], before)
    # save_buffer pickles the games (replay_buffer.py:100-106); a deep copy is what play_game makes of an environment
    back = pickle.loads(pickle.dumps(bg.buffer[:3]))
    for a, b in zip(back, want[:3]):
        assert a.game_length == b.game_length and list(a.rewards) == list(b.rewards)
        assert np.array_equal(np.array(a.policies), np.array(b.policies))
    c = copy.deepcopy(got[0])
    assert c.game_length == got[0].game_length and c.rewards == got[0].rewards


def test_image_observations_outside_the_record():
    sp = _sp()
    T, B, A, shape = 6, 3, 2, (3, 4, 4)
    d = make_chunk(T, B, 0, A, seed=9, p_end=0.2)
    frames = torch.rand(T, B, 48)
    ge = game_ends(d[..., 1].astype(np.int64), True)
    want = sp.chunk_to_games(d, 0, A, 0.97, after_end="new_game", observations=frames, observation_shape=shape)
    got = sp.records_from_host_copy(np.ascontiguousarray(d.transpose(1, 0, 2)), np.ascontiguousarray(ge.T), 0, A, 0.97,
                                    after_end="new_game", observations=frames.permute(1, 0, 2).contiguous(),
                                    observation_shape=shape)
    assert len(want) == len(got) > 0
    for a, b in zip(want, got):
        assert len(a.observations) == len(b.observations)
        for u, v in zip(a.observations, b.observations):
            assert u.shape == v.shape == (1, 3, 4, 4) and u.dtype == v.dtype and torch.equal(u, v)
        assert [x.shape for x in b.make_extended_image(0, 9)] == [(1, 3, 4, 4)] * 9


def test_stale_observation_width_fails_loudly():
    """a record whose observations moved to chunk.obs (obs_dim > TrajectoryChunk.SPLIT_OBS) sliced with the env's
    obs_dim must not mis-slice silently."""
    sp = _sp()
    d = make_chunk(4, 2, 0, 2, seed=1)              # F = 3 A + 3: observations are elsewhere
    with pytest.raises(AssertionError, match="chunk.obs"):
        sp.chunk_to_games(d, 65, 2, 0.97)
    with pytest.raises(AssertionError):
        sp.chunk_to_games(d[..., :8], 65, 2, 0.97)


@pytest.mark.parametrize("seed", range(24))
def test_random_chunks_cut_the_same_way_as_the_checker(seed):
    """Randomised: chunk shape, end-flag density, masked tails, envs switched off from the start, off-and-on envs, every cutting
    rule -- the array records are the checker's games, in its order."""
    r = np.random.RandomState(1000 + seed)
    T, B, A = int(r.randint(1, 20)), int(r.randint(1, 14)), int(r.randint(2, 5))
    o = int(r.choice([0, 1, 4, 9]))
    d = make_chunk(T, B, o, A, seed=seed, p_end=float(r.choice([0.0, 0.05, 0.3, 0.9])), mask_tail=bool(r.randint(2)),
                   odd_env=int(r.randint(B)) if (T >= 6 and r.rand() < 0.4) else None)
    if r.rand() < 0.3:
        d[:, r.randint(B), o + 1] = 3                       # an env that never played
    kw = dict(after_end=str(r.choice(["drop", "new_game"])), keep_partial=bool(r.randint(2)),
              ignore_termination=bool(r.rand() < 0.2))
    if r.rand() < 0.5:
        kw["limit_of_game_play"] = int(r.randint(1, 6))
    want, got = build(d, o, A, td=int(r.randint(1, 7)), **kw) if o else (None, None)
    if o == 0:                                              # observations outside the record
        sp = _sp()
        frames = torch.from_numpy(r.rand(T, B, 6).astype(np.float32))
        flags = np.zeros((T, B), np.int64) if kw["ignore_termination"] else d[..., 1].astype(np.int64)
        ge = game_ends(flags, kw["after_end"] == "new_game")
        want = sp.chunk_to_games(d, 0, A, 0.97, observations=frames, observation_shape=(2, 3), **kw)
        got = sp.records_from_host_copy(np.ascontiguousarray(d.transpose(1, 0, 2)), np.ascontiguousarray(ge.T), 0, A, 0.97,
                                        observations=frames.permute(1, 0, 2).contiguous(), observation_shape=(2, 3), **kw)
    assert len(want) == len(got)
    for a, b in zip(want, got):
        same_game(a, b, 3)


def test_reward_sums_are_pythons_sums_game_by_game_even_with_non_finite_rewards():
    """the per-game reward sums learning_cycle averages are sum(game.rewards) bit for bit, and a -inf reward (the
    reference's illegal-move reward with an unlimited game length, game.py:123-131) stays inside its own game."""
    sp = _sp()
    d = make_chunk(14, 7, 4, 3, seed=3, p_end=0.25)
    d[..., 4] += np.random.RandomState(0).rand(14, 7) * 1e-3             # rewards whose sum depends on the order of addition
    d[2, 3, 4] = -np.inf                                                # one illegal move in env 3's first game
    d[5, 1, 4] = np.inf
    want, got = build(d, 4, 3, after_end="new_game")
    assert got[0]._src.fresh is not None and got[0]._src.fresh[0] is got
    sums = sp._reward_sums(got)                                          # the list as built: summed from the windows' arrays
    assert got[0]._src.fresh is None and sp._reward_sums(got) == sums    # ... once; later calls check record by record
    assert len(sums) == len(want) > 7
    for g, s in zip(want, sums):
        ref = sum(g.rewards)
        assert (s == ref) or (np.isnan(s) and np.isnan(ref)), (s, ref)
    assert sum(np.isinf(s) for s in sums) == 2 and not any(np.isnan(s) for s in sums)
    # records whose list became a real list take the slow path and agree
    got[0].rewards.append(1.5)
    assert sp._reward_sums(got)[0] == sum(got[0].rewards)


def test_whole_chunk_reward_sums_accumulate_in_float64_like_pythons_sum():
    """the fast path for one whole-chunk game per env scanned the float32 reward column in float32; sum(g.rewards)
    adds Python floats.  Non-integer rewards make the two differ in the 7th digit -- and learning_cycle's save-model decision
    compares reward means for equality."""
    sp = _sp()
    T, B = 40, 6
    d = make_chunk(T, B, 4, 3, seed=5, p_end=0.0)
    d[..., 4] = np.random.RandomState(1).rand(T, B).astype(np.float32)    # float32-representable, non-integer
    want, got = build(d, 4, 3, after_end="drop", keep_partial=True)
    assert len(got) == B and all(len(g.rewards) == T for g in want)      # the whole-chunk shape: the fast path's condition
    fast = sp._reward_sums(got)
    slow = [sum(g.rewards) for g in want]
    assert fast == slow
    f32 = np.cumsum(d[..., 4].astype(np.float32).T, axis=1)[:, -1].astype(np.float64).tolist()
    assert f32 != slow                                                    # (the case does tell the two accumulations apart)


def test_make_priority_returns_an_array_the_caller_may_overwrite():
    """ReplayBuffer.update_value writes prio_position[game][h] in place (replay_buffer.py:222); the array a record
    hands out must not be a view of the chunk-wide priorities."""
    d = make_chunk(12, 5, 4, 3, seed=11)
    want, got = build(d, 4, 3, after_end="new_game")
    g = got[0]
    first, top = g.make_priority(5)
    keep = first.copy()
    first[:] = 123.0                                                    # what update_value does
    again, top2 = g.make_priority(5)
    assert np.array_equal(again, keep) and top2 == top
    assert np.array_equal(want[0].make_priority(5)[0], keep)


def test_every_open_records_job_owns_its_staging_buffers():
    """three jobs open at once (a pipelined iteration + an evaluation chunk of the same shape) must not share
    page-locked buffers; released sets are reused, and at most `keep` idle sets stay allocated."""
    sp = _sp()
    made = []
    pool = sp._StagingPool(keep=2, alloc=lambda shape, dtype: made.append(shape) or torch.empty(shape, dtype=dtype))
    lay = (((4, 3), torch.float64), ((4,), torch.int32))
    a, b, c = pool.acquire(lay), pool.acquire(lay), pool.acquire(lay)
    ptrs = {t.data_ptr() for s in (a, b, c) for t in s}
    assert len(ptrs) == 6 and len(made) == 6
    pool.release(lay, a)
    assert pool.acquire(lay) is a and len(made) == 6                    # reused, nothing new allocated
    other = (((2, 2), torch.float32),)
    x = pool.acquire(other)
    for s, l in ((a, lay), (b, lay), (c, lay), (x, other)):
        pool.release(l, s)
    assert len(pool.free) == 2 and pool.free[-1][0] == other            # bounded: the oldest idle sets are dropped
