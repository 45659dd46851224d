"""Batched self-play: the loop of self_play.py:63-98 for B environments at once, and the self-play half of
learning_cycle (self_play.py:236-271).

Per env step (all on the GPU, asynchronously on one stream):
    observation -> BatchedMCTS.run (root inference, num_simulations x [select -> heads -> expand/backup])
                -> smz_act (Game.policy_step's policy/action + store_search_statistics, game.py:179-235)
                -> env.step -> smz_traj_pack (the appends of game.py:193-195, 263-267)
The result is a fixed-length structure-of-arrays trajectory chunk [T][B][F] (float64) that is gathered to the learner
rank (gather.py) and converted into Game-compatible records for ReplayBuffer.save_game (replay_buffer.py:109-137).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def temperature_scheduler(epoch=1, actual_epoch=1, mode="static_temperature"):
    """epoch -> action-sampling temperature, the five modes of self_play.py:124-163."""
    if isinstance(mode, (float, int)):
        return mode
    if mode == "reversal_tanh_temperature":
        xs = np.arange(1, epoch + 1, dtype=np.float64)
        lo, hi = xs.min(), xs.max()
        scaled = np.full_like(xs, lo if 0.001 <= lo <= 0.75 else (0.001 if lo < 0.001 else 0.75)) if hi == lo else \
            (xs - lo) * ((0.75 - 0.001) / (hi - lo)) + 0.001
        return (1 - np.tanh(scaled)[xs == actual_epoch]) * 1.1
    if mode == "extreme_temperature":
        for k, t in zip((100, 200, 300, 400, 500, 600), (3, 2, 1, .7, .5, .4)):
            if actual_epoch < epoch * (k / 700):
                return t
        if actual_epoch < epoch * 1:
            return .0625
        return None
    if mode == "linear_decrease_temperature":
        if epoch * 0.5 > actual_epoch:
            return 1
        if epoch * 0.75 > actual_epoch:
            return 0.5
        return 0.2
    if mode == "static_temperature":
        return 0.0
    if mode == "static_one_temperature":
        return 1
    return None


class TrajectoryChunk:
    """[T][B][F] float64 device buffer + the field offsets of smz_traj_pack's record."""

    def __init__(self, T, B, obs_dim, A, device):
        self.T, self.B, self.obs_dim, self.A = int(T), int(B), int(obs_dim), int(A)
        self.F = _lib.load().smz_traj_floats(self.obs_dim, self.A)
        self.data = torch.zeros(self.T, self.B, self.F, dtype=torch.float64, device=device)

    def fields(self, data=None):
        d = self.data if data is None else data
        o, A = self.obs_dim, self.A
        return dict(observation=d[..., :o], reward=d[..., o], terminated=d[..., o + 1], policy=d[..., o + 2:o + 2 + A],
                    action_onehot=d[..., o + 2 + A:o + 2 + 2 * A], root_value=d[..., o + 2 + 2 * A],
                    child_visits=d[..., o + 3 + 2 * A:o + 3 + 3 * A])


def chunk_targets(chunk_data, obs_dim, A, discount, td_steps, ignore_termination=False):
    """Vectorised replay ingest on the device (smz_traj_targets): for a [T][B][F] chunk returns
    (length [B] i32, value_target [T][B] f64, abs_td_error [T][B] f64) -- per stored position the n-step return that
    GameRecord.make_target / make_priority (game.py:291-337) compute one position at a time, bit for bit, and
    |root value - return| (the priority before `** priority_scale`)."""
    lib = _lib.load()
    T, B, F = chunk_data.shape
    assert chunk_data.is_cuda and chunk_data.dtype == torch.float64 and chunk_data.is_contiguous()
    assert F == lib.smz_traj_floats(int(obs_dim), int(A))
    dev = chunk_data.device
    pows = torch.tensor([discount ** i for i in range(int(td_steps) + 1)], dtype=torch.float64).to(dev)   # Python's pow
    length = torch.empty(B, dtype=torch.int32, device=dev)
    target = torch.empty(T, B, dtype=torch.float64, device=dev)
    err = torch.empty(T, B, dtype=torch.float64, device=dev)
    P = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(lib.smz_traj_targets(P(chunk_data), T, int(obs_dim), int(A), B, int(td_steps), P(pows),
                                    int(bool(ignore_termination)), P(length), P(target), P(err),
                                    C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return length, target, err


class GameRecord:
    """What self-play hands to ReplayBuffer.save_game: the trajectory lists of game.py:72-77 plus the few members
    the buffer reads (game_length, reanalyzed, make_priority: replay_buffer.py:109-137, game.py:174-177, 316-337)."""

    def __init__(self, discount, action_space_size, priority_scale=1, limit_of_game_play=float("inf")):
        self.discount, self.action_space_size = discount, action_space_size
        self.priority_scale, self.limit_of_game_play = priority_scale, limit_of_game_play
        self.action_history, self.rewards, self.policies = [], [], []
        self.root_values, self.child_visits, self.observations = [], [], []
        self.done, self.reanalyzed, self.env = False, False, None

    @property
    def terminal(self):
        return self.done

    @property
    def game_length(self):
        return len(self.action_history)

    def make_target(self, state_index, num_unroll, td_steps):
        """[value target, last reward, child_visits] for num_unroll consecutive positions (game.py:291-314):
        n-step return bootstrapped from the search value td_steps ahead; positions past the end are absorbing."""
        n = len(self.root_values)
        targets = []
        for cur in range(state_index, state_index + num_unroll):
            b = cur + td_steps
            value = self.root_values[b] * self.discount ** td_steps if b < n else 0.0
            for i, reward in enumerate(self.rewards[cur:b]):
                value += reward * self.discount ** i
            last_reward = self.rewards[cur - 1] if 0 < cur <= len(self.rewards) else 0.0
            if cur < n:
                targets.append([value, last_reward, self.child_visits[cur]])
            else:
                targets.append([0.0, last_reward, np.zeros(self.action_space_size, dtype=np.float64)])
        return targets

    def make_priority(self, td_steps):
        """|root value - n-step return| ** priority_scale per position, and its maximum (game.py:316-337)."""
        n = len(self.root_values)
        rv = np.array(self.root_values)
        target = []
        for i in range(n):
            b = i + td_steps
            value = self.root_values[b] * self.discount ** td_steps if b < n else 0
            for k, r in enumerate(self.rewards[i:b]):
                value += r * self.discount ** k
            target.append(value)
        pos = np.abs(rv - np.array(target)) ** self.priority_scale
        return pos, np.max(pos)


def chunk_to_games(chunk_data, obs_dim, A, discount, priority_scale=1, limit_of_game_play=float("inf"),
                   ignore_termination=False):
    """[T][B][F] (host or device) -> list of B GameRecord, each cut after its first terminated step."""
    d = chunk_data.detach().cpu().numpy() if torch.is_tensor(chunk_data) else np.asarray(chunk_data)
    T, B, F = d.shape
    o = obs_dim
    games = []
    for e in range(B):
        g = GameRecord(discount, A, priority_scale, limit_of_game_play)
        for t in range(T):
            r = d[t, e]
            g.observations.append(torch.from_numpy(r[:o].astype(np.float32))[None, ...])   # game.py:145-167 shape [1,obs]
            g.rewards.append(float(r[o]))
            g.policies.append(r[o + 2:o + 2 + A].copy())
            g.action_history.append(r[o + 2 + A:o + 2 + 2 * A].copy())
            g.root_values.append(np.float32(r[o + 2 + 2 * A]))
            g.child_visits.append(r[o + 3 + 2 * A:o + 3 + 3 * A].copy())
            term = bool(r[o + 1]) and not ignore_termination
            g.done = term if limit_of_game_play != len(g.observations) else False             # game.py:270-271
            if term:
                break
        games.append(g)
    return games


def play_games(env, heads, mcts, temperature, steps, chunk=None, train=True):
    """Plays `steps` env steps of all env.B environments; returns the TrajectoryChunk (device resident).
    Everything is enqueued asynchronously on the current stream; the caller synchronises."""
    lib = _lib.load()
    dev = env.device
    A = env.num_actions
    if chunk is None:
        chunk = TrajectoryChunk(steps, env.B, env.obs_dim, A, dev)
    assert chunk.T >= steps and chunk.B == env.B
    obs = env.obs
    P = lambda t: C.c_void_p(t.data_ptr())
    for t in range(steps):
        eng = mcts.run(obs, heads, train=train, act_temperature=temperature)
        action, policy, child_visits, root_value = eng.act(temperature)
        if hasattr(env, "step_and_record"):          # built-in env: step + record in one launch
            obs, reward, terminated = env.step_and_record(action, chunk.data, t, policy, child_visits, root_value)
        else:
            obs, reward, terminated = env.step(action)
            _lib.check(lib.smz_traj_pack(P(chunk.data), chunk.T, t, env.obs_dim, A, P(obs), P(reward), P(terminated),
                                         P(action), P(policy), P(child_visits), P(root_value), env.B,
                                         C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return chunk


class StreamGroup:
    """One slice of a rank's environments with its own engine, head buffers, trajectory chunk and HIP stream.

    At 4096 envs every kernel of a simulation round is latency/launch bound and occupies a small part of the chip,
    so a rank splits its envs into G independent groups and runs each group's (captured) search on its own stream:
    the groups' kernels overlap on the GPU.  Trees are independent, so results do not depend on the grouping."""

    def __init__(self, env, heads, mcts, steps, stream=None):
        self.env, self.heads, self.mcts = env, heads, mcts
        self.stream = stream if stream is not None else torch.cuda.Stream(device=env.device)
        self.chunk = TrajectoryChunk(steps, env.B, env.obs_dim, env.num_actions, env.device)


def play_games_grouped(groups, temperature, steps, train=True):
    """play_games for several StreamGroups concurrently: step t of every group is enqueued before step t+1 of any,
    each on its group's stream; the caller's stream waits for all of them at the end."""
    dev = groups[0].env.device
    cur = torch.cuda.current_stream(dev)
    for g in groups:
        g.stream.wait_stream(cur)
    lib = _lib.load()
    P = lambda t: C.c_void_p(t.data_ptr())
    for t in range(steps):
        for g in groups:
            with torch.cuda.stream(g.stream):
                env = g.env
                eng = g.mcts.run(env.obs, g.heads, train=train, act_temperature=temperature)
                action, policy, child_visits, root_value = eng.act(temperature)
                if hasattr(env, "step_and_record"):      # built-in env: step + record in one launch
                    env.step_and_record(action, g.chunk.data, t, policy, child_visits, root_value)
                else:
                    obs, reward, terminated = env.step(action)
                    _lib.check(lib.smz_traj_pack(P(g.chunk.data), g.chunk.T, t, env.obs_dim, env.num_actions, P(obs),
                                                 P(reward), P(terminated), P(action), P(policy), P(child_visits),
                                                 P(root_value), env.B, C.c_void_p(g.stream.cuda_stream)))
    for g in groups:
        cur.wait_stream(g.stream)
    return [g.chunk for g in groups]


def reanalyse_games(games, model, mcts, device, train=False):
    """MuZero-Reanalyse on stored games: every stored position is searched again with the CURRENT networks by the
    same batched engine (pure search, no environment) and its root value / child-visit target is refreshed.
    The first position of a game is not in the record (game.py stores post-step observations, :264), so position t
    re-searches observations[t-1] for t >= 1 and position 0 keeps its stored statistics.
    `mcts.num_trees` must be >= the number of positions searched per call (the batch is padded)."""
    heads = model.heads(device)
    obs, index = [], []
    for gi, g in enumerate(games):
        for t in range(1, g.game_length):
            obs.append(g.observations[t - 1].reshape(-1))
            index.append((gi, t))
    if not obs:
        return 0
    B = mcts.num_trees
    done = 0
    for lo in range(0, len(obs), B):
        rows = obs[lo:lo + B]
        batch = torch.stack(rows + [rows[-1]] * (B - len(rows))).to(device=device, dtype=torch.float32).contiguous()
        eng = mcts.run(batch, heads, train=train)
        _, _, child_visits, root_value = eng.act(0.0)
        torch.cuda.synchronize(device)
        cv, rv = child_visits.cpu().numpy(), root_value.cpu().numpy()
        for k, (gi, t) in enumerate(index[lo:lo + B]):
            games[gi].child_visits[t] = cv[k].copy()
            games[gi].root_values[t] = np.float32(rv[k])
            games[gi].reanalyzed = True
        done += len(rows)
    return done


def self_play_iteration(env, model, mcts, temperature, steps, replay_buffer=None, gather=None, priority_scale=1,
                        ignore_termination=False):
    """Self-play half of one learning_cycle iteration (self_play.py:245-271): play, gather to the learner rank,
    hand the games to replay_buffer.save_game, return (games, mean reward) on the learner rank."""
    heads = model.heads(env.device)
    env.reset()
    chunk = play_games(env, heads, mcts, temperature, steps)
    data = chunk.data
    if gather is not None:
        parts = gather(data)
        if parts is None:
            return None, None
        data = torch.cat([p for p in parts], dim=1)
    torch.cuda.synchronize(env.device)
    games = chunk_to_games(data, env.obs_dim, env.num_actions, mcts.discount, priority_scale,
                           limit_of_game_play=steps, ignore_termination=ignore_termination)
    rewards = []
    for g in games:
        if replay_buffer is not None:
            replay_buffer.save_game(g)
        rewards.append(sum(g.rewards))
    return games, sum(rewards) / len(rewards)
