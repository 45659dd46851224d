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
from .game import ArrayGameRecord, ChunkHostCopy, Game, GameRecord  # noqa: F401  (re-exported: selfplay.GameRecord ...)


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
    """[T][B][F] float64 device buffer + the field offsets of smz_traj_pack's record.

    Image observations (obs_dim > SPLIT_OBS: a 98x98x3 frame is 28 812 floats) are NOT widened into the float64 record: they
    are kept as float32 -- what the heads consume and what the reference's Game stores (game.py:264) -- in `obs`
    [T][B][obs_dim], and the record carries the other fields only (`rec_obs_dim` = 0).  Vector observations stay inside the
    record (`obs` is None, `rec_obs_dim` = obs_dim)."""
    SPLIT_OBS = 64

    def __init__(self, T, B, obs_dim, A, device):
        self.T, self.B, self.obs_dim, self.A = int(T), int(B), int(obs_dim), int(A)
        self.rec_obs_dim = 0 if self.obs_dim > self.SPLIT_OBS else self.obs_dim
        self.F = _lib.load().smz_traj_floats(self.rec_obs_dim, self.A)
        self.data = torch.zeros(self.T, self.B, self.F, dtype=torch.float64, device=device)
        self.obs = None if self.rec_obs_dim else torch.zeros(self.T, self.B, self.obs_dim, dtype=torch.float32, device=device)
        self.owed_obs = None        # (slot, tensor): frames of step `slot` that the next representation launch will copy
        self.owed_patch = None      # (rows i32 [n], frames f32 [n, ...]): rows of that slot that show OTHER frames (envs reset in the step)

    def flush_obs(self):
        """Copies frames whose record was left to a representation launch that has not come (end of a play_games call, a
        search path that does not record)."""
        if self.owed_obs is not None:
            t, frames = self.owed_obs
            self.obs[t].copy_(frames.reshape(self.B, -1))
            self._patch(t)
            self.owed_obs = None

    def apply_patch(self):
        """After a representation launch has written the owed frames: the rows of envs that were reset inside that step."""
        if self.owed_patch is not None:
            self._patch(self._owed_slot)

    def _patch(self, t):
        if self.owed_patch is not None:
            rows, frames = self.owed_patch
            self.obs[t].index_copy_(0, rows.long(), frames.reshape(frames.shape[0], -1))
            self.owed_patch = None

    @property
    def owed_obs(self):
        return self._owed

    @owed_obs.setter
    def owed_obs(self, v):
        self._owed = v
        if v is not None:
            self._owed_slot = v[0]

    def fields(self, data=None):
        d = self.data if data is None else data
        o, A = self.rec_obs_dim, self.A
        # (a gathered `data` of other ranks has its frames in a gathered message of its own: no observation entry then)
        obs = d[..., :o] if self.obs is None else (self.obs if data is None else None)
        return dict(observation=obs, reward=d[..., o], terminated=d[..., o + 1],
                    policy=d[..., o + 2:o + 2 + A],
                    action_onehot=d[..., o + 2 + A:o + 2 + 2 * A], root_value=d[..., o + 2 + 2 * A],
                    child_visits=d[..., o + 3 + 2 * A:o + 3 + 3 * A])


_POWERS = {}


def _discount_powers(discount, td_steps, device):
    """discount ** i for i = 0 .. td_steps (Python's pow, as game.py:300-305 evaluates it) on the device, uploaded once per
    (discount, td_steps, device): an upload from pageable memory waits for the stream to drain, which would serialise a caller
    that has just enqueued a search (self_play_iterations)."""
    key = (discount, td_steps, str(device))
    t = _POWERS.get(key)
    if t is None:
        t = _POWERS[key] = torch.tensor([discount ** i for i in range(td_steps + 1)], dtype=torch.float64).to(device)
    return t


def chunk_targets(chunk_data, obs_dim, A, discount, td_steps, ignore_termination=False, after_end="drop",
                  return_game_end=False):
    """Vectorised replay ingest on the device (smz_traj_targets_games): for a [T][B][F] chunk returns
    (length [B] i32, value_target [T][B] f64, abs_td_error [T][B] f64) -- per stored position the n-step return that
    GameRecord.make_target / make_priority (game.py:291-337) compute one position at a time, bit for bit, and
    |root value - return| (the priority before `** priority_scale`).

    after_end as in chunk_to_games: "drop" -- an env's rows behind its first finished game belong to no game (targets 0);
    "new_game" -- they are its next games (on_end="reset" chunks), each with its own targets, the rows after the last end
    flag forming an unfinished game cut at the chunk's end (chunk_to_games(keep_partial=True)).  `length` is the end of
    each env's FIRST game either way; return_game_end adds game_end [T][B] i32 (one past the last row of the row's game,
    -1: no game) as a fourth result."""
    assert after_end in ("drop", "new_game")
    lib = _lib.load()
    if isinstance(chunk_data, TrajectoryChunk):
        chunk_data, obs_dim = chunk_data.data, chunk_data.rec_obs_dim
    T, B, F = chunk_data.shape
    assert chunk_data.is_cuda and chunk_data.dtype == torch.float64 and chunk_data.is_contiguous()
    assert F == lib.smz_traj_floats(int(obs_dim), int(A)), \
        f"a record of {F} floats is not obs_dim {obs_dim} + 3 * {A} + 3 (image observations: pass the TrajectoryChunk or obs_dim=0)"
    dev = chunk_data.device
    pows = _discount_powers(float(discount), int(td_steps), dev)
    length = torch.empty(B, dtype=torch.int32, device=dev)
    game_end = torch.empty(T, B, dtype=torch.int32, device=dev)
    target = torch.empty(T, B, dtype=torch.float64, device=dev)
    err = torch.empty(T, B, dtype=torch.float64, device=dev)
    P = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(lib.smz_traj_targets_games(P(chunk_data), T, int(obs_dim), int(A), B, int(td_steps), P(pows),
                                          int(bool(ignore_termination)), int(after_end == "new_game"), P(length),
                                          P(game_end), P(target), P(err),
                                          C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return (length, target, err, game_end) if return_game_end else (length, target, err)


def chunk_to_games(chunk_data, obs_dim, A, discount, priority_scale=1, limit_of_game_play=float("inf"),
                   ignore_termination=False, keep_partial=True, after_end="drop", observations=None, observation_shape=None):
    """[T][B][F] (host or device) -> list of GameRecord, env-major.  The record's flag slot cuts the games: 1 = terminated
    (Game.done True), 2 = stopped by limit_of_game_play (done False, game.py:270-271), 3 = no step (env switched off).
    after_end: what the rows behind an env's finished game are -- "drop": nothing (an env that is stepped on past its end,
    the fixed-length episodes of on_end="continue"), "new_game": its next game (on_end="reset": several games per env and
    chunk).  The rows behind the last finished game form an unfinished one, kept when keep_partial (a chunk without any
    end flag is then one game per env).  `observations` [T][B][n] float32 (TrajectoryChunk.obs): the observations live outside
    the record (obs_dim is then 0), each stored as [1, *observation_shape] like the reference's frames ([1,3,98,98])."""
    assert after_end in ("drop", "new_game")
    if isinstance(chunk_data, TrajectoryChunk):      # the chunk knows where its observations live
        chunk = chunk_data
        chunk_data, obs_dim, observations = chunk.data, chunk.rec_obs_dim, chunk.obs if observations is None else observations
    d = chunk_data.detach().cpu().numpy() if torch.is_tensor(chunk_data) else np.asarray(chunk_data)
    T, B, F = d.shape
    o = obs_dim
    assert F == o + 3 * A + 3, (f"a record of {F} floats is not obs_dim {o} + 3 * {A} + 3: observations wider than "
                                f"TrajectoryChunk.SPLIT_OBS live in chunk.obs (pass the TrajectoryChunk, or obs_dim=0 and observations=)")
    if observations is not None:
        assert o == 0
        observations = observations.detach().cpu() if torch.is_tensor(observations) else torch.as_tensor(np.asarray(observations))
        observations = observations.to(torch.float32)
    games = []
    for e in range(B):
        g = None
        for t in range(T):
            r = d[t, e]
            flag = 0 if ignore_termination else int(r[o + 1])
            if flag == 3:
                continue
            if g is None:
                g = GameRecord(discount, A, priority_scale, limit_of_game_play)
            if observations is None:
                g.observations.append(torch.from_numpy(r[:o].astype(np.float32))[None, ...])   # game.py:145-167 shape [1,obs]
            else:
                frame = observations[t, e].clone()
                g.observations.append(frame.reshape((1,) + tuple(observation_shape)) if observation_shape else frame[None, ...])
            g.rewards.append(float(r[o]))
            g.policies.append(r[o + 2:o + 2 + A].copy())
            g.action_history.append(r[o + 2 + A:o + 2 + 2 * A].copy())
            g.root_values.append(np.float32(r[o + 2 + 2 * A]))
            g.child_visits.append(r[o + 3 + 2 * A:o + 3 + 3 * A].copy())
            g.done = (flag == 1) if limit_of_game_play != len(g.observations) else False     # game.py:270-271
            if flag != 0:
                games.append(g)
                g = None
                if after_end == "drop":
                    break
        if g is not None and keep_partial:
            games.append(g)
    return games


def _segments(flags, game_end, keep_partial):
    """Game windows of an env-major chunk: flags [B][T] (the record's flag slot), game_end [B][T] (smz_traj_targets_games:
    one past the last row of the row's game, -1 = no game).  Returns (env, t0, t1, last_flag) arrays in chunk_to_games'
    order (env-major, games of an env in time order)."""
    B, T = game_end.shape
    start = game_end >= 0
    start[:, 1:] &= game_end[:, 1:] != game_end[:, :-1]
    e, t0 = np.nonzero(start)                                  # row-major: env-major, time order
    t1 = game_end[e, t0]
    last = flags[e, t1 - 1]
    if not keep_partial:
        keep = last != 0
        e, t0, t1, last = e[keep], t0[keep], t1[keep], last[keep]
    return e, t0, t1, last


def chunk_to_records(chunk_data, obs_dim, A, discount, priority_scale=1, limit_of_game_play=float("inf"),
                     ignore_termination=False, keep_partial=True, after_end="drop", observations=None, observation_shape=None,
                     td_steps=None):
    """chunk_to_games for a device chunk at the speed of the search that fills it: the same games, in the same order, as
    ArrayGameRecord windows into ONE env-major host copy of the chunk (game.py).  The device cuts the games
    (smz_traj_targets_games: game ends per row) and, when `td_steps` is given, computes every position's n-step value target and
    priority in the same launch pair, so that ReplayBuffer.save_game's make_priority (replay_buffer.py:109-137) and
    sample_batch's make_target (:185-214) are array reads.  Host work: two transfers, one numpy pass over the [B][T] flags, one
    small object per game.  The checker is chunk_to_games: field by field the same (tests/test_gpu_records.py).

    An env whose "no step" rows (flag 3) are followed by played rows again -- a caller that switches envs off and on inside a
    chunk -- is not a window: those envs go through chunk_to_games."""
    assert after_end in ("drop", "new_game")
    if isinstance(chunk_data, TrajectoryChunk):
        chunk = chunk_data
        chunk_data, obs_dim, observations = chunk.data, chunk.rec_obs_dim, chunk.obs if observations is None else observations
    assert torch.is_tensor(chunk_data) and chunk_data.is_cuda, "chunk_to_records reads a device chunk (host arrays: chunk_to_games)"
    T, B, F = chunk_data.shape
    o = int(obs_dim)
    assert F == o + 3 * A + 3, f"record of {F} floats is not obs_dim {o} + 3 * {A} + 3 (image observations live in chunk.obs: obs_dim 0)"
    job = RecordsJob(chunk_data, o, A, discount, td_steps, ignore_termination, after_end, observations)
    return job.finish(priority_scale, limit_of_game_play, keep_partial, observation_shape)


class _StagingPool:
    """Page-locked staging buffers for the transfers of RecordsJobs: one SET of buffers per open job (a process-wide
    parity counter handed a third open job the first one's buffers).  A job acquires a set for its list of (shape, dtype)
    in its constructor and releases it in finish(); released sets are kept for reuse, at most `keep` of them and only for the
    most recent layouts (chunk shapes that no longer occur give their pinned memory back)."""

    def __init__(self, keep=4, alloc=None):
        self.keep, self.free = keep, []                   # free: [(layout, [buffers])], most recently released last
        self.alloc = alloc or (lambda shape, dtype: torch.empty(shape, dtype=dtype, pin_memory=True))

    def acquire(self, layout):
        for i in range(len(self.free) - 1, -1, -1):
            if self.free[i][0] == layout:
                return self.free.pop(i)[1]
        return [self.alloc(shape, dtype) for shape, dtype in layout]

    def release(self, layout, bufs):
        self.free.append((layout, bufs))
        del self.free[:-self.keep]


_STAGING = _StagingPool()


_COPY_POOL = None


def _owned_copies(views):
    """Copies of the staging buffers' arrays that the records own.  The 27 MB record array of a 64 x 4096 chunk takes ~3 ms of a
    single thread; numpy's copy releases the GIL, so its row blocks are copied by four threads (the small arrays by the caller)."""
    global _COPY_POOL
    out = [np.empty_like(v) for v in views]
    big = max(range(len(views)), key=lambda i: views[i].nbytes)
    n = views[big].shape[0]
    if views[big].nbytes >= (4 << 20) and n >= 8:
        if _COPY_POOL is None:
            from concurrent.futures import ThreadPoolExecutor
            _COPY_POOL = ThreadPoolExecutor(max_workers=4, thread_name_prefix="smz-copy")
        cuts = [n * k // 4 for k in range(5)]
        jobs = [_COPY_POOL.submit(np.copyto, out[big][a:b], views[big][a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    else:
        jobs = []
        np.copyto(out[big], views[big])
    for i, v in enumerate(views):
        if i != big:
            np.copyto(out[i], v)
    for j in jobs:
        j.result()
    return out


class RecordsJob:
    """chunk_to_records in two halves.  The constructor ENQUEUES the device half on the current stream and returns at once: game
    ends + n-step targets + priorities (smz_traj_targets_games), the env-major transposes, and asynchronous copies into page-locked
    staging buffers, closed by an event.  finish() waits for that event only -- not for work enqueued later, e.g. the next
    iteration's search -- copies the staging buffers into arrays the records own, and builds the ArrayGameRecords.  Between the
    two a caller can enqueue more GPU work: the host half of iteration k then runs while the GPU searches iteration k + 1
    (self_play_iterations).  Every open job owns its staging buffers (_StagingPool); finish() (or close()) returns them."""

    def __init__(self, chunk_data, obs_dim, A, discount, td_steps=None, ignore_termination=False, after_end="drop",
                 observations=None):
        self.o, self.A, self.discount, self.td_steps = int(obs_dim), int(A), discount, td_steps
        self.ignore_termination, self.after_end = ignore_termination, after_end
        td = 0 if td_steps is None else int(td_steps)
        _, target, err, game_end = chunk_targets(chunk_data, self.o, A, discount, td, ignore_termination, after_end, return_game_end=True)
        # env-major on the device (a game's rows become one contiguous window), then to the host
        dev = [chunk_data.permute(1, 0, 2).contiguous(), game_end.t().contiguous()]
        if td_steps is not None:
            dev += [target.t().contiguous(), err.t().contiguous()]
        self.big_obs = None
        if observations is not None:
            assert self.o == 0
            frames = observations.permute(1, 0, 2).contiguous().to(torch.float32)
            if frames.numel() * 4 > (1 << 30):             # gigabytes of frames: no page-locked staging, a plain (synchronous) copy
                self.big_obs = frames.cpu()
            else:
                dev.append(frames)
        self.has_obs = observations is not None and self.big_obs is None
        self._layout = tuple((tuple(d.shape), d.dtype) for d in dev)
        self.host = _STAGING.acquire(self._layout)
        for h, d in zip(self.host, dev):
            h.copy_(d, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream(chunk_data.device))
        self._dev = dev                                    # (kept alive until the copies have run)

    def finish(self, priority_scale=1, limit_of_game_play=float("inf"), keep_partial=True, observation_shape=None):
        assert self.host is not None, "RecordsJob.finish() called twice"
        self.event.synchronize()
        self._dev = None
        arrays = _owned_copies([h.numpy() for h in self.host[:4 if self.td_steps is not None else 2]])   # the records own these
        rec, game_end = arrays[0], arrays[1]
        target, err = (arrays[2], arrays[3]) if self.td_steps is not None else (None, None)
        observations = self.host[-1].clone() if self.has_obs else self.big_obs
        self.close()
        return records_from_host_copy(rec, game_end, self.o, self.A, self.discount, priority_scale, limit_of_game_play,
                                      self.ignore_termination, keep_partial, self.after_end, observations, observation_shape,
                                      self.td_steps, target, err)

    def close(self):
        """Returns the staging buffers (after the copies into them have run); finish() does it by itself."""
        if self.host is not None:
            self.event.synchronize()
            _STAGING.release(self._layout, self.host)
            self.host = self._dev = None


def records_from_host_copy(rec, game_end, obs_dim, A, discount, priority_scale=1, limit_of_game_play=float("inf"),
                           ignore_termination=False, keep_partial=True, after_end="drop", observations=None,
                           observation_shape=None, td_steps=None, target=None, abs_td=None):
    """The host half of chunk_to_records: env-major arrays rec [B][T][F], game_end [B][T] (int32, -1 = no game) and optionally
    target / abs_td [B][T] for td_steps -> the list of ArrayGameRecord."""
    B, T, _ = rec.shape
    o = int(obs_dim)
    flags = np.zeros((B, T), np.int8) if ignore_termination else rec[:, :, o + 1].astype(np.int8)
    src = ChunkHostCopy(rec, o, A, discount, priority_scale, limit_of_game_play, observations, observation_shape,
                        td_steps, target, abs_td)
    odd = ((flags[:, :-1] == 3) & (flags[:, 1:] != 3)).any(1) if T > 1 else np.zeros(B, bool)
    if odd.any():
        game_end = game_end.copy()
        game_end[odd] = -1
    e, t0, t1, last = _segments(flags, game_end, keep_partial)
    done = (last == 1) & ((t1 - t0) != limit_of_game_play)                      # game.py:270-271
    if src.prio is not None and len(e):
        # every game's largest priority (make_priority's second result) in one pass: maxima over the flat [B * T] ranges
        # [start, end) of the windows -- reduceat over the interleaved (start, end) indices, every other result
        flat = src.prio.reshape(-1)
        lo, hi = e.astype(np.int64) * T + t0, e.astype(np.int64) * T + t1
        idx = np.stack([lo, hi], 1).reshape(-1)
        if idx[-1] >= flat.size:
            idx = idx[:-1]
        tops = [np.float64(v) for v in np.maximum.reduceat(flat, idx)[::2].tolist()]
    else:
        tops = [None] * len(e)
    games = [ArrayGameRecord(src, *w) for w in zip(e.tolist(), t0.tolist(), t1.tolist(), done.tolist(), tops)]
    src.windows = (e, t0, t1)                    # (the games' windows as arrays, in list order: _reward_sums)
    src.fresh = None
    if odd.any():                                                              # the rare envs: the general loop, merged in env order
        merged, k, envs = [], 0, e.tolist()
        for env in np.nonzero(odd)[0].tolist():
            sub = chunk_to_games(rec[env][:, None, :], o, A, discount, priority_scale, limit_of_game_play, ignore_termination,
                                 keep_partial, after_end,
                                 None if observations is None else observations[env][:, None, :], observation_shape)
            while k < len(games) and envs[k] < env:
                merged.append(games[k]); k += 1
            merged.extend(sub)
        merged.extend(games[k:])
        games = merged
    else:
        src.fresh = (games, len(games))          # this very list, unmodified: _reward_sums may sum it from the arrays
    return games


def _sync_active(env, mcts):
    """Hands the env's on/off array to the search (finished games stop consuming simulations, self_play.py:79)."""
    if getattr(env, "active", None) is not None or getattr(mcts, "_active", None) is not None:
        mcts.set_active(getattr(env, "active", None))


def _search_phase(env, heads, mcts, chunk, t, temperature, train=True):
    """First half of ONE env step of all env.B environments on the current stream -- self_play.py:79-94 up to the env's move:
    search the current observation and pick the action (game.py:179-232).  For a host env with split stepping
    (envs.HostVecEnv.step_begin) the download of the actions is enqueued too, so that the host can turn to another env group
    while this search runs.  Returns what _record_phase needs (None: the launch stepped and recorded the built-in env itself)."""
    fused = getattr(env, "fused_step", None)     # built-in env + single-launch search: ONE launch per env step
    env_step = fused(chunk.data, t) if fused is not None else None
    kw = {} if env_step is None else dict(env_step=env_step)
    # image observations: the frame appended at the previous step IS the observation this search starts from, so the
    # representation launch writes it into the record while reading it (smz_vision_initial_record); a search path that
    # does not do that leaves the copy to flush_obs
    owed = getattr(chunk, "owed_obs", None)
    if owed is not None and owed[1] is env.obs and getattr(heads, "records_frames", False):
        kw["record_obs"] = chunk.obs[owed[0]]
    eng = mcts.run(env.obs, heads, train=train, act_temperature=temperature, **kw)
    if owed is not None:
        if getattr(eng, "obs_recorded", False):
            chunk.owed_obs = None
            chunk.apply_patch()
        else:
            chunk.flush_obs()
    if env_step is not None and getattr(eng, "env_stepped", False):
        return None
    out = eng.act(temperature)
    if hasattr(env, "step_begin"):
        env.step_begin(out[0])
    return out


def _record_phase(env, chunk, t, out):
    """Second half: step the env with the chosen actions and append the record (game.py:193-195, 263-267)."""
    if out is None:
        return
    action, policy, child_visits, root_value = out
    if hasattr(env, "step_and_record"):          # built-in env: step + record in one launch
        env.step_and_record(action, chunk.data, t, policy, child_visits, root_value)
        return
    obs, reward, terminated = env.step_end() if hasattr(env, "step_begin") else env.step(action)
    rec_obs = getattr(env, "record_obs", None)   # post-step observation when `obs` already is the next game's reset one
    rec_obs = obs if rec_obs is None else rec_obs
    P = lambda x: C.c_void_p(x.data_ptr())
    split = getattr(chunk, "obs", None) is not None
    if split:                                    # image observations stay float32, outside the float64 record
        if rec_obs is obs:                       # the next search reads this very tensor: its launch copies it (see above)
            chunk.owed_obs = (t, obs)
            # ... except for the rows of envs that were reset inside this step: their post-step frames overwrite the copy
            chunk.owed_patch = getattr(env, "record_patch", None)
        else:
            chunk.obs[t].copy_(rec_obs.reshape(env.B, -1))
    _lib.check(_lib.load().smz_traj_pack(P(chunk.data), chunk.T, t, 0 if split else env.obs_dim, env.num_actions,
                                         None if split else P(rec_obs), P(reward), P(terminated), P(action),
                                         P(policy), P(child_visits), P(root_value), env.B,
                                         C.c_void_p(torch.cuda.current_stream(env.device).cuda_stream)))


def _play_step(env, heads, mcts, chunk, t, temperature, train=True):
    """ONE env step of all env.B environments on the current stream -- the loop body of self_play.py:79-94.  The single code
    path behind play_games and play_games_grouped (which runs the two halves of different env groups interleaved)."""
    _record_phase(env, chunk, t, _search_phase(env, heads, mcts, chunk, t, temperature, train))


def play_games(env, heads, mcts, temperature, steps, chunk=None, train=True, t0=0):
    """Plays `steps` env steps of all env.B environments into rows [t0, t0 + steps) of the TrajectoryChunk (device resident)
    and returns it.  Everything is enqueued asynchronously on the current stream; the caller synchronises."""
    if chunk is None:
        chunk = TrajectoryChunk(t0 + steps, env.B, env.obs_dim, env.num_actions, env.device)
    assert chunk.T >= t0 + steps and chunk.B == env.B
    _sync_active(env, mcts)
    for t in range(t0, t0 + steps):
        _play_step(env, heads, mcts, chunk, t, temperature, train)
    if getattr(chunk, "owed_obs", None) is not None:
        chunk.flush_obs()
    return chunk


class StreamGroup:
    """One slice of a rank's environments with its own engine, head buffers, trajectory chunk and HIP stream.

    A rank may split its envs into G independent groups and run each group's search on its own stream: the groups'
    kernels overlap on the GPU.  Where that pays (measured, tools/groups_probe.sh): from ~262 k envs per GPU on, where the
    step-wise tree kernel is memory-bound and the matrix-core network kernel is not -- two groups run 6-14 % faster than one
    (bench.py does this by itself); at 65 k envs it costs 4 %, and the single-launch search of small batches fills the
    chip with one group.  Trees are independent, so results do not depend on the grouping."""

    def __init__(self, env, heads, mcts, steps, stream=None):
        self.env, self.heads, self.mcts = env, heads, mcts
        self.stream = stream if stream is not None else torch.cuda.Stream(device=env.device)
        self.chunk = TrajectoryChunk(steps, env.B, env.obs_dim, env.num_actions, env.device)


def play_games_grouped(groups, temperature, steps, train=True, t0=0):
    """play_games for several StreamGroups concurrently: step t of every group is enqueued before step t+1 of any,
    each on its group's stream; the caller's stream waits for all of them at the end.  Per group it is play_games' own
    step (_play_step), so the chunks equal the ungrouped ones env by env."""
    dev = groups[0].env.device
    cur = torch.cuda.current_stream(dev)
    for g in groups:
        g.stream.wait_stream(cur)
        assert g.chunk.T >= t0 + steps
        _sync_active(g.env, g.mcts)
    # software pipeline: the search of step t + 1 of a group is enqueued right after its step t, BEFORE the host turns to the
    # next group -- while the host waits for / steps the envs of one group (host envs: envs.HostVecEnv.step_end), the GPU runs
    # the other groups' searches.  Per group the order of operations is play_games' own.
    pend = []
    for g in groups:
        with torch.cuda.stream(g.stream):
            pend.append(_search_phase(g.env, g.heads, g.mcts, g.chunk, t0, temperature, train) if steps > 0 else None)
    for t in range(t0, t0 + steps):
        for k, g in enumerate(groups):
            with torch.cuda.stream(g.stream):
                _record_phase(g.env, g.chunk, t, pend[k])
                if t + 1 < t0 + steps:
                    pend[k] = _search_phase(g.env, g.heads, g.mcts, g.chunk, t + 1, temperature, train)
    for g in groups:
        if getattr(g.chunk, "owed_obs", None) is not None:
            with torch.cuda.stream(g.stream):
                g.chunk.flush_obs()
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


def reanalyse_replay_games(games, model, mcts, device, temperature=0.0, train=True):
    """The reanalyse branch of the reference's play_game (self_play.py:70-81) for many stored games at once: the stored
    game is the environment.  Step i of a stored game with n observations searches observations[i] with the CURRENT
    networks, picks an action from the new tree (game.py:197-216) and records observations[i + 1], the reward
    `rewards[action + 1]` (indexed by the ACTION: game.py:255), the new policy / root value / child visits; the replay
    ends with the step for which i + 2 >= n - 1 (game.py:256), i.e. after n - 2 steps.  The observation sequence does
    not depend on the new actions, so all steps of all games are searched as ONE batch (tree k <-> (game, step) in
    game-major order; its random stream is tree k's).  Returns one GameRecord per stored game with at least 3
    observations, `reanalyzed` set.  `mcts.num_trees` is the batch size per launch (the last batch is padded)."""
    heads = model.heads(device)
    obs, index = [], []
    for gi, g in enumerate(games):
        n = len(g.observations)
        for i in range(max(0, n - 2)):
            obs.append(torch.as_tensor(g.observations[i]).reshape(-1))
            index.append((gi, i))
    out = {}
    B = mcts.num_trees
    for lo in range(0, len(obs), B):
        rows = obs[lo:lo + B]
        batch = torch.stack(rows + [rows[-1]] * (B - len(rows))).to(device=device, dtype=torch.float32).contiguous()
        eng = mcts.run(batch, heads, train=train, act_temperature=temperature)
        action, policy, child_visits, root_value = eng.act(temperature)
        torch.cuda.synchronize(device)
        a, p, cv, rv = (t.cpu().numpy() for t in (action, policy, child_visits, root_value))
        for k, (gi, i) in enumerate(index[lo:lo + B]):
            src = games[gi]
            rec = out.get(gi)
            if rec is None:
                rec = out[gi] = GameRecord(src.discount, src.action_space_size, src.priority_scale, src.limit_of_game_play)
                rec.reanalyzed = True
            if rec.done or rec.game_length >= rec.limit_of_game_play:
                continue                                              # the reference's loop has already stopped
            onehot = np.zeros(src.action_space_size)
            onehot[int(a[k])] = 1
            rec.observations.append(src.observations[i + 1])
            rec.rewards.append(src.rewards[int(a[k]) + 1])
            rec.policies.append(p[k].copy())
            rec.action_history.append(onehot)
            rec.root_values.append(np.float32(rv[k]))
            rec.child_visits.append(cv[k].copy())
            done = i + 2 >= len(src.observations) - 1
            rec.done = done if rec.limit_of_game_play != len(rec.observations) else False
    return [out[gi] for gi in sorted(out)]


def reanalyse_replay_records(games, model, mcts, device, temperature=0.0, train=True, td_steps=None):
    """reanalyse_replay_games for stored ArrayGameRecords, without a Python step per position: the observations of all
    positions are gathered from the records' shared host arrays into one batch, searched, and the reanalysed games come back
    as ArrayGameRecords over ONE new host block (rows: observations[i + 1], rewards[action + 1] -- indexed by the ACTION,
    game.py:255 --, the new policy / action / root value / child visits; the game ends with the step for which
    i + 2 >= n - 1, game.py:256; at most limit_of_game_play steps).  Same games as reanalyse_replay_games, field by field
    (tests/test_gpu_records.py); records that are not pristine ArrayGameRecords go through reanalyse_replay_games."""
    fast = all(isinstance(g, ArrayGameRecord) and g._pristine(*("observations", "rewards")) and g._src.observations is None
               for g in games)
    if not fast or not games:
        return reanalyse_replay_games(games, model, mcts, device, temperature, train)
    heads = model.heads(device)
    src0 = games[0]._src
    o, A = src0.o, src0.A
    F = o + 3 * A + 3
    n = np.array([g._t1 - g._t0 for g in games])
    limit = np.array([min(g.limit_of_game_play, 1 << 40) for g in games], np.int64)
    steps = np.minimum(np.maximum(n - 2, 0), limit)                 # positions searched (and rows produced) per game
    keep = np.nonzero(steps > 0)[0]
    if not len(keep):
        return []
    # position k <-> (game, step i): game-major, as reanalyse_replay_games orders its trees
    gi = np.repeat(keep, steps[keep])
    first = np.cumsum(steps[keep]) - steps[keep]
    step = np.arange(len(gi)) - np.repeat(first, steps[keep])
    # NB reanalyse_replay_games searches n - 2 positions per game even when the limit cuts the replay earlier (tree k's stream
    # belongs to position k of THAT numbering): positions are numbered over min(.., limit) here only when no game is cut
    full = np.maximum(n - 2, 0)
    if (steps != full).any():
        return reanalyse_replay_games(games, model, mcts, device, temperature, train)
    rows = [g._src.rec[g._e, g._t0:g._t1] for g in games]              # views [n_g, F]
    obs = np.concatenate([rows[g][:steps[g], :o] for g in keep]).astype(np.float32)
    P = len(obs)
    B = mcts.num_trees
    a = np.empty(P, np.int64); pol = np.empty((P, A)); cv = np.empty((P, A)); rv = np.empty(P, np.float32)
    for lo in range(0, P, B):
        part = obs[lo:lo + B]
        batch = torch.from_numpy(np.concatenate([part, np.repeat(part[-1:], B - len(part), 0)]) if len(part) < B else part)
        eng = mcts.run(batch.to(device).contiguous(), heads, train=train, act_temperature=temperature)
        action, policy, child_visits, root_value = eng.act(temperature)
        torch.cuda.synchronize(device)
        m = len(part)
        a[lo:lo + m] = action.cpu().numpy()[:m]; pol[lo:lo + m] = policy.cpu().numpy()[:m]
        cv[lo:lo + m] = child_visits.cpu().numpy()[:m]; rv[lo:lo + m] = root_value.cpu().numpy()[:m]
    # the new records, one env-major block [G][Tmax][F]
    G, Tmax = len(keep), int(steps[keep].max())
    rec = np.zeros((G, Tmax, F))
    rec[:, :, o + 1] = 3                                                  # rows behind a game's end: "no step"
    slot = np.repeat(np.arange(G), steps[keep])
    flat = np.concatenate([rows[g] for g in keep])                        # all stored rows of the kept games, game-major
    base = np.repeat(np.cumsum(n[keep]) - n[keep], steps[keep])           # first stored row of position k's game
    if (a + 1 >= np.repeat(n[keep], steps[keep])).any():
        raise IndexError("rewards[action + 1] past the end of a stored game (game.py:255 indexes the rewards by the action)")
    out = np.zeros((P, F))
    out[:, :o] = flat[base + step + 1, :o]
    out[:, o] = flat[base + a + 1, o]
    last = step == np.repeat(steps[keep], steps[keep]) - 1
    out[:, o + 1] = np.where(last, 1, 0)
    out[:, o + 2:o + 2 + A] = pol
    out[np.arange(P), o + 2 + A + a] = 1.0
    out[:, o + 2 + 2 * A] = rv
    out[:, o + 3 + 2 * A:] = cv
    rec[slot, step] = out
    game_end = np.full((G, Tmax), -1, np.int32)
    game_end[slot, step] = np.repeat(steps[keep], steps[keep]).astype(np.int32)
    target = err = None
    g0 = games[int(keep[0])]
    if td_steps is not None:
        dev_chunk = torch.from_numpy(np.ascontiguousarray(rec.transpose(1, 0, 2))).to(device)
        _, t_dev, e_dev = chunk_targets(dev_chunk, o, A, g0.discount, td_steps)
        torch.cuda.synchronize(device)
        target, err = np.ascontiguousarray(t_dev.cpu().numpy().T), np.ascontiguousarray(e_dev.cpu().numpy().T)
    # (one ChunkHostCopy: discount / priority scale / limit of the first game -- stored games of one buffer share them)
    new = records_from_host_copy(rec, game_end, o, A, g0.discount, g0.priority_scale, g0.limit_of_game_play, td_steps=td_steps,
                                 target=target, abs_td=err)
    for g in new:
        g.reanalyzed = True
    return new


def play_game(environment=None, model=None, monte_carlo_tree_search=None, temperature=1, replay_buffer=None):
    """The reference's per-game loop (self_play.py:63-98) with its exact call sequence, for ONE game: `environment` is a
    Game (this package's or the reference's), `monte_carlo_tree_search` any object with run(observation=, model=, train=)
    -> root and `.cycle.global_reset()` -- this package's Monte_carlo_tree_search keeps the tree on the GPU.  Returns
    the played copy of `environment`.  Many games at once: play_games / self_play_iteration."""
    import copy
    environment = copy.deepcopy(environment)
    reanalyse = replay_buffer.should_reanalyse()
    stored = replay_buffer.reanalyse_buffer_sample_game() if reanalyse else None
    if not reanalyse and environment.env.metadata["render_fps"] is None:
        environment.env.metadata["render_fps"] = 30
    counter, step_output = 0, None
    while not environment.terminal and counter < environment.limit_of_game_play:
        feedback = stored if reanalyse else step_output
        state = environment.observation(iteration=counter, feedback=feedback)
        tree = monte_carlo_tree_search.run(observation=state, model=model, train=True)
        step_output = environment.policy_step(root=tree, temperature=temperature, feedback=feedback, iteration=counter)
        environment.store_search_statistics(tree)
        counter += 1
    monte_carlo_tree_search.cycle.global_reset()
    environment.close()
    return environment


def _play_and_gather(env, heads, mcts, temperature, steps, gather):
    """env.reset + `steps` env steps (+ the exchange): (records [T][B_all][F], frames | None, chunk) on the learner, None on actors."""
    env.reset()
    if hasattr(gather, "start"):
        # gather.TrajectoryGather: the chunk is played in slices and every finished slice's rows travel to the learner on a side
        # stream while the next slice is searched; only the last slice's transfer is exposed
        n = max(1, min(gather.slices, steps))
        cuts = [steps * k // n for k in range(n + 1)]
        chunk = TrajectoryChunk(steps, env.B, env.obs_dim, env.num_actions, env.device)
        for k in range(n):
            play_games(env, heads, mcts, temperature, cuts[k + 1] - cuts[k], chunk=chunk, t0=cuts[k])
            gather.start(chunk.data[cuts[k]:cuts[k + 1]], None if chunk.obs is None else chunk.obs[cuts[k]:cuts[k + 1]])
        got = gather.finish()
        return None if got is None else (got[0], got[1], chunk)
    chunk = play_games(env, heads, mcts, temperature, steps)
    data, frames = chunk.data, chunk.obs
    if gather is not None:
        parts = gather(data)
        fparts = gather(frames) if frames is not None else None      # image observations: a float32 message of their own
        if parts is None:
            return None
        data = torch.cat([p for p in parts], dim=1)
        frames = torch.cat([p for p in fparts], dim=1) if fparts is not None else None
    return data, frames, chunk


def _cut_rules(env, steps, ignore_termination, limit_of_game_play):
    on_end = getattr(env, "on_end", "continue")
    limit = limit_of_game_play if limit_of_game_play is not None else (getattr(env, "limit", 0) or steps)
    return dict(limit_of_game_play=limit, ignore_termination=ignore_termination, keep_partial=on_end != "reset",
                after_end="new_game" if on_end == "reset" else "drop", observation_shape=getattr(env, "frame", None))


def _store(games, replay_buffer):
    """save_game of every game + the games' mean reward (self_play.py:266-271).  The reward sums are taken BEFORE the buffer sees
    the games: the records are then exactly what records_from_host_copy built, and a list of array records is summed from the
    windows' arrays without a Python step per game (_reward_sums)."""
    rewards = _reward_sums(games)
    if replay_buffer is not None:
        save = replay_buffer.save_game
        for g in games:
            save(g)
    return games, (sum(rewards) / len(rewards) if rewards else float("nan"))


def self_play_iteration(env, model, mcts, temperature, steps, replay_buffer=None, gather=None, priority_scale=1,
                        ignore_termination=False, limit_of_game_play=None, td_steps=None, records="array"):
    """Self-play half of one learning_cycle iteration (self_play.py:245-271): play, gather to the learner rank,
    hand the games to replay_buffer.save_game, return (games, mean reward) on the learner rank.
    An env built with on_end="reset" plays game after game inside the chunk (only finished games are handed on, the
    unfinished tail is dropped); on_end="mask" plays one game per env and stops searching it when it ends.

    records="array" (default): the games are ArrayGameRecord windows into one host copy of the chunk (chunk_to_records), with
    the value targets and priorities of `td_steps` (default: replay_buffer.td_steps when the buffer has one) computed on the
    device; records="lists": chunk_to_games' per-step Python lists (the checker; ~100x slower at 4096 envs x 64 steps).
    `gather`: a callable slab -> parts (gather.gather_to_learner) or a gather.TrajectoryGather (sliced, overlapped exchange)."""
    got = _play_and_gather(env, model.heads(env.device), mcts, temperature, steps, gather)
    if got is None:
        return None, None
    data, frames, chunk = got
    kw = _cut_rules(env, steps, ignore_termination, limit_of_game_play)
    if records == "array":
        if td_steps is None:
            td_steps = getattr(replay_buffer, "td_steps", None)
        games = chunk_to_records(data, chunk.rec_obs_dim, env.num_actions, mcts.discount, priority_scale, td_steps=td_steps,
                                 observations=frames, **kw)
    else:
        torch.cuda.synchronize(env.device)
        games = chunk_to_games(data, chunk.rec_obs_dim, env.num_actions, mcts.discount, priority_scale, observations=frames, **kw)
    return _store(games, replay_buffer)


def self_play_iterations(env, model, mcts, temperature, steps, iterations, replay_buffer=None, gather=None, priority_scale=1,
                         ignore_termination=False, limit_of_game_play=None, td_steps=None):
    """`iterations` x self_play_iteration as a generator of (games, mean reward), PIPELINED: the search of iteration k + 1 is
    enqueued before the host turns iteration k's chunk into games and stores them, so the host half (transfer, records,
    save_game: ~8 ms for 64 x 4096) hides behind the ~29 ms of search -- the loop then runs at the search kernel's rate.

    For callers whose networks do not change between iterations -- actor ranks, evaluation, number_of_training_before_self_play
    = 0: iteration k + 1 is already running with the weights of the moment it was enqueued when iteration k's games are yielded
    (the reference's Ray fan-out has the same property inside one iteration: every task of it carries the model pickled at its
    start, self_play.py:249-256).  `temperature` may be a callable iteration -> temperature.  Actor ranks yield (None, None)."""
    heads_of = lambda: model.heads(env.device)                                   # noqa: E731  (re-packed if the weights changed)
    if td_steps is None:
        td_steps = getattr(replay_buffer, "td_steps", None)
    kw = _cut_rules(env, steps, ignore_termination, limit_of_game_play)
    cut = {k: kw[k] for k in ("limit_of_game_play", "keep_partial", "observation_shape")}
    job = new = None
    try:
        for it in range(int(iterations)):
            T = temperature(it) if callable(temperature) else temperature
            got = _play_and_gather(env, heads_of(), mcts, T, steps, gather)
            new = None
            if got is not None:
                data, frames, chunk = got
                new = RecordsJob(data, chunk.rec_obs_dim, env.num_actions, mcts.discount, td_steps, kw["ignore_termination"],
                                 kw["after_end"], frames)
            if it > 0:
                done, job = job, None
                yield (None, None) if done is None else _store(done.finish(priority_scale, **cut), replay_buffer)
            job, new = new, None
        if int(iterations) > 0:
            done, job = job, None
            yield (None, None) if done is None else _store(done.finish(priority_scale, **cut), replay_buffer)
    finally:
        # closed early (the consumer raised, or stopped iterating): the jobs still pending give their page-locked staging back
        #  -- RecordsJob.close() waits for the job's own copies first
        for j in (job, new):
            if j is not None:
                j.close()


def _window_sums(src, e, t0, n):
    """Left-to-right sums of the reward windows [t0, t0 + n) of envs e in the chunk copy `src` (arrays)."""
    B_, T = src.rec.shape[:2]
    col = getattr(src, "_reward_col", None)
    if col is None:
        col = src._reward_col = np.ascontiguousarray(src.rec[:, :, src.o])      # [B][T] contiguous, made once per chunk
    if len(e) == B_ and (n == T).all() and (t0 == 0).all() and (e == np.arange(B_)).all():
        # one game per env over the whole chunk: a row scan each, accumulated in float64 like Python's sum of the floats
        # (a float32 scan differed from sum(g.rewards) in the 7th digit for non-integer rewards)
        return np.cumsum(col, axis=1, dtype=np.float64)[:, -1]
    flat = col.reshape(-1)
    order = np.argsort(-n, kind="stable")                          # games ordered by length: step k touches a prefix
    lo_s, n_s = (e * T + t0)[order], n[order]
    L = int(n_s[0]) if len(n_s) else 0
    alive = len(e) - np.searchsorted(n_s[::-1], np.arange(1, L + 1), side="left")
    acc = np.zeros(len(e))
    for k, m in enumerate(alive.tolist()):                         # m games have a step k: one gather + one add each
        acc[:m] += flat[lo_s[:m] + k]
    sums = np.empty(len(e))
    sums[order] = acc
    return sums


def _reward_sums(games):
    """[sum(game.rewards) for game in games] for ArrayGameRecords without touching their lists, bit for bit Python's
    left-to-right sum: the windows of one shared host copy are accumulated together, step k of every game that has one in one
    vector add (a row scan when every env holds one whole-chunk game).  A window is never combined with another game's rows: a
    non-finite reward (the reference's illegal-move reward is -inf when limit_of_game_play is unlimited) stays in its own game
    (differences of a per-env cumulative sum turned inf - inf into nan for every later game of the env).
    The list records_from_host_copy has just built is recognised as a whole (its windows are kept as arrays: no Python step per
    game); any other list is checked record by record, and records whose reward list became a real list are summed the slow way."""
    if not games:
        return []
    g0 = games[0]
    src = getattr(g0, "_src", None)
    fresh = getattr(src, "fresh", None)
    if fresh is not None and fresh[0] is games and fresh[1] == len(games):
        src.fresh = None                                           # (once: after that the records may have been modified)
        e, t0, t1 = src.windows
        e, t0 = np.asarray(e, np.int64), np.asarray(t0, np.int64)
        return _window_sums(src, e, t0, np.maximum(np.asarray(t1, np.int64) - t0, 0)).tolist()
    out = [None] * len(games)
    by_src = {}
    for i, g in enumerate(games):
        if isinstance(g, ArrayGameRecord) and g._pristine("rewards"):
            by_src.setdefault(id(g._src), (g._src, []))[1].append(i)
        else:
            out[i] = sum(g.rewards)
    for src, idx in by_src.values():
        e = np.fromiter((games[i]._e for i in idx), np.int64, len(idx))
        t0 = np.fromiter((games[i]._t0 for i in idx), np.int64, len(idx))
        t1 = np.fromiter((games[i]._t1 for i in idx), np.int64, len(idx))
        sums = _window_sums(src, e, t0, np.maximum(t1 - t0, 0)).tolist()
        for j, i in enumerate(idx):
            out[i] = sums[j]
    return out


def learning_cycle(number_of_iteration=10000, number_of_self_play_before_training=1, number_of_training_before_self_play=1,
                   model_tag_number=124, number_of_worker_selfplay=1, temperature_type="static_temperature", verbose=True,
                   muzero_model=None, gameplay=None, monte_carlo_tree_search=None, replay_buffer=None,
                   steps_per_iteration=None, gather=None, model_directory="model_checkpoint", broadcast=None, pipeline=None):
    """The reference's learning_cycle (self_play.py:168-306), same keyword arguments, assertions and return value
    (epoch_pr, loss, reward, configuration).  What plays the games is chosen the way the reference chooses its backend
    (self_play.py:237-243), by `number_of_worker_selfplay` and by what `gameplay` is:

      * `gameplay` is a vectorised device environment (it has `.B`; envs.CartPoleVec, envs.HostVecEnv ...) and
        `monte_carlo_tree_search` a BatchedMCTS -- or number_of_worker_selfplay == "gpu": the batched GPU engine plays
        all gameplay.B games of an iteration at once (self_play_iteration; `steps_per_iteration` env steps, default
        the env's limit or 500) -- this replaces the Ray fan-out of self_play.py:248-256;
      * otherwise: `number_of_self_play_before_training` sequential play_game calls (self_play.py:258-265; the Ray
        workers of the reference are not part of this build -- a worker count >= 2 with a single-game `gameplay` is
        served sequentially).

    Several ranks (one process per GPU): pass `gather=gather.gather_to_learner`.  Every rank plays its env shard; the
    learner rank (0) receives all trajectories, stores the games, saves and trains; the actor ranks do none of that (their
    reward / loss entries are nan); after the training phase the learner's weights are broadcast to every rank
    (`broadcast`, default gather.broadcast_model) so that the next iteration's searches use them -- the place where the
    reference re-pickles the model into its Ray tasks (self_play.py:249-256).

    The training half (self_play.py:285-288) calls muzero_model.train(replay_buffer.sample_batch()) exactly as the
    reference does; this package's Muzero raises NotImplementedError there (training is the reference's), any model
    object with the reference's train() works.

    `pipeline` (batched engine only): None = pipelined when it cannot change a result, i.e. when
    number_of_training_before_self_play == 0 -- the loop is then driven by self_play_iterations: the search of iteration
    k + 1 is enqueued before the host turns iteration k's chunk into games, stores them and saves the model, so the host half
    hides behind the search (the games, rewards and buffer contents are those of the synchronous loop; the weights never
    change, so the per-iteration broadcast is skipped too).  Note that muzero_model.save_model -- file IO -- then runs while
    the next iteration's search is in flight on the device.  True forces it for a caller whose train() leaves the searching
    weights alone until the loop ends (iteration k + 1 is already running with the weights of the moment it was enqueued when
    iteration k trains); False keeps the synchronous self_play_iteration per iteration.  With training between the
    iterations the loop stays synchronous: iteration k + 1 must search with the weights iteration k's training produced
    (self_play.py:285-288 -> :249-256), so neither the learner nor an actor rank may start it earlier without playing it
    with stale weights."""
    assert isinstance(number_of_iteration, int) and number_of_iteration >= 1, "number_of_iterationt ∈ int | {1 < number_of_iteration < +inf)"
    assert isinstance(number_of_self_play_before_training, int) and number_of_self_play_before_training >= 0, "number_of_self_play_before_training ∈ int | {0 < number_of_self_play_before_training < +inf)"
    assert isinstance(number_of_training_before_self_play, int) and number_of_training_before_self_play >= 0, "number_of_training_before_self_play ∈ int | {0 < number_of_training_before_self_play < +inf)"
    assert isinstance(model_tag_number, int) and model_tag_number >= 0, "model_tag_number ∈ int | {0 < model_tag_number < +inf)"
    assert number_of_worker_selfplay in ("max", "all", "gpu") or (isinstance(number_of_worker_selfplay, int) and number_of_worker_selfplay >= 0), "number_of_worker_selfplay ∈ float | {0 < discount < +inf)"
    assert isinstance(temperature_type, str) and temperature_type in ["reversal_tanh_temperature", "extreme_temperature", "linear_decrease_temperature", "static_temperature", "static_one_temperature"], "temperature_type ∈ {reversal_tanh_temperature,extreme_temperature,linear_decrease_temperature,static_temperature,static_one_temperature} ⊆ str "
    assert isinstance(verbose, bool), "verbose ∈ bool"
    batched = number_of_worker_selfplay == "gpu" or hasattr(gameplay, "B")
    if broadcast is None and gather is not None:
        from . import gather as _gather                       # the learner -> actors hand-off that goes with `gather`
        broadcast = lambda m: _gather.broadcast_model(m, src=0, device=getattr(gameplay, "device", None))
    reward, epoch_pr, loss = [-float("inf")], [], []
    if broadcast is not None:
        broadcast(muzero_model)               # every rank starts from the learner's weights (Ray pickles the learner's model)

    def temperature_of(ep):
        t = temperature_scheduler(number_of_iteration + 1, ep, mode=temperature_type)
        return float(t.reshape(-1)[0]) if isinstance(t, np.ndarray) else t
    pipelined = batched and (number_of_training_before_self_play == 0 if pipeline is None else bool(pipeline))
    piped = None
    if pipelined:
        steps = steps_per_iteration or getattr(gameplay, "limit", 0) or 500
        piped = self_play_iterations(gameplay, muzero_model, monte_carlo_tree_search, lambda it: temperature_of(it + 1), steps,
                                     number_of_iteration, replay_buffer=replay_buffer, gather=gather)
    try:
        for ep in range(1, number_of_iteration + 1):
            temperature = temperature_of(ep)
            learner = True
            if pipelined:
                game, batched_reward = next(piped)                # (iteration ep + 1's search is already enqueued)
                learner = game is not None
                game = game or []
            elif batched:
                steps = steps_per_iteration or getattr(gameplay, "limit", 0) or 500
                # (the games are stored by self_play_iteration itself -- replay_buffer.save_game per game, self_play.py:266-268 --
                #  and their mean reward comes back with them)
                game, batched_reward = self_play_iteration(gameplay, muzero_model, monte_carlo_tree_search, temperature, steps,
                                                           replay_buffer=replay_buffer, gather=gather)
                learner = game is not None        # with `gather`, only the learner rank receives the games (self_play.py:240-256)
                game = game or []
            else:
                game = [play_game(environment=gameplay, model=muzero_model, monte_carlo_tree_search=monte_carlo_tree_search,
                                  temperature=temperature, replay_buffer=replay_buffer)
                        for _ in range(number_of_self_play_before_training)]
            cache_reward, cache_loss = [], []
            if learner:
                if batched:
                    cache_reward = [batched_reward] if game else []
                else:
                    for g in game:
                        replay_buffer.save_game(g)
                        cache_reward.append(sum(g.rewards))
                # (the reference divides by zero when an iteration yields no game; a chunk of on_end="reset" envs may hold no
                # FINISHED game: nan, which never equals max(reward), so nothing is saved for it)
                reward.append(sum(cache_reward) / len(cache_reward) if cache_reward else float("nan"))
                did_better = None if (game and reward[-1] == max(r for r in reward if r == r)
                                      and not all(g.reanalyzed for g in game)) else "do not save"
                if did_better is None and verbose:
                    print("save model with : ", reward[-1], " reward")
                muzero_model.save_model(directory=model_directory, tag=model_tag_number, model_update_or_backtrack=did_better)
                for _ in range(number_of_training_before_self_play):
                    new_priority, batch_game_position = muzero_model.train(replay_buffer.sample_batch())
                    replay_buffer.update_value(new_priority, batch_game_position)
                    cache_loss.append(muzero_model.store_loss[-1][0])
            else:
                reward.append(float("nan"))       # an actor rank neither stores games, nor saves, nor trains
            if broadcast is not None and not (pipelined and number_of_training_before_self_play == 0):
                broadcast(muzero_model)           # the new weights reach the actors (self_play.py:285-288 -> next :249-256)
            loss.append(sum(cache_loss) / len(cache_loss) if cache_loss else float("nan"))   # (the reference divides by 0 here)
            epoch_pr.append(f"EPOCH {ep} || selfplay reward: {reward[-1]} || training loss: {loss[-1]}||")
            if verbose and learner:
                print(epoch_pr[-1], end="\r")
    finally:
        if piped is not None:                 # also when save_model / train / the generator raised: the job in flight gives its
            piped.close()                     # staging buffers back
    configuration = {"number_of_iteration": number_of_iteration,
                     "number_of_self_play_before_training": number_of_self_play_before_training,
                     "number_of_training_before_self_play": number_of_training_before_self_play,
                     "model_tag_number": model_tag_number, "number_of_worker_selfplay": number_of_worker_selfplay,
                     "temperature_type": temperature_type, "verbose": verbose}
    return epoch_pr, loss, reward, configuration
