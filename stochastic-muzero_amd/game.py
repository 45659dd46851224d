"""Game records and the single-environment wrapper of the self-play loop.

  GameRecord -- what self-play hands to ReplayBuffer.save_game: the six trajectory lists of game.py:72-77 plus the members
                the buffers read (game_length, reanalyzed, make_target, make_priority; replay_buffer.py:109-137,
                game.py:174-177, 291-337).  The batched engine fills it from a trajectory chunk (selfplay.chunk_to_games).
  Game       -- GameRecord + a live gym-style environment behind the interface the reference's loop drives
                (self_play.py:63-98): observation(), policy_step(root, temperature, feedback, iteration),
                store_search_statistics(root), terminal, limit_of_game_play, close().  `root` is whatever
                Monte_carlo_tree_search.run returns (children keyed by action with visit_count / prior / reward, value()).
                Constructor arguments, attribute names and the assertion rules are the reference's (game.py:12-70).

Behaviour reproduced on purpose (the reference's own tests are its recorded games, tests/golden/selfplay*.npz,
reanalyse*.npz, game_illegal*.npz):
  * the stored observation of step i is the one AFTER action i; the first observation is never stored (game.py:264);
  * an env.step that raises is an illegal move: the observation stays, the reward is min(-len(rewards), -limit, -1) and
    `done` keeps its value (game.py:123-131);
  * at step number limit_of_game_play `done` is forced False (game.py:270-271);
  * the reanalyse branch replays a stored game: observation i+1, reward `feedback.rewards[action + 1]` (indexed by the
    ACTION, as game.py:255 has it), done once iteration + 2 >= len(observations) - 1 (game.py:254-257).
gymnasium and torchvision are not part of this build: the environment is duck-typed (reset(seed=) -> obs | (obs, info);
step(a) -> (obs, reward, terminated, ...)), and the RGB path resizes with torch's bilinear interpolation (what
torchvision's Resize does for tensors); both are parity-unpinned against the reference's third-party stack.
"""
import random

import numpy as np
import torch


class GameRecord:
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

    def _n_step_return(self, cur, td_steps, n, past_end):
        b = cur + td_steps
        value = self.root_values[b] * self.discount ** td_steps if b < n else past_end
        for i, reward in enumerate(self.rewards[cur:b]):
            value += reward * self.discount ** i
        return value

    def make_target(self, state_index, num_unroll, td_steps):
        """[value target, last reward, child_visits] for num_unroll consecutive positions (game.py:291-314):
        n-step return bootstrapped from the search value td_steps ahead; positions past the end are absorbing."""
        n = len(self.root_values)
        targets = []
        for cur in range(state_index, state_index + num_unroll):
            value = self._n_step_return(cur, td_steps, n, 0.0)
            last_reward = self.rewards[cur - 1] if 0 < cur <= len(self.rewards) else 0.0
            if cur < n:
                targets.append([value, last_reward, self.child_visits[cur]])
            else:
                targets.append([0.0, last_reward, np.zeros(self.action_space_size, dtype=np.float64)])
        return targets

    def make_priority(self, td_steps):
        """|root value - n-step return| ** priority_scale per position, and its maximum (game.py:316-337)."""
        n = len(self.root_values)
        target = [self._n_step_return(i, td_steps, n, 0) for i in range(n)]
        pos = np.abs(np.array(self.root_values) - np.array(target)) ** self.priority_scale
        return pos, np.max(pos)

    def make_image(self, index):
        return self.observations[index]

    def make_extended_image(self, index, num_unroll):
        out = []
        for i in range(index, index + num_unroll):
            out.append(self.observations[i] if i < len(self.observations) else out[-1] * 0)
        return out


class _Column:
    """One of a game's six per-step lists (game.py:72-77) as a window into the host copy of a trajectory chunk: `rows` is an
    array slice [n, ...] of the chunk, `item` turns a row into what the reference's list holds at that position.  Reads
    (len, index, slice, iteration, `+`) convert on the fly; the first mutation (append, item assignment, ...) turns the column
    into a real list of its items, after which it simply is that list."""
    __slots__ = ("_rows", "_item", "_list")

    def __init__(self, rows, item):
        self._rows, self._item, self._list = rows, item, None

    def _real(self):
        if self._list is None:
            item = self._item
            self._list = [item(r) for r in self._rows]
            self._rows = None
        return self._list

    def __len__(self):
        return len(self._rows) if self._list is None else len(self._list)

    def __getitem__(self, i):
        if self._list is not None:
            return self._list[i]
        if isinstance(i, slice):
            item = self._item
            return [item(r) for r in self._rows[i]]
        return self._item(self._rows[i])             # (numpy raises IndexError past the end, like a list)

    def __iter__(self):
        if self._list is not None:
            return iter(self._list)
        return map(self._item, self._rows)

    def __add__(self, other):
        return list(self) + list(other)

    def __radd__(self, other):
        return list(other) + list(self)

    def __eq__(self, other):
        return list(self) == list(other)

    def __repr__(self):
        return repr(list(self))

    def __reduce__(self):                             # pickled (ReplayBuffer.save_buffer) as the plain list it stands for
        return (list, (list(self),))

    # mutations: become the list
    def __setitem__(self, i, v): self._real()[i] = v
    def __delitem__(self, i): del self._real()[i]
    def append(self, v): self._real().append(v)
    def extend(self, v): self._real().extend(v)
    def insert(self, i, v): self._real().insert(i, v)
    def pop(self, *a): return self._real().pop(*a)
    def __iadd__(self, other):
        self._real().extend(other)
        return self


class ChunkHostCopy:
    """What the games of one trajectory chunk share: ONE env-major host copy `rec` [B][T][F] (float64) of the chunk's records,
    optionally the float32 observations [B][T][n] that live outside the record, the device-computed n-step value targets /
    |root value - target| [B][T] for `td_steps` (selfplay.chunk_targets: bit-identical to make_target / make_priority), and
    the constants of the games.  ArrayGameRecord reads windows of these arrays."""

    def __init__(self, rec, obs_dim, A, discount, priority_scale, limit_of_game_play, observations=None, observation_shape=None,
                 td_steps=None, target=None, abs_td=None):
        self.rec, self.o, self.A = rec, int(obs_dim), int(A)
        self.discount, self.priority_scale, self.limit_of_game_play = discount, priority_scale, limit_of_game_play
        self.observations, self.observation_shape = observations, tuple(observation_shape) if observation_shape else None
        self.td_steps, self.target, self.abs_td = td_steps, target, abs_td
        self.prio = None if abs_td is None else abs_td ** priority_scale     # make_priority's positions, every game at once


def _obs_vector(row):
    return torch.from_numpy(row.astype(np.float32))[None, ...]       # game.py:145-167: [1, obs] float32


_COLUMNS = ("observations", "rewards", "policies", "action_history", "root_values", "child_visits")
_SHARED = ("discount", "action_space_size", "priority_scale", "limit_of_game_play")


class ArrayGameRecord(GameRecord):
    """A GameRecord whose six lists are windows [t0, t1) of env `e` into a ChunkHostCopy instead of per-step Python objects
    (selfplay.chunk_to_records builds 4096 of them from a 64-step chunk in milliseconds where chunk_to_games -- the checker --
    appends 1.6 M list items).  Everything the reference reads from a stored game works unchanged: len / index / slice /
    iterate / append on the lists (game.py:72-77, replay_buffer.py:185-214), game_length, make_target, make_priority,
    make_image, make_extended_image, reanalyzed, done, env.  make_target / make_priority are served from the device-computed
    arrays when they were computed for the same td_steps and the game's lists have not been modified; otherwise by
    GameRecord's own loops over the lists."""

    # (class-level defaults: a record that was never reanalysed / never had an env carries neither in its dict -- 4096 records are
    #  built per chunk, and every attribute store is ~0.1 us of the loop's exposed host half)
    reanalyzed, env = False, None

    def __init__(self, src, e, t0, t1, done, top=None):
        d = self.__dict__
        d["_src"] = src; d["_e"] = e; d["_t0"] = t0; d["_t1"] = t1; d["_top"] = top; d["done"] = done   # top: max priority of the window (or None)

    def __getattr__(self, name):                      # only reached for attributes not set yet
        if name in _COLUMNS:
            col = self._column(name)
            self.__dict__[name] = col
            return col
        if name in _SHARED:
            src = self.__dict__["_src"]
            return src.A if name == "action_space_size" else getattr(src, name)
        raise AttributeError(name)

    def _column(self, name):
        src, e, t0, t1 = self._src, self._e, self._t0, self._t1
        o, A = src.o, src.A
        rows = src.rec[e, t0:t1]
        if name == "observations":
            if src.observations is None:
                return _Column(rows[:, :o], _obs_vector)
            shape = src.observation_shape
            frames = torch.from_numpy(src.observations[e, t0:t1]) if isinstance(src.observations, np.ndarray) else src.observations[e, t0:t1]
            return _Column(frames, (lambda f: f.reshape((1,) + shape)) if shape else (lambda f: f[None, ...]))
        if name == "rewards":
            return _Column(rows[:, o], float)
        if name == "policies":
            return _Column(rows[:, o + 2:o + 2 + A], _same)
        if name == "action_history":
            return _Column(rows[:, o + 2 + A:o + 2 + 2 * A], _same)
        if name == "root_values":
            return _Column(rows[:, o + 2 + 2 * A], np.float32)
        return _Column(rows[:, o + 3 + 2 * A:o + 3 + 3 * A], _same)

    def _pristine(self, *names):
        d = self.__dict__
        for n in names:
            c = d.get(n)
            if c is not None and not (isinstance(c, _Column) and c._list is None):
                return False
        return True

    @property
    def game_length(self):
        if "action_history" in self.__dict__:
            return len(self.__dict__["action_history"])
        return self._t1 - self._t0

    def make_priority(self, td_steps):
        src, d = self._src, self.__dict__
        if src.prio is not None and td_steps == src.td_steps and td_steps >= 1 and self._t1 > self._t0 and \
                "priority_scale" not in d and self._pristine("rewards", "root_values"):
            # a fresh array per call, as the reference builds one (ReplayBuffer.update_value writes prio_position[game][h] in
            # place, replay_buffer.py:222: a view would let it rewrite the chunk-wide array every record of the chunk reads)
            pos = src.prio[self._e, self._t0:self._t1].copy()
            return pos, (np.max(pos) if self._top is None else self._top)
        return super().make_priority(td_steps)

    def make_target(self, state_index, num_unroll, td_steps):
        src = self._src
        if src.target is None or td_steps != src.td_steps or not self._pristine("rewards", "root_values", "child_visits"):
            return super().make_target(state_index, num_unroll, td_steps)
        e, t0, n = self._e, self._t0, self._t1 - self._t0
        o, A = src.o, src.A
        rec, tgt = src.rec[e], src.target[e]
        targets = []
        for cur in range(state_index, state_index + num_unroll):
            last_reward = float(rec[t0 + cur - 1, o]) if 0 < cur <= n else 0.0
            if 0 <= cur < n:
                # (NEP 50: a bootstrapped return is a numpy float32 in the reference, a return past the end a Python float)
                v = tgt[t0 + cur]
                targets.append([np.float32(v) if cur + td_steps < n else float(v), last_reward,
                                rec[t0 + cur, o + 3 + 2 * A:o + 3 + 3 * A]])
            elif cur < 0:
                return super().make_target(state_index, num_unroll, td_steps)      # negative indices: the lists' own rules
            else:
                targets.append([0.0, last_reward, np.zeros(A, dtype=np.float64)])
        return targets


def _same(row):
    return row


def _resize_frame(shape):
    def transform(frame):
        x = torch.from_numpy(np.ascontiguousarray(frame).astype(np.uint8)).permute(2, 0, 1).to(torch.float32) / 255
        x = torch.nn.functional.interpolate(x[None], size=tuple(shape), mode="bilinear", align_corners=False)
        return x
    return transform


class Game(GameRecord):
    def __init__(self, gym_env=None, discount=0.95, limit_of_game_play=float("inf"), observation_dimension=None,
                 action_dimension=None, rgb_observation=None, action_map=None, priority_scale=1, env_seed=None):
        assert isinstance(discount, float) and discount >= 0, "discount ∈ float | {0 < discount < +inf)"
        assert isinstance(limit_of_game_play, (float, int)) and limit_of_game_play >= 0, "limit_of_game_play ∈ int || float | {1 < limit_of_game_play < +inf)"
        assert isinstance(action_dimension, int) and action_dimension >= 1, "action_dimension ∈ float | {1 < action_dimension < +inf)"
        assert isinstance(rgb_observation, bool), "rgb_observation ∈ bool "
        assert isinstance(priority_scale, (float, int)) and 0 <= priority_scale <= 1, "priority_scale ∈ float | {0 < priority_scale < 1)"
        super().__init__(discount, action_dimension, priority_scale, limit_of_game_play)
        self.env, self.action_map, self.rgb_observation = gym_env, action_map, rgb_observation
        # the reference draws the reset seed from Python's `random` (game.py:102); a fixed env_seed replaces that draw
        self.env_seed = env_seed
        shape = observation_dimension[:-1] if type(observation_dimension) == tuple else None
        self.transform_rgb = _resize_frame(shape) if shape is not None else None

    # ---- environment side -------------------------------------------------------------------------------------------
    def tuple_test_obs(self, x):
        return x[0] if isinstance(x, tuple) else x

    def observation(self, observation_shape=None, iteration=0, feedback=None):
        if iteration == 0 and feedback is None:                       # first observation of a fresh game
            seed = random.randint(0, 100000) if self.env_seed is None else self.env_seed
            state = self.env.reset(seed=seed)
            if self.rgb_observation:
                try:
                    state = self.tuple_test_obs(self.render())
                except Exception:
                    state = self.transform_rgb(self.tuple_test_obs(state))
            else:
                state = self.flatten_state(self.tuple_test_obs(state))
        elif not isinstance(feedback, (tuple, type(None))):           # reanalyse: a stored game is the feedback
            state = feedback.observations[iteration]
            if iteration == 0:
                self.reanalyzed = True
        else:                                                         # the previous step's output
            state = feedback[0]
        self.feedback_state = state
        return state

    def step(self, action):
        try:
            return self.env.step(action)
        except Exception:                                             # illegal move (game.py:123-131)
            return (self.feedback_state, min(-len(self.rewards), -self.limit_of_game_play, -1), self.done)

    def close(self):
        return self.env.close()

    def reset(self):
        self.env.reset()

    def vision(self):
        return self.env.render()

    def render(self):
        return self.transform_rgb(self.env.render())

    def flatten_state(self, state):
        if isinstance(state, tuple):
            rows = [i.tolist() for i in state if isinstance(i, np.ndarray)]
        elif isinstance(state, list):
            rows = state
        elif isinstance(state, np.ndarray):
            rows = state.tolist()
        else:
            try:
                rows = [float(i) for i in state]
            except Exception:
                rows = [float(state)]
        return torch.tensor(rows, dtype=torch.float).flatten()[None, ...]

    # ---- search side ------------------------------------------------------------------------------------------------
    def store_search_statistics(self, root):
        visits = np.array([c.visit_count for c in root.children.values()], dtype=np.float64)
        if visits.sum() >= 3:
            policy = visits / visits.sum()
        else:
            policy = self.softmax_stable(np.array([root.children[u].prior for u in list(root.children.keys())],
                                                  dtype=np.float64), temperature=0)
        self.child_visits.append(policy)
        self.root_values.append(root.value())

    def policy_action_reward_from_tree(self, root):
        keys = list(root.children.keys())
        policy = np.array([root.children[u].visit_count for u in keys], dtype=np.float64)
        if policy.sum() <= 1:
            policy = np.array([root.children[u].prior for u in keys], dtype=np.float64)
        reward = np.array([root.children[u].reward for u in keys], dtype=np.float64)
        return np.array(keys), policy, reward

    def softmax_stable(self, tensor, temperature=1):
        if temperature >= 0.3:
            tensor = tensor ** (1 / temperature)
        return tensor / tensor.sum()

    def select_action(self, action, policy, temperature):
        if temperature > 0.1 or len(set(policy)) == 1:
            return np.random.choice(action, p=policy)
        return action[np.argmax(policy)]

    def onehot_action_encode(self, selected_action):
        out = np.zeros(self.action_space_size)
        out[selected_action] = 1
        return out

    def policy_step(self, root=None, temperature=0, feedback=None, iteration=0):
        action, policy, _ = self.policy_action_reward_from_tree(root)
        policy = self.softmax_stable(policy, temperature=temperature)
        selected = self.select_action(action, policy, temperature)
        if isinstance(feedback, (tuple, type(None))):
            out = self.step(self.action_map[selected])
            if self.rgb_observation:
                try:
                    obs = self.render()
                except Exception:
                    obs = self.transform_rgb(out[0])
            else:
                obs = self.flatten_state(out[0])
            step_val = (obs,) + tuple(out[1:])
        else:
            step_val = [feedback.observations[iteration + 1], feedback.rewards[selected + 1],
                        iteration + 2 >= len(feedback.observations) - 1]
        self.observations.append(step_val[0])
        self.rewards.append(step_val[1])
        self.policies.append(policy)
        self.action_history.append(self.onehot_action_encode(selected))
        self.done = step_val[2] if self.limit_of_game_play != len(self.observations) else False
        return step_val
