"""Shared pieces of the self-play seam tests (test infrastructure).

  TapePlayer      -- a `model` whose five *_inference methods replay the network outputs the REFERENCE recorded for a
                     whole game (tests/golden/selfplay*, reanalyse*, game_illegal*.npz), checking that it is asked for the
                     evaluations the reference asked for;
  OracleSearch    -- a search object with the reference's run(observation=, model=, train=) -> root surface whose tree
                     arithmetic is the CPU oracle's (oracle/smz_oracle.c) and whose random stream is numpy's process-global
                     one, carried in and out -- lets the CPU suite drive play_game / Game end to end without a GPU;
  PickyWalk       -- the one-dimensional env of oracle/gen_golden_r2.py whose step() rejects some moves;
  FakeBuffer      -- the three replay-buffer calls the self-play loop makes.
"""
import math

import numpy as np
import torch


class TapePlayer:
    def __init__(self, data):
        self.d, self.i, self.s, self.root_done = data, -1, 0, True

    def next_step(self):
        self.i, self.s, self.root_done = self.i + 1, 0, False

    def representation_function_inference(self, obs):
        self.next_step()
        np.testing.assert_array_equal(np.asarray(obs, np.float32).reshape(-1), self.d["obs"][self.i])
        return torch.from_numpy(self.d["root_hidden"][self.i][None].copy())

    def _pv(self):
        out = self.d["tape_policy"][self.i][self.s][None].copy(), self.d["tape_value"][self.i][self.s]
        self.s += 1
        return out

    def prediction_function_inference(self, h):
        if not self.root_done:
            self.root_done = True
            return self.d["root_policy"][self.i][None].copy(), np.float32(0)
        return self._pv()

    def afterstate_prediction_function_inference(self, h):
        return self._pv()

    def _check(self, h, a, branch):
        assert self.d["tape_branch"][self.i][self.s] == branch and a == self.d["tape_action"][self.i][self.s]
        assert np.array_equal(np.asarray(h, np.float32).reshape(-1), self.d["tape_hidden_in"][self.i][self.s])

    def afterstate_dynamics_function_inference(self, h, a):
        self._check(h, a, 0)
        return torch.from_numpy(self.d["tape_hidden_out"][self.i][self.s][None].copy())

    def dynamics_function_inference(self, h, a):
        self._check(h, a, 1)
        return self.d["tape_reward"][self.i][self.s], torch.from_numpy(self.d["tape_hidden_out"][self.i][self.s][None].copy())


class _Cycle:
    def __init__(self):
        self.resets = 0

    def global_reset(self):
        self.resets += 1


class OracleSearch:
    def __init__(self, cfg, A=2):
        import orc
        self.orc, self.A = orc, A
        self.kw = dict(pb_c_base=int(cfg["pb_c_base"]), pb_c_init=float(cfg["pb_c_init"]), discount=float(cfg["discount"]),
                       alpha=float(cfg["root_dirichlet_alpha"]), frac=float(cfg["root_exploration_fraction"]))
        self.sims, self.K = int(cfg["num_simulations"]), int(cfg["maxium_action_sample"])
        self.cycle = _Cycle()
        self.runs = 0

    def run(self, observation=None, model=None, train=True):
        from importlib import import_module
        import stochastic_muzero_amd  # noqa: F401
        views = import_module("stochastic-muzero_amd.mcts")._build_views
        h0 = model.representation_function_inference(observation)
        policy, _ = model.prediction_function_inference(h0)
        S = int(np.asarray(h0).size)
        t = self.orc.Tree(self.orc.make_cfg(self.A, self.K, S, self.sims, **self.kw))
        _, key, pos, *_ = np.random.get_state()
        t.set_rng(key, pos)
        t.root_init(np.asarray(policy, np.float32).reshape(-1), hidden=np.asarray(h0, np.float32).reshape(-1), train=train)
        for _ in range(self.sims):
            leaf, parent, act, flag, ph = t.select(want_hidden=True)
            ph = torch.from_numpy(ph[:S].copy()).reshape(tuple(torch.as_tensor(h0).shape))
            if flag:
                reward, h2 = model.dynamics_function_inference(ph, act)
                pol, val = model.prediction_function_inference(h2)
            else:
                reward, h2 = 0.0, model.afterstate_dynamics_function_inference(ph, act)
                pol, val = model.afterstate_prediction_function_inference(h2)
            t.expand_backup(np.asarray(pol, np.float32).reshape(-1), float(val), reward=float(reward),
                            hidden=np.asarray(h2, np.float32).reshape(-1))
        key, pos = t.get_rng()
        np.random.set_state(("MT19937", key, pos, 0, 0.0))
        d = t.dump()
        d["root_priors"] = t.root_stats()[1]
        self.runs += 1
        return views(d, self.A, min(self.K, self.A))


class PickyWalk:
    metadata = {"render_fps": 50}

    def __init__(self):
        self.x, self.n = 0.0, 0

    def reset(self, seed=None):
        self.x, self.n = float(np.random.RandomState(seed).uniform(-0.05, 0.05)), 0
        return np.array([self.x], np.float32), {}

    def step(self, action):
        self.n += 1
        if action == 1 and self.n % 2 == 1:
            raise ValueError("illegal move")
        self.x += 0.1 if action == 1 else -0.1
        return np.array([self.x], np.float32), 1.0, bool(abs(self.x) > 0.35), False, {}

    def close(self):
        pass


class MathCartPole:
    """The stand-in CartPole env oracle/gen_golden.py played the reference's games on (math.cos / math.sin physics)."""
    metadata = {"render_fps": 50}

    def reset(self, seed=None):
        self.state = np.random.RandomState(seed).uniform(-0.05, 0.05, size=4)
        return self.state.astype(np.float32), {}

    def step(self, action):
        x, xd, th, thd = (float(v) for v in self.state)
        force = 10.0 if action == 1 else -10.0
        ct, sn = math.cos(th), math.sin(th)
        temp = (force + 0.05 * thd * thd * sn) / 1.1
        tha = (9.8 * sn - ct * temp) / (0.5 * (4.0 / 3.0 - 0.1 * ct * ct / 1.1))
        xa = temp - 0.05 * tha * ct / 1.1
        self.state = np.array([x + 0.02 * xd, xd + 0.02 * xa, th + 0.02 * thd, thd + 0.02 * tha])
        term = bool(abs(self.state[0]) > 2.4 or abs(self.state[2]) > 12 * 2 * math.pi / 360)
        return self.state.astype(np.float32), 1.0, term, False, {}

    def close(self):
        pass


class FakeBuffer:
    """should_reanalyse / reanalyse_buffer_sample_game / save_game, as self_play.py:70-72, 267-268 call them.  `stored`:
    the game the reanalyse branch replays; `np_state`: numpy's stream as the reference had it right before the replay's
    first search (its own buffer classes draw from that stream while sampling, replay_buffer.py:231-235, 305)."""

    def __init__(self, stored=None, np_state=None):
        self.stored, self.np_state, self.saved = stored, np_state, []

    def should_reanalyse(self):
        return self.stored is not None

    def reanalyse_buffer_sample_game(self):
        if self.np_state is not None:
            np.random.set_state(("MT19937", self.np_state[0], int(self.np_state[1]), 0, 0.0))
        return self.stored

    def save_game(self, g):
        self.saved.append(g)

    def sample_batch(self):
        return ("batch", len(self.saved))

    def update_value(self, new_priority, position):
        self.updated = (new_priority, position)


def assert_game_equals(g, data, prefix="game_"):
    assert g.game_length == int(data[prefix + "length"])
    assert [int(np.argmax(a)) for a in g.action_history] == list(data[prefix + "actions"])
    assert np.array_equal(np.array(g.policies), data[prefix + "policies"])
    assert np.array_equal(np.array(g.child_visits), data[prefix + "child_visits"])
    assert np.array_equal(np.array(g.root_values, np.float32), data[prefix + "root_values"])
    assert np.array_equal(np.array(g.rewards, np.float64), data[prefix + "rewards"])
    obs = np.array([np.asarray(o, np.float32).reshape(-1) for o in g.observations], np.float32)
    assert np.array_equal(obs, data[prefix + "observations"])
    assert bool(g.done) == bool(data[prefix + "done"])
