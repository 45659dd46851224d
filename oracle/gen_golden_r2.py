"""Round-2 golden vectors, again produced by running the REFERENCE ITSELF (imported from /root/reference, this container
only).  TEST INFRASTRUCTURE.  oracle/gen_golden.py and its fixtures are untouched; this script adds

    python oracle/gen_golden_r2.py

  mlpnet_seed7.npz            the six mlp_model modules as the reference's Muzero(...) constructor initialises them under
                              torch.manual_seed(7) (muzero_model.py:300-358) -- pins compat_mlp's construction order;
  reanalyse421_sims10_T1.npz  the reference's own play_game on its REANALYSE branch (self_play.py:70-81, game.py:112-115,
  reanalyse421_sims10_T0.npz  254-257): a game it has just played is the stored game; per-step tapes + the resulting lists;
  game_illegal_moves.npz      the reference's Game over an env whose step() raises for some actions: the illegal-move
                              reward rule of game.py:123-131, through the reference's own play_game;
  weights_cfg434shape.npz     a random-init mlp_model of the OTHER network shape the reference's configs ship
  cfg434shape_sims11.npz      (config/experiment_434_config.json: state_space_dimensions 61, hidden_layer_dimensions 126,
                              number_of_hidden_layer 0, CartPole, num_simulations 11) and the reference's searches on it --
                              too wide for the LDS-resident kernels: the torch-GEMM heads + step-wise tree kernels serve it.
"""
import os
import random
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as R  # noqa: E402
import gen_golden as G  # noqa: E402

OUT = G.OUT
KW = dict(pb_c_base=19652, pb_c_init=1.25, discount=0.999, root_dirichlet_alpha=0.25, root_exploration_fraction=0.1,
          num_simulations=10, maxium_action_sample=2, number_of_player=1, custom_loop=None)


def recording_search(ref, mz, sims, steps, states=None):
    """The reference's search object with run() wrapped so that every call leaves its net-output tape in `steps` (and the
    numpy stream state right before the call in `states`)."""
    kw = dict(KW, num_simulations=sims)
    m = ref.mcts.Monte_carlo_tree_search(**kw)
    tape = G.TapeModel(mz)
    orig_run = m.run

    def run(observation=None, model=None, train=True):
        if states is not None:
            _, key, pos, *_ = np.random.get_state()
            states.append((key.copy(), int(pos)))
        tape.reset()
        root = orig_run(observation=observation, model=tape, train=train)
        A = len(root.children)
        steps.append(dict(
            obs=np.asarray(observation, np.float32).reshape(-1),
            root_hidden=tape.root["hidden"], root_policy=tape.root["policy"],
            tape_branch=np.array([s["branch"] for s in tape.sims], np.int8),
            tape_action=np.array([s["action"] for s in tape.sims], np.int32),
            tape_reward=np.array([s["reward"] for s in tape.sims], np.float32),
            tape_value=np.array([s["value"] for s in tape.sims], np.float32),
            tape_policy=np.array([s["policy"] for s in tape.sims], np.float32).reshape(sims, A),
            tape_hidden_in=np.array([s["hidden_in"] for s in tape.sims], np.float32).reshape(sims, -1),
            tape_hidden_out=np.array([s["hidden_out"] for s in tape.sims], np.float32).reshape(sims, -1),
            root_visits=np.array([c.visit_count for c in root.children.values()], np.int32),
            root_priors=np.array([np.float64(c.prior) for c in root.children.values()], np.float64),
            search_root_value=np.float32(root.value())))
        return root

    m.run = run
    return m, kw


def game_arrays(g, prefix):
    return {prefix + "actions": np.array([int(np.argmax(a)) for a in g.action_history], np.int32),
            prefix + "policies": np.array(g.policies, np.float64),
            prefix + "child_visits": np.array(g.child_visits, np.float64),
            prefix + "root_values": np.array(g.root_values, np.float32),
            prefix + "rewards": np.array(g.rewards, np.float64),
            prefix + "observations": np.array([np.asarray(o, np.float32).reshape(-1) for o in g.observations], np.float32),
            prefix + "done": np.int8(bool(g.done)), prefix + "length": np.int32(g.game_length),
            prefix + "reanalyzed": np.int8(bool(g.reanalyzed))}


def save(name, data, kw):
    for k, v in kw.items():
        if v is not None:
            data["cfg_" + k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **data)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def gen_reanalyse(ref, mz, name, temperature, limit, seed):
    game = ref.game.Game(gym_env=G.CartPoleEnv(), discount=0.999, limit_of_game_play=limit, observation_dimension=4,
                         action_dimension=2, rgb_observation=False, action_map=[0, 1], priority_scale=0.5)
    stack = ref.replay_buffer.ReanalyseBuffer()
    rb = ref.replay_buffer.ReplayBuffer(window_size=500, batch_size=128, num_unroll=10, td_steps=50,
                                        game_sampling="priority", position_sampling="priority",
                                        reanalyze_stack=[stack], reanalyse_fraction=1.0, reanalyse_fraction_mode="chance")
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    steps0 = []
    m, kw = recording_search(ref, mz, 10, steps0)
    assert rb.should_reanalyse() is False                       # nothing stored yet: the first game is a fresh one
    src = ref.self_play.play_game(environment=game, model=mz, monte_carlo_tree_search=m, temperature=1.0, replay_buffer=rb)
    rb.save_game(src)                                           # also lands in the reanalyse stack (replay_buffer.py:135-136)
    assert len(stack.buffer) == 1 and not src.reanalyzed
    steps, states = [], []
    m, kw = recording_search(ref, mz, 10, steps, states)
    assert bool(rb.should_reanalyse())
    g = ref.self_play.play_game(environment=game, model=mz, monte_carlo_tree_search=m, temperature=temperature,
                                replay_buffer=rb)
    probe = np.float64(np.random.random_sample())
    assert g.reanalyzed
    data = G.stack_cases(steps)
    data.update(game_arrays(src, "src_"))
    data.update(game_arrays(g, "game_"))
    data.update(np_key_before_first_search=states[0][0], np_pos_before_first_search=np.int32(states[0][1]),
                probe=probe, seed=np.int64(seed), temperature=np.float64(temperature), limit=np.int32(limit))
    save(name, data, kw)


class PickyWalk:
    """One-dimensional random walk behind the gym call shape (observation = position, actions: 0 left / 1 right by 0.1,
    terminated beyond |x| > 0.35), except that stepping right on an odd step is rejected with an exception -- what an env
    does with an illegal move.  The observation has ONE component on purpose: the reference's handler (game.py:123-131)
    returns the previous, already flattened observation, and Game.flatten_state (game.py:145-167) only takes that back
    for single-component observations (for wider ones it raises ValueError)."""
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


def gen_illegal(ref):
    limit, seed = 14, 5
    mz = G.fresh_mlp(ref, 1, 2, S=7, H=8, L=0, seed=9)
    game = ref.game.Game(gym_env=PickyWalk(), discount=0.999, limit_of_game_play=limit, observation_dimension=1,
                         action_dimension=2, rgb_observation=False, action_map=[0, 1], priority_scale=1)
    rb = ref.replay_buffer.ReplayBuffer(window_size=500, batch_size=128, num_unroll=10, td_steps=50,
                                        reanalyze_stack=[], reanalyse_fraction=0.0, reanalyse_fraction_mode="chance")
    random.seed(seed)
    np.random.seed(seed)
    steps = []
    m, kw = recording_search(ref, mz, 10, steps)
    g = ref.self_play.play_game(environment=game, model=mz, monte_carlo_tree_search=m, temperature=1.0, replay_buffer=rb)
    probe = np.float64(np.random.random_sample())
    data = G.stack_cases(steps)
    data.update(game_arrays(g, "game_"))
    assert (data["game_rewards"] < 0).any(), "no illegal move happened: pick another seed"
    data.update(probe=probe, seed=np.int64(seed), temperature=np.float64(1.0), limit=np.int32(limit))
    save("game_illegal_moves", data, kw)


def main():
    ref = R.import_reference()
    torch.set_num_threads(1)
    if "--only-cfg434" in sys.argv:
        gen_cfg434(ref)
        return
    net = G.fresh_mlp(ref, 4, 2, S=7, H=8, L=1, seed=7)
    G.export_state_dicts(net, os.path.join(OUT, "mlpnet_seed7.npz"), model_structure="mlp_model", A=2, S=7, H=8, L=1, obs=4,
                         torch_seed=7)
    mz = G.load_ckpt(ref, 421)
    gen_reanalyse(ref, mz, "reanalyse421_sims10_T1", temperature=1.0, limit=20, seed=4)
    gen_reanalyse(ref, mz, "reanalyse421_sims10_T0", temperature=0.0, limit=12, seed=6)
    gen_illegal(ref)
    gen_cfg434(ref)


def gen_cfg434(ref):
    wide = G.fresh_mlp(ref, 4, 2, S=61, H=126, L=0, seed=3)
    G.export_mlp_weights(wide, os.path.join(OUT, "weights_cfg434shape.npz"))
    kw = dict(KW, num_simulations=11)
    cases = [G.run_case(ref, wide, torch.tensor(np.random.RandomState(4340 + s).uniform(-0.05, 0.05, (1, 4)).astype(np.float32)), s, kw)
             for s in range(8)]
    G.save("cfg434shape_sims11", {k: v for k, v in kw.items() if v is not None}, cases)


if __name__ == "__main__":
    main()
