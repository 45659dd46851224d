"""Golden-vector generator: runs the REFERENCE ITSELF (imported from /root/reference, this container only) and
writes small fixtures to tests/golden/.  TEST INFRASTRUCTURE.

    python oracle/gen_golden.py            # regenerates every fixture (deterministic)

What is recorded per search (`Monte_carlo_tree_search.run`, monte_carlo_tree_search.py:311-349, under
`np.random.seed(seed)`):
  inputs : hyper-parameters, seed, observation, and the NET-OUTPUT TAPE -- every value the five `*_inference`
           wrappers (muzero_model.py:802-909) returned, in call order, together with the arguments they were
           called with (parent hidden state, action) so a consumer can check it asked for the same evaluations;
  outputs: root child visit counts, root child priors (float64), root value, MinMaxStats, per-simulation search
           paths (node ids in creation order), the whole final tree as flat arrays, the numpy stream position
           (probe = next random_sample) and, for several temperatures, what game.py:179-232 (the real
           Game.policy_step / store_search_statistics) makes of the root.
Also written: checkpoint-421 weights and random-init weights as plain float32 arrays (data, for the C/GPU heads),
and whole self-play games produced by the reference's own play_game (self_play.py:63-98) over a stand-in
CartPole env (gymnasium is not installed; the env's physics is ours and is recorded, not pinned).
"""
import copy
import os
import random
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as R  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


# ---------------------------------------------------------------------------------------------------------------
# recording wrappers
# ---------------------------------------------------------------------------------------------------------------
class TapeModel:
    """Delegates to a model exposing the five inference methods and records every call."""

    def __init__(self, inner):
        self.inner = inner
        self.reset()

    def reset(self):
        self.root = {}
        self.sims = []      # one dict per simulation
        self._pending = None

    def representation_function_inference(self, obs):
        h = self.inner.representation_function_inference(obs)
        self.root["hidden"] = np.asarray(h, dtype=np.float32).reshape(-1)
        return h

    def prediction_function_inference(self, h):
        p, v = self.inner.prediction_function_inference(h)
        rec = dict(policy=np.asarray(p, np.float32).reshape(-1), value=np.float32(v))
        if "policy" not in self.root:
            self.root.update(rec)
        else:
            self._pending.update(rec)
            self.sims.append(self._pending)
            self._pending = None
        return p, v

    def afterstate_prediction_function_inference(self, h):
        p, v = self.inner.afterstate_prediction_function_inference(h)
        self._pending.update(policy=np.asarray(p, np.float32).reshape(-1), value=np.float32(v))
        self.sims.append(self._pending)
        self._pending = None
        return p, v

    def afterstate_dynamics_function_inference(self, h, action):
        h2 = self.inner.afterstate_dynamics_function_inference(h, action)
        self._pending = dict(branch=0, action=int(action), reward=np.float32(0),
                             hidden_in=np.asarray(h, np.float32).reshape(-1),
                             hidden_out=np.asarray(h2, np.float32).reshape(-1))
        return h2

    def dynamics_function_inference(self, h, action):
        r, h2 = self.inner.dynamics_function_inference(h, action)
        self._pending = dict(branch=1, action=int(action), reward=np.float32(r),
                             hidden_in=np.asarray(h, np.float32).reshape(-1),
                             hidden_out=np.asarray(h2, np.float32).reshape(-1))
        return r, h2


class CraftedModel:
    """Not a network: returns fixed outputs, to pin degenerate cases (constant policy, equal values)."""

    def __init__(self, A, S, policy, value=0.5, reward=0.25):
        self.A, self.S = A, S
        self.policy = np.asarray(policy, np.float32)[None]
        self.value, self.reward = np.float32(value), np.float32(reward)

    def representation_function_inference(self, obs):
        return torch.zeros(1, self.S)

    def prediction_function_inference(self, h):
        return self.policy.copy(), self.value

    afterstate_prediction_function_inference = prediction_function_inference

    def afterstate_dynamics_function_inference(self, h, action):
        return h + 0.0

    def dynamics_function_inference(self, h, action):
        return self.reward, h + 0.0


def instrument_paths(mcts):
    """Wraps backup to record the search path as node ids in creation order (root 0, its children 1..A, then K
    per expansion) -- the same numbering a flat-array implementation allocates."""
    ids, paths = {}, []
    orig = mcts.back_propagate_and_update_min_max_bound

    def wrapped(search_path, value):
        if not ids:
            ids[id(mcts.root)] = 0
            for c in mcts.root.children.values():
                ids[id(c)] = len(ids)
        for c in search_path[-1].children.values():
            ids[id(c)] = len(ids)
        paths.append([ids[id(n)] for n in search_path])
        return orig(search_path, value)

    mcts.back_propagate_and_update_min_max_bound = wrapped
    return ids, paths


def flatten_tree(root, ids, n_nodes, A):
    if not ids:  # zero simulations: ids were never assigned
        ids[id(root)] = 0
        for c in root.children.values():
            ids[id(c)] = len(ids)
    visit = np.zeros(n_nodes, np.int32); value_sum = np.zeros(n_nodes, np.float32)
    reward = np.zeros(n_nodes, np.float32); prior = np.zeros(n_nodes, np.float32)
    child_base = np.zeros(n_nodes, np.int32); action = np.zeros(n_nodes, np.int32)
    flag = np.zeros(n_nodes, np.int8)
    stack = [(root, 0)]
    while stack:
        node, act = stack.pop()
        i = ids[id(node)]
        visit[i] = node.visit_count; value_sum[i] = np.float32(node.value_sum); reward[i] = np.float32(node.reward)
        prior[i] = np.float32(node.prior); action[i] = act; flag[i] = bool(node.is_chance)
        kids = list(node.children.items())
        if kids:
            child_base[i] = ids[id(kids[0][1])]
            for j, (a, c) in enumerate(kids):
                assert ids[id(c)] == child_base[i] + j
                stack.append((c, int(a)))
    return dict(visit=visit, value_sum=value_sum, reward=reward, prior=prior, child_base=child_base,
                action=action, flag=flag)


class StepEnv:
    """Stand-in for the gym env object that game.py drives (reset/step/close/metadata)."""
    metadata = {"render_fps": 30}

    def __init__(self, obs_dim):
        self.obs_dim = obs_dim

    def reset(self, seed=None):
        return np.zeros(self.obs_dim, np.float32), {}

    def step(self, action):
        return np.zeros(self.obs_dim, np.float32), 1.0, False, False, {}

    def close(self):
        pass


TEMPERATURES = (0.0, 0.2, 0.5, 1.0)


def post_search(ref, root, A, obs_dim, discount):
    """Runs the real Game.policy_step / store_search_statistics for each temperature from the same stream state."""
    state = np.random.get_state()
    out = {}
    for T in TEMPERATURES:
        np.random.set_state(state)
        g = ref.game.Game(gym_env=StepEnv(obs_dim), discount=float(discount), limit_of_game_play=500,
                          observation_dimension=obs_dim, action_dimension=A, rgb_observation=False,
                          action_map=list(range(A)), priority_scale=1)
        g.policy_step(root=root, temperature=T, feedback=None, iteration=0)
        g.store_search_statistics(root)
        key = f"T{T}"
        out[key + "_action"] = np.int32(np.argmax(g.action_history[-1]))
        out[key + "_policy"] = np.asarray(g.policies[-1], np.float64)
        out[key + "_child_visits"] = np.asarray(g.child_visits[-1], np.float64)
        out[key + "_root_value"] = np.float32(g.root_values[-1])
        out[key + "_probe"] = np.float64(np.random.random_sample())
    np.random.set_state(state)
    return out


def run_case(ref, model, obs, seed, mcts_kwargs, train=True, obs_dim=None, with_post=True):
    m = ref.mcts.Monte_carlo_tree_search(**mcts_kwargs)
    ids, paths = instrument_paths(m)
    tape = TapeModel(model)
    np.random.seed(seed)
    root = m.run(observation=obs, model=tape, train=train)
    after = np.random.get_state()
    A = len(root.children)
    K = min(mcts_kwargs["maxium_action_sample"], A)
    sims = mcts_kwargs["num_simulations"]
    n_nodes = 1 + A + sims * K
    rec = dict(seed=np.int64(seed), train=np.int8(train),
               obs=np.asarray(obs, np.float32).reshape(-1),
               root_hidden=tape.root["hidden"], root_policy=tape.root["policy"], root_value_net=tape.root["value"])
    S = tape.root["hidden"].size
    rec["tape_branch"] = np.array([s["branch"] for s in tape.sims], np.int8).reshape(sims)
    rec["tape_action"] = np.array([s["action"] for s in tape.sims], np.int32).reshape(sims)
    rec["tape_reward"] = np.array([s["reward"] for s in tape.sims], np.float32).reshape(sims)
    rec["tape_value"] = np.array([s["value"] for s in tape.sims], np.float32).reshape(sims)
    rec["tape_policy"] = np.array([s["policy"] for s in tape.sims], np.float32).reshape(sims, A)
    rec["tape_hidden_in"] = np.array([s["hidden_in"] for s in tape.sims], np.float32).reshape(sims, S)
    rec["tape_hidden_out"] = np.array([s["hidden_out"] for s in tape.sims], np.float32).reshape(sims, S)
    plen = np.array([len(p) for p in paths], np.int32).reshape(sims)
    pmat = np.full((sims, sims + 2), -1, np.int32)
    for i, p in enumerate(paths):
        pmat[i, :len(p)] = p
    rec["path_len"], rec["paths"] = plen, pmat
    rec.update({"tree_" + k: v for k, v in flatten_tree(root, ids, n_nodes, A).items()})
    kids = list(root.children.values())
    rec["root_visits"] = np.array([c.visit_count for c in kids], np.int32)
    rec["root_priors"] = np.array([np.float64(c.prior) for c in kids], np.float64)
    rec["root_value"] = np.float32(root.value())
    rec["root_visit_count"] = np.int32(root.visit_count)
    rec["minmax"] = np.array([m.min_max_stats.minimum, m.min_max_stats.maximum], np.float32)
    np.random.set_state(after)
    rec["probe"] = np.float64(np.random.random_sample())
    np.random.set_state(after)
    if with_post:
        rec.update(post_search(ref, root, A, obs_dim if obs_dim is not None else rec["obs"].size,
                               mcts_kwargs["discount"]))
    return rec


def stack_cases(cases):
    out = {}
    for k in cases[0]:
        out[k] = np.stack([np.asarray(c[k]) for c in cases])
    return out


def save(name, cfg, cases):
    data = stack_cases(cases)
    for k, v in cfg.items():
        data["cfg_" + k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **data)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB, {len(cases)} cases)")


# ---------------------------------------------------------------------------------------------------------------
# weights as plain arrays
# ---------------------------------------------------------------------------------------------------------------
def export_mlp_weights(mz, path):
    """mlp_model heads -> float32 arrays named as in oracle/orc.py:_MLP_PTRS."""
    def seq(s):  # nn.Sequential: [in, act, (mid, act)*L, out]
        lin = [m for m in s if isinstance(m, torch.nn.Linear)]
        return lin[0], (lin[1] if len(lin) > 2 else None), lin[-1]
    w = {}

    def put(prefix, lin_in, lin_mid, outs):
        H = lin_in.weight.shape[0]
        w[prefix + "_in_w"], w[prefix + "_in_b"] = lin_in.weight, lin_in.bias
        w[prefix + "_mid_w"] = lin_mid.weight if lin_mid is not None else torch.zeros(H, H)
        w[prefix + "_mid_b"] = lin_mid.bias if lin_mid is not None else torch.zeros(H)
        for tag, lin in outs.items():
            w[f"{prefix}_{tag}_w"], w[f"{prefix}_{tag}_b"] = lin.weight, lin.bias

    i, m_, o = seq(mz.representation_function.state_norm); put("rep", i, m_, {"out": o})
    i, m_, o = seq(mz.prediction_function.policy); _, _, ov = seq(mz.prediction_function.value)
    put("pre", i, m_, {"pol": o, "val": ov})
    i, m_, o = seq(mz.afterstate_prediction_function.policy); _, _, ov = seq(mz.afterstate_prediction_function.value)
    put("apr", i, m_, {"pol": o, "val": ov})
    i, m_, o = seq(mz.afterstate_dynamics_function.next_state_normalized); put("ady", i, m_, {"st": o})
    i, m_, o = seq(mz.dynamics_function.next_state_normalized); _, _, orw = seq(mz.dynamics_function.reward)
    put("dyn", i, m_, {"rw": orw, "st": o})
    arrays = {k: v.detach().cpu().numpy().astype(np.float32) for k, v in w.items()}
    enc = [m for m in mz.encoder_function.encoder if isinstance(m, torch.nn.Linear)]
    arrays["enc_in_w"], arrays["enc_in_b"] = (t.detach().numpy().astype(np.float32) for t in (enc[0].weight, enc[0].bias))
    arrays["enc_out_w"], arrays["enc_out_b"] = (t.detach().numpy().astype(np.float32) for t in (enc[-1].weight, enc[-1].bias))
    if len(enc) > 2:
        arrays["enc_mid_w"], arrays["enc_mid_b"] = (t.detach().numpy().astype(np.float32) for t in (enc[1].weight, enc[1].bias))
    dims = dict(obs=int(mz.observation_dimension), A=int(mz.action_dimension), S=int(mz.state_dimension),
                H=int(mz.hidden_layer_dimension), L=int(mz.number_of_hidden_layer))
    np.savez_compressed(path, **arrays, **{"dim_" + k: np.int32(v) for k, v in dims.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def load_ckpt(ref, tag):
    cwd = os.getcwd()
    os.chdir(R.REF)
    try:
        mz = ref.model.Muzero(load=True, type_format=torch.float32)
        mz.load_model(tag=tag, device="cpu")
    finally:
        os.chdir(cwd)
    return mz


def fresh_mlp(ref, obs_dim, A, S=31, H=64, L=0, seed=0):
    torch.manual_seed(seed)
    np_state = np.random.get_state()
    mz = ref.model.Muzero(model_structure="mlp_model",
                          observation_space_dimensions=ref.Box(-1.0, 1.0, shape=(obs_dim,)),
                          action_space_dimensions=ref.Discrete(A), state_space_dimensions=S,
                          hidden_layer_dimensions=H, number_of_hidden_layer=L, k_hypothetical_steps=5,
                          learning_rate=1e-3, device="cpu", use_amp=False, scaler_on=False, num_of_epoch=10)
    np.random.set_state(np_state)
    return mz


def fresh_vision(ref, A=2, L=1, seed=0):
    torch.manual_seed(seed)
    np_state = np.random.get_state()
    mz = ref.model.Muzero(model_structure="vision_model",
                          observation_space_dimensions=ref.Box(0.0, 1.0, shape=(98, 98, 3)),
                          action_space_dimensions=ref.Discrete(A), state_space_dimensions=31,
                          hidden_layer_dimensions=64, number_of_hidden_layer=L, k_hypothetical_steps=5,
                          learning_rate=1e-3, device="cpu", use_amp=False, scaler_on=False, num_of_epoch=10)
    np.random.set_state(np_state)
    return mz


# ---------------------------------------------------------------------------------------------------------------
# stand-in CartPole env for the reference's own play_game
# ---------------------------------------------------------------------------------------------------------------
class CartPoleEnv:
    """CartPole-v1 shaped env (Euler step, gymnasium's published constants).  float64 state, float32 obs."""
    metadata = {"render_fps": 50}

    def __init__(self):
        self.state = None
        self.log_reset_obs = []

    def reset(self, seed=None):
        self.state = np.random.RandomState(seed).uniform(-0.05, 0.05, size=4)
        obs = self.state.astype(np.float32)
        self.log_reset_obs.append(obs.copy())
        return obs, {}

    def step(self, action):
        import math
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


def gen_selfplay(ref, mz, name, sims, temperature, limit, seed):
    """The reference's own play_game (self_play.py:63-98) with per-step tapes."""
    kw = dict(pb_c_base=19652, pb_c_init=1.25, discount=0.999, root_dirichlet_alpha=0.25,
              root_exploration_fraction=0.1, num_simulations=sims, maxium_action_sample=2, number_of_player=1,
              custom_loop=None)
    m = ref.mcts.Monte_carlo_tree_search(**kw)
    tape = TapeModel(mz)
    steps = []
    orig_run = m.run

    def run(observation=None, model=None, train=True):
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
    env = CartPoleEnv()
    game = ref.game.Game(gym_env=env, discount=0.999, limit_of_game_play=limit, observation_dimension=4,
                         action_dimension=2, rgb_observation=False, action_map=[0, 1], priority_scale=0.5)
    rb = ref.replay_buffer.ReplayBuffer(window_size=500, batch_size=128, num_unroll=10, td_steps=50,
                                        game_sampling="priority", position_sampling="priority",
                                        reanalyze_stack=[], reanalyse_fraction=0.2, reanalyse_fraction_mode="chance")
    random.seed(seed)
    np.random.seed(seed)
    g = ref.self_play.play_game(environment=game, model=mz, monte_carlo_tree_search=m, temperature=temperature,
                                replay_buffer=rb)
    probe = np.float64(np.random.random_sample())
    rb.save_game(g)   # replay_buffer.py:109-137 accepts it; record what it derived
    data = stack_cases(steps)
    # make_target (game.py:291-314) for every position: what the training side reads from a stored game
    U, TD = 5, 10
    tv = np.zeros((len(g.root_values), U)); tr = np.zeros((len(g.root_values), U)); tp = np.zeros((len(g.root_values), U, 2))
    for i in range(len(g.root_values)):
        for u, (val, rew, pol) in enumerate(g.make_target(i, U, TD)):
            tv[i, u], tr[i, u], tp[i, u] = val, rew, pol
    data.update(target_values=tv, target_rewards=tr, target_policies=tp, target_unroll=np.int32(U), target_td=np.int32(TD))
    data.update(
        game_actions=np.array([int(np.argmax(a)) for a in g.action_history], np.int32),
        game_action_onehot=np.array(g.action_history, np.float64),
        game_policies=np.array(g.policies, np.float64),
        game_child_visits=np.array(g.child_visits, np.float64),
        game_root_values=np.array(g.root_values, np.float32),
        game_rewards=np.array(g.rewards, np.float64),
        game_observations=np.array([np.asarray(o, np.float32).reshape(-1) for o in g.observations], np.float32),
        game_done=np.int8(g.done), game_length=np.int32(g.game_length),
        buffer_prio_position=np.asarray(rb.prio_position[0], np.float64),
        buffer_prio_game=np.float64(rb.prio_game[0]),
        probe=probe, seed=np.int64(seed), temperature=np.float64(temperature), limit=np.int32(limit))
    for k, v in kw.items():
        if v is not None:
            data["cfg_" + k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **data)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB, {len(steps)} steps)")


def gen_temperature_schedule(ref):
    """temperature_scheduler (self_play.py:124-163) tabulated for every mode over a few horizon lengths."""
    modes = ["reversal_tanh_temperature", "extreme_temperature", "linear_decrease_temperature",
             "static_temperature", "static_one_temperature"]
    rows = []
    for mi, mode in enumerate(modes):
        for epoch in (1, 2, 7, 100, 701):
            for actual in sorted(set([1, 2, epoch // 7 + 1, epoch // 2, epoch // 2 + 1, (3 * epoch) // 4 + 1, epoch])):
                if not 1 <= actual <= epoch:
                    continue
                t = ref.self_play.temperature_scheduler(epoch, actual, mode)
                t = np.nan if t is None else float(np.asarray(t).reshape(-1)[0])
                rows.append((mi, epoch, actual, t))
    path = os.path.join(OUT, "temperature_schedule.npz")
    np.savez_compressed(path, modes=np.array(modes), rows=np.array(rows, np.float64))
    print(f"wrote {path} ({len(rows)} rows)")


def export_state_dicts(mz, path, **meta):
    """The six head modules' state_dicts as flat "<function>/<key>" arrays (+ meta_* scalars)."""
    out = {"meta_" + k: np.asarray(v) for k, v in meta.items()}
    for f in ("representation", "prediction", "afterstate_prediction", "afterstate_dynamics", "dynamics", "encoder"):
        for k, v in getattr(mz, f + "_function").state_dict().items():
            out[f + "/" + k] = v.detach().cpu().numpy()
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def gen_vision_nets(ref):
    """Weights of the reference's vision family + head-level input/output vectors (the reference's own
    *_inference functions, muzero_model.py:802-909)."""
    # (1) the net behind vision_sims50.npz: its tape already holds the head outputs for these weights
    vz = fresh_vision(ref, A=2, L=1, seed=0)
    export_state_dicts(vz, os.path.join(OUT, "visionnet_L1_seed0.npz"), model_structure="vision_model", A=2, S=31,
                       H=64, L=1, obs=0, torch_seed=0)
    # (2) a deeper net with non-trivial batch-norm statistics / affine terms (a trained net's situation)
    A, L, H, S = 3, 2, 32, 21
    torch.manual_seed(5)
    np_state = np.random.get_state()
    vz = ref.model.Muzero(model_structure="vision_model", observation_space_dimensions=ref.Box(0.0, 1.0, shape=(98, 98, 3)),
                          action_space_dimensions=ref.Discrete(A), state_space_dimensions=S, hidden_layer_dimensions=H,
                          number_of_hidden_layer=L, k_hypothetical_steps=5, learning_rate=1e-3, device="cpu",
                          use_amp=False, scaler_on=False, num_of_epoch=10)
    np.random.set_state(np_state)
    g = torch.Generator().manual_seed(11)
    for f in ("representation", "prediction", "afterstate_prediction", "afterstate_dynamics", "dynamics", "encoder"):
        for m in getattr(vz, f + "_function").modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                with torch.no_grad():
                    m.running_mean.copy_(0.1 * torch.randn(m.num_features, generator=g))
                    m.running_var.copy_(0.5 + torch.rand(m.num_features, generator=g))
                    m.weight.copy_(0.5 + torch.rand(m.num_features, generator=g))
                    m.bias.copy_(0.1 * torch.randn(m.num_features, generator=g))
    export_state_dicts(vz, os.path.join(OUT, "visionnet_L2_bn.npz"), model_structure="vision_model", A=A, S=S, H=H,
                       L=L, obs=0, torch_seed=5)
    io = {"obs_seed": np.arange(4000, 4003)}
    hs, pol, val, apol, aval, adyn, dyn_r, dyn_h = [], [], [], [], [], [], [], []
    for sd in io["obs_seed"]:
        obs = torch.tensor(np.random.RandomState(int(sd)).rand(1, 3, 98, 98).astype(np.float32))
        h = vz.representation_function_inference(obs)
        hs.append(h.numpy()[0])
        p, v = vz.prediction_function_inference(h); pol.append(p[0]); val.append(v)
        for a in range(A):
            ha = vz.afterstate_dynamics_function_inference(h, a)
            adyn.append(ha.numpy()[0])
            p, v = vz.afterstate_prediction_function_inference(ha); apol.append(p[0]); aval.append(v)
            r, hn = vz.dynamics_function_inference(ha, a)
            dyn_r.append(r); dyn_h.append(hn.numpy()[0])
    io.update(hidden=np.array(hs, np.float32), policy=np.array(pol, np.float32), value=np.array(val, np.float32),
              afterstate=np.array(adyn, np.float32), apolicy=np.array(apol, np.float32), avalue=np.array(aval, np.float32),
              reward=np.array(dyn_r, np.float32), next_hidden=np.array(dyn_h, np.float32))
    kw = dict(pb_c_base=19652, pb_c_init=1.25, discount=0.997, root_dirichlet_alpha=0.25, root_exploration_fraction=0.25,
              maxium_action_sample=2, number_of_player=1, custom_loop=None, num_simulations=16)
    cases = [run_case(ref, vz, torch.tensor(np.random.RandomState(4100 + s).rand(1, 3, 98, 98).astype(np.float32)),
                      s, kw, obs_dim=4) for s in range(4)]
    for c in cases:
        c["obs"] = c["obs"][:16]
    save("visionL2_sims16", {k: v for k, v in kw.items() if v is not None}, cases)
    np.savez_compressed(os.path.join(OUT, "visionnet_L2_bn_io.npz"), **io)
    print("wrote visionnet_L2_bn_io.npz")


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = R.import_reference()
    torch.set_num_threads(1)
    gen_temperature_schedule(ref)
    if os.environ.get("SMZ_GOLDEN_ONLY") == "temperature":
        return
    if os.environ.get("SMZ_GOLDEN_ONLY") == "vision":
        gen_vision_nets(ref)
        return
    if os.environ.get("SMZ_GOLDEN_ONLY") == "selfplay":
        mz = load_ckpt(ref, 421)
        gen_selfplay(ref, mz, "selfplay421_sims10_T1", sims=10, temperature=1.0, limit=24, seed=0)
        gen_selfplay(ref, mz, "selfplay421_sims11_T02", sims=11, temperature=0.2, limit=24, seed=1)
        gen_selfplay(ref, mz, "selfplay421_sims10_T05", sims=10, temperature=0.5, limit=16, seed=2)
        gen_selfplay(ref, mz, "selfplay421_sims10_T0", sims=10, temperature=0.0, limit=16, seed=3)
        return

    base = dict(pb_c_base=19652, pb_c_init=1.25, discount=0.999, root_dirichlet_alpha=0.25,
                root_exploration_fraction=0.1, maxium_action_sample=2, number_of_player=1, custom_loop=None)

    def cfgd(kw):
        return {k: v for k, v in kw.items() if v is not None}

    # --- checkpoint 421 (CartPole MLP S31/H64/L0) ---------------------------------------------------------------
    mz = load_ckpt(ref, 421)
    export_mlp_weights(mz, os.path.join(OUT, "weights_ckpt421.npz"))
    anchor = torch.tensor([[0.01, -0.02, 0.03, 0.04]])
    for sims, nseeds in ((0, 4), (1, 4), (2, 4), (10, 16), (11, 8), (50, 16), (100, 8)):
        kw = dict(base, num_simulations=sims)
        cases = []
        for seed in range(nseeds):
            obs = anchor if seed < 2 else torch.tensor(
                np.random.RandomState(1000 + seed).uniform(-0.05, 0.05, (1, 4)).astype(np.float32))
            cases.append(run_case(ref, mz, obs, seed, kw))
        save(f"ckpt421_sims{sims}", cfgd(kw), cases)
    # evaluation mode: no Dirichlet noise (mcts:214-218)
    kw = dict(base, num_simulations=25)
    save("ckpt421_sims25_notrain", cfgd(kw), [run_case(ref, mz, anchor, s, kw, train=False) for s in range(8)])
    # the constructor defaults (discount 0.95, exploration fraction 0.25, mcts:76-85)
    kw = dict(pb_c_base=19652, pb_c_init=1.25, discount=0.95, root_dirichlet_alpha=0.25,
              root_exploration_fraction=0.25, num_simulations=30, maxium_action_sample=2, number_of_player=1,
              custom_loop=None)
    save("ckpt421_sims30_defaults", cfgd(kw), [run_case(ref, mz, anchor, s, kw) for s in range(8)])
    # alpha = 1 (exponential branch of the gamma sampler), small pb_c_base
    kw = dict(base, num_simulations=20, root_dirichlet_alpha=1.0, pb_c_base=50, pb_c_init=0.5)
    save("ckpt421_sims20_alpha1", cfgd(kw), [run_case(ref, mz, anchor, s, kw) for s in range(8)])

    # --- LunarLander-shaped random-init MLPs ---------------------------------------------------------------------
    ll = fresh_mlp(ref, 8, 4, L=0, seed=0)
    export_mlp_weights(ll, os.path.join(OUT, "weights_lunar_L0.npz"))
    for K, sims in ((2, 50), (4, 30), (6, 12)):   # K=6 > A exercises min(K, A) (mcts:293)
        kw = dict(base, num_simulations=sims, maxium_action_sample=K)
        cases = [run_case(ref, ll, torch.tensor(np.random.RandomState(2000 + s).randn(1, 8).astype(np.float32)), s, kw)
                 for s in range(12)]
        save(f"lunar_K{K}_sims{sims}", cfgd(kw), cases)
    ll1 = fresh_mlp(ref, 8, 4, S=16, H=32, L=2, seed=1)   # repeated (weight-shared) hidden layer, mlp:36-37
    export_mlp_weights(ll1, os.path.join(OUT, "weights_lunar_L2.npz"))
    kw = dict(base, num_simulations=24, maxium_action_sample=3)
    save("lunarL2_K3_sims24", cfgd(kw),
         [run_case(ref, ll1, torch.tensor(np.random.RandomState(2100 + s).randn(1, 8).astype(np.float32)), s, kw)
          for s in range(8)])
    # wide action space: numpy's 8-lane pairwise summation kicks in at >= 8 addends
    wide = fresh_mlp(ref, 6, 11, S=16, H=32, L=1, seed=2)
    export_mlp_weights(wide, os.path.join(OUT, "weights_wide_A11.npz"))
    kw = dict(base, num_simulations=24, maxium_action_sample=9)
    save("wideA11_K9_sims24", cfgd(kw),
         [run_case(ref, wide, torch.tensor(np.random.RandomState(2200 + s).randn(1, 6).astype(np.float32)), s, kw)
          for s in range(8)])

    # --- vision ResNet-v2 random-init (tree-level parity only needs the tape; hidden state is 3x7x7) ------------
    try:
        vz = fresh_vision(ref, A=2, L=1, seed=0)
        kw = dict(base, num_simulations=50)
        cases = [run_case(ref, vz, torch.tensor(np.random.RandomState(3000 + s).rand(1, 3, 98, 98).astype(np.float32)),
                          s, kw, obs_dim=4) for s in range(4)]
        for c in cases:
            c["obs"] = c["obs"][:16]     # keep the fixture small: the image itself is not needed downstream
        save("vision_sims50", cfgd(kw), cases)
    except Exception as e:  # pragma: no cover
        print("vision goldens skipped:", repr(e))

    gen_vision_nets(ref)

    # --- degenerate crafted cases -------------------------------------------------------------------------------
    kw = dict(base, num_simulations=40)
    cm = CraftedModel(2, 5, [0.5, 0.5], value=0.5, reward=0.25)
    save("crafted_constant_policy", cfgd(kw), [run_case(ref, cm, torch.zeros(1, 4), s, kw) for s in range(8)])
    kw = dict(base, num_simulations=40, maxium_action_sample=2)
    cm = CraftedModel(4, 5, [1.0, 0.0, 0.0, 0.0], value=-1.5, reward=0.0)   # zero-probability actions (+1e-12)
    save("crafted_onehot_policy", cfgd(kw), [run_case(ref, cm, torch.zeros(1, 4), s, kw) for s in range(8)])

    # --- whole games through the reference's own play_game -------------------------------------------------------
    gen_selfplay(ref, mz, "selfplay421_sims10_T1", sims=10, temperature=1.0, limit=24, seed=0)
    gen_selfplay(ref, mz, "selfplay421_sims11_T02", sims=11, temperature=0.2, limit=24, seed=1)
    gen_selfplay(ref, mz, "selfplay421_sims10_T05", sims=10, temperature=0.5, limit=16, seed=2)
    gen_selfplay(ref, mz, "selfplay421_sims10_T0", sims=10, temperature=0.0, limit=16, seed=3)


if __name__ == "__main__":
    main()
