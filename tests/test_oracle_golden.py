"""Pins the CPU oracle (oracle/smz_oracle.c) to golden vectors produced by the reference itself
(oracle/gen_golden.py imported /root/reference: monte_carlo_tree_search.py:311-349, game.py:179-232,
self_play.py:63-98).  Everything here is bit-exact except where a tolerance is written."""
import numpy as np
import pytest

import golden_util as gu
import harness
import orc


def _tree(cfg, case):
    A, K, S, sims = gu.dims(cfg, case)
    c = orc.make_cfg(A, K, S, sims, pb_c_base=int(cfg["pb_c_base"]), pb_c_init=float(cfg["pb_c_init"]),
                     discount=float(cfg["discount"]), alpha=float(cfg["root_dirichlet_alpha"]),
                     frac=float(cfg["root_exploration_fraction"]))
    return orc.Tree(c)


@pytest.mark.parametrize("name", gu.SEARCH_FIXTURES)
def test_tape_driven_search_matches_reference(name):
    cfg, cases = gu.cases(name)
    for case in cases:
        t = harness.drive_tape(_tree(cfg, case), cfg, case)
        harness.check_search_outputs(t, cfg, case)
        key, pos = t.get_rng()
        t2 = _tree(cfg, case); t2.set_rng(key, pos)
        assert t2.random_sample() == case["probe"]          # stream position after the search


@pytest.mark.parametrize("name", gu.SEARCH_FIXTURES)
@pytest.mark.parametrize("T", gu.TEMPERATURES)
def test_post_search_policy_and_action(name, T):
    """game.py:197-232 + :179-195 on the finished root, from the same stream state."""
    cfg, cases = gu.cases(name)
    for case in cases:
        t = harness.drive_tape(_tree(cfg, case), cfg, case, check_inputs=False)
        action, policy, child_visits, root_value = t.act(T)
        k = f"T{T}"
        assert action == int(case[k + "_action"])
        assert np.array_equal(policy, case[k + "_policy"])
        assert np.array_equal(child_visits, case[k + "_child_visits"])
        assert root_value == case[k + "_root_value"]
        assert t.random_sample() == case[k + "_probe"]


@pytest.mark.parametrize("name", gu.SELFPLAY_FIXTURES)
def test_whole_game_tape_driven(name):
    """The reference's own play_game: one numpy stream across all env steps (search -> action draw -> search)."""
    cfg, data = gu.load(name)
    steps = data["obs"].shape[0]
    T = float(data["temperature"])
    case0 = {k: data[k][0] for k in ("root_policy", "root_hidden")}
    t = _tree(cfg, case0)
    t.seed(int(data["seed"]))
    sims = int(cfg["num_simulations"])
    for i in range(steps):
        t.root_init(data["root_policy"][i], hidden=data["root_hidden"][i], train=True)
        for s in range(sims):
            leaf, parent, act, flag, ph = t.select(want_hidden=True)
            assert flag == data["tape_branch"][i][s] and act == data["tape_action"][i][s]
            assert np.array_equal(ph, data["tape_hidden_in"][i][s])
            t.expand_backup(data["tape_policy"][i][s], data["tape_value"][i][s], reward=data["tape_reward"][i][s],
                            hidden=data["tape_hidden_out"][i][s])
        visits, priors, rv, _ = t.root_stats()
        assert np.array_equal(visits, data["root_visits"][i])
        assert np.array_equal(priors, data["root_priors"][i])
        action, policy, child_visits, root_value = t.act(T)
        assert action == data["game_actions"][i]
        assert np.array_equal(policy, data["game_policies"][i])
        assert np.array_equal(child_visits, data["game_child_visits"][i])
        assert root_value == data["game_root_values"][i]
    assert t.random_sample() == data["probe"]


WEIGHTED = [("ckpt421_sims10", "weights_ckpt421"), ("ckpt421_sims50", "weights_ckpt421"),
            ("ckpt421_sims100", "weights_ckpt421"), ("lunar_K2_sims50", "weights_lunar_L0"),
            ("lunar_K4_sims30", "weights_lunar_L0"), ("lunarL2_K3_sims24", "weights_lunar_L2"),
            ("wideA11_K9_sims24", "weights_wide_A11"), ("ckpt450_sims11", "weights_ckpt450")]


@pytest.mark.parametrize("name,wname", WEIGHTED)
def test_c_heads_match_reference_heads(name, wname):
    """The oracle's plain-C mlp_model heads vs the recorded torch-CPU outputs: float tolerance 1e-5
    (muzero_model.py:802-909; neural_network_mlp_model.py:5-250)."""
    import ctypes as C
    import os
    cfg, cases = gu.cases(name)
    w = orc.MlpWeights.from_npz(os.path.join(gu.GOLDEN, wname + ".npz"))
    L = orc.lib()
    A, S = w.dims["A"], w.dims["S"]
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    for case in cases[:4]:
        h = np.zeros(S, np.float32)
        L.orc_mlp_representation(C.byref(w.struct), p(np.ascontiguousarray(case["obs"])), p(h))
        np.testing.assert_allclose(h, case["root_hidden"], atol=1e-5)
        pol = np.zeros(A, np.float32); val = C.c_float()
        L.orc_mlp_prediction(C.byref(w.struct), p(np.ascontiguousarray(case["root_hidden"])), p(pol), C.byref(val))
        np.testing.assert_allclose(pol, case["root_policy"], atol=1e-5)
        np.testing.assert_allclose(val.value, case["root_value_net"], rtol=1e-4, atol=5e-4)
        for s in range(len(case["tape_branch"])):
            hin = np.ascontiguousarray(case["tape_hidden_in"][s]); h2 = np.zeros(S, np.float32)
            r = C.c_float(0.0)
            if case["tape_branch"][s]:
                L.orc_mlp_dynamics(C.byref(w.struct), p(hin), int(case["tape_action"][s]), C.byref(r), p(h2))
                L.orc_mlp_prediction(C.byref(w.struct), p(h2), p(pol), C.byref(val))
            else:
                L.orc_mlp_afterstate_dynamics(C.byref(w.struct), p(hin), int(case["tape_action"][s]), p(h2))
                L.orc_mlp_afterstate_prediction(C.byref(w.struct), p(h2), p(pol), C.byref(val))
            np.testing.assert_allclose(h2, case["tape_hidden_out"][s], atol=2e-5)
            # the inverse support transform (muzero_model.py:589-590) cancels twice in float32: one ulp of the
            # softmax expectation moves the decoded scalar by ~1e-4 near zero, so decoded scalars get 5e-4
            np.testing.assert_allclose(r.value, case["tape_reward"][s], rtol=1e-4, atol=5e-4)
            np.testing.assert_allclose(pol, case["tape_policy"][s], atol=1e-5)
            np.testing.assert_allclose(val.value, case["tape_value"][s], rtol=1e-4, atol=5e-4)


def test_end_to_end_with_c_heads_agrees_with_reference_search():
    """Full search with the C heads (no tape): head outputs differ from torch-CPU in the last bits, so the tree
    is allowed to diverge on a small fraction of seeds; visit counts must agree on the rest."""
    import os
    cfg, cases = gu.cases("ckpt421_sims50")
    w = orc.MlpWeights.from_npz(os.path.join(gu.GOLDEN, "weights_ckpt421.npz"))
    same = 0
    for case in cases:
        t = _tree(cfg, case); t.seed(int(case["seed"]))
        t.run_mlp(w, case["obs"], train=True)
        visits, priors, rv, _ = t.root_stats()
        same += int(np.array_equal(visits, case["root_visits"]))
        np.testing.assert_allclose(priors, case["root_priors"], rtol=1e-5)
        assert abs(float(rv) - float(case["root_value"])) < 1e-2 * max(1.0, abs(float(case["root_value"])))
    assert same >= len(cases) - 2
