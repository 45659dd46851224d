"""Tape-driven search harness (test infrastructure).

`drive_tape(tree, cfg, case)` replays one golden case on an engine exposing the oracle.Tree step interface
(seed / root_init / select / expand_backup / root_stats / dump / act / random_sample) and checks at every
simulation that the engine asked for the evaluation the reference asked for (same parent hidden state, same
action, same branch) before feeding it the recorded network outputs.
"""
import numpy as np

import golden_util as gu


def drive_tape(tree, cfg, case, check_inputs=True):
    A, K, S, sims = gu.dims(cfg, case)
    tree.seed(int(case["seed"]))
    tree.root_init(case["root_policy"], hidden=case["root_hidden"], train=bool(case["train"]))
    for s in range(sims):
        leaf, parent, act, flag, ph = tree.select(want_hidden=True)
        if check_inputs:
            assert flag == int(case["tape_branch"][s]), f"sim {s}: branch"
            assert act == int(case["tape_action"][s]), f"sim {s}: action"
            assert np.array_equal(ph[:S], case["tape_hidden_in"][s]), f"sim {s}: parent hidden"
            pl = int(case["path_len"][s])
            assert leaf == int(case["paths"][s][pl - 1]) and parent == int(case["paths"][s][pl - 2]), f"sim {s}: path"
        tree.expand_backup(case["tape_policy"][s], case["tape_value"][s], reward=case["tape_reward"][s],
                           hidden=case["tape_hidden_out"][s])
    return tree


def check_search_outputs(tree, cfg, case, prior_exact=True):
    A, K, S, sims = gu.dims(cfg, case)
    visits, priors, root_value, child_reward = tree.root_stats()
    assert np.array_equal(visits, case["root_visits"])
    if prior_exact:
        assert np.array_equal(priors, case["root_priors"])
    else:
        np.testing.assert_allclose(priors, case["root_priors"], rtol=1e-12, atol=0)
    assert np.float32(root_value) == case["root_value"]
    d = tree.dump()
    n = 1 + A + sims * K
    assert d["n_nodes"] == n
    for f in ("visit", "value_sum", "reward", "child_base", "action"):
        assert np.array_equal(d[f][:n], case["tree_" + f]), f
    # node 0 has no prior in our layout; root children keep the float32 pre-noise prior, the reference overwrites
    # it with the float64 mixed value (compared above), so compare from the first non-root-child node on
    assert np.array_equal(d["prior"][1 + A:n], case["tree_prior"][1 + A:n])
    if sims > 0:
        assert np.array_equal(d["minmax"], case["minmax"])
