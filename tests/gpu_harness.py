"""Drives the HIP engine through golden tapes, all cases of a fixture as one batch (test infrastructure)."""
import numpy as np
import torch

import golden_util as gu
import stochastic_muzero_amd as smz


def make_engine(cfg, A, S, sims, B, K=None):
    return smz.SearchEngine(num_trees=B, num_actions=A, hidden_size=S, num_simulations=sims,
                            maxium_action_sample=int(cfg["maxium_action_sample"]) if K is None else K,
                            pb_c_base=int(cfg["pb_c_base"]), pb_c_init=float(cfg["pb_c_init"]),
                            discount=float(cfg["discount"]), root_dirichlet_alpha=float(cfg["root_dirichlet_alpha"]),
                            root_exploration_fraction=float(cfg["root_exploration_fraction"]))


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def drive_fixture(name, fused=False, noise_override=None, check_inputs=True):
    """Returns (engine, cfg, data) after replaying every case of the fixture on its own tree."""
    cfg, data = gu.load(name)
    B = data["seed"].shape[0]
    A = data["root_policy"].shape[-1]
    S = data["root_hidden"].shape[-1]
    sims = int(cfg["num_simulations"])
    eng = make_engine(cfg, A, S, sims, B)
    eng.seed(data["seed"].astype(np.uint64))
    train = bool(data["train"][0])
    assert (data["train"] == data["train"][0]).all()
    eng.root_init(dev(data["root_hidden"]), dev(data["root_policy"]), train=train,
                  noise_override=None if noise_override is None else dev(noise_override))
    if sims > 0:
        ph, la, br, xin = eng.select()
    for s in range(sims):
        torch.cuda.synchronize()
        if check_inputs:
            assert np.array_equal(br.cpu().numpy(), data["tape_branch"][:, s].astype(np.uint8)), f"sim {s}: branch"
            assert np.array_equal(la.cpu().numpy(), data["tape_action"][:, s]), f"sim {s}: last action"
            assert np.array_equal(ph.cpu().numpy()[:, :S], data["tape_hidden_in"][:, s]), f"sim {s}: parent hidden"
            x = xin.cpu().numpy()
            assert np.array_equal(x[:, :S], data["tape_hidden_in"][:, s])
            assert np.array_equal(x[:, S:], np.eye(A, dtype=np.float32)[data["tape_action"][:, s]]), f"sim {s}: one-hot"
        args = (dev(data["tape_hidden_out"][:, s]), dev(data["tape_reward"][:, s]), dev(data["tape_policy"][:, s]),
                dev(data["tape_value"][:, s]))
        if fused and s + 1 < sims:
            ph, la, br, xin = eng.expand_backup_select(*args)
        else:
            eng.expand_backup(*args)
            if s + 1 < sims:
                ph, la, br, xin = eng.select()
    torch.cuda.synchronize()
    return eng, cfg, data


def check_fixture_outputs(eng, cfg, data, prior_exact):
    B = data["seed"].shape[0]
    A = data["root_policy"].shape[-1]
    K = min(int(cfg["maxium_action_sample"]), A)
    sims = int(cfg["num_simulations"])
    visits, priors, rv, _ = eng.root_stats()
    torch.cuda.synchronize()
    assert np.array_equal(visits.cpu().numpy(), data["root_visits"])
    # prior_exact=False: the device DREW the Dirichlet noise itself; =True: the reference's sample was injected.  Both are held
    # bit for bit since round 6 (glibc's log / pow restated on the device, csrc/smz_glibc_math.hpp; rounds 1-5: 1e-13 relative)
    assert np.array_equal(priors.cpu().numpy(), data["root_priors"]), ("device-drawn noise" if not prior_exact else "injected noise")
    assert np.array_equal(rv.cpu().numpy(), data["root_value"])
    n = 1 + A + sims * K
    for i in range(B):
        d = eng.dump_tree(i)
        assert d["n_nodes"] == n
        for f in ("visit", "value_sum", "reward", "child_base", "action"):
            assert np.array_equal(d[f][:n], data["tree_" + f][i]), (i, f)
        assert np.array_equal(d["prior"][1 + A:n], data["tree_prior"][i][1 + A:n])
        if sims > 0:
            assert np.array_equal(d["minmax"], data["minmax"][i])
            pl = int(data["path_len"][i][sims - 1])
            assert np.array_equal(d["path"], data["paths"][i][sims - 1][:pl])
        key, pos = eng.get_rng_state(i)
        rs = np.random.RandomState(0)
        rs.set_state(("MT19937", key, pos, 0, 0.0))
        assert rs.random_sample() == data["probe"][i], f"tree {i}: stream position"
