"""smz_mlp_recurrent at large batches: 16-leaf tiles on the matrix cores (k_mlp_recurrent_mfma, smz_mlp.hip) against the
vector-unit kernel (k_mlp_recurrent) on the same inputs -- the two must agree bit for bit (even / odd accumulator chains,
sums in wave_sum's association), ragged last tile and all-one-branch tiles included."""
import ctypes as C
import os
import sys
from importlib import import_module

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stochastic_muzero_amd  # noqa: F401,E402
import golden_util as gu  # noqa: E402

pytestmark = pytest.mark.gpu


class _Eng:
    def __init__(self, x, branch):
        self.mlp_input, self.branch, self.B = x, branch, x.shape[0]


@pytest.mark.parametrize("wname,B,branches", [("weights_ckpt421", 20000, "mixed"), ("weights_lunar_L0", 16391, "mixed"),
                                               ("weights_ckpt421", 4099, "dyn"), ("weights_ckpt421", 37, "ady")])
def test_matrix_core_recurrent_heads_equal_vector_unit_heads(wname, B, branches, monkeypatch):
    model_mod = import_module("stochastic-muzero_amd.model")
    model = model_mod.Muzero.from_arrays(os.path.join(gu.GOLDEN, wname + ".npz"))
    heads = model.heads("cuda:0", backend="hip")
    g = torch.Generator().manual_seed(5)
    S, A = heads.S, heads.A
    hidden = torch.rand(B, S, generator=g)
    act = torch.randint(0, A, (B,), generator=g)
    x = torch.cat([hidden, torch.nn.functional.one_hot(act, A).float()], dim=1).contiguous().cuda()
    if branches == "mixed":
        br = torch.randint(0, 2, (B,), generator=g).to(torch.uint8)
        br[:16] = 1; br[16:32] = 0                       # whole tiles of one branch
    else:
        br = torch.full((B,), 1 if branches == "dyn" else 0, dtype=torch.uint8)
    br = br.cuda()
    outs = []
    for min_rows in ("0", "-1"):                          # 0: matrix cores for any batch; -1: never
        monkeypatch.setenv("SMZ_MLP_MFMA_MIN", min_rows)
        h, r, p, v = heads.recurrent(_Eng(x, br))
        torch.cuda.synchronize()
        outs.append([t.cpu().numpy().copy() for t in (h, r, p, v)])
    for name, a, b in zip(("hidden", "reward", "policy", "value"), outs[0], outs[1]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), name
    assert np.all(outs[0][1][(br == 0).cpu().numpy()] == 0.0)          # afterstate branch: reward 0


@pytest.mark.parametrize("wname,B,sims,masked", [("weights_ckpt421", 20000, 12, False), ("weights_lunar_L0", 16500, 9, True)])
def test_rows_left_in_the_tree_give_the_same_search(wname, B, sims, masked, monkeypatch):
    """Step-wise search at a large batch with the network kernel taking / putting its rows in the tree's own hidden-state
    storage (smz_set_leaf_ids_out + smz_mlp_recurrent_rows; the tree kernels move no rows) against the same search with
    the rows copied through the mlp_input / hidden staging arrays: identical trees, hidden rows, statistics and streams.
    With some trees switched off (smz_set_active) those must stay untouched."""
    mcts_mod = import_module("stochastic-muzero_amd.mcts")
    model_mod = import_module("stochastic-muzero_amd.model")
    model = model_mod.Muzero.from_arrays(os.path.join(gu.GOLDEN, wname + ".npz"))
    heads = model.heads("cuda:0", backend="hip")
    obs = torch.randn(B, model.observation_dimension, generator=torch.Generator().manual_seed(4)).mul(0.3).cuda()
    active = None
    if masked:
        active = (torch.arange(B) % 5 != 0).to(torch.uint8).cuda()
    res = []
    for lim in ("0", "-1"):
        monkeypatch.setenv("SMZ_MLP_IN_PLACE_MIN", lim)
        m = mcts_mod.BatchedMCTS(B, num_simulations=sims, maxium_action_sample=2, discount=0.997, root_exploration_fraction=0.25,
                                 use_graph=False, single_launch=False)
        m.seed(np.arange(B, dtype=np.uint64) + 3)
        e = m.run(obs, heads, train=True)
        if masked:
            m.set_active(active)
            e = m.run(obs, heads, train=True)
        assert (heads._in_place is e) == (lim == "0")
        visits, priors, rv, cr = e.root_stats()
        torch.cuda.synchronize()
        out = [t.cpu().numpy().copy() for t in (visits, priors, rv, cr)]
        dumps = [e.dump_tree(i) for i in (0, 1, 5, 6, B // 2, B - 1)]
        states = [e.get_rng_state(i) for i in (0, 1, B - 1)]
        res.append((out, dumps, states))
    for a, b in zip(res[0][0], res[1][0]):
        assert np.array_equal(a, b)
    for da, db in zip(res[0][1], res[1][1]):
        for k in da:
            assert np.array_equal(np.asarray(da[k]), np.asarray(db[k])), k
    for (ka, pa), (kb, pb) in zip(res[0][2], res[1][2]):
        assert np.array_equal(ka, kb) and pa == pb


@pytest.mark.gpu
def test_rows_left_in_the_tree_inside_a_captured_graph():
    """The default configuration beyond the single launch's range: step-wise kernels replayed from a HIP graph with the
    network kernel working on the tree's own rows.  Graph replay against eager launches of the same path, two searches
    each (the second replays the captured graph): identical statistics, trees and stream positions."""
    mcts_mod = import_module("stochastic-muzero_amd.mcts")
    model_mod = import_module("stochastic-muzero_amd.model")
    model = model_mod.Muzero.from_arrays(os.path.join(gu.GOLDEN, "weights_ckpt421.npz"))
    heads = model.heads("cuda:0", backend="hip")
    B, sims = 8200, 7
    assert B >= heads.IN_PLACE_MIN
    obs = torch.randn(B, model.observation_dimension, generator=torch.Generator().manual_seed(9)).mul(0.3).cuda()
    res = []
    for graph in (True, False):
        m = mcts_mod.BatchedMCTS(B, num_simulations=sims, maxium_action_sample=2, discount=0.997, root_exploration_fraction=0.25,
                                 use_graph=graph, single_launch=False)
        m.seed(np.arange(B, dtype=np.uint64) + 11)
        for rep in range(2):
            e = m.run(obs, heads, train=True)
        assert heads._in_place is e
        visits, priors, rv, cr = e.root_stats()
        torch.cuda.synchronize()
        out = [t.cpu().numpy().copy() for t in (visits, priors, rv, cr)]
        dumps = [e.dump_tree(i) for i in (0, 17, B - 1)]
        states = [e.get_rng_state(i) for i in (0, B - 1)]
        res.append((out, dumps, states))
    for a, b in zip(res[0][0], res[1][0]):
        assert np.array_equal(a, b)
    for da, db in zip(res[0][1], res[1][1]):
        for k in da:
            assert np.array_equal(np.asarray(da[k]), np.asarray(db[k])), k
    for (ka, pa), (kb, pb) in zip(res[0][2], res[1][2]):
        assert np.array_equal(ka, kb) and pa == pb
