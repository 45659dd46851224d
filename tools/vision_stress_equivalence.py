"""Randomised equivalence stress for the vision family (GPU): smz_search_vision (one launch: block-parallel selection, early parent
planes) against the step-wise kernels on the same seeds -- root statistics, actions, tree dumps and stream positions identical.
python tools/vision_stress_equivalence.py [seed] [cases]"""
import os, sys
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import stochastic_muzero_amd  # noqa: F401
from importlib import import_module
mcts_mod = import_module("stochastic-muzero_amd.mcts"); model_mod = import_module("stochastic-muzero_amd.model")
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nets = {w: model_mod.Muzero.from_state_dicts(os.path.join(R, "tests", "golden", w + ".npz")) for w in ("visionnet_L1_seed0", "visionnet_L2_bn")}
for case in range(n_cases):
    w = list(nets)[rs.randint(len(nets))]
    heads = nets[w].heads("cuda:0", backend="hip")
    B = int(rs.choice([1, 3, 4, 5, 37, 64, 200, 1024]))
    sims = int(rs.choice([0, 1, 2, 7, 16, 31, 50, 62, 63, 64, 65, 100, 126, 127]))
    T = float(rs.choice([0.0, 0.5, 1.0]))
    train = bool(rs.randint(2))
    obs = torch.rand(B, 3, 98, 98, generator=torch.Generator().manual_seed(case)).cuda()
    res = []
    for single in (True, False):
        m = mcts_mod.BatchedMCTS(B, num_simulations=sims, maxium_action_sample=2, discount=0.997, root_exploration_fraction=0.25,
                                 use_graph=False, single_launch=single)
        m.seed(np.arange(B, dtype=np.uint64) * 3 + case)
        for rep in range(2):
            e = m.run(obs, heads, train=train)
        a = [t.clone() for t in e.act(T)]
        st = e.root_stats()
        torch.cuda.synchronize()
        out = [t.cpu().numpy().copy() for t in a] + [t.cpu().numpy().copy() for t in st]
        dumps = [e.dump_tree(i) for i in sorted({0, B // 2, B - 1})]
        res.append((out, dumps, [e.get_rng_state(i) for i in (0, B - 1)]))
    ok = all(np.array_equal(x, y, equal_nan=True) for x, y in zip(res[0][0], res[1][0]))
    ok = ok and all(np.array_equal(np.asarray(da[k]), np.asarray(db[k])) for da, db in zip(res[0][1], res[1][1]) for k in da)
    ok = ok and all(np.array_equal(p[0], q[0]) and p[1] == q[1] for p, q in zip(res[0][2], res[1][2]))
    print(case, w, "B", B, "sims", sims, "T", T, "train", train, "OK" if ok else "MISMATCH", flush=True)
    assert ok
print("all", n_cases, "cases identical")
