"""Randomised equivalence stress (GPU): the single-launch search (all its instantiations, chosen by geometry / shape)
against the step-wise kernels on the same seeds -- root statistics, actions and stream positions must be identical."""
import os, sys
import numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import stochastic_muzero_amd  # noqa: F401
from importlib import import_module
mcts_mod = import_module("stochastic-muzero_amd.mcts"); model_mod = import_module("stochastic-muzero_amd.model")
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
nets = {w: model_mod.Muzero.from_arrays(os.path.join(R, "tests", "golden", w + ".npz")) for w in
        ("weights_ckpt421", "weights_lunar_L0", "weights_lunar_L2", "weights_wide_A11")}
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for case in range(n_cases):
    w = list(nets)[rs.randint(len(nets))]
    model = nets[w]; heads = model.heads("cuda:0", backend="hip")
    A = model.action_dimension
    B = int(rs.choice([1, 2, 3, 63, 64, 65, 511, 2047, 2049, 4096, 4096, 4096, 4097, 6001]))
    sims = int(rs.choice([0, 1, 2, 5, 13, 31, 50, 53, 54, 70, 100, 126, 127]))     # (> 53: trees in global memory; <= 126: block-parallel selection there)
    K = 2 if rs.randint(3) == 0 else int(rs.randint(1, A + 2))      # (K = 2: the specialised instantiations)
    T = float(rs.choice([0.0, 0.2, 0.5, 1.0]))
    train = bool(rs.randint(2))
    obs = torch.randn(B, model.observation_dimension, generator=torch.Generator().manual_seed(case)).mul(0.3).cuda()
    out = []
    for single in (True, False):
        m = mcts_mod.BatchedMCTS(B, num_simulations=sims, maxium_action_sample=K, discount=0.997, root_exploration_fraction=0.25,
                                 use_graph=False, single_launch=single)
        m.seed(np.arange(B, dtype=np.uint64) * 7 + case)
        for rep in range(2):
            e = m.run(obs, heads, train=train, act_temperature=T if (single and rep) else None)
            a = [t.clone() for t in e.act(T)]
        st = e.root_stats()
        torch.cuda.synchronize()
        out.append(([t.cpu().numpy() for t in a] + [t.cpu().numpy().copy() for t in st], [e.get_rng_state(i) for i in (0, B - 1)]))
    ok = all(np.array_equal(x, y, equal_nan=True) for x, y in zip(out[0][0], out[1][0])) and \
        all(np.array_equal(p[0], q[0]) and p[1] == q[1] for p, q in zip(out[0][1], out[1][1]))
    print(case, w, "B", B, "sims", sims, "K", K, "T", T, "train", train, "OK" if ok else "MISMATCH", flush=True)
    assert ok
print("all", n_cases, "cases identical")
