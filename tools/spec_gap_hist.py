#!/usr/bin/env python3
"""Stage 0 of the speculative-evaluation study, CPU only: for the BASELINE workloads, how many simulation
rounds lie between the expansion that CREATES a node and the simulation that selects it as its leaf (the round in which its
network outputs are needed) -- a child's outputs depend only on (parent hidden state, action), known when the parent is
expanded (monte_carlo_tree_search.py:270-286, 333-342) -- and which share of the created children is never evaluated at all
(what evaluating every child ahead of time would waste).

Trees come from the CPU oracle with its own plain-C heads (oracle/smz_oracle.c), seeds 0..n-1, train=True, the bench's
observations.  Everything is read off the finished tree: node n was created by expansion (n - 1 - A) // K (root children:
the root expansion, round -1) and is the leaf of simulation (child_base[n] - 1 - A) / K.

    python tools/spec_gap_hist.py [--trees 4096] > profiles/r05_spec_gap_histogram.json"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import orc  # noqa: E402

WORK = [("cartpole_mlp_4096x50 (C2)", "weights_ckpt421.npz", 4, 2, 2, 50),
        ("lunarlander_mlp_4096x50 (C3, K 2)", "weights_lunar_L0.npz", 8, 4, 2, 50),
        ("lunarlander_mlp_4096x50_K4 (C3 stress)", "weights_lunar_L0.npz", 8, 4, 4, 50),
        ("cartpole_mlp_4096x100 (C5 per rank)", "weights_ckpt421.npz", 4, 2, 2, 100)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trees", type=int, default=4096)
    a = ap.parse_args()
    out = {"what": __doc__.split("\n\n")[0].replace("\n", " "), "trees_per_workload": a.trees, "workloads": {}}
    for name, wfile, obs_dim, A, K, sims in WORK:
        w = orc.MlpWeights.from_npz(os.path.join(ROOT, "tests", "golden", wfile))
        cfg = orc.make_cfg(A, K, w.dims["S"], sims, discount=0.999, alpha=0.25, frac=0.1)
        rs = np.random.RandomState(0)
        obs = (rs.uniform(-0.05, 0.05, (a.trees, obs_dim)) if obs_dim == 4 else rs.standard_normal((a.trees, obs_dim))).astype(np.float32)
        gaps = np.zeros(sims + 2, np.int64)
        depth_of_leaf = np.zeros(sims + 2, np.int64)
        created = evaluated = 0
        same_parent_next = 0
        for i in range(a.trees):
            t = orc.Tree(cfg)
            t.seed(i)
            t.run_mlp(w, obs[i], train=True)
            d = t.dump()
            n = d["n_nodes"]
            cb = d["child_base"][:n]
            nodes = np.nonzero(cb[1:] > 0)[0] + 1                      # expanded nodes other than the root
            sim = (cb[nodes] - 1 - A) // K                             # the simulation whose leaf the node was
            made = np.where(nodes <= A, -1, (nodes - 1 - A) // K)      # the simulation whose expansion created it
            g = sim - made
            np.add.at(gaps, g, 1)
            created += n - 1
            evaluated += len(nodes)
        tot = gaps.sum()
        cum = np.cumsum(gaps)
        med = int(np.searchsorted(cum, tot / 2.0))
        out["workloads"][name] = {
            "actions": A, "children_per_expansion": K, "simulations": sims,
            "leaf_selections": int(tot),
            "gap_histogram_rounds_between_creation_and_selection": {str(k): int(v) for k, v in enumerate(gaps) if v},
            "share_gap_1": gaps[1] / tot, "share_gap_2": gaps[2] / tot, "share_gap_ge_3": gaps[3:].sum() / tot,
            "median_gap": med, "mean_gap": float((np.arange(len(gaps)) * gaps).sum() / tot),
            "children_created_per_tree": created / a.trees, "children_evaluated_per_tree": evaluated / a.trees,
            "share_of_children_never_evaluated": 1.0 - evaluated / created,
            "speculative_rows_per_simulation": created / a.trees / sims,
        }
        print(name, json.dumps({k: v for k, v in out["workloads"][name].items() if not k.startswith("gap_hist")}), file=sys.stderr)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
