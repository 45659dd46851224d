#!/usr/bin/env python3
"""Round-3 golden vectors, written by the REFERENCE itself (imported from /root/reference in this container; never shipped).

TEST INFRASTRUCTURE: only tests/ read what this writes.

  weights_ckpt450.npz      the reference's shipped checkpoint 450 (config/experiment_450_config.json:18-20:
  ckpt450_sims11.npz       state_space_dimensions 61, hidden_layer_dimensions 126, number_of_hidden_layer 4 -- the deep
                           shape among the reference's configs; neural_network_mlp_model.py:5-250), exported as plain arrays,
                           and 8 searches of 11 simulations (the config's own num_simulations is 0; 11 is what the other shipped
                           configs use) with the reference's Monte_carlo_tree_search.run on that model: network-output tape,
                           per-simulation leaves, final trees, post-search policy / action.
Run:  python oracle/gen_golden_r3.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as R   # noqa: E402
import gen_golden as G    # noqa: E402

OUT = G.OUT
KW = dict(pb_c_base=19652, pb_c_init=1.25, discount=0.997, root_dirichlet_alpha=0.25, root_exploration_fraction=0.25,
          num_simulations=11, maxium_action_sample=2, number_of_player=1, custom_loop=None)   # experiment_450_config.json's search block


def main():
    ref = R.import_reference()
    torch.set_num_threads(1)
    mz = G.load_ckpt(ref, 450)
    G.export_mlp_weights(mz, os.path.join(OUT, "weights_ckpt450.npz"))
    cases = [G.run_case(ref, mz, torch.tensor(np.random.RandomState(4500 + s).uniform(-0.05, 0.05, (1, 4)).astype(np.float32)), s, KW)
             for s in range(8)]
    G.save("ckpt450_sims11", {k: v for k, v in KW.items() if v is not None}, cases)


if __name__ == "__main__":
    main()
