"""Helpers shared by the parity tests: golden fixtures as per-case dicts (test infrastructure)."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SEARCH_FIXTURES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz"))
                         if not os.path.basename(p).startswith(("weights_", "selfplay", "temperature_", "visionnet_", "mlpnet_", "reanalyse",
                                                                 "game_")))
SELFPLAY_FIXTURES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "selfplay*.npz")))
TEMPERATURES = (0.0, 0.2, 0.5, 1.0)


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = {k[4:]: z[k].item() for k in z.files if k.startswith("cfg_")}
    data = {k: z[k] for k in z.files if not k.startswith("cfg_")}
    return cfg, data


def cases(name):
    cfg, data = load(name)
    n = data["seed"].shape[0]
    return cfg, [{k: v[i] for k, v in data.items()} for i in range(n)]


def dims(cfg, case):
    A = int(case["root_policy"].shape[-1])
    K = min(int(cfg["maxium_action_sample"]), A)
    S = int(case["root_hidden"].shape[-1])
    return A, K, S, int(cfg["num_simulations"])
