#!/usr/bin/env python3
"""Measured error of the batched heads against the reference's recorded head outputs (the tapes in tests/golden):
max abs / max rel error of hidden states, policies, decoded values and rewards, per fixture and backend.  Writes a JSON
report (default gpurun_out/head_errors.json); the committed copy under profiles/ is what the test tolerances follow."""
import json
import os
import sys
from importlib import import_module

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import golden_util as gu  # noqa: E402
import stochastic_muzero_amd  # noqa: E402,F401


class FE:
    pass


def err(a, b, decoded=False):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = np.abs(a - b)
    steps = gu.decode_steps(a, b) if decoded else None
    extra = dict(max_stairs=float(steps.max())) if decoded else {}
    return dict(**extra, max_abs=float(d.max()), max_rel=float((d / np.maximum(np.abs(b), 1e-30))[np.abs(b) > 1e-3].max()) if (np.abs(b) > 1e-3).any() else 0.0,
                mean_abs=float(d.mean()))


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "head_errors.json")
    model_mod = import_module("stochastic-muzero_amd.model")
    rep = {}
    for name, wname, loader in (("ckpt421_sims50", "weights_ckpt421", "arrays"), ("ckpt421_sims100", "weights_ckpt421", "arrays"),
                                ("lunar_K2_sims50", "weights_lunar_L0", "arrays"), ("lunarL2_K3_sims24", "weights_lunar_L2", "arrays"),
                                ("wideA11_K9_sims24", "weights_wide_A11", "arrays"), ("vision_sims50", "visionnet_L1_seed0", "sd")):
        cfg, data = gu.load(name)
        model = (model_mod.Muzero.from_arrays if loader == "arrays" else model_mod.Muzero.from_state_dicts)(os.path.join(gu.GOLDEN, wname + ".npz"))
        ncase, sims = data["tape_branch"].shape
        A = data["root_policy"].shape[-1]
        for backend in ("hip", "torch"):
            heads = model.heads("cuda:0", backend=backend)
            if loader == "sd":      # the module path learns the hidden-state shape from a representation call
                frames = np.stack([np.random.RandomState(3000 + int(s)).rand(3, 98, 98).astype(np.float32) for s in data["seed"]])
                heads.initial(torch.from_numpy(frames).cuda())
            fe = FE()
            hin = torch.from_numpy(data["tape_hidden_in"].reshape(ncase * sims, -1))
            fe.B, fe.S = ncase * sims, hin.shape[1]
            fe.parent_hidden = hin.cuda().contiguous()
            fe.last_action = torch.from_numpy(data["tape_action"].reshape(-1).astype(np.int32)).cuda()
            fe.mlp_input = torch.cat([hin, torch.eye(A)[fe.last_action.cpu().long()]], 1).cuda().contiguous()
            fe.branch = torch.from_numpy(data["tape_branch"].reshape(-1).astype(np.uint8)).cuda()
            h2, rw, p2, v2 = heads.recurrent(fe)
            torch.cuda.synchronize()
            dyn = data["tape_branch"].reshape(-1) == 1
            r = dict(hidden=err(h2.cpu().numpy(), data["tape_hidden_out"].reshape(ncase * sims, -1)),
                     policy=err(p2.cpu().numpy(), data["tape_policy"].reshape(ncase * sims, -1)),
                     value=err(v2.cpu().numpy(), data["tape_value"].reshape(-1), decoded=True),
                     reward=err(rw.cpu().numpy()[dyn], data["tape_reward"].reshape(-1)[dyn], decoded=True) if dyn.any() else None,
                     evaluations=int(ncase * sims), heads=type(heads).__name__)
            rep[f"{name}/{backend}"] = r
            print(name, backend, "hidden %.2e policy %.2e value abs %.2e rel %.2e stairs %.3f reward rel %.2e stairs %.3f" % (
                r["hidden"]["max_abs"], r["policy"]["max_abs"], r["value"]["max_abs"], r["value"]["max_rel"], r["value"]["max_stairs"],
                r["reward"]["max_rel"] if r["reward"] else 0.0, r["reward"]["max_stairs"] if r["reward"] else 0.0))
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(rep, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
