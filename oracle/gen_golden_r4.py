#!/usr/bin/env python3
"""Round-4 golden vectors, written by the REFERENCE itself (imported from /root/reference in this container; never shipped).

TEST INFRASTRUCTURE: only tests/ and tools/decode_floor_report.py read what this writes.

  decode_floor_ckpt421.npz   every call the reference makes to Muzero.inverse_transform_with_support
  decode_floor_lunar.npz     (muzero_model.py:575-591) while it runs the searches of the committed fixtures
  decode_floor_vision.npz    ckpt421_sims50 / ckpt421_sims100 (checkpoint 421), lunar_K2_sims50 (random-init MLP, values near
                             zero) and vision_sims50: the float32 LOGITS that enter the transform, the reference's own float32
                             result, and the reference's own formula evaluated on the same logits in float64
                             (the same method called with `logits.double()`).  |f32 - f64| is the noise floor of the
                             reference's decode: what any other float32 evaluation order of softmax / sum may differ by.
Run:  python oracle/gen_golden_r4.py
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
BASE = dict(pb_c_base=19652, pb_c_init=1.25, discount=0.999, root_dirichlet_alpha=0.25,
            root_exploration_fraction=0.1, maxium_action_sample=2, number_of_player=1, custom_loop=None)


class DecodeTap:
    """Wraps model.inverse_transform_with_support: records (logits, float32 result) of every call and evaluates the
    reference's own method on the same logits in float64."""

    def __init__(self, mz):
        self.mz, self.inner = mz, mz.inverse_transform_with_support
        self.logits, self.f32, self.f64 = [], [], []
        mz.inverse_transform_with_support = self

    def __call__(self, x):
        y = self.inner(x)
        with torch.no_grad():
            self.logits.append(x.detach().to(torch.float32).cpu().numpy().copy())
            self.f32.append(y.detach().to(torch.float32).cpu().numpy().reshape(-1).copy())
            self.f64.append(self.inner(x.detach().double()).cpu().numpy().reshape(-1).copy())
        return y

    def save(self, name, **meta):
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, logits=np.concatenate(self.logits).astype(np.float32),
                            ref_f32=np.concatenate(self.f32).astype(np.float32),
                            ref_f64=np.concatenate(self.f64).astype(np.float64),
                            **{k: np.asarray(v) for k, v in meta.items()})
        n = sum(len(a) for a in self.f32)
        print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB, {n} decodes)")
        self.mz.inverse_transform_with_support = self.inner


def main():
    ref = R.import_reference()
    torch.set_num_threads(1)
    anchor = torch.tensor([[0.01, -0.02, 0.03, 0.04]])

    mz = G.load_ckpt(ref, 421)
    tap = DecodeTap(mz)
    for sims, nseeds in ((50, 16), (100, 8)):            # the searches of ckpt421_sims50 / ckpt421_sims100 (gen_golden.main)
        kw = dict(BASE, num_simulations=sims)
        for seed in range(nseeds):
            obs = anchor if seed < 2 else torch.tensor(
                np.random.RandomState(1000 + seed).uniform(-0.05, 0.05, (1, 4)).astype(np.float32))
            G.run_case(ref, mz, obs, seed, kw)
    tap.save("decode_floor_ckpt421", fixtures="ckpt421_sims50,ckpt421_sims100")

    ll = G.fresh_mlp(ref, 8, 4, L=0, seed=0)             # lunar_K2_sims50
    tap = DecodeTap(ll)
    kw = dict(BASE, num_simulations=50, maxium_action_sample=2)
    for s in range(12):
        G.run_case(ref, ll, torch.tensor(np.random.RandomState(2000 + s).randn(1, 8).astype(np.float32)), s, kw)
    tap.save("decode_floor_lunar", fixtures="lunar_K2_sims50")

    vz = G.fresh_vision(ref, A=2, L=1, seed=0)           # vision_sims50
    tap = DecodeTap(vz)
    kw = dict(BASE, num_simulations=50)
    for s in range(4):
        G.run_case(ref, vz, torch.tensor(np.random.RandomState(3000 + s).rand(1, 3, 98, 98).astype(np.float32)), s, kw,
                   obs_dim=4)
    tap.save("decode_floor_vision", fixtures="vision_sims50")


if __name__ == "__main__":
    main()
