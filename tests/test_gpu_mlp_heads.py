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
        self.mlp_input, self.branch = x, branch


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
