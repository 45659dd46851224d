"""The device's value / reward decode against the reference's own decodes (tests/golden/decode_floor_*.npz: float32 logits,
the reference's float32 result, the reference's formula in float64 on the same logits -- oracle/gen_golden_r4.py).

north_star asks for 1e-5 on backed-up values.  The reference's float32 decode is itself up to 0.752 stairs = 3.6e-5 relative
away from the exact value of its formula (tests/test_decode_floor.py), so the contract that can be held, and is held here, is:
the device decode is no further from EXACT than the reference's own float32 result is (<= DECODE_FLOOR_STEPS stairs), hence
at most two floors from the reference's; and it is the reference's formula bit for bit wherever the float32 support expectation
agrees (the large majority of decodes)."""
import ctypes as C

import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu

FLOORS = ("decode_floor_ckpt421", "decode_floor_lunar", "decode_floor_vision")


@pytest.mark.parametrize("name", FLOORS)
def test_device_decode_is_as_close_to_exact_as_the_references_own_float32_decode(name):
    import stochastic_muzero_amd as smz
    lib = smz._lib.load()
    z = np.load(f"{gu.GOLDEN}/{name}.npz")
    logits = torch.from_numpy(z["logits"]).cuda().contiguous()
    B, S = logits.shape
    out = torch.empty(B, device="cuda")
    smz._lib.check(lib.smz_support_decode(C.c_void_p(logits.data_ptr()), S, C.c_void_p(out.data_ptr()), B,
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    to_exact, to_ref = gu.decode_steps(out, z["ref_f64"]), gu.decode_steps(out, z["ref_f32"])
    ref_to_exact = gu.decode_steps(z["ref_f32"], z["ref_f64"])
    print(f"{name}: device-exact {to_exact.max():.3f} stairs, reference-exact {ref_to_exact.max():.3f}, device-reference "
          f"{to_ref.max():.3f}; bit-identical to the reference {100 * (out == z['ref_f32']).mean():.1f} %")
    assert to_exact.max() <= gu.DECODE_FLOOR_STEPS
    assert to_ref.max() <= 2 * gu.DECODE_FLOOR_STEPS
    assert (out == z["ref_f32"]).mean() > 0.9
