"""The noise floor of the reference's value / reward decode (inverse_transform_with_support, muzero_model.py:575-591), as
the reference itself measures it: tests/golden/decode_floor_*.npz hold the float32 logits of every decode the reference made
while it produced the search fixtures, its own float32 result, and its own formula on the same logits in float64
(oracle/gen_golden_r4.py).  CPU tests: the fixtures' floor is what golden_util states, and the oracle's decode sits inside
it.  The GPU counterpart is tests/test_gpu_decode_floor.py."""
import numpy as np
import pytest

import golden_util as gu

FLOORS = ("decode_floor_ckpt421", "decode_floor_lunar", "decode_floor_vision")


def _load(name):
    z = np.load(f"{gu.GOLDEN}/{name}.npz")
    return z["logits"], z["ref_f32"], z["ref_f64"]


@pytest.mark.parametrize("name", FLOORS)
def test_the_references_own_float32_decode_is_up_to_three_quarters_of_a_stair_from_exact(name):
    logits, f32, f64 = _load(name)
    steps = gu.decode_steps(f32, f64)
    assert steps.max() <= gu.DECODE_FLOOR_STEPS
    if name == "decode_floor_ckpt421":
        # the statement the tolerance of the parity tests rests on: the reference's float32 output is itself further than
        # north_star's 1e-5 from the exact value of its own formula (values 3 .. 111 here)
        rel = np.abs(f32.astype(np.float64) - f64) / np.abs(f64)
        assert rel.max() > 3e-5 and rel.mean() > 0.5e-5
        assert steps.max() > 0.7


@pytest.mark.parametrize("name", FLOORS)
def test_exact_value_of_the_formula_is_what_the_fixture_says(name):
    """ref_f64 is the reference's method called with float64 logits; a numpy restatement of the formula agrees to 1e-12."""
    logits, _, f64 = _load(name)
    L = logits.astype(np.float64)
    p = np.exp(L - L.max(1, keepdims=True))
    p /= p.sum(1, keepdims=True)
    S = L.shape[1]
    y = (p * (np.arange(S) - S // 2)).sum(1)
    v = np.sign(y) * (((np.sqrt(1 + 4 * 0.001 * (np.abs(y) + 1 + 0.001)) - 1) / (2 * 0.001)) ** 2 - 1)
    np.testing.assert_allclose(v, f64, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("name", FLOORS)
def test_oracle_decode_is_within_the_floor_of_exact_and_two_floors_of_the_reference(name):
    import orc
    logits, f32, f64 = _load(name)
    out = np.array([orc.lib().orc_support_decode(np.ascontiguousarray(row).ctypes.data, logits.shape[1]) for row in logits],
                   np.float32)
    assert gu.decode_steps(out, f64).max() <= gu.DECODE_FLOOR_STEPS
    assert gu.decode_steps(out, f32).max() <= 2 * gu.DECODE_FLOOR_STEPS
    # and most decodes are bit-identical to the reference's: same formula, same float32 operations after the expectation
    assert (out == f32).mean() > 0.9
