"""Helpers shared by the parity tests: golden fixtures as per-case dicts (test infrastructure)."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SEARCH_FIXTURES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz"))
                         if not os.path.basename(p).startswith(("weights_", "selfplay", "temperature_", "visionnet_", "mlpnet_", "reanalyse",
                                                                 "game_", "decode_floor")))
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


# ---- the staircase of the reference's value / reward decode (muzero_model.py:575-591) ----------------------------------
# value = r**2 - 1 with r = (sqrt(1 + 4 eps (|y| + 1 + eps)) - 1) / (2 eps), eps = 0.001, all float32.  The square root's
# argument and result lie in [1, 2): one float32 ulp there is 2**-23, and `- 1` exposes it -- as a function of the support
# expectation y the result is a staircase whose step is d value = 2 r / (2 eps) * 2**-23 = DECODE_STEP * sqrt(|value| + 1).
# The reference's own float32 result sits up to DECODE_FLOOR_STEPS steps from the exact value of its formula on the same
# logits (tests/golden/decode_floor_*.npz, written by oracle/gen_golden_r4.py from the reference itself: 0.752 steps =
# 3.6e-5 relative on checkpoint 421); another float32 evaluation order of the softmax / expectation lands on a neighbouring
# stair, so two correct float32 decodes differ by up to 2 * DECODE_FLOOR_STEPS steps.
DECODE_STEP = 2.0 ** -23 / 0.001
DECODE_FLOOR_STEPS = 0.76        # observed: the reference's own float32 decodes on the fixtures' logits (max 0.752)
# What ANY float32 evaluation of the formula can be from exact: `1 + x` and the square root each round to half an ulp of
# [1, 2) -- together one stair -- plus the relative roundings of the division, the square and the final subtraction.
DECODE_BOUND_STEPS = 1.05


def decode_steps(value, reference):
    """|value - reference| in units of the decode's staircase step at `reference`."""
    value, reference = np.asarray(value, np.float64), np.asarray(reference, np.float64)
    return np.abs(value - reference) / (DECODE_STEP * np.sqrt(np.abs(reference) + 1.0))


def assert_decoded_like_the_reference(value, tape, what="value", max_steps=1.05, min_identical=0.6):
    """A decoded value / reward of a whole head evaluation against the reference's recorded float32 one: at most ONE stair apart
    (measured on MI355X: 1.012 stairs, profiles/r04_head_errors.json; the reference's own distance from the exact value of its
    formula is 0.752 stairs).  How many land on the reference's very stair depends on how far the logits are apart: from the
    SAME logits 99.3 % are bit-identical (tests/test_gpu_decode_floor.py); through the head's own matrix products (another
    summation order than ATen's: logits differ in the 7th digit, the support expectation -- a sum weighted with -15 .. 15 --
    by ~1e-5, a third of a stair's width) 76 % on checkpoint 421."""
    value, tape = np.asarray(value, np.float32).reshape(-1), np.asarray(tape, np.float32).reshape(-1)
    steps = decode_steps(value, tape)
    assert steps.max() <= max_steps, f"{what}: {steps.max():.3f} stairs from the reference's decode (row {int(steps.argmax())})"
    near = steps <= 0.01          # same stair (the hidden state that entered the head differs in the 7th digit)
    assert near.mean() >= min_identical, f"{what}: only {100 * near.mean():.1f} % on the reference's stair"
