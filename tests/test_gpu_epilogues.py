"""Head epilogue kernels against plain torch float32 restatements of muzero_model.py:575-591, 837-839 and
neural_network_mlp_model.py:349-357 (floating point: tolerance written per check)."""
import ctypes as C

import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


def _ref_decode(logits):
    S = logits.shape[1]
    p = torch.softmax(logits, 1)
    half = S // 2
    sup = torch.arange(-half, -half + S, device=logits.device, dtype=logits.dtype)
    y = (sup * p).sum(1)
    return torch.sign(y) * (((torch.sqrt(1 + 4 * 0.001 * (torch.abs(y) + 1 + 0.001)) - 1) / (2 * 0.001)) ** 2 - 1)


def _ref_scale(x):
    mn = x.min(1, keepdim=True)[0]; mx = x.max(1, keepdim=True)[0]
    sc = mx - mn
    sc[sc < 1e-5] += 1e-5
    return (x - mn) / sc


def _within_the_floor(out, exact64):
    steps = gu.decode_steps(out.cpu().numpy(), exact64.cpu().numpy())
    # random logits: the bound of the formula's own float32 roundings (on the reference's recorded logits the device is held to
    # the reference's own observed distance, tests/test_gpu_decode_floor.py)
    assert steps.max() <= gu.DECODE_BOUND_STEPS, f"{steps.max():.3f} stairs from the exact value of the formula"


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr())


@pytest.mark.parametrize("B,S", [(1, 31), (4096, 31), (1000, 30), (777, 61), (64, 5)])
def test_support_decode(B, S):
    import stochastic_muzero_amd as smz
    lib = smz._lib.load()
    g = torch.Generator(device="cpu"); g.manual_seed(B + S)
    logits = (torch.randn(B, S, generator=g) * 3).cuda()
    out = torch.empty(B, device="cuda")
    smz._lib.check(lib.smz_support_decode(_p(logits), S, _p(out), B, _s()))
    # the float32 formula is a staircase in the support expectation (golden_util.DECODE_STEP): any float32 evaluation is within
    # about one stair of the exact value; the reference's own float32 results were seen up to 0.752 stairs away
    _within_the_floor(out, _ref_decode(logits.double()))
    assert gu.decode_steps(out.cpu().numpy(), _ref_decode(logits).cpu().numpy()).max() <= 2 * gu.DECODE_BOUND_STEPS   # torch-ROCm float32


@pytest.mark.parametrize("B,A", [(4096, 2), (513, 4), (100, 18)])
def test_policy_softmax(B, A):
    import stochastic_muzero_amd as smz
    lib = smz._lib.load()
    g = torch.Generator(device="cpu"); g.manual_seed(B + A)
    logits = (torch.randn(B, A, generator=g) * 4).cuda()
    out = torch.empty(B, A, device="cuda")
    smz._lib.check(lib.smz_policy_softmax(_p(logits), A, _p(out), B, _s()))
    torch.testing.assert_close(out, torch.softmax(logits, -1), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("B,A,S", [(4096, 2, 31), (300, 4, 16), (65, 11, 7)])
def test_dynamics_and_prediction_epilogues(B, A, S):
    import stochastic_muzero_amd as smz
    lib = smz._lib.load()
    g = torch.Generator(device="cpu"); g.manual_seed(B)
    sd, sa, rl = (torch.randn(B, S, generator=g).cuda() for _ in range(3))
    sd[3] = 0.25                                        # constant row: range < 1e-5 -> +1e-5 rule
    br = (torch.rand(B, generator=g) < 0.5).to(torch.uint8).cuda()
    hid = torch.empty(B, S, device="cuda"); rw = torch.empty(B, device="cuda")
    smz._lib.check(lib.smz_dynamics_epilogue(_p(sd), _p(sa), _p(rl), S, _p(br), S, _p(hid), _p(rw), B, _s()))
    m = br.bool()
    ref_h = torch.where(m[:, None], _ref_scale(sd.clone()), _ref_scale(sa.clone()))
    torch.testing.assert_close(hid, ref_h, rtol=1e-6, atol=1e-6)
    _within_the_floor(rw, torch.where(m, _ref_decode(rl.double()), torch.zeros_like(rw, dtype=torch.float64)))
    assert (rw[~m] == 0).all()
    # prediction epilogue: the four logit blocks are column slices of one [B, 2A+2S] GEMM output (row stride ld)
    ld = 2 * A + 2 * S
    q = (torch.randn(B, ld, generator=g) * 2).cuda()
    pp, vp, pa, va = q[:, :A], q[:, A:A + S], q[:, A + S:2 * A + S], q[:, 2 * A + S:]
    fs = q.element_size()
    ptr = lambda off: C.c_void_p(q.data_ptr() + off * fs)
    pol = torch.empty(B, A, device="cuda"); val = torch.empty(B, device="cuda")
    smz._lib.check(lib.smz_prediction_epilogue(ptr(0), ptr(A), ptr(A + S), ptr(2 * A + S), ld, _p(br), A, S, _p(pol), _p(val), B, _s()))
    torch.testing.assert_close(pol, torch.where(m[:, None], torch.softmax(pp, -1), torch.softmax(pa, -1)), rtol=1e-5, atol=1e-6)
    _within_the_floor(val, torch.where(m, _ref_decode(vp.double()), _ref_decode(va.double())))


def test_cartpole_step_and_traj_pack():
    import orc
    import stochastic_muzero_amd as smz
    lib = smz._lib.load()
    B, A, T = 300, 2, 3
    rs = np.random.RandomState(0)
    st = rs.uniform(-0.05, 0.05, (B, 4))
    act = rs.randint(0, 2, B).astype(np.int32)
    d_st = torch.from_numpy(st.copy()).cuda(); d_act = torch.from_numpy(act).cuda()
    obs = torch.empty(B, 4, device="cuda"); rw = torch.empty(B, device="cuda"); term = torch.empty(B, dtype=torch.uint8, device="cuda")
    ref = st.copy()
    for step in range(5):
        smz._lib.check(lib.smz_cartpole_step(_p(d_st), _p(d_act), _p(obs), _p(rw), _p(term), B, _s()))
        for i in range(B):
            orc.lib().orc_cartpole_step(ref[i].ctypes.data_as(C.c_void_p), int(act[i]))
    # device sin/cos are not glibc's: float64 state to 1e-12, observations to float32 resolution
    np.testing.assert_allclose(d_st.cpu().numpy(), ref, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(obs.cpu().numpy(), ref.astype(np.float32), rtol=1e-6, atol=1e-9)
    assert (rw == 1).all() and (term == 0).all()
    F = lib.smz_traj_floats(4, A)
    assert F == 4 + 3 * A + 3
    traj = torch.zeros(T, B, F, dtype=torch.float64, device="cuda")
    pol = torch.rand(B, A, dtype=torch.float64, device="cuda"); cv = torch.rand(B, A, dtype=torch.float64, device="cuda")
    rv = torch.rand(B, device="cuda")
    smz._lib.check(lib.smz_traj_pack(_p(traj), T, 1, 4, A, _p(obs), _p(rw), _p(term), _p(d_act), _p(pol), _p(cv), _p(rv), B, _s()))
    t = traj.cpu().numpy()
    assert (t[0] == 0).all() and (t[2] == 0).all()
    assert np.array_equal(t[1][:, :4], obs.cpu().numpy().astype(np.float64))
    assert np.array_equal(t[1][:, 4], rw.cpu().numpy().astype(np.float64))
    assert (t[1][:, 5] == 0).all()
    assert np.array_equal(t[1][:, 6:6 + A], pol.cpu().numpy())
    assert np.array_equal(t[1][:, 6 + A:6 + 2 * A], np.eye(A)[act])
    assert np.array_equal(t[1][:, 6 + 2 * A], rv.cpu().numpy().astype(np.float64))
    assert np.array_equal(t[1][:, 7 + 2 * A:], cv.cpu().numpy())
