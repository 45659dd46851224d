"""The DEVICE build of csrc/smz_glibc_math.hpp (v_fma_f64 / v_mul_f64 / v_add_f64 on gfx950) against the host's libm, bit for
bit, through the C ABI's inspection entry smz_debug_glibc_log_pow -- the GPU half of tests/test_glibc_math.py.  What rides on
it: the float64 root priors after device-drawn Dirichlet noise are the reference's exactly (tests/gpu_harness.py,
test_gpu_fullsize_parity.py hold them with rtol 0 since round 6).  monte_carlo_tree_search.py:214-225."""
import ctypes as C

import numpy as np
import pytest
import torch

import stochastic_muzero_amd as smz
from test_glibc_math import _same_libm_as_restated, glc

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not _same_libm_as_restated(), reason="the host's libm is not glibc 2.35's FMA build (the code restated)")]


def P(t):
    return C.c_void_p(t.data_ptr())


def device_log_pow(x, y=None):
    lib = smz._lib.load()
    dx = torch.from_numpy(np.ascontiguousarray(x, np.float64)).cuda()
    ol = torch.empty_like(dx)
    if y is None:
        assert lib.smz_debug_glibc_log_pow(P(dx), None, dx.numel(), P(ol), None, None) == 0
        torch.cuda.synchronize()
        return ol.cpu().numpy(), None
    dy = torch.from_numpy(np.array(np.broadcast_to(y, np.shape(x)), np.float64)).cuda()      # (a writable copy)
    op = torch.empty_like(dx)
    assert lib.smz_debug_glibc_log_pow(P(dx), P(dy), dx.numel(), P(ol), P(op), None) == 0
    torch.cuda.synchronize()
    return ol.cpu().numpy(), op.cpu().numpy()


def libm_log(x):
    x = np.ascontiguousarray(x, np.float64); out = np.empty_like(x)
    glc().glc_libm_log_array(x.ctypes.data, out.ctypes.data, x.size)
    return out


def libm_pow(x, y):
    x = np.ascontiguousarray(x, np.float64); y = np.ascontiguousarray(np.broadcast_to(y, x.shape), np.float64); out = np.empty_like(x)
    glc().glc_libm_pow_array(x.ctypes.data, y.ctypes.data, out.ctypes.data, x.size)
    return out


def same(a, b):
    return np.array_equal(a.view(np.uint64), b.view(np.uint64)) or np.array_equal(a, b, equal_nan=True)


def test_device_log_is_the_hosts_libm_log():
    r = np.random.RandomState(1)
    n = 4_000_000
    u = r.random_sample(n)
    shape = r.choice([0.25, 0.3, 0.03, 0.5, 1 / 3, 0.9], n)
    bits = r.randint(0, 2 ** 63 - 1, n, dtype=np.int64).astype(np.uint64) & np.uint64(0x7fefffffffffffff)
    for name, x in (("1 - U", 1.0 - u), ("(1 - U) / shape", (1.0 - u) / shape), ("any positive double", bits.view(np.float64)),
                    ("around 1", 1.0 + (u - 0.5) * 0.15)):
        got, _ = device_log_pow(x)
        want = libm_log(x)
        bad = np.flatnonzero(got.view(np.uint64) != want.view(np.uint64))
        assert bad.size == 0, (name, bad.size, x[bad[:3]], got[bad[:3]], want[bad[:3]])
    edge = np.array([1.0, 0.0, -0.0, np.inf, 5e-324, 2.2250738585072014e-308, 0.9375, 1.064697265625])
    assert same(device_log_pow(edge)[0], libm_log(edge))
    assert np.isnan(device_log_pow(np.array([-1.0, np.nan]))[0]).all()


def test_device_pow_is_the_hosts_libm_pow():
    r = np.random.RandomState(2)
    n = 4_000_000
    u = r.random_sample(n)
    shape = r.choice([0.25, 0.3, 0.03, 0.5, 1 / 3, 0.9, 0.01], n)
    # the sampler's two call sites
    x0 = u * (1.0 - shape)
    U1 = 1.0 - shape + r.random_sample(n) * shape
    Y = -libm_log((1.0 - U1) / shape)
    x1 = 1.0 - shape + shape * Y
    bits = r.randint(0, 2 ** 63 - 1, n, dtype=np.int64).astype(np.uint64) & np.uint64(0x7fefffffffffffff)
    yw = np.ldexp(1.0 + r.random_sample(n), r.randint(-70, 71, n))
    xa = bits.view(np.float64)
    # results around the under- / overflow thresholds
    with np.errstate(all="ignore"):
        lx = np.log(xa)
        yt = np.where(lx != 0, (690.0 + 70.0 * r.random_sample(n)) * np.sign(r.random_sample(n) - 0.5) / lx, 1.0)
        yt = np.where((yt > 0) & (yt < 1e300), yt, 3.5)
    for name, x, y in (("U ^ (1 / shape)", x0, 1.0 / shape), ("(1 - s + s Y) ^ (1 / s)", x1, 1.0 / shape), ("wide", xa, yw),
                       ("thresholds", xa, yt), ("p ^ (1 / T)", u, 1.0 / (0.3 + 0.7 * r.random_sample(n)))):
        _, got = device_log_pow(x, y)
        want = libm_pow(x, y)
        bad = np.flatnonzero(got.view(np.uint64) != want.view(np.uint64))
        assert bad.size == 0, (name, bad.size, x[bad[:3]], np.broadcast_to(y, x.shape)[bad[:3]], got[bad[:3]], want[bad[:3]])
    assert np.isnan(device_log_pow(np.array([-2.0, 2.0, 2.0]), np.array([2.0, -1.0, 0.0]))[1]).all()
