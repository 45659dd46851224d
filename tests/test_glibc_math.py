"""CPU pin of csrc/smz_glibc_math.hpp -- the product's operation-by-operation restatement of glibc's log() / pow(), which the
device uses inside the Dirichlet root noise (numpy's legacy gamma sampler calls libm's; monte_carlo_tree_search.py:220).  The same
source is compiled by gcc into oracle/libglibccheck.so (test infrastructure) and compared with THIS machine's libm bit for bit:
> 10^7 arguments of each of the sampler's call sites, wide nets over all positive doubles, the near-1 interval's ends, the
under- / overflow special cases -- and against Python's math.log / math.pow, and numpy's own legacy sampler end to end.
VERDICT r5 next #3."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_L = None


def _same_libm_as_restated():
    """The restatement follows the FMA build of glibc 2.35's log / pow (what this image's libm resolves to on a CPU with FMA + AVX2).
    On another glibc, or a CPU without FMA (libm then resolves to its SSE2 build: other fusions, other last bits), THIS machine's
    libm is not the thing restated and cannot serve as the checker: the comparisons are skipped, not failed."""
    import platform
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        flags = ""
    return platform.libc_ver()[1].startswith("2.35") and " fma " in flags and " avx2 " in flags


pytestmark = pytest.mark.skipif(not _same_libm_as_restated(), reason="this machine's libm is not glibc 2.35's FMA build (the code restated)")


def glc():
    global _L
    if _L is None:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "libglibccheck.so"])
        L = C.CDLL(os.path.join(ROOT, "oracle", "libglibccheck.so"))
        for n in ("glc_check_log", "glc_check_pow"):
            getattr(L, n).restype = C.c_int64
            getattr(L, n).argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_void_p]
        L.glc_log_array.argtypes = L.glc_libm_log_array.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.glc_pow_array.argtypes = L.glc_libm_pow_array.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        _L = L
    return _L


def log_arr(x):
    x = np.ascontiguousarray(x, np.float64); out = np.empty_like(x)
    glc().glc_log_array(x.ctypes.data, out.ctypes.data, x.size)
    return out


def pow_arr(x, y):
    x = np.ascontiguousarray(x, np.float64); y = np.ascontiguousarray(np.broadcast_to(y, x.shape), np.float64); out = np.empty_like(x)
    glc().glc_pow_array(x.ctypes.data, y.ctypes.data, out.ctypes.data, x.size)
    return out


@pytest.mark.parametrize("mode,n", [(0, 12_000_000), (1, 12_000_000), (2, 4_000_000), (3, 4_000_000), (4, 4_000_000)])
def test_log_equals_libm_bit_for_bit(mode, n):
    """mode 0: 1 - U (legacy_standard_exponential); 1: (1 - U) / shape; 2: any finite non-negative double (subnormals included);
    3, 4: around 1 and across the ends of the near-1 interval."""
    bad = (C.c_double * 2)()
    wrong = glc().glc_check_log(1000 + mode, n, mode, bad)
    assert wrong == 0, f"{wrong} of {n} differ, first x = {bad[0].hex()}"


@pytest.mark.parametrize("mode,n", [(0, 12_000_000), (1, 12_000_000), (2, 4_000_000), (3, 4_000_000), (4, 4_000_000)])
def test_pow_equals_libm_bit_for_bit(mode, n):
    """mode 0: U ^ (1 / shape), U <= 1 - shape; 1: (1 - shape + shape Y) ^ (1 / shape); 2: any positive finite x, y over
    2^-70 .. 2^70; 3: results around the under- / overflow thresholds (exp's special case, subnormal results); 4: p ^ (1 / T)."""
    bad = (C.c_double * 2)()
    wrong = glc().glc_check_pow(2000 + mode, n, mode, bad)
    assert wrong == 0, f"{wrong} of {n} differ, first (x, y) = ({bad[0].hex()}, {bad[1].hex()})"


def test_against_pythons_math_module_and_the_edge_values():
    r = np.random.RandomState(5)
    x = np.concatenate([1.0 - r.random_sample(20000), r.random_sample(20000) * 4, [1.0, 0.9375, 1.064697265625, 5e-324, 2.2250738585072014e-308,
                                                                              1.7976931348623157e308, 0.5, 2.0]])
    assert np.array_equal(log_arr(x), np.array([math.log(v) for v in x]))
    for v, want in ((0.0, -np.inf), (-0.0, -np.inf), (np.inf, np.inf)):
        assert log_arr([v])[0] == want
    assert np.isnan(log_arr([-1.0, np.nan])).all()
    u = r.random_sample(20000)
    for shape in (0.25, 0.3, 0.03, 1.0, 1 / 3):
        assert np.array_equal(pow_arr(u, 1.0 / shape), np.array([math.pow(v, 1.0 / shape) for v in u])), shape
    # special cases inside the documented domain
    cases = [(0.0, 4.0), (1.0, 1e300), (0.5, 1e300), (2.0, 1e300), (np.inf, 2.0), (0.5, 1e-300), (2.0, 1e-300), (1e-310, 0.5), (1e-310, 3.0),
             (0.999, 7e5), (1.001, 7e5), (0.999, 7.3e5), (0.3, 617.0), (0.3, 618.5), (1.5, 1750.0), (1.5, 1751.0), (3.0, 1e-20)]
    for xv, yv in cases:
        try:
            want = math.pow(xv, yv)
        except OverflowError:
            want = np.inf
        got = pow_arr([xv], yv)[0]
        assert got == want and math.copysign(1, got) == math.copysign(1, want), (xv, yv, got, want)
    # ... and outside it the restatement is loud, not quietly different
    assert np.isnan(pow_arr([-2.0, 2.0, 2.0, np.nan], [2.0, -1.0, 0.0, 1.0])).all()


def test_a_python_transcription_of_numpys_legacy_dirichlet_on_these_routines_is_numpys_sample():
    """End to end on the CPU: legacy_standard_gamma + the Dirichlet normalisation (numpy/random/mtrand.pyx, legacy-distributions.c)
    driven by RandomState's own doubles, with log / pow from the restatement -> RandomState.dirichlet's sample bit for bit,
    including the stream position afterwards."""
    lg = lambda v: float(log_arr([v])[0])            # noqa: E731
    pw = lambda a, b: float(pow_arr([a], b)[0])      # noqa: E731

    def gamma(rs, shape):
        if shape == 1.0:
            return -lg(1.0 - rs.random_sample())
        while True:
            U = rs.random_sample()
            V = -lg(1.0 - rs.random_sample())
            if U <= 1.0 - shape:
                X = pw(U, 1.0 / shape)
                if X <= V:
                    return X
            else:
                Y = -lg((1.0 - U) / shape)
                X = pw(1.0 - shape + shape * Y, 1.0 / shape)
                if X <= V + Y:
                    return X
    for seed in range(400):
        for alpha, A in ((0.25, 2), (0.3, 4), (0.03, 3), (1.0, 2), (0.9, 6)):
            a, b = np.random.RandomState(seed), np.random.RandomState(seed)
            want = a.dirichlet([alpha] * A)
            g = [gamma(b, alpha) for _ in range(A)]
            acc = 0.0
            for v in g:
                acc = acc + v
            inv = 1.0 / acc
            got = np.array([v * inv for v in g])
            assert np.array_equal(got, want), (seed, alpha, A)
            assert a.random_sample() == b.random_sample()
