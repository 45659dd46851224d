"""The oracle's numpy-legacy RNG restatement against numpy.random.RandomState itself (the third-party dependency
the reference draws from: monte_carlo_tree_search.py:208,220,243,254,294; game.py:213)."""
import ctypes as C

import numpy as np
import pytest

import orc

SEEDS = [0, 1, 2, 12345, 2**31 - 1, 2**32 - 1]


def _tree(A=4, K=2, sims=4):
    return orc.Tree(orc.make_cfg(A, K, 3, sims))


@pytest.mark.parametrize("seed", SEEDS)
def test_random_sample_stream(seed):
    t = _tree(); t.seed(seed)
    rs = np.random.RandomState(seed)
    ours = np.array([t.random_sample() for _ in range(2000)])   # crosses three MT19937 regenerations
    assert np.array_equal(ours, rs.random_sample(2000))


def test_state_roundtrip_matches_numpy():
    rs = np.random.RandomState(7)
    rs.random_sample(1000)
    _, key, pos, *_ = rs.get_state()
    t = _tree(); t.set_rng(key, pos)
    assert [t.random_sample() for _ in range(700)] == list(rs.random_sample(700))
    key2, pos2 = t.get_rng()
    _, k, p, *_ = rs.get_state()
    assert pos2 == p and np.array_equal(key2, k)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 8, 9, 11, 16, 17, 31, 33, 64, 100, 128])
def test_numpy_sum_order(n):
    L = orc.lib()
    rs = np.random.RandomState(n)
    for _ in range(300):
        a = rs.rand(n).astype(np.float32)
        assert np.float32(L.orc_np_sum_f32(a.ctypes.data_as(C.c_void_p), n)) == a.sum()
        d = rs.rand(n)
        assert L.orc_np_sum_f64(d.ctypes.data_as(C.c_void_p), n) == d.sum()


def _policy(rs, A, peaked):
    x = rs.randn(A) * (6.0 if peaked else 1.0)
    p = np.exp(x - x.max()); p = (p / p.sum()).astype(np.float32)
    return p


@pytest.mark.parametrize("A,K", [(2, 2), (4, 2), (4, 4), (6, 3), (11, 9), (18, 5)])
@pytest.mark.parametrize("peaked", [False, True])
def test_root_expansion_draws_and_dirichlet(A, K, peaked):
    """root choice(A, A, p, replace=False) + dirichlet consume exactly numpy's draws and give numpy's noise."""
    for seed in range(40):
        rs = np.random.RandomState(seed)
        pol = _policy(np.random.RandomState(100 + seed), A, peaked)
        t = orc.Tree(orc.make_cfg(A, K, 0, 8, alpha=0.25, frac=0.25)); t.seed(seed)
        noise = t.root_init(pol, train=True)
        p = (pol + 1e-12); p = p / p.sum()
        picks = np.sort(rs.choice(A, A, p=p, replace=False))
        assert list(picks) == list(range(A))
        ref_noise = rs.dirichlet([0.25] * A)
        assert np.array_equal(noise, ref_noise)
        _, pri, _, _ = t.root_stats()
        ref_pri = np.array([p[a] * (1 - 0.25) + ref_noise[a] * 0.25 for a in range(A)])
        assert np.array_equal(pri, ref_pri)
        assert t.random_sample() == rs.random_sample()


@pytest.mark.parametrize("alpha", [0.03, 0.25, 0.5, 0.9, 1.0])
def test_dirichlet_alphas(alpha):
    for seed in range(60):
        rs = np.random.RandomState(seed)
        t = orc.Tree(orc.make_cfg(3, 2, 0, 4, alpha=alpha, frac=0.5)); t.seed(seed)
        pol = np.array([0.2, 0.3, 0.5], np.float32)
        noise = t.root_init(pol, train=True)
        p = (pol + 1e-12); p = p / p.sum()
        rs.choice(3, 3, p=p, replace=False)
        assert np.array_equal(noise, rs.dirichlet([alpha] * 3))
        assert t.random_sample() == rs.random_sample()


@pytest.mark.parametrize("A,K", [(4, 2), (6, 3), (11, 9), (18, 5)])
def test_leaf_expansion_sampled_subset(A, K):
    """choice(A, K, p, replace=False): sampled subset, sorted, un-renormalised priors, stream position."""
    for seed in range(60):
        rs = np.random.RandomState(seed)
        pr = np.random.RandomState(500 + seed)
        t = orc.Tree(orc.make_cfg(A, K, 0, 1)); t.seed(seed)
        root_pol = _policy(pr, A, False)
        t.root_init(root_pol, train=False)
        p0 = root_pol + 1e-12; p0 = p0 / p0.sum()
        rs.choice(A, A, p=p0, replace=False)
        leaf, parent, act, flag = t.select()
        for _ in range(A):
            rs.uniform(low=1e-7, high=2e-7, size=1)
        assert parent == 0 and flag == 0 and act == leaf - 1
        pol = _policy(pr, A, seed % 2 == 0)
        t.expand_backup(pol, 0.5)
        p = pol + 1e-12; p = p / p.sum()
        picks = np.sort(rs.choice(A, K, p=p, replace=False))
        d = t.dump()
        cb = d["child_base"][leaf]
        assert cb == 1 + A
        assert list(d["action"][cb:cb + K]) == list(picks)
        assert np.array_equal(d["prior"][cb:cb + K], p[picks])
        assert t.random_sample() == rs.random_sample()


def test_philox_known_answers_and_the_librarys_host_evaluation():
    """Philox4x32-10: the published known-answer vectors (Random123 kat_vectors: all-zero input; the pi / e digit
    input) on the oracle's restatement, and the library's host evaluation of a stream (smz_philox_words: word i =
    component i & 3 of philox(counter i / 4, key = seed)) against it, across a 624-word block boundary."""
    import ctypes as C
    import orc
    import stochastic_muzero_amd as smz
    L = orc.lib()
    out = (C.c_uint32 * 4)()
    L.orc_philox_block(0, 0, 0, 0, out)
    assert list(out) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]

    def py(c, k):                       # the published round function, all four counter words
        c, k = list(c), list(k)
        for _ in range(10):
            p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
            c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xffffffff, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xffffffff]
            k = [(k[0] + 0x9E3779B9) & 0xffffffff, (k[1] + 0xBB67AE85) & 0xffffffff]
        return c
    assert py([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    seed = 0x0123456789abcdef
    lib = smz._lib.load()
    w = np.zeros(40, np.uint32)
    assert lib.smz_philox_words(seed, 3, 610, 40, w.ctypes.data_as(C.c_void_p)) == 0
    want = []
    block, idx = 3, 610
    for _ in range(40):
        n = block * 156 + idx // 4
        want.append(py([n & 0xffffffff, n >> 32, 0, 0], [seed & 0xffffffff, seed >> 32])[idx & 3])
        idx += 1
        if idx == 624:
            idx, block = 0, block + 1
    assert list(w) == want
    t = orc.Tree(orc.make_cfg(2, 2, 0, 4))
    t.seed_philox(seed)
    a = [t.random_sample() for _ in range(3)]
    w6 = np.zeros(6, np.uint32)
    lib.smz_philox_words(seed, 0, 0, 6, w6.ctypes.data_as(C.c_void_p))
    assert a == [((int(w6[2 * i]) >> 5) * 67108864.0 + (int(w6[2 * i + 1]) >> 6)) / 9007199254740992.0 for i in range(3)]
    assert t.philox_position() == (0, 6)
