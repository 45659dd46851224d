"""ctypes binding of the CPU oracle (oracle/smz_oracle.c).

TEST INFRASTRUCTURE.  Import this only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (stochastic-muzero_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    src = os.path.join(_HERE, "smz_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liborc.so"])
    return so


class Cfg(C.Structure):
    _fields_ = [("A", C.c_int32), ("K", C.c_int32), ("S", C.c_int32), ("sims", C.c_int32),
                ("pb_c_base", C.c_int32), ("pb_c_init", C.c_double), ("discount", C.c_double),
                ("alpha", C.c_double), ("frac", C.c_double)]


_MLP_PTRS = ["rep_in_w", "rep_in_b", "rep_mid_w", "rep_mid_b", "rep_out_w", "rep_out_b",
             "pre_in_w", "pre_in_b", "pre_mid_w", "pre_mid_b", "pre_pol_w", "pre_pol_b", "pre_val_w", "pre_val_b",
             "apr_in_w", "apr_in_b", "apr_mid_w", "apr_mid_b", "apr_pol_w", "apr_pol_b", "apr_val_w", "apr_val_b",
             "ady_in_w", "ady_in_b", "ady_mid_w", "ady_mid_b", "ady_st_w", "ady_st_b",
             "dyn_in_w", "dyn_in_b", "dyn_mid_w", "dyn_mid_b", "dyn_rw_w", "dyn_rw_b", "dyn_st_w", "dyn_st_b"]


class Mlp(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("obs", "A", "S", "H", "L")] + [(n, C.c_void_p) for n in _MLP_PTRS]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_tree_new.restype = C.c_void_p
        L.orc_tree_new.argtypes = [C.POINTER(Cfg), C.c_void_p]
        L.orc_tree_free.argtypes = [C.c_void_p]
        L.orc_tree_seed.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_tree_seed_philox.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_tree_philox_position.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_philox_block.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_tree_set_rng.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.orc_tree_get_rng.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_tree_random_sample.restype = C.c_double
        L.orc_tree_random_sample.argtypes = [C.c_void_p]
        L.orc_root_init.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_select.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.orc_expand_backup.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_float]
        L.orc_root_stats.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.orc_act.argtypes = [C.c_void_p, C.c_double, C.c_void_p] + [C.c_void_p] * 4
        L.orc_dump.restype = C.c_int32
        L.orc_dump.argtypes = [C.c_void_p] + [C.c_void_p] * 9
        L.orc_run_mlp.argtypes = [C.c_void_p, C.POINTER(Mlp), C.c_void_p, C.c_int]
        L.orc_np_sum_f32.restype = C.c_float
        L.orc_np_sum_f32.argtypes = [C.c_void_p, C.c_int]
        L.orc_np_sum_f64.restype = C.c_double
        L.orc_np_sum_f64.argtypes = [C.c_void_p, C.c_int]
        L.orc_support_decode.restype = C.c_float
        L.orc_support_decode.argtypes = [C.c_void_p, C.c_int]
        L.orc_cartpole_step.argtypes = [C.c_void_p, C.c_int]
        L.orc_selfplay_cartpole.restype = C.c_int64
        L.orc_selfplay_cartpole.argtypes = [C.POINTER(Cfg), C.POINTER(Mlp), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_selfplay_observations.restype = C.c_int64
        L.orc_selfplay_observations.argtypes = [C.POINTER(Cfg), C.POINTER(Mlp), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p,
                                                C.c_void_p, C.c_void_p]
        for name in ("orc_mlp_representation",):
            getattr(L, name).argtypes = [C.POINTER(Mlp), C.c_void_p, C.c_void_p]
        L.orc_mlp_prediction.argtypes = [C.POINTER(Mlp), C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mlp_afterstate_prediction.argtypes = [C.POINTER(Mlp), C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mlp_afterstate_dynamics.argtypes = [C.POINTER(Mlp), C.c_void_p, C.c_int, C.c_void_p]
        L.orc_mlp_dynamics.argtypes = [C.POINTER(Mlp), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def pbc_table(pb_c_base, pb_c_init, n):
    """pb_c[Np] exactly as monte_carlo_tree_search.py:236 evaluates it (numpy's own log, not libm's)."""
    return np.array([float(np.log((v + pb_c_base + 1) / pb_c_base) + pb_c_init) for v in range(n)], dtype=np.float64)


def pow_table(temperature, n):
    """float64(v) ** (1/T) for v in 0..n-1 as game.py:208 evaluates it on the visit-count vector."""
    return np.arange(n, dtype=np.float64) ** (1 / temperature)


def make_cfg(A, K, S, sims, pb_c_base=19652, pb_c_init=1.25, discount=0.95, alpha=0.25, frac=0.25):
    return Cfg(A, min(K, A), S, sims, pb_c_base, pb_c_init, discount, alpha, frac)


class MlpWeights:
    """Holds float32 weight arrays (name -> ndarray, names as in _MLP_PTRS) and the matching C struct."""

    def __init__(self, dims, arrays):
        self.arrays = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in arrays.items()}
        self.dims = dict(dims)
        self.struct = Mlp()
        for k in ("obs", "A", "S", "H", "L"):
            setattr(self.struct, k, int(dims[k]))
        for k in _MLP_PTRS:
            a = self.arrays.get(k)
            setattr(self.struct, k, None if a is None else a.ctypes.data)

    @classmethod
    def from_npz(cls, path):
        z = np.load(path)
        dims = {k: int(z["dim_" + k]) for k in ("obs", "A", "S", "H", "L")}
        return cls(dims, {k: z[k] for k in _MLP_PTRS if k in z.files})


class Tree:
    """One oracle search tree with its own numpy-legacy MT19937 stream."""

    def __init__(self, cfg, pbc=None):
        self.cfg = cfg
        self.L = lib()
        if pbc is None:
            pbc = pbc_table(cfg.pb_c_base, cfg.pb_c_init, cfg.sims + 2)
        self._pbc = np.ascontiguousarray(pbc, dtype=np.float64)
        assert self._pbc.size >= cfg.sims + 2
        self.h = self.L.orc_tree_new(C.byref(cfg), _p(self._pbc))
        self.N = 1 + cfg.A + cfg.sims * cfg.K

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_tree_free(self.h)
            self.h = None

    def seed(self, s):
        self.L.orc_tree_seed(self.h, int(s) & 0xFFFFFFFF)

    def seed_philox(self, s):
        """The engine's throughput mode (SMZ_RNG_PHILOX): words from Philox4x32-10 keyed by the 64-bit seed."""
        self.L.orc_tree_seed_philox(self.h, int(s) & 0xFFFFFFFFFFFFFFFF)

    def philox_position(self):
        b, p = C.c_uint32(), C.c_int32()
        self.L.orc_tree_philox_position(self.h, C.byref(b), C.byref(p))
        return b.value, p.value

    def set_rng(self, key, pos):
        key = np.ascontiguousarray(key, dtype=np.uint32)
        self.L.orc_tree_set_rng(self.h, _p(key), int(pos))

    def get_rng(self):
        key = np.zeros(624, np.uint32)
        pos = C.c_int32()
        self.L.orc_tree_get_rng(self.h, _p(key), C.byref(pos))
        return key, pos.value

    def random_sample(self):
        return self.L.orc_tree_random_sample(self.h)

    def root_init(self, policy, hidden=None, train=True):
        policy = np.ascontiguousarray(policy, dtype=np.float32).reshape(-1)
        hidden = None if hidden is None else np.ascontiguousarray(hidden, dtype=np.float32).reshape(-1)
        noise = np.zeros(self.cfg.A, np.float64)
        self.L.orc_root_init(self.h, _p(hidden), _p(policy), int(bool(train)), _p(noise))
        return noise

    def select(self, want_hidden=False):
        out = [C.c_int32() for _ in range(4)]
        ph = np.zeros(max(self.cfg.S, 1), np.float32) if want_hidden else None
        self.L.orc_select(self.h, *[C.byref(o) for o in out], _p(ph))
        leaf, parent, act, flag = (o.value for o in out)
        return (leaf, parent, act, flag, ph) if want_hidden else (leaf, parent, act, flag)

    def expand_backup(self, policy, value, reward=0.0, hidden=None):
        policy = np.ascontiguousarray(policy, dtype=np.float32).reshape(-1)
        hidden = None if hidden is None else np.ascontiguousarray(hidden, dtype=np.float32).reshape(-1)
        self.L.orc_expand_backup(self.h, _p(hidden), C.c_float(float(reward)), _p(policy), C.c_float(float(value)))

    def root_stats(self):
        A = self.cfg.A
        v = np.zeros(A, np.int32); p = np.zeros(A, np.float64); cr = np.zeros(A, np.float32); rv = C.c_float()
        self.L.orc_root_stats(self.h, _p(v), _p(p), C.byref(rv), _p(cr))
        return v, p, np.float32(rv.value), cr

    def act(self, temperature, pow_lut=None):
        A = self.cfg.A
        if pow_lut is None and temperature >= 0.3:
            pow_lut = pow_table(temperature, self.cfg.sims + 1)
        pow_lut = None if pow_lut is None else np.ascontiguousarray(pow_lut, dtype=np.float64)
        a = C.c_int32(); pol = np.zeros(A, np.float64); cv = np.zeros(A, np.float64); rv = C.c_float()
        self.L.orc_act(self.h, float(temperature), _p(pow_lut), C.byref(a), _p(pol), _p(cv), C.byref(rv))
        return a.value, pol, cv, np.float32(rv.value)

    def dump(self):
        N = self.N
        d = dict(visit=np.zeros(N, np.int32), value_sum=np.zeros(N, np.float32), reward=np.zeros(N, np.float32),
                 prior=np.zeros(N, np.float32), child_base=np.zeros(N, np.int32), action=np.zeros(N, np.int32),
                 minmax=np.zeros(2, np.float32), path=np.zeros(self.cfg.sims + 2, np.int32))
        pl = C.c_int32()
        n = self.L.orc_dump(self.h, _p(d["visit"]), _p(d["value_sum"]), _p(d["reward"]), _p(d["prior"]),
                            _p(d["child_base"]), _p(d["action"]), _p(d["minmax"]), _p(d["path"]), C.byref(pl))
        d["n_nodes"] = n
        d["path"] = d["path"][:pl.value]
        return d

    def run_mlp(self, weights, obs, train=True):
        obs = np.ascontiguousarray(obs, dtype=np.float32).reshape(-1)
        self.L.orc_run_mlp(self.h, C.byref(weights.struct), _p(obs), int(bool(train)))


def selfplay_cartpole(cfg, weights, obs0, seeds, steps, temperature=0.0, train=True, threads=1, pbc=None,
                      record=True):
    """CPU baseline / end-to-end oracle: fixed-length synthetic CartPole self-play.  Returns dict."""
    L = lib()
    obs0 = np.ascontiguousarray(obs0, dtype=np.float64)
    n_env = obs0.shape[0]
    seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
    if pbc is None:
        pbc = pbc_table(cfg.pb_c_base, cfg.pb_c_init, cfg.sims + 2)
    pbc = np.ascontiguousarray(pbc, dtype=np.float64)
    pow_lut = pow_table(temperature, cfg.sims + 1) if temperature >= 0.3 else None
    acts = np.zeros((n_env, steps), np.int32) if record else None
    vis = np.zeros((n_env, steps, cfg.A), np.int32) if record else None
    rv = np.zeros((n_env, steps), np.float32) if record else None
    n = L.orc_selfplay_cartpole(C.byref(cfg), C.byref(weights.struct), _p(pbc), _p(pow_lut), _p(obs0), _p(seeds),
                                n_env, steps, float(temperature), int(bool(train)), int(threads), _p(acts), _p(vis), _p(rv))
    return dict(simulations=int(n), actions=acts, visits=vis, root_values=rv)


def selfplay_observations(cfg, weights, obs_seq, seeds, temperature=0.0, train=True, threads=1, pbc=None, record=True):
    """As selfplay_cartpole for observation-only stand-in envs: obs_seq [n_env][steps][obs_dim] float32 is what the env
    shows the agent at each step (the actions do not influence it)."""
    L = lib()
    obs_seq = np.ascontiguousarray(obs_seq, dtype=np.float32)
    n_env, steps, obs_dim = obs_seq.shape
    seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
    if pbc is None:
        pbc = pbc_table(cfg.pb_c_base, cfg.pb_c_init, cfg.sims + 2)
    pbc = np.ascontiguousarray(pbc, dtype=np.float64)
    pow_lut = pow_table(temperature, cfg.sims + 1) if temperature >= 0.3 else None
    acts = np.zeros((n_env, steps), np.int32) if record else None
    vis = np.zeros((n_env, steps, cfg.A), np.int32) if record else None
    rv = np.zeros((n_env, steps), np.float32) if record else None
    n = L.orc_selfplay_observations(C.byref(cfg), C.byref(weights.struct), _p(pbc), _p(pow_lut), _p(obs_seq), obs_dim,
                                    _p(seeds), n_env, steps, float(temperature), int(bool(train)), int(threads), _p(acts),
                                    _p(vis), _p(rv))
    return dict(simulations=int(n), actions=acts, visits=vis, root_values=rv)
