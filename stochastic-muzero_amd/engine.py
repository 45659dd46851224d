"""SearchEngine: B independent Stochastic-MuZero search trees resident on one MI355X.

Thin host object over the C ABI (include/smz.h).  It owns a libsmz handle plus the torch tensors the kernels
write their per-simulation outputs into (fixed addresses, so a simulation round can be captured in a HIP graph).
torch is used here only as the device-memory / stream provider; all tree work happens in libsmz's HIP kernels.

Reference semantics: tree i behaves as `np.random.seed(seed_i); Monte_carlo_tree_search(...).run(obs_i, model)`
(monte_carlo_tree_search.py:311-349) followed by Game.policy_step / store_search_statistics (game.py:179-273).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def pb_c_table(pb_c_base, pb_c_init, n):
    """pb_c[Np] with numpy's own log, exactly the expression of monte_carlo_tree_search.py:236."""
    return np.array([float(np.log((v + pb_c_base + 1) / pb_c_base) + pb_c_init) for v in range(n)], dtype=np.float64)


def pow_table(temperature, n):
    """float64(visits) ** (1/T) as game.py:208 evaluates it (numpy's vectorised pow, not libm's)."""
    return np.ascontiguousarray(np.arange(n, dtype=np.float64) ** (1 / temperature))


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class SearchEngine:
    def __init__(self, num_trees, num_actions, hidden_size, num_simulations=10, maxium_action_sample=2,
                 pb_c_base=19652, pb_c_init=1.25, discount=0.95, root_dirichlet_alpha=0.25,
                 root_exploration_fraction=0.25, device=None, rng_mode=_lib.RNG_MT19937_NUMPY):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("SearchEngine needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", int(device))
        self.B, self.A, self.S, self.sims = int(num_trees), int(num_actions), int(hidden_size), int(num_simulations)
        self.K = min(int(maxium_action_sample), self.A)
        self.cfg = _lib.Config(self.B, self.A, int(maxium_action_sample), self.S, self.sims, int(pb_c_base),
                               float(pb_c_init), float(discount), float(root_dirichlet_alpha),
                               float(root_exploration_fraction), int(rng_mode), int(self.device.index))
        h = C.c_void_p()
        _lib.check(self.lib.smz_create(C.byref(self.cfg), C.byref(h)))
        self.h = h
        self.N = self.lib.smz_node_capacity(self.h)
        tab = pb_c_table(int(pb_c_base), float(pb_c_init), self.sims + 2)
        _lib.check(self.lib.smz_set_pb_c_table(self.h, tab.ctypes.data_as(C.c_void_p), tab.size))
        self._pow_tables = {}
        dev = self.device
        B, A, S = self.B, self.A, self.S
        # outputs of select (inputs of the heads)
        self.parent_hidden = torch.empty(B, max(S, 1), dtype=torch.float32, device=dev)
        self.last_action = torch.empty(B, dtype=torch.int32, device=dev)
        self.branch = torch.empty(B, dtype=torch.uint8, device=dev)
        self.mlp_input = torch.empty(B, S + A, dtype=torch.float32, device=dev)
        # outputs of root_stats / act
        self.visits = torch.empty(B, A, dtype=torch.int32, device=dev)
        self.priors = torch.empty(B, A, dtype=torch.float64, device=dev)
        self.root_value = torch.empty(B, dtype=torch.float32, device=dev)
        self.child_reward = torch.empty(B, A, dtype=torch.float32, device=dev)
        self.action = torch.empty(B, dtype=torch.int32, device=dev)
        self.policy = torch.empty(B, A, dtype=torch.float64, device=dev)
        self.child_visits = torch.empty(B, A, dtype=torch.float64, device=dev)

    # ---- lifetime ---------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self.lib.smz_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def set_active(self, active):
        """Per-tree on/off switch (smz_set_active): `active` is a uint8 [B] device tensor (kept alive here) or None.
        Trees whose byte is 0 are skipped by every search phase and by act(): the batched `while not
        environment.terminal` of self_play.py:79."""
        if active is not None:
            assert active.dtype == torch.uint8 and active.is_contiguous() and active.device == self.device and \
                tuple(active.shape) == (self.B,)
        self._active = active
        _lib.check(self.lib.smz_set_active(self.h, _ptr(active)))

    def enable_leaf_ids(self, on=True):
        """smz_set_leaf_ids_out: every selection also writes (leaf node id, parent node id) per tree to `self.leaf_ids`
        [B,2] int32, so that a network kernel can take and put its rows in the tree's own hidden-state storage
        (heads.HipMlpHeads at large batches) and the tree kernels move no rows."""
        if on and getattr(self, "leaf_ids", None) is None:
            self.leaf_ids = torch.full((self.B, 2), -1, dtype=torch.int32, device=self.device)
        if not on:
            self.leaf_ids = None
        _lib.check(self.lib.smz_set_leaf_ids_out(self.h, _ptr(self.leaf_ids if on else None)))

    def hidden_layout(self):
        """(device address of the hidden-state storage, nodes per tree, floats between rows): smz_get_hidden_layout."""
        base, n, hs = C.c_void_p(), C.c_int(), C.c_int()
        _lib.check(self.lib.smz_get_hidden_layout(self.h, C.byref(base), C.byref(n), C.byref(hs)))
        return base, n.value, hs.value

    # ---- random streams ------------------------------------------------------------------------------------------
    def seed(self, seeds):
        """numpy `seed(int)` per tree; a scalar s seeds tree i with s + i."""
        if np.isscalar(seeds):
            seeds = np.arange(self.B, dtype=np.uint64) + np.uint64(seeds)
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        assert seeds.shape == (self.B,)
        _lib.check(self.lib.smz_seed(self.h, seeds.ctypes.data_as(C.c_void_p), self._stream()))

    def set_rng_state(self, tree, key, pos):
        key = np.ascontiguousarray(key, dtype=np.uint32)
        assert key.shape == (624,)
        _lib.check(self.lib.smz_set_rng_state(self.h, int(tree), key.ctypes.data_as(C.c_void_p), int(pos)))

    def get_rng_state(self, tree):
        key = np.zeros(624, np.uint32)
        pos = C.c_int()
        _lib.check(self.lib.smz_get_rng_state(self.h, int(tree), key.ctypes.data_as(C.c_void_p), C.byref(pos)))
        return key, pos.value

    def philox_position(self, tree):
        """(rng_mode PHILOX) (block, idx): tree `tree` has consumed block * 624 + idx words of its stream."""
        b, i = C.c_uint32(), C.c_int()
        _lib.check(self.lib.smz_get_philox_position(self.h, int(tree), C.byref(b), C.byref(i)))
        return b.value, i.value

    def snapshot_rng(self):
        _lib.check(self.lib.smz_rng_snapshot(self.h, self._stream()))

    def restore_rng(self):
        _lib.check(self.lib.smz_rng_restore(self.h, self._stream()))

    # ---- search phases ----------------------------------------------------------------------------------------
    def _f32(self, t, shape):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.device == self.device and tuple(t.shape) == shape, \
            (t.dtype, t.shape, shape, t.device)
        return t

    def root_init(self, hidden, policy, train=True, noise_override=None):
        self._act_done = None
        self.env_stepped = False
        hidden = self._f32(hidden.reshape(self.B, -1), (self.B, self.S)) if self.S > 0 else None
        policy = self._f32(policy, (self.B, self.A))
        if noise_override is not None:
            assert noise_override.dtype == torch.float64 and tuple(noise_override.shape) == (self.B, self.A)
        _lib.check(self.lib.smz_root_init(self.h, _ptr(hidden), _ptr(policy), _ptr(noise_override), int(bool(train)),
                                          self._stream()))

    def select(self, want_mlp_input=True, want_parent_hidden=True):
        _lib.check(self.lib.smz_select(self.h, _ptr(self.parent_hidden) if (want_parent_hidden and self.S > 0) else None,
                                       _ptr(self.last_action), _ptr(self.branch),
                                       _ptr(self.mlp_input) if (want_mlp_input and self.S > 0) else None, self._stream()))
        return self.parent_hidden, self.last_action, self.branch, self.mlp_input

    def expand_backup(self, hidden, reward, policy, value):
        # (hidden None: the leaf rows are in the tree already -- smz_mlp_recurrent_rows wrote them)
        hidden = self._f32(hidden.reshape(self.B, -1), (self.B, self.S)) if (self.S > 0 and hidden is not None) else None
        _lib.check(self.lib.smz_expand_backup(self.h, _ptr(hidden), _ptr(None if reward is None else self._f32(reward, (self.B,))),
                                              _ptr(self._f32(policy, (self.B, self.A))), _ptr(self._f32(value, (self.B,))),
                                              self._stream()))

    def expand_backup_select(self, hidden, reward, policy, value, want_mlp_input=True, want_parent_hidden=True):
        hidden = self._f32(hidden.reshape(self.B, -1), (self.B, self.S)) if (self.S > 0 and hidden is not None) else None
        _lib.check(self.lib.smz_expand_backup_select(
            self.h, _ptr(hidden), _ptr(None if reward is None else self._f32(reward, (self.B,))),
            _ptr(self._f32(policy, (self.B, self.A))), _ptr(self._f32(value, (self.B,))),
            _ptr(self.parent_hidden) if (want_parent_hidden and self.S > 0) else None, _ptr(self.last_action),
            _ptr(self.branch), _ptr(self.mlp_input) if (want_mlp_input and self.S > 0) else None, self._stream()))
        return self.parent_hidden, self.last_action, self.branch, self.mlp_input

    def _pow_table(self, temperature):
        if temperature < 0.3:
            return None
        tab = self._pow_tables.get(temperature)
        if tab is None:
            tab = self._pow_tables[temperature] = pow_table(temperature, self.sims + 1)
        return tab

    def search_mlp(self, mlp_desc, weights, obs, train=True, act_temperature=None, env_step=None):
        """Whole search (root + num_simulations rounds) in one launch with LDS-resident mlp_model heads.  With
        `act_temperature` the action selection of act() runs in the tail of the same launch; the next act() call with
        that temperature returns its outputs without launching anything.  `env_step` (a _lib.CartPoleEnv whose obs_dev is
        `obs`; needs act_temperature): the built-in env's step + trajectory record run in the tail too
        (smz_search_mlp_act_cartpole) -- one launch per env step."""
        assert obs.dtype == torch.float32 and obs.is_contiguous() and obs.shape[0] == self.B
        self._act_done = None
        self.env_stepped = False
        if env_step is not None:
            assert act_temperature is not None and env_step.obs_dev == obs.data_ptr()
            T = float(act_temperature)
            tab = self._pow_table(T)
            _lib.check(self.lib.smz_search_mlp_act_cartpole(self.h, C.byref(mlp_desc), _ptr(weights), int(bool(train)), T,
                                                            None if tab is None else tab.ctypes.data_as(C.c_void_p),
                                                            _ptr(self.action), _ptr(self.policy), _ptr(self.child_visits),
                                                            _ptr(self.root_value), C.byref(env_step), self._stream()))
            self._act_done = T
            self.env_stepped = True
            return
        if act_temperature is None:
            _lib.check(self.lib.smz_search_mlp(self.h, C.byref(mlp_desc), _ptr(weights), _ptr(obs), int(bool(train)),
                                               self._stream()))
            return
        T = float(act_temperature)
        tab = self._pow_table(T)
        _lib.check(self.lib.smz_search_mlp_act(self.h, C.byref(mlp_desc), _ptr(weights), _ptr(obs), int(bool(train)), T,
                                               None if tab is None else tab.ctypes.data_as(C.c_void_p), _ptr(self.action),
                                               _ptr(self.policy), _ptr(self.child_visits), _ptr(self.root_value),
                                               self._stream()))
        self._act_done = T

    def search_vision(self, vision_desc, weights, hidden0, policy0, train=True, act_temperature=None):
        """Whole search in one launch for `vision_model` heads (smz_search_vision): hidden0 [B,147] / policy0 [B,A] are
        smz_vision_initial's outputs.  `act_temperature` as in search_mlp."""
        hidden0 = self._f32(hidden0.reshape(self.B, -1), (self.B, self.S))
        policy0 = self._f32(policy0, (self.B, self.A))
        self._act_done = None
        self.env_stepped = False
        if act_temperature is None:
            _lib.check(self.lib.smz_search_vision(self.h, C.byref(vision_desc), _ptr(weights), _ptr(hidden0), _ptr(policy0),
                                                  int(bool(train)), self._stream()))
            return
        T = float(act_temperature)
        tab = self._pow_table(T)
        _lib.check(self.lib.smz_search_vision_act(self.h, C.byref(vision_desc), _ptr(weights), _ptr(hidden0), _ptr(policy0),
                                                  int(bool(train)), T, None if tab is None else tab.ctypes.data_as(C.c_void_p),
                                                  _ptr(self.action), _ptr(self.policy), _ptr(self.child_visits),
                                                  _ptr(self.root_value), self._stream()))
        self._act_done = T

    def root_stats(self):
        _lib.check(self.lib.smz_root_stats(self.h, _ptr(self.visits), _ptr(self.priors), _ptr(self.root_value),
                                           _ptr(self.child_reward), self._stream()))
        return self.visits, self.priors, self.root_value, self.child_reward

    def act(self, temperature):
        """Game.policy_step's policy/action + store_search_statistics (game.py:179-235) for every tree."""
        temperature = float(temperature)
        if getattr(self, "_act_done", None) == temperature:      # already computed in the tail of the search launch
            self._act_done = None
            return self.action, self.policy, self.child_visits, self.root_value
        self._act_done = None
        tab = self._pow_table(temperature)
        _lib.check(self.lib.smz_act(self.h, temperature, None if tab is None else tab.ctypes.data_as(C.c_void_p),
                                    _ptr(self.action), _ptr(self.policy), _ptr(self.child_visits), _ptr(self.root_value),
                                    self._stream()))
        return self.action, self.policy, self.child_visits, self.root_value

    # ---- inspection -------------------------------------------------------------------------------------------
    def dump_tree(self, tree):
        nodes = (_lib.NodeView * self.N)()
        minmax = np.zeros(2, np.float32)
        path = np.zeros(self.sims + 2, np.int32)
        plen = C.c_int32()
        rp = np.zeros(self.A, np.float64)
        n = _lib.check(self.lib.smz_debug_dump_tree(self.h, int(tree), nodes, self.N, minmax.ctypes.data_as(C.c_void_p),
                                                    path.ctypes.data_as(C.c_void_p), path.size, C.byref(plen),
                                                    rp.ctypes.data_as(C.c_void_p)))
        arr = np.frombuffer(nodes, dtype=np.dtype([("visit", "<i4"), ("value_sum", "<f4"), ("reward", "<f4"),
                                                   ("prior", "<f4"), ("child_base", "<i4"), ("action", "<i4")]))
        out = {k: arr[k].copy() for k in arr.dtype.names}
        out.update(n_nodes=n, minmax=minmax, path=path[:plen.value].copy(), root_priors=rp)
        return out

    def last_kernel(self):
        """Name of the single-launch search kernel instantiation launched last (as rocprofv3 prints it), "" if none."""
        buf = C.create_string_buffer(128)
        _lib.check(self.lib.smz_last_kernel(self.h, buf, 128))
        return buf.value.decode()

    def enable_stats(self, on=True):
        _lib.check(self.lib.smz_enable_stats(self.h, int(bool(on))))

    def read_stats(self, reset=True):
        v = np.zeros(4, np.uint64)
        _lib.check(self.lib.smz_read_stats(self.h, v.ctypes.data_as(C.c_void_p), int(bool(reset))))
        return dict(decision_levels=int(v[0]), chance_levels=int(v[1]), descents=int(v[2]), children_scored=int(v[3]))
