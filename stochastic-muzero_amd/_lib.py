"""ctypes binding of libsmz.so (the C ABI declared in include/smz.h).

There is no fallback: if the HIP library is missing or was not built, importing the engine fails loudly.
Build it with `python __graft_entry__.py build` (or `make -C stochastic-muzero_amd/csrc`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMZ_LIB_PATH") or os.path.join(_HERE, "libsmz.so")   # (SMZ_LIB_PATH: A/B builds, tools/)

SMZ_OK, SMZ_ERR_INVALID, SMZ_ERR_HIP, SMZ_ERR_NOMEM, SMZ_ERR_STATE, SMZ_ERR_TOO_LARGE = 0, -1, -2, -3, -4, -5
RNG_MT19937_NUMPY, RNG_PHILOX = 0, 1
MAX_ACTIONS = 32


class SmzError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"libsmz error {code}: {text}")
        self.code = code


class Config(C.Structure):
    """smz_config (include/smz.h) -- kwargs of Monte_carlo_tree_search.__init__ + batch geometry."""
    _fields_ = [("num_trees", C.c_int32), ("num_actions", C.c_int32), ("max_action_sample", C.c_int32),
                ("hidden_size", C.c_int32), ("num_simulations", C.c_int32), ("pb_c_base", C.c_int32),
                ("pb_c_init", C.c_double), ("discount", C.c_double), ("root_dirichlet_alpha", C.c_double),
                ("root_exploration_fraction", C.c_double), ("rng_mode", C.c_int32), ("device", C.c_int32)]


class MlpDesc(C.Structure):
    """smz_mlp_desc (include/smz.h): dimensions + float offsets of the packed mlp_model weight buffer."""
    _fields_ = [("obs", C.c_int32), ("A", C.c_int32), ("S", C.c_int32), ("H", C.c_int32), ("L", C.c_int32),
                ("OP", C.c_int32), ("total_floats", C.c_int32), ("off", C.c_int32 * 30)]


class VisionDesc(C.Structure):
    """smz_vision_desc (include/smz.h): dimensions + float offsets of the packed vision_model weight buffer."""
    _fields_ = [("A", C.c_int32), ("S", C.c_int32), ("H", C.c_int32), ("L", C.c_int32), ("OP", C.c_int32),
                ("total_floats", C.c_int32), ("small_floats", C.c_int32), ("off", C.c_int32 * 80)]


class EpisodeCtl(C.Structure):
    """smz_episode_ctl (include/smz.h): per-env game bookkeeping of smz_cartpole_step_ctl."""
    _fields_ = [("step_count_dev", C.c_void_p), ("episode_dev", C.c_void_p), ("active_dev", C.c_void_p),
                ("limit", C.c_int32), ("on_end", C.c_int32), ("reset_seed", C.c_uint64), ("first_env", C.c_int64)]


class CartPoleEnv(C.Structure):
    """smz_cartpole_env (include/smz.h): the built-in env's buffers for smz_search_mlp_act_cartpole."""
    _fields_ = [("state_dev", C.c_void_p), ("obs_dev", C.c_void_p), ("reward_dev", C.c_void_p), ("flag_dev", C.c_void_p),
                ("ctl", C.POINTER(EpisodeCtl)), ("traj_dev", C.c_void_p), ("T", C.c_int32), ("t", C.c_int32)]


class NodeView(C.Structure):
    _fields_ = [("visit_count", C.c_int32), ("value_sum", C.c_float), ("reward", C.c_float), ("prior", C.c_float),
                ("child_base", C.c_int32), ("action", C.c_int32)]


_P = C.c_void_p
# every exported symbol of include/smz.h: name -> (restype, argtypes)
SIGNATURES = {
    "smz_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "smz_destroy": (C.c_int, [_P]),
    "smz_abi_version": (C.c_int, []),
    "smz_build_features": (C.c_int, []),
    "smz_last_error": (C.c_char_p, []),
    "smz_node_capacity": (C.c_int, [_P]),
    "smz_last_kernel": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "smz_set_pb_c_table": (C.c_int, [_P, _P, C.c_int]),
    "smz_seed": (C.c_int, [_P, _P, _P]),
    "smz_set_rng_state": (C.c_int, [_P, C.c_int, _P, C.c_int]),
    "smz_get_rng_state": (C.c_int, [_P, C.c_int, _P, C.POINTER(C.c_int)]),
    "smz_philox_words": (C.c_int, [C.c_uint64, C.c_uint32, C.c_int, C.c_int, _P]),
    "smz_get_philox_position": (C.c_int, [_P, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "smz_rng_snapshot": (C.c_int, [_P, _P]),
    "smz_rng_restore": (C.c_int, [_P, _P]),
    "smz_root_init": (C.c_int, [_P, _P, _P, _P, C.c_int, _P]),
    "smz_select": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "smz_expand_backup": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "smz_expand_backup_select": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "smz_root_stats": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "smz_act": (C.c_int, [_P, C.c_double, _P, _P, _P, _P, _P, _P]),
    "smz_support_decode": (C.c_int, [_P, C.c_int, _P, C.c_int, _P]),
    "smz_policy_softmax": (C.c_int, [_P, C.c_int, _P, C.c_int, _P]),
    "smz_dynamics_epilogue": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_int, _P, _P, C.c_int, _P]),
    "smz_prediction_epilogue": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, C.c_int, C.c_int, _P, _P, C.c_int, _P]),
    "smz_mlp_layout": (C.c_int, [C.POINTER(MlpDesc)]),
    "smz_mlp_initial": (C.c_int, [C.POINTER(MlpDesc), _P, _P, _P, _P, C.c_int, _P]),
    "smz_mlp_recurrent": (C.c_int, [C.POINTER(MlpDesc), _P, _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "smz_mlp_layout_wide": (C.c_int, [C.POINTER(MlpDesc)]),
    "smz_mlp_recurrent_wide": (C.c_int, [C.POINTER(MlpDesc), _P, _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "smz_vision_layout": (C.c_int, [C.POINTER(VisionDesc)]),
    "smz_vision_initial": (C.c_int, [C.POINTER(VisionDesc), _P, _P, _P, _P, C.c_int, _P]),
    "smz_vision_initial_record": (C.c_int, [C.POINTER(VisionDesc), _P, _P, _P, _P, _P, C.c_int, _P]),
    "smz_vision_recurrent": (C.c_int, [C.POINTER(VisionDesc), _P, _P, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "smz_search_mlp_act": (C.c_int, [_P, C.POINTER(MlpDesc), _P, _P, C.c_int, C.c_double, _P, _P, _P, _P, _P, _P]),
    "smz_search_mlp_act_cartpole": (C.c_int, [_P, C.POINTER(MlpDesc), _P, C.c_int, C.c_double, _P, _P, _P, _P, _P,
                                              C.POINTER(CartPoleEnv), _P]),
    "smz_search_mlp": (C.c_int, [_P, C.POINTER(MlpDesc), _P, _P, C.c_int, _P]),
    "smz_frames_resize_u8": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "smz_frames_resize_taps_u8": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "smz_host_register": (C.c_int, [_P, C.c_size_t]),
    "smz_host_unregister": (C.c_int, [_P]),
    "smz_copy_async": (C.c_int, [_P, _P, C.c_size_t, C.c_int, _P]),
    "smz_search_vision": (C.c_int, [_P, C.POINTER(VisionDesc), _P, _P, _P, C.c_int, _P]),
    "smz_search_vision_act": (C.c_int, [_P, C.POINTER(VisionDesc), _P, _P, _P, C.c_int, C.c_double, _P, _P, _P, _P, _P, _P]),
    "smz_cartpole_step": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, _P]),
    "smz_cartpole_step_pack": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P, C.c_int, _P]),
    "smz_cartpole_step_ctl": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(EpisodeCtl), _P, C.c_int, C.c_int, _P, _P, _P, C.c_int, _P]),
    "smz_cartpole_reset_state": (C.c_int, [C.c_uint64, C.c_int64, C.c_int64, C.POINTER(C.c_double * 4)]),
    "smz_synthetic_obs": (C.c_int, [_P, C.c_int, C.c_int, C.c_uint64, C.c_int64, C.c_int64, _P]),
    "smz_set_active": (C.c_int, [_P, _P]),
    "smz_set_leaf_ids_out": (C.c_int, [_P, _P]),
    "smz_host_cartpole_step": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, C.c_int]),
    "smz_get_hidden_layout": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "smz_mlp_recurrent_rows": (C.c_int, [C.POINTER(MlpDesc), _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "smz_traj_floats": (C.c_int, [C.c_int, C.c_int]),
    "smz_traj_pack": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "smz_traj_targets": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P, _P]),
    "smz_traj_targets_games": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "smz_debug_div_by_count": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "smz_debug_glibc_log_pow": (C.c_int, [_P, _P, C.c_int, _P, _P, _P]),
    "smz_debug_dump_tree": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P, C.c_int, C.POINTER(C.c_int32), _P]),
    "smz_enable_stats": (C.c_int, [_P, C.c_int]),
    "smz_read_stats": (C.c_int, [_P, _P, C.c_int]),
}

_lib = None
MISSING = []          # symbols an SMZ_LIB_PATH library lacks (A/B runs against older builds)


class _Partial:
    """An older library loaded on purpose (SMZ_LIB_PATH): using an entry point it lacks says so by name instead of failing
    inside ctypes."""

    def __init__(self, lib, missing):
        self.__dict__["_lib"], self.__dict__["_missing"] = lib, missing

    def __getattr__(self, name):
        if name in self._missing:
            # (an AttributeError, so that `hasattr(lib, name)` feature checks keep working)
            raise AttributeError(f"{LIB_PATH} (SMZ_LIB_PATH) does not export {name}; it was skipped at load time")
        return getattr(self._lib, name)


def load():
    """Loads libsmz.so and declares every signature; raises if the library is absent (no CPU fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. Run `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the search engine.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)   # AttributeError here = header/library mismatch
            except AttributeError:
                if os.environ.get("SMZ_LIB_PATH"):      # an older build loaded on purpose for an A/B run (tools/ab_lib.sh)
                    MISSING.append(name)
                    continue
                raise RuntimeError(f"{LIB_PATH} does not export {name}: the library is older than include/smz.h "
                                   "(rebuild: python __graft_entry__.py build)") from None
            fn.restype, fn.argtypes = res, args
        if lib.smz_abi_version() != 1:
            raise RuntimeError("libsmz.so ABI version mismatch")
        if MISSING:
            lib = _Partial(lib, tuple(MISSING))
        _lib = lib
    return _lib


def check(rc):
    if rc < 0:
        raise SmzError(rc, load().smz_last_error().decode("utf-8", "replace"))
    return rc
