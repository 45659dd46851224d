"""Import plumbing for the Python reference (this container only).

TEST INFRASTRUCTURE.  Used by oracle/gen_golden.py to import /root/reference so that golden vectors can be
generated from the reference's own code.  Nothing here travels to the GPU box as a dependency of the product:
the reference is absent there and only the committed fixtures under tests/golden/ are used.

The reference imports four third-party packages that are not installed in this image (gymnasium, torchvision,
ray; matplotlib is present).  None of them is on the arithmetic path that the goldens pin:
 * gymnasium  -- only `gym.spaces.*` type checks in muzero_model.py:484-494,1008-1058 and the env object that
                 game.py drives through reset/step/close/metadata (game.py:102,125,134; self_play.py:74).
 * torchvision-- only builds the RGB resize pipeline (game.py:82-89), never touched for vector observations.
 * ray        -- decorator on play_game_ray (self_play.py:21), unused by the sequential play_game.
Import stand-ins below provide just those names.
"""
import sys, types, os, functools

REF = os.environ.get("SMZ_REFERENCE_DIR", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REF, "monte_carlo_tree_search.py"))


class _Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = "int64"


class _Box:
    def __init__(self, low, high, shape=None, dtype="float32"):
        import numpy as np
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape if shape is not None else np.shape(low)).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), shape if shape is not None else np.shape(high)).copy()
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)


class _Tuple(tuple):
    pass


def install_stubs():
    if "gymnasium" not in sys.modules:
        gym = types.ModuleType("gymnasium")
        spaces = types.ModuleType("gymnasium.spaces")
        box = types.ModuleType("gymnasium.spaces.box")
        tup = types.ModuleType("gymnasium.spaces.tuple")
        spaces.Discrete, spaces.Box = _Discrete, _Box
        box.Box, tup.Tuple = _Box, _Tuple
        spaces.box, spaces.tuple = box, tup
        spaces.Tuple = _Tuple
        gym.spaces = spaces
        sys.modules.update({"gymnasium": gym, "gymnasium.spaces": spaces,
                            "gymnasium.spaces.box": box, "gymnasium.spaces.tuple": tup})
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tr = types.ModuleType("torchvision.transforms")

        class _Compose:
            def __init__(self, fns):
                self.fns = fns

            def __call__(self, x):
                raise RuntimeError("torchvision stand-in: RGB pipeline is not available in this image")
        tr.Compose = _Compose
        tr.ToTensor = lambda *a, **k: None
        tr.Resize = lambda *a, **k: None
        tv.transforms = tr
        sys.modules.update({"torchvision": tv, "torchvision.transforms": tr})
    if "ray" not in sys.modules:
        ray = types.ModuleType("ray")
        ray.remote = lambda f: f
        ray.init = lambda *a, **k: None
        ray.get = lambda x: x
        sys.modules["ray"] = ray
    try:
        import matplotlib
        matplotlib.use("Agg")
    except Exception:
        mpl = types.ModuleType("matplotlib")
        plt = types.ModuleType("matplotlib.pyplot")
        mpl.pyplot = plt
        sys.modules.update({"matplotlib": mpl, "matplotlib.pyplot": plt})


def import_reference():
    """Returns a namespace with the reference modules imported from REF."""
    import torch
    if not available():
        raise RuntimeError(f"reference not found at {REF}")
    sys.dont_write_bytecode = True
    install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if not getattr(torch.load, "_smz_patched", False):
        patched = functools.partial(torch.load, weights_only=False)   # whole-module pickles, muzero_model.py:983-988
        patched._smz_patched = True
        torch.load = patched
    import monte_carlo_tree_search as M
    import muzero_model as MM
    import game as G
    import replay_buffer as RB
    import self_play as SP
    return types.SimpleNamespace(mcts=M, model=MM, game=G, replay_buffer=RB, self_play=SP,
                                 Discrete=_Discrete, Box=_Box)
