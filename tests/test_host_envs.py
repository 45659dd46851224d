"""Host-resident envs: the rules around env.step (host_envs.HostSlice, game.py:96-131, 223-273) and the parallel
stepper -- worker processes over a shared block -- against the serial adapter, env by env.  CPU tests (device "cpu": the
transfers are plain copies); the GPU side is tests/test_gpu_host_envs.py."""
import os
import subprocess
import sys
from importlib import import_module

import numpy as np
import pytest
import torch

import stochastic_muzero_amd  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _envs():
    return import_module("stochastic-muzero_amd.envs")


def _he():
    return import_module("stochastic-muzero_amd.host_envs")


class Picky(import_module("stochastic-muzero_amd.host_envs").HostCartPole):
    """A CartPole that refuses action 1 on every third step (the illegal-move rule, game.py:123-131) -- a module-level class of
    this test file, so the worker processes must import it by name (the tests directory is the workers' working directory)."""

    def __init__(self):
        super().__init__()
        self.n = 0

    def step(self, action):
        self.n += 1
        if self.n % 3 == 0 and action == 1:
            raise ValueError("illegal")
        return super().step(action)


def _run(env, T, seed):
    r = np.random.RandomState(seed)
    out = [env.reset().clone()]
    for t in range(T):
        a = torch.from_numpy(r.randint(0, 2, env.B).astype(np.int32))
        obs, rew, flag = env.step(a)
        out.append((obs.clone(), rew.clone(), flag.clone(), env.record_obs.clone(),
                    None if env.active is None else env.active.clone()))
    return out


@pytest.mark.parametrize("on_end,limit", [("reset", 5), ("mask", 7), ("reset", 0)])
def test_parallel_stepper_equals_the_serial_adapter_env_by_env(on_end, limit, monkeypatch):
    monkeypatch.chdir(os.path.join(ROOT, "tests"))
    envs_mod = _envs()
    B, T = 23, 14
    res = []
    for workers in (0, 3, 5):
        env = envs_mod.HostVecEnv([Picky() for _ in range(B)], 4, 2, "cpu", env_seed=11, limit=limit, on_end=on_end,
                                  first_env=100, workers=workers)
        try:
            res.append(_run(env, T, 4))
        finally:
            env.close()
    for other in res[1:]:
        assert torch.equal(res[0][0], other[0])
        for a, b in zip(res[0][1:], other[1:]):
            for x, y in zip(a, b):
                assert (x is None and y is None) or torch.equal(x, y)
    flags = torch.stack([s[2] for s in res[0][1:]])
    assert (flags == 2).any() or limit == 0
    rewards = torch.stack([s[1] for s in res[0][1:]])
    assert (rewards < 0).any()                                 # illegal moves happened (and got the rule's reward)


def test_record_keeps_the_post_step_observation_of_an_env_that_was_reset():
    """on_end="reset": the next search sees the fresh observation, the record the post-step one (game.py:264)."""
    envs_mod, he = _envs(), _he()
    env = envs_mod.HostVecEnv([he.HostCartPole() for _ in range(6)], 4, 2, "cpu", env_seed=0, limit=3, on_end="reset")
    env.reset()
    twin = [he.HostCartPole() for _ in range(6)]
    for i, e in enumerate(twin):
        e.reset(seed=i)
    for t in range(3):
        a = torch.zeros(6, dtype=torch.int32)
        obs, rew, flag = env.step(a)
        post = np.stack([e.step(0)[0] for e in twin])
        assert np.array_equal(env.record_obs.numpy(), post)
    assert (flag == 2).all()
    fresh = np.stack([he.HostCartPole().reset(seed=i + 1000003)[0] for i in range(6)])
    assert np.array_equal(obs.numpy(), fresh) and not np.array_equal(fresh, post)
    env.close()


def test_workers_never_import_torch_and_exit_with_their_parent_object():
    envs_mod, he = _envs(), _he()
    env = envs_mod.HostVecEnv([he.HostCartPole for _ in range(8)], 4, 2, "cpu", workers=2)      # callables: built in the worker
    pids = [p.pid for p in env._procs]
    maps = open(f"/proc/{pids[0]}/maps").read()
    assert "libtorch" not in maps and "libamdhip64" not in maps and "smz_hostenv_" in maps
    env.reset()
    env.step(torch.ones(8, dtype=torch.int32))
    env.close()
    assert all(p.poll() is not None for p in env._procs) or not env._procs
    for pid in pids:
        assert not os.path.exists(f"/proc/{pid}") or open(f"/proc/{pid}/stat").read().split()[2] == "Z"


def test_sixty_four_workers_get_a_control_region_that_holds_their_done_words():
    """the per-worker done words (one 64-byte line each) outgrew the single 4 KB control page at 61 workers; the
    region is now sized from the worker count, and 64 workers (the bench / CLI default ceiling) step like the serial adapter."""
    envs_mod, he = _envs(), _he()
    assert he.ctrl_bytes(1) == 4096 and he.ctrl_bytes(60) == 4096 and he.ctrl_bytes(61) == 8192 and he.ctrl_bytes(64) == 8192 and he.ctrl_bytes(128) == 12288
    for w in (1, 60, 61, 64, 128, 500):
        assert 4 * he.done_word(w - 1) + 4 <= he.ctrl_bytes(w)
        assert he.block_layout(8, 4, np.float32, w)["action"] == he.ctrl_bytes(w)
    B = 130
    res = []
    for workers in (0, 64):
        env = envs_mod.HostVecEnv([he.HostCartPole for _ in range(B)], 4, 2, "cpu", env_seed=3, limit=6, workers=workers)
        try:
            assert len(env._procs) == workers
            res.append(_run(env, 8, 9))
        finally:
            env.close()
    assert torch.equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1:], res[1][1:]):
        for x, y in zip(a, b):
            assert (x is None and y is None) or torch.equal(x, y)


def test_tap_index_is_atens_source_index_rule():
    he = _he()
    for n_in, n_out in ((400, 98), (600, 98), (210, 98), (160, 98), (97, 98), (133, 98), (98, 98), (7, 3)):
        idx = he.tap_index(n_in, n_out).reshape(n_out, 2)
        # the same rule through torch's own CPU kernel: resizing a ramp picks exactly these source positions
        ramp = torch.arange(n_in, dtype=torch.float32).view(1, 1, 1, n_in)
        out = torch.nn.functional.interpolate(ramp, size=(1, n_out), mode="bilinear", align_corners=False).view(-1).numpy()
        scale = np.float32(n_in) / np.float32(n_out)
        src = np.maximum((np.float64(scale) * (np.arange(n_out) + 0.5) - 0.5).astype(np.float32), 0)
        l1 = np.clip(src - idx[:, 0].astype(np.float32), 0, 1)
        want = idx[:, 0] * (1 - l1) + idx[:, 1] * l1
        np.testing.assert_allclose(out, want, rtol=2e-6, atol=1e-4)
        assert (idx[:, 0] <= idx[:, 1]).all() and idx.max() == n_in - 1 or n_in > 2 * n_out


def test_frame_adapter_taps_are_the_pixels_the_resize_reads():
    he = _he()
    H, W = 40, 60
    frame = np.random.RandomState(0).randint(0, 256, (H, W, 3)).astype(np.uint8)

    class E:
        def render(self):
            return frame
    taps = he.FrameAdapter((H, W), (9, 11), upload="taps").observe(E(), None).reshape(18, 22, 3).copy()
    iy, ix = he.tap_index(H, 9), he.tap_index(W, 11)
    for oy in range(9):
        for ox in range(11):
            for r in range(2):
                for q in range(2):
                    assert np.array_equal(taps[2 * oy + r, 2 * ox + q], frame[iy[2 * oy + r], ix[2 * ox + q]])
    full = he.FrameAdapter((H, W), (9, 11), upload="frames").observe(E(), None)
    assert np.array_equal(full.reshape(H, W, 3), frame)
    assert np.array_equal(taps, frame[np.ix_(iy, ix)])          # the C gather (libsmzhost.so) == numpy's fancy indexing


def test_host_library_exports_what_its_header_declares():
    import ctypes
    import re
    header = open(os.path.join(ROOT, "include", "smz_host.h")).read()
    names = re.findall(r"^(?:int|void)\s+(smzh_\w+)\s*\(", header, re.M)
    assert "smzh_gather_taps_u8" in names and "smzh_abi_version" in names
    lib = ctypes.CDLL(os.path.join(ROOT, "stochastic-muzero_amd", "libsmzhost.so"))
    for n in names:
        getattr(lib, n)
    assert lib.smzh_abi_version() == 2
    lib.smzh_gather_taps_u8.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    out = np.zeros(12, np.uint8)
    bad = np.array([0, 5], np.int32); ok = np.array([0, 1], np.int32)
    fr = np.zeros((2, 2, 3), np.uint8)
    assert lib.smzh_gather_taps_u8(fr.ctypes.data, 2, 2, bad.ctypes.data, 2, ok.ctypes.data, 2, out.ctypes.data) == -1
    # the last pixel of the frame and of every row (the 4-byte moves must not run past either), guard bytes around the output
    fr = np.arange(5 * 7 * 3, dtype=np.uint8).reshape(5, 7, 3)
    iy, ix = np.array([4, 4, 0, 3], np.int32), np.array([6, 6, 0, 6, 5], np.int32)
    out = np.full(4 * 5 * 3 + 8, 0xEE, np.uint8)
    assert lib.smzh_gather_taps_u8(fr.ctypes.data, 5, 7, iy.ctypes.data, 4, ix.ctypes.data, 5, out[4:].ctypes.data) == 0
    assert np.array_equal(out[4:-4].reshape(4, 5, 3), fr[np.ix_(iy, ix)]) and (out[:4] == 0xEE).all() and (out[-4:] == 0xEE).all()


@pytest.mark.parametrize("on_end,limit", [("reset", 5), ("mask", 7), ("reset", 0), ("mask", 0)])
def test_batched_slice_stepping_equals_the_per_env_path_env_by_env(on_end, limit):
    """envs whose class offers make_batch (host_envs.CartPoleBatch) are stepped with array arithmetic, a
    slice at a time; the per-env loop (batch_step=False) is the checker.  Action 2 maps to an illegal env action (the
    illegal-move rule, game.py:123-131), action 3 lies outside the action map; long runs so that games end and restart."""
    envs_mod, he = _envs(), _he()
    B, T = 37, 60
    res = []
    for batch_step, workers in ((False, 0), (True, 0), (True, 3)):
        env = envs_mod.HostVecEnv([he.HostCartPole for _ in range(B)], 4, 3, "cpu", action_map=[0, 1, 7], env_seed=5, limit=limit,
                                  on_end=on_end, first_env=40, workers=workers, batch_step=batch_step)
        try:
            assert workers or (env._slice.batch is not None) == batch_step
            r = np.random.RandomState(9)
            out = [env.reset().clone()]
            for t in range(T):
                a = r.choice([0, 1, 2, 3], env.B, p=[0.46, 0.46, 0.05, 0.03]).astype(np.int32)
                obs, rew, flag = env.step(torch.from_numpy(a))
                out.append((obs.clone(), rew.clone(), flag.clone(), env.record_obs.clone(),
                            None if env.active is None else env.active.clone()))
            res.append(out)
        finally:
            env.close()
    for other in res[1:]:
        assert torch.equal(res[0][0], other[0])
        for t, (a, b) in enumerate(zip(res[0][1:], other[1:])):
            for k, (x, y) in enumerate(zip(a, b)):
                assert (x is None and y is None) or torch.equal(x, y), (t, k)
    flags = torch.stack([s[2] for s in res[0][1:]])
    rewards = torch.stack([s[1] for s in res[0][1:]])
    assert (rewards < 0).any() and ((flags == 2).any() or limit == 0) and ((flags == 1).any() or limit > 0)
    if limit == 0:
        assert torch.isinf(rewards).any()                      # unlimited game length: the reference's -inf illegal-move reward


def test_a_subclass_that_overrides_step_is_not_stepped_by_its_parents_batch():
    he = _he()
    arrays = {n: np.zeros((4,) + s, d) for n, s, d in (("action", (), np.int32), ("reward", (), np.float32), ("flag", (), np.uint8),
                                                        ("active", (), np.uint8), ("ended", (), np.uint8), ("obs", (4,), np.float32),
                                                        ("rec", (4,), np.float32))}
    mk = lambda envs, **kw: he.HostSlice(envs, 0, he.VectorAdapter(4), arrays, [0, 1], 0, 0, "reset", 0, **kw)     # noqa: E731
    assert mk([Picky() for _ in range(4)]).batch is None
    assert mk([he.HostCartPole() for _ in range(4)]).batch is not None
    assert mk([he.HostCartPole() for _ in range(3)] + [Picky()]).batch is None           # mixed classes: env by env
    assert mk([he.HostCartPole() for _ in range(4)], batch=False).batch is None
    sl = mk([he.HostCartPole() for _ in range(4)])
    sl.reset_all()
    assert all(e.state.base is sl.batch.state for e in sl.envs)                          # the env objects see the batch's state
