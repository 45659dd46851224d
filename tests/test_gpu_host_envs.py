"""Host-resident envs on the GPU box: the parallel stepper (worker processes writing into one shared,
page-locked block) and the pipelined env groups give the chunks of the serial adapter env by env; the tap-compacted frame
upload gives the frames of the full-frame upload bit for bit."""
import os
import time
from importlib import import_module

import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


def _pkg(name):
    import stochastic_muzero_amd  # noqa: F401
    return import_module("stochastic-muzero_amd." + name)


def _model():
    return _pkg("model").Muzero.from_arrays(os.path.join(gu.GOLDEN, "weights_ckpt421.npz"))


def _mcts(B, sims=6):
    m = _pkg("mcts").BatchedMCTS(B, num_simulations=sims, discount=0.999, root_exploration_fraction=0.1, use_graph=False)
    return m


@pytest.mark.parametrize("on_end,limit", [("reset", 5), ("mask", 6)])
def test_worker_processes_and_pipelined_groups_reproduce_the_serial_adapter(on_end, limit):
    """test_host_resident_envs_through_the_pinned_memory_adapter's serial HostVecEnv is the reference point: (a) the same envs
    stepped by 3 worker processes, (b) two env groups with two workers each whose searches and host steps are interleaved by
    play_games_grouped -- every record of every env identical."""
    envs_mod, sp = _pkg("envs"), _pkg("selfplay")
    B, T = 48, 13
    heads = _model().heads("cuda:0")

    def make(lo, n, workers):
        env = envs_mod.HostVecEnv([envs_mod.HostCartPole() for _ in range(n)], 4, 2, "cuda:0", env_seed=3, limit=limit,
                                  on_end=on_end, first_env=lo, workers=workers)
        env.reset()
        return env

    def play(workers):
        env = make(0, B, workers)
        m = _mcts(B)
        m.seed(np.arange(B, dtype=np.uint64))
        out = sp.play_games(env, heads, m, 1.0, T).data.clone()
        torch.cuda.synchronize()
        env.close()
        return out
    serial, pooled = play(0), play(3)
    assert torch.equal(serial, pooled)
    groups = []
    for gi, lo in enumerate((0, B // 2)):
        gm = _mcts(B // 2)
        gm.seed(np.arange(lo, lo + B // 2, dtype=np.uint64))
        groups.append(sp.StreamGroup(make(lo, B // 2, 2), _model().heads("cuda:0", instance=gi), gm, T))
    parts = sp.play_games_grouped(groups, 1.0, T)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([p.data for p in parts], dim=1), serial)
    flags = serial[..., 5].cpu().numpy()
    assert (flags == 2).any() and ((flags == 3).any() if on_end == "mask" else (flags != 3).all())
    for g in groups:
        g.env.close()


def test_tap_compacted_upload_gives_the_full_frame_resize_bit_for_bit():
    lib_mod = _pkg("_lib")
    he = _pkg("host_envs")
    lib = lib_mod.load()
    import ctypes as C
    P = lambda x: None if x is None else C.c_void_p(x.data_ptr())
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for (H, W), (oh, ow) in (((400, 600), (98, 98)), ((210, 160), (98, 98)), ((97, 133), (98, 98)), ((64, 48), (20, 30))):
        n = 5
        frames = torch.from_numpy(np.random.RandomState(H).randint(0, 256, (n, H, W, 3)).astype(np.uint8))
        iy, ix = he.tap_index(H, oh), he.tap_index(W, ow)
        taps = torch.from_numpy(np.ascontiguousarray(frames.numpy()[:, iy][:, :, ix]))
        assert tuple(taps.shape) == (n, 2 * oh, 2 * ow, 3)
        a = torch.zeros(n, 3, oh, ow, device="cuda"); b = torch.zeros(n + 2, 3, oh, ow, device="cuda")
        lib_mod.check(lib.smz_frames_resize_u8(P(frames.cuda()), n, H, W, oh, ow, None, P(a), s))
        rows = torch.tensor([6, 0, 3, 2, 5], dtype=torch.int32, device="cuda")
        lib_mod.check(lib.smz_frames_resize_taps_u8(P(taps.cuda()), n, H, W, oh, ow, P(rows), P(b), s))
        torch.cuda.synchronize()
        assert torch.equal(a, b[rows.long()]) and (b[1] == 0).all() and (b[4] == 0).all()


def test_registered_shared_block_copies_are_asynchronous_and_exact():
    """smz_host_register / smz_copy_async: a file-backed shared mapping page-locked in place; rows written by another process
    arrive on the device unchanged."""
    lib_mod, he = _pkg("_lib"), _pkg("host_envs")
    lib = lib_mod.load()
    import ctypes as C
    blk = he.SharedBlock(1 << 22)
    base = torch.frombuffer(blk.mm, dtype=torch.uint8)
    lib_mod.check(lib.smz_host_register(C.c_void_p(base.data_ptr()), blk.nbytes))
    try:
        base.numpy()[:] = np.random.RandomState(0).randint(0, 256, blk.nbytes).astype(np.uint8)
        dev = torch.zeros(blk.nbytes, dtype=torch.uint8, device="cuda")
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        lib_mod.check(lib.smz_copy_async(C.c_void_p(dev.data_ptr()), C.c_void_p(base.data_ptr()), blk.nbytes, 1, s))
        back = torch.zeros(blk.nbytes, dtype=torch.uint8)
        torch.cuda.synchronize()
        assert torch.equal(dev.cpu(), base)
        dev += 1
        lib_mod.check(lib.smz_copy_async(C.c_void_p(base.data_ptr()), C.c_void_p(dev.data_ptr()), blk.nbytes, 0, s))
        torch.cuda.synchronize()
        assert torch.equal(dev.cpu(), base) and not torch.equal(back, base)
    finally:
        lib_mod.check(lib.smz_host_unregister(C.c_void_p(base.data_ptr())))
        blk.unlink()


def test_parallel_host_step_rate():
    """What a host step of 4096 Python CartPoles costs with the worker pool (printed; the bench line is bench.py --host-env python)."""
    envs_mod = _pkg("envs")
    B = 4096
    workers = max(2, min(32, (os.cpu_count() or 2) // 2))
    env = envs_mod.HostVecEnv([envs_mod.HostCartPole for _ in range(B)], 4, 2, "cuda:0", on_end="reset", workers=workers)
    env.reset()
    act = torch.zeros(B, dtype=torch.int32, device="cuda")
    for _ in range(5):
        env.step(act)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        env.step(act)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"host step of {B} Python CartPoles on {workers} workers: {1e3 * dt:.3f} ms ({B / dt / 1e6:.2f} M env steps/s)")
    env.close()
    assert dt < 0.05
