"""The N > 1 path on CPU: world_size-2 gloo run of the trajectory gather and the env sharding rule."""
import os
import sys
from importlib import import_module

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import stochastic_muzero_amd  # noqa: F401
    g = import_module("stochastic-muzero_amd.gather")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total, T, F = 10, 3, 7
    lo, hi = g.shard_range(total, rank, world)
    full = torch.arange(T * total * F, dtype=torch.float64).reshape(T, total, F)     # what one GPU would have produced
    slab = full[:, lo:hi].contiguous()
    parts = g.gather_to_learner(slab)
    if rank == 0:
        assert parts is not None and len(parts) == world
        torch.save(torch.cat(parts, dim=1), os.path.join(out_dir, "gathered.pt"))
    else:
        assert parts is None
    dist.barrier()
    dist.destroy_process_group()


def _bcast_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import stochastic_muzero_amd  # noqa: F401
    g = import_module("stochastic-muzero_amd.gather")
    model_mod = import_module("stochastic-muzero_amd.model")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                      # every rank starts from DIFFERENT random weights
    m = model_mod.Muzero(model_structure="mlp_model", observation_space_dimensions=4, action_space_dimensions=2,
                         state_space_dimensions=7, hidden_layer_dimensions=8, number_of_hidden_layer=1, random_tag=1)
    g.broadcast_model(m, src=0)
    arrays = model_mod.mlp_arrays_from_modules(m.representation_function, m.prediction_function,
                                               m.afterstate_prediction_function, m.afterstate_dynamics_function,
                                               m.dynamics_function)
    torch.save({k: v.clone() for k, v in arrays.items()}, os.path.join(out_dir, f"w{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_weight_broadcast_world_size_2(tmp_path):
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_bcast_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    w0, w1 = (torch.load(os.path.join(tmp_path, f"w{r}.pt")) for r in (0, 1))
    torch.manual_seed(100)
    assert all(torch.equal(w0[k], w1[k]) for k in w0)
    assert any(w0[k].abs().sum() > 0 for k in w0)


def test_gather_to_learner_world_size_2(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(os.path.join(tmp_path, "gathered.pt"))
    T, total, F = 3, 10, 7
    assert torch.equal(got, torch.arange(T * total * F, dtype=torch.float64).reshape(T, total, F))


def test_shard_ranges_cover_every_env_once():
    sys.path.insert(0, ROOT)
    import stochastic_muzero_amd  # noqa: F401
    g = import_module("stochastic-muzero_amd.gather")
    for total in (1, 7, 4096, 32768, 1000):
        for world in (1, 2, 3, 4, 8):
            seen = []
            if total < world:          # no empty shards -- an engine for zero trees cannot be built
                with pytest.raises(ValueError):
                    g.shard_range(total, 0, world)
                continue
            sizes = []
            for r in range(world):
                lo, hi = g.shard_range(total, r, world)
                assert 0 <= lo < hi <= total
                seen += list(range(lo, hi))
                sizes.append(hi - lo)
            assert seen == list(range(total)) and max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        g.shard_range(8, 8, 8)
    assert g.gather_to_learner(torch.zeros(2, 2))[0].shape == (2, 2)     # no process group: identity


def _record(T, B, o, A, seed):
    r = np.random.RandomState(seed)
    d = np.zeros((T, B, o + 3 * A + 3))
    d[..., :o] = r.randn(T, B, o).astype(np.float32)
    d[..., o] = r.randn(T, B)                                   # rewards: full float64 values
    d[..., o + 1] = r.randint(0, 4, (T, B))
    d[..., o + 2:o + 2 + A] = r.rand(T, B, A)
    d[..., o + 2 + A:o + 2 + 2 * A] = np.eye(A)[r.randint(0, A, (T, B))]
    d[..., o + 2 + 2 * A] = (10 * r.randn(T, B)).astype(np.float32)
    d[..., o + 3 + 2 * A:] = r.rand(T, B, A)
    return torch.from_numpy(d)


def _tg_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import stochastic_muzero_amd  # noqa: F401
    g = import_module("stochastic-muzero_amd.gather")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, total, o, A = 9, 11, 4, 3
    full = _record(T, total, o, A, 0)
    frames = torch.arange(T * total * 6, dtype=torch.float32).reshape(T, total, 6)
    lo, hi = g.shard_range(total, rank, world)
    for compact in (True, False):
        tg = g.TrajectoryGather(o, A, slices=4, compact=compact)
        cuts = [T * k // 4 for k in range(5)]
        for k in range(4):                                      # slice after slice, as self_play_iteration plays them
            tg.start(full[cuts[k]:cuts[k + 1], lo:hi], frames[cuts[k]:cuts[k + 1], lo:hi])
        got = tg.finish()
        if rank == 0:
            assert torch.equal(got[0], full) and torch.equal(got[1], frames), compact
        else:
            assert got is None
    if rank == 0:
        torch.save(torch.ones(1), os.path.join(out_dir, "ok.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_sliced_compact_trajectory_gather_world_size_2(tmp_path):
    """gather.TrajectoryGather: slices of a chunk sent one after the other in the compact wire format (float32 for what is
    float32, float64 for rewards / policies / child visits) reassemble to the chunk bit for bit on the learner."""
    port = 27500 + (os.getpid() % 2000)
    mp.spawn(_tg_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(os.path.join(tmp_path, "ok.pt"))


def _resize_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import stochastic_muzero_amd  # noqa: F401
    g = import_module("stochastic-muzero_amd.gather")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from datetime import timedelta
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=60))
    o, A, T = 4, 2, 4
    # 9 envs (shards 4 | 5), then 8 (4 | 4): rank 0's local size does not change, rank 1's does
    for use_total in (False, True):
        for total in (9, 8, 9):
            full = _record(T, total, o, A, total)
            lo, hi = g.shard_range(total, rank, world)
            parts = g.gather_to_learner(full[:, lo:hi].contiguous(), total_envs=total if use_total else None)
            if rank == 0:
                assert torch.equal(torch.cat(parts, 1), full), (use_total, total)
        tg = g.TrajectoryGather(o, A, slices=2)                  # ONE object, chunks of different widths
        for total in (9, 8):
            full = _record(T, total, o, A, 100 + total)
            lo, hi = g.shard_range(total, rank, world)
            tg.total_envs = total if use_total else None
            tg.start(full[:2, lo:hi]); tg.start(full[2:, lo:hi])
            got = tg.finish()
            if rank == 0:
                assert torch.equal(got[0], full), (use_total, total)
    # a slab that is not this rank's shard of the stated total fails before anything is posted (on every rank alike)
    with pytest.raises(ValueError):
        g.gather_to_learner(torch.zeros(T, 3, 5, dtype=torch.float64), total_envs=10)
    if rank == 0:
        torch.save(torch.ones(1), os.path.join(out_dir, "ok.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_gathers_for_different_env_totals_in_one_process_stay_in_step(tmp_path):
    """the size exchange was cached under the LOCAL shard size, so after a 9-env gather an 8-env one
    deadlocked (rank 0: cache hit, irecv for a stale 5-env buffer; rank 1: all_gather).  Now every rank either derives the
    sizes from the job's total or enters the size exchange on every call."""
    port = 25500 + (os.getpid() % 2000)
    mp.spawn(_resize_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(os.path.join(tmp_path, "ok.pt"))


def _loopback_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import stochastic_muzero_amd  # noqa: F401
    g = import_module("stochastic-muzero_amd.gather")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    T, B, o, A = 6, 5, 4, 2
    full = _record(T, B, o, A, 3)
    frames = torch.arange(T * B * 6, dtype=torch.float32).reshape(T, B, 6)
    for compact in (True, False):
        tg = g.TrajectoryGather(o, A, slices=3, compact=compact, loopback=True, total_envs=B)
        for k in range(3):
            tg.start(full[2 * k:2 * k + 2], frames[2 * k:2 * k + 2])
        got = tg.finish()
        assert torch.equal(got[0], full) and torch.equal(got[1], frames)
        assert got[0].data_ptr() != full.data_ptr()
    plain = g.TrajectoryGather(o, A, slices=3)                   # world of one without the switch: pass-through
    plain.start(full)
    assert torch.equal(plain.finish()[0], full)
    torch.save(torch.ones(1), os.path.join(out_dir, "ok.pt"))
    dist.destroy_process_group()


def test_loopback_exchange_in_a_world_of_one(tmp_path):
    """The `loopback` switch of TrajectoryGather (the single-GPU RCCL test drives it with the nccl backend): a world of one
    still packs, 'sends', 'receives' and reassembles."""
    port = 23500 + (os.getpid() % 2000)
    mp.spawn(_loopback_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    assert os.path.exists(os.path.join(tmp_path, "ok.pt"))


class _FakeChunk:
    def __init__(self, data, obs):
        self.data, self.obs = data, obs


def _fallback_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import stochastic_muzero_amd  # noqa: F401
    g = import_module("stochastic-muzero_amd.gather")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from datetime import timedelta
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=60))
    T, total, o, A = 8, 11, 4, 3
    full = _record(T, total, o, A, 5)
    frames = torch.arange(T * total * 6, dtype=torch.float32).reshape(T, total, 6)
    lo, hi = g.shard_range(total, rank, world)
    mid = (lo + hi) // 2                                        # two env groups per rank, as bench.py's --groups
    for with_frames in (False, True):
        played = []

        def play(n, t0):                                        # "plays" rows [t0, t0 + n) of this rank's two groups
            played.append((n, t0))
            return [_FakeChunk(full[:, a:b], frames[:, a:b] if with_frames else None) for a, b in ((lo, mid), (mid, hi))]

        def rows(chunks, name, t0, t1):
            parts = [getattr(c, name) for c in chunks]
            return None if parts[0] is None else torch.cat([p[t0:t1] for p in parts], 1)

        want = (full, frames if with_frames else None)
        # the overlapped mode, undisturbed
        x = g.ChunkExchange(g.TrajectoryGather(o, A, slices=4, total_envs=total), play, rows, total_envs=total)
        _, got = x.warm_up(T)
        assert x.mode == {"kind": "overlapped"} and played == [(2, 0), (2, 2), (2, 4), (2, 6)]
        if rank == 0:
            assert torch.equal(got[0], want[0]) and (want[1] is None or torch.equal(got[1], want[1]))
        else:
            assert got is None
        # the first exchange raises at its second slice (on every rank, as a collective that cannot be built does): the warm-up
        # switches to the plain gather, replays the block whole, and every later block stays plain -- same tensors on the learner
        del played[:]
        tg = g.TrajectoryGather(o, A, slices=4, total_envs=total)
        real_start, calls, said = tg.start, [], []

        def broken_start(data, fr=None):
            calls.append(1)
            if len(calls) == 2:
                raise RuntimeError("injected: ncclCommInitRank failed")
            return real_start(data, fr)
        tg.start = broken_start
        x = g.ChunkExchange(tg, play, rows, total_envs=total, log=said.append)
        _, got = x.warm_up(T)
        assert x.mode["kind"] == "plain" and "injected" in x.mode["error"] and len(said) == 1
        assert played == [(2, 0), (2, 2), (T, 0)]
        for _ in range(2):
            if rank == 0:
                assert torch.equal(got[0], want[0]) and (want[1] is None or torch.equal(got[1], want[1])), with_frames
            else:
                assert got is None
            _, got = x.run(T)
        assert len(calls) == 2                                   # the overlapped exchange was not tried again
    # a failure in plain mode is not swallowed
    x = g.ChunkExchange(g.TrajectoryGather(o, A), lambda n, t0: (_ for _ in ()).throw(ValueError("play failed")), None, mode="plain")
    with pytest.raises(ValueError):
        x.warm_up(2)
    if rank == 0:
        torch.save(torch.ones(1), os.path.join(out_dir, "ok.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_the_plain_gather_fallback_delivers_the_same_chunk(tmp_path):
    """bench.py's fallback from the overlapped to the plain trajectory gather (gather.ChunkExchange.warm_up) is
    driven by an injected exception, 2 gloo ranks x 2 env groups, with and without image records."""
    port = 21500 + (os.getpid() % 2000)
    mp.spawn(_fallback_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(os.path.join(tmp_path, "ok.pt"))


def test_compact_wire_format_is_lossless_and_smaller():
    sys.path.insert(0, ROOT)
    import stochastic_muzero_amd  # noqa: F401
    g = import_module("stochastic-muzero_amd.gather")
    for o, A in ((4, 2), (8, 4), (0, 2)):
        d = _record(6, 5, o, A, o + A)
        n, w = g.pack_records(d, o, A)
        assert n.dtype == torch.float32 and w.dtype == torch.float64
        assert torch.equal(g.unpack_records(n, w, o, A), d)
        assert n.numel() * 4 + w.numel() * 8 < d.numel() * 8
    n, w = g.pack_records(_record(1, 1, 4, 2, 0), 4, 2)
    assert n.numel() * 4 + w.numel() * 8 == 72                   # CartPole: 72 bytes per env step instead of 104
