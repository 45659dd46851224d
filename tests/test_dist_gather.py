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
            for r in range(world):
                lo, hi = g.shard_range(total, r, world)
                assert 0 <= lo <= hi <= total
                seen += list(range(lo, hi))
            assert seen == list(range(total))
    assert g.gather_to_learner(torch.zeros(2, 2))[0].shape == (2, 2)     # no process group: identity
