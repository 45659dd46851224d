"""RCCL on ONE GPU (started as a fresh child by tests/test_gpu_rccl_loopback.py, before the child has made any GPU call): an
"nccl" process group of world size 1, and through it everything the N > 1 path uses -- the sliced, side-stream trajectory
gather with the rank posting isend + irecv to ITSELF in one group (gather.TrajectoryGather(loopback=True)), vector records and
image records; broadcast_model on device tensors; bench.py's barrier / all_reduce(MAX) / all_gather_object; and a whole
self_play_iteration whose games must equal the run without a process group.  Writes result.json into --out."""
import argparse
import json
import os
import sys
from importlib import import_module

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import numpy as np
    import torch
    import torch.distributed as dist
    from datetime import timedelta
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev, timeout=timedelta(seconds=120))
    res = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    import stochastic_muzero_amd  # noqa: F401
    mcts_mod, model_mod, envs_mod, sp, g = (import_module("stochastic-muzero_amd." + m)
                                            for m in ("mcts", "model", "envs", "selfplay", "gather"))

    # (1) bench.py's collectives under nccl with one rank
    t = torch.tensor([3.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ones = torch.ones(1, dtype=torch.float64, device=dev)
    dist.all_reduce(ones)
    objs = [None]
    dist.all_gather_object(objs, {"rate": 1.5})
    dist.barrier()
    torch.cuda.synchronize(dev)
    assert float(t.item()) == 3.25 and int(ones.item()) == 1 and objs == [{"rate": 1.5}]
    res["collectives"] = "all_reduce(MAX), all_reduce(SUM), all_gather_object, barrier"

    # (2) the sliced exchange to self: vector records (compact and plain wire format) and image records
    rs = np.random.RandomState(0)
    T, B, o, A = 8, 300, 4, 2
    F = o + 3 * A + 3
    d = np.zeros((T, B, F))
    d[..., :o] = rs.randn(T, B, o).astype(np.float32)
    d[..., o] = rs.randn(T, B)
    d[..., o + 1] = rs.randint(0, 4, (T, B))
    d[..., o + 2:o + 2 + A] = rs.rand(T, B, A)
    d[..., o + 2 + A:o + 2 + 2 * A] = np.eye(A)[rs.randint(0, A, (T, B))]
    d[..., o + 2 + 2 * A] = (10 * rs.randn(T, B)).astype(np.float32)
    d[..., o + 3 + 2 * A:] = rs.rand(T, B, A)
    data = torch.from_numpy(d).to(dev)
    frames = torch.from_numpy(rs.rand(T, B, 3 * 14 * 14).astype(np.float32)).to(dev)
    for compact in (True, False):
        for with_frames in (False, True):
            tg = g.TrajectoryGather(o, A, slices=4, compact=compact, loopback=True, total_envs=B)
            for k in range(4):
                # (work on the search stream between the slices, as the real loop has: the side stream must wait for it)
                data[2 * k:2 * k + 2].mul_(1.0)
                tg.start(data[2 * k:2 * k + 2], frames[2 * k:2 * k + 2] if with_frames else None)
            got = tg.finish()
            torch.cuda.synchronize(dev)
            assert got is not None and torch.equal(got[0], data), (compact, with_frames)
            assert got[0].data_ptr() != data.data_ptr()
            assert (got[1] is None) if not with_frames else torch.equal(got[1], frames)
            assert tg.exposed_gather_ms() is not None and tg._side is not None          # the nccl branch ran: side stream, events
    res["exchange"] = "TrajectoryGather(loopback): 4 slices x {compact, plain} x {vector, vector + image} == the chunk"
    # without total_envs: the size exchange itself (an all_gather under nccl), once per chunk
    tg = g.TrajectoryGather(o, A, slices=2, loopback=True)
    tg.start(data[:4]); tg.start(data[4:])
    assert torch.equal(tg.finish()[0], data)
    # the plain gather's callable protocol at world size 1 is the identity
    assert g.gather_to_learner(data)[0] is data

    # (3) broadcast_model on device tensors
    model = model_mod.Muzero.from_arrays(os.path.join(ROOT, "tests", "golden", "weights_ckpt421.npz"))
    stale = model.heads(dev)
    before = stale.weights.clone()
    g.broadcast_model(model, src=0, device=dev, loopback=True)
    heads = model.heads(dev)
    assert heads is not stale and torch.equal(heads.weights, before)
    res["broadcast"] = "broadcast_model(loopback) on cuda tensors: weights unchanged, evaluators re-packed"

    # (4) a whole self_play_iteration through the exchange == the same iteration without one
    def iteration(gather):
        env = envs_mod.CartPoleVec(512, dev, seed=0, on_end="reset", limit=5)
        m = mcts_mod.BatchedMCTS(512, num_simulations=10, discount=0.999, root_exploration_fraction=0.1, device=0, use_graph=False)
        m.seed(np.arange(512, dtype=np.uint64))
        games, mean = sp.self_play_iteration(env, model, m, 1.0, 12, gather=gather, td_steps=5)
        return games, mean
    g0, r0 = iteration(None)
    g1, r1 = iteration(g.TrajectoryGather(4, 2, slices=3, loopback=True, total_envs=512))
    assert len(g0) == len(g1) > 512 and r0 == r1
    for x, y in zip(g0, g1):
        assert x.game_length == y.game_length
        assert np.array_equal(np.asarray(x.rewards), np.asarray(y.rewards))
        assert all(np.array_equal(u, v) for u, v in zip(x.child_visits, y.child_visits))
        assert all(np.array_equal(np.asarray(u), np.asarray(v)) for u, v in zip(x.root_values, y.root_values))
        assert all(torch.equal(u, v) for u, v in zip(x.observations, y.observations))
    res["self_play_iteration"] = f"{len(g1)} games through the loopback exchange == the games without a process group"
    try:
        import torch.cuda.nccl as nccl
        res["nccl_version"] = list(nccl.version())
    except Exception as e:                           # noqa: BLE001
        res["nccl_version"] = repr(e)
    dist.barrier()
    dist.destroy_process_group()
    with open(os.path.join(a.out, "result.json"), "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main()
