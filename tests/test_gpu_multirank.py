"""The N > 1 path on real devices: 2 rank processes (RCCL when two GPUs are visible, gloo when they share the one GPU of
the test box) run weight broadcast -> sharded self-play -> trajectory gather, and the result must be the single-rank
result env by env (shard invariance); bench.py --gpus 2 must start its own ranks and report a 2-rank line."""
import json
import os
import socket
import subprocess
import sys
from importlib import import_module

import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e["OMP_NUM_THREADS"] = "1"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return e


@pytest.mark.parametrize("overlap", [0, 3])
def test_two_ranks_reproduce_the_single_rank_self_play(tmp_path, overlap):
    """overlap = 3: gather.TrajectoryGather -- the chunk played in 3 slices, each slice's rows sent in the compact wire format
    while the next slice is searched -- must deliver the very same chunk.  97 envs: unequal shards."""
    total, steps, sims, limit = 96 + (1 if overlap else 0), 6, 8, 4
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "tests", "dist_selfplay_worker.py"), "--out", str(tmp_path),
           "--total", str(total), "--steps", str(steps), "--sims", str(sims), "--limit", str(limit), "--overlap", str(overlap)]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = torch.load(os.path.join(tmp_path, "gathered.pt"))
    assert got["world"] == 2 and tuple(got["data"].shape) == (steps, total, 13)
    # the same job on one rank, in this process
    import stochastic_muzero_amd  # noqa: F401
    mcts_mod, model_mod, envs_mod, sp = (import_module("stochastic-muzero_amd." + m) for m in ("mcts", "model", "envs", "selfplay"))
    model = model_mod.Muzero.from_arrays(os.path.join(gu.GOLDEN, "weights_ckpt421.npz"))
    heads = model.heads("cuda:0")
    assert torch.equal(heads.weights.cpu(), got["weights"])            # the broadcast weights, as packed on rank 0
    env = envs_mod.CartPoleVec(total, "cuda:0", seed=0, on_end="reset", limit=limit)
    env.reset()
    m = mcts_mod.BatchedMCTS(total, num_simulations=sims, discount=0.999, root_exploration_fraction=0.1, use_graph=False)
    m.seed(np.arange(total, dtype=np.uint64))
    chunk = sp.play_games(env, heads, m, 1.0, steps)
    torch.cuda.synchronize()
    assert torch.equal(chunk.data.cpu(), got["data"]), f"shard invariance broken (backend {got['backend']})"
    assert (got["data"][..., 5] == 2).any()                             # games ended and restarted inside the chunk
    print("2-rank run over", got["backend"])


def test_c5_shape_eight_ranks_of_4096_envs_x_100_simulations_equal_one_32768_env_run(tmp_path):
    """BASELINE configs[4] at its real shape, functionally (VERDICT r2 #1a): 8 rank processes -- every one a fresh child
    that owns a 4096-env shard x 100 simulations (the production single-launch kernel, one launch per env step) -- play 4
    env steps with restarting games and gather their chunks to rank 0 (RCCL when 8 GPUs are visible; gloo through the host
    when the ranks share the test box's one GPU).  The gathered [4][32768][13] chunk must equal, env by env and bit for bit,
    a single-process run of all 32 768 envs (step-wise kernels there: 32 768 trees are beyond the single launch's range, and
    the two paths are bit-identical).  shard_range at world 8, the 8-way concatenation order, 8 engines on one device."""
    total, steps, sims, limit, world = 32768, 4, 100, 3, 8
    # (round 4: through the overlapped, sliced exchange -- gather.TrajectoryGather, 2 slices of 2 steps)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "tests", "dist_selfplay_worker.py"), "--out", str(tmp_path),
           "--total", str(total), "--steps", str(steps), "--sims", str(sims), "--limit", str(limit), "--overlap", "2"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = torch.load(os.path.join(tmp_path, "gathered.pt"))
    assert got["world"] == world and tuple(got["data"].shape) == (steps, total, 13)
    assert got["ranks_seen_by_collective"] == world and got["single_launch"] == [True] * world
    import stochastic_muzero_amd  # noqa: F401
    mcts_mod, model_mod, envs_mod, sp = (import_module("stochastic-muzero_amd." + m) for m in ("mcts", "model", "envs", "selfplay"))
    model = model_mod.Muzero.from_arrays(os.path.join(gu.GOLDEN, "weights_ckpt421.npz"))
    heads = model.heads("cuda:0")
    env = envs_mod.CartPoleVec(total, "cuda:0", seed=0, on_end="reset", limit=limit)
    env.reset()
    m = mcts_mod.BatchedMCTS(total, num_simulations=sims, discount=0.999, root_exploration_fraction=0.1, use_graph=False)
    m.seed(np.arange(total, dtype=np.uint64))
    chunk = sp.play_games(env, heads, m, 1.0, steps)
    torch.cuda.synchronize()
    assert m._single is not True                                         # the step-wise kernels served the 32 768-tree run
    want = chunk.data.cpu()
    same = (want == got["data"]).all(dim=2).all(dim=0)
    assert bool(same.all()), f"{int((~same).sum())} of {total} envs differ (first: {int((~same).nonzero()[0])}); backend {got['backend']}"
    assert (got["data"][..., 5] == 2).any() and (got["data"][..., 5] == 0).any()
    print(f"C5 shape: {world} ranks x 4096 envs x {sims} sims over {got['backend']} == one 32768-env run, env by env")


def test_bench_starts_its_own_ranks():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--envs", "256",
           "--no-cpu-baseline", "--min-timed-seconds", "0.05", "--gather-mode", "overlapped"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2 and len(out["per_rank_simulations_per_s"]) == 2
    assert out["value"] > 0 and out["scaling"] == "weak" and out["timing"]["blocks"] >= 1
    assert out["roofline"]["frac"] > 0 and "cpu_baseline" not in out
    assert out["timing"]["gather_overlap"]["slices"] == 4 and out["timing"]["gather_ms_median"] > 0
    assert out["timing"]["gather_overlap"]["mode"]["kind"] == "overlapped"
    assert len(r.stdout.strip().splitlines()) == 1                      # stdout is the JSON line and nothing else
    # the default exchange (one grouped send / receive per block, stream-ordered behind the block's last search)
    cmd = [c for c in cmd if c not in ("--gather-mode", "overlapped")]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads(r.stdout)
    assert out["n_gpus"] == 2 and out["timing"]["gather_overlap"]["mode"]["kind"] == "plain" and out["timing"]["gather_ms_median"] > 0
    assert out["config"]["ranks_seen_by_collective"] == 2 and len(out["per_rank_simulations_per_s"]) == 2
    # a launcher / flag mismatch is an error, not a silent single-GPU run
    bad = dict(_env(), WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=bad, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)


@pytest.mark.parametrize("launcher", ["bench", "torchrun"])
def test_the_drivers_eight_rank_command_yields_one_eight_rank_line(launcher):
    """Scale-run readiness (no 8-GPU node has seen this code): the driver's command SHAPE for the scaling
    bench -- `python bench.py --gpus 8 --steps K --warmup W`, and the same under `python -m torch.distributed.run --nnodes=1
    --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...` -- on the test box (the eight ranks share
    its GPU, so the exchange goes through gloo and the line says so).  One JSON line on stdout, eight ranks seen by the collective
    itself, eight per-rank rates, weak scaling, no cpu_baseline (N = 1 only), no `also` blocks.  The launcher starts the ranks as
    fresh children before anything has touched the GPU (never an exec from a process that has)."""
    import socket
    tail = ["--gpus", "8", "--steps", "2", "--warmup", "1", "--envs", "256", "--min-timed-seconds", "0.05"]
    if launcher == "bench":
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + tail
    else:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py")] + tail
    r = subprocess.run(cmd, env=dict(_env(), OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["ranks"] == 8 and out["config"]["ranks_seen_by_collective"] == 8
    assert len(out["per_rank_simulations_per_s"]) == 8 and all(v > 0 for v in out["per_rank_simulations_per_s"])
    assert out["scaling"] == "weak" and out["value"] > 0 and out["steps"] == 2 and out["warmup"] == 1
    assert "cpu_baseline" not in out and "also" not in out
    assert out["unit"] == "simulations/s" and out["higher_is_better"] is True and out["vs_baseline"] is None
    assert out["config"]["envs_per_gpu"] == 256 and "x8" in out["config"]["parallelism"]
    assert out["timing"]["gather_overlap"]["mode"]["kind"] == "plain" and out["timing"]["gather_ms_median"] > 0
