"""RCCL executes on the single test GPU: a fresh child process builds an "nccl" process group of world size
1 and drives the multi-GPU exchange against itself -- the first contact of gather.TrajectoryGather's nccl branch (grouped
isend / irecv on a side stream, record_stream bookkeeping), broadcast_model on device tensors and bench.py's collectives with
RCCL, before an 8-GPU box ever sees this code.  The child runs with NCCL_DEBUG=INFO; the log goes to
gpurun_out/rccl_loopback_nccl_debug.log (an excerpt is kept under profiles/)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e["OMP_NUM_THREADS"] = "1"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        e["MASTER_PORT"] = str(s.getsockname()[1])
    e["MASTER_ADDR"] = "127.0.0.1"
    return e


def _keep(name, text):
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, name), "w") as f:
            f.write(text)
    except OSError:
        pass


def test_rccl_world_of_one_runs_the_exchange_the_broadcast_and_the_bench_collectives(tmp_path):
    env = dict(_env(), NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,COLL,P2P")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_loopback_worker.py"), "--out", str(tmp_path)], env=env,
                       capture_output=True, text=True, timeout=900)
    _keep("rccl_loopback_nccl_debug.log", r.stdout[-200000:] + "\n==== stderr ====\n" + r.stderr[-200000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    res = json.load(open(os.path.join(tmp_path, "result.json")))
    assert res["backend"] == "nccl" and res["world"] == 1
    assert "NCCL INFO" in (r.stdout + r.stderr), "RCCL did not log: was the nccl backend really used?"
    _keep("rccl_loopback_result.json", json.dumps(res, indent=1))
    print(res)


def test_bench_rccl_loopback_line():
    """bench.py --rccl-loopback: the N > 1 line's code path (process group, ChunkExchange, per-rank rates, gather timing) with one
    rank over RCCL."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--rccl-loopback", "--steps", "4", "--warmup", "1", "--envs", "512",
           "--min-timed-seconds", "0.2", "--no-roofline"]
    # the default exchange is the plain gather (one grouped send / receive per block) ...
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-5000:]
    plain = json.loads(r.stdout)
    assert plain["timing"]["gather_overlap"]["mode"]["kind"] == "plain" and plain["timing"]["gather_ms_median"] > 0
    # ... and the overlapped, sliced one is a switch away
    cmd += ["--gather-mode", "overlapped"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-5000:]
    # stdout is the JSON line and nothing else (RCCL's version banner, written through C stdio, must not trail it)
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout[-1500:]
    out = json.loads(r.stdout)
    assert out["n_gpus"] == 1 and "LOOPBACK" in out["config"]["collective_backend"] and out["config"]["ranks_seen_by_collective"] == 1
    go = out["timing"]["gather_overlap"]
    assert go["mode"]["kind"] == "overlapped" and go["slices"] == 4 and go["exposed_ms_median"] is not None
    assert out["timing"]["gather_ms_median"] > 0 and len(out["per_rank_simulations_per_s"]) == 1
    assert "LOOPBACK" in out["metric"] and "cpu_baseline" not in out
    _keep("bench_rccl_loopback_small.json", json.dumps(out))
