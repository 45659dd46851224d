#!/usr/bin/env python3
"""bench.py -- MCTS simulations/sec of the batched self-play engine (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

N > 1 is served either way the driver may start it: under `python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N ...` (RANK / WORLD_SIZE in the environment), or as plain `python bench.py --gpus N`, in which case this process
touches no GPU at all: it starts the N rank processes through torch.distributed.run as a child and exits with its status.
A WORLD_SIZE that disagrees with --gpus is an error, not a silent single-GPU run.

A "step" is ONE env step of every environment of this rank: root inference, num_simulations x (select -> six-head
evaluation -> expand + backup), action selection, env step and trajectory record -- the loop body of
self_play.py:79-94 for 4096 envs at once (two launches for the MLP workloads: smz_search_mlp_act and the env step +
record).  Workload at N = 1 = BASELINE.json configs[1]: CartPole-v1 shaped synthetic episodes, checkpoint-421 MLP heads
(S31/H64/L0), 4096 envs x 50 simulations, per-tree numpy-legacy MT19937 streams (parity mode, the mode the parity tests
pin).  Inputs (weights, env state, trees) are resident in HBM before the timed region.  N > 1: each rank owns 4096 envs
(weak scaling) and the finished K-step trajectory chunk is gathered to rank 0 over RCCL inside the timed region.

Timing: after W warm-up steps, R blocks of EXACTLY K steps are timed, each bracketed by barrier + torch.cuda.synchronize
on both sides, the block time being the MAX over ranks; R is chosen so that the timed region is about 0.5 s or more
(one 10 ms block is at the mercy of a host hiccup).  `value` and `ms_per_step` come from the MEDIAN block; the spread
is in `timing`.

The JSON line also carries, for every workload,
  roofline     -- the dominant kernel's algorithmic bytes (SURVEY.md 8d formula evaluated on this run's own level
                  histogram) / its mean launch duration measured with events on the launching stream;
  cpu_baseline -- the CPU oracle (oracle/smz_oracle.c with plain-C heads, one game per thread; for the vision family the
                  oracle tree driven by torch-CPU batch-1 heads, the reference's own shape) timed on this box's host cores
                  on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# (the host driver of this pool only supports dmabuf IPC: without this RCCL's hipIpcGetMemHandle fails; the launcher exports it,
#  a rank started by someone else's launcher gets it here, before anything touches the GPU)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (weights fixture, env kind, obs, A, K, sims)
    "cartpole_mlp_4096x50": dict(weights="weights_ckpt421.npz", env="cartpole", obs=4, A=2, K=2, sims=50, envs=4096),
    "lunarlander_mlp_4096x50": dict(weights="weights_lunar_L0.npz", env="synthetic", obs=8, A=4, K=2, sims=50, envs=4096),
    # SURVEY 8d(3)'s stress variant: every expansion keeps all four actions (N = 1 + 4 + 50 x 4 = 205 nodes per tree)
    "lunarlander_mlp_4096x50_K4": dict(weights="weights_lunar_L0.npz", env="synthetic", obs=8, A=4, K=4, sims=50, envs=4096),
    "cartpole_mlp_4096x100": dict(weights="weights_ckpt421.npz", env="cartpole", obs=4, A=2, K=2, sims=100, envs=4096),
    # the reference's ResNet-v2 vision family (random init, L=1), 98x98x3 frames, hidden 3x7x7; heads =
    # smz_vision_initial / smz_vision_recurrent between the HIP tree kernels, one HIP graph per env step
    # (--heads torch: the same modules through torch-ROCm, for comparison)
    "vision_resnet_1024x50": dict(weights="visionnet_L1_seed0.npz", env="image", obs=3 * 98 * 98, A=2, K=2, sims=50, envs=1024),
}


def _strip_comments(text):
    """C / C++ source without comments and with runs of white space collapsed (string and character literals kept as they are)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c in "\"'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    return " ".join("".join(out).split())


def kernel_source_sha16():
    """sha256 (first 16 hex digits) of the kernel sources + the ABI header, comments and white space aside: counter files under
    profiles/ carry the value they were measured with, so a later change of the kernel CODE drops stale `roofline.traffic`
    figures by itself (and a corrected comment does not)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "stochastic-muzero_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.hpp")) + [os.path.join(ROOT, "include", "smz.h")]):
        h.update(os.path.basename(f).encode())
        h.update(_strip_comments(open(f, encoding="utf-8").read()).encode())
    return h.hexdigest()[:16]


def find_traffic(workload, kernel):
    """The committed TCC counter file (profiles/*traffic*.json) measured on exactly this kernel instantiation, this workload
    and these kernel sources -- or (None, why not)."""
    import glob
    sha = kernel_source_sha16()
    seen = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json")), reverse=True):
        try:
            tj = json.load(open(f))
        except Exception:
            continue
        if tj.get("workload") != workload or tj.get("kernel") != kernel:
            continue
        if tj.get("source_sha16") != sha:
            seen.append(os.path.basename(f))
            continue
        return tj, os.path.basename(f)
    return None, ("no counter file for kernel %s on workload %s with the current kernel sources (sha %s)%s" %
                  (kernel, workload, sha, "; stale: " + ", ".join(seen[:3]) if seen else ""))


def find_pmc(workload, kernel):
    """The committed SQ counter file (profiles/*pmc_k_search*.json: tools/pmc_search.sh, separate --pmc passes) of exactly this
    kernel instantiation, workload and kernel sources -- or (None, why not).  Same rule as find_traffic."""
    import glob
    sha = kernel_source_sha16()
    seen = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_k_search*.json")), reverse=True):
        try:
            pj = json.load(open(f))
        except Exception:
            continue
        if pj.get("workload") != workload or pj.get("kernel") != kernel or "per_launch_mean" not in pj:
            continue
        if pj.get("source_sha16") != sha:
            seen.append(os.path.basename(f))
            continue
        return pj, os.path.basename(f)
    return None, ("no SQ counter file for kernel %s on workload %s with the current kernel sources (sha %s)%s" %
                  (kernel, workload, sha, "; stale: " + ", ".join(seen[:3]) if seen else ""))


def valu_issue_bound(pj, fname, sims):
    """VERDICT r5 next #4a: the hardware-anchored bound that applies to the single-launch MLP search -- vector-instruction issue.
    From the SQ counters of the committed pass: a wavefront has a VALU instruction executing during ACTIVE_INST_VALU / WAVE_CYCLES
    of its life; with w wavefronts per SIMD (SQ_WAVES / 1024 SIMDs) the SIMD's vector unit is busy w times that; lanes enabled =
    THREAD_CYCLES_VALU / (64 x ACTIVE_INST_VALU)."""
    c = pj["per_launch_mean"]
    waves_per_simd = c["SQ_WAVES"] / 1024.0
    per_wave = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]
    busy = per_wave * waves_per_simd
    return {"bound": "valu_issue", "unit": "fraction of SIMD cycles with a vector instruction executing", "achieved": busy, "peak": 1.0,
            "frac": busy, "per_wave_frac": per_wave, "waves_per_simd": waves_per_simd,
            "lanes_enabled": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]),
            "valu_insts_per_wave_round": c["SQ_INSTS_VALU"] / c["SQ_WAVES"] / max(1, sims),
            "wait_any_frac": c.get("SQ_WAIT_ANY", float("nan")) / c["SQ_WAVE_CYCLES"],
            "source": fname,
            "note": "SQ counters of a committed rocprofv3 --pmc pass on this kernel instantiation, workload and kernel sources; the "
                    "launch is a dependent chain per wavefront whose vector unit is this busy -- the headroom a perfect overlap of "
                    "the two wavefronts per SIMD could still claim is 1 - frac"}


def algorithmic_bytes(stats, A, K, S, launches):
    """bytes per simulation per tree for select (K2) and expand+backup (K5), evaluated on the
    measured level histogram.  Returns (k2, k5) in bytes per tree per simulation."""
    desc = max(1, stats["descents"])
    dec = stats["decision_levels"] / desc          # decision-flag levels per descent (includes the root level)
    ch = stats["chance_levels"] / desc
    depth = dec + ch
    k2 = (12 + 16 * A + 4 * A) + max(0.0, dec - 1) * (12 + 16 * K) + ch * (12 + 4 * K) + 8 + 4 * depth + (12 + 4 * S)
    k5 = K * 20 + 4 * S + 4 + (depth + 1) * 20 + 16
    return k2, k5, depth


def host_cores():
    """Cores this process can actually USE (affinity mask capped by the cgroup CPU quota): host_envs.usable_cores."""
    from importlib import import_module
    import stochastic_muzero_amd  # noqa: F401
    return import_module("stochastic-muzero_amd.host_envs").usable_cores()


def cpu_baseline_mlp(wl, weights_path, seconds_target=15.0):
    """The CPU oracle on all host cores, on a bounded sample of the same workload (about 10-30 s of CPU work)."""
    import numpy as np
    import orc
    w = orc.MlpWeights.from_npz(weights_path)
    cores = host_cores()
    cfg = orc.make_cfg(wl["A"], wl["K"], w.dims["S"], wl["sims"], discount=0.999, alpha=0.25, frac=0.1)
    rs = np.random.RandomState(0)
    steps = 64                                                 # the synthetic episode length of the workload
    cart = wl["env"] == "cartpole"

    def run(n_env, n_steps, threads):
        seeds = np.arange(n_env, dtype=np.uint32)
        t0 = time.perf_counter()
        if cart:
            out = orc.selfplay_cartpole(cfg, w, rs.uniform(-0.05, 0.05, (n_env, 4)), seeds, n_steps, temperature=1.0,
                                        train=True, threads=threads, record=False)
        else:
            out = orc.selfplay_observations(cfg, w, rs.standard_normal((n_env, n_steps, wl["obs"])).astype(np.float32), seeds,
                                            temperature=1.0, train=True, threads=threads, record=False)
        return out["simulations"], time.perf_counter() - t0
    sims, dt = run(4, steps, 1)
    single = sims / dt
    sims, dt = run(2 * cores, steps, cores)                    # calibration pass (also warms the thread pool)
    rate = sims / dt
    n_env = int(max(cores, rate * seconds_target / (steps * wl["sims"])))
    n_env -= n_env % cores
    sims, dt = run(n_env, steps, cores)
    what = "CartPole" if cart else f"N(0,1)-observation (obs {wl['obs']}, {wl['A']} actions)"
    return dict(value=sims / dt, unit="simulations/s", cores=cores, kind="port",
                sample=f"{n_env} envs x {steps} steps x {wl['sims']} sims of the same {what} workload, C oracle "
                       f"(oracle/smz_oracle.c) with plain-C MLP heads, {cores} threads (one game per thread), "
                       f"{dt:.1f} s wall = {dt * cores:.0f} core-seconds; one thread alone: {single:.0f} simulations/s "
                       f"(os.cpu_count() = {os.cpu_count()}; cores = affinity mask capped by the cgroup CPU quota)")


def _vision_cpu_worker(weights_path, A, K, sims, seconds, seed):
    """One process of the vision cpu_baseline: the oracle's tree arithmetic (C) driven by the model's batch-1 torch-CPU inference
    functions -- the call shape of the reference's own search (muzero_model.py:802-909) -- for `seconds`; returns simulations done."""
    import numpy as np
    import torch
    import orc
    from importlib import import_module
    import stochastic_muzero_amd  # noqa: F401
    torch.set_num_threads(1)
    model = import_module("stochastic-muzero_amd.model").Muzero.from_state_dicts(weights_path)
    cfg = orc.make_cfg(A, K, 147, sims, discount=0.999, alpha=0.25, frac=0.1)
    rs = np.random.RandomState(seed)

    def search(tree, frame):
        h = model.representation_function_inference(frame)
        pol, _ = model.prediction_function_inference(h)
        tree.root_init(np.asarray(pol, np.float32).reshape(-1), hidden=h.numpy().reshape(-1), train=True)
        for _ in range(sims):
            leaf, parent, act, flag, ph = tree.select(want_hidden=True)
            ph = torch.from_numpy(ph[:147].reshape(1, 3, 7, 7))
            if flag:
                reward, h2 = model.dynamics_function_inference(ph, act)
                pol, val = model.prediction_function_inference(h2)
            else:
                reward, h2 = 0.0, model.afterstate_dynamics_function_inference(ph, act)
                pol, val = model.afterstate_prediction_function_inference(h2)
            tree.expand_backup(np.asarray(pol, np.float32).reshape(-1), float(val), reward=float(reward), hidden=h2.numpy().reshape(-1))
    tree = orc.Tree(cfg)
    tree.seed(seed)
    search(tree, torch.from_numpy(rs.rand(1, 3, 98, 98).astype(np.float32)))          # (warm-up: lazy initialisation is not work)
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < seconds:
        search(tree, torch.from_numpy(rs.rand(1, 3, 98, 98).astype(np.float32)))
        done += sims
    return done, time.perf_counter() - t0


def cpu_baseline_vision(wl, weights_path, seconds_target=12.0):
    """Vision family on ALL usable host cores (SURVEY 8d's shape: one game per worker process, the reference's Ray fan-out
    without Ray): `cores` child processes -- started before this process touches the GPU, torch-CPU only (the GPU is hidden from
    them) -- each running _vision_cpu_worker for the same wall-clock window; the figure is the sum."""
    cores = host_cores()
    code = ("import sys, json; sys.path[:0] = [%r, %r]; import bench; "
            "print(json.dumps(bench._vision_cpu_worker(%r, %d, %d, %d, %f, int(sys.argv[1]))))"
            % (ROOT, os.path.join(ROOT, "oracle"), weights_path, wl["A"], wl["K"], wl["sims"], seconds_target))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, "-c", code, str(i)], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
             for i in range(cores)]
    outs = []
    for p_ in procs:
        o, _ = p_.communicate(timeout=seconds_target * 6 + 300)
        lines = [ln for ln in o.splitlines() if ln.startswith("[")]
        if p_.returncode == 0 and lines:
            outs.append(json.loads(lines[-1]))
    done = sum(o[0] for o in outs)
    dt = max((o[1] for o in outs), default=float("nan"))
    one = max((o[0] / o[1] for o in outs), default=float("nan"))
    return dict(value=done / dt if outs else float("nan"), unit="simulations/s", cores=len(outs), kind="port",
                sample=f"{done // wl['sims']} searches x {wl['sims']} sims on 98x98x3 frames in {dt:.1f} s: oracle tree (oracle/smz_oracle.c) driven by "
                       f"batch-1 torch-CPU heads of the same ResNet-v2 weights, {len(outs)} worker processes x 1 torch thread (the shape of the "
                       f"reference's Ray fan-out: one game per worker) on the {cores} cores the cgroup quota grants; fastest single "
                       f"process {one:.0f} simulations/s")


class ReplaySink:
    """What ReplayBuffer.save_game does with a game (replay_buffer.py:109-137), restated as the bench's sink: evict beyond the
    window, make_priority(td_steps) -> per-position and per-game priorities, append, count positions; games that were not
    reanalysed would also go to the reanalyse stack (empty here).  Left out: the re-normalisation of soft_prio_game after every
    save (np.array(prio_game) / sum: O(buffer) per game, the reference's own quadratic cost -- not this engine's)."""

    def __init__(self, td_steps=50, window_size=10 ** 9):
        self.td_steps, self.window_size = td_steps, window_size
        self.buffer, self.prio_position, self.prio_game, self.positions = [], [], [], 0

    def save_game(self, game):
        if len(self.buffer) > self.window_size:
            self.positions -= self.buffer.pop(0).game_length
            self.prio_position.pop(0)
            self.prio_game.pop(0)
        pos, top = game.make_priority(self.td_steps)
        self.prio_position.append(pos)
        self.prio_game.append(top)
        self.buffer.append(game)
        self.positions += game.game_length
        if not game.reanalyzed:
            pass                                           # reanalyse_buffer_save_game: no reanalyse buffers in the bench

    def clear(self):
        self.buffer, self.prio_position, self.prio_game, self.positions = [], [], [], 0


def self_launch(args, argv):
    """--gpus N without a launcher: start the N ranks as a CHILD (torch.distributed.run) before anything here has
    touched a GPU; this process only waits and passes the status on."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    rc = subprocess.call(cmd, env=env)
    if rc != 0 and args.gather_mode == "overlapped":
        # The overlapped exchange's failure modes under RCCL are mostly not Python exceptions (an abort, or a hang that the
        # process-group timeout turns into one): nothing inside a rank can fall back then, but this launcher can -- one more
        # run with the plain, synchronous gather, and the line it prints says so (timing.gather_overlap.mode.kind = "plain").
        print(f"bench.py: the {args.gpus}-rank run exited with status {rc}; running again with --gather-mode plain", file=sys.stderr)
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            cmd[cmd.index("--master-port") + 1] = str(s.getsockname()[1])
        rc = subprocess.call(cmd + ["--gather-mode", "plain"], env=env)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="env steps per timed block (default 16; 64 with --end-to-end)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cartpole_mlp_4096x50", choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (default: the workload's)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--temperature", type=float, default=1.0)
    ap.add_argument("--heads", default="auto", choices=["auto", "hip", "torch"],
                    help="hip: fused LDS-resident HIP heads kernel; torch: torch-ROCm GEMMs + HIP epilogues")
    ap.add_argument("--stepwise", action="store_true", help="never use the single-launch search kernel")
    ap.add_argument("--groups", type=int, default=int(os.environ.get("SMZ_STREAM_GROUPS", "0")),
                    help="independent env groups per GPU, each on its own HIP stream (0 = 2 groups from 262 144 envs on -- "
                         "one group's tree kernel overlaps the other's network kernel: +6..14 %% measured -- else 1)")
    ap.add_argument("--min-timed-seconds", type=float, default=10.0,
                    help="repeat the K-step block until this much is timed (default 10 s: ~1400 blocks of the headline workload, so "
                         "that the timed region is the larger part of the command's run time)")
    ap.add_argument("--max-blocks", type=int, default=4000)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=5.0, help="wall-clock budget of the cpu_baseline sample (all host cores)")
    ap.add_argument("--end-to-end", action="store_true",
                    help="time selfplay.self_play_iteration, the function learning_cycle calls (self_play.py:245-271): K env steps, the "
                         "chunk's transfer to the host, Game records (selfplay.chunk_to_records) and replay_buffer.save_game of every "
                         "game into a sink that does ReplayBuffer.save_game's per-game work -- its own labelled line, never the headline")
    ap.add_argument("--rng", default="auto", choices=["auto", "mt19937", "philox"],
                    help="mt19937: per-tree numpy-legacy streams (parity mode, the headline); philox: counter-based "
                         "streams (throughput mode: same distributions, different numbers) -- reported as its own workload; "
                         "auto (default): mt19937 up to the single-launch crossover (17 408 envs per GPU: every BASELINE config), "
                         "philox above it, where the step-wise tree kernel is bandwidth-bound (VERDICT r5 next #1c)")
    ap.add_argument("--host-env", nargs="?", const="python", default=None, choices=["python", "native"],
                    help="cartpole workloads: step the envs on the HOST -- the PCIe-inclusive rate of the boundary's "
                         "host-buffer variant: 'python' = envs.HostVecEnv over Python CartPoles (measures the Python), 'native' = "
                         "envs.HostCartPoleVec (compiled host step, smz_host_cartpole_step)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="--end-to-end: iterations per timed block through selfplay.self_play_iterations (the search of iteration k + 1 is "
                         "enqueued before the host builds and stores iteration k's games); 1 = one synchronous self_play_iteration per block")
    ap.add_argument("--learning-cycle", action="store_true",
                    help="--end-to-end through selfplay.learning_cycle itself (the reference's loop, self_play.py:168-306): "
                         "--pipeline iterations per timed block with number_of_training_before_self_play = 0 and a model whose "
                         "save_model is a no-op; --pipeline 1 = the synchronous loop")
    ap.add_argument("--per-env-step", action="store_true",
                    help="--host-env python: step every env with its own Python call (HostSlice's per-env loop, the checker of the "
                         "batched-slice path) instead of one array step per worker slice (host_envs.CartPoleBatch)")
    ap.add_argument("--host-workers", type=int, default=None,
                    help="--host-env python: env worker processes per env group (default min(64, 3 x usable host cores); "
                         "0 = step the envs serially in this process)")
    ap.add_argument("--gather-slices", type=int, default=4,
                    help="N > 1: parts a K-step block's trajectory gather is cut into (each part's transfer overlaps the next part's search)")
    ap.add_argument("--gather-mode", default=os.environ.get("SMZ_GATHER_MODE", "plain"), choices=["overlapped", "plain"],
                    help="N > 1: plain (default) = one grouped send / receive of the chunk per block, stream-ordered behind the block's last "
                         "search; overlapped = sliced side-stream exchange in the compact wire format (gather.TrajectoryGather).  Measured "
                         "over RCCL with --rccl-loopback (profiles/r05_c_loopback_gather_sweep.txt): plain 469 M, overlapped 458 / 452 / 447 M "
                         "with 1 / 2 / 4 slices against 473 M without a process group -- a 16-step chunk is 4.7-6.8 MB per rank, its transfer "
                         "(0.03 ms to self) is far below what the side stream's packing kernels cost the latency-bound search.  "
                         "`python bench.py --gpus N --gather-mode overlapped` re-runs the ranks with plain if the overlapped run dies")
    ap.add_argument("--rccl-loopback", action="store_true",
                    help="ONE rank, but through everything the N > 1 line uses: an nccl (RCCL) process group of world size 1, the "
                         "sliced trajectory gather with the rank sending to and receiving from itself, barrier / all_reduce / "
                         "all_gather_object -- the multi-GPU code path on a single-GPU box (its own labelled line)")
    ap.add_argument("--dist-timeout-seconds", type=float, default=300.0,
                    help="process-group timeout: a collective that hangs ends as an error instead of stalling the run")
    ap.add_argument("--frame-upload", default="taps", choices=["taps", "frames"],
                    help="--host-env on the vision workload: upload only the pixels the 98x98 resize reads (taps) or whole frames")
    ap.add_argument("--also-seconds", type=float, default=2.0,
                    help="headline run only (N = 1, default workload): after the headline's timed region, each of the other single-GPU "
                         "BASELINE configs -- lunarlander_mlp_4096x50 (configs[2]), its K = 4 stress variant, cartpole_mlp_4096x100 "
                         "(configs[4]'s per-rank shard) and vision_resnet_1024x50 (configs[3]) -- is timed for this long with the same "
                         "block rule and appended under the `also` key (value, ms_per_step, kernel, roofline fraction); 0 = off.  The "
                         "headline fields are not touched")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 64 if args.end_to_end else 16

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # The job's stdout carries the JSON line and nothing else: RCCL writes a version banner to stdout through C stdio (seen on the
    # loopback run, flushed at process exit BEHIND the line).  Every rank points fd 1 at stderr now; rank 0 keeps a private
    # duplicate of the real stdout for the line.
    sys.stdout.flush()
    json_fd = os.dup(1) if rank == 0 else None
    os.dup2(2, 1)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}.  Run `python bench.py --gpus N` on its own "
                         f"(it starts the N ranks itself) or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`.")

    import numpy as np
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the search engine has no CPU fallback")
    n_dev = torch.cuda.device_count()
    shared_gpu = world > n_dev                                  # functional runs: several ranks on one GPU
    local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    from datetime import timedelta
    backend = None
    multi = world > 1 or args.rccl_loopback          # the code path of the N > 1 line (a loopback run takes it with one rank)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            if "MASTER_PORT" not in os.environ:
                with socket.socket() as s_:
                    s_.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # "nccl" is RCCL on ROCm; it cannot place two ranks on one device, so a run with more ranks than GPUs (the
        # 1-GPU functional test of the N > 1 path) exchanges through gloo and says so in its JSON line
        backend = os.environ.get("SMZ_DIST_BACKEND", "gloo" if shared_gpu else "nccl")
        tmo = timedelta(seconds=args.dist_timeout_seconds)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)

    import stochastic_muzero_amd as smz  # noqa: F401
    from importlib import import_module
    mcts_mod = import_module("stochastic-muzero_amd.mcts")
    model_mod = import_module("stochastic-muzero_amd.model")
    envs_mod = import_module("stochastic-muzero_amd.envs")
    sp = import_module("stochastic-muzero_amd.selfplay")
    gather_mod = import_module("stochastic-muzero_amd.gather")

    wl = dict(WORKLOADS[args.workload])
    B = args.envs or wl["envs"]
    if args.rng == "auto":
        args.rng = "philox" if mcts_mod.resolve_rng_mode("auto", B) == smz._lib.RNG_PHILOX else "mt19937"
    wpath = os.path.join(ROOT, "tests", "golden", wl["weights"])
    if wl["env"] == "image":
        model = model_mod.Muzero.from_state_dicts(wpath)
    else:
        model = model_mod.Muzero.from_arrays(wpath)      # trained ckpt-421 weights exported as plain arrays
    total = B * world
    lo = rank * B
    # the CPU leg runs FIRST (rank 0, N = 1 only, bounded wall time): the GPU part then fills the rest of the command's run time
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.rccl_loopback:
        cpu_baseline = (cpu_baseline_vision(wl, wpath, args.cpu_baseline_seconds) if wl["env"] == "image"
                        else cpu_baseline_mlp(wl, wpath, args.cpu_baseline_seconds))
    T = max(args.steps, args.warmup, 1)
    # host envs behind Python (--host-env python): two env groups, so that one group's search runs while the other's envs step
    # (CartPole envs stepped slice-wise: a worker's step is ~40 numpy operations, the hand-off dominates -- one group, few workers:
    #  measured on the 16-CPU-quota box 8 / 16 workers x 1 group 308 / 312 M, x 2 groups 224 M, in-process 212 M simulations/s)
    batched_slices = args.host_env == "python" and wl["env"] == "cartpole" and not args.per_env_step
    G = args.groups if args.groups > 0 else (2 if ((B >= 262144 or (args.host_env == "python" and not batched_slices)) and B % 2 == 0) else 1)
    host_workers = 0
    if batched_slices:
        host_workers = args.host_workers if args.host_workers is not None else min(16, max(1, host_cores()))
    elif args.host_env == "python":
        # worker processes PER ENV GROUP: idle workers sleep in the kernel (futex), so while one group's envs step, the other
        # group's workers cost nothing and every group may use all usable cores; 3 x the cores hides the wake-up latencies
        # (measured on the 16-CPU-quota bench box: 14 workers 76 M, 16 113 M, 24 135 M simulations/s with one group)
        # (rendering envs keep a core busy for ~130 us per step: there 1.5 x the cores is the measured optimum -- 7.5 M against 5.9 M
        #  simulations/s with 3 x on the 16-CPU-quota box; 4 us CartPole steps want the 3 x: 182-200 M against 161 M)
        per_core = 1.5 if wl["env"] == "image" else 3
        host_workers = args.host_workers if args.host_workers is not None else min(64, max(1, int(per_core * host_cores())))
    assert B % G == 0, "--groups must divide the env count"
    Bg = B // G
    groups = []
    for gi in range(G):
        glo = lo + gi * Bg
        if wl["env"] == "cartpole" and args.host_env == "native":
            env = envs_mod.HostCartPoleVec(Bg, dev, seed=0, first_env=glo)
        elif wl["env"] == "cartpole" and args.host_env:
            env = envs_mod.HostVecEnv([envs_mod.HostCartPole for _ in range(Bg)], 4, 2, dev, env_seed=0, limit=0,
                                      on_end="reset", first_env=glo, workers=host_workers, batch_step=not args.per_env_step)
        elif wl["env"] == "cartpole":
            env = envs_mod.CartPoleVec(Bg, dev, seed=0, first_env=glo, total_envs=total)
        elif wl["env"] == "image" and args.host_env:
            # host envs observed through rendered 400x600x3 uint8 frames (CartPole-v1's render size), uploaded
            # through pinned memory and resized to 98x98 on the engine's stream (smz_frames_resize_u8)
            env = envs_mod.HostImageVecEnv([envs_mod.HostCartPoleRender((400, 600)) for _ in range(Bg)], (400, 600), wl["A"], dev,
                                           env_seed=0, limit=0, on_end="reset", first_env=glo, workers=host_workers,
                                           upload=args.frame_upload)
        elif wl["env"] == "image":
            env = envs_mod.ImageVec(Bg, wl["A"], dev, seed=0, first_env=glo, total_envs=total)
        else:
            env = envs_mod.SyntheticVec(Bg, wl["obs"], wl["A"], dev, seed=0, first_env=glo, total_envs=total)
        m = mcts_mod.BatchedMCTS(Bg, num_simulations=wl["sims"], maxium_action_sample=wl["K"], discount=0.999,
                                 root_dirichlet_alpha=0.25, root_exploration_fraction=0.1, device=local_rank,
                                 use_graph=not args.no_graph, fused=True, single_launch=not args.stepwise,
                                 rng_mode=smz._lib.RNG_PHILOX if args.rng == "philox" else smz._lib.RNG_MT19937_NUMPY)
        m.seed(np.arange(glo, glo + Bg, dtype=np.uint64))
        env.reset()
        groups.append(sp.StreamGroup(env, model.heads(dev, instance=gi, backend=args.heads), m, T))
    env, heads, mcts = groups[0].env, groups[0].heads, groups[0].mcts

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if not multi:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # N > 1: the chunk is played in `--gather-slices` slices; every finished slice's rows go to rank 0 on a side stream (compact
    # wire format) while the next slice is searched (gather.TrajectoryGather) -- only the last slice's transfer is exposed
    tg = gather_mod.TrajectoryGather(groups[0].chunk.rec_obs_dim, wl["A"], slices=args.gather_slices, total_envs=total,
                                     loopback=args.rccl_loopback) if multi else None
    exposed = []

    def rows(chunks, name, t0, t1):
        parts = [getattr(c, name) for c in chunks]
        if parts[0] is None:
            return None
        parts = [p[t0:t1] for p in parts]
        return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)

    # (gather.ChunkExchange: slices + overlapped exchange, or -- mode "plain" -- the whole chunk with one synchronous grouped
    #  send / receive; its warm_up() falls back to plain when the first overlapped exchange raises)
    xch = gather_mod.ChunkExchange(tg, lambda n, t0: sp.play_games_grouped(groups, args.temperature, n, t0=t0), rows,
                                   mode=args.gather_mode, total_envs=total,
                                   log=lambda m: print("bench.py: " + m, file=sys.stderr)) if multi else None
    gather_mode = xch.mode if xch is not None else {"kind": None}

    def play_and_gather(n):
        """n env steps of every group; with N > 1 in slices, each slice's rows handed to the overlapped gather."""
        if xch is None:
            return sp.play_games_grouped(groups, args.temperature, n)
        return xch.run(n)[0]

    def timed_block():
        """EXACTLY K steps (+ the trajectory gather when N > 1) between two barrier + synchronize pairs; max over ranks."""
        barrier()
        t0 = time.perf_counter()
        play_and_gather(args.steps)
        barrier()
        dt_block = max_over_ranks(time.perf_counter() - t0)
        if tg is not None and tg.exposed_gather_ms() is not None:
            exposed.append(tg.exposed_gather_ms())
        return dt_block

    sink = ReplaySink(td_steps=50)                            # config/experiment_421_config.json: td_steps 50
    e2e_parts = []
    lc_model = None
    if args.learning_cycle:
        assert args.end_to_end, "--learning-cycle is a variant of --end-to-end"
        import copy
        lc_model = copy.copy(model)
        lc_model.save_model = lambda **k: None                # (no checkpoint file per iteration inside the timed region)
    if args.end_to_end:
        assert G == 1, "--end-to-end times self_play_iteration: one env group"

        def timed_block():                                    # noqa: F811  (replaces the search-only block)
            """ONE self_play_iteration of K steps -- play, (gather,) transfer, Game records, save_game x games -- between two
            barrier + synchronize pairs; max over ranks."""
            sink.clear()
            barrier()
            t0 = time.perf_counter()
            if args.learning_cycle:
                before = len(sink.buffer)
                sp.learning_cycle(number_of_iteration=args.pipeline, number_of_self_play_before_training=1,
                                  number_of_training_before_self_play=0, model_tag_number=1, number_of_worker_selfplay="gpu",
                                  temperature_type="static_one_temperature", verbose=False, muzero_model=lc_model, gameplay=env,
                                  monte_carlo_tree_search=mcts, replay_buffer=sink, steps_per_iteration=args.steps,
                                  gather=tg if multi else None, pipeline=None if args.pipeline > 1 else False)
                n_games = len(sink.buffer) - before
            elif args.pipeline > 1:
                n_games = 0
                for games, _ in sp.self_play_iterations(env, model, mcts, args.temperature, args.steps, args.pipeline, replay_buffer=sink,
                                                        gather=tg if multi else None, ignore_termination=True):
                    n_games += len(games or ())
            else:
                games, _ = sp.self_play_iteration(env, model, mcts, args.temperature, args.steps, replay_buffer=sink,
                                                  gather=tg if multi else None, ignore_termination=True)
                n_games = len(games or ())
            barrier()
            dt_block = max_over_ranks(time.perf_counter() - t0)
            if rank == 0:
                e2e_parts.append((n_games // max(1, args.pipeline), sink.positions // max(1, args.pipeline)))
            return dt_block

    # one priming step outside everything: code-object upload, LDS opt-in and allocator warm-up are initialisation, not
    # part of a step (a run with --warmup 0 would otherwise time them)
    sp.play_games_grouped(groups, args.temperature, 1)
    torch.cuda.synchronize(dev)
    # Python's cyclic collector pauses the host for 30-50 ms when a full collection falls into the loop (torch keeps
    # ~10^6 objects alive); the GPU drains its queue in a few ms, so such a pause inside the timed region would be charged
    # to the engine.  Freeze what exists and keep the collector off while measuring.
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
    # W untimed warm-up steps (with N > 1 the first grouped send / recv builds the RCCL communicators here, outside the timed
    # region; if that first overlapped exchange RAISES, ChunkExchange.warm_up falls back to the plain gather)
    if xch is not None:
        chunks = xch.warm_up(max(1, args.warmup))[0]
    else:
        chunks = play_and_gather(args.warmup)
    torch.cuda.synchronize(dev)
    first = timed_block()                                      # block 1 (timed like the others; also sizes R)
    R = int(min(args.max_blocks, max(1, np.ceil(args.min_timed_seconds / max(first, 1e-6)))))
    blocks = [first] + [timed_block() for _ in range(R - 1)]
    dt = float(np.median(blocks))
    per_rank_rate, ranks_seen, gather_ms = None, None, None
    if multi:                                                  # every rank's own rate of its last block, for the record
        barrier()
        t0 = time.perf_counter()
        chunks = sp.play_games_grouped(groups, args.temperature, args.steps)
        torch.cuda.synchronize(dev)
        mine = B * wl["sims"] * args.steps / (time.perf_counter() - t0)
        rates = [None] * world
        dist.all_gather_object(rates, mine)
        per_rank_rate = [float(r) for r in rates]
        # the rank count as the collective itself sees it, and the gather alone (a K-step chunk per rank -> rank 0), timed
        # between barrier + synchronize pairs, max over ranks: what of a block's time is the exchange
        ones = torch.ones(1, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        gts = []
        for _ in range(5):
            barrier()
            t0 = time.perf_counter()
            if gather_mode["kind"] == "plain":
                gather_mod.gather_to_learner(rows(chunks, "data", 0, args.steps), total_envs=total)
            else:
                tg.start(rows(chunks, "data", 0, args.steps), rows(chunks, "obs", 0, args.steps))
                tg.finish()
            barrier()
            gts.append(max_over_ranks(time.perf_counter() - t0))
        gather_ms = 1e3 * float(np.median(gts))
    block_steps = args.steps * (args.pipeline if args.end_to_end else 1)      # env steps inside one timed block
    sims_total = total * wl["sims"] * block_steps
    headline = (args.workload == "cartpole_mlp_4096x50" and B == 4096 and not args.host_env and args.rng == "mt19937"
                and not args.end_to_end and not args.rccl_loopback)
    single = getattr(mcts, "_single", None) is True
    data_note = {"cartpole": "synthetic (CartPole-shaped Euler env, fixed-length episodes; checkpoint-421 weights)",
                 "synthetic": "synthetic (N(0,1) observations of LunarLander width generated on the device; random-init weights, reference init rule)",
                 "image": "synthetic (uniform 98x98x3 frames scrolled per step; random-init ResNet-v2 weights)"}[wl["env"]]
    config = {"workload": args.workload, "envs_per_gpu": B, "num_simulations": wl["sims"],
              "actions": wl["A"], "children_per_expansion": wl["K"],
              "hidden_floats": int(groups[0].heads.S) if hasattr(groups[0].heads, "S") else model.state_dimension,
              "rng": "per-tree MT19937 (numpy-legacy, parity mode)" if args.rng == "mt19937" else
                     "per-tree Philox4x32-10 (throughput mode: the numpy-legacy algorithms on counter-based words; NOT the reference's draws)",
              "search": ("one launch per env step (smz_vision_initial + smz_search_vision_act)" if wl["env"] == "image" else
                         "one launch per env step (smz_search_mlp_act)") if single else
                        ("step-wise kernels" + ("" if args.no_graph else ", one HIP graph per env step")),
              "stream_groups": G, "heads": type(groups[0].heads).__name__,
              "env": ("host, compiled step (envs.HostCartPoleVec: pinned-memory action download + observation upload per step)" if args.host_env == "native"
                      else f"host, Python envs rendering 400x600x3 uint8 frames (envs.HostImageVecEnv: {host_workers} worker processes for each of {G} env group(s), page-locked shared block, "
                           f"upload = {args.frame_upload}: " + ("115 KB of resize taps per frame + smz_frames_resize_taps_u8" if args.frame_upload == "taps" else "720 KB frames + smz_frames_resize_u8") + " per step)" if (args.host_env and wl["env"] == "image")
                      else f"host, Python envs (envs.HostVecEnv: {host_workers} worker processes for each of {G} env group(s) (sleeping on a futex while idle) writing into a page-locked shared block; action download + observation upload per step; "
                           + ("every worker steps its slice of the envs with ONE array step (host_envs.CartPoleBatch, the batched-slice protocol; identical env by env to the per-env loop))" if batched_slices else
                              "one Python env.step call per env; one group's search overlaps the other's host step)") if args.host_env else "device"),
              "host_workers": host_workers if args.host_env == "python" else None,
              "parallelism": f"envs sharded x{world}, trajectory gather to rank 0" if world > 1 else
                              ("single GPU, trajectory gather to itself over RCCL (loopback)" if args.rccl_loopback else "single GPU")}
    if multi:
        config["collective_backend"] = ("nccl (RCCL)" if backend == "nccl" else f"{backend} ({world} ranks share {n_dev} GPU(s))") + \
            (" -- LOOPBACK: one rank, every send received by the sender itself" if args.rccl_loopback else "")
        config["ranks"] = world
        config["ranks_seen_by_collective"] = ranks_seen
        config["gpus_visible"] = n_dev
    out = {"metric": "MCTS simulations/sec (whole node), CartPole MLP 4096 envs \u00d7 50 sims" if headline else
                     f"MCTS simulations/sec (whole node), {args.workload} at {B} envs/GPU" + (" (host-resident envs)" if args.host_env else "")
                     + (" (RCCL LOOPBACK: the N > 1 code path -- process group, sliced trajectory gather to itself -- with one rank)" if args.rccl_loopback else "")
                     + (" (Philox throughput-mode random streams)" if args.rng == "philox" else "")
                     + (" (END TO END: self_play_iteration = search + transfer + Game records + save_game)" if args.end_to_end else "")
                     + (f" (pipelined: {args.pipeline} iterations per block, the next search enqueued before a chunk's host half)" if args.end_to_end and args.pipeline > 1 else "")
                     + (" (through selfplay.learning_cycle, no training between iterations, save_model a no-op)" if args.learning_cycle else ""),
           "value": sims_total / dt, "unit": "simulations/s", "n_gpus": world, "steps": block_steps,
           "warmup": args.warmup, "ms_per_step": 1e3 * dt / block_steps, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32 (tree values, heads) + f64 (pUCT scores, root priors) + i32 (counts)",
           "data": data_note, "config": config,
           "timing": {"blocks": R, "steps_per_block": block_steps, "block_ms_median": 1e3 * dt,
                      "block_ms_min": 1e3 * float(np.min(blocks)), "block_ms_max": 1e3 * float(np.max(blocks)),
                      "block_ms_p10_p90": [1e3 * float(np.percentile(blocks, 10)), 1e3 * float(np.percentile(blocks, 90))],
                      "timed_region_s": float(np.sum(blocks)),
                      "rule": "value and ms_per_step from the median block; every block is K steps between barrier + "
                              "synchronize pairs, max over ranks"}}
    if args.end_to_end and rank == 0:
        # where one iteration's time goes (one extra iteration, each part between synchronisations)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        env.reset()
        chunk = sp.play_games(env, heads, mcts, args.temperature, args.steps)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        games = sp.chunk_to_records(chunk, None, env.num_actions, mcts.discount, limit_of_game_play=args.steps,
                                    ignore_termination=True, td_steps=sink.td_steps, observation_shape=getattr(env, "frame", None))
        t2 = time.perf_counter()
        sink.clear()
        for g in games:
            sink.save_game(g)
        t3 = time.perf_counter()
        out["end_to_end"] = {"games_per_iteration": e2e_parts[-1][0], "positions_per_iteration": e2e_parts[-1][1],
                             "search_ms": 1e3 * (t1 - t0), "records_ms": 1e3 * (t2 - t1), "save_game_ms": 1e3 * (t3 - t2),
                             "records": "selfplay.chunk_to_records: game ends + n-step targets + priorities on the device, env-major "
                                        "transposes, transfer to the host, one ArrayGameRecord per game",
                             "sink": "bench.ReplaySink.save_game per game: make_priority(td_steps=50) + priority / position "
                                     "bookkeeping (replay_buffer.py:109-137 without its O(buffer) re-normalisation per save)",
                             "note": "the three parts of ONE extra iteration, each between synchronisations; `value` is the whole "
                                     "self_play_iteration call (env.reset + play + records + save), median block"}
    if per_rank_rate is not None:
        out["per_rank_simulations_per_s"] = per_rank_rate
        out["timing"]["gather_ms_median"] = gather_ms
        rec_bytes = sum(sum(m.numel() * m.element_size() for m in gather_mod.pack_records(c.data[:args.steps], c.rec_obs_dim, wl["A"]))
                        for c in chunks)
        out["timing"]["gather_bytes_per_rank"] = int(rec_bytes + sum(c.obs[:args.steps].numel() * 4 for c in chunks if c.obs is not None))
        out["timing"]["gather_overlap"] = {
            "mode": gather_mode, "slices": tg.slices, "exposed_ms_median": float(np.median(exposed)) if exposed else None,
            "how": "the K-step block is played in `slices` parts; each finished part's rows travel to rank 0 on a side stream "
                   "(float32 for observations / flags / actions / root values, float64 for rewards / policies / child visits) while "
                   "the next part is searched; exposed = device time between the end of the last search and the end of the exchange "
                   "(None: host-staged gloo exchange); gather_ms_median = one whole K-step chunk exchanged alone"}

    # ---- roofline (rank 0) --------------------------------------------------------------------------------------
    if rank == 0 and not args.no_roofline:
        eng = mcts.engine
        S, A, K = eng.S, eng.A, eng.K
        # (1) level histogram of this workload (stats atomics on; not timed)
        eng.enable_stats(True)
        eng.read_stats(reset=True)
        mcts.run(env.obs, heads, train=True) if single else None
        mcts_e = mcts_mod.BatchedMCTS(Bg, num_simulations=wl["sims"], maxium_action_sample=wl["K"], discount=0.999,
                                      root_dirichlet_alpha=0.25, root_exploration_fraction=0.1, device=local_rank,
                                      use_graph=False, fused=True, single_launch=False)
        mcts_e.engine = eng
        if not single:
            mcts_e._search(env.obs, heads, True)
        torch.cuda.synchronize(dev)
        stats = eng.read_stats(reset=True)
        eng.enable_stats(False)
        k2, k5, depth = algorithmic_bytes(stats, A, K, S, wl["sims"])
        reps = max(1, min(args.steps, 8))
        if single:
            # dominant kernel = k_search_mlp (the whole search, one launch per env step): event pairs around each launch
            # PER_PAIR back-to-back launches of that kernel ALONE between one pair (an event costs a barrier packet on
            # each side, ~10 us; spread over PER_PAIR launches the mean agrees with rocprofv3's per-dispatch average).
            PER_PAIR = 4
            if wl["env"] == "image":
                hidden0, policy0 = heads.initial(env.obs)          # the representation launch is not this kernel
                def launch():
                    eng.search_vision(heads.desc, heads.weights, hidden0, policy0, train=True, act_temperature=args.temperature)
            else:
                def launch():
                    eng.search_mlp(heads.desc, heads.weights, env.obs, train=True, act_temperature=args.temperature)
            durs = []
            for _ in range(reps + 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda._sleep(2_000_000)      # the launches are queued behind this before e0 is reached
                e0.record()
                for _k in range(PER_PAIR):
                    launch()
                e1.record()
                durs.append((e0, e1))
            torch.cuda.synchronize(dev)
            ms = np.array([a.elapsed_time(b) for a, b in durs[1:]]) / PER_PAIR
            mean_us = float(ms.mean() * 1e3)
            bytes_launch = (k2 + k5) * Bg * wl["sims"]
            kernel = ("k_search_vision<MAXA> (root expansion + num_simulations x [select, conv nets + MFMA towers, expand, backup] in one launch)"
                      if wl["env"] == "image" else
                      "k_search_mlp<MAXA,KS,U,INSTR,AEX> (root + num_simulations x [select, heads, expand, backup] in one launch)")
        else:
            kernel = "k_expand_backup<MAXA,KS,true,AEX> (expand + backup + next select)"
            durs = []
        # the tree kernel on its own (step-wise path): events around each fused expand+backup+select launch.  A spin
        # kernel in front of every repetition lets the host enqueue the whole search before the GPU starts on it, so an
        # event pair brackets the kernel alone (an idle queue would stamp e0 early and add the host's launch latency).
        hidden, policy = heads.initial(env.obs)
        want = (heads.bind_engine(eng) if hasattr(heads, "bind_engine") else
                dict(want_mlp_input=getattr(heads, "wants_mlp_input", True), want_parent_hidden=getattr(heads, "wants_parent_hidden", False)))
        tdurs = []
        for _ in range(min(reps, 4)):
            eng.root_init(hidden, policy, train=True)
            eng.select(**want)
            torch.cuda._sleep(2_000_000)
            for s in range(wl["sims"] - 1):
                o = heads.recurrent(eng)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                eng.expand_backup_select(*o, **want)
                e1.record()
                tdurs.append((e0, e1))
            eng.expand_backup(*heads.recurrent(eng))
        torch.cuda.synchronize(dev)
        per_rep = wl["sims"] - 1
        tms = np.array([a.elapsed_time(b) for i, (a, b) in enumerate(tdurs) if i % per_rep])   # first pair of a
        # repetition dropped: it can absorb the tail of the spin kernel / a queue wake-up (seen: 0.07 - 60 ms)
        if os.environ.get("SMZ_BENCH_DEBUG"):
            print("tree event pairs (us): first", (tms[:8] * 1e3).round(1), "median", np.median(tms) * 1e3, "max", tms.max() * 1e3,
                  "argmax", int(tms.argmax()), file=sys.stderr)
        tms = tms[tms <= 3.0 * np.median(tms)]     # a host hiccup while the queue is short shows up as a 10-60 ms pair
        tree_us = float(tms.mean() * 1e3)
        # rows left in the tree (large batches, heads.bind_engine): the tree kernel neither gathers the parent's hidden row
        # (4 S bytes of K2) nor scatters the new one (4 S bytes of K5) -- the network kernel does; they are not its bytes
        in_place = getattr(heads, "_in_place", None) is eng
        tree_bytes = (k2 + k5 - (8 * S if in_place else 0)) * Bg
        if not single:
            ms, mean_us, bytes_launch = tms, tree_us, tree_bytes
        achieved = bytes_launch / (mean_us * 1e-6) / 1e9
        # ---- what actually bounds the launch (profiles/r03_ceiling.md) -------------------------------------------------------
        # At 4096 trees there are 16 trees per CU: a launch is a chain of num_simulations dependent rounds per wavefront
        # (descent levels, network layers, expansion and backup each wait for the one before), not a stream of bytes.  The
        # chain's own duration is measured live: the same kernel on HALF the trees with ONE wavefront per SIMD (4-wave
        # workgroups) -- nothing to contend with, every latency exposed.  frac = chain / launch says how much of the launch the
        # dependent chain alone explains; the second wavefront per SIMD then adds its whole work for the remaining 1 - frac.
        bound_actual = None
        # (a geometry the user forced with SMZ_SEARCH_WAVES applies to the production launch too: the comparison would be void)
        if single and wl["env"] != "image" and Bg % 2 == 0 and Bg // 2 >= 1024 and "SMZ_SEARCH_WAVES" not in os.environ:
            os.environ["SMZ_SEARCH_WAVES"] = "4"
            try:
                Bh = Bg // 2
                m_h = mcts_mod.BatchedMCTS(Bh, num_simulations=wl["sims"], maxium_action_sample=wl["K"], discount=0.999,
                                           root_dirichlet_alpha=0.25, root_exploration_fraction=0.1, device=local_rank,
                                           use_graph=False, fused=True, single_launch=True,
                                           rng_mode=smz._lib.RNG_PHILOX if args.rng == "philox" else smz._lib.RNG_MT19937_NUMPY)
                m_h.seed(np.arange(Bh, dtype=np.uint64))
                obs_h = env.obs[:Bh].clone()
                eng_h = m_h.run(obs_h, heads, train=True, act_temperature=args.temperature)
                hd = []
                for _ in range(reps + 1):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda._sleep(2_000_000)
                    e0.record()
                    for _k in range(PER_PAIR):
                        eng_h.search_mlp(heads.desc, heads.weights, obs_h, train=True, act_temperature=args.temperature)
                    e1.record()
                    hd.append((e0, e1))
                torch.cuda.synchronize(dev)
                chain_us = float(np.mean([a.elapsed_time(b) for a, b in hd[1:]]) / PER_PAIR * 1e3)
                bound_actual = {"kind": "latency: dependent chain of one wavefront's simulation rounds",
                                "label": "self-relative (the kernel timed against itself at half occupancy), not a roofline",
                                "chain_us": chain_us, "launch_us": mean_us, "frac": chain_us / mean_us,
                                "how": f"{eng_h.last_kernel()} on {Bh} trees in 4-wave workgroups (one wavefront per SIMD, same "
                                       "trees per wavefront): its launch time is the chain; the production launch runs two "
                                       "wavefronts per SIMD"}
                eng_h.close()
            finally:
                del os.environ["SMZ_SEARCH_WAVES"]
        elif single and wl["env"] == "image":
            # k_search_vision runs ONE wavefront per SIMD (4 trees per CU at 1024 trees): the launch is its own dependent chain
            bound_actual = {"kind": "latency: dependent chain of one wavefront's simulation rounds (one wavefront per SIMD)",
                            "label": "self-relative (the launch is its own chain), not a roofline",
                            "chain_us": mean_us, "launch_us": mean_us, "frac": 1.0,
                            "how": "1024 trees = 4 per CU = one 4-wave workgroup per CU: nothing overlaps a wavefront's own chain"}
        # HBM bytes per launch from the TCC counters, when a PMC pass of this workload/kernel has been committed
        # ... of the kernel instantiation that actually ran (smz_last_kernel), same workload, same kernel sources
        traffic, traffic_note = None, None
        launched = eng.last_kernel() if single else ""
        if os.environ.get("SMZ_LIB_PATH"):
            traffic_note = "SMZ_LIB_PATH is set: a variant library does not inherit counter files measured on the default build"
        elif single and Bg == wl["envs"]:
            tj, tname = find_traffic(args.workload + ("" if args.rng == "mt19937" else "+philox"), launched)
            if tj is not None:
                traffic = tj["hbm_bytes_per_launch_raw"]
                traffic_note = ("(FETCH_SIZE + WRITE_SIZE) x 1024 per launch from " + tname +
                                "; read side may be under-counted up to 2x on gfx950 (upper bound %.0f)" % tj["hbm_bytes_per_launch_read_x2"])
            else:
                traffic_note = tname
        compute, compute_mlp = None, None
        if single and wl["env"] != "image" and Bg == wl["envs"] and not os.environ.get("SMZ_LIB_PATH"):
            pj, pname = find_pmc(args.workload + ("" if args.rng == "mt19937" else "+philox"), launched)
            compute_mlp = valu_issue_bound(pj, pname, wl["sims"]) if pj is not None else {"bound": "valu_issue", "frac": None, "note": pname}
        if single and wl["env"] == "image":
            # f32 multiply-adds of one leaf evaluation (neural_network_vision_model.py:41-515 at 3x7x7): 3x3 convolutions (49 pixels,
            # 3 output channels), 1x1 mixes, the 147 -> H -> [H ->] S / A towers; dynamics leaves also evaluate the reward tower
            Hh, Lh, Ss, Ah = heads.H, heads.L, heads.Ssup, heads.A
            conv = lambda cin: 49 * 3 * cin * 9
            tower = lambda n_out: 147 * Hh + Lh * Hh * Hh + Hh * n_out
            common = conv(4) + Lh * 3 * conv(3) + Lh * 3 * conv(3) + 2 * 49 * 3 * 3 + tower(Ss) + tower(Ah)
            macs_dyn, macs_aft = common + 49 * 3 * 4 + tower(Ss), common
            macs = 0.5 * (macs_dyn + macs_aft)
            flops_launch = 2.0 * macs * Bg * wl["sims"]
            tf = flops_launch / (mean_us * 1e-6) / 1e12
            compute = {"bound": "compute", "unit": "TFLOP/s", "achieved": tf, "peak": 157.3, "frac": tf / 157.3,
                       "flops_per_launch": flops_launch, "macs_per_leaf": {"dynamics": macs_dyn, "afterstate": macs_aft},
                       "note": "f32 multiply-adds of the leaf networks (mean of the two branches) x leaves per launch / launch "
                               "time against the 157.3 TFLOP/s f32 vector peak (MI355X_MICROARCH.md); the launch is latency-bound "
                               "(bound_actual), neither figure is a ceiling it approaches"}
        out["roofline"] = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "bound_actual": bound_actual, "compute": compute if compute is not None else compute_mlp,
                           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                           "kernel_launched": launched or None, "kernel_source_sha16": kernel_source_sha16(),
                           "bytes_per_launch": bytes_launch,
                           "mean_launch_us": mean_us, "median_launch_us": float(np.median(ms) * 1e3),
                           "launches_timed": int(ms.size) * (PER_PAIR if single else 1), "bytes_per_tree_select": k2,
                           "bytes_per_tree_expand_backup": k5, "mean_depth": depth,
                           "tree_kernel_alone": {"kernel": "k_expand_backup<MAXA,KS,true,AEX>" + (" (hidden rows moved by the network kernel: 8 S bytes per tree less)" if in_place else ""),
                                                 "mean_launch_us": tree_us,
                                                 "bytes_per_launch": tree_bytes,
                                                 "achieved": tree_bytes / (tree_us * 1e-6) / 1e9,
                                                 "frac": tree_bytes / (tree_us * 1e-6) / 1e9 / HBM_PEAK_GBS},
                           "method": "HIP event pairs on the launching (torch current) stream, after the timed region: "
                                     "around 4 back-to-back launches of the search kernel alone (elapsed / 4), around each "
                                     "launch for the step-wise tree kernel; bytes = SURVEY 8d formula on this run's level "
                                     "histogram"}
    if rank == 0 and wl["env"] == "image" and out.get("roofline", {}).get("compute"):
        # vision family: the leaf networks are arithmetic, not byte movement -- the headline fraction is the f32 one; the
        # bytes-based figures (SURVEY 8d formula) stay beside it
        r = out["roofline"]
        c = r.pop("compute")
        r["hbm"] = {k: r[k] for k in ("achieved", "peak", "unit", "frac", "traffic", "traffic_note", "bytes_per_launch")}
        r.update(bound="compute", achieved=c["achieved"], peak=c["peak"], unit=c["unit"], frac=c["frac"],
                 flops_per_launch=c["flops_per_launch"], macs_per_leaf=c["macs_per_leaf"], compute_note=c["note"])
    if cpu_baseline is not None:                                                            # N=1 only (contract)
        out["cpu_baseline"] = cpu_baseline

    def also_line(name, seconds, steps=16):
        """One of the other single-GPU BASELINE configs, timed like the headline (blocks of `steps` env steps between
        synchronisations, median block) for ~`seconds`, + the dominant kernel's launch time and algorithmic-bytes fraction."""
        w = dict(WORKLOADS[name])
        Bw = w["envs"]
        wp = os.path.join(ROOT, "tests", "golden", w["weights"])
        mdl = model_mod.Muzero.from_state_dicts(wp) if w["env"] == "image" else model_mod.Muzero.from_arrays(wp)
        if w["env"] == "cartpole":
            e_ = envs_mod.CartPoleVec(Bw, dev, seed=0, first_env=0, total_envs=Bw)
        elif w["env"] == "image":
            e_ = envs_mod.ImageVec(Bw, w["A"], dev, seed=0, first_env=0, total_envs=Bw)
        else:
            e_ = envs_mod.SyntheticVec(Bw, w["obs"], w["A"], dev, seed=0, first_env=0, total_envs=Bw)
        m_ = mcts_mod.BatchedMCTS(Bw, num_simulations=w["sims"], maxium_action_sample=w["K"], discount=0.999,
                                  root_dirichlet_alpha=0.25, root_exploration_fraction=0.1, device=local_rank,
                                  use_graph=True, fused=True, single_launch=True)
        m_.seed(np.arange(Bw, dtype=np.uint64))
        e_.reset()
        h_ = mdl.heads(dev, instance=7, backend="auto")
        g_ = [sp.StreamGroup(e_, h_, m_, steps)]
        sp.play_games_grouped(g_, args.temperature, 3)
        torch.cuda.synchronize(dev)
        bl, t_end = [], time.perf_counter() + seconds
        while not bl or (time.perf_counter() < t_end and len(bl) < args.max_blocks):
            torch.cuda.synchronize(dev)
            t0_ = time.perf_counter()
            sp.play_games_grouped(g_, args.temperature, steps)
            torch.cuda.synchronize(dev)
            bl.append(time.perf_counter() - t0_)
        d_ = float(np.median(bl))
        line = {"workload": name, "value": Bw * w["sims"] * steps / d_, "unit": "simulations/s", "ms_per_step": 1e3 * d_ / steps,
                "steps": steps, "blocks": len(bl), "timed_region_s": float(np.sum(bl)), "envs": Bw, "num_simulations": w["sims"],
                "actions": w["A"], "children_per_expansion": w["K"]}
        eng_ = m_.engine
        if getattr(m_, "_single", None) is True:
            eng_.enable_stats(True)
            eng_.read_stats(reset=True)
            m_.run(e_.obs, h_, train=True)
            torch.cuda.synchronize(dev)
            st_ = eng_.read_stats(reset=True)
            eng_.enable_stats(False)
            k2_, k5_, _ = algorithmic_bytes(st_, eng_.A, eng_.K, eng_.S, w["sims"])
            if w["env"] == "image":
                hid_, pol_ = h_.initial(e_.obs)
                go = lambda: eng_.search_vision(h_.desc, h_.weights, hid_, pol_, train=True, act_temperature=args.temperature)   # noqa: E731
            else:
                go = lambda: eng_.search_mlp(h_.desc, h_.weights, e_.obs, train=True, act_temperature=args.temperature)        # noqa: E731
            ev = []
            for _ in range(5):
                a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda._sleep(2_000_000)
                a_.record()
                for _k in range(4):
                    go()
                b_.record()
                ev.append((a_, b_))
            torch.cuda.synchronize(dev)
            us_ = float(np.mean([x.elapsed_time(y) for x, y in ev[1:]]) / 4 * 1e3)
            by_ = (k2_ + k5_) * Bw * w["sims"]
            line.update(kernel=eng_.last_kernel(), kernel_mean_launch_us=us_, bytes_per_launch=by_,
                        frac=by_ / (us_ * 1e-6) / 1e9 / HBM_PEAK_GBS, frac_of="HBM peak 8 TB/s, algorithmic bytes (SURVEY 8d) / launch time")
        else:
            line.update(kernel="step-wise kernels", frac=None)
        if hasattr(e_, "close"):
            e_.close()
        torch.cuda.synchronize(dev)
        del g_
        eng_.close()
        return line

    if rank == 0 and headline and world == 1 and args.also_seconds > 0 and not args.stepwise and not os.environ.get("SMZ_LIB_PATH_NO_ALSO"):
        out["also"] = []
        for name in ("lunarlander_mlp_4096x50", "lunarlander_mlp_4096x50_K4", "cartpole_mlp_4096x100", "vision_resnet_1024x50"):
            try:
                out["also"].append(also_line(name, args.also_seconds))
            except Exception as e:                                     # noqa: BLE001  (the headline line must not die with an extra)
                out["also"].append({"workload": name, "error": f"{type(e).__name__}: {e}"})
    for g in groups:
        if hasattr(g.env, "close"):
            g.env.close()                                  # host envs: the worker processes exit
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
        os.close(json_fd)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
