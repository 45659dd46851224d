"""Vectorised synthetic environments resident on the GPU (the "synthetic fixed-length episodes" of BASELINE.json).

gymnasium is not part of this image, so the environments here are this engine's own: a CartPole-v1 shaped Euler
integrator (float64 state, float32 observations; physics constants as published for CartPole-v1) and a
LunarLander-shaped stand-in that only provides observations of the right width.  Env i of a job draws its initial
state from `RandomState(seed).uniform(...)[i]`, so a shard sees exactly the rows it would see in a single-GPU run.
"""
import ctypes as C
import os

import numpy as np
import time

import torch

from . import _lib


ON_END = {"continue": 0, "mask": 1, "reset": 2}


class CartPoleVec:
    """B CartPole-v1 shaped games stepped on the device.

    on_end says what happens to an env whose game is over (terminated, or `limit` steps played -- the two exits of the
    reference's loop, self_play.py:79):
      "continue"  keep stepping (the fixed-length synthetic episodes of the benchmark; flags are recorded, nothing else);
      "mask"      switch the env off: `active[e]` drops to 0 on the device, the search skips it from then on
                  (SearchEngine.set_active) and its later records carry flag 3;
      "reset"     start its next game at once (counter-based reset state, smz_cartpole_step_ctl), so that every
                  simulation of a chunk belongs to some game.
    """
    obs_dim, num_actions = 4, 2

    def __init__(self, num_envs, device, seed=0, first_env=0, total_envs=None, on_end="continue", limit=0):
        self.lib = _lib.load()
        self.B, self.device = int(num_envs), torch.device(device)
        self.seed, self.first_env = int(seed), int(first_env)
        self.total = int(total_envs) if total_envs is not None else self.first_env + self.B
        assert on_end in ON_END
        self.on_end, self.limit = on_end, int(limit)
        self.state = torch.empty(self.B, 4, dtype=torch.float64, device=self.device)
        self.obs = torch.empty(self.B, 4, dtype=torch.float32, device=self.device)
        self.reward = torch.empty(self.B, dtype=torch.float32, device=self.device)
        self.terminated = torch.empty(self.B, dtype=torch.uint8, device=self.device)     # the flag of the last step
        self.step_count = torch.zeros(self.B, dtype=torch.int32, device=self.device)
        self.episode = torch.zeros(self.B, dtype=torch.int32, device=self.device)
        self.active = torch.ones(self.B, dtype=torch.uint8, device=self.device) if on_end == "mask" else None
        self._ctl = None

    def reset(self):
        all_states = np.random.RandomState(self.seed).uniform(-0.05, 0.05, size=(self.total, 4))
        st = all_states[self.first_env:self.first_env + self.B]
        self.state.copy_(torch.from_numpy(np.ascontiguousarray(st)))
        self.obs.copy_(self.state.to(torch.float32))
        self.terminated.zero_()
        self.step_count.zero_()
        self.episode.zero_()
        if self.active is not None:
            self.active.fill_(1)
        return self.obs

    def reset_state_of(self, env, episode):
        """Host copy of the state env `env` (global index) starts its game number `episode` >= 1 from (on_end="reset")."""
        out = (C.c_double * 4)()
        _lib.check(self.lib.smz_cartpole_reset_state(self.seed, int(env), int(episode), C.byref(out)))
        return np.array(list(out))

    def _controlled(self):
        return self.on_end != "continue" or self.limit > 0

    def _ctl_struct(self):
        if self._ctl is None:
            P = lambda x: None if x is None else x.data_ptr()
            self._ctl = _lib.EpisodeCtl(P(self.step_count), P(self.episode), P(self.active), self.limit, ON_END[self.on_end],
                                        self.seed, self.first_env)
        return self._ctl

    def step(self, action):
        """action: int32 [B] device tensor (index into action_map).  Asynchronous on the current stream."""
        return self.step_and_record(action, None, 0, None, None, None)

    def fused_step(self, chunk_data, t):
        """The arguments of smz_search_mlp_act_cartpole for this env's next step (the step + record run in the tail of the
        search launch: one launch per env step).  SMZ_FUSED_ENV_STEP=0 keeps the two launches (A/B runs)."""
        if os.environ.get("SMZ_FUSED_ENV_STEP", "1") == "0":
            return None
        P = lambda x: None if x is None else x.data_ptr()
        ctl = C.pointer(self._ctl_struct()) if self._controlled() else None
        return _lib.CartPoleEnv(P(self.state), P(self.obs), P(self.reward), P(self.terminated), ctl, P(chunk_data),
                                0 if chunk_data is None else chunk_data.shape[0], int(t))

    def step_and_record(self, action, chunk_data, t, policy, child_visits, root_value):
        """step(action) + the trajectory record of this step (smz_traj_pack's layout) in one launch."""
        s = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        P = lambda x: None if x is None else C.c_void_p(x.data_ptr())
        T = 0 if chunk_data is None else chunk_data.shape[0]
        if self._controlled():
            _lib.check(self.lib.smz_cartpole_step_ctl(P(self.state), P(action), P(self.obs), P(self.reward), P(self.terminated),
                                                      C.byref(self._ctl_struct()), P(chunk_data), T, int(t), P(policy),
                                                      P(child_visits), P(root_value), self.B, s))
        elif chunk_data is None:
            _lib.check(self.lib.smz_cartpole_step(P(self.state), P(action), P(self.obs), P(self.reward), P(self.terminated),
                                                  self.B, s))
        else:
            _lib.check(self.lib.smz_cartpole_step_pack(P(self.state), P(action), P(self.obs), P(self.reward),
                                                       P(self.terminated), P(chunk_data), T, int(t), P(policy),
                                                       P(child_visits), P(root_value), self.B, s))
        return self.obs, self.reward, self.terminated


class SyntheticVec:
    """Observation-only stand-in (e.g. LunarLander-shaped: obs 8 ~ N(0,1), 4 actions; Box2D is absent here).  The
    observations are generated on the device (smz_synthetic_obs): element (env, k) of step t is a pure function of
    (seed, global env index, t, k), so a shard sees the rows it would see in a single-GPU run and an env step costs no
    host work and no copy."""

    def __init__(self, num_envs, obs_dim, num_actions, device, seed=0, first_env=0, total_envs=None):
        self.lib = _lib.load()
        self.B, self.obs_dim, self.num_actions = int(num_envs), int(obs_dim), int(num_actions)
        self.device = torch.device(device)
        self.seed, self.first_env = int(seed), int(first_env)
        self.total = int(total_envs) if total_envs is not None else self.first_env + self.B
        self.obs = torch.empty(self.B, self.obs_dim, dtype=torch.float32, device=self.device)
        self.reward = torch.zeros(self.B, dtype=torch.float32, device=self.device)
        self.terminated = torch.zeros(self.B, dtype=torch.uint8, device=self.device)
        self._t = 0

    def _draw(self):
        _lib.check(self.lib.smz_synthetic_obs(C.c_void_p(self.obs.data_ptr()), self.B, self.obs_dim, self.seed, self.first_env,
                                              self._t, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        self._t += 1

    def reset(self):
        self._t = 0
        self._draw()
        return self.obs

    def step(self, action):
        self._draw()
        return self.obs, self.reward, self.terminated


class ImageVec:
    """Frame-observation stand-in for the vision family: [B,3,98,98] float32 frames in [0,1) (the shape
    muzero_model.py:400-404 fixes; the reference's resize to it, game.py:82-89, needs torchvision and is not pinned).
    Frame j = RandomState(seed + j).rand(3,98,98); env i (global index) shows frame i + (t mod POOL) at step t.  The
    frames of a shard sit in one device-resident pool and `obs` is a contiguous window into it, so an env step moves no
    data (the frames of a real env arrive from outside the engine; a device-side scroll here only measured torch.roll)."""
    frame = (3, 98, 98)
    POOL = 16

    def __init__(self, num_envs, num_actions, device, seed=0, first_env=0, total_envs=None):
        self.B, self.num_actions = int(num_envs), int(num_actions)
        self.obs_dim = int(np.prod(self.frame))
        self.device = torch.device(device)
        self.seed, self.first_env = int(seed), int(first_env)
        self.pool = torch.empty((self.B + self.POOL,) + self.frame, dtype=torch.float32, device=self.device)
        self.t = 0
        self.obs = self.pool[:self.B]
        self.reward = torch.zeros(self.B, dtype=torch.float32, device=self.device)
        self.terminated = torch.zeros(self.B, dtype=torch.uint8, device=self.device)

    def reset(self):
        rows = np.stack([np.random.RandomState(self.seed + self.first_env + j).rand(*self.frame).astype(np.float32)
                         for j in range(self.B + self.POOL)])
        self.pool.copy_(torch.from_numpy(rows))
        self.t = 0
        self.obs = self.pool[:self.B]
        return self.obs

    def step(self, action):
        self.t += 1
        k = self.t % (self.POOL + 1)
        self.obs = self.pool[k:k + self.B]
        return self.obs, self.reward, self.terminated


from .host_envs import HostCartPole, HostCartPoleRender  # noqa: E402,F401  (numpy-only module: the worker processes import it too)
from . import host_envs as _he  # noqa: E402


class _Done:
    def synchronize(self):
        pass


class HostVecEnv:
    """B environments that live on the HOST (gymnasium-style objects) behind the interface the batched loop drives
    (SURVEY 8f-4): per env step one download of the B actions and one upload of the B observations / rewards / flags through
    page-locked memory, both asynchronous on the engine's stream; the only host wait is for the actions.

    `envs`: a list of B single environments (reset(seed=) -> obs | (obs, info); step(a) -> (obs, reward, terminated, ...)) or
    of zero-argument callables that build one (the gymnasium.vector convention).  The rules the reference's Game keeps around
    env.step (first observation from reset(seed=env_seed + global index), illegal-move rule, limit / termination flags, mask /
    reset at a game's end: game.py:96-131, 223-273) live in host_envs.HostSlice.

    workers = 0: the envs are stepped in this process, one after the other (the serial adapter).
    workers = N: N child processes (fresh interpreters that import numpy and the env's module, never torch or the GPU runtime),
    each stepping a contiguous slice of the envs with the same HostSlice code and writing its rows straight into ONE shared,
    page-locked block the GPU copies from -- the Ray fan-out of self_play.py:240-256 without pickling a model or a game per
    task.  Results are identical to workers = 0 env by env (tests/test_host_envs.py, test_gpu_host_envs.py).

    step(action) = step_begin(action) + step_end(): the split lets selfplay.play_games_grouped search one env group on the
    GPU while another group's envs step on the host.
    Observations are flattened float32 vectors (game.py:145-167); rendered RGB frames: HostImageVecEnv."""

    def __init__(self, envs, obs_dim, num_actions, device, action_map=None, env_seed=0, limit=0, on_end="reset", first_env=0,
                 transform=None, workers=0, spin=50, _adapter=None, batch_step=True):
        assert on_end in ("mask", "reset")
        self.lib = _lib.load()
        self.B = len(envs)
        self.obs_dim, self.num_actions = int(obs_dim), int(num_actions)
        self.device = torch.device(device)
        self.action_map = list(action_map) if action_map is not None else list(range(self.num_actions))
        self.env_seed, self.limit, self.on_end, self.first_env = int(env_seed), int(limit), on_end, int(first_env)
        self.adapter = _adapter if _adapter is not None else _he.VectorAdapter(self.obs_dim, transform)
        self.workers = int(min(max(0, workers), self.B))
        self.batch_step = bool(batch_step)    # envs whose class offers make_batch are stepped slice-wise (host_envs.CartPoleBatch)
        self.spin = int(spin)        # polls of a shared word before a waiter sleeps in the kernel (futex); 0 on a CPU-quota'd box is fine
        B, row, dtype = self.B, self.adapter.row, self.adapter.dtype
        lay = _he.block_layout(B, row, dtype, max(1, self.workers))
        self._procs, self._block, self._registered, self._seq = [], None, False, 0
        if self.workers:
            self._block = _he.SharedBlock(lay["total"])
            buf = self._block.mm
            self._base = torch.frombuffer(buf, dtype=torch.uint8)
            if self.device.type != "cpu":             # (device "cpu": the CPU tests of the stepping logic, no GPU runtime)
                _lib.check(self.lib.smz_host_register(C.c_void_p(self._base.data_ptr()), lay["total"]))
                self._registered = True
        else:
            self._base = torch.zeros(lay["total"], dtype=torch.uint8, pin_memory=self.device.type != "cpu")
            buf = memoryview(self._base.numpy())
        self._arr, self._ctrl, self._lay = _he.map_arrays(buf, B, row, dtype, max(1, self.workers))
        self._words = _he.control_words(buf, max(1, self.workers))
        self._host_ptr = self._base.data_ptr()
        tdt = torch.float32 if dtype is np.float32 else torch.uint8
        self._d_action = torch.zeros(B, dtype=torch.int32, device=self.device)
        self.reward = torch.zeros(B, dtype=torch.float32, device=self.device)
        self.terminated = torch.zeros(B, dtype=torch.uint8, device=self.device)
        self.active = torch.ones(B, dtype=torch.uint8, device=self.device) if on_end == "mask" else None
        self._d_rows = torch.zeros(B, row, dtype=tdt, device=self.device)          # the envs' rows as uploaded
        self._side_cap = 0
        self.transfer_seconds = 0.0            # host time spent waiting for the action download (diagnostic)
        self.host_step_seconds = 0.0           # host time spent stepping (or waiting for the workers)
        self.upload_bytes = 0
        self._pending = None
        self._alloc_observations()
        if self.workers:
            self._start_workers(envs)
            self._slice = None
        else:
            envs = [_he.build_env(e) for e in envs]
            self._slice = _he.HostSlice(envs, 0, self.adapter, self._arr, self.action_map, self.env_seed, self.limit, self.on_end,
                                        self.first_env, batch=self.batch_step)
        self.envs = self._slice.envs if self._slice is not None else None

    @property
    def episode(self):
        """Game number of every env (serial adapter only: with workers the counters live in the worker processes)."""
        if self._slice is None:
            raise AttributeError("episode counters live in the worker processes (workers > 0)")
        return self._slice.episode

    @property
    def step_count(self):
        if self._slice is None:
            raise AttributeError("step counters live in the worker processes (workers > 0)")
        return self._slice.step_count

    # ---- observation tensors (vector observations; HostImageVecEnv overrides these) --------------------------------------
    def _alloc_observations(self):
        self.obs = self._d_rows                                                   # [B, obs_dim] float32: the uploaded rows themselves
        self.record_obs = torch.zeros_like(self._d_rows)                          # post-step observations (the record's)
        self.record_patch = None

    def _rows_to_observations(self, rows_dev, n, index_dev, out):
        """Turns n uploaded rows into observation rows of `out` (row index_dev[i], or i)."""
        if index_dev is None:
            if out is not rows_dev:
                out.copy_(rows_dev)
        else:
            out.index_copy_(0, index_dev[:n].long(), rows_dev[:n])

    def _publish(self, after_step):
        """The rows of this step are on the device (self._d_rows): derive `obs` and what the record shows."""
        if after_step:
            self.record_obs.copy_(self._d_rows)

    def _patch_record(self, n):
        self._rows_to_observations(self._d_side, n, self._d_side_rows, self.record_obs)

    # ---- transfers ----------------------------------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _h2d(self, dst, name, nbytes=None):
        off = self._lay[name]
        nbytes = dst.numel() * dst.element_size() if nbytes is None else nbytes
        if self.device.type == "cpu":              # (CPU tests of the stepping logic: no GPU runtime, a plain copy)
            dst.view(-1).view(torch.uint8).numpy()[:] = np.frombuffer(self._base.numpy(), np.uint8, nbytes, off)
            return
        _lib.check(self.lib.smz_copy_async(C.c_void_p(dst.data_ptr()), C.c_void_p(self._host_ptr + off), nbytes, 1, self._stream()))

    def _upload(self, after_step):
        B = self.B
        self._h2d(self._d_rows, "obs")
        self.upload_bytes += self._d_rows.numel() * self._d_rows.element_size()
        self._publish(after_step)
        if after_step:
            self._h2d(self.reward, "reward")
            self._h2d(self.terminated, "flag")
            ended = np.nonzero(self._arr["ended"])[0]
            n = len(ended)
            if n:                                  # envs reset inside this step: their post-step rows go into the record
                if n > self._side_cap:
                    self._side_cap = max(n, 2 * self._side_cap, 8)
                    pin = dict(pin_memory=self.device.type != "cpu")
                    self._h_side = torch.zeros(self._side_cap, self.adapter.row, dtype=self._d_rows.dtype, **pin)
                    self._h_side_rows = torch.zeros(self._side_cap, dtype=torch.int32, **pin)
                    self._d_side = torch.zeros(self._side_cap, self.adapter.row, dtype=self._d_rows.dtype, device=self.device)
                    self._d_side_rows = torch.zeros(self._side_cap, dtype=torch.int32, device=self.device)
                self._h_side.numpy()[:n] = self._arr["rec"][ended]
                self._h_side_rows.numpy()[:n] = ended
                self._d_side[:n].copy_(self._h_side[:n], non_blocking=True)
                self._d_side_rows[:n].copy_(self._h_side_rows[:n], non_blocking=True)
                self.upload_bytes += n * self.adapter.row * self._d_rows.element_size()
                self._patch_record(n)
            else:
                self.record_patch = None
        if self.active is not None:
            self._h2d(self.active, "active")

    # ---- workers --------------------------------------------------------------------------------------------------------------
    def _start_workers(self, envs):
        import pickle
        import subprocess
        import sys
        import tempfile
        here = os.path.dirname(os.path.abspath(__file__))
        W, B = self.workers, self.B
        cuts = [B * w // W for w in range(W + 1)]
        self._spec_files = []
        env_vars = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        for w in range(W):
            spec = dict(block_path=self._block.path, nbytes=self._block.nbytes, B=B, row=self.adapter.row, dtype=self.adapter.dtype,
                        workers=W, worker=w, lo=cuts[w], envs=list(envs[cuts[w]:cuts[w + 1]]), adapter=self.adapter,
                        action_map=self.action_map, env_seed=self.env_seed, limit=self.limit, on_end=self.on_end,
                        first_env=self.first_env, parent_pid=os.getpid(), spin=self.spin, batch=self.batch_step)
            f = tempfile.NamedTemporaryFile(prefix="smz_hostenv_spec_", suffix=".pkl", delete=False)
            pickle.dump(spec, f, protocol=pickle.HIGHEST_PROTOCOL)
            f.close()
            self._spec_files.append(f.name)
            self._procs.append(subprocess.Popen([sys.executable, os.path.join(here, "host_worker.py"), f.name], env=env_vars))
        alive = lambda: all(p.poll() is None for p in self._procs)           # noqa: E731
        t0 = time.perf_counter()
        while not all(self._words[_he.done_word(w)] == -1 for w in range(W)):   # every worker has mapped the block
            if not alive():
                self.close()
                raise RuntimeError("a host-env worker process exited while starting (its traceback is above)")
            if time.perf_counter() - t0 > 120:
                self.close()
                raise TimeoutError("host-env workers did not start within 120 s")
            time.sleep(0.002)
        self._block.unlink()
        for f in self._spec_files:
            os.unlink(f)
        self._spec_files = []

    def _command(self, cmd):
        """Starts `cmd` on every worker (returns at once) -- or runs it here (workers = 0)."""
        if self._slice is not None:
            self._slice.reset_all() if cmd == _he.CMD_RESET else self._slice.step_all()
            return
        self._seq += 1
        self._ctrl[1] = cmd
        _he.store_release(self._words, _he.GO, self._seq)      # (after the actions and the command)
        _he.futex_wake_all(self._words, _he.GO)

    def _wait_workers(self):
        """Sleeps on the completion bell until every worker has finished the current command."""
        if self._slice is not None:
            return
        seq, words = self._seq, self._words
        n = 0
        for w in range(self.workers):                  # in worker order: each word has ONE writer, so a value read here is final
            idx = _he.done_word(w)
            while True:
                seen = _he.load_acquire(words, idx)
                if seen == seq:
                    break
                _he.futex_wait_change(words, idx, seen, timeout_s=0.01, spin=self.spin)
                n += 1
                if n % 50 == 0 and not all(p.poll() is None for p in self._procs):
                    raise RuntimeError("a host-env worker process died (its traceback is above)")

    # ---- the loop's interface --------------------------------------------------------------------------------------------
    def reset(self):
        self._command(_he.CMD_RESET)
        self._wait_workers()
        self._upload(after_step=False)
        return self.obs

    def step_begin(self, action):
        """Enqueues the download of this step's actions; the search that produced them is still running."""
        if self.device.type == "cpu":
            self._arr["action"][:] = action.numpy()
            self._pending = _Done()
            return
        stream = torch.cuda.current_stream(self.device)
        _lib.check(self.lib.smz_copy_async(C.c_void_p(self._host_ptr + self._lay["action"]), C.c_void_p(action.data_ptr()),
                                           4 * self.B, 0, self._stream()))
        ev = torch.cuda.Event()
        ev.record(stream)
        self._pending = ev

    def step_end(self):
        """Waits for the actions, steps the envs (here or on the workers), enqueues the uploads."""
        t0 = time.perf_counter()
        self._pending.synchronize()                       # the search of this step has to finish before the env can move
        self._pending = None
        t1 = time.perf_counter()
        self._command(_he.CMD_STEP)
        self._wait_workers()
        self.transfer_seconds += t1 - t0
        self.host_step_seconds += time.perf_counter() - t1
        self._upload(after_step=True)
        return self.obs, self.reward, self.terminated

    def step(self, action):
        self.step_begin(action)
        return self.step_end()

    def close(self):
        if self._procs:
            try:
                self._seq += 1
                self._ctrl[1] = _he.CMD_EXIT
                _he.store_release(self._words, _he.GO, self._seq)
                _he.futex_wake_all(self._words, _he.GO)
            except Exception:
                pass
            for p in self._procs:
                try:
                    p.wait(timeout=5)
                except Exception:
                    p.kill()
            self._procs = []
        if self._registered:
            self.lib.smz_host_unregister(C.c_void_p(self._host_ptr))
            self._registered = False
        if self._block is not None:
            self._block.unlink()
        for f in getattr(self, "_spec_files", []):
            try:
                os.unlink(f)
            except OSError:
                pass
        if self._slice is not None:
            self._slice.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostImageVecEnv(HostVecEnv):
    """HostVecEnv for environments observed through rendered RGB frames (the reference's rgb_observation games:
    game.py:82-89, 105-107, 142-143 -- every frame through ToTensor + Resize(98, 98), one at a time on the CPU).

    Per env step the B uint8 frames go up through the page-locked block as they are and ONE launch on the engine's stream turns
    them into the [B,3,98,98] float32 tensor the vision heads read.  upload="frames": whole [H][W][3] frames (3 bytes per
    pixel; smz_frames_resize_u8); upload="taps" (default): only the source pixels the bilinear resize reads, gathered by the
    env's process ([2 out_h][2 out_w][3]: 115 KB instead of 720 KB for a 400 x 600 frame; smz_frames_resize_taps_u8) -- the
    blend runs on the device with the full-frame kernel's arithmetic, so both give bit-identical observations.
    frame_source "render": the frame is env.render() (what the reference's Game.render shows the agent); "obs": the env's
    observation itself is the frame.

    The record (round 4): `obs` is the next search's input and -- except for the few envs that finished a game
    this step -- also the frame the record shows, so `record_obs` is None (the representation launch of the next search copies
    the frames into the trajectory chunk while reading them, smz_vision_initial_record) and `record_patch` = (rows, frames)
    carries the post-step frames of the ended envs, which selfplay._play_step writes over those rows of the record."""
    frame = (3, 98, 98)

    def __init__(self, envs, frame_hw, num_actions, device, out_hw=(98, 98), frame_source="render", upload="taps", **kw):
        self.H, self.W = int(frame_hw[0]), int(frame_hw[1])
        self.out_h, self.out_w = int(out_hw[0]), int(out_hw[1])
        self.frame = (3, self.out_h, self.out_w)
        self.frame_source, self.upload = frame_source, upload
        adapter = _he.FrameAdapter(frame_hw, out_hw, frame_source, upload)
        super().__init__(envs, 3 * self.out_h * self.out_w, num_actions, device, _adapter=adapter, **kw)

    def _alloc_observations(self):
        self.obs = torch.zeros((self.B,) + self.frame, dtype=torch.float32, device=self.device)
        self.record_obs = None
        self.record_patch = None
        self._patch_frames = None

    def _rows_to_observations(self, rows_dev, n, index_dev, out):
        P = lambda x: None if x is None else C.c_void_p(x.data_ptr())
        fn = self.lib.smz_frames_resize_taps_u8 if self.upload == "taps" else self.lib.smz_frames_resize_u8
        _lib.check(fn(P(rows_dev), n, self.H, self.W, self.out_h, self.out_w, P(index_dev), P(out), self._stream()))

    def _publish(self, after_step):
        self._rows_to_observations(self._d_rows, self.B, None, self.obs)
        self.record_patch = None

    def _patch_record(self, n):
        if self._patch_frames is None or self._patch_frames.shape[0] < self._side_cap:
            self._patch_frames = torch.zeros((self._side_cap,) + self.frame, dtype=torch.float32, device=self.device)
        self._rows_to_observations(self._d_side, n, None, self._patch_frames)
        self.record_patch = (self._d_side_rows[:n], self._patch_frames[:n])


class HostCartPoleVec:
    """B CartPole-v1 shaped environments stepped on the HOST by compiled code (smz_host_cartpole_step), behind the same
    interface as the device envs: per env step the B actions come down and the B observations / rewards / flags go up
    through pinned memory on the engine's stream, and the host waits once, for the actions.  This is the boundary's
    host-buffer variant at its best -- what a compiled vector env costs -- where HostVecEnv over Python envs measures
    the Python.  Fixed-length episodes as CartPoleVec(on_end="continue")."""
    obs_dim, num_actions = 4, 2

    def __init__(self, num_envs, device, seed=0, first_env=0, limit=0):
        self.lib = _lib.load()
        self.B, self.device = int(num_envs), torch.device(device)
        self.seed, self.first_env, self.limit = int(seed), int(first_env), int(limit)
        pin = dict(pin_memory=torch.cuda.is_available())
        B = self.B
        self._state = torch.zeros(B, 4, dtype=torch.float64, **pin)
        self._h_obs = torch.zeros(B, 4, dtype=torch.float32, **pin)
        self._h_reward = torch.zeros(B, dtype=torch.float32, **pin)
        self._h_flag = torch.zeros(B, dtype=torch.uint8, **pin)
        self._h_action = torch.zeros(B, dtype=torch.int32, **pin)
        self._count = torch.zeros(B, dtype=torch.int32)
        self.obs = torch.zeros(B, 4, dtype=torch.float32, device=self.device)
        self.reward = torch.zeros(B, dtype=torch.float32, device=self.device)
        self.terminated = torch.zeros(B, dtype=torch.uint8, device=self.device)
        self.transfer_seconds = 0.0

    def reset(self):
        rows = np.stack([np.random.RandomState(self.seed + self.first_env + i).uniform(-0.05, 0.05, 4) for i in range(self.B)])
        self._state.copy_(torch.from_numpy(rows))
        self._h_obs.copy_(self._state.to(torch.float32))
        self._count.zero_()
        self.obs.copy_(self._h_obs, non_blocking=True)
        return self.obs

    def step_begin(self, action):
        """Enqueues the download of this step's actions (the split lets play_games_grouped search another env group meanwhile)."""
        stream = torch.cuda.current_stream(self.device)
        self._h_action.copy_(action, non_blocking=True)
        self._pending = torch.cuda.Event()
        self._pending.record(stream)

    def step_end(self):
        t0 = time.perf_counter()
        self._pending.synchronize()                       # the search of this step has to finish before the env can move
        self.transfer_seconds += time.perf_counter() - t0
        P = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(self.lib.smz_host_cartpole_step(P(self._state), P(self._h_action), P(self._h_obs), P(self._h_reward),
                                                   P(self._h_flag), P(self._count), self.limit, self.B))
        self.obs.copy_(self._h_obs, non_blocking=True)
        self.reward.copy_(self._h_reward, non_blocking=True)
        self.terminated.copy_(self._h_flag, non_blocking=True)
        return self.obs, self.reward, self.terminated

    def step(self, action):
        self.step_begin(action)
        return self.step_end()
