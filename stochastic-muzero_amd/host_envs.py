"""Host-resident environments: the per-env rules of the reference's Game around env.step (game.py:96-131, 223-273) over a
SLICE of a vector env's rows, the built-in CartPole stand-ins, and the worker process of the parallel stepper.

numpy + standard library only: envs.HostVecEnv runs a HostSlice in-process (the serial adapter); with `workers=N` it starts N
child processes (host_worker.py -> worker_main) that never import torch or touch the GPU, each owning a contiguous slice of
the envs and running the SAME HostSlice code on views of one shared-memory block -- the parallel stepper is the serial one by
construction, env by env.  This replaces the Ray fan-out of the reference's self-play (self_play.py:240-256: one game per
worker process) for envs that live on the host.

Shared block (SharedBlock): one file-backed mapping (/dev/shm) holding, for all B envs of the vector env,
    action i32 [B] | reward f32 [B] | flag u8 [B] | active u8 [B] | ended u8 [B] | obs <dtype> [B][row] | rec <dtype> [B][row]
and a control page: `go` (step sequence number written by the parent), `cmd`, one `done` sequence number per worker.  The
parent registers the mapping with the HIP runtime (page-locked), so rows written by a worker are DMA'd to the GPU from where
they lie.  Hand-off is by sequence numbers in shared memory (spin, then short sleeps): a step of 4096 CartPoles is ~0.1 ms of
work per worker, less than a pipe round trip.
"""
import ctypes
import mmap
import os
import pickle
import sys
import tempfile
import time

import numpy as np


def usable_cores():
    """Cores this process can actually USE: its affinity mask, capped by the container's CPU quota (cgroup v2 cpu.max / v1
    cfs_quota_us) -- the bench box shows 256 hardware threads and grants 16 CPUs' worth of time; threads beyond the quota only
    get the others throttled."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(period)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            if q > 0:
                quota = q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError):
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n



# ---------------------------------------------------------------------------------------------------------------------------
# built-in stand-ins (gymnasium is not part of this build)
# ---------------------------------------------------------------------------------------------------------------------------
_RS = None


def _seeded_uniform4(seed):
    """np.random.RandomState(seed).uniform(-0.05, 0.05, size=4), value for value, from ONE reseeded generator object per process:
    constructing a RandomState costs ~130 us (it first seeds itself from the OS), reseeding one ~5 us -- with thousands of envs
    the resets of finished games were most of a step."""
    global _RS
    if _RS is None:
        _RS = np.random.RandomState(0)
    _RS.seed(seed)
    return _RS.uniform(-0.05, 0.05, size=4)


class HostCartPole:
    """One CartPole-v1 shaped game on the host behind the gym call shape (reset(seed=) -> (obs, info); step(a) -> (obs,
    reward, terminated, truncated, info)): float64 Euler physics with CartPole-v1's published constants, the arithmetic
    of smz_cartpole_step.  gymnasium is not part of this build; this class is what the host-environment path is
    exercised with, and what `muzero_cli.py` uses for a single-game run."""
    metadata = {"render_fps": 50}

    def __init__(self):
        self.state = None

    def reset(self, seed=None):
        self.state = _seeded_uniform4(seed)
        return self.state.astype(np.float32), {}

    def step(self, action):
        if action not in (0, 1):
            raise ValueError(f"illegal action {action!r}")
        x, xd, th, thd = (float(v) for v in self.state)
        force = 10.0 if action == 1 else -10.0
        ct, sn = np.cos(th), np.sin(th)
        temp = (force + 0.05 * thd * thd * sn) / 1.1
        tha = (9.8 * sn - ct * temp) / (0.5 * (4.0 / 3.0 - 0.1 * ct * ct / 1.1))
        xa = temp - 0.05 * tha * ct / 1.1
        self.state = np.array([x + 0.02 * xd, xd + 0.02 * xa, th + 0.02 * thd, thd + 0.02 * tha])
        term = bool(abs(self.state[0]) > 2.4 or abs(self.state[2]) > 12 * 2 * np.pi / 360)
        return self.state.astype(np.float32), 1.0, term, False, {}

    def close(self):
        pass

    @classmethod
    def make_batch(cls, envs):
        """The batched-slice protocol (HostSlice): one object stepping all of a slice's envs with array arithmetic."""
        return CartPoleBatch(envs)


class CartPoleBatch:
    """HostCartPole.step for a whole slice at once: the state of the slice's n envs is ONE [n][4] float64 array (every env
    object's `state` becomes a row view of it) and a step is ~40 numpy operations whatever n is, instead of n Python calls of
    ~4 us.  Operation by operation the arithmetic of HostCartPole.step (same association, same numpy cos / sin), so the
    results are identical env by env -- the per-env path is the checker (tests/test_host_envs.py).

    The protocol a HostSlice asks of an env class (`make_batch(envs)` defined ON the class itself: a subclass that overrides
    step() without its own make_batch is stepped env by env):
        reset_one(i, seed) -> observation of env i after reset(seed=seed)
        step_batch(actions int64 [n], go bool [n]) -> (obs [n][...], reward float64 [n], terminated bool [n], illegal bool [n])
            steps the envs with go[i] set; an env whose action is illegal does not move and is reported in `illegal`;
            rows of envs that did not move are unspecified."""

    def __init__(self, envs):
        self.envs = list(envs)
        self.state = np.zeros((len(self.envs), 4))
        for i, e in enumerate(self.envs):
            if e.state is not None:
                self.state[i] = e.state
            e.state = self.state[i]                           # a view: render() and friends keep seeing the env's state

    def reset_one(self, i, seed):
        self.state[i] = _seeded_uniform4(seed)
        return self.state[i].astype(np.float32)

    def step_batch(self, actions, go):
        illegal = go & (actions != 0) & (actions != 1)
        move = go & ~illegal
        idx = np.nonzero(move)[0]
        n = len(self.envs)
        obs, reward, term = np.empty((n, 4), np.float32), np.zeros(n), np.zeros(n, bool)
        if len(idx):
            s = self.state[idx]
            x, xd, th, thd = s[:, 0], s[:, 1], s[:, 2], s[:, 3]
            force = np.where(actions[idx] == 1, 10.0, -10.0)
            ct, sn = np.cos(th), np.sin(th)
            temp = (force + 0.05 * thd * thd * sn) / 1.1
            tha = (9.8 * sn - ct * temp) / (0.5 * (4.0 / 3.0 - 0.1 * ct * ct / 1.1))
            xa = temp - 0.05 * tha * ct / 1.1
            new = np.stack([x + 0.02 * xd, xd + 0.02 * xa, th + 0.02 * thd, thd + 0.02 * tha], 1)
            self.state[idx] = new
            obs[idx] = new.astype(np.float32)
            reward[idx] = 1.0
            term[idx] = (np.abs(new[:, 0]) > 2.4) | (np.abs(new[:, 2]) > 12 * 2 * np.pi / 360)
        return obs, reward, term, illegal


class HostCartPoleRender(HostCartPole):
    """HostCartPole with a render(): an H x W x 3 uint8 picture of the cart and the pole (white background, black cart,
    brown pole, a track line) -- a stand-in for CartPole-v1's pygame renderer (400 x 600 frames), which is not part of this
    image.  Drawn with numpy slices: cheap enough to feed a thousand envs from Python."""

    def __init__(self, frame_hw=(400, 600)):
        super().__init__()
        self.H, self.W = int(frame_hw[0]), int(frame_hw[1])
        self._img = None

    def __getstate__(self):                                 # (sent to a worker process: the frame buffer stays behind)
        d = dict(self.__dict__)
        d["_img"] = None
        return d

    def render(self):
        """The frame buffer is reused from call to call (a fresh 720 KB array costs 0.5 ms of page faults): copy it to keep it."""
        H, W = self.H, self.W
        if self._img is None:
            self._img = np.empty((H, W, 3), np.uint8)
            dy, dx = np.meshgrid([-1, 0, 1], [-1, 0, 1], indexing="ij")
            self._dy, self._dx, self._col = dy.ravel(), dx.ravel(), np.array((202, 152, 101), np.uint8)
        img = self._img
        img[...] = 255
        x, _, th, _ = (float(v) for v in self.state)
        cy = int(H * 0.75)
        img[cy + H // 40:cy + H // 40 + 1, :, :] = 0                                       # track
        cx = int(np.clip((x / 4.8 + 0.5) * W, 0, W - 1))
        cw, ch = W // 12, H // 13
        img[max(0, cy - ch // 2):cy + ch // 2, max(0, cx - cw // 2):min(W, cx + cw // 2)] = 0
        k = np.arange(0, H // 4, 2)                                                         # pole: a run of 3x3 dots
        px, py = (cx + k * np.sin(th)).astype(np.int64), (cy - ch // 2 - k * np.cos(th)).astype(np.int64)
        ok = (px >= 1) & (px < W - 1) & (py >= 1) & (py < H - 1)
        img[(py[ok][:, None] + self._dy).ravel(), (px[ok][:, None] + self._dx).ravel()] = self._col
        return img


# ---------------------------------------------------------------------------------------------------------------------------
# observation adapters: what of env.reset / env.step's observation goes into the env's row
# ---------------------------------------------------------------------------------------------------------------------------
def tap_index(n_in, n_out):
    """Source indices of a bilinear resize n_in -> n_out as ATen's upsample_bilinear2d (align_corners False) and
    smz_frames.hip's src_index compute them: src = fma(scale, dst + 0.5, -0.5) in float32, clamped at 0; i0 = min(int(src),
    n_in - 1), i1 = min(i0 + 1, n_in - 1).  Returns int32 [2 * n_out]: (i0, i1) of every output index.  (scale * (dst + 0.5)
    and the subtraction are exact in float64, so rounding the float64 result once IS the float32 fma.)"""
    scale = np.float32(n_in) / np.float32(n_out)
    dst = np.arange(n_out, dtype=np.float64) + 0.5
    src = (np.float64(scale) * dst - 0.5).astype(np.float32)
    src = np.maximum(src, np.float32(0))
    i0 = np.minimum(src.astype(np.int32), n_in - 1)
    i1 = np.minimum(i0 + 1, n_in - 1)
    return np.stack([i0, i1], 1).reshape(-1).astype(np.int32)


class VectorAdapter:
    """Flattened float32 observation vectors (game.py:145-167)."""
    dtype = np.float32

    def __init__(self, obs_dim, transform=None):
        self.row, self.transform = int(obs_dim), transform

    def observe(self, env, obs):
        obs = obs[0] if isinstance(obs, tuple) else obs
        if self.transform is not None:
            obs = self.transform(obs)
        return np.asarray(obs, dtype=np.float32).reshape(-1)


class FrameAdapter:
    """Rendered uint8 frames [H][W][3] (the reference's rgb_observation games, game.py:82-89, 105-107, 142-143).
    upload="frames": the row is the whole frame.  upload="taps": the row holds only the source pixels a bilinear resize to
    out_hw reads -- rows y0/y1 and columns x0/x1 of every output pixel, [2 out_h][2 out_w][3] (98 x 98 from 400 x 600: 115 KB
    instead of 720 KB per frame) -- and the device blends them with the full-frame kernel's own arithmetic
    (smz_frames_resize_taps_u8), so the two uploads give bit-identical observations."""
    dtype = np.uint8

    def __init__(self, frame_hw, out_hw, frame_source="render", upload="frames"):
        assert frame_source in ("render", "obs") and upload in ("frames", "taps")
        self.H, self.W = int(frame_hw[0]), int(frame_hw[1])
        self.out_h, self.out_w = int(out_hw[0]), int(out_hw[1])
        self.frame_source, self.upload = frame_source, upload
        self.row = 2 * self.out_h * 2 * self.out_w * 3 if upload == "taps" else self.H * self.W * 3
        self._lib = None

    def __getstate__(self):                                 # (sent to a worker process: the library handle is per process)
        d = dict(self.__dict__)
        d["_lib"] = None
        return d

    def _taps(self, frame):
        """The gather runs in libsmzhost.so (plain C, smzh_gather_taps_u8: ~12 us per 400 x 600 frame; numpy's fancy indexing
        needs 90-600 us for the same 115 KB) -- the one native piece a worker loads, and not a GPU library."""
        if self._lib is None:
            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsmzhost.so")
            if not os.path.exists(path):
                raise RuntimeError(f"{path} is missing: build it with `make -C stochastic-muzero_amd/csrc host`")
            lib = ctypes.CDLL(path)
            lib.smzh_gather_taps_u8.restype = ctypes.c_int
            lib.smzh_gather_taps_u8.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                                ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
            self._iy, self._ix = tap_index(self.H, self.out_h), tap_index(self.W, self.out_w)
            self._buf = np.empty(self.row, np.uint8)
            self._lib = lib
        frame = np.ascontiguousarray(frame)
        rc = self._lib.smzh_gather_taps_u8(frame.ctypes.data, self.H, self.W, self._iy.ctypes.data, self._iy.size,
                                           self._ix.ctypes.data, self._ix.size, self._buf.ctypes.data)
        assert rc == 0
        return self._buf                                    # (reused from call to call: HostSlice copies it into the env's row)

    def observe(self, env, obs):
        frame = env.render() if self.frame_source == "render" else (obs[0] if isinstance(obs, tuple) else obs)
        frame = np.asarray(frame)
        assert frame.shape == (self.H, self.W, 3), f"frame {frame.shape}, expected {(self.H, self.W, 3)}"
        frame = frame.astype(np.uint8, copy=False)          # (the reference: x.copy().astype(np.uint8), game.py:84)
        if self.upload == "taps":
            return self._taps(frame)
        return frame.reshape(-1)


# ---------------------------------------------------------------------------------------------------------------------------
# the rules around env.step for a slice of envs
# ---------------------------------------------------------------------------------------------------------------------------
class HostSlice:
    """Envs [lo, hi) of a vector env and their rows in the (shared or private) buffers.  Per env it keeps what the reference's
    Game keeps around env.step (game.py:96-131, 223-273):
      * the first observation comes from env.reset(seed=env_seed + global env index [+ 1000003 * game number]);
      * a step that raises is an illegal move: observation unchanged, reward min(-steps so far, -limit, -1), termination
        flag unchanged (game.py:123-131);
      * flags: 1 terminated, 2 stopped by `limit` (game.py:270-271), 3 no step (switched off);
      * on_end "mask": a finished env is switched off (`active`); "reset": it is reset at once -- `obs` gets the fresh
        observation for the next search, `rec` keeps the post-step one for the record and `ended` marks the row."""

    def __init__(self, envs, lo, adapter, arrays, action_map, env_seed, limit, on_end, first_env, batch=True):
        self.envs, self.lo, self.n = list(envs), int(lo), len(envs)
        self.adapter, self.action_map = adapter, list(action_map)
        self.env_seed, self.limit, self.on_end, self.first_env = int(env_seed), int(limit), on_end, int(first_env)
        hi = self.lo + self.n
        for name in ("action", "reward", "flag", "active", "ended", "obs", "rec"):
            setattr(self, name, arrays[name][self.lo:hi])
        self.step_count = np.zeros(self.n, np.int64)
        self.episode = np.zeros(self.n, np.int64)
        self.done = np.zeros(self.n, bool)
        self.batch = self._make_batch() if batch else None

    def _make_batch(self):
        """The slice's batch stepper, when its envs offer one (CartPoleBatch documents the protocol): all envs of ONE class that
        defines make_batch itself, plain vector observations, an integer action map."""
        if not self.envs or not isinstance(self.adapter, VectorAdapter) or self.adapter.transform is not None:
            return None
        cls = type(self.envs[0])
        if any(type(e) is not cls for e in self.envs) or "make_batch" not in vars(cls):
            return None
        if not all(isinstance(a, (int, np.integer)) and not isinstance(a, bool) for a in self.action_map):
            return None
        self._amap = np.asarray(self.action_map, np.int64)
        return cls.make_batch(self.envs)

    def _reset_one(self, i):
        seed = self.env_seed + self.first_env + self.lo + i + 1000003 * int(self.episode[i])
        if self.batch is not None:
            self.obs[i] = np.asarray(self.batch.reset_one(i, seed), np.float32).reshape(-1)
        else:
            self.obs[i] = self.adapter.observe(self.envs[i], self.envs[i].reset(seed=seed))
        self.step_count[i] = 0
        self.done[i] = False

    def reset_all(self):
        self.episode[:] = 0
        for i in range(self.n):
            self._reset_one(i)
        self.active[:] = 1
        self.ended[:] = 0

    def _step_all_batch(self):
        """step_all's rules (below, env by env) as array operations around ONE batch.step_batch call."""
        n, limit = self.n, self.limit
        live = self.active.astype(bool)
        acts = self.action.astype(np.int64)
        in_map = (acts >= 0) & (acts < len(self._amap))       # (amap[acts[i]] raising IndexError is an illegal move too)
        mapped = self._amap[np.where(in_map, acts, 0)]
        seen, r, term, illegal = self.batch.step_batch(mapped, live & in_map)
        illegal = live & (illegal | ~in_map)
        moved = live & ~illegal
        count = self.step_count
        lim = float(limit) if limit > 0 else float("inf")
        if illegal.any():                                     # game.py:123-131: the observation stays, termination flag unchanged
            r = np.where(illegal, np.minimum(np.minimum(-count.astype(np.float64), -lim), -1.0), r)
            term = np.where(illegal, self.done, term)
        count[live] += 1
        f = np.where((limit > 0) & (count == limit), 2, np.where(term, 1, 0)).astype(np.uint8)
        self.done[live] = (term & (f != 2))[live]
        self.reward[:] = np.where(live, r, 0.0)
        self.flag[:] = np.where(live, f, 3)
        self.ended[:] = 0
        if moved.any():
            self.obs[moved] = np.asarray(seen, np.float32).reshape(n, -1)[moved]
        over = np.nonzero(live & (f != 0))[0]
        if len(over):
            if self.on_end == "reset":
                self.rec[over] = self.obs[over]
                self.ended[over] = 1
                self.episode[over] += 1
                for i in over.tolist():
                    self._reset_one(i)
            else:
                self.active[over] = 0

    def step_all(self):
        if self.batch is not None:
            return self._step_all_batch()
        envs, amap, observe = self.envs, self.action_map, self.adapter.observe
        acts, rew, flag, active, ended = self.action.tolist(), self.reward, self.flag, self.active, self.ended
        obs, rec, count = self.obs, self.rec, self.step_count
        limit, reset_mode = self.limit, self.on_end == "reset"
        ended[:] = 0
        live = active.tolist()
        for i in range(self.n):
            if not live[i]:
                flag[i], rew[i] = 3, 0.0
                continue
            env = envs[i]
            try:
                out = env.step(amap[acts[i]])
                seen, r, term = observe(env, out[0]), float(out[1]), bool(out[2])
            except Exception:                             # illegal move (game.py:123-131): the observation stays
                lim = limit if limit > 0 else float("inf")                       # Game's default limit_of_game_play
                seen, r, term = None, float(min(-int(count[i]), -lim, -1)), bool(self.done[i])
            c = count[i] = count[i] + 1
            f = 2 if (limit > 0 and c == limit) else (1 if term else 0)
            self.done[i] = term and f != 2
            rew[i], flag[i] = r, f
            if seen is not None:
                obs[i] = seen
            if f:
                if reset_mode:
                    rec[i] = obs[i]                       # the record keeps the post-step observation ...
                    ended[i] = 1
                    self.episode[i] += 1
                    self._reset_one(i)                    # ... and the next search starts from the fresh one
                else:
                    active[i] = 0

    def close(self):
        for e in self.envs:
            try:
                e.close()
            except Exception:
                pass


# ---------------------------------------------------------------------------------------------------------------------------
# the shared block
# ---------------------------------------------------------------------------------------------------------------------------
def build_env(e):
    """An env object as it is; a class or a zero-argument factory (the gymnasium.vector `env_fns` convention) is called."""
    return e() if isinstance(e, type) or (callable(e) and not hasattr(e, "step")) else e


CMD_STEP, CMD_RESET, CMD_EXIT = 1, 2, 3


def ctrl_bytes(workers):
    """Size of the control region in front of the arrays: whole 4 KB pages holding the GO word, the command slot and one 64-byte
    line per worker's done word (a fixed single page capped the worker count at 60 with a bare assertion)."""
    return (4 * done_word(max(1, int(workers))) + 4095) // 4096 * 4096


def block_layout(B, row, dtype, workers=1):
    """Byte offsets of the arrays of a B-env block, every array on a 4 KB boundary (rows of different workers never share a
    page with another array); the control region in front grows with the worker count."""
    item = np.dtype(dtype).itemsize
    out, off = {"ctrl": ctrl_bytes(workers)}, ctrl_bytes(workers)
    for name, nbytes in (("action", 4 * B), ("reward", 4 * B), ("flag", B), ("active", B), ("ended", B),
                         ("obs", B * row * item), ("rec", B * row * item)):
        out[name] = off
        off += (nbytes + 4095) // 4096 * 4096
    out["total"] = off
    return out


def map_arrays(buf, B, row, dtype, workers):
    lay = block_layout(B, row, dtype, workers)
    arr = dict(action=np.frombuffer(buf, np.int32, B, lay["action"]), reward=np.frombuffer(buf, np.float32, B, lay["reward"]),
               flag=np.frombuffer(buf, np.uint8, B, lay["flag"]), active=np.frombuffer(buf, np.uint8, B, lay["active"]),
               ended=np.frombuffer(buf, np.uint8, B, lay["ended"]),
               obs=np.frombuffer(buf, dtype, B * row, lay["obs"]).reshape(B, row),
               rec=np.frombuffer(buf, dtype, B * row, lay["rec"]).reshape(B, row))
    # control page: int32 word [GO] step sequence (futex, written by the parent); int64 slot [1] command; int32 words
    # [done_word(w)] sequence number worker w has finished (-1: attached, nothing done yet) -- written by worker w ONLY, and the
    # futex word the parent sleeps on while that worker is the one it is waiting for
    ctrl = np.frombuffer(buf, np.int64, lay["ctrl"] // 8, 0)
    if 4 * done_word(workers) > lay["ctrl"]:
        raise ValueError(f"control region of {lay['ctrl']} bytes cannot hold the done words of {workers} workers")
    return arr, ctrl, lay


def done_word(w):
    """int32 index (control page) of worker w's `done` word: one 64-byte line per worker (no false sharing between workers)."""
    return 64 + 16 * w


def control_words(buf, workers=1):
    return np.frombuffer(buf, np.int32, ctrl_bytes(workers) // 4, 0)


class SharedBlock:
    """A file-backed shared mapping created by the parent; the file is unlinked once every worker has attached."""

    def __init__(self, nbytes):
        d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
        fd, self.path = tempfile.mkstemp(prefix="smz_hostenv_", dir=d)
        try:
            os.ftruncate(fd, nbytes)
            self.mm = mmap.mmap(fd, nbytes)
        finally:
            os.close(fd)
        self.nbytes = nbytes

    def unlink(self):
        if self.path:
            try:
                os.unlink(self.path)
            except OSError:
                pass
            self.path = None


# ---- blocking hand-off: futex words in the control page ------------------------------------------------------------------------
# Spinning workers are the wrong default: a container's CPU quota (cgroup cpu.max: 16 CPUs' worth on the 256-thread bench box)
# is burnt by every spinner, and throttling then stalls everybody -- measured: 16 spinning workers 1.9 ms per step, 64 workers
# 6.0 ms, 128 workers 12.4 ms.  Waiters sleep in the kernel on a 32-bit word of the shared control page instead (futex: a
# syscall only when there is something to wait for or somebody to wake) after a spin of a few microseconds.
#   word GO            (ctrl32[0]) step sequence number, written by the parent; workers wait for it to change
#   word done_word(w)  the sequence number worker w has finished: ONE writer per word (a bell word that every worker
#                      incremented was a non-atomic read-modify-write across processes and could go backwards, sending the parent
#                      into a full 10 ms time-out).  The parent sleeps on the word of the first worker that is not done yet.
# The store of a done word is a release store and the parent's read an acquire load (libsmzhost.so, smzh_store_release_i32 /
# smzh_load_acquire_i32) so that a worker's rows are visible before its done word on every architecture; without the library
# (not built) plain numpy stores are used, which is enough on x86's total store order.
import platform  # noqa: E402

# SYS_futex of this machine (None: unknown architecture -> waiters fall back to short sleeps); shared (not PRIVATE) futexes: the
# waiters are other processes
_SYS_FUTEX = {"x86_64": 202, "aarch64": 98, "arm64": 98}.get(platform.machine()) if sys.platform.startswith("linux") else None
_FUTEX_WAIT, _FUTEX_WAKE = 0, 1
_libc = None


class _Timespec(ctypes.Structure):
    _fields_ = [("tv_sec", ctypes.c_long), ("tv_nsec", ctypes.c_long)]


def _futex(addr, op, val, timeout_s=None):
    global _libc
    if _SYS_FUTEX is None:                                  # no futex here: a wait is a nap (the caller re-checks), a wake is nothing
        if op == _FUTEX_WAIT:
            time.sleep(min(timeout_s or 100e-6, 100e-6))
        return 0
    if _libc is None:
        _libc = ctypes.CDLL(None, use_errno=True)
        _libc.syscall.restype = ctypes.c_long
    ts = None
    if timeout_s is not None:
        ts = ctypes.byref(_Timespec(int(timeout_s), int((timeout_s % 1.0) * 1e9)))
    return _libc.syscall(_SYS_FUTEX, ctypes.c_void_p(addr), ctypes.c_int(op), ctypes.c_int(val), ts, None, ctypes.c_int(0))


def futex_wait_change(words, index, seen, timeout_s=0.05, spin=50):
    """Waits until words[index] != seen or the timeout has passed once (a short spin, then ONE kernel wait on the word: the
    caller re-checks and calls again -- spurious wake-ups and time-outs look the same to it)."""
    for _ in range(spin):
        if words[index] != seen:
            return
    _futex(words.ctypes.data + 4 * index, _FUTEX_WAIT, int(seen), timeout_s)


def futex_wake_all(words, index):
    _futex(words.ctypes.data + 4 * index, _FUTEX_WAKE, 0x7fffffff)


GO = 0                                                  # int32 word index into the control page (int64 slot 1 is the command)
_sync_lib = False


def _sync():
    """libsmzhost.so's release / acquire helpers, or None when the library is not there."""
    global _sync_lib
    if _sync_lib is False:
        _sync_lib = None
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsmzhost.so")
        if os.path.exists(path):
            try:
                lib = ctypes.CDLL(path)
                lib.smzh_store_release_i32.restype = None
                lib.smzh_store_release_i32.argtypes = [ctypes.c_void_p, ctypes.c_int32]
                lib.smzh_load_acquire_i32.restype = ctypes.c_int32
                lib.smzh_load_acquire_i32.argtypes = [ctypes.c_void_p]
                _sync_lib = lib
            except (OSError, AttributeError):
                _sync_lib = None
    return _sync_lib


def store_release(words, index, value):
    lib = _sync()
    if lib is None:
        words[index] = value
    else:
        lib.smzh_store_release_i32(words.ctypes.data + 4 * index, int(value))


def load_acquire(words, index):
    lib = _sync()
    return int(words[index]) if lib is None else int(lib.smzh_load_acquire_i32(words.ctypes.data + 4 * index))


def worker_main(spec_path):
    """Entry of a worker process (host_worker.py): attach the block, build the slice, serve step / reset commands."""
    with open(spec_path, "rb") as f:
        spec = pickle.load(f)
    fd = os.open(spec["block_path"], os.O_RDWR)
    try:
        mm = mmap.mmap(fd, spec["nbytes"])
    finally:
        os.close(fd)
    arr, ctrl, _ = map_arrays(mm, spec["B"], spec["row"], spec["dtype"], spec["workers"])
    envs = [build_env(e) for e in spec["envs"]]
    sl = HostSlice(envs, spec["lo"], spec["adapter"], arr, spec["action_map"], spec["env_seed"], spec["limit"], spec["on_end"],
                   spec["first_env"], batch=spec.get("batch", True))
    words = control_words(mm, spec["workers"])
    w, parent = spec["worker"], spec["parent_pid"]
    seq = 0
    mine = done_word(w)
    store_release(words, mine, -1)                         # attached (the parent waits for every worker's -1 -> then unlinks the file)
    try:
        while True:
            futex_wait_change(words, GO, seq, timeout_s=0.25, spin=spec.get("spin", 50))
            if words[GO] == seq:
                if os.getppid() != parent:                 # an orphaned worker exits instead of waiting for ever
                    break
                continue
            seq = load_acquire(words, GO)
            cmd = int(ctrl[1])
            if cmd == CMD_EXIT:
                break
            if cmd == CMD_RESET:
                sl.reset_all()
            else:
                sl.step_all()
            store_release(words, mine, seq)                # (this worker's own word: no read-modify-write shared with others)
            futex_wake_all(words, mine)
    finally:
        sl.close()
    return 0


if __name__ == "__main__":
    sys.exit(worker_main(sys.argv[1]))
