"""Vectorised synthetic environments resident on the GPU (the "synthetic fixed-length episodes" of BASELINE.json).

gymnasium is not part of this image, so the environments here are this engine's own: a CartPole-v1 shaped Euler
integrator (float64 state, float32 observations; physics constants as published for CartPole-v1) and a
LunarLander-shaped stand-in that only provides observations of the right width.  Env i of a job draws its initial
state from `RandomState(seed).uniform(...)[i]`, so a shard sees exactly the rows it would see in a single-GPU run.
"""
import ctypes as C
import os

import numpy as np
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


class HostCartPole:
    """One CartPole-v1 shaped game on the host behind the gym call shape (reset(seed=) -> (obs, info); step(a) -> (obs,
    reward, terminated, truncated, info)): float64 Euler physics with CartPole-v1's published constants, the arithmetic
    of smz_cartpole_step.  gymnasium is not part of this build; this class is what the host-environment path is
    exercised with, and what `muzero_cli.py` uses for a single-game run."""
    metadata = {"render_fps": 50}

    def __init__(self):
        self.state = None

    def reset(self, seed=None):
        self.state = np.random.RandomState(seed).uniform(-0.05, 0.05, size=4)
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


class HostVecEnv:
    """B environments that live on the HOST (gymnasium-style objects) behind the interface the batched loop drives
    (SURVEY 8f-4): per env step one pinned-memory download of the B actions and one pinned-memory upload of the B
    observations / rewards / flags, both asynchronous on the engine's stream; the only host wait is for the actions.

    `envs`: a list of B single environments (reset(seed=) -> obs | (obs, info); step(a) -> (obs, reward, terminated, ...)).
    Per env the wrapper keeps what the reference's Game keeps around env.step (game.py:96-131, 223-273):
      * the first observation comes from env.reset(seed=env_seed + global env index) (game.py:102 draws that seed from
        Python's unseeded `random`; a reproducible rule replaces the draw);
      * a step that raises is an illegal move: observation unchanged, reward min(-steps so far, -limit, -1), termination
        flag unchanged (game.py:123-131);
      * flags as smz_cartpole_step_ctl: 1 terminated, 2 stopped by `limit` (game.py:270-271), 3 no step (switched off);
      * on_end "mask": a finished env is switched off (`active`, handed to the search); "reset": it is reset at once and
        the NEXT search sees the fresh observation while the record keeps the post-step one (`record_obs`).
    Observations are flattened float32 vectors (game.py:145-167); rendered RGB frames: HostImageVecEnv."""

    def __init__(self, envs, obs_dim, num_actions, device, action_map=None, env_seed=0, limit=0, on_end="reset", first_env=0,
                 transform=None):
        assert on_end in ("mask", "reset")
        self.envs, self.B = list(envs), len(envs)
        self.obs_dim, self.num_actions = int(obs_dim), int(num_actions)
        self.device = torch.device(device)
        self.action_map = list(action_map) if action_map is not None else list(range(self.num_actions))
        self.env_seed, self.limit, self.on_end, self.first_env = int(env_seed), int(limit), on_end, int(first_env)
        self.transform = transform
        B = self.B
        pin = dict(pin_memory=torch.cuda.is_available())
        self._h_reward = torch.zeros(B, dtype=torch.float32, **pin)
        self._h_flag = torch.zeros(B, dtype=torch.uint8, **pin)
        self._h_active = torch.ones(B, dtype=torch.uint8, **pin)
        self._h_action = torch.zeros(B, dtype=torch.int32, **pin)
        self.reward = torch.zeros(B, dtype=torch.float32, device=self.device)
        self.terminated = torch.zeros(B, dtype=torch.uint8, device=self.device)
        self.active = torch.ones(B, dtype=torch.uint8, device=self.device) if on_end == "mask" else None
        self.step_count = np.zeros(B, np.int64)
        self.episode = np.zeros(B, np.int64)
        self.done = np.zeros(B, bool)
        self.transfer_seconds = 0.0            # host time spent waiting for the action download (diagnostic)
        self._alloc_observations(pin)

    # ---- observation storage (vector observations; HostImageVecEnv overrides these four) -------------------------------
    def _alloc_observations(self, pin):
        B, o = self.B, self.obs_dim
        self._h_obs = torch.zeros(B, o, dtype=torch.float32, **pin)          # next search's input
        self._h_rec = torch.zeros(B, o, dtype=torch.float32, **pin)          # post-step observation (the record's)
        self.obs = torch.zeros(B, o, dtype=torch.float32, device=self.device)
        self.record_obs = torch.zeros(B, o, dtype=torch.float32, device=self.device)

    def _observe(self, env, obs):
        """What the agent sees after env.reset / env.step returned `obs`."""
        obs = obs[0] if isinstance(obs, tuple) else obs
        if self.transform is not None:
            obs = self.transform(obs)
        return np.asarray(obs, dtype=np.float32).reshape(-1)

    def _store(self, i, seen, after_step):
        """Keeps env i's observation: after a step it is both the record's and (until a reset replaces it) the next
        search's input; after a reset only the latter."""
        row = torch.from_numpy(seen)
        self._h_obs[i] = row
        if after_step:
            self._h_rec[i] = row

    def _keep(self, i):
        """A step that did not change env i's observation (illegal move, game.py:123-131): the record's row is the current one."""
        self._h_rec[i] = self._h_obs[i]

    def _upload_observations(self, after_step):
        self.obs.copy_(self._h_obs, non_blocking=True)
        if after_step:
            self.record_obs.copy_(self._h_rec, non_blocking=True)

    # ---- the loop's interface --------------------------------------------------------------------------------------------
    def _reset_one(self, i):
        seed = self.env_seed + self.first_env + i + 1000003 * int(self.episode[i])
        self._store(i, self._observe(self.envs[i], self.envs[i].reset(seed=seed)), after_step=False)
        self.step_count[i] = 0
        self.done[i] = False

    def reset(self):
        self.episode[:] = 0
        for i in range(self.B):
            self._reset_one(i)
        self._h_active.fill_(1)
        self._upload_observations(after_step=False)
        if self.active is not None:
            self.active.copy_(self._h_active, non_blocking=True)
        return self.obs

    def step(self, action):
        import time
        stream = torch.cuda.current_stream(self.device)
        self._h_action.copy_(action, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(stream)
        t0 = time.perf_counter()
        ev.synchronize()                                  # the search of this step has to finish before the env can move
        self.transfer_seconds += time.perf_counter() - t0
        acts = self._h_action.numpy()
        rew, flag, act_h = self._h_reward.numpy(), self._h_flag.numpy(), self._h_active.numpy()
        for i, env in enumerate(self.envs):
            if not act_h[i]:
                flag[i], rew[i] = 3, 0.0
                continue
            try:
                out = env.step(self.action_map[int(acts[i])])
                seen, r, term = self._observe(env, out[0]), float(out[1]), bool(out[2])
            except Exception:                             # illegal move (game.py:123-131): the observation stays
                limit = self.limit if self.limit > 0 else float("inf")          # Game's default limit_of_game_play
                seen, r, term = None, float(min(-int(self.step_count[i]), -limit, -1)), bool(self.done[i])
            self.step_count[i] += 1
            f = 2 if (self.limit > 0 and self.step_count[i] == self.limit) else (1 if term else 0)
            self.done[i] = term and f != 2
            rew[i], flag[i] = r, f
            if seen is not None:
                self._store(i, seen, after_step=True)
            else:
                self._keep(i)                             # illegal move: the record shows the unchanged observation
            if f:
                if self.on_end == "reset":
                    self.episode[i] += 1
                    self._reset_one(i)
                else:
                    act_h[i] = 0
        self._upload_observations(after_step=True)
        self.reward.copy_(self._h_reward, non_blocking=True)
        self.terminated.copy_(self._h_flag, non_blocking=True)
        if self.active is not None:
            self.active.copy_(self._h_active, non_blocking=True)
        return self.obs, self.reward, self.terminated

    def close(self):
        for e in self.envs:
            e.close()


class HostImageVecEnv(HostVecEnv):
    """HostVecEnv for environments observed through rendered RGB frames (the reference's rgb_observation games:
    game.py:82-89, 105-107, 142-143 -- every frame through ToTensor + Resize(98, 98), one at a time on the CPU).

    Per env step the B uint8 frames [H][W][3] go up through ONE pinned buffer as they are (3 bytes per pixel instead of 12)
    and one launch of smz_frames_resize_u8 on the engine's stream turns them into the [B,3,98,98] float32 tensor the vision
    heads read.  frame_source "render": the frame is env.render() (what the reference's Game.render shows the agent);
    "obs": the env's observation itself is the frame.  With on_end="reset" the few envs that finished a game this step have
    two frames -- the post-step one for the record and the reset one for the next search: the post-step ones travel in a
    small side buffer and are resized into `record_obs` rows by a second, indexed launch."""
    frame = (3, 98, 98)

    def __init__(self, envs, frame_hw, num_actions, device, out_hw=(98, 98), frame_source="render", **kw):
        assert frame_source in ("render", "obs")
        self.H, self.W = int(frame_hw[0]), int(frame_hw[1])
        self.out_h, self.out_w = int(out_hw[0]), int(out_hw[1])
        self.frame = (3, self.out_h, self.out_w)
        self.frame_source = frame_source
        self.lib = _lib.load()
        super().__init__(envs, 3 * self.out_h * self.out_w, num_actions, device, **kw)

    def _alloc_observations(self, pin):
        B, H, W = self.B, self.H, self.W
        self._h_frames = torch.zeros(B, H, W, 3, dtype=torch.uint8, **pin)        # next search's frames
        self._h_frames_np = self._h_frames.numpy()
        self._d_frames = torch.zeros(B, H, W, 3, dtype=torch.uint8, device=self.device)
        self.obs = torch.zeros((B,) + self.frame, dtype=torch.float32, device=self.device)
        self.record_obs = torch.zeros((B,) + self.frame, dtype=torch.float32, device=self.device) if self.on_end == "reset" else None
        self._side_cap = 0
        self._ended = []                                                          # (env, post-step frame) of this step
        self.upload_bytes = 0                                                     # diagnostic: PCIe bytes of frames so far

    def _observe(self, env, obs):
        frame = env.render() if self.frame_source == "render" else (obs[0] if isinstance(obs, tuple) else obs)
        frame = np.asarray(frame)
        assert frame.shape == (self.H, self.W, 3), f"frame {frame.shape}, expected {(self.H, self.W, 3)}"
        return frame.astype(np.uint8, copy=False)          # (the reference: x.copy().astype(np.uint8), game.py:84)

    def _store(self, i, seen, after_step):
        if not after_step and self._stepping:
            # a reset inside step(): the frame in the main buffer is this env's post-step one -- the record needs it
            self._ended.append((i, self._h_frames[i].clone()))
        np.copyto(self._h_frames_np[i], seen)          # (a numpy view of the pinned buffer: no tensor-indexing overhead per env)

    def _keep(self, i):
        pass                                               # the frame in the main buffer is still the current one

    _stepping = False

    def step(self, action):
        self._stepping, self._ended = True, []
        try:
            return super().step(action)
        finally:
            self._stepping = False

    def _resize(self, frames_dev, n, rows_dev, out):
        P = lambda x: None if x is None else C.c_void_p(x.data_ptr())
        _lib.check(self.lib.smz_frames_resize_u8(P(frames_dev), n, self.H, self.W, self.out_h, self.out_w, P(rows_dev), P(out),
                                                 C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def _upload_observations(self, after_step):
        self._d_frames.copy_(self._h_frames, non_blocking=True)
        self.upload_bytes += self._h_frames.numel()
        self._resize(self._d_frames, self.B, None, self.obs)
        if not after_step or self.record_obs is None:
            return
        self.record_obs.copy_(self.obs)
        n = len(self._ended)
        if n == 0:
            return
        if n > self._side_cap:                         # grows to the largest number of simultaneous game ends seen
            pin = dict(pin_memory=torch.cuda.is_available())
            self._side_cap = max(n, 2 * self._side_cap, 8)
            self._h_side = torch.zeros(self._side_cap, self.H, self.W, 3, dtype=torch.uint8, **pin)
            self._h_rows = torch.zeros(self._side_cap, dtype=torch.int32, **pin)
            self._d_side = torch.zeros(self._side_cap, self.H, self.W, 3, dtype=torch.uint8, device=self.device)
            self._d_rows = torch.zeros(self._side_cap, dtype=torch.int32, device=self.device)
        else:
            torch.cuda.current_stream(self.device).synchronize()      # the side buffers of the previous use have been read
        for k, (i, fr) in enumerate(self._ended):
            self._h_side[k] = fr
            self._h_rows[k] = i
        self._d_side[:n].copy_(self._h_side[:n], non_blocking=True)
        self._d_rows[:n].copy_(self._h_rows[:n], non_blocking=True)
        self.upload_bytes += n * self.H * self.W * 3
        self._resize(self._d_side, n, self._d_rows, self.record_obs)


class HostCartPoleRender(HostCartPole):
    """HostCartPole with a render(): an H x W x 3 uint8 picture of the cart and the pole (white background, black cart,
    brown pole, a track line) -- a stand-in for CartPole-v1's pygame renderer (400 x 600 frames), which is not part of this
    image.  Drawn with numpy slices: cheap enough to feed a thousand envs from Python."""

    def __init__(self, frame_hw=(400, 600)):
        super().__init__()
        self.H, self.W = int(frame_hw[0]), int(frame_hw[1])
        self._img = None

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

    def step(self, action):
        import time
        stream = torch.cuda.current_stream(self.device)
        self._h_action.copy_(action, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(stream)
        t0 = time.perf_counter()
        ev.synchronize()                                  # the search of this step has to finish before the env can move
        self.transfer_seconds += time.perf_counter() - t0
        P = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(self.lib.smz_host_cartpole_step(P(self._state), P(self._h_action), P(self._h_obs), P(self._h_reward),
                                                   P(self._h_flag), P(self._count), self.limit, self.B))
        self.obs.copy_(self._h_obs, non_blocking=True)
        self.reward.copy_(self._h_reward, non_blocking=True)
        self.terminated.copy_(self._h_flag, non_blocking=True)
        return self.obs, self.reward, self.terminated
