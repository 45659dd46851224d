"""Vectorised synthetic environments resident on the GPU (the "synthetic fixed-length episodes" of BASELINE.json).

gymnasium is not part of this image, so the environments here are this engine's own: a CartPole-v1 shaped Euler
integrator (float64 state, float32 observations; physics constants as published for CartPole-v1) and a
LunarLander-shaped stand-in that only provides observations of the right width.  Env i of a job draws its initial
state from `RandomState(seed).uniform(...)[i]`, so a shard sees exactly the rows it would see in a single-GPU run.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


class CartPoleVec:
    obs_dim, num_actions = 4, 2

    def __init__(self, num_envs, device, seed=0, first_env=0, total_envs=None):
        self.lib = _lib.load()
        self.B, self.device = int(num_envs), torch.device(device)
        self.seed, self.first_env = int(seed), int(first_env)
        self.total = int(total_envs) if total_envs is not None else self.first_env + self.B
        self.state = torch.empty(self.B, 4, dtype=torch.float64, device=self.device)
        self.obs = torch.empty(self.B, 4, dtype=torch.float32, device=self.device)
        self.reward = torch.empty(self.B, dtype=torch.float32, device=self.device)
        self.terminated = torch.empty(self.B, dtype=torch.uint8, device=self.device)

    def reset(self):
        all_states = np.random.RandomState(self.seed).uniform(-0.05, 0.05, size=(self.total, 4))
        st = all_states[self.first_env:self.first_env + self.B]
        self.state.copy_(torch.from_numpy(np.ascontiguousarray(st)))
        self.obs.copy_(self.state.to(torch.float32))
        self.terminated.zero_()
        return self.obs

    def step(self, action):
        """action: int32 [B] device tensor (index into action_map).  Asynchronous on the current stream."""
        s = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self.lib.smz_cartpole_step(C.c_void_p(self.state.data_ptr()), C.c_void_p(action.data_ptr()),
                                              C.c_void_p(self.obs.data_ptr()), C.c_void_p(self.reward.data_ptr()),
                                              C.c_void_p(self.terminated.data_ptr()), self.B, s))
        return self.obs, self.reward, self.terminated

    def step_and_record(self, action, chunk_data, t, policy, child_visits, root_value):
        """step(action) + the trajectory record of this step (smz_traj_pack's layout) in one launch."""
        s = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        P = lambda x: C.c_void_p(x.data_ptr())
        _lib.check(self.lib.smz_cartpole_step_pack(P(self.state), P(action), P(self.obs), P(self.reward), P(self.terminated),
                                                   P(chunk_data), chunk_data.shape[0], int(t), P(policy), P(child_visits),
                                                   P(root_value), self.B, s))
        return self.obs, self.reward, self.terminated


class SyntheticVec:
    """Observation-only stand-in (e.g. LunarLander-shaped: obs 8 ~ N(0,1), 4 actions; Box2D is absent here)."""

    def __init__(self, num_envs, obs_dim, num_actions, device, seed=0, first_env=0, total_envs=None):
        self.B, self.obs_dim, self.num_actions = int(num_envs), int(obs_dim), int(num_actions)
        self.device = torch.device(device)
        self.seed, self.first_env = int(seed), int(first_env)
        self.total = int(total_envs) if total_envs is not None else self.first_env + self.B
        self.obs = torch.empty(self.B, self.obs_dim, dtype=torch.float32, device=self.device)
        self.reward = torch.zeros(self.B, dtype=torch.float32, device=self.device)
        self.terminated = torch.zeros(self.B, dtype=torch.uint8, device=self.device)
        self._t = 0

    def _draw(self):
        g = np.random.RandomState(self.seed + 7919 * self._t)
        rows = g.standard_normal(size=(self.total, self.obs_dim)).astype(np.float32)
        self.obs.copy_(torch.from_numpy(rows[self.first_env:self.first_env + self.B]), non_blocking=False)
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
    Env i starts from RandomState(seed + i).rand(3,98,98); a step scrolls the frame one pixel, on the device."""
    frame = (3, 98, 98)

    def __init__(self, num_envs, num_actions, device, seed=0, first_env=0, total_envs=None):
        self.B, self.num_actions = int(num_envs), int(num_actions)
        self.obs_dim = int(np.prod(self.frame))
        self.device = torch.device(device)
        self.seed, self.first_env = int(seed), int(first_env)
        self.obs = torch.empty((self.B,) + self.frame, dtype=torch.float32, device=self.device)
        self.reward = torch.zeros(self.B, dtype=torch.float32, device=self.device)
        self.terminated = torch.zeros(self.B, dtype=torch.uint8, device=self.device)

    def reset(self):
        rows = np.stack([np.random.RandomState(self.seed + self.first_env + i).rand(*self.frame).astype(np.float32)
                         for i in range(self.B)])
        self.obs.copy_(torch.from_numpy(rows))
        return self.obs

    def step(self, action):
        self.obs.copy_(torch.roll(self.obs, shifts=1, dims=3))
        return self.obs, self.reward, self.terminated
