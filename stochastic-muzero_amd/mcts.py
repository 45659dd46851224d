"""Search objects: the batched many-tree search and the reference-compatible single-tree facade.

  BatchedMCTS              -- B trees on one MI355X; `run(observations, heads, train)` performs a whole
                              Monte_carlo_tree_search.run (monte_carlo_tree_search.py:311-349) for every tree and can
                              replay it as ONE captured HIP graph (select -> heads -> expand/backup x num_simulations).
  Monte_carlo_tree_search  -- same constructor, attributes, assertions and `.run(observation, model, train) -> root`
                              as the reference class (mcts:75-177, 311-349), one tree, any model object exposing the
                              reference's five `*_inference` methods.  The tree lives on the GPU; the process-global
                              numpy stream is carried in and out so the draws are the reference's draws.
"""
import warnings

import numpy as np
import torch

from . import _lib
from .engine import SearchEngine


class Player_cycle:
    """Turn order bookkeeping of mcts:38-72, single-player form (every config of the reference uses 1 player; the
    backup sign is then always + , mcts:302-305)."""

    def __init__(self, number_of_player=None, custom_loop=None):
        if custom_loop is not None and isinstance(custom_loop, str):
            self.cycle_map = [float(i) for i in custom_loop.split(">")]
        elif number_of_player is not None and number_of_player >= 1:
            self.cycle_map = list(range(number_of_player))
        else:
            raise Exception("You have to provide a number of player >= 1 or a custom loop like : \"1>2>3\" ")
        self.number_of_player, self.custom_loop = number_of_player, custom_loop
        self.global_count = 0

    def global_step(self):
        p = self.global_count % len(self.cycle_map)
        self.global_count = (1 + self.global_count) % len(self.cycle_map)
        return p

    def global_reset(self):
        self.global_count = 0

    def proximate_player_step(self, player_index):
        return (player_index + 1) % len(self.cycle_map)


def _validate(pb_c_base, pb_c_init, discount, root_dirichlet_alpha, root_exploration_fraction, num_simulations,
              maxium_action_sample, number_of_player, custom_loop):
    # the reference's assertions, mcts:148-173
    assert isinstance(pb_c_base, int) and pb_c_base >= 1, "pb_c_base ∈ int | {1 < pb_c_base < +inf)"
    assert isinstance(pb_c_init, float) and pb_c_init >= 0, "pb_c_init ∈ float | {0 < pb_c_init < +inf)"
    assert isinstance(discount, (int, float)) and discount >= 0, "discount ∈ float | {0 < discount < +inf)"
    assert isinstance(root_dirichlet_alpha, float) and 0 <= root_dirichlet_alpha <= 1, "root_dirichlet_alpha ∈ float | {0< root_dirichlet_alpha < 1)"
    assert isinstance(root_exploration_fraction, float) and 0 <= root_exploration_fraction <= 1, "root_exploration_fraction ∈ float | {0 < root_exploration_fraction < 1)"
    assert isinstance(maxium_action_sample, int) and maxium_action_sample >= 1, "maxium_action_sample ∈ int | {1 < maxium_action_sample < +inf)"
    assert isinstance(num_simulations, int) and num_simulations >= 0, "num_simulations ∈ int | {0 < num_simulations < +inf)"
    assert isinstance(number_of_player, int) and number_of_player >= 1, "number_of_player ∈ int | {1 < number_of_player < +inf)"
    assert isinstance(custom_loop, str) or custom_loop is None, "custom_loop ∈ str | 1>2>3>3 "
    if number_of_player != 1 or custom_loop is not None:
        raise NotImplementedError("the GPU engine implements the single-player backup (number_of_player=1), the only "
                                  "setting any reference config uses")


class _Hyper:
    def _set_hyper(self, pb_c_base, pb_c_init, discount, root_dirichlet_alpha, root_exploration_fraction,
                   num_simulations, maxium_action_sample, number_of_player, custom_loop):
        _validate(pb_c_base, pb_c_init, discount, root_dirichlet_alpha, root_exploration_fraction, num_simulations,
                  maxium_action_sample, number_of_player, custom_loop)
        self.pb_c_base, self.pb_c_init, self.discount = pb_c_base, pb_c_init, discount
        self.root_dirichlet_alpha, self.root_exploration_fraction = root_dirichlet_alpha, root_exploration_fraction
        self.num_simulations, self.maxium_action_sample = num_simulations, maxium_action_sample
        self.number_of_player, self.custom_loop = number_of_player, custom_loop
        self.cycle = Player_cycle(number_of_player=number_of_player, custom_loop=custom_loop)

    def _engine_kwargs(self):
        return dict(num_simulations=self.num_simulations, maxium_action_sample=self.maxium_action_sample,
                    pb_c_base=self.pb_c_base, pb_c_init=self.pb_c_init, discount=float(self.discount),
                    root_dirichlet_alpha=self.root_dirichlet_alpha,
                    root_exploration_fraction=self.root_exploration_fraction)


SINGLE_LAUNCH_MAX_TREES = 17408      # where the step-wise kernels overtake the single launch (tools/crossover.sh)


def resolve_rng_mode(rng_mode, num_trees):
    if isinstance(rng_mode, str):
        if rng_mode == "auto":
            return _lib.RNG_PHILOX if int(num_trees) > SINGLE_LAUNCH_MAX_TREES else _lib.RNG_MT19937_NUMPY
        return {"mt19937": _lib.RNG_MT19937_NUMPY, "philox": _lib.RNG_PHILOX}[rng_mode]
    return int(rng_mode)


class BatchedMCTS(_Hyper):
    def __init__(self, num_trees, pb_c_base=19652, pb_c_init=1.25, discount=0.95, root_dirichlet_alpha=0.25,
                 root_exploration_fraction=0.25, num_simulations=10, maxium_action_sample=2, number_of_player=1,
                 custom_loop=None, device=None, use_graph=True, fused=True, single_launch=True,
                 rng_mode=_lib.RNG_MT19937_NUMPY):
        self._set_hyper(pb_c_base, pb_c_init, discount, root_dirichlet_alpha, root_exploration_fraction,
                        num_simulations, maxium_action_sample, number_of_player, custom_loop)
        self.num_trees = int(num_trees)
        self.device = device
        # RNG_MT19937_NUMPY (default): the reference's draws, tree i == np.random.seed(seed_i) (parity mode); RNG_PHILOX: counter
        # streams (throughput mode: same numpy-legacy distributions, different numbers); "auto" (what bench.py passes): parity
        # mode up to the single-launch crossover, Philox above it, where the step-wise tree kernel is bandwidth-bound and a
        # 2.5 KB MT19937 state per tree is 28 % of its traffic (+14 % whole job at 262 144 trees, profiles/r05_l_philox_large.txt).
        # NOT the constructor's default: the mode then depends on the tree count, so the same envs sharded over more ranks
        # would draw other numbers (shard invariance, tests/test_gpu_multirank.py, holds within one mode)
        self.rng_mode = resolve_rng_mode(rng_mode, self.num_trees)
        self.use_graph, self.fused, self.single_launch = bool(use_graph), bool(fused), bool(single_launch)
        self.engine = None
        self._graph = None
        self._graph_key = None
        self._graph_heads = None
        self._single = None
        # beyond ~17 k trees the step-wise kernels (64 trees per wavefront, networks as 16-leaf tiles on the matrix cores, rows
        # left in the tree) overtake the single launch: measured on one box (tools/crossover.sh) 440 vs 424 M simulations/s at
        # 16 384 trees, 436 vs 505 at 20 480, 427 vs 586 at 24 576
        self.single_launch_max_trees = SINGLE_LAUNCH_MAX_TREES

    def _ensure_engine(self, num_actions, hidden_size):
        if self.engine is None or (self.engine.A, self.engine.S) != (num_actions, hidden_size):
            if self.engine is not None:
                self.engine.close()
            self.engine = SearchEngine(self.num_trees, num_actions, hidden_size, device=self.device,
                                       rng_mode=self.rng_mode, **self._engine_kwargs())
            if getattr(self, "_active", None) is not None:
                self.engine.set_active(self._active)
            self._graph = None
        return self.engine

    def seed(self, seeds):
        """numpy-style seeding of the per-tree streams; call after the first run() or pass dims explicitly."""
        if self.engine is None:
            self._pending_seed = seeds
        else:
            self.engine.seed(seeds)

    def set_active(self, active):
        """uint8 [num_trees] device tensor (or None): trees whose byte is 0 are skipped by run() and act() -- finished
        games stop consuming simulations (self_play.py:79).  A captured graph holds the pointer, so it is dropped."""
        if active is getattr(self, "_active", None) and (active is None or self.engine is None or
                                                         getattr(self.engine, "_active", None) is active):
            return                          # the same array is bound already: a captured graph stays valid
        self._active = active
        self._graph = None
        if self.engine is not None:
            self.engine.set_active(active)

    def _search(self, obs, heads, train):
        hidden, policy = heads.initial(obs)
        eng = self._ensure_engine(policy.shape[1], hidden.shape[1])
        if getattr(self, "_pending_seed", None) is not None:
            eng.seed(self._pending_seed)
            self._pending_seed = None
        eng.root_init(hidden, policy, train=train)
        if hasattr(heads, "bind_engine"):            # heads that may leave the rows in the tree (large batches)
            want = heads.bind_engine(eng)
        else:
            want = dict(want_mlp_input=getattr(heads, "wants_mlp_input", True),
                        want_parent_hidden=getattr(heads, "wants_parent_hidden", True))
        if self.num_simulations > 0:
            eng.select(**want)
        for s in range(self.num_simulations):
            out = heads.recurrent(eng)
            if self.fused and s + 1 < self.num_simulations:
                eng.expand_backup_select(*out, **want)
            else:
                eng.expand_backup(*out)
                if s + 1 < self.num_simulations:
                    eng.select(**want)

    def _build_graph(self, observations, heads, train, key):
        dev = observations.device
        self._static_obs = observations.clone()
        hidden, policy = heads.initial(self._static_obs)            # learns A and S; draws nothing
        eng = self._ensure_engine(policy.shape[1], hidden.shape[1])
        if getattr(self, "_pending_seed", None) is not None:
            eng.seed(self._pending_seed)
            self._pending_seed = None
        # One throw-away search warms the allocator / hipBLASLt on a side stream (as graph capture requires);
        # the per-tree random streams are snapshotted around it so tree i still is np.random.seed(seed_i).
        eng.snapshot_rng()
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self._search(self._static_obs, heads, train)
        cur.wait_stream(side)
        eng.restore_rng()
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._search(self._static_obs, heads, train)
        self._graph, self._graph_key = g, key

    def run(self, observations, heads, train=True, act_temperature=None, env_step=None, record_obs=None):
        """observations: [B, ...] float32 tensor on the engine's device.  Returns the engine; the search has been
        enqueued on the current stream (read results with engine.root_stats() / engine.act()).
        With HipMlpHeads the whole search is ONE kernel launch (smz_search_mlp) when it fits in LDS; otherwise the
        step-wise kernels run, captured in a HIP graph unless use_graph is off.
        `env_step` (envs.CartPoleVec.fused_step): the single launch also steps the built-in env and appends the record;
        `engine.env_stepped` says whether it did (any other path leaves the env to the caller).
        `record_obs` (a float32 tensor the size of `observations`): the representation launch of the vision family copies the
        frames it reads there (smz_vision_initial_record); `engine.obs_recorded` says whether that happened."""
        eng = self._run(observations, heads, train, act_temperature, env_step, record_obs)
        eng.obs_recorded = record_obs is not None and self._recorded
        return eng

    def _run(self, observations, heads, train, act_temperature, env_step, record_obs):
        self._recorded = False
        # (single_launch_max_trees: where the step-wise kernels overtake the single launch)
        if (self.single_launch and isinstance(getattr(heads, "desc", None), _lib.MlpDesc) and self._single is not False
                and self.num_trees <= self.single_launch_max_trees):
            eng = self._ensure_engine(heads.A, heads.S)
            if getattr(self, "_pending_seed", None) is not None:
                eng.seed(self._pending_seed)
                self._pending_seed = None
            try:
                eng.search_mlp(heads.desc, heads.weights, observations, train=train, act_temperature=act_temperature,
                               env_step=env_step if act_temperature is not None else None)
                self._single = True
                return eng
            except _lib.SmzError as err:
                # only "the working set does not fit a CU's LDS for this geometry" selects the step-wise path (once,
                # said aloud); a HIP error, a bad descriptor or a bad observation tensor is a failure, not a slow path
                if err.code != _lib.SMZ_ERR_TOO_LARGE or self._single is True:
                    raise
                self._single = False
                warnings.warn("single-launch search does not fit in LDS for this batch geometry "
                              f"({self.num_trees} trees x {self.num_simulations} simulations): using the step-wise kernels")
        # vision_model heads: representation per frame (its own launch), then the whole search in one launch
        if (self.single_launch and isinstance(getattr(heads, "desc", None), _lib.VisionDesc) and self._single is not False
                and self.num_trees <= self.single_launch_max_trees):
            hidden, policy = heads.initial(observations, **({} if record_obs is None else dict(record=record_obs)))
            self._recorded = record_obs is not None
            eng = self._ensure_engine(policy.shape[1], hidden.shape[1])
            if getattr(self, "_pending_seed", None) is not None:
                eng.seed(self._pending_seed)
                self._pending_seed = None
            try:
                eng.search_vision(heads.desc, heads.weights, hidden, policy, train=train, act_temperature=act_temperature)
                self._single = True
                return eng
            except _lib.SmzError as err:
                if err.code != _lib.SMZ_ERR_TOO_LARGE or self._single is True:
                    raise
                self._single = False
                warnings.warn("single-launch vision search is outside its limits for this configuration "
                              f"({err}): using the step-wise kernels")
        if not self.use_graph:
            self._search(observations, heads, train)
            return self.engine
        # the captured graph holds raw pointers into `heads` (weights, output buffers): the object is kept alive with the
        # graph and compared by identity -- Muzero.heads() builds a NEW evaluator after every weight update, and a bare
        # id() of a freed one can be handed out again by CPython
        key = (tuple(observations.shape), bool(train))
        if self._graph is None or self._graph_key != key or self._graph_heads is not heads:
            self._build_graph(observations, heads, train, key)
            self._graph_heads = heads
        self._static_obs.copy_(observations)
        self._graph.replay()
        return self.engine


class ChildView:
    __slots__ = ("visit_count", "prior", "value_sum", "reward", "to_play", "is_chance", "children", "hidden_state")

    def value(self):
        return 0 if self.visit_count == 0 else self.value_sum / self.visit_count

    def expanded(self):
        return len(self.children) > 0


def _build_views(dump, A, K, hidden_rows=None):
    """Node objects (mcts:6-21) for one dumped tree; children dicts keyed by action in ascending order."""
    n = dump["n_nodes"]
    nodes = []
    for i in range(n):
        v = ChildView()
        v.visit_count = int(dump["visit"][i])
        v.prior = np.float32(dump["prior"][i])
        v.value_sum = np.float32(dump["value_sum"][i]) if v.visit_count else 0
        v.reward = np.float32(dump["reward"][i])
        v.to_play = 0
        v.is_chance = False
        v.children = {}
        v.hidden_state = None if hidden_rows is None else hidden_rows[i]
        nodes.append(v)
    depth = [0] * n
    for i in range(n):
        cb = int(dump["child_base"][i])
        if cb:
            cnt = A if i == 0 else K
            for j in range(cnt):
                c = cb + j
                depth[c] = depth[i] + 1
                nodes[i].children[np.int64(dump["action"][c])] = nodes[c]
    for i in range(n):
        nodes[i].is_chance = bool((depth[i] >> 1) & 1)
    for a in range(A):
        nodes[1 + a].prior = np.float64(dump["root_priors"][a])
    nodes[0].prior = 0
    return nodes[0]


class Monte_carlo_tree_search(_Hyper):
    """Drop-in for the reference class of the same name (mcts:75-349), one tree per object."""

    def __init__(self, pb_c_base=19652, pb_c_init=1.25, discount=0.95, root_dirichlet_alpha=0.25,
                 root_exploration_fraction=0.25, num_simulations=10, maxium_action_sample=2, number_of_player=1,
                 custom_loop=None):
        self.reset(pb_c_base, pb_c_init, discount, root_dirichlet_alpha, root_exploration_fraction, num_simulations,
                   maxium_action_sample, number_of_player, custom_loop)

    def reset(self, pb_c_base=19652, pb_c_init=1.25, discount=0.95, root_dirichlet_alpha=0.25,
              root_exploration_fraction=0.25, num_simulations=10, maxium_action_sample=2, number_of_player=1,
              custom_loop=None):
        self._set_hyper(pb_c_base, pb_c_init, discount, root_dirichlet_alpha, root_exploration_fraction,
                        num_simulations, maxium_action_sample, number_of_player, custom_loop)
        self.node = None
        self.model = None
        self.root = None
        self._engine = None

    def _dev(self, a, dtype=torch.float32):
        t = torch.as_tensor(np.ascontiguousarray(a)) if not torch.is_tensor(a) else a
        return t.detach().to(dtype).reshape(1, -1).contiguous().to(self._engine.device if self._engine else "cuda")

    def run(self, observation=None, model=None, train=True, use_global_numpy_stream=True):
        """The reference's run(): `model` is any object with the five *_inference methods (batch 1, CPU tensors).
        The numpy global stream is imported before and exported after the search, so `np.random.seed(s)` followed
        by this run() consumes the same draws as the reference and leaves the stream where the reference leaves it
        (game.py:213 continues from there)."""
        self.model = model
        h0 = model.representation_function_inference(observation)
        self.cycle.global_step()
        policy, _value = model.prediction_function_inference(h0)     # value discarded (mcts:319-321)
        h0_flat = torch.as_tensor(h0).detach().reshape(1, -1).float()
        A, S = int(np.asarray(policy).shape[-1]), int(h0_flat.shape[1])
        if self._engine is None or (self._engine.A, self._engine.S) != (A, S):
            self._engine = SearchEngine(1, A, S, **self._engine_kwargs())
        eng = self._engine
        if use_global_numpy_stream:
            _, key, pos, *_ = np.random.get_state()
            eng.set_rng_state(0, key, pos)
        shape = tuple(torch.as_tensor(h0).shape)
        eng.root_init(h0_flat.cuda().contiguous(), torch.as_tensor(np.asarray(policy, np.float32)).reshape(1, A).cuda(),
                      train=bool(train))
        for _ in range(self.num_simulations):
            ph, la, br, _x = eng.select(want_mlp_input=False)
            torch.cuda.synchronize()
            parent_h = ph[:, :S].cpu().reshape(shape)
            action = int(la[0])
            if int(br[0]):
                reward, hidden = model.dynamics_function_inference(parent_h, action)
                pol, value = model.prediction_function_inference(hidden)
            else:
                reward = 0.0
                hidden = model.afterstate_dynamics_function_inference(parent_h, action)
                pol, value = model.afterstate_prediction_function_inference(hidden)
            eng.expand_backup(torch.as_tensor(hidden).detach().reshape(1, -1).float().cuda().contiguous(),
                              torch.tensor([float(reward)], dtype=torch.float32, device="cuda"),
                              torch.as_tensor(np.asarray(pol, np.float32)).reshape(1, A).cuda(),
                              torch.tensor([np.float32(value)], dtype=torch.float32, device="cuda"))
        torch.cuda.synchronize()
        if use_global_numpy_stream:
            key, pos = eng.get_rng_state(0)
            np.random.set_state(("MT19937", key, pos, 0, 0.0))
        self.root = _build_views(eng.dump_tree(0), A, eng.K)
        self.root.hidden_state = h0
        return self.root
