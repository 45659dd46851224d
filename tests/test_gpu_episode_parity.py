"""Episode-length parity of the loop bench.py times: T consecutive env steps of the PRODUCTION path -- one
launch per env step, smz_search_mlp_act_cartpole = k_search_mlp + the env step in its tail -- at BASELINE's size, every
tree's random stream carried from launch to launch (never re-seeded), against the CPU oracle on every tree at every step.

Per step t (self_play.py:79-94 for 4096 envs at once):
  1. the observations the production env holds are searched by a step-wise engine S (same seeds at step 0, streams
     continuing) whose network outputs are recorded as a tape, with the oracle's Dirichlet sample injected;
  2. the 4096 oracle trees (oracle/smz_oracle.c, pinned to the reference's goldens; streams continuing too) replay the tape:
     same leaf / parent / action / branch at every simulation, then S == oracle on visits, all tree arrays, MinMax, root
     value, f64 priors and stream position, bit for bit;
  3. the production launch runs on the same observations: its trees equal the oracle's (float64 priors included, with device-drawn noise),
     its action / policy / child_visits / root value equal orc_act's (which draws from the continuing stream where
     game.py:213 draws), and its stream positions AFTER the action draw equal the oracle's -- so the next step starts from
     the reference's stream state, including the partially twisted MT19937 block (`rng_pos` = ready << 16 | idx) that is
     handed from launch to launch;
  4. the env bookkeeping of the launch's tail: records, flags, switched-off envs (on_end="mask": their trees, streams and
     oracle twins stand still), restarted envs (on_end="reset": the next observation is smz_cartpole_reset_state's).

A search of 50 simulations draws ~1000-1500 words per tree, so every tree crosses a 624-word MT19937 block boundary once or
twice per step, at a different offset of the staged window each time.
monte_carlo_tree_search.py:311-349, game.py:179-232, self_play.py:79-94.
"""
import os
from importlib import import_module

import numpy as np
import pytest
import torch

import golden_util as gu
from test_gpu_fullsize_parity import ALPHA, DISCOUNT, FRAC, assert_engine_equals_oracle, oracle_replay

pytestmark = pytest.mark.gpu


def _pkg(name):
    import stochastic_muzero_amd  # noqa: F401
    return import_module("stochastic-muzero_amd." + name)


def stepwise_tape_step(eng, heads, trees, live, obs, sims, train=True):
    """One whole search of the persistent step-wise engine `eng` on `obs` (streams continue); returns the tape.  Oracle trees
    of live envs get their root (the oracle's Dirichlet sample is injected into the device run)."""
    B, A, S = eng.B, eng.A, eng.S
    hidden, policy = heads.initial(obs)
    torch.cuda.synchronize()
    root_hidden, root_policy = hidden.cpu().numpy().copy(), policy.cpu().numpy().copy()
    noise = np.zeros((B, A), np.float64)
    for i in np.flatnonzero(live):
        noise[i] = trees[i].root_init(root_policy[i], hidden=root_hidden[i], train=train)
    eng.root_init(hidden, policy, train=train, noise_override=torch.from_numpy(noise).cuda())
    tape = []
    for s in range(sims):
        eng.select()
        h2, rw, pol, val = heads.recurrent(eng)
        torch.cuda.synchronize()
        tape.append(dict(action=eng.last_action.cpu().numpy().copy(), branch=eng.branch.cpu().numpy().copy(),
                         parent_hidden=eng.parent_hidden.cpu().numpy()[:, :S].copy(), hidden=h2.cpu().numpy().copy(),
                         reward=rw.cpu().numpy().copy(), policy=pol.cpu().numpy().copy(), value=val.cpu().numpy().copy()))
        eng.expand_backup(h2, rw, pol, val)
    torch.cuda.synchronize()
    return tape


class _Sub:
    """The live trees of an engine, for the every-tree comparison helpers (they index trees 0..len-1)."""

    def __init__(self, eng, idx):
        self.eng, self.idx, self.cfg = eng, idx, eng.cfg

    def root_stats(self):
        sel = torch.from_numpy(self.idx).to(self.eng.device)
        return tuple(t.index_select(0, sel) for t in self.eng.root_stats())

    def dump_tree(self, k):
        return self.eng.dump_tree(int(self.idx[k]))

    def get_rng_state(self, k):
        return self.eng.get_rng_state(int(self.idx[k]))


@pytest.mark.parametrize("on_end,B,sims,T", [("continue", 4096, 50, 8), ("reset", 4096, 50, 8), ("mask", 4096, 50, 8),
                                             # more simulations than LDS holds: the kernels whose trees stay in global memory, with
                                             # the per-tree mask and with restarts (round 5: block-parallel selection there too,
                                             # three passes of blocks at 70 simulations)
                                             ("mask", 4096, 70, 4), ("reset", 4096, 70, 6)])
def test_every_tree_of_every_step_of_the_timed_loop_equals_the_oracle(on_end, B, sims, T):
    import orc
    import stochastic_muzero_amd as smz
    mcts_mod, model_mod, envs_mod, sp = (_pkg(m) for m in ("mcts", "model", "envs", "selfplay"))
    model = model_mod.Muzero.from_arrays(os.path.join(gu.GOLDEN, "weights_ckpt421.npz"))
    heads = model.heads("cuda:0", backend="hip")
    # (mask: everybody is stopped by the limit exactly with the last step, so that every step still has live trees)
    A, S, K, temperature, limit = 2, heads.S, 2, 1.0, {"continue": 0, "reset": 5, "mask": T}[on_end]
    seeds = np.arange(B, dtype=np.uint64) + 4242
    # ---- the production loop's objects (what bench.py builds) ----
    env = envs_mod.CartPoleVec(B, "cuda:0", seed=0, on_end=on_end, limit=limit)
    env.reset()
    env.state[::7, 2] = 0.2; env.state[::7, 3] = 3.0          # falling poles: terminations from step 1 on
    env.obs.copy_(env.state.float())
    m = mcts_mod.BatchedMCTS(B, num_simulations=sims, maxium_action_sample=K, discount=DISCOUNT, root_dirichlet_alpha=ALPHA,
                             root_exploration_fraction=FRAC, use_graph=False, single_launch=True)
    m.seed(seeds)
    chunk = sp.TrajectoryChunk(T, B, 4, A, "cuda:0")
    sp._sync_active(env, m)
    # ---- the step-wise twin + the oracle trees ----
    twin = smz.SearchEngine(B, A, S, num_simulations=sims, maxium_action_sample=K, discount=DISCOUNT,
                            root_dirichlet_alpha=ALPHA, root_exploration_fraction=FRAC)
    twin.seed(seeds)
    twin_active = torch.ones(B, dtype=torch.uint8, device="cuda") if on_end == "mask" else None
    if twin_active is not None:
        twin.set_active(twin_active)
    cfg = orc.make_cfg(A, K, S, sims, discount=DISCOUNT, alpha=ALPHA, frac=FRAC)
    trees = []
    for i in range(B):
        t = orc.Tree(cfg); t.seed(int(seeds[i]))
        trees.append(t)
    n_exact, n_searched, n_resets = 0, 0, 0
    prev_episode = np.zeros(B, np.int64)
    for t in range(T):
        live = np.ones(B, bool) if env.active is None else env.active.cpu().numpy().astype(bool)
        idx = np.flatnonzero(live)
        assert len(idx) > 0
        obs = env.obs.clone()                                 # the launch overwrites env.obs with the next observation
        if twin_active is not None:
            twin_active.copy_(env.active)
        # (1) + (2): step-wise twin == oracle, live trees, bit for bit
        tape = stepwise_tape_step(twin, heads, trees, live, obs, sims)
        live_trees = [trees[i] for i in idx]
        oracle_replay(live_trees, [{k: v[idx] for k, v in rec.items()} for rec in tape])
        assert_engine_equals_oracle(_Sub(twin, idx), live_trees, sims, prior_rtol=0)
        # (3): the production launch (search + action + env step + record) on the same observations
        sp._play_step(env, heads, m, chunk, t, temperature, train=True)
        torch.cuda.synchronize()
        assert m._single is True and m.engine.env_stepped
        eng = m.engine
        action, policy, child_visits, root_value = (x.cpu().numpy().copy() for x in (eng.action, eng.policy, eng.child_visits, eng.root_value))
        oa = [tr.act(temperature) for tr in live_trees]       # draws where game.py:213 draws: the stream goes on
        twin.act(temperature)                                 # ... and so does the twin's
        n_exact += assert_engine_equals_oracle(_Sub(eng, idx), live_trees, sims, prior_rtol=0)
        assert_engine_equals_oracle(_Sub(twin, idx), live_trees, sims, prior_rtol=0)       # (stream position after the draw)
        assert np.array_equal(action[idx], np.array([a[0] for a in oa], np.int32))
        assert np.array_equal(policy[idx], np.stack([a[1] for a in oa]))
        assert np.array_equal(child_visits[idx], np.stack([a[2] for a in oa]))
        assert np.array_equal(root_value[idx], np.array([a[3] for a in oa], np.float32))
        n_searched += len(idx)
        # (4): the record of the step and the env bookkeeping
        row = chunk.data[t].cpu().numpy()
        flags = row[:, 5]
        assert (flags[~live] == 3).all() and (row[~live][:, [0, 1, 2, 3, 4, 6, 7, 8, 9, 10, 11, 12]] == 0).all()
        assert np.array_equal(row[idx, 6:8], policy[idx]) and np.array_equal(row[idx, 11:13], child_visits[idx])
        assert np.array_equal(row[idx, 8 + action[idx]], np.ones(len(idx))) and np.array_equal(row[idx, 10], root_value[idx].astype(np.float64))
        assert set(np.unique(flags[idx])) <= ({0.0, 1.0} if limit == 0 else {0.0, 1.0, 2.0})
        if limit:
            count = env.step_count.cpu().numpy()
            assert ((flags[idx] == 2) <= (count[idx] == (0 if on_end == "reset" else limit))).all()
        if on_end == "mask":
            now = env.active.cpu().numpy().astype(bool)
            assert np.array_equal(now[idx], flags[idx] == 0) and not now[~live].any()
        if on_end == "reset":
            ep = env.episode.cpu().numpy()
            ended = flags != 0
            assert np.array_equal(ep - prev_episode, ended.astype(np.int64))
            nxt = env.obs.cpu().numpy()
            for e in np.flatnonzero(ended)[:64]:               # the next game starts from the counter-based reset state
                assert np.array_equal(nxt[e], env.reset_state_of(int(e), int(ep[e])).astype(np.float32))
                assert not np.array_equal(nxt[e], row[e, :4].astype(np.float32))    # the record keeps the post-step one
            n_resets += int(ended.sum())
            prev_episode = ep.copy()
        else:
            assert np.array_equal(env.obs.cpu().numpy()[idx], row[idx, :4].astype(np.float32))
    if on_end == "mask":
        assert n_searched < B * T and not env.active.cpu().numpy().any()      # everybody stopped by the limit
    if on_end == "reset":
        assert n_resets > B
    # every tree has crossed 624-word block boundaries several times by now (stream position = words drawn so far)
    print(f"[{on_end}] {T} steps x {B} envs x {sims} sims: {n_searched} searches of the one-launch-per-step path == oracle, every "
          f"tree, every step; f64 root priors bit-identical with device-drawn noise in {n_exact}/{n_searched}")
