"""Trajectory gather: the only inter-GPU exchange of the self-play path.

Games are independent, so ranks never exchange tree state; after a fixed-length chunk each rank owns one contiguous
[T][B][F] float64 slab and the learner rank (0) needs all of them for ReplayBuffer.save_game.  With torch.distributed's
"nccl" backend (= RCCL on ROCm) this is one grouped send/recv: every GPU has a direct xGMI link to rank 0, so the 7
receives proceed on 7 distinct links and no ring is involved.  "gloo" runs the same code on CPU tensors (tests).
"""
import torch
import torch.distributed as dist


def peer_sizes(n_local, device, group=None, total_envs=None):
    """Env count (dim 1 of a slab) of every rank's shard.  Shards may differ by one env (shard_range), so the learner cannot
    size its receive buffers from its own slab.  Whether a collective is issued must NOT depend on anything rank-local
    (a cache keyed on the local size let rank 0 skip the all_gather that rank 1 entered when a process gathered
    for 9 envs and then for 8 -- shards 4|5, then 4|4):
      * `total_envs` given: the sizes are shard_range's on every rank, no communication; a slab that is not this rank's
        shard of that total is an error here, before any receive is posted;
      * otherwise the sizes are exchanged with an 8-byte all_gather on EVERY call (every rank enters it)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if total_envs is not None:
        sizes = [hi - lo for lo, hi in (shard_range(int(total_envs), r, world) for r in range(world))]
        if sizes[rank] != int(n_local):
            raise ValueError(f"rank {rank} holds {int(n_local)} envs, but its shard of {int(total_envs)} envs over {world} "
                             f"ranks has {sizes[rank]} (gather.shard_range)")
        return sizes
    mine = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    return [int(t.item()) for t in out]


def _like(slab, n_envs):
    shape = list(slab.shape)
    shape[1] = n_envs
    return torch.empty(shape, dtype=slab.dtype, device=slab.device)


def gather_to_learner(slab, dst=0, group=None, total_envs=None, sizes=None):
    """Returns the list of every rank's slab [T][B_rank][...] on rank `dst` (rank order), None elsewhere.
    `total_envs` (the job's env count) or `sizes` (every rank's shard size) spare the size exchange (peer_sizes)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [slab]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    slab = slab.contiguous()
    if dist.get_backend(group) == "gloo" and slab.is_cuda:     # functional runs without RCCL: stage through the host
        parts = gather_to_learner(slab.cpu(), dst, group, total_envs, sizes)
        return None if parts is None else [p.to(slab.device) for p in parts]
    if sizes is None:
        sizes = peer_sizes(slab.shape[1], slab.device, group, total_envs)
    if rank == dst:
        parts = [slab if r == dst else _like(slab, sizes[r]) for r in range(world)]
        ops = [dist.P2POp(dist.irecv, parts[r], r, group) for r in range(world) if r != dst]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        return parts
    for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, slab, dst, group)]):
        w.wait()
    return None


def shard_range(total_envs, rank, world):
    """Contiguous env shard of a rank: env i keeps its seed and initial state whatever the world size.  Shards differ by at
    most one env; a job with fewer envs than ranks is refused -- an engine for
    zero trees cannot be built, and a rank without work would still sit in every collective."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside a world of {world}")
    if total_envs < world:
        raise ValueError(f"{total_envs} envs cannot be sharded over {world} ranks: every rank needs at least one env")
    return total_envs * rank // world, total_envs * (rank + 1) // world


def pack_records(data, obs_dim, A):
    """Wire format of a [..., F] float64 trajectory record (F = obs_dim + 3 A + 3: observation | reward | flag | policy |
    action one-hot | root value | child visits): the fields that ARE float32 values or small integers -- observation (the env's
    float32 row), flag, one-hot action, root value (numpy float32 in the reference, game.py:172) -- travel as float32, the
    float64 ones (reward: a Python float from the env; policy, child visits: numpy float64, game.py:184-209) stay float64.
    CartPole: 8 x 4 + 5 x 8 = 72 bytes per env step instead of 13 x 8 = 104.  Lossless (unpack_records inverts it bit for bit)."""
    o = int(obs_dim)
    narrow = torch.cat([data[..., :o], data[..., o + 1:o + 2], data[..., o + 2 + A:o + 3 + 2 * A]], -1).to(torch.float32)
    wide = torch.cat([data[..., o:o + 1], data[..., o + 2:o + 2 + A], data[..., o + 3 + 2 * A:]], -1).contiguous()
    return narrow, wide


def unpack_records(narrow, wide, obs_dim, A):
    o = int(obs_dim)
    n = narrow.to(torch.float64)
    return torch.cat([n[..., :o], wide[..., :1], n[..., o:o + 1], wide[..., 1:1 + A], n[..., o + 1:o + 2 + A], wide[..., 1 + A:]], -1)


class TrajectoryGather:
    """The trajectory exchange as an object: compact wire format and overlap with the search.

        tg = TrajectoryGather(obs_dim, A, slices=4)
        for each slice of the chunk's steps:   play the slice;  tg.start(chunk.data[t0:t1] [, chunk.obs[t0:t1]])
        parts = tg.finish()                    # learner: (records [T][B_total][F] float64, frames or None); actors: None

    start() returns at once: on a side stream that first waits for the current (search) stream, the rows are packed
    (pack_records) and handed to the grouped send / receive, so slice k's transfer over xGMI runs while the search of slice
    k + 1 computes; only the last slice's transfer is exposed.  finish() makes the current stream wait for everything and
    assembles the learner's tensors.  With "gloo" (functional runs: ranks sharing a GPU, CPU tests) the same calls go
    through the host synchronously.  Usable wherever a plain `gather(slab) -> parts` callable is (it is one: the whole
    slab as a single slice)."""

    def __init__(self, obs_dim, A, dst=0, group=None, slices=4, compact=True, total_envs=None, loopback=False):
        """`total_envs`: the job's env count -- every rank then derives the shard sizes itself (shard_range) and no size
        exchange happens; without it the sizes are exchanged once per chunk (at the first start() after a finish(), an
        8-byte all_gather every rank enters).  `loopback`: a world of ONE rank still goes through the exchange -- the rank
        posts its isend and the matching irecv to itself in one group (RCCL executes a self send / receive inside a
        group call), so the side stream, the grouped P2P and the record_stream bookkeeping run on a single-GPU box
        (tests/test_gpu_rccl_loopback.py)."""
        self.o, self.A, self.dst, self.group, self.slices, self.compact = int(obs_dim), int(A), dst, group, max(1, int(slices)), compact
        self.total_envs, self.loopback = total_envs, bool(loopback)
        self._side, self._pending, self._sizes = None, [], None
        self.exposed_ms = None

    def _active(self):
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size(self.group) > 1 or self.loopback

    def start(self, data, frames=None):
        if not self._active():
            self._pending.append(((data, None), frames, None))
            return
        nccl = dist.get_backend(self.group) == "nccl" and data.is_cuda
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        if self._sizes is None:                                  # once per chunk, on every rank alike
            self._sizes = [int(data.shape[1])] if world == 1 else peer_sizes(data.shape[1], data.device, self.group, self.total_envs)
        sizes = self._sizes
        if sizes[rank] != int(data.shape[1]):
            raise ValueError(f"slice with {int(data.shape[1])} envs inside a chunk that started with {sizes[rank]}")
        if not nccl:                                             # gloo: staged through the host, synchronous
            msgs = list(pack_records(data, self.o, self.A)) if self.compact else [data.contiguous()]
            if frames is not None:
                msgs.append(frames.contiguous())
            if world == 1:                                       # (loopback: what a receive from oneself yields)
                got = [[m.clone()] for m in msgs]
            else:
                got = [gather_to_learner(m, self.dst, self.group, sizes=sizes) for m in msgs]
            self._pending.append((None, None, got))
            return
        cur = torch.cuda.current_stream(data.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=data.device)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            msgs = list(pack_records(data, self.o, self.A)) if self.compact else [data.contiguous()]
            if frames is not None:
                msgs.append(frames.contiguous())
            for m in msgs:
                m.record_stream(self._side)
            if world == 1:                                       # loopback: send to and receive from oneself, one group
                bufs = [[_like(m, sizes[0])] for m in msgs]
                ops = [dist.P2POp(dist.isend, m, 0, self.group) for m in msgs] + [dist.P2POp(dist.irecv, b[0], 0, self.group) for b in bufs]
            elif rank == self.dst:
                bufs = [[m if r == rank else _like(m, sizes[r]) for r in range(world)] for m in msgs]
                ops = [dist.P2POp(dist.irecv, b[r], r, self.group) for b in bufs for r in range(world) if r != rank]
            else:
                bufs, ops = None, [dist.P2POp(dist.isend, m, self.dst, self.group) for m in msgs]
            works = dist.batch_isend_irecv(ops)
        self._pending.append((works, msgs, bufs))

    def finish(self):
        """Learner: (records, frames) gathered over ranks (dim 1) and slices (dim 0); other ranks: None."""
        pend, self._pending, self._sizes = self._pending, [], None
        if not pend:
            return None
        if not self._active():
            data = torch.cat([p[0][0] for p in pend], 0)
            frames = torch.cat([p[1] for p in pend], 0) if pend[0][1] is not None else None
            return data, frames
        learner = dist.get_rank(self.group) == self.dst
        staged = pend[0][0] is None
        if not staged:
            dev = pend[0][1][0].device
            cur = torch.cuda.current_stream(dev)
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record(cur)
            with torch.cuda.stream(self._side):
                for works, _, _ in pend:
                    for w in works:
                        w.wait()
            cur.wait_stream(self._side)
            t1.record(cur)
            self._exposed = (t0, t1)
            for _, msgs, bufs in pend:             # allocated / filled under the side stream, consumed on the current one
                for t in list(msgs) + ([b for bl in bufs for b in bl] if bufs else []):
                    t.record_stream(cur)
        if not learner:
            return None
        rows, frame_rows = [], []
        for _, _, bufs in pend:
            if bufs is None or bufs[0] is None:
                return None
            k = 0
            if self.compact:
                rows.append(torch.cat([unpack_records(n, w, self.o, self.A) for n, w in zip(bufs[0], bufs[1])], 1))
                k = 2
            else:
                rows.append(torch.cat(list(bufs[0]), 1))
                k = 1
            if len(bufs) > k:
                frame_rows.append(torch.cat(list(bufs[k]), 1))
        return torch.cat(rows, 0), (torch.cat(frame_rows, 0) if frame_rows else None)

    def abandon(self):
        """Drops a half-posted exchange (ChunkExchange.warm_up's fallback): the outstanding work handles are WAITED for first
        (clearing the list under live sends / receives left them writing into freed buffers), errors of the wait are
        swallowed -- the exchange has failed already -- and the per-chunk sizes are forgotten."""
        pend, self._pending, self._sizes = self._pending, [], None
        for works, _, _ in pend:                    # (staged / inactive entries carry no work handles)
            for w in (works if isinstance(works, list) else ()):
                if hasattr(w, "wait"):
                    try:
                        w.wait()
                    except Exception:               # noqa: BLE001
                        pass

    def exposed_gather_ms(self):
        """After a synchronisation: device time between the end of the search and the end of the exchange in the last finish()
        (what of the transfer the search did not hide); None for staged (gloo) exchanges."""
        ev = getattr(self, "_exposed", None)
        return None if ev is None else float(ev[0].elapsed_time(ev[1]))

    def __call__(self, slab):                                    # the plain-callable protocol: one slab, one slice, list of parts
        return gather_to_learner(slab, self.dst, self.group, self.total_envs)


class ChunkExchange:
    """A K-step block on N > 1 ranks the way bench.py runs it: played in slices, every finished slice handed to the overlapped
    exchange (TrajectoryGather) -- or, in mode "plain", played whole and sent with one synchronous grouped send / receive per
    message (gather_to_learner).  Both modes deliver the same tensors to the learner.

        play(n_steps, t0) -> chunks            plays rows [t0, t0 + n_steps) of every env group's chunk
        rows(chunks, name, t0, t1) -> tensor   the groups' `name` rows ("data" | "obs") side by side (dim 1), or None

    warm_up(n) is the first use of the exchange: an exception raised there (outside any timed region) switches to "plain" and
    plays again -- what `gather_mode` in bench.py's JSON line reports.  The switch is AGREED between the ranks (a
    rank-local decision left one rank in the synchronous plain gather while its peers were in the sliced one): after the first
    run every rank, failed or not, enters one MAX all-reduce of its failed flag, and all of them switch and replay when any
    did.  (A peer that is stuck INSIDE the sliced exchange because the failing rank never posted its side cannot reach that
    all-reduce; like an RCCL failure that hangs or aborts instead of raising, that ends at the process group's timeout as an
    error, and bench.py's launcher re-runs the ranks with --gather-mode plain.)"""

    def __init__(self, tg, play, rows, mode="overlapped", total_envs=None, log=None):
        assert mode in ("overlapped", "plain")
        self.tg, self.play, self.rows, self.total_envs, self.log = tg, play, rows, total_envs, log
        self.mode = {"kind": mode}

    def run(self, n):
        """n env steps + the exchange.  Returns (chunks, gathered): gathered = (records, frames | None) on the learner, None on
        the other ranks."""
        tg = self.tg
        if self.mode["kind"] == "plain":
            chunks = self.play(n, 0)
            parts = gather_to_learner(self.rows(chunks, "data", 0, n), tg.dst, tg.group, self.total_envs)
            obs = self.rows(chunks, "obs", 0, n)
            fparts = gather_to_learner(obs, tg.dst, tg.group, self.total_envs) if obs is not None else None
            if parts is None:
                return chunks, None
            return chunks, (torch.cat(list(parts), 1), torch.cat(list(fparts), 1) if fparts is not None else None)
        k = max(1, min(tg.slices, n))
        cuts = [n * i // k for i in range(k + 1)]
        chunks = None
        for i in range(k):
            chunks = self.play(cuts[i + 1] - cuts[i], cuts[i])
            tg.start(self.rows(chunks, "data", cuts[i], cuts[i + 1]), self.rows(chunks, "obs", cuts[i], cuts[i + 1]))
        return chunks, tg.finish()

    def _any_rank_failed(self, failed):
        """MAX over the ranks of `failed` (every rank calls this exactly once per warm-up in overlapped mode)."""
        tg = self.tg
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(tg.group) == 1:
            return bool(failed)
        dev = "cuda" if dist.get_backend(tg.group) == "nccl" else "cpu"
        flag = torch.tensor([1 if failed else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=tg.group)
        return bool(int(flag.item()))

    def warm_up(self, n):
        if self.mode["kind"] == "plain":
            return self.run(n)
        out, err = None, None
        try:
            out = self.run(n)
        except Exception as e:                      # noqa: BLE001  (whatever the first contact with the collective library raises)
            err = e
            self.tg.abandon()
        if not self._any_rank_failed(err is not None):
            return out
        why = f"{type(err).__name__}: {err}" if err is not None else "another rank's overlapped exchange failed"
        if self.log is not None:
            self.log(f"overlapped trajectory gather failed at warm-up ({why}); using the plain gather")
        self.mode.update(kind="plain", error=why)
        return self.run(n)


def broadcast_model(model, src=0, group=None, device=None, loopback=False):
    """Learner -> actors weight hand-off after a training step (replaces Ray re-pickling the whole model into every
    task, self_play.py:249-256): every parameter/buffer of the six head modules is broadcast from rank `src` in one
    flattened message per module (checkpoint 421: 115 KB in total), then the cached batched evaluators are dropped so
    the next search packs the new weights.  "nccl" (= RCCL) broadcasts from device memory; "gloo" from the host.
    `loopback`: a world of one rank still flattens, broadcasts (RCCL executes the collective with a single participant) and
    copies back -- the single-GPU test of this path."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not loopback):
        return model
    use_cuda = dist.get_backend(group) == "nccl"
    # (gloo = the functional runs without RCCL: the message goes through the host whatever `device` says)
    dev = torch.device((device if device is not None else "cuda") if use_cuda else "cpu")
    for name in ("representation", "prediction", "afterstate_prediction", "afterstate_dynamics", "dynamics", "encoder"):
        module = getattr(model, name + "_function")
        tensors, seen = [], set()
        for t in list(module.parameters()) + list(module.buffers()):
            if id(t) not in seen and t.is_floating_point():          # shared trunks appear once
                seen.add(id(t))
                tensors.append(t)
        if not tensors:
            continue
        flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in tensors]).to(dev)
        dist.broadcast(flat, src=src, group=group)
        off = 0
        with torch.no_grad():
            for t in tensors:
                n = t.numel()
                t.copy_(flat[off:off + n].reshape(t.shape).to(t.device, t.dtype))
                off += n
    model.refresh_heads()
    return model
