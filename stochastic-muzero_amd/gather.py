"""Trajectory gather: the only inter-GPU exchange of the self-play path.

Games are independent, so ranks never exchange tree state; after a fixed-length chunk each rank owns one contiguous
[T][B][F] float64 slab and the learner rank (0) needs all of them for ReplayBuffer.save_game.  With torch.distributed's
"nccl" backend (= RCCL on ROCm) this is one grouped send/recv: every GPU has a direct xGMI link to rank 0, so the 7
receives proceed on 7 distinct links and no ring is involved.  "gloo" runs the same code on CPU tensors (tests).
"""
import torch
import torch.distributed as dist


def gather_to_learner(slab, dst=0, group=None):
    """Returns the list of every rank's slab on rank `dst` (rank order), None elsewhere."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [slab]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    slab = slab.contiguous()
    if dist.get_backend(group) == "gloo" and slab.is_cuda:     # functional runs without RCCL: stage through the host
        parts = gather_to_learner(slab.cpu(), dst, group)
        return None if parts is None else [p.to(slab.device) for p in parts]
    if rank == dst:
        parts = [slab if r == dst else torch.empty_like(slab) for r in range(world)]
        ops = [dist.P2POp(dist.irecv, parts[r], r, group) for r in range(world) if r != dst]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        return parts
    for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, slab, dst, group)]):
        w.wait()
    return None


def shard_range(total_envs, rank, world):
    """Contiguous env shard of a rank: env i keeps its seed and initial state whatever the world size."""
    per = (total_envs + world - 1) // world
    lo = min(total_envs, rank * per)
    return lo, min(total_envs, lo + per)


def broadcast_model(model, src=0, group=None, device=None):
    """Learner -> actors weight hand-off after a training step (replaces Ray re-pickling the whole model into every
    task, self_play.py:249-256): every parameter/buffer of the six head modules is broadcast from rank `src` in one
    flattened message per module (checkpoint 421: 115 KB in total), then the cached batched evaluators are dropped so
    the next search packs the new weights.  "nccl" (= RCCL) broadcasts from device memory; "gloo" from the host."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
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
