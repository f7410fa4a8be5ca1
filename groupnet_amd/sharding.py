"""Batch-parallel sharding of the MS-HGNN block over the GPUs of one node.

Scenes are independent (batch dim leads every tensor, weights are shared, no BatchNorm —
SURVEY.md §8e), so rank r owns the contiguous scene range ``shard_range(B, r, R)``, runs the
block on it with replicated weights and no data-path collective, and ONE all-gather (RCCL over
xGMI through ``torch.distributed``; backend "nccl" is RCCL on ROCm) returns the concatenated
output embeddings ``(B, N, 64*(2+S))`` to every rank.  The reference has no distributed code at
all; this is new design.

Noise under sharding: to stay identical to a single-device run, a rank must use the rows it owns
of the FULL-batch uniform stream.  In 'device' mode that is an offset into the Philox counter;
with host noise the caller slices the full-batch draw (``slice_noise``).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

Tensor = torch.Tensor


def shard_range(B: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of B scenes over `world` ranks; the first B % world ranks get one extra."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    base, extra = divmod(B, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def slice_noise(noise_full: Sequence, start: int, stop: int) -> List:
    """Rows [start, stop) of every full-batch uniform tensor (nested lists allowed)."""
    out = []
    for u in noise_full:
        if isinstance(u, torch.Tensor):
            out.append(u[start:stop].contiguous())
        else:
            out.append(slice_noise(u, start, stop))
    return out


def philox_offsets(shapes: Sequence[Tuple[int, int, int]], start: int, base_offset: int = 0) -> List[int]:
    """Element offset of this rank's first row inside each module's full-batch (B,E,K) stream,
    when the modules' streams are laid out back to back starting at `base_offset`."""
    offs, cur = [], base_offset
    for (B, E, K) in shapes:
        offs.append(cur + start * E * K)
        cur += B * E * K
    return offs


def all_gather_rows(local: Tensor, B: int, group=None) -> Tensor:
    """All-gather along dim 0 of per-rank row blocks whose sizes follow ``shard_range``.

    Equal shards use one ``all_gather_into_tensor`` (a single RCCL call on the full buffer);
    ragged shards fall back to ``all_gather`` on per-rank views of the preallocated output.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    out = torch.empty((B,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    s, e = shard_range(B, rank, world)
    if local.shape[0] != e - s:
        raise ValueError(f"rank {rank}: local block has {local.shape[0]} rows, expected {e - s}")
    local = local.contiguous()
    if B % world == 0:
        dist.all_gather_into_tensor(out, local, group=group)
    else:
        views = []
        for r in range(world):
            rs, re = shard_range(B, r, world)
            views.append(out[rs:re])
        if all(v.shape[0] == views[0].shape[0] for v in views):
            dist.all_gather(views, local, group=group)
        else:
            # ragged: pad every block to the largest, gather, then copy the live rows out
            mx = max(v.shape[0] for v in views)
            pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            pad[: local.shape[0]].copy_(local)
            bufs = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(bufs, pad, group=group)
            for v, b in zip(views, bufs):
                v.copy_(b[: v.shape[0]])
    return out


def sharded_forward(block: Callable[..., Tuple[Tensor, Optional[Tensor]]], f_full: Tensor,
                    noise_full: Optional[Sequence] = None, group=None, gather_H: bool = False):
    """Run `block` on this rank's scenes of `f_full` (every rank holds the full input, as a
    data-parallel caller would after its own loader) and all-gather the features.

    `block(f_local, noise_u=...)` -> `(features (b, N, F), H (b, E, N) or None)`;
    ``groupnet_amd.multiscale.MultiScaleHGNN`` has this signature.
    Returns `(features_full (B, N, F), H_full or local H)`.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    B = f_full.shape[0]
    s, e = shard_range(B, rank, world)
    f_local = f_full[s:e].contiguous()
    noise_local = None if noise_full is None else slice_noise(noise_full, s, e)
    feats, H = block(f_local, noise_u=noise_local)
    feats_full = all_gather_rows(feats, B, group)
    if gather_H and H is not None:
        H = all_gather_rows(H, B, group)
    return feats_full, H


def allreduce_gradients(params: Sequence[torch.nn.Parameter], group=None, average: bool = True,
                        bucket_bytes: int = 256 << 20) -> None:
    """Data-parallel training across the batch shards (SURVEY §8f rank 2 on more than one GPU): the one
    exchange a training step adds is the sum of the parameter gradients.  They are flattened into as few
    buckets as `bucket_bytes` allows (the whole MS-HGNN block is 4.3 MB: one bucket) and reduced with ONE
    all-reduce each — xGMI rings are per-link bound, so few large messages — then scattered back in place.
    A parameter without a gradient on this rank contributes zeros (every rank must issue identical collectives).
    With `average` the sum is divided by the world size (the mean-over-batch loss of equal shards)."""
    world = dist.get_world_size(group)
    params = [p for p in params if p.requires_grad]
    if world == 1 or not params:
        return
    buckets: List[List[torch.nn.Parameter]] = [[]]
    size = 0
    for p in params:
        nbytes = p.numel() * p.element_size()
        if buckets[-1] and size + nbytes > bucket_bytes:
            buckets.append([])
            size = 0
        buckets[-1].append(p)
        size += nbytes
    for bucket in buckets:
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat /= world
        off = 0
        for p in bucket:
            n = p.numel()
            g = flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
