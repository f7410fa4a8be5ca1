"""Batch-parallel sharding of the MS-HGNN block over the GPUs of one node.

Scenes are independent (batch dim leads every tensor, weights are shared, no BatchNorm —
SURVEY.md §8e), so rank r owns the contiguous scene range ``shard_range(B, r, R)``, runs the
block on it with replicated weights and no data-path collective, and ONE all-gather (RCCL over
xGMI through ``torch.distributed``; backend "nccl" is RCCL on ROCm) returns the concatenated
output embeddings — the computed columns ``(B, N, 64*(1+S))`` of the ``(B, N, 64*(2+S))`` feature tensor, whose first
64 columns are a copy of the input every rank already holds — to every rank.  The reference has no distributed code at
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


def default_shard_noise(block, B: int, N: int, start: int, stop: int, device) -> List:
    """The noise a rank owning scenes [start, stop) of a B-scene batch must use so that the sharded run equals
    the single-device run with default noise: its rows of the FULL-batch streams.  `block` provides
    ``noise_shapes(B, N)`` (one (B,E,K) per module) and ``interaction.nmp_layers``.
      host mode   — every rank draws the full-batch uniforms exactly as one device would (same global CPU
                    generator state on every rank, module-major: all rounds of the pairwise module, then scale by
                    scale) and keeps its rows;
      device mode — per module and round a `PhiloxNoise` positioned at the rank's first row inside the module's
                    span of the stream (spans laid out back to back in the same module-major order)."""
    from . import MS_HGNN_batch as M
    from .ops import PhiloxNoise
    shapes = block.noise_shapes(B, N)
    nmp = block.interaction.nmp_layers
    if M._NoiseState.mode == "host":
        full = [[torch.rand(shp).float() for _ in range(nmp)] for shp in shapes]
        return [[u[start:stop].contiguous().to(device, non_blocking=True) for u in per] for per in full]
    out, cur = [], M._NoiseState.offset
    for (b, e, k) in shapes:
        per = []
        for _ in range(nmp):
            per.append(PhiloxNoise(M._NoiseState.seed, cur + start * e * k, M._NoiseState.counter))
            cur += b * e * k
        out.append(per)
    M._NoiseState.offset = cur
    return out


def sharded_forward(block: Callable[..., Tuple[Tensor, Optional[Tensor]]], f_full: Tensor,
                    noise_full: Optional[Sequence] = None, group=None, gather_H: bool = False,
                    input_prefix: bool = True):
    """Run `block` on this rank's scenes of `f_full` (every rank holds the full input, as a
    data-parallel caller would after its own loader) and all-gather the features.

    `block(f_local, noise_u=...)` -> `(features (b, N, F), H (b, E, N) or None)`;
    ``groupnet_amd.multiscale.MultiScaleHGNN`` has this signature.
    ``noise_full`` = the full-batch uniforms (sliced here), or None: every rank then takes its rows of the
    full-batch default streams (`default_shard_noise`), so that the result equals a single-device run —
    never the same noise on different scene shards.
    ``input_prefix``: the block's features start with a copy of its input (`final = cat(f, inter, hyper...)`,
    model/GroupNet_nba.py:301-309).  Every rank already holds those columns for the whole batch (`f_full`), so only the
    COMPUTED columns — the output embeddings, 64*(1+S) of the 64*(2+S) — cross xGMI (5.8 MB instead of 7.2 MB
    per 512 scenes) and the result is assembled as cat(f_full, gathered).  False gathers the whole tensor.
    Returns `(features_full (B, N, F), H_full or local H)`.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    B = f_full.shape[0]
    s, e = shard_range(B, rank, world)
    f_local = f_full[s:e].contiguous()
    if noise_full is not None:
        noise_local = slice_noise(noise_full, s, e)
    elif hasattr(block, "noise_shapes"):
        noise_local = default_shard_noise(block, B, f_full.shape[1], s, e, f_full.device)
    else:
        noise_local = None
    feats, H = block(f_local, noise_u=noise_local)
    D = f_full.shape[-1]
    if input_prefix and feats.shape[-1] > D and feats.dtype == f_full.dtype:
        feats_full = torch.cat((f_full, all_gather_rows(feats[..., D:].contiguous(), B, group)), dim=-1)
    else:
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


class BucketedGather:
    """The one exchange of a batch-sharded forward — the all-gather of the output embeddings — bucketed.

    xGMI is point-to-point (7 links per GPU), so an all-gather is per-link bound and small messages are dominated
    by launch latency: `slots` consecutive steps (e.g. one per compute stream of a throughput caller) fill one
    BANK of a double-buffered staging area and are gathered by ONE `all_gather_into_tensor` call — `slots` x
    larger messages — issued on a side stream so that it overlaps the next bank's compute.

        bg = BucketedGather(slots=4, local_shape=(512, 11, 320), device=dev)
        for k in range(steps):
            out = run_step(k)                       # on whatever stream the caller likes
            bg.put(out)                             # copies into the bank; gathers when the bank is full
        bg.flush()                                  # a partial last bank is gathered too
        bg.wait()                                   # host-side join of the gather stream
        g = bg.gathered(bank)                       # (world, slots, *local_shape); the caller's current stream is made
        consume(g)                                  # to wait for that bank's gather, so it may be read right away
        bg.release(bank)                            # ... and the NEXT gather into `bank` waits for this consumer

    Bank reuse is ordered by events: a step that writes slot i of a bank first waits for the gather that last read
    that bank; `gathered(bank)` orders the reader's stream behind the gather that wrote `out[bank]`, and a reader
    that keeps the view beyond the next `slots` puts calls `release(bank)` when done so that the gather two banks
    later does not overwrite it under its hands.  Works on CPU tensors with the gloo backend too (no streams there: everything is synchronous),
    which is how the logic is tested without GPUs."""

    def __init__(self, slots: int, local_shape: Sequence[int], device, dtype=torch.float32, group=None):
        if slots < 1:
            raise ValueError("slots >= 1")
        self.slots, self.group = int(slots), group
        self.world = dist.get_world_size(group)
        self.local_shape = tuple(int(d) for d in local_shape)
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.banks = torch.empty((2, self.slots) + self.local_shape, dtype=dtype, device=self.device)
        self.out = torch.empty((2, self.world * self.slots) + self.local_shape, dtype=dtype, device=self.device)
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.bank_free = [None, None]        # event: the gather that last read this bank has finished
        self.consumed = [None, None]         # event: the last reader of out[bank] is done (release)
        self.ready = [None] * self.slots     # event: slot i of the current bank has been written
        self.k = 0                           # steps put so far
        self.count = [0, 0]                  # valid slots in the last gather of each bank
        self.gathers = 0

    def put(self, out: Tensor) -> int:
        """Stage one step's local output (on the caller's current stream); returns the bank it went to."""
        if tuple(out.shape) != self.local_shape:
            raise ValueError(f"expected {self.local_shape}, got {tuple(out.shape)}")
        i, bank = self.k % self.slots, (self.k // self.slots) % 2
        self.k += 1
        if self.cuda:
            cur = torch.cuda.current_stream(self.device)
            if self.bank_free[bank] is not None:
                cur.wait_event(self.bank_free[bank])
            if out.is_contiguous():
                self.banks[bank, i].copy_(out, non_blocking=True)
            else:
                from . import ops            # (a column block of a wider tensor: one pitched copy launch)
                ops.copy_cols(self.banks[bank, i], out)
            ev = torch.cuda.Event()
            ev.record(cur)
            self.ready[i] = ev
        else:
            self.banks[bank, i].copy_(out)
        if i == self.slots - 1:
            self._gather(bank, self.slots)
        return bank

    def _gather(self, bank: int, count: int) -> None:
        src = self.banks[bank].view((self.slots * self.local_shape[0],) + self.local_shape[1:])
        dst = self.out[bank].view((self.world * self.slots * self.local_shape[0],) + self.local_shape[1:])
        if self.cuda:
            with torch.cuda.stream(self.stream):
                for ev in self.ready[:count]:
                    if ev is not None:
                        self.stream.wait_event(ev)
                if self.consumed[bank] is not None:          # a reader of the previous gather into this bank
                    self.stream.wait_event(self.consumed[bank])
                    self.consumed[bank] = None
                dist.all_gather_into_tensor(dst, src, group=self.group)
                ev = torch.cuda.Event()
                ev.record(self.stream)
                self.bank_free[bank] = ev
        else:
            dist.all_gather_into_tensor(dst, src, group=self.group)
        self.count[bank] = count
        self.gathers += 1

    def flush(self) -> None:
        """Gather a partially filled bank (step count not a multiple of `slots`); the unfilled slots carry stale
        bytes — `count[bank]` says how many are valid.  Every rank must call it at the same step."""
        rem = self.k % self.slots
        if rem:
            self._gather((self.k // self.slots) % 2, rem)
            self.k += self.slots - rem          # the next put starts a fresh bank

    def wait(self) -> None:
        if self.cuda:
            self.stream.synchronize()

    def gathered(self, bank: int) -> Tensor:
        """(world, slots, *local_shape) view of the last gather of `bank`: [r, i] = rank r's step in slot i.  The
        caller's current stream waits for that gather (no host block)."""
        if self.cuda and self.bank_free[bank] is not None:
            torch.cuda.current_stream(self.device).wait_event(self.bank_free[bank])
        return self.out[bank].view((self.world, self.slots) + self.local_shape)

    def release(self, bank: int) -> None:
        """The caller's current stream is done reading `gathered(bank)`: the next gather into that bank waits for it."""
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.consumed[bank] = ev
