"""hipGraph capture of the multiscale forward.

One MS-HGNN forward is ~35 short kernel launches on 1+S streams; at B=512 the GPU work is a few
hundred microseconds, less than what the host needs to issue those launches one by one.  The
launchers of libgroupnet_hip.so never synchronise or allocate, so the whole forward (side-stream
fork/join included) is captured once into a hipGraph and replayed with one host call.

Noise inside a graph: the host draw of the reference (torch.rand on the CPU) cannot be captured, so
a graphed forward uses the device Philox stream.  Its position lives in a device counter that the
graph itself advances at the end of every replay — each replay draws fresh, reproducible noise.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import torch

from . import MS_HGNN_batch as _mods
from . import ops
from .multiscale import MultiScaleHGNN

Tensor = torch.Tensor


class GraphedMultiScale:
    """Static-shape, replayable forward of a ``MultiScaleHGNN`` block.

        g = GraphedMultiScale(block, B, N, seed=1234)
        feats, H = g(f)          # f is copied into the graph's input buffer; outputs are the
                                 # graph's static buffers (overwritten by the next call)
    """

    def __init__(self, block: MultiScaleHGNN, B: int, N: int, seed: int = 0, device: Optional[torch.device] = None,
                 warmup: int = 2, dtype: torch.dtype = torch.float32, affinity_tail: Optional[bool] = None):
        """``affinity_tail``: capture with `block.affinity_tail` set to this value (None: as the block has it) — True =
        the latency form (affinity + top-k in the first node stage's tail), False = the throughput form for graphs that
        are replayed side by side on several streams (see `MultiScaleHGNN.affinity_tail`)."""
        p = next(block.parameters())
        self.device = device or p.device
        if self.device.type != "cuda":
            raise ValueError("GraphedMultiScale needs the block on a GPU")
        self.block = block
        self.B, self.N = B, N
        self.seed = int(seed)
        self.f_in = torch.zeros((B, N, block.h_dim), dtype=dtype, device=self.device)    # bf16: the twins (config 4)
        self.counter = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.draws_per_step = sum(b * e * k for (b, e, k) in block.noise_shapes(B, N)) * block.interaction.nmp_layers
        self.graph = torch.cuda.CUDAGraph()
        prev = (_mods._NoiseState.mode, _mods._NoiseState.seed, _mods._NoiseState.offset, _mods._NoiseState.counter)
        prev_tail = block.affinity_tail
        if affinity_tail is not None:
            block.affinity_tail = bool(affinity_tail)
        try:
            with torch.no_grad(), torch.cuda.device(self.device):
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):      # warm-up off the default stream: packs weights, sizes the pool
                    for _ in range(max(1, warmup)):
                        self._step()
                torch.cuda.current_stream(self.device).wait_stream(side)
                torch.cuda.synchronize(self.device)
                # every replay first advances the counter by one step's worth of draws: start one step back
                self.counter.fill_(-self.draws_per_step)
                with torch.cuda.graph(self.graph):
                    self.out, self.H = self._step()
        finally:
            block.affinity_tail = prev_tail
            _mods.set_noise_mode(prev[0], prev[1], prev[2], prev[3])

    def _step(self) -> Tuple[Tensor, Optional[Tensor]]:
        # offset 0 + device counter: the position is entirely on the device; the first launch of the
        # forward moves the counter past the previous replay's draws
        _mods.set_noise_mode("device", seed=self.seed, offset=0, counter=self.counter)
        return self.block(self.f_in, advance=(self.counter, self.draws_per_step))

    def __call__(self, f: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
        if f is not None:
            self.f_in.copy_(f, non_blocking=True)
        self.graph.replay()
        return self.out, self.H


class GraphedPastEncoder:
    """Static-shape, replayable inference forward of a `PastEncoder` drop-in (embedding front-end, affinity,
    incidences, all modules, concat — `model/GroupNet_nba.py:266-315`) as one hipGraph.

        g = GraphedPastEncoder(enc.eval(), B, N, seed=7)
        feats, new_H = g(inputs)       # inputs (B*N, T, in_dim) copied into the static buffer; outputs static
    """

    def __init__(self, enc, B: int, N: int, seed: int = 0, warmup: int = 2):
        p = next(enc.parameters())
        self.device = p.device
        if self.device.type != "cuda":
            raise ValueError("GraphedPastEncoder needs the encoder on a GPU")
        if enc.training:
            raise ValueError("GraphedPastEncoder captures the inference path: call enc.eval() first")
        self.enc, self.B, self.N, self.seed = enc, B, N, int(seed)
        self.x_in = torch.zeros((B * N, enc._length, enc.input_fc.in_features), dtype=torch.float32, device=self.device)
        self.counter = torch.zeros(1, dtype=torch.int64, device=self.device)
        E = [1 if m.scale == N else N for m in (getattr(enc, n) for n in enc._hyper_names)]
        self.draws_per_step = B * N * N * enc.interaction.edge_types + sum(B * e * 10 for e in E)
        self.graph = torch.cuda.CUDAGraph()
        prev = (_mods._NoiseState.mode, _mods._NoiseState.seed, _mods._NoiseState.offset, _mods._NoiseState.counter)
        try:
            with torch.no_grad(), torch.cuda.device(self.device):
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):
                    for _ in range(max(1, warmup)):
                        self._step()
                torch.cuda.current_stream(self.device).wait_stream(side)
                torch.cuda.synchronize(self.device)
                self.counter.fill_(-self.draws_per_step)
                with torch.cuda.graph(self.graph):
                    self.out, self.H = self._step()
        finally:
            enc.__dict__.pop("_advance", None)
            _mods.set_noise_mode(prev[0], prev[1], prev[2], prev[3])

    def _step(self):
        _mods.set_noise_mode("device", seed=self.seed, offset=0, counter=self.counter)
        self.enc._advance = (self.counter, self.draws_per_step)
        return self.enc(self.x_in, self.B, self.N)

    def __call__(self, inputs: Optional[Tensor] = None):
        if inputs is not None:
            self.x_in.copy_(inputs, non_blocking=True)
        self.graph.replay()
        return self.out, self.H


class GraphedTrainStep:
    """One training step of a ``MultiScaleHGNN`` block — forward, loss, backward and the optimizer update —
    captured in ONE hipGraph (`train_hyper_nba.py:107-118` is this loop).

    Eagerly a step is ~150 launches whose host side (autograd bookkeeping, descriptor tables, allocations)
    takes several times longer than the GPU work; replayed from a graph the step runs at GPU speed.

        step = GraphedTrainStep(block, opt, loss_fn, B, N, target_shapes=[(B, N, 2)], seed=7)
        loss = step(f, target)        # copies into the graph's static inputs, replays, returns the loss tensor

    ``loss_fn(final_feature, new_H, *targets) -> scalar``.  Noise is the device Philox stream, advanced inside
    the graph so every replay draws fresh Gumbel noise.  Everything that depends on parameter values (packed
    weight images) is rebuilt inside the graph; after a replay the eager caches are dropped, so eager calls
    between replays see the updated parameters.  Optimizers must be capturable (SGD is; Adam with
    ``capturable=True``)."""

    def __init__(self, block: MultiScaleHGNN, optimizer: torch.optim.Optimizer, loss_fn: Callable, B: int, N: int,
                 target_shapes: Sequence[Tuple[int, ...]] = (), seed: int = 0, warmup: int = 3):
        p = next(block.parameters())
        self.device = p.device
        if self.device.type != "cuda":
            raise ValueError("GraphedTrainStep needs the block on a GPU")
        self.block, self.optimizer, self.loss_fn = block, optimizer, loss_fn
        self.seed = int(seed)
        self.f_in = torch.zeros((B, N, block.h_dim), dtype=torch.float32, device=self.device)
        self.targets = [torch.zeros(tuple(s), dtype=torch.float32, device=self.device) for s in target_shapes]
        self.counter = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.draws_per_step = sum(b * e * k for (b, e, k) in block.noise_shapes(B, N)) * block.interaction.nmp_layers
        self.graph = torch.cuda.CUDAGraph()
        self._repack: dict = {}
        prev = (_mods._NoiseState.mode, _mods._NoiseState.seed, _mods._NoiseState.offset, _mods._NoiseState.counter)
        try:
            with torch.cuda.device(self.device):
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):
                    for _ in range(max(1, warmup)):
                        self._step()
                torch.cuda.current_stream(self.device).wait_stream(side)
                torch.cuda.synchronize(self.device)
                self.counter.fill_(-self.draws_per_step)
                _mods.invalidate_weight_caches(block)       # every packing kernel must be part of the graph
                optimizer.zero_grad(set_to_none=True)       # gradients are allocated from the graph's pool
                with torch.cuda.graph(self.graph):
                    self.loss = self._step()
                _mods.invalidate_weight_caches(block)
        finally:
            _mods.set_noise_mode(prev[0], prev[1], prev[2], prev[3])

    def _step(self) -> Tensor:
        _mods.set_noise_mode("device", seed=self.seed, offset=0, counter=self.counter)
        self.optimizer.zero_grad(set_to_none=True)
        # every packed weight image of the step from TWO launches at its head (ops.repack_scope: the first warm-up step
        # records which plans / images a step touches)
        with ops.repack_scope(self._repack):
            with torch.enable_grad():
                out, H = self.block(self.f_in, advance=(self.counter, self.draws_per_step))
                loss = self.loss_fn(out, H, *self.targets)
            loss.backward()
        self.optimizer.step()
        return loss.detach()

    def __call__(self, f: Optional[Tensor] = None, *targets: Tensor) -> Tensor:
        if f is not None:
            self.f_in.copy_(f, non_blocking=True)
        for dst, src in zip(self.targets, targets):
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        _mods.invalidate_weight_caches(self.block)
        return self.loss
