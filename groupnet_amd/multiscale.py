"""The multiscale block around the path: what ``PastEncoder.forward`` does between computing the
agent embedding and concatenating the per-scale features (model/GroupNet_nba.py:284-311).

    corr  = normalize(f) normalize(f)^T                         :284-286
    inter = MS_HGNN_oridinary(f)                                :290
    hyper_s, H_s = MS_HGNN_hyper_s(f, corr)  for each scale     :293-299
    final = cat(f, inter, hyper_1, ...)                         :301-309
    new_H = cat(H_1, H_2, ...) on dim 1                         :296,299

MI355X-first choices: affinity and the incidence of EVERY scale come from one fused launch
(corr never makes a trip through HBM at all); the 1+S modules are independent given (f, H_s), so
each stage of all of them is ONE grouped launch (the hyper modules have only B*N edge rows each
and would leave most of the 256 CUs idle if launched one behind the other).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import contextlib
import os
import torch
import torch.nn as nn

from . import ops
from .MS_HGNN_batch import MS_HGNN_hyper, MS_HGNN_oridinary, _needs_grad, _plist, run_message_passing

Tensor = torch.Tensor


def multiscale_autograd(pair: MS_HGNN_oridinary, hypers: Sequence[MS_HGNN_hyper], scales: Sequence[int], f: Tensor,
                        noise_u: Optional[Sequence] = None) -> Tuple[Tensor, Optional[Tensor]]:
    """cat(f, pairwise(f), hyper_s(f, corr)...) and cat(H_s) with autograd attached: incidences from one fused
    launch (constants of the backward), then ONE `MSHGNNFunction` node over all modules."""
    from .backward import MSHGNNFunction
    S = len(hypers)
    fd = f.detach().contiguous()
    if S and ops.fused_affinity_fits(fd.shape[1], fd.shape[2]):
        _, Hs, new_H = ops.affinity_topk(fd, list(scales), want_corr=False, want_H_cat=True)
    elif S:
        Hs = ops.topk_incidence(ops.affinity(fd), list(scales))
        new_H = torch.cat(Hs, dim=1)
    else:
        Hs, new_H = [], None
    mods = (pair, *hypers)
    if noise_u is None:
        # draw as the no-grad path (and the reference) does: module-major — every round of the pairwise module,
        # then scale by scale — so that a seeded training forward sees the noise of the seeded inference forward
        from .MS_HGNN_batch import _draw_uniform
        B, N = fd.shape[0], fd.shape[1]
        shapes = [(B, N * N, pair.edge_types)] + [(B, H.shape[1], m.edge_types) for H, m in zip(Hs, hypers)]
        noise_u = [[_draw_uniform(shp, fd.device) for _ in range(pair.nmp_layers)] for shp in shapes]
    nz = tuple(noise_u)
    if len(nz) != 1 + S:
        raise ValueError(f"noise_u: need {1 + S} entries (pairwise + one per scale)")
    params = [p for m in mods for p in _plist(m)]
    res = MSHGNNFunction.apply(mods, (None, *Hs), nz, *([f] * (1 + S)), *params)
    return torch.cat([f, *res[0::2]], dim=-1), new_H


_FORK_STREAMS = {}


def _fork_stream(device: torch.device) -> "torch.cuda.Stream":
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    st = _FORK_STREAMS.get(key)
    if st is None:
        st = _FORK_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


class MultiScaleHGNN(nn.Module):
    """One pairwise module + one hyper module per scale on the same (f, corr).

    ``forward(f)`` -> ``(final_feature (B, N, 64*(2+S)), new_H (B, sum_s E_s, N))``.
    Unlike ``PastEncoder`` (three scales at most, model/GroupNet_nba.py:218-248) any number of
    scales up to 8 is accepted (BASELINE configs 4 and 5 use four).
    """

    def __init__(self, hyper_scales: Sequence[int] = (2, 5, 11), h_dim: int = 64, nmp_layers: int = 1,
                 grouped: bool = True):
        super().__init__()
        if not 0 <= len(hyper_scales) <= 8:
            raise ValueError("0..8 scales")
        self.hyper_scales = [int(s) for s in hyper_scales]
        self.h_dim = h_dim
        # same constructor arguments as PastEncoder.__init__ (model/GroupNet_nba.py:209-248)
        self.interaction = MS_HGNN_oridinary(embedding_dim=16, h_dim=h_dim, mlp_dim=64, bottleneck_dim=h_dim,
                                             batch_norm=0, nmp_layers=nmp_layers)
        self.interaction_hyper = nn.ModuleList(
            MS_HGNN_hyper(embedding_dim=h_dim, h_dim=h_dim, mlp_dim=64, bottleneck_dim=h_dim, batch_norm=0,
                          nmp_layers=nmp_layers, scale=s) for s in self.hyper_scales)
        self.grouped = grouped
        # The fused affinity + top-k launch rides as the tail workgroups of the first node-stage launch (one launch and
        # one boundary fewer: -5.5 us of 116 on a dependent chain of forwards at B = 512, N = 11).  A caller that overlaps
        # independent forwards on several streams may prefer it as its own small launch, which fits beside other
        # streams' kernels (+2.6 % throughput on 4 streams in the same measurement): set False.  The same switch selects
        # whether the typed-aggregation launch applies the closing MLPs itself (4 launches instead of 5: same latency,
        # -2 % throughput side by side — `run_message_passing(fuse_closing=)`).
        self.affinity_tail = True

    @property
    def out_features(self) -> int:
        return self.h_dim * (2 + len(self.hyper_scales))

    def noise_shapes(self, B: int, N: int) -> List[Tuple[int, int, int]]:
        """(B,E,K) of the uniforms each module draws per message-passing round, pairwise first."""
        out = [(B, N * N, self.interaction.edge_types)]
        for s in self.hyper_scales:
            out.append((B, 1 if s == N else N, 10))
        return out

    def forward(self, f: Tensor, noise_u: Optional[Sequence] = None, advance=None
                ) -> Tuple[Tensor, Optional[Tensor]]:
        """``noise_u``: optional list with one entry per module (pairwise first), each a tensor or a
        list of ``nmp_layers`` tensors; default draws as the modules do (reference order).
        ``advance`` = (counter, n): add n to the device Philox counter at the START of this forward
        (used by the captured graph so that every replay draws fresh noise)."""
        ops._req(f, "f", (None, None, self.h_dim), ops._ACT_DTYPES)
        B, N, D = f.shape
        S = len(self.hyper_scales)
        nmp = self.interaction.nmp_layers
        if _needs_grad(self, f):
            if f.dtype == torch.bfloat16:
                # bf16 storage under autograd: the fp32 training path on the up-cast features, results back in bf16
                # (see MS_HGNN_batch._forward_autograd) — config 4 can train, with fp32 intermediates
                if advance:
                    ops.counter_add(advance[0], advance[1])
                out, new_H = multiscale_autograd(self.interaction, list(self.interaction_hyper), self.hyper_scales,
                                                 f.float(), noise_u)
                return out.to(torch.bfloat16), (None if new_H is None else new_H.to(torch.bfloat16))
            if f.dtype != torch.float32:
                raise NotImplementedError("activations must be fp32 or bf16")
            # training: ONE autograd node for the 1+S modules (grouped fused forward, grouped HIP backward);
            # the concat is an ordinary differentiable torch.cat
            if advance:
                ops.counter_add(advance[0], advance[1])
            return multiscale_autograd(self.interaction, list(self.interaction_hyper), self.hyper_scales, f, noise_u)
        if noise_u is None:
            # reference order: every draw of the pairwise module first, then scale by scale
            from .MS_HGNN_batch import _draw_uniform
            noise_u = [[_draw_uniform(shp, f.device) for _ in range(nmp)] for shp in self.noise_shapes(B, N)]
        elif len(noise_u) != 1 + S:
            raise ValueError(f"noise_u: need {1 + S} entries (pairwise + one per scale)")
        final = torch.empty((B, N, self.out_features), dtype=f.dtype, device=f.device)
        cols = [final[..., D * (1 + i):D * (2 + i)] for i in range(1 + S)]   # written in place by the last MLP
        join = None
        if S and ops.fused_affinity_fits(N, D):
            # one launch: affinity, incidence of every scale, f -> final[..., :D], cat(H_s), Philox bump.
            # The node stage of the first round needs only f: inside a graph capture the two launches are forked
            # (the graph then runs them side by side; eager launches stay on one stream)
            # — worth it only when the launches are long: at B*N = 5.6 k rows the extra graph edges cost more
            # (+9 us per replay) than the 7-us overlap saves; at 51 k rows the replay is 33 us shorter
            fork = self.grouped and f.is_cuda and B * N >= 32768 and torch.cuda.is_current_stream_capturing()
            main = torch.cuda.current_stream(f.device) if fork else None
            side = _fork_stream(f.device) if fork else None
            if fork:
                side.wait_stream(main)
            tail = None
            if self.grouped and not fork and self.affinity_tail:
                # the launch is DEFERRED: it rides as the tail workgroups of the first node-stage launch (which needs only
                # f), or is issued right before it when that launch cannot take it — one launch and one boundary fewer
                tail = ops.AffinityTail(f, self.hyper_scales, want_corr=False, f_out=final[..., :D], want_H_cat=True,
                                        counter=advance[0] if advance else None, counter_add=advance[1] if advance else 0)
                Hs, new_H = tail.Hs, tail.H_cat
            else:
                with (torch.cuda.stream(side) if fork else contextlib.nullcontext()):
                    _, Hs, new_H = ops.affinity_topk(f, self.hyper_scales, want_corr=False, f_out=final[..., :D],
                                                     want_H_cat=True, counter=advance[0] if advance else None,
                                                     counter_add=advance[1] if advance else 0)
            if fork:
                join = lambda: main.wait_stream(side)
        elif S:
            # large N (N*(N+68)*4 B > LDS tile): banded affinity and banded top-k launches, plain copies
            Hs = ops.topk_incidence(ops.affinity(f if f.dtype == torch.float32 else f.float()), self.hyper_scales)
            new_H = torch.cat(Hs, dim=1).to(f.dtype)
            final[..., :D].copy_(f)
            if advance:
                ops.counter_add(advance[0], advance[1])
        else:
            Hs, new_H = [], None
            final[..., :D].copy_(f)
            if advance:
                ops.counter_add(advance[0], advance[1])
        mods = [self.interaction, *self.interaction_hyper]
        if self.grouped:
            # every stage of the 1+S modules in ONE launch: launches always carry enough workgroups
            # to fill the chip, and nothing depends on how streams map to hardware queues
            # (latency form — `affinity_tail` — also folds the closing MLPs into the aggregation launch: 4 launches)
            run_message_passing(mods, [f] * (1 + S), [None, *Hs], list(noise_u), cols, join=join,
                                affinity=tail if (S and ops.fused_affinity_fits(N, D)) else None,
                                fuse_closing=self.affinity_tail or os.environ.get("GN_FUSE_CLOSING") == "2")
        else:
            for m, H, u, c in zip(mods, [None, *Hs], noise_u, cols):
                run_message_passing([m], [f], [H], [u], [c])
        return final, new_H
