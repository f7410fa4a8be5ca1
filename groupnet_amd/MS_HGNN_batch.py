"""Drop-in replacement of the reference's ``model/MS_HGNN_batch.py`` for MI355X.

    from groupnet_amd.MS_HGNN_batch import MS_HGNN_oridinary, MS_HGNN_hyper, MLP

keeps the constructor arguments, ``forward()`` signatures, return tuples and
``state_dict`` keys of the reference classes (model/MS_HGNN_batch.py:55-198,
201-229, 270-443), so ``model/GroupNet_nba.py:9,209-248,290-299`` works unchanged
and ``load_state_dict(strict=True)`` of a reference checkpoint succeeds.  The
modules are parameter containers plus orchestration: every tensor operation of the
forward is a hand-written gfx950 kernel of libgroupnet_hip.so (``groupnet_amd.ops``).
There is no CPU path and no torch-math fallback; CPU tensors raise ``ValueError``.

Differences from the reference that a caller can observe
  * tensors must be on the GPU (the reference fork only runs on the CPU);
  * under autograd the two MS_HGNN modules return outputs with a HIP backward attached
    (``groupnet_amd.backward``); their sub-modules (``MLP``,
    ``MLP_dict_softmax``, ``edge_aggregation``) called on their own are forward-only;
  * optional ``noise_u=`` lets the caller inject the uniforms of the Gumbel noise.
    By default they are drawn exactly as the reference does — ``torch.rand`` on the
    global CPU generator, one ``(B,E,K)`` draw per MLP_dict_softmax call
    (model/MS_HGNN_batch.py:45,454) — and uploaded; ``set_noise_mode('device', seed)``
    switches to the on-device Philox stream.
"""
from __future__ import annotations

import warnings
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple, Union

import os
import torch
import torch.nn as nn

from . import ops

Tensor = torch.Tensor
_HDIM_EXTEND = 64      # model/MS_HGNN_batch.py:72,292
_GUMBEL_TAU = 0.5      # model/MS_HGNN_batch.py:45
# Beyond these N the stand-alone gather / scatter kernels (LDS-tiled, high occupancy) beat a per-lane scan of H rows /
# a per-lane gather of N feature rows in the prologue of an MFMA kernel (latency-bound at 1-2 waves per SIMD; measured
# again in round 2 at N = 50, B = 1024, bf16: typed MLP 390 -> 439 us against a 38-us gather launch, closing MLP
# 54 -> 303 us against 71 + 56 us of scatter launches).
_FUSED_GATHER_MAX_N = int(os.environ.get("GN_FUSED_GATHER_MAX_N", "16"))     # eo = H @ ori inside the typed MLP kernel
_FUSED_SCATTER_MAX_N = int(os.environ.get("GN_FUSED_SCATTER_MAX_N", "16"))   # cat(H^T feat, ori)/N inside the closing MLP
_FUSE_POOL = os.environ.get("GN_FUSE_POOL", "1") != "0"   # node->edge pooling inside the edge kernel (inference)


# ---------------------------------------------------------------------------------------------
# noise source
# ---------------------------------------------------------------------------------------------
class _NoiseState:
    mode = "host"   # "host": torch.rand on the CPU generator (reference contract); "device": Philox
    seed = 0
    offset = 0      # running element offset of the device stream (host-side bookkeeping)
    counter = None  # optional 1-element int64 GPU tensor added to `offset` on the device (graph replays)


def set_noise_mode(mode: str, seed: Optional[int] = None, offset: int = 0, counter: Optional[Tensor] = None) -> None:
    """'host' (default, reference-identical stream) or 'device' (Philox4x32-10, no host work).
    ``counter``: a 1-element int64 GPU tensor holding the stream position on the DEVICE; a captured
    hipGraph that ends with ``ops.counter_add(counter, n)`` then draws fresh noise on every replay."""
    if mode not in ("host", "device"):
        raise ValueError("mode must be 'host' or 'device'")
    _NoiseState.mode = mode
    _NoiseState.counter = counter if mode == "device" else None
    if seed is not None:
        _NoiseState.seed = int(seed)
        _NoiseState.offset = int(offset)


def _draw_uniform(shape: Tuple[int, int, int], device: torch.device):
    """The uniforms of one MLP_dict_softmax call: a tensor drawn as the reference does (host mode), or
    a handle on the next shape-sized span of the device Philox stream, expanded inside the edge kernel."""
    if _NoiseState.mode == "host":
        return torch.rand(shape).float().to(device, non_blocking=True)
    u = ops.PhiloxNoise(_NoiseState.seed, _NoiseState.offset, _NoiseState.counter)
    _NoiseState.offset += shape[0] * shape[1] * shape[2]
    return u


def _noise_iter(noise_u: Union[None, Tensor, Sequence[Tensor]]) -> Optional[Iterator[Tensor]]:
    if noise_u is None:
        return None
    if isinstance(noise_u, (torch.Tensor, ops.PhiloxNoise)):
        return iter([noise_u])
    return iter(list(noise_u))


def _plist(m: nn.Module) -> tuple:
    """Parameters of `m` as a cached tuple (walking the module tree costs ~10 us per call and the engine asks
    hundreds of times per step); dropped by `invalidate_weight_caches`."""
    pl = m.__dict__.get("_gn_plist")
    if pl is None:
        pl = m.__dict__["_gn_plist"] = tuple(m.parameters())
    return pl


def _needs_grad(mod: nn.Module, *inputs: Optional[Tensor]) -> bool:
    """True when autograd is recording and the call involves anything that wants a gradient."""
    if not torch.is_grad_enabled():
        return False
    return any(t is not None and t.requires_grad for t in inputs) or any(p.requires_grad for p in _plist(mod))


_warned_grad = False


def _check_forward_only(*inputs: Tensor) -> None:
    global _warned_grad
    if not torch.is_grad_enabled():
        return
    if any(t is not None and t.requires_grad for t in inputs):
        raise RuntimeError("this sub-module is forward-only when called on its own: an input requires grad. "
                           "Differentiate through MS_HGNN_oridinary / MS_HGNN_hyper (groupnet_amd.backward) or call "
                           "it under torch.no_grad().")
    if not _warned_grad:
        warnings.warn("groupnet_amd sub-modules called on their own are forward-only; outputs carry no autograd graph.")
        _warned_grad = True


# ---------------------------------------------------------------------------------------------
# parameter containers (same names / shapes as the reference so state_dicts interchange)
# ---------------------------------------------------------------------------------------------
class MLP(nn.Module):
    """Stack of nn.Linear with an activation between layers (model/MS_HGNN_batch.py:201-229).

    Inside the MS-HGNN modules an MLP is only a parameter container — the fused HIP chains read
    its weights.  Called on its own (the reference's L3 model imports it for unrelated heads,
    model/GroupNet_nba.py:9,31-32) every Linear runs on the HIP GEMM (`groupnet_amd.linear.hip_linear`,
    differentiable); activations / dropout between the layers are elementwise torch ops.  GPU tensors only,
    like everything else in this package.
    """

    def __init__(self, input_dim, output_dim, hidden_size=(1024, 512), activation='relu', discrim=False,
                 dropout=-1):
        super().__init__()
        widths = [input_dim, *hidden_size, output_dim]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(widths[:-1], widths[1:]))
        if activation == 'relu':
            self.activation = nn.ReLU()
        elif activation == 'sigmoid':
            self.activation = nn.Sigmoid()
        self.sigmoid = nn.Sigmoid() if discrim else None
        self.dropout = dropout

    def forward(self, x):
        from .linear import hip_linear
        ops._req(x, "x")
        last = len(self.layers) - 1
        for i, layer in enumerate(self.layers):
            x = hip_linear(x, layer)
            if i != last:
                x = self.activation(x)
                if self.dropout != -1:
                    p = min(0.1, self.dropout / 3) if i == 1 else self.dropout
                    x = nn.functional.dropout(x, p=p, training=True)
            elif self.sigmoid is not None:
                x = self.sigmoid(x)
        return x


def _two_layer(mlp: MLP) -> Tuple[nn.Linear, nn.Linear]:
    if len(mlp.layers) != 2:
        raise NotImplementedError("the HIP chains are built for MLPs with one hidden layer")
    return mlp.layers[0], mlp.layers[1]


class MLP_dict_softmax(nn.Module):
    """Edge-type head (model/MS_HGNN_batch.py:31-53): (factor * distribution, distribution)."""

    def __init__(self, input_dim, output_dim, hidden_size=(1024, 512), activation='relu', discrim=False,
                 dropout=-1, edge_types=5):
        super().__init__()
        if input_dim != _HDIM_EXTEND or tuple(hidden_size) != (128,):
            raise NotImplementedError("HIP edge-MLP chain is specialised to 64 -> 128 -> {64, K, 1}")
        if not 1 <= edge_types <= 15:
            raise NotImplementedError("edge_types must be in 1..15")
        self.bottleneck_dim = edge_types
        self.MLP_distribution = MLP(input_dim=input_dim, output_dim=edge_types, hidden_size=hidden_size)
        self.MLP_factor = MLP(input_dim=input_dim, output_dim=1, hidden_size=hidden_size)
        self.init_MLP = MLP(input_dim=input_dim, output_dim=input_dim, hidden_size=hidden_size)
        self._pk: Optional[dict] = None
        self._pk_key = None
        self._plan = None

    def _packed(self) -> dict:
        """Weight stream of the edge-MLP kernel (layout: `ops.edge_stream`) and its biases, refreshed from the
        parameters by one `PackPlan` launch whenever they changed."""
        params = _plist(self)
        if self._plan is None or self._plan[0] != tuple(p.data_ptr() for p in params):
            K = self.bottleneck_dim
            i0, i1 = _two_layer(self.init_MLP)
            d0, d1 = _two_layer(self.MLP_distribution)
            f0, f1 = _two_layer(self.MLP_factor)
            plan, T = ops.PackPlan(params[0].device), ops.PackPlan.TILE
            w0 = plan.alloc(0)
            a0 = lambda o: plan.block(plan.alloc(2 * T), i0.weight, 2, r0=32 * o, rows=32)         # hidden tile o of layer 0
            b0 = lambda o: plan.block(plan.alloc(2 * T), (d0 if o < 4 else f0).weight, 2, r0=32 * (o % 4), rows=32)

            def sa(t):      # both output tiles of init_MLP layer 1 over hidden tile t
                off = plan.alloc(2 * T)
                for o in range(2):
                    plan.block(off + o * T, i1.weight, 1, r0=32 * o, c0=32 * t, rows=32, cols=32)

            def sb(t):      # (logits | factor) head over hidden tile t: d1 rows 0..K-1, f1 row K
                off = plan.alloc(T)
                if t < 4:
                    plan.block(off, d1.weight, 1, c0=32 * t, cols=32)
                else:
                    plan.block(off, f1.weight, 1, c0=32 * (t - 4), cols=32, place_r=K)
            # pair A: T0 T1 S0 T2 S1 T3 S2 S3; pair B: T0 T1 S0 T2 S1 ... T7 S6 S7; then the ring's 8-step run-out
            a0(0), a0(1), sa(0), a0(2), sa(1), a0(3), sa(2), sa(3)
            b0(0), b0(1)
            for t in range(8):
                sb(t)
                if t < 6:
                    b0(t + 2)
            plan.alloc(2 * T)
            w_len = plan.size - w0
            # the same four layers in the pipeline order of the bf16-core kernel (source of its image): 40 tiles
            wh = plan.alloc(40 * T)
            off = wh
            for kind, t in ops.pipeline_order(4):
                if kind == "A":
                    plan.block(off, i0.weight, 2, r0=32 * t, rows=32)
                else:
                    plan.block(off, i1.weight, 1, r0=0, c0=32 * t, rows=32, cols=32)
                    plan.block(off + T, i1.weight, 1, r0=32, c0=32 * t, rows=32, cols=32)
                off += 2 * T
            for kind, t in ops.pipeline_order(8):
                if kind == "A":
                    plan.block(off, (d0 if t < 4 else f0).weight, 2, r0=32 * (t % 4), rows=32)
                    off += 2 * T
                else:
                    if t < 4:
                        plan.block(off, d1.weight, 1, c0=32 * t, cols=32)
                    else:
                        plan.block(off, f1.weight, 1, c0=32 * (t - 4), cols=32, place_r=K)
                    off += T
            bo = plan.alloc(128 + 64 + 256 + 32)
            plan.vector(bo, i0.bias)
            plan.vector(bo + 128, i1.bias)
            plan.vector(bo + 192, d0.bias)
            plan.vector(bo + 320, f0.bias)
            plan.vector(bo + 448, d1.bias)
            plan.vector(bo + 448, f1.bias, place=K)
            plan.finish()
            self._plan = (plan.sources[:0] + tuple(p.data_ptr() for p in params), plan)
            xi = ops.XImages()
            xi.add("edge", plan.view(wh, 40 * T))
            self._pk = dict(W=plan.view(w0, w_len), bias=plan.view(bo, 480), xi=xi)
            self._pk_key = None
        key = _param_key(params)
        if key != self._pk_key or _volatile(params):
            self._plan[1].refresh()
            self._pk["xi"].bump()
            self._pk_key = key
        return self._pk

    def forward(self, x, noise_u: Optional[Tensor] = None):
        _check_forward_only(x)
        if x.dim() != 3:
            raise ValueError("MLP_dict_softmax expects (B, E, 64); the reference's softmax is only "
                             "well-defined for 3-D input (model/MS_HGNN_batch.py:517-520)")
        K = self.bottleneck_dim
        U = noise_u if noise_u is not None else _draw_uniform((x.shape[0], x.shape[1], K), x.device)
        return ops.edge_mlp_gumbel(x, U, self._packed(), K, _GUMBEL_TAU)


class edge_aggregation(nn.Module):
    """Typed edge -> node aggregation (model/MS_HGNN_batch.py:247-268): cat(H^T feat, ori)."""

    def __init__(self, input_dim, output_dim, hidden_size=(1024, 512), activation='relu', discrim=False,
                 dropout=-1, edge_types=5):
        super().__init__()
        if input_dim != 64:
            raise NotImplementedError("HIP aggregation kernels are specialised to 64-wide features")
        self.edge_types = edge_types
        self.dict_dim = input_dim
        self.agg_mlp = nn.ModuleList(MLP(input_dim=input_dim, output_dim=input_dim, hidden_size=(128,))
                                     for _ in range(edge_types))
        self.mlp = MLP(input_dim=input_dim, output_dim=input_dim, hidden_size=(128,))  # unused, kept for state_dict
        self._pk: Optional[dict] = None
        self._pk_key = None
        self._plan = None

    def _packed(self) -> dict:
        """Packed images of the K typed MLPs, refreshed by one `PackPlan` launch whenever they changed:
        W (both layers, type by type), b1 / b2, and for the pairwise form layer 1 of all types as one
        (K*128 x 64) matrix applied per node (half the bias rides with each of the two nodes of a pair) and
        layer 2 re-ordered hidden-tile-major."""
        params = _plist(self.agg_mlp)
        if self._plan is None or self._plan[0] != tuple(p.data_ptr() for p in params):
            K = self.edge_types
            l0 = [m.layers[0] for m in self.agg_mlp]
            l1 = [m.layers[1] for m in self.agg_mlp]
            plan, T = ops.PackPlan(params[0].device), ops.PackPlan.TILE
            w0 = plan.alloc(0)
            for a, b in zip(l0, l1):
                plan.matrix(a.weight)
                plan.matrix(b.weight)
            w_len = plan.size - w0
            b1o, b2o, bho = plan.alloc(K * 128), plan.alloc(K * 64), plan.alloc(K * 128)
            w1c, w2t, w12 = plan.alloc(K * 8 * T), plan.alloc(K * 8 * T), plan.alloc(K * 16 * T)
            for k in range(K):
                off = w12 + k * 16 * T  # both layers in the pipeline order of the bf16-core kernel (its image's source)
                for kind, o in ops.pipeline_order(4):
                    if kind == "A":
                        plan.block(off, l0[k].weight, 2, r0=32 * o, rows=32)
                    else:
                        plan.block(off, l1[k].weight, 1, r0=0, c0=32 * o, rows=32, cols=32)
                        plan.block(off + T, l1[k].weight, 1, r0=32, c0=32 * o, rows=32, cols=32)
                    off += 2 * T
                plan.vector(b1o + 128 * k, l0[k].bias)
                plan.vector(b2o + 64 * k, l1[k].bias)
                plan.vector(bho + 128 * k, l0[k].bias, scale=0.5)
                plan.block(w1c + k * 8 * T, l0[k].weight, 2)
                for t in range(4):
                    for o in range(2):
                        plan.block(w2t + (k * 8 + t * 2 + o) * T, l1[k].weight, 1, r0=32 * o, c0=32 * t, rows=32, cols=32)
            plan.finish()
            self._plan = (tuple(p.data_ptr() for p in params), plan)
            self._pk = dict(W=plan.view(w0, w_len), b1=plan.view(b1o, K * 128).view(K, 128),
                            b2=plan.view(b2o, K * 64).view(K, 64), W1cat=plan.view(w1c, K * 8 * T),
                            b1half=plan.view(bho, K * 128), W2t=plan.view(w2t, K * 8 * T))
            xi = ops.XImages()
            xi.add("W2t", self._pk["W2t"])                   # layer 2 per hidden tile (pair form)
            xi.add("W12", plan.view(w12, K * 16 * T))        # both layers, hidden-tile-major (two-layer form)
            xi.add("W1cat", self._pk["W1cat"])               # layer 1 of all types per node (node stage)
            self._pk["xi"] = xi
            self._pk_key = None
        key = _param_key(params)
        if key != self._pk_key or _volatile(params):
            self._plan[1].refresh()
            self._pk["xi"].bump()
            self._pk_key = key
        return self._pk

    def forward(self, edge_distribution, H, ori):
        """Returns cat(H^T feat, ori) WITHOUT the division by N (that is edge2node's,
        model/MS_HGNN_batch.py:120,355).  ``H=None`` selects the implicit pairwise graph."""
        _check_forward_only(edge_distribution, ori)
        return self._aggregate(edge_distribution, H, ori, divisor=1.0)

    def _aggregate(self, edge_feat, H, ori, divisor=None):
        """gather -> typed MLP -> scatter; divisor=None applies the / N of edge2node."""
        eo = ops.agg_gather(ori, H)
        feat = ops.agg_mlp(eo, edge_feat, self._packed(), self.edge_types)
        return ops.agg_scatter(feat, H, ori, divisor)


def _param_key(params: Iterable[nn.Parameter]):
    """Cheap fingerprint of a parameter set: storage address + in-place version counter."""
    return tuple((p.data_ptr(), p._version) for p in params)


def _volatile(params: Iterable[nn.Parameter]) -> bool:
    """Whether the packed-image cache must not be trusted for this call.  The fingerprint above misses writes
    made through ``p.data`` / raw pointers (they do not bump ``_version``) — the idiom of hand-written
    optimizers, EMA updates, weight clamping and the reference's own ``m.bias.data.fill_`` re-initialisation.
    While autograd is recording for these parameters (a training step) every call therefore re-runs the one
    refresh launch of its pack plan; only inference (no-grad / frozen parameters) trusts the cache, and code
    that rewrites weights there behind autograd's back calls `invalidate_weight_caches`."""
    return _TRAINING_CALL[0] or (torch.is_grad_enabled() and any(p.requires_grad for p in params))


_TRAINING_CALL = [False]


class training_call:
    """Context of a forward that belongs to a training step (`backward.MSHGNNFunction.forward` runs under
    no_grad, so `_volatile` cannot see it from the grad mode)."""

    def __enter__(self):
        self.prev, _TRAINING_CALL[0] = _TRAINING_CALL[0], True

    def __exit__(self, *exc):
        _TRAINING_CALL[0] = self.prev
        return False


def invalidate_weight_caches(module: nn.Module) -> None:
    """Mark every packed / concatenated weight image below `module` stale.  They are keyed on the parameters'
    (address, in-place version); anything that rewrites parameters behind autograd's back — a replayed
    hipGraph containing the optimizer step — must call this before the next eager use.  (The pack plans
    themselves — arenas and segment tables — stay; the next use re-runs their one refresh launch.)"""
    for m in module.modules():
        d = m.__dict__
        d.pop("_gn_plist", None)
        if "_pk_key" in d:
            m._pk_key = None
        for name in ("_pk_n2e", "_pk_mlp", "_bwd_cat"):
            for hit in d.get(name, {}).values():
                hit[3] = None
        if "_affine" in d:
            m._affine = None


class _MessagePassing(nn.Module):
    """What both reference modules share: one or more node->edge->node rounds
    (model/MS_HGNN_batch.py:174-195, 425-441)."""

    edge_types: int

    def _build(self, h_dim: int, bottleneck_dim: int, nmp_layers: int) -> None:
        if h_dim != 64:
            raise NotImplementedError("the gfx950 kernels are specialised to h_dim == 64 "
                                      "(every caller in the reference uses 64, model/GroupNet_nba.py:209-248)")
        if nmp_layers < 1:
            raise ValueError("nmp_layers must be >= 1")
        K = self.edge_types
        self.nmp_mlp_start = MLP_dict_softmax(input_dim=_HDIM_EXTEND, output_dim=h_dim, hidden_size=(128,),
                                              edge_types=K)
        rounds = []
        for _ in range(nmp_layers - 1):
            rounds.append(MLP(input_dim=h_dim * 2, output_dim=h_dim, hidden_size=(128,)))
            rounds.append(MLP_dict_softmax(input_dim=_HDIM_EXTEND, output_dim=h_dim, hidden_size=(128,),
                                           edge_types=K))
        self.nmp_mlps = nn.ModuleList(rounds)
        self.nmp_mlp_end = MLP(input_dim=h_dim * 2, output_dim=bottleneck_dim, hidden_size=(128,))
        self.attention_mlp = nn.ModuleList(MLP(input_dim=_HDIM_EXTEND * 2, output_dim=1, hidden_size=(32,))
                                           for _ in range(nmp_layers))
        self.node2edge_start_mlp = nn.ModuleList(MLP(input_dim=h_dim, output_dim=_HDIM_EXTEND, hidden_size=(256,))
                                                 for _ in range(nmp_layers))
        self.edge_aggregation_list = nn.ModuleList(
            edge_aggregation(input_dim=h_dim, output_dim=bottleneck_dim, hidden_size=(128,), edge_types=K)
            for _ in range(nmp_layers))
        self._pk_n2e: Dict[int, list] = {}      # idx -> [param addresses, packed dict, PackPlan, version key]
        self._pk_mlp: Dict[int, list] = {}

    # -- packed weights ------------------------------------------------------------------------
    def _packed_n2e(self, idx: int) -> dict:
        start, att = self.node2edge_start_mlp[idx], self.attention_mlp[idx]
        params = _plist(start) + _plist(att)
        ptrs = tuple(p.data_ptr() for p in params)
        hit = self._pk_n2e.get(idx)
        if hit is None or hit[0] != ptrs:
            s0, s1 = _two_layer(start)
            a0, a1 = _two_layer(att)
            D = _HDIM_EXTEND
            plan = ops.PackPlan(params[0].device)
            w0 = plan.matrix(s0.weight)
            plan.matrix(s1.weight)
            # attention layer 0 acts on cat(x'_n, e0_e): split it into the node half (with the bias) and the
            # edge half, which by linearity is applied to x' before the H-pooling: Wpq = [W[:, :D]; W[:, D:]]
            wpq = plan.alloc(4 * plan.TILE)
            plan.block(wpq, a0.weight, 2, c0=0, cols=D)
            plan.block(wpq, a0.weight, 2, c0=D, cols=D, place_r=32)
            w_len = plan.size - w0
            # the same chain in the pipeline order of the bf16-core kernel (source of its image): A_t = [W0(t,in0),
            # W0(t,in1)], B_t = [W1(0,t), W1(1,t)], then Wpq as above: 36 tiles = 72 sub-steps
            T = plan.TILE
            wc = plan.alloc(36 * T)
            off = wc
            for kind, t in ops.pipeline_order(8):
                if kind == "A":
                    plan.block(off, s0.weight, 2, r0=32 * t, rows=32)
                else:
                    plan.block(off, s1.weight, 1, r0=0, c0=32 * t, rows=32, cols=32)
                    plan.block(off + T, s1.weight, 1, r0=32, c0=32 * t, rows=32, cols=32)
                off += 2 * T
            plan.block(wc + 32 * T, a0.weight, 2, c0=0, cols=D)
            plan.block(wc + 32 * T, a0.weight, 2, c0=D, cols=D, place_r=32)
            bo = plan.alloc(256 + 64 + 64)
            plan.vector(bo, s0.bias)
            plan.vector(bo + 256, s1.bias)
            plan.vector(bo + 320, a0.bias)
            plan.finish()
            xi = ops.XImages()
            xi.add("chain", plan.view(wc, 36 * T))
            pk = dict(W=plan.view(w0, w_len), bias=plan.view(bo, 384), xi=xi,
                      w2=a1.weight.detach()[0], b2=a1.bias.detach())    # views of the parameters: no host sync
            hit = self._pk_n2e[idx] = [ptrs, pk, plan, None]
        key = _param_key(params)
        if key != hit[3] or _volatile(params):
            hit[2].refresh()
            hit[1]["xi"].bump()
            hit[3] = key
        return hit[1]

    def _packed_mlp2(self, mlp: MLP) -> dict:
        params = _plist(mlp)
        ptrs = tuple(p.data_ptr() for p in params)
        hit = self._pk_mlp.get(id(mlp))
        if hit is None or hit[0] != ptrs:
            l0, l1 = _two_layer(mlp)
            plan = ops.PackPlan(params[0].device)
            w0 = plan.matrix(l0.weight)
            plan.matrix(l1.weight)
            w_len = plan.size - w0
            pad = lambda n: (n + 31) // 32 * 32
            din, dh, dout = l0.in_features, l0.out_features, l1.out_features
            xi = ops.XImages()
            wh = n_t = 0
            if dout <= 64 and din % 32 == 0 and dh % 32 == 0:
                # pipeline order of the bf16-core kernel (source of its image): A_t = W0(t, in *), B_t = W1(*, t)
                T, IT, HT, OT = plan.TILE, din // 32, dh // 32, (dout + 31) // 32
                n_t = HT * (IT + OT)
                wh = plan.alloc(n_t * T)
                off = wh
                for kind, t in ops.pipeline_order(HT):
                    if kind == "A":
                        plan.block(off, l0.weight, IT, r0=32 * t, rows=32)
                        off += IT * T
                    else:
                        for o in range(OT):
                            plan.block(off + o * T, l1.weight, 1, r0=32 * o, c0=32 * t, rows=min(32, dout - 32 * o),
                                       cols=32)
                        off += OT * T
            bo = plan.alloc(pad(dh) + pad(dout))
            plan.vector(bo, l0.bias)
            plan.vector(bo + pad(dh), l1.bias)
            plan.finish()
            if n_t:
                xi.add("mlp2", plan.view(wh, n_t * plan.TILE))
            pk = dict(W=plan.view(w0, w_len), bias=plan.view(bo, pad(dh) + pad(dout)), xi=xi,
                      din=din, dh=dh, dout=dout)
            hit = self._pk_mlp[id(mlp)] = [ptrs, pk, plan, None]
        key = _param_key(params)
        if key != hit[3] or _volatile(params):
            hit[2].refresh()
            hit[1]["xi"].bump()
            hit[3] = key
        return hit[1]

    # -- stages (single-module faces of the grouped engine below) -----------------------------------
    def _node2edge(self, x: Tensor, H: Optional[Tensor], idx: int) -> Tensor:
        pk = self._packed_n2e(idx)
        xp, pq = ops.node_mlp(x, pk)
        return ops.node2edge(xp, pq, H, pk["w2"], pk["b2"])

    def _edge2node(self, edge_feat: Tensor, ori: Tensor, H: Optional[Tensor], idx: int) -> Tensor:
        return self.edge_aggregation_list[idx]._aggregate(edge_feat, H, ori)

    def _run(self, h: Tensor, H: Optional[Tensor], E: int, noise_u, out: Optional[Tensor] = None
             ) -> Tuple[Tensor, Tensor]:
        return run_message_passing([self], [h], [H], [noise_u], [out])[0]

    def _forward_autograd(self, h: Tensor, H: Optional[Tensor], noise_u, out: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
        """Training path: fused forward + HIP backward through torch.autograd."""
        from .backward import MSHGNNFunction
        if out is not None:
            raise ValueError("out= is an inference-time extra; under autograd the module returns a new tensor")
        if h.dtype == torch.bfloat16:
            # bf16 activations under autograd (config 4's storage type in training): the twins' kernels are
            # forward-only, so the training step runs the fp32 training path on the up-cast inputs — forward values
            # then carry fp32 intermediates (within the twins' own tolerance of them), gradients are those of the fp32
            # function at the bf16-rounded inputs, and outputs / the input gradient come back in bf16 (the casts are
            # ordinary differentiable torch ops).
            nf, fac = MSHGNNFunction.apply((self,), (None if H is None else H.float(),), (noise_u,), h.float(),
                                           *_plist(self))
            return nf.to(torch.bfloat16), fac.to(torch.bfloat16)
        if h.dtype != torch.float32:
            raise NotImplementedError("activations must be fp32 or bf16")
        return MSHGNNFunction.apply((self,), (H,), (noise_u,), h, *_plist(self))


class _Closed(list):
    """Outputs of closing MLPs that the aggregation launch applied itself."""


def _pair_form() -> bool:
    """Pairwise module, fp32 entry points: first layer of the typed aggregation MLP per NODE in the node stage + pair
    form (True), or both layers per unordered pair inside the aggregation kernel (False: no `A` tensor)."""
    return os.environ.get("GN_PAIR_FORM", "1") != "0"


def run_message_passing(mods: Sequence["_MessagePassing"], hs: Sequence[Tensor], Hs: Sequence[Optional[Tensor]],
                        noises: Sequence, outs: Sequence[Optional[Tensor]], traces=None, join=None, affinity=None,
                        fuse_closing: bool = False) -> List[Tuple[Tensor, Tensor]]:
    """The message-passing rounds of SEVERAL modules over the same scenes, stage by stage, each stage
    ONE grouped launch (model/MS_HGNN_batch.py:174-195 and :425-441 for every module at once).

    mods[i] runs on hs[i] (B,N,64) with incidence Hs[i] (None = the implicit pairwise graph);
    noises[i] is None (draw), a tensor / PhiloxNoise, or a list of nmp_layers of them; outs[i]
    optionally receives node_feat.  Returns [(node_feat, factors)] per module.  All modules must
    share nmp_layers and bottleneck_dim (they do in every caller of the reference).
    `traces` (training): one `backward.ModuleTrace` per module, which receives the node features entering
    every round and the dist every round sampled — all the backward needs besides the inputs.
    `join`: called once after the first node stage has been launched and before anything reads Hs (a caller that
    builds the incidences on a forked stream joins it here).
    `affinity`: an `ops.AffinityTail` — the deferred affinity + top-k launch that produces Hs; it rides in the first
    node-stage launch (its tail workgroups), or is issued beside it when that launch cannot take it.
    `fuse_closing`: let the typed-aggregation launch apply the closing MLP of every stage itself where its launch shape
    allows (`ops.closing_fusable`): one launch fewer per stage, bit-identical rows.  At B = 512, N = 11 the chain costs
    inside that launch what the closing launch cost on its own (single-stream forward 0.104 -> 0.103 ms, 4-stream
    throughput -2 %): the block asks for it in its latency form only."""
    n = len(mods)
    if not (n == len(hs) == len(Hs) == len(noises) == len(outs)) or n == 0:
        raise ValueError("run_message_passing: one h, H, noise and out per module")
    nmp = mods[0].nmp_layers
    if any(m.nmp_layers != nmp or m.bottleneck_dim != mods[0].bottleneck_dim for m in mods):
        raise ValueError("grouped modules must share nmp_layers and bottleneck_dim")
    B, N = hs[0].shape[0], hs[0].shape[1]
    Es = [N * N if H is None else H.shape[1] for H in Hs]   # ordered edges: the shape of noise and factors
    # The pairwise graph is symmetric: edges (i,j) and (j,i) pool the same feature, meet the same typed
    # MLP output and are summed into the same two nodes.  Its per-edge MLPs therefore run once per
    # unordered pair (N(N+1)/2 rows instead of N*N); only the Gumbel softmax runs per ordered edge.
    syms = [H is None for H in Hs]
    given = [_noise_iter(u) for u in noises]

    def next_u(i: int):
        if given[i] is not None:
            try:
                return next(given[i])
            except StopIteration:
                raise ValueError(f"noise_u: {nmp} uniform tensors of shape ({B},{Es[i]},{mods[i].edge_types}) needed")
        return _draw_uniform((B, Es[i], mods[i].edge_types), hs[i].device)

    twin = hs[0].dtype == torch.bfloat16
    if twin and traces is not None:
        raise NotImplementedError("the bf16 twins are forward-only")
    pair_A: List[Optional[Tensor]] = [None] * n     # per-node first layer of the typed MLP (pairwise groups)

    def node2edge(xs: Sequence[Tensor], idx: int) -> List[Tensor]:
        pks = [m._packed_n2e(idx) for m in mods]
        keep = [] if traces is not None else None
        # the node rows entering this round also feed the typed aggregation MLP that closes it: for the pairwise
        # graph its first layer is linear in the two nodes (eo = ori_i + ori_j) and runs once per NODE, in this
        # same launch (fp32 path; the bf16 twin runs both layers per pair on the matrix cores instead)
        specs = [((m.edge_aggregation_list[idx]._packed(), m.edge_aggregation_list[idx].edge_types)
                  if (sy and not twin and _pair_form()) else None) for m, sy in zip(mods, syms)]
        xpq, As = ops.node_stage_grouped([(x, pk) for x, pk in zip(xs, pks)], keep, specs,
                                         affinity if idx == 0 else None)
        pair_A[:] = As
        if join is not None and idx == 0:
            join()       # the incidences were built on a forked stream beside the node stage (graph capture)
        # Inference: the pooled edge rows feed only the edge MLP, so its kernel forms them itself (ops.PoolSpec) and
        # `edges` never exists in HBM — always for the pairwise graph, for hyper modules up to ops.POOL_MAX_N nodes;
        # larger hyper modules (and training, whose backward reads `edges`) keep the node2edge launch.
        fuse = [_FUSE_POOL and traces is None and (twin or ops.BF16X6) and (sy or N <= ops.POOL_MAX_N) for sy in syms]
        edges: List = [None] * n
        rest = [i for i in range(n) if not fuse[i]]
        if rest:
            for i, e in zip(rest, ops.node2edge_grouped([(xpq[i][0], xpq[i][1], Hs[i], pks[i]["w2"], pks[i]["b2"], syms[i])
                                                         for i in rest])):
                edges[i] = e
        for i in range(n):
            if fuse[i]:
                edges[i] = ops.PoolSpec(xpq[i][0], xpq[i][1], Hs[i], pks[i]["w2"], pks[i]["b2"], syms[i])
        if traces is not None:      # kept for the backward: nothing of this round is re-computed there
            for t, kd, (xp, pq), e in zip(traces, keep, xpq, edges):
                t.n2e.append(dict(x1=kd["hid"], xp=xp, pq=pq, edges=e))
        return edges

    def edge_mlp(stages, edges: Sequence[Tensor], want_dist: bool):
        # draws happen module by module, in the order given — the reference's RNG order per call site
        us = [next_u(i) for i in range(n)]
        keep = [] if traces is not None else None
        res = ops.edge_mlp_gumbel_grouped([(e, u, st._packed(), st.bottleneck_dim, N if sy else 0, want_dist)
                                           for e, u, st, sy in zip(edges, us, stages, syms)], _GUMBEL_TAU, keep)
        if traces is not None:
            for t, kd in zip(traces, keep):
                t.estage.append(kd)
        return res

    def edge2node(edge_feats: Sequence[Tensor], oris: Sequence[Tensor], idx: int, closing=None) -> List[Tensor]:
        """-> the inputs of the stage's closing MLPs (tensors / ScatterSpec / NodeAggSpec); with ``closing`` = [(packed
        MLP, out or None)] per module and a launch shape that allows it, the closing MLPs' OUTPUTS (the aggregation launch
        applies them itself) as a `_Closed` list."""
        aggs = [m.edge_aggregation_list[idx] for m in mods]
        items = []
        # larger graphs: eo = H @ ori of every hyper module from ONE stand-alone gather launch
        standalone = [i for i in range(n) if not syms[i] and N > _FUSED_GATHER_MAX_N]
        eos = dict(zip(standalone, ops.agg_gather_grouped([(oris[i], Hs[i]) for i in standalone]))) if standalone else {}
        for i in range(n):
            pk, K = aggs[i]._packed(), aggs[i].edge_types
            if syms[i] and not twin and pair_A[i] is not None:
                # pairwise: eo = ori_i + ori_j makes the typed MLP's first layer linear in the two nodes, so
                # it ran once per node (N rows instead of N(N+1)/2 pairs) in this round's node stage; the
                # pair form does the rest
                src = ops.PairSpec(pair_A[i], node=(ops.node_form_enabled() and ops.BF16X6 and N <= ops.NODE_FORM_MAX_N
                                                     and K <= ops.NODE_FORM_MAX_K and N <= _FUSED_SCATTER_MAX_N))
            elif syms[i]:
                # bf16 twin: both layers per NODE, one scene per workgroup (node form), or per unordered pair
                src = ops.GatherSpec(oris[i], None, True,
                                     node=twin and ops.node_form_enabled() and N <= ops.SCENE_FORM_MAX_N)
            elif N <= _FUSED_GATHER_MAX_N:
                src = ops.GatherSpec(oris[i], Hs[i], False)   # eo = H @ ori formed inside the kernel
            else:
                src = eos[i]
            items.append((src, edge_feats[i], pk, K))
        if (fuse_closing and closing is not None and traces is None and not twin and N <= _FUSED_SCATTER_MAX_N
                and len({(-1 if o is None else o.stride(-2)) for _, o in closing}) == 1
                and ops.closing_fusable(items, [pk2 for pk2, _ in closing])):
            return _Closed(ops.agg_mlp_grouped(items, [(pk2, o, ori) for (pk2, o), ori in zip(closing, oris)]))
        feats = ops.agg_mlp_grouped(items)

        def node_item(it) -> bool:
            return isinstance(it[0], (ops.PairSpec, ops.GatherSpec)) and it[0].node
        if N <= _FUSED_SCATTER_MAX_N:
            # cat(H^T feat, ori) / N is formed inside the MLP kernel that consumes it (node form: H^T feat is what the
            # aggregation kernel wrote)
            return [ops.NodeAggSpec(f, o) if node_item(it) else ops.ScatterSpec(f, H, o, sy)
                    for f, H, o, sy, it in zip(feats, Hs, oris, syms, items)]
        # larger graphs: one stand-alone scatter launch for the modules whose aggregation wrote per-edge features
        rest = [i for i in range(n) if not node_item(items[i])]
        scat = dict(zip(rest, ops.agg_scatter_grouped([(feats[i], Hs[i], oris[i], syms[i]) for i in rest]))) if rest else {}
        return [scat[i] if i in scat else ops.NodeAggSpec(feats[i], oris[i]) for i in range(n)]

    res = edge_mlp([m.nmp_mlp_start for m in mods], node2edge(hs, 0), True)
    edge_feats, factors = [r[0] for r in res], [r[1] for r in res]
    if traces is not None:
        for t, r in zip(traces, res):
            t.dists.append(r[1])
    node_feats, idx = list(hs), 0
    for l in range(2 * (nmp - 1)):
        stages = [m.nmp_mlps[l] for m in mods]
        if l % 2 == 0:
            agg = edge2node(edge_feats, node_feats, idx, [(m._packed_mlp2(st), None) for m, st in zip(mods, stages)])
            keep = [] if traces is not None else None
            node_feats = (list(agg) if isinstance(agg, _Closed) else
                          ops.mlp2_grouped([(a, m._packed_mlp2(st), None) for a, m, st in zip(agg, mods, stages)], keep))
            idx += 1
            if traces is not None:
                for t, x, kd in zip(traces, node_feats, keep):
                    t.xs.append(x)
                    t.tails.append(kd)
        else:
            res = edge_mlp(stages, node2edge(node_feats, idx), traces is not None)
            edge_feats = [r[0] for r in res]
            if traces is not None:
                for t, r in zip(traces, res):
                    t.dists.append(r[1])
    agg = edge2node(edge_feats, node_feats, idx, [(m._packed_mlp2(m.nmp_mlp_end), o) for m, o in zip(mods, outs)])
    if isinstance(agg, _Closed):
        return list(zip(list(agg), factors))
    ends = [(a, m._packed_mlp2(m.nmp_mlp_end), o) for a, m, o in zip(agg, mods, outs)]
    # the last MLP writes in place when `out` is given; grouped when every group has the same stride
    strides = {(-1 if o is None else o.stride(-2)) for o in outs}
    keep = [] if traces is not None else None
    if len(strides) == 1:
        node_feats = ops.mlp2_grouped(ends, keep)
    else:
        node_feats = [ops.mlp2_grouped([e], keep)[0] for e in ends]
    if traces is not None:
        for t, kd in zip(traces, keep):
            t.tails.append(kd)
    return list(zip(node_feats, factors))


class MS_HGNN_oridinary(_MessagePassing):
    """Pairwise (fully connected, self-loops included) message passing — drop-in for the reference
    class of the same (misspelt) name, model/MS_HGNN_batch.py:55-198.  ``forward(h_states)`` ->
    ``(node_feat (B,N,bottleneck), factors (B,N*N,6))``.  The N^2 x N incidence the reference
    rebuilds with numpy on every call (:143-160) is never materialised."""

    def __init__(self, embedding_dim=64, h_dim=64, mlp_dim=1024, bottleneck_dim=1024, activation='relu',
                 batch_norm=True, dropout=0.0, nmp_layers=4, vis=False):
        super().__init__()
        self.mlp_dim = mlp_dim
        self.h_dim = h_dim
        self.bottleneck_dim = bottleneck_dim
        self.embedding_dim = embedding_dim
        self.nmp_layers = nmp_layers
        self.batch_norm = batch_norm
        self.activation = activation
        self.vis = vis
        self.hdim_extend = _HDIM_EXTEND
        self.edge_types = 6   # model/MS_HGNN_batch.py:74
        self._build(h_dim, bottleneck_dim, nmp_layers)

    # reference-named stage methods (rel_rec / rel_send are accepted and ignored: the graph is implicit)
    def node2edge(self, x, rel_rec=None, rel_send=None, idx=0):
        return self._node2edge(x, None, idx)

    def edge2node(self, x, rel_rec, rel_send, ori, idx):
        return self._edge2node(x, ori, None, idx)

    def forward(self, h_states, noise_u=None, out=None):
        """``out`` (optional): where node_feat is written, e.g. a column block of the caller's
        concatenated feature tensor.  Under autograd (an input or a parameter requires grad) the call
        goes through `groupnet_amd.backward.MSHGNNFunction`: same fused forward, HIP backward."""
        ops._req(h_states, "h_states", (None, None, self.h_dim), ops._ACT_DTYPES)
        N = h_states.shape[1]
        if h_states.shape[0] and N and _needs_grad(self, h_states):
            return self._forward_autograd(h_states, None, noise_u, out)
        if h_states.shape[0] == 0 or N == 0:      # empty batch: nothing to launch
            nf = out if out is not None else h_states.new_empty((h_states.shape[0], N, self.bottleneck_dim))
            return nf, h_states.new_empty((h_states.shape[0], N * N, self.edge_types))
        return self._run(h_states, None, N * N, noise_u, out)


class MS_HGNN_hyper(_MessagePassing):
    """Top-k hypergraph message passing at one group size — drop-in for
    model/MS_HGNN_batch.py:270-443 (listall=False, the only mode the reference runs, :312).
    ``forward(h_states, corr)`` -> ``(node_feat (B,N,bottleneck), factor (B,E,10), H (B,E,N))``
    with E = 1 when scale == N, else N."""

    def __init__(self, embedding_dim=64, h_dim=64, mlp_dim=1024, bottleneck_dim=1024, activation='relu',
                 batch_norm=True, dropout=0.0, nmp_layers=4, scale=2, vis=False, actor_number=11):
        super().__init__()
        self.mlp_dim = mlp_dim
        self.h_dim = h_dim
        self.bottleneck_dim = bottleneck_dim
        self.embedding_dim = embedding_dim
        self.nmp_layers = nmp_layers
        self.batch_norm = batch_norm
        self.activation = activation
        self.scale = scale
        self.vis = vis
        # never used by forward, present in reference checkpoints (model/MS_HGNN_batch.py:290-291)
        self.spatial_embedding = nn.Linear(2, embedding_dim)
        self.spatial_transform = nn.Linear(h_dim, h_dim)
        self.hdim_extend = _HDIM_EXTEND
        self.edge_types = 10  # model/MS_HGNN_batch.py:294
        self._build(h_dim, bottleneck_dim, nmp_layers)
        self.listall = False

    def init_adj_attention(self, feat, feat_corr, scale_factor=2):
        """H (B,E,N) from the affinity matrix (model/MS_HGNN_batch.py:372-388)."""
        if feat_corr.dtype == torch.bfloat16:     # ranked in fp32 either way; fp32 corr keeps near-ties apart
            feat_corr = feat_corr.float()
        ops._req(feat_corr, "corr", (feat.shape[0], feat.shape[1], feat.shape[1]))
        return ops.topk_incidence(feat_corr, [int(scale_factor)])[0]

    def init_adj_attention_listall(self, feat, feat_corr, scale_factor=2):
        """H (B,E,N) by exhaustive search of the best group per agent (model/MS_HGNN_batch.py:390-414);
        used by forward when ``self.listall`` is set (hard-coded False in the reference, :312)."""
        ops._req(feat_corr, "corr", (feat.shape[0], feat.shape[1], feat.shape[1]))
        return ops.listall_incidence(feat_corr, int(scale_factor))

    def _build_H(self, h_states, corr):
        build = self.init_adj_attention_listall if self.listall else self.init_adj_attention
        return build(h_states, corr, scale_factor=self.scale)

    def node2edge(self, x, H, idx=0):
        return self._node2edge(x, H, idx)

    def edge2node(self, x, ori, H, idx):
        return self._edge2node(x, ori, H, idx)

    def forward(self, h_states, corr, noise_u=None, H=None, out=None):
        """``H`` (optional) lets a caller that already built the incidence for every scale in one
        fused launch (``ops.affinity_topk``) hand it in; by default it is built here from ``corr``."""
        ops._req(h_states, "h_states", (None, None, self.h_dim), ops._ACT_DTYPES)
        if h_states.shape[0] and _needs_grad(self, h_states):
            # H is a constant of the backward (top-k selection has no gradient; corr is only used to build it)
            if H is None:
                H = self._build_H(h_states.detach(), corr.detach())
            node_feat, factor = self._forward_autograd(h_states, H, noise_u, out)
            return node_feat, factor, H
        if h_states.shape[0] == 0:                  # empty batch: nothing to launch
            B, N = h_states.shape[0], h_states.shape[1]
            if self.scale > N:
                raise RuntimeError("selected index k out of range")
            E = 1 if self.scale == N else N
            nf = out if out is not None else h_states.new_empty((B, N, self.bottleneck_dim))
            return nf, h_states.new_empty((B, E, self.edge_types)), h_states.new_empty((B, E, N))
        if H is None:
            H = self._build_H(h_states, corr)
        else:
            ops._req(H, "H", (h_states.shape[0], None, h_states.shape[1]))
        node_feat, factor = self._run(h_states, H, H.shape[1], noise_u, out)
        return node_feat, factor, (H if H.dtype == h_states.dtype else H.to(h_states.dtype))    # type_as(feat), :376,384
