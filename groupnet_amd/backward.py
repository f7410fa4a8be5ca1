"""Backward of the MS-HGNN modules on HIP, behind torch.autograd.

`train_hyper_nba.py:116` back-propagates through `MS_HGNN_oridinary` / `MS_HGNN_hyper`.  The forward
of the drop-in modules is the fused matrix-core path; when a gradient is needed a call goes through
`MSHGNNFunction`: forward = that same fused path (keeping only the node features entering each
message-passing round and the `dist` each round sampled, which carries the Gumbel noise), backward =
this file — the chain rule of model/MS_HGNN_batch.py:41-53, 116-141, 247-268, 357-370 written out stage
by stage for SEVERAL modules at once, every stage one grouped launch of libgroupnet_hip.so:

  * all dense contractions (re-computation of hidden activations, dX = dY W, dW = dY^T X with the bias
    gradient as a side output) are problems of `gn_gemm_grouped_f32` — fp32 matrix cores, many problems
    per launch; every weight gradient of a round goes into ONE launch at the end of the round;
  * the K typed MLPs of the aggregation are handled as one wide layer (hidden (rows, K*128)) plus
    `gn_typed_bwd_f32`;
  * `gn_gumbel_bwd_f32`, `gn_node2edge_bwd_f32` for the two non-GEMM stages; gather and scatter are each
    other's adjoints and reuse the forward kernels.

A round j of a module is two blocks: x_j --node2edge_j, edge MLP_j--> (ef_j, dist_j) and
(ef_j, x_j) --edge2node_j, MLP--> x_{j+1} (or node_feat).  `nmp_layers > 1` (:186-194, :432-440) is
the same two blocks repeated, walked backwards.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import _P, check, load, stream_handle

Tensor = torch.Tensor
_TAU = 0.5
_HID = 128           # hidden width of the typed aggregation MLPs (model/MS_HGNN_batch.py:253-255)
_LGF_LD = 32         # leading dimension of the (logits | factor pre-activation) buffer


def _p(t: Optional[Tensor]):
    return _P(0 if t is None else t.data_ptr())


class GemmBatch:
    """Collects GEMM problems (2-D row-major views, unit column stride) and runs them as grouped launches."""

    def __init__(self):
        self.descs: List[_lib.GemmDesc] = []
        self.keep: List[Tensor] = []
        self.device = None

    def add(self, A: Tensor, Bm: Tensor, C: Tensor, tA=False, tB=False, bias=None, mask=None, relu=False, alpha=1.0,
            beta=0.0, rs: Optional[Tensor] = None, colsum: Optional[Tensor] = None, accum=False, tC=False) -> Tensor:
        for t in (A, Bm, C) + ((mask,) if mask is not None else ()):
            if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1) or t.dtype != torch.float32:
                raise ValueError("gemm: 2-D fp32 views with unit column stride")
        M, K = (A.shape[1], A.shape[0]) if tA else (A.shape[0], A.shape[1])
        K2, N = (Bm.shape[1], Bm.shape[0]) if tB else (Bm.shape[0], Bm.shape[1])
        if K != K2 or tuple(C.shape) != ((N, M) if tC else (M, N)) or (mask is not None and tuple(mask.shape) != (M, N)):
            raise ValueError(f"gemm: ({M},{K}) x ({K2},{N}) -> {tuple(C.shape)}")
        if rs is not None and (rs.dim() != 1 or rs.shape[0] != A.shape[0]):
            raise ValueError("gemm: rs scales the stored rows of A")
        if colsum is not None and (not tA or colsum.numel() != M or not colsum.is_contiguous()):
            raise ValueError("gemm: colsum needs transA and M contiguous entries")
        flags = (_lib.GEMM_TRANS_A if tA else 0) | (_lib.GEMM_TRANS_B if tB else 0) | \
                (_lib.GEMM_RELU if relu else 0) | (_lib.GEMM_ACCUM if accum else 0) | (_lib.GEMM_TRANS_C if tC else 0)
        if tC and not accum:
            raise ValueError("gemm: tC (write the product transposed) exists for accumulate mode only")
        ld = lambda t: max(t.stride(0), t.shape[1])
        self.descs.append(_lib.GemmDesc(A.data_ptr(), Bm.data_ptr(), C.data_ptr(), _p(bias).value, _p(mask).value,
                                        _p(rs).value, _p(colsum).value, M, N, K, ld(A), ld(Bm), ld(C),
                                        0 if mask is None else ld(mask), 0 if rs is None else rs.stride(0), flags,
                                        float(alpha), float(beta)))
        self.keep += [t for t in (A, Bm, C, bias, mask, rs, colsum) if t is not None]
        self.device = A.device
        return C

    def run(self) -> None:
        if not self.descs:
            return
        arr = (_lib.GemmDesc * len(self.descs))(*self.descs)
        with torch.cuda.device(self.device):
            check(load().gn_gemm_grouped_f32(arr, len(self.descs), stream_handle()), "gn_gemm_grouped_f32")
        self.descs, self.keep = [], []


def gemm(A: Tensor, Bm: Tensor, transA=False, transB=False, bias=None, mask=None, relu=False, alpha=1.0,
         out: Optional[Tensor] = None, beta=0.0) -> Tensor:
    """out (M,N) = beta*out + alpha*op(A) op(B) (+bias) (relu) (zeroed where mask <= 0) — one problem."""
    M = A.shape[1] if transA else A.shape[0]
    N = Bm.shape[0] if transB else Bm.shape[1]
    if out is None:
        out, beta = torch.empty((M, N), dtype=A.dtype, device=A.device), 0.0
    gb = GemmBatch()
    gb.add(A, Bm, out, transA, transB, bias, mask, relu, alpha, beta)
    gb.run()
    return out


def axpby(out: Tensor, a: Tensor, alpha=1.0, beta=0.0) -> Tensor:
    """out = alpha*a + beta*out on 2-D (possibly column-sliced) views."""
    assert out.shape == a.shape and out.stride(1) == 1 and a.stride(1) == 1
    with torch.cuda.device(a.device):
        check(load().gn_axpby2d_f32(_p(out), out.stride(0), _p(a), a.stride(0), a.shape[0], a.shape[1], float(alpha),
                                    float(beta), stream_handle()), "gn_axpby2d_f32")
    return out


def pairwise_incidence(B: int, N: int, device, dtype=torch.float32) -> Tensor:
    """The (B, N*N, N) incidence rel_rec + rel_send of the pairwise graph (model/MS_HGNN_batch.py:118,
    143-160): edge e = i*N + j has weight 1 on i and on j, 2 when i == j.  A constant; only the backward
    of node2edge materialises it."""
    e = torch.arange(N * N, device=device)
    H = torch.zeros(N * N, N, dtype=dtype, device=device)
    H[e, e % N] += 1
    H[e, e // N] += 1
    return H[None].expand(B, -1, -1).contiguous()


_pair_w_cache: Dict[tuple, Tensor] = {}


def _pair_row_weights(B: int, N: int, device) -> Tensor:
    """(B * N(N+1)/2) incidence weight of each unordered-pair row on its nodes: 2 on self-loops, else 1."""
    key = (B, N, str(device))
    w = _pair_w_cache.get(key)
    if w is None:
        i = torch.arange(N, device=device)
        one = torch.ones(ops.pair_count(N), dtype=torch.float32, device=device)
        one[i * N - (i * (i - 1)) // 2] = 2.0          # row of the pair (i, i)
        w = _pair_w_cache[key] = one.repeat(B).contiguous()
        if len(_pair_w_cache) > 16:
            _pair_w_cache.pop(next(iter(_pair_w_cache)))
    return w


class _Pool:
    """Zero-initialised scratch handed out in slices: every accumulate-by-atomics target of a round comes
    from one fill instead of one fill each."""

    def __init__(self, numel: int, device, buf: Optional[Tensor] = None):
        # (buf: a zeroed slice of a larger fill — the modules of a round share ONE fill)
        self.buf = torch.zeros(numel, dtype=torch.float32, device=device) if buf is None else buf
        self.used = 0

    def take(self, *shape: int) -> Tensor:
        n = 1
        for s in shape:
            n *= s
        n4 = (n + 3) // 4 * 4
        if self.used + n4 > self.buf.numel():
            return torch.zeros(shape, dtype=torch.float32, device=self.buf.device)
        out = self.buf[self.used:self.used + n].view(*shape)
        self.used += n4
        return out


class ModuleTrace:
    """What one module's fused forward keeps for its backward."""

    def __init__(self, mod, h: Tensor, H: Optional[Tensor]):
        self.mod, self.H = mod, H
        self.xs: List[Tensor] = [h]          # node features entering round j
        self.dists: List[Tensor] = []        # dist of round j (B,E,K)
        self.tails: List[dict] = []          # round j's closing MLP: {"x": cat(H^T feat, ori)/N, "hid": relu(layer 0)}
        self.n2e: List[dict] = []            # round j: {"x1" hidden of node2edge_start_mlp, "xp", "pq", "edges"}
        self.estage: List[dict] = []         # round j's edge MLP: {"z1", "z", "dh1", "lgf"}


def _round_layers(mod, j: int):
    L = mod.nmp_layers - 1
    stage = mod.nmp_mlp_start if j == 0 else mod.nmp_mlps[2 * j - 1]
    tail = mod.nmp_mlp_end if j == L else mod.nmp_mlps[2 * j]
    return (mod.node2edge_start_mlp[j].layers, mod.attention_mlp[j].layers, stage, mod.edge_aggregation_list[j],
            tail.layers)


def _bwd_weights(mod, j: int) -> dict:
    """The concatenated weight matrices round j's backward GEMMs read — the K typed MLPs as one wide layer,
    MLP_distribution | MLP_factor side by side, the split attention layer 0 — assembled from the parameters
    by ONE `PackPlan` launch (no torch.cat, capturable).  A backward is by definition part of a training
    step, where parameters may have been rewritten through `.data` without a version bump: the plan (arena and
    segment table) is cached per parameter addresses, its one refresh launch runs on every call."""
    from .MS_HGNN_batch import _param_key
    cache = mod.__dict__.setdefault("_bwd_cat", {})
    hit = cache.get(j)
    (s0, s1), (a0, a1), st, agg, _ = _round_layers(mod, j)
    d0, d1 = st.MLP_distribution.layers
    f0, f1 = st.MLP_factor.layers
    l0 = [m.layers[0] for m in agg.agg_mlp]
    l1 = [m.layers[1] for m in agg.agg_mlp]
    params = [a0.weight, a0.bias, d0.weight, d0.bias, d1.weight, d1.bias, f0.weight, f0.bias, f1.weight, f1.bias]
    params += [p for l in l0 + l1 for p in (l.weight, l.bias)]
    ptrs = tuple(p.data_ptr() for p in params)
    if hit is None or hit[0] != ptrs:
        K, D = mod.edge_types, ops.FEAT
        plan = ops.PackPlan(params[0].device)
        off = dict(W1cat=plan.alloc(K * _HID * D), b1cat=plan.alloc(K * _HID), W2cat=plan.alloc(D * K * _HID),
                   b2mat=plan.alloc(K * D), Wd0=plan.alloc(256 * D), bd0=plan.alloc(256), Wd1=plan.alloc(_LGF_LD * 256),
                   bd1=plan.alloc(_LGF_LD), Wpq=plan.alloc(D * D), bpq=plan.alloc(D))
        for k in range(K):
            plan.place(off["W1cat"], D, l0[k].weight, place_r=k * _HID)           # (K*128, 64)
            plan.place(off["b1cat"], 0, l0[k].bias, place_c=k * _HID)
            plan.place(off["W2cat"], K * _HID, l1[k].weight, place_c=k * _HID)     # (64, K*128)
            plan.place(off["b2mat"], 0, l1[k].bias, place_c=k * D)                 # (K, 64)
        plan.place(off["Wd0"], D, d0.weight)                                       # hidden layers side by side
        plan.place(off["Wd0"], D, f0.weight, place_r=128)
        plan.place(off["bd0"], 0, d0.bias)
        plan.place(off["bd0"], 0, f0.bias, place_c=128)
        plan.place(off["Wd1"], 256, d1.weight)                                     # rows 0..K-1: logits over hidden[:128]
        plan.place(off["Wd1"], 256, f1.weight, place_r=K, place_c=128)             # row K: factor over hidden[128:]
        plan.place(off["bd1"], 0, d1.bias)
        plan.place(off["bd1"], 0, f1.bias, place_c=K)
        plan.place(off["Wpq"], D, a0.weight[:, :D])                                # P = W[:, :64] x' + b
        plan.place(off["Wpq"], D, a0.weight[:, D:], place_r=32)                    # Qn = W[:, 64:] x'
        plan.place(off["bpq"], 0, a0.bias)
        plan.finish()
        shapes = dict(W1cat=(K * _HID, D), b1cat=(K * _HID,), W2cat=(D, K * _HID), b2mat=(K, D), Wd0=(256, D), bd0=(256,),
                      Wd1=(_LGF_LD, 256), bd1=(_LGF_LD,), Wpq=(D, D), bpq=(D,))
        cat = {n: plan.view(off[n], math.prod(shp)).view(*shp) for n, shp in shapes.items()}
        hit = cache[j] = [ptrs, cat, plan, None, params]
    hit[4] = params
    hit[2].refresh()
    hit[3] = _param_key(params)
    return hit[1]


def round_backward(traces: Sequence[ModuleTrace], j: int, g_ys: Sequence[Optional[Tensor]],
                   g_dists: Sequence[Optional[Tensor]], grads: Dict[nn.Parameter, Tensor]) -> List[Tensor]:
    """Back through round j of several modules at once.

    g_ys[i] = gradient w.r.t. the output of round j's edge2node + MLP block (node_feat for the last round),
    g_dists[i] = gradient w.r.t. dist_j (the returned `factors` for round 0), either may be None.
    Adds the parameter gradients to `grads`, returns d x_j per module."""
    n = len(traces)
    W = lambda l: l.weight.detach()
    b = lambda l: l.bias.detach()
    D = ops.FEAT
    S = []           # per-module state of this round
    for t, g_y, g_d in zip(traces, g_ys, g_dists):
        mod, x = t.mod, t.xs[j]
        B, N, _ = x.shape
        Hx = t.H
        scene_fits = (4 * N * D + 64 + 16 * N) * 4 <= 150 * 1024     # gn_node2edge_bwd_f32's per-scene form
        # The pairwise graph is symmetric: the ordered edges (i,j) and (j,i) pool the same feature and feed
        # the same typed MLP, so (as in the forward) every per-edge stage runs on the N(N+1)/2 unordered
        # pairs; only dist (its own Gumbel noise per ordered edge) stays ordered.
        sym = Hx is None and scene_fits
        if Hx is None and not sym:
            Hx = t.__dict__.setdefault("_pair_H", None)
            if Hx is None:
                Hx = t._pair_H = pairwise_incidence(B, N, x.device, x.dtype)
        E = ops.pair_count(N) if sym else Hx.shape[1]
        K = mod.edge_types
        (s0, s1), (a0, a1), st, agg, (e0, e1) = _round_layers(mod, j)
        npar = t.__dict__.setdefault("_npar", {}).get(j)
        if npar is None:
            npar = t._npar[j] = sum(p.numel() for m in (mod.node2edge_start_mlp[j], mod.attention_mlp[j], st, agg,
                                                        nn.ModuleList([e0, e1])) for p in m.parameters())
        S.append(dict(mod=mod, x=x, x2=x.reshape(B * N, D), H=Hx, sym=sym, B=B, N=N, E=E, K=K, R=B * E, s0=s0, s1=s1, a0=a0, a1=a1, i=st.init_MLP.layers, d=st.MLP_distribution.layers,
                      f=st.MLP_factor.layers, agg=agg, tw=_bwd_weights(mod, j), e0=e0, e1=e1,
                      tail=t.tails[j], dist=t.dists[j].reshape(-1, K), g_y=None if g_y is None else g_y.reshape(B * N, -1).contiguous(),
                      g_d=None if g_d is None else g_d.reshape(-1, K).contiguous(),
                      pool_n=(npar + 2 * B * N * D + B * E * (2 * K + 4 + 2 * D) + 4096 + 63) // 64 * 64))
    dev = S[0]["x"].device
    # one zero fill for the accumulate-by-atomics targets of every module of the round
    arena = torch.zeros(sum(c["pool_n"] for c in S), dtype=torch.float32, device=dev)
    off = 0
    for c in S:
        c["pool"] = _Pool(c["pool_n"], dev, arena[off:off + c["pool_n"]])
        off += c["pool_n"]
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    gb = GemmBatch()

    def stage(fn):
        for c in S:
            fn(c)
        gb.run()

    with torch.no_grad():
        # ---------------- what the fused forward kept (training mode writes these on request) ----------------
        # The forward runs the pairwise module on unordered pairs; if this backward has to use ordered edge rows
        # (N too large for the per-scene node2edge backward), its edge-row activations are re-computed instead.
        kept = all(c["sym"] or c["H"] is not None and tr.H is not None for c, tr in zip(S, traces))
        for c, tr in zip(S, traces):
            c["w2"], c["b2"] = W(c["a1"])[0], b(c["a1"])
            if kept:
                a, e = tr.n2e[j], tr.estage[j]
                c.update(x1=a["x1"], xp=a["xp"].view(-1, D), pq=a["pq"].view(-1, D), edges=a["edges"].view(c["R"], D),
                         z1=e["z1"], z=e["z"], dh1=e["dh1"], lgf=e["lgf"])
        if not kept:
            stage(lambda c: c.update(x1=gb.add(c["x2"], W(c["s0"]), new(c["B"] * c["N"], 256), tB=True, bias=b(c["s0"]),
                                               relu=True)))
            stage(lambda c: c.update(xp=gb.add(c["x1"], W(c["s1"]), new(c["B"] * c["N"], D), tB=True, bias=b(c["s1"]))))
            # attention layer 0 on cat(x'_n, e0_e), split by linearity: P = W[:, :64] x' + b, Qn = W[:, 64:] x'
            stage(lambda c: c.update(pq=gb.add(c["xp"], c["tw"]["Wpq"], new(c["B"] * c["N"], D), tB=True, bias=c["tw"]["bpq"])))
            edges = ops.node2edge_grouped([(c["xp"].view(c["B"], c["N"], D), c["pq"].view(c["B"], c["N"], D), c["H"], c["w2"],
                                            c["b2"], c["sym"]) for c in S])
            for c, e in zip(S, edges):
                c["edges"] = e.view(c["R"], D)
            stage(lambda c: c.update(z1=gb.add(c["edges"], W(c["i"][0]), new(c["R"], 128), tB=True, bias=b(c["i"][0]),
                                               relu=True)))
            stage(lambda c: c.update(z=gb.add(c["z1"], W(c["i"][1]), new(c["R"], D), tB=True, bias=b(c["i"][1]))))
            # hidden layers of MLP_distribution | MLP_factor side by side, then (logits | factor pre-activation)
            stage(lambda c: c.update(dh1=gb.add(c["z"], c["tw"]["Wd0"], new(c["R"], 256), tB=True, bias=c["tw"]["bd0"],
                                                relu=True)))
            stage(lambda c: c.update(lgf=gb.add(c["dh1"], c["tw"]["Wd1"], new(c["R"], _LGF_LD), tB=True, bias=c["tw"]["bd1"])))
        for c in S:
            # ef per edge row; for pair rows ef_ij + ef_ji (self-loop rows not doubled: the typed backward below
            # works with dfeat = the pair gather of d(H^T feat), which carries the self-loop's 2)
            # (leading dimension padded to a multiple of 4, zero-filled: the GEMMs reading ef stay on the vector path)
            c["Kp"] = (c["K"] + 3) // 4 * 4
            c["ef"] = c["pool"].take(c["R"], c["Kp"])
            with torch.cuda.device(dev):
                check(load().gn_gumbel_ef_f32(_p(c["dist"]), _p(c["lgf"]), _p(c["ef"]), c["R"], c["K"], _LGF_LD,
                                              c["N"] if c["sym"] else 0, 1.0, c["Kp"], stream_handle()), "gn_gumbel_ef_f32")
            c["def"] = c["pool"].take(c["R"], c["K"])
            c["dx"] = None

        live = [c for c in S if c["g_y"] is not None]
        if live:
            S_all, S = S, live
            eos = ops.agg_gather_grouped([(c["x"], c["H"], c["sym"]) for c in S])
            for c, eo in zip(S, eos):
                c["eo"], c["eo2"] = eo, eo.view(c["R"], D)
            stage(lambda c: c.update(Hc=gb.add(c["eo2"], c["tw"]["W1cat"], new(c["R"], c["K"] * _HID), tB=True,
                                               bias=c["tw"]["b1cat"], relu=True)))
            # cat(H^T feat, ori) / N and the closing MLP's hidden layer were kept by the forward (the fused
            # scatter+MLP kernel writes them on request), so feat is not re-computed
            for c in S:
                c["agg2"], c["y1"] = c["tail"]["x"], c["tail"]["hid"]
            # ---------------- back through MLP(edge2node) ----------------
            stage(lambda c: c.update(dy1=gb.add(c["g_y"], W(c["e1"]), new(c["B"] * c["N"], 128), mask=c["y1"])))

            def dagg_stage(c):
                inv = 1.0 / c["N"]
                c["da"] = gb.add(c["dy1"], W(c["e0"])[:, :D], new(c["B"] * c["N"], D), alpha=inv)   # d(H^T feat)
                c["dx"] = gb.add(c["dy1"], W(c["e0"])[:, D:], new(c["B"] * c["N"], D), alpha=inv)   # ori half of the cat
            stage(dagg_stage)
            dfeats = ops.agg_gather_grouped([(c["da"].view(c["B"], c["N"], D), c["H"], c["sym"]) for c in S])   # adjoint of H^T feat
            for c, df in zip(S, dfeats):
                c["dfeat"] = df.view(c["R"], D)
            stage(lambda c: c.update(T=gb.add(c["dfeat"], c["tw"]["W2cat"], new(c["R"], c["K"] * _HID))))
            for c in S:
                with torch.cuda.device(dev):
                    check(load().gn_typed_bwd_f32(_p(c["T"]), _p(c["Hc"]), _p(c["ef"]), c["Kp"], _p(c["dfeat"]), _p(c["tw"]["b2mat"]),
                                                  _p(c["def"]), c["R"], c["K"], _HID, stream_handle()), "gn_typed_bwd_f32")
            # (pair rows: the self-loop's eo = 2 ori, and the pair-form scatter below weighs every row once)
            # (K = K_types*128 is long and the output has few tiles: split K over workgroups, atomic accumulate)
            stage(lambda c: c.update(deo=gb.add(c["T"], c["tw"]["W1cat"], c["pool"].take(c["R"], D), accum=True,
                                                rs=_pair_row_weights(c["B"], c["N"], dev) if c["sym"] else None)))
            # eo = H ori  ->  d ori += H^T d eo  (the scatter kernel with divisor 1; its ori half is unused)
            scs = ops.agg_scatter_grouped([(c["deo"].view(c["B"], c["E"], D), c["H"], c["x"], c["sym"]) for c in S], 1.0)
            for c, sc in zip(S, scs):
                axpby(c["dx"], sc.view(c["B"] * c["N"], 2 * D)[:, :D], 1.0, 1.0)

            def e2n_weight_grads(c):
                K, pool = c["K"], c["pool"]
                for lin, dY, X in ((c["e1"], c["g_y"], c["y1"]), (c["e0"], c["dy1"], c["agg2"])):
                    grads[lin.weight] = gb.add(dY, X, pool.take(*lin.weight.shape), tA=True, accum=True,
                                               colsum=grads.setdefault(lin.bias, pool.take(*lin.bias.shape)))
                gW1 = gb.add(c["T"], c["eo2"], pool.take(K * _HID, D), tA=True, accum=True,
                             colsum=c.setdefault("gb1", pool.take(K * _HID)))
                gb2 = gb.add(c["ef"], c["dfeat"], pool.take(c["Kp"], D), tA=True, accum=True)     # rows >= K stay zero
                for k, m in enumerate(c["agg"].agg_mlp):
                    l0, l1 = m.layers
                    grads[l0.weight], grads[l0.bias] = gW1[k * _HID:(k + 1) * _HID], c["gb1"][k * _HID:(k + 1) * _HID]
                    grads[l1.bias] = gb2[k]
                    # dW2_k = sum_r ef[r,k] dfeat[r] (x) h_k[r]
                    # (as (ef_k h_k)^T dfeat written transposed: 128-row tiles instead of half-empty 64-row ones)
                    grads[l1.weight] = gb.add(c["Hc"][:, k * _HID:(k + 1) * _HID], c["dfeat"], pool.take(D, _HID), tA=True,
                                              accum=True, rs=c["ef"][:, k], tC=True)
            stage(e2n_weight_grads)
            S = S_all

        # ---------------- back through the edge MLP + Gumbel softmax ----------------
        # (one grouped launch for all modules: these launches take ~14 us whatever their size)
        arr = (_lib.GumbelBwdGroup * len(S))()
        for i, c in enumerate(S):
            c["dlgf"] = new(c["R"], _LGF_LD)
            arr[i] = _lib.GumbelBwdGroup(_p(c["dist"]), _p(c["lgf"]), _p(c["def"]), _p(c["g_d"]), _p(c["dlgf"]), c["R"], c["K"],
                                         c["N"] if c["sym"] else 0)
        with torch.cuda.device(dev):
            check(load().gn_gumbel_bwd_grouped_f32(arr, len(S), _LGF_LD, _TAU, stream_handle()), "gn_gumbel_bwd_grouped_f32")

        stage(lambda c: c.update(dd1=gb.add(c["dlgf"], c["tw"]["Wd1"], new(c["R"], 256), mask=c["dh1"])))
        stage(lambda c: c.update(dz=gb.add(c["dd1"], c["tw"]["Wd0"], new(c["R"], D))))
        stage(lambda c: c.update(dz1=gb.add(c["dz"], W(c["i"][1]), new(c["R"], 128), mask=c["z1"])))
        stage(lambda c: c.update(dedges=gb.add(c["dz1"], W(c["i"][0]), new(c["R"], D))))
        # ---------------- back through the attention-weighted pooling ----------------
        for c in S:
            pool, BN = c["pool"], c["B"] * c["N"]
            c["dxp"], c["dpq"] = pool.take(BN, D), pool.take(BN, D)
            grads[c["a1"].weight], grads[c["a1"].bias] = pool.take(1, 32), pool.take(1)
        N0 = S[0]["N"]
        scene_form = (4 * N0 * D + 64 + 16 * N0) * 4 <= 150 * 1024 and all(c["B"] == S[0]["B"] and c["N"] == N0 for c in S)
        if scene_form:
            # every module in one launch (blockIdx.y = module): the pairwise module's 88 us and the hyper modules' 24-41 us
            # side by side instead of end to end
            arr = (_lib.N2EBwdGroup * len(S))()
            for i, c in enumerate(S):
                arr[i] = _lib.N2EBwdGroup(_p(c["xp"]), _p(c["pq"]), _p(c["H"]), _p(c["w2"]), _p(c["b2"]), _p(c["dedges"]),
                                          _p(c["dxp"]), _p(c["dpq"]), _p(grads[c["a1"].weight]), _p(grads[c["a1"].bias]),
                                          c["E"], int(c["sym"]))
            with torch.cuda.device(dev):
                check(load().gn_node2edge_bwd_grouped_f32(arr, len(S), S[0]["B"], N0, stream_handle()),
                      "gn_node2edge_bwd_grouped_f32")
        else:
            for c in S:
                with torch.cuda.device(dev):
                    check(load().gn_node2edge_bwd_f32(_p(c["xp"]), _p(c["pq"]), _p(c["H"]), _p(c["w2"]), _p(c["b2"]),
                                                      _p(c["dedges"]), _p(c["dxp"]), _p(c["dpq"]), _p(grads[c["a1"].weight]),
                                                      _p(grads[c["a1"].bias]), c["B"], c["N"], c["E"], int(c["sym"]),
                                                      stream_handle()),
                          "gn_node2edge_bwd_f32")
        stage(lambda c: gb.add(c["dpq"], c["tw"]["Wpq"], c["dxp"], beta=1.0))
        stage(lambda c: c.update(dx1=gb.add(c["dxp"], W(c["s1"]), new(c["B"] * c["N"], 256), mask=c["x1"])))

        def dx_stage(c):
            if c["dx"] is None:
                c["dx"] = gb.add(c["dx1"], W(c["s0"]), new(c["B"] * c["N"], D))
            else:
                gb.add(c["dx1"], W(c["s0"]), c["dx"], beta=1.0)
        stage(dx_stage)

        def n2e_weight_grads(c):
            K, pool = c["K"], c["pool"]

            def wgrad(lin, dY, X):
                grads[lin.weight] = gb.add(dY, X, pool.take(*lin.weight.shape), tA=True, accum=True,
                                           colsum=grads.setdefault(lin.bias, pool.take(*lin.bias.shape)))
            # MLP_distribution | MLP_factor: gradients of the side-by-side matrices, handed out as views
            gWd1, gbd1 = pool.take(_LGF_LD, 256), pool.take(_LGF_LD)
            gb.add(c["dlgf"], c["dh1"], gWd1, tA=True, accum=True, colsum=gbd1)
            grads[c["d"][1].weight], grads[c["f"][1].weight] = gWd1[:K, :128], gWd1[K:K + 1, 128:]
            grads[c["d"][1].bias], grads[c["f"][1].bias] = gbd1[:K], gbd1[K:K + 1]
            gWd0, gbd0 = pool.take(256, D), pool.take(256)
            gb.add(c["dd1"], c["z"], gWd0, tA=True, accum=True, colsum=gbd0)
            grads[c["d"][0].weight], grads[c["f"][0].weight] = gWd0[:128], gWd0[128:]
            grads[c["d"][0].bias], grads[c["f"][0].bias] = gbd0[:128], gbd0[128:]
            wgrad(c["i"][1], c["dz"], c["z1"])
            wgrad(c["i"][0], c["dz1"], c["edges"])
            ga0 = pool.take(32, 2 * D)                               # the (32,128) layout of attention layer 0
            gb.add(c["dpq"][:, :32], c["xp"], ga0[:, :D], tA=True, accum=True,
                   colsum=grads.setdefault(c["a0"].bias, pool.take(32)))
            gb.add(c["dpq"][:, 32:], c["xp"], ga0[:, D:], tA=True, accum=True)
            grads[c["a0"].weight] = ga0
            wgrad(c["s1"], c["dxp"], c["x1"])
            wgrad(c["s0"], c["dx1"], c["x2"])
        stage(n2e_weight_grads)
    return [c["dx"].view(c["B"], c["N"], D) for c in S]


def modules_backward(traces: Sequence[ModuleTrace], g_nfs: Sequence[Optional[Tensor]],
                     g_facs: Sequence[Optional[Tensor]]) -> Tuple[List[Tensor], Dict[nn.Parameter, Tensor]]:
    """Gradients of (node_feat, factors) of several modules (same nmp_layers) w.r.t. their h_states and
    parameters: the rounds walked backwards, each round grouped over the modules."""
    L = traces[0].mod.nmp_layers - 1
    grads: Dict[nn.Parameter, Tensor] = {}
    g_ys = list(g_nfs)
    for j in range(L, -1, -1):
        g_ds = list(g_facs) if j == 0 else [None] * len(traces)
        g_ys = round_backward(traces, j, g_ys, g_ds, grads)
    return g_ys, grads


class MSHGNNFunction(torch.autograd.Function):
    """Several modules on their inputs: forward = the grouped fused HIP path, backward = `modules_backward`.

    apply(mods, Hs, noises, n_params_per_module, h_0..h_{n-1}, *params) -> (nf_0, fac_0, nf_1, fac_1, ...)."""

    @staticmethod
    def forward(ctx, mods, Hs, noises, *tensors):
        from . import MS_HGNN_batch as M
        n = len(mods)
        hs, params = tensors[:n], tensors[n:]
        traces = [ModuleTrace(m, h.detach(), H) for m, h, H in zip(mods, hs, Hs)]
        with torch.no_grad(), M.training_call():     # a training step: packed-weight caches are not trusted
            res = M.run_message_passing(list(mods), [t.xs[0] for t in traces], list(Hs), list(noises), [None] * n,
                                        traces=traces)
        ctx.traces, ctx.params, ctx.n = traces, params, n
        # the backward reads the LIVE weights (and re-packs them): remember which versions the activations belong to
        ctx.param_key = M._param_key(params)
        out = []
        for nf, fac in res:
            out += [nf, fac]
        return tuple(out)

    @staticmethod
    def backward(ctx, *gs):
        from .MS_HGNN_batch import _param_key
        if _param_key(ctx.params) != ctx.param_key:
            raise RuntimeError("one of the parameters of an MS-HGNN module was modified in place between the forward and "
                               "this backward (e.g. an optimizer step before backward()): the saved activations belong "
                               "to the old weights.  torch's own autograd raises here too.")
        n = ctx.n
        g_nfs = [None if g is None else g.contiguous() for g in gs[0::2]]
        g_facs = [None if g is None else g.contiguous() for g in gs[1::2]]
        # a module none of whose outputs is used downstream contributes nothing
        live = [i for i in range(n) if g_nfs[i] is not None or g_facs[i] is not None]
        dhs: List[Optional[Tensor]] = [None] * n
        grads: Dict[nn.Parameter, Tensor] = {}
        if live:
            d, grads = modules_backward([ctx.traces[i] for i in live], [g_nfs[i] for i in live],
                                        [g_facs[i] for i in live])
            for i, dh in zip(live, d):
                dhs[i] = dh
        return (None, None, None) + tuple(dhs) + tuple(grads.get(p) for p in ctx.params)
