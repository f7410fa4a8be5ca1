"""Backward of one MS-HGNN module assembled from HIP building blocks.

`train_hyper_nba.py:116` back-propagates through `MS_HGNN_oridinary` / `MS_HGNN_hyper`.  The forward
of the drop-in modules is the fused matrix-core path; when a gradient is needed the module goes through
`MSHGNNFunction`: forward = that same fused path, backward = this file — the chain rule written out
stage by stage (model/MS_HGNN_batch.py:41-53, 116-141, 247-268, 357-370), every tensor operation a kernel
of libgroupnet_hip.so (`gn_gemm_f32`, `gn_colsum_f32`, `gn_rowscale_f32`, `gn_rowdot_f32`,
`gn_gumbel_bwd_f32`, `gn_gumbel_ef_f32`, `gn_node2edge_bwd_f32`, `gn_axpby2d_f32` and the forward
gather / scatter / node2edge / typed-MLP kernels).  Hidden activations are not kept by the fused forward,
so the backward first re-computes them layer by layer with the generic GEMM.

Scope: `nmp_layers == 1` (every caller of the reference), explicit incidence H — the pairwise graph is
materialised as its (B, N*N, N) incidence (weights 1, self-loops 2) for the backward, which bounds the
trainable pairwise module to moderate N.  Correctness first; this path is not tuned.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from ._lib import _P, check, load, stream_handle

Tensor = torch.Tensor
_TAU = 0.5


# ---- thin faces of the generic kernels -------------------------------------------------------------
def _p(t: Optional[Tensor]):
    return _P(0 if t is None else t.data_ptr())


def gemm(A: Tensor, Bm: Tensor, transA=False, transB=False, bias=None, mask=None, relu=False, alpha=1.0,
         out: Optional[Tensor] = None, beta=0.0) -> Tensor:
    """out (M,N) = beta*out + alpha*op(A) op(B) (+bias) (relu) (zeroed where mask <= 0); 2-D row-major views."""
    assert A.dim() == 2 and Bm.dim() == 2 and A.stride(1) == 1 and Bm.stride(1) == 1
    M, K = (A.shape[1], A.shape[0]) if transA else (A.shape[0], A.shape[1])
    K2, N = (Bm.shape[1], Bm.shape[0]) if transB else (Bm.shape[0], Bm.shape[1])
    if K != K2:
        raise ValueError(f"gemm: inner dimensions {K} vs {K2}")
    if out is None:
        out = torch.empty((M, N), dtype=A.dtype, device=A.device)
        beta = 0.0
    assert out.shape == (M, N) and out.stride(1) == 1
    if mask is not None:
        assert mask.shape == (M, N) and mask.stride(1) == 1
    with torch.cuda.device(A.device):
        check(load().gn_gemm_f32(_p(A), _p(Bm), _p(out), M, N, K, A.stride(0), Bm.stride(0), out.stride(0), int(transA),
                                 int(transB), _p(bias), _p(mask), 0 if mask is None else mask.stride(0), int(relu),
                                 float(alpha), float(beta), stream_handle()), "gn_gemm_f32")
    return out


def colsum(X: Tensor) -> Tensor:
    out = torch.zeros(X.shape[1], dtype=X.dtype, device=X.device)
    with torch.cuda.device(X.device):
        check(load().gn_colsum_f32(_p(X), _p(out), X.shape[0], X.shape[1], X.stride(0), stream_handle()), "gn_colsum_f32")
    return out


def rowscale(src: Tensor, s: Tensor, off: int) -> Tensor:
    """dst[r,:] = s[r, off] * src[r,:]"""
    dst = torch.empty_like(src)
    with torch.cuda.device(src.device):
        check(load().gn_rowscale_f32(_p(dst), _p(src), _p(s), src.shape[0], src.shape[1], s.stride(0), off,
                                     stream_handle()), "gn_rowscale_f32")
    return dst


def rowdot_into(a: Tensor, b: Tensor, out: Tensor, off: int) -> None:
    """out[r, off] = <a[r], b[r]>"""
    with torch.cuda.device(a.device):
        check(load().gn_rowdot_f32(_p(a), _p(b), _p(out), a.shape[0], a.shape[1], out.stride(0), off, stream_handle()),
              "gn_rowdot_f32")


def axpby(out: Tensor, a: Tensor, alpha=1.0, beta=0.0) -> Tensor:
    """out = alpha*a + beta*out on 2-D (possibly column-sliced) views."""
    assert out.shape == a.shape and out.stride(1) == 1 and a.stride(1) == 1
    with torch.cuda.device(a.device):
        check(load().gn_axpby2d_f32(_p(out), out.stride(0), _p(a), a.stride(0), a.shape[0], a.shape[1], float(alpha),
                                    float(beta), stream_handle()), "gn_axpby2d_f32")
    return out


def _lin(X: Tensor, layer: nn.Linear, relu=False) -> Tensor:
    return gemm(X, layer.weight.detach(), transB=True, bias=layer.bias.detach(), relu=relu)


def pairwise_incidence(B: int, N: int, device, dtype=torch.float32) -> Tensor:
    """The (B, N*N, N) incidence rel_rec + rel_send of the pairwise graph (model/MS_HGNN_batch.py:118,
    143-160): edge e = i*N + j has weight 1 on i and on j, 2 when i == j.  A constant; only the backward
    materialises it."""
    e = torch.arange(N * N, device=device)
    H = torch.zeros(N * N, N, dtype=dtype, device=device)
    H[e, e % N] += 1
    H[e, e // N] += 1
    return H[None].expand(B, -1, -1).contiguous()


# ---- the backward of one module ----------------------------------------------------------------------
def module_backward(mod, h: Tensor, H: Tensor, dist: Tensor, g_nf: Optional[Tensor], g_dist: Optional[Tensor]
                    ) -> Tuple[Tensor, Dict[nn.Parameter, Tensor]]:
    """Gradients of (node_feat, factors) of one module w.r.t. h_states and its parameters.

    h (B,N,64); H (B,E,N) explicit; dist = the `factors` the forward returned (B,E,K) (it carries the
    Gumbel noise, which is a constant of the backward); g_nf (B,N,bottleneck) / g_dist (B,E,K) the incoming
    gradients (either may be None)."""
    if mod.nmp_layers != 1:
        raise NotImplementedError("backward is built for nmp_layers == 1")
    B, N, D = h.shape
    E, K = H.shape[1], mod.edge_types
    dev = h.device
    grads: Dict[nn.Parameter, Tensor] = {}
    h2 = h.reshape(B * N, D)
    s0, s1 = mod.node2edge_start_mlp[0].layers
    a0, a1 = mod.attention_mlp[0].layers
    st = mod.nmp_mlp_start
    i0, i1 = st.init_MLP.layers
    d0, d1 = st.MLP_distribution.layers
    f0, f1 = st.MLP_factor.layers
    agg_mod = mod.edge_aggregation_list[0]
    e0, e1 = mod.nmp_mlp_end.layers
    with torch.no_grad():
        # ---------------- re-computation of what the fused forward did not keep ----------------
        x1 = _lin(h2, s0, relu=True)                                   # (BN,256)
        xp = _lin(x1, s1)                                              # (BN,64)
        Wpq = torch.cat((a0.weight[:, :D], a0.weight[:, D:]), 0).detach().contiguous()      # (64,64): [P ; Qn]
        bpq = torch.cat((a0.bias, torch.zeros_like(a0.bias)), 0).detach().contiguous()
        pq = gemm(xp, Wpq, transB=True, bias=bpq)
        w2 = a1.weight.detach()[0].contiguous()
        b2 = float(a1.bias.detach()[0].item())
        edges = ops.node2edge(xp.view(B, N, D), pq.view(B, N, D), H, w2, b2)                 # (B,E,64)
        edges2 = edges.view(B * E, D)
        z1 = _lin(edges2, i0, relu=True)                               # (BE,128)
        z = _lin(z1, i1)                                               # (BE,64)
        Wd0 = torch.cat((d0.weight, f0.weight), 0).detach().contiguous()                     # (256,64)
        bd0 = torch.cat((d0.bias, f0.bias), 0).detach().contiguous()
        dh1 = gemm(z, Wd0, transB=True, bias=bd0, relu=True)           # (BE,256)
        Wd1 = torch.zeros(32, 256, dtype=h.dtype, device=dev)
        Wd1[:K, :128] = d1.weight.detach()
        Wd1[K, 128:] = f1.weight.detach()[0]
        bd1 = torch.zeros(32, dtype=h.dtype, device=dev)
        bd1[:K] = d1.bias.detach()
        bd1[K] = f1.bias.detach()[0]
        lgf = gemm(dh1, Wd1, transB=True, bias=bd1)                    # (BE,32): K logits, then the factor pre-activation
        dist2 = dist.reshape(B * E, K).contiguous()
        ef = torch.empty_like(dist2)
        with torch.cuda.device(dev):
            check(load().gn_gumbel_ef_f32(_p(dist2), _p(lgf), _p(ef), B * E, K, 32, stream_handle()), "gn_gumbel_ef_f32")
        eo = ops.agg_gather(h, H)                                      # (B,E,64)
        eo2 = eo.view(B * E, D)
        l0s = [m.layers[0] for m in agg_mod.agg_mlp]
        l1s = [m.layers[1] for m in agg_mod.agg_mlp]
        hks = [_lin(eo2, l, relu=True) for l in l0s]                   # K x (BE,128)
        mks = [_lin(hk, l) for hk, l in zip(hks, l1s)]                 # K x (BE,64)
        feat = ops.agg_mlp(eo, ef.view(B, E, K), agg_mod._packed(), K)                       # (B,E,64)
        agg = ops.agg_scatter(feat, H, h)                              # (B,N,128) = cat(H^T feat, ori)/N
        agg2 = agg.view(B * N, 2 * D)
        y1 = _lin(agg2, e0, relu=True)                                 # (BN,128)

        # ---------------- backward ----------------
        dh = torch.zeros_like(h2)
        def_ = torch.zeros((B * E, K), dtype=h.dtype, device=dev)
        if g_nf is not None:
            g = g_nf.reshape(B * N, -1).contiguous()
            dy1 = gemm(g, e1.weight.detach(), mask=y1)                 # (BN,128)
            grads[e1.weight] = gemm(g, y1, transA=True)
            grads[e1.bias] = colsum(g)
            daggN = gemm(dy1, e0.weight.detach(), alpha=1.0 / N)       # d(agg) / N  (BN,128)
            grads[e0.weight] = gemm(dy1, agg2, transA=True)
            grads[e0.bias] = colsum(dy1)
            axpby(dh, daggN[:, D:], 1.0, 1.0)                          # ori half of the concat
            da = torch.empty((B * N, D), dtype=h.dtype, device=dev)
            axpby(da, daggN[:, :D])
            dfeat = ops.agg_gather(da.view(B, N, D), H).view(B * E, D)  # adjoint of H^T feat
            deo = torch.zeros((B * E, D), dtype=h.dtype, device=dev)
            for k in range(K):
                rowdot_into(dfeat, mks[k], def_, k)                    # d ef_k = <dfeat, M_k(eo)>
                dmk = rowscale(dfeat, ef, k)                           # ef_k * dfeat
                dhk = gemm(dmk, l1s[k].weight.detach(), mask=hks[k])   # (BE,128)
                gemm(dhk, l0s[k].weight.detach(), out=deo, beta=1.0)
                grads[l1s[k].weight] = gemm(dmk, hks[k], transA=True)
                grads[l1s[k].bias] = colsum(dmk)
                grads[l0s[k].weight] = gemm(dhk, eo2, transA=True)
                grads[l0s[k].bias] = colsum(dhk)
            # eo = H ori  ->  d ori += H^T d eo   (the scatter kernel with divisor 1; its ori half is unused)
            sc = ops.agg_scatter(deo.view(B, E, D), H, torch.zeros_like(h), divisor=1.0).view(B * N, 2 * D)
            axpby(dh, sc[:, :D], 1.0, 1.0)
        if g_nf is not None or g_dist is not None:
            gd = None if g_dist is None else g_dist.reshape(B * E, K).contiguous()
            dlgf = torch.empty((B * E, 32), dtype=h.dtype, device=dev)
            with torch.cuda.device(dev):
                check(load().gn_gumbel_bwd_f32(_p(dist2), _p(lgf), _p(def_), _p(gd), _p(dlgf), B * E, K, 32, _TAU,
                                               stream_handle()), "gn_gumbel_bwd_f32")
            dd1 = gemm(dlgf, Wd1, mask=dh1)                            # (BE,256)
            gWd1 = gemm(dlgf, dh1, transA=True)                        # (32,256)
            gbd1 = colsum(dlgf)
            grads[d1.weight] = gWd1[:K, :128].contiguous()
            grads[f1.weight] = gWd1[K:K + 1, 128:].contiguous()
            grads[d1.bias] = gbd1[:K].contiguous()
            grads[f1.bias] = gbd1[K:K + 1].contiguous()
            dz = gemm(dd1, Wd0)                                        # (BE,64)
            gWd0 = gemm(dd1, z, transA=True)                           # (256,64)
            gbd0 = colsum(dd1)
            grads[d0.weight], grads[f0.weight] = gWd0[:128].contiguous(), gWd0[128:].contiguous()
            grads[d0.bias], grads[f0.bias] = gbd0[:128].contiguous(), gbd0[128:].contiguous()
            dz1 = gemm(dz, i1.weight.detach(), mask=z1)                # (BE,128)
            grads[i1.weight] = gemm(dz, z1, transA=True)
            grads[i1.bias] = colsum(dz)
            dedges = gemm(dz1, i0.weight.detach())                     # (BE,64)
            grads[i0.weight] = gemm(dz1, edges2, transA=True)
            grads[i0.bias] = colsum(dz1)
            # node -> edge pooling
            dxp = torch.zeros((B * N, D), dtype=h.dtype, device=dev)
            dpq = torch.zeros((B * N, D), dtype=h.dtype, device=dev)
            dw2 = torch.zeros(32, dtype=h.dtype, device=dev)
            db2 = torch.zeros(1, dtype=h.dtype, device=dev)
            with torch.cuda.device(dev):
                check(load().gn_node2edge_bwd_f32(_p(xp), _p(pq), _p(H), _p(w2), b2, _p(dedges), _p(dxp), _p(dpq), _p(dw2),
                                                  _p(db2), B, N, E, stream_handle()), "gn_node2edge_bwd_f32")
            grads[a1.weight] = dw2.view(1, 32)
            grads[a1.bias] = db2
            gemm(dpq, Wpq, out=dxp, beta=1.0)                          # pq = Wpq x' + bpq
            gWpq = gemm(dpq, xp, transA=True)                          # (64,64)
            gbpq = colsum(dpq)
            grads[a0.weight] = torch.cat((gWpq[:32], gWpq[32:]), dim=1).contiguous()   # back to the (32,128) layout
            grads[a0.bias] = gbpq[:32].contiguous()
            dx1 = gemm(dxp, s1.weight.detach(), mask=x1)               # (BN,256)
            grads[s1.weight] = gemm(dxp, x1, transA=True)
            grads[s1.bias] = colsum(dxp)
            gemm(dx1, s0.weight.detach(), out=dh, beta=1.0)
            grads[s0.weight] = gemm(dx1, h2, transA=True)
            grads[s0.bias] = colsum(dx1)
    return dh.view(B, N, D), grads


class MSHGNNFunction(torch.autograd.Function):
    """forward = the fused HIP path of the module; backward = `module_backward`."""

    @staticmethod
    def forward(ctx, mod, H_or_none, noise_u, h, *params):
        from .MS_HGNN_batch import run_message_passing
        with torch.no_grad():
            (node_feat, factors), = run_message_passing([mod], [h.detach()], [H_or_none], [noise_u], [None])
        ctx.mod = mod
        ctx.pairwise = H_or_none is None
        ctx.params = params
        ctx.save_for_backward(h.detach(), factors, *( [] if H_or_none is None else [H_or_none] ))
        return node_feat, factors

    @staticmethod
    def backward(ctx, g_nf, g_fac):
        saved = ctx.saved_tensors
        h, factors = saved[0], saved[1]
        B, N = h.shape[0], h.shape[1]
        H = pairwise_incidence(B, N, h.device, h.dtype) if ctx.pairwise else saved[2]
        g_nf = None if g_nf is None else g_nf.contiguous()
        g_fac = None if g_fac is None else g_fac.contiguous()
        dh, grads = module_backward(ctx.mod, h, H, factors, g_nf, g_fac)
        return (None, None, None, dh) + tuple(grads.get(p) for p in ctx.params)
