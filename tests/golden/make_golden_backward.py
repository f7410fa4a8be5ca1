#!/usr/bin/env python3
"""Gradient goldens for the backward, produced by the REFERENCE's own autograd.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_backward.py

Imports /root/reference/model/MS_HGNN_batch.py unmodified, loads the committed golden weights
(weights_*.npz) into reference modules, runs forward + `loss.backward()` on CPU with the loss
sum(node_feat * R1) + sum(factors * R2) for fixed random R1, R2, and stores inputs, the uniforms the
forward drew, R1/R2, dL/dh in full, a few parameter gradients in full, and for EVERY parameter the
triple (sum, sum of |.|, max |.|) of its gradient in float64.  Only data is written.
"""
import os
import sys
import warnings

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
warnings.filterwarnings("ignore")
from model import MS_HGNN_batch as ref  # noqa: E402

sys.path.insert(0, OUT)
from make_golden import CALLER_KW, Recorder  # noqa: E402

FULL = ("attention_mlp.0.layers.0.weight", "attention_mlp.0.layers.0.bias", "attention_mlp.0.layers.1.weight",
        "nmp_mlp_start.MLP_factor.layers.1.weight", "nmp_mlp_start.MLP_distribution.layers.1.bias",
        "nmp_mlp_start.init_MLP.layers.0.weight", "edge_aggregation_list.0.agg_mlp.0.layers.0.weight",
        "edge_aggregation_list.0.agg_mlp.5.layers.1.weight", "nmp_mlp_end.layers.1.weight",
        "node2edge_start_mlp.0.layers.0.weight")


def load(name):
    with np.load(os.path.join(OUT, f"weights_{name}.npz")) as z:
        return {k: torch.from_numpy(z[k].copy()) for k in z.files}


def record(prefix, rec, module, outputs_fn, h):
    h = h.clone().requires_grad_(True)
    with Recorder() as r:
        outs = outputs_fn(module, h)
    nf, fac = outs[0], outs[1]
    g = torch.Generator().manual_seed(len(prefix) * 1000 + nf.numel())
    R1, R2 = torch.randn(nf.shape, generator=g), torch.randn(fac.shape, generator=g)
    loss = (nf * R1).sum() + (fac * R2).sum()
    names = [n for n, _ in module.named_parameters()]
    grads = torch.autograd.grad(loss, [h] + [p for _, p in module.named_parameters()], allow_unused=True)
    rec[f"{prefix}_R1"], rec[f"{prefix}_R2"] = R1.numpy(), R2.numpy()
    rec[f"{prefix}_node_feat"], rec[f"{prefix}_factors"] = nf.detach().numpy(), fac.detach().numpy()
    rec[f"{prefix}_g_h"] = grads[0].numpy()
    for i, u in enumerate(r.draws):
        rec[f"{prefix}_U{i}"] = u.numpy()
    stats = []
    for n, gr in zip(names, grads[1:]):
        if gr is None:
            stats.append((n, np.array([np.nan, np.nan, np.nan])))
            continue
        g64 = gr.double()
        stats.append((n, np.array([float(g64.sum()), float(g64.abs().sum()), float(g64.abs().max())])))
        if n in FULL:
            rec[f"{prefix}_g/{n}"] = gr.numpy()
    rec[f"{prefix}_stat_names"] = np.array([n for n, _ in stats])
    rec[f"{prefix}_stats"] = np.stack([v for _, v in stats])
    if len(outs) > 2:
        rec[f"{prefix}_H"] = outs[2].numpy()


def run(name, nmp, B, N, scales, seed):
    sfx = "_nmp2" if nmp == 2 else ""
    pair = ref.MS_HGNN_oridinary(embedding_dim=16, nmp_layers=nmp, **CALLER_KW)
    hyper = ref.MS_HGNN_hyper(embedding_dim=64, nmp_layers=nmp, scale=2, **CALLER_KW)
    pair.load_state_dict(load("pairwise" + sfx))
    hyper.load_state_dict(load("hyper" + sfx))
    g = torch.Generator().manual_seed(seed)
    h = torch.randn(B, N, 64, generator=g)
    q = torch.nn.functional.normalize(h, p=2, dim=2)
    corr = torch.matmul(q, q.permute(0, 2, 1))
    rec = dict(h=h.numpy(), corr=corr.numpy(), scales=np.asarray(scales), nmp=np.int64(nmp))
    torch.manual_seed(seed + 1)
    record("pair", rec, pair, lambda m, x: m(x), h)
    for s in scales:
        hyper.scale = s
        record(f"hyper{s}", rec, hyper, lambda m, x: m(x, corr), h)
    np.savez_compressed(os.path.join(OUT, f"grad_{name}.npz"), **rec)
    print(name, len(rec), "arrays")


if __name__ == "__main__":
    run("n11_b5", 1, 5, 11, [3, 11], 501)
    run("n7_b3_nmp2", 2, 3, 7, [4], 502)
