#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Run once, in the build container (the only place /root/reference exists):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports /root/reference/model/MS_HGNN_batch.py unmodified, seeds the
generator, builds one MS_HGNN_oridinary and one MS_HGNN_hyper with the
constructor arguments the reference's callers use
(model/GroupNet_nba.py:209-248), runs their forward on seeded inputs and
stores inputs, uniforms, outputs and a few intermediates as .npz.  Only DATA is
written — no reference source text.  The weights of MS_HGNN_hyper do not
depend on `scale`, so one state_dict serves every scale.

Files written:
  weights_pairwise.npz / weights_hyper.npz   state_dicts (reference key names)
  weights_*_nmp2.npz                         same for nmp_layers=2
  case_<name>.npz                            h, corr, U*, and the reference outputs
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
import warnings

warnings.filterwarnings("ignore")
from model import MS_HGNN_batch as ref  # noqa: E402

CALLER_KW = dict(h_dim=64, mlp_dim=64, bottleneck_dim=64, batch_norm=0)  # GroupNet_nba.py:209-248


def sd_to_np(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def build_modules(nmp_layers):
    torch.manual_seed(20240 + nmp_layers)
    pair = ref.MS_HGNN_oridinary(embedding_dim=16, nmp_layers=nmp_layers, **CALLER_KW).eval()
    hyper = ref.MS_HGNN_hyper(embedding_dim=64, nmp_layers=nmp_layers, scale=2, **CALLER_KW).eval()
    # default nn.Linear init gives tiny attention logits; scale a few weights up so that the
    # softmax / gumbel / sigmoid paths are exercised away from their flat regions.
    with torch.no_grad():
        for m in (pair, hyper):
            for name, p in m.named_parameters():
                if "attention_mlp" in name or "MLP_distribution" in name or "MLP_factor" in name:
                    p.mul_(3.0)
    return pair, hyper


class Recorder:
    """Wraps torch.rand to record the uniforms a forward draws (MS_HGNN_batch.py:454)."""

    def __init__(self):
        self.draws = []
        self._orig = torch.rand

    def __enter__(self):
        def rec(*a, **k):
            u = self._orig(*a, **k)
            self.draws.append(u.clone())
            return u
        torch.rand = rec
        return self

    def __exit__(self, *exc):
        torch.rand = self._orig


def trace_intermediates(module, h, H, draws, hyper):
    """Re-run the stages of the reference by calling ITS methods, to save intermediates."""
    it = iter(draws)
    orig = torch.rand
    torch.rand = lambda *a, **k: next(it)
    try:
        if hyper:
            edges = module.node2edge(h, H, 0)
        else:
            rel_rec, rel_send = module.init_adj(h.shape[1], h.shape[0])
            edges = module.node2edge(h, rel_rec, rel_send, 0)
        xp = module.node2edge_start_mlp[0](h)
        edge_feat, dist = module.nmp_mlp_start(edges)
    finally:
        torch.rand = orig
    out = dict(xp=xp, edges=edges, edge_feat=edge_feat)
    if module.nmp_layers == 1:
        Hm = H if hyper else (rel_rec + rel_send)
        eo = torch.matmul(Hm, h)
        agg = module.edge_aggregation_list[0](edge_feat, Hm, h)
        out.update(eo=eo, agg=agg / agg.size(1), feat_scattered=agg[..., :h.shape[-1]])
    return out


def nba_features(B_rep=1):
    """h_states-like features from the 10-scene NBA sample shipped with the reference
    (datasets/nba/test_nba.npy, (10,15,11,2)); the front-end is out of scope here, so
    the 5 past steps x (pos, vel) are embedded with a fixed seeded projection to 64-d."""
    a = np.load(os.path.join(REF, "datasets/nba/test_nba.npy")).astype(np.float32)
    a = a / np.float32(94 / 28)                           # data/dataloader_nba.py:36
    a = np.transpose(a, (0, 2, 1, 3))                     # (S,N,T,2)   dataloader_nba.py:49
    past = a[:, :, :5, :]
    vel = np.concatenate([np.zeros_like(past[:, :, :1]), np.diff(past, axis=2)], axis=2)
    x = np.concatenate([past, vel], axis=-1).reshape(a.shape[0], a.shape[1], 20)
    g = torch.Generator().manual_seed(7)
    W = torch.randn(20, 64, generator=g) * 0.05
    h = torch.from_numpy(x) @ W
    return h.repeat(B_rep, 1, 1).contiguous()


def run_case(name, pair, hyper, h, scales, seed, with_pair=True):
    h = h.float().contiguous()
    B, N = h.shape[0], h.shape[1]
    q = torch.nn.functional.normalize(h, p=2, dim=2)       # GroupNet_nba.py:284
    corr = torch.matmul(q, q.permute(0, 2, 1))             # GroupNet_nba.py:285
    rec = dict(h=h.numpy(), corr=corr.numpy(), scales=np.asarray(scales, dtype=np.int64), seed=np.int64(seed))
    torch.manual_seed(seed)
    with torch.no_grad():
        if with_pair:
            with Recorder() as r:
                nf, fac = pair(h)
            rec.update(pair_node_feat=nf.numpy(), pair_factors=fac.numpy())
            for i, u in enumerate(r.draws):
                rec[f"pair_U{i}"] = u.numpy()
            for k, v in trace_intermediates(pair, h, None, r.draws, hyper=False).items():
                rec[f"pair_{k}"] = v.numpy()
        for s in scales:
            hyper.scale = s
            with Recorder() as r:
                nf, fac, H = hyper(h, corr)
            rec.update({f"hyper{s}_node_feat": nf.numpy(), f"hyper{s}_factor": fac.numpy(),
                        f"hyper{s}_H": H.numpy()})
            for i, u in enumerate(r.draws):
                rec[f"hyper{s}_U{i}"] = u.numpy()
            for k, v in trace_intermediates(hyper, h, H, r.draws, hyper=True).items():
                rec[f"hyper{s}_{k}"] = v.numpy()
            # tie check: top-k must be well separated for the fixture to be tie-free
            if s != N:
                k = max(s, 1)
                vals, _ = torch.topk(corr, k=k + 1, dim=2)
                gap = (vals[..., k - 1] - vals[..., k]).min().item()
                rec[f"hyper{s}_min_gap"] = np.float32(gap)
                assert gap > 0, (name, s, gap)
    np.savez_compressed(os.path.join(OUT, f"case_{name}.npz"), **rec)
    print(name, "B", B, "N", N, "scales", scales, "keys", len(rec))


def main():
    pair1, hyper1 = build_modules(1)
    np.savez_compressed(os.path.join(OUT, "weights_pairwise.npz"), **sd_to_np(pair1.state_dict()))
    np.savez_compressed(os.path.join(OUT, "weights_hyper.npz"), **sd_to_np(hyper1.state_dict()))
    pair2, hyper2 = build_modules(2)
    np.savez_compressed(os.path.join(OUT, "weights_pairwise_nmp2.npz"), **sd_to_np(pair2.state_dict()))
    np.savez_compressed(os.path.join(OUT, "weights_hyper_nmp2.npz"), **sd_to_np(hyper2.state_dict()))

    # (1) real NBA sample scenes, B=10, N=11, the caller's scales + the BASELINE scales
    run_case("nba_b10", pair1, hyper1, nba_features(), [2, 5, 11], seed=101)
    # (2) synthetic N=11, ragged batch (not a multiple of any tile)
    g = torch.Generator().manual_seed(11)
    run_case("syn_n11_b37", pair1, hyper1, torch.randn(37, 11, 64, generator=g), [1, 2, 3, 5, 8, 11], seed=102)
    # (3) N=50 (SDD-like), small B — hyper for the C4 scales, pairwise at B=2 (E=2500)
    g = torch.Generator().manual_seed(12)
    run_case("syn_n50_b3", pair1, hyper1, torch.randn(3, 50, 64, generator=g), [2, 4, 8, 16, 50], seed=103)
    # (4) tiny edge cases: B=1, N=2 and N=1-like scale clamps
    g = torch.Generator().manual_seed(13)
    run_case("syn_n2_b1", pair1, hyper1, torch.randn(1, 2, 64, generator=g), [1, 2], seed=104)
    g = torch.Generator().manual_seed(14)
    run_case("syn_n5_b4", pair1, hyper1, torch.randn(4, 5, 64, generator=g), [0, 3, 5], seed=105)
    # (5) nmp_layers = 2 (the loop of MS_HGNN_batch.py:186-194,432-440)
    g = torch.Generator().manual_seed(15)
    run_case("syn_n11_b6_nmp2", pair2, hyper2, torch.randn(6, 11, 64, generator=g), [3, 11], seed=106)
    # (6) N=70 > 64 (more nodes than lanes), hyper only (pairwise E=4900 materialises 2.7 GB/scene-batch)
    g = torch.Generator().manual_seed(16)
    run_case("syn_n70_b2", pair1, hyper1, torch.randn(2, 70, 64, generator=g), [2, 8, 32], seed=107, with_pair=False)


if __name__ == "__main__":
    main()
