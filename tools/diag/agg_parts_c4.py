"""Diagnostic (not product): the typed-aggregation launch of config 4 (B=1024, N=50, bf16 twins) with subsets of its
groups — all five modules, the pairwise group alone (scene form / per-pair form), the four hyper groups alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from groupnet_amd import ops
from groupnet_amd.multiscale import MultiScaleHGNN
dev = torch.device("cuda")
torch.manual_seed(0)
B, N, SC = 1024, 50, [2, 4, 8, 16]
blk = MultiScaleHGNN(SC).to(dev).eval()
f = torch.randn(B, N, 64, device=dev).bfloat16()
_, Hs, _ = ops.affinity_topk(f, SC, want_corr=False)
mods = [blk.interaction, *blk.interaction_hyper]
aggs = [m.edge_aggregation_list[0] for m in mods]
pks = [a._packed() for a in aggs]
Ks = [a.edge_types for a in aggs]
efs = [torch.rand(B, ops.pair_count(N), Ks[0], device=dev)] + [torch.rand(B, H.shape[1], 10, device=dev) for H in Hs]
eos = ops.agg_gather_grouped([(f, H) for H in Hs])
hyper = [(eo, ef, pk, K) for eo, ef, pk, K in zip(eos, efs[1:], pks[1:], Ks[1:])]
def t(sub, reps=10):
    for _ in range(2): ops.agg_mlp_grouped(sub)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(6_000_000)
    a.record()
    for _ in range(reps): ops.agg_mlp_grouped(sub)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
scene = (ops.GatherSpec(f, None, True, node=True), efs[0], pks[0], Ks[0])
pairf = (ops.GatherSpec(f, None, True), efs[0], pks[0], Ks[0])
for name, sub in (("scene form + hyper x4", [scene] + hyper), ("scene form only", [scene]), ("hyper x4 only", hyper),
                  ("per-pair form + hyper x4", [pairf] + hyper), ("per-pair form only", [pairf])):
    print(f"{name:28s} {t(sub):7.1f} us")
