"""Diagnostic (not product): the typed-aggregation launch at B=512, N=11 with subsets of its groups — all four modules
(as the forward launches it), the pairwise group alone, the three hyper groups alone — timed with events."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from groupnet_amd import ops
from groupnet_amd.multiscale import MultiScaleHGNN
dev = torch.device("cuda")
torch.manual_seed(0)
B, N, SC = 512, 11, [2, 5, 11]
blk = MultiScaleHGNN(SC).to(dev).eval()
f = torch.randn(B, N, 64, device=dev)
_, Hs, _ = ops.affinity_topk(f, SC, want_corr=False)
mods = [blk.interaction, *blk.interaction_hyper]
aggs = [m.edge_aggregation_list[0] for m in mods]
pks = [a._packed() for a in aggs]
Ks = [a.edge_types for a in aggs]
A = ops.node_linear(f, pks[0]["W1cat"], pks[0]["b1half"], Ks[0] * 128)
efs = [torch.rand(B, ops.pair_count(N), Ks[0], device=dev)] + [torch.rand(B, H.shape[1], 10, device=dev) for H in Hs]
NODE = os.environ.get("GN_NODE_FORM", "1") != "0"
items = [(ops.PairSpec(A, node=NODE), efs[0], pks[0], Ks[0])] + [(ops.GatherSpec(f, H, False), ef, pk, K) for H, ef, pk, K in zip(Hs, efs[1:], pks[1:], Ks[1:])]
print("node form" if NODE else "pair form")

def t(sub, reps=30):
    for _ in range(3): ops.agg_mlp_grouped(sub)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(3_000_000)
    a.record()
    for _ in range(reps): ops.agg_mlp_grouped(sub)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for name, sub in (("all", items), ("pair only", items[:1]), ("hyper x3 only", items[1:]), ("hyper s=2 only", items[1:2]), ("pair + hyper s=2", items[:2])):
    print(f"{name:18s} {t(sub):7.1f} us (incl. ~2-3 us launch gap)")
