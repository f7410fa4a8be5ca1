"""Diagnostic (not product): the closing-MLP launch at B=512, N=11 with its inputs formed by the fused scatter (as the
forward launches it) and from per-node aggregates (E = 0) for the pairwise group / for every group."""
import os, sys
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
pks = [m._packed_mlp2(m.nmp_mlp_end) for m in mods]
feat_pair = torch.randn(B, ops.pair_count(N), 64, device=dev)
feats = [torch.randn(B, H.shape[1], 64, device=dev) for H in Hs]
node = [torch.randn(B, N, 64, device=dev) for _ in range(4)]
out = torch.empty(B, N, 320, device=dev)
cols = [out[..., 64 * (i + 1):64 * (i + 2)] for i in range(4)]
scat = [ops.ScatterSpec(feat_pair, None, f, True)] + [ops.ScatterSpec(ft, H, f, False) for ft, H in zip(feats, Hs)]
nodes = [ops.NodeAggSpec(n, f) for n in node]
def t(items, reps=30):
    for _ in range(3): ops.mlp2_grouped(items)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(3_000_000)
    a.record()
    for _ in range(reps): ops.mlp2_grouped(items)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for name, src in (("all fused scatter", scat), ("pair from node aggregate", [nodes[0]] + scat[1:]),
                  ("pair + two hyper from node aggregates", nodes[:3] + scat[3:]), ("all from node aggregates", nodes)):
    print(f"{name:40s} {t([(s, pk, c) for s, pk, c in zip(src, pks, cols)]):6.1f} us (incl. ~2-3 us launch gap)")
