import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groupnet_amd import ops
from groupnet_amd.multiscale import MultiScaleHGNN
torch.manual_seed(0)
dev = torch.device("cuda")
blk = MultiScaleHGNN([2, 5, 11]).to(dev).eval()
B, N = 512, 11
def t(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(3_000_000); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
h = torch.randn(B, N, 64, device=dev)
corr, Hs, _ = ops.affinity_topk(h, [2, 5, 11])
def items(which):
    out = []
    if "p" in which:
        a = blk.interaction.edge_aggregation_list[0]; pk = a._packed()
        A = ops.node_linear(h, pk["W1cat"], pk["b1half"], 768)
        out.append((ops.PairSpec(A), torch.rand(B, 66, 6, device=dev), pk, 6))
    for i, m in enumerate(blk.interaction_hyper):
        if str(i) in which:
            a = m.edge_aggregation_list[0]
            out.append((ops.GatherSpec(h, Hs[i], False), torch.rand(B, Hs[i].shape[1], 10, device=dev), a._packed(), 10))
    return out
for which in ["p", "012", "p012"]:
    it = items(which)
    for wp in ["1", "2", "4"]:
        for wh in ["1", "2", "4"]:
            if which == "p" and wh != "1": continue
            if which == "012" and wp != "1": continue
            os.environ["GN_AGG_WPR_PAIR"] = wp; os.environ["GN_AGG_WPR"] = wh
            print(f"groups={which:5s} wpr_pair={wp} wpr_hyper={wh} {t(lambda: ops.agg_mlp_grouped(it)):8.1f} us", flush=True)
