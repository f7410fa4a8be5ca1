"""Diagnostic: node2edge at a given shape, pairwise group alone / hyper groups alone / all (HIP event timing)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from groupnet_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50
SC = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2, 4, 8, 16]
dt = torch.bfloat16 if (len(sys.argv) > 4 and sys.argv[4] == "bf16") else torch.float32
dev = torch.device("cuda")
torch.manual_seed(0)
xp = torch.randn(B, N, 64, device=dev).to(dt)
pq = torch.randn(B, N, 64, device=dev).to(dt)
w2 = torch.randn(32, device=dev) * 0.2
b2 = torch.zeros(1, device=dev)
def mkH(k):
    E = 1 if k >= N else N
    H = torch.zeros(B, E, N, device=dev)
    idx = torch.rand(B, E, N, device=dev).argsort(-1)[..., :min(k, N)]
    H.scatter_(2, idx, 1.0)
    return H
pair = (xp, pq, None, w2, b2, True)
hyper = [(xp, pq, mkH(k), w2, b2) for k in SC]
def t(items, tag):
    for _ in range(3): ops.node2edge_grouped(items)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): ops.node2edge_grouped(items)
    b.record(); torch.cuda.synchronize()
    print(f"{tag:12s} {a.elapsed_time(b)/20*1e3:8.1f} us")
t([pair], "pair")
t(hyper, "hyper")
for k, h in zip(SC, hyper): t([h], f"hyper k={k}")
t([pair] + hyper, "all")
