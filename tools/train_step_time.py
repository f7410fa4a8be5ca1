import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root
from groupnet_amd.multiscale import MultiScaleHGNN
from groupnet_amd.graphs import GraphedTrainStep
dev = torch.device('cuda:0')
torch.manual_seed(0)
B, N = 512, 11
blk = MultiScaleHGNN([2, 5, 11]).to(dev).train()
f = torch.randn(B, N, 64, device=dev)
tgt = torch.randn(B, N, 320, device=dev)
step = GraphedTrainStep(blk, torch.optim.SGD(blk.parameters(), lr=1e-3), lambda o, H, t: ((o - t) ** 2).mean(), B, N, [tuple(tgt.shape)], seed=1)
step(f, tgt)
torch.cuda.synchronize()
t = time.perf_counter(); K = 50
for _ in range(K): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / K
print(f"graphed train step B={B}: {dt*1e3:.3f} ms -> {B/dt:.0f} scenes/s, loss {float(step.loss):.4f}")
