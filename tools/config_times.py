import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import groupnet_amd as G
from groupnet_amd.multiscale import MultiScaleHGNN
from groupnet_amd.graphs import GraphedMultiScale
from oracle import ms_hgnn_oracle as O   # checker only (this is a measurement tool, not the product path)
dev = torch.device("cuda")
def run(name, B, N, scales, check_B=0):
    torch.manual_seed(0)
    blk = MultiScaleHGNN(scales)
    sp = {k: v.detach().clone() for k, v in blk.interaction.state_dict().items()}
    shs = [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in blk.interaction_hyper]
    blk.to(dev).eval()
    f = torch.randn(B, N, 64, device=dev)
    with torch.no_grad():
        if check_B:
            fc = f[:check_B].cpu()
            noise = [[torch.rand(s)] for s in blk.noise_shapes(check_B, N)]
            ref, Href, _ = O.ms_hgnn_multiscale_forward(sp, shs, scales, fc, noise[0], noise[1:], decomposed=True)
            out, H = blk(f[:check_B].contiguous(), noise_u=[[u.to(dev) for u in n] for n in noise])
            print(name, "parity vs oracle at B=%d: H equal %s, max err %.2e" % (check_B, torch.equal(H.cpu(), Href), (out.cpu() - ref).abs().max().item()), flush=True)
        g = GraphedMultiScale(blk, B, N, seed=1)
        g.f_in.copy_(f)
        for _ in range(3): g()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 20
        for _ in range(K): g()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
        print(f"{name}: B={B} N={N} scales={scales}: {dt*1e3:.3f} ms/forward, {B/dt:,.0f} scenes/s (single stream)", flush=True)
run("C2", 512, 11, [2, 5, 11], check_B=16)
run("C4-fp32", 1024, 50, [2, 4, 8, 16], check_B=2)
run("C5-hyper+pairwise", 32, 256, [2, 8, 32, 128])
run("C3-per-GPU-share", 512, 11, [2, 5, 11])


def train(name, B, N, scales):
    from groupnet_amd.graphs import GraphedTrainStep
    torch.manual_seed(1)
    blk = MultiScaleHGNN(scales).to(dev).train()
    f = torch.randn(B, N, 64, device=dev)
    tgt = torch.randn(B, N, blk.out_features, device=dev)
    step = GraphedTrainStep(blk, torch.optim.SGD(blk.parameters(), lr=1e-3), lambda o, H, t: ((o - t) ** 2).mean(), B, N,
                            [tuple(tgt.shape)], seed=1)
    step(f, tgt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 10
    for _ in range(K): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print(f"{name}: graphed training step B={B} N={N} scales={scales}: {dt*1e3:.3f} ms, {B/dt:,.0f} scenes/s", flush=True)


train("C2", 512, 11, [2, 5, 11])
train("C4-fp32", 1024, 50, [2, 4, 8, 16])
