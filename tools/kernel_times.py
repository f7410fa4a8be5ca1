"""Per-kernel times of one forward (B=512, N=11, scales {2,5,11}) with HIP events, single stream, eager launches
with the host kept ahead; prints the probe summary of bench.py without the rest of the bench."""
import sys, os, json, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import groupnet_amd as G
from groupnet_amd import ops
from groupnet_amd.multiscale import MultiScaleHGNN
from groupnet_amd.graphs import GraphedMultiScale

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 11
scales = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2, 5, 11]
dt = torch.bfloat16 if (len(sys.argv) > 4 and sys.argv[4] == "bf16") else torch.float32
dev = torch.device("cuda")
torch.manual_seed(0)
blk = MultiScaleHGNN(scales).to(dev).eval()
f = torch.randn(B, N, 64, device=dev).to(dt)
with torch.no_grad():
    G.set_noise_mode("device", seed=99)
    for _ in range(3):
        blk(f)
    probe = bench.Probe()
    ops.launch_probe = probe
    torch.cuda.synchronize()
    for _ in range(30):
        torch.cuda._sleep(3_000_000)
        blk(f)
    ops.launch_probe = None
    torch.cuda.synchronize()
    ov = bench.empty_bracket_ms()
    tot = 0.0
    for k, (ms, fl, n, _ref) in probe.summary(ov).items():
        print(f"{k:28s} {ms*1e3:8.2f} us  {fl/(ms*1e-3)/1e12:8.1f} TFLOP/s")
        tot += ms
    print(f"sum of probed kernels {tot*1e3:.1f} us")
    if True:
        g = GraphedMultiScale(blk, B, N, seed=5, dtype=dt)
        g.f_in.copy_(f)
        for _ in range(5):
            g()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200):
            g()
        b.record()
        torch.cuda.synchronize()
        print(f"graph replay, single stream: {a.elapsed_time(b)/200*1e3:.1f} us per forward = {B/(a.elapsed_time(b)/200*1e-3):.0f} scenes/s")
