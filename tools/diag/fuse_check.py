"""Diagnostic (not product): fused closing stage (GN_FUSE_CLOSING=1) against the two launches (=0) on the multiscale block."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from groupnet_amd.multiscale import MultiScaleHGNN
from groupnet_amd import MS_HGNN_batch as M, ops
dev = torch.device("cuda")
calls = {"0": 0, "1": 0}
real = ops.mlp2_grouped
def spy(*a, **k):
    calls[os.environ["GN_FUSE_CLOSING"]] += 1
    return real(*a, **k)
ops.mlp2_grouped = spy
for (B, N, SC) in ((3, 11, [2, 5, 11]), (37, 11, [2, 5, 11]), (512, 11, [2, 5, 11]), (5, 2, [2]), (7, 16, [2, 4]), (4, 1, [1]), (9, 5, [2, 5]), (64, 13, [3, 13])):
    torch.manual_seed(B * 100 + N)
    blk = MultiScaleHGNN(SC).to(dev).eval()
    f = torch.randn(B, N, 64, device=dev)
    outs = {}
    for mode in ("0", "1"):
        os.environ["GN_FUSE_CLOSING"] = mode
        M.set_noise_mode("device", 1234)
        with torch.no_grad():
            outs[mode] = blk(f)
    torch.cuda.synchronize()
    a, b = outs["0"][0], outs["1"][0]
    print(f"B={B} N={N} scales={SC}: max|diff| {float((a - b).abs().max()):.2e} equal={torch.equal(a, b)} scale {float(a.abs().max()):.2e}  mlp2 launches {calls}")
