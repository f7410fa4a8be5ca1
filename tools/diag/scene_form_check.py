"""Diagnostic (not product): scene form of the node form (bf16 twins, N <= 64) against the per-pair twin (GN_NODE_FORM=0)
and against the fp32 path on the up-cast inputs: max |difference| relative to the fp32 result's scale."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from groupnet_amd.multiscale import MultiScaleHGNN
from groupnet_amd import MS_HGNN_batch as M
dev = torch.device("cuda")
for (B, N, SC) in ((3, 50, [2, 4, 8, 16]), (5, 33, [2, 8]), (4, 64, [2, 16]), (6, 20, [2, 5]), (64, 50, [2, 4, 8, 16]), (7, 11, [2, 5, 11]), (2, 1, [1])):
    torch.manual_seed(B * 100 + N)
    blk = MultiScaleHGNN(SC).to(dev).eval()
    f = torch.randn(B, N, 64, device=dev)
    outs = {}
    for mode, x in (("0", f.bfloat16()), ("1", f.bfloat16()), ("fp32", f.bfloat16().float())):
        os.environ["GN_NODE_FORM"] = "1" if mode == "fp32" else mode
        M.set_noise_mode("device", 1234)
        with torch.no_grad():
            outs[mode] = blk(x)[0].float()
    torch.cuda.synchronize()
    ref = outs["fp32"]
    sc = float(ref.abs().max())
    print(f"B={B} N={N} scales={SC}: pair-form twin vs fp32 {float((outs['0'] - ref).abs().max()) / sc:.2e}  "
          f"scene-form twin vs fp32 {float((outs['1'] - ref).abs().max()) / sc:.2e}  twin vs twin {float((outs['0'] - outs['1']).abs().max()) / sc:.2e}")
