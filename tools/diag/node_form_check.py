"""Diagnostic (not product): node form of the pairwise typed aggregation (GN_NODE_FORM=1) against the per-pair form
(GN_NODE_FORM=0) on the multiscale block, several shapes: max |difference| of every output."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from groupnet_amd.multiscale import MultiScaleHGNN
from groupnet_amd import MS_HGNN_batch as M
dev = torch.device("cuda")
for (B, N, SC) in ((3, 11, [2, 5, 11]), (37, 11, [2, 5, 11]), (512, 11, [2, 5, 11]), (5, 2, [2]), (7, 16, [2, 4]), (4, 1, [1]), (9, 5, [2, 5])):
    torch.manual_seed(B * 100 + N)
    blk = MultiScaleHGNN(SC).to(dev).eval()
    f = torch.randn(B, N, 64, device=dev)
    outs = {}
    for mode in ("0", "1"):
        os.environ["GN_NODE_FORM"] = mode
        M.set_noise_mode("device", 1234)
        with torch.no_grad():
            outs[mode] = blk(f)
    torch.cuda.synchronize()
    a, b = outs["0"], outs["1"]
    flat = lambda o: [t for t in (o if isinstance(o, (tuple, list)) else [o]) for t in (flat(t) if isinstance(t, (tuple, list)) else [t])]
    d = [float((x.float() - y.float()).abs().max()) for x, y in zip(flat(a), flat(b)) if torch.is_tensor(x)]
    s = [float(x.float().abs().max()) for x in flat(a) if torch.is_tensor(x)]
    print(f"B={B} N={N} scales={SC}: max|diff| per output {['%.2e' % v for v in d]} scales {['%.2e' % v for v in s]}")
