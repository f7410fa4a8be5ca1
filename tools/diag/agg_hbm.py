"""Diagnostic: the stand-alone hyper gather / scatter pair at N=11, B=4096 (bench.py's agg_hbm leg) on its own."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from groupnet_amd import ops
import bench
dev = torch.device("cuda")
for Bb, Nn, sc in ((4096, 11, 5), (512, 11, 5), (1024, 50, 8), (64, 50, 8), (1024, 24, 6), (1024, 32, 8)):
    ori = torch.randn(Bb, Nn, 64, device=dev)
    _, Hs, _ = ops.affinity_topk(ori, [sc], want_corr=False)
    H = Hs[0]
    feat = torch.randn(Bb, Nn, 64, device=dev)
    tg = bench.time_kernel_ms(lambda: ops.agg_gather(ori, H))
    ts = bench.time_kernel_ms(lambda: ops.agg_scatter(feat, H, ori))
    by = bench.agg_hbm_bytes(Bb, Nn, Nn)
    print(f"GN_GS_ROWS={os.environ.get('GN_GS_ROWS')} B={Bb} N={Nn}: gather {tg*1e3:.2f} us scatter {ts*1e3:.2f} us -> {by/((tg+ts)*1e-3)/1e9:.0f} GB/s")
