"""Diagnostic: hyper node->edge pooling, banded form vs one lane pair per hyperedge (GN_N2E_ROWS = 0 / 1), by launch size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from groupnet_amd import ops
import bench
dev = torch.device("cuda")
for N, scales, Bs in ((11, [2, 5, 11], (512, 1024, 2048, 4096)), (50, [2, 4, 8, 16], (32, 64, 128, 256, 1024))):
    for B in Bs:
        items = []
        for s in scales:
            xp = torch.randn(B, N, 64, device=dev); pq = torch.randn(B, N, 64, device=dev)
            H = ops.topk_incidence(torch.rand(B, N, N, device=dev), [s])[0]
            items.append((xp, pq, H, torch.randn(32, device=dev), torch.randn(1, device=dev)))
        rows = sum(it[2].shape[0] * it[2].shape[1] for it in items)
        t = {}
        for v in ("0", "1"):
            os.environ["GN_N2E_ROWS"] = v
            t[v] = bench.time_kernel_ms(lambda: ops.node2edge_grouped(items)) * 1e3
        print(f"N={N:3d} B={B:5d} hyper rows {rows:7d}: banded {t['0']:7.2f} us   rows {t['1']:7.2f} us")
