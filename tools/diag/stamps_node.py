"""Diagnostic (not product): per-wave cycle stamps of node_stage_kernel from the -DGN_STAMPS build."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["GROUPNET_HIP_LIB"] = os.path.join(ROOT, "tools", "diag", sys.argv[1])
import numpy as np
import torch
from groupnet_amd import _lib, ops
from groupnet_amd.multiscale import MultiScaleHGNN
import groupnet_amd as G

B, N = 512, 11
dev = torch.device("cuda")
torch.manual_seed(0)
blk = MultiScaleHGNN([2, 5, 11]).to(dev).eval()
f = torch.randn(B, N, 64, device=dev)
mods = [blk.interaction, *blk.interaction_hyper]
with torch.no_grad():
    pks = [m._packed_n2e(0) for m in mods]
    agg = blk.interaction.edge_aggregation_list[0]
    specs = [(agg._packed(), 6), None, None, None]
    for _ in range(20):
        ops.node_stage_grouped([(f, pk) for pk in pks], None, specs)
    torch.cuda.synchronize()
lib = _lib.load()
lib.gn_debug_read_stamps.restype = ctypes.c_int
lib.gn_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
n_units = 704 + 528
buf = np.zeros((n_units, 16), dtype=np.uint64)
rc = lib.gn_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
print("rc", rc)
s = buf.astype(np.int64)
rt0 = s[:, 8].min()
print("kernel span (realtime 100MHz): %.2f us" % ((s[:, 9].max() - rt0) / 100.0))
for name, sl in (("chain", slice(0, 704)), ("A", slice(704, n_units))):
    u = s[sl]
    start = (u[:, 8] - rt0) / 100.0
    end = (u[:, 9] - rt0) / 100.0
    cyc = u[:, 4] - u[:, 0]
    dur = end - start
    print(f"{name}: start us min/med/max {start.min():.2f}/{np.median(start):.2f}/{start.max():.2f}  end {end.min():.2f}/{np.median(end):.2f}/{end.max():.2f}")
    print(f"   duration us med {np.median(dur):.2f} max {dur.max():.2f}; cycles med {np.median(cyc):.0f} -> clock {np.median(cyc/np.maximum(dur,1e-3))/1e3:.2f} GHz")
    print(f"   prologue cycles (t0->t1) med {np.median(u[:,1]-u[:,0]):.0f}", end="")
    if name == "chain":
        print(f"; pair (t1->t2) med {np.median(u[:,2]-u[:,1]):.0f} min {np.min(u[:,2]-u[:,1]):.0f} max {np.max(u[:,2]-u[:,1]):.0f}; pq (t2->t3) {np.median(u[:,3]-u[:,2]):.0f}; store {np.median(u[:,4]-u[:,3]):.0f}")
    else:
        print(f"; body (t1->t4) med {np.median(u[:,4]-u[:,1]):.0f}")
