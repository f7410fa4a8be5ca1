"""Diagnostic (not product): per-wave cycle stamps of the bf16-core kernels of one forward, -DGN_STAMPS build.
Each stage is run alone (the stamp buffer is shared) after a full forward produced its inputs."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["GROUPNET_HIP_LIB"] = os.path.join(ROOT, "tools", "diag", sys.argv[1])
import numpy as np
import torch
from groupnet_amd import _lib, ops
from groupnet_amd.multiscale import MultiScaleHGNN
import groupnet_amd as G
import groupnet_amd.MS_HGNN_batch as M

B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
N = int(sys.argv[3]) if len(sys.argv) > 3 else 11
SC = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [2, 5, 11]
DT = torch.bfloat16 if (len(sys.argv) > 5 and sys.argv[5] == "bf16") else torch.float32
dev = torch.device("cuda")
torch.manual_seed(0)
blk = MultiScaleHGNN(SC).to(dev).eval()
f = torch.randn(B, N, 64, device=dev).to(DT)
lib = _lib.load()
lib.gn_debug_read_stamps.restype = ctypes.c_int
lib.gn_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]

def report(tag, n_units, groups=None):
    torch.cuda.synchronize()
    buf = np.zeros((n_units, 16), dtype=np.uint64)
    lib.gn_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
    s = buf.astype(np.int64)
    live = s[:, 9] > 0
    rt0 = s[live, 8].min()
    print(f"== {tag}: {live.sum()} waves, span {(s[live, 9].max() - rt0) / 100.0:.2f} us")
    for name, sl in (groups or [("all", slice(0, n_units))]):
        u = s[sl][live[sl]]
        if len(u) == 0:
            continue
        start, end = (u[:, 8] - rt0) / 100.0, (u[:, 9] - rt0) / 100.0
        pro = u[:, 1] - u[:, 0]
        body = u[:, 2] - u[:, 1]
        tail = u[:, 4] - u[:, 2]
        mid = u[:, 3] - u[:, 2]
        epi = u[:, 4] - u[:, 3]
        if (u[:, 5] > 0).all() and (u[:, 7] > u[:, 5]).all():
            print(f"  {name:10s} second type: slot 5->6 {np.median(u[:, 6] - u[:, 5]):6.0f}  6->7 {np.median(u[:, 7] - u[:, 6]):6.0f} cycles (min {(u[:, 7] - u[:, 6]).min():6.0f})")
        print(f"  {name:10s} n={len(u):5d} start med/max {np.median(start):5.2f}/{start.max():5.2f}  end med/max {np.median(end):5.2f}/{end.max():5.2f} us |"
              f" cycles: prologue {np.median(pro):6.0f}  body med {np.median(body):6.0f} min {body.min():6.0f} max {body.max():6.0f}  tail {np.median(tail):6.0f} (2->3 {np.median(mid):6.0f}, 3->4 {np.median(epi):6.0f})")

calls = []
orig = {}
def wrap(name, n_units_fn, groups_fn=None):
    fn = getattr(ops, name)
    orig[name] = fn
    def w(*a, **k):
        for _ in range(5):
            r = fn(*a, **k)
        report(name, n_units_fn(*a, **k), groups_fn(*a, **k) if groups_fn else None)
        return r
    setattr(ops, name, w)

NCH = 704
wrap("node_stage_grouped", lambda items, keep, specs, *rest: 8192 if B > 512 else NCH + 1056, (lambda *a: None) if B > 512 else (lambda *a: [("chain", slice(0, NCH)), ("A", slice(NCH, NCH + 1056))]))
def edge_rows(e):
    if isinstance(e, ops.PoolSpec):       # rows formed inside the kernel: unordered pairs / hyperedges
        Bn, Nn = e.xp.shape[0], e.xp.shape[1]
        return Bn * (Nn * (Nn + 1) // 2 if e.H is None else e.H.shape[1])
    return e.shape[0] * e.shape[1]
def edge_units(items, *a, **k):
    return sum(((edge_rows(it[0]) + 127) // 128) * 4 for it in items)
wrap("edge_mlp_gumbel_grouped", (lambda *a, **k: 8192) if B > 512 else edge_units, (lambda *a, **k: None) if B > 512 else (lambda items, *a, **k: [("pair", slice(0, 1056)), ("hyper", slice(1056, 1056 + 400))]))
def agg_groups(items, closing=None):
    if closing is not None:      # fused closing stage: hyper modules 103 workgroups each (5 scenes), scale = N 37 (14 scenes)
        return [("hyperA", slice(0, 412)), ("hyperB", slice(412, 824)), ("small", slice(824, 972)), ("node", slice(972, 1676))]
    return None
wrap("agg_mlp_grouped", lambda items, closing=None: 8192 if B > 512 else 4 * 700, (lambda items, closing=None: None) if B > 512 else (lambda items, closing=None: agg_groups(items, closing) or [("hyperA", slice(0, 352)), ("hyperB", slice(352, 704)), ("small", slice(704, 768)), ("node", slice(768, 1472))]
                                   if any(isinstance(it[0], ops.PairSpec) and it[0].node for it in items) else
                                   [("hyperA", slice(0, 352)), ("hyperB", slice(352, 704)), ("pair", slice(704, 1760)), ("small", slice(1760, 1824))]))
wrap("mlp2_grouped", lambda items, keep=None: 8192 if B > 512 else (4 * 176 * 4 if os.environ.get("GN_MLP2_XS", "1") != "0" else 4 * 44 * 4), None)
M.ops = ops
with torch.no_grad():
    G.set_noise_mode("device", seed=3)
    blk(f)
torch.cuda.synchronize()
