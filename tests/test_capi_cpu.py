"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol the
header declares, the modules keep the reference's state_dict layout, and the product never
imports the oracle.  No kernel is launched."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT, load_state

HEADER = os.path.join(ROOT, "include", "groupnet_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from groupnet_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 14, names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/groupnet_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), (set(names) ^ set(_lib.SIGNATURES))
    assert lib.gn_abi_version() == _lib.ABI_VERSION
    assert lib.gn_strerror(-3) == b"selected index k out of range"
    assert lib.gn_packed_elems(256, 64) == 8 * 2 * 1024 and lib.gn_packed_elems(11, 256) == 8 * 1024


def test_null_and_shape_errors_do_not_launch():
    """Argument validation happens on the host before any launch, so it can run without a GPU."""
    from groupnet_amd import _lib
    lib = _lib.load()
    P = ctypes.c_void_p
    assert lib.gn_affinity_f32(P(0), P(0), 1, 1, 64, P(0)) == -1
    assert lib.gn_affinity_f32(P(16), P(16), 0, 1, 64, P(0)) == -2
    assert lib.gn_affinity_f32(P(16), P(16), 1, 1, 63, P(0)) == -2
    assert lib.gn_affinity_f32(P(8), P(16), 1, 1, 64, P(0)) == -4
    Hs = (P * 1)(16)
    ks = (ctypes.c_int * 1)(12)
    assert lib.gn_topk_incidence_f32(P(16), Hs, ks, 1, 2, 11, P(0)) == -3      # k > N
    # grouped stages: descriptors are validated on the host too
    g = (_lib.Mlp2Group * 1)(_lib.Mlp2Group(16, 16, 16, 16))
    assert lib.gn_mlp2_f32(g, 1, 5, 96, 128, 64, 64, 0, 1.0, P(0)) == -2    # unsupported widths
    assert lib.gn_mlp2_f32(g, 0, 5, 128, 128, 64, 64, 0, 1.0, P(0)) == -2   # no groups
    assert lib.gn_mlp2_f32(g, 11, 5, 128, 128, 64, 64, 0, 1.0, P(0)) == -2  # > GN_MAX_GROUPS
    assert lib.gn_mlp2_f32(None, 1, 5, 128, 128, 64, 64, 0, 1.0, P(0)) == -1
    g[0].x = 0                                                               # fused scatter needs din=128, N, divisor
    assert lib.gn_mlp2_f32(g, 1, 22, 64, 128, 64, 64, 11, 11.0, P(0)) == -2
    assert lib.gn_mlp2_f32(g, 1, 22, 128, 128, 64, 64, 11, 0.0, P(0)) == -2
    a = (_lib.AggGroup * 1)(_lib.AggGroup(16, 16, 16, 16, 16, 16, 5, 17))
    assert lib.gn_agg_mlp_f32(a, 1, P(0)) == -2                             # K > GN_MAX_TYPES
    a[0].K, a[0].W = 6, 8
    assert lib.gn_agg_mlp_f32(a, 1, P(0)) == -4                             # misaligned weight stream
    n = (_lib.N2EGroup * 1)(_lib.N2EGroup(16, 16, 0, 16, 16, 16, 8))
    assert lib.gn_node2edge_f32(n, 1, 2, 3, P(0)) == -2                     # pairwise needs E == N*N
    n[0].b2 = 0
    assert lib.gn_node2edge_f32(n, 1, 2, 3, P(0)) == -1                     # the bias is a device pointer
    # backward blocks validate on the host as well
    d = (_lib.GemmDesc * 1)(_lib.GemmDesc(16, 16, 16, 0, 0, 0, 0, 4, 4, 4, 4, 4, 3, 0, 0, 0, 1.0, 0.0))
    assert lib.gn_gemm_grouped_f32(d, 1, P(0)) == -2                        # ldc < N
    d[0].ldc, d[0].colsum = 4, 16
    assert lib.gn_gemm_grouped_f32(d, 1, P(0)) == -2                        # colsum needs GN_GEMM_TRANS_A
    assert lib.gn_gemm_grouped_f32(None, 1, P(0)) == -1
    assert lib.gn_node2edge_bwd_f32(P(16), P(16), P(0), P(16), P(16), P(16), P(16), P(16), P(16), P(16), 2, 3, 8, 0,
                                    P(0)) == -2                             # pairwise: E == N*N
    assert lib.gn_gumbel_bwd_f32(P(16), P(16), P(16), P(0), P(16), 7, 6, 32, 0.5, 3, P(0)) == -2   # rows % pairs
    e = (_lib.EdgeGroup * 1)(_lib.EdgeGroup(16, 0, 16, 16, 16, 16, 0, 10, 16))
    assert lib.gn_edge_mlp_gumbel_f32(e, 1, 0.5, 0, P(0), P(0)) == -2       # K > 15
    e[0].K = 10
    assert lib.gn_edge_mlp_gumbel_f32(e, 1, 0.0, 0, P(0), P(0)) == -2       # tau must be > 0


def test_state_dict_layout_matches_reference_checkpoints():
    import groupnet_amd as G
    for nmp, suffix in ((1, ""), (2, "_nmp2")):
        pair = G.MS_HGNN_oridinary(embedding_dim=16, h_dim=64, mlp_dim=64, bottleneck_dim=64, batch_norm=0,
                                   nmp_layers=nmp)
        hyper = G.MS_HGNN_hyper(embedding_dim=64, h_dim=64, mlp_dim=64, bottleneck_dim=64, batch_norm=0,
                                nmp_layers=nmp, scale=5)
        for mod, name in ((pair, "pairwise"), (hyper, "hyper")):
            ref = load_state(name + suffix)
            own = mod.state_dict()
            assert list(own.keys()) == list(ref.keys())          # same names, same order
            assert all(own[k].shape == ref[k].shape for k in ref)
            mod.load_state_dict(ref, strict=True)
    assert sum(p.numel() for p in pair.parameters()) > 212168   # nmp=2 has more
    p1 = G.MS_HGNN_oridinary(embedding_dim=16, h_dim=64, mlp_dim=64, bottleneck_dim=64, batch_norm=0, nmp_layers=1)
    h1 = G.MS_HGNN_hyper(embedding_dim=64, h_dim=64, mlp_dim=64, bottleneck_dim=64, batch_norm=0, nmp_layers=1)
    assert sum(p.numel() for p in p1.parameters()) == 212168
    assert sum(p.numel() for p in h1.parameters()) == 283340


@pytest.mark.skipif(not os.path.isdir("/root/reference/model"), reason="reference only exists in the build container")
def test_same_seed_same_init_as_reference():
    """Construction order mirrors the reference, so a seeded default init is identical."""
    import sys
    import groupnet_amd as G
    sys.path.insert(0, "/root/reference")
    sys.dont_write_bytecode = True
    from model import MS_HGNN_batch as ref
    kw = dict(h_dim=64, mlp_dim=64, bottleneck_dim=64, batch_norm=0, nmp_layers=2)
    for own_cls, ref_cls, extra in ((G.MS_HGNN_oridinary, ref.MS_HGNN_oridinary, dict(embedding_dim=16)),
                                    (G.MS_HGNN_hyper, ref.MS_HGNN_hyper, dict(embedding_dim=64, scale=3))):
        torch.manual_seed(5)
        a = own_cls(**extra, **kw).state_dict()
        torch.manual_seed(5)
        b = ref_cls(**extra, **kw).state_dict()
        assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)


def test_cpu_tensors_are_refused_not_emulated():
    import groupnet_amd as G
    from groupnet_amd import ops
    m = G.MS_HGNN_hyper(embedding_dim=64, h_dim=64, mlp_dim=64, bottleneck_dim=64, batch_norm=0, nmp_layers=1)
    with torch.no_grad(), pytest.raises(ValueError, match="GPU"):
        m(torch.zeros(2, 11, 64), torch.zeros(2, 11, 11))
    with pytest.raises(ValueError):
        ops.affinity(torch.zeros(2, 11, 64))
    with pytest.raises(NotImplementedError):
        G.MS_HGNN_hyper(h_dim=32)
    with pytest.raises(ValueError):      # also with autograd on: there is no CPU path to differentiate either
        m(torch.zeros(2, 11, 64, requires_grad=True), torch.zeros(2, 11, 11))


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "groupnet_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), fn
                assert "ms_hgnn_oracle" not in text, fn


def test_mlp_standalone_has_no_cpu_path_either():
    """`MLP` called on its own (the reference imports it for heads outside the path) runs its Linears on the
    HIP GEMM; like every other module it refuses CPU tensors instead of falling back to torch math."""
    from groupnet_amd import MLP
    torch.manual_seed(0)
    m = MLP(8, 3, hidden_size=(16, 5))
    assert [tuple(l.weight.shape) for l in m.layers] == [(16, 8), (5, 16), (3, 5)]
    with pytest.raises(ValueError):
        m(torch.randn(4, 8))


def test_packed_weight_cache_policy():
    """The cache fingerprint (address, in-place version) misses `.data` writes — so while autograd records for the
    parameters the cache must not be trusted (`_volatile`), and only then."""
    from groupnet_amd import MS_HGNN_batch as M
    lin = torch.nn.Linear(4, 3)
    params = list(lin.parameters())
    k0 = M._param_key(params)
    lin.weight.data.mul_(2.0)
    assert M._param_key(params) == k0                    # the blind spot
    with torch.no_grad():
        lin.weight.mul_(2.0)
    assert M._param_key(params) != k0                    # ordinary in-place updates are seen
    assert M._volatile(params)                           # grad mode on, parameters trainable: re-pack every call
    with torch.no_grad():
        assert not M._volatile(params)                   # inference: cache trusted
        with M.training_call():
            assert M._volatile(params)                   # the forward of an autograd Function runs under no_grad
        assert not M._volatile(params)
    for p in params:
        p.requires_grad_(False)
    assert not M._volatile(params)                       # frozen parameters
