"""Parity of the HIP path (through the C ABI) with the CPU oracle and with the golden vectors
generated from the reference.  Needs an MI355X: `pytest -m gpu`.

Bars (BASELINE.json north_star): top-k incidence H bit-identical; fp32 features within 1e-5 abs.
"""
import os

import numpy as np
import pytest
import torch

from conftest import case_names, load_case, load_state, uniforms, weights_for
from oracle import ms_hgnn_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-5       # north_star: fp32 features within 1e-5
TOL_CORR = 2e-6  # affinity entries are O(1); a 64-term fp32 dot differs from MKL's by a few ulp


def dev():
    return torch.device("cuda:0")


def to_dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev())


def maxerr(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else b
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0


def build_modules(nmp, scale=2, bottleneck=64):
    import groupnet_amd as G
    pair = G.MS_HGNN_oridinary(embedding_dim=16, h_dim=64, mlp_dim=64, bottleneck_dim=bottleneck, batch_norm=0,
                               nmp_layers=nmp)
    hyper = G.MS_HGNN_hyper(embedding_dim=64, h_dim=64, mlp_dim=64, bottleneck_dim=bottleneck, batch_norm=0,
                            nmp_layers=nmp, scale=scale)
    return pair, hyper


def loaded_modules(case):
    sp, sh, nmp = weights_for(case)
    pair, hyper = build_modules(nmp)
    pair.load_state_dict(sp, strict=True)    # reference key names / shapes (test_nba.py:603)
    hyper.load_state_dict(sh, strict=True)
    return pair.to(dev()).eval(), hyper.to(dev()).eval(), sp, sh, nmp


# ---------------------------------------------------------------------------------------------
# golden vectors of the reference
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("xs", ["1", "0"])
@pytest.mark.parametrize("name", case_names())
def test_modules_match_reference_goldens(name, xs, monkeypatch):
    """xs: the closing MLP (with its fused scatter) through its 4-waves-per-row-block kernel ("1", what the launcher picks
    at these sizes) and through the one-wave-per-row-block kernel ("0", what it picks for large launches)."""
    monkeypatch.setenv("GN_MLP2_XS", xs)
    c = load_case(name)
    pair, hyper, sp, sh, nmp = loaded_modules(name)
    h, corr = to_dev(c["h"]), to_dev(c["corr"])
    with torch.no_grad():
        if "pair_node_feat" in c:
            U = [u.to(dev()) for u in uniforms(c, "pair")]
            nf, fac = pair(h, noise_u=U)
            assert maxerr(nf, c["pair_node_feat"]) <= TOL
            assert maxerr(fac, c["pair_factors"]) <= TOL
        for s in c["scales"].tolist():
            hyper.scale = s
            U = [u.to(dev()) for u in uniforms(c, f"hyper{s}")]
            nf, fac, H = hyper(h, corr, noise_u=U)
            assert np.array_equal(H.cpu().numpy(), c[f"hyper{s}_H"]), (name, s)   # bit-exact
            assert H.dtype == h.dtype
            assert maxerr(nf, c[f"hyper{s}_node_feat"]) <= TOL, (name, s)
            assert maxerr(fac, c[f"hyper{s}_factor"]) <= TOL, (name, s)


@pytest.mark.parametrize("name", [n for n in case_names() if not n.endswith("nmp2")])
def test_each_kernel_against_reference_intermediates(name):
    """Stage by stage, feeding each kernel the REFERENCE's inputs for that stage."""
    from groupnet_amd import ops
    c = load_case(name)
    pair, hyper, sp, sh, nmp = loaded_modules(name)
    h, corr = to_dev(c["h"]), to_dev(c["corr"])
    B, N = h.shape[:2]
    with torch.no_grad():
        assert maxerr(ops.affinity(h), c["corr"]) <= TOL_CORR
        scales = c["scales"].tolist()
        Hs = ops.topk_incidence(corr, scales)
        for s, H in zip(scales, Hs):
            assert np.array_equal(H.cpu().numpy(), c[f"hyper{s}_H"])
        todo = [("pair", pair, None)] if "pair_node_feat" in c else []
        todo += [(f"hyper{s}", hyper, H) for s, H in zip(scales, Hs)]
        for tag, mod, H in todo:
            pk = mod._packed_n2e(0)
            xp, pq = ops.node_mlp(h, pk)
            assert maxerr(xp, c[f"{tag}_xp"]) <= TOL, tag
            edges = ops.node2edge(to_dev(c[f"{tag}_xp"]), pq, H, pk["w2"], pk["b2"])
            assert maxerr(edges, c[f"{tag}_edges"]) <= TOL, tag
            K = mod.edge_types
            ef, dist = ops.edge_mlp_gumbel(to_dev(c[f"{tag}_edges"]), to_dev(c[f"{tag}_U0"]),
                                           mod.nmp_mlp_start._packed(), K)
            assert maxerr(ef, c[f"{tag}_edge_feat"]) <= TOL, tag
            fac_key = "pair_factors" if tag == "pair" else f"{tag}_factor"
            assert maxerr(dist, c[fac_key]) <= TOL, tag
            eo = ops.agg_gather(h, H)
            assert maxerr(eo, c[f"{tag}_eo"]) <= TOL, tag
            agg = mod.edge_aggregation_list[0]
            feat = ops.agg_mlp(to_dev(c[f"{tag}_eo"]), to_dev(c[f"{tag}_edge_feat"]), agg._packed(), K)
            out = ops.agg_scatter(feat, H, h)
            assert maxerr(out, c[f"{tag}_agg"]) <= TOL, tag
            raw = agg(to_dev(c[f"{tag}_edge_feat"]), H, h)     # edge_aggregation.forward: no / N
            assert maxerr(raw[..., :64], c[f"{tag}_feat_scattered"]) <= TOL * N, tag
            y = ops.mlp2(to_dev(c[f"{tag}_agg"]), mod._packed_mlp2(mod.nmp_mlp_end))
            ref_key = "pair_node_feat" if tag == "pair" else f"{tag}_node_feat"
            assert maxerr(y, c[ref_key]) <= TOL, tag


def test_fused_affinity_topk_matches_separate():
    from groupnet_amd import ops
    c = load_case("syn_n11_b37")
    h = to_dev(c["h"])
    scales = c["scales"].tolist()
    corr, Hs, _ = ops.affinity_topk(h, scales)
    assert maxerr(corr, c["corr"]) <= TOL_CORR
    # ranked from the kernel's own corr: identical to ranking that corr with the oracle's rule
    for s, H in zip(scales, Hs):
        assert np.array_equal(H.cpu().numpy(), O.topk_incidence_ranked(corr.cpu(), s).numpy())
    # and on these well-separated fixtures also identical to the reference's H
    for s, H in zip(scales, Hs):
        if float(c.get(f"hyper{s}_min_gap", 1.0)) > 1e-4:
            assert np.array_equal(H.cpu().numpy(), c[f"hyper{s}_H"])
    final = torch.zeros(h.shape[0], h.shape[1], 320, device=dev())
    ctr = torch.tensor([5], dtype=torch.int64, device=dev())
    _, Hs2, Hcat = ops.affinity_topk(h, scales, want_corr=False, f_out=final[..., :64], want_H_cat=True,
                                     counter=ctr, counter_add=7)
    for a, b in zip(Hs, Hs2):
        assert torch.equal(a, b)
    assert torch.equal(Hcat, torch.cat(Hs, dim=1))
    assert torch.equal(final[..., :64], h) and bool((final[..., 64:] == 0).all())
    assert int(ctr.item()) == 12


# ---------------------------------------------------------------------------------------------
# oracle on fresh seeded inputs (sizes the oracle finishes in seconds)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,N,scales,nmp,bott", [(64, 11, [2, 5, 11], 1, 64), (5, 23, [4, 23], 3, 64),
                                                  (130, 7, [3], 1, 96), (2, 50, [16], 1, 1024), (1, 1, [1], 1, 64)])
def test_modules_match_oracle_random(B, N, scales, nmp, bott):
    torch.manual_seed(1000 + B + N)
    pair, hyper = build_modules(nmp, bottleneck=bott)
    with torch.no_grad():
        for m in (pair, hyper):
            for n_, p in m.named_parameters():
                if "attention_mlp" in n_ or "MLP_distribution" in n_ or "MLP_factor" in n_:
                    p.mul_(4.0)
    sp = {k: v.detach().clone() for k, v in pair.state_dict().items()}
    sh = {k: v.detach().clone() for k, v in hyper.state_dict().items()}
    pair.to(dev()).eval()
    hyper.to(dev()).eval()
    h = torch.randn(B, N, 64)
    corr = O.affinity(h)
    Up = [torch.rand(s) for s in O.noise_shapes(B, N, None, nmp)]
    with torch.no_grad():
        nf_o, fac_o = O.ms_hgnn_pairwise_forward(sp, h, Up, nmp, decomposed=True)
        nf, fac = pair(h.to(dev()), noise_u=[u.to(dev()) for u in Up])
        assert maxerr(nf, nf_o) <= TOL and maxerr(fac, fac_o) <= TOL
        for s in scales:
            Uh = [torch.rand(x) for x in O.noise_shapes(B, N, s, nmp)]
            nf_o, fac_o, H_o = O.ms_hgnn_hyper_forward(sh, h, corr, s, Uh, nmp, decomposed=True)
            hyper.scale = s
            nf, fac, H = hyper(h.to(dev()), corr.to(dev()), noise_u=[u.to(dev()) for u in Uh])
            assert torch.equal(H.cpu(), O.topk_incidence_ranked(corr, s))
            assert torch.equal(H.cpu(), H_o)
            assert maxerr(nf, nf_o) <= TOL and maxerr(fac, fac_o) <= TOL


def test_general_incidence_values_and_dense_rows():
    """node2edge / gather / scatter take ANY float H (the pairwise graph has weight-2 entries);
    check a dense random H against the oracle, and the implicit pairwise graph against the
    materialised one."""
    from groupnet_amd import ops
    torch.manual_seed(5)
    B, N, E = 6, 9, 13
    pair, hyper = build_modules(1)
    sh = {k: v.detach().clone() for k, v in hyper.state_dict().items()}
    hyper.to(dev())
    h = torch.randn(B, N, 64)
    H = torch.randint(0, 3, (B, E, N)).float() * torch.rand(B, E, N).round()
    H[0, 0] = 0.0   # an empty hyperedge
    H[1, 1] = 1.0   # a full one
    edges_o, xp_o = O.node2edge(sh, h, H, 0, decomposed=True)
    pk = hyper._packed_n2e(0)
    xp, pq = ops.node_mlp(h.to(dev()), pk)
    edges = ops.node2edge(xp, pq, H.to(dev()), pk["w2"], pk["b2"])
    assert maxerr(xp, xp_o) <= TOL and maxerr(edges, edges_o) <= TOL
    assert maxerr(ops.agg_gather(h.to(dev()), H.to(dev())), O.aggregate_gather(H, h)) <= TOL
    feat = torch.randn(B, E, 64)
    assert maxerr(ops.agg_scatter(feat.to(dev()), H.to(dev()), h.to(dev())), O.aggregate_scatter(H, feat, h)) <= TOL
    # implicit pairwise == explicit pairwise incidence
    Hp = O.pairwise_incidence(N, B)
    e_imp = ops.node2edge(xp, pq, None, pk["w2"], pk["b2"])
    e_exp = ops.node2edge(xp, pq, Hp.to(dev()), pk["w2"], pk["b2"])
    assert maxerr(e_imp, e_exp) <= 1e-6
    assert maxerr(ops.agg_gather(h.to(dev()), None), O.aggregate_gather(Hp, h)) <= TOL
    featp = torch.randn(B, N * N, 64)
    assert maxerr(ops.agg_scatter(featp.to(dev()), None, h.to(dev())), O.aggregate_scatter(Hp, featp, h)) <= TOL


def test_topk_ties_nan_and_errors():
    from groupnet_amd import ops
    corr = torch.tensor([[[1.0, 1.0, 0.5, 1.0], [0.0, float("nan"), 2.0, 2.0],
                          [3.0, 2.0, 1.0, 0.0], [0.0, 0.0, 0.0, 0.0]]])
    (H,) = ops.topk_incidence(corr.to(dev()), [2])
    assert torch.equal(H.cpu(), O.topk_incidence_ranked(corr, 2))
    assert H[0, 0].tolist() == [1, 1, 0, 0] and H[0, 1].tolist() == [0, 1, 1, 0]
    H0, H4 = ops.topk_incidence(corr.to(dev()), [0, 4])
    assert H0.sum().item() == 4 and H4.shape == (1, 1, 4) and bool((H4 == 1).all())
    with pytest.raises(RuntimeError):
        ops.topk_incidence(corr.to(dev()), [5])           # torch.topk: k out of range (MS_HGNN_batch.py:382)
    with pytest.raises(ValueError):
        ops.topk_incidence(corr, [2])                     # CPU tensor: no fallback
    with pytest.raises(ValueError):
        ops.affinity(torch.zeros(2, 3, 64, dtype=torch.float64, device=dev()))


def test_topk_large_n_banded():
    """N=256 (BASELINE config 5): banded affinity + banded top-k, 4 scales in one pass."""
    from groupnet_amd import ops
    torch.manual_seed(9)
    h = torch.randn(3, 256, 64)
    corr_o = O.affinity(h)
    corr = ops.affinity(h.to(dev()))
    assert maxerr(corr, corr_o) <= TOL_CORR
    scales = [2, 8, 32, 128]
    Hs = ops.topk_incidence(corr, scales)
    cc = corr.cpu()
    for s, H in zip(scales, Hs):
        assert torch.equal(H.cpu(), O.topk_incidence_ranked(cc, s))
        assert torch.equal(H.cpu(), O.topk_incidence(cc, s))


def test_mlp2_shapes_and_pack():
    from groupnet_amd import MLP, ops
    torch.manual_seed(3)
    for din, dh, dout, rows in [(128, 128, 64, 300), (64, 256, 64, 33), (128, 128, 1024, 70), (64, 128, 10, 129),
                                (128, 256, 7, 1)]:
        m = MLP(din, dout, hidden_size=(dh,))
        x = torch.randn(rows, din)
        l0, l1 = m.layers
        with torch.no_grad():      # plain torch fp32 layer math on the CPU as the reference of this op
            y_ref = torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(x, l0.weight, l0.bias)),
                                               l1.weight, l1.bias)
        pk = dict(W=ops.pack_stream([l0.weight.detach().to(dev()), l1.weight.detach().to(dev())]),
                  bias=ops.bias_stream([l0.bias.detach().to(dev()), l1.bias.detach().to(dev())]),
                  din=din, dh=dh, dout=dout)
        y = ops.mlp2(x.to(dev()), pk)
        assert maxerr(y, y_ref) <= TOL, (din, dh, dout, rows)


def test_philox_matches_oracle_bit_exact():
    from groupnet_amd import ops
    for n, seed, off in [(4, 0, 0), (1000, 12345, 0), (777, 2**40 + 17, 3), (10, 5, 2**33 + 1)]:
        u = ops.philox_uniform((n,), seed, off, dev()).cpu().numpy()
        assert np.array_equal(u, O.philox_uniform(n, seed, off)), (n, seed, off)
    a = ops.philox_uniform((100,), 9, 0, dev())
    b = ops.philox_uniform((60,), 9, 40, dev())
    assert torch.equal(a[40:], b)   # a shard draws exactly its slice of the full stream


# ---------------------------------------------------------------------------------------------
# full-size, size-independent properties (BASELINE configs 2 and 3: N=11, B=512 / 4096)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B", [512, 4096])
def test_full_size_properties(B):
    import groupnet_amd as G
    from groupnet_amd import ops
    torch.manual_seed(77)
    N, scales = 11, [2, 5, 11]
    pair, hyper = build_modules(1)
    pair.to(dev()).eval()
    hyper.to(dev()).eval()
    h = torch.randn(B, N, 64, device=dev())
    G.set_noise_mode("device", seed=4242)
    try:
        with torch.no_grad():
            corr, Hs, _ = ops.affinity_topk(h, scales)
            assert torch.allclose(torch.diagonal(corr, dim1=1, dim2=2), torch.ones(B, N, device=dev()), atol=1e-5)
            assert torch.equal(corr, corr.transpose(1, 2))
            for s, H in zip(scales, Hs):
                assert bool(((H == 0) | (H == 1)).all())
                assert bool((H.sum(-1) == s).all())                     # every hyperedge has `scale` members
                if s != N:
                    assert bool((torch.diagonal(H, dim1=1, dim2=2) == 1).all())   # self is always selected
            U = ops.philox_uniform((B, N * N, 6), 1, 0, dev())
            nf, fac = pair(h, noise_u=U)
            assert nf.shape == (B, N, 64) and fac.shape == (B, N * N, 6)
            assert torch.allclose(fac.sum(-1), torch.ones(B, N * N, device=dev()), atol=1e-5)
            assert bool(torch.isfinite(nf).all())
            # batch-shard invariance (scenes are independent): halves == whole to rounding.  Not bit for
            # bit in general: the launchers pick how many waves share a row block from the launch size,
            # which changes the order partial sums are added in; the same size is bit-reproducible (the
            # permutation check below).
            half = B // 2
            nf_a, fac_a = pair(h[:half].contiguous(), noise_u=U[:half].contiguous())
            nf_b, fac_b = pair(h[half:].contiguous(), noise_u=U[half:].contiguous())
            assert maxerr(torch.cat((nf_a, nf_b)), nf) <= 1e-6 and maxerr(torch.cat((fac_a, fac_b)), fac) <= 1e-6
            # scene-permutation equivariance
            perm = torch.randperm(B, device=dev())
            nf_p, _ = pair(h[perm].contiguous(), noise_u=U[perm].contiguous())
            assert torch.equal(nf_p, nf[perm])
            for s, H in zip(scales, Hs):
                hyper.scale = s
                E = H.shape[1]
                Uh = ops.philox_uniform((B, E, 10), 2, 0, dev())
                nf, fac, H2 = hyper(h, corr, noise_u=Uh)
                assert torch.equal(H2, H)
                assert torch.allclose(fac.sum(-1), torch.ones(B, E, device=dev()), atol=1e-5)
                nf_a, _, _ = hyper(h[:half].contiguous(), corr[:half].contiguous(), noise_u=Uh[:half].contiguous())
                assert maxerr(nf_a, nf[:half]) <= 1e-6
            # aggregation linearity at full size: gather(a + b) == gather(a) + gather(b) up to rounding
            a, b = torch.randn_like(h), torch.randn_like(h)
            H = Hs[1]
            lhs = ops.agg_gather(a + b, H)
            rhs = ops.agg_gather(a, H) + ops.agg_gather(b, H)
            assert torch.allclose(lhs, rhs, atol=1e-5)
            # scatter of ones counts memberships: (H^T 1)/N on the first half, ori/N on the second
            ones = torch.ones(B, H.shape[1], 64, device=dev())
            out = ops.agg_scatter(ones, H, a)
            assert torch.allclose(out[..., :64], (H.sum(1) / N).unsqueeze(-1).expand(-1, -1, 64), atol=1e-6)
            # true division as the CPU reference does (torch's GPU `a / N` multiplies by 1/N instead)
            assert torch.equal(out[..., 64:].cpu(), a.cpu() / N)
            # default noise path in device mode runs and is reproducible from (seed, offset)
            G.set_noise_mode("device", seed=7, offset=0)
            x1, f1 = pair(h)
            G.set_noise_mode("device", seed=7, offset=0)
            x2, f2 = pair(h)
            assert torch.equal(x1, x2) and torch.equal(f1, f2)
    finally:
        G.set_noise_mode("host")


def test_host_noise_mode_is_the_reference_stream():
    """Default mode draws torch.rand on the CPU generator exactly like the reference: seeding the
    generator and calling pair then each hyper scale reproduces the golden outputs with no
    injected noise."""
    c = load_case("nba_b10")
    pair, hyper, *_ = loaded_modules("nba_b10")
    h, corr = to_dev(c["h"]), to_dev(c["corr"])
    torch.manual_seed(int(c["seed"]))
    with torch.no_grad():
        nf, fac = pair(h)
        assert maxerr(nf, c["pair_node_feat"]) <= TOL and maxerr(fac, c["pair_factors"]) <= TOL
        for s in c["scales"].tolist():
            hyper.scale = s
            nf, fac, H = hyper(h, corr)
            assert maxerr(nf, c[f"hyper{s}_node_feat"]) <= TOL and maxerr(fac, c[f"hyper{s}_factor"]) <= TOL


def test_graph_capture_replays():
    """The launchers never sync or allocate, so a forward can be captured into a hipGraph."""
    pair, hyper = build_modules(1)
    pair.to(dev()).eval()
    torch.manual_seed(0)
    h = torch.randn(32, 11, 64, device=dev())
    U = torch.rand(32, 121, 6, device=dev())
    with torch.no_grad():
        ref, _ = pair(h, noise_u=U)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            pair(h, noise_u=U)   # warm-up on the side stream
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            out, _ = pair(h, noise_u=U)
        h.copy_(torch.randn_like(h))
        g.replay()
        torch.cuda.synchronize()
        ref2, _ = pair(h, noise_u=U)
        assert torch.equal(out, ref2) and not torch.equal(ref, ref2)


@pytest.mark.parametrize("grouped", [False, True])
def test_multiscale_block_matches_oracle(grouped):
    """The PastEncoder-shaped block (model/GroupNet_nba.py:284-311): fused affinity+top-k, the
    1+S modules (every stage one grouped launch, or module by module), features written in place."""
    from groupnet_amd.multiscale import MultiScaleHGNN
    torch.manual_seed(21)
    scales = [2, 5, 11]
    blk = MultiScaleHGNN(scales, grouped=grouped)
    sp = {k: v.detach().clone() for k, v in blk.interaction.state_dict().items()}
    shs = [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in blk.interaction_hyper]
    blk.to(dev()).eval()
    B, N = 48, 11
    h = torch.randn(B, N, 64)
    noise = [[torch.rand(s)] for s in blk.noise_shapes(B, N)]
    with torch.no_grad():
        ref, Href, corr_ref = O.ms_hgnn_multiscale_forward(sp, shs, scales, h, noise[0], noise[1:], decomposed=True)
        out, H = blk(h.to(dev()), noise_u=[[u.to(dev()) for u in n] for n in noise])
    assert out.shape == (B, N, 64 * 5) and H.shape == (B, N + N + 1, N)
    assert torch.equal(H.cpu(), Href)
    assert maxerr(out, ref) <= TOL
    assert torch.equal(out[..., :64].cpu(), h)


def test_graphed_multiscale_fresh_reproducible_noise():
    """A captured forward draws from a device-side stream position that the graph advances itself:
    replay k equals an eager forward at offset k * draws_per_step."""
    import groupnet_amd as G
    from groupnet_amd.graphs import GraphedMultiScale
    from groupnet_amd.multiscale import MultiScaleHGNN
    torch.manual_seed(4)
    blk = MultiScaleHGNN([2, 5, 11]).to(dev()).eval()
    B, N = 16, 11
    f = torch.randn(B, N, 64, device=dev())
    g = GraphedMultiScale(blk, B, N, seed=321)
    o1 = g(f)[0].clone()
    o2 = g(f)[0].clone()
    assert not torch.equal(o1, o2)
    try:
        with torch.no_grad():
            G.set_noise_mode("device", seed=321, offset=0)
            e1 = blk(f)[0]
            G.set_noise_mode("device", seed=321, offset=g.draws_per_step)
            e2 = blk(f)[0]
    finally:
        G.set_noise_mode("host")
    assert torch.equal(o1, e1) and torch.equal(o2, e2)
    assert int(g.counter.item()) == g.draws_per_step   # base of the last replay


def test_in_kernel_philox_equals_uniform_tensor():
    """U == NULL path of the edge kernel draws exactly the stream gn_philox_uniform_f32 writes."""
    from groupnet_amd import ops
    torch.manual_seed(8)
    pair, hyper = build_modules(1)
    hyper.to(dev())
    B, E, K = 37, 11, 10
    edges = torch.randn(B, E, 64, device=dev())
    pk = hyper.nmp_mlp_start._packed()
    ctr = torch.tensor([1000], dtype=torch.int64, device=dev())
    U = ops.philox_uniform((B, E, K), 77, 123 + 1000, dev())
    ef_a, d_a = ops.edge_mlp_gumbel(edges, U, pk, K)
    ef_b, d_b = ops.edge_mlp_gumbel(edges, ops.PhiloxNoise(77, 123, ctr), pk, K)
    assert torch.equal(ef_a, ef_b) and torch.equal(d_a, d_b)
    ef_c, _ = ops.edge_mlp_gumbel(edges, ops.PhiloxNoise(77, 124, ctr), pk, K)
    assert not torch.equal(ef_a, ef_c)


def _pairs(N):
    return [(i, j) for i in range(N) for j in range(i, N)]


@pytest.mark.parametrize("B,N", [(5, 11), (2, 50), (3, 1), (1, 3)])
def test_symmetric_pairwise_stages_equal_ordered_ones(B, N):
    """The pairwise graph is symmetric; the engine runs its per-edge MLPs once per unordered pair.
    Each symmetric stage must agree with the ordered stage it replaces."""
    from groupnet_amd import ops
    torch.manual_seed(31 + N)
    pair, _ = build_modules(1)
    pair.to(dev())
    with torch.no_grad():
        for p in pair.attention_mlp.parameters():
            p.mul_(4.0)
    h = torch.randn(B, N, 64, device=dev())
    pk = pair._packed_n2e(0)
    xp, pq = ops.node_mlp(h, pk)
    P = ops.pair_count(N)
    pr = _pairs(N)
    assert len(pr) == P
    idx_ij = torch.tensor([i * N + j for i, j in pr], device=dev())
    idx_ji = torch.tensor([j * N + i for i, j in pr], device=dev())
    diag = torch.tensor([i == j for i, j in pr], device=dev())
    # node -> edge
    e_ord = ops.node2edge(xp, pq, None, pk["w2"], pk["b2"])
    e_sym = ops.node2edge(xp, pq, None, pk["w2"], pk["b2"], sym=True)
    assert e_sym.shape == (B, P, 64)
    assert maxerr(e_sym, e_ord[:, idx_ij]) <= 1e-6 and maxerr(e_sym, e_ord[:, idx_ji]) <= 1e-6
    # gather
    g_ord = ops.agg_gather(h, None)
    g_sym = ops.agg_gather(h, None, sym=True)
    assert torch.equal(g_sym, g_ord[:, idx_ij])
    # edge MLP + gumbel: ordered distributions identical, pair weight = sum of the two ordered ones
    K = pair.edge_types
    U = torch.rand(B, N * N, K, device=dev())
    spk = pair.nmp_mlp_start._packed()
    ef_ord, d_ord = ops.edge_mlp_gumbel(e_ord, U, spk, K)
    (ef_sym, d_sym), = ops.edge_mlp_gumbel_grouped([(e_sym, U, spk, K, N, True)])
    assert d_sym.shape == (B, N * N, K) and ef_sym.shape == (B, P, K)
    assert maxerr(d_sym, d_ord) <= 1e-6
    want = torch.where(diag[None, :, None], 2 * ef_ord[:, idx_ij], ef_ord[:, idx_ij] + ef_ord[:, idx_ji])
    assert maxerr(ef_sym, want) <= 1e-6
    # in-kernel Philox in the symmetric form reads the ordered stream positions
    Uph = ops.philox_uniform((B, N * N, K), 5, 17, dev())
    (ef_a, d_a), = ops.edge_mlp_gumbel_grouped([(e_sym, Uph, spk, K, N, True)])
    (ef_b, d_b), = ops.edge_mlp_gumbel_grouped([(e_sym, ops.PhiloxNoise(5, 17), spk, K, N, True)])
    assert torch.equal(ef_a, ef_b) and torch.equal(d_a, d_b)
    (ef_c, d_c), = ops.edge_mlp_gumbel_grouped([(e_sym, Uph, spk, K, N, False)])
    assert d_c is None and torch.equal(ef_c, ef_a)
    # scatter: pair sums == ordered sums
    f_ord = torch.randn(B, N * N, 64, device=dev())
    f_pair = torch.where(diag[None, :, None], 2 * f_ord[:, idx_ij], f_ord[:, idx_ij] + f_ord[:, idx_ji])
    s_ord = ops.agg_scatter(f_ord, None, h)
    s_sym = ops.agg_scatter(f_pair, None, h, sym=True)
    assert maxerr(s_sym, s_ord) <= 1e-5


def test_fused_gather_equals_standalone_gather():
    """eo == NULL form of the typed-MLP kernel (gather in the prologue) == gather kernel + typed MLP."""
    from groupnet_amd import ops
    torch.manual_seed(17)
    pair, hyper = build_modules(1)
    pair.to(dev())
    hyper.to(dev())
    B, N = 37, 11
    h = torch.randn(B, N, 64, device=dev())
    corr = ops.affinity(h)
    cases = []
    for s in (3, 11):
        (H,) = ops.topk_incidence(corr, [s])
        cases.append((hyper, H, False))
    Hd = torch.randint(0, 3, (B, 7, N), device=dev()).float()   # dense rows with weight-2 entries
    cases.append((hyper, Hd, False))
    cases += [(pair, None, False), (pair, None, True)]
    items_a, items_b = [], []
    for mod, H, sym in cases:
        K = mod.edge_types
        agg = mod.edge_aggregation_list[0]
        eo = ops.agg_gather(h, H, sym)
        ef = torch.rand(B, eo.shape[1], K, device=dev())
        items_a.append((eo, ef, agg._packed(), K))
        items_b.append((ops.GatherSpec(h, H, sym), ef, agg._packed(), K))
    fa = ops.agg_mlp_grouped(items_a)
    fb = ops.agg_mlp_grouped(items_b)
    for a, b in zip(fa, fb):
        assert torch.equal(a, b)


def test_fused_scatter_equals_standalone_scatter():
    """x == NULL form of the 128-wide MLP kernel (scatter in the prologue) == scatter kernel + MLP,
    in the split (few row blocks) and the whole (many row blocks) form of the kernel."""
    from groupnet_amd import ops
    torch.manual_seed(23)
    pair, hyper = build_modules(1)
    pair.to(dev())
    hyper.to(dev())
    for B in (9, 1500):     # 4 / 516 row blocks: split and whole kernel
        N = 11
        h = torch.randn(B, N, 64, device=dev())
        corr = ops.affinity(h)
        (H3,) = ops.topk_incidence(corr, [3])
        (H11,) = ops.topk_incidence(corr, [11])
        cases = [(hyper, H3, False), (hyper, H11, False), (pair, None, True), (pair, None, False)]
        a_items, b_items = [], []
        for mod, H, sym in cases:
            E = ops._edge_count(H, B, N, sym)
            feat = torch.randn(B, E, 64, device=dev())
            pk = mod._packed_mlp2(mod.nmp_mlp_end)
            a_items.append((ops.agg_scatter(feat, H, h, sym=sym), pk, None))
            b_items.append((ops.ScatterSpec(feat, H, h, sym), pk, None))
        ya = ops.mlp2_grouped(a_items)
        yb = ops.mlp2_grouped(b_items)
        for a, b in zip(ya, yb):
            assert maxerr(a, b) <= 1e-6


@pytest.mark.parametrize("B,N", [(37, 11), (3, 50), (5, 1), (700, 4)])
def test_pairwise_node_level_first_layer_equals_typed_mlp(B, N):
    """gn_node_linear_f32 + the pair form of gn_agg_mlp_f32 == gather + typed MLP on pair rows."""
    from groupnet_amd import ops
    torch.manual_seed(41 + N)
    pair, _ = build_modules(1)
    pair.to(dev())
    agg = pair.edge_aggregation_list[0]
    K = pair.edge_types
    pk = agg._packed()
    h = torch.randn(B, N, 64, device=dev())
    P = ops.pair_count(N)
    ef = torch.rand(B, P, K, device=dev())
    want = ops.agg_mlp(ops.agg_gather(h, None, sym=True), ef, pk, K)
    A = ops.node_linear(h, pk["W1cat"], pk["b1half"], K * 128)
    # A == W1 ori + b1/2 for every type
    W1 = torch.cat([m.layers[0].weight for m in agg.agg_mlp], 0)
    b1 = torch.cat([m.layers[0].bias for m in agg.agg_mlp], 0)
    assert maxerr(A, (h @ W1.t() + 0.5 * b1).detach()) <= 1e-5
    (got,) = ops.agg_mlp_grouped([(ops.PairSpec(A), ef, pk, K)])
    assert got.shape == (B, P, 64)
    assert maxerr(got, want) <= 1e-5


def test_hyper_module_large_n_256():
    """BASELINE config 5 shape (N=256, scales {2,8,32,128}), small B: the hyper module against the
    oracle (decomposed attention — the reference's (B,E,N,128) tensor would be 8.6 GB per scene for the
    pairwise module, SURVEY §7, so only the hyper module has an oracle at this N)."""
    torch.manual_seed(256)
    _, hyper = build_modules(1)
    sh = {k: v.detach().clone() for k, v in hyper.state_dict().items()}
    hyper.to(dev()).eval()
    B, N = 2, 256
    h = torch.randn(B, N, 64)
    corr = O.affinity(h)
    with torch.no_grad():
        for s in (2, 8, 32, 128):
            U = [torch.rand(x) for x in O.noise_shapes(B, N, s, 1)]
            nf_o, fac_o, H_o = O.ms_hgnn_hyper_forward(sh, h, corr, s, U, 1, decomposed=True)
            hyper.scale = s
            nf, fac, H = hyper(h.to(dev()), corr.to(dev()), noise_u=[u.to(dev()) for u in U])
            assert torch.equal(H.cpu(), H_o)
            assert maxerr(nf, nf_o) <= TOL and maxerr(fac, fac_o) <= TOL


def test_empty_batch_returns_empty_tensors():
    """B = 0 (the reference's torch ops accept it): same shapes, no launch."""
    pair, hyper = build_modules(1)
    pair.to(dev())
    hyper.to(dev())
    h = torch.zeros(0, 11, 64, device=dev())
    with torch.no_grad():
        nf, fac = pair(h)
        assert nf.shape == (0, 11, 64) and fac.shape == (0, 121, 6)
        hyper.scale = 5
        nf, fac, H = hyper(h, torch.zeros(0, 11, 11, device=dev()))
        assert nf.shape == (0, 11, 64) and fac.shape == (0, 11, 10) and H.shape == (0, 11, 11)
        hyper.scale = 11
        assert hyper(h, torch.zeros(0, 11, 11, device=dev()))[2].shape == (0, 1, 11)


def test_multiscale_block_n70_standalone_aggregation_path():
    """N = 70 > 64: the engine uses the stand-alone LDS-tiled gather/scatter kernels instead of the fused
    prologues (and more nodes than lanes in the node->edge kernel); 3 scales incl. scale == N."""
    from groupnet_amd.multiscale import MultiScaleHGNN
    torch.manual_seed(70)
    scales = [2, 8, 70]
    blk = MultiScaleHGNN(scales)
    sp = {k: v.detach().clone() for k, v in blk.interaction.state_dict().items()}
    shs = [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in blk.interaction_hyper]
    blk.to(dev()).eval()
    B, N = 2, 70
    h = torch.randn(B, N, 64)
    noise = [[torch.rand(s)] for s in blk.noise_shapes(B, N)]
    with torch.no_grad():
        ref, Href, _ = O.ms_hgnn_multiscale_forward(sp, shs, scales, h, noise[0], noise[1:], decomposed=True)
        out, H = blk(h.to(dev()), noise_u=[[u.to(dev()) for u in n] for n in noise])
    assert torch.equal(H.cpu(), Href) and maxerr(out, ref) <= TOL


def test_multiscale_block_n256_matches_oracle():
    """BASELINE config 5's shape: N=256, scales {2,8,32,128}, pairwise module included (32 896 pair rows per scene,
    E = 65 536 ordered edges).  The pairwise module is pinned by the slab-wise oracle
    (`ms_hgnn_pairwise_forward_chunked`, itself checked against the golden-pinned oracle on the CPU), the hyper
    modules by the ordinary oracle; banded affinity / banded top-k / stand-alone gather-scatter paths (N > 64)."""
    from groupnet_amd.multiscale import MultiScaleHGNN
    torch.manual_seed(5)
    scales = [2, 8, 32, 128]
    blk = MultiScaleHGNN(scales)
    sp = {k: v.detach().clone() for k, v in blk.interaction.state_dict().items()}
    shs = [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in blk.interaction_hyper]
    blk.to(dev()).eval()
    B, N = 1, 256
    h = torch.randn(B, N, 64)
    noise = [[torch.rand(s)] for s in blk.noise_shapes(B, N)]
    with torch.no_grad():
        out, H = blk(h.to(dev()), noise_u=[[u.to(dev()) for u in n] for n in noise])
        _, fac = blk.interaction(h.to(dev()), noise_u=[noise[0][0].to(dev())])
        corr = O.affinity(h)
        ref_pair, ref_fac = O.ms_hgnn_pairwise_forward_chunked(sp, h, noise[0], slab=4096)
        refs, Hs = [], []
        for st, s, U in zip(shs, scales, noise[1:]):
            nf, _, Hr = O.ms_hgnn_hyper_forward(st, h, corr, s, U, decomposed=True)
            refs.append(nf)
            Hs.append(Hr)
    assert out.shape == (B, N, 64 * 6) and H.shape == (B, 4 * N, N)
    assert torch.equal(out[..., :64].cpu(), h)
    # incidence: exact wherever the k-th / (k+1)-th affinity gap exceeds float rounding (ties: unspecified in torch.topk)
    Href = torch.cat(Hs, dim=1)
    srt = torch.sort(corr, dim=-1, descending=True).values
    for i, s in enumerate(scales):
        ok = (srt[..., s - 1] - srt[..., s]) > 1e-6
        assert float(ok.float().mean()) > 0.99
        assert torch.equal(H[:, i * N:(i + 1) * N].cpu()[ok], Href[:, i * N:(i + 1) * N][ok])
        assert bool((H[:, i * N:(i + 1) * N].sum(-1) == s).all())
    e_pair = maxerr(out[..., 64:128], ref_pair)
    e_fac = maxerr(fac, ref_fac)
    e_hyp = [maxerr(out[..., 64 * (2 + i):64 * (3 + i)], r) for i, r in enumerate(refs)]
    print(f"\nN=256: pairwise module vs slab-wise oracle {e_pair:.1e} (factors {e_fac:.1e}); hyper modules {['%.1e' % e for e in e_hyp]}")
    assert e_pair <= TOL and e_fac <= TOL and max(e_hyp) <= TOL


def test_config5_per_gpu_share_properties():
    """Config 5's per-GPU share at full size (B = 256 / 8 = 32 scenes, N = 256, scales {2,8,32,128}), device noise:
    shapes, finiteness, incidence row sums, f copied through, bit-exact scene-permutation equivariance of the
    incidence and of the features (per-scene math, same launch shapes)."""
    from groupnet_amd.multiscale import MultiScaleHGNN
    from groupnet_amd import ops
    torch.manual_seed(6)
    scales = [2, 8, 32, 128]
    blk = MultiScaleHGNN(scales).to(dev()).eval()
    B, N = 32, 256
    h = torch.randn(B, N, 64, device=dev())
    shapes = blk.noise_shapes(B, N)
    U = [[ops.philox_uniform(s, 11, 1000 * i, dev())] for i, s in enumerate(shapes)]
    perm = torch.randperm(B, device=dev())
    with torch.no_grad():
        out, H = blk(h, noise_u=U)
        out2, H2 = blk(h[perm].contiguous(), noise_u=[[u[0][perm].contiguous()] for u in U])
    assert out.shape == (B, N, 64 * 6) and H.shape == (B, 4 * N, N)
    assert bool(torch.isfinite(out).all()) and torch.equal(out[..., :64], h)
    for i, s in enumerate(scales):
        assert bool((H[:, i * N:(i + 1) * N].sum(-1) == s).all())
    assert torch.equal(H2, H[perm]) and torch.equal(out2, out[perm])


def test_topk_heavy_ties_against_c_oracle():
    """Index work must be bit-exact: small-integer affinities (many exact ties, some NaN) through the HIP
    rank kernel vs the plain-C arg-max oracle (oracle/topk_incidence.c), fused and stand-alone form."""
    from groupnet_amd import ops
    from test_oracle_golden import _c_oracle, _c_topk
    lib = _c_oracle()
    rng = np.random.default_rng(1)
    for B, N in ((7, 11), (3, 64), (2, 129)):
        corr = rng.integers(0, 3, size=(B, N, N)).astype(np.float32)
        corr[0, 0, 1] = np.nan
        scales = [1, 2, N // 2, N - 1, N]
        Hs = ops.topk_incidence(to_dev(corr), scales)
        for s, H in zip(scales, Hs):
            rc, want = _c_topk(lib, corr, s)
            assert rc == 0 and np.array_equal(H.cpu().numpy(), want), (B, N, s)


# ---- exhaustive hyperedge search (init_adj_attention_listall) -------------------------
@pytest.mark.parametrize("name", ["n11_b6", "n7_b3", "n13_b2"])
def test_listall_matches_reference_goldens(name):
    from groupnet_amd import ops
    import numpy as np
    c = np.load(os.path.join(os.path.dirname(__file__), "golden", f"listall_{name}.npz"))
    corr = torch.from_numpy(c["corr"]).to(dev())
    for s in c["scales"]:
        H = ops.listall_incidence(corr, int(s))
        assert torch.equal(H.cpu(), torch.from_numpy(c[f"H_s{int(s)}"])), (name, int(s))


@pytest.mark.parametrize("B,N,s", [(5, 11, 1), (4, 11, 10), (3, 12, 7), (2, 16, 8), (70, 11, 5), (3, 2, 1), (2, 20, 3)])
def test_listall_matches_oracle(B, N, s):
    """Scales the reference cannot even tabulate (torch.combinations with r = 9) and larger candidate
    counts (C(15,7) = 6435), against the oracle's itertools statement of the same search."""
    from groupnet_amd import ops
    g = torch.Generator().manual_seed(1000 + 17 * N + s)
    h = torch.randn(B, N, 64, generator=g)
    corr = O.affinity(h)
    H = ops.listall_incidence(corr.to(dev()), s)
    assert torch.equal(H.cpu(), O.listall_incidence(corr, s))


def test_listall_ties_first_candidate_wins():
    """Small-integer affinities make every group total exact in fp32 whatever the summation order, so ties
    are real ties: the kernel must resolve them like torch.max over the candidate axis (first maximum)."""
    from groupnet_amd import ops
    g = torch.Generator().manual_seed(77)
    corr = torch.randint(0, 3, (6, 9, 9), generator=g).float()
    corr = corr + corr.transpose(1, 2)
    for s in (2, 3, 5, 8):
        H = ops.listall_incidence(corr.to(dev()), s)
        assert torch.equal(H.cpu(), O.listall_incidence(corr, s)), s
    ones = torch.ones(2, 7, 7)
    H = ops.listall_incidence(ones.to(dev()), 3).cpu()      # all tied: candidate 0 = i plus the two lowest others
    for i in range(7):
        want = sorted([i] + [a for a in range(7) if a != i][:2])
        assert torch.nonzero(H[0, i]).flatten().tolist() == want


def test_hyper_module_with_listall_builder():
    """forward with `listall` set (model/MS_HGNN_batch.py:420-421) = the oracle forward on the oracle's
    exhaustive-search incidence; scale > N is refused as by the reference's table builder."""
    from groupnet_amd import ops
    torch.manual_seed(41)
    hyper = build_modules(1)[1]
    hyper.listall = True
    hyper.scale = 4
    state = {k: v.detach().cpu().clone() for k, v in hyper.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    h = torch.randn(7, 11, 64, generator=g)
    corr = O.affinity(h)
    U = [torch.rand(7, 11, 10, generator=g)]
    H_ref = O.listall_incidence(corr, 4)
    nf_ref, fac_ref = O._message_passing(state, h, H_ref, U, 1, False, None)
    hyper.to(dev())
    with torch.no_grad():
        nf, fac, H = hyper(h.to(dev()), corr.to(dev()), noise_u=[u.to(dev()) for u in U])
    assert torch.equal(H.cpu(), H_ref)
    assert float((nf.cpu() - nf_ref).abs().max()) <= 1e-5
    assert float((fac.cpu() - fac_ref).abs().max()) <= 1e-5
    with pytest.raises(RuntimeError):
        ops.listall_incidence(corr.to(dev()), 12)


def test_pack_plan_equals_matrix_by_matrix_packing():
    """The one-launch refresh of a module's packed weights (`ops.PackPlan`, gn_pack_segments_f32) writes
    bit for bit what packing matrix by matrix (gn_pack_linear_f32 + concatenation) produces — for the node,
    edge, typed-aggregation and closing MLP streams — and follows in-place parameter updates."""
    from groupnet_amd import ops
    torch.manual_seed(17)
    pair, hyper = build_modules(2)
    for m in (pair.to(dev()), hyper.to(dev())):
        K = m.edge_types
        for rnd in range(2):
            s0, s1 = m.node2edge_start_mlp[1].layers
            a0, a1 = m.attention_mlp[1].layers
            pk = m._packed_n2e(1)
            Wpq = torch.cat((a0.weight[:, :64], a0.weight[:, 64:]), 0).detach().contiguous()
            bpq = torch.cat((a0.bias, torch.zeros_like(a0.bias)), 0).detach()
            assert torch.equal(pk["W"], ops.pack_stream([s0.weight, s1.weight, Wpq]))
            assert torch.equal(pk["bias"], ops.bias_stream([s0.bias, s1.bias, bpq]))
            st = m.nmp_mlps[1]
            i0, i1 = st.init_MLP.layers
            d0, d1 = st.MLP_distribution.layers
            f0, f1 = st.MLP_factor.layers
            Wd1 = torch.zeros(32, 256, device=dev())
            Wd1[:K, :128] = d1.weight.detach()
            Wd1[K, 128:] = f1.weight.detach()[0]
            bd1 = torch.zeros(32, device=dev())
            bd1[:K] = d1.bias.detach()
            bd1[K] = f1.bias.detach()[0]
            pk = st._packed()
            assert torch.equal(pk["W"], ops.edge_stream(i0.weight, i1.weight, torch.cat((d0.weight, f0.weight), 0).detach(), Wd1))
            assert torch.equal(pk["bias"], ops.bias_stream([i0.bias, i1.bias, torch.cat((d0.bias, f0.bias), 0), bd1]))
            agg = m.edge_aggregation_list[0]
            l0 = [x.layers[0] for x in agg.agg_mlp]
            l1 = [x.layers[1] for x in agg.agg_mlp]
            pk = agg._packed()
            assert torch.equal(pk["W"], ops.pack_stream([w for a, b in zip(l0, l1) for w in (a.weight, b.weight)]))
            assert torch.equal(pk["b1"], torch.stack([l.bias.detach() for l in l0]))
            assert torch.equal(pk["b2"], torch.stack([l.bias.detach() for l in l1]))
            assert torch.equal(pk["W1cat"], ops.pack_linear(torch.cat([l.weight.detach() for l in l0], 0).contiguous()))
            assert torch.equal(pk["b1half"], torch.cat([l.bias.detach() for l in l0]) * 0.5)
            w2t = [ops.pack_linear(l.weight.detach().contiguous()).view(2, 4, 4, 256).permute(1, 0, 2, 3).reshape(-1) for l in l1]
            assert torch.equal(pk["W2t"], torch.cat(w2t))
            e0, e1 = m.nmp_mlp_end.layers
            pk = m._packed_mlp2(m.nmp_mlp_end)
            assert torch.equal(pk["W"], ops.pack_stream([e0.weight, e1.weight]))
            assert torch.equal(pk["bias"], ops.bias_stream([e0.bias, e1.bias]))
            with torch.no_grad():          # in-place update (an optimizer step): the next access re-packs
                for p in m.parameters():
                    p.add_(torch.randn_like(p) * 0.1)


def test_mlp_standalone_on_the_hip_gemm():
    """`MLP.forward` on its own (model/MS_HGNN_batch.py:220-229: Linear, activation between layers, optional
    final sigmoid) against plain torch fp32 layer math on the same device, forward and gradients; odd widths
    take the ragged GEMM kernel."""
    from groupnet_amd import MLP
    torch.manual_seed(4)
    for kw, shape in ((dict(input_dim=8, output_dim=3, hidden_size=(16, 5)), (37, 8)),
                      (dict(input_dim=128, output_dim=64, hidden_size=(128,)), (5, 11, 128)),
                      (dict(input_dim=20, output_dim=1, hidden_size=(32,), discrim=True, activation='sigmoid'), (9, 20))):
        m = MLP(**kw).to(dev())
        x = torch.randn(*shape, device=dev(), requires_grad=True)
        y = m(x)
        z = x
        for i, l in enumerate(m.layers):
            z = torch.nn.functional.linear(z, l.weight, l.bias)
            if i != len(m.layers) - 1:
                z = m.activation(z)
            elif m.sigmoid is not None:
                z = torch.sigmoid(z)
        assert float((y - z).abs().max()) <= 1e-5 * max(1.0, float(z.abs().max()))
        R = torch.randn_like(y)
        gx, *gw = torch.autograd.grad((y * R).sum(), [x, *m.parameters()])
        rx, *rw = torch.autograd.grad((z * R).sum(), [x, *m.parameters()])
        for a, b in zip([gx, *gw], [rx, *rw]):
            assert float((a - b).abs().max()) <= 1e-4 * max(1e-3, float(b.abs().max()))


def test_full_membership_softmax_with_large_negative_logits():
    """softmax(att*H) over all N nodes: when every node is a member (scale == N, or the pairwise graph at
    N <= 2) there is no exp(0) term, and with strongly negative logits exp(0 - max) overflows — the kernels must
    drop the term rather than compute 0 * inf.  (Found by the training-mode PastEncoder golden, whose
    un-normalised embeddings drive the attention logits to about -150.)"""
    from groupnet_amd import ops
    torch.manual_seed(8)
    pair, hyper = build_modules(1)
    for N, s in ((11, 11), (2, 2), (1, 1)):
        hyper.scale = s
        for m in (pair, hyper):
            m.cpu()
            with torch.no_grad():           # strongly negative attention logits for every node
                m.attention_mlp[0].layers[1].bias.fill_(-400.0)
        sp = {k: v.detach().clone() for k, v in pair.state_dict().items()}
        sh = {k: v.detach().clone() for k, v in hyper.state_dict().items()}
        h = torch.randn(3, N, 64)
        corr = O.affinity(h)
        Up = [torch.rand(x) for x in O.noise_shapes(3, N, None)]
        Uh = [torch.rand(x) for x in O.noise_shapes(3, N, s)]
        nf_p, fac_p = O.ms_hgnn_pairwise_forward(sp, h, Up, decomposed=True)
        nf_h, fac_h, H = O.ms_hgnn_hyper_forward(sh, h, corr, s, Uh, decomposed=True)
        assert bool(torch.isfinite(nf_p).all()) and bool(torch.isfinite(nf_h).all())
        pair.to(dev()), hyper.to(dev())
        with torch.no_grad():
            a, b = pair(h.to(dev()), noise_u=[u.to(dev()) for u in Up])
            c, d, H2 = hyper(h.to(dev()), corr.to(dev()), noise_u=[u.to(dev()) for u in Uh])
        assert maxerr(a, nf_p.numpy()) <= TOL and maxerr(b, fac_p.numpy()) <= TOL, N
        assert maxerr(c, nf_h.numpy()) <= TOL and maxerr(d, fac_h.numpy()) <= TOL, N
        # and the backward of the same rows stays finite
        for m in (pair, hyper):
            m.train()
        x = h.to(dev()).requires_grad_(True)
        (pair(x, noise_u=[u.to(dev()) for u in Up])[0].sum() + hyper(x, corr.to(dev()), noise_u=[u.to(dev()) for u in Uh])[0].sum()).backward()
        assert bool(torch.isfinite(x.grad).all())
        assert all(bool(torch.isfinite(p.grad).all()) for m in (pair, hyper) for p in m.parameters() if p.grad is not None)
        for m in (pair, hyper):
            m.eval()


@pytest.mark.parametrize("mode", ["f16x3", "bf16x6"])
@pytest.mark.parametrize("in_scale,w_scale", [(1e-3, 1.0), (30.0, 1.0), (1.0, 12.0), (30.0, 6.0), (200.0, 3.0)])
def test_modules_under_extreme_magnitudes(in_scale, w_scale, mode):
    """Un-normalised inputs and sharp attention / distribution heads (saturated softmax, sigmoid and ReLU
    regimes, logits in the hundreds, one all-zero agent): the HIP modules against the oracle, relative to the
    output scale; nothing may turn into inf / NaN.  Probabilities: a logit of magnitude 10^3 carries 1e-4 of fp32
    rounding, which the exponential turns into that much probability — the fp32 oracle ITSELF is that far from its
    float64 evaluation (measured 0.6e-4 ... 1.0e-4 at in_scale 30 / w_scale 6, depending on the host's BLAS).  The
    gate is therefore stated against the truth, the FLOAT64 oracle: the six-product bf16 path (24-bit operands) within
    max(1e-4, 3 x the fp32 oracle's own distance); the two-part fp16 path, whose products carry 22 significant bits
    (hi + lo = 11 + 11) instead of fp32's 24, within 4 x that (measured 1.8e-4 where the fp32 oracle is 0.6e-4 off:
    logits of ~3e3 x 2.4e-7 / (4 tau)).  Features keep the same relative gate on both paths."""
    from groupnet_amd import ops
    prev = ops.precision()
    ops.set_precision(mode)
    try:
        _extreme_magnitudes(in_scale, w_scale, 4.0 if mode == "f16x3" else 1.0)
    finally:
        ops.set_precision(prev)


def _extreme_magnitudes(in_scale, w_scale, slack):
    torch.manual_seed(int(in_scale * 7 + w_scale))
    pair, hyper = build_modules(1)
    with torch.no_grad():
        for m in (pair, hyper):
            for n_, p in m.named_parameters():
                if "attention_mlp" in n_ or "MLP_distribution" in n_ or "MLP_factor" in n_:
                    p.mul_(w_scale)
    B, N = 6, 11
    h = torch.randn(B, N, 64) * in_scale
    h[1, 3] = 0.0        # a zero feature row: its affinities are all exactly 0, i.e. its top-k is one big tie, whose
    #                      order is implementation-defined — the incidence is therefore taken from the oracle
    corr = O.affinity(h)
    sp = {k: v.detach().clone() for k, v in pair.state_dict().items()}
    sh = {k: v.detach().clone() for k, v in hyper.state_dict().items()}
    pair.to(dev()), hyper.to(dev())
    Up = [torch.rand(x) for x in O.noise_shapes(B, N, None)]
    nf_p, fac_p = O.ms_hgnn_pairwise_forward(sp, h, Up, decomposed=True)
    d64 = lambda st: {k: v.double() for k, v in st.items()}
    _, fac_p64 = O.ms_hgnn_pairwise_forward(d64(sp), h.double(), [u.double() for u in Up], decomposed=True)

    def prob_gate(got, ref32, ref64):
        own = float((ref32.double() - ref64).abs().max())               # what fp32 rounding alone does to the oracle
        err = float((got.detach().cpu().double() - ref64).abs().max())
        print(f"probabilities: HIP vs float64 oracle {err:.2e}, fp32 oracle vs float64 {own:.2e}")
        return err <= slack * max(1e-4, 3.0 * own)

    with torch.no_grad():
        a, b = pair(h.to(dev()), noise_u=[u.to(dev()) for u in Up])
    assert bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all())
    assert maxerr(a, nf_p.numpy()) <= 2e-5 * max(1.0, float(nf_p.abs().max()))
    assert prob_gate(b, fac_p, fac_p64)
    for s in (2, 5, 11):
        hyper.scale = s
        Uh = [torch.rand(x) for x in O.noise_shapes(B, N, s)]
        nf_h, fac_h, H = O.ms_hgnn_hyper_forward(sh, h, corr, s, Uh, decomposed=True)
        _, fac_h64, _ = O.ms_hgnn_hyper_forward(d64(sh), h.double(), corr.double(), s, [u.double() for u in Uh],
                                                decomposed=True)
        with torch.no_grad():
            c, d, H2 = hyper(h.to(dev()), corr.to(dev()), noise_u=[u.to(dev()) for u in Uh], H=H.to(dev()))
            H3 = hyper.init_adj_attention(h.to(dev()), corr.to(dev()), scale_factor=s).cpu()
        tied = torch.zeros(B, N, dtype=torch.bool)
        tied[1, 3] = True
        if s != N:
            assert torch.equal(H3[~tied], H[~tied])          # every row but the all-tied one
            assert float(H3[1, 3].sum()) == float(s)
        assert bool(torch.isfinite(c).all()) and bool(torch.isfinite(d).all())
        assert maxerr(c, nf_h.numpy()) <= 2e-5 * max(1.0, float(nf_h.abs().max())), s
        assert prob_gate(d, fac_h, fac_h64), s


def test_pair_form_on_the_bf16_cores_is_fp32_accurate():
    """The pair form of the typed aggregation at a size where its workgroups stage node rows in LDS (B=512)
    forms layer 2 from bf16 part-products (x = x1+x2+x3, six products): against the fp32-MFMA pair form
    (`ops.BF16X6 = False`) and against the ordered-edge oracle-checked form — fp32 accuracy, not bf16's."""
    from groupnet_amd import ops
    torch.manual_seed(31)
    pair, _ = build_modules(1)
    pair.to(dev())
    B, N, K = 512, 11, 6
    ori = torch.randn(B, N, 64, device=dev()) * 2.0
    ef = torch.rand(B, ops.pair_count(N), K, device=dev())
    pk = pair.edge_aggregation_list[0]._packed()
    A = ops.node_linear(ori, pk["W1cat"], pk["b1half"], K * 128)
    assert ops.BF16X6
    got = ops.agg_mlp_grouped([(ops.PairSpec(A), ef, pk, K)])[0]
    ops.BF16X6 = False
    try:
        ref = ops.agg_mlp_grouped([(ops.PairSpec(A), ef, pk, K)])[0]
    finally:
        ops.BF16X6 = True
    eo = ops.agg_gather(ori, None, sym=True)
    plain = ops.agg_mlp(eo, ef, pk, K)                 # both layers on the fp32 matrix cores, explicit eo
    scale = float(plain.abs().max())
    assert float((got - ref).abs().max()) <= 2e-6 * scale
    assert float((got - plain).abs().max()) <= 4e-6 * scale


def test_bf16_core_paths_match_fp32_mfma_paths_at_full_size():
    """B=512, N=11, scales {2,5,11}: the whole multiscale forward with the edge and aggregation kernels on the
    bf16 cores (three-part split, six products) against the same forward with every kernel on the fp32 matrix
    cores (`ops.BF16X6 = False`), same noise: identical incidence, features within 2e-6 of the feature scale."""
    from groupnet_amd import ops
    from groupnet_amd.multiscale import MultiScaleHGNN
    torch.manual_seed(77)
    B, N = 512, 11
    blk = MultiScaleHGNN([2, 5, 11]).to(dev()).eval()
    f = torch.randn(B, N, 64, device=dev())
    noise = [[torch.rand(s, device=dev())] for s in blk.noise_shapes(B, N)]
    with torch.no_grad():
        assert ops.BF16X6
        a, Ha = blk(f, noise_u=noise)
        ops.BF16X6 = False
        try:
            b, Hb = blk(f, noise_u=noise)
        finally:
            ops.BF16X6 = True
    assert torch.equal(Ha, Hb)
    assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max()))


def test_fused_node2edge_pooling_equals_the_node2edge_launch(monkeypatch):
    """ops.PoolSpec: the edge kernel forms the pooled edge rows itself (pairwise graph by default; hyper modules when
    ops.POOL_MAX_N allows) — same outputs as with the node2edge launch and an `edges` tensor in HBM (the 32 attention
    channels are summed in two halves instead of one chain: 1e-6, incidence and factors included)."""
    import groupnet_amd.MS_HGNN_batch as M
    from groupnet_amd import ops
    from groupnet_amd.multiscale import MultiScaleHGNN
    torch.manual_seed(21)
    B, N, scales = 7, 11, [2, 5, 11]
    blk = MultiScaleHGNN(scales).to(dev()).eval()
    f = torch.randn(B, N, 64, device=dev())
    U = [[torch.rand(s, device=dev())] for s in blk.noise_shapes(B, N)]
    outs = {}
    with torch.no_grad():
        for tag, fuse, maxn in (("launch", False, 0), ("pair", True, 0), ("all", True, 16)):
            monkeypatch.setattr(M, "_FUSE_POOL", fuse)
            monkeypatch.setattr(ops, "POOL_MAX_N", maxn)
            outs[tag] = blk(f, noise_u=U)
        # the modules' own forward (factors) through the fused path
        monkeypatch.setattr(M, "_FUSE_POOL", True)
        _, fac_f = blk.interaction(f, noise_u=U[0])
        monkeypatch.setattr(M, "_FUSE_POOL", False)
        _, fac_u = blk.interaction(f, noise_u=U[0])
    ref = outs["launch"]
    scale = float(ref[0].abs().max())
    for tag in ("pair", "all"):
        err = float((outs[tag][0] - ref[0]).abs().max()) / scale
        print(f"\nfused pooling ({tag}) vs node2edge launch: max rel diff {err:.2e}")
        assert err <= 1e-6 and torch.equal(outs[tag][1], ref[1])
    assert float((fac_f - fac_u).abs().max()) <= 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N,scales,nmp", [(7, 11, [2, 5, 11], 1), (3, 6, [2, 6], 2), (33, 5, [3], 1)])
def test_mlp2_xs_kernel_equals_the_one_wave_per_block_kernel(B, N, scales, nmp, dtype, monkeypatch):
    """mlp2_xs_kernel (4 waves share a row block: hidden tiles dealt over the waves, partial sums through LDS; fused
    scatter accumulated in line layout) against mlp2_x_kernel on the same block, inputs and uniforms — pairwise and hyper
    groups, the plain-input form (nmp_layers = 2: the MLP between the rounds), ragged last row block: identical
    incidence, features equal up to the summation order of the four partial sums (fp32: 2e-6 relative; bf16 storage:
    one rounding of the stored result, 1.6e-2)."""
    from groupnet_amd.multiscale import MultiScaleHGNN
    torch.manual_seed(5)
    blk = MultiScaleHGNN(scales, nmp_layers=nmp).to(dev()).eval()
    f = torch.randn(B, N, 64, device=dev()).to(dtype)
    U = [[torch.rand(s, device=dev()) for _ in range(nmp)] for s in blk.noise_shapes(B, N)]
    outs = {}
    with torch.no_grad():
        for xs in ("0", "1"):
            monkeypatch.setenv("GN_MLP2_XS", xs)
            outs[xs] = blk(f, noise_u=U)
    a, b = outs["0"][0].float(), outs["1"][0].float()
    err = float((a - b).abs().max()) / max(1.0, float(a.abs().max()))
    print(f"\nmlp2 xs vs one-wave-per-block kernel ({dtype}, B={B} N={N} nmp={nmp}): max rel diff {err:.2e}")
    assert torch.equal(outs["0"][1], outs["1"][1])
    assert torch.isfinite(b).all()
    assert err <= (2e-6 if dtype == torch.float32 else 1.6e-2)


@pytest.mark.parametrize("dtype,B,N,scales", [(torch.float32, 7, 11, [2, 5, 11]), (torch.float32, 3, 5, [2]),
                                               (torch.bfloat16, 3, 50, [4, 16]), (torch.float32, 2, 70, [8])])
def test_staged_pairwise_pooling_is_bit_identical(dtype, B, N, scales, monkeypatch):
    """The fused node->edge pooling of the pairwise graph reads the scenes' x' / pq rows from an LDS stage (the workgroup
    copies them once, coalesced) or, GN_POOL_STAGE=0 / rows that do not fit, per lane from L2: same arithmetic in the
    same order, so every output is identical bit for bit (N=50 bf16: the two-row-blocks-per-wave kernel and a workgroup
    spanning two scenes; N=70: 2485 pairs per scene)."""
    from groupnet_amd.multiscale import MultiScaleHGNN
    torch.manual_seed(9)
    blk = MultiScaleHGNN(scales).to(dev()).eval()
    f = torch.randn(B, N, 64, device=dev()).to(dtype)
    U = [[torch.rand(s, device=dev())] for s in blk.noise_shapes(B, N)]
    outs = {}
    with torch.no_grad():
        for st in ("0", "1"):
            monkeypatch.setenv("GN_POOL_STAGE", st)
            if dtype == torch.bfloat16:
                monkeypatch.setenv("GN_EDGE_RB2", "1")
                monkeypatch.setenv("GN_AGG_RB2", "1")     # (its pairwise gather ori_i + ori_j has the same stage)
            outs[st] = blk(f, noise_u=U)
            _, fac = blk.interaction(f, noise_u=U[0])
            outs[st] = (*outs[st], fac)
    for a, b in zip(outs["0"], outs["1"]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N,scales", [(5, 11, [2, 5, 11]), (3, 50, [2, 4, 8, 16]), (4, 30, [20, 30]), (70, 3, [1, 2]),
                                        (2, 64, [7])])
def test_node2edge_row_form_equals_the_banded_form(B, N, scales, dtype, monkeypatch):
    """The hyper modules' node->edge pooling with one lane pair per hyperedge (what large launches run; GN_N2E_ROWS=1
    forces it) against the banded form on the same x', pq and incidences — incl. rows with more than 16 members (the
    generic path: running max / sum), scale == N (one full hyperedge), a mix of groups in one launch, ragged last
    workgroups, and hand-made rows: empty, and weights other than 1.  fp32: 2e-6 of the output scale (different summation
    order of the softmax); bf16 storage: one rounding of the stored result."""
    from groupnet_amd import ops
    torch.manual_seed(17)
    items = []
    for s in scales:
        xp = torch.randn(B, N, 64, device=dev()).to(dtype)
        pq = torch.randn(B, N, 64, device=dev()).to(dtype)
        corr = torch.rand(B, N, N, device=dev())
        H = ops.topk_incidence(corr, [s])[0].clone()
        if s != N and B > 1:
            H[0, 0, :] = 0.0                          # an empty hyperedge
            H[1, min(1, H.shape[1] - 1), :] *= 1.5    # weights other than 0 / 1
        w2 = torch.randn(32, device=dev())
        b2 = torch.randn(1, device=dev())
        items.append((xp, pq, H, w2, b2))
    outs = {}
    for v in ("0", "1"):
        monkeypatch.setenv("GN_N2E_ROWS", v)
        outs[v] = [e.float() for e in ops.node2edge_grouped(items)]
    for a, b, s in zip(outs["0"], outs["1"], scales):
        assert a.shape == b.shape and torch.isfinite(b).all()
        err = float((a - b).abs().max()) / max(1.0, float(a.abs().max()))
        print(f"\nnode2edge rows vs banded ({dtype}, B={B} N={N} scale={s}): max rel diff {err:.2e}")
        assert err <= (2e-6 if dtype == torch.float32 else 8e-3)


# ---------------------------------------------------------------------------------------------
# the two-part fp16 path ("f16x3") of the fp32 entry points: accuracy, range vote, fallback
# ---------------------------------------------------------------------------------------------
def _with_precision(mode, fn):
    from groupnet_amd import ops
    old = ops.precision()
    ops.set_precision(mode)
    try:
        return fn()
    finally:
        ops.set_precision(old)


def _block_forward(blk, f, noise):
    with torch.no_grad():
        out, H = blk(f, noise_u=noise)
    return out.clone(), H.clone()


def test_f16x3_is_the_default_and_is_fp32_accurate():
    """B=512, N=11, scales {2,5,11}: the forward on the fp16 cores (two parts per operand, three part-products) against
    the six-product bf16 path and the fp32 matrix cores, same noise: identical incidence, features within 2e-6 of the
    feature scale of either (both are a few 1e-7 from the oracle at small sizes; see the golden tests)."""
    from groupnet_amd import ops
    from groupnet_amd.multiscale import MultiScaleHGNN
    assert ops.precision() == os.environ.get("GN_PRECISION", "f16x3").lower() or not ops.BF16X6
    torch.manual_seed(78)
    B, N = 512, 11
    blk = MultiScaleHGNN([2, 5, 11]).to(dev()).eval()
    f = torch.randn(B, N, 64, device=dev())
    noise = [[torch.rand(s, device=dev())] for s in blk.noise_shapes(B, N)]
    a, Ha = _with_precision("f16x3", lambda: _block_forward(blk, f, noise))
    b, Hb = _with_precision("bf16x6", lambda: _block_forward(blk, f, noise))
    c, Hc = _with_precision("fp32", lambda: _block_forward(blk, f, noise))
    assert torch.equal(Ha, Hb) and torch.equal(Ha, Hc)
    scale = max(1.0, float(c.abs().max()))
    e_ab, e_ac, e_bc = (float((x - y).abs().max()) for x, y in ((a, b), (a, c), (b, c)))
    print(f"f16x3 vs bf16x6 {e_ab:.2e}, f16x3 vs fp32 cores {e_ac:.2e}, bf16x6 vs fp32 cores {e_bc:.2e} (scale {scale:.2f})")
    assert e_ab <= 2e-6 * scale and e_ac <= 2e-6 * scale
    assert not torch.equal(a, b)          # (the two paths really are different code)


@pytest.mark.parametrize("in_scale", [3.0e3, 1.0e6, 1.0e12])
def test_f16x3_range_vote_falls_back_to_bf16x6(in_scale):
    """Operands beyond the fp16 range (|x| > 65504): the workgroups that meet one repeat their rows on the bf16x6 path
    inside the same launch — results finite, equal to the oracle within fp32 accuracy of the output scale, and where
    EVERY workgroup falls back bit-identical to the bf16x6 mode.  3e3: inputs fit, some hidden activations may not."""
    from groupnet_amd.multiscale import MultiScaleHGNN
    torch.manual_seed(79)
    B, N, scales = 6, 11, [2, 5, 11]
    blk = MultiScaleHGNN(scales)
    sp = {k: v.detach().clone() for k, v in blk.interaction.state_dict().items()}
    shs = [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in blk.interaction_hyper]
    blk.to(dev()).eval()
    h = torch.randn(B, N, 64) * in_scale
    noise = [[torch.rand(s)] for s in blk.noise_shapes(B, N)]
    with torch.no_grad():
        ref, Href, _ = O.ms_hgnn_multiscale_forward(sp, shs, scales, h, noise[0], noise[1:], decomposed=True)
    nd = [[u.to(dev()) for u in n] for n in noise]
    a, Ha = _with_precision("f16x3", lambda: _block_forward(blk, h.to(dev()), nd))
    b, Hb = _with_precision("bf16x6", lambda: _block_forward(blk, h.to(dev()), nd))
    assert bool(torch.isfinite(a).all())
    assert torch.equal(Ha.cpu(), Href)
    scale = max(1.0, float(ref.abs().max()))
    assert maxerr(a, ref.numpy()) <= 2e-5 * scale, (maxerr(a, ref.numpy()), scale)
    assert maxerr(b, ref.numpy()) <= 2e-5 * scale
    # (no bit-identity with the bf16x6 mode here: a workgroup whose operands all fit — e.g. hidden rows scaled by edge
    # weights that saturated to 0 — legitimately stays on the fp16 path; test_f16x3_fallback_is_the_bf16x6_path pins that)


def test_f16x3_fallback_is_the_bf16x6_path():
    """Stand-alone stages whose EVERY workgroup meets an out-of-range operand: the f16x3 launch must then produce the
    bf16x6 mode's bits (the fallback is that path, run inside the same launch), for the node stage and both closing-MLP
    kernels (4 waves per row block / one wave per row block)."""
    from groupnet_amd import ops
    torch.manual_seed(81)
    pair, _ = build_modules(1)
    pair.to(dev()).eval()
    x = torch.randn(40, 11, 64, device=dev()) * 1.0e6
    pk_end = pair._packed_mlp2(pair.nmp_mlp_end)
    pk_node = pair._packed_n2e(0)
    big = torch.cat((x, x), -1).contiguous()
    for xs in ("1", "0"):
        os.environ["GN_MLP2_XS"] = xs
        try:
            fast = _with_precision("f16x3", lambda: ops.mlp2(big, pk_end).clone())
            slow = _with_precision("bf16x6", lambda: ops.mlp2(big, pk_end).clone())
        finally:
            del os.environ["GN_MLP2_XS"]
        assert bool(torch.isfinite(fast).all()) and torch.equal(fast, slow), xs
    fn = _with_precision("f16x3", lambda: [t.clone() for t in ops.node_mlp(x, pk_node)])
    sn = _with_precision("bf16x6", lambda: [t.clone() for t in ops.node_mlp(x, pk_node)])
    assert torch.equal(fn[0], sn[0]) and torch.equal(fn[1], sn[1]) and bool(torch.isfinite(fn[1]).all())
    # in range: the two modes differ in the last bits (different part-products), i.e. the fast path really ran above
    y = torch.randn(40, 11, 128, device=dev())
    f2 = _with_precision("f16x3", lambda: ops.mlp2(y, pk_end).clone())
    s2 = _with_precision("bf16x6", lambda: ops.mlp2(y, pk_end).clone())
    assert not torch.equal(f2, s2) and float((f2 - s2).abs().max()) <= 2e-6 * max(1.0, float(s2.abs().max()))


def test_f16x3_flagged_weight_image_falls_back():
    """A weight that does not fit fp16 raises the flag word behind the fp16 image (gn_split_bf16_f32, parts = 2): every
    workgroup that walks that image then runs on the bf16x6 path — bit-identical to the bf16x6 mode, finite, and still
    matching the oracle."""
    torch.manual_seed(80)
    pair, hyper = build_modules(1, scale=5)
    with torch.no_grad():
        w = hyper.nmp_mlp_end.layers[0].weight
        w[3, 7] = 1.0e5                      # the closing MLP of the hyper module
        w2 = pair.nmp_mlp_start.init_MLP.layers[0].weight
        w2[1, 2] = -7.0e4                    # the edge MLP of the pairwise module
    sp = {k: v.detach().clone() for k, v in pair.state_dict().items()}
    sh = {k: v.detach().clone() for k, v in hyper.state_dict().items()}
    pair.to(dev()).eval(), hyper.to(dev()).eval()
    B, N = 5, 11
    h = torch.randn(B, N, 64)
    corr = O.affinity(h)
    Up = [torch.rand(x) for x in O.noise_shapes(B, N, None)]
    Uh = [torch.rand(x) for x in O.noise_shapes(B, N, 5)]
    nf_p, fac_p = O.ms_hgnn_pairwise_forward(sp, h, Up, decomposed=True)
    nf_h, fac_h, H = O.ms_hgnn_hyper_forward(sh, h, corr, 5, Uh, decomposed=True)

    def run():
        with torch.no_grad():
            a, b = pair(h.to(dev()), noise_u=[u.to(dev()) for u in Up])
            c, d, _ = hyper(h.to(dev()), corr.to(dev()), noise_u=[u.to(dev()) for u in Uh])
        return a.clone(), b.clone(), c.clone(), d.clone()

    fast = _with_precision("f16x3", run)
    for x in fast:
        assert bool(torch.isfinite(x).all())
    # the flagged stage alone, on in-range inputs: the bf16x6 mode's bits (everything upstream of it inside a module
    # runs on the fp16 path, so whole-module outputs agree only to rounding)
    from groupnet_amd import ops
    pk_end = hyper._packed_mlp2(hyper.nmp_mlp_end)
    y = torch.randn(7, 11, 128, device=dev())
    assert torch.equal(_with_precision("f16x3", lambda: ops.mlp2(y, pk_end).clone()),
                       _with_precision("bf16x6", lambda: ops.mlp2(y, pk_end).clone()))
    assert maxerr(fast[0], nf_p.numpy()) <= 2e-5 * max(1.0, float(nf_p.abs().max()))
    assert maxerr(fast[2], nf_h.numpy()) <= 2e-5 * max(1.0, float(nf_h.abs().max()))
    assert maxerr(fast[1], fac_p.numpy()) <= 1e-4 and maxerr(fast[3], fac_h.numpy()) <= 1e-4


def test_split_fp16_image_layout_and_flag():
    """gn_split_bf16_f32 with parts = 2: hi + lo reproduces every weight to 2^-22 relative (2^-25 absolute below the
    fp16 normal range), pieces in the (sub-step, part, lane, j) order of the bf16 images, flag word zero unless a
    weight exceeds the fp16 range."""
    from groupnet_amd import ops
    torch.manual_seed(5)
    packed = (torch.randn(6 * 1024, device=dev()) * torch.logspace(-6, 2, 6 * 1024, device=dev())).contiguous()
    img = ops.split_bf16(packed, parts=2)
    n = 6 * 2 * 2 * 64 * 8
    assert img.numel() == n + 8 and int(img[n:].abs().sum()) == 0
    parts = img[:n].view(torch.float16).view(6, 2, 2, 64, 8).float()          # (tile, half, part, lane, j)
    ref = ops.split_bf16(packed, parts=3).view(torch.bfloat16).view(6, 2, 3, 64, 8).float().sum(2)   # same element order
    got = parts.sum(2)
    err = (got - ref).abs()
    assert float((err / ref.abs().clamp_min(2.0 ** -3)).max()) <= 2.0 ** -21
    bad = packed.clone()
    bad[100] = 7.0e4
    img2 = ops.split_bf16(bad, parts=2)
    assert int(img2[n:].view(torch.int32)[0]) != 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N", [2, 11, 50, 64])
def test_pair_scatter_read_once_is_bit_identical(N, dtype, monkeypatch):
    """agg_scatter over unordered pairs, one workgroup per scene with the pair rows streamed once through LDS bands
    (the launcher's pick at B >= 256, N <= 64), against the direct kernel that fetches every pair row for both of its
    nodes: same members in the same order, identical bits — and against the dense definition H^T feat at small N."""
    from groupnet_amd import ops
    torch.manual_seed(N)
    B = 256
    P = ops.pair_count(N)
    feat = torch.randn(B, P, 64, device=dev()).to(dtype)
    ori = torch.randn(B, N, 64, device=dev()).to(dtype)
    got = ops.agg_scatter(feat, None, ori, sym=True)
    monkeypatch.setenv("GN_SCATTER_PAIRS", "0")
    ref = ops.agg_scatter(feat, None, ori, sym=True)
    assert torch.equal(got, ref)
    if N <= 11 and dtype == torch.float32:
        # dense check: node n sums the pair rows {n, j} for every j (the self pair once: its row already carries both
        # ordered self-loops), then cat(., ori) / N
        want = torch.zeros(B, N, 64, device=dev())
        for i in range(N):
            for j in range(i, N):
                p = i * N - i * (i - 1) // 2 + (j - i)
                want[:, i] += feat[:, p]
                if j != i:
                    want[:, j] += feat[:, p]
        assert float((got[..., :64] - want / N).abs().max()) <= 1e-5
        assert float((got[..., 64:] - ori / N).abs().max()) <= 1e-6 * float(ori.abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N,scales", [(37, 11, [2, 5, 11]), (512, 11, [2, 5, 11]), (3, 30, [4, 30])])
def test_affinity_tail_of_the_node_stage_is_bit_identical(B, N, scales, dtype):
    """The fused affinity + top-k launch riding as the tail workgroups of the first node-stage launch
    (gn_node_mlp_affinity_*, the block's default) against the two separate launches (`ops._AFFINITY_TAIL = False`):
    same code for a scene either way — features, incidence and the f copy identical, and the launch really is gone
    (the block issues one launch fewer)."""
    from groupnet_amd import ops
    from groupnet_amd.multiscale import MultiScaleHGNN
    torch.manual_seed(5)
    blk = MultiScaleHGNN(scales).to(dev()).eval()
    f = torch.randn(B, N, 64, device=dev()).to(dtype)
    noise = [[torch.rand(s, device=dev())] for s in blk.noise_shapes(B, N)]
    counts = {}
    orig = ops.AffinityTail.launch

    def counting(self):
        counts["alone"] = counts.get("alone", 0) + (0 if self.done else 1)
        return orig(self)
    ops.AffinityTail.launch = counting
    try:
        with torch.no_grad():
            a, Ha = blk(f, noise_u=noise)
            a, Ha = a.clone(), Ha.clone()
            rode = counts.get("alone", 0) == 0
            ops._AFFINITY_TAIL = False
            try:
                b, Hb = blk(f, noise_u=noise)
            finally:
                ops._AFFINITY_TAIL = True
    finally:
        ops.AffinityTail.launch = orig
    assert rode and counts.get("alone", 0) == 1          # first call rode in the node stage, second was its own launch
    assert torch.equal(a, b) and torch.equal(Ha, Hb)
    assert torch.equal(a[..., :64], f)


@pytest.mark.parametrize("knob", ["GN_XCD", "GN_AGG_HSTAGE", "GN_AGG_LINES"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_order_and_gather_knobs_are_bit_identical(knob, dtype, monkeypatch):
    """Knobs that change WHERE work runs or where operands are read from, never what is computed: GN_XCD=0 (workgroups in
    dispatch order instead of the XCD-aware permutation), GN_AGG_HSTAGE=0 (hyper gather of the typed aggregation from L2
    in line layout instead of LDS-staged rows), GN_AGG_LINES=0 (per-lane gather).  The library reads them at every
    launch (a `static` getenv froze the first value and made such A/B tests vacuous), so the two settings
    run different code; outputs must be identical bit for bit.  B = 512 so that the XCD order is active (>= 64
    workgroups) and the hyper groups run two waves per row block."""
    from groupnet_amd.multiscale import MultiScaleHGNN
    torch.manual_seed(13)
    B, N, scales = 512, 11, [2, 5, 11]
    blk = MultiScaleHGNN(scales).to(dev()).eval()
    f = torch.randn(B, N, 64, device=dev()).to(dtype)
    U = [[torch.rand(s, device=dev())] for s in blk.noise_shapes(B, N)]
    outs = {}
    with torch.no_grad():
        for v in ("1", "0"):
            monkeypatch.setenv(knob, v)
            if knob == "GN_AGG_LINES":
                monkeypatch.setenv("GN_AGG_HSTAGE", "0")      # (the line layout is what runs without the stage)
            a, H = blk(f, noise_u=U)
            outs[v] = (a.clone(), H.clone())
    assert torch.equal(outs["1"][0], outs["0"][0]) and torch.equal(outs["1"][1], outs["0"][1])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_copy_cols_equals_torch_copy(dtype):
    """ops.copy_cols (gn_copy_2d: the pitched copy that stages a rank's output columns for the all-gather) against
    torch's strided copy, incl. the fallback for shapes the kernel does not take."""
    from groupnet_amd import ops
    torch.manual_seed(2)
    full = torch.randn(37, 11, 320, device=dev()).to(dtype)
    for lo, hi in ((64, 320), (0, 64), (8, 72), (3, 67)):          # the last one is not 16-byte aligned: torch fallback
        src = full[..., lo:hi]
        dst = torch.zeros(37, 11, hi - lo, device=dev(), dtype=dtype)
        ops.copy_cols(dst, src)
        assert torch.equal(dst, src.contiguous()), (lo, hi)


# ---------------------------------------------------------------------------------------------
# node form of the pairwise typed aggregation (gn_agg_group_t.node_form): layer 2 once per node
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["f16x3", "bf16x6"])
@pytest.mark.parametrize("B,N", [(37, 11), (512, 11), (5, 1), (9, 2), (7, 16), (3, 5), (130, 13)])
def test_node_form_equals_scattered_pair_form(B, N, precision):
    """edge_aggregation.forward consumes the per-edge feature only as H^T feat (model/MS_HGNN_batch.py:267): the node
    form evaluates that sum directly, layer 2 once per node.  Against the pair form scattered by gn_agg_scatter (which the
    goldens pin), ragged row blocks, one to three scenes per row block, both matrix paths."""
    from groupnet_amd import ops
    torch.manual_seed(300 + N)
    pair, _ = build_modules(1)
    pair.to(dev())
    agg = pair.edge_aggregation_list[0]
    K = pair.edge_types
    pk = agg._packed()
    ori = torch.randn(B, N, 64, device=dev()) * 1.5
    ef = torch.rand(B, ops.pair_count(N), K, device=dev())
    A = ops.node_linear(ori, pk["W1cat"], pk["b1half"], K * 128)
    old = ops.precision()
    ops.set_precision(precision)
    try:
        (feat,) = ops.agg_mlp_grouped([(ops.PairSpec(A), ef, pk, K)])
        (node,) = ops.agg_mlp_grouped([(ops.PairSpec(A, node=True), ef, pk, K)])
    finally:
        ops.set_precision(old)
    assert node.shape == (B, N, 64)
    want = ops.agg_scatter(feat, None, ori, divisor=1.0, sym=True)[..., :64]      # H^T feat
    scale = max(1.0, float(want.abs().max()))
    err = maxerr(node, want)
    print(f"node form vs scattered pair form, B={B} N={N} {precision}: {err:.2e} (scale {scale:.2e})")
    assert err <= 2e-6 * scale


def test_node_form_feeds_the_closing_mlp_like_the_fused_scatter():
    """gn_mlp2_f32 with E = 0 (feat already per node) == the fused scatter of the pair rows."""
    from groupnet_amd import ops
    torch.manual_seed(77)
    pair, _ = build_modules(1)
    pair.to(dev())
    B, N, K = 41, 11, pair.edge_types
    pk = pair.edge_aggregation_list[0]._packed()
    ori = torch.randn(B, N, 64, device=dev())
    ef = torch.rand(B, ops.pair_count(N), K, device=dev())
    A = ops.node_linear(ori, pk["W1cat"], pk["b1half"], K * 128)
    (feat,) = ops.agg_mlp_grouped([(ops.PairSpec(A), ef, pk, K)])
    (node,) = ops.agg_mlp_grouped([(ops.PairSpec(A, node=True), ef, pk, K)])
    pk2 = pair._packed_mlp2(pair.nmp_mlp_end)
    for xs in ("1", "0"):
        os.environ["GN_MLP2_XS"] = xs
        try:
            (ya,) = ops.mlp2_grouped([(ops.ScatterSpec(feat, None, ori, True), pk2, None)])
            (yb,) = ops.mlp2_grouped([(ops.NodeAggSpec(node, ori), pk2, None)])
        finally:
            os.environ.pop("GN_MLP2_XS", None)
        assert maxerr(ya, yb) <= 2e-6 * max(1.0, float(ya.abs().max())), xs


def test_node_form_clamp_scaling_is_exact_and_has_a_fallback():
    """The node form evaluates relu(a + b) as clamp01(s a + s b) / s with s a power of two chosen from the largest staged
    pre-activation: results must not depend on the magnitude class of the inputs beyond fp32 rounding — tiny (below the
    2^-27 floor of the scale), ordinary, large (bf16x6 fallback of the range vote), and beyond 2^72 where the kernel takes
    the max form — each against the scattered pair form."""
    from groupnet_amd import ops
    torch.manual_seed(5)
    pair, _ = build_modules(1)
    pair.to(dev())
    B, N, K = 19, 11, pair.edge_types
    pk = pair.edge_aggregation_list[0]._packed()
    ef = torch.rand(B, ops.pair_count(N), K, device=dev())
    base = torch.randn(B, N, K * 128, device=dev())
    for mag in (1e-12, 1.0, 3e4, 1e9, 1e24):
        A = (base * mag).contiguous()
        (feat,) = ops.agg_mlp_grouped([(ops.PairSpec(A), ef, pk, K)])
        (node,) = ops.agg_mlp_grouped([(ops.PairSpec(A, node=True), ef, pk, K)])
        want = ops.agg_scatter(feat, None, torch.zeros(B, N, 64, device=dev()), divisor=1.0, sym=True)[..., :64]
        assert bool(torch.isfinite(node).all()), mag
        scale = float(want.abs().max())
        assert maxerr(node, want) <= 3e-6 * scale + 1e-30, (mag, maxerr(node, want), scale)


def test_node_form_switch_changes_the_path_not_the_result(monkeypatch):
    """GN_NODE_FORM is read per call: 0 keeps the per-pair form (feat (B,E,64) + fused scatter), 1 the node form; the
    block's outputs agree to fp32 rounding and the typed-aggregation launch really differs (its output shape)."""
    import groupnet_amd as G
    from groupnet_amd import ops
    from groupnet_amd.multiscale import MultiScaleHGNN
    torch.manual_seed(9)
    blk = MultiScaleHGNN([2, 5, 11]).to(dev()).eval()
    f = torch.randn(64, 11, 64, device=dev())
    shapes, outs = {}, {}
    real = ops.agg_mlp_grouped

    def spy(items, *a, **k):
        r = real(items, *a, **k)
        shapes[os.environ.get("GN_NODE_FORM", "1")] = tuple(r[0].shape)
        return r
    monkeypatch.setattr(ops, "agg_mlp_grouped", spy)
    monkeypatch.setenv("GN_FUSE_CLOSING", "0")      # (the closing stage on its own: the aggregation's output is what is looked at)
    for mode in ("0", "1"):
        monkeypatch.setenv("GN_NODE_FORM", mode)
        G.set_noise_mode("device", seed=11)
        with torch.no_grad():
            outs[mode] = blk(f)
    G.set_noise_mode("host")
    assert shapes["0"] == (64, 66, 64) and shapes["1"] == (64, 11, 64)
    a, b = outs["0"][0], outs["1"][0]
    assert maxerr(a, b) <= 1e-6 * max(1.0, float(a.abs().max()))


# ---------------------------------------------------------------------------------------------
# closing MLP fused into the typed-aggregation launch (gn_agg_group_t.y)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["f16x3", "bf16x6"])
@pytest.mark.parametrize("B,N,scales,nmp", [(37, 11, [2, 5, 11], 1), (512, 11, [2, 5, 11], 1), (5, 2, [2], 1), (7, 16, [2, 4], 1),
                                            (4, 1, [1], 1), (9, 5, [2, 5], 2), (64, 13, [3, 13], 1)])
def test_closing_stage_fused_into_the_aggregation_launch_is_bit_identical(B, N, scales, nmp, precision, monkeypatch):
    """gn_agg_group_t.y: the workgroups that finish a scene's aggregate apply the closing MLP themselves (scene-aligned
    hyper row blocks, LDS scatter in the fused scatter's order, the 4-wave chain of mlp2_xs_body).  The block's latency
    form (affinity_tail) uses it: no gn_mlp2 launch, rows bit-identical to the two launches; ragged last workgroups,
    scale == N modules (many scenes per workgroup), one and two node row blocks per workgroup, nmp_layers = 2."""
    import groupnet_amd as G
    from groupnet_amd import ops
    from groupnet_amd.multiscale import MultiScaleHGNN
    torch.manual_seed(7 * B + N)
    blk = MultiScaleHGNN(scales, nmp_layers=nmp).to(dev()).eval()
    f = torch.randn(B, N, 64, device=dev())
    calls = {"0": 0, "1": 0}
    real = ops.mlp2_grouped

    def spy(*a, **k):
        calls[os.environ["GN_FUSE_CLOSING"]] += 1
        return real(*a, **k)
    monkeypatch.setattr(ops, "mlp2_grouped", spy)
    old = ops.precision()
    ops.set_precision(precision)
    outs = {}
    try:
        for mode in ("0", "1"):
            monkeypatch.setenv("GN_FUSE_CLOSING", mode)
            G.set_noise_mode("device", seed=21)
            with torch.no_grad():
                outs[mode] = blk(f)[0]
    finally:
        ops.set_precision(old)
        G.set_noise_mode("host")
    assert calls["0"] >= nmp and calls["1"] == 0
    assert torch.equal(outs["0"], outs["1"])
