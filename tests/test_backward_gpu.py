"""Backward of the modules: gradients of the HIP path (fused forward + HIP backward
through torch.autograd) against torch autograd on the CPU oracle, same weights, same inputs, same noise."""
import contextlib

import pytest
import torch

from oracle import ms_hgnn_oracle as O
from relu_probe import WINDOW, relu_probe

pytestmark = pytest.mark.gpu

TOL_CLEAN = 2e-5     # of max|grad| (measured <= 3.3e-6), on scenes without a ReLU unit inside the rounding window of zero (relu_probe.py)
TOL_ANY = 2e-3       # any batch: a unit inside the window may be on in one implementation and off in the other


def _modules(seed, nmp=1):
    import groupnet_amd as G
    torch.manual_seed(seed)
    pair = G.MS_HGNN_oridinary(embedding_dim=16, h_dim=64, mlp_dim=64, bottleneck_dim=64, batch_norm=0, nmp_layers=nmp)
    hyper = G.MS_HGNN_hyper(embedding_dim=64, h_dim=64, mlp_dim=64, bottleneck_dim=64, batch_norm=0, nmp_layers=nmp,
                            scale=3)
    with torch.no_grad():
        for m in (pair, hyper):
            for n_, p in m.named_parameters():
                if "attention_mlp" in n_ or "MLP_distribution" in n_ or "MLP_factor" in n_:
                    p.mul_(4.0)
    return pair, hyper


def _check(grads_hip, grads_ref, names, tol=TOL_ANY):
    """Every parameter gradient element-wise against the reference, relative to its own scale (floored at 1 % of the
    model's largest: softmax-shift directions are exactly zero).  Returns (worst error, its parameter)."""
    worst, where = 0.0, None
    gmax = max(float(grads_ref[n].abs().max()) for n in names)
    for name in names:
        a, b = grads_hip[name], grads_ref[name]
        assert a is not None and b is not None, name
        assert a.shape == b.shape, (name, a.shape, b.shape)
        scale = max(float(b.abs().max()), 1e-2 * gmax)
        err = float((a.cpu() - b).abs().max()) / scale
        if err > worst:
            worst, where = err, name
        assert err <= tol, (name, err, scale, tol)
    return worst, where


def _grad_compare(tag, module, oracle_fwd, hip_fwd, h, noise, min_used):
    """Gradients of L = <node_feat, R1> + <factors, R2> w.r.t. h and every used parameter, HIP vs torch autograd on
    the CPU oracle — on the whole batch (gate TOL_ANY) and on its clean scenes alone (gate TOL_CLEAN); prints the
    measured errors.  oracle_fwd(state, h, noise) / hip_fwd(h_dev, noise_dev) -> (node_feat, factors)."""
    dev = torch.device("cuda:0")
    B = h.shape[0]
    R1 = torch.randn(B, h.shape[1], 64)
    state0 = {k: v.detach().clone() for k, v in module.state_dict().items()}
    module.to(dev).train()
    R2 = None

    def both(rows, tol):
        nonlocal R2
        state = {k: v.clone().requires_grad_(True) for k, v in state0.items()}
        hh = h[rows].clone().requires_grad_(True)
        nz = [u[rows].contiguous() for u in noise]
        with relu_probe(len(rows)) as probe:
            nf, fac = oracle_fwd(state, hh, nz)
        if R2 is None:
            R2 = torch.randn(B, *fac.shape[1:])
        ((nf * R1[rows]).sum() + (fac * R2[rows]).sum()).backward()
        for p in module.parameters():
            p.grad = None
        x = h[rows].clone().to(dev).requires_grad_(True)
        nf2, fac2 = hip_fwd(x, [u.to(dev) for u in nz])
        assert float((nf2.detach().cpu() - nf.detach()).abs().max()) <= 1e-5
        ((nf2 * R1[rows].to(dev)).sum() + (fac2 * R2[rows].to(dev)).sum()).backward()
        eh = float((x.grad.cpu() - hh.grad).abs().max()) / float(hh.grad.abs().max())
        assert eh <= tol, (tag, "dL/dh", eh, tol)
        used = [k for k, v in state.items() if v.grad is not None and float(v.grad.abs().max()) > 0]
        assert len(used) >= min_used
        ew, where = _check({k: p.grad for k, p in module.named_parameters()}, {k: v.grad for k, v in state.items()}, used,
                           tol)
        return probe, eh, ew, where, {k: p.grad for k, p in module.named_parameters()}

    rows_all = torch.arange(B)
    probe, eh, ew, where, hip_grads = both(rows_all, TOL_ANY)
    clean = probe.clean()
    msg = (f"\n{tag}: whole batch dL/dh {eh:.1e}, worst parameter {ew:.1e} ({where}); {int(clean.sum())}/{B} scenes clean "
           f"({probe.units} ReLU units per scene, window {WINDOW:g})")
    if bool(clean.any()) and not bool(clean.all()):
        _, eh2, ew2, where2, _ = both(rows_all[clean], TOL_CLEAN)
        msg += f"; clean scenes alone dL/dh {eh2:.1e}, worst parameter {ew2:.1e} ({where2})"
    elif bool(clean.all()):
        assert eh <= TOL_CLEAN and ew <= TOL_CLEAN, (tag, eh, ew, where)
        msg += "; all clean: gated at 2e-5"
    print(msg)
    return hip_grads, bool(clean.any())


def test_grouped_gemm_matches_torch():
    """Every mode of gn_gemm_grouped_f32 (the backward's workhorse) against torch.matmul, ragged sizes,
    strided views, several problems in one launch (more than one table's worth)."""
    from groupnet_amd.backward import GemmBatch
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    R = lambda *s: torch.randn(*s, generator=g).to(dev)
    gb, checks = GemmBatch(), []
    for i, (M, N, K) in enumerate([(1, 1, 1), (130, 70, 33), (257, 64, 256), (64, 200, 5), (1000, 32, 128)] * 4):
        tA, tB = bool(i & 1), bool(i & 2)
        A = R(K, M + 3)[:, :M] if tA else R(M, K + 5)[:, :K]
        Bm = R(N, K) if tB else R(K, N + 2)[:, :N]
        C0 = R(M, N + 1)[:, :N].clone()
        bias = R(N) if i % 3 == 0 else None
        mask = R(M, N) if i % 5 == 1 else None
        relu = i % 4 == 2
        alpha, beta = (0.5, 0.25) if i % 2 else (1.0, 0.0)
        C = torch.zeros(M, N + 1, device=dev)[:, :N]
        C.copy_(C0)
        gb.add(A, Bm, C, tA, tB, bias, mask, relu, alpha, beta)
        ref = alpha * ((A.t() if tA else A).double() @ (Bm.t() if tB else Bm).double())
        if bias is not None:
            ref = ref + bias.double()
        ref = ref + beta * C0.double()
        if relu:
            ref = ref.clamp_min(0)
        if mask is not None:
            ref = torch.where(mask > 0, ref, torch.zeros_like(ref))
        checks.append((C, ref))
    gb.run()
    for C, ref in checks:
        assert float((C.double() - ref).abs().max()) <= 2e-5 * (1 + float(ref.abs().max()))
    # accumulate mode: split-K atomics, scaled stored rows, bias gradient as a side output
    rows = 20000
    dY, X, rs = R(rows, 96), R(rows, 64), torch.rand(rows, 3, generator=g).to(dev)
    C, cs = torch.zeros(96, 64, device=dev), torch.zeros(96, device=dev)
    gb.add(dY, X, C, tA=True, accum=True, rs=rs[:, 1], colsum=cs)
    gb.run()
    ref = (dY.double() * rs[:, 1:2].double()).t() @ X.double()
    assert float((C.double() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    ref_cs = (dY.double() * rs[:, 1:2].double()).sum(0)
    assert float((cs.double() - ref_cs).abs().max()) <= 1e-4 * float(ref_cs.abs().max())


@pytest.mark.parametrize("B,N,scale,nmp", [(5, 11, 3, 1), (2, 7, 7, 1), (3, 20, 2, 1), (3, 11, 5, 2), (2, 6, 6, 3),
                                           (2, 50, 8, 1), (2, 70, 16, 1)])      # config-4 shape; N > 64 (unfused gather)
def test_hyper_module_gradients(B, N, scale, nmp):
    _, hyper = _modules(100 + N, nmp)
    hyper.scale = scale
    h = torch.randn(B, N, 64)
    corr = O.affinity(h)
    U = [torch.rand(s) for s in O.noise_shapes(B, N, scale, nmp)]
    Hs = {}

    def oracle_fwd(state, hh, nz):
        rows = slice(None) if hh.shape[0] == B else None
        c = corr if hh.shape[0] == B else O.affinity(hh.detach())
        nf, fac, H = O.ms_hgnn_hyper_forward(state, hh, c, scale, nz, nmp_layers=nmp, decomposed=True)
        Hs["ref"], Hs["corr"] = H, c
        return nf, fac

    def hip_fwd(x, nz):
        nf, fac, H = hyper(x, Hs["corr"].to(x.device), noise_u=nz)
        assert torch.equal(H.cpu(), Hs["ref"])
        return nf, fac

    hip, _ = _grad_compare(f"hyper B={B} N={N} s={scale} nmp={nmp}", hyper, oracle_fwd, hip_fwd, h, U, 40 * nmp)
    # parameters the forward never touches get no gradient, as with the reference
    assert hip["spatial_embedding.weight"] is None and hip["edge_aggregation_list.0.mlp.layers.0.weight"] is None


@pytest.mark.parametrize("B,N,nmp", [(8, 11, 1), (4, 5, 1), (4, 6, 2), (1, 50, 1)])      # last: config-4 shape, 1275 pair rows
def test_pairwise_module_gradients(B, N, nmp):
    pair, _ = _modules(200 + N, nmp)
    h = torch.randn(B, N, 64)
    U = [torch.rand(s) for s in O.noise_shapes(B, N, None, nmp)]
    _grad_compare(f"pairwise B={B} N={N} nmp={nmp}", pair,
                  lambda state, hh, nz: O.ms_hgnn_pairwise_forward(state, hh, nz, nmp_layers=nmp, decomposed=True),
                  lambda x, nz: pair(x, noise_u=nz), h, U, 30)


def test_multiscale_block_trains_one_sgd_step():
    """End to end: loss through the multiscale block decreases after one SGD step (gradients have the
    right sign and scale), and inference mode afterwards still takes the grouped fused path."""
    from groupnet_amd.multiscale import MultiScaleHGNN
    dev = torch.device("cuda:0")
    torch.manual_seed(9)
    blk = MultiScaleHGNN([2, 11]).to(dev).train()
    f = torch.randn(6, 11, 64, device=dev)
    target = torch.randn(6, 11, 64 * 4, device=dev)
    noise = [[torch.rand(s, device=dev)] for s in blk.noise_shapes(6, 11)]
    opt = torch.optim.SGD(blk.parameters(), lr=0.05)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        out, _ = blk(f, noise_u=noise)
        loss = ((out - target) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[2] < losses[0]
    with torch.no_grad():
        out2, H = blk.eval()(f, noise_u=noise)
    assert out2.shape == (6, 11, 256) and bool(torch.isfinite(out2).all())


def test_graphed_train_step_matches_eager_and_learns():
    """fwd + loss + bwd + SGD in one hipGraph: the first replay reproduces an eager step with the same
    Philox noise bit for bit in the loss and to fp32 rounding (split-K atomics) in the updated weights;
    further replays keep reducing the loss; eager inference afterwards sees the updated weights."""
    import copy
    import groupnet_amd as G
    from groupnet_amd.graphs import GraphedTrainStep
    from groupnet_amd.multiscale import MultiScaleHGNN
    from groupnet_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    B, N = 16, 11
    blk = MultiScaleHGNN([2, 5, 11]).to(dev).train()
    ref = copy.deepcopy(blk)
    f = torch.randn(B, N, 64, device=dev)
    tgt = torch.randn(B, N, blk.out_features, device=dev)
    loss_fn = lambda out, H, t: ((out - t) ** 2).mean()
    step = GraphedTrainStep(blk, torch.optim.SGD(blk.parameters(), lr=0.05), loss_fn, B, N,
                            target_shapes=[tuple(tgt.shape)], seed=11, warmup=2)
    # the warm-up steps trained `blk`; restart both from the same weights
    blk.load_state_dict(ref.state_dict())
    G.MS_HGNN_batch.invalidate_weight_caches(blk)
    l0 = float(step(f, tgt))
    # the same step eagerly on the copy: replay k draws from Philox position k * draws_per_step
    opt = torch.optim.SGD(ref.parameters(), lr=0.05)
    counter = torch.zeros(1, dtype=torch.int64, device=dev)
    G.set_noise_mode("device", seed=11, offset=0, counter=counter)
    try:
        out, _ = ref(f)
        loss = loss_fn(out, None, tgt)
        loss.backward()
        opt.step()
    finally:
        G.set_noise_mode("host")
    assert abs(float(loss) - l0) <= 1e-6 * max(1.0, abs(l0))
    for (n1, p1), (_, p2) in zip(blk.named_parameters(), ref.named_parameters()):
        assert float((p1 - p2).abs().max()) <= 1e-5 * (1.0 + float(p2.abs().max())), n1
    losses = [l0] + [float(step()) for _ in range(4)]
    assert losses[-1] < losses[0]
    with torch.no_grad():       # eager call between replays sees the trained weights (caches were dropped)
        G.set_noise_mode("device", seed=11, offset=0, counter=counter)
        try:
            out_now, _ = blk.eval()(f)
        finally:
            G.set_noise_mode("host")
    assert float(((out_now - tgt) ** 2).mean()) < l0


@pytest.mark.parametrize("xs", ["1", "0"])
@pytest.mark.parametrize("name", ["n11_b5", "n7_b3_nmp2"])
def test_hip_backward_matches_reference_gradients(name, xs, monkeypatch):
    """The HIP backward against gradients produced by the REFERENCE's own autograd
    (tests/golden/grad_*.npz, generated by make_golden_backward.py): same weights (golden state_dicts), same
    inputs, same uniforms, same loss.  dL/dh and the stored parameter gradients element-wise, every parameter
    gradient through its (sum, sum|.|, max|.|).  Gate: 1e-4 of max|g| when no scene of the golden batch holds a ReLU
    unit inside the rounding window of zero (relu_probe.py), else 2e-3; the measured errors are printed.
    xs: the forward's closing MLP (which also WRITES the activations the backward reads) through the
    4-waves-per-row-block kernel ("1", the launcher's choice at this size) and the one-wave-per-block kernel ("0")."""
    monkeypatch.setenv("GN_MLP2_XS", xs)
    import numpy as np
    import os
    import groupnet_amd as G
    from conftest import load_state
    dev = torch.device("cuda:0")
    with np.load(os.path.join(os.path.dirname(__file__), "golden", f"grad_{name}.npz")) as z:
        c = {k: z[k].copy() for k in z.files}
    nmp = int(c["nmp"])
    sfx = "_nmp2" if nmp == 2 else ""
    kw = dict(h_dim=64, mlp_dim=64, bottleneck_dim=64, batch_norm=0, nmp_layers=nmp)
    corr = torch.from_numpy(c["corr"]).to(dev)
    for prefix in ["pair"] + [f"hyper{int(s)}" for s in c["scales"]]:
        if prefix == "pair":
            m = G.MS_HGNN_oridinary(embedding_dim=16, **kw)
            m.load_state_dict(load_state("pairwise" + sfx))
        else:
            m = G.MS_HGNN_hyper(embedding_dim=64, scale=int(prefix[5:]), **kw)
            m.load_state_dict(load_state("hyper" + sfx))
        m.to(dev).train()
        U, i = [], 0
        while f"{prefix}_U{i}" in c:
            U.append(torch.from_numpy(c[f"{prefix}_U{i}"]).to(dev))
            i += 1
        # which scenes are clean, from the oracle's pre-activations on the golden inputs
        st_cpu = load_state(("pairwise" if prefix == "pair" else "hyper") + sfx)
        hc, Uc = torch.from_numpy(c["h"]), [u.cpu() for u in U]
        with torch.no_grad(), relu_probe(hc.shape[0]) as probe:
            if prefix == "pair":
                O.ms_hgnn_pairwise_forward(st_cpu, hc, Uc, nmp_layers=nmp, decomposed=True)
            else:
                O.ms_hgnn_hyper_forward(st_cpu, hc, corr.cpu(), int(prefix[5:]), Uc, nmp_layers=nmp, decomposed=True)
        tol = TOL_CLEAN if bool(probe.clean().all()) else TOL_ANY
        h = torch.from_numpy(c["h"]).to(dev).requires_grad_(True)
        out = m(h, noise_u=U) if prefix == "pair" else m(h, corr, noise_u=U)
        nf, fac = out[0], out[1]
        assert float((nf.detach().cpu() - torch.from_numpy(c[f"{prefix}_node_feat"])).abs().max()) <= 1e-5
        assert float((fac.detach().cpu() - torch.from_numpy(c[f"{prefix}_factors"])).abs().max()) <= 1e-5
        ((nf * torch.from_numpy(c[f"{prefix}_R1"]).to(dev)).sum() + (fac * torch.from_numpy(c[f"{prefix}_R2"]).to(dev)).sum()).backward()
        gh = torch.from_numpy(c[f"{prefix}_g_h"])
        eh = float((h.grad.cpu() - gh).abs().max()) / float(gh.abs().max())
        assert eh <= tol, (prefix, eh, tol)
        worst = 0.0
        params = dict(m.named_parameters())
        for n, ref in zip(c[f"{prefix}_stat_names"], c[f"{prefix}_stats"]):
            g = params[str(n)].grad
            if np.isnan(ref[0]):
                assert g is None, n
                continue
            g64 = g.double().cpu()
            tl = tol * ref[1] + 1e-5
            assert abs(float(g64.abs().sum()) - ref[1]) <= tl and abs(float(g64.sum()) - ref[0]) <= tl, (prefix, n)
            assert abs(float(g64.abs().max()) - ref[2]) <= tol * ref[2] + 1e-6, (prefix, n)
            if f"{prefix}_g/{n}" in c:
                full = torch.from_numpy(c[f"{prefix}_g/{n}"])
                e = float((g.cpu() - full).abs().max()) / (float(full.abs().max()) + 1e-12)
                worst = max(worst, e)
                assert e <= tol + 1e-6, (prefix, n, e)
        print(f"\ngolden {name}/{prefix}: dL/dh {eh:.1e}, worst stored parameter gradient {worst:.1e}, "
              f"{int(probe.clean().sum())}/{hc.shape[0]} scenes clean -> gate {tol:g}")


def test_full_size_backward_is_additive_over_shards():
    """BASELINE size (B=512, N=11, scales {2,5,11}), size-independent properties of the backward: scenes are
    independent, so dL/df of a scene does not depend on what else is in the batch, and every parameter gradient
    of a batch is the sum of its half-batch gradients (same weights, same noise rows).
    Kernel forms depend on the launch size, so pre-activations differ in the last bits between the runs, and a
    ReLU unit inside that window of zero may flip (relu_probe.py).  The claim is checked, not asserted:
      (1) dL/df: every scene that disagrees between the full and the half-batch run by more than 1e-4 of max|g|
          IS a scene with such a unit (found from the CPU oracle's pre-activations), and no scene disagrees by
          more than 2e-2;
      (2) on the clean scenes alone (~40 % of 512) the same experiment must hold to 1e-4: dL/df element-wise and
          every parameter gradient additive in Frobenius norm."""
    from groupnet_amd.multiscale import MultiScaleHGNN
    dev = torch.device("cuda:0")
    torch.manual_seed(12)
    B, N, scales = 512, 11, [2, 5, 11]
    blk = MultiScaleHGNN(scales)
    sp = {k: v.detach().clone() for k, v in blk.interaction.state_dict().items()}
    shs = [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in blk.interaction_hyper]
    blk.to(dev).train()
    f = torch.randn(B, N, 64)
    R = torch.randn(B, N, blk.out_features, device=dev)
    noise = [[torch.rand(s)] for s in blk.noise_shapes(B, N)]
    with torch.no_grad(), relu_probe(B) as probe:
        O.ms_hgnn_multiscale_forward(sp, shs, scales, f, noise[0], noise[1:], decomposed=True)
    clean = probe.clean()
    f, noise = f.to(dev), [[u[0].to(dev)] for u in noise]

    def run(rows):
        for p in blk.parameters():
            p.grad = None
        x = f[rows].clone().requires_grad_(True)
        out, _ = blk(x, noise_u=[[u[0][rows].contiguous()] for u in noise])
        (out * R[rows]).sum().backward()
        return x.grad, {n: p.grad.clone() for n, p in blk.named_parameters() if p.grad is not None}

    def additive(rows):
        half = len(rows) // 2
        g_full, w_full = run(rows)
        g_a, w_a = run(rows[:half])
        g_b, w_b = run(rows[half:])
        gscale = float(g_full.abs().max())
        per_scene = (torch.cat((g_a, g_b)) - g_full).abs().flatten(1).max(dim=1).values / gscale
        assert len(w_full) >= 200
        werr = max(float((w_a[n] + w_b[n] - v).norm()) / (float(v.norm()) + 1e-12) for n, v in w_full.items())
        return per_scene.cpu(), werr

    rows = torch.arange(B, device=dev)
    per_scene, werr = additive(rows)
    bad = per_scene > TOL_CLEAN
    print(f"\nadditivity B={B}: {int(bad.sum())} scenes differ by more than 1e-4 (max {float(per_scene.max()):.1e}); "
          f"{int((bad & clean).sum())} of them are clean scenes; {int(clean.sum())} clean scenes; "
          f"parameter gradients additive to {werr:.1e}")
    assert not bool((bad & clean).any()), "a scene without a unit near zero changed with the batch composition"
    assert float(per_scene.max()) <= 2e-2 and werr <= 2e-2
    per_scene_c, werr_c = additive(rows[clean.to(dev)])
    print(f"   clean scenes alone ({int(clean.sum())}): dL/df max {float(per_scene_c.max()):.1e}, parameter gradients additive "
          f"to {werr_c:.1e}")
    assert float(per_scene_c.max()) <= TOL_CLEAN and werr_c <= TOL_CLEAN


def test_graphed_train_step_with_adam():
    """The reference trains with Adam (train_hyper_nba.py); `capturable=True` keeps its step inside the graph."""
    from groupnet_amd.graphs import GraphedTrainStep
    from groupnet_amd.multiscale import MultiScaleHGNN
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    B, N = 8, 11
    blk = MultiScaleHGNN([5, 11]).to(dev).train()
    f = torch.randn(B, N, 64, device=dev)
    tgt = torch.randn(B, N, blk.out_features, device=dev)
    opt = torch.optim.Adam(blk.parameters(), lr=1e-3, capturable=True)
    step = GraphedTrainStep(blk, opt, lambda out, H, t: ((out - t) ** 2).mean(), B, N, target_shapes=[tuple(tgt.shape)],
                            seed=3, warmup=2)
    losses = [float(step(f, tgt)) for _ in range(12)]
    assert all(l == l for l in losses) and losses[-1] < losses[0]


def test_pairwise_backward_beyond_the_per_scene_kernel():
    """N = 150: the scene's rows no longer fit the per-scene node2edge backward, so the pairwise module's
    backward falls back to ordered edge rows with the explicit (B, N*N, N) incidence, re-computes its edge-row
    activations (the forward ran on unordered pairs) and uses the one-wave-per-hyperedge kernel with global
    atomics.  Gradient of h and a sample of parameter gradients against the oracle's autograd."""
    dev = torch.device("cuda:0")
    B, N = 1, 150
    pair, _ = _modules(321)
    state = {k: v.detach().clone().requires_grad_(True) for k, v in pair.state_dict().items()}
    g = torch.Generator().manual_seed(4)
    h = torch.randn(B, N, 64, generator=g)
    U = [torch.rand(s, generator=g) for s in O.noise_shapes(B, N, None)]
    R1 = torch.randn(B, N, 64, generator=g)
    h_ref = h.clone().requires_grad_(True)
    nf, fac = O.ms_hgnn_pairwise_forward(state, h_ref, U, decomposed=True)
    (nf * R1).sum().backward()
    pair.to(dev).train()
    x = h.clone().to(dev).requires_grad_(True)
    nf2, _ = pair(x, noise_u=[u.to(dev) for u in U])
    assert float((nf2.detach().cpu() - nf.detach()).abs().max()) <= 1e-5
    (nf2 * R1.to(dev)).sum().backward()
    eh = float((x.grad.cpu() - h_ref.grad).abs().max()) / float(h_ref.grad.abs().max())
    assert eh <= TOL_ANY
    used = [k for k, v in state.items() if v.grad is not None and float(v.grad.abs().max()) > 0]
    ew, where = _check({k: p.grad for k, p in pair.named_parameters()}, {k: v.grad for k, v in state.items()}, used)
    print(f"\npairwise N=150 (22 500 edge rows, ~26 M ReLU units: never clean): dL/dh {eh:.1e}, worst parameter {ew:.1e} ({where})")


def test_results_do_not_depend_on_stale_memory():
    """Every buffer of the forward and the backward comes from torch.empty / a zeroed pool; poisoning the caching
    allocator's free blocks with NaN between two identical steps must change nothing (beyond the order of
    atomic additions in the weight gradients)."""
    from groupnet_amd.multiscale import MultiScaleHGNN
    dev = torch.device("cuda:0")
    torch.manual_seed(6)

    def poison():
        junk = [torch.full((n,), float("nan"), device=dev) for n in (1 << k for k in range(8, 25))]
        junk += [torch.full((3 * n,), float("nan"), device=dev) for n in (1 << k for k in range(8, 23))]
        del junk

    for B, N, scales in ((37, 11, [2, 5, 11]), (3, 50, [4, 50])):
        blk = MultiScaleHGNN(scales).to(dev).train()
        f = torch.randn(B, N, 64, device=dev)
        noise = [[torch.rand(s, device=dev)] for s in blk.noise_shapes(B, N)]
        R = torch.randn(B, N, blk.out_features, device=dev)

        def run():
            for p in blk.parameters():
                p.grad = None
            x = f.clone().requires_grad_(True)
            out, _ = blk(x, noise_u=noise)
            (out * R).sum().backward()
            return out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in blk.parameters() if p.grad is not None]

        o1, g1, w1 = run()
        torch.cuda.synchronize()
        poison()
        torch.cuda.synchronize()
        o2, g2, w2 = run()
        assert torch.equal(o1, o2)
        assert float((g1 - g2).abs().max()) <= 1e-5 * float(g1.abs().max())
        for a, b in zip(w1, w2):
            assert bool(torch.isfinite(b).all())
            assert float((a - b).abs().max()) <= 1e-4 * (float(a.abs().max()) + 1e-6)


def test_data_writes_and_the_packed_weight_caches():
    """`p.data.mul_()` does not bump `p._version`, so a cache keyed on versions cannot see it.
    Contract: while autograd records for the parameters (a training step) every call re-packs — the write is
    seen with no further action; under no_grad the cache is trusted and `invalidate_weight_caches` is what a
    caller that rewrites weights behind autograd's back must call."""
    from groupnet_amd.multiscale import MultiScaleHGNN
    from groupnet_amd.MS_HGNN_batch import invalidate_weight_caches
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    blk = MultiScaleHGNN([2, 11]).to(dev)
    f = torch.randn(5, 11, 64, device=dev)
    noise = [[torch.rand(s, device=dev)] for s in blk.noise_shapes(5, 11)]
    # training mode: no invalidation needed
    blk.train()
    o1, _ = blk(f, noise_u=noise)
    v0 = [p._version for p in blk.parameters()]
    for p in blk.parameters():
        p.data.mul_(1.5)
    assert [p._version for p in blk.parameters()] == v0          # the premise: versions did not move
    o2, _ = blk(f, noise_u=noise)
    assert float((o2 - o1).abs().max()) > 1e-3
    # inference: trusted cache, explicit invalidation
    blk.eval()
    with torch.no_grad():
        a, _ = blk(f, noise_u=noise)
        for p in blk.parameters():
            p.data.mul_(1.25)
        invalidate_weight_caches(blk)
        b, _ = blk(f, noise_u=noise)
    assert float((b - a).abs().max()) > 1e-3
    # and an ordinary in-place update (version bump) is seen without any call
    with torch.no_grad():
        for p in blk.parameters():
            p.mul_(0.5)
        c, _ = blk(f, noise_u=noise)
    assert float((c - b).abs().max()) > 1e-3


def test_backward_after_an_in_place_parameter_update_raises():
    """two forwards, optimizer.step(), then backward of the first — torch's autograd raises for its own
    ops ("modified by an inplace operation"); so does the HIP backward, instead of silently pairing old
    activations with new weights."""
    import groupnet_amd as G
    dev = torch.device("cuda:0")
    torch.manual_seed(8)
    hyper = G.MS_HGNN_hyper(embedding_dim=64, h_dim=64, mlp_dim=64, bottleneck_dim=64, batch_norm=0, nmp_layers=1,
                            scale=3).to(dev).train()
    h = torch.randn(3, 7, 64, device=dev)
    corr = O.affinity(h.cpu()).to(dev)
    opt = torch.optim.SGD(hyper.parameters(), lr=0.1)
    nf1, _, _ = hyper(h, corr)
    nf2, _, _ = hyper(h, corr)
    nf2.sum().backward()
    opt.step()
    with pytest.raises(RuntimeError, match="modified in place"):
        nf1.sum().backward()


def test_seeded_training_forward_sees_the_noise_of_the_seeded_inference_forward():
    """with default noise and nmp_layers > 1 the training path used to draw round-major (round 0 of
    every module, then round 1) while inference and the reference draw module-major.  Same seed, same features."""
    from groupnet_amd.multiscale import MultiScaleHGNN
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    blk = MultiScaleHGNN([2, 5], nmp_layers=2).to(dev)
    f = torch.randn(4, 11, 64, device=dev)
    torch.manual_seed(77)
    with torch.no_grad():
        a, _ = blk.eval()(f)
    torch.manual_seed(77)
    b, _ = blk.train()(f)                    # parameters require grad: the autograd path
    assert b.requires_grad
    assert float((a - b.detach()).abs().max()) <= 1e-6


def test_repack_scope_two_launches_equal_the_per_plan_refreshes():
    """ops.repack_scope (what GraphedTrainStep wraps every step in): the first step records the pack plans / bf16-core
    images a training step touches, later steps rebuild ALL of them with two launches up front and skip the recorded
    per-plan launches.  Two blocks with identical weights take the same three SGD steps (the second and third step see
    parameters rewritten through `.data`, the case the per-step refresh exists for) — one inside the scope, one without:
    identical losses and parameters, bit for bit, and the scoped block's later steps really were served by the batch."""
    import copy
    from groupnet_amd import ops
    from groupnet_amd.multiscale import MultiScaleHGNN
    import groupnet_amd as G
    dev = lambda: torch.device("cuda:0")
    torch.manual_seed(31)
    a = MultiScaleHGNN([2, 5]).to(dev()).train()
    b = copy.deepcopy(a)
    B, N = 6, 7
    f = torch.randn(B, N, 64, device=dev())
    tgt = torch.randn(B, N, a.out_features, device=dev())
    U = [[torch.rand(s, device=dev())] for s in a.noise_shapes(B, N)]
    holder = {}
    calls = {"pack": 0}
    orig = ops.PackPlan.refresh

    def counting(self):
        before = ops._REPACK["done_plans"]
        if not (before is not None and id(self) in before):
            calls["pack"] += 1
        return orig(self)

    losses = {"a": [], "b": []}
    ops.PackPlan.refresh = counting
    try:
        for step in range(3):
            for name, blk in (("a", a), ("b", b)):
                ctx = ops.repack_scope(holder) if name == "a" else contextlib.nullcontext()
                calls["pack"] = 0
                with ctx:
                    out, _ = blk(f, noise_u=U)
                    loss = ((out - tgt) ** 2).mean()
                    blk.zero_grad(set_to_none=True)
                    loss.backward()
                if name == "a" and step > 0:
                    assert calls["pack"] == 0, "a recorded plan took its own refresh launch"
                if name == "b":
                    assert calls["pack"] > 0
                with torch.no_grad():
                    for p in blk.parameters():
                        if p.grad is not None:
                            p.data.add_(p.grad, alpha=-0.05)       # behind autograd's back: no version bump
                losses[name].append(float(loss.detach()))
    finally:
        ops.PackPlan.refresh = orig
    assert holder.get("batch") is not None and len(holder["batch"].plans) > 4
    print(f"\nrepack_scope: {len(holder['batch'].plans)} plans / {len(holder['batch'].splits)} images per step in two launches; "
          f"losses {losses['a']}")
    assert losses["a"] == losses["b"]
    assert losses["a"][2] != losses["a"][0]
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa, pb)
