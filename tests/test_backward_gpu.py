"""Backward of the modules: gradients of the HIP path (fused forward + HIP backward
through torch.autograd) against torch autograd on the CPU oracle, same weights, same inputs, same noise."""
import pytest
import torch

from oracle import ms_hgnn_oracle as O

pytestmark = pytest.mark.gpu


def _modules(seed):
    import groupnet_amd as G
    torch.manual_seed(seed)
    pair = G.MS_HGNN_oridinary(embedding_dim=16, h_dim=64, mlp_dim=64, bottleneck_dim=64, batch_norm=0, nmp_layers=1)
    hyper = G.MS_HGNN_hyper(embedding_dim=64, h_dim=64, mlp_dim=64, bottleneck_dim=64, batch_norm=0, nmp_layers=1,
                            scale=3)
    with torch.no_grad():
        for m in (pair, hyper):
            for n_, p in m.named_parameters():
                if "attention_mlp" in n_ or "MLP_distribution" in n_ or "MLP_factor" in n_:
                    p.mul_(4.0)
    return pair, hyper


def _check(grads_hip, grads_ref, names):
    worst = 0.0
    for name in names:
        a, b = grads_hip[name], grads_ref[name]
        assert a is not None and b is not None, name
        assert a.shape == b.shape, (name, a.shape, b.shape)
        scale = float(b.abs().max()) + 1e-6
        err = float((a.cpu() - b).abs().max()) / scale
        worst = max(worst, err)
        assert err <= 2e-3, (name, err, scale)
    return worst


@pytest.mark.parametrize("B,N,scale", [(5, 11, 3), (2, 7, 7), (3, 20, 2)])
def test_hyper_module_gradients(B, N, scale):
    dev = torch.device("cuda:0")
    _, hyper = _modules(100 + N)
    hyper.scale = scale
    state = {k: v.detach().clone().requires_grad_(True) for k, v in hyper.state_dict().items()}
    h = torch.randn(B, N, 64)
    corr = O.affinity(h)
    U = [torch.rand(s) for s in O.noise_shapes(B, N, scale)]
    R1, R2 = torch.randn(B, N, 64), None
    # oracle
    h_ref = h.clone().requires_grad_(True)
    nf, fac, H = O.ms_hgnn_hyper_forward(state, h_ref, corr, scale, U, decomposed=True)
    R2 = torch.randn_like(fac)
    ((nf * R1).sum() + (fac * R2).sum()).backward()
    # HIP
    hyper.to(dev).train()
    h_hip = h.clone().to(dev).requires_grad_(True)
    nf2, fac2, H2 = hyper(h_hip, corr.to(dev), noise_u=[u.to(dev) for u in U])
    assert torch.equal(H2.cpu(), H)
    assert float((nf2.detach().cpu() - nf.detach()).abs().max()) <= 1e-5
    ((nf2 * R1.to(dev)).sum() + (fac2 * R2.to(dev)).sum()).backward()
    assert float((h_hip.grad.cpu() - h_ref.grad).abs().max()) <= 2e-3 * float(h_ref.grad.abs().max())
    used = [k for k, v in state.items() if v.grad is not None and float(v.grad.abs().max()) > 0]
    hip = {k: p.grad for k, p in hyper.named_parameters()}
    ref = {k: v.grad for k, v in state.items()}
    assert len(used) >= 40
    _check(hip, ref, used)
    # parameters the forward never touches get no gradient, as with the reference
    assert hip["spatial_embedding.weight"] is None and hip["edge_aggregation_list.0.mlp.layers.0.weight"] is None


@pytest.mark.parametrize("B,N", [(4, 11), (2, 5)])
def test_pairwise_module_gradients(B, N):
    dev = torch.device("cuda:0")
    pair, _ = _modules(200 + N)
    state = {k: v.detach().clone().requires_grad_(True) for k, v in pair.state_dict().items()}
    h = torch.randn(B, N, 64)
    U = [torch.rand(s) for s in O.noise_shapes(B, N, None)]
    R1 = torch.randn(B, N, 64)
    h_ref = h.clone().requires_grad_(True)
    nf, fac = O.ms_hgnn_pairwise_forward(state, h_ref, U, decomposed=True)
    R2 = torch.randn_like(fac)
    ((nf * R1).sum() + (fac * R2).sum()).backward()
    pair.to(dev).train()
    h_hip = h.clone().to(dev).requires_grad_(True)
    nf2, fac2 = pair(h_hip, noise_u=[u.to(dev) for u in U])
    ((nf2 * R1.to(dev)).sum() + (fac2 * R2.to(dev)).sum()).backward()
    assert float((h_hip.grad.cpu() - h_ref.grad).abs().max()) <= 2e-3 * float(h_ref.grad.abs().max())
    used = [k for k, v in state.items() if v.grad is not None and float(v.grad.abs().max()) > 0]
    _check({k: p.grad for k, p in pair.named_parameters()}, {k: v.grad for k, v in state.items()}, used)


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
