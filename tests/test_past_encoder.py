"""Trajectory encoders: against goldens produced by the reference's own classes
(tests/golden/make_golden_past_encoder.py) and, for further shapes and the training path, against the oracle
restatement those goldens pin (oracle/past_encoder_oracle.py)."""
import types

import pytest
import torch

from oracle import ms_hgnn_oracle as O
from oracle import past_encoder_oracle as PO

def close_to(got, want, tag, rel=1e-5):
    """north_star's bar is 1e-5 ABSOLUTE on O(1) features.  The encoders' outputs are un-normalised (max|value| 200 .. 800 on
    these inputs: the fp32 spacing there is 1.5e-5 .. 6e-5, so 1e-5 absolute is below one ulp), so the gate is
    1e-5 * max(1, max|want|); the absolute error and the scale are printed (measured on MI355X: 0.9e-4 .. 4.8e-4 absolute
    = 4e-7 .. 1e-6 of the scale, a handful of ulps)."""
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    err, scale = float((got - want).abs().max()), float(want.abs().max())
    print(f"\n{tag}: max abs err {err:.2e} at max|want| {scale:.2f} -> {err / max(1.0, scale):.2e} of scale "
          f"({'meets' if err <= 1e-5 else 'above'} 1e-5 absolute)")
    assert err <= rel * max(1.0, scale), tag



def make(scales, seed=0):
    from groupnet_amd.past_encoder import PastEncoder
    torch.manual_seed(seed)
    args = types.SimpleNamespace(hidden_dim=64, hyper_scales=list(scales), past_length=5)
    enc = PastEncoder(args).eval()
    with torch.no_grad():      # default init gives an almost constant embedding; spread it out
        for p in (enc.input_fc.weight, enc.input_fc2.weight, enc.input_fc3.weight, enc.pos_encoder.fc.weight):
            p.mul_(3.0)
    return enc


def test_state_dict_keys_and_affine_composition_cpu():
    enc = make([5, 11])
    keys = list(enc.state_dict().keys())
    assert keys[:6] == ["input_fc.weight", "input_fc.bias", "input_fc2.weight", "input_fc2.bias",
                        "input_fc3.weight", "input_fc3.bias"]
    assert keys[6].startswith("interaction.") and any(k.startswith("interaction_hyper2.") for k in keys)
    assert keys[-3:] == ["pos_encoder.pe", "pos_encoder.fc.weight", "pos_encoder.fc.bias"]
    assert torch.equal(enc.pos_encoder.pe, PO.build_pos_enc(200, 64))
    # the composed affine map == the layer-by-layer embedding (lines 269-280)
    B, N, T = 7, 11, 5
    x = torch.randn(B * N, T, 4) * 5
    sd = {k: v.detach() for k, v in enc.state_dict().items()}
    M, c = enc._compose(T, N)
    f_aff = (x.reshape(B, N, T * 4) @ M.t()) + c[None]
    f_ref = PO.embed(sd, x, B, N)
    assert float((f_aff - f_ref).abs().max()) <= 2e-5 * float(f_ref.abs().max())
    with pytest.raises(IndexError):
        enc._affine = None
        enc._compose(T, 10)          # add_category hard-codes slot 10 (model/GroupNet_nba.py:261)
    with pytest.raises(ValueError):      # training mode exists now, but there is still no CPU path
        enc.train()(x, B, N)


@pytest.mark.gpu
@pytest.mark.parametrize("scales", [[5, 11], [2, 5, 11], [3]])
def test_past_encoder_matches_oracle(scales):
    enc = make(scales, seed=3)
    sd = {k: v.detach().clone() for k, v in enc.state_dict().items()}
    dev = torch.device("cuda:0")
    enc.to(dev)
    B, N, T = 21, 11, 5
    traj = torch.cumsum(torch.randn(B * N, T, 2), dim=1) + torch.rand(B * N, 1, 2) * 20
    vel = traj[:, 1:] - traj[:, :-1]
    x = torch.cat((traj, torch.cat([vel[:, [0]], vel], dim=1)), dim=-1)     # GroupNet_nba.py:792-797
    # oracle: embedding restatement + pinned MS-HGNN oracle, same host noise stream
    torch.manual_seed(77)
    f, corr = PO.embed_and_affinity(sd, x, B, N)
    sub = lambda pre: {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    Up = [O.draw_uniform(s) for s in O.noise_shapes(B, N, None)]
    inter, _ = O.ms_hgnn_pairwise_forward(sub("interaction."), f, Up, decomposed=True)
    feats, Hs = [f, inter], []
    for name, s in zip(["interaction_hyper.", "interaction_hyper2.", "interaction_hyper3."], scales):
        Uh = [O.draw_uniform(sh) for sh in O.noise_shapes(B, N, s)]
        nf, _, H = O.ms_hgnn_hyper_forward(sub(name), f, corr, s, Uh, decomposed=True)
        feats.append(nf)
        Hs.append(H)
    want = torch.cat(feats, dim=-1).view(B * N, -1)
    torch.manual_seed(77)
    with torch.no_grad():
        out, new_H = enc(x.to(dev), B, N)
    assert out.shape == want.shape
    close_to(out, want, f"PastEncoder vs oracle, scales {scales}")
    if len(scales) > 1:
        # H is built from the kernel's own affinity; these seeded inputs have well-separated neighbours
        assert torch.equal(new_H.cpu(), torch.cat(Hs, dim=1))
    else:
        assert new_H is None


def make_future(scales, seed=0):
    from groupnet_amd.past_encoder import FutureEncoder
    torch.manual_seed(seed)
    args = types.SimpleNamespace(hidden_dim=64, hyper_scales=list(scales), past_length=5, future_length=10, zdim=32)
    enc = FutureEncoder(args).eval()
    with torch.no_grad():
        for p in (enc.input_fc.weight, enc.input_fc2.weight, enc.input_fc3.weight, enc.pos_encoder.fc.weight):
            p.mul_(3.0)
        enc.out_mlp.affine_layers[0].weight.mul_(20.0)      # N(0, 0.01) init: lift the head out of the noise floor
        enc.qz_layer.weight.mul_(20.0)
    return enc


def test_future_encoder_state_dict_layout_cpu():
    """Registration order and shapes of model/GroupNet_nba.py:317-375 (the strict load of the reference state_dict in the golden tests pins them)."""
    enc = make_future([5, 11])
    sd = enc.state_dict()
    keys = list(sd.keys())
    assert keys[:6] == ["input_fc.weight", "input_fc.bias", "input_fc2.weight", "input_fc2.bias",
                        "input_fc3.weight", "input_fc3.bias"]
    assert keys[-7:] == ["pos_encoder.pe", "pos_encoder.fc.weight", "pos_encoder.fc.bias",
                         "out_mlp.affine_layers.0.weight", "out_mlp.affine_layers.0.bias", "qz_layer.weight",
                         "qz_layer.bias"]
    assert tuple(sd["input_fc2.weight"].shape) == (64, 640)                      # future_length 10
    assert tuple(sd["out_mlp.affine_layers.0.weight"].shape) == (128, 4 * 2 * 64)
    assert tuple(sd["qz_layer.weight"].shape) == (64, 128)
    assert tuple(sd["interaction_hyper.spatial_embedding.weight"].shape) == (16, 2)   # embedding_dim=16, :339


def _inputs(B, N, T, gen):
    traj = torch.cumsum(torch.randn(B * N, T, 2, generator=gen), dim=1) + torch.rand(B * N, 1, 2, generator=gen) * 20
    vel = traj[:, 1:] - traj[:, :-1]
    return torch.cat((traj, torch.cat([vel[:, [0]], vel], dim=1)), dim=-1)


@pytest.mark.gpu
@pytest.mark.parametrize("scales", [[5, 11], [2, 5, 11]])
def test_future_encoder_matches_oracle(scales):
    enc = make_future(scales, seed=5)
    sd = {k: v.detach().clone() for k, v in enc.state_dict().items()}
    dev = torch.device("cuda:0")
    enc.to(dev)
    g = torch.Generator().manual_seed(9)
    B, N, T = 13, 11, 10
    x = _inputs(B, N, T, g)
    past = torch.randn(B * N, 64 * (2 + len(scales)), generator=g)
    torch.manual_seed(123)
    want = PO.future_encoder_forward(sd, x, B, N, past, scales)
    torch.manual_seed(123)
    with torch.no_grad():
        got = enc(x.to(dev), B, N, past.to(dev))
    assert got.shape == (B * N, 64)
    close_to(got, want, "encoder output")


def _encoder_gradients(which, scales, rows, x_all, noise_all, R_all, past_all, N, gate):
    """Gradients of every used parameter of an encoder, HIP training path vs torch autograd on the oracle, on the scenes
    `rows` of the batch; returns (worst error of a parameter relative to its scale, number of used parameters)."""
    enc = make(scales, seed=8) if which == "past" else make_future(scales, seed=8)
    enc.pos_encoder.dropout.p = 0.0
    state = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and k != "pos_encoder.pe")
             for k, v in enc.state_dict().items()}
    dev = torch.device("cuda:0")
    B = len(rows)
    node_rows = (rows[:, None] * N + torch.arange(N)[None]).reshape(-1)
    x, R, past = x_all[node_rows], R_all[node_rows], past_all[node_rows]
    noise = [[u[0][rows]] for u in noise_all]
    if which == "past":
        want, _ = PO.encode(state, x, B, N, scales, noise)
    else:
        want = PO.future_encoder_forward(state, x, B, N, past, scales, noise)
    (want * R).sum().backward()
    enc.to(dev).train()
    from groupnet_amd import MS_HGNN_batch as M
    it = iter([u[0].to(dev) for u in noise])
    orig = M._draw_uniform
    M._draw_uniform = lambda shape, device: next(it)          # hand the modules the oracle's uniforms
    try:
        out = enc(x.to(dev), B, N)[0] if which == "past" else enc(x.to(dev), B, N, past.to(dev))
    finally:
        M._draw_uniform = orig
    close_to(out, want, f"{which} encoder (training mode, {B} scenes) vs oracle")
    (out * R.to(dev)).sum().backward()
    used = [k for k, v in state.items() if v.grad is not None and float(v.grad.abs().max()) > 0]
    assert "input_fc.weight" in used and "pos_encoder.fc.weight" in used
    hip = dict(enc.named_parameters())
    gmax = max(float(state[k].grad.abs().max()) for k in used)
    worst, where = 0.0, ""
    for k in used:
        a, b = hip[k].grad, state[k].grad
        assert a is not None, k
        # relative to the parameter's own gradient, floored at 1 % of the largest gradient in the model: some
        # directions (the attention bias under a full softmax) have an exactly-zero gradient and only carry noise
        scale = max(float(b.abs().max()), 1e-2 * gmax)
        e = float((a.cpu() - b).abs().max()) / scale
        if e > worst:
            worst, where = e, k
        assert e <= gate, (k, e, gate)
    return worst, where, len(used)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["past", "future"])
def test_encoder_training_gradients(which):
    """Training path of the encoders (embedding on the HIP GEMM with HIP backward, modules through
    MSHGNNFunction, head): gradients of every used parameter against torch autograd on the oracle, dropout
    probability 0 so that both sides are deterministic, same noise.  Gates as the module-level backward tests
    (tests/relu_probe.py): the whole batch at 2e-3 — a ReLU unit within rounding of zero may be on in one forward and off
    in the other, which switches its whole backward contribution — and the batch's CLEAN scenes alone (no ReLU
    pre-activation of the oracle's forward within 2e-6 of zero) at 2e-5 of the parameter's gradient scale."""
    from relu_probe import relu_probe
    scales = [3, 11]
    g = torch.Generator().manual_seed(21)
    B, N, T = 24, 11, 5 if which == "past" else 10
    x = _inputs(B, N, T, g)
    noise = [[torch.rand(s, generator=g)] for s in [(B, N * N, 6), (B, N, 10), (B, 1, 10)]]
    R = torch.randn(B * N, 64 * 4 if which == "past" else 64, generator=g)
    past = torch.randn(B * N, 64 * 4, generator=g)
    # which scenes are clean: the oracle's forward on the whole batch under the probe
    probe_enc = make(scales, seed=8) if which == "past" else make_future(scales, seed=8)
    sd = {k: v.detach().clone() for k, v in probe_enc.state_dict().items()}
    with torch.no_grad(), relu_probe(B) as probe:
        if which == "past":
            PO.encode(sd, x, B, N, scales, noise)
        else:
            PO.future_encoder_forward(sd, x, B, N, past, scales, noise)
    clean = probe.clean()
    rows_all = torch.arange(B)
    w_all, k_all, n_used = _encoder_gradients(which, scales, rows_all, x, noise, R, past, N, 2e-3)
    msg = (f"\n{which} encoder gradients vs torch autograd on the oracle ({n_used} parameters): whole batch worst {w_all:.2e} "
           f"({k_all}), gate 2e-3; {int(clean.sum())}/{B} scenes clean ({probe.units} ReLU units per scene probed)")
    assert int(clean.sum()) >= 2, "seed gives too few clean scenes for the tight gate"
    w_c, k_c, _ = _encoder_gradients(which, scales, rows_all[clean], x, noise, R, past, N, 2e-5)
    print(msg + f"; clean scenes alone worst {w_c:.2e} ({k_c}), gate 2e-5")


# ---- pinned by the reference itself: goldens from tests/golden/make_golden_past_encoder.py --------------------
def _pe_golden(name):
    import numpy as np
    import os
    with np.load(os.path.join(os.path.dirname(__file__), "golden", f"past_encoder_{name}.npz")) as z:
        c = {k: z[k].copy() for k in z.files}
    sd = {k[3:]: torch.from_numpy(v) for k, v in c.items() if k.startswith("sd/")}
    return c, sd


@pytest.mark.parametrize("name", ["s5_11_b9", "s2_5_11_b4", "nba_s5_11_b10"])
def test_past_encoder_oracle_matches_reference_goldens(name):
    """The restatement (oracle/past_encoder_oracle.py + the pinned MS-HGNN oracle) reproduces what the
    REFERENCE's own `PastEncoder.forward` returned (its class definitions executed from /root/reference by the
    golden script), given the recorded uniforms: this pins the front-end oracle."""
    c, sd = _pe_golden(name)
    B, scales = int(c["B"]), [int(s) for s in c["scales"]]
    noise = [[torch.from_numpy(c[f"U{i}"])] for i in range(1 + len(scales))]
    out, Hs = PO.encode(sd, torch.from_numpy(c["x"]), B, 11, scales, noise)
    want = torch.from_numpy(c["output_feature"])
    assert float((out - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))
    assert torch.equal(torch.cat(Hs, dim=1), torch.from_numpy(c["new_H"]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["s5_11_b9", "s2_5_11_b4", "nba_s5_11_b10"])
def test_past_encoder_matches_reference_goldens(name):
    """The HIP `PastEncoder` with the reference's state_dict (strict load), the reference's inputs and the same
    seeded host noise stream against the reference's own outputs.  `nba_s5_11_b10` = the 10 real NBA scenes the
    reference ships (datasets/nba/test_nba.npy, BASELINE config 1's data), prepared as `GroupNet.inference`
    prepares them, through the reference's own PastEncoder class."""
    from groupnet_amd.past_encoder import PastEncoder
    c, sd = _pe_golden(name)
    B, scales = int(c["B"]), [int(s) for s in c["scales"]]
    enc = PastEncoder(types.SimpleNamespace(hidden_dim=64, hyper_scales=scales, past_length=5)).eval()
    enc.load_state_dict(sd, strict=True)
    dev = torch.device("cuda:0")
    enc.to(dev)
    torch.manual_seed(int(c["seed"]))
    with torch.no_grad():
        out, new_H = enc(torch.from_numpy(c["x"]).to(dev), B, 11)
    want = torch.from_numpy(c["output_feature"])
    close_to(out, want, f"PastEncoder vs reference golden {name}")
    assert torch.equal(new_H.cpu(), torch.from_numpy(c["new_H"]))


def _fe_golden():
    import numpy as np
    import os
    with np.load(os.path.join(os.path.dirname(__file__), "golden", "future_encoder_noscale_b6.npz")) as z:
        c = {k: z[k].copy() for k in z.files}
    return c, {k[3:]: torch.from_numpy(v) for k, v in c.items() if k.startswith("sd/")}


def test_future_encoder_oracle_matches_reference_golden():
    """`FutureEncoder.forward` of the reference only runs without hyper scales (it raises on its 3-tuple unpack
    otherwise, model/GroupNet_nba.py:408-413); that configuration — front-end, pairwise module, out_mlp,
    qz_layer — is pinned here by the reference's own output."""
    c, sd = _fe_golden()
    B = int(c["B"])
    got = PO.future_encoder_forward(sd, torch.from_numpy(c["x"]), B, 11, torch.from_numpy(c["past"]), [],
                                    [[torch.from_numpy(c["U0"])]])
    want = torch.from_numpy(c["q_z_params"])
    assert float((got - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))


@pytest.mark.gpu
def test_future_encoder_matches_reference_golden():
    from groupnet_amd.past_encoder import FutureEncoder
    c, sd = _fe_golden()
    B = int(c["B"])
    enc = FutureEncoder(types.SimpleNamespace(hidden_dim=64, hyper_scales=[], past_length=5, future_length=10,
                                              zdim=32)).eval()
    enc.load_state_dict(sd, strict=True)
    dev = torch.device("cuda:0")
    enc.to(dev)
    torch.manual_seed(int(c["seed"]))
    with torch.no_grad():
        got = enc(torch.from_numpy(c["x"]).to(dev), B, 11, torch.from_numpy(c["past"]).to(dev))
    want = torch.from_numpy(c["q_z_params"])
    close_to(got, want, "encoder output")


@pytest.mark.gpu
def test_past_encoder_training_mode_matches_reference_golden():
    """Training mode (dropout of the positional encoder active): with host noise the HIP encoder draws the
    dropout mask and the Gumbel uniforms from the global CPU generator in the reference's order, so a seeded
    forward reproduces the reference module's training-mode output."""
    from groupnet_amd.past_encoder import PastEncoder
    c, sd = _pe_golden("train_s5_11_b6")
    B, scales = int(c["B"]), [int(s) for s in c["scales"]]
    enc = PastEncoder(types.SimpleNamespace(hidden_dim=64, hyper_scales=scales, past_length=5))
    enc.load_state_dict(sd, strict=True)
    dev = torch.device("cuda:0")
    enc.to(dev).train()
    torch.manual_seed(int(c["seed"]))
    out, new_H = enc(torch.from_numpy(c["x"]).to(dev), B, 11)
    assert out.requires_grad
    want = torch.from_numpy(c["output_feature"])
    close_to(out, want, "training-mode golden")
    assert torch.equal(new_H.cpu(), torch.from_numpy(c["new_H"]))


@pytest.mark.gpu
def test_graphed_past_encoder_replays_eager_with_fresh_noise():
    """One hipGraph for the whole encoder: replay k equals an eager forward at Philox position k * draws, and
    consecutive replays draw different noise."""
    import groupnet_amd as G
    from groupnet_amd.graphs import GraphedPastEncoder
    enc = make([5, 11], seed=2)
    dev = torch.device("cuda:0")
    enc.to(dev)
    B, N = 16, 11
    x = _inputs(B, N, 5, torch.Generator().manual_seed(3)).to(dev)
    g = GraphedPastEncoder(enc, B, N, seed=21)
    out0, H0 = [t.clone() for t in g(x)]
    out1, _ = [t.clone() for t in g()]
    assert float((out0[:, 64:] - out1[:, 64:]).abs().max()) > 0          # fresh Gumbel noise
    assert torch.equal(out0[:, :64], out1[:, :64])                        # the embedding itself is noise-free
    for k, want in ((0, out0), (1, out1)):
        counter = torch.full((1,), k * g.draws_per_step, dtype=torch.int64, device=dev)
        G.set_noise_mode("device", seed=21, offset=0, counter=counter)
        try:
            with torch.no_grad():
                eager, H = enc(x, B, N)
        finally:
            G.set_noise_mode("host")
        assert torch.equal(eager, want) and torch.equal(H, H0)
