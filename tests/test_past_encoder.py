"""PastEncoder front-end.  Parity here is against the oracle RESTATEMENT only
(oracle/past_encoder_oracle.py: the reference file cannot be imported in the build container)."""
import types

import pytest
import torch

from oracle import ms_hgnn_oracle as O
from oracle import past_encoder_oracle as PO


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
    with pytest.raises(RuntimeError):
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
    scale = float(want.abs().max())
    assert float((out.cpu() - want).abs().max()) <= 1e-5 * max(1.0, scale)
    if len(scales) > 1:
        # H is built from the kernel's own affinity; these seeded inputs have well-separated neighbours
        assert torch.equal(new_H.cpu(), torch.cat(Hs, dim=1))
    else:
        assert new_H is None
