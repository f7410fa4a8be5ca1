"""CPU oracle for the trajectory encoders.  TEST INFRASTRUCTURE ONLY.

PINNED: `model/GroupNet_nba.py` cannot be imported as a module in the build container (its line 2 imports
`tkinter`, `model/utils.py:8` imports `glob2`; neither exists here), but the classes this file restates use
neither.  `tests/golden/make_golden_past_encoder.py` executes the reference's own, unmodified class definitions
(`PositionalAgentEncoding`, `PastEncoder`, `MLP2`, `FutureEncoder`, `initialize_weights`) straight from
/root/reference and stores their outputs; `tests/test_past_encoder.py` checks this restatement against them
(<= 2e-6): `PastEncoder.forward` for hyper_scales [5,11] and [2,5,11], `FutureEncoder.forward` for
hyper_scales [] — the only configuration in which the reference's FutureEncoder runs at all (with a hyper
scale it raises on its own 3-tuple unpack, model/GroupNet_nba.py:408-413).  FutureEncoder WITH hyper scales is
therefore pinned only piecewise (its front-end, modules and head are each pinned; their composition follows
the "take node_feat" reading of :408-413).  Training-mode dropout masks are unpinned (the reference draws them
on the CPU generator).

What it restates: `PositionalAgentEncoding` (model/GroupNet_nba.py:156-195), the embedding lines of
`PastEncoder.forward` (:266-286) in eval mode (dropout is the identity), the module calls and concat
(:290-311), and `FutureEncoder.forward` (:393-438).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]


def build_pos_enc(max_len: int, d_model: int) -> Tensor:
    """`PositionalAgentEncoding.build_pos_enc` (model/GroupNet_nba.py:168-174): sin on even, cos on odd
    feature indices, frequencies 10000^(-2i/d)."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def category_onehot(N: int, like: Tensor) -> Tensor:
    """`PastEncoder.add_category` (model/GroupNet_nba.py:252-264): rows 0..4 -> team A, 5..9 -> team B,
    row 10 -> ball, hard-coded (so N <= 10 raises IndexError exactly as the reference does)."""
    category = torch.zeros(N, 3).type_as(like)
    category[0:5, 0] = 1
    category[5:10, 1] = 1
    category[10, 2] = 1
    return category


def embed(state: State, inputs: Tensor, batch_size: int, agent_num: int) -> Tensor:
    """`PastEncoder.forward` lines 269-280, eval mode: inputs (B*N, T, in_dim) -> ftraj_input (B,N,D)."""
    D = state["input_fc.weight"].shape[0]
    T = inputs.shape[1]
    tf_in = F.linear(inputs, state["input_fc.weight"], state["input_fc.bias"]).view(batch_size * agent_num, T, D)  # :269
    pe = state["pos_encoder.pe"][0:T, :][None].repeat(batch_size * agent_num, 1, 1)                               # :177-178
    x = torch.cat([tf_in, pe], dim=-1)                                                                             # :190-191
    tf_in_pos = F.linear(x, state["pos_encoder.fc.weight"], state["pos_encoder.fc.bias"])                          # :192 (dropout = id)
    tf_in_pos = tf_in_pos.view(batch_size, agent_num, T, D)                                                        # :273
    ftraj = F.linear(tf_in_pos.contiguous().view(batch_size, agent_num, T * D),
                     state["input_fc2.weight"], state["input_fc2.bias"])                                           # :276-277
    cat = category_onehot(agent_num, ftraj).repeat(batch_size, 1, 1)                                              # :262
    return F.linear(torch.cat((ftraj, cat), dim=-1), state["input_fc3.weight"], state["input_fc3.bias"])          # :279-280


def embed_and_affinity(state: State, inputs: Tensor, batch_size: int, agent_num: int) -> Tuple[Tensor, Tensor]:
    """Lines 269-286: (ftraj_input, feat_corr)."""
    f = embed(state, inputs, batch_size, agent_num)
    q = F.normalize(f, p=2, dim=2)
    return f, torch.matmul(q, q.permute(0, 2, 1))


def encode(state: State, inputs: Tensor, batch_size: int, agent_num: int, scales, noise=None):
    """Lines 269-311 (PastEncoder) / 395-425 (FutureEncoder), eval mode: embedding, affinity, the pairwise
    module, one hyper module per scale, concat -> (final_feature (B*N, D*(2+S)), [H_s]).  `noise`: optional
    list (pairwise first) of lists of uniforms; default draws from the global CPU generator in the
    reference's call order."""
    from . import ms_hgnn_oracle as O
    B, N = batch_size, agent_num
    f, corr = embed_and_affinity(state, inputs, B, N)
    sub = lambda pre: {k[len(pre):]: v for k, v in state.items() if k.startswith(pre)}
    draw = lambda i, shapes: noise[i] if noise is not None else [O.draw_uniform(s) for s in shapes]
    inter, _ = O.ms_hgnn_pairwise_forward(sub("interaction."), f, draw(0, O.noise_shapes(B, N, None)), decomposed=True)
    feats, Hs = [f, inter], []
    for i, (name, s) in enumerate(zip(["interaction_hyper.", "interaction_hyper2.", "interaction_hyper3."], scales)):
        nf, _, H = O.ms_hgnn_hyper_forward(sub(name), f, corr, s, draw(1 + i, O.noise_shapes(B, N, s)), decomposed=True)
        feats.append(nf)
        Hs.append(H)
    return torch.cat(feats, dim=-1).view(B * N, -1), Hs


def future_encoder_forward(state: State, inputs: Tensor, batch_size: int, agent_num: int, past_feature: Tensor,
                           scales, noise=None) -> Tensor:
    """`FutureEncoder.forward` (model/GroupNet_nba.py:393-438) with the 3-tuple unpack of :408-413 read as
    "take node_feat": q_z_params = qz_layer(relu(out_mlp.affine_layers.0(cat(past_feature, final_feature))))."""
    final, _ = encode(state, inputs, batch_size, agent_num, scales, noise)
    h = torch.cat((past_feature, final), dim=-1)                                                              # :428
    h = torch.relu(F.linear(h, state["out_mlp.affine_layers.0.weight"], state["out_mlp.affine_layers.0.bias"]))  # :431, MLP2 :147-150
    return F.linear(h, state["qz_layer.weight"], state["qz_layer.bias"])                                     # :436
