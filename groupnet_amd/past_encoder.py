"""Drop-in `PastEncoder` (model/GroupNet_nba.py:198-315) on top of the HIP path — SURVEY.md §8f rank 1,
the step immediately before the hot path and the producer of `h_states` / `corr`.

Same constructor (`PastEncoder(args, in_dim=4)` reading `args.hidden_dim`, `args.hyper_scales`,
`args.past_length`), same `forward(inputs, batch_size, agent_num)` -> `(output_feature (B*N, 64*(2+S)),
new_H)`, same `state_dict` keys (`input_fc`, `input_fc2`, `input_fc3`, `pos_encoder.fc`,
`pos_encoder.pe`, `interaction`, `interaction_hyper`, `interaction_hyper2`, `interaction_hyper3`).

MI355X-first: in eval mode (dropout = identity) the embedding lines 269-280 — input_fc per time step,
concat with the positional encoding, fc, flatten, input_fc2, concat with the agent-slot one-hot,
input_fc3 — contain no non-linearity, so they ARE one affine map per agent slot,
    f[b,n] = M x[b,n] + c[n],   x = the T*in_dim raw inputs of the agent,
M (64 x T*in_dim) and c (N x 64) composed once per parameter version in fp64.  That map is evaluated
inside the affinity+top-k launch (`gn_affinity_topk_f32`, `extras.x_raw`), so the whole front-end costs
no launch and f never makes an extra trip through HBM.  Training mode (dropout active) is not affine
and is refused.

Parity of this block is UNPINNED against the reference (GroupNet_nba.py cannot be imported in the build
container; see oracle/past_encoder_oracle.py); it is tested against that restatement.
"""
from __future__ import annotations

import math
from typing import Tuple

import torch
import torch.nn as nn

from . import ops
from .MS_HGNN_batch import (MS_HGNN_hyper, MS_HGNN_oridinary, _check_forward_only, _draw_uniform, _param_key,
                            run_message_passing)

Tensor = torch.Tensor


class PositionalAgentEncoding(nn.Module):
    """Parameter/buffer container of the reference class (model/GroupNet_nba.py:156-195)."""

    def __init__(self, d_model, dropout=0.1, max_t_len=200, concat=True):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        self.concat = concat
        self.d_model = d_model
        if concat:
            self.fc = nn.Linear(2 * d_model, d_model)
        pe = torch.zeros(max_t_len, d_model)
        position = torch.arange(0, max_t_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer('pe', pe)


class PastEncoder(nn.Module):
    def __init__(self, args, in_dim=4):
        super().__init__()
        self.args = args
        self.model_dim = args.hidden_dim
        self.scale_number = len(args.hyper_scales)
        if self.scale_number > 3:
            raise ValueError("PastEncoder takes at most 3 hyper scales (model/GroupNet_nba.py:218-248); "
                             "use groupnet_amd.multiscale.MultiScaleHGNN for more")
        d = self.model_dim
        self.input_fc = nn.Linear(in_dim, d)
        self.input_fc2 = nn.Linear(d * args.past_length, d)
        self.input_fc3 = nn.Linear(d + 3, d)
        self.interaction = MS_HGNN_oridinary(embedding_dim=16, h_dim=d, mlp_dim=64, bottleneck_dim=d, batch_norm=0,
                                             nmp_layers=1)
        names = ["interaction_hyper", "interaction_hyper2", "interaction_hyper3"]
        for name, s in zip(names, args.hyper_scales):
            setattr(self, name, MS_HGNN_hyper(embedding_dim=d, h_dim=d, mlp_dim=64, bottleneck_dim=d, batch_norm=0,
                                              nmp_layers=1, scale=s))
        self._hyper_names = names[:self.scale_number]
        self.pos_encoder = PositionalAgentEncoding(d, 0.1, concat=True)
        self._affine = None

    # -- the embedding as one affine map ---------------------------------------------------------------
    def _front_params(self):
        return [self.input_fc.weight, self.input_fc.bias, self.input_fc2.weight, self.input_fc2.bias,
                self.input_fc3.weight, self.input_fc3.bias, self.pos_encoder.fc.weight, self.pos_encoder.fc.bias]

    def _compose(self, T: int, N: int) -> Tuple[Tensor, Tensor]:
        """(M (D, T*in_dim), c (N, D)) with f = M x + c[n]; lines 269-280 composed in fp64."""
        key = (_param_key(self._front_params()), T, N, self.pos_encoder.pe.data_ptr())
        if self._affine is None or self._affine[0] != key:
            with torch.no_grad():
                D = self.model_dim
                dd = torch.float64
                Win, b_in = self.input_fc.weight.to(dd), self.input_fc.bias.to(dd)
                Wfc, bfc = self.pos_encoder.fc.weight.to(dd), self.pos_encoder.fc.bias.to(dd)
                W2, b2 = self.input_fc2.weight.to(dd), self.input_fc2.bias.to(dd)
                W3, b3 = self.input_fc3.weight.to(dd), self.input_fc3.bias.to(dd)
                pe = self.pos_encoder.pe[:T].to(dd)
                Wa, Wb = Wfc[:, :D], Wfc[:, D:]
                W3a, W3c = W3[:, :D], W3[:, D:]
                in_dim = Win.shape[1]
                M = torch.zeros(D, T * in_dim, dtype=dd, device=Win.device)
                c0 = b2.clone()
                for t in range(T):
                    W2t = W2[:, t * D:(t + 1) * D]
                    M[:, t * in_dim:(t + 1) * in_dim] = W3a @ W2t @ Wa @ Win
                    c0 = c0 + W2t @ (Wa @ b_in + Wb @ pe[t] + bfc)
                c0 = W3a @ c0 + b3
                cat = torch.zeros(N, 3, dtype=dd, device=Win.device)    # add_category, :252-264
                cat[0:5, 0] = 1
                cat[5:10, 1] = 1
                cat[10, 2] = 1                                            # IndexError for N <= 10, as the reference
                c = c0[None, :] + cat @ W3c.t()
                self._affine = (key, M.float().contiguous(), c.float().contiguous())
        return self._affine[1], self._affine[2]

    def forward(self, inputs, batch_size, agent_num):
        _check_forward_only(inputs)
        if self.training and self.pos_encoder.dropout.p > 0:
            raise RuntimeError("groupnet_amd.PastEncoder is inference-only: call .eval() (with dropout active the "
                               "embedding is not the affine map the fused kernel evaluates)")
        ops._req(inputs, "inputs", (batch_size * agent_num, None, self.input_fc.in_features))
        B, N, D = batch_size, agent_num, self.model_dim
        T = inputs.shape[1]
        if T * D != self.input_fc2.in_features:
            raise ValueError(f"inputs: {T} time steps, but input_fc2 was built for past_length={self.args.past_length}")
        M, c = self._compose(T, N)
        x_raw = inputs.reshape(B, N, T * inputs.shape[2])
        S = self.scale_number
        hypers = [getattr(self, n) for n in self._hyper_names]
        final = torch.empty((B, N, D * (2 + S)), dtype=inputs.dtype, device=inputs.device)
        cols = [final[..., D * (1 + i):D * (2 + i)] for i in range(1 + S)]
        scales = [m.scale for m in hypers] or [N]     # the launch needs >= 1 scale; N = the cheap all-ones edge
        _, Hs, new_H, f = ops.affinity_topk(None, scales, want_corr=False, f_out=final[..., :D],
                                            want_H_cat=S > 1, embed=(x_raw, M, c))
        if S == 0:
            Hs, new_H = [], None
        elif S == 1:
            new_H = None      # the reference only builds new_H from two scales on (:296); its S==1 path raises
        mods = [self.interaction, *hypers]
        noise = [[_draw_uniform((B, N * N, 6), f.device)]] + [[_draw_uniform((B, H.shape[1], 10), f.device)] for H in Hs]
        run_message_passing(mods, [f] * (1 + S), [None, *Hs], noise, cols)
        return final.view(B * N, -1), new_H
