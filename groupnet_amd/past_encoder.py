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
no launch and f never makes an extra trip through HBM.  In training mode (dropout active, or gradients
wanted) the embedding runs layer by layer on the HIP GEMM with its HIP backward, the modules through the
autograd path of `groupnet_amd.backward`.  `FutureEncoder` is the same encoder at the
reference's second call site plus its output head.

Parity: pinned by goldens the reference's own `PastEncoder` / `FutureEncoder` classes produced
(tests/golden/make_golden_past_encoder.py; tests/test_past_encoder.py).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .MS_HGNN_batch import (MS_HGNN_hyper, MS_HGNN_oridinary, _NoiseState, _draw_uniform, _needs_grad, _param_key,
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


class _TrajectoryEncoder(nn.Module):
    """What `PastEncoder` and `FutureEncoder` share (model/GroupNet_nba.py:198-286 and :316-413): the
    embedding front-end, the pairwise module, up to three hyper modules, the concat."""

    def _build_encoder(self, args, in_dim: int, length: int) -> None:
        self.args = args
        self.model_dim = args.hidden_dim
        self.scale_number = len(args.hyper_scales)
        if self.scale_number > 3:
            raise ValueError("the reference encoders take at most 3 hyper scales (model/GroupNet_nba.py:218-248); "
                             "use groupnet_amd.multiscale.MultiScaleHGNN for more")
        d = self.model_dim
        self.input_fc = nn.Linear(in_dim, d)
        self.input_fc2 = nn.Linear(d * length, d)
        self.input_fc3 = nn.Linear(d + 3, d)
        self.interaction = MS_HGNN_oridinary(embedding_dim=16, h_dim=d, mlp_dim=64, bottleneck_dim=d, batch_norm=0,
                                             nmp_layers=1)
        names = ["interaction_hyper", "interaction_hyper2", "interaction_hyper3"]
        emb = self._hyper_embedding_dim
        for name, s in zip(names, args.hyper_scales):
            setattr(self, name, MS_HGNN_hyper(embedding_dim=emb, h_dim=d, mlp_dim=64, bottleneck_dim=d, batch_norm=0,
                                              nmp_layers=1, scale=s))
        self._hyper_names = names[:self.scale_number]
        self.pos_encoder = PositionalAgentEncoding(d, 0.1, concat=True)
        self._length = length
        self._affine = None

    _hyper_embedding_dim = 64

    # -- the embedding as one affine map (eval mode) ------------------------------------------------------
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
                c = c0[None, :] + self._category(N, dd, Win.device) @ W3c.t()
                self._affine = (key, M.float().contiguous(), c.float().contiguous())
        return self._affine[1], self._affine[2]

    @staticmethod
    def _category(N: int, dtype, device) -> Tensor:
        """add_category (model/GroupNet_nba.py:252-264): team A, team B, ball one-hot by agent slot."""
        cat = torch.zeros(N, 3, dtype=dtype, device=device)
        cat[0:5, 0] = 1
        cat[5:10, 1] = 1
        cat[10, 2] = 1                                            # IndexError for N <= 10, as the reference
        return cat

    def _check_inputs(self, inputs: Tensor, B: int, N: int) -> int:
        ops._req(inputs, "inputs", (B * N, None, self.input_fc.in_features))
        T = inputs.shape[1]
        if T * self.model_dim != self.input_fc2.in_features:
            raise ValueError(f"inputs: {T} time steps, but input_fc2 was built for {self._length}")
        return T

    def _embed_autograd(self, inputs: Tensor, B: int, N: int, T: int) -> Tensor:
        """Lines 269-280 layer by layer (dropout active or gradients wanted): every Linear is the HIP GEMM with
        its HIP backward (`hip_linear`); concat / dropout / views are torch glue."""
        from .linear import hip_linear
        D = self.model_dim
        tf_in = hip_linear(inputs.reshape(B * N * T, -1), self.input_fc)                             # :269
        pe = self.pos_encoder.pe[:T].to(inputs.dtype).repeat(B * N, 1)                               # :177-178
        x = hip_linear(torch.cat([tf_in, pe], dim=-1), self.pos_encoder.fc)                          # :190-192
        x = self._dropout(x)                                                                         # :195
        ftraj = hip_linear(x.view(B * N, T * D), self.input_fc2)                                     # :276-277
        cat = self._category(N, inputs.dtype, inputs.device).repeat(B, 1)                            # :262
        return hip_linear(torch.cat((ftraj, cat), dim=-1), self.input_fc3).view(B, N, D)             # :279-280

    def _dropout(self, x: Tensor) -> Tensor:
        """nn.Dropout of the positional encoder.  Host noise mode (the default, reference contract): the mask is
        drawn exactly as the reference's CPU module draws it — torch's dropout on a CPU tensor of the same shape,
        global CPU generator, before the modules' uniforms — and uploaded; device mode: torch's device generator."""
        drop = self.pos_encoder.dropout
        if not (self.training and drop.p > 0):
            return x
        if _NoiseState.mode == "host":
            mask = torch.nn.functional.dropout(torch.ones(x.shape, dtype=x.dtype), drop.p, True)
            return x * mask.to(x.device, non_blocking=True)
        return drop(x)

    def _encode(self, inputs: Tensor, B: int, N: int) -> Tuple[Tensor, Optional[Tensor]]:
        """(final_feature (B,N,64*(2+S)), new_H): embedding, affinity, incidences, all modules, concat."""
        T = self._check_inputs(inputs, B, N)
        D, S = self.model_dim, self.scale_number
        hypers = [getattr(self, n) for n in self._hyper_names]
        scales = [m.scale for m in hypers]
        dropout_on = self.training and self.pos_encoder.dropout.p > 0
        if dropout_on or _needs_grad(self, inputs):
            from .multiscale import multiscale_autograd
            f = self._embed_autograd(inputs, B, N, T)
            final, new_H = multiscale_autograd(self.interaction, hypers, scales, f)
            return final, (new_H if S > 1 else None)
        M, c = self._compose(T, N)
        x_raw = inputs.reshape(B, N, T * inputs.shape[2])
        final = torch.empty((B, N, D * (2 + S)), dtype=inputs.dtype, device=inputs.device)
        cols = [final[..., D * (1 + i):D * (2 + i)] for i in range(1 + S)]
        adv = self.__dict__.get("_advance")      # (device counter, draws per call): set by graphs.GraphedPastEncoder
        _, Hs, new_H, f = ops.affinity_topk(None, scales or [N], want_corr=False, f_out=final[..., :D],
                                            want_H_cat=S > 1, embed=(x_raw, M, c),
                                            counter=adv[0] if adv else None,
                                            counter_add=adv[1] if adv else 0)        # >= 1 scale per launch; N = the
        if S == 0:                                                                     # cheap all-ones edge
            Hs, new_H = [], None
        elif S == 1:
            new_H = None      # the reference only builds new_H from two scales on (:296); its S==1 path raises
        mods = [self.interaction, *hypers]
        noise = [[_draw_uniform((B, N * N, 6), f.device)]] + [[_draw_uniform((B, H.shape[1], 10), f.device)] for H in Hs]
        run_message_passing(mods, [f] * (1 + S), [None, *Hs], noise, cols)
        return final, new_H


class PastEncoder(_TrajectoryEncoder):
    def __init__(self, args, in_dim=4):
        super().__init__()
        self._build_encoder(args, in_dim, args.past_length)

    def forward(self, inputs, batch_size, agent_num):
        """-> (output_feature (B*N, 64*(2+S)), new_H).  Eval / no-grad: the fused path (the embedding is one
        affine map evaluated inside the affinity launch).  Training (dropout active or gradients wanted): the
        embedding layer by layer on the HIP GEMM, the modules through `MSHGNNFunction`."""
        final, new_H = self._encode(inputs, batch_size, agent_num)
        return final.view(batch_size * agent_num, -1), new_H


class MLP2(nn.Module):
    """Parameter container of the reference class (model/GroupNet_nba.py:128-150): Linear + activation per
    hidden size; weights N(0, 0.01), biases 0 (model/utils.py:19-21)."""

    def __init__(self, input_dim, hidden_dims=(128, 128), activation='tanh'):
        super().__init__()
        self.activation = {'tanh': torch.tanh, 'relu': torch.relu, 'sigmoid': torch.sigmoid}[activation]
        self.out_dim = hidden_dims[-1]
        self.affine_layers = nn.ModuleList()
        last = input_dim
        for nh in hidden_dims:
            self.affine_layers.append(nn.Linear(last, nh))
            last = nh
        for m in self.affine_layers:
            nn.init.normal_(m.weight, 0, 0.01)
            nn.init.constant_(m.bias, 0)

    def forward(self, x):
        from .linear import hip_linear
        for affine in self.affine_layers:
            x = self.activation(hip_linear(x, affine))
        return x


class FutureEncoder(_TrajectoryEncoder):
    """Drop-in `FutureEncoder` (model/GroupNet_nba.py:316-438), the posterior q(z | past, future) of training
    — SURVEY.md §8f rank 3: the same two modules at a second call site, after the same front-end, followed by
    `out_mlp` and `qz_layer`.  `forward(inputs, batch_size, agent_num, past_feature)` -> q_z_params
    (B*N, 2*zdim).  The reference unpacks the hyper modules' 3-tuples into two names (:408-413) and therefore
    raises; this class does what those lines mean (take node_feat)."""

    _hyper_embedding_dim = 16      # model/GroupNet_nba.py:339,350,361

    def __init__(self, args, in_dim=4):
        super().__init__()
        self._build_encoder(args, in_dim, args.future_length)
        scale_num = 2 + len(args.hyper_scales)
        self.out_mlp = MLP2(scale_num * 2 * self.model_dim, [128], 'relu')
        self.qz_layer = nn.Linear(self.out_mlp.out_dim, 2 * args.zdim)
        nn.init.normal_(self.qz_layer.weight, 0, 0.01)
        nn.init.constant_(self.qz_layer.bias, 0)

    def forward(self, inputs, batch_size, agent_num, past_feature):
        from .linear import hip_linear
        final, _ = self._encode(inputs, batch_size, agent_num)
        final = final.view(batch_size * agent_num, -1)
        ops._req(past_feature, "past_feature", (batch_size * agent_num, final.shape[1]))
        h = torch.cat((past_feature, final), dim=-1)                                                # :428
        return hip_linear(self.out_mlp(h), self.qz_layer)                                           # :431-436
