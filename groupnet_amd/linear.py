"""`y = x W^T + b` on the HIP GEMM with a HIP backward — the dense layers AROUND the message-passing
modules in the reference's encoders (model/GroupNet_nba.py:269-280 front-end, :374-376/:431-436 head)
when they have to be differentiable (training mode: dropout sits between them, so the fused affine
front-end of eval mode does not apply)."""
from __future__ import annotations

import torch

from .backward import GemmBatch

Tensor = torch.Tensor


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Tensor) -> Tensor:
        x = x.contiguous()
        y = torch.empty((x.shape[0], weight.shape[0]), dtype=x.dtype, device=x.device)
        gb = GemmBatch()
        gb.add(x, weight.detach(), y, tB=True, bias=bias.detach())
        gb.run()
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        gb = GemmBatch()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = gb.add(dy, weight.detach(), torch.empty_like(x))
        dW = db = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            z = torch.zeros(weight.numel() + weight.shape[0], dtype=x.dtype, device=x.device)
            dW, db = z[:weight.numel()].view_as(weight), z[weight.numel():]
            gb.add(dy, x, dW, tA=True, accum=True, colsum=db)     # dW = dy^T x, db = column sums of dy
        gb.run()
        return dx, dW, db


def hip_linear(x: Tensor, layer: torch.nn.Linear) -> Tensor:
    """layer(x) for x (..., in_features) through `gn_gemm_grouped_f32`, differentiable."""
    lead = x.shape[:-1]
    y = _LinearFn.apply(x.reshape(-1, x.shape[-1]), layer.weight, layer.bias)
    return y.view(*lead, layer.out_features)
