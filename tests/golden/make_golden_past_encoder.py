#!/usr/bin/env python3
"""Golden vectors for `PastEncoder.forward`, produced by the REFERENCE's own class.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_past_encoder.py

`model/GroupNet_nba.py` cannot be imported as a module here (its top-level `from tkinter import TRUE` and, via
`model/utils.py`, `import glob2` name packages this image lacks), but `PositionalAgentEncoding` and `PastEncoder`
use neither.  This script parses the file, takes exactly those two class definitions — unmodified, straight from
/root/reference at run time, nothing is copied into the repo — and executes them in a namespace holding what
their bodies reference (torch, nn, F, np and the reference's own MS_HGNN classes, which import normally).  It
then runs the reference `PastEncoder.forward` on seeded inputs and stores the state_dict, the inputs, the
uniforms the forward drew (torch.rand, in call order) and the outputs.  Only data is written.
"""
import ast
import os
import sys
import types
import warnings

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
warnings.filterwarnings("ignore")
from model import MS_HGNN_batch as ref  # noqa: E402

sys.path.insert(0, OUT)
from make_golden import Recorder  # noqa: E402


def reference_classes():
    """`PastEncoder` and `FutureEncoder` of the reference (with `MLP2`, `PositionalAgentEncoding` and
    model/utils.py's `initialize_weights`, which they use), executed from the reference's own source text."""
    ns = dict(torch=torch, nn=torch.nn, F=torch.nn.functional, np=np, MS_HGNN_oridinary=ref.MS_HGNN_oridinary,
              MS_HGNN_hyper=ref.MS_HGNN_hyper, MLP=ref.MLP)
    for rel, kind, wanted in (("model/utils.py", ast.FunctionDef, ("initialize_weights",)),
                              ("model/GroupNet_nba.py", ast.ClassDef,
                               ("MLP2", "PositionalAgentEncoding", "PastEncoder", "FutureEncoder"))):
        path = os.path.join(REF, rel)
        tree = ast.parse(open(path).read())
        body = [n for n in tree.body if isinstance(n, kind) and n.name in wanted]
        assert [n.name for n in body] == list(wanted), [n.name for n in body]
        exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns["PastEncoder"], ns["FutureEncoder"]


def run(name, scales, B, seed, train=False, nba=False):
    """train=True: the module in training mode, i.e. with the dropout of the positional encoder active
    (model/GroupNet_nba.py:195; mask drawn on the global CPU generator before the modules' uniforms)."""
    PastEncoder, _ = reference_classes()
    torch.manual_seed(seed)
    args = types.SimpleNamespace(hidden_dim=64, hyper_scales=list(scales), past_length=5)
    enc = PastEncoder(args).train(train)
    with torch.no_grad():      # default init gives an almost constant embedding and flat attention: spread them out
        for p in (enc.input_fc.weight, enc.input_fc2.weight, enc.input_fc3.weight, enc.pos_encoder.fc.weight):
            p.mul_(3.0)
        for n, p in enc.named_parameters():
            if "attention_mlp" in n or "MLP_distribution" in n or "MLP_factor" in n:
                p.mul_(3.0)
    N, T = 11, 5
    if nba:
        # the 10 NBA scenes shipped with the reference, prepared as its loader and `GroupNet.inference` do:
        # feet -> metres (data/dataloader_nba.py:36), (scene, agent, time, xy) (:49), the first past_length steps
        a = np.load(os.path.join(REF, "datasets/nba/test_nba.npy")).astype(np.float32) / np.float32(94 / 28)
        traj = torch.from_numpy(np.transpose(a, (0, 2, 1, 3))[:, :, :T, :].copy()).reshape(-1, T, 2)
        B = a.shape[0]
    else:
        g = torch.Generator().manual_seed(seed + 1)
        traj = torch.cumsum(torch.randn(B * N, T, 2, generator=g), dim=1) + torch.rand(B * N, 1, 2, generator=g) * 20
    vel = traj[:, 1:] - traj[:, :-1]
    x = torch.cat((traj, torch.cat([vel[:, [0]], vel], dim=1)), dim=-1)          # (B*N, T, 4), GroupNet_nba.py:792-797
    torch.manual_seed(seed + 2)
    with torch.no_grad(), Recorder() as r:
        out, new_H = enc(x, B, N)
    rec = {"sd/" + k: v.detach().numpy() for k, v in enc.state_dict().items()}
    rec.update(x=x.numpy(), B=np.int64(B), scales=np.asarray(scales), seed=np.int64(seed + 2), output_feature=out.numpy(),
               new_H=new_H.numpy())
    for i, u in enumerate(r.draws):
        rec[f"U{i}"] = u.numpy()
    np.savez_compressed(os.path.join(OUT, f"past_encoder_{name}.npz"), **rec)
    print(name, tuple(out.shape), tuple(new_H.shape), len(r.draws), "draws")


def run_future(name, B, seed):
    """`FutureEncoder.forward` raises on its own 3-tuple unpack as soon as a hyper scale exists
    (model/GroupNet_nba.py:408-413), so the reference can only produce a golden with hyper_scales = []: the
    front-end, the pairwise module and the head (out_mlp, qz_layer)."""
    _, FutureEncoder = reference_classes()
    torch.manual_seed(seed)
    args = types.SimpleNamespace(hidden_dim=64, hyper_scales=[], past_length=5, future_length=10, zdim=32)
    enc = FutureEncoder(args).eval()
    with torch.no_grad():
        for p in (enc.input_fc.weight, enc.input_fc2.weight, enc.input_fc3.weight, enc.pos_encoder.fc.weight):
            p.mul_(3.0)
        enc.out_mlp.affine_layers[0].weight.mul_(20.0)      # N(0, 0.01) init: lift the head out of the noise floor
        enc.qz_layer.weight.mul_(20.0)
    N, T = 11, 10
    g = torch.Generator().manual_seed(seed + 1)
    traj = torch.cumsum(torch.randn(B * N, T, 2, generator=g), dim=1) + torch.rand(B * N, 1, 2, generator=g) * 20
    vel = traj[:, 1:] - traj[:, :-1]
    x = torch.cat((traj, torch.cat([vel[:, [0]], vel], dim=1)), dim=-1)
    past = torch.randn(B * N, 128, generator=g)
    torch.manual_seed(seed + 2)
    with torch.no_grad(), Recorder() as r:
        q = enc(x, B, N, past)
    rec = {"sd/" + k: v.detach().numpy() for k, v in enc.state_dict().items()}
    rec.update(x=x.numpy(), past=past.numpy(), B=np.int64(B), seed=np.int64(seed + 2), q_z_params=q.numpy())
    for i, u in enumerate(r.draws):
        rec[f"U{i}"] = u.numpy()
    np.savez_compressed(os.path.join(OUT, f"future_encoder_{name}.npz"), **rec)
    print(name, tuple(q.shape), len(r.draws), "draws")


if __name__ == "__main__":
    run_future("noscale_b6", 6, 703)
    run("s5_11_b9", [5, 11], 9, 701)
    run("train_s5_11_b6", [5, 11], 6, 704, train=True)
    run("nba_s5_11_b10", [5, 11], 10, 705, nba=True)      # BASELINE config 1's data: datasets/nba/test_nba.npy
    run("s2_5_11_b4", [2, 5, 11], 4, 702)
