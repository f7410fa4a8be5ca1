"""CPU oracle for the GroupNet MS-HGNN hot path.  TEST INFRASTRUCTURE ONLY.

This file is a checker, not a product path: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  ``groupnet_amd`` never does (tests/test_capi_cpu.py::
test_product_never_touches_the_oracle enforces that).

It restates, op for op and in plain fp32 torch on the CPU, what the reference
does in ``model/MS_HGNN_batch.py`` and in the three affinity lines of
``model/GroupNet_nba.py``.  Every function cites the reference lines it
follows (paths relative to /root/reference).  The functions are *functional*:
they take a ``state_dict`` (the reference's key names)
instead of owning parameters, so the very same weights can be fed to the
reference, to this oracle and to the HIP path.

Parity pin: the reference holds no tests or golden vectors for this path
(SURVEY.md §4), so the oracle is pinned by outputs of the reference itself,
generated in the build container by ``tests/golden/make_golden.py`` (which
imports /root/reference/model/MS_HGNN_batch.py unmodified) and committed as
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this file
against every one of them.

Stochastic part: the reference draws ``torch.rand(B,E,K)`` from the global
CPU generator on every ``MLP_dict_softmax.forward`` (MS_HGNN_batch.py:45,
454,466).  Every oracle entry point therefore takes the uniforms ``U``
explicitly; ``draw_uniform`` reproduces the reference's draw.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]

HDIM_EXTEND = 64          # MS_HGNN_batch.py:72,292
EDGE_TYPES_PAIRWISE = 6   # MS_HGNN_batch.py:74
EDGE_TYPES_HYPER = 10     # MS_HGNN_batch.py:294
GUMBEL_TAU = 0.5          # MS_HGNN_batch.py:45
GUMBEL_EPS = 1e-10        # MS_HGNN_batch.py:446


# --------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------
def mlp(state: State, prefix: str, x: Tensor) -> Tensor:
    """``MLP.forward`` with activation='relu', dropout=-1, discrim=False
    (MS_HGNN_batch.py:220-229): Linear, ReLU between layers, none after the last."""
    n_layers = 0
    while f"{prefix}.layers.{n_layers}.weight" in state:
        n_layers += 1
    assert n_layers > 0, prefix
    for i in range(n_layers):
        x = F.linear(x, state[f"{prefix}.layers.{i}.weight"], state[f"{prefix}.layers.{i}.bias"])
        if i != n_layers - 1:
            x = torch.relu(x)
    return x


def draw_uniform(shape: Sequence[int]) -> Tensor:
    """The draw of ``sample_gumbel`` (MS_HGNN_batch.py:454): global CPU generator."""
    return torch.rand(tuple(shape)).float()


def gumbel_softmax_from_uniform(logits: Tensor, U: Tensor, tau: float = GUMBEL_TAU,
                                eps: float = GUMBEL_EPS) -> Tensor:
    """``gumbel_softmax(hard=False)`` (MS_HGNN_batch.py:446-473,517-520) with the
    uniforms given.  ``my_softmax`` transposes axis -1 with 0 and calls the
    implicit-dim softmax, which for a 3-D tensor picks dim 0 of the transposed
    tensor, i.e. the last axis of the original."""
    g = -torch.log(eps - torch.log(U + eps))
    y = logits + g
    return torch.softmax(y / tau, dim=-1)


def affinity(f: Tensor) -> Tensor:
    """``PastEncoder.forward`` affinity lines (model/GroupNet_nba.py:284-286)."""
    q = F.normalize(f, p=2, dim=2)
    return torch.matmul(q, q.permute(0, 2, 1))


def topk_incidence(corr: Tensor, scale: int, like: Optional[Tensor] = None) -> Tensor:
    """``MS_HGNN_hyper.init_adj_attention`` (MS_HGNN_batch.py:372-388).

    Ties: the reference inherits whatever ``torch.topk`` does on the CPU; the
    build defines lowest-index-wins, which is what
    ``topk_incidence_ranked`` below states; the goldens are tie-free."""
    B, N = corr.shape[0], corr.shape[1]
    dtype = corr.dtype if like is None else like.dtype
    if scale == N:
        return torch.ones(B, 1, N, dtype=dtype)
    k = max(int(scale), 1)
    _, idx = torch.topk(corr, dim=2, k=k, largest=True)
    H = torch.zeros(B, N, N, dtype=dtype)
    return H.scatter(2, idx, 1)


def topk_incidence_ranked(corr: Tensor, scale: int) -> Tensor:
    """Same selection stated as a rank: column c of row r is chosen iff fewer
    than k entries of the row beat it, where j beats c when v_j > v_c, or
    v_j == v_c and j < c (lowest index wins ties); NaN ranks above every
    number, as in ``torch.topk``.  This is the rule the HIP kernel implements."""
    B, N = corr.shape[0], corr.shape[1]
    if scale == N:
        return torch.ones(B, 1, N, dtype=corr.dtype)
    k = max(int(scale), 1)
    if k > N:
        raise RuntimeError("selected index k out of range")
    v = corr
    key_nan = torch.isnan(v)
    a = v.unsqueeze(-1)            # (B,N,N,1): candidate c
    b = v.unsqueeze(-2)            # (B,N,1,N): rival j
    a_nan = key_nan.unsqueeze(-1)
    b_nan = key_nan.unsqueeze(-2)
    gt = (b > a) | (b_nan & ~a_nan)
    eq = (b == a) | (b_nan & a_nan)
    jj = torch.arange(N).view(1, 1, 1, N)
    cc = torch.arange(N).view(1, 1, N, 1)
    beats = gt | (eq & (jj < cc))
    rank = beats.sum(-1)
    return (rank < k).to(corr.dtype)


def listall_groups(N: int, group_size: int) -> Tensor:
    """The constant candidate table of the exhaustive builder (MS_HGNN_batch.py:313-326, `all_combs`):
    for every agent i, every group of `group_size` agents that contains i — i first, then a
    (group_size-1)-subset of the others in lexicographic order (the order of torch.combinations).
    Shape (N, C(N-1, group_size-1), group_size), int64."""
    import itertools
    rows = []
    for i in range(N):
        others = [a for a in range(N) if a != i]
        rows.append([[i, *c] for c in itertools.combinations(others, group_size - 1)])
    return torch.tensor(rows, dtype=torch.long)


def listall_incidence(corr: Tensor, scale: int, like: Optional[Tensor] = None) -> Tensor:
    """``MS_HGNN_hyper.init_adj_attention_listall`` (MS_HGNN_batch.py:390-414): hyperedge i = the group
    of `scale` agents containing i whose affinity sub-matrix has the largest total (all ordered pairs,
    diagonal included), searched exhaustively.  scale == N -> one all-ones edge.

    Stated with the reference's own reduction ops on the same tensor shape ((B,N,C,s,s) contiguous,
    torch.sum over the last two dims, torch.max over C), so score rounding and the tie rule (first
    maximum) are the reference's; goldens are tie-free."""
    B, N = corr.shape[0], corr.shape[1]
    dtype = corr.dtype if like is None else like.dtype
    if scale == N:
        return torch.ones(B, 1, N, dtype=dtype)
    s = max(int(scale), 1)
    if s > N:
        raise RuntimeError("group size larger than the number of agents")
    groups = listall_groups(N, s)                                   # (N, C, s)
    sub = corr[:, groups[:, :, :, None], groups[:, :, None, :]].contiguous()    # (B, N, C, s, s)
    score = torch.sum(sub, dim=(3, 4))
    _, best = torch.max(score, dim=2)                               # (B, N)
    chosen = groups[torch.arange(N)[None, :], best]                 # (B, N, s)
    H = torch.zeros(B, N, N, dtype=dtype)
    return H.scatter(2, chosen, 1)


def pairwise_incidence(N: int, B: int, dtype=torch.float32) -> Tensor:
    """``init_adj`` + ``H = rel_rec + rel_send`` (MS_HGNN_batch.py:143-160,118,124):
    edge e = i*N + j touches node j (rel_rec) and node i (rel_send); a
    self-loop (i == j) therefore has weight 2."""
    H = torch.zeros(N * N, N, dtype=dtype)
    e = torch.arange(N * N)
    H[e, e % N] += 1   # rel_rec: np.where(off_diag)[1]
    H[e, e // N] += 1  # rel_send: np.where(off_diag)[0]
    return H[None].repeat(B, 1, 1)


# --------------------------------------------------------------------------
# the three stages of one message-passing round
# --------------------------------------------------------------------------
def node2edge(state: State, x: Tensor, H: Tensor, idx: int = 0, decomposed: bool = False
              ) -> Tuple[Tensor, Tensor]:
    """``node2edge`` (hyper: MS_HGNN_batch.py:357-370; pairwise :122-141, the same
    once H = rel_rec + rel_send).  Returns (edges, x') — x' is exposed for
    per-kernel checks.  ``decomposed=True`` evaluates the attention MLP without
    materialising the (B,E,N,128) tensor."""
    xp = mlp(state, f"node2edge_start_mlp.{idx}", x)                 # :358
    edge_init = torch.matmul(H, xp)                                   # :359
    N, E = xp.shape[1], edge_init.shape[1]
    if not decomposed:
        x_rep = (xp[:, :, None, :].transpose(2, 1)).repeat(1, E, 1, 1)        # :362
        edge_rep = edge_init[:, :, None, :].repeat(1, 1, N, 1)                # :363
        cat = torch.cat((x_rep, edge_rep), dim=-1)                            # :364
        att = mlp(state, f"attention_mlp.{idx}", cat)[:, :, :, 0]             # :365
    else:
        W1 = state[f"attention_mlp.{idx}.layers.0.weight"]
        b1 = state[f"attention_mlp.{idx}.layers.0.bias"]
        W2 = state[f"attention_mlp.{idx}.layers.1.weight"]
        b2 = state[f"attention_mlp.{idx}.layers.1.bias"]
        D = xp.shape[-1]
        P = F.linear(xp, W1[:, :D], b1)            # (B,N,32)
        Q = F.linear(edge_init, W1[:, D:])         # (B,E,32)
        hid = torch.relu(P[:, None, :, :] + Q[:, :, None, :])
        att = torch.matmul(hid, W2[0]) + b2[0]
    Hw = att * H                                                       # :366
    Hw = torch.softmax(Hw, dim=2)                                      # :367
    Hw = Hw * H                                                        # :368
    edges = torch.matmul(Hw, xp)                                       # :369
    return edges, xp


def edge_mlp_gumbel(state: State, prefix: str, edges: Tensor, U: Tensor) -> Tuple[Tensor, Tensor]:
    """``MLP_dict_softmax.forward`` (MS_HGNN_batch.py:41-53): returns
    (factor * distribution, distribution)."""
    z = mlp(state, f"{prefix}.init_MLP", edges)                                       # :43
    dist = gumbel_softmax_from_uniform(mlp(state, f"{prefix}.MLP_distribution", z), U)  # :45
    fac = torch.sigmoid(mlp(state, f"{prefix}.MLP_factor", z))                        # :47
    return fac * dist, dist                                                           # :50,53


def aggregate_gather(H: Tensor, ori: Tensor) -> Tensor:
    """``edges = torch.matmul(H, ori)`` (MS_HGNN_batch.py:263)."""
    return torch.matmul(H, ori)


def aggregate_typed_mlp(state: State, prefix: str, edge_feat: Tensor, eo: Tensor) -> Tensor:
    """The typed sum of ``edge_aggregation.forward`` (MS_HGNN_batch.py:262,264-265)."""
    K = edge_feat.shape[-1]
    out = torch.zeros(eo.shape[0], eo.shape[1], eo.shape[-1], dtype=eo.dtype)
    for k in range(K):
        out += edge_feat[:, :, k:k + 1] * mlp(state, f"{prefix}.agg_mlp.{k}", eo)
    return out


def aggregate_scatter(H: Tensor, feat: Tensor, ori: Tensor) -> Tensor:
    """``cat(Hᵀ@feat, ori)`` (MS_HGNN_batch.py:267) followed by the division by
    ``incoming.size(1)`` = N of ``edge2node`` (:120,355)."""
    node = torch.cat((torch.matmul(H.permute(0, 2, 1), feat), ori), dim=-1)
    return node / node.size(1)


def edge2node(state: State, edge_feat: Tensor, ori: Tensor, H: Tensor, idx: int = 0) -> Tensor:
    """``edge2node`` (MS_HGNN_batch.py:116-120,352-355) = gather, typed MLP, scatter."""
    p = f"edge_aggregation_list.{idx}"
    eo = aggregate_gather(H, ori)
    feat = aggregate_typed_mlp(state, p, edge_feat, eo)
    return aggregate_scatter(H, feat, ori)


# --------------------------------------------------------------------------
# whole modules
# --------------------------------------------------------------------------
def _message_passing(state: State, h: Tensor, H: Tensor, U_list: List[Tensor], nmp_layers: int,
                     decomposed: bool, trace: Optional[dict]) -> Tuple[Tensor, Tensor]:
    """Shared skeleton of both forwards (MS_HGNN_batch.py:174-195 and :425-441)."""
    u = iter(U_list)
    edges, xp = node2edge(state, h, H, 0, decomposed)
    edge_feat, factors = edge_mlp_gumbel(state, "nmp_mlp_start", edges, next(u))
    if trace is not None:
        trace.update(xp=xp, edges=edges, edge_feat=edge_feat)
    node_feat = h
    idx = 0
    if nmp_layers > 1:
        for l in range(2 * (nmp_layers - 1)):
            if l % 2 == 0:
                node_feat = mlp(state, f"nmp_mlps.{l}", edge2node(state, edge_feat, node_feat, H, idx))
                idx += 1
            else:
                e2, _ = node2edge(state, node_feat, H, idx, decomposed)
                edge_feat, _ = edge_mlp_gumbel(state, f"nmp_mlps.{l}", e2, next(u))
    if trace is not None:
        p = f"edge_aggregation_list.{idx}"
        eo = aggregate_gather(H, node_feat)
        feat = aggregate_typed_mlp(state, p, edge_feat, eo)
        trace.update(eo=eo, feat=feat, agg=aggregate_scatter(H, feat, node_feat))
    node_feat = mlp(state, "nmp_mlp_end", edge2node(state, edge_feat, node_feat, H, idx))
    return node_feat, factors


def noise_shapes(B: int, N: int, scale: Optional[int], nmp_layers: int = 1) -> List[Tuple[int, int, int]]:
    """Shapes of the uniform draws one forward makes, in order: one (B,E,K) per
    ``MLP_dict_softmax`` call.  ``scale=None`` means the pairwise module."""
    if scale is None:
        E, K = N * N, EDGE_TYPES_PAIRWISE
    else:
        E, K = (1 if scale == N else N), EDGE_TYPES_HYPER
    return [(B, E, K)] * nmp_layers


def ms_hgnn_pairwise_forward(state: State, h: Tensor, U_list: List[Tensor], nmp_layers: int = 1,
                             decomposed: bool = False, trace: Optional[dict] = None
                             ) -> Tuple[Tensor, Tensor]:
    """``MS_HGNN_oridinary.forward`` (MS_HGNN_batch.py:162-198) → (node_feat, factors)."""
    B, N = h.shape[0], h.shape[1]
    H = pairwise_incidence(N, B, h.dtype)
    return _message_passing(state, h, H, U_list, nmp_layers, decomposed, trace)


def ms_hgnn_hyper_forward(state: State, h: Tensor, corr: Tensor, scale: int, U_list: List[Tensor],
                          nmp_layers: int = 1, decomposed: bool = False,
                          trace: Optional[dict] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """``MS_HGNN_hyper.forward`` with listall=False (MS_HGNN_batch.py:417-443)
    → (node_feat, factor, H)."""
    H = topk_incidence(corr, scale, like=h)
    node_feat, factor = _message_passing(state, h, H, U_list, nmp_layers, decomposed, trace)
    return node_feat, factor, H


def ms_hgnn_multiscale_forward(state_pair: State, states_hyper: Sequence[State], scales: Sequence[int],
                               h: Tensor, U_pair: List[Tensor], U_hyper: Sequence[List[Tensor]],
                               decomposed: bool = False):
    """What ``PastEncoder.forward`` does around the path (model/GroupNet_nba.py:284-311):
    affinity, the pairwise module, one hyper module per scale, and the two concats."""
    corr = affinity(h)
    inter, _ = ms_hgnn_pairwise_forward(state_pair, h, U_pair, decomposed=decomposed)
    feats, Hs = [h, inter], []
    for st, s, U in zip(states_hyper, scales, U_hyper):
        nf, _, H = ms_hgnn_hyper_forward(st, h, corr, s, U, decomposed=decomposed)
        feats.append(nf)
        Hs.append(H)
    return torch.cat(feats, dim=-1), (torch.cat(Hs, dim=1) if Hs else None), corr


# --------------------------------------------------------------------------
# large N: the pairwise module in slabs of edges (BASELINE config 5, N = 256: E = 65 536 edges per scene)
# --------------------------------------------------------------------------
def pairwise_incidence_rows(N: int, e0: int, e1: int, dtype=torch.float32) -> Tensor:
    """Rows [e0, e1) of ``pairwise_incidence`` (MS_HGNN_batch.py:143-160,118,124) without the other N*N - (e1-e0)."""
    e = torch.arange(e0, e1)
    H = torch.zeros(e1 - e0, N, dtype=dtype)
    r = torch.arange(e1 - e0)
    H[r, e % N] += 1
    H[r, e // N] += 1
    return H


def ms_hgnn_pairwise_forward_chunked(state: State, h: Tensor, U_list: List[Tensor], slab: int = 4096
                                     ) -> Tuple[Tensor, Tensor]:
    """``MS_HGNN_oridinary.forward`` with nmp_layers = 1 (MS_HGNN_batch.py:162-198) evaluated in slabs of `slab`
    edges, for N where neither the (B,E,N,128) attention input of the reference (:129-132) nor the (B,E,N,32)
    hidden tensor of the decomposed form fits in memory.  Edges are independent in ``node2edge`` (:122-141) and in
    the typed MLP of ``edge2node`` (:116-120, 259-268); the only coupling is the sum over edges
    ``H^T @ feat`` (:267), which is accumulated slab by slab (in float64, rounded once — the reference's single fp32
    matmul sums the same terms).  Same operations per edge as `node2edge(decomposed=True)` / `edge_mlp_gumbel` /
    `aggregate_typed_mlp`; `tests/test_oracle_golden.py` checks it against the unchunked oracle (itself pinned by
    the reference's goldens) at every golden N, with slabs that do not divide E."""
    B, N, D = h.shape
    E = N * N
    U = U_list[0]
    xp = mlp(state, "node2edge_start_mlp.0", h)                              # :125
    W1 = state["attention_mlp.0.layers.0.weight"]
    b1 = state["attention_mlp.0.layers.0.bias"]
    W2 = state["attention_mlp.0.layers.1.weight"]
    b2 = state["attention_mlp.0.layers.1.bias"]
    P = F.linear(xp, W1[:, :D], b1)                                          # node half of attention layer 0
    node = torch.zeros(B, N, D, dtype=torch.float64)
    factors = torch.empty(B, E, EDGE_TYPES_PAIRWISE, dtype=h.dtype)
    for e0 in range(0, E, slab):
        e1 = min(E, e0 + slab)
        Hs = pairwise_incidence_rows(N, e0, e1, h.dtype)[None]               # (1, slab, N)
        edge_init = torch.matmul(Hs, xp)                                      # :127
        Q = F.linear(edge_init, W1[:, D:])
        hid = torch.relu(P[:, None, :, :] + Q[:, :, None, :])                # (B, slab, N, 32)
        att = torch.matmul(hid, W2[0]) + b2[0]                                # :131-134
        del hid
        Hw = torch.softmax(att * Hs, dim=2) * Hs                              # :135-137
        edges = torch.matmul(Hw, xp)                                          # :138
        edge_feat, dist = edge_mlp_gumbel(state, "nmp_mlp_start", edges, U[:, e0:e1])    # :177
        factors[:, e0:e1] = dist
        eo = torch.matmul(Hs, h)                                              # edge_aggregation :263
        feat = aggregate_typed_mlp(state, "edge_aggregation_list.0", edge_feat, eo)
        node += torch.matmul(Hs.permute(0, 2, 1).double(), feat.double())     # :267, summed over slabs
    agg = torch.cat((node.to(h.dtype), h), dim=-1)
    agg = agg / agg.size(1)                                                   # :120
    return mlp(state, "nmp_mlp_end", agg), factors


# --------------------------------------------------------------------------
# Philox4x32-10 (for the device-noise mode of the HIP path)
# --------------------------------------------------------------------------
def philox_uniform(n: int, seed: int, offset: int = 0):
    """Counter-based uniforms as ``gn_philox_uniform_f32`` produces them (the
    build's own design — the reference has no device RNG): element i comes from
    Philox4x32-10 block (i + offset) // 4, lane (i + offset) % 4, key =
    (seed lo32, seed hi32); u = (x >> 8) * 2**-24 in [0, 1)."""
    import numpy as np
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    W0, W1 = 0x9E3779B9, 0xBB67AE85
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    blk = idx >> np.uint64(2)
    c = [(blk & np.uint64(0xFFFFFFFF)).astype(np.uint64), (blk >> np.uint64(32)).astype(np.uint64),
         np.zeros_like(blk), np.zeros_like(blk)]
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    out = np.stack(c, axis=1)                      # (n, 4)
    x = out[np.arange(n), (idx & np.uint64(3)).astype(np.int64)]
    return ((x >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)
