"""The CPU oracle against every golden vector generated from the reference itself
(tests/golden/make_golden.py).  Runs without a GPU."""
import numpy as np
import pytest
import torch

from conftest import case_names, load_case, uniforms, weights_for
from oracle import ms_hgnn_oracle as O

TOL = 1e-6  # restatement == reference within 1e-6 on all goldens


def _close(a, b, tol=TOL):
    a = a.numpy() if isinstance(a, torch.Tensor) else a
    err = float(np.max(np.abs(a - b))) if a.size else 0.0
    assert a.shape == b.shape and err <= tol, (a.shape, b.shape, err)


@pytest.mark.parametrize("name", case_names())
@pytest.mark.parametrize("decomposed", [False, True])
def test_oracle_matches_reference(name, decomposed):
    c = load_case(name)
    sp, sh, nmp = weights_for(name)
    h, corr = torch.from_numpy(c["h"]), torch.from_numpy(c["corr"])
    N = h.shape[1]
    _close(O.affinity(h), c["corr"], 1e-6)
    if "pair_node_feat" in c:
        tr = {}
        nf, fac = O.ms_hgnn_pairwise_forward(sp, h, uniforms(c, "pair"), nmp, decomposed, tr)
        _close(nf, c["pair_node_feat"])
        _close(fac, c["pair_factors"])
        for k in ("xp", "edges", "edge_feat"):
            _close(tr[k], c[f"pair_{k}"])
        if nmp == 1:
            _close(tr["eo"], c["pair_eo"])
            _close(tr["agg"], c["pair_agg"])
    for s in c["scales"].tolist():
        tr = {}
        nf, fac, H = O.ms_hgnn_hyper_forward(sh, h, corr, s, uniforms(c, f"hyper{s}"), nmp, decomposed, tr)
        assert np.array_equal(H.numpy(), c[f"hyper{s}_H"]), (name, s)
        assert np.array_equal(O.topk_incidence_ranked(corr, s).numpy(), c[f"hyper{s}_H"]), (name, s)
        _close(nf, c[f"hyper{s}_node_feat"])
        _close(fac, c[f"hyper{s}_factor"])
        for k in ("xp", "edges", "edge_feat"):
            _close(tr[k], c[f"hyper{s}_{k}"])
        if nmp == 1:
            _close(tr["eo"], c[f"hyper{s}_eo"])
            _close(tr["agg"], c[f"hyper{s}_agg"])
        E = 1 if s == N else N
        assert H.shape == (h.shape[0], E, N) and fac.shape == (h.shape[0], E, 10)


def test_uniform_draw_reproduces_reference_stream():
    """The recorded uniforms are exactly torch.manual_seed(seed); torch.rand(shape) in call order
    (pairwise first, then each scale) — the RNG contract of SURVEY.md §7."""
    c = load_case("nba_b10")
    torch.manual_seed(int(c["seed"]))
    B, N = c["h"].shape[:2]
    (shape,) = O.noise_shapes(B, N, None)
    assert np.array_equal(O.draw_uniform(shape).numpy(), c["pair_U0"])
    for s in c["scales"].tolist():
        (shape,) = O.noise_shapes(B, N, s)
        assert np.array_equal(O.draw_uniform(shape).numpy(), c[f"hyper{s}_U0"])


def test_pairwise_incidence_self_loops():
    H = O.pairwise_incidence(4, 2)
    assert H.shape == (2, 16, 4)
    assert H[0, 5, 1] == 2 and H[0, 6].tolist() == [0, 1, 1, 0]
    assert torch.all(H.sum(-1) == 2)


def test_topk_rank_rule_ties_and_nan():
    corr = torch.tensor([[[1.0, 1.0, 0.5, 1.0], [0.0, float("nan"), 2.0, 2.0],
                          [3.0, 2.0, 1.0, 0.0], [0.0, 0.0, 0.0, 0.0]]])
    H = O.topk_incidence_ranked(corr, 2)[0]
    assert H[0].tolist() == [1, 1, 0, 0]      # lowest index wins ties
    assert H[1].tolist() == [0, 1, 1, 0]      # NaN ranks first, then the first 2.0
    assert H[2].tolist() == [1, 1, 0, 0]
    assert H[3].tolist() == [1, 1, 0, 0]
    with pytest.raises(RuntimeError):
        O.topk_incidence_ranked(corr, 5)
    with pytest.raises(RuntimeError):
        O.topk_incidence(corr, 5)
    assert O.topk_incidence_ranked(corr, 0).sum() == 4  # scale < 1 clamps to 1
    assert O.topk_incidence_ranked(corr, 4).shape == (1, 1, 4)


def test_philox_known_answer():
    """Philox4x32-10 known-answer vectors (Random123 kat_vectors): counter 0 / key 0 and the
    all-ones case."""
    import numpy as np
    # counter (0,0,0,0), key (0,0) -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8
    u = O.philox_uniform(4, seed=0, offset=0)
    exp = np.array([0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8], dtype=np.uint64)
    assert np.array_equal(u, ((exp >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -24)))
    assert u.min() >= 0.0 and u.max() < 1.0


def _c_oracle():
    import ctypes
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "oracle", "_build", "liboracle_topk.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(root, "oracle")])
    lib = ctypes.CDLL(so)
    lib.gn_oracle_topk_incidence.restype = ctypes.c_int
    lib.gn_oracle_topk_incidence.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 3
    return lib


def _c_topk(lib, corr, s):
    B, N = corr.shape[:2]
    corr = np.ascontiguousarray(corr, dtype=np.float32)
    H = np.empty((B, 1 if s == N else N, N), dtype=np.float32)
    rc = lib.gn_oracle_topk_incidence(corr.ctypes.data, H.ctypes.data, B, N, s)
    return rc, H


@pytest.mark.parametrize("name", case_names())
def test_c_oracle_topk_matches_reference_incidence(name):
    """The plain-C restatement of the index work (k rounds of arg-max) reproduces every golden H bit
    for bit, and agrees with the rank-count statement the HIP kernel implements."""
    lib = _c_oracle()
    c = load_case(name)
    for s in c["scales"].tolist():
        rc, H = _c_topk(lib, c["corr"], s)
        assert rc == 0 and np.array_equal(H, c[f"hyper{s}_H"]), (name, s)


def test_c_oracle_topk_ties_nan_range():
    lib = _c_oracle()
    corr = np.array([[[1.0, 1.0, 0.5, 1.0], [0.0, np.nan, 2.0, 2.0], [3.0, 2.0, 1.0, 0.0], [0.0] * 4]], dtype=np.float32)
    for s in (0, 1, 2, 3, 4):
        rc, H = _c_topk(lib, corr, s)
        assert rc == 0 and np.array_equal(H, O.topk_incidence_ranked(torch.from_numpy(corr), s).numpy()), s
    assert _c_topk(lib, corr, 5)[0] == -1
    rng = np.random.default_rng(0)
    big = rng.integers(0, 4, size=(3, 40, 40)).astype(np.float32)      # heavy ties
    for s in (1, 7, 39):
        rc, H = _c_topk(lib, big, s)
        assert np.array_equal(H, O.topk_incidence_ranked(torch.from_numpy(big), s).numpy())


@pytest.mark.parametrize("name", ["n11_b6", "n7_b3", "n13_b2"])
def test_listall_oracle_matches_reference_goldens(name):
    """oracle.listall_incidence == the reference's init_adj_attention_listall (MS_HGNN_batch.py:390-414) on
    fixtures produced by the reference method itself (tests/golden/make_golden_listall.py); every row's
    winner leads the runner-up by >= 2e-4, far above fp32 summation noise."""
    import numpy as np
    import os
    c = np.load(os.path.join(os.path.dirname(__file__), "golden", f"listall_{name}.npz"))
    corr = torch.from_numpy(c["corr"])
    for s in c["scales"]:
        H = O.listall_incidence(corr, int(s))
        assert torch.equal(H, torch.from_numpy(c[f"H_s{int(s)}"]))
        if int(s) < corr.shape[1]:
            assert float(H.sum(-1).min()) == float(H.sum(-1).max()) == float(max(int(s), 1))
            assert bool((torch.diagonal(H, dim1=1, dim2=2) == 1).all())      # agent i is in its own group
            assert float(c[f"margin_s{int(s)}"].min()) > 1e-4


def _grad_case(name):
    import numpy as np
    import os
    with np.load(os.path.join(os.path.dirname(__file__), "golden", f"grad_{name}.npz")) as z:
        return {k: z[k].copy() for k in z.files}


def _grad_stats(g):
    g64 = g.double()
    return float(g64.sum()), float(g64.abs().sum()), float(g64.abs().max())


@pytest.mark.parametrize("name", ["n11_b5", "n7_b3_nmp2"])
def test_oracle_autograd_matches_reference_gradients(name):
    """Pins the backward oracle: torch autograd through oracle.ms_hgnn_oracle reproduces the gradients the
    REFERENCE's autograd produced (tests/golden/make_golden_backward.py) — dL/dh and selected parameter
    gradients element-wise, every parameter gradient through its (sum, sum|.|, max|.|)."""
    from conftest import load_state
    c = _grad_case(name)
    nmp = int(c["nmp"])
    sfx = "_nmp2" if nmp == 2 else ""
    h0, corr = torch.from_numpy(c["h"]), torch.from_numpy(c["corr"])
    for prefix in ["pair"] + [f"hyper{int(s)}" for s in c["scales"]]:
        state = {k: v.clone().requires_grad_(True) for k, v in load_state(("pairwise" if prefix == "pair" else "hyper") + sfx).items()}
        U, i = [], 0
        while f"{prefix}_U{i}" in c:
            U.append(torch.from_numpy(c[f"{prefix}_U{i}"]))
            i += 1
        h = h0.clone().requires_grad_(True)
        if prefix == "pair":
            nf, fac = O.ms_hgnn_pairwise_forward(state, h, U, nmp_layers=nmp)
        else:
            nf, fac, H = O.ms_hgnn_hyper_forward(state, h, corr, int(prefix[5:]), U, nmp_layers=nmp)
            assert torch.equal(H, torch.from_numpy(c[f"{prefix}_H"]))
        assert float((nf.detach() - torch.from_numpy(c[f"{prefix}_node_feat"])).abs().max()) <= 1e-6
        ((nf * torch.from_numpy(c[f"{prefix}_R1"])).sum() + (fac * torch.from_numpy(c[f"{prefix}_R2"])).sum()).backward()
        gh = torch.from_numpy(c[f"{prefix}_g_h"])
        assert float((h.grad - gh).abs().max()) <= 1e-5 * float(gh.abs().max())
        for n, ref in zip(c[f"{prefix}_stat_names"], c[f"{prefix}_stats"]):
            g = state[str(n)].grad
            if np.isnan(ref[0]):
                assert g is None, n
                continue
            got = _grad_stats(g)
            assert abs(got[1] - ref[1]) <= 1e-4 * ref[1] + 1e-7 and abs(got[2] - ref[2]) <= 1e-4 * ref[2] + 1e-7, (n, got, ref)
            assert abs(got[0] - ref[0]) <= 1e-4 * ref[1] + 1e-7, (n, got, ref)
            if f"{prefix}_g/{n}" in c:
                full = torch.from_numpy(c[f"{prefix}_g/{n}"])
                assert float((g - full).abs().max()) <= 1e-5 * float(full.abs().max()) + 1e-8, n


@pytest.mark.parametrize("name,slab", [("syn_n11_b37", 50), ("syn_n5_b4", 7), ("syn_n50_b3", 999), ("nba_b10", 121)])
def test_chunked_pairwise_oracle_equals_the_pinned_oracle(name, slab):
    """The slab-wise pairwise oracle used at N = 256 (config 5) against the reference's golden outputs and the
    unchunked oracle, with slabs that do not divide E."""
    c = load_case(name)
    sp, _, nmp = weights_for(name)
    assert nmp == 1
    h = torch.from_numpy(c["h"])
    U = uniforms(c, "pair")
    with torch.no_grad():
        nf, fac = O.ms_hgnn_pairwise_forward_chunked(sp, h, U, slab=slab)
        nf0, fac0 = O.ms_hgnn_pairwise_forward(sp, h, U, decomposed=True)
    assert float((nf - nf0).abs().max()) <= 1e-6 and float((fac - fac0).abs().max()) <= 1e-6
    assert float((nf - torch.from_numpy(c["pair_node_feat"])).abs().max()) <= 1e-6
    assert float((fac - torch.from_numpy(c["pair_factors"])).abs().max()) <= 1e-6
