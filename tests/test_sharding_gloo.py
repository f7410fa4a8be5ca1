"""world_size=2 (and 3, ragged) runs of the batch-sharding path on CPU with the gloo backend.
The per-rank compute is stood in for by the CPU oracle (tests may use it); what is under test is
the partition, the noise slicing and the all-gather — the data-path collective of SURVEY.md §8e."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_state
from groupnet_amd import sharding
from oracle import ms_hgnn_oracle as O

SCALES = [2, 5, 11]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_block(sp, sh):
    def block(f_local, noise_u=None):
        feats, H, _ = O.ms_hgnn_multiscale_forward(sp, [sh] * len(SCALES), SCALES, f_local, noise_u[0],
                                                   noise_u[1:], decomposed=True)
        return feats, H
    return block


def _inputs(B, N=11):
    g = torch.Generator().manual_seed(99)
    f = torch.randn(B, N, 64, generator=g)
    shapes = [O.noise_shapes(B, N, None)[0]] + [O.noise_shapes(B, N, s)[0] for s in SCALES]
    noise = [[torch.rand(s, generator=g)] for s in shapes]
    return f, noise


def _worker(rank, world, port, B, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sp, sh = load_state("pairwise"), load_state("hyper")
        f, noise = _inputs(B)
        with torch.no_grad():
            feats, H = sharding.sharded_forward(_oracle_block(sp, sh), f, noise, gather_H=True)
        np.save(os.path.join(out_dir, f"feats_{rank}.npy"), feats.numpy())
        np.save(os.path.join(out_dir, f"H_{rank}.npy"), H.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 8), (3, 7)])
def test_sharded_equals_single(tmp_path, world, B):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
    sp, sh = load_state("pairwise"), load_state("hyper")
    f, noise = _inputs(B)
    with torch.no_grad():
        ref, Href = _oracle_block(sp, sh)(f, noise)
    for r in range(world):
        feats = np.load(tmp_path / f"feats_{r}.npy")
        H = np.load(tmp_path / f"H_{r}.npy")
        assert feats.shape == (B, 11, 64 * 5)
        assert np.array_equal(H, Href.numpy())
        assert np.max(np.abs(feats - ref.numpy())) <= 1e-6   # per-scene math; only BLAS blocking differs


def test_shard_range_and_offsets():
    for B in (1, 7, 8, 4096):
        for R in (1, 2, 3, 8):
            spans = [sharding.shard_range(B, r, R) for r in range(R)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(e - s for s, e in spans) - min(e - s for s, e in spans) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(4, 4, 4)
    shapes = [(8, 121, 6), (8, 11, 10), (8, 1, 10)]
    assert sharding.philox_offsets(shapes, 0) == [0, 8 * 726, 8 * 726 + 8 * 110]
    assert sharding.philox_offsets(shapes, 4, 100) == [100 + 4 * 726, 100 + 8 * 726 + 4 * 110,
                                                       100 + 8 * 726 + 8 * 110 + 4 * 10]
    full = [torch.arange(8.0).view(8, 1, 1), [torch.arange(8.0).view(8, 1, 1)]]
    sl = sharding.slice_noise(full, 2, 5)
    assert sl[0].flatten().tolist() == [2, 3, 4] and sl[1][0].flatten().tolist() == [2, 3, 4]


def _grad_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(5)
        params = [torch.nn.Parameter(torch.randn(s)) for s in [(3, 4), (7,), (2, 2, 2), (5,)]]
        params[3].requires_grad_(False)
        g = torch.Generator().manual_seed(100 + rank)
        for i, p in enumerate(params[:3]):
            if not (rank == 1 and i == 1):          # rank 1 has no gradient for parameter 1
                p.grad = torch.randn(p.shape, generator=g)
        sharding.allreduce_gradients(params, average=True, bucket_bytes=64)      # tiny buckets: several collectives
        torch.save([None if p.grad is None else p.grad.clone() for p in params], os.path.join(out_dir, f"g{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_allreduce_gradients_buckets_and_missing_grads(tmp_path):
    world = 2
    mp.spawn(_grad_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(os.path.join(tmp_path, f"g{r}.pt")) for r in range(world)]
    # expected: mean over ranks of the per-rank gradients (zeros where a rank had none)
    want = []
    for i, shape in enumerate([(3, 4), (7,), (2, 2, 2)]):
        acc = torch.zeros(shape)
        for r in range(world):
            g = torch.Generator().manual_seed(100 + r)
            gs = [torch.randn(s, generator=g) if not (r == 1 and j == 1) else None for j, s in enumerate([(3, 4), (7,), (2, 2, 2)])]
            if gs[i] is not None:
                acc += gs[i]
        want.append(acc / world)
    for r in range(world):
        for i in range(3):
            assert torch.allclose(got[r][i], want[i], atol=1e-6), (r, i)
        assert got[r][3] is None        # frozen parameter untouched
