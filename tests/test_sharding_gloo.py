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


# ---------------------------------------------------------------------------------------------
# default noise under sharding: every rank must use ITS rows of the full-batch streams
# ---------------------------------------------------------------------------------------------
class _OracleBlock:
    """Stand-in with the attributes `sharded_forward` / `default_shard_noise` use of a MultiScaleHGNN block."""

    class _I:
        nmp_layers = 1
    interaction = _I()

    def __init__(self, sp, sh):
        self.sp, self.sh = sp, sh

    def noise_shapes(self, B, N):
        return [O.noise_shapes(B, N, None)[0]] + [O.noise_shapes(B, N, s)[0] for s in SCALES]

    def __call__(self, f_local, noise_u=None):
        feats, H, _ = O.ms_hgnn_multiscale_forward(self.sp, [self.sh] * len(SCALES), SCALES, f_local, noise_u[0],
                                                   noise_u[1:], decomposed=True)
        return feats, H


def _default_noise_worker(rank, world, port, B, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        blk = _OracleBlock(load_state("pairwise"), load_state("hyper"))
        f, _ = _inputs(B)
        torch.manual_seed(4242)                 # the caller's seed, the same on every rank (as on one device)
        with torch.no_grad():
            feats, _ = sharding.sharded_forward(blk, f)          # noise_full=None: default noise
        np.save(os.path.join(out_dir, f"dn_{rank}.npy"), feats.numpy())
    finally:
        dist.destroy_process_group()


def test_sharded_default_host_noise_equals_single_device(tmp_path):
    world, B = 2, 6
    mp.spawn(_default_noise_worker, args=(world, _free_port(), B, str(tmp_path)), nprocs=world, join=True)
    blk = _OracleBlock(load_state("pairwise"), load_state("hyper"))
    f, _ = _inputs(B)
    torch.manual_seed(4242)
    noise = [[torch.rand(s)] for s in blk.noise_shapes(B, 11)]      # what one device draws, module-major
    with torch.no_grad():
        ref, _ = blk(f, noise)
    for r in range(world):
        got = np.load(tmp_path / f"dn_{r}.npy")
        assert np.max(np.abs(got - ref.numpy())) <= 1e-6, r
    # and the two halves did NOT see the same noise: a run that re-seeds per shard differs from the reference
    torch.manual_seed(4242)
    same = [[torch.rand((B // 2,) + tuple(s[1:]))] for s in blk.noise_shapes(B, 11)]
    with torch.no_grad():
        wrong, _ = blk(f[B // 2:], same)
    assert np.max(np.abs(wrong.numpy() - ref.numpy()[B // 2:])) > 1e-4


def test_default_shard_noise_device_mode_offsets():
    """Device mode: per module a PhiloxNoise at offset(module span) + first_row * E * K, spans back to back."""
    import groupnet_amd.MS_HGNN_batch as M
    blk = _OracleBlock(None, None)
    B, N = 8, 11
    prev = (M._NoiseState.mode, M._NoiseState.seed, M._NoiseState.offset, M._NoiseState.counter)
    try:
        M.set_noise_mode("device", seed=31, offset=1000)
        nz = sharding.default_shard_noise(blk, B, N, 4, 8, torch.device("cpu"))
        offs = [per[0].offset for per in nz]
        shapes = blk.noise_shapes(B, N)
        assert offs == sharding.philox_offsets(shapes, 4, 1000)
        assert all(per[0].seed == 31 for per in nz)
        assert M._NoiseState.offset == 1000 + sum(b * e * k for b, e, k in shapes)     # the stream advanced one batch
    finally:
        M.set_noise_mode(prev[0], prev[1], prev[2], prev[3])


# ---------------------------------------------------------------------------------------------
# bucketed double-buffered all-gather (bench.py --gpus N uses it; here with gloo on CPU)
# ---------------------------------------------------------------------------------------------
def _bucket_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        slots, shape = 3, (4, 2, 5)
        bg = sharding.BucketedGather(slots, shape, torch.device("cpu"))
        log = []
        val = lambda r, k: torch.full(shape, float(1000 * r + k))
        for k in range(8):                       # banks: [0,1,2] [3,4,5] [6,7,-] -> two full gathers + a flush
            bank = bg.put(val(rank, k))
            if k % slots == slots - 1:
                log.append((k, bank, bg.count[bank], bg.gathered(bank).clone()))
        bg.flush()
        log.append((7, 0, bg.count[0], bg.gathered(0).clone()))
        for k in range(8, 11):                   # after a flush the next put starts a fresh bank (bank 1)
            bank = bg.put(val(rank, k))
        log.append((10, bank, bg.count[bank], bg.gathered(bank).clone()))
        bg.wait()
        with pytest.raises(ValueError):
            bg.put(torch.zeros(1))
        torch.save(dict(log=log, gathers=bg.gathers), os.path.join(out_dir, f"bg{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_bucketed_gather_bank_reuse_partial_flush_ragged_steps(tmp_path):
    world = 2
    mp.spawn(_bucket_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        d = torch.load(os.path.join(tmp_path, f"bg{r}.pt"))
        assert d["gathers"] == 4
        (k0, b0, c0, g0), (k1, b1, c1, g1), (k2, b2, c2, g2), (k3, b3, c3, g3) = d["log"]
        assert (b0, c0, b1, c1, b2, c2, b3, c3) == (0, 3, 1, 3, 0, 2, 1, 3)
        for src in range(world):                 # [rank, slot] = that rank's step
            assert g0[src, :, 0, 0, 0].tolist() == [1000 * src + k for k in (0, 1, 2)]
            assert g1[src, :, 0, 0, 0].tolist() == [1000 * src + k for k in (3, 4, 5)]
            assert g2[src, :2, 0, 0, 0].tolist() == [1000 * src + k for k in (6, 7)]      # bank 0 reused, 2 valid slots
            assert g3[src, :, 0, 0, 0].tolist() == [1000 * src + k for k in (8, 9, 10)]   # bank 1 reused after the flush
