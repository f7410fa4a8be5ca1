"""bench.py's host logic without a GPU: the self-launch of `python bench.py --gpus N` (argv, exit-code propagation)
and the distributed control flow of its timed regions (groupnet_amd/timing.py) under gloo at world size 2 with a
stub step — fences, max-over-ranks clock, the region count agreed through rank 0."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from groupnet_amd.timing import RegionTimer, launch_argv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_launch_argv_shape():
    argv = launch_argv(8, "/x/bench.py", ["--gpus", "8", "--steps", "5"], port=1234)
    assert argv[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=8" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert argv[argv.index("--master-port") + 1] == "1234"
    assert argv[-5:] == ["/x/bench.py", "--gpus", "8", "--steps", "5"]
    with pytest.raises(ValueError):
        launch_argv(1, "/x/bench.py", [])


def test_bench_dry_launch_prints_the_command():
    """`python bench.py --gpus 4 ...` with no WORLD_SIZE must start the ranks itself: --dry-launch shows the command."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "3", "--warmup", "1", "--dry-launch",
                        "--master-port", "29999"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    argv = json.loads(r.stdout.strip().splitlines()[-1])["launch"]
    assert argv[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in argv
    assert argv[argv.index("--master-port") + 1] == "29999"
    tail = argv[argv.index(BENCH) + 1:]
    assert tail[:6] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and "--dry-launch" not in tail


def test_bench_self_launch_runs_ranks_and_propagates_their_exit_code():
    """The real self-launch path on a box without GPUs: two ranks start under torch.distributed.run, each refuses
    to run ("needs a GPU"), and the parent exits non-zero with the ranks' message relayed — it neither dies on host
    logic before launching nor swallows the failure."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if torch.cuda.is_available():
        pytest.skip("CPU-only rehearsal")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--master-port", str(_free_port())], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0
    assert "bench.py needs a GPU" in (r.stderr + r.stdout)


def _timer_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fences = [0]

        def fence():
            fences[0] += 1
            dist.barrier()

        timer = RegionTimer(fence, dist, torch.device("cpu"))
        calls = [0]

        def step():                      # rank 1 is the slow rank
            calls[0] += 1
            time.sleep(0.004 if rank == 1 else 0.0005)

        regions = timer.measure(5, step)                   # short regions: repeated, count agreed through rank 0
        again = timer.measure(5, step, n_regions=len(regions))
        # a rank-local opinion about the count must not matter: rank 1 "thinks" one region is enough
        n = timer.agree(7 if rank == 0 else 1)
        with open(os.path.join(out_dir, f"r{rank}.json"), "w") as f:
            json.dump(dict(regions=regions, again=again, calls=calls[0], fences=fences[0], agreed=n), f)
    finally:
        dist.destroy_process_group()


def test_region_timer_under_gloo_world2(tmp_path):
    mp.spawn(_timer_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (json.load(open(tmp_path / f"r{r}.json")) for r in (0, 1))
    # same number of regions on both ranks, an odd count > 1 (every region was far below 50 ms)
    assert len(r0["regions"]) == len(r1["regions"]) >= 3 and len(r0["regions"]) % 2 == 1
    assert len(r0["again"]) == len(r0["regions"]) == len(r1["again"])
    # the reported time is the slow rank's (max over ranks): identical on both ranks, >= 5 x 4 ms
    assert r0["regions"] == r1["regions"] and r0["again"] == r1["again"]
    assert min(r0["regions"]) >= 5 * 0.004
    # exactly K steps per region on every rank, two fences per region
    n = len(r0["regions"]) + len(r0["again"])
    assert r0["calls"] == r1["calls"] == 5 * n
    assert r0["fences"] == r1["fences"] == 2 * n
    assert r0["agreed"] == r1["agreed"] == 7


def test_region_count_rule():
    assert RegionTimer.region_count(0.2) == 1
    assert RegionTimer.region_count(0.003) == 25
    c = RegionTimer.region_count(0.04)
    assert c % 2 == 1 and 3 <= c <= 25
