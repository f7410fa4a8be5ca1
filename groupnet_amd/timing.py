"""The timed-region protocol of bench.py, factored out so that its distributed control flow — fences, the
max-over-ranks clock, the agreed number of repeated regions — runs in the CPU suite under gloo with a stub step
(tests/test_bench_control.py), and `launch_argv`, the command a plain `python bench.py --gpus N` re-launches itself
with (one rank per GPU under torch.distributed.run).  Nothing here touches a GPU by itself: the caller passes the
device fence."""
from __future__ import annotations

import statistics
import sys
import time
from typing import Callable, List, Optional, Sequence

import torch


def launch_argv(n_gpus: int, script: str, script_args: Sequence[str], port: int = 29533) -> List[str]:
    """argv of the N-rank launch of `script` on ONE node: one process per GPU, rendezvous on 127.0.0.1 (the
    container hostname may not resolve)."""
    if n_gpus < 2:
        raise ValueError("launch_argv is for N > 1 ranks")
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), script, *script_args]


class RegionTimer:
    """Times regions of exactly `n_steps` calls of `stepper` between two fences.

    `fence()` must leave the device idle and every rank at the same point (bench.py: flush the bucketed gather,
    torch.cuda.synchronize(), dist.barrier(), synchronize again).  With a process group the region time is the MAX
    over ranks (one all_reduce of a float64 on `device`), and every decision that changes the number of collectives
    a rank issues — how many regions to run — is taken by rank 0 and broadcast, so that the ranks cannot diverge."""

    def __init__(self, fence: Callable[[], None], dist=None, device: Optional[torch.device] = None,
                 clock: Callable[[], float] = time.perf_counter):
        self.fence, self.dist, self.device, self.clock = fence, dist, device, clock

    def region(self, n_steps: int, stepper: Callable[[], object]) -> float:
        self.fence()
        t0 = self.clock()
        for _ in range(n_steps):
            stepper()
        self.fence()
        el = self.clock() - t0
        if self.dist is not None:
            t = torch.tensor([el], device=self.device, dtype=torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)          # the slowest rank's clock
            el = float(t.item())
        return el

    def agree(self, value: int) -> int:
        """rank 0's `value` on every rank."""
        if self.dist is None:
            return int(value)
        t = torch.tensor([int(value)], device=self.device, dtype=torch.int64)
        self.dist.broadcast(t, 0)
        return int(t.item())

    @staticmethod
    def region_count(first: float, min_region_s: float = 0.05, target_s: float = 0.25, max_regions: int = 25) -> int:
        """How many regions to run given the first one's time: 1 when it is long enough to stand on its own, else
        an odd number (a median) that fills ~target_s, at most max_regions."""
        if first >= min_region_s:
            return 1
        return min(max_regions, max(3, int(target_s / max(first, 1e-4))) | 1)

    def measure(self, n_steps: int, stepper: Callable[[], object], n_regions: Optional[int] = None) -> List[float]:
        """Region times; short regions are repeated (same n_steps each) — every rank runs the same count."""
        regions = [self.region(n_steps, stepper)]
        n = self.agree(self.region_count(regions[0]) if n_regions is None else n_regions)
        while len(regions) < n:
            regions.append(self.region(n_steps, stepper))
        return regions


def median(xs: Sequence[float]) -> float:
    return statistics.median(xs)
