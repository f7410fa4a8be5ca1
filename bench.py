#!/usr/bin/env python3
"""bench.py — scenes/s of the GroupNet MS-HGNN forward on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W          (N > 1, one rank per GPU)

One "step" = one MS-HGNN forward over one batch of synthetic agent features already resident in
HBM: cosine affinity + top-k incidence of every scale (one fused launch), the pairwise module and
one hyper module per scale {2,5,11} (model/GroupNet_nba.py:284-311), Gumbel noise drawn on the
device inside the step, features written into the concatenated (B, N, 320) tensor; with N > 1 ranks
each rank owns 512 scenes (BASELINE config 3: 4096 scenes over 8 GPUs — weak scaling) and the step
ends with ONE all-gather of the output embeddings over RCCL/xGMI.  The step is a captured hipGraph;
consecutive steps are issued round-robin on --streams (default 4) HIP streams, each with its own
graph and output buffers, so the tail / small kernels / all-gather of one step overlap the next
step's matrix work (every step is still a complete forward of its own batch).

Rank 0 prints one JSON line.  Besides the contract fields it carries
  roofline      the dominant kernel (typed aggregation MLP, all modules in one grouped launch, fp32 MFMA):
                algorithmic FLOPs per launch / its average duration, measured here with HIP events on
                the stream it is launched on, in an instrumented pass over the same K steps;
  agg_hbm       the hyperedge aggregation gather+scatter kernels against the HBM roofline at
                N=11/B=4096 (north_star target >= 30 %), measured the same way;
  cpu_baseline  the CPU oracle (a port of the reference's PyTorch path) timed on this box's host
                cores on the same workload — a baseline, not a target.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_AGENTS = 11
SCALES = [2, 5, 11]
B_PER_GPU = 512
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: fp32 matrix peak
MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16 matrix peak (same guide)


def agg_mlp_flops(rows, K):
    """Algorithmic FLOPs of feat = sum_k ef_k * (W2k relu(W1k eo + b1k) + b2k) per launch:
    per row and type 2*(64*128) + 2*(128*64) MAC-flops, + 2*64 for the typed scale-and-add."""
    return rows * K * (2 * 64 * 128 + 2 * 128 * 64 + 2 * 64)


def agg_hbm_bytes(B, N, E):
    """gather 4(N*D + E*N + E*D) + scatter 4(E*D + E*N + N*D + 2N*D) per scene."""
    D = 64
    return B * (4 * (N * D + E * N + E * D) + 4 * (E * D + E * N + N * D + 2 * N * D))


class Probe:
    """Brackets every launch of the matrix-core kernels with HIP events on the stream it is launched on."""

    def __init__(self):
        self.pairs, self._open = {}, {}

    def __call__(self, name, flops, before):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        if before:
            self._open[name] = ev
        else:
            self.pairs.setdefault(name, []).append((self._open.pop(name), ev, flops))

    def summary(self, overhead_ms):
        out = {}
        for name, pr in self.pairs.items():
            ms = max(statistics.fmean(a.elapsed_time(b) for a, b, _ in pr) - overhead_ms, 1e-6)
            out[name] = (ms, pr[0][2], len(pr))
        return out


def pmc_traffic(kernel_prefixes):
    """HBM bytes per launch from the committed PMC passes (profiles/*pmc_traffic.json), or None.
    PMC counters cannot be read from inside the process; the file says how they were collected."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None
    try:
        ks = json.load(open(files[-1]))["kernels"]
        tot = 0
        for pre in kernel_prefixes:
            hit = [v for k, v in ks.items() if k.startswith(pre)]
            if not hit:
                return None
            tot += hit[0]["hbm_bytes"]
        return tot
    except Exception:
        return None


def empty_bracket_ms(n=50):
    """What an event pair with NOTHING between costs on this stream (subtracted from the brackets)."""
    pairs = []
    torch.cuda._sleep(1_000_000)
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        b.record()
        pairs.append((a, b))
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in pairs)


def time_kernel_ms(fn, reps=20, warm=3):
    """Average duration of the launches `fn` enqueues, with events on the current stream and the
    host kept ahead of the GPU by a spin kernel."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(2_000_000)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def train_step_leg(N, B, dev, replays=30):
    """Forward + MSE loss + backward + SGD update of the same block, captured once (groupnet_amd.graphs.
    GraphedTrainStep) and replayed; a side figure next to the forward metric, not part of `value`."""
    from groupnet_amd.graphs import GraphedTrainStep
    from groupnet_amd.multiscale import MultiScaleHGNN
    torch.manual_seed(1)
    blk = MultiScaleHGNN(SCALES).to(dev).train()
    f = torch.randn(B, N, 64, device=dev)
    tgt = torch.randn(B, N, blk.out_features, device=dev)
    step = GraphedTrainStep(blk, torch.optim.SGD(blk.parameters(), lr=1e-3), lambda o, H, t: ((o - t) ** 2).mean(),
                            B, N, [tuple(tgt.shape)], seed=1)
    step(f, tgt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(replays):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / replays * 1e3
    return dict(what="forward + MSE loss + backward + SGD step of the same block, one hipGraph replay per step",
                scenes=B, ms_per_step=round(ms, 3), scenes_per_s=round(B / (ms * 1e-3), 1), replays=replays)


def cpu_baseline(block_state, B, N, threads, budget_s=20.0):
    """The oracle on the host cores: same workload (B scenes, pairwise + 3 scales), host noise drawn
    as the reference does.  Bounded sample: as many full forwards as fit in ~budget_s (>= 2)."""
    from oracle import ms_hgnn_oracle as O
    torch.set_num_threads(threads)
    sp, shs = block_state
    g = torch.Generator().manual_seed(1234)
    h = torch.randn(B, N, 64, generator=g)
    times = []
    with torch.no_grad():
        t_end = time.time() + budget_s
        it = 0
        while it < 2 or (time.time() < t_end and it < 12):
            t0 = time.perf_counter()
            Up = [O.draw_uniform(s) for s in O.noise_shapes(B, N, None)]
            Uh = [[O.draw_uniform(s) for s in O.noise_shapes(B, N, sc)] for sc in SCALES]
            O.ms_hgnn_multiscale_forward(sp, shs, SCALES, h, Up, Uh, decomposed=False)
            times.append(time.perf_counter() - t0)
            it += 1
    best = statistics.median(times[1:]) if len(times) > 1 else times[0]
    return dict(value=B / best, unit="scenes/s", cores=threads, kind="port",
                sample=f"{len(times)} full forwards at B={B}, N={N}, scales {SCALES} (median of all but the first); "
                       f"torch {torch.__version__} CPU, materialised attention tensor as the reference executes")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch-per-gpu", type=int, default=B_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--streams", type=int, default=4,
                    help="consecutive steps are issued round-robin on this many HIP streams (each with its own "
                         "captured graph and output buffers), so the tail of one step overlaps the head of the next")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the training-step side measurement")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: run the multi-GPU code path (process group, bucketed all-gather) with one rank")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 through torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: groupnet_amd has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    distributed = world > 1 or args.force_dist
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from groupnet_amd import ops
    from groupnet_amd.graphs import GraphedMultiScale
    from groupnet_amd.multiscale import MultiScaleHGNN
    import groupnet_amd as G

    Bl, N = args.batch_per_gpu, N_AGENTS
    B_total = Bl * world
    torch.manual_seed(0)                      # same seeded default-init weights on every rank
    block = MultiScaleHGNN(SCALES)
    block_state = ({k: v.detach().clone() for k, v in block.interaction.state_dict().items()},
                   [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in block.interaction_hyper])
    block.to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    f = torch.randn(Bl, N, 64, generator=g, device=dev)     # synthetic agent embeddings, resident in HBM
    S = 1 if args.no_graph else max(1, args.streams)
    # Multi-GPU: the only exchange is the all-gather of the output embeddings.  S consecutive steps
    # (one per stream) fill one bank of a double-buffered staging area and are gathered by ONE RCCL call on a
    # side stream — a 4x larger message per collective than step by step, overlapped with the next S steps.
    if distributed:
        Fo = block.out_features
        outs = torch.empty((2, S, Bl, N, Fo), device=dev)
        gathered = torch.empty((2, world * S * Bl, N, Fo), device=dev)
        gather_stream = torch.cuda.Stream(device=dev)
        bank_free = [None, None]        # event: the gather that last read this bank has finished
        ready = [None] * S

    with torch.no_grad():
        if args.no_graph:
            G.set_noise_mode("device", seed=99)
            streams = [torch.cuda.current_stream()]
            runs = [lambda: block(f)[0]]
        else:
            streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else [torch.cuda.current_stream()]
            graphs = [GraphedMultiScale(block, Bl, N, seed=99 + i) for i in range(S)]
            for gr in graphs:
                gr.f_in.copy_(f)
            runs = [(lambda gr=gr: gr()[0]) for gr in graphs]
        torch.cuda.synchronize()
        step_no = [0]

        def gather(bank):
            with torch.cuda.stream(gather_stream):
                for ev in ready:
                    if ev is not None:
                        gather_stream.wait_event(ev)
                dist.all_gather_into_tensor(gathered[bank], outs[bank].view(S * Bl, N, Fo))
                ev = torch.cuda.Event()
                ev.record(gather_stream)
                bank_free[bank] = ev

        def step():
            k = step_no[0]
            i, bank = k % S, (k // S) % 2
            step_no[0] += 1
            with torch.cuda.stream(streams[i]):
                out = runs[i]()
                if distributed:
                    if bank_free[bank] is not None:
                        streams[i].wait_event(bank_free[bank])
                    outs[bank, i].copy_(out, non_blocking=True)
                    ready[i] = torch.cuda.Event()
                    ready[i].record(streams[i])
            if distributed and i == S - 1:
                gather(bank)
            return out

        def flush():
            # steps not yet gathered (step count not a multiple of S): gather their bank now
            k = step_no[0]
            if distributed and k % S != 0:
                gather((k // S) % 2)

        def fence():
            flush()
            torch.cuda.synchronize()
            if distributed:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        # ---- roofline leg: instrumented eager pass over the same K steps (rank 0) -------------------
        roof = agg = mfma_kernels = train = None
        if rank == 0:
            G.set_noise_mode("device", seed=99)
            probe = Probe()
            ops.launch_probe = probe
            torch.cuda.synchronize()
            for _ in range(min(args.steps, 50)):
                torch.cuda._sleep(1_500_000)        # keep the host ahead: no launch gaps inside the brackets
                block(f)
            ops.launch_probe = None
            torch.cuda.synchronize()
            overhead = empty_bracket_ms()
            summ = probe.summary(overhead)
            mfma_kernels = {k: dict(avg_launch_us=round(ms * 1e3, 2), flops_per_launch=fl,
                                    achieved_tflops=round(fl / (ms * 1e-3) / 1e12, 2),
                                    frac=round(fl / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4))
                            for k, (ms, fl, n) in summ.items()}
            x6 = ("agg_mlp_kernel", "edge_mlp_gumbel_kernel") if ops.BF16X6 else ()
            for name in x6:
                if name in mfma_kernels:
                    # these kernels form their fp32-accurate products from six bf16 part-products on the bf16 cores:
                    # `frac` is their algorithmic (fp32) work against the fp32 matrix peak; against the cores they
                    # actually run on, the executed work is 6x and the peak the dense bf16 one
                    mk = mfma_kernels[name]
                    mk["matrix_path"] = "v_mfma_f32_32x32x16_bf16, x = x1+x2+x3 (bf16 parts), six part-products (fp32-accurate)"
                    mk["executed_bf16_tflops"] = round(6 * mk["achieved_tflops"], 1)
                    mk["frac_of_bf16_peak"] = round(6 * mk["achieved_tflops"] / MFMA_BF16_PEAK_TFLOPS, 4)
            dom = max(summ, key=lambda k: summ[k][0])       # the kernel with the largest launch time
            ms, fl, nl = summ[dom]
            ach = fl / (ms * 1e-3) / 1e12
            roof = dict(kernel=dom + (" (typed aggregation MLP: pair form of the pairwise module + 3 hyper modules, "
                                      "one grouped launch)" if dom == "agg_mlp_kernel" else
                                      " (edge MLP 64-128-64 + distribution/factor heads + Gumbel softmax epilogue, pair rows "
                                      "of the pairwise module + 3 hyper modules, one grouped launch; fp32 MFMA)"
                                      if dom == "edge_mlp_gumbel_kernel" else ""),
                        bound="mfma", achieved=round(ach, 2), peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=round(ach / MFMA_F32_PEAK_TFLOPS, 4), traffic=pmc_traffic([dom]),
                        avg_launch_us=round(ms * 1e3, 2), flops_per_launch=fl, launches_timed=nl,
                        event_pair_overhead_us=round(overhead * 1e3, 2),
                        measured="single-stream instrumented pass (in the timed region steps overlap across streams, "
                                 "which stretches every kernel; profiles/ holds both views)")
            if dom in x6:
                roof.update(note="algorithmic fp32 FLOPs against the fp32 matrix peak; the kernel executes them as six "
                                 "bf16 part-products per product on the bf16 cores (fp32-accurate), see mfma_kernels",
                            executed_bf16_tflops=mfma_kernels[dom]["executed_bf16_tflops"],
                            frac_of_bf16_peak=mfma_kernels[dom]["frac_of_bf16_peak"])
            # ---- north_star: hyperedge aggregation gather+scatter vs HBM at N=11 / B=4096 ----------------
            Bb = 4096
            ori = torch.randn(Bb, N, 64, device=dev)
            _, Hs, _ = ops.affinity_topk(ori, [5], want_corr=False)
            H = Hs[0]
            feat = torch.randn(Bb, N, 64, device=dev)
            t_g = time_kernel_ms(lambda: ops.agg_gather(ori, H))
            t_s = time_kernel_ms(lambda: ops.agg_scatter(feat, H, ori))
            by = agg_hbm_bytes(Bb, N, N)
            gbs = by / ((t_g + t_s) * 1e-3) / 1e9
            agg = dict(kernel="agg_gather_kernel + agg_scatter_kernel (hyper, E=N=11, B=4096)", bound="hbm",
                       achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
                       traffic=pmc_traffic(["agg_gather_kernel", "agg_scatter_kernel"]),
                       gather_us=round(t_g * 1e3, 2), scatter_us=round(t_s * 1e3, 2),
                       bytes_per_launch_pair=by)
            G.set_noise_mode("host")
            # ---- one training step (fwd + loss + bwd + SGD) replayed from one hipGraph ------
            try:
                train = train_step_leg(N, Bl, dev) if (world == 1 and not args.no_train_leg) else None
            except Exception as e:      # the headline forward numbers stand on their own
                train = dict(error=f"{type(e).__name__}: {e}")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        threads = max(1, min(16, len(os.sched_getaffinity(0))))
        cpu = cpu_baseline(block_state, Bl, N, threads)

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        line = {
            "metric": "scenes/sec GroupNet MS-HGNN forward, NBA N=11 B=512, 1/2/4/8 MI355X",
            "value": round(B_total * args.steps / elapsed, 1),
            "unit": "scenes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"MS-HGNN forward: affinity + top-k + pairwise + hyper scales {SCALES}, "
                                   f"N={N} agents, {Bl} scenes per GPU (global batch {B_total}), fp32 in and out"
                                   + (" (edge and aggregation MLPs: fp32-accurate three-part bf16 products on the bf16 "
                                      "matrix cores, other kernels fp32 MFMA), " if ops.BF16X6 else ", ")
                                   + f"device Philox noise, {'eager' if args.no_graph else f'hipGraph replay on {S} alternating streams'}"
                                   + (f", + RCCL all-gather of the (B,N,320) embeddings, one call per {S} steps, "
                                      f"overlapped on a side stream" if distributed else ""),
                       "global_batch": B_total, "agents": N, "scales": SCALES,
                       "parallelism": f"batch-sharded x{world}"},
            "roofline": roof, "agg_hbm": agg, "mfma_kernels": mfma_kernels, "train_step": train,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
