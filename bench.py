#!/usr/bin/env python3
"""bench.py — scenes/s of the GroupNet MS-HGNN forward on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W                       (BASELINE config 2 / 3: N=11, fp32)
    python bench.py --config c4 ...                                     (BASELINE config 4: N=50, B=1024, bf16)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W          (N > 1, one rank per GPU)

One "step" = one MS-HGNN forward over one batch of synthetic agent features already resident in HBM: cosine
affinity + top-k incidence of every scale (one fused launch), the pairwise module and one hyper module per scale
(model/GroupNet_nba.py:284-311), Gumbel noise drawn on the device inside the step, features written into the
concatenated (B, N, 64*(2+S)) tensor; with N > 1 ranks each rank owns its own scenes (config 3: 4096 scenes over
8 GPUs — weak scaling) and the outputs are all-gathered over RCCL/xGMI (`sharding.BucketedGather`: one call per
`--streams` steps, on a side stream).  The step is a captured hipGraph; consecutive steps are issued round-robin
on --streams (default 4) HIP streams, each with its own graph and output buffers, so the tail / small kernels /
all-gather of one step overlap the next step's matrix work (every step is still a complete forward of its own
batch).  A caller with a dependency between steps gets the single-stream figure, reported next to it
(`value_single_stream`).

When K steps take less than 50 ms the timed region (exactly K steps between two fences) is repeated and the
MEDIAN region is reported (`timed_regions`), so that a 3 ms measurement is not at the mercy of one hiccup.

Rank 0 prints one JSON line.  Besides the contract fields it carries
  roofline      the matrix-core kernel with the largest launch time: algorithmic FLOPs per launch / its average
                duration, measured here with HIP events on the stream it is launched on, in an instrumented pass;
  agg_hbm       the hyperedge aggregation gather+scatter kernels against the HBM roofline at N=11/B=4096
                (north_star target >= 30 %), measured the same way;
  cpu_baseline  the CPU oracle (a port of the reference's PyTorch path) timed on this box's host cores on a
                bounded sample of the same workload — a baseline, not a target.
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

CONFIGS = {
    # BASELINE.json configs[1] (the metric's configuration; configs[2] = the same per GPU on 8 GPUs)
    "c2": dict(N=11, scales=[2, 5, 11], B=512, dtype="f32",
               metric="scenes/sec GroupNet MS-HGNN forward, NBA N=11 B=512, 1/2/4/8 MI355X"),
    # BASELINE.json configs[3]: synthetic N=50 (SDD-like), B=1024, scales {2,4,8,16}, bf16 storage
    "c4": dict(N=50, scales=[2, 4, 8, 16], B=1024, dtype="bf16",
               metric="scenes/sec GroupNet MS-HGNN forward, synthetic N=50 B=1024 scales {2,4,8,16} bf16 (BASELINE config 4)"),
}
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: fp32 matrix peak
MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16 matrix peak (same guide)
X6_CEILING_TFLOPS = MFMA_BF16_PEAK_TFLOPS / 6.0   # bf16x6: fp32-accurate products = six bf16 part-products each
X3_CEILING_TFLOPS = MFMA_BF16_PEAK_TFLOPS / 3.0   # f16x3: three fp16 part-products each (fp16 MFMA rate = bf16 rate)


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
        """name -> (ms per launch, executed flops, launches, reference-form flops or None)"""
        out = {}
        for name, pr in self.pairs.items():
            ms = max(statistics.fmean(a.elapsed_time(b) for a, b, _ in pr) - overhead_ms, 1e-6)
            fl = pr[0][2]
            ex, ref = (fl if isinstance(fl, tuple) else (fl, None))
            out[name] = (ms, ex, len(pr), ref)
        return out


def pmc_traffic(kernel_prefixes, cfg_name):
    """HBM bytes per launch from the committed PMC passes of the SAME configuration (profiles/rNN_pmc_traffic_<cfg>.json;
    PMC counters cannot be read from inside the process; the file says how they were collected).  Returns
    (bytes or None, file name or None).  A kernel that is not in the newest file (renamed, or a different grid) yields
    None — never a stale number under a new name."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*pmc_traffic_{cfg_name}.json")))
    if not files:
        return None, None
    name = os.path.relpath(files[-1], ROOT)
    try:
        ks = json.load(open(files[-1]))["kernels"]
        tot = 0
        for pre in kernel_prefixes:
            alts = pre if isinstance(pre, (tuple, list)) else (pre,)     # alternatives: the one launched most often
            hit = [v for k, v in ks.items() if any(k.startswith(a) for a in alts)]
            if not hit:
                return None, name
            tot += max(hit, key=lambda v: v.get("launches", 0))["hbm_bytes"]
        return tot, name
    except Exception:
        return None, name


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


def train_step_leg(N, B, scales, dev, replays=30):
    """Forward + MSE loss + backward + SGD update of the same block, captured once (groupnet_amd.graphs.
    GraphedTrainStep) and replayed; a side figure next to the forward metric, not part of `value`."""
    from groupnet_amd.graphs import GraphedTrainStep
    from groupnet_amd.multiscale import MultiScaleHGNN
    torch.manual_seed(1)
    blk = MultiScaleHGNN(scales).to(dev).train()
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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_share():
    """(cores this process should use, cores visible).  A GPU box hands a one-GPU job a SHARE of its host cores
    (cgroup cpu.max; 16 per GPU on this pool) while sched_getaffinity still shows every core of the machine: 256
    torch threads on a 16-core share ran the oracle 140x slower than 16 threads.  Use the quota when there is one,
    else 16 per visible GPU, never more than what is visible."""
    visible = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(round(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    if quota is None:
        quota = 16 * max(1, torch.cuda.device_count())
    return max(1, min(visible, quota)), visible


def cpu_baseline(block_state, cfg, budget_s=24.0):
    """The oracle on the host cores: the same workload (pairwise + one hyper module per scale, host noise drawn as
    the reference does) on this job's share of the host cores (`cpu_share`).  SURVEY 8d: B = 32 / 512 / 4096 at N = 11
    (the configuration's own batch otherwise), the faithful form (materialised attention tensor, as the reference
    executes) and the decomposed form, median / min / max of the forwards that fit a bounded budget (>= 2 after one
    warm-up each).  `value` = the faithful form at the configuration's batch (scenes are independent: a slice of a
    large-N batch carries over)."""
    from oracle import ms_hgnn_oracle as O
    cores, visible = cpu_share()
    torch.set_num_threads(cores)
    sp, shs = block_state
    N, scales, B = cfg["N"], cfg["scales"], cfg["B"]
    # the (Bs, N*N, N, 128) attention input of the reference costs Bs*N^3*512 bytes (x3 live copies)
    cap = B if N <= 16 else max(1, min(B, int(1.5e9 // (N ** 3 * 512 * 3))))
    sizes = sorted({32, B, 4096}) if N <= 16 else [cap]

    def run(Bs, decomposed, budget, max_it):
        g = torch.Generator().manual_seed(1234)
        h = torch.randn(Bs, N, 64, generator=g)
        times = []
        with torch.no_grad():
            t_end = time.time() + budget
            while len(times) < 3 or (time.time() < t_end and len(times) < max_it):
                t0 = time.perf_counter()
                Up = [O.draw_uniform(s) for s in O.noise_shapes(Bs, N, None)]
                Uh = [[O.draw_uniform(s) for s in O.noise_shapes(Bs, N, sc)] for sc in scales]
                O.ms_hgnn_multiscale_forward(sp, shs, scales, h, Up, Uh, decomposed=decomposed)
                times.append(time.perf_counter() - t0)
                if len(times) >= 3 and time.time() >= t_end:
                    break
        t = times[1:]                      # (the first forward warms the allocator / thread pool)
        return dict(scenes=Bs, forwards=len(t), scenes_per_s=round(Bs / statistics.median(t), 1),
                    min=round(Bs / max(t), 1), max=round(Bs / min(t), 1))

    per = budget_s / (2.0 * len(sizes))
    table = {}
    for Bs in sizes:
        table[f"B{Bs}"] = dict(faithful=run(Bs, False, per, 10), decomposed=run(Bs, True, per, 10))
    main = table[f"B{min(B, cap)}"]["faithful"]
    return dict(value=main["scenes_per_s"], unit="scenes/s", min=main["min"], max=main["max"], cores=cores,
                cores_visible=visible, cpu_model=cpu_model(), kind="port", by_batch=table,
                sample=f"{main['forwards']} forwards of {main['scenes']} scenes at N={N}, scales {scales} (median, min, max "
                       f"of all but the first); torch {torch.__version__} CPU fp32 on {cores} threads, materialised "
                       f"attention tensor as the reference executes; by_batch: the same at the other batch sizes and "
                       f"the decomposed-attention variant")


def matrix_path(twin):
    """(description, peak TFLOP/s, name) of the matrix path the fp32 / bf16 entry points run on, for the roofline: the
    peak is the dense 16-bit MFMA peak divided by the part-products ONE product of the path costs."""
    from groupnet_amd import ops
    if twin:
        return ("v_mfma_f32_32x32x16_bf16, bf16 operands, fp32 accumulate", MFMA_BF16_PEAK_TFLOPS, "bf16")
    mode = ops.precision()
    if mode == "f16x3":
        return ("v_mfma_f32_32x32x16_f16, x = xh + xl (two fp16 parts), three part-products per product (fp32-accurate; "
                "operands beyond the fp16 range fall back to bf16x6 inside the launch)", X3_CEILING_TFLOPS, "f16x3")
    if mode == "bf16x6":
        return ("v_mfma_f32_32x32x16_bf16, x = x1+x2+x3 (bf16 parts), six part-products per product (fp32-accurate)",
                X6_CEILING_TFLOPS, "bf16x6")
    return ("v_mfma_f32_32x32x2_f32", MFMA_F32_PEAK_TFLOPS, "fp32")


def roofline_leg(block, f, cfg_name, twin, N, n_steps):
    """Brackets every launch of the matrix-core kernels of `n_steps` eager single-stream forwards with HIP events on the
    stream they are launched on.  Returns (roofline of the kernel with the largest launch time, all kernels).
    `achieved` = algorithmic fp32-equivalent FLOPs per launch / average launch duration; `peak` = what the matrix path
    can issue at best: 2.5 PF dense 16-bit MFMA / part-products per product (3 for f16x3, 6 for bf16x6, 1 for the
    twins); the fraction of the fp32 matrix peak (157.3 TF, the metric's precision) is reported next to it."""
    import groupnet_amd as G
    from groupnet_amd import ops
    G.set_noise_mode("device", seed=99)
    probe = Probe()
    for _ in range(2):
        block(f)
    ops.launch_probe = probe
    torch.cuda.synchronize()
    for _ in range(n_steps):
        torch.cuda._sleep(3_000_000 if N <= 16 else 12_000_000)   # keep the host ahead: no launch gaps in the brackets
        block(f)
    ops.launch_probe = None
    torch.cuda.synchronize()
    overhead = empty_bracket_ms()
    summ = probe.summary(overhead)
    desc, peak, pname = matrix_path(twin)
    parts = {"f16x3": 3, "bf16x6": 6}.get(pname, 1)
    kernels = {}
    for k, (ms, fl, n, ref) in summ.items():
        tf = fl / (ms * 1e-3) / 1e12
        kernels[k] = dict(avg_launch_us=round(ms * 1e3, 2), flops_per_launch=fl, achieved_tflops=round(tf, 2),
                          frac=round(tf / peak, 4), frac_of_fp32_matrix_peak=round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                          executed_16bit_tflops=round(parts * tf, 1))
        if ref is not None:
            kernels[k].update(reference_form_flops_per_launch=ref, reference_form_tflops=round(ref / (ms * 1e-3) / 1e12, 2))
    dom = max(summ, key=lambda k: summ[k][0])       # the kernel with the largest launch time
    ms, fl, nl, ref = summ[dom]
    ach = fl / (ms * 1e-3) / 1e12
    # (the stage's device kernels: large bf16 launches run the two-row-blocks-per-wave kernels; the twins' typed
    # aggregation is two launches, the scene form of the pairwise module and the hyper modules')
    names = [{"agg_mlp_kernel": ("agg_x_kernel", "agg_rb2_kernel"),
              "edge_mlp_gumbel_kernel": ("edge_x_kernel", "edge_rb2_kernel"),
              "node_stage_kernel": "node_stage_kernel", "mlp2_kernel": "mlp2_x_kernel"}.get(dom, dom)]
    traffic, tfile = pmc_traffic(names, cfg_name)
    if dom == "agg_mlp_kernel" and twin and traffic is not None:
        extra, _ = pmc_traffic(["agg_scene_kernel"], cfg_name)
        traffic = None if extra is None else traffic + extra
    roof = dict(kernel=dom + {"agg_mlp_kernel": " (typed aggregation MLP, all modules in one grouped launch)",
                              "edge_mlp_gumbel_kernel": " (edge MLP 64-128-64 + distribution/factor heads + Gumbel "
                                                        "softmax epilogue, all modules in one grouped launch)",
                              "node_stage_kernel": " (node MLP 64-256-64 + attention projections + per-node typed "
                                                   "layer 1, all modules in one grouped launch)"}.get(dom, ""),
                bound="mfma", achieved=round(ach, 2), peak=round(peak, 1), unit="TFLOP/s",
                frac=round(ach / peak, 4), traffic=traffic, traffic_source=tfile,
                avg_launch_us=round(ms * 1e3, 2), flops_per_launch=fl, launches_timed=nl,
                event_pair_overhead_us=round(overhead * 1e3, 2), matrix_path=desc,
                peak_is=f"{MFMA_BF16_PEAK_TFLOPS:.0f} TFLOP/s dense 16-bit MFMA / {parts} part-product(s) per product",
                frac_of_fp32_matrix_peak=round(ach / MFMA_F32_PEAK_TFLOPS, 4),
                executed_16bit_tflops=round(parts * ach, 1),
                measured="single-stream instrumented eager pass (in the timed region steps overlap across streams, "
                         "which stretches every kernel; profiles/ holds both views)")
    if ref is not None:
        # `achieved` counts what the launch EXECUTES.  The typed aggregation runs algebraically reduced forms (pairwise
        # graph: layer 1 once per node, layer 2 and the type weights once per node behind the sum H^T feat — the node
        # form; symmetric pairs), so it does the reference's work with a fraction of the reference's operations:
        # the same launch time against the reference form's count (every ordered edge through both
        # layers) is reported next to it, and named for what it is.
        roof.update(reference_form_flops_per_launch=ref,
                    reference_form_tflops=round(ref / (ms * 1e-3) / 1e12, 2),
                    reference_form_frac=round(ref / (ms * 1e-3) / 1e12 / peak, 4),
                    flops_note="flops_per_launch = executed (matrix cores + the 3 flops per pair member and hidden value of "
                               "the node form's VALU sum); reference_form_* = every ordered edge through both layers of "
                               "every type (model/MS_HGNN_batch.py:262-265) in the same launch time")
    return roof, kernels


def agg_hbm_leg(dev, Bb=4096, Nn=11, sets=16):
    """north_star: the hyperedge aggregation gather + scatter kernels against the HBM roofline at N=11 / B=4096 (fp32).
    Two figures: `frac` = the launch pair repeated on ONE set of buffers (60 MB: resident in the 256 MiB Infinity Cache
    after the first pass — an on-die figure) and `frac_cold` = the same launches rotating over `sets` distinct buffer
    sets (~1 GB footprint, 4x the Infinity Cache: every launch streams from HBM).  `frac_cold` is the HBM figure."""
    from groupnet_amd import ops
    by = agg_hbm_bytes(Bb, Nn, Nn)
    bufs = []
    for i in range(sets):
        ori = torch.randn(Bb, Nn, 64, device=dev)
        _, Hs, _ = ops.affinity_topk(ori, [5], want_corr=False)
        bufs.append((ori, Hs[0], torch.randn(Bb, Nn, 64, device=dev)))
    ori, H, feat = bufs[0]
    t_g = time_kernel_ms(lambda: ops.agg_gather(ori, H))
    t_s = time_kernel_ms(lambda: ops.agg_scatter(feat, H, ori))
    gbs = by / ((t_g + t_s) * 1e-3) / 1e9
    it = [0]

    def rot(fn):
        def go():
            o, h, ft = bufs[it[0] % sets]
            it[0] += 1
            fn(o, h, ft)
        return go
    tc_g = time_kernel_ms(rot(lambda o, h, ft: ops.agg_gather(o, h)), reps=4 * sets, warm=sets)
    tc_s = time_kernel_ms(rot(lambda o, h, ft: ops.agg_scatter(ft, h, o)), reps=4 * sets, warm=sets)
    gbs_c = by / ((tc_g + tc_s) * 1e-3) / 1e9
    footprint = sets * (3 * Bb * Nn * 64 * 4 + Bb * Nn * Nn * 4 + Bb * Nn * 64 * 4 + Bb * Nn * 128 * 4)   # + outputs
    atr, afile = pmc_traffic(["agg_gather_kernel", "agg_scatter_kernel"], "c2")
    return dict(kernel="agg_gather_kernel + agg_scatter_kernel (hyper, E=N=11, B=4096, fp32)", bound="hbm",
                achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
                achieved_cold=round(gbs_c, 1), frac_cold=round(gbs_c / HBM_PEAK_GBS, 4),
                cold_is=f"{sets} rotating buffer sets, ~{footprint / 2**20:.0f} MiB touched between two uses of a line "
                        f"(Infinity Cache: 256 MiB)",
                traffic=atr, traffic_source=afile,
                gather_us=round(t_g * 1e3, 2), scatter_us=round(t_s * 1e3, 2),
                gather_us_cold=round(tc_g * 1e3, 2), scatter_us_cold=round(tc_s * 1e3, 2), bytes_per_launch_pair=by)


def parity_mode_leg(block, f, B, N, steps=30):
    """SURVEY 8d "parity mode": the reference's noise contract — every module call draws torch.rand((B,E,K)) from the
    global CPU generator and the uniforms are uploaded (MS_HGNN_batch.py:454) — eager, one stream.  A side figure: the
    step is bound by the host draw + four H2D copies, not by the GPU."""
    import groupnet_amd as G
    G.set_noise_mode("host")
    for _ in range(3):
        block(f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        block(f)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return dict(what="eager forward with host noise: torch.rand on the CPU generator per module call + upload (the "
                     "reference's RNG contract; outputs match the reference's for the same seed)",
                scenes_per_s=round(B * steps / el, 1), ms_per_step=round(el / steps * 1e3, 3), steps=steps)


def per_module_leg(block, f):
    """Device time of each module of the block ALONE (eager launches bracketed by events, device noise): the pairwise
    module and one hyper module per scale — each includes its own node / edge / aggregation / closing launches; the hyper
    figures include the affinity + top-k launch that builds their incidence."""
    import groupnet_amd as G
    from groupnet_amd import ops
    G.set_noise_mode("device", seed=7)
    out = {}
    t = time_kernel_ms(lambda: block.interaction(f), reps=20, warm=3)
    out["pairwise"] = round(t * 1e3, 1)
    for s, m in zip(block.hyper_scales, block.interaction_hyper):
        def go(m=m, s=s):
            corr, Hs, _ = ops.affinity_topk(f, [s], want_corr=True)
            m(f, corr, H=Hs[0])
        out[f"hyper_scale_{s}"] = round(time_kernel_ms(go, reps=20, warm=3) * 1e3, 1)
    out["unit"] = "us per forward of the module alone (the grouped block launches all of them together)"
    return out


def c4_leg(dev, steps=40):
    """BASELINE config 4 (N=50, B=1024, scales {2,4,8,16}, bf16 twins) as a short leg of the default run, so that its
    number is observed by the driver: hipGraph replay on one stream, scenes/s, and the dominant kernel against the bf16
    MFMA peak (the instrumented pass of `roofline_leg`)."""
    from groupnet_amd.graphs import GraphedMultiScale
    from groupnet_amd.multiscale import MultiScaleHGNN
    cfg = CONFIGS["c4"]
    torch.manual_seed(0)
    blk = MultiScaleHGNN(cfg["scales"]).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(1234)
    f4 = torch.randn(cfg["B"], cfg["N"], 64, generator=g, device=dev).to(torch.bfloat16)
    gr = GraphedMultiScale(blk, cfg["B"], cfg["N"], seed=5, dtype=torch.bfloat16)
    gr.f_in.copy_(f4)
    for _ in range(5):
        gr()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gr()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    roof, kernels = roofline_leg(blk, f4, "c4", True, cfg["N"], 10)
    return dict(workload=cfg["metric"], scenes_per_s=round(cfg["B"] * steps / el, 1), ms_per_step=round(el / steps * 1e3, 4),
                steps=steps, streams=1, dtype="bf16", roofline=roof, mfma_kernels=kernels)


def self_launch(n_gpus, port, dry):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>` as a CHILD process and return its exit
    code.  The parent never initialises the GPU; the children's stdout (rank 0 prints the one JSON line) and stderr
    pass straight through."""
    import subprocess
    from groupnet_amd.timing import launch_argv
    passed = [a for a in sys.argv[1:] if a != "--dry-launch"]
    argv = launch_argv(n_gpus, os.path.abspath(__file__), passed, port)
    if dry:
        print(json.dumps({"launch": argv}), flush=True)
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(argv, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2",
                    help="c2: BASELINE metric configuration (N=11, B=512 per GPU, fp32); c4: N=50, B=1024, bf16 twins")
    ap.add_argument("--batch-per-gpu", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--streams", type=int, default=4,
                    help="consecutive steps are issued round-robin on this many HIP streams (each with its own "
                         "captured graph and output buffers), so the tail of one step overlaps the head of the next")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the training-step side measurement")
    ap.add_argument("--no-c4-leg", action="store_true", help="skip the short BASELINE-config-4 side leg of the default run")
    ap.add_argument("--no-side-legs", action="store_true",
                    help="profiling runs: only the timed regions, the roofline pass and the HBM leg (no parity-mode / "
                         "per-module / training / config-4 legs, whose launches of other sizes would mix into kernel statistics)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: run the multi-GPU code path (process group, bucketed all-gather) with one rank")
    ap.add_argument("--gather-full", action="store_true",
                    help="N > 1: all-gather the whole (B,N,64(2+S)) feature tensor instead of its computed columns")
    ap.add_argument("--master-port", type=int, default=29533, help="rendezvous port of the self-launched N > 1 run")
    ap.add_argument("--dry-launch", action="store_true",
                    help="with --gpus N > 1 and no WORLD_SIZE: print the launch command (JSON) instead of running it")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    if args.batch_per_gpu:
        cfg["B"] = args.batch_per_gpu

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: start the N ranks ourselves — from THIS process, which has not touched the GPU
        # (no exec, no re-launch after a HIP call) — relay the ranks' output (rank 0's JSON line) and exit with their code
        sys.exit(self_launch(args.gpus, args.master_port, args.dry_launch))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
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

    from groupnet_amd import ops, sharding
    from groupnet_amd.timing import RegionTimer
    from groupnet_amd.graphs import GraphedMultiScale
    from groupnet_amd.multiscale import MultiScaleHGNN
    import groupnet_amd as G

    Bl, N, SCALES = cfg["B"], cfg["N"], cfg["scales"]
    twin = cfg["dtype"] == "bf16"
    tdt = torch.bfloat16 if twin else torch.float32
    B_total = Bl * world
    torch.manual_seed(0)                      # same seeded default-init weights on every rank
    block = MultiScaleHGNN(SCALES)
    block_state = ({k: v.detach().clone() for k, v in block.interaction.state_dict().items()},
                   [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in block.interaction_hyper])
    block.to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    f = torch.randn(Bl, N, 64, generator=g, device=dev).to(tdt)     # synthetic agent embeddings, resident in HBM
    S = 1 if args.no_graph else max(1, args.streams)
    # what crosses xGMI: the output embeddings = the COMPUTED columns (Bl, N, 64 (1 + scales)) of the feature tensor
    # (SURVEY 8e); its first 64 columns are a copy of the rank's own input, which no rank needs back
    gather_cols = block.out_features if args.gather_full else block.out_features - block.h_dim
    bg = sharding.BucketedGather(S, (Bl, N, gather_cols), dev, dtype=tdt) if distributed else None

    with torch.no_grad():
        if args.no_graph:
            G.set_noise_mode("device", seed=99 + 1000 * rank)
            streams = [torch.cuda.current_stream()]
            graphs = []
            runs = [lambda: block(f)[0]]
        else:
            streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else [torch.cuda.current_stream()]
            # every rank and every stream draws its own noise (a shard is not a copy of another shard)
            # graphs replayed side by side on S > 1 streams: the throughput form (affinity + top-k as its own small launch,
            # which fits beside the other streams' kernels); one stream: the latency form (it rides in the node stage)
            graphs = [GraphedMultiScale(block, Bl, N, seed=99 + i + 1000 * rank, dtype=tdt, affinity_tail=(S == 1))
                      for i in range(S)]
            for gr in graphs:
                gr.f_in.copy_(f)
            runs = [(lambda gr=gr: gr()[0]) for gr in graphs]
            # the single-stream figure (a caller with a dependency between steps) uses the latency form
            lat_graph = None
            if not distributed:
                lat_graph = graphs[0] if S == 1 else GraphedMultiScale(block, Bl, N, seed=99 + 1000 * rank, dtype=tdt,
                                                                       affinity_tail=True)
                lat_graph.f_in.copy_(f)
        torch.cuda.synchronize()
        step_no = [0]

        def step():
            i = step_no[0] % S
            step_no[0] += 1
            with torch.cuda.stream(streams[i]):
                out = runs[i]()
                if bg is not None:
                    bg.put(out if args.gather_full else out[..., block.h_dim:])
            return out

        def fence():
            if bg is not None:
                bg.flush()                # steps not yet gathered (count not a multiple of S)
                step_no[0] = 0
            torch.cuda.synchronize()
            if distributed:
                dist.barrier()
            torch.cuda.synchronize()

        # the timed-region protocol (fences, max-over-ranks clock, agreed region count): groupnet_amd/timing.py,
        # rehearsed under gloo in tests/test_bench_control.py
        timer = RegionTimer(fence, dist if distributed else None, dev)
        for _ in range(args.warmup):
            step()
        # short regions are repeated (same K steps each) and the median is reported; every rank takes the same count
        regions = timer.measure(args.steps, step)
        n_regions = len(regions)
        elapsed = statistics.median(regions)

        # ---- the same K steps on ONE stream, back to back (what a caller with a dependency between steps sees) ----
        def step1():
            with torch.cuda.stream(streams[0]):
                if args.no_graph:
                    runs[0]()
                else:
                    lat_graph()
        single = None
        if not distributed:
            for _ in range(min(args.warmup, 10)):
                step1()
            single = statistics.median(timer.measure(args.steps, step1, n_regions))

        # ---- roofline leg: instrumented eager pass (rank 0) ---------------------------------------------
        roof = agg = mfma_kernels = train = parity = modules = c4 = None
        if rank == 0:
            roof, mfma_kernels = roofline_leg(block, f, args.config, twin, N, min(args.steps, 50))
            agg = agg_hbm_leg(dev)         # (its two kernels run in no forward at N = 11: they cannot mix into its statistics)
            if world == 1 and not args.no_side_legs:
                parity = parity_mode_leg(block, f, Bl, N)
                modules = per_module_leg(block, f)
            G.set_noise_mode("host")
            # ---- one training step (fwd + loss + bwd + SGD) replayed from one hipGraph ------
            try:
                train = (train_step_leg(N, Bl, SCALES, dev)
                         if (world == 1 and not args.no_train_leg and not args.no_side_legs and not twin) else None)
            except Exception as e:      # the headline forward numbers stand on their own
                train = dict(error=f"{type(e).__name__}: {e}")
            # ---- BASELINE config 4 beside the metric's configuration (a short leg: driver-observed, not the headline) ----
            if world == 1 and args.config == "c2" and not args.no_c4_leg and not args.no_side_legs:
                try:
                    c4 = c4_leg(dev)
                except Exception as e:
                    c4 = dict(error=f"{type(e).__name__}: {e}")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(block_state, cfg)

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        line = {
            "metric": cfg["metric"],
            "value": round(B_total * args.steps / elapsed, 1),
            "unit": "scenes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": cfg["dtype"], "data": "synthetic",
            "timed_regions": len(regions),
            "ms_per_step_min_max": [round(min(regions) / args.steps * 1e3, 4), round(max(regions) / args.steps * 1e3, 4)],
            "value_single_stream": None if single is None else round(Bl * args.steps / single, 1),
            "ms_per_step_single_stream": None if single is None else round(single / args.steps * 1e3, 4),
            "config": {"workload": f"MS-HGNN forward ({args.config}): affinity + top-k + pairwise + hyper scales {SCALES}, "
                                   f"N={N} agents, {Bl} scenes per GPU (global batch {B_total}), "
                                   + ("bf16 storage / fp32 accumulate (the *_bf16 twins of every stage), " if twin else
                                      f"fp32 in and out (matrix stages: {matrix_path(False)[2]} — {matrix_path(False)[0]}), ")
                                   + f"device Philox noise, {'eager' if args.no_graph else f'hipGraph replay on {S} alternating streams'}"
                                   + ("" if args.no_graph or S == 1 else " (5 + 1 launches per forward: affinity + top-k as its own "
                                      "launch; value_single_stream: one stream, 5 launches, affinity + top-k in the node stage's tail)")
                                   + (f", + RCCL all-gather of the (B,N,{block.out_features - block.h_dim}) output embeddings (the computed "
                                      f"columns of the (B,N,{block.out_features}) feature tensor; the rest is the rank's input), one call per {S} "
                                      f"steps, overlapped on a side stream" if distributed else ""),
                       "global_batch": B_total, "agents": N, "scales": SCALES,
                       "parallelism": f"batch-sharded x{world}"},
            "roofline": roof, "agg_hbm": agg, "mfma_kernels": mfma_kernels, "train_step": train,
            "value_parity_mode": None if parity is None else parity["scenes_per_s"], "parity_mode": parity,
            "per_module_us": modules, "configs": {"c4": c4},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
