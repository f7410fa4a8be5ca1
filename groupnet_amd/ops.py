"""Python faces of the HIP kernels: argument checking, output allocation, launch.

One function per C-ABI entry point of include/groupnet_hip.h.  Tensors must be fp32,
contiguous and on a HIP device; anything else raises ValueError (the reference would
instead crash on device tensors, SURVEY.md §8b "Errors").  Nothing here touches the
CPU oracle or falls back to torch math.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import _P, check, load, stream_handle

Tensor = torch.Tensor
FEAT = 64


_ACT_DTYPES = (torch.float32, torch.bfloat16)   # storage types of activations: the *_f32 kernels and their *_bf16 twins


def _req(t: Tensor, name: str, shape: Optional[Sequence[Optional[int]]] = None, dtype=torch.float32) -> Tensor:
    """`dtype`: the required dtype, or a tuple of admissible ones (activations: fp32 or the bf16 twins)."""
    if not isinstance(t, torch.Tensor):
        raise ValueError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise ValueError(f"{name}: must live on the GPU (got device {t.device}); groupnet_amd has no CPU path")
    if (t.dtype not in dtype) if isinstance(dtype, tuple) else (t.dtype != dtype):
        want = " or ".join(str(d) for d in dtype) if isinstance(dtype, tuple) else str(dtype)
        raise ValueError(f"{name}: must be {want} (got {t.dtype})")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if shape is not None:
        if t.dim() != len(shape) or any(s is not None and int(d) != int(s) for d, s in zip(t.shape, shape)):
            raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


def _same_device(*ts: Tensor) -> None:
    dev = ts[0].device
    for t in ts[1:]:
        if t is not None and t.device != dev:
            raise ValueError(f"tensors on different devices: {dev} vs {t.device}")


def _ptr(t: Optional[Tensor]) -> ctypes.c_void_p:
    return _P(0 if t is None else t.data_ptr())


# ---- optional instrumentation ----------------------------------------------------------------------
# bench.py brackets the launches of the matrix-core kernels with HIP events on the stream they are
# launched on (the roofline leg of the bench contract).  `launch_probe(name, flops, before)` is None in
# normal use and costs one comparison per launch.
launch_probe = None


BF16X6 = os.environ.get("GN_BF16X6", "1") != "0"   # fp32 entry points: fp32-accurate products on the bf16 cores
# ... and, on top of it, the two-part fp16 path ("f16x3": three part-products per product instead of six, bf16x6 as the
# in-kernel fallback for operands beyond the fp16 range).  GN_PRECISION = bf16x6 keeps the six-product path.
F16X3 = BF16X6 and os.environ.get("GN_PRECISION", "f16x3").lower() != "bf16x6"


# the fused affinity + top-k launch as the tail workgroups of the first node stage (GN_AFFINITY_TAIL = 0: its own launch)
_AFFINITY_TAIL = os.environ.get("GN_AFFINITY_TAIL", "1") != "0"


def precision() -> str:
    """Matrix path of the fp32 entry points: 'f16x3' (default), 'bf16x6' or 'fp32' (the fp32 matrix cores)."""
    return "f16x3" if (BF16X6 and F16X3) else ("bf16x6" if BF16X6 else "fp32")


def set_precision(mode: str) -> None:
    """Select the matrix path of the fp32 entry points for launches issued from now on: 'f16x3' | 'bf16x6' | 'fp32'.
    (Captured graphs keep the path they were captured with.)"""
    global BF16X6, F16X3
    mode = mode.lower()
    if mode not in ("f16x3", "bf16x6", "fp32"):
        raise ValueError("precision: 'f16x3', 'bf16x6' or 'fp32'")
    BF16X6 = mode != "fp32"
    F16X3 = mode == "f16x3"


def _twin(dtype: torch.dtype) -> bool:
    return dtype == torch.bfloat16


def _fn(stem: str, dtype: torch.dtype):
    """The C entry point of a stage for a storage type: `<stem>_f32` or its `<stem>_bf16` twin."""
    return getattr(load(), stem + ("_bf16" if _twin(dtype) else "_f32"))


class XImages:
    """bf16-core weight images of one packed-weight set (`gn_split_bf16_f32`), built on demand per number of
    parts (3: the fp32-accurate path of the fp32 entry points, 1: the bf16 twins) from hidden-tile-major fp32
    tile streams of a `PackPlan` arena, and rebuilt whenever the owner bumps `version` after a refresh."""

    def __init__(self):
        self.src = {}
        self.img = {}
        self.version = 0

    def add(self, name: str, tiles: Tensor) -> None:
        self.src[name] = tiles

    def bump(self) -> None:
        self.version += 1

    def get(self, name: str, parts: int) -> Tensor:
        hit = self.img.get((name, parts))
        src = self.src[name]
        if hit is None:
            n = src.numel() // 1024 * 2 * parts * 64 * 8
            # (two fp16 parts: + the 16-byte flag word behind the image, zero-initialised once)
            buf = (torch.zeros(n + 8, dtype=torch.int16, device=src.device) if parts == 2 else
                   torch.empty(n, dtype=torch.int16, device=src.device))
            hit = self.img[(name, parts)] = [buf, -1]
        if hit[1] != self.version:
            split_bf16(src, hit[0], parts)
            hit[1] = self.version
        return hit[0]


def _ximg(pk: dict, name: str, dtype: torch.dtype) -> int:
    """Device address of image `name` for a launch on `dtype` activations, or 0 (fp32 entry point with the
    bf16-core path switched off: the launch then uses the plain packed stream on the fp32 matrix cores)."""
    xi = pk.get("xi")
    if _twin(dtype):
        if xi is None or name not in xi.src:
            raise ValueError(f"no bf16 image '{name}' for this weight set")
        return xi.get(name, 1).data_ptr()
    if not BF16X6 or xi is None or name not in xi.src:
        return 0
    return xi.get(name, 3).data_ptr()


def _himg(pk: dict, name: str, dtype: torch.dtype) -> int:
    """Device address of the two-part fp16 image `name` (f16x3 path of the fp32 entry points), or 0."""
    xi = pk.get("xi")
    if _twin(dtype) or not (BF16X6 and F16X3) or xi is None or name not in xi.src:
        return 0
    return xi.get(name, 2).data_ptr()


class _Probed:
    """flops: the EXECUTED fp32-equivalent count of the launch, or (executed, reference form) where a stage runs an
    algebraically reduced form (typed aggregation: the reference form is every ordered edge through both layers,
    SURVEY.md 8d)."""
    __slots__ = ("name", "flops", "probe")

    def __init__(self, name: str, flops):
        self.name, self.flops, self.probe = name, flops, launch_probe

    def __enter__(self):
        if self.probe is not None:
            self.probe(self.name, self.flops, True)

    def __exit__(self, *exc):
        if self.probe is not None:
            self.probe(self.name, self.flops, False)
        return False


# ---- A0 / A1 -----------------------------------------------------------------------------------
def affinity(f: Tensor) -> Tensor:
    """corr = normalize(f) @ normalize(f)^T   (model/GroupNet_nba.py:284-286)."""
    _req(f, "f", (None, None, None))
    B, N, D = f.shape
    corr = torch.empty((B, N, N), dtype=f.dtype, device=f.device)
    with torch.cuda.device(f.device):
        check(load().gn_affinity_f32(_ptr(f), _ptr(corr), B, N, D, stream_handle()), "gn_affinity_f32")
    return corr


def _alloc_incidence(B: int, N: int, scales: Sequence[int], device) -> List[Tensor]:
    out = []
    for s in scales:
        s = int(s)
        if s > N:
            raise RuntimeError("selected index k out of range")  # what torch.topk raises (MS_HGNN_batch.py:382)
        E = 1 if s == N else N
        out.append(torch.empty((B, E, N), dtype=torch.float32, device=device))
    return out


def _scale_args(Hs: List[Tensor], scales: Sequence[int]):
    n = len(Hs)
    Hl = (_P * n)(*[h.data_ptr() for h in Hs])
    kl = (ctypes.c_int * n)(*[int(s) for s in scales])
    return Hl, kl, n


def topk_incidence(corr: Tensor, scales: Sequence[int]) -> List[Tensor]:
    """H per scale from corr (MS_HGNN_hyper.init_adj_attention, model/MS_HGNN_batch.py:372-388)."""
    _req(corr, "corr", (None, None, None))
    B, N, N2 = corr.shape
    if N != N2:
        raise ValueError(f"corr: expected (B,N,N), got {tuple(corr.shape)}")
    if not 1 <= len(scales) <= 8:
        raise ValueError("between 1 and 8 scales per call")
    Hs = _alloc_incidence(B, N, scales, corr.device)
    Hl, kl, n = _scale_args(Hs, scales)
    with torch.cuda.device(corr.device):
        check(load().gn_topk_incidence_f32(_ptr(corr), Hl, kl, n, B, N, stream_handle()), "gn_topk_incidence_f32")
    return Hs


def listall_incidence(corr: Tensor, scale: int) -> Tensor:
    """H of MS_HGNN_hyper.init_adj_attention_listall (model/MS_HGNN_batch.py:390-414): (B,1,N) of ones when
    scale == N, else (B,N,N) with row i = the best group of max(scale,1) agents containing i (exhaustive
    search over C(N-1, scale-1) candidates, first maximum in torch.combinations order)."""
    _req(corr, "corr", (None, None, None))
    B, N, N2 = corr.shape
    if N != N2:
        raise ValueError("corr must be (B, N, N)")
    scale = int(scale)
    if scale > N:
        raise RuntimeError("group size larger than the number of agents")   # the reference cannot build its table either
    H = torch.empty((B, 1 if scale == N else N, N), dtype=corr.dtype, device=corr.device)
    if B == 0:
        return H
    with torch.cuda.device(corr.device):
        check(load().gn_listall_incidence_f32(_ptr(corr), _ptr(H), B, N, scale, stream_handle()),
              "gn_listall_incidence_f32")
    return H


def fused_affinity_fits(N: int, D: int, x_dim: int = 0) -> bool:
    """Whether one scene's tile of the fused affinity+top-k launch fits its 128 KiB LDS budget."""
    return N * (D + 4 + x_dim) * 4 + 8 + N * N * 8 <= 128 * 1024      # rows + 64-bit ranking keys (+ raw inputs)


def affinity_topk(f: Optional[Tensor], scales: Sequence[int], want_corr: bool = True, f_out: Optional[Tensor] = None,
                  want_H_cat: bool = False, counter: Optional[Tensor] = None, counter_add: int = 0,
                  embed: Optional[Tuple[Tensor, Tensor, Tensor]] = None):
    """Fused A0+A1: f -> (corr, [H_s], H_cat) in one launch.

    Extras for the multiscale block (no copy kernels after this launch): ``f_out`` — a last-dim slice
    (B, N, D) of a wider contiguous tensor that also receives f; ``want_H_cat`` — also build
    cat(H_s, dim=1); ``counter``/``counter_add`` — advance the device Philox position.
    ``embed`` = (x_raw (B,N,xd), M (D,xd), c (N,D)): f itself is computed in the launch as M x + c[n]
    (pass f=None); a fourth return value then carries f (B,N,D)."""
    f_contig = None
    if embed is not None:
        x_raw, M, c = embed
        _req(x_raw, "x_raw", (None, None, None))
        B, N, xd = x_raw.shape
        _req(M, "M", (None, xd))
        D = M.shape[0]
        _req(c, "c", (N, D))
        _same_device(x_raw, M, c)
        f_contig = torch.empty((B, N, D), dtype=x_raw.dtype, device=x_raw.device)
        f = f_contig          # shapes / device for the allocations below; the kernel ignores its contents
    _req(f, "f", (None, None, None), _ACT_DTYPES)
    if _twin(f.dtype) and embed is not None:
        raise ValueError("the embedding front-end is fp32 only")
    B, N, D = f.shape
    corr = torch.empty((B, N, N), dtype=torch.float32, device=f.device) if want_corr else None
    Hs = _alloc_incidence(B, N, scales, f.device)
    Hl, kl, n = _scale_args(Hs, scales)
    ex = _lib.BlockExtras()
    H_cat = None
    if f_out is not None:
        if not (f_out.is_cuda and f_out.dtype == f.dtype and tuple(f_out.shape) == (B, N, D)
                and f_out.stride(2) == 1 and f_out.stride(0) == N * f_out.stride(1)):
            raise ValueError("f_out: a (B,N,D) last-dim slice of a contiguous GPU tensor of f's dtype")
        ex.f_out, ex.f_out_ld = f_out.data_ptr(), f_out.stride(1)
    if want_H_cat:
        H_cat = torch.empty((B, sum(h.shape[1] for h in Hs), N), dtype=f.dtype, device=f.device)
        ex.H_cat = H_cat.data_ptr()
    if counter is not None:
        if not (counter.is_cuda and counter.dtype == torch.int64 and counter.numel() == 1):
            raise ValueError("counter: a 1-element int64 GPU tensor")
        ex.counter, ex.counter_add = counter.data_ptr(), int(counter_add) & (2**64 - 1)
    if embed is not None:
        ex.x_raw, ex.x_dim, ex.M, ex.c, ex.f_contig = (embed[0].data_ptr(), embed[0].shape[2], embed[1].data_ptr(),
                                                      embed[2].data_ptr(), f_contig.data_ptr())
    with torch.cuda.device(f.device):
        check(_fn("gn_affinity_topk", f.dtype)(_ptr(f), _ptr(corr), Hl, kl, n, B, N, D, ctypes.byref(ex),
                                               stream_handle()), "gn_affinity_topk")
    if embed is not None:
        return corr, Hs, H_cat, f_contig
    return corr, Hs, H_cat


class AffinityTail:
    """The fused affinity + top-k launch of a forward, DEFERRED: outputs are allocated now, the work is issued as the tail
    workgroups of the first node-stage launch (`node_stage_grouped(..., affinity=job)` -> gn_node_mlp_affinity_*), or —
    when nothing picks it up — by `launch()` as the stand-alone launch.  Same arguments as `affinity_topk` (without the
    embedding front-end)."""

    def __init__(self, f: Tensor, scales: Sequence[int], want_corr: bool = False, f_out: Optional[Tensor] = None,
                 want_H_cat: bool = False, counter: Optional[Tensor] = None, counter_add: int = 0):
        _req(f, "f", (None, None, None), _ACT_DTYPES)
        self.f, self.scales = f, [int(s) for s in scales]
        B, N, D = f.shape
        self.corr = torch.empty((B, N, N), dtype=torch.float32, device=f.device) if want_corr else None
        self.Hs = _alloc_incidence(B, N, self.scales, f.device)
        self._Hl, self._kl, n = _scale_args(self.Hs, self.scales)
        self._ex = _lib.BlockExtras()
        self.H_cat = None
        if f_out is not None:
            if not (f_out.is_cuda and f_out.dtype == f.dtype and tuple(f_out.shape) == (B, N, D)
                    and f_out.stride(2) == 1 and f_out.stride(0) == N * f_out.stride(1)):
                raise ValueError("f_out: a (B,N,D) last-dim slice of a contiguous GPU tensor of f's dtype")
            self._ex.f_out, self._ex.f_out_ld = f_out.data_ptr(), f_out.stride(1)
        if want_H_cat:
            self.H_cat = torch.empty((B, sum(h.shape[1] for h in self.Hs), N), dtype=f.dtype, device=f.device)
            self._ex.H_cat = self.H_cat.data_ptr()
        if counter is not None:
            if not (counter.is_cuda and counter.dtype == torch.int64 and counter.numel() == 1):
                raise ValueError("counter: a 1-element int64 GPU tensor")
            self._ex.counter, self._ex.counter_add = counter.data_ptr(), int(counter_add) & (2**64 - 1)
        self._keep = (f_out, counter)
        self.job = _lib.AffinityJob(f.data_ptr(), 0 if self.corr is None else self.corr.data_ptr(), self._Hl, self._kl, n,
                                    B, N, D, ctypes.pointer(self._ex))
        self.done = False

    def fits_tail(self) -> bool:
        B, N, D = self.f.shape
        return N * (D + 4) * 4 + 8 + N * N * 8 <= load().gn_affinity_tail_lds_limit()

    def launch(self) -> None:
        """The stand-alone launch (nothing took the job along)."""
        if self.done:
            return
        B, N, D = self.f.shape
        with torch.cuda.device(self.f.device):
            check(_fn("gn_affinity_topk", self.f.dtype)(_ptr(self.f), _ptr(self.corr), self._Hl, self._kl, len(self.Hs), B, N, D,
                                                        ctypes.byref(self._ex), stream_handle()), "gn_affinity_topk")
        self.done = True


# ---- weight packing ------------------------------------------------------------------------------
def pack_linear(W: Tensor, col_offset: int = 0, in_features: Optional[int] = None) -> Tensor:
    """Packed image of an nn.Linear weight (out x in), or of the column block
    [col_offset, col_offset + in_features) of it."""
    _req(W, "W", (None, None))
    out_f, ld = W.shape
    in_f = ld - col_offset if in_features is None else in_features
    lib = load()
    Wp = torch.empty(lib.gn_packed_elems(out_f, in_f), dtype=W.dtype, device=W.device)
    with torch.cuda.device(W.device):
        check(lib.gn_pack_linear_f32(_ptr(W), _ptr(Wp), out_f, in_f, ld, col_offset, stream_handle()),
              "gn_pack_linear_f32")
    return Wp


# ---- A3 ------------------------------------------------------------------------------------------
def pack_stream(weights: Sequence[Tensor]) -> Tensor:
    """One weight stream: the packed images of `weights` (nn.Linear layout, out x in) back to back, in
    the order a kernel consumes them."""
    return torch.cat([pack_linear(w.detach().contiguous()) for w in weights])


def edge_stream(Wi0: Tensor, Wi1: Tensor, Wd0: Tensor, Wd1: Tensor) -> Tensor:
    """Weight stream of the edge-MLP kernel: the packed images of its four layers cut into hidden tiles
    (T, 8 steps) and second-layer slices (S) and laid out in the order the kernel consumes them —
    pair A: T0 T1 S0 T2 S1 T3 S2 S3 (S = both output tiles over one hidden tile, 8 steps);
    pair B: T0 T1 S0 T2 S1 ... T7 S6 S7 (S = 4 steps).  One step = 256 floats."""
    a0 = pack_linear(Wi0.detach().contiguous()).view(4, 8, 256)          # (128 x 64): 4 tiles x 8 steps
    a1 = pack_linear(Wi1.detach().contiguous()).view(2, 4, 4, 256)       # (64 x 128): (o, t) x 4 steps
    b0 = pack_linear(Wd0.detach().contiguous()).view(8, 8, 256)          # (256 x 64): 8 tiles x 8 steps
    b1 = pack_linear(Wd1.detach().contiguous()).view(1, 8, 4, 256)       # (32 x 256): (0, t) x 4 steps
    sa = lambda t: a1[:, t].reshape(8, 256)                               # slices (0,t), (1,t)
    sb = lambda t: b1[0, t]
    parts = [a0[0], a0[1], sa(0), a0[2], sa(1), a0[3], sa(2), sa(3)]
    parts += [b0[0], b0[1]]
    for t in range(8):
        parts.append(sb(t))
        if t < 6:
            parts.append(b0[t + 2])
    parts.append(a0.new_zeros(8, 256))      # the kernel's ring reads 8 steps ahead of the last one it uses
    return torch.cat([p.reshape(-1) for p in parts]).contiguous()


def bias_stream(biases: Sequence[Tensor]) -> Tensor:
    """Biases back to back, each zero-padded to a multiple of 32 (one 32-float tile per output tile)."""
    parts = []
    for b in biases:
        b = b.detach().reshape(-1)
        pad = (-b.numel()) % 32
        parts.append(torch.cat((b, b.new_zeros(pad))) if pad else b)
    return torch.cat(parts).contiguous()


class PackPlan:
    """All packed weight images of one module as ONE arena refreshed by ONE launch.

    Built once (per module and parameter addresses): `matrix` / `block` / `vector` record segments of
    `gn_pack_segments_f32` and hand out arena offsets; `finish()` uploads the segment table to the device.
    `refresh()` = one kernel launch, reading the parameters in place — what has to happen after every
    optimizer step, capturable in a hipGraph."""

    TILE = 1024

    def __init__(self, device: torch.device):
        self.device = device
        self.size = 0
        self._segs: List[tuple] = []
        self._keep: List[Tensor] = []
        self.arena: Optional[Tensor] = None
        self.table: Optional[Tensor] = None
        self.max_elems = 1

    def alloc(self, numel: int) -> int:
        off = self.size
        self.size += (numel + 63) // 64 * 64          # keeps every image 256-byte aligned
        return off

    def block(self, dst_off: int, W: Tensor, IT: int, r0=0, c0=0, rows=None, cols=None, place_r=0, place_c=0,
              scale=1.0) -> None:
        """W[r0:r0+rows, c0:c0+cols] -> the packed image at arena offset dst_off (IT tiles per packed row),
        at (place_r, place_c) of its virtual matrix."""
        W = W.detach()
        if W.dim() != 2 or W.stride(1) != 1 or W.dtype != torch.float32 or W.device != self.device:
            raise ValueError("PackPlan.block: 2-D fp32 row-major matrix on the plan's device")
        rows = W.shape[0] - r0 if rows is None else rows
        cols = W.shape[1] - c0 if cols is None else cols
        self._keep.append(W)
        self._segs.append((W.data_ptr() + 4 * (r0 * W.stride(0) + c0), dst_off, W.stride(0), rows, cols, place_r, place_c,
                           IT, scale, 0))
        self.max_elems = max(self.max_elems, rows * cols)

    def place(self, dst_off: int, dst_ld: int, W: Tensor, place_r=0, place_c=0, scale=1.0) -> None:
        """W (2-D, or a vector as one row) -> rows [place_r, ...) x columns [place_c, ...) of the plain row-major
        (.., dst_ld) matrix at arena offset dst_off: concatenations without torch.cat."""
        W = W.detach()
        W = W.reshape(1, -1) if W.dim() == 1 else W
        if W.dim() != 2 or W.stride(1) != 1 or W.dtype != torch.float32 or W.device != self.device:
            raise ValueError("PackPlan.place: fp32 row-major matrix or vector on the plan's device")
        self._keep.append(W)
        self._segs.append((W.data_ptr(), dst_off, W.stride(0), W.shape[0], W.shape[1], place_r, place_c, 0, scale, dst_ld))
        self.max_elems = max(self.max_elems, W.numel())

    def matrix(self, W: Tensor) -> int:
        """The whole (out x in) weight as a standard packed image; returns its arena offset."""
        OT, IT = (W.shape[0] + 31) // 32, (W.shape[1] + 31) // 32
        off = self.alloc(OT * IT * self.TILE)
        self.block(off, W, IT)
        return off

    def vector(self, dst_off: int, v: Tensor, place=0, scale=1.0) -> None:
        v = v.detach().reshape(1, -1)
        if v.stride(1) != 1 or v.dtype != torch.float32 or v.device != self.device:
            raise ValueError("PackPlan.vector: contiguous fp32 vector on the plan's device")
        self._keep.append(v)
        self._segs.append((v.data_ptr(), dst_off, v.shape[1], 1, v.shape[1], 0, place, 0, scale, 0))
        self.max_elems = max(self.max_elems, v.shape[1])

    def finish(self) -> "PackPlan":
        self.arena = torch.zeros(max(self.size, 64), dtype=torch.float32, device=self.device)
        base = self.arena.data_ptr()
        arr = (_lib.PackSeg * len(self._segs))()
        for i, (src, off, ld, rows, cols, pr, pc, IT, scale, dst_ld) in enumerate(self._segs):
            arr[i] = _lib.PackSeg(src, base + 4 * off, ld, rows, cols, pr, pc, IT, float(scale), dst_ld)
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self.table = raw.to(self.device)
        self.sources = tuple(t.data_ptr() for t in self._keep)
        return self

    def view(self, off: int, numel: int) -> Tensor:
        return self.arena[off:off + numel]

    def refresh(self) -> None:
        # (segments always write the same positions: the padding zeroed in finish() stays zero)
        if _REPACK["rec"] is not None:
            _REPACK["rec"][0].append(self)
        elif _REPACK["done_plans"] is not None and id(self) in _REPACK["done_plans"]:
            return      # this step's RepackBatch refreshed it already
        with torch.cuda.device(self.device):
            check(load().gn_pack_segments_f32(_ptr(self.table), len(self._segs), self.max_elems, stream_handle()),
                  "gn_pack_segments_f32")


# ---- one re-pack per training step -----------------------------------------------------------------------------
# A training step re-derives every packed weight image from the parameters (the optimizer just rewrote them): one
# `refresh` launch per pack plan and one `split_bf16` launch per bf16-core image — 21 + 17 launches of ~4.6 us per step
# of the multiscale block, 9 % of the graphed step.  All of them read only the parameters, so they can run first and
# together: `repack_scope` RECORDS which plans / images one step touches (first use), then runs them as TWO launches —
# `gn_pack_segments_f32` over the concatenated segment tables, `gn_split_bf16_batch_f32` over all images — at the head
# of every later step and turns the recorded per-plan / per-image launches of that step into no-ops.  A plan or image
# that is not in the batch (rebuilt because parameter storage moved) simply takes its own launch as before.
_REPACK = {"rec": None, "done_plans": None, "done_splits": None}


class RepackBatch:
    def __init__(self, plans: Sequence["PackPlan"], splits: Sequence[Tuple[Tensor, Tensor, int]]):
        self.plans, self.splits = list(plans), list(splits)          # (keeps arenas, tables and images alive)
        dev = self.plans[0].device
        self.device = dev
        self.table = torch.cat([p.table for p in self.plans])
        self.n_segs = sum(len(p._segs) for p in self.plans)
        self.max_elems = max(p.max_elems for p in self.plans)
        if self.n_segs > 65535:
            raise ValueError("RepackBatch: too many segments for one launch")
        self.parts = sorted({pt for _, _, pt in self.splits})
        self.jobs = {}
        for pt in self.parts:
            js = [(a, b) for a, b, q in self.splits if q == pt]
            arr = (_lib.SplitJob * len(js))()
            for i, (a, b) in enumerate(js):
                arr[i] = _lib.SplitJob(a.data_ptr(), b.data_ptr(), a.numel() // 1024, 0)
            raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            self.jobs[pt] = (raw, len(js), max(a.numel() // 1024 for a, _ in js))
        self.plan_ids = {id(p) for p in self.plans}
        self.split_keys = {(a.data_ptr(), b.data_ptr(), q) for a, b, q in self.splits}

    def run(self) -> None:
        with torch.cuda.device(self.device):
            check(load().gn_pack_segments_f32(_ptr(self.table), self.n_segs, self.max_elems, stream_handle()),
                  "gn_pack_segments_f32")
            for pt, (raw, n, mx) in self.jobs.items():
                check(load().gn_split_bf16_batch_f32(_ptr(raw), n, mx, pt, stream_handle()), "gn_split_bf16_batch_f32")


class repack_scope:
    """``with repack_scope(holder):`` around ONE whole training step (forward and backward).  `holder` is a dict owned by
    the caller; its first use records, later uses replay the batch (see above).  Not re-entrant."""

    def __init__(self, holder: dict):
        self.h = holder

    def __enter__(self):
        if _REPACK["rec"] is not None or _REPACK["done_plans"] is not None:
            raise RuntimeError("repack_scope is not re-entrant")
        batch = self.h.get("batch")
        if batch is None:
            _REPACK["rec"] = ([], [])
        else:
            batch.run()
            _REPACK["done_plans"], _REPACK["done_splits"] = batch.plan_ids, batch.split_keys
        return self

    def __exit__(self, et, ev, tb):
        rec = _REPACK["rec"]
        _REPACK["rec"] = _REPACK["done_plans"] = _REPACK["done_splits"] = None
        if rec is not None and et is None:
            plans, seen = [], set()
            for p in rec[0]:
                if id(p) not in seen:
                    seen.add(id(p))
                    plans.append(p)
            splits, seen2 = [], set()
            for a, b, q in rec[1]:
                k = (a.data_ptr(), b.data_ptr(), q)
                if k not in seen2:
                    seen2.add(k)
                    splits.append((a, b, q))
            self.h["batch"] = RepackBatch(plans, splits) if plans else None
        return False


def pipeline_order(HT: int) -> List[Tuple[str, int]]:
    """Order in which the bf16-core kernels consume the tiles of one layer pair with HT hidden tiles
    (gn_mlp_bf16.hpp, layer_pair): A_t = the first-layer tiles producing hidden tile t, B_t = the second-layer
    tiles consuming it;  A0 A1 B0 A2 B1 ... A(HT-1) B(HT-2) B(HT-1)."""
    out = [("A", 0)]
    for t in range(HT):
        if t + 1 < HT:
            out.append(("A", t + 1))
        out.append(("B", t))
    return out


def _groups(n: int) -> None:
    if not 1 <= n <= _lib.MAX_GROUPS:
        raise ValueError(f"1..{_lib.MAX_GROUPS} groups per launch, got {n}")


def node_stage_grouped(items: Sequence[Tuple[Tensor, dict]], keep: Optional[List[dict]] = None,
                       a_specs: Optional[Sequence[Optional[Tuple[dict, int]]]] = None,
                       affinity: Optional["AffinityTail"] = None
                       ) -> Tuple[List[Tuple[Tensor, Tensor]], List[Optional[Tensor]]]:
    """One launch for the node rows of several modules: items = [(x (B,N,64), pk{"W","bias","xi"})] with equal
    shapes -> [(x', pq)].  ``keep`` (training) receives per group {"hid": relu(W0 x + b0) (rows, 256)}.
    ``a_specs[g]`` = (agg_pk, K) asks for the per-node first layer of the typed aggregation MLP of the pairwise
    graph, A = W1cat x + b1/2 (B,N,K*128), from the SAME launch (bf16-core path of the fp32 entry point; with
    that path switched off it is a `node_linear` launch of its own); returned as the second list."""
    _groups(len(items))
    x0 = _req(items[0][0], "x", (None, None, FEAT), _ACT_DTYPES)
    dt = x0.dtype
    rows = x0.shape[0] * x0.shape[1]
    arr = (_lib.NodeGroup * len(items))()
    outs, As, late = [], [], []
    for g, (x, pk) in enumerate(items):
        _req(x, "x", tuple(x0.shape), dt)
        _same_device(x0, x)
        xp, pq = torch.empty_like(x), torch.empty_like(x)
        hid_ptr = 0
        if keep is not None:
            if _twin(dt):
                raise ValueError("the bf16 twins are forward-only")
            keep.append(dict(hid=torch.empty((rows, 256), dtype=x.dtype, device=x.device)))
            hid_ptr = keep[-1]["hid"].data_ptr()
        wx = _ximg(pk, "chain", dt)
        a_fields = (0, 0, 0, 0)
        A = None
        wah = 0
        spec = a_specs[g] if a_specs is not None else None
        if spec is not None:
            if _twin(dt):
                raise ValueError("the per-node first layer (pair form) belongs to the fp32 entry points")
            apk, K = spec
            if wx:
                A = torch.empty(tuple(x.shape[:-1]) + (K * 128,), dtype=x.dtype, device=x.device)
                a_fields = (_ximg(apk, "W1cat", dt), apk["b1half"].data_ptr(), A.data_ptr(), K)
                wah = _himg(apk, "W1cat", dt)
            else:
                late.append((g, x, apk, K))
        As.append(A)
        arr[g] = _lib.NodeGroup(x.data_ptr(), pk["W"].data_ptr(), pk["bias"].data_ptr(), xp.data_ptr(), pq.data_ptr(),
                                hid_ptr, wx, *a_fields, _himg(pk, "chain", dt) if wx else 0, wah)
        outs.append((xp, pq))
    flops = sum(rows * 2 * (64 * 256 + 256 * 64 + 64 * 64 + (64 * 128 * int(a.KA) if a.A else 0)) for a in arr)
    # ``affinity``: a deferred fused affinity + top-k launch rides as this launch's tail workgroups (bf16-/fp16-core
    # kernels, a scene tile within the tail's LDS limit); otherwise it is issued on its own right here
    ride = (affinity is not None and not affinity.done and all(bool(a.Wx) for a in arr) and affinity.fits_tail()
            and affinity.f.dtype == dt and affinity.f.device == x0.device and _AFFINITY_TAIL)
    if affinity is not None and not ride:
        affinity.launch()
    with torch.cuda.device(x0.device), _Probed("node_stage_kernel", flops):
        if ride:
            check(_fn("gn_node_mlp_affinity", dt)(arr, len(items), rows, ctypes.byref(affinity.job), stream_handle()),
                  "gn_node_mlp_affinity")
            affinity.done = True
        else:
            check(_fn("gn_node_mlp", dt)(arr, len(items), rows, stream_handle()), "gn_node_mlp")
    for g, x, apk, K in late:
        As[g] = node_linear(x, apk["W1cat"], apk["b1half"], K * 128)
    return outs, As


def node_mlp_grouped(items: Sequence[Tuple[Tensor, dict]], keep: Optional[List[dict]] = None
                     ) -> List[Tuple[Tensor, Tensor]]:
    return node_stage_grouped(items, keep)[0]


def node_mlp(x: Tensor, pk: dict) -> Tuple[Tensor, Tensor]:
    """x (B,N,64) -> x' = MLP_{64->256->64}(x), pq = x' Wpq^T + bpq.  pk: {"W": stream, "bias": stream}."""
    return node_mlp_grouped([(x, pk)])[0]


def pair_count(N: int) -> int:
    """Unordered pairs (i <= j) of N nodes — rows per scene of the symmetric pairwise form."""
    return N * (N + 1) // 2


def node2edge_grouped(items: Sequence[tuple]) -> List[Tensor]:
    """items = [(xp, pq, H or None, w2 (32,), b2 (1,)[, sym])] over the same (B, N); returns [edges (B,E,64)].
    w2 / b2 = weight row and bias of attention layer 1, device tensors (the parameters themselves).
    H=None selects the implicit pairwise graph: E = N*N ordered edges, or with sym=True the
    N(N+1)/2 unordered pairs (edges (i,j) and (j,i) carry the same feature)."""
    _groups(len(items))
    xp0 = _req(items[0][0], "xp", (None, None, FEAT), _ACT_DTYPES)
    dt = xp0.dtype
    B, N, _ = xp0.shape
    arr = (_lib.N2EGroup * len(items))()
    outs = []
    for g, item in enumerate(items):
        xp, pq, H, w2, b2 = item[:5]
        sym = bool(item[5]) if len(item) > 5 else False
        _req(xp, "xp", (B, N, FEAT), dt)
        _req(pq, "pq", (B, N, FEAT), dt)
        if H is None:
            E = pair_count(N) if sym else N * N
        else:
            if sym:
                raise ValueError("sym applies to the pairwise graph (H=None) only")
            _req(H, "H", (B, None, N))
            E = H.shape[1]
        _req(w2, "w2", (32,))
        _req(b2, "b2", (1,))
        _same_device(xp0, xp, pq, H, w2, b2)
        edges = torch.empty((B, E, FEAT), dtype=xp.dtype, device=xp.device)
        arr[g] = _lib.N2EGroup(xp.data_ptr(), pq.data_ptr(), 0 if H is None else H.data_ptr(), w2.data_ptr(),
                               edges.data_ptr(), b2.data_ptr(), E, int(sym))
        outs.append(edges)
    with torch.cuda.device(xp0.device):
        check(_fn("gn_node2edge", dt)(arr, len(items), B, N, stream_handle()), "gn_node2edge")
    return outs


def node2edge(xp: Tensor, pq: Tensor, H: Optional[Tensor], w2: Tensor, b2: Tensor, sym: bool = False) -> Tensor:
    return node2edge_grouped([(xp, pq, H, w2, b2, sym)])[0]


# ---- A4 ------------------------------------------------------------------------------------------
class PhiloxNoise:
    """Uniforms generated inside the edge kernel: element (row, k) of a (B,E,K) draw is element
    offset (+ device counter) + row*K + k of the Philox4x32-10 stream `seed`."""
    __slots__ = ("seed", "offset", "counter")

    def __init__(self, seed: int, offset: int = 0, counter: Optional[Tensor] = None):
        if counter is not None and not (counter.is_cuda and counter.dtype == torch.int64 and counter.numel() == 1):
            raise ValueError("counter: a 1-element int64 GPU tensor")
        self.seed, self.offset, self.counter = int(seed) & (2**64 - 1), int(offset), counter


class PoolSpec:
    """Rows of the edge MLP to be formed inside its kernel instead of read from an `edges` tensor: the
    attention-weighted node -> edge pooling of `node2edge` (xp, pq (B,N,64) from the node stage; H (B,E,N) or None =
    the implicit pairwise graph, sym -> unordered pairs; w2 (32,), b2 (1,)).  bf16-core kernels only; H needs N <= 16
    (run_message_passing uses the hyper form only up to POOL_MAX_N)."""
    __slots__ = ("xp", "pq", "H", "w2", "b2", "sym")

    def __init__(self, xp: Tensor, pq: Tensor, H: Optional[Tensor], w2: Tensor, b2: Tensor, sym: bool = False):
        self.xp, self.pq, self.H, self.w2, self.b2, self.sym = xp, pq, H, w2, b2, bool(sym)


# Hyper modules: largest N whose pooling the edge kernel may form itself (the kernel's bound is 16: a row's incidence
# stays in registers).  Default 0 = pairwise graph only: without an LDS stage the hyper form pays one L2 latency per
# member and pass — measured at B=512, N=11: edge kernel +15.7 us against the ~10 us the hyper groups cost in the
# node2edge launch; the pairwise form costs +3 us and removes 5.5 us (and, at N=50 / B=1024, 2 x 167 MB of traffic).
POOL_MAX_N = int(os.environ.get("GN_POOL_MAX_N", "0"))
POOL_KERNEL_MAX_N = 16


def edge_mlp_gumbel_grouped(items: Sequence[tuple], tau: float = 0.5, keep: Optional[List[dict]] = None
                            ) -> List[Tuple[Tensor, Optional[Tensor]]]:
    """items = [(edges (B,E,64) or PoolSpec, U tensor (B,E,K) or PhiloxNoise, pk, K[, sym_N[, want_dist]])];
    returns [(edge_feat, dist)].  All PhiloxNoise entries of one call must share seed and counter.

    sym_N = N > 0: `edges` holds the (B, N(N+1)/2, 64) unordered-pair rows of the pairwise graph; U /
    the Philox positions and `dist` are those of the ORDERED (B, N*N, K) tensor; edge_feat is
    (B, N(N+1)/2, K) = fac * (dist_ij + dist_ji).  want_dist=False skips the ordered dist output.
    ``keep`` (training) receives per group the activations the kernel otherwise holds in registers:
    {"z1" (rows,128), "z" (rows,64), "dh1" (rows,256), "lgf" (rows,32)}."""
    _groups(len(items))
    first = items[0][0]
    e0 = _req(first.xp if isinstance(first, PoolSpec) else first, "edges", (None, None, FEAT), _ACT_DTYPES)
    dt = e0.dtype
    arr = (_lib.EdgeGroup * len(items))()
    outs = []
    seed, ctr = None, None
    for g, item in enumerate(items):
        edges, U, pk, K = item[:4]
        sym_N = int(item[4]) if len(item) > 4 else 0
        want_dist = bool(item[5]) if len(item) > 5 else True
        pool = edges if isinstance(edges, PoolSpec) else None
        if pool is not None:
            if keep is not None or not BF16X6 and not _twin(dt):
                raise ValueError("PoolSpec: forward-only, bf16-core kernels only")
            _req(pool.xp, "xp", (None, None, FEAT), dt)
            B, Np, _ = pool.xp.shape
            _req(pool.pq, "pq", (B, Np, FEAT), dt)
            _req(pool.w2, "w2", (32,))
            _req(pool.b2, "b2", (1,))
            if pool.H is None:
                if bool(sym_N) != pool.sym or (sym_N and sym_N != Np):
                    raise ValueError("PoolSpec: sym must agree with sym_N = N")
                E = pair_count(Np) if pool.sym else Np * Np
            else:
                _req(pool.H, "H", (B, None, Np))
                if sym_N or Np > POOL_KERNEL_MAX_N:
                    raise ValueError(f"PoolSpec with H: a hyper module with N <= {POOL_KERNEL_MAX_N}")
                E = pool.H.shape[1]
            _same_device(e0, pool.xp, pool.pq, pool.H, pool.w2, pool.b2)
            edges = pool.xp              # (device / dtype of the outputs)
        else:
            _req(edges, "edges", (None, None, FEAT), dt)
            _same_device(e0, edges)
            B, E, _ = edges.shape
        if sym_N and E != pair_count(sym_N):
            raise ValueError(f"edges: symmetric form needs {pair_count(sym_N)} pair rows per scene, got {E}")
        Eo = sym_N * sym_N if sym_N else E          # ordered edges per scene (noise / dist layout)
        if isinstance(U, PhiloxNoise):
            if seed is None:
                seed, ctr = U.seed, U.counter
            elif (seed, ctr) != (U.seed, U.counter) and not (seed == U.seed and ctr is U.counter):
                raise ValueError("all PhiloxNoise groups of one launch must share seed and counter")
            u_ptr, off = 0, U.offset
        else:
            _req(U, "noise_u", (B, Eo, K))
            _same_device(edges, U)
            u_ptr, off = U.data_ptr(), 0
        edge_feat = torch.empty((B, E, K), dtype=torch.float32, device=edges.device)    # always fp32: a VALU scale
        dist = torch.empty((B, Eo, K), dtype=dt, device=edges.device) if (want_dist or not sym_N) else None
        kp = (0, 0, 0, 0)
        if keep is not None:
            if _twin(dt):
                raise ValueError("the bf16 twins are forward-only")
            mk = lambda w: torch.empty((B * E, w), dtype=edges.dtype, device=edges.device)
            keep.append(dict(z1=mk(128), z=mk(64), dh1=mk(256), lgf=mk(32)))
            kp = tuple(keep[-1][n].data_ptr() for n in ("z1", "z", "dh1", "lgf"))
        pool_args = (0, 0, 0, 0, 0, 0, 0) if pool is None else (
            pool.xp.data_ptr(), pool.pq.data_ptr(), 0 if pool.H is None else pool.H.data_ptr(), pool.w2.data_ptr(),
            pool.b2.data_ptr(), pool.xp.shape[1], 0 if pool.H is None else pool.H.shape[1])
        arr[g] = _lib.EdgeGroup(0 if pool is not None else edges.data_ptr(), u_ptr, pk["W"].data_ptr(),
                                pk["bias"].data_ptr(), edge_feat.data_ptr(), 0 if dist is None else dist.data_ptr(), off,
                                B * E, K, sym_N, *kp, _ximg(pk, "edge", dt), *pool_args, _himg(pk, "edge", dt))
        outs.append((edge_feat, dist))
    flops = sum(int(a.rows) for a in arr) * 2 * (64 * 128 + 128 * 64 + 64 * 256 + 256 * 32)
    with torch.cuda.device(e0.device), _Probed("edge_mlp_gumbel_kernel", flops):
        check(_fn("gn_edge_mlp_gumbel", dt)(arr, len(items), float(tau), seed or 0, _ptr(ctr), stream_handle()),
              "gn_edge_mlp_gumbel")
    return outs


def edge_mlp_gumbel(edges: Tensor, U, pk: dict, K: int, tau: float = 0.5) -> Tuple[Tensor, Tensor]:
    """(edge_feat, dist) of MLP_dict_softmax.  `U`: a (B,E,K) tensor of uniforms, or a PhiloxNoise."""
    return edge_mlp_gumbel_grouped([(edges, U, pk, K)], tau)[0]


# ---- A5 ------------------------------------------------------------------------------------------
def _edge_count(H: Optional[Tensor], B: int, N: int, sym: bool = False) -> int:
    if H is None:
        return pair_count(N) if sym else N * N
    if sym:
        raise ValueError("sym applies to the pairwise graph (H=None) only")
    _req(H, "H", (B, None, N))
    return H.shape[1]


def agg_gather_grouped(items: Sequence[tuple]) -> List[Tensor]:
    """items = [(ori (B,N,64), H (B,E,N) or None[, sym])] -> [eo (B,E,64)]."""
    _groups(len(items))
    o0 = _req(items[0][0], "ori", (None, None, FEAT), _ACT_DTYPES)
    dt = o0.dtype
    B, N, _ = o0.shape
    arr = (_lib.GatherGroup * len(items))()
    outs = []
    for g, item in enumerate(items):
        ori, H = item[:2]
        sym = bool(item[2]) if len(item) > 2 else False
        _req(ori, "ori", (B, N, FEAT), dt)
        E = _edge_count(H, B, N, sym)
        _same_device(o0, ori, H)
        eo = torch.empty((B, E, FEAT), dtype=ori.dtype, device=ori.device)
        arr[g] = _lib.GatherGroup(ori.data_ptr(), 0 if H is None else H.data_ptr(), eo.data_ptr(), E, int(sym))
        outs.append(eo)
    with torch.cuda.device(o0.device):
        check(_fn("gn_agg_gather", dt)(arr, len(items), B, N, stream_handle()), "gn_agg_gather")
    return outs


def agg_gather(ori: Tensor, H: Optional[Tensor], sym: bool = False) -> Tensor:
    return agg_gather_grouped([(ori, H, sym)])[0]


class GatherSpec:
    """Input rows of the typed MLP to be formed inside the kernel instead of read from an `eo` tensor:
    eo = H @ ori (H (B,E,N)), or the pairwise rows ori_i + ori_j (H=None; sym -> unordered pairs).
    ``node=True`` (bf16 twins, pairwise graph, unordered pairs, N <= SCENE_FORM_MAX_N): the NODE form with one scene per
    workgroup — both layers run once per node (layer 1 is linear in the two nodes, layer 2 and the type weighting commute
    with H^T, see PairSpec); the result is H^T feat (B,N,64) for a ``NodeAggSpec``."""
    __slots__ = ("ori", "H", "sym", "node")

    def __init__(self, ori: Tensor, H: Optional[Tensor], sym: bool = False, node: bool = False):
        self.ori, self.H, self.sym, self.node = ori, H, bool(sym), bool(node)


SCENE_FORM_MAX_N = 64


class PairSpec:
    """Pair form of the typed MLP for the pairwise graph: A (B,N,K*128) = node_linear(ori) holds the first
    layer per node; rows are the N(N+1)/2 unordered pairs.  Uses pk["W2t"] instead of pk["W"].
    ``node=True`` (N <= NODE_FORM_MAX_N, bf16-core images): the NODE form — the per-pair feature is consumed only as
    H^T feat (model/MS_HGNN_batch.py:267) and type weighting + layer 2 are linear, so layer 2 runs once per node on
    S[n,k] = sum_j ef[p(n,j),k] relu(A[n,k] + A[j,k]); the result is H^T feat (B,N,64), to be fed to the closing MLP
    through a ``NodeAggSpec``."""
    __slots__ = ("A", "node")

    def __init__(self, A: Tensor, node: bool = False):
        self.A, self.node = A, bool(node)


NODE_FORM_MAX_N = 16
NODE_FORM_MAX_K = 12


def node_form_enabled() -> bool:
    """GN_NODE_FORM=0 keeps the per-pair form (A/B switch, read per call)."""
    return os.environ.get("GN_NODE_FORM", "1") != "0"


def split_bf16(packed: Tensor, out: Optional[Tensor] = None, parts: int = 3) -> Tensor:
    """bf16-core image (16-bit words, as int16) of packed fp32 32x32 weight tiles: `gn_split_bf16_f32`
    (parts = 3: x = p1 + p2 + p3, the fp32-accurate path; parts = 1: x rounded to bf16, the twins; parts = 2: two fp16
    parts x = hi + lo — the f16x3 path — followed by the 16-byte range flag)."""
    _req(packed, "packed")
    n_tiles = packed.numel() // 1024
    if out is None:
        n = n_tiles * 2 * parts * 64 * 8
        out = (torch.zeros(n + 8, dtype=torch.int16, device=packed.device) if parts == 2 else     # (+ the flag word)
               torch.empty(n, dtype=torch.int16, device=packed.device))
    if _REPACK["rec"] is not None:
        _REPACK["rec"][1].append((packed, out, int(parts)))
    elif _REPACK["done_splits"] is not None and (packed.data_ptr(), out.data_ptr(), int(parts)) in _REPACK["done_splits"]:
        return out      # this step's RepackBatch built it already
    with torch.cuda.device(packed.device):
        check(load().gn_split_bf16_f32(_ptr(packed), ctypes.c_void_p(out.data_ptr()), n_tiles, int(parts),
                                       stream_handle()), "gn_split_bf16_f32")
    return out


def closing_fusable(items: Sequence[Tuple[object, Tensor, dict, int]], pks2: Sequence[dict]) -> bool:
    """Can gn_agg_mlp_f32 apply the closing MLPs itself (gn_agg_group_t.y)?  Mirrors the launcher's rules: fp32
    results on the 16-bit matrix cores, N <= 16, every group the node form of the pairwise graph or a hyper module with
    the fused gather whose edge rows run more than one wave per row block (< 768 row blocks), a 128 -> 128 -> dout <= 64 MLP."""
    if not (BF16X6 and closing_fusion_enabled()):
        return False
    if os.environ.get("GN_AGG_HSTAGE", "1") == "0" or os.environ.get("GN_AGG_LINES", "1") == "0":
        return False      # (diagnostic gathers: the fused stage reads its scenes' rows from the LDS stage)
    for (src, ef, pk, K), pk2 in zip(items, pks2):
        if (pk2["din"], pk2["dh"]) != (2 * FEAT, 128) or not (32 < pk2["dout"] <= 64) or _ximg(pk2, "mlp2", torch.float32) == 0:
            return False
        if isinstance(src, PairSpec):
            if not src.node or src.A.dtype != torch.float32:
                return False
        elif isinstance(src, GatherSpec):
            if src.H is None or src.ori.dtype != torch.float32:
                return False
            B, E, N = src.H.shape
            if N > 16 or E > 16 or (B * E + 31) // 32 >= 768 or (B * E + 31) // 32 < 1:
                return False
            if (B * E + 31) // 32 < 128 and K < 4:      # (the launcher would run one wave per row block)
                return False
        else:
            return False
    return True


def closing_fusion_enabled() -> bool:
    """GN_FUSE_CLOSING=0: the closing MLP as a launch of its own (read per call)."""
    return os.environ.get("GN_FUSE_CLOSING", "1") != "0"


def agg_mlp_grouped(items: Sequence[Tuple[object, Tensor, dict, int]],
                    closing: Optional[Sequence[Tuple[dict, Optional[Tensor], Tensor]]] = None) -> List[Tensor]:
    """items = [(eo (B,E,64) | GatherSpec | PairSpec, edge_feat (B,E,K) fp32, pk{"W","b1","b2","xi"[,"W2t"]}, K)]
    -> [feat (B,E,64)] in the storage type of the inputs (bf16 twin: eo / GatherSpec only).
    ``closing`` (see `closing_fusable`): per item (pk of the closing MLP, out or None, ori (B,N,64)) — the launch then also
    applies y = MLP(cat(H^T feat, ori) / N) and returns [y (B,N,dout)] instead of the features."""
    _groups(len(items))
    arr = (_lib.AggGroup * len(items))()
    outs = []
    flops = ref_flops = flops2 = 0
    dev0 = dt = None
    for g, (eo, edge_feat, pk, K) in enumerate(items):
        wkey = "W"
        if isinstance(eo, PairSpec):
            A = eo.A
            _req(A, "A", (None, None, K * 128))
            B, N = A.shape[0], A.shape[1]
            E = pair_count(N)
            _same_device(A, edge_feat)
            like, eo_ptr, wkey = A, 0, "W2t"
            extra = (0, 0, E, N, 1, A.data_ptr(), _ximg(pk, "W2t", A.dtype), 0, _himg(pk, "W2t", A.dtype), 0,
                     1 if eo.node else 0)
            if eo.node and (N > NODE_FORM_MAX_N or K > NODE_FORM_MAX_K or not extra[6]):
                raise ValueError("PairSpec(node=True): needs N <= 16, K <= 12 and the bf16-core weight images")
        elif isinstance(eo, GatherSpec):
            ori, H = eo.ori, eo.H
            _req(ori, "ori", (None, None, FEAT), _ACT_DTYPES)
            B, N, _ = ori.shape
            E = _edge_count(H, B, N, eo.sym)
            _same_device(ori, H, edge_feat)
            like, eo_ptr = ori, 0
            if eo.node and not (_twin(ori.dtype) and H is None and eo.sym and N <= SCENE_FORM_MAX_N):
                raise ValueError("GatherSpec(node=True): bf16 storage, the pairwise graph's unordered pairs and N <= 64")
            extra = (ori.data_ptr(), 0 if H is None else H.data_ptr(), E, N, int(eo.sym), 0, 0,
                     _ximg(pk, "W12", ori.dtype), 0, _himg(pk, "W12", ori.dtype), 1 if eo.node else 0)
        else:
            _req(eo, "eo", (None, None, FEAT), _ACT_DTYPES)
            B, E, _ = eo.shape
            _same_device(eo, edge_feat)
            like, eo_ptr = eo, eo.data_ptr()
            extra = (0, 0, 0, 0, 0, 0, 0, _ximg(pk, "W12", eo.dtype), 0, _himg(pk, "W12", eo.dtype))
        dev0, dt = dev0 or like.device, dt or like.dtype
        if like.device != dev0 or like.dtype != dt:
            raise ValueError("grouped launch: every group must be on the same device and of the same storage type")
        _req(edge_feat, "edge_feat", (B, E, K))
        node = isinstance(eo, (PairSpec, GatherSpec)) and eo.node
        if closing is not None:
            pk2, out2, ori2 = closing[g]
            _req(ori2, "ori", (B, N, FEAT))
            y, ldy = _mlp2_out((B, N), pk2["dout"], out2, ori2)
            if len(extra) == 10:
                extra = extra + (0,)
            # (ori: the pairwise group's extra[0] is unused by its node form — the fused stage reads it from there)
            extra = (ori2.data_ptr(),) + tuple(extra[1:]) + (_ximg(pk2, "mlp2", like.dtype), _himg(pk2, "mlp2", like.dtype),
                                                             pk2["bias"].data_ptr(), y.data_ptr(), ldy, pk2["dout"], float(N))
            arr[g] = _lib.AggGroup(eo_ptr, edge_feat.data_ptr(), pk[wkey].data_ptr(), pk["b1"].data_ptr(),
                                   pk["b2"].data_ptr(), 0, B * E, K, *extra)
            outs.append(y)
            flops2 += B * N * 2 * (128 * 128 + 128 * (((pk2["dout"] + 31) // 32) * 32))
        else:
            feat = torch.empty((B, N if node else E, FEAT), dtype=like.dtype, device=like.device)
            arr[g] = _lib.AggGroup(eo_ptr, edge_feat.data_ptr(), pk[wkey].data_ptr(), pk["b1"].data_ptr(),
                                   pk["b2"].data_ptr(), feat.data_ptr(), B * E, K, *extra)
            outs.append(feat)
        # executed FLOPs: both layers, or the second layer only in the pair form; node form: layer 2 per node plus the
        # 3 flops (add, max, fma) per pair-member and hidden value that form S
        if node:
            flops += (B * N * K * ((2 * 128 * 64 + 2 * 64) + (0 if wkey == "W2t" else 2 * 64 * 128))
                      + B * N * N * K * 128 * 3)
        else:
            flops += B * E * K * ((2 * 128 * 64 + 2 * 64) + (0 if wkey == "W2t" else 2 * 64 * 128))
        # the reference's form of the stage (model/MS_HGNN_batch.py:262-265): every edge row — all N*N ordered edges of
        # the pairwise graph — through both layers of every type
        pairwise = isinstance(eo, PairSpec) or (isinstance(eo, GatherSpec) and eo.H is None)
        ref_flops += B * (N * N if pairwise else E) * K * (2 * 64 * 128 + 2 * 128 * 64 + 2 * 64)
    flops = (flops + flops2, ref_flops + flops2)
    # (the scene-form groups of the twins run in their own kernel ahead of the others' launch, on the same stream: forked
    # onto a side stream beside it they were measured at config 4 — single-stream 0.791 -> 0.784 ms, but 4-stream
    # throughput 1.445 -> 1.338 M scenes/s — and the fork was not kept)
    with torch.cuda.device(dev0), _Probed("agg_mlp_kernel", flops):
        check(_fn("gn_agg_mlp", dt)(arr, len(items), stream_handle()), "gn_agg_mlp")
    return outs


def agg_mlp(eo: Tensor, edge_feat: Tensor, pk: dict, K: int) -> Tensor:
    return agg_mlp_grouped([(eo, edge_feat, pk, K)])[0]


def node_linear(x: Tensor, W: Tensor, bias: Tensor, dout: int) -> Tensor:
    """y = W x + bias for x (..., 64): W = packed (dout x 64) image, dout a multiple of 128."""
    _req(x, "x")
    if x.shape[-1] != FEAT:
        raise ValueError("x: last dim must be 64")
    rows = x.numel() // FEAT
    y = torch.empty(tuple(x.shape[:-1]) + (dout,), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device), _Probed("node_linear_kernel", rows * 2 * 64 * dout):
        check(load().gn_node_linear_f32(_ptr(x), _ptr(W), _ptr(bias), _ptr(y), rows, dout, stream_handle()),
              "gn_node_linear_f32")
    return y


def agg_scatter_grouped(items: Sequence[tuple], divisor: Optional[float] = None) -> List[Tensor]:
    """items = [(feat (B,E,64), H or None, ori (B,N,64)[, sym])] -> [cat(H^T feat, ori) / divisor
    (B,N,128)]; divisor defaults to N (edge2node, model/MS_HGNN_batch.py:120,355)."""
    _groups(len(items))
    o0 = _req(items[0][2], "ori", (None, None, FEAT), _ACT_DTYPES)
    dt = o0.dtype
    B, N, _ = o0.shape
    arr = (_lib.ScatterGroup * len(items))()
    outs = []
    for g, item in enumerate(items):
        feat, H, ori = item[:3]
        sym = bool(item[3]) if len(item) > 3 else False
        _req(ori, "ori", (B, N, FEAT), dt)
        E = _edge_count(H, B, N, sym)
        _req(feat, "feat", (B, E, FEAT), dt)
        _same_device(o0, feat, ori, H)
        out = torch.empty((B, N, 2 * FEAT), dtype=ori.dtype, device=ori.device)
        arr[g] = _lib.ScatterGroup(feat.data_ptr(), 0 if H is None else H.data_ptr(), ori.data_ptr(), out.data_ptr(), E,
                                   int(sym))
        outs.append(out)
    with torch.cuda.device(o0.device):
        check(_fn("gn_agg_scatter", dt)(arr, len(items), B, N, float(N if divisor is None else divisor),
                                        stream_handle()), "gn_agg_scatter")
    return outs


def agg_scatter(feat: Tensor, H: Optional[Tensor], ori: Tensor, divisor: Optional[float] = None,
                sym: bool = False) -> Tensor:
    return agg_scatter_grouped([(feat, H, ori, sym)], divisor)[0]


# ---- A6 ------------------------------------------------------------------------------------------
def _mlp2_out(lead: Tuple[int, ...], dout: int, out: Optional[Tensor], like: Tensor) -> Tuple[Tensor, int]:
    if out is None:
        return torch.empty(tuple(lead) + (dout,), dtype=like.dtype, device=like.device), dout
    if not (out.is_cuda and out.dtype == like.dtype and out.device == like.device):
        raise ValueError("out: must be a tensor of the input's dtype on the input's device")
    if tuple(out.shape) != tuple(lead) + (dout,) or out.stride(-1) != 1:
        raise ValueError(f"out: expected shape {tuple(lead) + (dout,)} with unit inner stride")
    ldy = out.stride(-2) if out.dim() >= 2 else dout
    for d in range(out.dim() - 2):   # leading dims must be row-contiguous w.r.t. ldy
        if out.stride(d) != out.stride(d + 1) * out.shape[d + 1]:
            raise ValueError("out: leading dimensions must be contiguous")
    return out, ldy


class ScatterSpec:
    """Input rows of a 128-wide MLP to be formed inside the kernel instead of read from a tensor:
    cat(H^T feat, ori) / divisor (divisor defaults to N) — what agg_scatter would have produced."""
    __slots__ = ("feat", "H", "ori", "sym", "divisor")

    def __init__(self, feat: Tensor, H: Optional[Tensor], ori: Tensor, sym: bool = False,
                 divisor: Optional[float] = None):
        self.feat, self.H, self.ori, self.sym, self.divisor = feat, H, ori, bool(sym), divisor


class NodeAggSpec:
    """Input rows of a 128-wide MLP formed inside the kernel from H^T feat per NODE (the node form of the typed
    aggregation, ``PairSpec(node=True)``) and ori: cat(agg, ori) / divisor (divisor defaults to N)."""
    __slots__ = ("agg", "ori", "divisor")

    def __init__(self, agg: Tensor, ori: Tensor, divisor: Optional[float] = None):
        self.agg, self.ori, self.divisor = agg, ori, divisor


def mlp2_grouped(items: Sequence[Tuple[object, dict, Optional[Tensor]]], keep: Optional[List[dict]] = None
                 ) -> List[Tensor]:
    """items = [(x (..., din) or ScatterSpec, pk{"W","bias","din","dh","dout"}, out or None)], same
    shapes and row stride for every group.  ``out`` may be a last-dim slice of a contiguous tensor
    (row stride > dout): the kernel writes the column block in place.  ``keep`` (training): a list that
    receives, per group, {"x": the MLP's input rows as evaluated (rows, din), "hid": relu(W0 x + b0) (rows, dh)}
    — what the backward needs and the fused kernel otherwise never writes."""
    _groups(len(items))
    pk0 = items[0][1]
    din, dh, dout = pk0["din"], pk0["dh"], pk0["dout"]
    arr = (_lib.Mlp2Group * len(items))()
    outs, ld0, shape0, dev0, N, divisor, dt = [], None, None, None, 0, 1.0, None
    for g, (x, pk, out) in enumerate(items):
        if (pk["din"], pk["dh"], pk["dout"]) != (din, dh, dout):
            raise ValueError("grouped mlp2: every group must have the same layer widths")
        if isinstance(x, NodeAggSpec):
            if din != 2 * FEAT:
                raise ValueError("NodeAggSpec feeds a 128-wide MLP")
            _req(x.ori, "ori", (None, None, FEAT), _ACT_DTYPES)
            B, Nn, _ = x.ori.shape
            _req(x.agg, "agg", (B, Nn, FEAT), x.ori.dtype)
            _same_device(x.ori, x.agg)
            d = float(Nn if x.divisor is None else x.divisor)
            if N and (N, divisor) != (Nn, d):
                raise ValueError("grouped mlp2: every fused-scatter group must share N and divisor")
            N, divisor = Nn, d
            lead, like = (B, Nn), x.ori
            fields = (0, pk["W"].data_ptr(), pk["bias"].data_ptr(), None, x.agg.data_ptr(), 0, x.ori.data_ptr(), 0, 0)
        elif isinstance(x, ScatterSpec):
            if din != 2 * FEAT:
                raise ValueError("ScatterSpec feeds a 128-wide MLP")
            _req(x.ori, "ori", (None, None, FEAT), _ACT_DTYPES)
            B, Nn, _ = x.ori.shape
            E = _edge_count(x.H, B, Nn, x.sym)
            _req(x.feat, "feat", (B, E, FEAT), x.ori.dtype)
            _same_device(x.ori, x.feat, x.H)
            d = float(Nn if x.divisor is None else x.divisor)
            if N and (N, divisor) != (Nn, d):
                raise ValueError("grouped mlp2: every fused-scatter group must share N and divisor")
            N, divisor = Nn, d
            lead, like = (B, Nn), x.ori
            fields = (0, pk["W"].data_ptr(), pk["bias"].data_ptr(), None, x.feat.data_ptr(),
                      0 if x.H is None else x.H.data_ptr(), x.ori.data_ptr(), E, int(x.sym))
        else:
            _req(x, "x", None, _ACT_DTYPES)
            if x.shape[-1] != din:
                raise ValueError(f"x: last dim {x.shape[-1]} != {din}")
            lead, like = tuple(x.shape[:-1]), x
            fields = (x.data_ptr(), pk["W"].data_ptr(), pk["bias"].data_ptr(), None, 0, 0, 0, 0, 0)
        if shape0 is None:
            shape0, dev0, dt = lead, like.device, like.dtype
        elif lead != shape0 or like.device != dev0 or like.dtype != dt:
            raise ValueError("grouped mlp2: every group must have the same leading shape, device and storage type")
        y, ldy = _mlp2_out(lead, dout, out, like)
        if ld0 is None:
            ld0 = ldy
        elif ldy != ld0:
            raise ValueError("grouped mlp2: every group must have the same output row stride")
        kp = (0, 0)
        if keep is not None:
            if _twin(dt):
                raise ValueError("the bf16 twins are forward-only")
            nrow = 1
            for d_ in lead:
                nrow *= int(d_)
            kd = dict(x=torch.empty((nrow, din), dtype=like.dtype, device=like.device),
                      hid=torch.empty((nrow, dh), dtype=like.dtype, device=like.device))
            keep.append(kd)
            kp = (kd["x"].data_ptr(), kd["hid"].data_ptr())
        arr[g] = _lib.Mlp2Group(fields[0], fields[1], fields[2], y.data_ptr(), *fields[4:], *kp,
                                _ximg(pk, "mlp2", dt) if dout <= 64 else 0, _himg(pk, "mlp2", dt) if dout <= 64 else 0)
        outs.append(y)
    rows = 1
    for d_ in shape0:
        rows *= int(d_)
    flops = len(items) * rows * 2 * (din * dh + dh * (((dout + 31) // 32) * 32))
    with torch.cuda.device(dev0), _Probed("mlp2_kernel", flops):
        check(_fn("gn_mlp2", dt)(arr, len(items), rows, din, dh, dout, ld0, N, divisor, stream_handle()),
              "gn_mlp2")
    return outs


def mlp2(x: Tensor, pk: dict, out: Optional[Tensor] = None) -> Tensor:
    """y = W1 relu(W0 x + b0) + b1 over the last dim of x."""
    return mlp2_grouped([(x, pk, out)])[0]


# ---- noise ---------------------------------------------------------------------------------------
def philox_uniform(shape: Sequence[int], seed: int, offset: int, device, offset_dev: Optional[Tensor] = None) -> Tensor:
    """Uniforms in [0,1) from the Philox4x32-10 stream `seed` at element position
    `offset` (+ the int64 device counter `offset_dev`, if given)."""
    U = torch.empty(tuple(shape), dtype=torch.float32, device=device)
    if not U.is_cuda:
        raise ValueError("philox_uniform: device must be a GPU")
    if offset_dev is not None and not (offset_dev.is_cuda and offset_dev.dtype == torch.int64 and offset_dev.numel() == 1):
        raise ValueError("offset_dev: a 1-element int64 GPU tensor")
    with torch.cuda.device(U.device):
        check(load().gn_philox_uniform_f32(_ptr(U), U.numel(), int(seed) & (2**64 - 1), int(offset), _ptr(offset_dev),
                                           stream_handle()), "gn_philox_uniform_f32")
    return U


def counter_add(counter: Tensor, add: int) -> None:
    """counter += add, in stream order (counter: 1-element int64 GPU tensor)."""
    if not (counter.is_cuda and counter.dtype == torch.int64 and counter.numel() == 1):
        raise ValueError("counter: a 1-element int64 GPU tensor")
    with torch.cuda.device(counter.device):
        check(load().gn_counter_add_u64(_ptr(counter), int(add), stream_handle()), "gn_counter_add_u64")




def copy_cols(dst: Tensor, src: Tensor) -> None:
    """dst (contiguous) <- src, where src is a last-dim slice of a wider contiguous tensor (a column block of the
    concatenated features): one pitched copy launch (`gn_copy_2d`) instead of a strided elementwise copy."""
    if not (src.is_cuda and dst.is_cuda and src.dtype == dst.dtype and tuple(src.shape) == tuple(dst.shape)):
        raise ValueError("copy_cols: GPU tensors of equal shape and dtype")
    w = src.shape[-1] * src.element_size()
    rows = src.numel() // src.shape[-1]
    ok = (dst.is_contiguous() and src.stride(-1) == 1 and w % 16 == 0 and src.dim() >= 2
          and (src.stride(-2) * src.element_size()) % 16 == 0 and src.data_ptr() % 16 == 0 and dst.data_ptr() % 16 == 0
          and all(src.stride(d) == src.stride(d + 1) * src.shape[d + 1] for d in range(src.dim() - 2)))
    if not ok:
        dst.copy_(src, non_blocking=True)
        return
    with torch.cuda.device(src.device):
        check(load().gn_copy_2d(_P(dst.data_ptr()), w, _P(src.data_ptr()), src.stride(-2) * src.element_size(), w, rows,
                                stream_handle()), "gn_copy_2d")
