// Row-wise fused MLP chains on the bf16 matrix cores of gfx950 (v_mfma_f32_32x32x16_bf16), included by
// gn_mlp_mfma.hip (one translation unit, shared launch tables).
//
// Every kernel here exists in two precisions, selected by the template parameter P ("parts"):
//
//   P = 3, T = float  — fp32 results on the bf16 cores ("bf16x6").  Every fp32 operand is x = x1 + x2 + x3 with
//          three bf16 parts (round to nearest; the remainders are exact in fp32): weights once per parameter
//          version (gn_split_bf16_f32, parts = 3), activations on the VALU when a tile is produced.  A product is
//          the six significant part-products, smallest first, accumulated in the fp32 accumulator:
//          w3x1 + w2x2 + w1x3 + w2x1 + w1x2 + w1x1.  As accurate as fp32 accumulation (2e-7 of max|result| at
//          K = 256), six 8-pass MFMAs per k = 16 instead of eight 16-pass fp32 MFMAs.  These are what the
//          *_f32 entry points run when a group carries the `Wx` image.
//   P = 1, T = __bf16 — the bf16 twins (BASELINE config 4): activations stored in HBM as bf16,
//          weights rounded once to bf16, one MFMA per k = 16, fp32 accumulation, bias / ReLU / softmax in fp32;
//          a layer's fp32 result is rounded to bf16 when it becomes the next layer's operand or is stored.
//
// Orientation and register layout are those of gn_mlp_mfma.hip (Y^T = W X^T: A operand = weights, B operand =
// activations; the 16 accumulator registers of lane (j, h) are 16 features of ITS row j): registers 8*hf .. 8*hf+7
// of a 32-feature tile are exactly the eight k-values lane (j, h) supplies to the k = 16 MFMA of half hf, once
// converted to bf16 — the k-permutation this implies is baked into the weight image by gn_split_bf16_f32.  A whole
// chain therefore runs in registers, hidden tile by hidden tile (one hidden tile live), with no LDS round trip.
//
// Weight image: a "sub-step" = the (32 outputs x 16 k) operand of one MFMA position, P pieces of 64 lanes x 16 B:
// piece p of sub-step s sits at ((s*P + p)*64 + lane) * 16 B.  Images are laid out in CONSUMPTION order (built by
// ops.PackPlan as fp32 tiles, two sub-steps per 32x32 tile, then split), so a kernel walks its image linearly
// through a register ring that runs D sub-steps ahead of the matrix pipe (D = 4 for P = 3: 24 MFMAs; D = 16 for
// P = 1: 16 MFMAs).
#pragma once
#include "gn_mlp_common.hpp"

#ifdef GN_STAMPS
// Diagnostic build only (never the product): per-wave cycle stamps, read back with gn_debug_read_stamps.
__device__ unsigned long long gn_stamp_buf[1 << 17];
#define GN_STAMP(unit, slot)                                                              \
  do {                                                                                    \
    if ((threadIdx.x & 63) == 0 && (unit) < (1 << 13))                                    \
      gn_stamp_buf[(size_t)(unit) * 16 + (slot)] = ((slot) & 8) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); \
  } while (0)
extern "C" int gn_debug_read_stamps(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(gn_stamp_buf), bytes);
}
#else
#define GN_STAMP(unit, slot) do {} while (0)
#endif

namespace {

template <int P>
struct Parts {
  bf16x8 p[P];
};

// bf16 operand(s) of half `hf` of a 32-feature fp32 tile held in a lane's 16 registers
template <int P>
__device__ __forceinline__ void make_parts(const f32x16& v, int hf, Parts<P>& x) {
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    if constexpr (P == 3) {
      __bf16 a, b, c;
      split3(v[8 * hf + jj], a, b, c);
      x.p[0][jj] = a;
      x.p[1][jj] = b;
      x.p[2][jj] = c;
    } else {
      x.p[0][jj] = (__bf16)v[8 * hf + jj];
    }
  }
}
template <int P, int NT>
__device__ __forceinline__ void make_parts_tiles(const f32x16 (&v)[NT], Parts<P> (&x)[NT][2]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    make_parts<P>(v[t], 0, x[t][0]);
    make_parts<P>(v[t], 1, x[t][1]);
  }
}

template <int P>
struct XRing {
#ifndef GN_RING3
#define GN_RING3 4
#endif
  static constexpr int D = P == 3 ? GN_RING3 : 16;   // ring depth in sub-steps
  f32x4 q[D * P];
  __device__ __forceinline__ void prime(const f32x4* __restrict__ src) {   // src: this lane's pointer at sub-step 0
#pragma unroll
    for (int u = 0; u < D * P; ++u) q[u] = src[u * 64];
  }
  // acc += W[sub-step in ring slot s % D] . x, then the slot is refilled from `next` (this lane's pointer at the
  // sub-step D ahead).  `s` must be a compile-time constant at every call site (fully unrolled callers).
  // FENCE: close the scheduling region behind the step (the default); a caller that interleaves other work with a
  // run of steps passes false and fences the run itself.
  template <bool FENCE = true>
  __device__ __forceinline__ void step(int s, const Parts<P>& x, f32x16& acc, const f32x4* __restrict__ next) {
    const int u = (s % D) * P;
    if constexpr (P == 3) {
      const bf16x8 w1 = __builtin_bit_cast(bf16x8, q[u + 0]);
      const bf16x8 w2 = __builtin_bit_cast(bf16x8, q[u + 1]);
      const bf16x8 w3 = __builtin_bit_cast(bf16x8, q[u + 2]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3, x.p[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x.p[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x.p[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x.p[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x.p[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x.p[0], acc, 0, 0, 0);
    } else {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, q[u]), x.p[0], acc, 0, 0, 0);
    }
#ifndef GN_NO_REFILL     // (diagnostic builds only: what the kernels would take with the weights already in registers)
#pragma unroll
    for (int p = 0; p < P; ++p) q[u + p] = next[p * 64];
#endif
    // hipcc otherwise sinks the run-ahead loads down to their use and collapses the ring
    if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
  }
};

// ---- one layer pair, hidden tile by hidden tile, software-pipelined ------------------------------------------------
//   out[o] += W1(o, :) post(W0 x + b0)      x: IT input tiles (as parts), HT hidden tiles, OT output tiles
// A_t = the 2*IT sub-steps that produce hidden tile t, V_t = its VALU work (post: ReLU / scale / optional store,
// then the bf16 part(s)), B_t = the 2*OT sub-steps that consume it.  B_t needs V_t needs A_t, so executed in that
// order a lone wave leaves the matrix pipe idle during every V_t (~120 VALU instructions for three parts).  The
// pipeline issues A_{t+1} between A_t and B_t and interleaves V_t with its MFMAs (sched_group_barrier: one MFMA,
// then a few VALU), so the splitting runs in the shadow of the matrix pipe.  The weight image is laid out in this
// order:  A0 A1 B0 A2 B1 ... A(HT-1) B(HT-2) B(HT-1).  Bias tiles ride two hidden tiles ahead of their use: a
// load that is waited for right after it is issued would also wait for every run-ahead load of the weight ring
// (vmcnt counts in order).
//   pos0: stream position (sub-steps) of A0 — a compile-time constant at the call site; nxt(pos): this lane's
//   pointer at position pos + D;  hid0: bias tile 0 (loaded early by the caller).
template <int P, int IT, int OT, int HT, typename NextFn, typename PostFn>
__device__ __forceinline__ void layer_pair(XRing<P>& ring, int pos0, NextFn nxt, const Parts<P> (&xi)[IT][2],
                                           const f32x16& hid0, const float* __restrict__ b0, int h,
                                           f32x16 (&out)[OT], PostFn post) {
  constexpr int NA = 2 * IT, NB = 2 * OT;
  constexpr int kMfma = NA * (P == 3 ? 6 : 1);                 // MFMAs of one A phase
#ifndef GN_KVALU
#define GN_KVALU 96
#endif
  constexpr int kValu = P == 3 ? (GN_KVALU + kMfma - 1) / kMfma : (40 + kMfma - 1) / kMfma;   // VALU slots per MFMA
  int pos = pos0;
  f32x16 hidn = hid0;
  f32x16 bias_n;
  if (HT > 1) bias_n = load_bias_tile(b0 + 32, h);
#pragma unroll
  for (int u = 0; u < NA; ++u) {
    ring.step(pos, xi[u >> 1][u & 1], hidn, nxt(pos));
    ++pos;
  }
#pragma unroll
  for (int t = 0; t < HT; ++t) {
    f32x16 cur = hidn;
    if (t + 1 < HT) {
      hidn = bias_n;
      if (t + 2 < HT) bias_n = load_bias_tile(b0 + 32 * (t + 2), h);
    }
    post(t, cur);
    Parts<P> xh[2];
    make_parts<P>(cur, 0, xh[0]);
    make_parts<P>(cur, 1, xh[1]);
    if (t + 1 < HT) {
#pragma unroll
      for (int u = 0; u < NA; ++u) {
        ring.template step<false>(pos, xi[u >> 1][u & 1], hidn, nxt(pos));
        ++pos;
      }
#pragma unroll
      for (int i = 0; i < kMfma; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, kValu, 0);      // VALU of V_t in its shadow
        if (P == 3 ? (i & 1) == 1 : true) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // a ring refill
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      ring.step(pos, xh[u & 1], out[u >> 1], nxt(pos));
      ++pos;
    }
  }
}

template <typename T>
__device__ __forceinline__ void store_tile(T* __restrict__ p, const f32x16& a) {   // p: row base + 32*tile + 4*h
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = {a[4 * q + 0], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
    st4(p + 8 * q, v);
  }
}

// ---- node stage: x' = MLP(64->256->64)(x), pq = x' Wpq^T + bpq, and (pairwise module) A = WA x + bA ------------
// A3 first half (MS_HGNN_batch.py:125,131-134,358,362-365) and the per-node first layer of the typed aggregation
// MLP of the pairwise graph (MS_HGNN_batch.py:264-265; see gn_node_linear_f32) in ONE launch: both read the same
// node rows.  Work units, one wave each: a "chain" unit = one 32-row block of one group through the whole chain
// (72 sub-steps); an "A" unit = one row block x 8 output tiles of WA (32 sub-steps).  Long units first.
struct NodeTable {
  gn_node_group_t g[GN_MAX_GROUPS];
  int a_first[GN_MAX_GROUPS + 1];   // prefix of A units per group, relative to chain_units
  int n, rows, blocks32, chain_units, total_units;
};
constexpr int kATiles = 8;   // output tiles of WA per A unit

template <int P, typename T>
__global__ __launch_bounds__(256, 2) void node_stage_kernel(NodeTable Tb) {
  constexpr int D = XRing<P>::D;
  const int unit = blockIdx.x * 4 + wave_id();
  if (unit >= Tb.total_units) return;
  const int lane = threadIdx.x & 63, h = lane >> 5;
  if (unit < Tb.chain_units) {
    const int gi = gn_uniform(unit / Tb.blocks32);
    const gn_node_group_t G = Tb.g[gi];
    const RowBlock rb = row_block(Tb.rows, unit - gi * Tb.blocks32);
    GN_STAMP(unit, 0);
    GN_STAMP(unit, 8);
    const float* b0 = G.bias;
    const float* b1 = G.bias + 256;
    const float* bpq = G.bias + 320;
    const f32x4* Wx = reinterpret_cast<const f32x4*>(G.Wx) + lane;
    XRing<P> ring;
    ring.prime(Wx);
    f32x16 in[2];
    load_rows<2>(reinterpret_cast<const T*>(G.x), GN_FEAT, rb.row_ld, h, in);
    const f32x16 hid0 = load_bias_tile(b0, h);
    f32x16 xp[2], pq[2];
    xp[0] = load_bias_tile(b1, h);
    xp[1] = load_bias_tile(b1 + 32, h);
    pq[0] = load_bias_tile(bpq, h);
    pq[1] = load_bias_tile(bpq + 32, h);
    Parts<P> xi[2][2];
    make_parts_tiles<P, 2>(in, xi);
    GN_STAMP(unit, 1);
    constexpr int kSub = 72;
    auto nxt = [&](int s) { return Wx + (size_t)(s + D < kSub ? s + D : s + D - kSub) * P * 64; };
    // image: the 64->256->64 pair in pipeline order (A_t = [W0(t,in0), W0(t,in1)], B_t = [W1(0,t), W1(1,t)]), then
    // [Wpq(0,in0), Wpq(0,in1), Wpq(1,in0), Wpq(1,in1)]
    layer_pair<P, 2, 2, 8>(ring, 0, nxt, xi, hid0, b0, h, xp, [&](int t, f32x16& hid) {
      relu16(hid);
      if (G.hid_out != nullptr && rb.live) store_tile(G.hid_out + (size_t)rb.row * 256 + 32 * t + 4 * h, hid);
    });
    GN_STAMP(unit, 2);
    store_rows<2>(reinterpret_cast<T*>(G.xp), GN_FEAT, rb.row, h, rb.live, xp);
    Parts<P> xq[2][2];
    make_parts_tiles<P, 2>(xp, xq);
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int s = 64 + 4 * o;
      ring.step(s + 0, xq[0][0], pq[o], nxt(s + 0));
      ring.step(s + 1, xq[0][1], pq[o], nxt(s + 1));
      ring.step(s + 2, xq[1][0], pq[o], nxt(s + 2));
      ring.step(s + 3, xq[1][1], pq[o], nxt(s + 3));
    }
    GN_STAMP(unit, 3);
    store_rows<2>(reinterpret_cast<T*>(G.pq), GN_FEAT, rb.row, h, rb.live, pq);
    GN_STAMP(unit, 4);
    GN_STAMP(unit, 9);
    return;
  }
  // ---- A unit ----
  const int v = unit - Tb.chain_units;
  int gi = 0;
  while (gi + 1 < Tb.n && v >= Tb.a_first[gi + 1]) ++gi;
  gi = gn_uniform(gi);
  const gn_node_group_t G = Tb.g[gi];
  const int OTA = 4 * G.KA;                              // output tiles of WA (128 per type)
  const int chunks = (OTA + kATiles - 1) / kATiles;
  const int local = v - Tb.a_first[gi];
  const int blk = local / chunks, c = local - blk * chunks;
  const RowBlock rb = row_block(Tb.rows, blk);
  GN_STAMP(unit, 0);
  GN_STAMP(unit, 8);
  f32x16 in[2];
  load_rows<2>(reinterpret_cast<const T*>(G.x), GN_FEAT, rb.row_ld, h, in);
  Parts<P> xi[2][2];
  make_parts_tiles<P, 2>(in, xi);
  GN_STAMP(unit, 1);
  const int o0 = c * kATiles;
  const int nt = min(kATiles, OTA - o0);                 // a multiple of 4
  const int nsub = 4 * nt;
  const f32x4* Wa = reinterpret_cast<const f32x4*>(G.WAx) + lane + (size_t)o0 * 4 * P * 64;
  XRing<P> ring;
  ring.prime(Wa);
  const size_t ldA = (size_t)OTA * 32;
  T* arow = reinterpret_cast<T*>(G.A) + (size_t)rb.row * ldA + 4 * h;
  f32x16 bn = load_bias_tile(G.bA + 32 * o0, h);          // bias rides one tile ahead (see layer_pair)
#pragma unroll 1
  for (int o4 = 0; o4 < nt; o4 += 4) {
#pragma unroll
    for (int oo = 0; oo < 4; ++oo) {
      const int o = o0 + o4 + oo;
      f32x16 acc = bn;
      bn = load_bias_tile(G.bA + 32 * min(o + 1, o0 + nt - 1), h);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int s = 4 * oo + i;                        // ring slot: 16 sub-steps per iteration of the outer loop
        const int sg = 4 * o4 + s + D;
        ring.step(s, xi[i >> 1][i & 1], acc, Wa + (size_t)(sg < nsub ? sg : sg - nsub) * P * 64);
      }
      if (rb.live) store_tile(arow + 32 * o, acc);
    }
  }
  GN_STAMP(unit, 4);
  GN_STAMP(unit, 9);
}

// ---- A4: z = MLP(64->128->64); [dist|fac] heads; gumbel softmax; sigmoid -------------------------------------
// Image (80 sub-steps), hidden-tile-major: per hidden tile t of init_MLP the tiles [Wi0(t,in0), Wi0(t,in1),
// Wi1(0,t), Wi1(1,t)], then per hidden tile t of [Wd0] the tiles [Wd0(t,in0), Wd0(t,in1), Wd1(0,t)].
template <int P, typename T>
__global__ __launch_bounds__(256, 2) void edge_x_kernel(GroupTable<gn_edge_group_t> Tb, float tau,
                                                        unsigned long long seed,
                                                        const unsigned long long* __restrict__ offset_dev) {
  constexpr int D = XRing<P>::D;
  const int gi = find_group(Tb, blockIdx.x);
  const gn_edge_group_t G = Tb.g[gi];
  const int rows = G.rows, K = G.K;
  const int blk = (blockIdx.x - Tb.first_wg[gi]) * 4 + wave_id();
  if (blk * 32 >= rows) return;
  const RowBlock rb = row_block(rows, blk);
  const int lane = rb.lane, h = rb.h;
  const int unit = blockIdx.x * 4 + wave_id();
  GN_STAMP(unit, 0);
  GN_STAMP(unit, 8);
  f32x16 in[2], z[2], lg;
  load_rows<2>(reinterpret_cast<const T*>(G.edges), GN_FEAT, rb.row_ld, h, in);
  // ordered edge rows whose uniforms this row consumes: itself, or — symmetric pairwise form — the two ordered
  // edges (i,j) and (j,i) of its unordered pair
  long long o1 = rb.row_ld, o2 = rb.row_ld;
  bool diag = true;
  if (G.sym_N > 0) {
    const int N = G.sym_N, Pn = gn_pair_count(N);
    const int b = rb.row_ld / Pn, p = rb.row_ld - b * Pn;
    int i, j;
    gn_pair_decode(p, N, i, j);
    o1 = (long long)b * N * N + i * N + j;
    o2 = (long long)b * N * N + j * N + i;
    diag = i == j;
  }
  float u1[8], u2[8];
  const unsigned long long pbase = G.philox_offset + (offset_dev ? *offset_dev : 0ull);
  const float* bi0 = G.bias;
  const float* bi1 = G.bias + 128;
  const float* bd0 = G.bias + 192;
  const float* bd1 = G.bias + 448;
  const f32x4* Wx = reinterpret_cast<const f32x4*>(G.Wx) + lane;
  XRing<P> ring;
  ring.prime(Wx);
  constexpr int kSub = 80;
  auto nxt = [&](int s) { return Wx + (size_t)(s + D < kSub ? s + D : s + D - kSub) * P * 64; };
  const f32x16 hidA0 = load_bias_tile(bi0, h);
  const f32x16 hidB0 = load_bias_tile(bd0, h);
  z[0] = load_bias_tile(bi1, h);
  z[1] = load_bias_tile(bi1 + 32, h);
  f32x16 lgv[1];
  lgv[0] = load_bias_tile(bd1, h);
  Parts<P> xi[2][2];
  make_parts_tiles<P, 2>(in, xi);
  GN_STAMP(unit, 1);
  // ---- pair A: 64 -> 128 -> 64, 4 hidden tiles x (4 + 4) sub-steps, pipeline order ----
  layer_pair<P, 2, 2, 4>(ring, 0, nxt, xi, hidA0, bi0, h, z, [&](int t, f32x16& hid) {
    relu16(hid);
    if (G.keep_z1 != nullptr && rb.live) store_tile(G.keep_z1 + (size_t)rb.row * 128 + 32 * t + 4 * h, hid);
  });
  GN_STAMP(unit, 2);
  if (G.keep_z != nullptr) store_rows<2>(G.keep_z, GN_FEAT, rb.row, h, rb.live, z);
  make_parts_tiles<P, 2>(z, xi);
  // ---- pair B: 64 -> 256 -> (logits | factor), 8 hidden tiles x (4 + 2) sub-steps, pipeline order ----
  layer_pair<P, 2, 1, 8>(ring, 32, nxt, xi, hidB0, bd0, h, lgv, [&](int t, f32x16& hid) {
    relu16(hid);
    if (G.keep_dh1 != nullptr && rb.live) store_tile(G.keep_dh1 + (size_t)rb.row * 256 + 32 * t + 4 * h, hid);
  });
  lg = lgv[0];
  GN_STAMP(unit, 3);
  // uniforms: read from U or generated from the Philox stream — after the chains (nothing is in flight any more)
  fetch_uniforms(G.U, pbase, seed, o1, K, h, u1);
  if (G.sym_N > 0) fetch_uniforms(G.U, pbase, seed, o2, K, h, u2);
  if (G.keep_lgf != nullptr && rb.live) store_tile(G.keep_lgf + (size_t)rb.row * 32 + 4 * h, lg);

  // Epilogue.  Features 0..K-1 of `lg` are the logits of this lane's row, feature K the factor pre-activation; a
  // row's features are split over its two lanes (j, h=0) and (j, h=1).
  float facv = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r)
    if (feat_of(r, h) == K) facv = lg[r];
  facv += __shfl_xor(facv, 32, GN_WAVE);   // exactly one of the two lanes holds it, the other has 0
  const float sig = 1.f / (1.f + expf(-facv));
  float d1[8], d2[8];
  gumbel_softmax_row(lg, u1, K, tau, h, d1);
  if (G.sym_N > 0) gumbel_softmax_row(lg, u2, K, tau, h, d2);
  if (rb.live) {
    float* frow = G.edge_feat + (size_t)rb.row * K;
    T* dist = reinterpret_cast<T*>(G.dist);
    if (G.sym_N == 0) {
      T* drow = dist + (size_t)rb.row * K;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int f = feat_of(r, h);
        if (f < K) {
          st1(drow + f, d1[r]);
          frow[f] = sig * d1[r];
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int f = feat_of(r, h);
        if (f < K) {
          if (dist != nullptr) {
            st1(dist + (size_t)o1 * K + f, d1[r]);
            if (!diag) st1(dist + (size_t)o2 * K + f, d2[r]);
          }
          // both ordered edges meet the same typed MLP output downstream; the self-loop has weight 2
          frow[f] = diag ? 2.f * (sig * d1[r]) : sig * d1[r] + sig * d2[r];
        }
      }
    }
  }
  GN_STAMP(unit, 4);
  GN_STAMP(unit, 9);
}

// ---- A5 typed MLP on the bf16 cores: feat = sum_k ef[:,k] * (W2k relu(W1k eo + b1k) + b2k) -----------------------
// Same work shapes as agg_mlp_kernel (wpr waves share a row block, partial sums meet in LDS).  Forms:
//   two-layer  (W12x): per type and hidden tile o the tiles [W1k(o,in0), W1k(o,in1), W2k(0,o), W2k(1,o)] (32
//              sub-steps per type); input rows from eo, or gathered on the fly from ori (H / pairwise);
//   pair form  (A, W2x; P = 3 only): layer 1 was applied per node (node stage), per type and hidden tile t the tiles
//              [W2k(0,t), W2k(1,t)] (16 sub-steps per type); the two nodes' pre-activations are read from LDS
//              (staged: the workgroup's scenes fit) or straight from HBM/L2.
// b2k enters as one fp32 MFMA per output tile: lane (i, h=0) carries b2k[32o + i], paired with B = ef_k on k-index 0
__device__ __forceinline__ void add_b2(const float* __restrict__ b2k, float efk, int lane, int h, f32x16 (&out)[2]) {
  const float f0 = h == 0 ? b2k[lane & 31] : 0.f;
  const float f1 = h == 0 ? b2k[32 + (lane & 31)] : 0.f;
  const float efb = h == 0 ? efk : 0.f;
  out[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f0, efb, out[0], 0, 0, 0);
  out[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f1, efb, out[1], 0, 0, 0);
}

template <int P, typename T>
__global__ __launch_bounds__(256, 2) void agg_x_kernel(GroupTable<AggGroup> Tb) {
  constexpr int D = XRing<P>::D;
  __shared__ float part[4][32][64 + 8];   // wpr > 1: [wave][register 0..31][lane]; staged pair form: 2 x node rows
  const int gi = find_group(Tb, blockIdx.x);
  const gn_agg_group_t G = Tb.g[gi].a;
  const int wpr = Tb.g[gi].wpr;
  const int rows = G.rows, K = G.K;
  const int wave = wave_id();
  const int wg = blockIdx.x - Tb.first_wg[gi];
  const int sub = wave % wpr;                       // which share of the types
  const int blk = wg * (4 / wpr) + wave / wpr;      // which row block
  const bool any_rows = blk * 32 < rows;
  const bool staged = Tb.g[gi].stage != 0;
  if (wpr == 1 && !any_rows && !staged) return;     // (a staged workgroup keeps all its waves for the barriers)
  const RowBlock rb = row_block(rows, any_rows ? blk : 0);
  const int lane = rb.lane, h = rb.h;
  f32x16 out[2];
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[o][r] = 0.f;
  const float* efrow = G.edge_feat + (size_t)rb.row_ld * K;
  const float* b1 = G.b1;
  const float* b2 = G.b2;
  XRing<P> ring;
  const int unit = blockIdx.x * 4 + wave;
  GN_STAMP(unit, 0);
  GN_STAMP(unit, 8);

  bool pair_form = false;
  if constexpr (P == 3) pair_form = G.A != nullptr;
  if (pair_form) {
    // ---- pair form: hid_t = relu(A_i + A_j) * ef_k is VALU work (V_t), its layer-2 slice the matrix work (B_t).
    // Pipelined: V of the NEXT tile (the next type's tile 0 after t == 3) is interleaved with the MFMAs of B_t, the
    // pre-activations it needs were loaded one tile earlier still.
    const int N = G.N, Pn = G.E;
    int i, j;
    {
      const int b = rb.row_ld / Pn, p = rb.row_ld - b * Pn;
      gn_pair_decode(p, N, i, j);
      i += b * N;
      j += b * N;
    }
    const size_t ldA = (size_t)K * 128;
    const T* Abase = reinterpret_cast<const T*>(G.A);
    const f32x4* Wx = reinterpret_cast<const f32x4*>(G.W2x) + lane;   // sub-step s of type k: (k*16 + s)
    auto hidden = [&](const PreTile& pa, const PreTile& pb, float efk, Parts<P> (&xh)[2]) {
      f32x16 hid;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) hid[4 * q + c] = fmaxf(pa.v[q][c] + pb.v[q][c], 0.f) * efk;
      make_parts<P>(hid, 0, xh[0]);
      make_parts<P>(hid, 1, xh[1]);
    };
    // B_t with the next tile's V in its shadow
    auto slice = [&](int t, const Parts<P> (&xh)[2], const f32x4* cur, const f32x4* nx, const PreTile& pa,
                     const PreTile& pb, float ef_next, Parts<P> (&xh_next)[2]) {
      hidden(pa, pb, ef_next, xh_next);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int s = 4 * t + u;                      // [W2(0,t) hf0, hf1, W2(1,t) hf0, hf1]
        const f32x4* src = s + D < 16 ? cur + (size_t)(s + D) * P * 64 : nx + (size_t)(s + D - 16) * P * 64;
        ring.template step<false>(s, xh[u & 1], out[u >> 1], src);
      }
#pragma unroll
      for (int m = 0; m < 24; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
        if (m & 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    if (staged) {
      // The 4 row blocks of this workgroup touch a short run of consecutive node rows: the workgroup copies that
      // run (one type at a time, coalesced, prefetched in registers a type ahead) into one of two LDS buffers.
      float* stage = &part[0][0][0];
      const int r0 = wg * 128, r1 = min(rows - 1, r0 + 127);
      const int node0 = (r0 / Pn) * N;
      const int nodes = (r1 / Pn + 1) * N - node0;
      const f32x4* Ag = reinterpret_cast<const f32x4*>(Abase + (size_t)node0 * ldA);
      const int total4 = nodes * 32;                      // 16-byte pieces per type (fp32 storage)
      f32x4 pre[kStageLoadsX];
      auto fetch = [&](int kk) {
#pragma unroll
        for (int it = 0; it < kStageLoadsX; ++it) {
          const int idx = min((int)threadIdx.x + it * 256, total4 - 1);
          pre[it] = Ag[(size_t)(idx >> 5) * (ldA / 4) + kk * 32 + (idx & 31)];
        }
      };
      auto commit = [&](int buf) {
#pragma unroll
        for (int it = 0; it < kStageLoadsX; ++it) {
          const int idx = (int)threadIdx.x + it * 256;
          if (idx < total4)
            *reinterpret_cast<f32x4*>(stage + buf * kStageBuf + (idx >> 5) * kStagePitch + (idx & 31) * 4) = pre[it];
        }
      };
      const int oi = (i - node0) * kStagePitch, oj = (j - node0) * kStagePitch;
      ring.prime(Wx);
      fetch(0);
      commit(0);
      fetch(K > 1 ? 1 : 0);
      __syncthreads();
      float efk = efrow[0];
      Parts<P> xh[2];
      PreTile pa = load_pre(stage + oi, h), pb = load_pre(stage + oj, h);
      hidden(pa, pb, efk, xh);                            // V of (type 0, tile 0): the only exposed one
      pa = load_pre(stage + oi + 32, h);
      pb = load_pre(stage + oj + 32, h);
#pragma unroll 1
      for (int k = 0; k < K; ++k) {
        const int kc = k + 1 < K ? k + 1 : k;
        const float efk_next = efrow[kc];
        const float* cb = stage + (k & 1) * kStageBuf;        // this type's rows
        const float* nb = stage + (kc & 1) * kStageBuf;       // the next type's
        if (k + 1 < K) commit((k + 1) & 1);                   // (that buffer was last read during type k - 1)
        if (k + 2 < K) fetch(k + 2);
        add_b2(b2 + k * 64, efk, lane, h, out);
        const f32x4* cur = Wx + (size_t)k * 16 * P * 64;
        const f32x4* nx = Wx + (size_t)kc * 16 * P * 64;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          // pa/pb hold tile t+1 (tile 0 of the next type when t == 3); fetch the one after it
          const PreTile qa = pa, qb = pb;
          if (t == 2) __syncthreads();                        // the next type's rows are committed by every wave
          if (t < 2) {
            pa = load_pre(cb + oi + 32 * (t + 2), h);
            pb = load_pre(cb + oj + 32 * (t + 2), h);
          } else {
            pa = load_pre(nb + oi + 32 * (t - 2), h);
            pb = load_pre(nb + oj + 32 * (t - 2), h);
          }
          Parts<P> xn[2];
          slice(t, xh, cur, nx, qa, qb, t < 3 ? efk : efk_next, xn);
          xh[0] = xn[0];
          xh[1] = xn[1];
        }
        efk = efk_next;
      }
      if (!any_rows) return;
    } else if (any_rows && sub < K) {
      const T* Ai = Abase + (size_t)i * ldA;
      const T* Aj = Abase + (size_t)j * ldA;
      int k = sub;
      ring.prime(Wx + (size_t)k * 16 * P * 64);
      float efk = efrow[k];
      Parts<P> xh[2];
      PreTile pa = load_pre(Ai + k * 128, h), pb = load_pre(Aj + k * 128, h);
      hidden(pa, pb, efk, xh);
      pa = load_pre(Ai + k * 128 + 32, h);
      pb = load_pre(Aj + k * 128 + 32, h);
#pragma unroll 1
      while (k < K) {
        const int kn = k + wpr;
        const int kc = kn < K ? kn : k;
        const float efk_next = efrow[kc];
        add_b2(b2 + k * 64, efk, lane, h, out);
        const f32x4* cur = Wx + (size_t)k * 16 * P * 64;
        const f32x4* nx = Wx + (size_t)kc * 16 * P * 64;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const PreTile qa = pa, qb = pb;
          const int off = t < 2 ? k * 128 + 32 * (t + 2) : kc * 128 + 32 * (t - 2);
          pa = load_pre(Ai + off, h);
          pb = load_pre(Aj + off, h);
          Parts<P> xn[2];
          slice(t, xh, cur, nx, qa, qb, t < 3 ? efk : efk_next, xn);
          xh[0] = xn[0];
          xh[1] = xn[1];
        }
        efk = efk_next;
        k = kn;
      }
    }
  } else if (any_rows && sub < K) {
    // ---- two-layer form, hidden tile by hidden tile: tile o of layer 1 (4 sub-steps), ReLU * ef_k, its bf16
    // part(s), then its contribution to both output tiles (4 sub-steps) — one hidden tile live ----------------------
    f32x16 in[2];
    if (G.eo != nullptr)
      load_rows<2>(reinterpret_cast<const T*>(G.eo), GN_FEAT, rb.row_ld, h, in);
    else
      gather_rows<T>(G, rb.row_ld, h, in);
    Parts<P> xi[2][2];
    make_parts_tiles<P, 2>(in, xi);
    GN_STAMP(unit, 1);
    const f32x4* Wx = reinterpret_cast<const f32x4*>(G.W12x) + lane;   // sub-step s of type k: (k*32 + s)
    int k = sub;
    ring.prime(Wx + (size_t)k * 32 * P * 64);
    f32x16 hid0 = load_bias_tile(b1 + k * 128, h);
    float efk = efrow[k];
#pragma unroll 1
    while (k < K) {
      const int kn = k + wpr;
      const int kc = kn < K ? kn : k;
      const f32x4* cur = Wx + (size_t)k * 32 * P * 64;
      const f32x4* nx = Wx + (size_t)kc * 32 * P * 64;
      const float efk_next = efrow[kc];
      const f32x16 hid0_next = load_bias_tile(b1 + kc * 128, h);
      add_b2(b2 + k * 64, efk, lane, h, out);
      auto src = [&](int s) { return s + D < 32 ? cur + (size_t)(s + D) * P * 64 : nx + (size_t)(s + D - 32) * P * 64; };
      layer_pair<P, 2, 2, 4>(ring, 0, src, xi, hid0, b1 + k * 128, h, out,
                             [&](int, f32x16& hid) { relu_scale16(hid, efk); });
      hid0 = hid0_next;
      efk = efk_next;
      k = kn;
    }
  }
  T* feat = reinterpret_cast<T*>(G.feat);
  GN_STAMP(unit, 2);
  if (wpr == 1) {
    store_rows<2>(feat, GN_FEAT, rb.row, h, rb.live, out);
    GN_STAMP(unit, 4);
    GN_STAMP(unit, 9);
    return;
  }
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][16 * o + r][lane] = out[o][r];
  __syncthreads();
  // the wpr waves of a row block each finish 32/wpr of its registers
  if (rb.live && any_rows) {
    T* p = feat + (size_t)rb.row * GN_FEAT + 4 * h;
    const int w0 = wave - sub;
    const int nreg = 32 / wpr;
    for (int rr = 0; rr < nreg; rr += 4) {
      const int reg0 = sub * nreg + rr;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      for (int jw = 0; jw < wpr; ++jw) {
        v[0] += part[w0 + jw][reg0 + 0][lane];
        v[1] += part[w0 + jw][reg0 + 1][lane];
        v[2] += part[w0 + jw][reg0 + 2][lane];
        v[3] += part[w0 + jw][reg0 + 3][lane];
      }
      const int o = reg0 >> 4, q = (reg0 & 15) >> 2;
      st4(p + 32 * o + 8 * q, v);
    }
  }
  GN_STAMP(unit, 4);
  GN_STAMP(unit, 9);
}

// ---- A6 / closing MLP on the bf16 cores: y = W1 relu(W0 x + b0) + b1, dout <= 64 ---------------------------------
// Image, hidden-tile-major: per hidden tile t the tiles [W0(t, in 0..IT-1), W1(0..OT-1, t)].  Input rows read from
// x or formed on the fly (fused scatter, IT == 4) exactly as in mlp2_kernel.  blockIdx.y = group.
template <int P, typename T, int IT, int HT, int OT>
__global__ __launch_bounds__(256, (P == 3 && IT == 4) ? 1 : 2) void mlp2_x_kernel(GroupTable<gn_mlp2_group_t> Tb, int rows, int dout, int ldy,
                                                        int N, float divisor) {
  constexpr int D = XRing<P>::D;
  const int blk = blockIdx.x * 4 + wave_id();
  if (blk * 32 >= rows) return;
  const gn_mlp2_group_t G = Tb.g[blockIdx.y];
  const RowBlock rb = row_block(rows, blk);
  const int lane = rb.lane, h = rb.h;
  const int unit = (blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave_id();
  GN_STAMP(unit, 0);
  GN_STAMP(unit, 8);
  const f32x4* Wx = reinterpret_cast<const f32x4*>(G.Wx) + lane;
  XRing<P> ring;
  ring.prime(Wx);
  const float* b0 = G.bias;
  const float* b1 = G.bias + 32 * HT;
  const f32x16 hid0 = load_bias_tile(b0, h);
  f32x16 out[OT];
#pragma unroll
  for (int o = 0; o < OT; ++o) out[o] = load_bias_tile(b1 + 32 * o, h);
  Parts<P> xi[IT][2];
  {
    f32x16 in[IT];
    mlp2_rows<IT, T>(G, rb.row_ld, h, N, divisor, in);
    if (G.in_out != nullptr) store_rows<IT>(G.in_out, 32 * IT, rb.row, h, rb.live, in);   // kept for the backward
    make_parts_tiles<P, IT>(in, xi);
  }
  GN_STAMP(unit, 1);
  constexpr int kSub = HT * (2 * IT + 2 * OT);
  auto nxt = [&](int s) { return Wx + (size_t)(s + D < kSub ? s + D : s + D - kSub) * P * 64; };
  layer_pair<P, IT, OT, HT>(ring, 0, nxt, xi, hid0, b0, h, out, [&](int t, f32x16& hid) {
    relu16(hid);
    if (G.hid_out != nullptr && rb.live) store_tile(G.hid_out + (size_t)rb.row * (32 * HT) + 32 * t + 4 * h, hid);
  });
  GN_STAMP(unit, 2);
  if (rb.live) {
#pragma unroll
    for (int o = 0; o < OT; ++o) store_out_tile(reinterpret_cast<T*>(G.y), rb.row, ldy, dout, o, h, out[o]);
  }
  GN_STAMP(unit, 4);
  GN_STAMP(unit, 9);
}

}  // namespace
