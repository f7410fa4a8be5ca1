// Row-wise fused MLP chains on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Every dense layer of the path is a small nn.Linear applied
// to B*N node rows or B*E edge rows.  A wave owns a block of 32 rows and evaluates the whole
// chain for them in registers, in the TRANSPOSED orientation  Y^T = W . X^T :
//
//   MFMA A operand = weight tile   A[i][k] = W[32*o + i][k]      (i = lane & 31, k = lane >> 5)
//   MFMA B operand = activations   B[k][j] = X[row j][k]          (j = lane & 31)
//   result D[i][j]: lane (j, h = lane >> 5), register r  <->  out feature 32*o + (r&3) + 8*(r>>2) + 4*h
//
// so the 16 result registers of a lane are 16 features of ITS row — exactly the k-values the
// next layer's B operand wants from that lane.  A layer's output therefore feeds the next
// layer with no LDS round trip and no lane movement; bias is the initial accumulator and ReLU
// is a register-wise max.  The only memory traffic is the row block in, the row block out and
// the weight stream, which gn_pack_linear_f32 has laid out in the order the lanes consume it
// (one coalesced 1 KiB dwordx4 load per wave — a "step" — feeds four MFMAs).  All the weights of a
// kernel form ONE stream in consumption order, read through a register ring that runs kP steps
// (32 MFMAs, ~2k cycles) ahead of the matrix pipe, across tile, layer and loop boundaries; biases
// ride one tile ahead in 16 registers.  A lone wave per SIMD therefore stays MFMA-bound, which is
// what the small hyper modules (B*N rows = fewer waves than SIMDs) need.
//
// fp32 MFMA is an exact k-ordered fmaf chain, so results differ from the reference's MKL
// GEMMs only by summation order (~1e-7 relative).
#include <stdlib.h>

#include "gn_mlp_common.hpp"
#include "gn_mlp_bf16.hpp"

namespace {


// ---- packing -------------------------------------------------------------------------------
// Wp[(((o*IT + t)*4 + q)*64 + lane)*4 + c] = W[32o + (lane&31)][32t + 8q + 4(lane>>5) + c]
__global__ void pack_linear_kernel(const float* __restrict__ W, float* __restrict__ Wp, int out_f, int in_f,
                                   int ld, int col_off, int OT, int IT) {
  const size_t total = (size_t)OT * IT * kTileFloats;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int c = idx & 3;
    const int lane = (idx >> 2) & 63;
    const int q = (idx >> 8) & 3;
    const size_t tile = idx >> 10;
    const int t = (int)(tile % IT), o = (int)(tile / IT);
    const int row = 32 * o + (lane & 31);
    const int col = 32 * t + 8 * q + 4 * (lane >> 5) + c;
    Wp[idx] = (row < out_f && col < in_f) ? W[(size_t)row * ld + col_off + col] : 0.f;
  }
}

// Many weights, one launch.  A segment drops a source block (rows x cols, row-major, ld) into a packed
// image at (place_r, place_c) of the image's virtual matrix — or, IT == 0, into a plain row-major matrix
// (dst_ld) / vector — scaled.  The
// table lives in DEVICE memory: it is built once per module and parameter addresses, after which refreshing
// every packed weight of the module after an optimizer step is one fill + this one launch (and capturable).
__global__ __launch_bounds__(256) void pack_segments_kernel(const gn_pack_seg_t* __restrict__ segs) {
  const gn_pack_seg_t S = segs[blockIdx.y];
  const int total = S.rows * S.cols;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int r = idx / S.cols, c = idx - r * S.cols;
    const float v = S.scale * S.src[(size_t)r * S.ld + c];
    const int row = S.place_r + r, col = S.place_c + c;
    if (S.IT == 0) {
      S.dst[(size_t)row * S.dst_ld + col] = v;
    } else {
      const int o = row >> 5, t = col >> 5, cc = col & 31;
      const int lane = (row & 31) + 32 * ((cc & 7) >> 2);
      S.dst[((((size_t)o * S.IT + t) * 4 + (cc >> 3)) * 64 + lane) * 4 + (cc & 3)] = v;
    }
  }
}

// one thread per (tile, half, lane, j) of one image; the batch kernel serves one image per blockIdx.y
__device__ __forceinline__ void split_bf16_image(const float* __restrict__ packed, __bf16* __restrict__ out, int n_tiles,
                                                 int parts);
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ packed, __bf16* __restrict__ out,
                                                         int n_tiles, int parts) {
  split_bf16_image(packed, out, n_tiles, parts);
}
__global__ __launch_bounds__(256) void split_bf16_batch_kernel(const gn_split_job_t* __restrict__ jobs, int parts) {
  const gn_split_job_t J = jobs[blockIdx.y];
  split_bf16_image(J.packed, reinterpret_cast<__bf16*>(J.out), J.n_tiles, parts);
}
__device__ __forceinline__ void split_bf16_image(const float* __restrict__ packed, __bf16* __restrict__ out, int n_tiles,
                                                 int parts) {
  const long long total = (long long)n_tiles * 2 * 64 * 8;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63), half = (int)((idx >> 9) & 1);
    const long long tile = idx >> 10;
    const float x = packed[tile * kTileFloats + (((j >> 2) + 2 * half) * 64 + lane) * 4 + (j & 3)];
    __bf16* o = out + ((tile * 2 + half) * parts * 64 + lane) * 8 + j;
    if (parts == 2) {
      // f16x3 image: x = hi + lo, two fp16 parts (the remainder is exact in fp32).  A weight that does not fit fp16
      // raises the flag word behind the image: its workgroups then take the bf16x6 path (gn_mlp_bf16.hpp).
      const _Float16 hi = (_Float16)x;
      const _Float16 lo = (_Float16)(x - (float)hi);
      reinterpret_cast<_Float16*>(o)[0] = hi;
      reinterpret_cast<_Float16*>(o)[64 * 8] = lo;
      if (!(fabsf(x) <= 65000.f)) atomicOr(reinterpret_cast<int*>(out + (size_t)n_tiles * 2 * 2 * 64 * 8), 1);
      continue;
    }
    __bf16 p1, p2, p3;
    split3(x, p1, p2, p3);
    o[0] = p1;
    if (parts == 3) {
      o[64 * 8] = p2;
      o[2 * 64 * 8] = p3;
    }
  }
}


#ifndef GN_RING_DEPTH
#define GN_RING_DEPTH 8
#endif
constexpr int kP = GN_RING_DEPTH;  // ring depth in steps
constexpr int kStep = 64;      // f32x4 elements per step (one per lane)
struct WRing {
  f32x4 s[kP];
};
__device__ __forceinline__ void ring_prime(WRing& ring, const f32x4* __restrict__ p) {
#pragma unroll
  for (int i = 0; i < kP; ++i) ring.s[i] = p[i * kStep];
}

// acc += W[tile] . in, the tile being 4*IT consecutive steps at `cur` (this lane's pointer).  The ring
// always holds the next kP steps of the stream; START is the ring slot of the tile's first step (the
// running step count of the kernel modulo kP — 0 whenever every tile before it was a whole number of
// ring turns).  `nxt` points at the steps consumed after this tile (by default the ones that follow in
// memory).  `side(s)` runs right after the MFMAs of step s are issued: VALU work on a PREVIOUS tile's
// accumulator placed there executes in the shadow of this tile's MFMAs instead of stalling the pipe.
struct NoSide {
  __device__ __forceinline__ void operator()(int) const {}
};
// LN: length in steps of the chunk at `nxt` when it is shorter than the ring (kP); the steps after it then
// come from `nxt2`.
template <int IT, int START = 0, typename Side = NoSide, int LN = kP>
__device__ __forceinline__ void mma_tile(const f32x4* __restrict__ cur, const f32x4* __restrict__ nxt, WRing& ring,
                                         const f32x16 (&in)[IT], f32x16& acc, Side side = Side(),
                                         const f32x4* __restrict__ nxt2 = nullptr) {
  constexpr int S = 4 * IT;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int t = s >> 2, q = s & 3;
    const int slot = (START + s) % kP;
    const f32x4 w = ring.s[slot];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], in[t][4 * q + 0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], in[t][4 * q + 1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], in[t][4 * q + 2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3], in[t][4 * q + 3], acc, 0, 0, 0);
    side(s);
    if (s + kP < S)
      ring.s[slot] = cur[(s + kP) * kStep];
    else if (s + kP - S < LN)
      ring.s[slot] = nxt[(s + kP - S) * kStep];
    else
      ring.s[slot] = nxt2[(s + kP - S - LN) * kStep];
    // hipcc otherwise sinks the run-ahead load down to its use and collapses the ring to depth 1-2
    __builtin_amdgcn_sched_barrier(0);
  }
}


// State of a wave walking the weight / bias streams of a chain of layers laid out back to back.
struct Chain {
  const f32x4* w;     // this lane's pointer to the tile consumed next
  const float* b;     // bias tile of the tile AFTER the one whose bias sits in `bnext`
  WRing ring;
  f32x16 bnext;       // bias of the tile consumed next
  int lane, h;
};
__device__ __forceinline__ void chain_begin(Chain& c, const float* __restrict__ W, const float* __restrict__ bias,
                                            int lane) {
  c.lane = lane;
  c.h = lane >> 5;
  c.w = reinterpret_cast<const f32x4*>(W) + lane;
  ring_prime(c.ring, c.w);
  c.bnext = load_bias_tile(bias, c.h);
  c.b = bias + 32;
}
// out = act(W in + b): OT tiles of 4*IT steps each.  `last` marks the final layer of the kernel: its last
// tile has nothing after it, so the run-ahead loads are pointed back at valid memory.
template <int OT, int IT, bool RELU>
__device__ __forceinline__ void chain_linear(Chain& c, const f32x16 (&in)[IT], f32x16 (&out)[OT], bool last = false) {
  constexpr int S = 4 * IT;
#pragma unroll
  for (int o = 0; o < OT; ++o) {
    const bool tail = last && o == OT - 1;
    out[o] = c.bnext;
    c.bnext = load_bias_tile(tail ? c.b - 32 : c.b, c.h);
    mma_tile<IT>(c.w, tail ? c.w : c.w + S * kStep, c.ring, in, out[o]);
    if (RELU) relu16(out[o]);
    c.w += S * kStep;
    c.b += 32;
  }
}


// ---- A3 first half: x' = MLP(64->256->64)(x); pq = x' Wpq^T + bpq -----------------------------
// W = [W0 (256x64) | W1 (64x256) | Wpq (64x64)] packed, bias = [b0 | b1 | bpq].  blockIdx.y = group.
__global__ __launch_bounds__(256) void node_mlp_kernel(GroupTable<gn_node_group_t> T, int rows) {
  const int blk = blockIdx.x * 4 + wave_id();
  if (blk * 32 >= rows) return;  // whole wave past the end
  const gn_node_group_t G = T.g[blockIdx.y];
  const RowBlock rb = row_block(rows, blk);
  Chain c;
  chain_begin(c, G.W, G.bias, rb.lane);
  f32x16 in[2], hid[8], o1[2], o2[2];
  load_rows<2>(G.x, GN_FEAT, rb.row_ld, rb.h, in);
  chain_linear<8, 2, true>(c, in, hid);
  if (G.hid_out != nullptr) store_rows<8>(G.hid_out, 256, rb.row, rb.h, rb.live, hid);   // kept for the backward
  chain_linear<2, 8, false>(c, hid, o1);
  store_rows<2>(G.xp, GN_FEAT, rb.row, rb.h, rb.live, o1);
  chain_linear<2, 2, false>(c, o1, o2, true);
  store_rows<2>(G.pq, GN_FEAT, rb.row, rb.h, rb.live, o2);
}

// ---- A4: z = MLP(64->128->64); [dist|fac] heads; gumbel softmax; sigmoid ------------------------
// W = [Wi0 (128x64) | Wi1 (64x128) | Wd0 (256x64) | Wd1 (32x256)] packed, bias likewise.


__global__ __launch_bounds__(256, 2) void edge_mlp_gumbel_kernel(GroupTable<gn_edge_group_t> T, float tau,
                                                              unsigned long long seed,
                                                              const unsigned long long* __restrict__ offset_dev) {
  const int gi = find_group(T, blockIdx.x);
  const gn_edge_group_t G = T.g[gi];
  const int rows = G.rows, K = G.K;
  const int blk = (blockIdx.x - T.first_wg[gi]) * 4 + wave_id();
  if (blk * 32 >= rows) return;
  const RowBlock rb = row_block(rows, blk);
  f32x16 in[2], z[2], lg[1];
  load_rows<2>(G.edges, GN_FEAT, rb.row_ld, rb.h, in);
  // Ordered edge rows whose uniforms this row consumes: itself, or — symmetric pairwise form — the two
  // ordered edges (i,j) and (j,i) of its unordered pair.
  long long o1 = rb.row_ld, o2 = rb.row_ld;
  bool diag = true;
  if (G.sym_N > 0) {
    const int N = G.sym_N, P = gn_pair_count(N);
    const int b = rb.row_ld / P, p = rb.row_ld - b * P;
    int i, j;
    gn_pair_decode(p, N, i, j);
    o1 = (long long)b * N * N + i * N + j;
    o2 = (long long)b * N * N + j * N + i;
    diag = i == j;
  }
  // With a U tensor the loads are issued now, far ahead of the epilogue; Philox values are computed on
  // the VALU after the MFMAs are queued (no HBM traffic, no extra launch).
  float u1[8], u2[8];
  const unsigned long long pbase = G.philox_offset + (offset_dev ? *offset_dev : 0ull);
  if (G.U != nullptr) {
    fetch_uniforms(G.U, 0ull, 0ull, o1, K, rb.h, u1);
    if (G.sym_N > 0) fetch_uniforms(G.U, 0ull, 0ull, o2, K, rb.h, u2);
  }
  // Two layer PAIRS (64->128->64 and 64->256->32), each evaluated hidden-tile by hidden-tile: hidden tile t
  // (8 steps) is produced, and while tile t+1 is being produced its ReLU runs in the MFMA shadow; then the
  // second layer's slice over tile t (4 steps per output tile) is accumulated.  Only two hidden tiles are
  // ever live (32 registers instead of 128), which is what lets 2-3 waves share a SIMD.  The weight stream
  // was packed in exactly this order (edge_stream_order in ops.py):
  //   A: T0 T1 S0 T2 S1 T3 S2 S3        (T = Wi0 tile, S = the two Wi1 slices over it: 8 steps each)
  //   B: T0 T1 S0 T2 S1 ... T7 S6 S7    (T = Wd0 tile: 8 steps, S = the Wd1 slice over it: 4 steps)
  // followed by 8 steps of padding, because the ring always reads 8 steps ahead.
  {
    const int lane = rb.lane, h = rb.h;
    const float* bi0 = G.bias;
    const float* bi1 = G.bias + 128;
    const float* bd0 = G.bias + 192;
    const float* bd1 = G.bias + 448;
    const f32x4* w = reinterpret_cast<const f32x4*>(G.W) + lane;
    WRing ring;
    ring_prime(ring, w);
    f32x16 ha[1], hb[1];   // the two live hidden tiles (named, so that every index is static)
    z[0] = load_bias_tile(bi1, h);
    z[1] = load_bias_tile(bi1 + 32, h);
    lg[0] = load_bias_tile(bd1, h);
    // ---- pair A ----
    ha[0] = load_bias_tile(bi0, h);
    hb[0] = load_bias_tile(bi0 + 32, h);
    mma_tile<2>(w, w + 8 * kStep, ring, in, ha[0]);                                   // T0
    w += 8 * kStep;
    mma_tile<2>(w, w + 8 * kStep, ring, in, hb[0], [&](int s) { if (s == 1) relu16(ha[0]); });   // T1
    w += 8 * kStep;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x16(&cur)[1] = (t & 1) ? hb : ha;     // tile t
      f32x16(&oth)[1] = (t & 1) ? ha : hb;     // tile t+1 (already produced, ReLU pending) / tile t+2 target
      if (G.keep_z1 != nullptr) store_rows<1>(G.keep_z1 + 32 * t, 128, rb.row, h, rb.live, cur);   // (ReLU done)
      // S_t: z[0] += Wi1(0,t) h_t ; z[1] += Wi1(1,t) h_t      (ReLU of tile t+1 rides here when t == 2)
      mma_tile<1, 0>(w, w + 4 * kStep, ring, cur, z[0], [&](int s) { if (t == 2 && s == 1) relu16(oth[0]); });
      mma_tile<1, 4>(w + 4 * kStep, w + 8 * kStep, ring, cur, z[1]);
      w += 8 * kStep;
      if (t < 2) {
        // T_{t+2} into the register set tile t just vacated; ReLU of tile t+1 in its shadow
        cur[0] = load_bias_tile(bi0 + 32 * (t + 2), h);
        mma_tile<2>(w, w + 8 * kStep, ring, in, cur[0], [&](int s) { if (s == 1) relu16(oth[0]); });
        w += 8 * kStep;
      }
    }
    if (G.keep_z != nullptr) store_rows<2>(G.keep_z, GN_FEAT, rb.row, h, rb.live, z);
    // ---- pair B ----
    ha[0] = load_bias_tile(bd0, h);
    hb[0] = load_bias_tile(bd0 + 32, h);
    mma_tile<2>(w, w + 8 * kStep, ring, z, ha[0]);                                    // T0
    w += 8 * kStep;
    mma_tile<2>(w, w + 8 * kStep, ring, z, hb[0], [&](int s) { if (s == 1) relu16(ha[0]); });    // T1
    w += 8 * kStep;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      f32x16(&cur)[1] = (t & 1) ? hb : ha;
      f32x16(&oth)[1] = (t & 1) ? ha : hb;
      if (G.keep_dh1 != nullptr) store_rows<1>(G.keep_dh1 + 32 * t, 256, rb.row, h, rb.live, cur);
      // S_t: lg += Wd1(0,t) h_t  (4 steps; the ring slot alternates 0 / 4)
      if ((t & 1) == 0)
        mma_tile<1, 0>(w, w + 4 * kStep, ring, cur, lg[0], [&](int s) { if (t == 6 && s == 1) relu16(oth[0]); });
      else
        mma_tile<1, 4>(w, w + 4 * kStep, ring, cur, lg[0]);
      w += 4 * kStep;
      if (t < 6) {
        cur[0] = load_bias_tile(bd0 + 32 * (t + 2), h);
        if ((t & 1) == 0)
          mma_tile<2, 4>(w, w + 8 * kStep, ring, z, cur[0], [&](int s) { if (s == 1) relu16(oth[0]); });
        else
          mma_tile<2, 0>(w, w + 8 * kStep, ring, z, cur[0], [&](int s) { if (s == 1) relu16(oth[0]); });
        w += 8 * kStep;
      }
    }
  }
  if (G.U == nullptr) {
    fetch_uniforms(nullptr, pbase, seed, o1, K, rb.h, u1);
    if (G.sym_N > 0) fetch_uniforms(nullptr, pbase, seed, o2, K, rb.h, u2);
  }
  if (G.keep_lgf != nullptr) store_rows<1>(G.keep_lgf, 32, rb.row, rb.h, rb.live, lg);

  // Epilogue.  Features 0..K-1 of `lg` are the logits of this lane's row, feature K the factor
  // pre-activation; a row's features are split over its two lanes (j, h=0) and (j, h=1).
  float facv = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r)
    if (feat_of(r, rb.h) == K) facv = lg[0][r];
  facv += __shfl_xor(facv, 32, GN_WAVE);  // exactly one of the two lanes holds it, the other has 0
  const float sig = 1.f / (1.f + expf(-facv));
  float d1[8], d2[8];
  gumbel_softmax_row(lg[0], u1, K, tau, rb.h, d1);
  if (G.sym_N > 0) gumbel_softmax_row(lg[0], u2, K, tau, rb.h, d2);
  if (rb.live) {
    float* frow = G.edge_feat + (size_t)rb.row * K;
    if (G.sym_N == 0) {
      float* drow = G.dist + (size_t)rb.row * K;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int f = feat_of(r, rb.h);
        if (f < K) {
          drow[f] = d1[r];
          frow[f] = sig * d1[r];
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int f = feat_of(r, rb.h);
        if (f < K) {
          if (G.dist != nullptr) {
            G.dist[(size_t)o1 * K + f] = d1[r];
            if (!diag) G.dist[(size_t)o2 * K + f] = d2[r];
          }
          // both ordered edges meet the same typed MLP output downstream; the self-loop has weight 2
          frow[f] = diag ? 2.f * (sig * d1[r]) : sig * d1[r] + sig * d2[r];
        }
      }
    }
  }
}

// ---- A5, pairwise graph: layer 1 per node, layer 2 per unordered pair -------------------------------
// y = W x + bias for 64-wide x and dout = 32*OT outputs, 4 waves per row block, wave w taking output
// tiles [w*OT/4, (w+1)*OT/4): the per-node half of the typed MLP's first layer for all K types at once.
__global__ __launch_bounds__(256) void node_linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          int rows, int OT) {
  const int wave = wave_id();
  const RowBlock rb = row_block(rows, blockIdx.x);
  const int lane = rb.lane, h = rb.h;
  const int per = OT >> 2, o0 = wave * per;
  const f32x4* Wl = reinterpret_cast<const f32x4*>(W) + lane + (size_t)o0 * 8 * kStep;
  WRing ring;
  ring_prime(ring, Wl);
  f32x16 in[2];
  load_rows<2>(x, GN_FEAT, rb.row_ld, h, in);
  f32x16 bnext = load_bias_tile(bias + 32 * o0, h);
  const int ld = 32 * OT;
#pragma unroll 1
  for (int o = 0; o < per; ++o) {
    const bool tail = o == per - 1;
    f32x16 acc = bnext;
    bnext = load_bias_tile(bias + 32 * (o0 + (tail ? o : o + 1)), h);
    const f32x4* cur = Wl + (size_t)o * 8 * kStep;
    mma_tile<2>(cur, tail ? cur : cur + 8 * kStep, ring, in, acc);
    if (rb.live) {
      float* p = y + (size_t)rb.row * ld + 32 * (o0 + o) + 4 * h;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {acc[4 * q + 0], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
        *reinterpret_cast<f32x4*>(p + 8 * q) = v;
      }
    }
  }
}


__global__ __launch_bounds__(256, 2) void agg_mlp_kernel(GroupTable<AggGroup> T) {
  __shared__ float part[4][32][64];  // wpr > 1 only: [wave][register 0..31][lane]
  const int gi = find_group(T, blockIdx.x);
  const gn_agg_group_t G = T.g[gi].a;
  const int wpr = T.g[gi].wpr;
  const int rows = G.rows, K = G.K;
  const int wave = wave_id();
  const int wg = blockIdx.x - T.first_wg[gi];
  const int sub = wave % wpr;                       // which share of the types
  const int blk = wg * (4 / wpr) + wave / wpr;      // which row block
  const bool any_rows = blk * 32 < rows;
  const bool staged = T.g[gi].stage != 0;
  if (wpr == 1 && !any_rows && !staged) return;   // (a staged workgroup keeps all its waves for the barriers)
  const RowBlock rb = row_block(rows, any_rows ? blk : 0);
  const int lane = rb.lane, h = rb.h;
  f32x16 out[2];
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[o][r] = 0.f;
  const float* efrow = G.edge_feat + (size_t)rb.row_ld * K;
  const f32x4* Wl = reinterpret_cast<const f32x4*>(G.W) + lane;
  const float* b1 = G.b1;
  const float* b2 = G.b2;

  int k = sub;
  if (G.A != nullptr && staged) {
    // ---- pair form, staged: every lane needs the pre-activations of ITS two nodes — 16-byte pieces scattered
    // over up to 32 rows per load, which makes the texture-address path, not the matrix cores, the limiter.
    // The 4 row blocks of this workgroup touch at most 3 scenes, i.e. a short run of consecutive node rows: the
    // workgroup copies that run (one type at a time, coalesced, prefetched in registers under the previous
    // type's MFMAs) into LDS, and the lanes pick their rows from there.
    float* stage = &part[0][0][0];
    const int N = G.N, P = G.E;
    const int r0 = wg * 128, r1 = min(rows - 1, r0 + 127);
    const int node0 = (r0 / P) * N;
    const int nodes = (r1 / P + 1) * N - node0;
    const size_t ldA = (size_t)K * 128;
    const f32x4* Ag = reinterpret_cast<const f32x4*>(G.A + (size_t)node0 * ldA);
    const int total4 = nodes * 32;                      // float4 pieces per type
    f32x4 pre[kStageLoads];
    auto fetch = [&](int kk) {
#pragma unroll
      for (int it = 0; it < kStageLoads; ++it) {
        const int idx = min((int)threadIdx.x + it * 256, total4 - 1);
        pre[it] = Ag[(size_t)(idx >> 5) * (ldA / 4) + kk * 32 + (idx & 31)];
      }
    };
    auto commit = [&]() {
#pragma unroll
      for (int it = 0; it < kStageLoads; ++it) {
        const int idx = (int)threadIdx.x + it * 256;
        if (idx < total4) *reinterpret_cast<f32x4*>(stage + (idx >> 5) * kStagePitch + (idx & 31) * 4) = pre[it];
      }
    };
    int i = 0, j = 0;
    {
      const int b = rb.row_ld / P, p = rb.row_ld - b * P;
      gn_pair_decode(p, N, i, j);
      i += b * N - node0;
      j += b * N - node0;
    }
    const float* Si = stage + i * kStagePitch;
    const float* Sj = stage + j * kStagePitch;
    WRing ring;
    ring_prime(ring, Wl);
    fetch(0);
    commit();
    __syncthreads();
    PreTile pa = load_pre(Si, h), pb = load_pre(Sj, h);
    float efk = efrow[0];
    float b2f0 = h == 0 ? b2[lane & 31] : 0.f;
    float b2f1 = h == 0 ? b2[32 + (lane & 31)] : 0.f;
#pragma unroll 1
    for (k = 0; k < K; ++k) {
      const int kc = k + 1 < K ? k + 1 : k;
      fetch(kc);                                        // next type's rows: in flight during this type's MFMAs
      const f32x4* base = Wl + (size_t)k * 32 * kStep;
      const f32x4* base_next = Wl + (size_t)kc * 32 * kStep;
      const float efk_next = efrow[kc];
      const float b2n0 = h == 0 ? b2[kc * 64 + (lane & 31)] : 0.f;
      const float b2n1 = h == 0 ? b2[kc * 64 + 32 + (lane & 31)] : 0.f;
      const float efb = h == 0 ? efk : 0.f;
      out[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b2f0, efb, out[0], 0, 0, 0);
      out[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b2f1, efb, out[1], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x16 hid1[1];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int cidx = 0; cidx < 4; ++cidx)
            hid1[0][4 * q + cidx] = fmaxf(pa.v[q][cidx] + pb.v[q][cidx], 0.f) * efk;
        if (t < 3) {
          pa = load_pre(Si + 32 * (t + 1), h);
          pb = load_pre(Sj + 32 * (t + 1), h);
        }
        const f32x4* cur = base + t * 8 * kStep;
        const f32x4* nxt = t < 3 ? cur + 8 * kStep : base_next;
        mma_tile<1, 0, NoSide, 4>(cur, cur + 4 * kStep, ring, hid1, out[0], NoSide(), nxt);
        mma_tile<1, 4>(cur + 4 * kStep, nxt, ring, hid1, out[1]);
      }
      __syncthreads();                                  // every wave has read type k's rows
      commit();
      __syncthreads();
      pa = load_pre(Si, h);
      pb = load_pre(Sj, h);
      efk = efk_next;
      b2f0 = b2n0;
      b2f1 = b2n1;
    }
    if (!any_rows) return;
  } else if (G.A != nullptr) {
    // ---- pair form: the first layer was applied per node (A = W1 ori + b1/2 for every type) ----------
    // W = per type 4 hidden tiles x (2 output tiles x 4 steps), consumed strictly in order: 8 steps
    // (one ring turn) per hidden tile.
    if (k < K && any_rows) {
      const int N = G.N, P = G.E;
      int i, j;
      {
        const int b = rb.row_ld / P, p = rb.row_ld - b * P;
        gn_pair_decode(p, N, i, j);
        i += b * N;
        j += b * N;
      }
      const size_t ldA = (size_t)K * 128;
      const float* Ai = G.A + (size_t)i * ldA;
      const float* Aj = G.A + (size_t)j * ldA;
      WRing ring;
      ring_prime(ring, Wl + (size_t)k * 32 * kStep);
      PreTile pa = load_pre(Ai + k * 128, h), pb = load_pre(Aj + k * 128, h);
      float efk = efrow[k];
      float b2f0 = h == 0 ? b2[k * 64 + (lane & 31)] : 0.f;
      float b2f1 = h == 0 ? b2[k * 64 + 32 + (lane & 31)] : 0.f;
#pragma unroll 1
      while (k < K) {
        const int kn = k + wpr;
        const int kc = kn < K ? kn : k;
        const f32x4* base = Wl + (size_t)k * 32 * kStep;
        const f32x4* base_next = Wl + (size_t)kc * 32 * kStep;
        const float efk_next = efrow[kc];
        const float b2n0 = h == 0 ? b2[kc * 64 + (lane & 31)] : 0.f;
        const float b2n1 = h == 0 ? b2[kc * 64 + 32 + (lane & 31)] : 0.f;
        const float efb = h == 0 ? efk : 0.f;
        out[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b2f0, efb, out[0], 0, 0, 0);
        out[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b2f1, efb, out[1], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          // hidden tile t of this type from the two nodes' pre-activations (b1k/2 rides in each of them)
          f32x16 hid1[1];
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int cidx = 0; cidx < 4; ++cidx)
              hid1[0][4 * q + cidx] = fmaxf(pa.v[q][cidx] + pb.v[q][cidx], 0.f) * efk;
          // fetch the next tile's pre-activations (next type's tile 0 after t == 3) under this tile's MFMAs
          const int off = t < 3 ? k * 128 + 32 * (t + 1) : kc * 128;
          pa = load_pre(Ai + off, h);
          pb = load_pre(Aj + off, h);
          const f32x4* cur = base + t * 8 * kStep;
          const f32x4* nxt = t < 3 ? cur + 8 * kStep : base_next;
          mma_tile<1, 0, NoSide, 4>(cur, cur + 4 * kStep, ring, hid1, out[0], NoSide(), nxt);
          mma_tile<1, 4>(cur + 4 * kStep, nxt, ring, hid1, out[1]);
        }
        efk = efk_next;
        b2f0 = b2n0;
        b2f1 = b2n1;
        k = kn;
      }
    }
  } else if (k < K && any_rows) {
    f32x16 in[2], hid[4];
    if (G.eo != nullptr)
      load_rows<2>(G.eo, GN_FEAT, rb.row_ld, h, in);
    else
      gather_rows(G, rb.row_ld, h, in);
    WRing ring;
    ring_prime(ring, Wl + (size_t)k * kTypeSteps * kStep);
    f32x16 bnext = load_bias_tile(b1 + k * 128, h);
    float efk = efrow[k];
    // b2k as an MFMA A fragment: lane (i, h=0) carries b2k[32o + i]; paired with B = ef_k on k-index 0
    float b2f0 = h == 0 ? b2[k * 64 + (lane & 31)] : 0.f;
    float b2f1 = h == 0 ? b2[k * 64 + 32 + (lane & 31)] : 0.f;
#pragma unroll 1
    while (k < K) {
      const int kn = k + wpr;
      const int kc = kn < K ? kn : k;  // what the run-ahead loads target (valid memory either way)
      const f32x4* base = Wl + (size_t)k * kTypeSteps * kStep;
      const f32x4* base_next = Wl + (size_t)kc * kTypeSteps * kStep;
      const float efk_next = efrow[kc];
      const float b2n0 = h == 0 ? b2[kc * 64 + (lane & 31)] : 0.f;
      const float b2n1 = h == 0 ? b2[kc * 64 + 32 + (lane & 31)] : 0.f;
      // layer 1: 4 tiles of 8 steps; relu * ef_k of tile o-1 rides in the shadow of tile o's MFMAs
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        hid[o] = bnext;
        bnext = load_bias_tile(o < 3 ? b1 + k * 128 + 32 * (o + 1) : b1 + kc * 128, h);
        mma_tile<2>(base + o * 8 * kStep, base + (o + 1) * 8 * kStep, ring, in, hid[o], [&](int s) {
          if (o > 0 && s == 1) relu_scale16(hid[o > 0 ? o - 1 : 0], efk);
        });
      }
      // layer 2: 2 tiles of 16 steps, accumulated over types (tile 0 touches hid[3] only from step 12 on)
      const float efb = h == 0 ? efk : 0.f;
      out[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b2f0, efb, out[0], 0, 0, 0);
      mma_tile<4>(base + 32 * kStep, base + 48 * kStep, ring, hid, out[0], [&](int s) {
        if (s == 1) relu_scale16(hid[3], efk);
      });
      out[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b2f1, efb, out[1], 0, 0, 0);
      mma_tile<4>(base + 48 * kStep, base_next, ring, hid, out[1]);
      efk = efk_next;
      b2f0 = b2n0;
      b2f1 = b2n1;
      k = kn;
    }
  }
  if (wpr == 1) {
    store_rows<2>(G.feat, GN_FEAT, rb.row, h, rb.live, out);
    return;
  }
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][16 * o + r][lane] = out[o][r];
  __syncthreads();
  // the wpr waves of a row block each finish 32/wpr of its registers
  if (rb.live && any_rows) {
    float* p = G.feat + (size_t)rb.row * GN_FEAT + 4 * h;
    const int w0 = wave - sub;
    const int nreg = 32 / wpr;
    for (int rr = 0; rr < nreg; rr += 4) {
      const int reg0 = sub * nreg + rr;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      for (int j = 0; j < wpr; ++j) {
        v[0] += part[w0 + j][reg0 + 0][lane];
        v[1] += part[w0 + j][reg0 + 1][lane];
        v[2] += part[w0 + j][reg0 + 2][lane];
        v[3] += part[w0 + j][reg0 + 3][lane];
      }
      const int o = reg0 >> 4, q = (reg0 & 15) >> 2;
      *reinterpret_cast<f32x4*>(p + 32 * o + 8 * q) = v;
    }
  }
}

// ---- A6 / generic: y = W1 relu(W0 x + b0) + b1 ----------------------------------------------------
// W = [W0 (dh x din) | W1 (dout x dh)] packed; bias = [b0 (dh) | b1 padded to a multiple of 32].


// Whole form: every wave owns a row block, output tiles streamed.  blockIdx.y = group.
template <int IT, int HT>
__global__ __launch_bounds__(256) void mlp2_kernel(GroupTable<gn_mlp2_group_t> T, int rows, int dout, int ldy, int N,
                                                   float divisor) {
  const int blk = blockIdx.x * 4 + wave_id();
  if (blk * 32 >= rows) return;
  const gn_mlp2_group_t G = T.g[blockIdx.y];
  const RowBlock rb = row_block(rows, blk);
  Chain c;
  chain_begin(c, G.W, G.bias, rb.lane);
  f32x16 in[IT], hid[HT];
  mlp2_rows<IT>(G, rb.row_ld, rb.h, N, divisor, in);
  if (G.in_out != nullptr) store_rows<IT>(G.in_out, 32 * IT, rb.row, rb.h, rb.live, in);     // kept for the backward
  chain_linear<HT, IT, true>(c, in, hid);
  if (G.hid_out != nullptr) store_rows<HT>(G.hid_out, 32 * HT, rb.row, rb.h, rb.live, hid);
  const int OT = (dout + 31) >> 5;
  constexpr int S = 4 * HT;
#pragma unroll 1
  for (int o = 0; o < OT; ++o) {
    const bool tail = o == OT - 1;
    f32x16 acc = c.bnext;
    c.bnext = load_bias_tile(tail ? c.b - 32 : c.b, rb.h);
    mma_tile<HT>(c.w, tail ? c.w : c.w + S * kStep, c.ring, hid, acc);
    c.w += S * kStep;
    c.b += 32;
    if (rb.live) store_out_tile(G.y, rb.row, ldy, dout, o, rb.h, acc);
  }
}

// Split form for few row blocks (fewer waves than SIMDs): the 4 waves of a workgroup share ONE row
// block.  Wave w computes hidden tile w (HT == 4) and the partial products of every output tile with
// that slice of the hidden layer; partial sums meet in LDS.  4x shorter critical path; dout <= 64.
template <int IT>
__global__ __launch_bounds__(256) void mlp2_split_kernel(GroupTable<gn_mlp2_group_t> T, int rows, int dout, int ldy,
                                                         int N, float divisor) {
  constexpr int HT = 4;
  __shared__ float part[4][32][64];
  const gn_mlp2_group_t G = T.g[blockIdx.y];
  const int wave = wave_id();
  const RowBlock rb = row_block(rows, blockIdx.x);
  const int lane = rb.lane, h = rb.h;
  const int OT = (dout + 31) >> 5;  // 1 or 2
  f32x16 in[IT], hid[1], out[2];
  const f32x4* Wl = reinterpret_cast<const f32x4*>(G.W) + lane;
  const f32x4* tile = Wl + (size_t)wave * (4 * IT) * kStep;                 // W0 tile `wave`
  const f32x4* w1 = Wl + (size_t)HT * (4 * IT) * kStep;                     // W1: tile (o, t) at (o*HT + t)*4 steps
  const f32x4* sl0 = w1 + (size_t)(0 * HT + wave) * 4 * kStep;
  const f32x4* sl1 = w1 + (size_t)((OT > 1 ? 1 : 0) * HT + wave) * 4 * kStep;
  WRing ring;
  ring_prime(ring, tile);
  hid[0] = load_bias_tile(G.bias + 32 * wave, h);
  mlp2_rows<IT>(G, rb.row_ld, h, N, divisor, in);
  mma_tile<IT, 0, NoSide, 4>(tile, sl0, ring, in, hid[0], NoSide(), sl1);
  relu16(hid[0]);
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[o][r] = 0.f;
  constexpr int ST = (4 * IT) % kP;  // ring slot after the tile
  mma_tile<1, ST, NoSide, 4>(sl0, sl1, ring, hid, out[0], NoSide(), sl1);
  if (OT > 1) mma_tile<1, (ST + 4) % kP, NoSide, 4>(sl1, sl1, ring, hid, out[1], NoSide(), sl1);
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][16 * o + r][lane] = out[o][r];
  __syncthreads();
  if (!rb.live) return;
  // wave w finishes 4*OT registers: the q = w quarter of every output tile
  const float* b1 = G.bias + 32 * HT;
  for (int o = 0; o < OT; ++o) {
    const int reg0 = 16 * o + 4 * wave;
    const int f = 32 * o + 8 * wave + 4 * h;
    float v[4];
#pragma unroll
    for (int cidx = 0; cidx < 4; ++cidx)
      v[cidx] = b1[f + cidx] + ((part[0][reg0 + cidx][lane] + part[1][reg0 + cidx][lane]) +
                                 (part[2][reg0 + cidx][lane] + part[3][reg0 + cidx][lane]));
    float* p = G.y + (size_t)rb.row * ldy;
    if (((dout | ldy) & 3) == 0) {
      if (f < dout) {
        f32x4 vv = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p + f) = vv;
      }
    } else {
#pragma unroll
      for (int cidx = 0; cidx < 4; ++cidx)
        if (f + cidx < dout) p[f + cidx] = v[cidx];
    }
  }
}

// ---- A3 first half, split form: 4 waves per row block -------------------------------------------------
// Wave w computes hidden tiles 2w, 2w+1 of the 64->256 layer and the partial 256->64 products over
// that quarter of the hidden layer; the partial x' tiles meet in LDS; waves 0 and 1 then finish x'
// (+ bias), store it and apply the 64->64 pq projection (one output tile each).
__global__ __launch_bounds__(256) void node_mlp_split_kernel(GroupTable<gn_node_group_t> T, int rows) {
  __shared__ float part[4][32][64];
  const gn_node_group_t G = T.g[blockIdx.y];
  const int wave = wave_id();
  const RowBlock rb = row_block(rows, blockIdx.x);
  const int lane = rb.lane, h = rb.h;
  const f32x4* Wl = reinterpret_cast<const f32x4*>(G.W) + lane;
  const f32x4* w0 = Wl + (size_t)(2 * wave) * 8 * kStep;                   // W0 tiles 2w, 2w+1 (8 steps each)
  const f32x4* w1 = Wl + (size_t)8 * 8 * kStep;                            // W1 tile (o, t) at (o*8 + t)*4 steps
  const f32x4* w1a = w1 + (size_t)(0 * 8 + 2 * wave) * 4 * kStep;          // (0,2w),(0,2w+1): 8 steps
  const f32x4* w1b = w1 + (size_t)(1 * 8 + 2 * wave) * 4 * kStep;          // (1,2w),(1,2w+1): 8 steps
  const f32x4* wpq = w1 + (size_t)2 * 8 * 4 * kStep + (size_t)(wave & 1) * 8 * kStep;  // Wpq tile w (8 steps)
  const float* b0 = G.bias;
  const float* b1 = G.bias + 256;
  const float* bpq = G.bias + 320;
  WRing ring;
  ring_prime(ring, w0);
  f32x16 in[2], hid[2], o1[2];
  hid[0] = load_bias_tile(b0 + 32 * (2 * wave), h);
  hid[1] = load_bias_tile(b0 + 32 * (2 * wave + 1), h);
  load_rows<2>(G.x, GN_FEAT, rb.row_ld, h, in);
  mma_tile<2>(w0, w0 + 8 * kStep, ring, in, hid[0]);
  mma_tile<2>(w0 + 8 * kStep, w1a, ring, in, hid[1], [&](int s) {
    if (s == 1) relu16(hid[0]);
  });
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) o1[o][r] = 0.f;
  mma_tile<2>(w1a, w1b, ring, hid, o1[0], [&](int s) {
    if (s == 1) relu16(hid[1]);   // hid[1] is first read at step 4
  });
  mma_tile<2>(w1b, wpq, ring, hid, o1[1]);
  if (G.hid_out != nullptr) store_rows<2>(G.hid_out + 64 * wave, 256, rb.row, h, rb.live, hid);   // tiles 2w, 2w+1
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][16 * o + r][lane] = o1[o][r];
  __syncthreads();
  if (wave >= 2) return;
  // full x' in the MFMA operand layout: bias + the four partial sums
  f32x16 xp[2];
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    xp[o] = load_bias_tile(b1 + 32 * o, h);
#pragma unroll
    for (int r = 0; r < 16; ++r)
      xp[o][r] += (part[0][16 * o + r][lane] + part[1][16 * o + r][lane]) +
                  (part[2][16 * o + r][lane] + part[3][16 * o + r][lane]);
  }
  f32x16 pq = load_bias_tile(bpq + 32 * wave, h);
  mma_tile<2>(wpq, wpq, ring, xp, pq);
  if (rb.live) {
    float* px = G.xp + (size_t)rb.row * GN_FEAT + 4 * h + 32 * wave;
    float* pp = G.pq + (size_t)rb.row * GN_FEAT + 4 * h + 32 * wave;
    const f32x16 mine = wave == 0 ? xp[0] : xp[1];   // a select, not a runtime-indexed register array
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = {mine[4 * q + 0], mine[4 * q + 1], mine[4 * q + 2], mine[4 * q + 3]};
      *reinterpret_cast<f32x4*>(px + 8 * q) = v;
      f32x4 u = {pq[4 * q + 0], pq[4 * q + 1], pq[4 * q + 2], pq[4 * q + 3]};
      *reinterpret_cast<f32x4*>(pp + 8 * q) = u;
    }
  }
}


}  // namespace

extern "C" size_t gn_packed_elems(int out_features, int in_features) {
  if (out_features <= 0 || in_features <= 0) return 0;
  return (size_t)((out_features + 31) / 32) * ((in_features + 31) / 32) * kTileFloats;
}

extern "C" int gn_pack_linear_f32(const float* W, float* Wp, int out_features, int in_features, int ld,
                                  int col_offset, gn_stream_t stream) {
  GN_REQUIRE_PTR(W);
  GN_REQUIRE_PTR(Wp);
  if (out_features <= 0 || in_features <= 0 || ld < in_features + col_offset || col_offset < 0) return GN_ERR_SHAPE;
  GN_REQUIRE_ALIGNED(Wp);
  const int OT = (out_features + 31) / 32, IT = (in_features + 31) / 32;
  const size_t total = (size_t)OT * IT * kTileFloats;
  const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  hipLaunchKernelGGL(pack_linear_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, W, Wp, out_features,
                     in_features, ld, col_offset, OT, IT);
  return gn_check_launch();
}

extern "C" int gn_split_bf16_f32(const float* packed, void* out, int n_tiles, int parts, gn_stream_t stream) {
  GN_REQUIRE_PTR(packed);
  GN_REQUIRE_PTR(out);
  if (n_tiles < 1 || parts < 1 || parts > 3) return GN_ERR_SHAPE;
  GN_REQUIRE_ALIGNED(out);
  const long long total = (long long)n_tiles * 1024;
  hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)),
                     dim3(256), 0, (hipStream_t)stream, packed, reinterpret_cast<__bf16*>(out), n_tiles, parts);
  return gn_check_launch();
}

extern "C" int gn_split_bf16_batch_f32(const gn_split_job_t* jobs_dev, int n_jobs, int max_tiles, int parts,
                                       gn_stream_t stream) {
  GN_REQUIRE_PTR(jobs_dev);
  if (n_jobs < 1 || n_jobs > 65535 || max_tiles < 1 || parts < 1 || parts > 3) return GN_ERR_SHAPE;
  const long long total = (long long)max_tiles * 1024;
  const unsigned gx = (unsigned)((total + 255) / 256 < 256 ? (total + 255) / 256 : 256);
  hipLaunchKernelGGL(split_bf16_batch_kernel, dim3(gx, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_dev, parts);
  return gn_check_launch();
}

extern "C" int gn_pack_segments_f32(const gn_pack_seg_t* segs_dev, int n_segs, int max_elems, gn_stream_t stream) {
  GN_REQUIRE_PTR(segs_dev);
  if (n_segs < 1 || n_segs > 65535 || max_elems < 1) return GN_ERR_SHAPE;
  int gx = (max_elems + 255) / 256;
  gx = gx > 16 ? 16 : gx;
  hipLaunchKernelGGL(pack_segments_kernel, dim3(gx, n_segs), dim3(256), 0, (hipStream_t)stream, segs_dev);
  return gn_check_launch();
}

// How many groups of a launch carry the bf16-core image: all (returns 1), none (0), or a mix (-1, an error).
template <typename G, typename F>
static int x_mode(const G* groups, int n, F has) {
  int cnt = 0;
  for (int g = 0; g < n; ++g) cnt += has(groups[g]) ? 1 : 0;
  return cnt == 0 ? 0 : (cnt == n ? 1 : -1);
}

// ---- node stage ----------------------------------------------------------------------------------------------
// largest LDS a scene of the fused affinity tail may ask for: beside the node stage's 36 KiB weight ring two workgroups
// per CU must still fit
constexpr size_t kAffTailLds = 24 * 1024;
template <int P, typename T>
static int node_stage_launch(const gn_node_group_t* groups, int n_groups, int rows, hipStream_t s,
                             const gn_affinity_job_t* job = nullptr) {
  NodeTable Tb{};
  Tb.n = n_groups;
  Tb.rows = rows;
  Tb.wgs_per_group = (rows + 127) / 128;          // 4 row blocks of one group per workgroup
  Tb.chain_wgs = n_groups * Tb.wgs_per_group;
  int a_wgs = 0;
  for (int g = 0; g < n_groups; ++g) {
    const gn_node_group_t& G = groups[g];
    const void* ptrs[] = {G.x, G.Wx, G.bias, G.xp, G.pq};
    for (const void* p : ptrs) GN_CHECK(need(p, true));
    if (P == 1 && G.hid_out != nullptr) return GN_ERR_SHAPE;   // the twins are forward-only
    if (P == 2) GN_CHECK(need(G.Wh, true));
    Tb.a_first[g] = a_wgs;
    if (G.A != nullptr) {
      GN_CHECK(need(G.WAx, true));
      if (P == 2) GN_CHECK(need(G.WAh, true));
      GN_CHECK(need(G.bA, true));
      GN_CHECK(need(G.A, true));
      if (G.KA < 1 || G.KA > GN_MAX_TYPES) return GN_ERR_SHAPE;
      a_wgs += Tb.wgs_per_group * ((4 * G.KA + kATiles - 1) / kATiles);
    }
    Tb.g[g] = G;
  }
  Tb.a_first[n_groups] = a_wgs;
  // sections of equal size: every group's chain, then every (group, chunk) of WA
  const int n_sec = (Tb.chain_wgs + a_wgs) / Tb.wgs_per_group;
  Tb.xs.n = n_sec <= GN_MAX_SECTIONS ? n_sec : 1;
  for (int i = 0; i <= Tb.xs.n; ++i) Tb.xs.first[i] = n_sec <= GN_MAX_SECTIONS ? i * Tb.wgs_per_group : i * (Tb.chain_wgs + a_wgs);
  const int node_grid = gn_xcd_grid(Tb.xs);
  size_t aff_lds = 0;
  if (job != nullptr) {
    // the affinity + top-k of job->B scenes as the launch's tail workgroups (same checks as gn_affinity_topk_*)
    const bool embed = job->extras != nullptr && job->extras->x_raw != nullptr;
    if (embed && sizeof(T) != sizeof(float)) return GN_ERR_SHAPE;
    if (!embed) GN_CHECK(need(job->f, true));
    if (job->B <= 0 || job->N <= 0 || job->D <= 0 || (job->D & 3) || job->D > 1024) return GN_ERR_SHAPE;
    GN_CHECK(fill_scales(Tb.aff_sl, job->H_list, job->k_list, job->n_scales, job->N));
    if (embed) {
      if (job->extras->x_dim <= 0 || !job->extras->M || !job->extras->c || !job->extras->f_contig) return GN_ERR_NULL;
      if (!gn_aligned16(job->extras->c) || !gn_aligned16(job->extras->f_contig)) return GN_ERR_ALIGN;
    }
    aff_lds = affinity_fused_lds(job->N, job->D, embed ? job->extras->x_dim : 0);
    if (aff_lds > kAffTailLds) return GN_ERR_LDS;
    if (job->extras != nullptr) {
      Tb.aff_ex = *job->extras;
      if (Tb.aff_ex.f_out != nullptr && (!gn_aligned16(Tb.aff_ex.f_out) || Tb.aff_ex.f_out_ld < job->D || (Tb.aff_ex.f_out_ld & 3)))
        return GN_ERR_ALIGN;
      Tb.aff_sl.H_cat = Tb.aff_ex.H_cat;
    }
    Tb.node_grid = node_grid;
    Tb.aff_scenes = job->B;
    Tb.aff_N = job->N;
    Tb.aff_D = job->D;
    Tb.aff_f = job->f;
    Tb.aff_corr = job->corr;
  }
  hipLaunchKernelGGL((node_stage_kernel<P, T>), dim3(node_grid + Tb.aff_scenes), dim3(256), aff_lds, s, Tb);
  return gn_check_launch();
}

extern "C" int gn_node_mlp_f32(const gn_node_group_t* groups, int n_groups, int rows, gn_stream_t stream) {
  GN_CHECK(check_groups(groups, n_groups));
  if (rows <= 0) return GN_ERR_SHAPE;
  const int xm = x_mode(groups, n_groups, [](const gn_node_group_t& G) { return G.Wx != nullptr; });
  if (xm < 0) return GN_ERR_SHAPE;
  if (xm == 1) {
    // f16x3 (two fp16 parts, bf16x6 fallback inside the launch) when every group carries the fp16 image too
    const int hm = x_mode(groups, n_groups, [](const gn_node_group_t& G) { return G.Wh != nullptr; });
    if (hm < 0) return GN_ERR_SHAPE;
    return hm ? node_stage_launch<2, float>(groups, n_groups, rows, (hipStream_t)stream)
              : node_stage_launch<3, float>(groups, n_groups, rows, (hipStream_t)stream);
  }
  GroupTable<gn_node_group_t> T{};
  T.n = n_groups;
  for (int g = 0; g < n_groups; ++g) {
    const gn_node_group_t& G = groups[g];
    const void* ptrs[] = {G.x, G.W, G.bias, G.xp, G.pq};
    for (const void* p : ptrs) GN_CHECK(need(p, true));
    if (G.A != nullptr) return GN_ERR_SHAPE;   // the fused per-node layer exists on the bf16-core path only
    T.g[g] = G;
  }
  const int blocks32 = (rows + 31) / 32;
  if ((long long)blocks32 * n_groups <= 1024)   // fewer row blocks than SIMDs: 4 waves per row block
    hipLaunchKernelGGL(node_mlp_split_kernel, dim3(blocks32, n_groups), dim3(256), 0, (hipStream_t)stream, T, rows);
  else
    hipLaunchKernelGGL(node_mlp_kernel, dim3(row_grid(rows), n_groups), dim3(256), 0, (hipStream_t)stream, T, rows);
  return gn_check_launch();
}

extern "C" int gn_node_mlp_bf16(const gn_node_group_t* groups, int n_groups, int rows, gn_stream_t stream) {
  GN_CHECK(check_groups(groups, n_groups));
  if (rows <= 0) return GN_ERR_SHAPE;
  return node_stage_launch<1, __bf16>(groups, n_groups, rows, (hipStream_t)stream);
}

// The node stage with the fused affinity + top-k launch as its tail workgroups (bf16-/fp16-core kernels only: every group
// must carry its `Wx` image).
extern "C" int gn_node_mlp_affinity_f32(const gn_node_group_t* groups, int n_groups, int rows, const gn_affinity_job_t* job,
                                        gn_stream_t stream) {
  GN_CHECK(check_groups(groups, n_groups));
  if (rows <= 0 || job == nullptr) return GN_ERR_SHAPE;
  if (x_mode(groups, n_groups, [](const gn_node_group_t& G) { return G.Wx != nullptr; }) != 1) return GN_ERR_SHAPE;
  const int hm = x_mode(groups, n_groups, [](const gn_node_group_t& G) { return G.Wh != nullptr; });
  if (hm < 0) return GN_ERR_SHAPE;
  return hm ? node_stage_launch<2, float>(groups, n_groups, rows, (hipStream_t)stream, job)
            : node_stage_launch<3, float>(groups, n_groups, rows, (hipStream_t)stream, job);
}
extern "C" int gn_node_mlp_affinity_bf16(const gn_node_group_t* groups, int n_groups, int rows, const gn_affinity_job_t* job,
                                         gn_stream_t stream) {
  GN_CHECK(check_groups(groups, n_groups));
  if (rows <= 0 || job == nullptr) return GN_ERR_SHAPE;
  return node_stage_launch<1, __bf16>(groups, n_groups, rows, (hipStream_t)stream, job);
}
extern "C" size_t gn_affinity_tail_lds_limit(void) { return kAffTailLds; }

// Row-block pairs from which the bf16-storage edge / aggregation launches run two row blocks per wave (the chip must
// still be filled: >= 2048 waves).  GN_RB2_MIN_PAIRS is a TEST knob: the parity suite lowers it so that the launcher's
// own choice falls on those kernels at sizes the CPU oracle can follow (tests/test_bf16_gpu.py).
static long long rb2_min_pairs() {
  if (const char* e = getenv("GN_RB2_MIN_PAIRS")) {
    const long long v = atoll(e);
    if (v > 0) return v;
  }
  return 2048;
}

// ---- edge MLP --------------------------------------------------------------------------------------------------
static int edge_launch(const gn_edge_group_t* groups, int n_groups, float tau, unsigned long long seed,
                       const unsigned long long* offset_dev, hipStream_t stream, bool twin) {
  GN_CHECK(check_groups(groups, n_groups));
  if (!(tau > 0.f)) return GN_ERR_SHAPE;
  const int xm = x_mode(groups, n_groups, [](const gn_edge_group_t& G) { return G.Wx != nullptr; });
  if (xm < 0 || (twin && xm != 1)) return GN_ERR_SHAPE;
  GroupTable<gn_edge_group_t> T{};
  T.n = n_groups;
  int wg = 0;
  for (int g = 0; g < n_groups; ++g) {
    const gn_edge_group_t& G = groups[g];
    if (G.edges != nullptr) {
      GN_CHECK(need(G.edges, true));
    } else {      // fused node -> edge pooling: bf16-core kernels only, nothing kept for a backward
      if (!xm || G.keep_z1 || G.keep_z || G.keep_dh1 || G.keep_lgf) return GN_ERR_NULL;
      GN_CHECK(need(G.xp, true));
      GN_CHECK(need(G.pq, true));
      GN_CHECK(need(G.w2, true));
      GN_CHECK(need(G.b2, false));
      if (G.pool_N <= 0) return GN_ERR_SHAPE;
      if (G.pool_H == nullptr) {
        const long long per = G.sym_N > 0 ? gn_pair_count(G.pool_N) : (long long)G.pool_N * G.pool_N;
        if ((G.sym_N > 0 && G.sym_N != G.pool_N) || G.rows % per != 0) return GN_ERR_SHAPE;
      } else {
        if (G.sym_N != 0 || G.pool_N > 16 || G.pool_E <= 0 || G.rows % G.pool_E != 0) return GN_ERR_SHAPE;
      }
    }
    GN_CHECK(need(xm ? G.Wx : (const void*)G.W, true));
    GN_CHECK(need(G.bias, true));
    GN_CHECK(need(G.edge_feat, false));
    if (G.sym_N == 0) GN_CHECK(need(G.dist, false));
    if (G.rows <= 0 || G.K < 1 || G.K > 15 || G.sym_N < 0) return GN_ERR_SHAPE;
    if (G.sym_N > 0 && G.rows % gn_pair_count(G.sym_N) != 0) return GN_ERR_SHAPE;
    if (twin && (G.keep_z1 || G.keep_z || G.keep_dh1 || G.keep_lgf)) return GN_ERR_SHAPE;
    T.g[g] = G;
    T.first_wg[g] = wg;
    wg += row_grid(G.rows);
  }
  T.first_wg[n_groups] = wg;
  // LDS for the staged node rows of the fused pairwise pooling (the largest any group wants, at most 48 KiB; a group
  // that would need more pools straight from L2).  GN_POOL_STAGE = 0 switches the stage off.
  const bool no_pool_stage = getenv("GN_POOL_STAGE") != nullptr && atoi(getenv("GN_POOL_STAGE")) == 0;   // (per call: tests toggle it)
  auto pool_bytes_for = [&](int wg_rows) {
    size_t need = 0;
    for (int g = 0; g < n_groups && !no_pool_stage; ++g) {
      const gn_edge_group_t& G = groups[g];
      if (G.edges != nullptr || G.pool_H != nullptr || G.sym_N <= 0) continue;
      const int nodes = pool_stage_nodes(wg_rows, gn_pair_count(G.pool_N), G.pool_N);
      const size_t b = twin ? PoolStage<__bf16>::bytes(nodes) : PoolStage<float>::bytes(nodes);
      if (b <= 48 * 1024 && b > need) need = b;
    }
    return need;
  };
  if (twin) {
    // a large launch: two row blocks per wave (edge_rb2_kernel); GN_EDGE_RB2 = 0 / 1 forces the choice (parity tests)
    long long pairs = 0;
    for (int g = 0; g < n_groups; ++g) pairs += ((groups[g].rows + 31) / 32 + 1) / 2;
    bool rb2 = pairs >= rb2_min_pairs();
    if (const char* e = getenv("GN_EDGE_RB2")) rb2 = atoi(e) != 0;
    if (rb2) {
      wg = 0;
      for (int g = 0; g < n_groups; ++g) {
        T.first_wg[g] = wg;
        wg += ((groups[g].rows + 31) / 32 + 7) / 8;
      }
      T.first_wg[n_groups] = wg;
      const size_t pb = pool_bytes_for(256);
      hipLaunchKernelGGL((edge_rb2_kernel<__bf16>), dim3(table_xcd_grid(T)), dim3(256), pb, stream, T, tau, seed, offset_dev,
                         (int)pb);
      return gn_check_launch();
    }
    const size_t pb = pool_bytes_for(128);
    hipLaunchKernelGGL((edge_x_kernel<1, __bf16>), dim3(table_xcd_grid(T)), dim3(256), pb, stream, T, tau, seed, offset_dev,
                       (int)pb);
  }
  else if (xm) {
    const int hm = x_mode(groups, n_groups, [](const gn_edge_group_t& G) { return G.Wh != nullptr; });
    if (hm < 0) return GN_ERR_SHAPE;
    const size_t pb = pool_bytes_for(128);
    if (hm)
      hipLaunchKernelGGL((edge_x_kernel<2, float>), dim3(table_xcd_grid(T)), dim3(256), pb, stream, T, tau, seed, offset_dev,
                         (int)pb);
    else
      hipLaunchKernelGGL((edge_x_kernel<3, float>), dim3(table_xcd_grid(T)), dim3(256), pb, stream, T, tau, seed, offset_dev,
                         (int)pb);
  }
  else
    hipLaunchKernelGGL(edge_mlp_gumbel_kernel, dim3(wg), dim3(256), 0, stream, T, tau, seed, offset_dev);
  return gn_check_launch();
}
extern "C" int gn_edge_mlp_gumbel_f32(const gn_edge_group_t* groups, int n_groups, float tau,
                                      unsigned long long seed, const unsigned long long* offset_dev,
                                      gn_stream_t stream) {
  return edge_launch(groups, n_groups, tau, seed, offset_dev, (hipStream_t)stream, false);
}
extern "C" int gn_edge_mlp_gumbel_bf16(const gn_edge_group_t* groups, int n_groups, float tau,
                                       unsigned long long seed, const unsigned long long* offset_dev,
                                       gn_stream_t stream) {
  return edge_launch(groups, n_groups, tau, seed, offset_dev, (hipStream_t)stream, true);
}

// ---- typed aggregation MLP ---------------------------------------------------------------------------------------
static int agg_launch(const gn_agg_group_t* groups_in, int n_groups_in, hipStream_t stream, bool twin) {
  GN_CHECK(check_groups(groups_in, n_groups_in));
  // twins: scene-form groups (node form without A) run in their own kernel (agg_scene_kernel: VALU-bound, compiled for
  // more waves per SIMD than the matrix-core kernels), ahead of the launch of the remaining groups
  gn_agg_group_t rest[GN_MAX_GROUPS];
  int n_groups = 0;
  for (int g = 0; g < n_groups_in; ++g) {
    const gn_agg_group_t& G = groups_in[g];
    if (!(G.node_form && G.A == nullptr)) {
      rest[n_groups++] = G;
      continue;
    }
    if (!twin || G.eo != nullptr || G.H != nullptr || !G.sym || G.N <= 0 || G.N > 64 || G.E != gn_pair_count(G.N) ||
        G.rows <= 0 || G.rows % G.E != 0 || G.K < 1 || G.K > GN_MAX_TYPES)
      return GN_ERR_SHAPE;
    GN_CHECK(need(G.ori, true));
    GN_CHECK(need(G.W12x, true));
    GN_CHECK(need(G.b1, true));
    GN_CHECK(need(G.b2, false));
    GN_CHECK(need(G.edge_feat, false));
    GN_CHECK(need(G.feat, true));
    const int B = G.rows / G.E, RBN = (G.N + 31) / 32;
    hipLaunchKernelGGL((agg_scene_kernel<__bf16>), dim3(B * RBN), dim3(256), (size_t)node_scene_lds_floats(G.N) * sizeof(float),
                       stream, G);
    GN_CHECK(gn_check_launch());
  }
  if (n_groups == 0) return GN_OK;
  const gn_agg_group_t* groups = rest;
  const int xm = x_mode(groups, n_groups, [](const gn_agg_group_t& G) {
    return (G.A != nullptr ? G.W2x : G.W12x) != nullptr;
  });
  if (xm < 0 || (twin && xm != 1)) return GN_ERR_SHAPE;
  GroupTable<AggGroup> T{};
  T.n = n_groups;
  int wg = 0;
  int n_fused = 0;
  for (int g = 0; g < n_groups; ++g) {
    const gn_agg_group_t& G = groups[g];
    if (G.A != nullptr) {
      if (twin) return GN_ERR_SHAPE;
      GN_CHECK(need(G.A, true));
      if (G.N <= 0 || G.E != gn_pair_count(G.N) || G.rows % G.E != 0) return GN_ERR_SHAPE;
      // node form: layer 2 per node (bf16-core images only; the type weights of a row block fit the weight ring's LDS)
      if (G.node_form && (!xm || G.N > 16 || G.K > 12)) return GN_ERR_SHAPE;
    } else if (G.node_form) {
      return GN_ERR_SHAPE;
    } else if (G.eo != nullptr) {
      GN_CHECK(need(G.eo, true));
    } else {
      GN_CHECK(need(G.ori, true));
      if (G.E <= 0 || G.N <= 0 || G.rows % G.E != 0) return GN_ERR_SHAPE;
      if (G.H == nullptr && G.E != (G.sym ? gn_pair_count(G.N) : G.N * G.N)) return GN_ERR_SHAPE;
      if (G.H != nullptr && G.sym) return GN_ERR_SHAPE;
    }
    GN_CHECK(need(xm ? (G.A != nullptr ? G.W2x : G.W12x) : (const void*)G.W, true));
    if (G.A == nullptr) GN_CHECK(need(G.b1, true));
    if (G.y == nullptr) GN_CHECK(need(G.feat, true));
    GN_CHECK(need(G.edge_feat, false));
    GN_CHECK(need(G.b2, false));
    if (G.rows <= 0 || G.K < 1 || G.K > GN_MAX_TYPES) return GN_ERR_SHAPE;
    const int blocks32 = (G.rows + 31) / 32;
    // waves per row block (measured on MI355X at B = 512, N = 11, scripts in DESIGN.md): a group that
    // alone covers most of the 1024 SIMDs runs one wave per block; a mid-sized group halves its units
    // (4 waves of a workgroup sit on ONE CU, so 4-way splitting 176..700 blocks stacks two workgroups on
    // some CUs and idles others); only small groups split 4 ways
    int wpr = blocks32 >= 768 ? 1 : (blocks32 >= 128 && G.K >= 2 ? 2 : (G.K >= 4 ? 4 : 1));
    if (const char* e = getenv(G.A != nullptr ? "GN_AGG_WPR_PAIR" : "GN_AGG_WPR")) {  // tuning knobs: force 1, 2 or 4
      const int v = atoi(e);
      if (v == 1 || v == 2 || v == 4) wpr = v;
    }
    if (G.node_form) wpr = 1;
    T.g[g].a = G;
    T.g[g].wpr = wpr;
    T.g[g].spw = 0;
    if (G.y != nullptr) {      // fused closing stage
      ++n_fused;
      // (the chain is built for two output tiles: 32 < dout <= 64, the image gn_mlp2's kernels take for that width)
      if (twin || !xm || G.dout <= 32 || G.dout > 64 || G.ldy < G.dout || !(G.divisor != 0.f) || G.N <= 0 || G.N > 16)
        return GN_ERR_SHAPE;
      GN_CHECK(need(G.m2x, true));
      GN_CHECK(need(G.m2bias, true));
      GN_CHECK(need(G.ori, true));
      GN_CHECK(need(G.y, false));
      if (G.A != nullptr) {
        if (!G.node_form) return GN_ERR_SHAPE;
      } else {
        // hyper group: whole scenes per workgroup, at most 64 nodes (two row blocks of the closing chain)
        if (G.eo != nullptr || G.H == nullptr || G.E > 16 || wpr == 1) return GN_ERR_SHAPE;
        // (a one-hyperedge module — scale == N — has so few edge rows that 64 nodes per workgroup would multiply its
        // workgroups sixfold; it takes up to 160 nodes, i.e. five row blocks of the closing chain, per workgroup: the
        // launch is as long as its waves live, and that depends on how many workgroups share a CU — measured)
        const int node_cap = G.E == 1 ? 160 : 64;
        int spw = (128 / wpr) / G.E;
        if (node_cap / G.N < spw) spw = node_cap / G.N;
        if (spw < 1) return GN_ERR_SHAPE;
        T.g[g].spw = spw;
      }
    }
    // pair form with one wave per row block: stage the scenes' node rows in LDS when they fit
    const bool no_stage = getenv("GN_AGG_NO_STAGE") != nullptr;
    T.g[g].stage = (!no_stage && G.A != nullptr && wpr == 1 && !G.node_form &&
                    (127 / G.E + 2) * G.N <= (xm ? kStageMaxNodesX : kStageMaxNodes)) ? 1 : 0;
  }
  // Workgroups are dispatched in index order: give the low indices to the group whose waves run longest
  // (types x layers per wave), so the long waves start first and the short ones fill the tail.
  const bool as_given = getenv("GN_AGG_ORDER_AS_GIVEN") != nullptr;
  if (!as_given) {
    auto cost = [](const AggGroup& a) {
      return a.a.node_form ? 1ll : (long long)a.a.K * (a.a.A != nullptr ? 1 : 2) * 4 / a.wpr;
    };
    for (int i = 1; i < n_groups; ++i)        // insertion sort, stable, n <= GN_MAX_GROUPS
      for (int j = i; j > 0 && cost(T.g[j]) > cost(T.g[j - 1]); --j) {
        const AggGroup tmp = T.g[j];
        T.g[j] = T.g[j - 1];
        T.g[j - 1] = tmp;
      }
  }
  for (int g = 0; g < n_groups; ++g) {
    T.first_wg[g] = wg;
    const gn_agg_group_t& a = T.g[g].a;
    wg += a.node_form ? (a.rows / a.E * a.N + 31) / 32      // one workgroup per 32-NODE row block
          : T.g[g].spw > 0 ? (a.rows / a.E + T.g[g].spw - 1) / T.g[g].spw      // fused closing stage: whole scenes
                           : ((a.rows + 31) / 32 * T.g[g].wpr + 3) / 4;
  }
  T.first_wg[n_groups] = wg;
  if (n_fused != 0 && n_fused != n_groups) return GN_ERR_SHAPE;      // every group of a launch or none
  // bf16 storage, a large launch: two row blocks per wave (agg_rb2_kernel)
  if (twin) {
    // (pairs of row blocks must still fill the chip: >= 2048 waves in all.  GN_AGG_RB2 = 0 / 1 forces the choice —
    // the parity tests run small cases through both kernels)
    long long pairs = 0;
    for (int g = 0; g < n_groups; ++g) pairs += ((groups[g].rows + 31) / 32 + 1) / 2;
    bool rb2 = pairs >= rb2_min_pairs();
    if (const char* e = getenv("GN_AGG_RB2")) rb2 = atoi(e) != 0;
    if (rb2) {
      wg = 0;
      for (int g = 0; g < n_groups; ++g) {
        T.first_wg[g] = wg;
        wg += ((T.g[g].a.rows + 31) / 32 + 7) / 8;
      }
      T.first_wg[n_groups] = wg;
      // LDS for the staged ori rows of the pairwise gather (GN_POOL_STAGE = 0 switches it off)
      const bool no_stage = getenv("GN_POOL_STAGE") != nullptr && atoi(getenv("GN_POOL_STAGE")) == 0;   // (per call)
      size_t sb = 0;
      for (int g = 0; g < n_groups && !no_stage; ++g) {
        const gn_agg_group_t& a = T.g[g].a;
        if (a.eo != nullptr || a.H != nullptr || !a.sym) continue;
        const size_t b = (size_t)pool_stage_nodes(256, a.E, a.N) * PoolStage<__bf16>::kPitch * sizeof(__bf16);
        if (b <= 48 * 1024 && b > sb) sb = b;
      }
      hipLaunchKernelGGL((agg_rb2_kernel<__bf16>), dim3(table_xcd_grid(T)), dim3(256), sb, stream, T, (int)sb);
      return gn_check_launch();
    }
  }
  // fused hyper gather in line layout (bf16-core kernels; GN_AGG_LINES = 0 keeps the per-lane gather): needs the LDS too
  const bool no_lines = getenv("GN_AGG_LINES") != nullptr && atoi(getenv("GN_AGG_LINES")) == 0;
  // ... or, when the scenes of a workgroup's rows fit the LDS, from their ori rows staged there (GN_AGG_HSTAGE = 0
  // keeps the line-layout gather): lines = 2
  const bool no_hstage = getenv("GN_AGG_HSTAGE") != nullptr && atoi(getenv("GN_AGG_HSTAGE")) == 0;
  size_t stage_need = 0;
  for (int g = 0; g < n_groups; ++g) {
    const gn_agg_group_t& a = T.g[g].a;
    T.g[g].lines = (xm && !no_lines && a.A == nullptr && a.eo == nullptr && a.H != nullptr && a.N <= 64) ? 1 : 0;
    if (T.g[g].lines && !no_hstage) {
      const int nodes = T.g[g].spw > 0 ? T.g[g].spw * a.N : pool_stage_nodes(128 / T.g[g].wpr, a.E, a.N);
      const size_t b = (size_t)nodes * (twin ? PoolStage<__bf16>::kPitch * sizeof(__bf16) : PoolStage<float>::kPitch * sizeof(float));
      if (b <= 44 * 1024) {
        T.g[g].lines = 2;
        stage_need = b > stage_need ? b : stage_need;
      }
    }
  }
  for (int g = 0; g < n_groups; ++g)      // the fused closing stage of a hyper group reads its scenes' rows from the stage
    if (T.g[g].spw > 0 && T.g[g].lines != 2) return GN_ERR_SHAPE;
  bool need_part = false;       // LDS for partial sums (wpr > 1), the staged node rows or the line-layout gather
  for (int g = 0; g < n_groups; ++g) {
    need_part = need_part || T.g[g].wpr > 1 || T.g[g].stage != 0 || T.g[g].lines != 0;
    if (T.g[g].a.node_form) {   // second stage buffer of the row block's scenes + the block's type weights / scene form
      const size_t b = (size_t)node_form_lds_floats(T.g[g].a.N, T.g[g].a.K) * sizeof(float);
      need_part = true;
      stage_need = b > stage_need ? b : stage_need;
    }
  }
  const size_t part_bytes = need_part ? (stage_need > kAggPartBytes ? stage_need : (size_t)kAggPartBytes) : 0;
  bool any_pair = false;        // a per-pair form of the pairwise graph in this launch: the instantiation that has them
  for (int g = 0; g < n_groups; ++g) any_pair = any_pair || (groups[g].A != nullptr && !groups[g].node_form);
  if (twin)
    hipLaunchKernelGGL((agg_x_kernel<1, __bf16, false>), dim3(table_xcd_grid(T)), dim3(256), part_bytes, stream, T);
  else if (xm) {
    const int hm = x_mode(groups, n_groups, [](const gn_agg_group_t& G) {
      return (G.A != nullptr ? G.W2h : G.W12h) != nullptr;
    });
    if (hm < 0) return GN_ERR_SHAPE;
    const dim3 grid(table_xcd_grid(T));
    if (hm && any_pair)
      hipLaunchKernelGGL((agg_x_kernel<2, float, true>), grid, dim3(256), part_bytes, stream, T);
    else if (hm)
      hipLaunchKernelGGL((agg_x_kernel<2, float, false>), grid, dim3(256), part_bytes, stream, T);
    else if (any_pair)
      hipLaunchKernelGGL((agg_x_kernel<3, float, true>), grid, dim3(256), part_bytes, stream, T);
    else
      hipLaunchKernelGGL((agg_x_kernel<3, float, false>), grid, dim3(256), part_bytes, stream, T);
  } else
    hipLaunchKernelGGL(agg_mlp_kernel, dim3(wg), dim3(256), 0, stream, T);
  return gn_check_launch();
}
extern "C" int gn_agg_mlp_f32(const gn_agg_group_t* groups, int n_groups, gn_stream_t stream) {
  return agg_launch(groups, n_groups, (hipStream_t)stream, false);
}
extern "C" int gn_agg_mlp_bf16(const gn_agg_group_t* groups, int n_groups, gn_stream_t stream) {
  return agg_launch(groups, n_groups, (hipStream_t)stream, true);
}

extern "C" int gn_node_linear_f32(const float* x, const float* W, const float* bias, float* y, int rows, int dout,
                                  gn_stream_t stream) {
  const void* ptrs[] = {x, W, bias, y};
  for (const void* p : ptrs) GN_CHECK(need(p, true));
  if (rows <= 0 || dout <= 0 || dout % 128 != 0) return GN_ERR_SHAPE;
  hipLaunchKernelGGL(node_linear_kernel, dim3((rows + 31) / 32), dim3(256), 0, (hipStream_t)stream, x, W, bias, y, rows,
                     dout / 32);
  return gn_check_launch();
}

template <int P, typename T>
static int mlp2_x_launch(GroupTable<gn_mlp2_group_t>& T_, int n_groups, int rows, int din, int dh, int dout, int ldy,
                         int N, float divisor, hipStream_t s) {
  const int OT = (dout + 31) / 32;
  // a small launch: 4 waves per row block (mlp2_xs_kernel; its fused scatter reads every hyperedge of the scene, so
  // hyper groups need E <= 16).  GN_MLP2_XS = 0 / 1 forces the choice (parity tests run both)
  const int blocks32 = (rows + 31) / 32;
  bool xs = (long long)blocks32 * n_groups <= 1536;
  if (const char* e = getenv("GN_MLP2_XS")) xs = atoi(e) != 0;
  for (int g = 0; g < n_groups; ++g)
    if (T_.g[g].x == nullptr && T_.g[g].H != nullptr && T_.g[g].E > 16) xs = false;
  if (xs) {
    for (int g = 0; g <= n_groups; ++g) T_.first_wg[g] = g * blocks32;
    const dim3 grid(table_xcd_grid(T_)), block(256);
#define GN_MLP2XS(IT, HT, OTv) \
  hipLaunchKernelGGL((mlp2_xs_kernel<P, T, IT, HT, OTv>), grid, block, 0, s, T_, rows, dout, ldy, N, divisor)
    if (din == 64 && dh == 128 && OT == 1) GN_MLP2XS(2, 4, 1);
    else if (din == 64 && dh == 128) GN_MLP2XS(2, 4, 2);
    else if (din == 128 && dh == 128 && OT == 1) GN_MLP2XS(4, 4, 1);
    else if (din == 128 && dh == 128) GN_MLP2XS(4, 4, 2);
    else if (din == 64 && dh == 256 && OT == 1) GN_MLP2XS(2, 8, 1);
    else if (din == 64 && dh == 256) GN_MLP2XS(2, 8, 2);
    else if (din == 128 && dh == 256 && OT == 1) GN_MLP2XS(4, 8, 1);
    else if (din == 128 && dh == 256) GN_MLP2XS(4, 8, 2);
    else return GN_ERR_SHAPE;
#undef GN_MLP2XS
    return gn_check_launch();
  }
  for (int g = 0; g <= n_groups; ++g) T_.first_wg[g] = g * row_grid(rows);
  const dim3 grid(table_xcd_grid(T_)), block(256);
#define GN_MLP2X(IT, HT, OTv) \
  hipLaunchKernelGGL((mlp2_x_kernel<P, T, IT, HT, OTv>), grid, block, 0, s, T_, rows, dout, ldy, N, divisor)
  if (din == 64 && dh == 128 && OT == 1) GN_MLP2X(2, 4, 1);
  else if (din == 64 && dh == 128) GN_MLP2X(2, 4, 2);
  else if (din == 128 && dh == 128 && OT == 1) GN_MLP2X(4, 4, 1);
  else if (din == 128 && dh == 128) GN_MLP2X(4, 4, 2);
  else if (din == 64 && dh == 256 && OT == 1) GN_MLP2X(2, 8, 1);
  else if (din == 64 && dh == 256) GN_MLP2X(2, 8, 2);
  else if (din == 128 && dh == 256 && OT == 1) GN_MLP2X(4, 8, 1);
  else if (din == 128 && dh == 256) GN_MLP2X(4, 8, 2);
  else return GN_ERR_SHAPE;
#undef GN_MLP2X
  return gn_check_launch();
}

static int mlp2_launch(const gn_mlp2_group_t* groups, int n_groups, int rows, int din, int dh, int dout, int ldy,
                       int N, float divisor, hipStream_t s, bool twin) {
  GN_CHECK(check_groups(groups, n_groups));
  if (rows <= 0 || dout <= 0 || ldy < dout) return GN_ERR_SHAPE;
  int xm = x_mode(groups, n_groups, [](const gn_mlp2_group_t& G) { return G.Wx != nullptr; });
  if (xm < 0) return GN_ERR_SHAPE;
  if (twin && (xm != 1 || dout > 64)) return GN_ERR_SHAPE;
  if (xm == 1 && dout > 64) {      // the bf16-core kernel keeps every output tile live: wide outputs use the plain stream
    xm = 0;
    for (int g = 0; g < n_groups; ++g)
      if (groups[g].W == nullptr) return GN_ERR_NULL;
  }
  GroupTable<gn_mlp2_group_t> T{};
  T.n = n_groups;
  for (int g = 0; g < n_groups; ++g) {
    const gn_mlp2_group_t& G = groups[g];
    if (G.x != nullptr) {
      GN_CHECK(need(G.x, true));
    } else {
      if (din != 128 || N <= 0 || !(divisor != 0.f) || rows % N != 0 || G.E < 0) return GN_ERR_SHAPE;
      GN_CHECK(need(G.feat, true));
      GN_CHECK(need(G.ori, true));
      if (G.E > 0) {      // (E == 0: feat is H^T feat per node already)
        if (G.H == nullptr && G.E != (G.sym ? gn_pair_count(N) : N * N)) return GN_ERR_SHAPE;
        if (G.H != nullptr && G.sym) return GN_ERR_SHAPE;
      } else if (G.H != nullptr || G.sym) return GN_ERR_SHAPE;
    }
    GN_CHECK(need(xm ? G.Wx : (const void*)G.W, true));
    GN_CHECK(need(G.bias, true));
    GN_CHECK(need(G.y, false));
    if (twin && (G.in_out != nullptr || G.hid_out != nullptr)) return GN_ERR_SHAPE;
    T.g[g] = G;
  }
  if (twin) return mlp2_x_launch<1, __bf16>(T, n_groups, rows, din, dh, dout, ldy, N, divisor, s);
  if (xm) {
    const int hm = x_mode(groups, n_groups, [](const gn_mlp2_group_t& G) { return G.Wh != nullptr; });
    if (hm < 0) return GN_ERR_SHAPE;
    return hm ? mlp2_x_launch<2, float>(T, n_groups, rows, din, dh, dout, ldy, N, divisor, s)
              : mlp2_x_launch<3, float>(T, n_groups, rows, din, dh, dout, ldy, N, divisor, s);
  }
  const dim3 block(256);
  const int blocks32 = (rows + 31) / 32;
  bool fused = false;   // (also set when activations are to be kept: only the whole-chain kernel writes them)
  for (int g = 0; g < n_groups; ++g)
    fused = fused || groups[g].x == nullptr || groups[g].in_out != nullptr || groups[g].hid_out != nullptr;
  int use_split = (dh == 128 && dout <= 64 && (long long)blocks32 * n_groups <= 1024 && (din == 64 || din == 128) &&
                   !fused)   // the fused-scatter prologue would be repeated by all 4 waves of a row block
                      ? 1
                      : 0;
  if (const char* e = getenv("GN_MLP2_SPLIT")) use_split = atoi(e) != 0 && dh == 128 && dout <= 64 && !fused;
  if (use_split) {
    const dim3 grid(blocks32, n_groups);
    if (din == 64)
      hipLaunchKernelGGL((mlp2_split_kernel<2>), grid, block, 0, s, T, rows, dout, ldy, N, divisor);
    else
      hipLaunchKernelGGL((mlp2_split_kernel<4>), grid, block, 0, s, T, rows, dout, ldy, N, divisor);
    return gn_check_launch();
  }
  const dim3 grid(row_grid(rows), n_groups);
  if (din == 64 && dh == 128)
    hipLaunchKernelGGL((mlp2_kernel<2, 4>), grid, block, 0, s, T, rows, dout, ldy, N, divisor);
  else if (din == 64 && dh == 256)
    hipLaunchKernelGGL((mlp2_kernel<2, 8>), grid, block, 0, s, T, rows, dout, ldy, N, divisor);
  else if (din == 128 && dh == 128)
    hipLaunchKernelGGL((mlp2_kernel<4, 4>), grid, block, 0, s, T, rows, dout, ldy, N, divisor);
  else if (din == 128 && dh == 256)
    hipLaunchKernelGGL((mlp2_kernel<4, 8>), grid, block, 0, s, T, rows, dout, ldy, N, divisor);
  else
    return GN_ERR_SHAPE;
  return gn_check_launch();
}
extern "C" int gn_mlp2_f32(const gn_mlp2_group_t* groups, int n_groups, int rows, int din, int dh, int dout, int ldy,
                           int N, float divisor, gn_stream_t stream) {
  return mlp2_launch(groups, n_groups, rows, din, dh, dout, ldy, N, divisor, (hipStream_t)stream, false);
}
extern "C" int gn_mlp2_bf16(const gn_mlp2_group_t* groups, int n_groups, int rows, int din, int dh, int dout, int ldy,
                            int N, float divisor, gn_stream_t stream) {
  return mlp2_launch(groups, n_groups, rows, din, dh, dout, ldy, N, divisor, (hipStream_t)stream, true);
}
