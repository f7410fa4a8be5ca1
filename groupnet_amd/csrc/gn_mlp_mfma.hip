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

#include "gn_common.hpp"

namespace {

constexpr int kTileFloats = 32 * 32;  // one packed 32x32 weight tile

// feature held by register r of a lane in half h, inside a 32-feature tile
__device__ __forceinline__ constexpr int feat_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

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

// ---- register-resident building blocks -----------------------------------------------------
template <int IT>
__device__ __forceinline__ void load_rows(const float* __restrict__ X, int ld, int row, int h, f32x16 (&a)[IT]) {
  const float* p = X + (size_t)row * ld + 4 * h;
#pragma unroll
  for (int t = 0; t < IT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p + 32 * t + 8 * q);
      a[t][4 * q + 0] = v[0];
      a[t][4 * q + 1] = v[1];
      a[t][4 * q + 2] = v[2];
      a[t][4 * q + 3] = v[3];
    }
}

template <int OT>
__device__ __forceinline__ void store_rows(float* __restrict__ Y, int ld, int row, int h, bool live,
                                           const f32x16 (&a)[OT]) {
  if (!live) return;
  float* p = Y + (size_t)row * ld + 4 * h;
#pragma unroll
  for (int o = 0; o < OT; ++o)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = {a[o][4 * q + 0], a[o][4 * q + 1], a[o][4 * q + 2], a[o][4 * q + 3]};
      *reinterpret_cast<f32x4*>(p + 32 * o + 8 * q) = v;
    }
}

// The 16 bias values a lane needs for one 32-feature output tile (its accumulator's initial value).
__device__ __forceinline__ f32x16 load_bias_tile(const float* __restrict__ bias_tile, int h) {
  f32x16 b;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(bias_tile + 8 * q + 4 * h);
    b[4 * q + 0] = v[0];
    b[4 * q + 1] = v[1];
    b[4 * q + 2] = v[2];
    b[4 * q + 3] = v[3];
  }
  return b;
}

constexpr int kP = 8;          // ring depth in steps
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
template <int IT, int START = 0, typename Side = NoSide>
__device__ __forceinline__ void mma_tile(const f32x4* __restrict__ cur, const f32x4* __restrict__ nxt, WRing& ring,
                                         const f32x16 (&in)[IT], f32x16& acc, Side side = Side()) {
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
    ring.s[slot] = (s + kP < S) ? cur[(s + kP) * kStep] : nxt[(s + kP - S) * kStep];
    // hipcc otherwise sinks the run-ahead load down to its use and collapses the ring to depth 1-2
#if !defined(GN_EXP_NO_SCHED_BARRIER)
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
}

__device__ __forceinline__ void relu16(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = fmaxf(a[r], 0.f);
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

struct RowBlock {
  int lane, h, row, row_ld;  // row = this lane's row; row_ld = clamped row used for loads
  bool live;
};
__device__ __forceinline__ RowBlock row_block(int rows, int block_index) {
  RowBlock rb;
  rb.lane = threadIdx.x & 63;
  rb.h = rb.lane >> 5;
  rb.row = block_index * 32 + (rb.lane & 31);
  rb.live = rb.row < rows;
  rb.row_ld = rb.live ? rb.row : rows - 1;
  return rb;
}
__device__ __forceinline__ int wave_id() { return gn_uniform((int)(threadIdx.x >> 6)); }

// ---- group tables (kernel arguments, by value) -------------------------------------------------
// blockIdx -> (group, workgroup inside the group).  Groups with equal work use blockIdx.y; ragged ones a
// prefix table in workgroup units, so a workgroup never straddles two groups and the lookup is scalar.
template <typename G>
struct GroupTable {
  G g[GN_MAX_GROUPS];
  int first_wg[GN_MAX_GROUPS + 1];
  int n;
};
template <typename G>
__device__ __forceinline__ int find_group(const GroupTable<G>& t, int wg) {
  int g = 0;
  while (g + 1 < t.n && wg >= t.first_wg[g + 1]) ++g;
  return gn_uniform(g);
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
  chain_linear<2, 8, false>(c, hid, o1);
  store_rows<2>(G.xp, GN_FEAT, rb.row, rb.h, rb.live, o1);
  chain_linear<2, 2, false>(c, o1, o2, true);
  store_rows<2>(G.pq, GN_FEAT, rb.row, rb.h, rb.live, o2);
}

// ---- A4: z = MLP(64->128->64); [dist|fac] heads; gumbel softmax; sigmoid ------------------------
// W = [Wi0 (128x64) | Wi1 (64x128) | Wd0 (256x64) | Wd1 (32x256)] packed, bias likewise.

// Gumbel softmax over the K logits of a row whose features are split over its two lanes (j, h=0/1):
// d[r] = softmax_f((lg_f + g_f) / tau), g = -log(eps - log(u + eps))   (MS_HGNN_batch.py:446-473).
__device__ __forceinline__ void gumbel_softmax_row(const f32x16& lg, const float (&u)[8], int K, float tau, int h,
                                                   float (&d)[8]) {
  const float eps = 1e-10f;  // MS_HGNN_batch.py:446
  float y[8];
  float m = -INFINITY;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const float g = -logf(eps - logf(u[r] + eps));
    y[r] = (lg[r] + g) / tau;
    if (feat_of(r, h) < K) m = fmaxf(m, y[r]);
  }
  m = fmaxf(m, __shfl_xor(m, 32, GN_WAVE));
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    d[r] = (feat_of(r, h) < K) ? expf(y[r] - m) : 0.f;
    s += d[r];
  }
  s += __shfl_xor(s, 32, GN_WAVE);
#pragma unroll
  for (int r = 0; r < 8; ++r) d[r] = d[r] / s;
}

// uniforms of this lane's features for ordered row `orow`: from U, or from the Philox stream
__device__ __forceinline__ void fetch_uniforms(const float* __restrict__ U, unsigned long long base,
                                               unsigned long long seed, long long orow, int K, int h, float (&u)[8]) {
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int f = feat_of(r, h);
    if (f >= K)
      u[r] = 0.5f;
    else if (U != nullptr)
      u[r] = U[(size_t)orow * K + f];
    else
      u[r] = gn_philox_uniform_at(base + (unsigned long long)orow * K + f, seed);
  }
}

__global__ __launch_bounds__(256) void edge_mlp_gumbel_kernel(GroupTable<gn_edge_group_t> T, float tau,
                                                              unsigned long long seed,
                                                              const unsigned long long* __restrict__ offset_dev) {
  const int gi = find_group(T, blockIdx.x);
  const gn_edge_group_t G = T.g[gi];
  const int rows = G.rows, K = G.K;
  const int blk = (blockIdx.x - T.first_wg[gi]) * 4 + wave_id();
  if (blk * 32 >= rows) return;
  const RowBlock rb = row_block(rows, blk);
  Chain c;
  chain_begin(c, G.W, G.bias, rb.lane);
  f32x16 in[2], h1[4], z[2], h2[8], lg[1];
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
  chain_linear<4, 2, true>(c, in, h1);
  chain_linear<2, 4, false>(c, h1, z);
  chain_linear<8, 2, true>(c, z, h2);
  chain_linear<1, 8, false>(c, h2, lg, true);
  if (G.U == nullptr) {
    fetch_uniforms(nullptr, pbase, seed, o1, K, rb.h, u1);
    if (G.sym_N > 0) fetch_uniforms(nullptr, pbase, seed, o2, K, rb.h, u2);
  }

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

// ---- A5 typed MLP: feat = sum_k ef[:,k] * (W2k relu(W1k eo + b1k) + b2k) --------------------------
// W = for each type k: [W1k (128x64) | W2k (64x128)] packed (64 steps per type); b1 (K,128); b2 (K,64).
// Work shape, chosen per group by the launcher (block-uniform): `wpr` waves share one 32-row block,
// wave w of them takes types w, w+wpr, ... and the partial sums meet in LDS.
//   wpr = 1 : every wave owns a row block and walks all K types (no LDS);
//   wpr = 2 : the pairwise module (K = 6 -> 3 types per wave): twice as many, half as long work units,
//             which is what lets the chip's 1024 SIMDs finish together (one 6-type unit is ~47 us);
//   wpr = 4 : groups with fewer row blocks than SIMDs (the hyper modules at B*N rows): 4x shorter
//             critical path.
constexpr int kTypeSteps = 64;
struct AggGroup {
  gn_agg_group_t a;
  int wpr;
};
__device__ __forceinline__ void relu_scale16(f32x16& a, float w) {
#if defined(GN_EXP_NO_VALU)
  (void)w;
#elif defined(GN_EXP_MED3)
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = __builtin_amdgcn_fmed3f(a[r], 0.f, __builtin_inff()) * w;
#else
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = fmaxf(a[r], 0.f) * w;
#endif
}
__global__ __launch_bounds__(256) void agg_mlp_kernel(GroupTable<AggGroup> T) {
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
  if (wpr == 1 && !any_rows) return;
  const RowBlock rb = row_block(rows, any_rows ? blk : 0);
  const int lane = rb.lane, h = rb.h;
  f32x16 in[2], hid[4], out[2];
  load_rows<2>(G.eo, GN_FEAT, rb.row_ld, h, in);
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[o][r] = 0.f;
  const float* efrow = G.edge_feat + (size_t)rb.row_ld * K;
  const f32x4* Wl = reinterpret_cast<const f32x4*>(G.W) + lane;
  const float* b1 = G.b1;
  const float* b2 = G.b2;

  int k = sub;
  if (k < K && any_rows) {
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

// ---- A6 / generic: y = W1 relu(W0 x + b0) + b1, output tiles streamed -----------------------------
// W = [W0 (dh x din) | W1 (dout x dh)] packed; bias = [b0 (dh) | b1 padded to a multiple of 32].
// blockIdx.y = group.
template <int IT, int HT>
__global__ __launch_bounds__(256) void mlp2_kernel(GroupTable<gn_mlp2_group_t> T, int rows, int dout, int ldy) {
  const int blk = blockIdx.x * 4 + wave_id();
  if (blk * 32 >= rows) return;
  const gn_mlp2_group_t G = T.g[blockIdx.y];
  const RowBlock rb = row_block(rows, blk);
  Chain c;
  chain_begin(c, G.W, G.bias, rb.lane);
  f32x16 in[IT], hid[HT];
  load_rows<IT>(G.x, IT * 32, rb.row_ld, rb.h, in);
  chain_linear<HT, IT, true>(c, in, hid);
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
    if (rb.live) {
      float* p = G.y + (size_t)rb.row * ldy;
      if (((dout | ldy) & 3) == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f = 32 * o + 8 * q + 4 * rb.h;
          if (f < dout) {
            f32x4 v = {acc[4 * q + 0], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            *reinterpret_cast<f32x4*>(p + f) = v;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int f = 32 * o + feat_of(r, rb.h);
          if (f < dout) p[f] = acc[r];
        }
      }
    }
  }
}

inline int row_grid(int rows) { return (rows + 127) / 128; }  // 4 waves x 32 rows per block

inline int check_groups(const void* groups, int n) {
  if (groups == nullptr) return GN_ERR_NULL;
  if (n < 1 || n > GN_MAX_GROUPS) return GN_ERR_SHAPE;
  return GN_OK;
}
#define GN_CHECK(expr)            \
  do {                            \
    const int rc_ = (expr);       \
    if (rc_ != GN_OK) return rc_; \
  } while (0)
inline int need(const void* p, bool aligned) {
  if (p == nullptr) return GN_ERR_NULL;
  if (aligned && !gn_aligned16(p)) return GN_ERR_ALIGN;
  return GN_OK;
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

extern "C" int gn_node_mlp_f32(const gn_node_group_t* groups, int n_groups, int rows, gn_stream_t stream) {
  GN_CHECK(check_groups(groups, n_groups));
  if (rows <= 0) return GN_ERR_SHAPE;
  GroupTable<gn_node_group_t> T{};
  T.n = n_groups;
  for (int g = 0; g < n_groups; ++g) {
    const gn_node_group_t& G = groups[g];
    const void* ptrs[] = {G.x, G.W, G.bias, G.xp, G.pq};
    for (const void* p : ptrs) GN_CHECK(need(p, true));
    T.g[g] = G;
  }
  hipLaunchKernelGGL(node_mlp_kernel, dim3(row_grid(rows), n_groups), dim3(256), 0, (hipStream_t)stream, T, rows);
  return gn_check_launch();
}

extern "C" int gn_edge_mlp_gumbel_f32(const gn_edge_group_t* groups, int n_groups, float tau,
                                      unsigned long long seed, const unsigned long long* offset_dev,
                                      gn_stream_t stream) {
  GN_CHECK(check_groups(groups, n_groups));
  if (!(tau > 0.f)) return GN_ERR_SHAPE;
  GroupTable<gn_edge_group_t> T{};
  T.n = n_groups;
  int wg = 0;
  for (int g = 0; g < n_groups; ++g) {
    const gn_edge_group_t& G = groups[g];
    GN_CHECK(need(G.edges, true));
    GN_CHECK(need(G.W, true));
    GN_CHECK(need(G.bias, true));
    GN_CHECK(need(G.edge_feat, false));
    if (G.sym_N == 0) GN_CHECK(need(G.dist, false));
    if (G.rows <= 0 || G.K < 1 || G.K > 15 || G.sym_N < 0) return GN_ERR_SHAPE;
    if (G.sym_N > 0 && G.rows % gn_pair_count(G.sym_N) != 0) return GN_ERR_SHAPE;
    T.g[g] = G;
    T.first_wg[g] = wg;
    wg += row_grid(G.rows);
  }
  T.first_wg[n_groups] = wg;
  hipLaunchKernelGGL(edge_mlp_gumbel_kernel, dim3(wg), dim3(256), 0, (hipStream_t)stream, T, tau, seed, offset_dev);
  return gn_check_launch();
}

extern "C" int gn_agg_mlp_f32(const gn_agg_group_t* groups, int n_groups, gn_stream_t stream) {
  GN_CHECK(check_groups(groups, n_groups));
  GroupTable<AggGroup> T{};
  T.n = n_groups;
  int wg = 0;
  for (int g = 0; g < n_groups; ++g) {
    const gn_agg_group_t& G = groups[g];
    GN_CHECK(need(G.eo, true));
    GN_CHECK(need(G.W, true));
    GN_CHECK(need(G.b1, true));
    GN_CHECK(need(G.feat, true));
    GN_CHECK(need(G.edge_feat, false));
    GN_CHECK(need(G.b2, false));
    if (G.rows <= 0 || G.K < 1 || G.K > GN_MAX_TYPES) return GN_ERR_SHAPE;
    const int blocks32 = (G.rows + 31) / 32;
    // waves per row block: 4 when the group has fewer row blocks than half the chip's 1024 SIMDs,
    // else 2 when the types split evenly (shorter, more uniform work units), else 1
    int wpr = (blocks32 <= 512 && G.K >= 4) ? 4 : ((G.K % 2 == 0 && G.K >= 4) ? 2 : 1);
    if (const char* e = getenv("GN_AGG_WPR")) {  // tuning knob: force 1, 2 or 4
      const int v = atoi(e);
      if (v == 1 || v == 2 || v == 4) wpr = v;
    }
    T.g[g].a = G;
    T.g[g].wpr = wpr;
    T.first_wg[g] = wg;
    wg += (blocks32 * wpr + 3) / 4;
  }
  T.first_wg[n_groups] = wg;
  hipLaunchKernelGGL(agg_mlp_kernel, dim3(wg), dim3(256), 0, (hipStream_t)stream, T);
  return gn_check_launch();
}

extern "C" int gn_mlp2_f32(const gn_mlp2_group_t* groups, int n_groups, int rows, int din, int dh, int dout, int ldy,
                           gn_stream_t stream) {
  GN_CHECK(check_groups(groups, n_groups));
  if (rows <= 0 || dout <= 0 || ldy < dout) return GN_ERR_SHAPE;
  GroupTable<gn_mlp2_group_t> T{};
  T.n = n_groups;
  for (int g = 0; g < n_groups; ++g) {
    const gn_mlp2_group_t& G = groups[g];
    GN_CHECK(need(G.x, true));
    GN_CHECK(need(G.W, true));
    GN_CHECK(need(G.bias, true));
    GN_CHECK(need(G.y, false));
    T.g[g] = G;
  }
  const dim3 grid(row_grid(rows), n_groups), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (din == 64 && dh == 128)
    hipLaunchKernelGGL((mlp2_kernel<2, 4>), grid, block, 0, s, T, rows, dout, ldy);
  else if (din == 64 && dh == 256)
    hipLaunchKernelGGL((mlp2_kernel<2, 8>), grid, block, 0, s, T, rows, dout, ldy);
  else if (din == 128 && dh == 128)
    hipLaunchKernelGGL((mlp2_kernel<4, 4>), grid, block, 0, s, T, rows, dout, ldy);
  else if (din == 128 && dh == 256)
    hipLaunchKernelGGL((mlp2_kernel<4, 8>), grid, block, 0, s, T, rows, dout, ldy);
  else
    return GN_ERR_SHAPE;
  return gn_check_launch();
}
