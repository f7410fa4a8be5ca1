// Device helpers shared by the MFMA kernels of the GroupNet MS-HGNN path (gn_mlp_mfma.hip: fp32 matrix cores;
// gn_mlp_bf16.hip: bf16 matrix cores): row-block addressing, the register layout of a 32-row block, fused
// gather / scatter prologues, the Gumbel-softmax epilogue and host-side argument checks.
#pragma once
#include <stdlib.h>

#include "gn_common.hpp"

namespace {

constexpr int kTileFloats = 32 * 32;  // one packed 32x32 weight tile

// feature held by register r of a lane in half h, inside a 32-feature tile
__device__ __forceinline__ constexpr int feat_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// fp32 -> three bf16 parts (round to nearest even each time; the remainders are exact in fp32)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split3(float x, __bf16& p1, __bf16& p2, __bf16& p3) {
  p1 = (__bf16)x;
  const float r1 = x - (float)p1;
  p2 = (__bf16)r1;
  p3 = (__bf16)(r1 - (float)p2);
}

// ---- storage types -----------------------------------------------------------------------------
// Activations live in HBM as fp32 (the *_f32 entry points) or bf16 (the *_bf16 twins, BASELINE config 4);
// registers always hold fp32.  Four consecutive features are one access: 16 bytes (fp32) or 8 bytes (bf16).
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4(const __bf16* p) {
  const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void st4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4(__bf16* p, const f32x4& v) {
  const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};   // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  *reinterpret_cast<bf16x4*>(p) = o;
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const __bf16* p) { return (float)*p; }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(__bf16* p, float v) { *p = (__bf16)v; }

// ---- register-resident building blocks -----------------------------------------------------
template <int IT, typename T>
__device__ __forceinline__ void load_rows(const T* __restrict__ X, int ld, int row, int h, f32x16 (&a)[IT]) {
  const T* p = X + (size_t)row * ld + 4 * h;
#pragma unroll
  for (int t = 0; t < IT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = ld4(p + 32 * t + 8 * q);
      a[t][4 * q + 0] = v[0];
      a[t][4 * q + 1] = v[1];
      a[t][4 * q + 2] = v[2];
      a[t][4 * q + 3] = v[3];
    }
}

template <int OT, typename T>
__device__ __forceinline__ void store_rows(T* __restrict__ Y, int ld, int row, int h, bool live,
                                           const f32x16 (&a)[OT]) {
  if (!live) return;
  T* p = Y + (size_t)row * ld + 4 * h;
#pragma unroll
  for (int o = 0; o < OT; ++o)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = {a[o][4 * q + 0], a[o][4 * q + 1], a[o][4 * q + 2], a[o][4 * q + 3]};
      st4(p + 32 * o + 8 * q, v);
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

__device__ __forceinline__ void relu16(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = fmaxf(a[r], 0.f);
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
  XcdSections xs;     // bf16-core kernels: XCD-aware order of the logical workgroups (sections = groups)
};
template <typename G>
inline int table_xcd_grid(GroupTable<G>& T) {
  T.xs.n = T.n;
  for (int g = 0; g <= T.n; ++g) T.xs.first[g] = T.first_wg[g];
  return gn_xcd_grid(T.xs);
}
template <typename G>
__device__ __forceinline__ int find_group(const GroupTable<G>& t, int wg) {
  int g = 0;
  while (g + 1 < t.n && wg >= t.first_wg[g + 1]) ++g;
  return gn_uniform(g);
}

// Gumbel softmax over the K logits of a row whose features are split over its two lanes (j, h=0/1):
// d[r] = softmax_f((lg_f + g_f) / tau), g = -log(eps - log(u + eps))   (MS_HGNN_batch.py:446-473).
// Registers 4..7 of a lane hold features 8..15: with K <= 8 types nothing there is live and the (uniform) branch skips
// their logarithms and exponentials.  FAST (bf16-storage kernels only): hardware log2 / exp2 based __logf / __expf.
template <bool FAST = false>
__device__ __forceinline__ void gumbel_softmax_row(const f32x16& lg, const float (&u)[8], int K, float tau, int h,
                                                   float (&d)[8]) {
#ifdef GN_DIAG_NO_GUMBEL     // diagnostic builds only (results are wrong)
  for (int r = 0; r < 8; ++r) d[r] = lg[r] * u[r];
  return;
#endif
  const float eps = 1e-10f;  // MS_HGNN_batch.py:446
  float y[8];
  float m = -INFINITY;
  const bool hi = K > 8;
  // FAST: one reciprocal per row instead of a division per element (an IEEE division is ~10 instructions; with 12-16 of
  // them per row the bf16-storage edge kernel spent as many issue slots dividing as on its matrix work)
  const float inv_tau = 1.f / tau;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    if (g == 0 || hi) {
#pragma unroll
      for (int r = 4 * g; r < 4 * g + 4; ++r) {
        const float gg = FAST ? -__logf(eps - __logf(u[r] + eps)) : -logf(eps - logf(u[r] + eps));
        y[r] = FAST ? (lg[r] + gg) * inv_tau : (lg[r] + gg) / tau;
        if (feat_of(r, h) < K) m = fmaxf(m, y[r]);
      }
    } else {
#pragma unroll
      for (int r = 4 * g; r < 4 * g + 4; ++r) y[r] = 0.f;
    }
  }
  m = fmaxf(m, __shfl_xor(m, 32, GN_WAVE));
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    if (g == 0 || hi) {
#pragma unroll
      for (int r = 4 * g; r < 4 * g + 4; ++r) {
        d[r] = (feat_of(r, h) < K) ? (FAST ? __expf(y[r] - m) : expf(y[r] - m)) : 0.f;
        s += d[r];
      }
    } else {
#pragma unroll
      for (int r = 4 * g; r < 4 * g + 4; ++r) d[r] = 0.f;
    }
  }
  s += __shfl_xor(s, 32, GN_WAVE);
  const float inv_s = 1.f / s;
#pragma unroll
  for (int g = 0; g < 2; ++g)
    if (g == 0 || hi) {                     // (the dead half stays 0: 0 / s with s >= 1)
#pragma unroll
      for (int r = 4 * g; r < 4 * g + 4; ++r) d[r] = FAST ? d[r] * inv_s : d[r] / s;
    }
}

// uniforms of this lane's features for ordered row `orow`: from U, or from the Philox stream.  A lane's features come
// in two runs of four consecutive stream positions (registers 0..3 and 4..7): each run lies in at most two Philox
// blocks, which are evaluated once and shared by the run's elements (one block per ELEMENT was 4 evaluations per run).
__device__ __forceinline__ void fetch_uniforms(const float* __restrict__ U, unsigned long long base,
                                               unsigned long long seed, long long orow, int K, int h, float (&u)[8]) {
#ifdef GN_DIAG_NO_PHILOX     // diagnostic builds only (results are wrong): what the edge kernels take without the generator
  for (int r = 0; r < 8; ++r) u[r] = 0.5f;
  return;
#endif
  if (U != nullptr) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int f = feat_of(r, h);
      u[r] = f < K ? U[(size_t)orow * K + f] : 0.5f;
    }
    return;
  }
  {
    // K <= 8: the row's K consecutive stream positions lie in at most three Philox blocks, in two whenever
    // (first position mod 4) + K <= 8 (always for the pairwise module's K = 6: its rows start at even positions).  Then
    // the row's two lanes evaluate ONE block each and swap: a Philox block is ten rounds of quarter-rate 32x32->64
    // multiplies, the most expensive piece of the epilogue, and each lane evaluating the block(s) of its own features
    // made the wave pay two evaluations per row and ordered edge.
    const unsigned long long pos_row = base + (unsigned long long)orow * K;
    const int o = (int)(pos_row & 3ull);
    if (K <= 8 && o + K <= 8) {                             // (same answer in both lanes of a row)
      uint32_t c[4], w[8];
      gn_philox_block((pos_row >> 2) + (unsigned long long)h, seed, c);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t other = (uint32_t)__shfl_xor((int)c[i], 32, GN_WAVE);
        w[i] = h ? other : c[i];
        w[4 + i] = h ? c[i] : other;
      }
      const int k0 = o + 4 * h;                             // word of this lane's first feature
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + j;
        const uint32_t lo = (k & 2) ? ((k & 1) ? w[3] : w[2]) : ((k & 1) ? w[1] : w[0]);
        const uint32_t hi = (k & 2) ? ((k & 1) ? w[7] : w[6]) : ((k & 1) ? w[5] : w[4]);
        u[j] = 4 * h + j < K ? gn_philox_to_uniform((k & 4) ? hi : lo) : 0.5f;
        u[4 + j] = 0.5f;
      }
      return;
    }
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int f0 = 8 * g + 4 * h;                           // feat_of(4 g, h)
#pragma unroll
    for (int j = 0; j < 4; ++j) u[4 * g + j] = 0.5f;
    if (f0 < K) {
      const unsigned long long pos0 = base + (unsigned long long)orow * K + f0;
      const int o = (int)(pos0 & 3ull);
      // words 0..6 of the run's (at most two) blocks as NAMED scalars: held in an array, the select of word o + j is
      // turned into a load at a run-time index by the compiler, i.e. the array lives in scratch memory (32 bytes per
      // lane, stores and dependent loads in the epilogue of every row of a K > 8 module)
      uint32_t c0[4], c1[4] = {0u, 0u, 0u, 0u};
      gn_philox_block(pos0 >> 2, seed, c0);
      if (o + min(K - f0, 4) > 4) gn_philox_block((pos0 >> 2) + 1ull, seed, c1);     // the run crosses into the next block
      const bool o1 = (o & 1) != 0, o2 = (o & 2) != 0;
      // shift by (o & 1), then by (o & 2): word o + j for j = 0..3 in 10 selects
      const uint32_t s0 = o1 ? c0[1] : c0[0], s1 = o1 ? c0[2] : c0[1], s2 = o1 ? c0[3] : c0[2], s3 = o1 ? c1[0] : c0[3],
                     s4 = o1 ? c1[1] : c1[0], s5 = o1 ? c1[2] : c1[1];
      const uint32_t a[4] = {o2 ? s2 : s0, o2 ? s3 : s1, o2 ? s4 : s2, o2 ? s5 : s3};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (f0 + j < K) u[4 * g + j] = gn_philox_to_uniform(a[j]);
    }
  }
}

// ---- fused node -> edge pooling: the rows of the edge MLP formed on the fly (gn_edge_group_t.xp) -------------------
// Lane (row, h) of a 32-row block produces its 32 features of edges[row] exactly as node2edge_kernel writes them
// (MS_HGNN_batch.py:127-141, 359-370; decomposed attention  att[e,n] = w2 . relu(P_n + (H Qn)_e) + b2).  The 32
// attention channels are split over the row's two lanes (16 each, one shuffle per logit), the node rows are read
// straight from L2 (xp / pq of a launch are a few MB).
template <typename T>
__device__ __forceinline__ void load16(const T* __restrict__ p, float (&v)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 x = ld4(p + 4 * q);
    v[4 * q] = x[0], v[4 * q + 1] = x[1], v[4 * q + 2] = x[2], v[4 * q + 3] = x[3];
  }
}
// acc[t][4q + c] += w * x[32 t + 8 q + 4 h + c]
template <typename T>
__device__ __forceinline__ void axpy_row(const T* __restrict__ x, float w, int h, f32x16 (&acc)[2], bool first) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = ld4(x + 32 * t + 8 * q + 4 * h);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[t][4 * q + c] = first ? w * v[c] : fmaf(w, v[c], acc[t][4 * q + c]);
    }
}
template <typename T>
__device__ __forceinline__ void pooled_rows(const gn_edge_group_t& G, int row, int h, f32x16 (&in)[2]) {
  const int N = G.pool_N;
  const T* xp = reinterpret_cast<const T*>(G.xp);
  const T* pq = reinterpret_cast<const T*>(G.pq);
  float w2[16];
  load16(G.w2 + 16 * h, w2);
  const float b2v = *G.b2;
  if (G.pool_H == nullptr) {
    int b, i, j;
    if (G.sym_N > 0) {
      const int Pn = gn_pair_count(N);
      b = row / Pn;
      gn_pair_decode(row - b * Pn, N, i, j);
    } else {
      b = row / (N * N);
      const int e = row - b * N * N;
      i = e / N;
      j = e - i * N;
    }
    const T* pi = pq + ((size_t)b * N + i) * GN_FEAT + 16 * h;
    const T* pj = pq + ((size_t)b * N + j) * GN_FEAT + 16 * h;
    float Pi[16], Qi[16], Pj[16], Qj[16];
    load16(pi, Pi);
    load16(pi + 32, Qi);
    load16(pj, Pj);
    load16(pj + 32, Qj);
    float ai = 0.f, aj = 0.f;
    float wi, wj;
    if (i == j) {
#pragma unroll
      for (int c = 0; c < 16; ++c) ai = fmaf(w2[c], fmaxf(Pi[c] + 2.f * Qi[c], 0.f), ai);
      ai += __shfl_xor(ai, 32, GN_WAVE);
      ai += b2v;
      // H = 2 on the self-loop: v = 2*att, the other N-1 nodes contribute exp(0)
      const float v = 2.f * ai;
      const float mx = N > 1 ? fmaxf(v, 0.f) : v;
      const float ev = expf(v - mx);
      const float sum = ev + gn_nonmember_sum(N - 1, mx);
      wi = ev / sum * 2.f;
      wj = 0.f;
    } else {
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float q = Qi[c] + Qj[c];
        ai = fmaf(w2[c], fmaxf(Pi[c] + q, 0.f), ai);
        aj = fmaf(w2[c], fmaxf(Pj[c] + q, 0.f), aj);
      }
      ai += __shfl_xor(ai, 32, GN_WAVE);
      aj += __shfl_xor(aj, 32, GN_WAVE);
      ai += b2v;
      aj += b2v;
      const float mx = N > 2 ? fmaxf(fmaxf(ai, aj), 0.f) : fmaxf(ai, aj);
      const float ei = expf(ai - mx), ej = expf(aj - mx);
      const float sum = (ei + ej) + gn_nonmember_sum(N - 2, mx);
      wi = ei / sum;
      wj = ej / sum;
    }
    axpy_row(xp + ((size_t)b * N + i) * GN_FEAT, wi, h, in, true);
    axpy_row(xp + ((size_t)b * N + j) * GN_FEAT, wj, h, in, false);
    return;
  }
  // hyper module (N <= 16): members = nodes with H != 0
  constexpr int NMAX = 16;
  const int b = row / G.pool_E;
  const float* Hrow = G.pool_H + (size_t)row * N;
  const T* pqb = pq + (size_t)b * N * GN_FEAT + 16 * h;
  float hv[NMAX];
  float Q[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) Q[c] = 0.f;
  int cnt = 0;
#pragma unroll
  for (int n = 0; n < NMAX; ++n) {
    hv[n] = n < N ? Hrow[n] : 0.f;
    if (hv[n] != 0.f) {
      ++cnt;
      float Qn[16];
      load16(pqb + (size_t)n * GN_FEAT + 32, Qn);
#pragma unroll
      for (int c = 0; c < 16; ++c) Q[c] = fmaf(hv[n], Qn[c], Q[c]);
    }
  }
  float v[NMAX];
  float mx = cnt < N ? 0.f : -INFINITY;
#pragma unroll
  for (int n = 0; n < NMAX; ++n) {
    v[n] = 0.f;
    float t = 0.f;
    if (hv[n] != 0.f) {
      float Pn[16];
      load16(pqb + (size_t)n * GN_FEAT, Pn);
#pragma unroll
      for (int c = 0; c < 16; ++c) t = fmaf(w2[c], fmaxf(Pn[c] + Q[c], 0.f), t);
    }
    t += __shfl_xor(t, 32, GN_WAVE);       // (both lanes of a row take the same branches: H is per row)
    if (hv[n] != 0.f) {
      v[n] = (t + b2v) * hv[n];
      mx = fmaxf(mx, v[n]);
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int n = 0; n < NMAX; ++n)
    if (hv[n] != 0.f) {
      v[n] = expf(v[n] - mx);
      sum += v[n];
    }
  sum += gn_nonmember_sum(N - cnt, mx);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) in[t][r] = 0.f;
#pragma unroll
  for (int n = 0; n < NMAX; ++n)
    if (hv[n] != 0.f) axpy_row(xp + ((size_t)b * N + n) * GN_FEAT, v[n] / sum * hv[n], h, in, false);
}

// ---- the pairwise pooling from node rows staged in LDS -----------------------------------------------------------
// The pair rows of a workgroup belong to a short run of consecutive scenes.  Read per lane from L2 the pooling costs 32
// load instructions per row block, each touching 32-64 different 128-byte lines (the L1 handles one line per clock):
// 1-2 k cycles of look-ups per row block — as long as the block's whole matrix work on bf16 storage.  Here the
// workgroup copies the x' and pq rows of its scenes into LDS once, coalesced (storage type, 16-byte row padding: the
// 16 lanes of a b128 access phase hit different banks), and the pooling reads them from there; same arithmetic, same
// order: results identical to pooled_rows.
template <typename T>
struct PoolStage {
  static constexpr int kPiece = 16 / (int)sizeof(T);        // elements per 16-byte piece
  static constexpr int kPitch = GN_FEAT + kPiece;           // elements per staged row
  static constexpr int kRowPieces = GN_FEAT / kPiece;
  __host__ __device__ static constexpr size_t bytes(int nodes) { return (size_t)2 * nodes * kPitch * sizeof(T); }
};
// scenes a workgroup of `wg_rows` consecutive pair rows can touch, times N
__host__ __device__ inline int pool_stage_nodes(int wg_rows, int Pn, int N) { return ((wg_rows - 1) / Pn + 2) * N; }
// copies the rows of the scenes of pair rows [r0, r1] (all threads of the workgroup; the caller barriers); returns the
// first staged node
template <typename T>
__device__ __forceinline__ int pool_stage_fill(const gn_edge_group_t& G, int r0, int r1, T* __restrict__ s_xp,
                                               T* __restrict__ s_pq) {
  using PS = PoolStage<T>;
  const int N = G.pool_N, Pn = gn_pair_count(N);
  const int b0 = r0 / Pn, b1 = r1 / Pn;
  const int node0 = b0 * N, nodes = (b1 - b0 + 1) * N;
  const T* xp = reinterpret_cast<const T*>(G.xp) + (size_t)node0 * GN_FEAT;
  const T* pq = reinterpret_cast<const T*>(G.pq) + (size_t)node0 * GN_FEAT;
  for (int idx = threadIdx.x; idx < nodes * PS::kRowPieces; idx += blockDim.x) {
    const int r = idx / PS::kRowPieces, c = idx - r * PS::kRowPieces;
    const f32x4 a = *reinterpret_cast<const f32x4*>(xp + (size_t)idx * PS::kPiece);
    const f32x4 b = *reinterpret_cast<const f32x4*>(pq + (size_t)idx * PS::kPiece);
    *reinterpret_cast<f32x4*>(s_xp + r * PS::kPitch + c * PS::kPiece) = a;
    *reinterpret_cast<f32x4*>(s_pq + r * PS::kPitch + c * PS::kPiece) = b;
  }
  return node0;
}
// pooled_rows, pairwise graph in unordered-pair form, from the staged rows
template <typename T>
__device__ __forceinline__ void pooled_rows_staged(const gn_edge_group_t& G, int row, int h, const T* __restrict__ s_xp,
                                                   const T* __restrict__ s_pq, int node0, f32x16 (&in)[2]) {
  using PS = PoolStage<T>;
#ifdef GN_DIAG_NO_POOL       // diagnostic builds only (results are wrong)
  for (int t = 0; t < 2; ++t)
    for (int r = 0; r < 16; ++r) in[t][r] = s_xp[(row & 15) * PS::kPitch + r];
  return;
#endif
  const int N = G.pool_N, Pn = gn_pair_count(N);
  float w2[16];
  load16(G.w2 + 16 * h, w2);
  const float b2v = *G.b2;
  const int b = row / Pn;
  int i, j;
  gn_pair_decode(row - b * Pn, N, i, j);
  const int ri = (b * N + i - node0) * PS::kPitch, rj = (b * N + j - node0) * PS::kPitch;
  const T* pi = s_pq + ri + 16 * h;
  const T* pj = s_pq + rj + 16 * h;
  float Pi[16], Qi[16], Pj[16], Qj[16];
  load16(pi, Pi);
  load16(pi + 32, Qi);
  load16(pj, Pj);
  load16(pj + 32, Qj);
  float ai = 0.f, aj = 0.f;
  float wi, wj;
  if (i == j) {
#pragma unroll
    for (int c = 0; c < 16; ++c) ai = fmaf(w2[c], fmaxf(Pi[c] + 2.f * Qi[c], 0.f), ai);
    ai += __shfl_xor(ai, 32, GN_WAVE);
    ai += b2v;
    const float v = 2.f * ai;
    const float mx = N > 1 ? fmaxf(v, 0.f) : v;
    const float ev = expf(v - mx);
    const float sum = ev + gn_nonmember_sum(N - 1, mx);
    wi = ev / sum * 2.f;
    wj = 0.f;
  } else {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float q = Qi[c] + Qj[c];
      ai = fmaf(w2[c], fmaxf(Pi[c] + q, 0.f), ai);
      aj = fmaf(w2[c], fmaxf(Pj[c] + q, 0.f), aj);
    }
    ai += __shfl_xor(ai, 32, GN_WAVE);
    aj += __shfl_xor(aj, 32, GN_WAVE);
    ai += b2v;
    aj += b2v;
    const float mx = N > 2 ? fmaxf(fmaxf(ai, aj), 0.f) : fmaxf(ai, aj);
    const float ei = expf(ai - mx), ej = expf(aj - mx);
    const float sum = (ei + ej) + gn_nonmember_sum(N - 2, mx);
    wi = ei / sum;
    wj = ej / sum;
  }
  axpy_row(s_xp + ri, wi, h, in, true);
  axpy_row(s_xp + rj, wj, h, in, false);
}

// The 16 pre-activation values lane (j,h) needs of hidden tile t of type k for ONE node: A row + offset.
struct PreTile {
  f32x4 v[4];
};
template <typename T>
__device__ __forceinline__ PreTile load_pre(const T* __restrict__ arow, int h) {
  PreTile p;
#pragma unroll
  for (int q = 0; q < 4; ++q) p.v[q] = ld4(arow + 8 * q + 4 * h);
  return p;
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
// Input rows of the typed MLP formed on the fly (fused gather): lane (j,h) accumulates its 32 features
// of row r = b*E + e from the member nodes' ori rows.
template <typename T>
__device__ __forceinline__ void add_row(const T* __restrict__ src, float w, int h, f32x16 (&a)[2]) {
  const T* p = src + 4 * h;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = ld4(p + 32 * t + 8 * q);
      a[t][4 * q + 0] = fmaf(w, v[0], a[t][4 * q + 0]);
      a[t][4 * q + 1] = fmaf(w, v[1], a[t][4 * q + 1]);
      a[t][4 * q + 2] = fmaf(w, v[2], a[t][4 * q + 2]);
      a[t][4 * q + 3] = fmaf(w, v[3], a[t][4 * q + 3]);
    }
}
// a += sum_m w[m] * rows[m] over the m < count with w[m] != 0, where w[m] = wptr[m * wstride] (a row or a column of an
// incidence matrix) and rows[m] = base + m * GN_FEAT.  Every lane has its own weights.  A loop "load w, branch, load
// row, accumulate" pays one memory latency per member; instead the nonzero pattern is collected first (independent
// loads, 8 at a time, a 64-bit mask per lane) and the members are then fetched four at a time (32 loads in flight);
// lanes with fewer members ride along with weight 0.  count > 64 falls back to the plain loop.
template <typename T>
__device__ __forceinline__ void weighted_rows(const float* __restrict__ wptr, int wstride, int count,
                                              const T* __restrict__ base, int h, f32x16 (&a)[2]) {
  if (count > 64) {
    for (int m = 0; m < count; ++m) {
      const float w = wptr[(size_t)m * wstride];
      if (w != 0.f) add_row(base + (size_t)m * GN_FEAT, w, h, a);
    }
    return;
  }
  unsigned long long mask = 0ull;
  for (int m0 = 0; m0 < count; m0 += 8) {
    float w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = m0 + u < count ? wptr[(size_t)(m0 + u) * wstride] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) mask |= (unsigned long long)(w[u] != 0.f) << (m0 + u);
  }
  while (__any(mask != 0ull)) {
    int m[4];
    float w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool ok = mask != 0ull;
      m[u] = ok ? __builtin_ctzll(mask) : 0;
      mask &= mask - 1ull;                       // (0 stays 0)
      w[u] = ok ? wptr[(size_t)m[u] * wstride] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) add_row(base + (size_t)m[u] * GN_FEAT, w[u], h, a);
  }
}

template <typename T = float>
__device__ __forceinline__ void gather_rows(const gn_agg_group_t& G, int row, int h, f32x16 (&a)[2]) {
  const int E = G.E, N = G.N;
  const int b = row / E, e = row - b * E;
  const T* ob = reinterpret_cast<const T*>(G.ori) + (size_t)b * N * GN_FEAT;
  if (G.H == nullptr) {
    int i, j;
    if (G.sym) {
      gn_pair_decode(e, N, i, j);
    } else {
      i = e / N;
      j = e - i * N;
    }
    load_rows<2>(ob, GN_FEAT, i, h, a);           // ori_i
    add_row(ob + (size_t)j * GN_FEAT, 1.f, h, a);  // + ori_j  (2 ori_i on the diagonal)
  } else {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) a[t][r] = 0.f;
    weighted_rows(G.H + (size_t)row * N, 1, N, ob, h, a);
  }
}

// The same stage for the typed MLP's pairwise gather (eo = ori_i + ori_j, bf16-storage twin): the scenes' ori rows once
// per workgroup, coalesced; a row's two member rows then come from LDS.  Same operations as gather_rows: identical bits.
template <typename T>
__device__ __forceinline__ int ori_stage_fill(const gn_agg_group_t& G, int r0, int r1, T* __restrict__ s_ori) {
  using PS = PoolStage<T>;
  const int N = G.N, E = G.E;
  const int b0 = r0 / E, b1 = r1 / E;
  const int node0 = b0 * N, nodes = (b1 - b0 + 1) * N;
  const T* src = reinterpret_cast<const T*>(G.ori) + (size_t)node0 * GN_FEAT;
  for (int idx = threadIdx.x; idx < nodes * PS::kRowPieces; idx += blockDim.x) {
    const int r = idx / PS::kRowPieces, c = idx - r * PS::kRowPieces;
    *reinterpret_cast<f32x4*>(s_ori + r * PS::kPitch + c * PS::kPiece) =
        *reinterpret_cast<const f32x4*>(src + (size_t)idx * PS::kPiece);
  }
  return node0;
}
template <typename T>
__device__ __forceinline__ void gather_pair_staged(const gn_agg_group_t& G, int row, int h, const T* __restrict__ s_ori,
                                                   int node0, f32x16 (&a)[2]) {
  using PS = PoolStage<T>;
  const int E = G.E, N = G.N;
  const int b = row / E, e = row - b * E;
  int i, j;
  gn_pair_decode(e, N, i, j);
  const T* ob = s_ori + (size_t)(b * N - node0) * PS::kPitch;
  load_rows<2>(ob, PS::kPitch, i, h, a);                 // ori_i
  add_row(ob + (size_t)j * PS::kPitch, 1.f, h, a);       // + ori_j  (2 ori_i on the diagonal)
}

// eo = H ori of a hyper module from the staged rows: the same members in the same (ascending) order and the same fmaf
// per (row, feature) as weighted_rows — identical bits — with the member rows read from LDS instead of one L2 round trip
// per batch of members (the per-lane / line-layout gathers kept a hyper wave of the typed aggregation in its prologue
// for 15 k cycles, a fifth of the launch).  The incidence row (N floats) is requested up front.
template <typename T>
__device__ __forceinline__ void gather_hyper_staged(const gn_agg_group_t& G, int row, int h, const T* __restrict__ s_ori,
                                                    int node0, f32x16 (&a)[2]) {
  using PS = PoolStage<T>;
  constexpr int NMAX = 16;
  const int E = G.E, N = G.N;
  const int b = row / E;
  const T* ob = s_ori + (size_t)(b * N - node0) * PS::kPitch;
  const float* Hrow = G.H + (size_t)row * N;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) a[t][r] = 0.f;
  for (int n0 = 0; n0 < N; n0 += NMAX) {
    float w[NMAX];
#pragma unroll
    for (int u = 0; u < NMAX; ++u) w[u] = n0 + u < N ? Hrow[n0 + u] : 0.f;
#pragma unroll
    for (int u = 0; u < NMAX; ++u)
      if (w[u] != 0.f) add_row(ob + (size_t)(n0 + u) * PS::kPitch, w[u], h, a);
  }
}

__device__ __forceinline__ void relu_scale16(f32x16& a, float w) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = fmaxf(a[r], 0.f) * w;
}

// Input rows of the MLP: read from x, or — fused scatter, IT == 4 — formed on the fly as
// cat(sum_e H[b,e,n] feat[b,e], ori[b,n]) / divisor  (edge_aggregation.forward + edge2node's / N).
template <int IT, typename T = float>
__device__ __forceinline__ void mlp2_rows(const gn_mlp2_group_t& G, int row, int h, int N, float divisor,
                                          f32x16 (&in)[IT]) {
  if (G.x != nullptr) {
    load_rows<IT>(reinterpret_cast<const T*>(G.x), IT * 32, row, h, in);
    return;
  }
  if constexpr (IT == 4) {
    const int E = G.E;
    const int b = row / N, n = row - b * N;
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const T* fb = reinterpret_cast<const T*>(G.feat) + (size_t)b * E * GN_FEAT;
    if (E == 0) {
      // feat already holds H^T feat per node (node form of the typed aggregation)
      load_rows<2>(reinterpret_cast<const T*>(G.feat), GN_FEAT, row, h, acc);
    } else if (G.H != nullptr && E <= 16) {
      // few hyperedges: read every feat row of the scene (they sit in L1/L2: the scene's lanes share them) weighted by
      // H, four rows in flight — a branch per edge would serialise one memory latency per member
      const float* hcol = G.H + (size_t)b * E * N + n;
      int e = 0;
      for (; e + 4 <= E; e += 4) {
        float w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = hcol[(size_t)(e + u) * N];
#pragma unroll
        for (int u = 0; u < 4; ++u) add_row(fb + (size_t)(e + u) * GN_FEAT, w[u], h, acc);
      }
      for (; e < E; ++e) add_row(fb + (size_t)e * GN_FEAT, hcol[(size_t)e * N], h, acc);
    } else if (G.H != nullptr) {
      weighted_rows(G.H + (size_t)b * E * N + n, N, E, fb, h, acc);
    } else if (G.sym) {
      int j = 0;
      for (; j + 4 <= N; j += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) add_row(fb + (size_t)gn_pair_index(n, j + u, N) * GN_FEAT, 1.f, h, acc);
      }
      for (; j < N; ++j) add_row(fb + (size_t)gn_pair_index(n, j, N) * GN_FEAT, 1.f, h, acc);
    } else {
      for (int j = 0; j < N; ++j) {
        add_row(fb + (size_t)(n * N + j) * GN_FEAT, 1.f, h, acc);
        add_row(fb + (size_t)(j * N + n) * GN_FEAT, 1.f, h, acc);
      }
    }
    f32x16 o[2];
    load_rows<2>(reinterpret_cast<const T*>(G.ori), GN_FEAT, row, h, o);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        in[t][r] = acc[t][r] / divisor;
        in[2 + t][r] = o[t][r] / divisor;
      }
  }
}

// one output tile (16 registers of this lane) -> y, honouring dout / ldy that are not multiples of 4
template <typename T>
__device__ __forceinline__ void store_out_tile(T* __restrict__ y, int row, int ldy, int dout, int o, int h,
                                               const f32x16& acc) {
  T* p = y + (size_t)row * ldy;
  if (((dout | ldy) & 3) == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = 32 * o + 8 * q + 4 * h;
      if (f < dout) {
        f32x4 v = {acc[4 * q + 0], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
        st4(p + f, v);
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = 32 * o + feat_of(r, h);
      if (f < dout) st1(p + f, acc[r]);
    }
  }
}

constexpr int kTypeSteps = 64;
struct AggGroup {
  gn_agg_group_t a;
  int wpr;
  int stage;   // pair form, wpr == 1: the workgroup stages the per-node pre-activations of its scenes in LDS
  int lines;   // fused hyper gather: 1 = accumulated in line layout (agg_x_kernel; needs the launch's `part` LDS),
               // 2 = from the scenes' ori rows staged in that LDS (the workgroup's scenes fit)
  int spw;     // fused closing stage (a.y != NULL): scenes per workgroup — its edge rows are whole scenes
};

// eo = H ori of one 32-row block in LINE layout (see scatter_tile_lines in gn_mlp_bf16.hpp for why): lane L = (sub = L / 8,
// piece = L % 8) accumulates the 16-byte piece `piece` of both 128-byte tiles of the edge rows rg * 8 + sub (rg = 0..3),
// so one load instruction covers 8 rows' whole lines — and the rows of one scene read the SAME node row, so an
// instruction touches 2-4 lines instead of 32.  Nodes in ascending order, zero incidences contribute nothing (a select,
// not a multiply by zero): bit-identical to weighted_rows.  scratch: 32 rows x 72 floats, private to the wave.
constexpr int kLineRow = 72, kLineTile = 36;
template <typename T>
__device__ __forceinline__ void gather_rows_lines(const gn_agg_group_t& G, int blk, int rows, int lane,
                                                  float* __restrict__ scratch) {
  const int E = G.E, N = G.N;
  const int sub = lane >> 3, piece = lane & 7;
  unsigned oo[4], ho[4];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const int r = min(blk * 32 + rg * 8 + sub, rows - 1);
    const int b = r / E;
    oo[rg] = (unsigned)b * N * GN_FEAT + 4 * piece;
    ho[rg] = (unsigned)r * N;
  }
  const T* ori = reinterpret_cast<const T*>(G.ori);
  f32x4 acc[4][2];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg)
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[rg][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int NB = 3;                          // nodes per batch (their rows are requested together)
  for (int n0 = 0; n0 < N; n0 += NB) {
    f32x4 v[NB][4][2];
    float w[NB][4];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int n = min(n0 + u, N - 1);
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        w[u][rg] = G.H[ho[rg] + n];
        v[u][rg][0] = ld4(ori + oo[rg] + (size_t)n * GN_FEAT);
        v[u][rg][1] = ld4(ori + oo[rg] + (size_t)n * GN_FEAT + 32);
      }
    }
#pragma unroll
    for (int u = 0; u < NB; ++u)
      if (n0 + u < N) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const float wv = w[u][rg];
          const bool on = wv != 0.f;
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[rg][t][c] = on ? fmaf(wv, v[u][rg][t][c], acc[rg][t][c]) : acc[rg][t][c];
        }
      }
  }
#pragma unroll
  for (int rg = 0; rg < 4; ++rg)
#pragma unroll
    for (int t = 0; t < 2; ++t)
      *reinterpret_cast<f32x4*>(scratch + (rg * 8 + sub) * kLineRow + t * kLineTile + 4 * piece) = acc[rg][t];
}
// ... and in the MFMA layout (same wave: LDS operations of a wave complete in order)
__device__ __forceinline__ void read_rows_lines(const float* __restrict__ scratch, int lane, f32x16 (&in)[2]) {
  const int j = lane & 31, h = lane >> 5;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(scratch + j * kLineRow + t * kLineTile + 8 * q + 4 * h);
      in[t][4 * q + 0] = v[0];
      in[t][4 * q + 1] = v[1];
      in[t][4 * q + 2] = v[2];
      in[t][4 * q + 3] = v[3];
    }
}
constexpr int kStagePitch = 128 + 4;                 // floats per staged node row (one type)
constexpr int kStageFloats = 4 * 32 * 64;            // the LDS agg_mlp_kernel owns (shared with the wpr > 1 partial sums)
constexpr int kStageMaxNodes = kStageFloats / kStagePitch;
constexpr int kStageBuf = 4 * 32 * (64 + 8) / 2;     // agg_x_kernel: floats per buffer of its double-buffered stage
constexpr int kStageMaxNodesX = kStageBuf / kStagePitch;
constexpr int kStageLoadsX = (kStageMaxNodesX * 32 + 255) / 256;
constexpr int kStageLoads = (kStageMaxNodes * 32 + 255) / 256;

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
