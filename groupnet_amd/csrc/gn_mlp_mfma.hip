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
// (one coalesced 1 KiB dwordx4 load per wave feeds four MFMAs).
//
// fp32 MFMA is an exact k-ordered fmaf chain, so results differ from the reference's MKL
// GEMMs only by summation order (~1e-7 relative).
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

__device__ __forceinline__ void bias_init(const float* __restrict__ bias, int h, f32x16& acc) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 8 * q + 4 * h);
    acc[4 * q + 0] = b[0];
    acc[4 * q + 1] = b[1];
    acc[4 * q + 2] = b[2];
    acc[4 * q + 3] = b[3];
  }
}

// acc += W[o-tile, all IT input tiles] . in        (Wp points at tile (o, 0))
template <int IT>
__device__ __forceinline__ void mma_tile(const float* __restrict__ Wp_o, int lane, const f32x16 (&in)[IT],
                                         f32x16& acc) {
  const f32x4* w4 = reinterpret_cast<const f32x4*>(Wp_o) + lane;
#pragma unroll
  for (int t = 0; t < IT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 w = w4[(t * 4 + q) * 64];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], in[t][4 * q + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], in[t][4 * q + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], in[t][4 * q + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3], in[t][4 * q + 3], acc, 0, 0, 0);
    }
}

__device__ __forceinline__ void relu16(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = fmaxf(a[r], 0.f);
}

// out = act(W in + b) for a layer with OT output tiles and IT input tiles
template <int OT, int IT, bool RELU>
__device__ __forceinline__ void linear(const float* __restrict__ Wp, const float* __restrict__ bias, int lane,
                                       const f32x16 (&in)[IT], f32x16 (&out)[OT]) {
  const int h = lane >> 5;
#pragma unroll
  for (int o = 0; o < OT; ++o) {
    bias_init(bias + 32 * o, h, out[o]);
    mma_tile<IT>(Wp + (size_t)o * IT * kTileFloats, lane, in, out[o]);
    if (RELU) relu16(out[o]);
  }
}

struct RowBlock {
  int lane, h, row, row_ld;  // row = this lane's row; row_ld = clamped row used for loads
  bool live;
};
__device__ __forceinline__ RowBlock row_block(int rows) {
  RowBlock rb;
  rb.lane = threadIdx.x & 63;
  rb.h = rb.lane >> 5;
  const int wave = gn_uniform((int)(threadIdx.x >> 6));
  rb.row = (blockIdx.x * (blockDim.x >> 6) + wave) * 32 + (rb.lane & 31);
  rb.live = rb.row < rows;
  rb.row_ld = rb.live ? rb.row : rows - 1;
  return rb;
}

// ---- A3 first half: x' = MLP(64->256->64)(x); pq = x' Wpq^T + bpq -----------------------------
__global__ __launch_bounds__(256) void node_mlp_kernel(const float* __restrict__ x, const float* __restrict__ W0p,
                                                       const float* __restrict__ b0, const float* __restrict__ W1p,
                                                       const float* __restrict__ b1, const float* __restrict__ Wpqp,
                                                       const float* __restrict__ bpq, float* __restrict__ xp,
                                                       float* __restrict__ pq, int rows) {
  const RowBlock rb = row_block(rows);
  if (gn_uniform(rb.row - (rb.lane & 31)) >= rows) return;  // whole wave past the end
  f32x16 in[2], hid[8], o1[2], o2[2];
  load_rows<2>(x, GN_FEAT, rb.row_ld, rb.h, in);
  linear<8, 2, true>(W0p, b0, rb.lane, in, hid);
  linear<2, 8, false>(W1p, b1, rb.lane, hid, o1);
  store_rows<2>(xp, GN_FEAT, rb.row, rb.h, rb.live, o1);
  linear<2, 2, false>(Wpqp, bpq, rb.lane, o1, o2);
  store_rows<2>(pq, GN_FEAT, rb.row, rb.h, rb.live, o2);
}

// ---- A4: z = MLP(64->128->64); [dist|fac] heads; gumbel softmax; sigmoid ------------------------
__global__ __launch_bounds__(256) void edge_mlp_gumbel_kernel(
    const float* __restrict__ edges, const float* __restrict__ U, const float* __restrict__ Wi0p,
    const float* __restrict__ bi0, const float* __restrict__ Wi1p, const float* __restrict__ bi1,
    const float* __restrict__ Wd0p, const float* __restrict__ bd0, const float* __restrict__ Wd1p,
    const float* __restrict__ bd1, float* __restrict__ edge_feat, float* __restrict__ dist, int rows, int K,
    float tau) {
  const RowBlock rb = row_block(rows);
  if (gn_uniform(rb.row - (rb.lane & 31)) >= rows) return;
  f32x16 in[2], h1[4], z[2], h2[8], lg[1];
  load_rows<2>(edges, GN_FEAT, rb.row_ld, rb.h, in);
  linear<4, 2, true>(Wi0p, bi0, rb.lane, in, h1);
  linear<2, 4, false>(Wi1p, bi1, rb.lane, h1, z);
  linear<8, 2, true>(Wd0p, bd0, rb.lane, z, h2);
  linear<1, 8, false>(Wd1p, bd1, rb.lane, h2, lg);

  // Epilogue.  Features 0..K-1 of `lg` are the logits of this lane's row, feature K the factor
  // pre-activation; a row's features are split over its two lanes (j, h=0) and (j, h=1).
  const float eps = 1e-10f;  // MS_HGNN_batch.py:446
  const float* urow = U + (size_t)rb.row_ld * K;
  float y[8], e[8];
  float m = -INFINITY, facv = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int f = feat_of(r, rb.h);
    const bool valid = f < K;
    const float u = valid ? urow[f] : 0.5f;
    const float g = -logf(eps - logf(u + eps));
    y[r] = (lg[0][r] + g) / tau;
    if (valid) m = fmaxf(m, y[r]);
    if (f == K) facv = lg[0][r];
  }
  m = fmaxf(m, __shfl_xor(m, 32, GN_WAVE));
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    e[r] = (feat_of(r, rb.h) < K) ? expf(y[r] - m) : 0.f;
    s += e[r];
  }
  s += __shfl_xor(s, 32, GN_WAVE);
  facv += __shfl_xor(facv, 32, GN_WAVE);  // exactly one of the two lanes holds it, the other has 0
  const float sig = 1.f / (1.f + expf(-facv));
  if (rb.live) {
    float* drow = dist + (size_t)rb.row * K;
    float* frow = edge_feat + (size_t)rb.row * K;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int f = feat_of(r, rb.h);
      if (f < K) {
        const float d = e[r] / s;
        drow[f] = d;
        frow[f] = sig * d;
      }
    }
  }
}

// ---- A5 typed MLP: feat = sum_k ef[:,k] * (W2k relu(W1k eo + b1k) + b2k) --------------------------
// KT = number of edge types when known at compile time (6: pairwise module, 10: hyper module — the two
// get distinct kernel names, which keeps their rocprof rows apart), 0 = runtime K.
template <int KT>
__global__ __launch_bounds__(256) void agg_mlp_kernel(const float* __restrict__ eo, const float* __restrict__ ef,
                                                      const float* __restrict__ W1p, const float* __restrict__ b1,
                                                      const float* __restrict__ W2p, const float* __restrict__ b2,
                                                      float* __restrict__ feat, int rows, int K_rt) {
  const int K = KT > 0 ? KT : K_rt;
  const RowBlock rb = row_block(rows);
  if (gn_uniform(rb.row - (rb.lane & 31)) >= rows) return;
  f32x16 in[2], hid[4], out[2];
  load_rows<2>(eo, GN_FEAT, rb.row_ld, rb.h, in);
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[o][r] = 0.f;
  const float* efrow = ef + (size_t)rb.row_ld * K;
#pragma unroll 1
  for (int k = 0; k < K; ++k) {
    const float w = efrow[k];
    linear<4, 2, true>(W1p + (size_t)k * 8 * kTileFloats, b1 + k * 128, rb.lane, in, hid);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) hid[t][r] *= w;
    const float* W2k = W2p + (size_t)k * 8 * kTileFloats;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      mma_tile<4>(W2k + (size_t)o * 4 * kTileFloats, rb.lane, hid, out[o]);
      // + ef_k * b2k
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(b2 + k * 64 + 32 * o + 8 * q + 4 * rb.h);
        out[o][4 * q + 0] = fmaf(w, b[0], out[o][4 * q + 0]);
        out[o][4 * q + 1] = fmaf(w, b[1], out[o][4 * q + 1]);
        out[o][4 * q + 2] = fmaf(w, b[2], out[o][4 * q + 2]);
        out[o][4 * q + 3] = fmaf(w, b[3], out[o][4 * q + 3]);
      }
    }
  }
  store_rows<2>(feat, GN_FEAT, rb.row, rb.h, rb.live, out);
}

// ---- A6 / generic: y = W1 relu(W0 x + b0) + b1, output tiles streamed -----------------------------
template <int IT, int HT>
__global__ __launch_bounds__(256) void mlp2_kernel(const float* __restrict__ x, const float* __restrict__ W0p,
                                                   const float* __restrict__ b0, const float* __restrict__ W1p,
                                                   const float* __restrict__ b1, float* __restrict__ y, int rows,
                                                   int dout, int ldy) {
  const RowBlock rb = row_block(rows);
  if (gn_uniform(rb.row - (rb.lane & 31)) >= rows) return;
  f32x16 in[IT], hid[HT];
  load_rows<IT>(x, IT * 32, rb.row_ld, rb.h, in);
  linear<HT, IT, true>(W0p, b0, rb.lane, in, hid);
  const int OT = (dout + 31) >> 5;
#pragma unroll 1
  for (int o = 0; o < OT; ++o) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = 32 * o + feat_of(r, rb.h);
      acc[r] = f < dout ? b1[f] : 0.f;
    }
    mma_tile<HT>(W1p + (size_t)o * HT * kTileFloats, rb.lane, hid, acc);
    if (rb.live) {
      float* p = y + (size_t)rb.row * ldy;
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

extern "C" int gn_node_mlp_f32(const float* x, const float* W0p, const float* b0, const float* W1p, const float* b1,
                               const float* Wpqp, const float* bpq, float* xp, float* pq, int rows,
                               gn_stream_t stream) {
  const void* ptrs[] = {x, W0p, b0, W1p, b1, Wpqp, bpq, xp, pq};
  for (const void* p : ptrs) {
    GN_REQUIRE_PTR(p);
    GN_REQUIRE_ALIGNED(p);
  }
  if (rows <= 0) return GN_ERR_SHAPE;
  hipLaunchKernelGGL(node_mlp_kernel, dim3(row_grid(rows)), dim3(256), 0, (hipStream_t)stream, x, W0p, b0, W1p, b1,
                     Wpqp, bpq, xp, pq, rows);
  return gn_check_launch();
}

extern "C" int gn_edge_mlp_gumbel_f32(const float* edges, const float* U, const float* Wi0p, const float* bi0,
                                      const float* Wi1p, const float* bi1, const float* Wd0p, const float* bd0,
                                      const float* Wd1p, const float* bd1, float* edge_feat, float* dist, int rows,
                                      int K, float tau, gn_stream_t stream) {
  const void* ptrs[] = {edges, U, Wi0p, bi0, Wi1p, bi1, Wd0p, bd0, Wd1p, bd1, edge_feat, dist};
  for (const void* p : ptrs) GN_REQUIRE_PTR(p);
  const void* al[] = {edges, Wi0p, bi0, Wi1p, bi1, Wd0p, bd0, Wd1p, bd1};
  for (const void* p : al) GN_REQUIRE_ALIGNED(p);
  if (rows <= 0 || K < 1 || K > 15 || !(tau > 0.f)) return GN_ERR_SHAPE;
  hipLaunchKernelGGL(edge_mlp_gumbel_kernel, dim3(row_grid(rows)), dim3(256), 0, (hipStream_t)stream, edges, U, Wi0p,
                     bi0, Wi1p, bi1, Wd0p, bd0, Wd1p, bd1, edge_feat, dist, rows, K, tau);
  return gn_check_launch();
}

extern "C" int gn_agg_mlp_f32(const float* eo, const float* edge_feat, const float* W1p, const float* b1,
                              const float* W2p, const float* b2, float* feat, int rows, int K, gn_stream_t stream) {
  const void* ptrs[] = {eo, edge_feat, W1p, b1, W2p, b2, feat};
  for (const void* p : ptrs) GN_REQUIRE_PTR(p);
  const void* al[] = {eo, W1p, b1, W2p, b2, feat};
  for (const void* p : al) GN_REQUIRE_ALIGNED(p);
  if (rows <= 0 || K < 1 || K > GN_MAX_TYPES) return GN_ERR_SHAPE;
  const dim3 grid(row_grid(rows)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (K == 6)
    hipLaunchKernelGGL((agg_mlp_kernel<6>), grid, block, 0, s, eo, edge_feat, W1p, b1, W2p, b2, feat, rows, K);
  else if (K == 10)
    hipLaunchKernelGGL((agg_mlp_kernel<10>), grid, block, 0, s, eo, edge_feat, W1p, b1, W2p, b2, feat, rows, K);
  else
    hipLaunchKernelGGL((agg_mlp_kernel<0>), grid, block, 0, s, eo, edge_feat, W1p, b1, W2p, b2, feat, rows, K);
  return gn_check_launch();
}

extern "C" int gn_mlp2_f32(const float* x, const float* W0p, const float* b0, const float* W1p, const float* b1,
                           float* y, int rows, int din, int dh, int dout, int ldy, gn_stream_t stream) {
  const void* ptrs[] = {x, W0p, b0, W1p, b1, y};
  for (const void* p : ptrs) GN_REQUIRE_PTR(p);
  const void* al[] = {x, W0p, b0, W1p, y};
  for (const void* p : al) GN_REQUIRE_ALIGNED(p);
  if (rows <= 0 || dout <= 0 || ldy < dout) return GN_ERR_SHAPE;
  const dim3 grid(row_grid(rows)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (din == 64 && dh == 128)
    hipLaunchKernelGGL((mlp2_kernel<2, 4>), grid, block, 0, s, x, W0p, b0, W1p, b1, y, rows, dout, ldy);
  else if (din == 64 && dh == 256)
    hipLaunchKernelGGL((mlp2_kernel<2, 8>), grid, block, 0, s, x, W0p, b0, W1p, b1, y, rows, dout, ldy);
  else if (din == 128 && dh == 128)
    hipLaunchKernelGGL((mlp2_kernel<4, 4>), grid, block, 0, s, x, W0p, b0, W1p, b1, y, rows, dout, ldy);
  else if (din == 128 && dh == 256)
    hipLaunchKernelGGL((mlp2_kernel<4, 8>), grid, block, 0, s, x, W0p, b0, W1p, b1, y, rows, dout, ldy);
  else
    return GN_ERR_SHAPE;
  return gn_check_launch();
}
