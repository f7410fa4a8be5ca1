// Fused cosine affinity + top-k incidence of one scene (A0 + A1: model/GroupNet_nba.py:284-286, model/MS_HGNN_batch.py:372-388)
// as a device function: the stand-alone launch (gn_graph.hip: one workgroup per scene) and the TAIL workgroups of the
// node stage's launch (gn_mlp_bf16.hpp: the first node stage of a forward needs only f, not the incidence, so the two
// are independent; dispatched behind the node stage's own workgroups, the scenes' workgroups fill the CUs its short "A"
// workgroups leave early instead of costing a launch of their own) run the same code.
#pragma once
#include "gn_mlp_common.hpp"

namespace {

struct ScaleList {
  float* H[GN_MAX_SCALES];
  int k[GN_MAX_SCALES];  // clamped to >= 1; k == N marks the single all-ones hyperedge
  int n;
  // optional second destination: the (B, cat_rows, N) concatenation of every H_s (scale s at row cat_off[s])
  float* H_cat;
  int cat_off[GN_MAX_SCALES];
  int cat_rows;
};


// Ranking key of affinity v in column j of its row: key_j > key_c  <=>  beats(v_j, j, v_c, c).  High word: the float's
// bits mapped to an order-preserving unsigned (-0 folded into +0: the two compare equal; every NaN -> the largest key:
// NaN ranks first), low word: ~j (the lower index wins a tie).  One 64-bit compare per (j, c) instead of the
// NaN / greater / equal / index cascade — the ranking was the larger half of this kernel at N = 50.
__device__ __forceinline__ unsigned long long rank_key(float v, int j) {
  const float z = v + 0.f;                                   // -0 -> +0
  const uint32_t b = __float_as_uint(z);
  uint32_t s = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  s = (z != z) ? 0xFFFFFFFFu : s;
  return ((unsigned long long)s << 32) | (uint32_t)(~(uint32_t)j);
}
// the writes of every scale for column c of row i, given its rank
template <typename T>
__device__ __forceinline__ void emit_ranked(int rank, int N, int b, int i, int c, const ScaleList& sl) {
  T* H_cat = reinterpret_cast<T*>(sl.H_cat);   // the concatenation is what the caller returns: storage type T
  for (int s = 0; s < sl.n; ++s) {
    if (sl.k[s] == N) {
      if (i == 0) {
        sl.H[s][(size_t)b * N + c] = 1.f;
        if (H_cat) st1(H_cat + ((size_t)b * sl.cat_rows + sl.cat_off[s]) * N + c, 1.f);
      }
    } else {
      const float v = rank < sl.k[s] ? 1.f : 0.f;
      sl.H[s][((size_t)b * N + i) * N + c] = v;
      if (H_cat) st1(H_cat + ((size_t)b * sl.cat_rows + sl.cat_off[s] + i) * N + c, v);
    }
  }
}

// One workgroup per scene.  f rows are normalised into LDS (stride D+4 floats keeps the
// 16-byte row reads of different rows on different banks); corr is formed once per UNORDERED pair (the dot product
// is symmetric term by term, so corr[j][i] is the same bits), optionally written out, and kept in LDS as ranking keys,
// which are then ranked in place.  Needs N*(D+4)*4 + N*N*8 bytes of LDS.
template <typename T>
__device__ __forceinline__ void affinity_topk_body(const T* __restrict__ f, float* __restrict__ corr, const ScaleList& sl,
                                                   int N, int D, const gn_block_extras_t& ex, int b, float* lds) {
  constexpr int kBlock = 256;
  const int ldq = D + 4;
  float* q = lds;             // N x ldq
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(lds + N * ldq + ((N * ldq) & 1));  // N x N, 8-byte aligned
  const T* fb = f + (size_t)b * N * D;
  const int d4 = D >> 2;
  if (ex.counter != nullptr && b == 0 && threadIdx.x == 0) *ex.counter += ex.counter_add;
  float* xs = reinterpret_cast<float*>(keys + N * N);  // N x x_dim raw inputs (embedding form only)
  if (ex.x_raw != nullptr) {
    const float* xb = ex.x_raw + (size_t)b * N * ex.x_dim;
    for (int idx = threadIdx.x; idx < N * ex.x_dim; idx += kBlock) xs[idx] = xb[idx];
    __syncthreads();
  }
  for (int idx = threadIdx.x; idx < N * d4; idx += kBlock) {
    const int r = idx / d4, cc = idx - r * d4;
    f32x4 v;
    if (ex.x_raw != nullptr) {
      // f = M x + c[agent slot]: the whole embedding front-end is one affine map in eval mode
      v = *reinterpret_cast<const f32x4*>(ex.c + (size_t)r * D + 4 * cc);
      const float* xr = xs + r * ex.x_dim;
      for (int k = 0; k < ex.x_dim; ++k) {
        const float xv = xr[k];
        v[0] = fmaf(ex.M[(size_t)(4 * cc + 0) * ex.x_dim + k], xv, v[0]);
        v[1] = fmaf(ex.M[(size_t)(4 * cc + 1) * ex.x_dim + k], xv, v[1]);
        v[2] = fmaf(ex.M[(size_t)(4 * cc + 2) * ex.x_dim + k], xv, v[2]);
        v[3] = fmaf(ex.M[(size_t)(4 * cc + 3) * ex.x_dim + k], xv, v[3]);
      }
      *reinterpret_cast<f32x4*>(ex.f_contig + ((size_t)b * N + r) * D + 4 * cc) = v;
    } else {
      v = ld4(fb + (size_t)r * D + 4 * cc);
    }
    *reinterpret_cast<f32x4*>(q + r * ldq + 4 * cc) = v;
    if (ex.f_out != nullptr) st4(reinterpret_cast<T*>(ex.f_out) + ((size_t)b * N + r) * ex.f_out_ld + 4 * cc, v);
  }
  __syncthreads();
  // row norms: one wave per row, lanes stride the row
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int r = wave; r < N; r += kBlock / 64) {
    float ss = 0.f;
    for (int d = lane; d < D; d += 64) ss += q[r * ldq + d] * q[r * ldq + d];
    ss = gn_wave_sum(ss);
    const float denom = fmaxf(sqrtf(ss), 1e-12f);  // F.normalize eps, GroupNet_nba.py:284
    for (int d = lane; d < D; d += 64) q[r * ldq + d] = q[r * ldq + d] / denom;
  }
  __syncthreads();
  const int Pn = gn_pair_count(N);
  for (int p = threadIdx.x; p < Pn; p += kBlock) {
    int i, j;
    gn_pair_decode(p, N, i, j);
    const f32x4* a = reinterpret_cast<const f32x4*>(q + i * ldq);
    const f32x4* c = reinterpret_cast<const f32x4*>(q + j * ldq);
    float acc = 0.f;
    for (int d = 0; d < d4; ++d) {
      const f32x4 x = a[d], y = c[d];
      acc = fmaf(x[0], y[0], acc);
      acc = fmaf(x[1], y[1], acc);
      acc = fmaf(x[2], y[2], acc);
      acc = fmaf(x[3], y[3], acc);
    }
    keys[i * N + j] = rank_key(acc, j);
    keys[j * N + i] = rank_key(acc, i);
    if (corr) {
      corr[(size_t)b * N * N + i * N + j] = acc;
      corr[(size_t)b * N * N + j * N + i] = acc;
    }
  }
  if (sl.n == 0) return;
  __syncthreads();
  for (int idx = threadIdx.x; idx < N * N; idx += kBlock) {
    const int i = idx / N, c = idx - i * N;
    const unsigned long long* row = keys + i * N;
    const unsigned long long kc = row[c];
    int rank = 0;
    for (int j = 0; j < N; ++j) rank += row[j] > kc ? 1 : 0;
    emit_ranked<T>(rank, N, b, i, c, sl);
  }
}


// host: the scale table of a launch from the caller's lists
inline int fill_scales(ScaleList& sl, float* const* H_list, const int* k_list, int n_scales, int N) {
  sl = ScaleList{};
  if (n_scales < 0 || n_scales > GN_MAX_SCALES) return GN_ERR_SHAPE;
  if (n_scales > 0 && (H_list == nullptr || k_list == nullptr)) return GN_ERR_NULL;
  sl.n = n_scales;
  for (int s = 0; s < n_scales; ++s) {
    if (H_list[s] == nullptr) return GN_ERR_NULL;
    if (k_list[s] > N) return GN_ERR_K_RANGE;
    sl.H[s] = H_list[s];
    sl.k[s] = k_list[s] == N ? N : (k_list[s] < 1 ? 1 : k_list[s]);
    sl.cat_off[s] = sl.cat_rows;
    sl.cat_rows += k_list[s] == N ? 1 : N;
  }
  return GN_OK;
}


// LDS bytes of one scene's workgroup (rows + 64-bit ranking keys (+ raw inputs of the embedding front-end))
inline size_t affinity_fused_lds(int N, int D, int x_dim) {
  return (size_t)N * (D + 4) * sizeof(float) + 8 + (size_t)N * N * 8 + (size_t)N * x_dim * sizeof(float);
}

}  // namespace
