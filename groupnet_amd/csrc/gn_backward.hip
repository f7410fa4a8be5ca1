// Backward (training) kernels of the GroupNet MS-HGNN path for gfx950 — SURVEY.md §8f rank 2.
//
// The forward is a handful of fused matrix-core kernels; the backward is deliberately built from a few
// GENERIC blocks, correctness first (train_hyper_nba.py:116 back-propagates through these modules, but
// training throughput is not the path's headline):
//   gn_gemm_f32            C = beta*C + op(A) op(B) (+ bias) (relu) (masked by another tensor's sign);
//                          LDS-tiled 64x64x16 VALU SGEMM with optional split-K (atomic) — used for the
//                          re-computation of hidden activations, for input gradients dX = dY W and for
//                          weight gradients dW = dY^T X (K = rows, split over workgroups);
//   gn_colsum_f32          bias gradients db = sum_rows dY;
//   gn_typed_scale/dot     the per-row, per-type scalings of the typed aggregation;
//   gn_gumbel_bwd_f32      back through fac * softmax((logits + g) / tau) and the sigmoid;
//   gn_node2edge_bwd_f32   back through the attention-weighted pooling (one wave per hyperedge, like the
//                          forward; node rows are shared by edges, so their gradients are atomic adds).
// Gather and scatter are each other's adjoints and reuse the forward kernels.
#include "gn_common.hpp"

namespace {

constexpr int kB = 256;
constexpr int TM = 64, TN = 64, TK = 16;

// element (r, c) of op(X): X is (rows x cols) row-major with leading dimension ld; trans reads X^T
__device__ __forceinline__ float at(const float* __restrict__ X, int ld, int trans, int r, int c) {
  return trans ? X[(size_t)c * ld + r] : X[(size_t)r * ld + c];
}

__global__ __launch_bounds__(kB) void gemm_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                  float* __restrict__ C, int M, int N, int K, int lda, int ldb,
                                                  int ldc, int transA, int transB, const float* __restrict__ bias,
                                                  const float* __restrict__ mask, int ldmask, int relu, float alpha,
                                                  float beta, int kchunk) {
  __shared__ float As[TK][TM + 4];
  __shared__ float Bs[TK][TN + 4];
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const int k_lo = blockIdx.z * kchunk, k_hi = min(K, k_lo + kchunk);
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16 x 16 threads, 4 x 4 outputs each
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = k_lo; k0 < k_hi; k0 += TK) {
    for (int idx = threadIdx.x; idx < TK * TM; idx += kB) {
      int kk, mm;
      if (transA) {  // A stored K x M: consecutive m contiguous
        kk = idx / TM;
        mm = idx - kk * TM;
      } else {       // A stored M x K: consecutive k contiguous
        mm = idx / TK;
        kk = idx - mm * TK;
      }
      const int m = m0 + mm, k = k0 + kk;
      As[kk][mm] = (m < M && k < k_hi) ? at(A, lda, transA, m, k) : 0.f;
    }
    for (int idx = threadIdx.x; idx < TK * TN; idx += kB) {
      int kk, nn;
      if (transB) {  // B stored N x K
        nn = idx / TK;
        kk = idx - nn * TK;
      } else {       // B stored K x N
        kk = idx / TN;
        nn = idx - kk * TN;
      }
      const int n = n0 + nn, k = k0 + kk;
      Bs[kk][nn] = (n < N && k < k_hi) ? at(Bm, ldb, transB, k, n) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
  const bool split = gridDim.z > 1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= N) continue;
      float* c = C + (size_t)m * ldc + n;
      if (split) {  // partial sums of a K split: C was prepared (zeroed or holding beta*C) by the launcher
        atomicAdd(c, alpha * acc[i][j]);
        continue;
      }
      float v = alpha * acc[i][j];
      if (bias) v += bias[n];
      if (beta != 0.f) v += beta * *c;
      if (relu) v = fmaxf(v, 0.f);
      if (mask && !(mask[(size_t)m * ldmask + n] > 0.f)) v = 0.f;
      *c = v;
    }
  }
}

__global__ __launch_bounds__(kB) void scale_kernel(float* __restrict__ C, long long total, int N, int ldc, float beta) {
  for (long long idx = (long long)blockIdx.x * kB + threadIdx.x; idx < total; idx += (long long)gridDim.x * kB) {
    const long long m = idx / N;
    const int n = (int)(idx - m * N);
    float* c = C + (size_t)m * ldc + n;
    *c = beta == 0.f ? 0.f : beta * *c;
  }
}

// out[c] (+)= sum_r X[r][c]
__global__ __launch_bounds__(kB) void colsum_kernel(const float* __restrict__ X, float* __restrict__ out, int rows,
                                                    int cols, int ld, int rows_per_block) {
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  for (int c = threadIdx.x; c < cols; c += kB) {
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += X[(size_t)r * ld + c];
    atomicAdd(out + c, s);
  }
}

// dst[r][c] = s[r*lds + off] * src[r][c]
__global__ __launch_bounds__(kB) void rowscale_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                      const float* __restrict__ s, long long rows, int cols, int lds,
                                                      int off) {
  const long long total = rows * cols;
  for (long long idx = (long long)blockIdx.x * kB + threadIdx.x; idx < total; idx += (long long)gridDim.x * kB) {
    const long long r = idx / cols;
    dst[idx] = s[r * lds + off] * src[idx];
  }
}

// out[r*ldo + off] = <a[r], b[r]>  (cols <= 64: one wave per row)
__global__ __launch_bounds__(kB) void rowdot_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                    float* __restrict__ out, long long rows, int cols, int ldo,
                                                    int off) {
  const int lane = threadIdx.x & 63;
  for (long long r = (long long)blockIdx.x * (kB / 64) + (threadIdx.x >> 6); r < rows;
       r += (long long)gridDim.x * (kB / 64)) {
    float v = 0.f;
    for (int c = lane; c < cols; c += 64) v += a[r * cols + c] * b[r * cols + c];
    v = gn_wave_sum(v);
    if (lane == 0) out[r * ldo + off] = v;
  }
}

// Back through ef = sig(f) * dist, dist = softmax((logits + g)/tau):
//   dsig = sum_k def_k dist_k;  ddist_k = def_k sig + gdist_k;  df = dsig sig (1 - sig);
//   dlogits_k = dist_k (ddist_k - sum_j ddist_j dist_j) / tau.
// lgf (rows, ldl): column K holds the factor pre-activation f.  dlgf (rows, ldl): columns 0..K-1 <- dlogits,
// column K <- df, the rest 0.
__global__ __launch_bounds__(kB) void gumbel_bwd_kernel(const float* __restrict__ dist, const float* __restrict__ lgf,
                                                        const float* __restrict__ def, const float* __restrict__ gdist,
                                                        float* __restrict__ dlgf, long long rows, int K, int ldl,
                                                        float tau) {
  for (long long r = (long long)blockIdx.x * kB + threadIdx.x; r < rows; r += (long long)gridDim.x * kB) {
    const float f = lgf[r * ldl + K];
    const float sig = 1.f / (1.f + expf(-f));
    float dsig = 0.f, dot = 0.f;
    for (int k = 0; k < K; ++k) {
      const float d = dist[r * K + k], de = def[r * K + k];
      dsig += de * d;
      const float dd = de * sig + (gdist ? gdist[r * K + k] : 0.f);
      dot += dd * d;
    }
    for (int k = 0; k < K; ++k) {
      const float d = dist[r * K + k];
      const float dd = def[r * K + k] * sig + (gdist ? gdist[r * K + k] : 0.f);
      dlgf[r * ldl + k] = d * (dd - dot) / tau;
    }
    dlgf[r * ldl + K] = dsig * sig * (1.f - sig);
    for (int k = K + 1; k < ldl; ++k) dlgf[r * ldl + k] = 0.f;
  }
}

// Back through node2edge (one wave per hyperedge; forward quantities recomputed exactly as the forward
// kernel does).  Accumulates (atomics) into dxp (B,N,64), dpq (B,N,64) = [dP | dQn], dw2 (32), db2 (1).
__global__ __launch_bounds__(kB) void node2edge_bwd_kernel(const float* __restrict__ xp, const float* __restrict__ pq,
                                                           const float* __restrict__ H, const float* __restrict__ w2,
                                                           float b2, const float* __restrict__ dedges,
                                                           float* __restrict__ dxp, float* __restrict__ dpq,
                                                           float* __restrict__ dw2, float* __restrict__ db2, int N,
                                                           int E, long long total_edges) {
  extern __shared__ __align__(16) float lds[];
  const int wave = gn_uniform((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63, c = lane & 31;
  const long long eg_raw = (long long)blockIdx.x * (kB / 64) + wave;
  const bool live = eg_raw < total_edges;
  const long long eg = live ? eg_raw : total_edges - 1;
  const int b = (int)(eg / E);
  float* base = lds + (size_t)wave * 4 * N;
  int* s_idx = reinterpret_cast<int*>(base);
  float* s_h = base + N;
  float* s_att = base + 2 * N;   // att_m, later dv_m
  float* s_dw = base + 3 * N;    // dw_m
  int cnt = 0;
  const float* Hrow = H + (size_t)eg * N;
  for (int n0 = 0; n0 < N; n0 += 64) {
    const int n = n0 + lane;
    const float hv = n < N ? Hrow[n] : 0.f;
    const unsigned long long mask = __ballot(hv != 0.f);
    if (hv != 0.f) {
      const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
      s_idx[pos] = n;
      s_h[pos] = hv;
    }
    cnt += __popcll(mask);
  }
  __syncthreads();
  const float* pqb = pq + (size_t)b * N * GN_FEAT;
  const float* xpb = xp + (size_t)b * N * GN_FEAT;
  float* dpqb = dpq + (size_t)b * N * GN_FEAT;
  float* dxpb = dxp + (size_t)b * N * GN_FEAT;
  float qe = 0.f;
  for (int m = 0; m < cnt; ++m) qe = fmaf(s_h[m], pqb[(size_t)s_idx[m] * GN_FEAT + lane], qe);
  const float qlo = __shfl(qe, 32 + c, GN_WAVE);   // Q_e[c] on every lane
  const float w2c = w2[c];
  const float de = dedges[(size_t)eg * GN_FEAT + lane];   // dE[lane]
  // forward logits and dw_m = <dE, x'_m>
  for (int m = 0; m < cnt; ++m) {
    const int n = s_idx[m];
    float t = lane < 32 ? w2c * fmaxf(pqb[(size_t)n * GN_FEAT + c] + qlo, 0.f) : 0.f;
    t = gn_wave_sum(t);
    float d = de * xpb[(size_t)n * GN_FEAT + lane];
    d = gn_wave_sum(d);
    if (lane == 0) {
      s_att[m] = t + b2;
      s_dw[m] = d;
    }
  }
  __syncthreads();
  float mx = cnt < N ? 0.f : -INFINITY;
  for (int m = lane; m < cnt; m += 64) mx = fmaxf(mx, s_att[m] * s_h[m]);
  mx = gn_wave_max(mx);
  float sum = 0.f;
  for (int m = lane; m < cnt; m += 64) sum += expf(s_att[m] * s_h[m] - mx);
  sum = gn_wave_sum(sum);
  sum += (float)(N - cnt) * expf(0.f - mx);
  // sum_j p_j dp_j over members (dp = dw * h; non-members have dp = 0)
  float pd = 0.f;
  for (int m = lane; m < cnt; m += 64) {
    const float p = expf(s_att[m] * s_h[m] - mx) / sum;
    pd += p * s_dw[m] * s_h[m];
  }
  pd = gn_wave_sum(pd);
  __syncthreads();
  float dq = 0.f;     // dQ_e[c], lanes < 32
  float dw2c = 0.f, db2l = 0.f;
  for (int m = 0; m < cnt; ++m) {
    const int n = s_idx[m];
    const float hv = s_h[m];
    const float p = expf(s_att[m] * hv - mx) / sum;
    const float wgt = p * hv;
    // d x'_n += w_m dE
    if (live) atomicAdd(dxpb + (size_t)n * GN_FEAT + lane, wgt * de);
    const float dv = p * (s_dw[m] * hv - pd);
    const float datt = dv * hv;
    if (lane < 32) {
      const float pre = pqb[(size_t)n * GN_FEAT + c] + qlo;
      const float act = fmaxf(pre, 0.f);
      dw2c = fmaf(datt, act, dw2c);
      const float dpre = pre > 0.f ? datt * w2c : 0.f;
      if (live) atomicAdd(dpqb + (size_t)n * GN_FEAT + c, dpre);   // dP_n[c]
      dq += dpre;
    }
    db2l += datt;
  }
  // dQn_n[c] += H[e,n] dQ_e[c]
  const float dq_hi = __shfl(dq, c, GN_WAVE);   // lanes 32..63 pick up dQ_e[c]
  if (live && lane >= 32)
    for (int m = 0; m < cnt; ++m) atomicAdd(dpqb + (size_t)s_idx[m] * GN_FEAT + lane, s_h[m] * dq_hi);
  if (live && lane < 32) atomicAdd(dw2 + c, dw2c);
  if (live && lane == 0) atomicAdd(db2, db2l);
}

// out[r][c] = alpha * a[r][c] + beta * out[r][c] on (rows x cols) blocks with leading dimensions
__global__ __launch_bounds__(kB) void axpby2d_kernel(float* __restrict__ out, int ldo, const float* __restrict__ a,
                                                     int lda, long long rows, int cols, float alpha, float beta) {
  const long long total = rows * cols;
  for (long long idx = (long long)blockIdx.x * kB + threadIdx.x; idx < total; idx += (long long)gridDim.x * kB) {
    const long long r = idx / cols;
    const int c = (int)(idx - r * cols);
    float* o = out + r * ldo + c;
    const float v = alpha * a[r * lda + c];
    *o = beta == 0.f ? v : v + beta * *o;
  }
}

// ef[r][k] = sigmoid(lgf[r][K]) * dist[r][k]   (edge_feat of MLP_dict_softmax from its saved pieces)
__global__ __launch_bounds__(kB) void gumbel_ef_kernel(const float* __restrict__ dist, const float* __restrict__ lgf,
                                                       float* __restrict__ ef, long long rows, int K, int ldl) {
  const long long total = rows * K;
  for (long long idx = (long long)blockIdx.x * kB + threadIdx.x; idx < total; idx += (long long)gridDim.x * kB) {
    const long long r = idx / K;
    const float sig = 1.f / (1.f + expf(-lgf[r * ldl + K]));
    ef[idx] = sig * dist[idx];
  }
}

inline int cap_grid(long long items, int per_block, int cap = 4096) {
  long long g = (items + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int gn_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                           int transA, int transB, const float* bias, const float* mask, int ldmask, int relu,
                           float alpha, float beta, gn_stream_t stream) {
  GN_REQUIRE_PTR(A);
  GN_REQUIRE_PTR(B);
  GN_REQUIRE_PTR(C);
  if (M <= 0 || N <= 0 || K <= 0 || ldc < N) return GN_ERR_SHAPE;
  if (lda < (transA ? M : K) || ldb < (transB ? K : N)) return GN_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const int gm = (M + TM - 1) / TM, gn = (N + TN - 1) / TN;
  // few output tiles and a long K (weight gradients): split K over workgroups, partial sums by atomics
  int splits = 1;
  if ((long long)gm * gn < 256 && K >= 4096 && !relu && !mask && !bias) {
    splits = (int)((512 + (long long)gm * gn - 1) / ((long long)gm * gn));
    const int max_splits = (K + 255) / 256;
    splits = splits > max_splits ? max_splits : splits;
  }
  int kchunk = ((K + splits - 1) / splits + TK - 1) / TK * TK;
  splits = (K + kchunk - 1) / kchunk;
  if (splits > 1) {
    // C <- beta*C (or 0) first; the workgroups of the split then add their partial sums atomically
    const long long total = (long long)M * N;
    hipLaunchKernelGGL(scale_kernel, dim3(cap_grid(total, kB)), dim3(kB), 0, s, C, total, N, ldc, beta);
  }
  hipLaunchKernelGGL(gemm_kernel, dim3(gn, gm, splits), dim3(kB), 0, s, A, B, C, M, N, K, lda, ldb, ldc, transA, transB,
                     bias, mask, ldmask, relu, alpha, beta, kchunk);
  return gn_check_launch();
}

extern "C" int gn_colsum_f32(const float* X, float* out, int rows, int cols, int ld, gn_stream_t stream) {
  GN_REQUIRE_PTR(X);
  GN_REQUIRE_PTR(out);
  if (rows <= 0 || cols <= 0 || ld < cols) return GN_ERR_SHAPE;
  const int rpb = rows < 4096 ? 64 : 256;
  hipLaunchKernelGGL(colsum_kernel, dim3((rows + rpb - 1) / rpb), dim3(kB), 0, (hipStream_t)stream, X, out, rows, cols,
                     ld, rpb);
  return gn_check_launch();
}

extern "C" int gn_rowscale_f32(float* dst, const float* src, const float* s, long long rows, int cols, int lds, int off,
                               gn_stream_t stream) {
  GN_REQUIRE_PTR(dst);
  GN_REQUIRE_PTR(src);
  GN_REQUIRE_PTR(s);
  if (rows <= 0 || cols <= 0) return GN_ERR_SHAPE;
  hipLaunchKernelGGL(rowscale_kernel, dim3(cap_grid(rows * cols, kB)), dim3(kB), 0, (hipStream_t)stream, dst, src, s,
                     rows, cols, lds, off);
  return gn_check_launch();
}

extern "C" int gn_rowdot_f32(const float* a, const float* b, float* out, long long rows, int cols, int ldo, int off,
                             gn_stream_t stream) {
  GN_REQUIRE_PTR(a);
  GN_REQUIRE_PTR(b);
  GN_REQUIRE_PTR(out);
  if (rows <= 0 || cols <= 0) return GN_ERR_SHAPE;
  hipLaunchKernelGGL(rowdot_kernel, dim3(cap_grid(rows, kB / 64)), dim3(kB), 0, (hipStream_t)stream, a, b, out, rows,
                     cols, ldo, off);
  return gn_check_launch();
}

extern "C" int gn_gumbel_bwd_f32(const float* dist, const float* lgf, const float* def, const float* gdist, float* dlgf,
                                 long long rows, int K, int ldl, float tau, gn_stream_t stream) {
  GN_REQUIRE_PTR(dist);
  GN_REQUIRE_PTR(lgf);
  GN_REQUIRE_PTR(def);
  GN_REQUIRE_PTR(dlgf);
  if (rows <= 0 || K < 1 || ldl <= K || !(tau > 0.f)) return GN_ERR_SHAPE;
  hipLaunchKernelGGL(gumbel_bwd_kernel, dim3(cap_grid(rows, kB)), dim3(kB), 0, (hipStream_t)stream, dist, lgf, def,
                     gdist, dlgf, rows, K, ldl, tau);
  return gn_check_launch();
}

extern "C" int gn_node2edge_bwd_f32(const float* xp, const float* pq, const float* H, const float* w2, float b2,
                                    const float* dedges, float* dxp, float* dpq, float* dw2, float* db2, int B, int N,
                                    int E, gn_stream_t stream) {
  const void* ptrs[] = {xp, pq, H, w2, dedges, dxp, dpq, dw2, db2};
  for (const void* p : ptrs)
    if (p == nullptr) return GN_ERR_NULL;
  if (B <= 0 || N <= 0 || E <= 0) return GN_ERR_SHAPE;
  const size_t lds = (size_t)(kB / 64) * 4 * N * sizeof(float);
  if (lds > 64 * 1024) return GN_ERR_LDS;
  const long long total = (long long)B * E;
  const long long grid = (total + 3) / 4;
  if (grid > 0x7fffffffLL) return GN_ERR_SHAPE;
  hipLaunchKernelGGL(node2edge_bwd_kernel, dim3((unsigned)grid), dim3(kB), lds, (hipStream_t)stream, xp, pq, H, w2, b2,
                     dedges, dxp, dpq, dw2, db2, N, E, total);
  return gn_check_launch();
}

extern "C" int gn_axpby2d_f32(float* out, int ldo, const float* a, int lda, long long rows, int cols, float alpha,
                              float beta, gn_stream_t stream) {
  GN_REQUIRE_PTR(out);
  GN_REQUIRE_PTR(a);
  if (rows <= 0 || cols <= 0 || ldo < cols || lda < cols) return GN_ERR_SHAPE;
  hipLaunchKernelGGL(axpby2d_kernel, dim3(cap_grid(rows * cols, kB)), dim3(kB), 0, (hipStream_t)stream, out, ldo, a,
                     lda, rows, cols, alpha, beta);
  return gn_check_launch();
}

extern "C" int gn_gumbel_ef_f32(const float* dist, const float* lgf, float* ef, long long rows, int K, int ldl,
                                gn_stream_t stream) {
  GN_REQUIRE_PTR(dist);
  GN_REQUIRE_PTR(lgf);
  GN_REQUIRE_PTR(ef);
  if (rows <= 0 || K < 1 || ldl <= K) return GN_ERR_SHAPE;
  hipLaunchKernelGGL(gumbel_ef_kernel, dim3(cap_grid(rows * K, kB)), dim3(kB), 0, (hipStream_t)stream, dist, lgf, ef,
                     rows, K, ldl);
  return gn_check_launch();
}
