// Backward (training) kernels of the GroupNet MS-HGNN path for gfx950 — SURVEY.md §8f rank 2.
//
// The forward is a handful of fused matrix-core kernels; the backward is deliberately built from a few
// GENERIC blocks, correctness first (train_hyper_nba.py:116 back-propagates through these modules, but
// training throughput is not the path's headline):
//   gn_gemm[_grouped]_f32  C = beta*C + op(A) op(B) (+ bias) (relu) (masked by another tensor's sign), many
//                          problems per launch; LDS-staged 128x64x32 tiles on the fp32 matrix cores with
//                          optional split-K (atomic) — used for the re-computation of hidden activations,
//                          for input gradients dX = dY W and for weight gradients dW = dY^T X (K = rows,
//                          split over workgroups, bias gradient as a side output);
//   gn_typed_bwd_f32       the per-type scalings / dot products of the typed aggregation;
//   gn_typed_scale/dot     the per-row, per-type scalings of the typed aggregation;
//   gn_gumbel_bwd_f32      back through fac * softmax((logits + g) / tau) and the sigmoid;
//   gn_node2edge_bwd_f32   back through the attention-weighted pooling (one wave per hyperedge, like the
//                          forward; node rows are shared by edges, so their gradients are atomic adds).
// Gather and scatter are each other's adjoints and reuse the forward kernels.
#include <stdlib.h>

#include "gn_common.hpp"

namespace {

constexpr int kB = 256;
constexpr int BM = 128, BN = 64, BK = 32;   // workgroup tile; each of the 4 waves owns 32 x 64 of it
constexpr int kMaxDescs = 16;

typedef float floatx16 __attribute__((ext_vector_type(16)));

// fp32 -> three bf16 parts per value (round to nearest, remainders exact in fp32), eight values at a time, written
// pair by pair so that every conversion is one two-source v_cvt_pk_bf16_f32.  x = p1 + p2 + p3 to 2^-24 |x|.
typedef __bf16 gbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gbf16x2 __attribute__((ext_vector_type(2)));
typedef float gf32x2 __attribute__((ext_vector_type(2)));
typedef unsigned gu32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split8(const float (&v)[8], gbf16x8& p1, gbf16x8& p2, gbf16x8& p3) {
  gu32x4 q1, q2, q3;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const gf32x2 x = {v[2 * j], v[2 * j + 1]};
    const unsigned u1 = __builtin_bit_cast(unsigned, __builtin_convertvector(x, gbf16x2));
    const gf32x2 r1 = {x[0] - __builtin_bit_cast(float, u1 << 16), x[1] - __builtin_bit_cast(float, u1 & 0xffff0000u)};
    const unsigned u2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, gbf16x2));
    const gf32x2 r2 = {r1[0] - __builtin_bit_cast(float, u2 << 16), r1[1] - __builtin_bit_cast(float, u2 & 0xffff0000u)};
    q1[j] = u1;
    q2[j] = u2;
    q3[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, gbf16x2));
  }
  p1 = __builtin_bit_cast(gbf16x8, q1);
  p2 = __builtin_bit_cast(gbf16x8, q2);
  p3 = __builtin_bit_cast(gbf16x8, q3);
}
// acc += a . b from the parts: the six significant part-products, smallest first (as the forward's mfma_substep<3>)
__device__ __forceinline__ void mfma_x6(const gbf16x8 (&a)[3], const gbf16x8 (&b)[3], floatx16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}

// one GEMM problem of a grouped launch (by value in the kernarg segment)
struct GemmDesc {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* mask;
  const float* rs;     // optional scale of A's STORED rows: A_eff[r][:] = rs[r*rs_ld] * A[r][:]
  float* colsum;       // transA only: colsum[m] += sum_k A_eff[k][m]  (bias gradient next to dW = dY^T X)
  int M, N, K, lda, ldb, ldc, ldmask, rs_ld;
  int flags;
  float alpha, beta;
  int tile0, gn, gmn, kchunk;   // filled by the launcher
};
struct GemmTable {
  GemmDesc d[kMaxDescs];
  int n;
};

// Register images of the next k-chunk of A (BM x BK) and B (BK x BN): all loads of a chunk are issued back
// to back with clamped addresses (no branches between them) and are in flight while the matrix cores work
// on the previous chunk; out-of-range elements are zeroed afterwards.
constexpr int kALoads = BM * BK / kB, kBLoads = BN * BK / kB;

// Issue-only: nothing here consumes a loaded value, so the waits land where the values are used — at the
// LDS stores of the NEXT iteration, after the matrix-core phase.
template <bool TRANS>   // TRANS: stored (K x M), m contiguous; else stored (M x K), k contiguous
__device__ __forceinline__ unsigned fetch_a(float (&va)[kALoads], float (&vs)[kALoads], const float* __restrict__ A,
                                            const float* __restrict__ rs, int lda, int rs_ld, int M, int m0, int k0,
                                            int k_hi, int tid) {
  unsigned ok = 0;
#pragma unroll
  for (int it = 0; it < kALoads; ++it) {
    const int idx = tid + it * kB;
    const int mm = TRANS ? idx % BM : idx / BK, kk = TRANS ? idx / BM : idx % BK;
    const int m = m0 + mm, k = k0 + kk;
    if (m < M && k < k_hi) ok |= 1u << it;
    const int mc = min(m, M - 1), kc = min(k, k_hi - 1);
    const size_t srow = TRANS ? kc : mc;
    va[it] = A[srow * lda + (TRANS ? mc : kc)];
    if (rs) vs[it] = rs[srow * rs_ld];
  }
  return ok;
}

template <bool TRANS>   // TRANS: stored (N x K), k contiguous; else stored (K x N), n contiguous
__device__ __forceinline__ unsigned fetch_b(float (&vb)[kBLoads], const float* __restrict__ Bm, int ldb, int N, int n0,
                                            int k0, int k_hi, int tid) {
  unsigned ok = 0;
#pragma unroll
  for (int it = 0; it < kBLoads; ++it) {
    const int idx = tid + it * kB;
    const int nn = TRANS ? idx / BK : idx % BN, kk = TRANS ? idx % BK : idx / BN;
    const int n = n0 + nn, k = k0 + kk;
    if (n < N && k < k_hi) ok |= 1u << it;
    const int nc = min(n, N - 1), kc = min(k, k_hi - 1);
    vb[it] = TRANS ? Bm[(size_t)nc * ldb + kc] : Bm[(size_t)kc * ldb + nc];
  }
  return ok;
}

// Vector form of the same staging for interior tiles of 16-byte-aligned operands: one dwordx4 per 4 elements,
// no clamps, no masks (4 + 2 loads per thread and chunk instead of 16 + 8).
constexpr int kA4 = kALoads / 4, kB4 = kBLoads / 4;

template <bool TRANS>
__device__ __forceinline__ void fetch_a4(f32x4 (&va)[kA4], float (&vs)[kA4], const float* __restrict__ A,
                                         const float* __restrict__ rs, int lda, int rs_ld, int M, int m0, int k0,
                                         int tid) {
#pragma unroll
  for (int it = 0; it < kA4; ++it) {
    const int idx = tid + it * kB;
    // TRANS: 4 consecutive m of one k; else 4 consecutive k of one m.  Rows past M are clamped, not zeroed:
    // they only feed output rows that are never stored.
    const int mm = TRANS ? (idx % (BM / 4)) * 4 : idx / (BK / 4), kk = TRANS ? idx / (BM / 4) : (idx % (BK / 4)) * 4;
    const int m = TRANS ? min(m0 + mm, M - 4) : min(m0 + mm, M - 1);
    const size_t srow = TRANS ? (size_t)(k0 + kk) : (size_t)m;
    va[it] = *reinterpret_cast<const f32x4*>(A + srow * lda + (TRANS ? m : k0 + kk));
    if (rs) vs[it] = rs[srow * rs_ld];
  }
}

template <bool TRANS>
__device__ __forceinline__ void fetch_b4(f32x4 (&vb)[kB4], const float* __restrict__ Bm, int ldb, int N, int n0,
                                         int k0, int tid) {
#pragma unroll
  for (int it = 0; it < kB4; ++it) {
    const int idx = tid + it * kB;
    // TRANS (stored N x K): 4 consecutive k of one n; else 4 consecutive n of one k (columns past N clamped)
    const int nn = TRANS ? idx / (BK / 4) : (idx % (BN / 4)) * 4, kk = TRANS ? (idx % (BK / 4)) * 4 : idx / (BN / 4);
    const int n = TRANS ? min(n0 + nn, N - 1) : min(n0 + nn, N - 4);
    vb[it] = *reinterpret_cast<const f32x4*>(TRANS ? Bm + (size_t)n * ldb + k0 + kk : Bm + (size_t)(k0 + kk) * ldb + n);
  }
}

template <bool TA, bool TB, bool FAST, bool X6>
__device__ __forceinline__ void gemm_body(const GemmDesc& D, float (&As)[BK][BM + 4], float (&Bs)[BK][BN + 4]) {
  const int local = (int)blockIdx.x - D.tile0;
  const int split = local / D.gmn, t = local - split * D.gmn;
  const int tm = t / D.gn, tn = t - tm * D.gn;
  const int m0 = tm * BM, n0 = tn * BN;
  const int M = D.M, N = D.N;
  const int k_lo = split * D.kchunk, k_hi = min(D.K, k_lo + D.kchunk);
  const float* __restrict__ A = D.A;
  const float* __restrict__ Bm = D.B;
  const float* __restrict__ rs = D.rs;
  const int lda = D.lda, ldb = D.ldb, rs_ld = D.rs_ld;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  floatx16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
  const bool want_cs = TA && D.colsum != nullptr && tn == 0;
  const bool transC = D.flags & GN_GEMM_TRANS_C;
  float cs = 0.f;
  float va[FAST ? 1 : kALoads], vs[FAST ? 1 : kALoads], vb[FAST ? 1 : kBLoads];
  f32x4 va4[FAST ? kA4 : 1], vb4[FAST ? kB4 : 1];
  float vs4[FAST ? kA4 : 1];
  unsigned okA = 0, okB = 0;
  if constexpr (FAST) {
    fetch_a4<TA>(va4, vs4, A, rs, lda, rs_ld, M, m0, k_lo, tid);
    fetch_b4<TB>(vb4, Bm, ldb, N, n0, k_lo, tid);
  } else {
    okA = fetch_a<TA>(va, vs, A, rs, lda, rs_ld, M, m0, k_lo, k_hi, tid);
    okB = fetch_b<TB>(vb, Bm, ldb, N, n0, k_lo, k_hi, tid);
  }
  for (int k0 = k_lo; k0 < k_hi; k0 += BK) {
    if constexpr (FAST) {
#pragma unroll
      for (int it = 0; it < kA4; ++it) {
        const int idx = tid + it * kB;
        f32x4 v = va4[it];
        if (rs) v *= vs4[it];
        if (TA) {
          *reinterpret_cast<f32x4*>(&As[idx / (BM / 4)][(idx % (BM / 4)) * 4]) = v;
        } else {
          const int mm = idx / (BK / 4), kk = (idx % (BK / 4)) * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) As[kk + j][mm] = v[j];
        }
      }
#pragma unroll
      for (int it = 0; it < kB4; ++it) {
        const int idx = tid + it * kB;
        if (TB) {
          const int nn = idx / (BK / 4), kk = (idx % (BK / 4)) * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) Bs[kk + j][nn] = vb4[it][j];
        } else {
          *reinterpret_cast<f32x4*>(&Bs[idx / (BN / 4)][(idx % (BN / 4)) * 4]) = vb4[it];
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < kALoads; ++it) {
        const int idx = tid + it * kB;
        float v = rs ? va[it] * vs[it] : va[it];
        v = (okA >> it & 1u) ? v : 0.f;
        if (TA) As[idx / BM][idx % BM] = v;
        else As[idx % BK][idx / BK] = v;
      }
#pragma unroll
      for (int it = 0; it < kBLoads; ++it) {
        const int idx = tid + it * kB;
        const float v = (okB >> it & 1u) ? vb[it] : 0.f;
        if (TB) Bs[idx % BK][idx / BK] = v;
        else Bs[idx / BN][idx % BN] = v;
      }
    }
    __syncthreads();
    if (k0 + BK < k_hi) {   // the next chunk is in flight while the matrix cores run
      if constexpr (FAST) {
        fetch_a4<TA>(va4, vs4, A, rs, lda, rs_ld, M, m0, k0 + BK, tid);
        fetch_b4<TB>(vb4, Bm, ldb, N, n0, k0 + BK, tid);
      } else {
        okA = fetch_a<TA>(va, vs, A, rs, lda, rs_ld, M, m0, k0 + BK, k_hi, tid);
        okB = fetch_b<TB>(vb, Bm, ldb, N, n0, k0 + BK, k_hi, tid);
      }
    }
    if (want_cs && tid < BM) {
#pragma unroll
      for (int kk = 0; kk < BK; ++kk) cs += As[kk][tid];
    }
    if constexpr (X6) {
      // fp32-accurate products on the bf16 matrix cores: per 16-deep k-step a lane reads its 8 k-values of the A row
      // and of the two B columns (k = ks + 8 hi + j: any mapping works as long as A and B agree), splits them into
      // three bf16 parts and issues 2 x 6 MFMAs (16x the fp32 instruction's rate: the splitting, ~130 VALU per
      // k-step, is what bounds this loop)
#pragma unroll
      for (int ks = 0; ks < BK; ks += 16) {
        float av[8], b0v[8], b1v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          av[j] = As[ks + 8 * hi + j][wave * 32 + l31];
          b0v[j] = Bs[ks + 8 * hi + j][l31];
          b1v[j] = Bs[ks + 8 * hi + j][32 + l31];
        }
        gbf16x8 ap[3], bp0[3], bp1[3];
        split8(av, ap[0], ap[1], ap[2]);
        split8(b0v, bp0[0], bp0[1], bp0[2]);
        split8(b1v, bp1[0], bp1[1], bp1[2]);
        if (!transC) {
          mfma_x6(ap, bp0, acc0);
          mfma_x6(ap, bp1, acc1);
        } else {    // operands swapped: the accumulators hold the transposed tile (lanes run along m)
          mfma_x6(bp0, ap, acc0);
          mfma_x6(bp1, ap, acc1);
        }
      }
    } else if (!transC) {
#pragma unroll 4
      for (int kk = 0; kk < BK; kk += 2) {
        const float a = As[kk + hi][wave * 32 + l31];
        const float b0 = Bs[kk + hi][l31], b1 = Bs[kk + hi][32 + l31];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
      }
    } else {   // operands swapped: the accumulators hold the transposed tile (lanes run along m)
#pragma unroll 4
      for (int kk = 0; kk < BK; kk += 2) {
        const float a = As[kk + hi][wave * 32 + l31];
        const float b0 = Bs[kk + hi][l31], b1 = Bs[kk + hi][32 + l31];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, a, acc1, 0, 0, 0);
      }
    }
    __syncthreads();
  }
  if (want_cs && tid < BM && m0 + tid < M) atomicAdd(D.colsum + m0 + tid, cs);
  const bool accum = D.flags & GN_GEMM_ACCUM, relu = D.flags & GN_GEMM_RELU;
  const float alpha = D.alpha, beta = D.beta;
  if (transC) {   // accumulate mode only: C is (N x M, ldc); lane = m, register = n
    const int m = m0 + wave * 32 + l31;
    if (m < M) {
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + half * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (n < N) atomicAdd(D.C + (size_t)n * D.ldc + m, alpha * (half ? acc1[r] : acc0[r]));
        }
    }
    return;
  }
  const float* __restrict__ bias = D.bias;
  const float* __restrict__ mask = D.mask;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int n = n0 + half * 32 + l31;
    if (n >= N) continue;
    const float bn = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (m >= M) continue;
      float v = alpha * (half ? acc1[r] : acc0[r]);
      float* c = D.C + (size_t)m * D.ldc + n;
      if (accum) {
        atomicAdd(c, v);
        continue;
      }
      v += bn;
      if (beta != 0.f) v += beta * *c;
      if (relu) v = fmaxf(v, 0.f);
      if (mask && !(mask[(size_t)m * D.ldmask + n] > 0.f)) v = 0.f;
      *c = v;
    }
  }
}

// C = beta*C + alpha*op(A)op(B) (+bias)(relu)(mask), or with GN_GEMM_ACCUM: C += alpha*op(A)op(B) by atomics
// (K split over workgroups).  fp32 matrix cores (v_mfma_f32_32x32x2_f32), operands staged through LDS
// k-major so that either storage order of A and B loads coalesced.  Two kernels so that each keeps its own
// register budget: the vector-staging one (16-byte aligned operands, K a multiple of 32 — every large
// problem of the backward) runs 4 workgroups per CU; the scalar one takes anything.
template <bool FAST, bool X6>
__device__ __forceinline__ void gemm_dispatch(const GemmTable& T, float (&As)[BK][BM + 4], float (&Bs)[BK][BN + 4]) {
  int g = 0;
  for (int i = 1; i < T.n; ++i)
    if ((int)blockIdx.x >= T.d[i].tile0) g = i;
  g = gn_uniform(g);
  const GemmDesc& D = T.d[g];
  const int tt = gn_uniform(D.flags & (GN_GEMM_TRANS_A | GN_GEMM_TRANS_B));
  if (tt == 0) gemm_body<false, false, FAST, X6>(D, As, Bs);
  else if (tt == GN_GEMM_TRANS_A) gemm_body<true, false, FAST, X6>(D, As, Bs);
  else if (tt == GN_GEMM_TRANS_B) gemm_body<false, true, FAST, X6>(D, As, Bs);
  else gemm_body<true, true, FAST, X6>(D, As, Bs);
}

__global__ __launch_bounds__(kB, 4) void gemm_mfma_kernel(const GemmTable T) {
  __shared__ float As[BK][BM + 4];
  __shared__ float Bs[BK][BN + 4];
  gemm_dispatch<true, false>(T, As, Bs);
}

// the same tiles with the products formed on the bf16 matrix cores from three-part splits (fp32-accurate, like the
// forward's P = 3 kernels): the default for the vector-staged problems; GN_GEMM_X6=0 selects the fp32-core kernel
__global__ __launch_bounds__(kB, 3) void gemm_x6_kernel(const GemmTable T) {
  __shared__ float As[BK][BM + 4];
  __shared__ float Bs[BK][BN + 4];
  gemm_dispatch<true, true>(T, As, Bs);
}

__global__ __launch_bounds__(kB) void gemm_mfma_edge_kernel(const GemmTable T) {
  __shared__ float As[BK][BM + 4];
  __shared__ float Bs[BK][BN + 4];
  gemm_dispatch<false, false>(T, As, Bs);
}

__global__ __launch_bounds__(kB) void scale_kernel(float* __restrict__ C, long long total, int N, int ldc, float beta) {
  for (long long idx = (long long)blockIdx.x * kB + threadIdx.x; idx < total; idx += (long long)gridDim.x * kB) {
    const long long m = idx / N;
    const int n = (int)(idx - m * N);
    float* c = C + (size_t)m * ldc + n;
    *c = beta == 0.f ? 0.f : beta * *c;
  }
}

// Back through ef = sig(f) * dist, dist = softmax((logits + g)/tau):
//   dsig = sum_k def_k dist_k;  ddist_k = def_k sig + gdist_k;  df = dsig sig (1 - sig);
//   dlogits_k = dist_k (ddist_k - sum_j ddist_j dist_j) / tau.
// lgf (rows, ldl): column K holds the factor pre-activation f.  dlgf (rows, ldl): columns 0..K-1 <- dlogits,
// column K <- df, the rest 0.  sym_N > 0: rows are the unordered pairs of the pairwise graph — one row of
// logits fed the two ordered edges (i,j), (j,i) of dist (each with its own noise), def is the gradient of
// ef_ij + ef_ji, and the row receives the sum over its ordered edges.
__device__ __forceinline__ void gumbel_bwd_body(const float* __restrict__ dist, const float* __restrict__ lgf,
                                                const float* __restrict__ def, const float* __restrict__ gdist,
                                                float* __restrict__ dlgf, long long rows, int K, int ldl, float tau,
                                                int sym_N) {
  const int P = sym_N > 0 ? gn_pair_count(sym_N) : 1;
  for (long long r = (long long)blockIdx.x * kB + threadIdx.x; r < rows; r += (long long)gridDim.x * kB) {
    const float f = lgf[r * ldl + K];
    const float sig = 1.f / (1.f + expf(-f));
    // rows of dist / gdist this row of logits fed: itself, or with sym_N the ordered edges (i,j) and (j,i)
    long long e[2] = {r, -1};
    if (sym_N > 0) {
      const long long b = r / P;
      int i, j;
      gn_pair_decode((int)(r - b * P), sym_N, i, j);
      e[0] = (b * sym_N + i) * sym_N + j;
      e[1] = i == j ? -1 : (b * sym_N + j) * sym_N + i;
    }
    float df = 0.f;
    for (int k = 0; k < ldl; ++k) dlgf[r * ldl + k] = 0.f;
    for (int s = 0; s < 2; ++s) {
      if (e[s] < 0) continue;
      const float* d_ = dist + e[s] * K;
      const float* gd = gdist ? gdist + e[s] * K : nullptr;
      float dsig = 0.f, dot = 0.f;
      for (int k = 0; k < K; ++k) {
        const float d = d_[k], de = def[r * K + k];
        dsig += de * d;
        dot += (de * sig + (gd ? gd[k] : 0.f)) * d;
      }
      for (int k = 0; k < K; ++k) {
        const float dd = def[r * K + k] * sig + (gd ? gd[k] : 0.f);
        dlgf[r * ldl + k] += d_[k] * (dd - dot) / tau;
      }
      df += dsig * sig * (1.f - sig);
    }
    dlgf[r * ldl + K] = df;
  }
}
__global__ __launch_bounds__(kB) void gumbel_bwd_kernel(const float* __restrict__ dist, const float* __restrict__ lgf,
                                                        const float* __restrict__ def, const float* __restrict__ gdist,
                                                        float* __restrict__ dlgf, long long rows, int K, int ldl,
                                                        float tau, int sym_N) {
  gumbel_bwd_body(dist, lgf, def, gdist, dlgf, rows, K, ldl, tau, sym_N);
}
// every module of a backward stage in one launch (blockIdx.y = module): these launches take ~14 us whatever their size
struct GumbelBwdTable {
  gn_gumbel_bwd_group_t g[GN_MAX_GROUPS];
};
__global__ __launch_bounds__(kB) void gumbel_bwd_grouped_kernel(GumbelBwdTable T, int ldl, float tau) {
  const gn_gumbel_bwd_group_t G = T.g[blockIdx.y];
  gumbel_bwd_body(G.dist, G.lgf, G.def, G.gdist, G.dlgf, G.rows, G.K, ldl, tau, G.sym_N);
}

// Back through node2edge (one wave per hyperedge; forward quantities recomputed exactly as the forward
// kernel does).  Accumulates (atomics) into dxp (B,N,64), dpq (B,N,64) = [dP | dQn], dw2 (32), db2 (1).
__global__ __launch_bounds__(kB) void node2edge_bwd_kernel(const float* __restrict__ xp, const float* __restrict__ pq,
                                                           const float* __restrict__ H, const float* __restrict__ w2,
                                                           const float* __restrict__ b2p,
                                                           const float* __restrict__ dedges,
                                                           float* __restrict__ dxp, float* __restrict__ dpq,
                                                           float* __restrict__ dw2, float* __restrict__ db2, int N,
                                                           int E, long long total_edges) {
  extern __shared__ __align__(16) float lds[];
  const float b2 = *b2p;
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
  sum += gn_nonmember_sum(N - cnt, mx);
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

// The same backward, one WORKGROUP per scene (N small enough for LDS): a scene owns its node rows, so the
// gradients of x' and pq are accumulated in LDS (ds_add) and written once — no global atomics except the
// 33 attention-layer-1 sums per scene.  H == nullptr selects the implicit pairwise graph (E = N*N, edge
// e = i*N + j joins i and j with weight 1, weight 2 on i when i == j; model/MS_HGNN_batch.py:143-160), or
// with sym its N(N+1)/2 unordered pairs (the two ordered edges of a pair pool the same feature).
__device__ __forceinline__ void node2edge_bwd_scene_body(
    const float* __restrict__ xp, const float* __restrict__ pq, const float* __restrict__ H,
    const float* __restrict__ w2, const float* __restrict__ b2p, const float* __restrict__ dedges,
    float* __restrict__ dxp,
    float* __restrict__ dpq, float* __restrict__ dw2, float* __restrict__ db2, int N, int E, int sym, const int b) {
  extern __shared__ __align__(16) float lds[];
  const float b2 = *b2p;
  const int NF = N * GN_FEAT;
  float* s_xp = lds;
  float* s_pq = s_xp + NF;
  float* s_dxp = s_pq + NF;
  float* s_dpq = s_dxp + NF;
  float* s_red = s_dpq + NF;               // 64: dw2 (32) and db2 of the workgroup
  const int wave = gn_uniform((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63, c = lane & 31;
  float* base = s_red + 64 + (size_t)wave * 4 * N;
  int* s_idx = reinterpret_cast<int*>(base);
  float* s_h = base + N;
  float* s_att = base + 2 * N;
  float* s_dw = base + 3 * N;
  const float* xpb = xp + (size_t)b * NF;
  const float* pqb = pq + (size_t)b * NF;
  for (int i = threadIdx.x; i < NF; i += kB) {
    s_xp[i] = xpb[i];
    s_pq[i] = pqb[i];
    s_dxp[i] = 0.f;
    s_dpq[i] = 0.f;
  }
  if (threadIdx.x < 64) s_red[threadIdx.x] = 0.f;
  __syncthreads();
  const float w2c = w2[c];
  float dw2c = 0.f, db2l = 0.f;
  for (int e = wave; e < E; e += kB / 64) {
    int cnt = 0;
    if (H == nullptr) {
      int i, j;
      if (sym) {       // E = N(N+1)/2 unordered pairs, dedges already summed over (i,j) and (j,i)
        gn_pair_decode(e, N, i, j);
      } else {
        i = e / N;
        j = e - i * N;
      }
      if (lane == 0) {
        if (i == j) {
          s_idx[0] = i;
          s_h[0] = 2.f;
        } else {
          s_idx[0] = min(i, j);
          s_h[0] = 1.f;
          s_idx[1] = max(i, j);
          s_h[1] = 1.f;
        }
      }
      cnt = i == j ? 1 : 2;
    } else {
      const float* Hrow = H + ((size_t)b * E + e) * N;
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
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float qe = 0.f;
    for (int m = 0; m < cnt; ++m) qe = fmaf(s_h[m], s_pq[s_idx[m] * GN_FEAT + lane], qe);
    const float qlo = __shfl(qe, 32 + c, GN_WAVE);
    const float de = dedges[((size_t)b * E + e) * GN_FEAT + lane];
    for (int m = 0; m < cnt; ++m) {
      const int n = s_idx[m];
      float t = lane < 32 ? w2c * fmaxf(s_pq[n * GN_FEAT + c] + qlo, 0.f) : 0.f;
      t = gn_wave_sum(t);
      float d = de * s_xp[n * GN_FEAT + lane];
      d = gn_wave_sum(d);
      if (lane == 0) {
        s_att[m] = t + b2;
        s_dw[m] = d;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float mx = cnt < N ? 0.f : -INFINITY;
    for (int m = lane; m < cnt; m += 64) mx = fmaxf(mx, s_att[m] * s_h[m]);
    mx = gn_wave_max(mx);
    float sum = 0.f;
    for (int m = lane; m < cnt; m += 64) sum += expf(s_att[m] * s_h[m] - mx);
    sum = gn_wave_sum(sum);
    sum += gn_nonmember_sum(N - cnt, mx);
    float pd = 0.f;
    for (int m = lane; m < cnt; m += 64) pd += expf(s_att[m] * s_h[m] - mx) / sum * s_dw[m] * s_h[m];
    pd = gn_wave_sum(pd);
    float dq = 0.f;
    for (int m = 0; m < cnt; ++m) {
      const int n = s_idx[m];
      const float hv = s_h[m];
      const float p = expf(s_att[m] * hv - mx) / sum;
      atomicAdd(s_dxp + n * GN_FEAT + lane, p * hv * de);
      const float datt = p * (s_dw[m] * hv - pd) * hv;
      if (lane < 32) {
        const float pre = s_pq[n * GN_FEAT + c] + qlo;
        dw2c = fmaf(datt, fmaxf(pre, 0.f), dw2c);
        const float dpre = pre > 0.f ? datt * w2c : 0.f;
        atomicAdd(s_dpq + n * GN_FEAT + c, dpre);
        dq += dpre;
      }
      db2l += datt;
    }
    const float dq_hi = __shfl(dq, c, GN_WAVE);
    if (lane >= 32)
      for (int m = 0; m < cnt; ++m) atomicAdd(s_dpq + s_idx[m] * GN_FEAT + lane, s_h[m] * dq_hi);
    __builtin_amdgcn_wave_barrier();   // the lists are rewritten by the next edge of this wave
  }
  if (lane < 32) atomicAdd(s_red + c, dw2c);
  if (lane == 0) atomicAdd(s_red + 32, db2l);
  __syncthreads();
  float* dxpb = dxp + (size_t)b * NF;
  float* dpqb = dpq + (size_t)b * NF;
  for (int i = threadIdx.x; i < NF; i += kB) {
    dxpb[i] += s_dxp[i];
    dpqb[i] += s_dpq[i];
  }
  if (threadIdx.x < 32) atomicAdd(dw2 + threadIdx.x, s_red[threadIdx.x]);
  if (threadIdx.x == 32) atomicAdd(db2, s_red[32]);
}
__global__ __launch_bounds__(kB) void node2edge_bwd_scene_kernel(
    const float* __restrict__ xp, const float* __restrict__ pq, const float* __restrict__ H,
    const float* __restrict__ w2, const float* __restrict__ b2p, const float* __restrict__ dedges,
    float* __restrict__ dxp, float* __restrict__ dpq, float* __restrict__ dw2, float* __restrict__ db2, int N, int E,
    int sym) {
  node2edge_bwd_scene_body(xp, pq, H, w2, b2p, dedges, dxp, dpq, dw2, db2, N, E, sym, (int)blockIdx.x);
}
// every module of the stage in one launch: blockIdx.y = module, blockIdx.x = scene (the pairwise module's 88 us and the
// hyper modules' 24-41 us side by side instead of end to end)
struct N2EBwdTable {
  gn_n2e_bwd_group_t g[GN_MAX_GROUPS];
};
__global__ __launch_bounds__(kB) void node2edge_bwd_scene_grouped_kernel(N2EBwdTable T, int N) {
  const gn_n2e_bwd_group_t G = T.g[blockIdx.y];
  node2edge_bwd_scene_body(G.xp, G.pq, G.H, G.w2, G.b2, G.dedges, G.dxp, G.dpq, G.dw2, G.db2, N, G.E, G.sym, (int)blockIdx.x);
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
                                                       float* __restrict__ ef, long long rows, int K, int ldl,
                                                       int sym_N, float diag_w, int ld_ef) {
  const long long total = rows * K;
  const int P = sym_N > 0 ? gn_pair_count(sym_N) : 1;
  for (long long idx = (long long)blockIdx.x * kB + threadIdx.x; idx < total; idx += (long long)gridDim.x * kB) {
    const long long r = idx / K;
    const int k = (int)(idx - r * K);
    const float sig = 1.f / (1.f + expf(-lgf[r * ldl + K]));
    float d;
    if (sym_N > 0) {   // pair row: ef_ij + ef_ji (the diagonal has one ordered edge)
      const long long b = r / P;
      int i, j;
      gn_pair_decode((int)(r - b * P), sym_N, i, j);
      d = dist[((b * sym_N + i) * sym_N + j) * K + k];
      if (i != j) d += dist[((b * sym_N + j) * sym_N + i) * K + k];
      else d *= diag_w;
    } else {
      d = dist[idx];
    }
    ef[r * ld_ef + k] = sig * d;
  }
}

// Typed aggregation MLP, middle of its backward (one wave per edge row, all K types):
//   T (rows, K*hid) = dfeat W2cat on entry; Hc (rows, K*hid) the hidden activations relu(W1_k eo + b1_k);
//   def[r][k] = <T[r,k,:], Hc[r,k,:]> + <dfeat[r], b2[k]>      (= <dfeat, MLP_k(eo)>, the gradient of ef_k)
//   T[r,k,:] <- ef[r][k] * T[r,k,:] where Hc > 0, else 0        (= the gradient of the pre-activation)
__global__ __launch_bounds__(kB) void typed_bwd_kernel(float* __restrict__ T, const float* __restrict__ Hc,
                                                       const float* __restrict__ ef, const float* __restrict__ dfeat,
                                                       const float* __restrict__ b2, float* __restrict__ def,
                                                       long long rows, int K, int hid, int ld_ef) {
  const int lane = threadIdx.x & 63;
  for (long long r = (long long)blockIdx.x * (kB / 64) + (threadIdx.x >> 6); r < rows;
       r += (long long)gridDim.x * (kB / 64)) {
    const float g = dfeat[r * GN_FEAT + lane];
    for (int k = 0; k < K; ++k) {
      const size_t base = ((size_t)r * K + k) * hid;
      const float e = ef[r * ld_ef + k];
      float v = g * b2[k * GN_FEAT + lane];
      for (int c = lane; c < hid; c += 64) {
        const float t = T[base + c], h = Hc[base + c];
        v = fmaf(t, h, v);
        T[base + c] = h > 0.f ? e * t : 0.f;
      }
      v = gn_wave_sum(v);
      if (lane == 0) def[r * K + k] = v;
    }
  }
}

inline int cap_grid(long long items, int per_block, int cap = 4096) {
  long long g = (items + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

static int gemm_validate(const gn_gemm_desc_t& d) {
  if (d.A == nullptr || d.B == nullptr || d.C == nullptr) return GN_ERR_NULL;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || d.ldc < ((d.flags & GN_GEMM_TRANS_C) ? d.M : d.N)) return GN_ERR_SHAPE;
  if ((d.flags & GN_GEMM_TRANS_C) && !(d.flags & GN_GEMM_ACCUM)) return GN_ERR_SHAPE;
  const bool tA = d.flags & GN_GEMM_TRANS_A, tB = d.flags & GN_GEMM_TRANS_B;
  if (d.lda < (tA ? d.M : d.K) || d.ldb < (tB ? d.K : d.N)) return GN_ERR_SHAPE;
  if (d.mask && d.ldmask < d.N) return GN_ERR_SHAPE;
  if (d.colsum && !tA) return GN_ERR_SHAPE;
  if ((d.flags & GN_GEMM_ACCUM) && (d.bias || d.mask || (d.flags & GN_GEMM_RELU))) return GN_ERR_SHAPE;
  return GN_OK;
}

extern "C" int gn_gemm_grouped_f32(const gn_gemm_desc_t* descs, int n, gn_stream_t stream) {
  GN_REQUIRE_PTR(descs);
  if (n < 1) return GN_ERR_SHAPE;
  for (int i = 0; i < n; ++i) {
    const int rc = gemm_validate(descs[i]);
    if (rc != GN_OK) return rc;
  }
  hipStream_t s = (hipStream_t)stream;
  GemmTable T[2];            // [0] vector staging, [1] scalar staging; flushed when full
  long long tiles[2] = {0, 0};
  T[0].n = T[1].n = 0;
  auto flush = [&](int which) {
    if (T[which].n == 0) return;
    static const bool x6 = !(getenv("GN_GEMM_X6") && atoi(getenv("GN_GEMM_X6")) == 0);
    if (which == 0 && x6) hipLaunchKernelGGL(gemm_x6_kernel, dim3((unsigned)tiles[0]), dim3(kB), 0, s, T[0]);
    else if (which == 0) hipLaunchKernelGGL(gemm_mfma_kernel, dim3((unsigned)tiles[0]), dim3(kB), 0, s, T[0]);
    else hipLaunchKernelGGL(gemm_mfma_edge_kernel, dim3((unsigned)tiles[1]), dim3(kB), 0, s, T[1]);
    T[which].n = 0;
    tiles[which] = 0;
  };
  for (int i = 0; i < n; ++i) {
    const gn_gemm_desc_t& d = descs[i];
    const bool tA = d.flags & GN_GEMM_TRANS_A, tB = d.flags & GN_GEMM_TRANS_B;
    const bool vec = ((uintptr_t)d.A & 15) == 0 && ((uintptr_t)d.B & 15) == 0 && d.lda % 4 == 0 && d.ldb % 4 == 0 &&
                     d.K % BK == 0 && (!tA || (d.M % 4 == 0)) && (tB || (d.N % 4 == 0)) &&
                     (!d.rs || true);
    const int which = vec ? 0 : 1;
    const int gm = (d.M + BM - 1) / BM, gn = (d.N + BN - 1) / BN;
    int splits = 1;
    if (d.flags & GN_GEMM_ACCUM) {   // long K over few tiles: split K, partial sums by atomics
      // (swept on the whole training step at B=512: >= 256 rows per split, ~512 workgroups per problem)
      splits = (512 + gm * gn - 1) / (gm * gn);
      const int max_splits = (d.K + 255) / 256;
      splits = splits > max_splits ? max_splits : splits;
    }
    int kchunk = ((d.K + splits - 1) / splits + BK - 1) / BK * BK;
    splits = (d.K + kchunk - 1) / kchunk;
    const long long mine = (long long)gm * gn * splits;
    if (mine > 0x7fffffffLL) return GN_ERR_SHAPE;
    if (T[which].n == kMaxDescs || tiles[which] + mine > 0x7fffffffLL) flush(which);
    GemmDesc& g = T[which].d[T[which].n++];
    g.A = d.A; g.B = d.B; g.C = d.C; g.bias = d.bias; g.mask = d.mask; g.rs = d.rs; g.colsum = d.colsum;
    g.M = d.M; g.N = d.N; g.K = d.K; g.lda = d.lda; g.ldb = d.ldb; g.ldc = d.ldc; g.ldmask = d.ldmask;
    g.rs_ld = d.rs_ld; g.flags = d.flags & 31; g.alpha = d.alpha; g.beta = d.beta;
    g.tile0 = (int)tiles[which]; g.gn = gn; g.gmn = gm * gn; g.kchunk = kchunk;
    tiles[which] += mine;
  }
  flush(0);
  flush(1);
  return gn_check_launch();
}

extern "C" int gn_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                           int transA, int transB, const float* bias, const float* mask, int ldmask, int relu,
                           float alpha, float beta, gn_stream_t stream) {
  gn_gemm_desc_t d{};
  d.A = A; d.B = B; d.C = C; d.bias = bias; d.mask = mask;
  d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb; d.ldc = ldc; d.ldmask = ldmask;
  d.flags = (transA ? GN_GEMM_TRANS_A : 0) | (transB ? GN_GEMM_TRANS_B : 0) | (relu ? GN_GEMM_RELU : 0);
  d.alpha = alpha; d.beta = beta;
  const int rc = gemm_validate(d);
  if (rc != GN_OK) return rc;
  const long long tiles = (long long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  if (tiles < 128 && K >= 4096 && !relu && !mask && !bias) {
    // weight-gradient shape: C <- beta*C first, then the K split adds its partial sums atomically
    const long long total = (long long)M * N;
    hipLaunchKernelGGL(scale_kernel, dim3(cap_grid(total, kB)), dim3(kB), 0, (hipStream_t)stream, C, total, N, ldc, beta);
    d.flags |= GN_GEMM_ACCUM;
    d.beta = 0.f;
  }
  return gn_gemm_grouped_f32(&d, 1, stream);
}

extern "C" int gn_gumbel_bwd_f32(const float* dist, const float* lgf, const float* def, const float* gdist, float* dlgf,
                                 long long rows, int K, int ldl, float tau, int sym_N, gn_stream_t stream) {
  GN_REQUIRE_PTR(dist);
  GN_REQUIRE_PTR(lgf);
  GN_REQUIRE_PTR(def);
  GN_REQUIRE_PTR(dlgf);
  if (rows <= 0 || K < 1 || ldl <= K || !(tau > 0.f) || sym_N < 0) return GN_ERR_SHAPE;
  if (sym_N > 0 && rows % gn_pair_count(sym_N) != 0) return GN_ERR_SHAPE;
  hipLaunchKernelGGL(gumbel_bwd_kernel, dim3(cap_grid(rows, kB)), dim3(kB), 0, (hipStream_t)stream, dist, lgf, def,
                     gdist, dlgf, rows, K, ldl, tau, sym_N);
  return gn_check_launch();
}

extern "C" int gn_node2edge_bwd_f32(const float* xp, const float* pq, const float* H, const float* w2, const float* b2,
                                    const float* dedges, float* dxp, float* dpq, float* dw2, float* db2, int B, int N,
                                    int E, int sym, gn_stream_t stream) {
  const void* ptrs[] = {xp, pq, w2, b2, dedges, dxp, dpq, dw2, db2};
  for (const void* p : ptrs)
    if (p == nullptr) return GN_ERR_NULL;
  if (B <= 0 || N <= 0 || E <= 0) return GN_ERR_SHAPE;
  if (sym && H != nullptr) return GN_ERR_SHAPE;
  if (H == nullptr && E != (sym ? gn_pair_count(N) : N * N)) return GN_ERR_SHAPE;
  // one workgroup per scene while the scene's node rows (x', pq and their gradients) fit in LDS
  const size_t scene_lds = ((size_t)4 * N * GN_FEAT + 64 + (size_t)(kB / 64) * 4 * N) * sizeof(float);
  if (scene_lds <= 150 * 1024) {
    if (scene_lds > 64 * 1024) gn_allow_big_lds(node2edge_bwd_scene_kernel);
    hipLaunchKernelGGL(node2edge_bwd_scene_kernel, dim3((unsigned)B), dim3(kB), scene_lds, (hipStream_t)stream, xp, pq,
                       H, w2, b2, dedges, dxp, dpq, dw2, db2, N, E, sym);
    return gn_check_launch();
  }
  if (H == nullptr) return GN_ERR_LDS;     // large-N pairwise: hand in the explicit incidence
  const size_t lds = (size_t)(kB / 64) * 4 * N * sizeof(float);
  if (lds > 64 * 1024) return GN_ERR_LDS;
  const long long total = (long long)B * E;
  const long long grid = (total + 3) / 4;
  if (grid > 0x7fffffffLL) return GN_ERR_SHAPE;
  hipLaunchKernelGGL(node2edge_bwd_kernel, dim3((unsigned)grid), dim3(kB), lds, (hipStream_t)stream, xp, pq, H, w2, b2,
                     dedges, dxp, dpq, dw2, db2, N, E, total);
  return gn_check_launch();
}

extern "C" int gn_gumbel_bwd_grouped_f32(const gn_gumbel_bwd_group_t* groups, int n_groups, int ldl, float tau,
                                         gn_stream_t stream) {
  if (groups == nullptr) return GN_ERR_NULL;
  if (n_groups < 1 || n_groups > GN_MAX_GROUPS || !(tau > 0.f)) return GN_ERR_SHAPE;
  GumbelBwdTable T{};
  long long max_rows = 0;
  for (int g = 0; g < n_groups; ++g) {
    const gn_gumbel_bwd_group_t& G = groups[g];
    if (!G.dist || !G.lgf || !G.def || !G.dlgf) return GN_ERR_NULL;
    if (G.rows <= 0 || G.K < 1 || ldl <= G.K || G.sym_N < 0) return GN_ERR_SHAPE;
    if (G.sym_N > 0 && G.rows % gn_pair_count(G.sym_N) != 0) return GN_ERR_SHAPE;
    T.g[g] = G;
    max_rows = G.rows > max_rows ? G.rows : max_rows;
  }
  hipLaunchKernelGGL(gumbel_bwd_grouped_kernel, dim3(cap_grid(max_rows, kB), n_groups), dim3(kB), 0, (hipStream_t)stream, T,
                     ldl, tau);
  return gn_check_launch();
}

extern "C" int gn_node2edge_bwd_grouped_f32(const gn_n2e_bwd_group_t* groups, int n_groups, int B, int N,
                                            gn_stream_t stream) {
  if (groups == nullptr) return GN_ERR_NULL;
  if (n_groups < 1 || n_groups > GN_MAX_GROUPS || B <= 0 || N <= 0) return GN_ERR_SHAPE;
  const size_t scene_lds = ((size_t)4 * N * GN_FEAT + 64 + (size_t)(kB / 64) * 4 * N) * sizeof(float);
  if (scene_lds > 150 * 1024) return GN_ERR_LDS;      // (the per-scene form only: larger N goes module by module)
  N2EBwdTable T{};
  for (int g = 0; g < n_groups; ++g) {
    const gn_n2e_bwd_group_t& G = groups[g];
    const void* ptrs[] = {G.xp, G.pq, G.w2, G.b2, G.dedges, G.dxp, G.dpq, G.dw2, G.db2};
    for (const void* p : ptrs)
      if (p == nullptr) return GN_ERR_NULL;
    if (G.E <= 0 || (G.sym && G.H != nullptr)) return GN_ERR_SHAPE;
    if (G.H == nullptr && G.E != (G.sym ? gn_pair_count(N) : N * N)) return GN_ERR_SHAPE;
    T.g[g] = G;
  }
  if (scene_lds > 64 * 1024) gn_allow_big_lds(node2edge_bwd_scene_grouped_kernel);
  hipLaunchKernelGGL(node2edge_bwd_scene_grouped_kernel, dim3((unsigned)B, n_groups), dim3(kB), scene_lds,
                     (hipStream_t)stream, T, N);
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
                                int sym_N, float diag_w, int ld_ef, gn_stream_t stream) {
  GN_REQUIRE_PTR(dist);
  GN_REQUIRE_PTR(lgf);
  GN_REQUIRE_PTR(ef);
  if (rows <= 0 || K < 1 || ldl <= K || sym_N < 0 || ld_ef < K) return GN_ERR_SHAPE;
  if (sym_N > 0 && rows % gn_pair_count(sym_N) != 0) return GN_ERR_SHAPE;
  hipLaunchKernelGGL(gumbel_ef_kernel, dim3(cap_grid(rows * K, kB)), dim3(kB), 0, (hipStream_t)stream, dist, lgf, ef,
                     rows, K, ldl, sym_N, diag_w, ld_ef);
  return gn_check_launch();
}

extern "C" int gn_typed_bwd_f32(float* T, const float* Hc, const float* ef, int ld_ef, const float* dfeat,
                                const float* b2, float* def, long long rows, int K, int hid, gn_stream_t stream) {
  const void* ptrs[] = {T, Hc, ef, dfeat, b2, def};
  for (const void* p : ptrs)
    if (p == nullptr) return GN_ERR_NULL;
  if (rows <= 0 || K < 1 || hid < 64 || hid % 64 || ld_ef < K) return GN_ERR_SHAPE;
  hipLaunchKernelGGL(typed_bwd_kernel, dim3(cap_grid(rows, kB / 64, 8192)), dim3(kB), 0, (hipStream_t)stream, T, Hc, ef,
                     dfeat, b2, def, rows, K, hid, ld_ef);
  return gn_check_launch();
}
