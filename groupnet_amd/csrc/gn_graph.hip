// Per-scene graph kernels of the GroupNet MS-HGNN path for gfx950: cosine affinity, top-k
// hyperedge incidence, attention-weighted node->edge pooling, hyperedge gather / scatter
// aggregation, and the Philox noise source.  All of them are byte-moving / VALU work bounded
// by HBM, so the design rules are: coalesced 16-byte global accesses, the
// scene's N x 64 feature tile and N x N (or E x N) incidence tile staged ONCE through LDS,
// wavefront-shuffle reductions, and grids of >> 256 workgroups.
#include "gn_mlp_common.hpp"
#include "gn_affinity.hpp"

// hyperedges per wave and band whose H rows the node->edge kernel holds in registers (band = 4 waves x this many)
#ifndef GN_N2E_U
#define GN_N2E_U 4
#endif

namespace {

constexpr int kBlock = 256;
constexpr size_t kLdsBudget = 128 * 1024;  // per workgroup (of the CU's 160 KiB)

// --------------------------------------------------------------------------------------------
// A0 + A1: affinity and top-k incidence
// --------------------------------------------------------------------------------------------
// NaN ranks above every number (torch.topk semantics); ties go to the lower index.
__device__ __forceinline__ bool beats(float vj, int j, float vc, int c) {
  const bool nj = vj != vj, nc = vc != vc;
  if (nj || nc) return (nj && !nc) || (nj && nc && j < c);
  return vj > vc || (vj == vc && j < c);
}

// rank of column c inside `row` (LDS, N entries) and the writes of every scale
template <typename T>
__device__ __forceinline__ void emit_incidence(const float* row, int N, int b, int i, int c, const ScaleList& sl) {
  T* H_cat = reinterpret_cast<T*>(sl.H_cat);   // the concatenation is what the caller returns: storage type T
  const float v = row[c];
  int rank = 0;
  for (int j = 0; j < N; ++j) rank += beats(row[j], j, v, c) ? 1 : 0;
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

// (rank_key, emit_ranked and the body of the fused affinity + top-k launch live in gn_affinity.hpp: the node stage's launch
// runs the same body in its tail workgroups)
template <typename T>
__global__ __launch_bounds__(kBlock) void affinity_topk_kernel(const T* __restrict__ f, float* __restrict__ corr,
                                                               ScaleList sl, int N, int D, gn_block_extras_t ex) {
  extern __shared__ __align__(16) float lds[];
  affinity_topk_body<T>(f, corr, sl, N, D, ex, (int)blockIdx.x, lds);
}

// Large-N affinity: one workgroup per (scene, 16-row band); the band and one 64-column
// panel of q at a time live in LDS.
__global__ __launch_bounds__(kBlock) void affinity_banded_kernel(const float* __restrict__ f,
                                                                 float* __restrict__ corr, int N, int D) {
  extern __shared__ __align__(16) float lds[];
  constexpr int RB = 16, CB = 64;
  const int b = blockIdx.y, i0 = blockIdx.x * RB;
  const int ldq = D + 4, d4 = D >> 2;
  float* qa = lds;             // RB x ldq
  float* qb = lds + RB * ldq;  // CB x ldq
  const float* fb = f + (size_t)b * N * D;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  auto stage = [&](float* dst, int r0, int nr) {
    for (int idx = threadIdx.x; idx < nr * d4; idx += kBlock) {
      const int r = idx / d4, cc = idx - r * d4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r0 + r < N) v = *reinterpret_cast<const f32x4*>(fb + (size_t)(r0 + r) * D + 4 * cc);
      *reinterpret_cast<f32x4*>(dst + r * ldq + 4 * cc) = v;
    }
    __syncthreads();
    for (int r = wave; r < nr; r += kBlock / 64) {
      float ss = 0.f;
      for (int d = lane; d < D; d += 64) ss += dst[r * ldq + d] * dst[r * ldq + d];
      ss = gn_wave_sum(ss);
      const float denom = fmaxf(sqrtf(ss), 1e-12f);
      for (int d = lane; d < D; d += 64) dst[r * ldq + d] = dst[r * ldq + d] / denom;
    }
    __syncthreads();
  };
  stage(qa, i0, RB);
  for (int j0 = 0; j0 < N; j0 += CB) {
    stage(qb, j0, CB);
    for (int idx = threadIdx.x; idx < RB * CB; idx += kBlock) {
      const int i = idx / CB, j = idx - i * CB;
      if (i0 + i < N && j0 + j < N) {
        const f32x4* a = reinterpret_cast<const f32x4*>(qa + i * ldq);
        const f32x4* c = reinterpret_cast<const f32x4*>(qb + j * ldq);
        float acc = 0.f;
        for (int d = 0; d < d4; ++d) {
          const f32x4 x = a[d], y = c[d];
          acc = fmaf(x[0], y[0], acc);
          acc = fmaf(x[1], y[1], acc);
          acc = fmaf(x[2], y[2], acc);
          acc = fmaf(x[3], y[3], acc);
        }
        corr[((size_t)b * N + i0 + i) * N + j0 + j] = acc;
      }
    }
    __syncthreads();
  }
}

// Stand-alone top-k incidence: one workgroup per (scene, band of RB rows of corr).
__global__ __launch_bounds__(kBlock) void topk_incidence_kernel(const float* __restrict__ corr, ScaleList sl, int N,
                                                                int RB) {
  extern __shared__ __align__(16) float lds[];  // RB x N
  const int b = blockIdx.y, i0 = blockIdx.x * RB;
  const int nr = min(RB, N - i0);
  const float* src = corr + ((size_t)b * N + i0) * N;
  for (int idx = threadIdx.x; idx < nr * N; idx += kBlock) lds[idx] = src[idx];
  __syncthreads();
  for (int idx = threadIdx.x; idx < nr * N; idx += kBlock) {
    const int i = idx / N, c = idx - i * N;
    emit_incidence<float>(lds + i * N, N, b, i0 + i, c, sl);
  }
}

// --------------------------------------------------------------------------------------------
// group tables (kernel arguments, by value): one launch serves the same stage of several modules
// --------------------------------------------------------------------------------------------
template <typename G>
struct WaveTable {
  G g[GN_MAX_GROUPS];
  long long first[GN_MAX_GROUPS + 1];  // prefix of work items (meaning depends on the kernel)
  int n;
};
template <typename G>
__device__ __forceinline__ int find_wave_group(const WaveTable<G>& t, long long item) {
  int g = 0;
  while (g + 1 < t.n && item >= t.first[g + 1]) ++g;
  return g;
}

// --------------------------------------------------------------------------------------------
// A3 second half: attention-weighted node -> edge pooling
// --------------------------------------------------------------------------------------------
// Hyper modules.  A workgroup stages the x' and pq rows of SG scenes of ONE module in LDS (pq rows padded to 65
// floats) and walks those scenes' hyperedges in bands of EB.  Per band, five thread-parallel phases (a barrier
// between them), every one with a THREAD per output element instead of a wave per hyperedge — the wave-per-edge form
// spent ~800 wave instructions per hyperedge on wave-uniform work (the softmax weight of every member evaluated by
// all 64 lanes, a 5-step shuffle reduction per member) and took 237 us at N = 50, B = 1024:
//   P0  member lists: the nodes with H != 0 of each edge, compacted with a ballot (one wave per edge; non-members
//       enter the softmax only as exp(0 - max) terms, exactly as softmax(att * H) treats them,
//       MS_HGNN_batch.py:135-137,366-368);
//   P1  Q_e[c] = sum_m H[e,m] Qn_m[c]                                   thread = (edge, channel)
//   P2  v[e,m] = H[e,m] (b2 + sum_c w2[c] relu(P_m[c] + Q_e[c]))        thread = (edge, member), w2 in SGPRs
//   P3  softmax over all N nodes, weight[e,m] = softmax * H[e,m]        half-wave = edge
//   P4  edges[e] = sum_m weight[e,m] x'_m                               thread = (edge, 4 features)
// `first` counts workgroups.
__device__ __forceinline__ float gn_half_max(float v) {
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, GN_WAVE));
  return v;
}
__host__ __device__ inline size_t n2e_hyper_scratch_floats(int EB, int N) {
  return (size_t)3 * EB * N + (size_t)EB * 33 + (size_t)3 * EB + 8;
}
template <typename TS>
__device__ __forceinline__ void node2edge_hyper_body(const WaveTable<gn_n2e_group_t>& T, int B, int N, int SG, int EB,
                                                     int wg) {
  extern __shared__ __align__(16) float lds[];
  constexpr int LDP = GN_FEAT + 1, LDQ = 33;
  const int wave = gn_uniform((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int gi = gn_uniform(find_wave_group(T, wg));
  const gn_n2e_group_t G = T.g[gi];
  const int E = G.E;
  const int b0 = (int)(wg - T.first[gi]) * SG;
  const int sg = min(SG, B - b0);
  float* s_xp = lds;                                   // sg x N x 64
  float* s_pq = s_xp + (size_t)SG * N * GN_FEAT;       // sg x N x 65
  float* sc = s_pq + (size_t)SG * N * LDP;             // band scratch
  int* s_idx = reinterpret_cast<int*>(sc);             // EB x N   member -> node
  float* s_h = sc + (size_t)EB * N;                    // EB x N   H[e, member]
  float* s_v = s_h + (size_t)EB * N;                   // EB x N   att * H, then softmax weight * H
  float* s_Q = s_v + (size_t)EB * N;                   // EB x 33
  int* s_cnt = reinterpret_cast<int*>(s_Q + (size_t)EB * LDQ);   // EB
  int* s_wmax = s_cnt + EB;                            // 4: the waves' largest member counts
  float w2r[32];                                        // (uniform addresses: scalar loads, the values stay in SGPRs)
#pragma unroll
  for (int c = 0; c < 32; ++c) w2r[c] = G.w2[c];
  const float b2v = *G.b2;
  const int total = sg * E;
  // H rows ride one band ahead in registers (N <= 64: one value per lane and edge, up to U edges per wave and band):
  // a wave that loaded each row when it needed it paid one memory latency per hyperedge.
  constexpr int U = GN_N2E_U;
  const bool pre = N <= 64 && EB <= U * (kBlock / 64);
  float hn[U];
  auto fetch = [&](int e0n) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int le = wave + u * (kBlock / 64);
      hn[u] = (pre && le < EB && e0n + le < total && lane < N) ? G.H[((size_t)b0 * E + e0n + le) * N + lane] : 0.f;
    }
  };
  fetch(0);
  {
    const TS* src = reinterpret_cast<const TS*>(G.xp) + (size_t)b0 * N * GN_FEAT;
    f32x4* dst = reinterpret_cast<f32x4*>(s_xp);
    for (int idx = threadIdx.x; idx < sg * N * 16; idx += kBlock) dst[idx] = ld4(src + 4 * idx);
    const TS* psrc = reinterpret_cast<const TS*>(G.pq) + (size_t)b0 * N * GN_FEAT;
    for (int idx = threadIdx.x; idx < sg * N * 16; idx += kBlock) {
      const f32x4 v = ld4(psrc + 4 * idx);
      float* d = s_pq + (idx >> 4) * LDP + (idx & 15) * 4;
      d[0] = v[0];
      d[1] = v[1];
      d[2] = v[2];
      d[3] = v[3];
    }
  }
  for (int e0 = 0; e0 < total; e0 += EB) {
    const int eb = min(EB, total - e0);
    __syncthreads();            // the staged rows are visible / the previous band's scratch is free
    // ---- P0: member lists ----
    int wmax = 0;
    if (pre) {
      float hc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) hc[u] = hn[u];
      fetch(e0 + EB);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int le = wave + u * (kBlock / 64);
        if (le < eb) {
          const float hv = hc[u];
          const unsigned long long mask = __ballot(hv != 0.f);
          if (hv != 0.f) {
            const int pos = __popcll(mask & ((1ull << lane) - 1ull));
            s_idx[le * N + pos] = lane;
            s_h[le * N + pos] = hv;
          }
          const int cnt = __popcll(mask);
          if (lane == 0) s_cnt[le] = cnt;
          wmax = max(wmax, cnt);
        }
      }
    } else {
      for (int le = wave; le < eb; le += kBlock / 64) {
        const float* Hrow = G.H + ((size_t)b0 * E + e0 + le) * N;
        int cnt = 0;
        for (int n0 = 0; n0 < N; n0 += 64) {
          const int n = n0 + lane;
          const float hv = n < N ? Hrow[n] : 0.f;
          const unsigned long long mask = __ballot(hv != 0.f);
          if (hv != 0.f) {
            const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
            s_idx[le * N + pos] = n;
            s_h[le * N + pos] = hv;
          }
          cnt += __popcll(mask);
        }
        if (lane == 0) s_cnt[le] = cnt;
        wmax = max(wmax, cnt);
      }
    }
    if (lane == 0) s_wmax[wave] = wmax;
    __syncthreads();
    const int kmax = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));
    // ---- P1: Q_e ----
    for (int it = threadIdx.x; it < eb * 32; it += kBlock) {
      const int le = it >> 5, c = it & 31;
      const float* pqb = s_pq + (size_t)((e0 + le) / E) * N * LDP + 32 + c;
      const int cnt = s_cnt[le];
      float q = 0.f;
#pragma unroll 4
      for (int m = 0; m < cnt; ++m) q = fmaf(s_h[le * N + m], pqb[s_idx[le * N + m] * LDP], q);
      s_Q[le * LDQ + c] = q;
    }
    __syncthreads();
    // ---- P2: attention logits times H ----
    for (int it = threadIdx.x; it < eb * kmax; it += kBlock) {
      const int le = it / kmax, m = it - le * kmax;
      if (m < s_cnt[le]) {
        const float* p = s_pq + ((size_t)((e0 + le) / E) * N + s_idx[le * N + m]) * LDP;
        const float* Qe = s_Q + le * LDQ;
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) t = fmaf(w2r[c], fmaxf(p[c] + Qe[c], 0.f), t);
        s_v[le * N + m] = (t + b2v) * s_h[le * N + m];
      }
    }
    __syncthreads();
    // ---- P3: softmax over all N nodes of v_n = att_n * H[e,n] (0 for non-members); weight = softmax * H ----
    for (int le = threadIdx.x >> 5; le < eb; le += kBlock / 32) {
      const int c = threadIdx.x & 31;
      const int cnt = s_cnt[le];
      float mx = cnt < N ? 0.f : -INFINITY;
      for (int m = c; m < cnt; m += 32) mx = fmaxf(mx, s_v[le * N + m]);
      mx = gn_half_max(mx);
      float sum = 0.f;
      for (int m = c; m < cnt; m += 32) sum += expf(s_v[le * N + m] - mx);
      sum = gn_half_sum(sum);
      sum += gn_nonmember_sum(N - cnt, mx);
      for (int m = c; m < cnt; m += 32) s_v[le * N + m] = expf(s_v[le * N + m] - mx) / sum * s_h[le * N + m];
    }
    __syncthreads();
    // ---- P4: edges[e] = sum_m weight_m x'_m ----
    for (int it = threadIdx.x; it < eb * 16; it += kBlock) {
      const int le = it >> 4, f4 = it & 15;
      const f32x4* xpb = reinterpret_cast<const f32x4*>(s_xp + (size_t)((e0 + le) / E) * N * GN_FEAT) + f4;
      const int cnt = s_cnt[le];
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int m = 0; m < cnt; ++m) {
        const float w = s_v[le * N + m];
        const f32x4 x = xpb[s_idx[le * N + m] * (GN_FEAT / 4)];
        acc[0] = fmaf(w, x[0], acc[0]);
        acc[1] = fmaf(w, x[1], acc[1]);
        acc[2] = fmaf(w, x[2], acc[2]);
        acc[3] = fmaf(w, x[3], acc[3]);
      }
      st4(reinterpret_cast<TS*>(G.edges) + ((size_t)b0 * E + e0 + le) * GN_FEAT + 4 * f4, acc);
    }
  }
}

// Hyper modules, one LANE PAIR per hyperedge (N <= 64).  The banded form above walks each band of hyperedges through five
// barrier-separated phases (member lists, Q, logits, softmax, pooling) that hand everything over through LDS scratch:
// at N = 50, B = 1024 a workgroup lived ~25 us for a few microseconds of arithmetic (20 barriers, 4 workgroups per
// CU).  Here the scenes' x' / pq rows are staged once (row pitch 68 floats: lanes that read different rows at the same
// offset hit different banks) and lane (r, h) of a row's pair then does the whole row on its own 16 of the 32 attention
// channels / 32 of the 64 features: members from a 64-bit mask of its incidence row, Q = sum H Qn, the logits twice
// (first pass: running max and sum of the softmax over ALL N nodes; second pass: the weights), the pooled features —
// no scratch, no barrier after the stage.  Same formulas as the banded form (softmax(att*H)*H incl. the non-members'
// exp(0 - max)); the running max/sum rounds differently in the last bits.
constexpr int kRowPitch = GN_FEAT + 4;
template <typename TS>
__device__ __forceinline__ void node2edge_hyper_rows_body(const WaveTable<gn_n2e_group_t>& T, int B, int N, int SG, int wg) {
  extern __shared__ __align__(16) float lds[];
  const int gi = gn_uniform(find_wave_group(T, wg));
  const gn_n2e_group_t G = T.g[gi];
  const int E = G.E;
  const int b0 = (int)(wg - T.first[gi]) * SG;
  const int sg = min(SG, B - b0);
  float* s_xp = lds;                                      // sg x N x 68
  float* s_pq = s_xp + (size_t)SG * N * kRowPitch;        // sg x N x 68
  {
    const TS* xs = reinterpret_cast<const TS*>(G.xp) + (size_t)b0 * N * GN_FEAT;
    const TS* ps = reinterpret_cast<const TS*>(G.pq) + (size_t)b0 * N * GN_FEAT;
    for (int idx = threadIdx.x; idx < sg * N * 16; idx += kBlock) {
      const int r = idx >> 4, c = idx & 15;
      *reinterpret_cast<f32x4*>(s_xp + r * kRowPitch + 4 * c) = ld4(xs + 4 * idx);
      *reinterpret_cast<f32x4*>(s_pq + r * kRowPitch + 4 * c) = ld4(ps + 4 * idx);
    }
  }
  const int lane = threadIdx.x & 63, h = lane >> 5;
  float w2[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) w2[c] = G.w2[16 * h + c];
  const float b2v = *G.b2;
  __syncthreads();
  const int total = sg * E;
  for (int r0 = 0; r0 < total; r0 += kBlock / 2) {
    const int r = r0 + (threadIdx.x >> 6) * 32 + (lane & 31);       // this lane pair's hyperedge (wave w: rows 32w ..)
    const bool live = r < total;
    const int rr = live ? r : total - 1;
    const int s = rr / E;
    const float* Hrow = G.H + ((size_t)b0 * E + rr) * N;
    const float* pqb = s_pq + (size_t)s * N * kRowPitch;
    const float* xpb = s_xp + (size_t)s * N * kRowPitch;
    // members: lane h scans the nodes n = h, h+2, ...; the pair ORs its halves
    unsigned long long mask = 0ull;
    for (int n = h; n < N; n += 2) mask |= (unsigned long long)(Hrow[n] != 0.f) << n;
    {
      const unsigned lo = (unsigned)mask, hi = (unsigned)(mask >> 32);
      const unsigned olo = (unsigned)__shfl_xor((int)lo, 32, GN_WAVE), ohi = (unsigned)__shfl_xor((int)hi, 32, GN_WAVE);
      mask |= ((unsigned long long)ohi << 32) | olo;
    }
    const int cnt = __popcll(mask);
    float Q[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) Q[c] = 0.f;
    f32x4 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto logit = [&](int n, float hv) {                    // (att[e,n] + b2) * H[e,n]
      const float* pp = pqb + n * kRowPitch + 16 * h;
      float t = 0.f;
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(pp + 4 * c4);
#pragma unroll
        for (int c = 0; c < 4; ++c) t = fmaf(w2[4 * c4 + c], fmaxf(v[c] + Q[4 * c4 + c], 0.f), t);
      }
      t += __shfl_xor(t, 32, GN_WAVE);
      return (t + b2v) * hv;
    };
    constexpr int MK = 16;
    if (__all(cnt <= MK)) {
      // up to 16 members (every top-k scale of the reference): member list and logits in registers, loops unrolled so
      // that the LDS reads of several members are in flight together; max first, then the sum, as the banded form
      int mem[MK];
      float hvv[MK], vv[MK];
      {
        unsigned long long m = mask;
#pragma unroll
        for (int k = 0; k < MK; ++k) {
          mem[k] = m != 0ull ? __builtin_ctzll(m) : 0;
          hvv[k] = k < cnt ? Hrow[mem[k]] : 0.f;
          m &= m - 1ull;                                   // (0 stays 0)
        }
      }
#pragma unroll
      for (int k = 0; k < MK; ++k)
        if (k < cnt) {
          const float* q = pqb + mem[k] * kRowPitch + 32 + 16 * h;
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(q + 4 * c4);
#pragma unroll
            for (int c = 0; c < 4; ++c) Q[4 * c4 + c] = fmaf(hvv[k], v[c], Q[4 * c4 + c]);
          }
        }
      float mx = cnt < N ? 0.f : -INFINITY;
#pragma unroll
      for (int k = 0; k < MK; ++k) {
        vv[k] = 0.f;
        if (__any(k < cnt)) {                              // (the shuffle inside needs both lanes of a pair)
          const float v = logit(mem[k], hvv[k]);
          if (k < cnt) {
            vv[k] = v;
            mx = fmaxf(mx, v);
          }
        }
      }
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < MK; ++k)
        if (k < cnt) {
          vv[k] = expf(vv[k] - mx);
          sum += vv[k];
        }
      sum += gn_nonmember_sum(N - cnt, mx);
#pragma unroll
      for (int k = 0; k < MK; ++k)
        if (k < cnt) {
          const float w = vv[k] / sum * hvv[k];
          const float* x = xpb + mem[k] * kRowPitch + 32 * h;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * q);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[q][c] = fmaf(w, v[c], acc[q][c]);
          }
        }
    } else {
      // any number of members: running max / sum of the softmax, logits evaluated a second time for the weights
      for (unsigned long long m = mask; m != 0ull; m &= m - 1ull) {
        const int n = __builtin_ctzll(m);
        const float hv = Hrow[n];
        const float* q = pqb + n * kRowPitch + 32 + 16 * h;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(q + 4 * c4);
#pragma unroll
          for (int c = 0; c < 4; ++c) Q[4 * c4 + c] = fmaf(hv, v[c], Q[4 * c4 + c]);
        }
      }
      // (lanes of rows with fewer members idle through the longer rows' iterations: the shuffle needs the whole pair)
      float mx = cnt < N ? 0.f : -INFINITY, sum = 0.f;
      for (unsigned long long m = mask; __any(m != 0ull); m &= m - 1ull) {
        const bool on = m != 0ull;
        const int n = on ? __builtin_ctzll(m) : 0;
        const float v = logit(n, Hrow[n]);
        if (on) {
          const float nm = fmaxf(mx, v);
          sum = sum * expf(mx - nm) + expf(v - nm);
          mx = nm;
        }
      }
      sum += gn_nonmember_sum(N - cnt, mx);
      for (unsigned long long m = mask; __any(m != 0ull); m &= m - 1ull) {
        const bool on = m != 0ull;
        const int n = on ? __builtin_ctzll(m) : 0;
        const float hv = Hrow[n];
        const float lv = logit(n, hv);
        if (on) {
          const float w = expf(lv - mx) / sum * hv;
          const float* x = xpb + n * kRowPitch + 32 * h;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * q);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[q][c] = fmaf(w, v[c], acc[q][c]);
          }
        }
      }
    }
    if (live) {
      TS* out = reinterpret_cast<TS*>(G.edges) + ((size_t)b0 * E + r) * GN_FEAT + 32 * h;
#pragma unroll
      for (int q = 0; q < 8; ++q) st4(out + 4 * q, acc[q]);
    }
  }
}

// Pairwise module (MS_HGNN_oridinary): edge e = i*N + j touches i and j with weight 1 (2 on the
// diagonal), so only att[e,i] and att[e,j] matter and Q_e = Qn_i + Qn_j.  A workgroup stages the
// x' and pq rows of SG scenes in LDS (pq rows padded to 65 floats: lanes read different rows at the
// same channel), then walks bands of 256 edges: phase A one thread per edge evaluates the two
// attention logits and the softmax weights, phase B the band's 256 x 16 float4 outputs are written
// as consecutive 16-byte pieces.
template <typename TS>
__device__ __forceinline__ void node2edge_pairwise_body(const gn_n2e_group_t& G, int B, int N, int SG, int bands,
                                                        int wg) {
  extern __shared__ __align__(16) float lds[];
  constexpr int LDP = GN_FEAT + 1;
  const bool sym = G.sym != 0;
  const float b2v = *G.b2;
  const int E = sym ? gn_pair_count(N) : N * N;   // edge rows per scene
  const int b0 = (wg / bands) * SG;
  const int band = wg % bands;
  const int sg = min(SG, B - b0);
  float* s_xp = lds;                                // sg x N x 64
  float* s_pq = s_xp + (size_t)SG * N * GN_FEAT;    // sg x N x 65
  float* s_w = s_pq + (size_t)SG * N * LDP;         // 256 x 2 edge weights of the current band
  float* s_w2 = s_w + 2 * kBlock;                   // 32
  int* s_ij = reinterpret_cast<int*>(s_w2 + 32);    // 256 packed (i, j) of the current band
  {
    const TS* src = reinterpret_cast<const TS*>(G.xp) + (size_t)b0 * N * GN_FEAT;
    f32x4* dst = reinterpret_cast<f32x4*>(s_xp);
    for (int idx = threadIdx.x; idx < sg * N * 16; idx += kBlock) dst[idx] = ld4(src + 4 * idx);
    const TS* psrc = reinterpret_cast<const TS*>(G.pq) + (size_t)b0 * N * GN_FEAT;
    for (int idx = threadIdx.x; idx < sg * N * GN_FEAT; idx += kBlock) {
      const int r = idx >> 6, cc = idx & 63;
      s_pq[r * LDP + cc] = ld1(psrc + idx);
    }
    if (threadIdx.x < 32) s_w2[threadIdx.x] = G.w2[threadIdx.x];
  }
  __syncthreads();
  const long long total = (long long)sg * E;  // edges of this workgroup's scenes
  const long long per_band = ((total + bands - 1) / bands + kBlock - 1) / kBlock * kBlock;
  const long long lo = band * per_band, hi = min(total, lo + per_band);
  for (long long base = lo; base < hi; base += kBlock) {
    const long long eidx = base + threadIdx.x;
    if (eidx < hi) {
      const int s = (int)(eidx / E), e = (int)(eidx - (long long)s * E);
      int i, j;
      if (sym) {
        gn_pair_decode(e, N, i, j);
      } else {
        i = e / N;
        j = e - i * N;
      }
      s_ij[threadIdx.x] = i | (j << 16);
      const float* pi = s_pq + (size_t)(s * N + i) * LDP;
      const float* pj = s_pq + (size_t)(s * N + j) * LDP;
      float ai = 0.f, aj = 0.f;
      if (i == j) {
#pragma unroll 8
        for (int cch = 0; cch < 32; ++cch) ai = fmaf(s_w2[cch], fmaxf(pi[cch] + 2.f * pi[32 + cch], 0.f), ai);
        ai += b2v;
        // H = 2 on the self-loop: v = 2*att, the other N-1 nodes contribute exp(0)
        const float v = 2.f * ai;
        const float mx = N > 1 ? fmaxf(v, 0.f) : v;
        const float ev = expf(v - mx);
        const float sum = ev + gn_nonmember_sum(N - 1, mx);
        s_w[2 * threadIdx.x] = ev / sum * 2.f;
        s_w[2 * threadIdx.x + 1] = 0.f;
      } else {
#pragma unroll 8
        for (int cch = 0; cch < 32; ++cch) {
          const float q = pi[32 + cch] + pj[32 + cch];
          ai = fmaf(s_w2[cch], fmaxf(pi[cch] + q, 0.f), ai);
          aj = fmaf(s_w2[cch], fmaxf(pj[cch] + q, 0.f), aj);
        }
        ai += b2v;
        aj += b2v;
        const float mx = N > 2 ? fmaxf(fmaxf(ai, aj), 0.f) : fmaxf(ai, aj);
        const float ei = expf(ai - mx), ej = expf(aj - mx);
        const float sum = (ei + ej) + gn_nonmember_sum(N - 2, mx);
        s_w[2 * threadIdx.x] = ei / sum;
        s_w[2 * threadIdx.x + 1] = ej / sum;
      }
    }
    __syncthreads();
    const int nb = (int)min((long long)kBlock, hi - base);
    TS* dst = reinterpret_cast<TS*>(G.edges) + ((size_t)b0 * E + base) * GN_FEAT;
    for (int idx = threadIdx.x; idx < nb * 16; idx += kBlock) {
      const int t = idx >> 4, d = idx & 15;
      const long long eidx = base + t;
      const int s = (int)(eidx / E);
      const int ij = s_ij[t];
      const int i = ij & 0xffff, j = ij >> 16;
      const float wi = s_w[2 * t], wj = s_w[2 * t + 1];
      const f32x4 xi = reinterpret_cast<const f32x4*>(s_xp + (size_t)(s * N + i) * GN_FEAT)[d];
      const f32x4 xj = reinterpret_cast<const f32x4*>(s_xp + (size_t)(s * N + j) * GN_FEAT)[d];
      f32x4 r = {fmaf(wj, xj[0], wi * xi[0]), fmaf(wj, xj[1], wi * xi[1]), fmaf(wj, xj[2], wi * xi[2]),
                 fmaf(wj, xj[3], wi * xi[3])};
      st4(dst + 4 * idx, r);
    }
    __syncthreads();
  }
}

// One launch for every module of a multiscale forward: the first `pair.first_wg[pair.n]` workgroups walk the
// pairwise groups, the rest the hyper groups (both scene-staged).
struct PairTable {
  gn_n2e_group_t g[GN_MAX_GROUPS];
  int first_wg[GN_MAX_GROUPS + 1];
  int SG[GN_MAX_GROUPS], bands[GN_MAX_GROUPS];
  int n;
};
// ROWS: the hyper groups in the lane-pair-per-hyperedge form (a kernel of its own: that form keeps member lists and
// logits in registers, which the banded form's occupancy must not pay for)
template <typename TS, bool ROWS>
__global__ __launch_bounds__(kBlock) void node2edge_kernel(WaveTable<gn_n2e_group_t> T, PairTable pair, int B, int N,
                                                           int SGh, int EBh, XcdSections xs) {
  const int wg = gn_uniform(gn_xcd_logical(xs, blockIdx.x));     // sections: the pairwise groups, then the hyper groups
  if (wg < 0) return;
  const int n_pair_wgs = pair.first_wg[pair.n];
  if (wg < n_pair_wgs) {
    int g = 0;
    while (g + 1 < pair.n && wg >= pair.first_wg[g + 1]) ++g;
    node2edge_pairwise_body<TS>(pair.g[g], B, N, pair.SG[g], pair.bands[g], wg - pair.first_wg[g]);
  } else if constexpr (ROWS) {
    node2edge_hyper_rows_body<TS>(T, B, N, SGh, wg - n_pair_wgs);      // one lane pair per hyperedge (N <= 64)
  } else {
    node2edge_hyper_body<TS>(T, B, N, SGh, EBh, wg - n_pair_wgs);
  }
}

// --------------------------------------------------------------------------------------------
// A5: hyperedge aggregation — gather (eo = H ori) and scatter (out = cat(H^T feat, ori) / N)
// --------------------------------------------------------------------------------------------
// A workgroup owns G consecutive scenes x an edge band [e0, e0+TE) of the group blockIdx.z.  The
// scenes' ori tiles and the H band are contiguous in HBM, so they stream into LDS as whole 16-byte
// pieces; each thread then produces float4 outputs that are again contiguous across the workgroup.
struct GatherTable {
  gn_gather_group_t g[GN_MAX_GROUPS];
};
template <typename TS>
__global__ __launch_bounds__(kBlock) void agg_gather_kernel(GatherTable T, int B, int N, int G, int TE) {
  extern __shared__ __align__(16) float lds[];
  const gn_gather_group_t Gr = T.g[blockIdx.z];
  const int E = Gr.E;
  const int b0 = blockIdx.x * G, e0 = blockIdx.y * TE;
  if (e0 >= E) return;  // groups with fewer edges (E = 1 when scale == N) need fewer bands
  const int g = min(G, B - b0), te = min(TE, E - e0);
  float* s_ori = lds;                           // g x N x 64
  float* s_H = lds + (size_t)G * N * GN_FEAT;   // g x te x N
  {
    const TS* src = reinterpret_cast<const TS*>(Gr.ori) + (size_t)b0 * N * GN_FEAT;
    f32x4* dst = reinterpret_cast<f32x4*>(s_ori);
    for (int idx = threadIdx.x; idx < g * N * (GN_FEAT / 4); idx += kBlock) dst[idx] = ld4(src + 4 * idx);
    for (int idx = threadIdx.x; idx < g * te * N; idx += kBlock) {
      const int s = idx / (te * N), r = idx - s * te * N;
      s_H[idx] = Gr.H[((size_t)(b0 + s) * E + e0) * N + r];
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < g * te * (GN_FEAT / 4); idx += kBlock) {
    const int d = idx & 15, se = idx >> 4;
    const int s = se / te, e = se - s * te;
    const float* hrow = s_H + (size_t)se * N;
    const f32x4* o4 = reinterpret_cast<const f32x4*>(s_ori + (size_t)s * N * GN_FEAT) + d;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int n = 0; n < N; ++n) {
      const float hv = hrow[n];
      const f32x4 v = o4[n * (GN_FEAT / 4)];
      acc[0] = fmaf(hv, v[0], acc[0]);
      acc[1] = fmaf(hv, v[1], acc[1]);
      acc[2] = fmaf(hv, v[2], acc[2]);
      acc[3] = fmaf(hv, v[3], acc[3]);
    }
    st4(reinterpret_cast<TS*>(Gr.eo) + ((size_t)(b0 + s) * E + e0 + e) * GN_FEAT + 4 * d, acc);
  }
}

// Pairwise graph: eo[(i,j)] = ori_i + ori_j (2 ori_i on the diagonal), H never materialised.
template <bool SYM, typename TS>
__global__ __launch_bounds__(kBlock) void agg_gather_pairwise_kernel(const TS* __restrict__ ori,
                                                                     TS* __restrict__ eo, int N,
                                                                     long long total4) {
  const int E = SYM ? gn_pair_count(N) : N * N;
  for (long long idx = (long long)blockIdx.x * kBlock + threadIdx.x; idx < total4;
       idx += (long long)gridDim.x * kBlock) {
    const int d = (int)(idx & 15);
    const long long be = idx >> 4;
    const int b = (int)(be / E), e = (int)(be - (long long)b * E);
    int i, j;
    if (SYM) {
      gn_pair_decode(e, N, i, j);
    } else {
      i = e / N;
      j = e - i * N;
    }
    const TS* o4 = ori + (size_t)b * N * GN_FEAT + 4 * d;
    const f32x4 a = ld4(o4 + (size_t)i * GN_FEAT), c = ld4(o4 + (size_t)j * GN_FEAT);
    f32x4 r = {a[0] + c[0], a[1] + c[1], a[2] + c[2], a[3] + c[3]};
    st4(eo + 4 * idx, r);
  }
}

struct ScatterTable {
  gn_scatter_group_t g[GN_MAX_GROUPS];
};
// blockIdx.y = group; LDS sized for the largest E of the launch
template <typename TS>
__global__ __launch_bounds__(kBlock) void agg_scatter_kernel(ScatterTable T, int B, int N, int G, int Emax,
                                                             float fN) {
  extern __shared__ __align__(16) float lds[];
  const gn_scatter_group_t Gr = T.g[blockIdx.y];
  const int E = Gr.E;
  const int b0 = blockIdx.x * G;
  const int g = min(G, B - b0);
  float* s_feat = lds;                              // g x E x 64
  float* s_H = lds + (size_t)G * Emax * GN_FEAT;    // g x E x N
  {
    const TS* src = reinterpret_cast<const TS*>(Gr.feat) + (size_t)b0 * E * GN_FEAT;
    f32x4* dst = reinterpret_cast<f32x4*>(s_feat);
    for (int idx = threadIdx.x; idx < g * E * (GN_FEAT / 4); idx += kBlock) dst[idx] = ld4(src + 4 * idx);
    const float* hs = Gr.H + (size_t)b0 * E * N;
    for (int idx = threadIdx.x; idx < g * E * N; idx += kBlock) s_H[idx] = hs[idx];
  }
  __syncthreads();
  // every thread a (node, 4 features) item of H^T feat (a zero incidence skips its row: top-k incidences are sparse) ...
  for (int idx = threadIdx.x; idx < g * N * 16; idx += kBlock) {
    const int d = idx & 15, sn = idx >> 4;
    const int s = sn / N, n = sn - s * N;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* hcol = s_H + (size_t)s * E * N + n;
    const f32x4* f4 = reinterpret_cast<const f32x4*>(s_feat + (size_t)s * E * GN_FEAT) + d;
    for (int e = 0; e < E; ++e) {
      const float hv = hcol[(size_t)e * N];
      if (hv != 0.f) {
        const f32x4 v = f4[e * 16];
        acc[0] = fmaf(hv, v[0], acc[0]);
        acc[1] = fmaf(hv, v[1], acc[1]);
        acc[2] = fmaf(hv, v[2], acc[2]);
        acc[3] = fmaf(hv, v[3], acc[3]);
      }
    }
    const f32x4 r = {acc[0] / fN, acc[1] / fN, acc[2] / fN, acc[3] / fN};
    st4(reinterpret_cast<TS*>(Gr.out) + ((size_t)(b0 + s) * N + n) * 2 * GN_FEAT + 4 * d, r);
  }
  // ... then the ori half of the concat
  for (int idx = threadIdx.x; idx < g * N * 16; idx += kBlock) {
    const int d = idx & 15, sn = idx >> 4;
    const f32x4 acc = ld4(reinterpret_cast<const TS*>(Gr.ori) + ((size_t)b0 * N + sn) * GN_FEAT + 4 * d);
    const f32x4 r = {acc[0] / fN, acc[1] / fN, acc[2] / fN, acc[3] / fN};
    st4(reinterpret_cast<TS*>(Gr.out) + ((size_t)b0 * N + sn) * 2 * GN_FEAT + GN_FEAT + 4 * d, r);
  }
}

// Large-E / pairwise scatter straight from global memory (feat rows are 256-byte lines; every
// row is read by exactly two nodes in the pairwise case, so the second read is an L2 hit).
// MODE 0: general H (large E); 1: pairwise, E = N*N ordered edges; 2: pairwise, E = N(N+1)/2 pair sums
template <int MODE, typename TS>
__global__ __launch_bounds__(kBlock) void agg_scatter_direct_kernel(const TS* __restrict__ feat,
                                                                    const float* __restrict__ H,
                                                                    const TS* __restrict__ ori,
                                                                    TS* __restrict__ out, int N, int E,
                                                                    long long total4, float fN) {
  for (long long idx = (long long)blockIdx.x * kBlock + threadIdx.x; idx < total4;
       idx += (long long)gridDim.x * kBlock) {
    const int d = (int)(idx & 31);
    const long long bn = idx >> 5;
    const int b = (int)(bn / N), n = (int)(bn - (long long)b * N);
    f32x4 acc;
    if (d < 16) {
      acc = {0.f, 0.f, 0.f, 0.f};
      const TS* f4 = feat + (size_t)b * E * GN_FEAT + 4 * d;
      if (MODE == 2) {
        // every pair {n,j} once: the pair row already holds both ordered edges (and the self-loop's 2)
#pragma unroll 8
        for (int j = 0; j < N; ++j) {
          const f32x4 v = ld4(f4 + (size_t)gn_pair_index(n, j, N) * GN_FEAT);
          acc[0] += v[0];
          acc[1] += v[1];
          acc[2] += v[2];
          acc[3] += v[3];
        }
      } else if (MODE == 1) {
        // (n,n) counts twice (H = 2 on self-loops); fp32 tolerance makes the edge order immaterial
#pragma unroll 4
        for (int j = 0; j < N; ++j) {
          const f32x4 v = ld4(f4 + (size_t)(n * N + j) * GN_FEAT);
          const f32x4 w = ld4(f4 + (size_t)(j * N + n) * GN_FEAT);
          acc[0] += v[0] + w[0];
          acc[1] += v[1] + w[1];
          acc[2] += v[2] + w[2];
          acc[3] += v[3] + w[3];
        }
      } else {
        const float* hcol = H + (size_t)b * E * N + n;
        for (int e = 0; e < E; ++e) {
          const float hv = hcol[(size_t)e * N];
          if (hv != 0.f) {
            const f32x4 v = ld4(f4 + (size_t)e * GN_FEAT);
            acc[0] = fmaf(hv, v[0], acc[0]);
            acc[1] = fmaf(hv, v[1], acc[1]);
            acc[2] = fmaf(hv, v[2], acc[2]);
            acc[3] = fmaf(hv, v[3], acc[3]);
          }
        }
      }
    } else {
      acc = ld4(ori + ((size_t)b * N + n) * GN_FEAT + 4 * (d - 16));
    }
    f32x4 r = {acc[0] / fN, acc[1] / fN, acc[2] / fN, acc[3] / fN};
    st4(out + 4 * idx, r);
  }
}

// Pairwise scatter over unordered pairs, every pair row read ONCE (agg_scatter_direct_kernel<2> fetches each row for
// both of its member nodes: measured 326 MB of HBM traffic for 187 MB of algorithmic bytes at N = 50, B = 1024, bf16 —
// the launch sits at the HBM roof, so the second fetch is paid in full).  A workgroup owns a scene: the scene's
// N(N+1)/2 pair rows are contiguous in HBM and stream through LDS in bands of kPairBand rows (coalesced 16-byte pieces,
// the next band's loads in flight while the current one is consumed: two buffers); thread t owns the items
// (node n, 4 features d) with n*16 + d = t, t + 256, ... and adds, band by band, the rows of the band that contain n —
// rows (i, n), i < n, have increasing pair indices and all precede the contiguous run (n, n .. N-1), so a cursor per
// item walks them in the order j = 0 .. N-1 of the direct kernel: identical sums, bit for bit.
constexpr int kPairBand = 128;        // pair rows per band
constexpr int kPairItems = 4;         // (node, 4 features) items per thread: N * 16 <= 256 * kPairItems  ->  N <= 64
template <typename TS>
__global__ __launch_bounds__(kBlock) void agg_scatter_pairs_kernel(const TS* __restrict__ feat, const TS* __restrict__ ori,
                                                                   TS* __restrict__ out, int N, float fN) {
  constexpr int kPiece = 16 / (int)sizeof(TS);                 // elements per 16-byte piece
  constexpr int kRowPieces = GN_FEAT / kPiece;                 // pieces per pair row
  constexpr int kLoads = kPairBand * kRowPieces / kBlock;      // pieces per thread and band (fp32: 8, bf16: 4)
  __shared__ __align__(16) TS band[2][kPairBand * GN_FEAT];
  const int b = blockIdx.x;
  const int P = gn_pair_count(N);
  const f32x4* src = reinterpret_cast<const f32x4*>(feat + (size_t)b * P * GN_FEAT);
  const int total_pieces = P * kRowPieces;
  f32x4 acc[kPairItems];
  int nn[kPairItems], dd[kPairItems], cur[kPairItems];        // node, feature quad, next partner i < n
#pragma unroll
  for (int u = 0; u < kPairItems; ++u) {
    const int idx = (int)threadIdx.x + u * kBlock;
    nn[u] = idx < N * 16 ? idx >> 4 : -1;
    dd[u] = idx & 15;
    cur[u] = 0;
    acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 st[kLoads];
  auto fetch = [&](int band_i) {
#pragma unroll
    for (int it = 0; it < kLoads; ++it) {
      const int pc = band_i * (kPairBand * kRowPieces) + (int)threadIdx.x + it * kBlock;
      st[it] = src[min(pc, total_pieces - 1)];
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int it = 0; it < kLoads; ++it)
      reinterpret_cast<f32x4*>(band[buf])[(int)threadIdx.x + it * kBlock] = st[it];
  };
  const int n_bands = (P + kPairBand - 1) / kPairBand;
  fetch(0);
  commit(0);
  if (n_bands > 1) fetch(1);
  __syncthreads();
  for (int bi = 0; bi < n_bands; ++bi) {
    const int r0 = bi * kPairBand, r1 = min(P, r0 + kPairBand);
    const TS* rows = band[bi & 1];
#pragma unroll
    for (int u = 0; u < kPairItems; ++u) {
      const int n = nn[u];
      if (n < 0) continue;
      const TS* col = rows + 4 * dd[u];
      // rows (i, n), i < n, that fall into this band
      int i = cur[u];
      while (i < n) {
        const int p = gn_pair_start(i, N) + (n - i);
        if (p >= r1) break;
        const f32x4 v = ld4(col + (size_t)(p - r0) * GN_FEAT);
        acc[u][0] += v[0], acc[u][1] += v[1], acc[u][2] += v[2], acc[u][3] += v[3];
        ++i;
      }
      cur[u] = i;
      // the run (n, j), j = n .. N-1: pair indices ps .. ps + N - 1 - n
      if (i == n) {
        const int ps = gn_pair_start(n, N);
        const int lo = max(ps, r0), hi = min(ps + N - n, r1);
        for (int p = lo; p < hi; ++p) {
          const f32x4 v = ld4(col + (size_t)(p - r0) * GN_FEAT);
          acc[u][0] += v[0], acc[u][1] += v[1], acc[u][2] += v[2], acc[u][3] += v[3];
        }
      }
    }
    if (bi + 1 < n_bands) {
      commit((bi + 1) & 1);                       // (that buffer was last read during band bi - 1: behind a barrier)
      if (bi + 2 < n_bands) fetch(bi + 2);
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < kPairItems; ++u) {
    const int n = nn[u];
    if (n < 0) continue;
    const size_t row = (size_t)b * N + n;
    const f32x4 r = {acc[u][0] / fN, acc[u][1] / fN, acc[u][2] / fN, acc[u][3] / fN};
    st4(out + row * 2 * GN_FEAT + 4 * dd[u], r);
    const f32x4 o = ld4(ori + row * GN_FEAT + 4 * dd[u]);
    const f32x4 ro = {o[0] / fN, o[1] / fN, o[2] / fN, o[3] / fN};
    st4(out + row * 2 * GN_FEAT + GN_FEAT + 4 * dd[u], ro);
  }
}

// --------------------------------------------------------------------------------------------
// Exhaustive hyperedge search (init_adj_attention_listall, model/MS_HGNN_batch.py:390-414): hyperedge i =
// the group of s agents containing i with the largest total affinity sum_{a,b in group} corr[a][b].
// Candidates of agent i: i plus an (s-1)-subset of the others, in lexicographic order (the order of the
// reference's torch.combinations table); the first maximum wins, NaN ranks above every number.
// A workgroup owns a scene: corr and a binomial table live in LDS; for each agent the 256 threads stride
// over the candidate ranks, un-rank each (combinatorial number system), score it from LDS and keep the
// best; a workgroup arg-max picks the winner, which is un-ranked once more to write the 0/1 row.
// --------------------------------------------------------------------------------------------
constexpr unsigned long long kBinomCap = 1ull << 62;

// members of candidate `rank` of agent i as a bit mask (N <= 64) and its score
__device__ __forceinline__ unsigned long long listall_unrank(long long rank, int i, int n, int k, int ldb,
                                                             const unsigned long long* __restrict__ binom,
                                                             const float* __restrict__ s_corr, int N, float& score) {
  unsigned long long mask = 1ull << i;
  float sc = s_corr[i * N + i];
  int x = 0;
  for (int p = 0; p < k; ++p) {
    for (; x < n - 1; ++x) {
      const unsigned long long cnt = binom[(n - x - 1) * ldb + (k - p - 1)];   // candidates continuing with x here
      if ((unsigned long long)rank < cnt) break;
      rank -= (long long)cnt;
    }
    const int v = x < i ? x : x + 1;   // x-th of the other agents
    ++x;
    float add = s_corr[v * N + v];
    unsigned long long m = mask;
    while (m) {
      const int u = __ffsll((long long)m) - 1;
      m &= m - 1;
      add += s_corr[v * N + u] + s_corr[u * N + v];
    }
    sc += add;
    mask |= 1ull << v;
  }
  score = sc;
  return mask;
}

__device__ __forceinline__ bool listall_better(float a, long long ra, float b, long long rb) {
  const bool an = a != a, bn = b != b;
  if (an != bn) return an;
  if (!an && a != b) return a > b;
  return ra < rb;
}

__global__ __launch_bounds__(kBlock) void listall_incidence_kernel(const float* __restrict__ corr, float* __restrict__ H,
                                                                   int N, int s, long long C) {
  extern __shared__ __align__(16) float lds[];
  const int n = N - 1, k = s - 1, ldb = k + 1;
  unsigned long long* binom = reinterpret_cast<unsigned long long*>(lds);          // (n+1) x (k+1)
  long long* s_rank = reinterpret_cast<long long*>(binom + (size_t)(n + 1) * ldb);  // kBlock/64 ranks
  float* s_best = reinterpret_cast<float*>(s_rank + kBlock / 64);                  // 8 scores
  float* s_corr = s_best + 8;                                                      // N x N
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int idx = tid; idx < N * N; idx += kBlock) s_corr[idx] = corr[(size_t)b * N * N + idx];
  // Pascal's triangle, one column per thread, row by row (saturating: only ranks < C <= 2^31 are compared)
  for (int r = 0; r <= n; ++r) {
    if (tid <= k) {
      unsigned long long v;
      if (tid == 0) v = 1;
      else if (r == 0) v = 0;
      else {
        v = binom[(r - 1) * ldb + tid - 1] + binom[(r - 1) * ldb + tid];
        if (v > kBinomCap) v = kBinomCap;
      }
      binom[r * ldb + tid] = v;
    }
    __syncthreads();
  }
  for (int i = 0; i < N; ++i) {
    float best = 0.f;
    long long best_rank = -1;
    for (long long rank = tid; rank < C; rank += kBlock) {
      float sc;
      listall_unrank(rank, i, n, k, ldb, binom, s_corr, N, sc);
      if (best_rank < 0 || listall_better(sc, rank, best, best_rank)) {
        best = sc;
        best_rank = rank;
      }
    }
    // wave arg-max, then across the waves
    for (int off = 32; off > 0; off >>= 1) {
      const float ob = __shfl_xor(best, off, GN_WAVE);
      const long long orank = __shfl_xor(best_rank, off, GN_WAVE);
      if (orank >= 0 && (best_rank < 0 || listall_better(ob, orank, best, best_rank))) {
        best = ob;
        best_rank = orank;
      }
    }
    if (lane == 0) {
      s_best[wave] = best;
      s_rank[wave] = best_rank;
    }
    __syncthreads();
    best = s_best[0];
    best_rank = s_rank[0];
    for (int w = 1; w < kBlock / 64; ++w)
      if (s_rank[w] >= 0 && (best_rank < 0 || listall_better(s_best[w], s_rank[w], best, best_rank))) {
        best = s_best[w];
        best_rank = s_rank[w];
      }
    float sc;
    const unsigned long long mask = listall_unrank(best_rank, i, n, k, ldb, binom, s_corr, N, sc);
    if (tid < N) H[((size_t)b * N + i) * N + tid] = (mask >> tid) & 1ull ? 1.f : 0.f;
    __syncthreads();
  }
}

__global__ __launch_bounds__(kBlock) void fill_ones_kernel(float* __restrict__ p, long long n) {
  for (long long idx = (long long)blockIdx.x * kBlock + threadIdx.x; idx < n; idx += (long long)gridDim.x * kBlock)
    p[idx] = 1.f;
}

// --------------------------------------------------------------------------------------------
// Philox4x32-10 uniforms
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void philox_uniform_kernel(float* __restrict__ U, unsigned long long n,
                                                                unsigned long long seed, unsigned long long offset,
                                                                const unsigned long long* __restrict__ offset_dev,
                                                                unsigned long long nblk) {
  if (offset_dev) offset += *offset_dev;   // stream position kept in device memory (graph replays advance it)
  const unsigned long long blk0 = offset >> 2;
  for (unsigned long long t = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; t < nblk;
       t += (unsigned long long)gridDim.x * kBlock) {
    const unsigned long long blk = blk0 + t;
    uint32_t c[4];
    gn_philox_block(blk, seed, c);
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const unsigned long long gidx = blk * 4 + l;
      if (gidx >= offset && gidx - offset < n) U[gidx - offset] = gn_philox_to_uniform(c[l]);
    }
  }
}

inline int capped_grid(long long work_items, int per_block, int cap = 256 * 16) {
  long long g = (work_items + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int gn_abi_version(void) { return 34; }

extern "C" const char* gn_strerror(int code) {
  switch (code) {
    case GN_OK: return "ok";
    case GN_ERR_NULL: return "null pointer";
    case GN_ERR_SHAPE: return "unsupported or non-positive size";
    case GN_ERR_K_RANGE: return "selected index k out of range";
    case GN_ERR_ALIGN: return "pointer not 16-byte aligned";
    case GN_ERR_LAUNCH: return "kernel launch failed";
    case GN_ERR_LDS: return "tile does not fit in LDS";
    default: return "unknown error";
  }
}

extern "C" int gn_affinity_f32(const float* f, float* corr, int B, int N, int D, gn_stream_t stream) {
  GN_REQUIRE_PTR(f);
  GN_REQUIRE_PTR(corr);
  GN_REQUIRE_ALIGNED(f);
  if (B <= 0 || N <= 0 || D <= 0 || (D & 3) || D > 1024) return GN_ERR_SHAPE;
  const size_t fused = (size_t)N * (D + 4) * sizeof(float) + 8 + (size_t)N * N * 8;
  if (fused <= kLdsBudget) {
    ScaleList sl{};
    sl.n = 0;
    gn_allow_big_lds(affinity_topk_kernel<float>);
    hipLaunchKernelGGL(affinity_topk_kernel<float>, dim3(B), dim3(kBlock), fused, (hipStream_t)stream, f, corr, sl, N,
                       D, gn_block_extras_t{});
  } else {
    const size_t lds = (size_t)(16 + 64) * (D + 4) * sizeof(float);
    gn_allow_big_lds(affinity_banded_kernel);
    hipLaunchKernelGGL(affinity_banded_kernel, dim3((N + 15) / 16, B), dim3(kBlock), lds, (hipStream_t)stream, f, corr,
                       N, D);
  }
  return gn_check_launch();
}

extern "C" int gn_topk_incidence_f32(const float* corr, float* const* H_list, const int* k_list, int n_scales, int B,
                                     int N, gn_stream_t stream) {
  GN_REQUIRE_PTR(corr);
  if (B <= 0 || N <= 0 || n_scales < 1) return GN_ERR_SHAPE;
  ScaleList sl;
  const int rc = fill_scales(sl, H_list, k_list, n_scales, N);
  if (rc != GN_OK) return rc;
  if ((size_t)N * sizeof(float) > kLdsBudget) return GN_ERR_LDS;
  int RB = (int)(kLdsBudget / 2 / ((size_t)N * sizeof(float)));
  RB = RB > N ? N : RB;
  // enough bands to give the chip >= ~1024 workgroups when B is small
  while (RB > 8 && (long long)B * ((N + RB - 1) / RB) < 1024) RB = (RB + 1) / 2;
  gn_allow_big_lds(topk_incidence_kernel);
  hipLaunchKernelGGL(topk_incidence_kernel, dim3((N + RB - 1) / RB, B), dim3(kBlock), (size_t)RB * N * sizeof(float),
                     (hipStream_t)stream, corr, sl, N, RB);
  return gn_check_launch();
}

extern "C" int gn_listall_incidence_f32(const float* corr, float* H, int B, int N, int scale, gn_stream_t stream) {
  GN_REQUIRE_PTR(corr);
  GN_REQUIRE_PTR(H);
  if (B <= 0 || N <= 0) return GN_ERR_SHAPE;
  if (scale > N) return GN_ERR_K_RANGE;
  hipStream_t st = (hipStream_t)stream;
  if (scale == N) {   // one hyperedge holding every agent
    const long long total = (long long)B * N;
    hipLaunchKernelGGL(fill_ones_kernel, dim3(capped_grid(total, kBlock)), dim3(kBlock), 0, st, H, total);
    return gn_check_launch();
  }
  const int s = scale < 1 ? 1 : scale;
  if (N > 64) return GN_ERR_SHAPE;    // group membership is a 64-bit mask
  // C(N-1, s-1) candidates per agent, bounded so that ranks and the search stay sane
  long double c = 1;
  for (int j = 1; j <= s - 1; ++j) c = c * (long double)(N - 1 - (s - 1) + j) / (long double)j;
  if (c > 2147483647.0L) return GN_ERR_SHAPE;
  const long long C = (long long)(c + 0.5L);
  const size_t lds = (size_t)N * s * sizeof(unsigned long long) + (size_t)N * N * sizeof(float) + 8 * sizeof(float) +
                     (kBlock / 64) * sizeof(long long);
  if (lds > kLdsBudget) return GN_ERR_LDS;
  if (lds > 64 * 1024) gn_allow_big_lds(listall_incidence_kernel);
  hipLaunchKernelGGL(listall_incidence_kernel, dim3(B), dim3(kBlock), lds, st, corr, H, N, s, C);
  return gn_check_launch();
}

template <typename TS>
static int affinity_topk_launch(const TS* f, float* corr, float* const* H_list, const int* k_list, int n_scales, int B,
                                int N, int D, const gn_block_extras_t* extras, hipStream_t stream) {
  const bool embed = extras != nullptr && extras->x_raw != nullptr;
  if (embed && sizeof(TS) != sizeof(float)) return GN_ERR_SHAPE;   // the embedding front-end is fp32 only
  if (!embed) {
    GN_REQUIRE_PTR(f);
    GN_REQUIRE_ALIGNED(f);
  }
  if (B <= 0 || N <= 0 || D <= 0 || (D & 3) || D > 1024) return GN_ERR_SHAPE;
  ScaleList sl;
  const int rc = fill_scales(sl, H_list, k_list, n_scales, N);
  if (rc != GN_OK) return rc;
  size_t fused = (size_t)N * (D + 4) * sizeof(float) + 8 + (size_t)N * N * 8;
  if (embed) {
    if (extras->x_dim <= 0 || !extras->M || !extras->c || !extras->f_contig) return GN_ERR_NULL;
    if (!gn_aligned16(extras->c) || !gn_aligned16(extras->f_contig)) return GN_ERR_ALIGN;
    fused += (size_t)N * extras->x_dim * sizeof(float);
  }
  if (fused > kLdsBudget) return GN_ERR_LDS;
  gn_block_extras_t ex{};
  if (extras != nullptr) {
    ex = *extras;
    if (ex.f_out != nullptr && (!gn_aligned16(ex.f_out) || ex.f_out_ld < D || (ex.f_out_ld & 3))) return GN_ERR_ALIGN;
    sl.H_cat = ex.H_cat;
  }
  gn_allow_big_lds(affinity_topk_kernel<TS>);
  hipLaunchKernelGGL(affinity_topk_kernel<TS>, dim3(B), dim3(kBlock), fused, stream, f, corr, sl, N, D, ex);
  return gn_check_launch();
}
extern "C" int gn_affinity_topk_f32(const float* f, float* corr, float* const* H_list, const int* k_list, int n_scales,
                                    int B, int N, int D, const gn_block_extras_t* extras, gn_stream_t stream) {
  return affinity_topk_launch<float>(f, corr, H_list, k_list, n_scales, B, N, D, extras, (hipStream_t)stream);
}
extern "C" int gn_affinity_topk_bf16(const void* f, float* corr, float* const* H_list, const int* k_list, int n_scales,
                                     int B, int N, int D, const gn_block_extras_t* extras, gn_stream_t stream) {
  return affinity_topk_launch<__bf16>(reinterpret_cast<const __bf16*>(f), corr, H_list, k_list, n_scales, B, N, D,
                                      extras, (hipStream_t)stream);
}

template <typename TS>
static int node2edge_launch(const gn_n2e_group_t* groups, int n_groups, int B, int N, hipStream_t s) {
  int rc = check_groups(groups, n_groups);
  if (rc != GN_OK) return rc;
  if (B <= 0 || N <= 0) return GN_ERR_SHAPE;
  WaveTable<gn_n2e_group_t> T{};
  PairTable P{};
  long long waves = 0;
  int pair_wgs = 0;
  size_t lds = 0;
  for (int g = 0; g < n_groups; ++g) {
    const gn_n2e_group_t& G = groups[g];
    if (!G.xp || !G.pq || !G.w2 || !G.b2 || !G.edges) return GN_ERR_NULL;
    if (G.E <= 0) return GN_ERR_SHAPE;
    if (!gn_aligned16(G.xp) || !gn_aligned16(G.pq) || !gn_aligned16(G.edges)) return GN_ERR_ALIGN;
    if (G.H == nullptr) {
      if ((long long)G.E != (G.sym ? (long long)gn_pair_count(N) : (long long)N * N)) return GN_ERR_SHAPE;
      // pairwise groups: scenes per workgroup so that the staged rows stay <= 32 KiB; edge bands when one
      // scene alone has many more edges than a workgroup should walk
      const size_t per_scene = (size_t)N * (GN_FEAT + GN_FEAT + 1) * sizeof(float);
      const size_t fixed = (3 * kBlock + 32) * sizeof(float);
      if (per_scene + fixed > 158 * 1024 || N > 32767) return GN_ERR_LDS;   // one workgroup may take the CU's 160 KiB
      int SG = 1;
      while (SG < 8 && (size_t)(2 * SG) * per_scene <= 32 * 1024 && (B + 2 * SG - 1) / (2 * SG) >= 512) SG *= 2;
      const long long edges_per_wg = (long long)SG * G.E;
      int bands = 1;
      while (bands < 64 && edges_per_wg / (bands * 2) >= 2048 && (long long)((B + SG - 1) / SG) * bands < 2048)
        bands *= 2;
      P.g[P.n] = G;
      P.SG[P.n] = SG;
      P.bands[P.n] = bands;
      P.first_wg[P.n] = pair_wgs;
      pair_wgs += ((B + SG - 1) / SG) * bands;
      ++P.n;
      lds = lds > (size_t)SG * per_scene + fixed ? lds : (size_t)SG * per_scene + fixed;
      continue;
    }
    if (G.sym) return GN_ERR_SHAPE;  // the symmetric form exists for the pairwise graph only
    T.g[T.n] = G;
    ++T.n;
  }
  P.first_wg[P.n] = pair_wgs;
  // hyper groups: scenes per workgroup so that the staged rows stay <= 32 KiB while the grid keeps >= ~1024 workgroups
  int SGh = 1, EBh = 1;
  if (T.n > 0) {
    const size_t per_scene = (size_t)N * (GN_FEAT + GN_FEAT + 1) * sizeof(float);
    while (SGh < 8 && (size_t)(2 * SGh) * per_scene <= 32 * 1024 && (long long)((B + 2 * SGh - 1) / (2 * SGh)) * T.n >= 1024)
      SGh *= 2;
    // hyperedges per band: ~12 KiB of member lists (3 words per (edge, node) slot), at least 4 edges, at most the
    // most edges any group's workgroup walks
    int maxE = 1;
    for (int g = 0; g < T.n; ++g) maxE = maxE > T.g[g].E ? maxE : T.g[g].E;
    EBh = (int)((size_t)3072 * GN_N2E_U / ((size_t)12 * N + 144));
    EBh = EBh < 4 ? 4 : EBh;
    if (N <= 64 && EBh > 4 * GN_N2E_U) EBh = 4 * GN_N2E_U;        // (the band whose H rows a workgroup can hold in registers, see the kernel)
    EBh = EBh > SGh * maxE ? SGh * maxE : EBh;
    size_t scratch = n2e_hyper_scratch_floats(EBh, N) * sizeof(float);
    size_t stage = (size_t)SGh * per_scene;
    // N <= 64: one lane pair per hyperedge (no scratch, no barriers after the stage); scenes per workgroup so that
    // ~100 of the 128 pairs have a row, the stage stays <= 56 KiB and the grid keeps >= 1024 workgroups.
    // GN_N2E_ROWS = 0 keeps the banded form, 1 forces this one (parity tests).
    const char* rows_env = getenv("GN_N2E_ROWS");                 // 0: never, 1: whenever N <= 64, unset: by launch size
    const bool no_rows = rows_env != nullptr && atoi(rows_env) == 0;
    const bool force_rows = rows_env != nullptr && atoi(rows_env) != 0;
    long long hyper_rows = 0;
    for (int g = 0; g < T.n; ++g) hyper_rows += (long long)B * T.g[g].E;
    // (few short scenes cannot fill the 128 lane pairs of a workgroup AND the chip.  Measured, banded vs rows, us —
    // N = 11: B = 512 10.6 / 21.0, 1024 15.3 / 19.1, 2048 25.0 / 24.3, 4096 42.1 / 39.1;
    // N = 50: B = 32 23.7 / 14.8, 128 25.9 / 18.4, 256 32.1 / 31.9, 1024 104 / 79)
    if (N <= 64 && !no_rows && (maxE >= 24 || hyper_rows >= 49152 || force_rows)) {
      const size_t row_scene = (size_t)2 * N * kRowPitch * sizeof(float);
      SGh = 1;
      while ((SGh + 1) * maxE <= kBlock / 2 && (size_t)(SGh + 1) * row_scene <= 56 * 1024 &&
             (long long)((B + SGh) / (SGh + 1)) * T.n >= 1024)
        ++SGh;
      if (const char* e = getenv("GN_N2E_ROWS_SG")) SGh = atoi(e) > 0 ? atoi(e) : SGh;     // (tuning)
      EBh = 0;
      scratch = 0;
      stage = (size_t)SGh * row_scene;
    }
    if (stage + scratch > 158 * 1024) return GN_ERR_LDS;
    const size_t l = stage + scratch;
    lds = lds > l ? lds : l;
    for (int g = 0; g < T.n; ++g) {
      T.first[g] = waves;
      waves += (B + SGh - 1) / SGh;                    // (`waves` counts workgroups here)
    }
  }
  T.first[T.n] = waves;
  const long long grid = pair_wgs + waves;
  if (grid > 0x3fffffffLL) return GN_ERR_SHAPE;
  XcdSections xs{};
  for (int g = 0; g < P.n; ++g) xs.first[xs.n++] = P.first_wg[g];
  for (int g = 0; g < T.n; ++g) xs.first[xs.n++] = pair_wgs + (int)T.first[g];
  xs.first[xs.n] = (int)grid;
  // (pairwise workgroups are (scene chunk, band) with the band fastest: scene order, like every other stage)
  if (T.n > 0 && EBh == 0) {
    gn_allow_big_lds(node2edge_kernel<TS, true>);
    hipLaunchKernelGGL((node2edge_kernel<TS, true>), dim3((unsigned)gn_xcd_grid(xs)), dim3(kBlock), lds, s, T, P, B, N, SGh,
                       EBh, xs);
  } else {
    gn_allow_big_lds(node2edge_kernel<TS, false>);
    hipLaunchKernelGGL((node2edge_kernel<TS, false>), dim3((unsigned)gn_xcd_grid(xs)), dim3(kBlock), lds, s, T, P, B, N, SGh,
                       EBh, xs);
  }
  return gn_check_launch();
}
extern "C" int gn_node2edge_f32(const gn_n2e_group_t* groups, int n_groups, int B, int N, gn_stream_t stream) {
  return node2edge_launch<float>(groups, n_groups, B, N, (hipStream_t)stream);
}
extern "C" int gn_node2edge_bf16(const gn_n2e_group_t* groups, int n_groups, int B, int N, gn_stream_t stream) {
  return node2edge_launch<__bf16>(groups, n_groups, B, N, (hipStream_t)stream);
}

// workgroups a gather / scatter launch keeps at least when it packs several scenes into one (GN_GS_MIN_WGS).  2048 = two
// rounds of the chip at B = 4096, N = 11: the second round's loads overlap the first one's stores (3.90 -> 4.12 TB/s for
// the pair; 4096 and 8192 measured the same as 2048)
static int gs_min_wgs() {
  static const int v = getenv("GN_GS_MIN_WGS") != nullptr ? atoi(getenv("GN_GS_MIN_WGS")) : 2048;
  return v;
}

template <typename TS>
static int gather_launch(const gn_gather_group_t* groups, int n_groups, int B, int N, hipStream_t s) {
  int rc = check_groups(groups, n_groups);
  if (rc != GN_OK) return rc;
  if (B <= 0 || N <= 0) return GN_ERR_SHAPE;
  GatherTable T{};
  int nh = 0, Emax = 0;
  for (int g = 0; g < n_groups; ++g) {
    const gn_gather_group_t& G = groups[g];
    if (!G.ori || !G.eo) return GN_ERR_NULL;
    if (!gn_aligned16(G.ori) || !gn_aligned16(G.eo)) return GN_ERR_ALIGN;
    if (G.E <= 0) return GN_ERR_SHAPE;
    if (G.H == nullptr) {
      if ((long long)G.E != (G.sym ? (long long)gn_pair_count(N) : (long long)N * N)) return GN_ERR_SHAPE;
    } else {
      if (G.sym) return GN_ERR_SHAPE;
      T.g[nh++] = G;
      Emax = G.E > Emax ? G.E : Emax;
    }
  }
  for (int g = 0; g < n_groups; ++g) {
    const gn_gather_group_t& G = groups[g];
    if (G.H != nullptr) continue;
    const long long total4 = (long long)B * G.E * 16;
    const dim3 grid(capped_grid(total4, kBlock * 4));
    const TS* ori = reinterpret_cast<const TS*>(G.ori);
    TS* eo = reinterpret_cast<TS*>(G.eo);
    if (G.sym)
      hipLaunchKernelGGL((agg_gather_pairwise_kernel<true, TS>), grid, dim3(kBlock), 0, s, ori, eo, N, total4);
    else
      hipLaunchKernelGGL((agg_gather_pairwise_kernel<false, TS>), grid, dim3(kBlock), 0, s, ori, eo, N, total4);
  }
  if (nh > 0) {
    const size_t ori_b = (size_t)N * GN_FEAT * sizeof(float);
    if (ori_b + (size_t)N * sizeof(float) > kLdsBudget) return GN_ERR_LDS;
    int G = 1, TE = Emax;
    const size_t per_scene = ori_b + (size_t)Emax * N * sizeof(float);
    if (per_scene <= kLdsBudget) {
      // several scenes per workgroup while the tile stays <= 24 KiB and the grid stays >= 1024
      while (G < 16 && (size_t)(2 * G) * per_scene <= 24 * 1024 && (long long)((B + 2 * G - 1) / (2 * G)) * nh >= gs_min_wgs())
        G *= 2;
    } else {
      TE = (int)((kLdsBudget - ori_b) / ((size_t)N * sizeof(float)));
      if (TE < 1) return GN_ERR_LDS;
    }
    const size_t lds = (size_t)G * ori_b + (size_t)G * TE * N * sizeof(float);
    gn_allow_big_lds(agg_gather_kernel<TS>);
    hipLaunchKernelGGL(agg_gather_kernel<TS>, dim3((B + G - 1) / G, (Emax + TE - 1) / TE, nh), dim3(kBlock), lds, s, T,
                       B, N, G, TE);
  }
  return gn_check_launch();
}
extern "C" int gn_agg_gather_f32(const gn_gather_group_t* groups, int n_groups, int B, int N, gn_stream_t stream) {
  return gather_launch<float>(groups, n_groups, B, N, (hipStream_t)stream);
}
extern "C" int gn_agg_gather_bf16(const gn_gather_group_t* groups, int n_groups, int B, int N, gn_stream_t stream) {
  return gather_launch<__bf16>(groups, n_groups, B, N, (hipStream_t)stream);
}

template <typename TS>
static int scatter_launch(const gn_scatter_group_t* groups, int n_groups, int B, int N, float divisor, hipStream_t s) {
  int rc = check_groups(groups, n_groups);
  if (rc != GN_OK) return rc;
  if (B <= 0 || N <= 0 || !(divisor != 0.f)) return GN_ERR_SHAPE;
  const long long total4 = (long long)B * N * 32;
  ScatterTable T{};
  int nh = 0, Emax = 0;
  for (int g = 0; g < n_groups; ++g) {
    const gn_scatter_group_t& G = groups[g];
    if (!G.feat || !G.ori || !G.out) return GN_ERR_NULL;
    if (!gn_aligned16(G.feat) || !gn_aligned16(G.ori) || !gn_aligned16(G.out)) return GN_ERR_ALIGN;
    if (G.E <= 0) return GN_ERR_SHAPE;
    if (G.H == nullptr && (long long)G.E != (G.sym ? (long long)gn_pair_count(N) : (long long)N * N))
      return GN_ERR_SHAPE;
    if (G.H != nullptr && G.sym) return GN_ERR_SHAPE;
  }
  for (int g = 0; g < n_groups; ++g) {
    const gn_scatter_group_t& G = groups[g];
    const TS* feat = reinterpret_cast<const TS*>(G.feat);
    const TS* ori = reinterpret_cast<const TS*>(G.ori);
    TS* out = reinterpret_cast<TS*>(G.out);
    if (G.H == nullptr && G.sym && N * 16 <= kBlock * kPairItems && B >= 256 &&
        !(getenv("GN_SCATTER_PAIRS") && atoi(getenv("GN_SCATTER_PAIRS")) == 0)) {
      // one workgroup per scene, every pair row read once (enough scenes to fill the chip; GN_SCATTER_PAIRS = 0 keeps
      // the direct kernel: parity tests run both)
      hipLaunchKernelGGL((agg_scatter_pairs_kernel<TS>), dim3(B), dim3(kBlock), 0, s, feat, ori, out, N, divisor);
    } else if (G.H == nullptr && G.sym) {
      hipLaunchKernelGGL((agg_scatter_direct_kernel<2, TS>), dim3(capped_grid(total4, kBlock)), dim3(kBlock), 0, s,
                         feat, G.H, ori, out, N, G.E, total4, divisor);
    } else if (G.H == nullptr) {
      hipLaunchKernelGGL((agg_scatter_direct_kernel<1, TS>), dim3(capped_grid(total4, kBlock)), dim3(kBlock), 0, s,
                         feat, G.H, ori, out, N, G.E, total4, divisor);
    } else if ((size_t)G.E * (GN_FEAT + N) * sizeof(float) > kLdsBudget / 2) {
      hipLaunchKernelGGL((agg_scatter_direct_kernel<0, TS>), dim3(capped_grid(total4, kBlock)), dim3(kBlock), 0, s,
                         feat, G.H, ori, out, N, G.E, total4, divisor);
    } else {
      T.g[nh++] = G;
      Emax = G.E > Emax ? G.E : Emax;
    }
  }
  if (nh > 0) {
    const size_t per_scene = (size_t)Emax * (GN_FEAT + N) * sizeof(float);
    int G = 1;
    while (G < 16 && (size_t)(2 * G) * per_scene <= 24 * 1024 && (long long)((B + 2 * G - 1) / (2 * G)) * nh >= gs_min_wgs())
      G *= 2;
    gn_allow_big_lds(agg_scatter_kernel<TS>);
    hipLaunchKernelGGL(agg_scatter_kernel<TS>, dim3((B + G - 1) / G, nh), dim3(kBlock), (size_t)G * per_scene, s, T, B,
                       N, G, Emax, divisor);
  }
  return gn_check_launch();
}
extern "C" int gn_agg_scatter_f32(const gn_scatter_group_t* groups, int n_groups, int B, int N, float divisor,
                                  gn_stream_t stream) {
  return scatter_launch<float>(groups, n_groups, B, N, divisor, (hipStream_t)stream);
}
extern "C" int gn_agg_scatter_bf16(const gn_scatter_group_t* groups, int n_groups, int B, int N, float divisor,
                                   gn_stream_t stream) {
  return scatter_launch<__bf16>(groups, n_groups, B, N, divisor, (hipStream_t)stream);
}

// Pitched copy (rows x width bytes, 16-byte pieces): the column block of the feature tensor that a rank ships into its
// all-gather staging bank (sharding.BucketedGather) — a strided torch copy_ of the same 5.8 MB took ~3x as long.
__global__ __launch_bounds__(kBlock) void copy_2d_kernel(const unsigned char* __restrict__ src, size_t spitch,
                                                         unsigned char* __restrict__ dst, size_t dpitch, int w16,
                                                         long long total) {
  for (long long idx = (long long)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (long long)gridDim.x * kBlock) {
    const long long r = idx / w16;
    const int c = (int)(idx - r * w16);
    *reinterpret_cast<f32x4*>(dst + r * dpitch + 16 * (size_t)c) = *reinterpret_cast<const f32x4*>(src + r * spitch + 16 * (size_t)c);
  }
}
extern "C" int gn_copy_2d(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width_bytes, int rows,
                          gn_stream_t stream) {
  GN_REQUIRE_PTR(dst);
  GN_REQUIRE_PTR(src);
  if (rows <= 0 || width_bytes == 0 || (width_bytes & 15) || (dst_pitch & 15) || (src_pitch & 15) || dst_pitch < width_bytes ||
      src_pitch < width_bytes)
    return GN_ERR_SHAPE;
  GN_REQUIRE_ALIGNED(dst);
  GN_REQUIRE_ALIGNED(src);
  const int w16 = (int)(width_bytes / 16);
  const long long total = (long long)rows * w16;
  hipLaunchKernelGGL(copy_2d_kernel, dim3(capped_grid(total, kBlock * 2)), dim3(kBlock), 0, (hipStream_t)stream,
                     reinterpret_cast<const unsigned char*>(src), src_pitch, reinterpret_cast<unsigned char*>(dst), dst_pitch, w16,
                     total);
  return gn_check_launch();
}

__global__ void counter_add_kernel(unsigned long long* ctr, unsigned long long add) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *ctr += add;
}

extern "C" int gn_philox_uniform_f32(float* U, size_t n, unsigned long long seed, unsigned long long offset,
                                     const unsigned long long* offset_dev, gn_stream_t stream) {
  GN_REQUIRE_PTR(U);
  if (n == 0) return GN_ERR_SHAPE;
  // with a device-side base the first block may be partial whatever `offset` is: one spare block
  const unsigned long long nblk =
      offset_dev ? (((unsigned long long)n + 3) >> 2) + 1 : ((offset + n + 3) >> 2) - (offset >> 2);
  hipLaunchKernelGGL(philox_uniform_kernel, dim3(capped_grid((long long)nblk, kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, U, (unsigned long long)n, seed, offset, offset_dev, nblk);
  return gn_check_launch();
}

extern "C" int gn_counter_add_u64(unsigned long long* counter, unsigned long long add, gn_stream_t stream) {
  GN_REQUIRE_PTR(counter);
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, add);
  return gn_check_launch();
}
