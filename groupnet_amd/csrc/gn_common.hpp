// Shared device/host helpers for the gfx950 kernels of the GroupNet MS-HGNN path.
// Written for CDNA4 only: 64-lane wavefronts, v_mfma_f32_32x32x2_f32, 160 KiB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/groupnet_hip.h"

#define GN_WAVE 64

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline bool gn_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static inline int gn_check_launch() { return hipGetLastError() == hipSuccess ? GN_OK : GN_ERR_LAUNCH; }

#define GN_REQUIRE_PTR(p) \
  do {                    \
    if ((p) == nullptr) return GN_ERR_NULL; \
  } while (0)
#define GN_REQUIRE_ALIGNED(p) \
  do {                        \
    if (!gn_aligned16(p)) return GN_ERR_ALIGN; \
  } while (0)

// Wave-uniform value → SGPR, so that addresses derived from it use the scalar path.
__device__ __forceinline__ int gn_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Reductions over the 32 lanes of one half-wave (lanes 0..31 and 32..63 separately).
__device__ __forceinline__ float gn_half_sum(float v) {
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, GN_WAVE);
  return v;
}
// Reductions over all 64 lanes.
__device__ __forceinline__ float gn_wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, GN_WAVE);
  return v;
}
__device__ __forceinline__ float gn_wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, GN_WAVE));
  return v;
}

// softmax(att*H) runs over ALL N nodes (model/MS_HGNN_batch.py:366-367): the `others` non-members each add
// exp(0 - mx).  With others == 0 the running maximum mx may be far below zero, exp(0 - mx) overflows and
// 0 * inf would poison the sum: the term is then absent, not multiplied by zero.
__device__ __forceinline__ float gn_nonmember_sum(int others, float mx) {
  return others > 0 ? (float)others * expf(0.f - mx) : 0.f;
}

// Kernels that may ask for more than 64 KiB of dynamic LDS must opt in once per process.
template <typename K>
static inline void gn_allow_big_lds(K kernel) {
  static bool done = false;
  if (!done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    done = true;
  }
}

// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3"): counter = (blk, 0),
// key = seed.  Element i of the uniform stream is word (i & 3) of block (i >> 2), mapped to [0,1) by
// (x >> 8) * 2^-24.
__device__ __forceinline__ void gn_philox_block(unsigned long long blk, unsigned long long seed, uint32_t (&c)[4]) {
  c[0] = (uint32_t)blk;
  c[1] = (uint32_t)(blk >> 32);
  c[2] = 0u;
  c[3] = 0u;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0;
    c[1] = n1;
    c[2] = n2;
    c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
__device__ __forceinline__ float gn_philox_to_uniform(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }
__device__ __forceinline__ float gn_philox_uniform_at(unsigned long long idx, unsigned long long seed) {
  uint32_t c[4];
  gn_philox_block(idx >> 2, seed, c);
  const int l = (int)(idx & 3);
  const uint32_t x = l == 0 ? c[0] : (l == 1 ? c[1] : (l == 2 ? c[2] : c[3]));
  return gn_philox_to_uniform(x);
}

// ---- XCD-aware workgroup order ---------------------------------------------------------------------------------
// The hardware deals the workgroups of a launch round-robin over the 8 XCDs (block p runs on the XCD of p % 8), and
// every XCD has its own 4 MiB L2.  Every stage of the forward is a set of SECTIONS (one per module, or per module and
// weight slice) whose workgroups cover the scenes in order.  XCD x is given the x-th eighth of EVERY section: the same
// scenes then meet the same L2 in every stage, so what a stage writes for a scene (x', pq, edges, edge_feat, A, feat)
// is still in that L2 when the next stage reads it, instead of being fetched from another XCD's L2 / the Infinity
// Cache.  Correctness never depends on the placement: this is only a permutation of the block index.
#define GN_MAX_SECTIONS 32
struct XcdSections {
  int first[GN_MAX_SECTIONS + 1];   // prefix of logical workgroup indices, section by section
  int n;
  int enabled;
};
// physical block -> logical workgroup, or -1 for a padding block (the grid is 8 x the largest per-XCD share)
__device__ __forceinline__ int gn_xcd_logical(const XcdSections& S, int p) {
  if (!S.enabled) return p;
  const int x = p & 7;
  int slot = p >> 3;
  for (int g = 0; g < S.n; ++g) {
    const int W = S.first[g + 1] - S.first[g];
    const int lo = (int)(((long long)W * x) >> 3), hi = (int)(((long long)W * (x + 1)) >> 3);
    if (slot < hi - lo) return S.first[g] + lo + slot;
    slot -= hi - lo;
  }
  return -1;
}
// host: finish a table whose first[0..n] is filled; returns the grid size
static inline int gn_xcd_grid(XcdSections& S) {
  const bool off = getenv("GN_XCD") != nullptr && atoi(getenv("GN_XCD")) == 0;      // (per call: tests toggle it)
  const int total = S.first[S.n];
  S.enabled = (!off && S.n <= GN_MAX_SECTIONS && total >= 64) ? 1 : 0;
  if (!S.enabled) return total;
  int worst = 0;
  for (int x = 0; x < 8; ++x) {
    int c = 0;
    for (int g = 0; g < S.n; ++g) {
      const int W = S.first[g + 1] - S.first[g];
      c += (int)(((long long)W * (x + 1)) >> 3) - (int)(((long long)W * x) >> 3);
    }
    worst = c > worst ? c : worst;
  }
  return 8 * worst;
}

// Unordered pairs (i <= j) of N nodes, row-major over i: p(i,j) = i*N - i(i-1)/2 + (j - i).
__host__ __device__ __forceinline__ int gn_pair_count(int N) { return N * (N + 1) / 2; }
__device__ __forceinline__ int gn_pair_start(int i, int N) { return i * N - (i * (i - 1)) / 2; }
__device__ __forceinline__ int gn_pair_index(int a, int b, int N) {  // any order
  const int i = a < b ? a : b, j = a < b ? b : a;
  return gn_pair_start(i, N) + (j - i);
}
__device__ __forceinline__ void gn_pair_decode(int p, int N, int& i, int& j) {
  const float t = (float)(2 * N + 1);
  int r = (int)((t - sqrtf(t * t - 8.f * (float)p)) * 0.5f);
  r = r < 0 ? 0 : (r > N - 1 ? N - 1 : r);
  while (r + 1 < N && gn_pair_start(r + 1, N) <= p) ++r;
  while (r > 0 && gn_pair_start(r, N) > p) --r;
  i = r;
  j = r + (p - gn_pair_start(r, N));
}
