// Shared device/host helpers for the gfx950 kernels of the GroupNet MS-HGNN path.
// Written for CDNA4 only: 64-lane wavefronts, v_mfma_f32_32x32x2_f32, 160 KiB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
