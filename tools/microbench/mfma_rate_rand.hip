#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// MODE 0: one dependent chain, operands in registers. MODE 1: two chains. MODE 2: one chain + ring loads (depth 8).
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ W, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 a0, a1;
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
  float b = W[lane + 64] , w = W[lane];
  const f32x4* p = reinterpret_cast<const f32x4*>(W) + lane;
  f32x4 ring[8];
  if (MODE == 2) for (int i = 0; i < 8; ++i) ring[i] = p[i * 64];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (MODE == 0) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b, a0, 0, 0, 0);
      } else if (MODE == 1) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b, a1, 0, 0, 0);
      } else {
        const f32x4 ww = ring[s];
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ww[0], b, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ww[1], b, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ww[2], b, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ww[3], b, a0, 0, 0, 0);
        ring[s] = p[((it * 8 + s + 8) & 1023) * 64];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float acc = 0.f;
  for (int r = 0; r < 16; ++r) acc += a0[r] + a1[r];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
  float *W, *out;
  hipMalloc(&W, 1024 * 64 * 16 + 4096);
  { std::vector<float> hbuf(1024 * 64 * 4 + 1024); for (auto& v : hbuf) v = (float)rand() / RAND_MAX * 2.f - 1.f; hipMemcpy(W, hbuf.data(), hbuf.size() * 4, hipMemcpyHostToDevice); }
  hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 400;  // 400*32 = 12800 MFMAs per wave
  for (int mode = 0; mode < 3; ++mode)
    for (int wgs : {256, 512, 1024}) {   // 1, 2, 4 WG (of 4 waves) per CU
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, W, out, iters);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, W, out, iters);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 0, 0, W, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 1) {
          double flop = (double)wgs * 4 * iters * 32 * 4096.0;
          printf("mode %d wgs %4d: %8.1f us  %6.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4GHz)\n", mode, wgs, ms * 1e3, flop / ms / 1e9,
                 ms * 1e-3 * 2.4e9 / (iters * 32.0 * (wgs / 256.0)));
        }
      }
    }
  return 0;
}
