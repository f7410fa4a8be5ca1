#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int UNROLL>
__global__ __launch_bounds__(256) void k(const float* __restrict__ W, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 a0;
  for (int r = 0; r < 16; ++r) a0[r] = 0.f;
  float b = W[lane + 64], w = W[lane];
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < UNROLL; ++s) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b, a0, 0, 0, 0);
      w += 1.0f;   // keep instructions distinct-ish, prevent CSE
    }
  }
  float acc = 0.f;
  for (int r = 0; r < 16; ++r) acc += a0[r];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int U> void run(const float* W, float* out, int total, int wgs) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<U>, dim3(wgs), dim3(256), 0, 0, W, out, total / U);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("unroll %4d (total %d MFMA/wave, %d WGs): %7.1f us   ideal@2.1GHz %.1f us\n", U, total, wgs, best * 1e3, total * 64.0 / 2.1e3);
}
int main() {
  float *W, *out; (void)hipMalloc(&W, 4096); (void)hipMemset(W, 0, 4096); (void)hipMalloc(&out, 4096 * 256 * 4);
  for (int wgs : {176, 704}) {
    run<16>(W, out, 576, wgs);
    run<64>(W, out, 576, wgs);
    run<192>(W, out, 576, wgs);
    run<576>(W, out, 576, wgs);
    run<16>(W, out, 160, wgs);
    run<160>(W, out, 160, wgs);
  }
  return 0;
}
