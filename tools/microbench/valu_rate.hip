// Issue rate of the VALU instructions the node form's partner loop is made of, one wave per SIMD and two:
// v_fma_f32, v_max_f32, v_pk_add_f32, v_pk_fma_f32 (plain, with an op_sel broadcast, with clamp).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
  f32x2 a[8], b[8], c = {1.0001f, 0.9999f};
  float s[16];
  for (int i = 0; i < 8; ++i) { a[i] = f32x2{(float)threadIdx.x * 1e-3f + i, 0.5f + i}; b[i] = f32x2{0.25f * i, 0.125f * i}; }
  for (int i = 0; i < 16; ++i) s[i] = threadIdx.x * 1e-4f + i;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[2 * i]) : "v"(s[(2 * i + 3) & 15]), "v"(c[0])); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[2 * i + 1]) : "v"(s[(2 * i + 5) & 15]), "v"(c[1])); }
      if (MODE == 1) { asm volatile("v_max_f32 %0, %1, %0" : "+v"(s[2 * i]) : "v"(c[0])); asm volatile("v_max_f32 %0, %1, %0" : "+v"(s[2 * i + 1]) : "v"(c[1])); }
      if (MODE == 2) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
      if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c));
      if (MODE == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(c), "v"(b[i]));
      if (MODE == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(a[i]) : "v"(b[i]), "v"(c), "v"(a[(i + 1) & 7]));
      if (MODE == 6) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
  for (int i = 0; i < 8; ++i) r += a[i][0] + a[i][1];
  for (int i = 0; i < 16; ++i) r += s[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
void run(const char* name, int per_iter, int blocks) {
  float* out; unsigned long long* cyc; hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, 8);
  const int iters = 20000;
  k<MODE><<<blocks, 256>>>(out, iters, cyc); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); k<MODE><<<blocks, 256>>>(out, iters, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-28s blocks %4d: %.2f memtime ticks per instruction (wave 0), %.3f ns per instruction per wave\n", name, blocks, (double)c / iters / per_iter, ms * 1e6 / iters / per_iter);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int blocks : {256, 512, 1024, 2048}) {
    run<0>("v_fma_f32", 16, blocks); run<1>("v_max_f32", 16, blocks); run<2>("v_pk_add_f32", 8, blocks); run<3>("v_pk_fma_f32", 8, blocks);
    run<4>("v_pk_fma_f32 op_sel_hi", 8, blocks); run<5>("v_pk_fma_f32 clamp", 8, blocks); run<6>("v_pk_mul_f32", 8, blocks);
  }
  return 0;
}
