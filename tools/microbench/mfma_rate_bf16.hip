// What v_mfma_f32_32x32x16_bf16 sustains on gfx950 in the shape the bf16 twins of the MLP chains would use it:
// a dependent accumulation chain whose B operand is produced from fp32 accumulator registers (v_cvt_pk_bf16_f32)
// and whose A operand comes from a weight ring (one dwordx4 = 8 bf16 = one MFMA of k = 16).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o mfma_rate_bf16 mfma_rate_bf16.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// MODE 0: one dependent chain, constant operands; MODE 1: two chains; MODE 2: one chain, A from a ring of
// depth 8 (one 16-byte load per MFMA), B converted from 8 fp32 registers each step.
// MODE 3: fp32-accurate product from bf16 parts ("bf16x6"): a = a1+a2+a3, b = b1+b2+b3 (8 mantissa bits each),
// a.b ~ a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1 — six MFMAs per k = 16 step; the three weight parts come from
// the ring (3 loads per step, pre-split offline), the activation parts are split on the VALU every step.
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ W, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 a0, a1, src;
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; src[r] = (float)(lane + r) * 1e-3f; }
  bf16x8 w, b;
  for (int r = 0; r < 8; ++r) { w[r] = (__bf16)1e-3f; b[r] = (__bf16)((float)lane * 1e-3f); }
  const f32x4* p = reinterpret_cast<const f32x4*>(W) + lane;
  f32x4 ring[8];
  if (MODE >= 2) for (int i = 0; i < 8; ++i) ring[i] = p[i * 64];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (MODE == 0) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, b, a0, 0, 0, 0);
      } else if (MODE == 1) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, b, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, b, a1, 0, 0, 0);
      } else if (MODE == 3) {
        bf16x8 b1, b2, b3;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float x = src[(8 * (s & 1)) + r];
          const __bf16 h1 = (__bf16)x;
          const float r1 = x - (float)h1;
          const __bf16 h2 = (__bf16)r1;
          const __bf16 h3 = (__bf16)(r1 - (float)h2);
          b1[r] = h1; b2[r] = h2; b3[r] = h3;
        }
        const bf16x8 w1 = __builtin_bit_cast(bf16x8, ring[(3 * s) % 8]);
        const bf16x8 w2 = __builtin_bit_cast(bf16x8, ring[(3 * s + 1) % 8]);
        const bf16x8 w3 = __builtin_bit_cast(bf16x8, ring[(3 * s + 2) % 8]);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3, b1, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, b2, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, b3, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, b1, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, b2, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, b1, a0, 0, 0, 0);
        ring[(3 * s) % 8] = p[((it * 24 + 3 * s + 8) & 1023) * 64];
        ring[(3 * s + 1) % 8] = p[((it * 24 + 3 * s + 9) & 1023) * 64];
        ring[(3 * s + 2) % 8] = p[((it * 24 + 3 * s + 10) & 1023) * 64];
        __builtin_amdgcn_sched_barrier(0);
      } else {
        const f32x4 ww = ring[s];
        bf16x8 wa = __builtin_bit_cast(bf16x8, ww);
        bf16x8 bb;
#pragma unroll
        for (int r = 0; r < 8; ++r) bb[r] = (__bf16)src[(8 * (s & 1)) + r];   // fp32 activations -> bf16 operand
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, bb, a0, 0, 0, 0);
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
  hipMemset(W, 0, 1024 * 64 * 16 + 4096);
  hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 1600;
  for (int mode = 0; mode < 4; ++mode)
    for (int wgs : {256, 512, 1024}) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, W, out, iters);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, W, out, iters);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 0, 0, W, out, iters);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(wgs), dim3(256), 0, 0, W, out, iters / 4);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 1) {
          // mode 3: one step = six MFMAs = ONE fp32-accurate 32x32x16 product; its rate is quoted in
          // fp32-equivalent FLOPs (2*32*32*16 per step)
          const double steps = (double)wgs * 4 * (mode == 3 ? iters / 4 : iters) * 8 * (mode == 1 ? 2 : 1);
          printf("mode %d wgs %4d: %8.1f us  %7.1f TFLOP/s%s\n", mode, wgs, ms * 1e3, steps * 2.0 * 32 * 32 * 16 / ms / 1e9,
                 mode == 3 ? " fp32-equivalent" : "");
        }
      }
    }
  return 0;
}
