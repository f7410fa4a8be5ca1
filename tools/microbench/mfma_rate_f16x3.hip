// Feasibility of fp32-accurate products from TWO fp16 parts per operand ("f16x3"): x = xh + xl (11 + 11 significant
// bits, remainder exact in fp32), w.x ~ wh.xl + wl.xh + wh.xh — three v_mfma_f32_32x32x16_f16 per k = 16 step instead
// of the six bf16 part-products of "bf16x6".  Measures (a) the chain rate in the shape the MLP chains use it (2 weight
// loads + 8 splits + 3 MFMAs per step) next to the bf16x6 form in the same process, (b) the numerics on the device
// against an fp64 host reference, including operands whose low parts are fp16 subnormals.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o mfma_rate_f16x3 mfma_rate_f16x3.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2(const float (&x)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x2 v = {x[2 * j], x[2 * j + 1]};
    const f16x2 h = __builtin_convertvector(v, f16x2);
    const f32x2 r = {v[0] - (float)h[0], v[1] - (float)h[1]};
    const f16x2 l = __builtin_convertvector(r, f16x2);
    hi[2 * j] = h[0], hi[2 * j + 1] = h[1];
    lo[2 * j] = l[0], lo[2 * j + 1] = l[1];
  }
}

// MODE 3: bf16x6 (as mfma_rate_bf16.hip); MODE 4: f16x3; MODE 5: f16x3 without the split (operands constant)
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ W, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 a0, src;
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; src[r] = (float)(lane + r) * 1e-3f; }
  const f32x4* p = reinterpret_cast<const f32x4*>(W) + lane;
  f32x4 ring[8];
  for (int i = 0; i < 8; ++i) ring[i] = p[i * 64];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (MODE == 3) {
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
        f16x8 xh, xl;
        if (MODE == 4) {
          float x[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) x[r] = src[(8 * (s & 1)) + r];
          split2(x, xh, xl);
        } else {
          for (int r = 0; r < 8; ++r) { xh[r] = (_Float16)1e-3f; xl[r] = (_Float16)1e-6f; }
        }
        const f16x8 wh = __builtin_bit_cast(f16x8, ring[(2 * s) % 8]);
        const f16x8 wl = __builtin_bit_cast(f16x8, ring[(2 * s + 1) % 8]);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, a0, 0, 0, 0);
        ring[(2 * s) % 8] = p[((it * 16 + 2 * s + 8) & 1023) * 64];
        ring[(2 * s + 1) % 8] = p[((it * 16 + 2 * s + 9) & 1023) * 64];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // keep the split live: the next iteration's operands depend on the accumulator
#pragma unroll
    for (int r = 0; r < 16; ++r) src[r] = fmaf(a0[r], 1e-30f, src[r]);
  }
  float acc = 0.f;
  for (int r = 0; r < 16; ++r) acc += a0[r];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// ---- numerics: Y (32 x 32) = W (32 x K) X (K x 32), one wave, K = 16 * steps -----------------------------------------
__global__ void gemm_f16x3(const float* __restrict__ W, const float* __restrict__ X, float* __restrict__ Y, int steps) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int K = 16 * steps;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int s = 0; s < steps; ++s) {
    float w[8], x[8];
    for (int j = 0; j < 8; ++j) {
      w[j] = W[r * K + 16 * s + 8 * h + j];        // A[row r][k = 8h + j]
      x[j] = X[(16 * s + 8 * h + j) * 32 + r];     // B[k = 8h + j][col r]
    }
    f16x8 wh, wl, xh, xl;
    split2(w, wh, wl);
    split2(x, xh, xl);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) Y[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

static double check(float wscale, float xscale, int steps) {
  const int K = 16 * steps;
  std::vector<float> W(32 * K), X(K * 32), Y(32 * 32);
  for (auto& v : W) v = wscale * (2.f * rand() / RAND_MAX - 1.f);
  for (auto& v : X) v = xscale * (2.f * rand() / RAND_MAX - 1.f);
  float *dW, *dX, *dY;
  hipMalloc(&dW, W.size() * 4); hipMalloc(&dX, X.size() * 4); hipMalloc(&dY, Y.size() * 4);
  hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(gemm_f16x3, dim3(1), dim3(64), 0, 0, dW, dX, dY, steps);
  hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost);
  double maxerr = 0, maxref = 0, maxf32 = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double ref = 0; float f = 0.f;
      for (int kk = 0; kk < K; ++kk) { ref += (double)W[i * K + kk] * X[kk * 32 + j]; f = fmaf(W[i * K + kk], X[kk * 32 + j], f); }
      maxerr = fmax(maxerr, fabs(Y[i * 32 + j] - ref));
      maxf32 = fmax(maxf32, fabs((double)f - ref));
      maxref = fmax(maxref, fabs(ref));
    }
  printf("numerics K=%4d |w|<=%g |x|<=%g: max|err| %.3e (fp32 fma chain %.3e)  max|ref| %.3e  rel %.2e\n", K, wscale, xscale,
         maxerr, maxf32, maxref, maxerr / maxref);
  hipFree(dW); hipFree(dX); hipFree(dY);
  return maxerr / maxref;
}

int main() {
  for (float ws : {0.125f, 0.01f, 1.f})
    for (float xs : {1.f, 0.01f, 1e-4f, 300.f}) check(ws, xs, 16);
  check(0.09f, 1.f, 4);
  check(0.06f, 1.f, 8);
  float *W, *out;
  hipMalloc(&W, 1024 * 64 * 16 + 4096);
  {
    std::vector<unsigned short> hw(1024 * 64 * 8);
    for (auto& v : hw) v = (unsigned short)(0x2000 + (rand() & 0x0fff));      // small positive fp16 / bf16 patterns
    hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  }
  hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 400;
  for (int round = 0; round < 2; ++round)
    for (int mode = 3; mode < 6; ++mode)
      for (int wgs : {256, 512, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(wgs), dim3(256), 0, 0, W, out, iters);
          if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(wgs), dim3(256), 0, 0, W, out, iters);
          if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(wgs), dim3(256), 0, 0, W, out, iters);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (rep == 1) {
            const double steps = (double)wgs * 4 * iters * 8;
            printf("mode %d wgs %4d: %8.1f us  %7.1f TFLOP/s fp32-equivalent\n", mode, wgs, ms * 1e3,
                   steps * 2.0 * 32 * 32 * 16 / ms / 1e9);
          }
        }
      }
  return 0;
}
