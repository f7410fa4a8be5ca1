// What paces a lone wave in the layer-pair loop of gn_mlp_bf16.hpp (P = 1: one v_mfma_f32_32x32x16_bf16 per sub-step)?
// One hidden tile = A (4 sub-steps into `hid`), V (ReLU + convert hid -> bf16 operands), B (4 sub-steps into `out`).
// MODE bits: 1 operands read from LDS every sub-step (QD sub-steps ahead)   2 chunk boundary every 8 sub-steps (raw
// barrier + staged ds_write + staging global loads)   4 V work   8 bias tile loaded per hidden tile (global, two tiles
// ahead)   32 barrier only (no staging) at the boundary.  Staging runs LOOK = 4 chunks ahead.  Without V the A phase is
// dead code (4 MFMAs per tile: ideal 128 cycles); with V 8 MFMAs (ideal 256).  Prints cycles per hidden tile.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short i16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int QD, int WAVES>
__global__ __launch_bounds__(256) void k(const f32x4* __restrict__ W, const float* __restrict__ bias, float* out_g,
                                         long long* cyc, int tiles) {
  __shared__ f32x4 ring[3 * 8 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= WAVES) return;
  // fill the ring once
  for (int i = threadIdx.x; i < 3 * 8 * 64; i += WAVES * 64) ring[i] = W[i];
  __syncthreads();
  const f32x4* rd = ring + lane;
  const f32x4 *rdc = rd, *rdn = rd + 8 * 64;       // running pointers as in WStream: current / next chunk
  f32x4* wr = ring + 2 * 8 * 64 + wave * 128 + lane;
  f32x4 q[QD];
#pragma unroll
  for (int j = 0; j + 1 < QD; ++j) q[j] = rd[j * 64];
  bf16x8 xi[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) xi[i] = __builtin_bit_cast(bf16x8, W[lane + 64 * i]);
  f32x16 out[2], hid, hidn;
  for (int r = 0; r < 16; ++r) out[0][r] = out[1][r] = 0.f, hid[r] = 0.f, hidn[r] = 0.f;
  f32x4 st[4][2];
  const f32x4* ld = W + lane;
  for (int j = 0; j < 4; ++j) st[j][0] = ld[0], st[j][1] = ld[64], ld += 128;
  int cl = 0;
  f32x4 bn[4];
  for (int i = 0; i < 4; ++i) bn[i] = *reinterpret_cast<const f32x4*>(bias + 8 * i + 4 * (lane >> 5));
  int slot = 0;
  const long long t0 = __builtin_amdgcn_s_memtime();
  auto operand = [&](int s) -> bf16x8 {
    if (MODE & 1) {
      const int idx = s % 8 + QD - 1;
      q[(s + QD - 1) % QD] = (idx < 8 ? rdc : rdn)[(idx % 8) * 64];
      return __builtin_bit_cast(bf16x8, q[s % QD]);
    }
    return __builtin_bit_cast(bf16x8, q[0]);
  };
  auto boundary = [&](int u) {
    if (MODE & 32) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (MODE & 2) {
      if (MODE & 64) asm volatile("s_barrier" ::: "memory");      // operand queue stays in flight across the boundary
      else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      slot = slot == 2 ? 0 : slot + 1;
      rdc = rdn;
      rdn = rd + (slot == 2 ? 0 : slot + 1) * 8 * 64;
      wr[0] = st[u][0];
      wr[64] = st[u][1];
      wr = slot == 0 ? wr - 2 * 8 * 64 : wr + 8 * 64;
      st[u][0] = ld[0];
      st[u][1] = ld[64];
      ld = ld + 128 >= W + 6144 ? W + lane : ld + 128;       // a 96 KiB image every wave walks: L2-resident, as in the kernels
    }
  };
  auto vwork = [&](const f32x16& hsrc, bf16x8 (&xh)[2]) {
    if (MODE & 4) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        bf16x8 tq;
#pragma unroll
        for (int j = 0; j < 8; ++j) tq[j] = (__bf16)hsrc[8 * hf + j];
        const i16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        xh[hf] = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(i16x8, tq), z));
      }
    } else {
      xh[0] = xi[0], xh[1] = xi[1];
    }
  };
  auto bias_tile = [&](f32x16& hdst, int tt) {
    if (MODE & 8) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        hdst[4 * i] = bn[i][0], hdst[4 * i + 1] = bn[i][1], hdst[4 * i + 2] = bn[i][2], hdst[4 * i + 3] = bn[i][3];
        bn[i] = *reinterpret_cast<const f32x4*>(bias + (tt & 63) * 32 + 8 * i + 4 * (lane >> 5));
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) hdst[r] = 0.f;
    }
  };
  if (MODE & 16) {
    // software-pipelined: A_{t+1} is issued before V_t / B_t, V_t interleaved with its MFMAs (as layer_pair does).
    // Stream positions per tile: 8 sub-steps; boundaries every 8.
    bias_tile(hid, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      hid = __builtin_amdgcn_mfma_f32_32x32x16_bf16(operand(s), xi[s], hid, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll 1
    for (int t = 0; t < tiles; t += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        boundary(u);
        f32x16 cur = hid;
        bias_tile(hid, t + u);
        bf16x8 xh[2];
        vwork(cur, xh);
#pragma unroll
        for (int s = 0; s < 4; ++s) hid = __builtin_amdgcn_mfma_f32_32x32x16_bf16(operand(s), xi[s], hid, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 4; s < 8; ++s) {
          out[(s >> 1) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(operand(s), xh[s & 1], out[(s >> 1) & 1], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  } else {
#pragma unroll 1
  for (int t = 0; t < tiles; t += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {                         // 4 hidden tiles per pass: 32 sub-steps = 4 chunks
      bias_tile(hid, t + u);
      boundary(u);
      // A: 4 sub-steps
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        hid = __builtin_amdgcn_mfma_f32_32x32x16_bf16(operand(s), xi[s], hid, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      bf16x8 xh[2];
      vwork(hid, xh);
      // B: 4 sub-steps
#pragma unroll
      for (int s = 4; s < 8; ++s) {
        out[(s >> 1) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(operand(s), xh[s & 1], out[(s >> 1) & 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0.f;
  for (int r = 0; r < 16; ++r) acc += out[0][r] + out[1][r] + hid[r];
  out_g[blockIdx.x * 256 + threadIdx.x] = acc;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

// ---- RB row blocks per wave: every weight operand read from LDS feeds RB MFMAs (one per row block) ----------------
// pipelined, LDS operands QD-1 ahead, chunk boundary (barrier without draining the operand queue, staged writes),
// bias tiles, V = convert + packed ReLU pair by pair.  Cycles are reported per hidden tile of ONE row block.
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 cvt_relu(const f32x16& v, int hf) {
  u32x4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x2 p = {v[8 * hf + 2 * j], v[8 * hf + 2 * j + 1]};
    const i16x2 z = {0, 0};
    r[j] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, __builtin_convertvector(p, bf16x2)), z));
  }
  return __builtin_bit_cast(bf16x8, r);
}
template <int RB, int QD, bool DRAIN, bool BIASC, int BND = 7, int MINB = 1>
__global__ __launch_bounds__(256, MINB) void k2(const f32x4* __restrict__ W, const float* __restrict__ bias, float* out_g,
                                          long long* cyc, int tiles) {
  __shared__ f32x4 ring[3 * 8 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 3 * 8 * 64; i += 256) ring[i] = W[i];
  __syncthreads();
  const f32x4* rd = ring + lane;
  const f32x4 *rdc = rd, *rdn = rd + 8 * 64;
  f32x4* wr = ring + 2 * 8 * 64 + wave * 128 + lane;
  f32x4 q[QD];
#pragma unroll
  for (int j = 0; j + 1 < QD; ++j) q[j] = rd[j * 64];
  bf16x8 xi[RB][4];
#pragma unroll
  for (int b = 0; b < RB; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i) xi[b][i] = __builtin_bit_cast(bf16x8, W[lane + 64 * (i + 4 * b)]);
  f32x16 out[RB][2], hid[RB];
  for (int b = 0; b < RB; ++b)
    for (int r = 0; r < 16; ++r) out[b][0][r] = out[b][1][r] = 0.f, hid[b][r] = 0.f;
  f32x4 st[4][2];
  const f32x4* ld = W + lane;
  for (int j = 0; j < 4; ++j) st[j][0] = ld[0], st[j][1] = ld[64], ld += 128;
  f32x16 bn;
  for (int i = 0; i < 4; ++i) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(bias + 8 * i + 4 * (lane >> 5));
    bn[4 * i] = v[0], bn[4 * i + 1] = v[1], bn[4 * i + 2] = v[2], bn[4 * i + 3] = v[3];
  }
  int slot = 0;
  const long long t0 = __builtin_amdgcn_s_memtime();
  auto operand = [&](int s) -> bf16x8 {
    const int idx = s % 8 + QD - 1;
    q[(s + QD - 1) % QD] = (idx < 8 ? rdc : rdn)[(idx % 8) * 64];
    return __builtin_bit_cast(bf16x8, q[s % QD]);
  };
  auto boundary = [&](int u) {
    // BND bits: 1 barrier, 2 staged LDS writes, 4 staging global loads
    if (BND & 1) {
      if (DRAIN) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_barrier" ::: "memory");
    }
    slot = slot == 2 ? 0 : slot + 1;
    rdc = rdn;
    rdn = rd + (slot == 2 ? 0 : slot + 1) * 8 * 64;
    if (BND & 2) {
      wr[0] = st[u][0];
      wr[64] = st[u][1];
    }
    wr = slot == 0 ? wr - 2 * 8 * 64 : wr + 8 * 64;
    if (BND & 4) {
      st[u][0] = ld[0];
      st[u][1] = ld[64];
    }
    ld = ld + 128 >= W + 6144 ? W + lane : ld + 128;
  };
  auto next_bias = [&](int tt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(bias + (tt & 63) * 32 + 8 * i + 4 * (lane >> 5));
      bn[4 * i] = v[0], bn[4 * i + 1] = v[1], bn[4 * i + 2] = v[2], bn[4 * i + 3] = v[3];
    }
  };
  // A_0
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const bf16x8 w = operand(s);
#pragma unroll
    for (int b = 0; b < RB; ++b) hid[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, xi[b][s], hid[b], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll 1
  for (int t = 0; t < tiles; t += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      boundary(u);
      f32x16 cur[RB];
      bf16x8 xh[RB][2];
#pragma unroll
      for (int b = 0; b < RB; ++b) {
        cur[b] = hid[b];
        xh[b][0] = cvt_relu(cur[b], 0);
        xh[b][1] = cvt_relu(cur[b], 1);
      }
      // A_{t+1}: the first sub-step takes the bias tile as its C operand (BIASC) or the accumulators are copies of it
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 w = operand(s);
#pragma unroll
        for (int b = 0; b < RB; ++b) {
          if (s == 0 && BIASC) hid[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, xi[b][s], bn, 0, 0, 0);
          else {
            if (s == 0) hid[b] = bn;
            hid[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, xi[b][s], hid[b], 0, 0, 0);
          }
        }
      }
      next_bias(t + u);
#pragma unroll
      for (int i = 0; i < 4 * RB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        if (i % RB == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 4; s < 8; ++s) {
        const bf16x8 w = operand(s);
#pragma unroll
        for (int b = 0; b < RB; ++b)
          out[b][(s >> 1) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, xh[b][s & 1], out[b][(s >> 1) & 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0.f;
  for (int b = 0; b < RB; ++b)
    for (int r = 0; r < 16; ++r) acc += out[b][0][r] + out[b][1][r] + hid[b][r];
  out_g[blockIdx.x * 256 + threadIdx.x] = acc;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int RB, int QD, bool DRAIN, bool BIASC, int BND = 7, int MINB = 1>
void run2(const char* what, const f32x4* W, const float* bias, float* out, long long* cyc, int wgs) {
  const int tiles = 400;
  hipLaunchKernelGGL((k2<RB, QD, DRAIN, BIASC, BND, MINB>), dim3(wgs), dim3(256), 0, 0, W, bias, out, cyc, tiles);
  hipLaunchKernelGGL((k2<RB, QD, DRAIN, BIASC, BND, MINB>), dim3(wgs), dim3(256), 0, 0, W, bias, out, cyc, tiles);
  (void)hipDeviceSynchronize();
  static long long h[4096];
  (void)hipMemcpy(h, cyc, sizeof(long long) * wgs * 4, hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < wgs * 4; ++i) s += h[i];
  printf("%-34s minb=%d bnd=%d RB=%d QD=%d drain=%d biasC=%d wgs=%4d : %7.1f cycles per hidden tile per row block (ideal 256)\n", what, MINB, BND, RB, QD,
         (int)DRAIN, (int)BIASC, wgs, s / (wgs * 4) / tiles / RB);
}

template <int MODE, int QD, int WAVES>
void run(const char* what, const f32x4* W, const float* bias, float* out, long long* cyc, int wgs) {
  const int tiles = 400;
  hipLaunchKernelGGL((k<MODE, QD, WAVES>), dim3(wgs), dim3(256), 0, 0, W, bias, out, cyc, tiles);
  hipLaunchKernelGGL((k<MODE, QD, WAVES>), dim3(wgs), dim3(256), 0, 0, W, bias, out, cyc, tiles);
  (void)hipDeviceSynchronize();
  static long long h[4096];
  (void)hipMemcpy(h, cyc, sizeof(long long) * wgs * 4, hipMemcpyDeviceToHost);
  double s = 0;
  int n = 0;
  for (int i = 0; i < wgs * 4; ++i)
    if (i % 4 < WAVES) s += h[i], ++n;
  printf("%-64s QD=%d waves/wg=%d wgs=%4d : %7.1f cycles per hidden tile (ideal 256)\n", what, QD, WAVES, wgs, s / n / tiles);
}

int main() {
  f32x4* W;
  float *bias, *out;
  long long* cyc;
  (void)hipMalloc(&W, 16 << 20);
  (void)hipMemset(W, 0x3c, 16 << 20);
  (void)hipMalloc(&bias, 1 << 20);
  (void)hipMemset(bias, 0, 1 << 20);
  (void)hipMalloc(&out, 4096 * 256 * 4);
  (void)hipMalloc(&cyc, 4096 * 8 * 4);
  for (int wgs : {256, 512, 768}) {      // forced two workgroups per CU
    run2<2, 4, false, true, 7, 2>("full pipeline", W, bias, out, cyc, wgs);
    run2<2, 8, false, true, 7, 2>("full pipeline", W, bias, out, cyc, wgs);
    run2<1, 4, false, true, 7, 3>("full pipeline", W, bias, out, cyc, wgs);
    run2<1, 4, false, true, 7, 4>("full pipeline", W, bias, out, cyc, wgs);
  }
  for (int wgs : {768, 1024}) {      // 3 and 4 waves per SIMD where the registers allow
    run2<1, 4, false, true>("full pipeline", W, bias, out, cyc, wgs);
    run2<1, 8, false, true>("full pipeline", W, bias, out, cyc, wgs);
    run2<2, 4, false, true>("full pipeline", W, bias, out, cyc, wgs);
  }
  for (int wgs : {256, 512}) {
    run<0, 4, 4>("MFMA only (dependent chains, no memory)", W, bias, out, cyc, wgs);
    run<1, 2, 4>("+ operands from LDS, 1 ahead", W, bias, out, cyc, wgs);
    run<1, 4, 4>("+ operands from LDS, 3 ahead", W, bias, out, cyc, wgs);
    run<1, 8, 4>("+ operands from LDS, 7 ahead", W, bias, out, cyc, wgs);
    run<33, 4, 4>("+ raw barrier per 8 sub-steps only", W, bias, out, cyc, wgs);
    run<3, 4, 4>("+ chunk boundary (barrier, staging 4 chunks ahead)", W, bias, out, cyc, wgs);
    run<5, 4, 4>("+ V (convert + packed ReLU), no boundary", W, bias, out, cyc, wgs);
    run<7, 4, 4>("+ V + boundary", W, bias, out, cyc, wgs);
    run<15, 4, 4>("+ V + boundary + bias tiles", W, bias, out, cyc, wgs);
    run<15, 8, 4>("+ V + boundary + bias tiles, 7 ahead", W, bias, out, cyc, wgs);
    run<20, 4, 4>("pipelined: V under A(t+1), no boundary", W, bias, out, cyc, wgs);
    run<21, 4, 4>("pipelined + LDS operands", W, bias, out, cyc, wgs);
    run<21, 8, 4>("pipelined + LDS operands, 7 ahead", W, bias, out, cyc, wgs);
    run<23, 4, 4>("pipelined + LDS operands + boundary", W, bias, out, cyc, wgs);
    run<31, 4, 4>("pipelined + LDS operands + boundary + bias", W, bias, out, cyc, wgs);
    run<31, 8, 4>("pipelined + LDS operands + boundary + bias, 7 ahead", W, bias, out, cyc, wgs);
    run2<1, 4, false, true, 0>("no boundary work", W, bias, out, cyc, wgs);
    run2<1, 4, false, true, 1>("barrier", W, bias, out, cyc, wgs);
    run2<1, 4, false, true, 2>("LDS writes", W, bias, out, cyc, wgs);
    run2<1, 4, false, true, 4>("staging loads", W, bias, out, cyc, wgs);
    run2<1, 4, false, true, 6>("writes + loads", W, bias, out, cyc, wgs);
    run2<1, 4, false, true, 3>("barrier + writes", W, bias, out, cyc, wgs);
    run2<1, 4, false, true, 5>("barrier + loads", W, bias, out, cyc, wgs);
    run2<2, 4, false, true, 0>("no boundary work", W, bias, out, cyc, wgs);
    run2<2, 4, false, true, 1>("barrier", W, bias, out, cyc, wgs);
    run2<2, 4, false, true, 6>("writes + loads", W, bias, out, cyc, wgs);
    run2<1, 4, true, false>("full pipeline", W, bias, out, cyc, wgs);
    run2<1, 4, false, false>("full pipeline", W, bias, out, cyc, wgs);
    run2<1, 4, false, true>("full pipeline", W, bias, out, cyc, wgs);
    run2<1, 8, false, true>("full pipeline", W, bias, out, cyc, wgs);
    run2<2, 4, true, false>("full pipeline", W, bias, out, cyc, wgs);
    run2<2, 4, false, false>("full pipeline", W, bias, out, cyc, wgs);
    run2<2, 4, false, true>("full pipeline", W, bias, out, cyc, wgs);
    run2<2, 8, false, true>("full pipeline", W, bias, out, cyc, wgs);
    run2<4, 4, false, true>("full pipeline", W, bias, out, cyc, wgs);
    run<64 + 3, 4, 4>("boundary without lgkmcnt(0), 3 ahead", W, bias, out, cyc, wgs);
    run<64 + 23, 4, 4>("pipelined + LDS + boundary without lgkmcnt(0)", W, bias, out, cyc, wgs);
    run<64 + 31, 4, 4>("pipelined + LDS + boundary w/o lgkmcnt(0) + bias", W, bias, out, cyc, wgs);
    run<64 + 31, 8, 4>("pipelined + LDS + boundary w/o lgkmcnt(0) + bias, 7 ahead", W, bias, out, cyc, wgs);
  }
  return 0;
}
