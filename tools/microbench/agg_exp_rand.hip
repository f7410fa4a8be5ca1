#include "../../groupnet_amd/csrc/gn_mlp_mfma.hip"
#include <stdio.h>
#include <vector>
#include <stdlib.h>
namespace {
// FLAGS bit0: bias loads, bit1: ef/b2 loads + b2 MFMA, bit2: relu side, bit3: B operand from arrays (else constant)
template <int FLAGS>
__global__ __launch_bounds__(256) void agg_exp(const float* __restrict__ eo, const float* __restrict__ ef,
                                               const float* __restrict__ W, const float* __restrict__ b1,
                                               const float* __restrict__ b2, float* __restrict__ feat, int rows, int K) {
  const int wave = wave_id();
  const int blk = blockIdx.x * 4 + wave;
  if (blk * 32 >= rows) return;
  const RowBlock rb = row_block(rows, blk);
  const int lane = rb.lane, h = rb.h;
  f32x16 in[2], hid[4], out[2];
  load_rows<2>(eo, GN_FEAT, rb.row_ld, h, in);
  for (int o = 0; o < 2; ++o) for (int r = 0; r < 16; ++r) out[o][r] = 0.f;
  const float* efrow = ef + (size_t)rb.row_ld * K;
  const f32x4* Wl = reinterpret_cast<const f32x4*>(W) + lane;
  int k = 0;
  WRing ring;
  ring_prime(ring, Wl);
  f32x16 bnext = load_bias_tile(b1, h);
  float efk = efrow[0];
  float b2f0 = h == 0 ? b2[(lane & 31)] : 0.f, b2f1 = h == 0 ? b2[32 + (lane & 31)] : 0.f;
#pragma unroll 1
  while (k < K) {
    const int kn = k + 1;
    const int kc = kn < K ? kn : k;
    const f32x4* base = Wl + (size_t)k * kTypeSteps * kStep;
    const f32x4* base_next = Wl + (size_t)kc * kTypeSteps * kStep;
    float efk_next = efk, b2n0 = b2f0, b2n1 = b2f1;
    if (FLAGS & 2) {
      efk_next = efrow[kc];
      b2n0 = h == 0 ? b2[kc * 64 + (lane & 31)] : 0.f;
      b2n1 = h == 0 ? b2[kc * 64 + 32 + (lane & 31)] : 0.f;
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      hid[o] = bnext;
      if (FLAGS & 1) bnext = load_bias_tile(o < 3 ? b1 + k * 128 + 32 * (o + 1) : b1 + kc * 128, h);
      mma_tile<2>(base + o * 8 * kStep, base + (o + 1) * 8 * kStep, ring, in, hid[o], [&](int s) {
        if ((FLAGS & 4) && o > 0 && s == 1) relu_scale16(hid[o > 0 ? o - 1 : 0], efk);
      });
    }
    const float efb = h == 0 ? efk : 0.f;
    if (FLAGS & 2) out[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b2f0, efb, out[0], 0, 0, 0);
    mma_tile<4>(base + 32 * kStep, base + 48 * kStep, ring, hid, out[0], [&](int s) {
      if ((FLAGS & 4) && s == 1) relu_scale16(hid[3], efk);
    });
    if (FLAGS & 2) out[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b2f1, efb, out[1], 0, 0, 0);
    mma_tile<4>(base + 48 * kStep, base_next, ring, hid, out[1]);
    efk = efk_next; b2f0 = b2n0; b2f1 = b2n1;
    k = kn;
  }
  store_rows<2>(feat, GN_FEAT, rb.row, h, rb.live, out);
}
}  // namespace
template <int F> float run(const float* eo, const float* ef, const float* W, const float* b1, const float* b2, float* feat, int rows, int K) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((agg_exp<F>), dim3((rows + 127) / 128), dim3(256), 0, 0, eo, ef, W, b1, b2, feat, rows, K);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    if (ms < best) best = ms;
  }
  return best * 1e3f;
}
int main() {
  const int K = 6;
  for (int rows : {61952, 65536}) {
  float *eo, *ef, *W, *b1, *b2, *feat;
  (void)hipMalloc(&eo, (size_t)rows * 64 * 4); (void)hipMalloc(&feat, (size_t)rows * 64 * 4); (void)hipMalloc(&ef, (size_t)rows * K * 4);
  (void)hipMalloc(&W, (size_t)K * 16384 * 4); (void)hipMalloc(&b1, K * 128 * 4); (void)hipMalloc(&b2, K * 64 * 4);
  auto fill = [](float* d, size_t n, float sc) { std::vector<float> hb(n); for (auto& v : hb) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * sc; (void)hipMemcpy(d, hb.data(), n * 4, hipMemcpyHostToDevice); };
  fill(eo, (size_t)rows * 64, 1.f); fill(ef, (size_t)rows * K, 0.5f); fill(W, (size_t)K * 16384, 0.1f); fill(b1, K * 128, 0.1f); fill(b2, K * 64, 0.1f);
  const double fl = (double)rows * K * (2 * 64 * 128 * 2);
  printf("rows %d\n", rows);
  float t;
  t = run<0>(eo, ef, W, b1, b2, feat, rows, K); printf("  flags 0 (bare)          %7.1f us %6.1f TF\n", t, fl / t / 1e6);
  t = run<1>(eo, ef, W, b1, b2, feat, rows, K); printf("  flags 1 (+bias loads)   %7.1f us %6.1f TF\n", t, fl / t / 1e6);
  t = run<3>(eo, ef, W, b1, b2, feat, rows, K); printf("  flags 3 (+ef/b2)        %7.1f us %6.1f TF\n", t, fl / t / 1e6);
  t = run<7>(eo, ef, W, b1, b2, feat, rows, K); printf("  flags 7 (+relu = real)  %7.1f us %6.1f TF\n", t, fl / t / 1e6);
  t = run<4>(eo, ef, W, b1, b2, feat, rows, K); printf("  flags 4 (relu only)     %7.1f us %6.1f TF\n", t, fl / t / 1e6);
  }
  return 0;
}
