// Row-wise fused MLP chains on the bf16 matrix cores of gfx950 (v_mfma_f32_32x32x16_bf16), included by
// gn_mlp_mfma.hip (one translation unit, shared launch tables).
//
// Every kernel here exists in two precisions, selected by the template parameter P ("parts"):
//
//   P = 3, T = float  — fp32 results on the bf16 cores ("bf16x6").  Every fp32 operand is x = x1 + x2 + x3 with
//          three bf16 parts (round to nearest; the remainders are exact in fp32): weights once per parameter
//          version (gn_split_bf16_f32, parts = 3), activations on the VALU when a tile is produced.  A product is
//          the six significant part-products, smallest first, accumulated in the fp32 accumulator:
//          w3x1 + w2x2 + w1x3 + w2x1 + w1x2 + w1x1.  As accurate as fp32 accumulation (2e-7 of max|result| at
//          K = 256), six 8-pass MFMAs per k = 16 instead of eight 16-pass fp32 MFMAs.  These are what the
//          *_f32 entry points run when a group carries the `Wx` image.
//   P = 2, T = float  — fp32 results on the fp16 cores ("f16x3", the default of the *_f32 entry points since round 3).
//          Every fp32 operand is x = xh + xl with two fp16 parts (11 + 11 significant bits, the remainder is exact in
//          fp32); a product is the three significant part-products, smallest first: wh.xl + wl.xh + wh.xh, three
//          v_mfma_f32_32x32x16_f16 per k = 16 — half the matrix work and half the splitting work of bf16x6 at the
//          same accuracy (3e-7 of max|result| at K = 256, measured next to an fp32 fma chain's 5e-7:
//          tools/microbench/mfma_rate_f16x3.hip; chain rate 380-390 vs 202-220 TFLOP/s fp32-equivalent).  fp16 has a
//          NARROW exponent: an operand beyond 65504 would become inf.  Every value that is split is therefore also
//          folded into a running maximum (v_max3_f32, 8 instructions per hidden tile), the workgroup votes at the end
//          of the chain, and a workgroup that saw |operand| > 65000 (or whose weight image is flagged: a weight out of
//          range) REPEATS its rows on the bf16x6 path of the same kernel (P = 3 below, full fp32 exponent range) and
//          overwrites what it stored — never a wrong result, and no cost on data within range.  Operands below 2^-14
//          lose their low part (fp16 subnormals): an ABSOLUTE error of 2^-25 per such operand, far below the 1e-5 gate.
//   P = 1, T = __bf16 — the bf16 twins (BASELINE config 4): activations stored in HBM as bf16,
//          weights rounded once to bf16, one MFMA per k = 16, fp32 accumulation, bias / ReLU / softmax in fp32;
//          a layer's fp32 result is rounded to bf16 when it becomes the next layer's operand or is stored.
//
// Orientation and register layout are those of gn_mlp_mfma.hip (Y^T = W X^T: A operand = weights, B operand =
// activations; the 16 accumulator registers of lane (j, h) are 16 features of ITS row j): registers 8*hf .. 8*hf+7
// of a 32-feature tile are exactly the eight k-values lane (j, h) supplies to the k = 16 MFMA of half hf, once
// converted to bf16 — the k-permutation this implies is baked into the weight image by gn_split_bf16_f32.  A whole
// chain therefore runs in registers, hidden tile by hidden tile (one hidden tile live), with no LDS round trip.
//
// Weight image: a "sub-step" = the (32 outputs x 16 k) operand of one MFMA position, P pieces of 64 lanes x 16 B:
// piece p of sub-step s sits at ((s*P + p)*64 + lane) * 16 B.  Images are laid out in CONSUMPTION order (built by
// ops.PackPlan as fp32 tiles, two sub-steps per 32x32 tile, then split), so a kernel walks its image linearly
// through a register ring that runs D sub-steps ahead of the matrix pipe (D = 4 for P = 3: 24 MFMAs; D = 16 for
// P = 1: 16 MFMAs).
#pragma once
#include "gn_mlp_common.hpp"
#include "gn_affinity.hpp"

#ifdef GN_STAMPS
// Diagnostic build only (never the product): per-wave cycle stamps, read back with gn_debug_read_stamps.
__device__ unsigned long long gn_stamp_buf[1 << 17];
#define GN_STAMP(unit, slot)                                                              \
  do {                                                                                    \
    if ((threadIdx.x & 63) == 0 && (unit) < (1 << 13))                                    \
      gn_stamp_buf[(size_t)(unit) * 16 + (slot)] = ((slot) & 8) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); \
  } while (0)
extern "C" int gn_debug_read_stamps(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(gn_stamp_buf), bytes);
}
#else
#define GN_STAMP(unit, slot) do { (void)(unit); } while (0)
#endif

// workgroups per CU the bf16-storage (P = 1) edge / aggregation kernels are compiled for
#ifndef GN_OCC_P1
#define GN_OCC_P1 2
#endif

#include <type_traits>

namespace {

template <int P>
struct Parts {
  bf16x8 p[P];
};

// eight fp32 registers -> four packed bf16 pairs.  Written as ONE vector conversion so that it lowers to four
// two-source v_cvt_pk_bf16_f32; element-by-element casts lower to eight one-source conversions plus four v_perm.
typedef float f32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 cvt_half(const f32x16& v, int hf) {
  f32x8 h;
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) h[jj] = v[8 * hf + jj];
  return __builtin_convertvector(h, bf16x8);
}

// operand part(s) of half `hf` of a 32-feature fp32 tile held in a lane's 16 registers.  P = 2 (two fp16 parts) also
// folds |v| into `ovf`, the wave's running maximum of everything it has split (see the header: range vote).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float kF16Limit = 65000.f;      // an operand beyond this sends the workgroup to the bf16x6 path
// Range state of a wave on the fp16 path: `mask` = lanes that have split a value beyond kF16Limit (wave-uniform, in
// SGPRs), `wf` = the weight image's flag word as loaded (NOT waited for until the vote at the end of the chain: a wait at
// the point of the load is a full L2 round trip in the prologue of every kernel).
struct ovf_t {
  unsigned long long mask;
  int wf;
};
template <int P>
__device__ __forceinline__ void make_parts(const f32x16& v, int hf, Parts<P>& x, ovf_t& ovf) {
#ifdef GN_DIAG_NO_SPLIT      // diagnostic builds only: what the kernels take without the VALU splitting work
  for (int p = 0; p < P; ++p) {
    const f32x4 w = {v[8 * hf], v[8 * hf + 1], v[8 * hf + 2], v[8 * hf + 3]};
    x.p[p] = __builtin_bit_cast(bf16x8, w);
  }
  return;
#endif
  if constexpr (P == 3) {
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      __bf16 a, b, c;
      split3(v[8 * hf + jj], a, b, c);
      x.p[0][jj] = a;
      x.p[1][jj] = b;
      x.p[2][jj] = c;
    }
  }
  if constexpr (P == 2) {
    f16x8 hi, lo;
    // (`ovf` is a wave-uniform lane mask in SCALAR registers: v_cmp + s_or_b64 per half tile.  Kept as a running fp32
    // maximum in a vector register instead, the one serial chain through every split of the kernel cost the node / edge
    // kernels 240 / 500 bytes of scratch per lane.)
    float m[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = v[8 * hf + 2 * j], b = v[8 * hf + 2 * j + 1];
      m[j] = __builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b));
      const f32x2 pr = {a, b};
      const f16x2 hh = __builtin_convertvector(pr, f16x2);                                       // v_cvt_pk_f16_f32 (RNE)
      const f32x2 rem = {a - (float)hh[0], b - (float)hh[1]};                                    // exact
      const f16x2 ll = __builtin_convertvector(rem, f16x2);
      hi[2 * j] = hh[0], hi[2 * j + 1] = hh[1];
      lo[2 * j] = ll[0], lo[2 * j + 1] = ll[1];
    }
    x.p[0] = __builtin_bit_cast(bf16x8, hi);
    x.p[1] = __builtin_bit_cast(bf16x8, lo);
#ifndef GN_NO_OVF
    ovf.mask |= __builtin_amdgcn_ballot_w64(__builtin_fmaxf(__builtin_fmaxf(m[0], m[1]), __builtin_fmaxf(m[2], m[3])) > kF16Limit);
#endif
  }
  if constexpr (P == 1) x.p[0] = cvt_half(v, hf);
}
// P = 1 with the ReLU folded in: convert first (v_cvt_pk_bf16_f32, two values per instruction), then clamp the packed
// bf16 pairs at zero as 16-bit integers (a negative float has its sign bit set, so max(x, 0) on the raw halves is
// ReLU; -0 and negative NaNs become +0): 4 + 4 instructions per half tile instead of 8 v_max_f32 + 4 conversions —
// the P = 1 kernels are bound by VALU issue, not by the matrix pipe.  Written pair by pair: clamping the whole
// 8-vector as i16x8 makes the compiler scalarise the conversion (8 one-source conversions + 4 v_perm per half).
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void make_parts_relu(const f32x16& v, int hf, Parts<1>& x) {
  u32x4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x2 pr = {v[8 * hf + 2 * j], v[8 * hf + 2 * j + 1]};
    const i16x2 zero = {0, 0};
    r[j] = __builtin_bit_cast(unsigned,
                              __builtin_elementwise_max(__builtin_bit_cast(i16x2, __builtin_convertvector(pr, bf16x2)), zero));
  }
  x.p[0] = __builtin_bit_cast(bf16x8, r);
}

template <int P, int NT>
__device__ __forceinline__ void make_parts_tiles(const f32x16 (&v)[NT], Parts<P> (&x)[NT][2], ovf_t& ovf) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    make_parts<P>(v[t], 0, x[t][0], ovf);
    make_parts<P>(v[t], 1, x[t][1], ovf);
  }
}
// MFMAs per sub-step
template <int P>
constexpr int kMfmaPerSub = P == 3 ? 6 : (P == 2 ? 3 : 1);

// ---- range vote of the f16x3 path ---------------------------------------------------------------------------------
// `body(parts, ovf)` runs a workgroup's whole chain with `parts` parts per operand.  P = 2: first on the fp16 cores; if
// any wave of the workgroup split a value beyond the fp16 range (or the weight image is flagged) the workgroup runs
// the same chain again on the bf16x6 path and overwrites its stores.  The vote is workgroup-wide because the waves of
// a workgroup share one weight stream (and its barriers).  Every wave of the workgroup must return from `body`.
template <int P, typename F>
__device__ __forceinline__ void run_with_fallback(F body) {
  ovf_t ovf = {0ull, 0};
  if constexpr (P == 2) {
    body(std::integral_constant<int, 2>{}, ovf);
    if (!__syncthreads_or(ovf.mask != 0ull || ovf.wf != 0)) return;
#ifndef GN_NO_FALLBACK       // (diagnostic builds: the fp16 path alone)
    ovf = {0ull, 0};
    body(std::integral_constant<int, 3>{}, ovf);
#endif
  } else {
    body(std::integral_constant<int, P>{}, ovf);
  }
}
// The flag word behind an fp16 weight image of `substeps` sub-steps (gn_split_f16_f32 sets it when a weight does not
// fit fp16): nonzero sends the workgroup to the bf16x6 path.
__device__ __forceinline__ int image_flag(const void* image, int substeps) {
  return *reinterpret_cast<const int*>(reinterpret_cast<const char*>(image) + (size_t)substeps * 2 * 1024);
}
// the image a group's chain walks with P parts (f16x3: `h`, bf16x6 / twins: `x`)
template <int P>
__device__ __forceinline__ const void* pick_image(const void* x, const void* h) { return P == 2 ? h : x; }

// acc += W[sub-step] . x from the sub-step's P 16-byte operand pieces
template <int P>
__device__ __forceinline__ void mfma_substep(const f32x4 (&w)[P], const Parts<P>& x, f32x16& acc) {
#ifdef GN_DIAG_NO_MFMA       // diagnostic builds only: everything but the matrix instructions
  for (int p = 0; p < P; ++p) acc[p] += w[p][0] * __builtin_bit_cast(f32x4, x.p[p])[0];
  return;
#endif
  if constexpr (P == 3) {
    const bf16x8 w1 = __builtin_bit_cast(bf16x8, w[0]);
    const bf16x8 w2 = __builtin_bit_cast(bf16x8, w[1]);
    const bf16x8 w3 = __builtin_bit_cast(bf16x8, w[2]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3, x.p[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x.p[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x.p[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x.p[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x.p[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x.p[0], acc, 0, 0, 0);
  } else if constexpr (P == 2) {
    const f16x8 wh = __builtin_bit_cast(f16x8, w[0]), wl = __builtin_bit_cast(f16x8, w[1]);
    const f16x8 xh = __builtin_bit_cast(f16x8, x.p[0]), xl = __builtin_bit_cast(f16x8, x.p[1]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[0]), x.p[0], acc, 0, 0, 0);
  }
}

// ---- a wave's PRIVATE weight stream: a register ring D sub-steps ahead of the matrix pipe, refilled straight from
// L2.  Used where the waves of a workgroup walk different parts of an image (the typed aggregation with the types
// dealt over the waves of a row block); everything else shares one stream per workgroup through LDS (WStream).
// A stream is a sequence of segments (e.g. one per edge type) of `seg` sub-steps at `cur`, followed by `nxt`.
template <int P>
struct XStream {
  static constexpr int D = P == 1 ? 8 : 4;    // ring depth in sub-steps (P = 2 at 6 / 8: launch time unchanged to 0.1 us)
  static constexpr int CH = 1;                // (positions are given in sub-steps: c0 is ignored)
  f32x4 q[D][P];
  const f32x4* cur;     // this lane's pointer at sub-step 0 of the current segment
  const f32x4* nxt;     // ... of the segment after it
  int seg;
  __device__ __forceinline__ void begin(const f32x4* first) {
#pragma unroll
    for (int u = 0; u < D; ++u)
#pragma unroll
      for (int p = 0; p < P; ++p) q[u][p] = first[(u * P + p) * 64];
  }
  __device__ __forceinline__ void segment(const f32x4* cur_, const f32x4* nxt_, int seg_) { cur = cur_, nxt = nxt_, seg = seg_; }
  template <bool FENCE = true>
  __device__ __forceinline__ void step(int, int s, const Parts<P>& x, f32x16& acc) {
    const int u = s % D;
    mfma_substep<P>(q[u], x, acc);
    const f32x4* src = s + D < seg ? cur + (size_t)(s + D) * P * 64 : nxt + (size_t)(s + D - seg) * P * 64;
#ifdef GN_DIAG_HALF_W        // diagnostic builds only (results are wrong): half of the ring's bytes from L2
    q[u][0] = src[0];
#pragma unroll
    for (int p = 1; p < P; ++p) q[u][p] = q[u][0];
#else
#pragma unroll
    for (int p = 0; p < P; ++p) q[u][p] = src[p * 64];
#endif
    // hipcc otherwise sinks the run-ahead loads down to their use and collapses the ring
    if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
  }
};

// ---- the weight stream of a workgroup, shared through LDS -----------------------------------------------------
// Every wave of a workgroup walks the SAME weight image (the four 32-row blocks of a workgroup belong to one
// module).  Read per wave straight from L2, the three 16-byte pieces per k = 16 sub-step are 64 B/clk per CU at
// full matrix rate — exactly the bandwidth of the CU's vector-memory path, which then, not the matrix pipe, paces
// the kernel (measured: lone waves at 50-55 % of the MFMA-bound time, 68 % with the loads removed).  So the
// workgroup fetches the image ONCE: it is cut into chunks of CH sub-steps; each of the 4 waves loads a quarter of
// every chunk into registers LOOK chunks ahead (ordinary global loads: counted vmcnt waits, nothing drains), writes
// it into a ring of R = 3 chunks in LDS two chunks before its use, and all waves read their MFMA A operands from
// there with ds_read_b128 (conflict-free: lane-linear 1-KiB pieces), one sub-step ahead.  One raw barrier per chunk:
//   boundary(c):  s_waitcnt lgkmcnt(0); s_barrier      -- every wave has finished chunk c-1, chunk c+1 is visible
//                 chunk c+2  : staging registers -> LDS slot (c+2) % 3   (the slot chunk c-1 occupied)
//                 chunk c+2+LOOK : global -> staging registers
// Requirements: all 4 waves of the workgroup call begin / step / skip with identical arguments (no early exits);
// segment lengths are multiples of CH * LOOK sub-steps wherever the chunk index is not a compile-time constant.
#ifndef GN_LOOK2
#define GN_LOOK2 2
#endif
#ifndef GN_QD2
#define GN_QD2 2
#endif
template <int P, int LOOK_ = (P == 1 ? 4 : (P == 2 ? GN_LOOK2 : 2)), int QD_ = (P == 1 ? 4 : (P == 2 ? GN_QD2 : 2))>
struct WStream {
  static constexpr int CH = P == 1 ? 8 : 4;      // sub-steps per chunk
  static constexpr int LOOK = LOOK_;             // chunks between a piece's global load and its LDS write (the kernels
                                                 // with two row blocks per wave spend twice as long per chunk: 2)
  static constexpr int R = 3;                    // chunks in the LDS ring
  static constexpr int PIECES = CH * P;          // 1-KiB pieces (64 lanes x 16 B) per chunk
  static constexpr int PW = PIECES / 4;          // pieces each of the 4 waves stages per chunk
  static constexpr int kChunkF4 = PIECES * 64;   // f32x4 elements of a chunk
  static constexpr int kRingF4 = R * kChunkF4;   // ... of the ring (36 KiB / 24 KiB)
  static constexpr int QD = QD_;                 // operand registers: sub-steps read ahead of their MFMA + 1.  A P = 3
                                                 // sub-step is 192 cycles of matrix work — one ahead covers the LDS
                                                 // latency; a P = 1 sub-step is 32 cycles, so three ahead
  // Running state instead of index arithmetic per access (a P = 1 sub-step is ONE 32-cycle MFMA: a handful of
  // address instructions per operand read would cost as much as the matrix work): the chunks are consumed strictly
  // in order, so the ring slot of the current chunk, the wave's write position and its global read position advance
  // by constants at every boundary; sub-step offsets inside a chunk are immediates of the ds_read.
  const f32x4* rd_cur;    // LDS: current chunk, this lane's view
  const f32x4* rd_next;   // LDS: next chunk
  f32x4* wr;              // LDS: where this wave's pieces of chunk (current + 2) go
  f32x4* ring0;           // LDS: slot 0, this lane's view
  const f32x4* ld;        // global: this wave's pieces of the chunk loaded next
  const f32x4* ld_last;   // ... of the image's last chunk (loads beyond it re-read that one)
  int slot;               // ring slot of the current chunk
  f32x4 st[LOOK][PW];
  f32x4 q[QD][P];

  // Two barriers.  begin(): the first chunks were just written, so the wave drains its LDS queue before it signals.
  // boundary(): NO drain — the operand reads already in flight (QD-1 sub-steps ahead) stay in flight across the
  // barrier.  That is safe because LDS operations of a wave complete in order: (1) the pieces this wave wrote at the
  // previous boundary are older than operand reads it has since waited for, so they have landed before it signals;
  // (2) the slot overwritten after the barrier held chunk c-1, whose last operand reads every wave consumed (waited
  // for) before its last MFMA of that chunk, i.e. before it arrived here.
#ifdef GN_DIAG_NO_BARRIER    // diagnostic builds only (results are wrong)
  __device__ __forceinline__ static void barrier_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
  __device__ __forceinline__ static void barrier() { asm volatile("" ::: "memory"); }
#else
  __device__ __forceinline__ static void barrier_drain() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
  __device__ __forceinline__ static void barrier() { asm volatile("s_barrier" ::: "memory"); }
#endif
  __device__ __forceinline__ void load_stage(int j) {
#pragma unroll
    for (int i = 0; i < PW; ++i) st[j][i] = ld[i * 64];
    ld = ld == ld_last ? ld : ld + kChunkF4;
  }
  // operands of sub-step `sub` (0 .. 2*CH-1: sub >= CH addresses the next chunk) into register set `set`
  __device__ __forceinline__ void read_ops(int sub, int set) {
    const f32x4* p = (sub < CH ? rd_cur : rd_next) + (sub % CH) * P * 64;
#pragma unroll
    for (int pp = 0; pp < P; ++pp) q[set][pp] = p[pp * 64];
  }
  // image: global address of the stream's first chunk; n_chunks: chunks from there to the end of the image
  __device__ __forceinline__ void begin(const void* image, f32x4* lds_ring, int lane, int wave, int n_chunks) {
    ring0 = lds_ring + lane;
    ld = reinterpret_cast<const f32x4*>(image) + lane + wave * PW * 64;
    ld_last = ld + (size_t)(n_chunks - 1) * kChunkF4;
    f32x4* w0 = ring0 + wave * PW * 64;
    // chunks 0 and 1 straight into slots 0 and 1, chunks 2 .. 2+LOOK-1 into the staging registers
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int i = 0; i < PW; ++i) w0[c * kChunkF4 + i * 64] = ld[i * 64];
      ld = ld == ld_last ? ld : ld + kChunkF4;
    }
#pragma unroll
    for (int j = 0; j < LOOK; ++j) load_stage(j);
    slot = R - 1;                       // (the first boundary advances to slot 0)
    rd_cur = ring0 + (R - 1) * kChunkF4;
    rd_next = ring0;
    wr = w0 + 2 * kChunkF4;             // chunk 2 -> slot 2
    barrier_drain();
    // operands of the first QD-1 sub-steps: "next chunk" is chunk 0 until the first boundary has run
#pragma unroll
    for (int j = 0; j + 1 < QD; ++j) read_ops(CH + j, j);
  }
  // chunk boundary; `cl` = index of the chunk that starts here, modulo LOOK (compile-time)
  __device__ __forceinline__ void boundary(int cl) {
    barrier();
    slot = slot + 1 == R ? 0 : slot + 1;
    rd_cur = rd_next;
    rd_next = ring0 + (slot + 1 == R ? 0 : slot + 1) * kChunkF4;
#pragma unroll
    for (int i = 0; i < PW; ++i) wr[i * 64] = st[cl % LOOK][i];      // chunk + 2 -> the slot chunk - 1 occupied
    wr = slot == 0 ? wr - 2 * kChunkF4 : wr + kChunkF4;               // slots of chunk+2: 2, 0, 1, 2, ... (R = 3)
    load_stage(cl % LOOK);                                            // chunk + 2 + LOOK
  }
  // acc += W[sub-step s of the current segment] . x.  `s` is a compile-time constant at every call site (segments
  // start at chunk boundaries and are multiples of CH * LOOK sub-steps long where they repeat in a runtime loop);
  // the first argument is unused (XStream has the same interface).  FENCE closes the scheduling region behind the step.
  template <bool FENCE = true>
  __device__ __forceinline__ void step(int, int s, const Parts<P>& x, f32x16& acc) {
    if (s % CH == 0) boundary(s / CH);
    // operands of the sub-step QD-1 ahead (LDS latency under the MFMAs); chunk c+1 is visible throughout chunk c
    read_ops(s % CH + QD - 1, (s + QD - 1) % QD);
    mfma_substep<P>(q[s % QD], x, acc);
    if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
  }
  // the same sub-step against RB row blocks: ONE operand read feeds RB MFMAs (x_of(b), acc_of(b): block b's operand
  // and accumulator)
  template <int RB, bool FENCE = true, typename XF, typename AF>
  __device__ __forceinline__ void step_rb(int s, XF x_of, AF acc_of) {
    if (s % CH == 0) boundary(s / CH);
    read_ops(s % CH + QD - 1, (s + QD - 1) % QD);
#pragma unroll
    for (int b = 0; b < RB; ++b) mfma_substep<P>(q[s % QD], x_of(b), acc_of(b));
    if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
  }
  // pass over sub-steps [s, s + n) without using them: the wave still takes part in every chunk boundary among them
  __device__ __forceinline__ void skip(int, int s, int n) {
#pragma unroll
    for (int p = s; p < s + n; ++p)
      if (p % CH == 0) boundary(p / CH);
    const int basec = (s + n - 1) / CH;          // the chunk the last boundary made current
#pragma unroll
    for (int j = 0; j + 1 < QD; ++j) read_ops(s + n + j - basec * CH, (s + n + j) % QD);
  }
};

// ---- one layer pair, hidden tile by hidden tile, software-pipelined ------------------------------------------------
//   out[o] += W1(o, :) post(W0 x + b0)      x: IT input tiles (as parts), HT hidden tiles, OT output tiles
// A_t = the 2*IT sub-steps that produce hidden tile t, V_t = its VALU work (post: ReLU / scale / optional store,
// then the bf16 part(s)), B_t = the 2*OT sub-steps that consume it.  B_t needs V_t needs A_t, so executed in that
// order a lone wave leaves the matrix pipe idle during every V_t (~100 VALU instructions for three parts).  The
// pipeline issues A_{t+1} between A_t and B_t and interleaves V_t with its MFMAs (sched_group_barrier: one MFMA,
// then a few VALU), so the splitting runs in the shadow of the matrix pipe.  The weight image is laid out in this
// order:  A0 A1 B0 A2 B1 ... A(HT-1) B(HT-2) B(HT-1).  Bias tiles ride two hidden tiles ahead of their use: a
// load that is waited for right after it is issued would also wait for every staging load of the weight stream
// (vmcnt counts in order).
//   c0, s0: the segment's first chunk (runtime) and the sub-step of A0 inside it (compile-time);
//   hid0: bias tile 0 (loaded early by the caller).
//   PACKED_RELU (P = 1 only): the hidden layer's ReLU is applied on the packed bf16 operands (make_parts_relu);
//   `post` then must not apply it.
template <int P, int IT, int OT, int HT, bool PACKED_RELU = false, typename Stream, typename PostFn>
__device__ __forceinline__ void layer_pair(Stream& ws, int c0, int s0, const Parts<P> (&xi)[IT][2],
                                           const f32x16& hid0, const float* __restrict__ b0, int h,
                                           f32x16 (&out)[OT], ovf_t& ovf, PostFn post) {
  constexpr int NA = 2 * IT, NB = 2 * OT;
  constexpr int kMfma = NA * kMfmaPerSub<P>;                   // MFMAs of one A phase
#ifndef GN_KVALU
#define GN_KVALU 96
#endif
#ifndef GN_KVALU2
#define GN_KVALU2 84
#endif
  // VALU slots per MFMA (V_t: ReLU / scale, the split into parts, the range maximum)
  constexpr int kValu = ((P == 3 ? GN_KVALU : (P == 2 ? GN_KVALU2 : 40)) + kMfma - 1) / kMfma;
  int pos = s0;
  f32x16 hidn = hid0;
  f32x16 bias_n;
  if (HT > 1) bias_n = load_bias_tile(b0 + 32, h);
#pragma unroll
  for (int u = 0; u < NA; ++u) {
    ws.step(c0, pos, xi[u >> 1][u & 1], hidn);
    ++pos;
  }
#pragma unroll
  for (int t = 0; t < HT; ++t) {
    f32x16 cur = hidn;
    if (t + 1 < HT) {
      hidn = bias_n;
      if (t + 2 < HT) bias_n = load_bias_tile(b0 + 32 * (t + 2), h);
    }
    post(t, cur);
    Parts<P> xh[2];
    if constexpr (PACKED_RELU) {
      make_parts_relu(cur, 0, xh[0]);
      make_parts_relu(cur, 1, xh[1]);
    } else {
      make_parts<P>(cur, 0, xh[0], ovf);
      make_parts<P>(cur, 1, xh[1], ovf);
    }
    if (t + 1 < HT) {
#pragma unroll
      for (int u = 0; u < NA; ++u) {
        ws.template step<false>(c0, pos, xi[u >> 1][u & 1], hidn);
        ++pos;
      }
#pragma unroll
      for (int i = 0; i < kMfma; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, kValu, 0);      // VALU of V_t in its shadow
        if ((i % kMfmaPerSub<P>) == 0)            // the sub-step's operand reads (LDS) / ring refills (global)
          __builtin_amdgcn_sched_group_barrier(Stream::CH == 1 ? 0x020 : 0x100, P, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      ws.step(c0, pos, xh[u & 1], out[u >> 1]);
      ++pos;
    }
  }
}

template <typename T>
__device__ __forceinline__ void store_tile(T* __restrict__ p, const f32x16& a) {   // p: row base + 32*tile + 4*h
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = {a[4 * q + 0], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
    st4(p + 8 * q, v);
  }
}

// ---- node stage: x' = MLP(64->256->64)(x), pq = x' Wpq^T + bpq, and (pairwise module) A = WA x + bA ------------
// A3 first half (MS_HGNN_batch.py:125,131-134,358,362-365) and the per-node first layer of the typed aggregation
// MLP of the pairwise graph (MS_HGNN_batch.py:264-265; see gn_node_linear_f32) in ONE launch: both read the same
// node rows.  A workgroup = 4 consecutive 32-row blocks of ONE group, one wave each, sharing one weight stream:
// "chain" workgroups run the whole chain (72 sub-steps), "A" workgroups 8 output tiles of WA (32 sub-steps).
// Long workgroups first.
struct NodeTable {
  gn_node_group_t g[GN_MAX_GROUPS];
  int a_first[GN_MAX_GROUPS + 1];   // prefix of A workgroups per group, relative to chain_wgs
  int n, rows, wgs_per_group, chain_wgs;
  XcdSections xs;                   // sections: every group's chain, then every (group, output-tile chunk) of WA
  // optional tail of the launch: blocks node_grid .. node_grid + aff_scenes - 1 each build the affinity / incidences of
  // one scene (gn_affinity.hpp) — independent of the node stage, dispatched behind it
  int node_grid, aff_scenes, aff_N, aff_D;
  const void* aff_f;
  float* aff_corr;
  ScaleList aff_sl;
  gn_block_extras_t aff_ex;
};
// output tiles of WA per "A" workgroup (a multiple of 4).  4: 264 workgroups of 16 sub-steps at B = 512 instead of 132 of
// 32 — the A workgroups were the node stage's last to finish (same-session A/B: launch 22.1 -> 19.9 us).
#ifndef GN_KATILES
#define GN_KATILES 4
#endif
constexpr int kATiles = GN_KATILES;

template <int P, typename T>
__device__ __forceinline__ void node_stage_body(const NodeTable& Tb, f32x4* wring, ovf_t& ovf) {
  using WS = WStream<P>;
  const int wave = wave_id();
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int wg = gn_uniform(gn_xcd_logical(Tb.xs, blockIdx.x));
  if (wg < 0) return;
  WS ws;
  if (wg < Tb.chain_wgs) {
    const int gi = gn_uniform(wg / Tb.wgs_per_group);
    const gn_node_group_t G = Tb.g[gi];
    const RowBlock rb = row_block(Tb.rows, (wg - gi * Tb.wgs_per_group) * 4 + wave);
    const int unit = wg * 4 + wave;
    GN_STAMP(unit, 0);
    GN_STAMP(unit, 8);
    const float* b0 = G.bias;
    const float* b1 = G.bias + 256;
    const float* bpq = G.bias + 320;
    const void* img = pick_image<P>(G.Wx, G.Wh);
    if constexpr (P == 2) ovf.wf |= image_flag(img, 72);
    ws.begin(img, wring, lane, wave, 72 / WS::CH);
    f32x16 in[2];
    load_rows<2>(reinterpret_cast<const T*>(G.x), GN_FEAT, rb.row_ld, h, in);
    const f32x16 hid0 = load_bias_tile(b0, h);
    f32x16 xp[2], pq[2];
    xp[0] = load_bias_tile(b1, h);
    xp[1] = load_bias_tile(b1 + 32, h);
    Parts<P> xi[2][2];
    make_parts_tiles<P, 2>(in, xi, ovf);
    GN_STAMP(unit, 1);
    // image: the 64->256->64 pair in pipeline order (A_t = [W0(t,in0), W0(t,in1)], B_t = [W1(0,t), W1(1,t)]), then
    // [Wpq(0,in0), Wpq(0,in1), Wpq(1,in0), Wpq(1,in1)]
    layer_pair<P, 2, 2, 8, P == 1>(ws, 0, 0, xi, hid0, b0, h, xp, ovf, [&](int t, f32x16& hid) {
      if constexpr (P != 1) {
        relu16(hid);
        if (G.hid_out != nullptr && rb.live) store_tile(G.hid_out + (size_t)rb.row * 256 + 32 * t + 4 * h, hid);
      }
    });
    GN_STAMP(unit, 2);
    pq[0] = load_bias_tile(bpq, h);         // (requested here: held across the layer pair they cost 32 registers)
    pq[1] = load_bias_tile(bpq + 32, h);
    store_rows<2>(reinterpret_cast<T*>(G.xp), GN_FEAT, rb.row, h, rb.live, xp);
    Parts<P> xq[2][2];
    make_parts_tiles<P, 2>(xp, xq, ovf);
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int i = 0; i < 4; ++i) ws.step(0, 64 + 4 * o + i, xq[i >> 1][i & 1], pq[o]);
    GN_STAMP(unit, 3);
    store_rows<2>(reinterpret_cast<T*>(G.pq), GN_FEAT, rb.row, h, rb.live, pq);
    GN_STAMP(unit, 4);
    GN_STAMP(unit, 9);
    return;
  }
  // ---- A workgroup: (output-tile chunk c, 4 row blocks) ----
  const int v = wg - Tb.chain_wgs;
  int gi = 0;
  while (gi + 1 < Tb.n && v >= Tb.a_first[gi + 1]) ++gi;
  gi = gn_uniform(gi);
  const gn_node_group_t G = Tb.g[gi];
  const int OTA = 4 * G.KA;                              // output tiles of WA (128 per type)
  const int local = v - Tb.a_first[gi];
  const int c = local / Tb.wgs_per_group, quad = local - c * Tb.wgs_per_group;
  const RowBlock rb = row_block(Tb.rows, quad * 4 + wave);
  const int unit = wg * 4 + wave;
  GN_STAMP(unit, 0);
  GN_STAMP(unit, 8);
  const int o0 = c * kATiles;
  const int nt = min(kATiles, OTA - o0);                 // a multiple of 4
  // 4 sub-steps per output tile; the stream of this workgroup starts at tile o0 of the image
  const void* imgA = pick_image<P>(G.WAx, G.WAh);
  if constexpr (P == 2) ovf.wf |= image_flag(imgA, OTA * 4);
  ws.begin(reinterpret_cast<const f32x4*>(imgA) + (size_t)o0 * 4 * P * 64, wring, lane, wave, (OTA - o0) * 4 / WS::CH);
  f32x16 in[2];
  load_rows<2>(reinterpret_cast<const T*>(G.x), GN_FEAT, rb.row_ld, h, in);
  Parts<P> xi[2][2];
  make_parts_tiles<P, 2>(in, xi, ovf);
  GN_STAMP(unit, 1);
  const size_t ldA = (size_t)OTA * 32;
  T* arow = reinterpret_cast<T*>(G.A) + (size_t)rb.row * ldA + 4 * h;
  f32x16 bn = load_bias_tile(G.bA + 32 * o0, h);          // bias rides one tile ahead (see layer_pair)
#pragma unroll 1
  for (int o4 = 0; o4 < nt; o4 += 4) {
#pragma unroll
    for (int oo = 0; oo < 4; ++oo) {
      const int o = o0 + o4 + oo;
      f32x16 acc = bn;
      bn = load_bias_tile(G.bA + 32 * min(o + 1, o0 + nt - 1), h);
#pragma unroll
      for (int i = 0; i < 4; ++i) ws.step(o4 * 4 / WS::CH, 4 * oo + i, xi[i >> 1][i & 1], acc);   // 16 sub-steps per pass
      if (rb.live) store_tile(arow + 32 * o, acc);
    }
  }
  GN_STAMP(unit, 4);
  GN_STAMP(unit, 9);
}
// largest weight ring of the precisions a kernel instantiation may run (P = 2 falls back to 3)
template <int P>
constexpr int kRingF4For = WStream<P == 2 ? 3 : P>::kRingF4;

template <int P, typename T>
__global__ __launch_bounds__(256, 2) void node_stage_kernel(NodeTable Tb) {
  __shared__ f32x4 wring[kRingF4For<P>];
  if (Tb.aff_scenes > 0 && (int)blockIdx.x >= Tb.node_grid) {       // (block-uniform)
    extern __shared__ __align__(16) float aff_lds[];
    affinity_topk_body<T>(reinterpret_cast<const T*>(Tb.aff_f), Tb.aff_corr, Tb.aff_sl, Tb.aff_N, Tb.aff_D, Tb.aff_ex,
                          (int)blockIdx.x - Tb.node_grid, aff_lds);
    return;
  }
  run_with_fallback<P>([&](auto pc, ovf_t& ovf) { node_stage_body<decltype(pc)::value, T>(Tb, wring, ovf); });
}

// ---- A4: z = MLP(64->128->64); [dist|fac] heads; gumbel softmax; sigmoid -------------------------------------
// Image (80 sub-steps), hidden-tile-major: per hidden tile t of init_MLP the tiles [Wi0(t,in0), Wi0(t,in1),
// Wi1(0,t), Wi1(1,t)], then per hidden tile t of [Wd0] the tiles [Wd0(t,in0), Wd0(t,in1), Wd1(0,t)].
template <int P, typename T>
__device__ __forceinline__ void edge_x_body(const GroupTable<gn_edge_group_t>& Tb, float tau, unsigned long long seed,
                                            const unsigned long long* __restrict__ offset_dev, int pool_bytes,
                                            f32x4* wring, ovf_t& ovf) {
  using WS = WStream<P>;
  extern __shared__ __align__(16) unsigned char pool_dyn[];      // staged x' / pq rows of the pairwise pooling
  const int lwg = gn_uniform(gn_xcd_logical(Tb.xs, blockIdx.x));
  if (lwg < 0) return;
  const int gi = find_group(Tb, lwg);
  const gn_edge_group_t G = Tb.g[gi];
  const int rows = G.rows, K = G.K;
  const int blk = (lwg - Tb.first_wg[gi]) * 4 + wave_id();
  const RowBlock rb = row_block(rows, blk);      // (a wave past the group's rows works on a clamped row, stores nothing)
  const int lane = rb.lane, h = rb.h;
  const int unit = lwg * 4 + wave_id();
  GN_STAMP(unit, 0);
  GN_STAMP(unit, 8);
  f32x16 in[2], z[2], lg;
  if (G.edges != nullptr) {
    load_rows<2>(reinterpret_cast<const T*>(G.edges), GN_FEAT, rb.row_ld, h, in);
  } else {
    // fused node -> edge pooling; unordered pairs: from the scenes' node rows staged in LDS when they fit
    const int nodes_max = G.pool_H == nullptr && G.sym_N > 0 ? pool_stage_nodes(128, gn_pair_count(G.pool_N), G.pool_N) : 0;
    if (nodes_max > 0 && PoolStage<T>::bytes(nodes_max) <= (size_t)pool_bytes) {       // (block-uniform)
      T* s_xp = reinterpret_cast<T*>(pool_dyn);
      T* s_pq = s_xp + (size_t)nodes_max * PoolStage<T>::kPitch;
      const int r0 = (lwg - Tb.first_wg[gi]) * 128;
      const int node0 = pool_stage_fill<T>(G, r0, min(rows - 1, r0 + 127), s_xp, s_pq);
      __syncthreads();
      pooled_rows_staged<T>(G, rb.row_ld, h, s_xp, s_pq, node0, in);
    } else {
      pooled_rows<T>(G, rb.row_ld, h, in);
    }
  }
  // ordered edge rows whose uniforms this row consumes: itself, or — symmetric pairwise form — the two ordered
  // edges (i,j) and (j,i) of its unordered pair
  long long o1 = rb.row_ld, o2 = rb.row_ld;
  bool diag = true;
  if (G.sym_N > 0) {
    const int N = G.sym_N, Pn = gn_pair_count(N);
    const int b = rb.row_ld / Pn, p = rb.row_ld - b * Pn;
    int i, j;
    gn_pair_decode(p, N, i, j);
    o1 = (long long)b * N * N + i * N + j;
    o2 = (long long)b * N * N + j * N + i;
    diag = i == j;
  }
  float u1[8], u2[8];
  const unsigned long long pbase = G.philox_offset + (offset_dev ? *offset_dev : 0ull);
  const float* bi0 = G.bias;
  const float* bi1 = G.bias + 128;
  const float* bd0 = G.bias + 192;
  const float* bd1 = G.bias + 448;
  WS ws;
  const void* img = pick_image<P>(G.Wx, G.Wh);
  if constexpr (P == 2) ovf.wf |= image_flag(img, 80);
  ws.begin(img, wring, lane, wave_id(), 80 / WS::CH);
  const f32x16 hidA0 = load_bias_tile(bi0, h);
  z[0] = load_bias_tile(bi1, h);
  z[1] = load_bias_tile(bi1 + 32, h);
  Parts<P> xi[2][2];
  make_parts_tiles<P, 2>(in, xi, ovf);
  GN_STAMP(unit, 1);
  // ---- pair A: 64 -> 128 -> 64, 4 hidden tiles x (4 + 4) sub-steps, pipeline order ----
  layer_pair<P, 2, 2, 4, P == 1>(ws, 0, 0, xi, hidA0, bi0, h, z, ovf, [&](int t, f32x16& hid) {
    if constexpr (P != 1) {
      relu16(hid);
      if (G.keep_z1 != nullptr && rb.live) store_tile(G.keep_z1 + (size_t)rb.row * 128 + 32 * t + 4 * h, hid);
    }
  });
  GN_STAMP(unit, 2);
  if (G.keep_z != nullptr) store_rows<2>(G.keep_z, GN_FEAT, rb.row, h, rb.live, z);
  // (the bias tiles of pair B are requested here, not at the top: held across pair A they cost 32 registers — and scratch)
  const f32x16 hidB0 = load_bias_tile(bd0, h);
  f32x16 lgv[1];
  lgv[0] = load_bias_tile(bd1, h);
  make_parts_tiles<P, 2>(z, xi, ovf);
  // ---- pair B: 64 -> 256 -> (logits | factor), 8 hidden tiles x (4 + 2) sub-steps, pipeline order ----
  layer_pair<P, 2, 1, 8, P == 1>(ws, 0, 32, xi, hidB0, bd0, h, lgv, ovf, [&](int t, f32x16& hid) {
    if constexpr (P != 1) {
      relu16(hid);
      if (G.keep_dh1 != nullptr && rb.live) store_tile(G.keep_dh1 + (size_t)rb.row * 256 + 32 * t + 4 * h, hid);
    }
  });
  lg = lgv[0];
  GN_STAMP(unit, 3);
  // uniforms: read from U or generated from the Philox stream — after the chains (nothing is in flight any more)
  fetch_uniforms(G.U, pbase, seed, o1, K, h, u1);
  if (G.sym_N > 0) fetch_uniforms(G.U, pbase, seed, o2, K, h, u2);
  if (G.keep_lgf != nullptr && rb.live) store_tile(G.keep_lgf + (size_t)rb.row * 32 + 4 * h, lg);

  // Epilogue.  Features 0..K-1 of `lg` are the logits of this lane's row, feature K the factor pre-activation; a
  // row's features are split over its two lanes (j, h=0) and (j, h=1).
  float facv = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r)
    if (feat_of(r, h) == K) facv = lg[r];
  facv += __shfl_xor(facv, 32, GN_WAVE);   // exactly one of the two lanes holds it, the other has 0
  const float sig = 1.f / (1.f + expf(-facv));
  float d1[8], d2[8];
  gumbel_softmax_row<P == 1>(lg, u1, K, tau, h, d1);
  if (G.sym_N > 0) gumbel_softmax_row<P == 1>(lg, u2, K, tau, h, d2);
  if (rb.live) {
    float* frow = G.edge_feat + (size_t)rb.row * K;
    T* dist = reinterpret_cast<T*>(G.dist);
    if (G.sym_N == 0) {
      T* drow = dist + (size_t)rb.row * K;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int f = feat_of(r, h);
        if (f < K) {
          st1(drow + f, d1[r]);
          frow[f] = sig * d1[r];
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int f = feat_of(r, h);
        if (f < K) {
          if (dist != nullptr) {
            st1(dist + (size_t)o1 * K + f, d1[r]);
            if (!diag) st1(dist + (size_t)o2 * K + f, d2[r]);
          }
          // both ordered edges meet the same typed MLP output downstream; the self-loop has weight 2
          frow[f] = diag ? 2.f * (sig * d1[r]) : sig * d1[r] + sig * d2[r];
        }
      }
    }
  }
  GN_STAMP(unit, 4);
  GN_STAMP(unit, 9);
}
template <int P, typename T>
__global__ __launch_bounds__(256, P == 1 ? GN_OCC_P1 : 2) void edge_x_kernel(GroupTable<gn_edge_group_t> Tb, float tau,
                                                        unsigned long long seed,
                                                        const unsigned long long* __restrict__ offset_dev, int pool_bytes) {
  __shared__ f32x4 wring[kRingF4For<P>];
  run_with_fallback<P>([&](auto pc, ovf_t& ovf) {
    edge_x_body<decltype(pc)::value, T>(Tb, tau, seed, offset_dev, pool_bytes, wring, ovf);
  });
}

// ---- A5 typed MLP on the bf16 cores: feat = sum_k ef[:,k] * (W2k relu(W1k eo + b1k) + b2k) -----------------------
// Same work shapes as agg_mlp_kernel (wpr waves share a row block, partial sums meet in LDS).  Forms:
//   two-layer  (W12x): per type and hidden tile o the tiles [W1k(o,in0), W1k(o,in1), W2k(0,o), W2k(1,o)] (32
//              sub-steps per type); input rows from eo, or gathered on the fly from ori (H / pairwise);
//   pair form  (A, W2x; P = 3 only): layer 1 was applied per node (node stage), per type and hidden tile t the tiles
//              [W2k(0,t), W2k(1,t)] (16 sub-steps per type); the two nodes' pre-activations are read from LDS
//              (staged: the workgroup's scenes fit) or straight from HBM/L2.
// b2k enters as one fp32 MFMA per output tile: lane (i, h=0) carries b2k[32o + i], paired with B = ef_k on k-index 0
__device__ __forceinline__ void add_b2(const float* __restrict__ b2k, float efk, int lane, int h, f32x16 (&out)[2]) {
  const float f0 = h == 0 ? b2k[lane & 31] : 0.f;
  const float f1 = h == 0 ? b2k[32 + (lane & 31)] : 0.f;
  const float efb = h == 0 ? efk : 0.f;
  out[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f0, efb, out[0], 0, 0, 0);
  out[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f1, efb, out[1], 0, 0, 0);
}

// the same with the type's b2 values and ef already in registers (requested one type ahead: loaded where they are used,
// the ef value and then the two b2 values each cost the wave a full L2 round trip per type — vmcnt counts in order)
__device__ __forceinline__ void add_b2_regs(float f0, float f1, float efk, int h, f32x16 (&out)[2]) {
  const float efb = h == 0 ? efk : 0.f;
  out[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(h == 0 ? f0 : 0.f, efb, out[0], 0, 0, 0);
  out[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(h == 0 ? f1 : 0.f, efb, out[1], 0, 0, 0);
}
constexpr int kAggPartBytes = 4 * 32 * (64 + 8) * 4;


// A wave's private walk over L sub-steps whose addresses `at(s)` gives (s a compile-time constant after unrolling):
// a ring of D sub-steps refilled straight from L2.
template <int P, int L, int D_ = (P == 1 ? 8 : 4)>
struct PStream {
  static constexpr int D = D_;
  f32x4 q[D][P];
  template <typename AT>
  __device__ __forceinline__ void begin(AT at) {
#pragma unroll
    for (int u = 0; u < D; ++u)
      if (u < L) {
        const f32x4* src = at(u);
#pragma unroll
        for (int p = 0; p < P; ++p) q[u][p] = src[p * 64];
      }
  }
  template <typename AT>
  __device__ __forceinline__ void step(int s, AT at, const Parts<P>& x, f32x16& acc) {
    mfma_substep<P>(q[s % D], x, acc);
    if (s + D < L) {
      const f32x4* src = at(s + D);
#pragma unroll
      for (int p = 0; p < P; ++p) q[s % D][p] = src[p * 64];
    }
    __builtin_amdgcn_sched_barrier(0);     // (keeps the run-ahead loads where they are issued)
  }
};


// operand exchange through LDS: the bf16 part(s) of one input tile, lane-linear 16-byte pieces (conflict-free)
template <int P>
__device__ __forceinline__ void put_parts(f32x4* lds, int tile, int lane, const f32x16& v, ovf_t& ovf) {
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    Parts<P> x;
    make_parts<P>(v, hf, x, ovf);
#pragma unroll
    for (int p = 0; p < P; ++p) lds[((tile * 2 + hf) * P + p) * 64 + lane] = __builtin_bit_cast(f32x4, x.p[p]);
  }
}
template <int P, int IT>
__device__ __forceinline__ void get_parts(const f32x4* lds, int lane, Parts<P> (&xi)[IT][2]) {
#pragma unroll
  for (int t = 0; t < IT; ++t)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int p = 0; p < P; ++p) xi[t][hf].p[p] = __builtin_bit_cast(bf16x8, lds[((t * 2 + hf) * P + p) * 64 + lane]);
}
// partial output tiles of the 4 waves: [wave][quad of 4 registers][lane] as 16-byte pieces
template <int OT>
__device__ __forceinline__ void put_partial(f32x4* lds, int wave, int lane, const f32x16 (&out)[OT]) {
#pragma unroll
  for (int q = 0; q < 4 * OT; ++q) {
    const f32x4 v = {out[q >> 2][4 * (q & 3) + 0], out[q >> 2][4 * (q & 3) + 1], out[q >> 2][4 * (q & 3) + 2],
                     out[q >> 2][4 * (q & 3) + 3]};
    lds[(wave * 4 * OT + q) * 64 + lane] = v;
  }
}
template <int OT>
__device__ __forceinline__ f32x4 sum_partial(const f32x4* lds, int q, int lane) {      // fixed order: ((w0+w1)+w2)+w3
  f32x4 v = lds[(0 * 4 * OT + q) * 64 + lane];
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const f32x4 a = lds[(w * 4 * OT + q) * 64 + lane];
    v[0] += a[0];
    v[1] += a[1];
    v[2] += a[2];
    v[3] += a[3];
  }
  return v;
}

// sub-step offsets of A_t / B_t inside a pipeline-ordered layer-pair image (A0 A1 B0 A2 B1 ... A(HT-1) B(HT-2) B(HT-1))
__device__ __forceinline__ int pipe_off_A(int t, int NA, int NB) { return t == 0 ? 0 : NA + (t - 1) * (NA + NB); }
__device__ __forceinline__ int pipe_off_B(int t, int HT, int NA, int NB) {
  return t < HT - 1 ? 2 * NA + t * (NA + NB) : HT * NA + (HT - 1) * NB;
}

// ---- the closing MLP (128 -> 128 -> dout, 32 < dout <= 64) of ONE 32-row block by the 4 waves of a workgroup, inputs in LDS
// What mlp2_xs_body does behind its input-forming prologue, as a function: the typed-aggregation workgroups call it on the
// node rows whose aggregate they have just finished (gn_agg_group_t.y: the closing stage fused into the aggregation
// launch).  X: 32 rows x kXPitch floats, cat(H^T feat, ori) / divisor; lds: 32 KiB of exchange space (operand parts, then
// the partial outputs); every wave of the workgroup calls it with the same arguments.  Same image, same sub-step order,
// same sums as mlp2_xs_body: bit-identical rows.  (Requesting a wave's 12 sub-steps once per workgroup and keeping them
// for every row block it closes was measured: 96 / 144 more live registers, 156 bytes of scratch, launch +2 us.)
constexpr int kXPitch = 128 + 4;
// a wave's 12 sub-steps of the closing image (hidden tile `wave`: 8 of layer 1, then 4 of layer 2) and the ring that
// walks them; begin() is called by the workgroup BEFORE it forms the row block's inputs, so that the first sub-steps'
// L2 round trip runs beside the scatter instead of ahead of the first MFMA
template <int P>
struct ClosingStream {
  const f32x4* segA;
  const f32x4* segB;
#ifndef GN_CLOSING_DEPTH
#define GN_CLOSING_DEPTH 4      // sub-steps in flight.  Four cover ~400 cycles of MFMA work against an L2 round trip of ~1 k
#endif                          // beside a second workgroup: a row block's chain runs three round trips long (per-wave stamps:
                                // ~8 k cycles per row block).  Five or more cost the kernel scratch (20 / 80 / 128 bytes per
                                // lane at 5 / 6 / 8: measured +2 us) — the depth stays at what 254 registers allow.
  PStream<P, 12, GN_CLOSING_DEPTH> ps;
  __device__ __forceinline__ void init(const void* image, int wave, int lane, ovf_t& ovf) {
    if constexpr (P == 2) ovf.wf |= image_flag(image, 48);
    const f32x4* img = reinterpret_cast<const f32x4*>(image) + lane;
    segA = img + (size_t)pipe_off_A(wave, 8, 4) * P * 64;
    segB = img + (size_t)pipe_off_B(wave, 4, 8, 4) * P * 64;
  }
  __device__ __forceinline__ const f32x4* at(int s) const { return s < 8 ? segA + s * P * 64 : segB + (s - 8) * P * 64; }
  __device__ __forceinline__ void begin() {
    ps.begin([&](int s) { return at(s); });
  }
  __device__ __forceinline__ void step(int s, const Parts<P>& x, f32x16& acc) {
    ps.step(s, [&](int t) { return at(t); }, x, acc);
  }
};
template <int P, typename T>
__device__ __forceinline__ void closing_chain(ClosingStream<P>& cs, const float* __restrict__ bias, T* __restrict__ y, int ldy,
                                              int dout, int row0, int nlive, const float* __restrict__ X, f32x4* lds,
                                              int wave, int lane, ovf_t& ovf) {
  constexpr int IT = 4, HT = 4, OT = 2, NA = 2 * IT, NB = 2 * OT;
  const int h = lane >> 5, r = lane & 31;
  {
    f32x16 in;                                       // wave w forms the operand parts of input tile w
    const float* xr = X + r * kXPitch + 32 * wave + 4 * h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 8 * q);
      in[4 * q + 0] = v[0], in[4 * q + 1] = v[1], in[4 * q + 2] = v[2], in[4 * q + 3] = v[3];
    }
    put_parts<P>(lds, wave, lane, in, ovf);
  }
  __syncthreads();
  auto xop = [&](int u, Parts<P>& x) {
#pragma unroll
    for (int p = 0; p < P; ++p) x.p[p] = __builtin_bit_cast(bf16x8, lds[(u * P + p) * 64 + lane]);
  };
  Parts<P> xa[2];
  xop(0, xa[0]);
  const f32x16 bt = load_bias_tile(bias + 32 * wave, h);
  f32x16 hid;
#pragma unroll
  for (int q = 0; q < 16; ++q) hid[q] = 0.f;
#pragma unroll
  for (int u = 0; u < NA; ++u) {
    if (u + 1 < NA) xop(u + 1, xa[(u + 1) & 1]);
    cs.step(u, xa[u & 1], hid);
  }
  f32x16 out[OT], bo[OT];
#pragma unroll
  for (int o = 0; o < OT; ++o) {
    if (wave == 0) bo[o] = load_bias_tile(bias + 32 * HT + 32 * o, h);
#pragma unroll
    for (int q = 0; q < 16; ++q) out[o][q] = 0.f;
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) hid[q] += bt[q];
  relu16(hid);
  Parts<P> xh[2];
  make_parts<P>(hid, 0, xh[0], ovf);
  make_parts<P>(hid, 1, xh[1], ovf);
#pragma unroll
  for (int u = 0; u < NB; ++u) cs.step(NA + u, xh[u & 1], out[u >> 1]);
  if (wave == 0) {
#pragma unroll
    for (int o = 0; o < OT; ++o)
#pragma unroll
      for (int q = 0; q < 16; ++q) out[o][q] += bo[o][q];
  }
  __syncthreads();                                   // every wave is past its reads of the exchanged operands
  put_partial<OT>(lds, wave, lane, out);
  __syncthreads();
  if (r < nlive) {
    T* yrow = y + (size_t)(row0 + r) * ldy;
    const bool vec = ((dout | ldy) & 3) == 0;
#pragma unroll
    for (int qq = 0; qq < OT; ++qq) {                // wave w finishes quads [w*OT, w*OT + OT) of the 4*OT
      const int q = wave * OT + qq;
      const f32x4 v = sum_partial<OT>(lds, q, lane);
      const int f = 32 * (q >> 2) + 8 * (q & 3) + 4 * h;
      if (vec) {
        if (f < dout) st4(yrow + f, v);
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (f + c < dout) st1(yrow + f + c, v[c]);
      }
    }
  }
  __syncthreads();                                   // (the caller may reuse X / lds)
}

// ---- node form of the pairwise typed aggregation (gn_agg_group_t.node_form; N <= 16, K <= 12) ----------------------------
// edge_aggregation.forward consumes the per-edge feature only through H^T feat (MS_HGNN_batch.py:267), and the type
// weighting and layer 2 are linear, so they commute with that sum:
//   (H^T feat)[n] = sum_k W2k S[n,k] + sum_k b2k c[n,k],   S[n,k] = sum_j ef[p(n,j),k] relu(A[n,k] + A[j,k]),
//                                                          c[n,k] = sum_j ef[p(n,j),k]
// (ef = the pair rows' type weights with the self-loop's H = 2 folded in, as the edge kernel writes them).  Layer 2 then
// runs once per NODE — B*N rows instead of B*N(N+1)/2 — and so does the split into matrix operands; forming S is plain
// VALU work (N x 128 values per node and type) on rows staged in LDS.  A workgroup owns one 32-node row block: wave w
// takes hidden tile w of every type (its 4 sub-steps of the type's 16 straight from L2 into registers, requested before
// the type's VALU work), the scenes' pre-activation rows are staged type by type in a double-buffered LDS stage (one
// barrier per type), the type weights of the block's rows sit in `efs` ([row][k][16 partners], zero padded), the four
// waves' partial outputs meet in LDS in a fixed order.
//
// relu without a max: there is no packed fp32 max, so e * relu(a + b) is 1 packed add + 2 max + 1 packed fma per value
// pair.  With s a power of two such that s * |a + b| < 1 for every staged value of the type, relu(a + b) =
// clamp01(s b + s a) / s is ONE v_pk_fma_f32 with the clamp modifier, and 1/s goes into the type weight: two packed
// instructions per value pair.  Scaling by powers of two commutes with rounding, so the result is bit-identical to the
// max form (GN_NODE_CLAMP = 0 builds that one; a type whose largest pre-activation is beyond 2^72 takes it at run time).
// s comes from the maximum |A| of the staged rows, which the waves record while they stage a type (one slot per wave and
// buffer, read behind the barrier that publishes the buffer).
#ifndef GN_NODE_CLAMP
#define GN_NODE_CLAMP 1
#endif
constexpr int kNodeLoads = 8;                        // 16-byte pieces per thread and type of the stage (<= 64 nodes)
__host__ __device__ __forceinline__ int node_stage_nodes(int N) { return (31 / N + 2) * N; }
// floats of the launch's dynamic LDS the node form needs: stage buffer 1, efs, the waves' maxima
__host__ __device__ __forceinline__ int node_form_lds_floats(int N, int K) {
  return node_stage_nodes(N) * kStagePitch + 32 * (K * 16 + 1) + 12;
}
__device__ __forceinline__ f32x2 pk_fma_clamp(f32x2 a, f32x2 b, f32x2 c) {      // clamp01(a * b + c), both halves
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
template <int P, typename T>
__device__ __forceinline__ void agg_node_body(const gn_agg_group_t& G, int wg, int wave, int lane, const void* img,
                                              float* stage0, float* dyn, ovf_t& ovf, int unit) {
  const int N = G.N, E = G.E, K = G.K;
  const int rowsN = G.rows / E * N;                  // node rows of the group
  const int h = lane >> 5, r = lane & 31;
  const int g0 = wg * 32;
  const int grow = min(g0 + r, rowsN - 1);
  const int b = grow / N, i = grow - b * N;
  const int sb0 = g0 / N, sb1 = min(g0 + 31, rowsN - 1) / N;
  const int node0 = sb0 * N, nodes = (sb1 - sb0 + 1) * N;
  // LDS: stage buffer 0 = the (here unused) weight ring, buffer 1 and the type weights in the launch's dynamic part —
  // the launch must keep two workgroups per CU for the groups that run beside this one
  const int bufsz = node_stage_nodes(N) * kStagePitch;
  float* const stage1 = dyn;
  float* const efs = dyn + bufsz;
  const int efp = K * 16 + 1;                        // floats per row of efs (+1: the rows of a ds_read_b32 on different banks)
  unsigned* const wmax = reinterpret_cast<unsigned*>(efs + 32 * efp);   // [3][4]: slots k % 3 = max |A| each wave staged of type k
  const size_t ldA = (size_t)K * 128;
  const f32x4* Ag = reinterpret_cast<const f32x4*>(reinterpret_cast<const T*>(G.A) + (size_t)node0 * ldA);
  const int total4 = nodes * 32;
  f32x4 pre[kNodeLoads];
  auto fetch = [&](int kk) {
#pragma unroll
    for (int it = 0; it < kNodeLoads; ++it) {
      const int idx = min((int)threadIdx.x + it * 256, total4 - 1);
      if (it * 256 < total4) pre[it] = Ag[(size_t)(idx >> 5) * (ldA / 4) + kk * 32 + (idx & 31)];
    }
  };
  auto commit = [&](int buf, int kk) {              // kk: the type the registers hold
    float m = 0.f;
#pragma unroll
    for (int it = 0; it < kNodeLoads; ++it) {
      const int idx = (int)threadIdx.x + it * 256;
      if (it * 256 < total4) {                       // (uniform; clamped loads repeat the last piece: harmless in the max)
        if (idx < total4)
          *reinterpret_cast<f32x4*>((buf ? stage1 : stage0) + (idx >> 5) * kStagePitch + (idx & 31) * 4) = pre[it];
        if constexpr (GN_NODE_CLAMP != 0)
          m = fmaxf(fmaxf(m, fmaxf(fabsf(pre[it][0]), fabsf(pre[it][1]))), fmaxf(fabsf(pre[it][2]), fabsf(pre[it][3])));
      }
    }
    // wave maximum on DPP moves (a __shfl butterfly is six dependent LDS round trips; LDS atomics on one word serialise
    // 256 lanes: both measured, thousands of cycles per type), one slot per wave
    if constexpr (GN_NODE_CLAMP != 0) {
      auto dpp_max = [&](auto ctrl) {
        const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), decltype(ctrl)::value, 0xf, 0xf, false);
        m = fmaxf(m, __builtin_bit_cast(float, o));
      };
      dpp_max(std::integral_constant<int, 0xb1>{});    // quad_perm [1,0,3,2]
      dpp_max(std::integral_constant<int, 0x4e>{});    // quad_perm [2,3,0,1]
      dpp_max(std::integral_constant<int, 0x141>{});   // row_half_mirror
      dpp_max(std::integral_constant<int, 0x140>{});   // row_mirror: every lane holds its 16-lane row's maximum
      const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 0));
      const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 16));
      const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 32));
      const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 48));
      if (lane == 0) wmax[(kk % 3) * 4 + wave] = __builtin_bit_cast(unsigned, fmaxf(fmaxf(r0, r1), fmaxf(r2, r3)));
    }
  };
  fetch(0);
  // type weights of the block's rows: efs[row * efp + k * 16 + j] = ef[pair(i_row, j), k], 0 for j >= N.  Eight threads
  // per row, element e = k * 16 + j = (t & 7) + 8 * it; twelve gathers in flight per thread, then their LDS writes (a
  // load-wait-write loop would pay one L2 round trip per element)
  {
    const int rr = (int)threadIdx.x >> 3, sub = (int)threadIdx.x & 7;
    const int gr = min(g0 + rr, rowsN - 1);
    const int bb = gr / N, ii = gr - bb * N;
    const float* efb = G.edge_feat + (size_t)bb * E * K;
    float* dst = efs + rr * efp + sub;
    for (int it0 = 0; it0 * 8 < K * 16; it0 += 12) {
      float v[12];
#pragma unroll
      for (int u = 0; u < 12; ++u) {
        const int e = min(sub + 8 * (it0 + u), K * 16 - 1);
        const int k = e >> 4, j = e & 15;
        v[u] = j < N ? efb[(size_t)gn_pair_index(ii, j, N) * K + k] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 12; ++u)
        if ((it0 + u) * 8 < K * 16) dst[8 * (it0 + u)] = v[u];
    }
  }
  commit(0, 0);
  if (K > 1) fetch(1);
  __syncthreads();
  GN_STAMP(unit, 1);
  // (these waves end the launch — their instructions go first where a SIMD is shared — unless the closing stage is
  // fused: then the hyper groups' waves, with two row blocks to close, do)
  if (G.y == nullptr) __builtin_amdgcn_s_setprio(2);
  f32x16 out[2];
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int q = 0; q < 16; ++q) out[o][q] = 0.f;
  const f32x4* Wl = reinterpret_cast<const f32x4*>(img) + lane;
  const int base = (b - sb0) * N;                    // stage row of this lane's scene, node 0
  const int my = base + i;
#pragma unroll 1
  for (int k = 0; k < K; ++k) {
    const float* cb = ((k & 1) ? stage1 : stage0) + 32 * wave;
    if (k == 1) GN_STAMP(unit, 5);
    // scale of the clamp form for this type (uniform): s = 2^-(floor(log2 M) + 2) for M = max |A| staged, so that
    // s |a + b| < 1; M below 2^-27 is treated as 2^-27, M beyond 2^72 (or not finite) takes the max form
    float sc = 1.f, isc = 1.f;
    bool clamp_form = false;
    if constexpr (GN_NODE_CLAMP != 0) {
      // slots k % 3 were filled while type k was staged (before the barrier that published it); they are next written
      // while type k + 3 is staged, two barriers from here
      const u32x4 wm = *reinterpret_cast<const u32x4*>(wmax + (k % 3) * 4);      // (non-negative floats order like their bits)
      const int eb = max(gn_uniform((int)(max(max(wm[0], wm[1]), max(wm[2], wm[3])) >> 23)), 100);
      clamp_form = eb <= 199;
      sc = __builtin_bit_cast(float, (unsigned)(252 - min(eb, 199)) << 23);
      isc = __builtin_bit_cast(float, (unsigned)(min(eb, 199) + 2) << 23);
    }
    if (k + 1 < K) commit((k + 1) & 1, k + 1);       // (that buffer was last read before the barrier that closed type k - 1)
    if (k + 2 < K) fetch(k + 2);
    f32x4 w[4][P];                                   // [W2k(0,t) hf0, hf1, W2k(1,t) hf0, hf1] of hidden tile t = wave
    {
      const f32x4* wp = Wl + (size_t)((k * 16 + 4 * wave) * P) * 64;
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int p = 0; p < P; ++p) w[u][p] = wp[(u * P + p) * 64];
    }
    // b2k (wave 0 adds sum_k b2k c[n,k]): requested here, used behind the partner loop — a load waited for where it is
    // issued would also wait for the weight and stage loads above (vmcnt counts in order): one L2 round trip per type
    // (every lane loads, the select by wave / half happens at the use: a load under an exec mask into a register that
    // was zeroed first made the compiler wait for the stage loads above before it)
    const float* b2k = G.b2 + k * 64 + r;
    const float bf0 = b2k[0], bf1 = b2k[32];
    const PreTile a = load_pre(cb + my * kStagePitch, h);
    if (k == 1) GN_STAMP(unit, 6);
    f32x2 acc2[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc2[q] = f32x2{0.f, 0.f};
    float csum = 0.f;
    const float* er = efs + r * efp + k * 16;
    const float* prow = cb + base * kStagePitch;     // partner 0 of this lane's scene
    f32x16 acc;
    // partners in order, the rows and weights of the partner two ahead requested (LDS) before the current one's
    // arithmetic (one partner's arithmetic is shorter than an LDS round trip); the scheduling barriers keep the
    // requests where they are written
    auto partners = [&](auto accumulate) {
      auto row = [&](int j) { return load_pre(prow + min(j, N - 1) * kStagePitch, h); };
      auto wgt = [&](int j) { return er[min(j, N - 1)]; };
      PreTile pA = row(0), pB = row(1), pC;
      float eA = wgt(0), eB = wgt(1), eC;
      int j = 0;
      for (; j + 2 < N; j += 3) {
        pC = row(j + 2), eC = wgt(j + 2);
        __builtin_amdgcn_sched_barrier(0);
        accumulate(pA, eA);
        pA = row(j + 3), eA = wgt(j + 3);
        __builtin_amdgcn_sched_barrier(0);
        accumulate(pB, eB);
        pB = row(j + 4), eB = wgt(j + 4);
        __builtin_amdgcn_sched_barrier(0);
        accumulate(pC, eC);
      }
      if (j < N) accumulate(pA, eA);
      if (j + 1 < N) accumulate(pB, eB);
    };
    if (clamp_form) {
      f32x2 as[8];                                   // s * a
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 2; ++c) as[2 * q + c] = f32x2{a.v[q][2 * c] * sc, a.v[q][2 * c + 1] * sc};
      const f32x2 s2 = {sc, sc};
      partners([&](const PreTile& pb, float e) {
        const float es = e * isc;
        const f32x2 e2 = {es, es};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const f32x2 rl = pk_fma_clamp(f32x2{pb.v[q][2 * c], pb.v[q][2 * c + 1]}, s2, as[2 * q + c]);   // s relu(a + b)
            acc2[2 * q + c] = __builtin_elementwise_fma(e2, rl, acc2[2 * q + c]);
          }
        csum += e;
      });
    } else {
      partners([&](const PreTile& pb, float e) {
        const f32x2 e2 = {e, e};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const f32x2 s2 = f32x2{a.v[q][2 * c], a.v[q][2 * c + 1]} + f32x2{pb.v[q][2 * c], pb.v[q][2 * c + 1]};
            const f32x2 m2 = {fmaxf(s2[0], 0.f), fmaxf(s2[1], 0.f)};
            acc2[2 * q + c] = __builtin_elementwise_fma(e2, m2, acc2[2 * q + c]);
          }
        csum += e;
      });
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      acc[2 * q] = acc2[q][0];
      acc[2 * q + 1] = acc2[q][1];
    }
    if (k == 1) GN_STAMP(unit, 7);
    Parts<P> xh[2];
    make_parts<P>(acc, 0, xh[0], ovf);
    make_parts<P>(acc, 1, xh[1], ovf);
#pragma unroll
    for (int u = 0; u < 4; ++u) mfma_substep<P>(w[u], xh[u & 1], out[u >> 1]);
    if (wave == 0) {                                 // (as add_b2: lane (i, h = 0) carries b2k[32 o + i] against c[row, k])
      const float cs = h == 0 ? csum : 0.f;
      out[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(h == 0 ? bf0 : 0.f, cs, out[0], 0, 0, 0);
      out[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(h == 0 ? bf1 : 0.f, cs, out[1], 0, 0, 0);
    }
    __syncthreads();                                 // type k + 1 is visible; every wave is past its reads of type k
  }
  // partial outputs of the four waves (one hidden tile each) meet in LDS; wave w finishes quads 2w, 2w + 1
  GN_STAMP(unit, 2);
  f32x4* lds = reinterpret_cast<f32x4*>(dyn);          // (32 KiB; the launch's dynamic part is at least kAggPartBytes)
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 v = {out[q >> 2][4 * (q & 3) + 0], out[q >> 2][4 * (q & 3) + 1], out[q >> 2][4 * (q & 3) + 2],
                     out[q >> 2][4 * (q & 3) + 3]};
    lds[(wave * 8 + q) * 64 + lane] = v;
  }
  __syncthreads();
  if (G.y != nullptr) {
    // ---- fused closing stage: X = cat(H^T feat, ori) / divisor of the block's 32 nodes in stage buffer 0 (free now),
    // the chain's exchange space in the dynamic part (behind a barrier: the partial outputs are read from there first)
    float* X = stage0;
    ClosingStream<P> cs;
    cs.init(pick_image<P>(G.m2x, G.m2h), wave, lane, ovf);
    cs.begin();
    f32x4 v2[2];
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const int q = wave * 2 + qq;
      f32x4 v = lds[(0 * 8 + q) * 64 + lane];
#pragma unroll
      for (int ww = 1; ww < 4; ++ww) {
        const f32x4 t = lds[(ww * 8 + q) * 64 + lane];
        v[0] += t[0];
        v[1] += t[1];
        v[2] += t[2];
        v[3] += t[3];
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = v[c] / G.divisor;
      v2[qq] = v;
    }
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const int q = wave * 2 + qq;
      *reinterpret_cast<f32x4*>(X + r * kXPitch + 32 * (q >> 2) + 8 * (q & 3) + 4 * h) = v2[qq];
    }
    {
      const int row = (int)threadIdx.x >> 3, piece = (int)threadIdx.x & 7;
      const T* orow = reinterpret_cast<const T*>(G.ori) + (size_t)min(g0 + row, rowsN - 1) * GN_FEAT + 4 * piece;
      f32x4 o0 = ld4(orow), o1 = ld4(orow + 32);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        o0[c] = o0[c] / G.divisor;
        o1[c] = o1[c] / G.divisor;
      }
      *reinterpret_cast<f32x4*>(X + row * kXPitch + 64 + 4 * piece) = o0;
      *reinterpret_cast<f32x4*>(X + row * kXPitch + 96 + 4 * piece) = o1;
    }
    __syncthreads();
    closing_chain<P, T>(cs, G.m2bias, reinterpret_cast<T*>(G.y), G.ldy, G.dout, g0, rowsN - g0, X,
                        reinterpret_cast<f32x4*>(dyn), wave, lane, ovf);
    GN_STAMP(unit, 3);
    GN_STAMP(unit, 4);
    GN_STAMP(unit, 9);
    return;
  }
  if (g0 + r < rowsN) {
    T* yrow = reinterpret_cast<T*>(G.feat) + (size_t)(g0 + r) * GN_FEAT;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const int q = wave * 2 + qq;
      f32x4 v = lds[(0 * 8 + q) * 64 + lane];
#pragma unroll
      for (int ww = 1; ww < 4; ++ww) {
        const f32x4 t = lds[(ww * 8 + q) * 64 + lane];
        v[0] += t[0];
        v[1] += t[1];
        v[2] += t[2];
        v[3] += t[3];
      }
      st4(yrow + 32 * (q >> 2) + 8 * (q & 3) + 4 * h, v);
    }
  }
  GN_STAMP(unit, 3);
  GN_STAMP(unit, 4);
  GN_STAMP(unit, 9);
}

// ---- node form, ONE SCENE per workgroup (bf16 twins, N <= 64): both layers of the pairwise typed MLP per NODE -------------
// The same identity as agg_node_body, for graphs whose scenes do not fit a 32-node row block (config 4: N = 50, 1275
// pair rows per scene).  Per type k the workgroup (wave w = hidden tile w, the scene's nodes as one or two row blocks)
//   1. forms A'[n] = W1k ori_n + b1k/2 for its nodes on the matrix cores (eo = ori_i + ori_j makes layer 1 linear in the
//      two nodes; the twins store nothing per node in HBM) and writes the tile to an fp32 LDS stage,
//   2. accumulates S[n] = sum_j ef[p(n,j),k] relu(A'[n] + A'[j]) on the VALU — every lane reads the SAME partner row, the
//      pair weight comes from the type's column of the scene's edge_feat block staged in LDS (index p(n,j) =
//      min(f(j) + n, f(n) + j), f(m) = m (N - 1) - m (m - 1) / 2: no branch), clamp form as in agg_node_body,
//   3. applies W2k to S (bf16 operand, as the twins do) — B*N rows of matrix work per layer instead of B*N(N+1)/2.
// The weights are the two-layer image of the twins (pipeline order: A_w and B_w of each type), straight from L2.
#ifndef GN_SCENE_OCC
#define GN_SCENE_OCC 3          // workgroups per CU the scene-form kernel is compiled for (register budget 512 / occupancy)
#endif
// floats of dynamic LDS: the stage (N rows), two columns of pair weights, the waves' maxima — and at least the 32 KiB the
// partial outputs need at the end
__host__ __device__ __forceinline__ int node_scene_lds_floats(int N) {
  const int EP = (N * (N + 1) / 2 + 3) & ~3;
  const int need = N * kStagePitch + 2 * EP + 8;
  return need > 8192 ? need : 8192;
}
// One workgroup per (scene, 32-node row block): the VALU work of step 2 is what bounds this form, a wave issues an
// instruction every ~5-10 cycles whatever its neighbours do, and the matrix-core kernels' 256 registers allow two waves
// per SIMD — so the form has its own kernel, compiled for GN_SCENE_OCC workgroups per CU (measured inside
// agg_rb2_kernel, two per CU: 250 us for the pairwise module of config 4).  Every workgroup of a scene forms A' for all
// of the scene's nodes (matrix work, cheap) and S / layer 2 for its own row block.
template <typename T>
__global__ __launch_bounds__(256, GN_SCENE_OCC) void agg_scene_kernel(gn_agg_group_t G) {
  constexpr int P = 1;
  extern __shared__ __align__(16) float dyn[];
  ovf_t ovf_unused = {0ull, 0};
  const int N = G.N, E = G.E, K = G.K;
  const int RBN = (N + 31) >> 5;                       // row blocks of nodes per scene: 1 or 2
  const int scene = blockIdx.x / RBN, rbI = blockIdx.x - scene * RBN;
  const int wave = wave_id();
  const int lane = threadIdx.x & 63, h = lane >> 5, r = lane & 31;
  const int EP = (E + 3) & ~3;
  float* const stage = dyn;                            // [N][kStagePitch]: A' of the current type
  float* const efk = dyn + N * kStagePitch;            // [2][EP]: the scene's pair weights of the current / next type
  unsigned* const wmax = reinterpret_cast<unsigned*>(efk + 2 * EP);      // [4]: max |A'| each wave wrote (bits)
  const T* ori = reinterpret_cast<const T*>(G.ori) + (size_t)scene * N * GN_FEAT;
  const float* efb = G.edge_feat + (size_t)scene * E * K;
  const f32x4* Wl = reinterpret_cast<const f32x4*>(G.W12x) + lane;
  // this wave's sub-steps of type k in the pipeline-ordered image: A_w (layer 1, hidden tile w), B_w (its layer-2 slice)
  const int offA = pipe_off_A(wave, 4, 4), offB = pipe_off_B(wave, 4, 4, 4);
  constexpr int kEfLoads = 9;                          // E <= 2080 (N <= 64) over 256 threads
  float efr[kEfLoads];
  auto ef_fetch = [&](int kk) {
#pragma unroll
    for (int it = 0; it < kEfLoads; ++it)
      if (it * 256 < E) efr[it] = efb[(size_t)min((int)threadIdx.x + it * 256, E - 1) * K + kk];
  };
  auto ef_commit = [&](int buf) {
#pragma unroll
    for (int it = 0; it < kEfLoads; ++it)
      if (it * 256 < E && (int)threadIdx.x + it * 256 < E) efk[buf * EP + threadIdx.x + it * 256] = efr[it];
  };
  f32x4 w[4][P];                                       // A_w during step 1, B_w during steps 2 / 3
  auto w_fetch = [&](int kk, int off) {
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u][0] = Wl[(size_t)(kk * 32 + off + u) * 64];
  };
  ef_fetch(0);
  w_fetch(0, offA);
  f32x16 out[2];
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int q = 0; q < 16; ++q) out[o][q] = 0.f;
  const int n = min(rbI * 32 + r, N - 1);              // this lane's node (clamped: dead rows compute, never store)
  const int fn = n * (N - 1) - n * (n - 1) / 2;        // pair index of (n, j): min(f(j) + n, f(n) + j)
#pragma unroll 1
  for (int k = 0; k < K; ++k) {
    // ---- 1. layer 1 per node: hidden tile `wave` of type k for ALL the scene's nodes -> the stage
    {
      f32x16 b1h = load_bias_tile(G.b1 + k * 128 + 32 * wave, h);
#pragma unroll
      for (int q = 0; q < 16; ++q) b1h[q] *= 0.5f;
      float m = 0.f;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
        if (rb < RBN) {
          // (the scene's ori rows as operands: re-read per type — L2 hits beside other waves' work — instead of 32
          // resident registers, which this kernel's register budget does not have)
          Parts<P> xi[2][2];
          {
            f32x16 in[2];
            load_rows<2>(ori, GN_FEAT, min(rb * 32 + r, N - 1), h, in);
            make_parts_tiles<P, 2>(in, xi, ovf_unused);
          }
          f32x16 hid = b1h;
#pragma unroll
          for (int u = 0; u < 4; ++u) mfma_substep<P>(w[u], xi[u >> 1][u & 1], hid);
          if (rb * 32 + r < N) {
            float* dst = stage + (rb * 32 + r) * kStagePitch + 32 * wave + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 v = {hid[4 * q], hid[4 * q + 1], hid[4 * q + 2], hid[4 * q + 3]};
              *reinterpret_cast<f32x4*>(dst + 8 * q) = v;
              m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
            }
          }
        }
      ef_commit(k & 1);
      auto dpp_max = [&](auto ctrl) {
        const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), decltype(ctrl)::value, 0xf, 0xf, false);
        m = fmaxf(m, __builtin_bit_cast(float, o));
      };
      dpp_max(std::integral_constant<int, 0xb1>{});
      dpp_max(std::integral_constant<int, 0x4e>{});
      dpp_max(std::integral_constant<int, 0x141>{});
      dpp_max(std::integral_constant<int, 0x140>{});
      const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 0));
      const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 16));
      const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 32));
      const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 48));
      if (lane == 0) wmax[wave] = __builtin_bit_cast(unsigned, fmaxf(fmaxf(r0, r1), fmaxf(r2, r3)));
    }
    // (B_w of this type: requested before the barrier, used behind the partner loop)
    w_fetch(k, offB);
    __syncthreads();
    // ---- 2. S = sum_j ef relu(A'_n + A'_j) for this workgroup's row block
    const int kn = k + 1 < K ? k + 1 : k;
    ef_fetch(kn);
    const float bf0 = G.b2[k * 64 + r], bf1 = G.b2[k * 64 + 32 + r];
    const u32x4 wm = *reinterpret_cast<const u32x4*>(wmax);
    const int eb = max(gn_uniform((int)(max(max(wm[0], wm[1]), max(wm[2], wm[3])) >> 23)), 100);
    const bool clamp_form = eb <= 199;
    const float sc = __builtin_bit_cast(float, (unsigned)(252 - min(eb, 199)) << 23);
    const float isc = __builtin_bit_cast(float, (unsigned)(min(eb, 199) + 2) << 23);
    const float* ec = efk + (k & 1) * EP;
    const float* prow = stage + 32 * wave;
    f32x2 acc2[8];
    float csum = 0.f;
    {
      const PreTile a = load_pre(prow + n * kStagePitch, h);
      f32x2 as[8];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          as[2 * q + c] = clamp_form ? f32x2{a.v[q][2 * c] * sc, a.v[q][2 * c + 1] * sc} : f32x2{a.v[q][2 * c], a.v[q][2 * c + 1]};
          acc2[2 * q + c] = f32x2{0.f, 0.f};
        }
      const f32x2 s2 = {sc, sc};
      // partner j: row j of the stage (the same for every lane), weight ef[p(n, j)]
      auto partner = [&](int j, auto clamped) {
        const PreTile pb = load_pre(prow + j * kStagePitch, h);
        const int fj = j * (N - 1) - j * (j - 1) / 2;
        const float e = ec[min(fj + n, fn + j)];
        csum += e;
        if constexpr (decltype(clamped)::value) {
          const float es = e * isc;
          const f32x2 e2 = {es, es};
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const f32x2 rl = pk_fma_clamp(f32x2{pb.v[q][2 * c], pb.v[q][2 * c + 1]}, s2, as[2 * q + c]);
              acc2[2 * q + c] = __builtin_elementwise_fma(e2, rl, acc2[2 * q + c]);
            }
        } else {
          const f32x2 e2 = {e, e};
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const f32x2 t2 = as[2 * q + c] + f32x2{pb.v[q][2 * c], pb.v[q][2 * c + 1]};
              const f32x2 m2 = {fmaxf(t2[0], 0.f), fmaxf(t2[1], 0.f)};
              acc2[2 * q + c] = __builtin_elementwise_fma(e2, m2, acc2[2 * q + c]);
            }
        }
      };
      // (no explicit software pipeline: with GN_SCENE_OCC waves per SIMD the other waves cover a partner's LDS round trip)
      if (clamp_form) {
#pragma unroll 2
        for (int j = 0; j < N; ++j) partner(j, std::integral_constant<bool, true>{});
      } else {
#pragma unroll 2
        for (int j = 0; j < N; ++j) partner(j, std::integral_constant<bool, false>{});
      }
    }
    // ---- 3. layer 2 on S (bf16 operand), this wave's hidden tile; wave 0 adds b2k c[n, k]
    {
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        acc[2 * q] = acc2[q][0];
        acc[2 * q + 1] = acc2[q][1];
      }
      Parts<P> xh[2];
      make_parts<P>(acc, 0, xh[0], ovf_unused);
      make_parts<P>(acc, 1, xh[1], ovf_unused);
#pragma unroll
      for (int u = 0; u < 4; ++u) mfma_substep<P>(w[u], xh[u & 1], out[u >> 1]);
      if (wave == 0) {
        const float cs = h == 0 ? csum : 0.f;
        out[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(h == 0 ? bf0 : 0.f, cs, out[0], 0, 0, 0);
        out[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(h == 0 ? bf1 : 0.f, cs, out[1], 0, 0, 0);
      }
    }
    w_fetch(kn, offA);                                 // A_w of the next type
    __syncthreads();                                   // every wave is past its reads of the stage and of efk[k & 1]
  }
  // partial outputs of the four waves meet in LDS; wave w finishes quads 2w, 2w + 1
  f32x4* lds = reinterpret_cast<f32x4*>(dyn);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 v = {out[q >> 2][4 * (q & 3) + 0], out[q >> 2][4 * (q & 3) + 1], out[q >> 2][4 * (q & 3) + 2],
                     out[q >> 2][4 * (q & 3) + 3]};
    lds[(wave * 8 + q) * 64 + lane] = v;
  }
  __syncthreads();
  if (rbI * 32 + r < N) {
    T* yrow = reinterpret_cast<T*>(G.feat) + ((size_t)scene * N + rbI * 32 + r) * GN_FEAT;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const int q = wave * 2 + qq;
      f32x4 v = lds[(0 * 8 + q) * 64 + lane];
#pragma unroll
      for (int ww = 1; ww < 4; ++ww) {
        const f32x4 t = lds[(ww * 8 + q) * 64 + lane];
        v[0] += t[0];
        v[1] += t[1];
        v[2] += t[2];
        v[3] += t[3];
      }
      st4(yrow + 32 * (q >> 2) + 8 * (q & 3) + 4 * h, v);
    }
  }
}

// PAIRF: the per-pair forms of the pairwise graph are compiled in.  They are what costs the kernel its last registers (28
// bytes of scratch per lane); since the node form replaced them in the default path the launcher picks the instantiation
// without them whenever no group of the launch needs one (fp32 path beyond N = 16, GN_NODE_FORM = 0).
template <int P, typename T, bool PAIRF>
__device__ __forceinline__ void agg_x_body(const GroupTable<AggGroup>& Tb, f32x4* wring, ovf_t& ovf) {
  using WS = WStream<P>;
  constexpr int CH = WS::CH;
  // wpr > 1: [wave][register 0..31][lane]; staged pair form: 2 x node rows.  Dynamic: a launch whose groups all run one
  // wave per row block without the stage (the large bf16 configurations) passes 0 bytes and fits more workgroups per CU.
  extern __shared__ __align__(16) float part_dyn[];
  float (*part)[32][64 + 8] = reinterpret_cast<float (*)[32][64 + 8]>(part_dyn);
  const int lwg = gn_uniform(gn_xcd_logical(Tb.xs, blockIdx.x));
  if (lwg < 0) return;
  const int gi = find_group(Tb, lwg);
  const gn_agg_group_t G = Tb.g[gi].a;
  const int wpr = Tb.g[gi].wpr;
  const int rows = G.rows, K = G.K;
  const int wave = wave_id();
  const int wg = lwg - Tb.first_wg[gi];
  // wpr == 1: every wave owns a 32-row block and walks all K types; the workgroup's 4 waves share ONE weight stream
  //           through LDS (WStream);
  // wpr  > 1: wpr waves share one row block (groups with few row blocks: shorter critical path), wave `sub` of them
  //           takes the types sub, sub + wpr, ... through a private register ring (XStream), the partial sums meet
  //           in LDS.
  const int sub = wave % wpr;
  const int blk = wg * (4 / wpr) + wave / wpr;
  const bool staged = Tb.g[gi].stage != 0;
  // Fused closing stage of a hyper group (spw > 0): the workgroup's edge rows are the `spw` whole scenes wg*spw ..., so
  // that it can form H^T feat of those scenes' nodes itself — rows_used = spw * E of its 128 / wpr row slots
  const int spw = Tb.g[gi].spw;
  const bool fused = G.y != nullptr;
  const int rows_used = spw * G.E;
  bool any_rows_ = blk * 32 < rows;
  RowBlock rb_ = row_block(rows, any_rows_ ? blk : 0);         // (no early exit: every wave takes part in the barriers)
  if (fused && !G.node_form) {
    const int lb = wave / wpr, lr = lb * 32 + (rb_.lane & 31);
    const int row = wg * rows_used + lr;
    any_rows_ = lb * 32 < rows_used && wg * rows_used + lb * 32 < rows;
    rb_.live = lr < rows_used && row < rows;
    rb_.row = row;
    rb_.row_ld = rb_.live ? row : min(wg * rows_used + min(lr, rows_used - 1), rows - 1);
  }
  const bool any_rows = any_rows_;
  const RowBlock rb = rb_;
  const int lane = rb.lane, h = rb.h;
  f32x16 out[2];
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[o][r] = 0.f;
  const float* efrow = G.edge_feat + (size_t)rb.row_ld * K;
  const float* b1 = G.b1;
  const float* b2 = G.b2;
  WS ws;
  const int unit = lwg * 4 + wave;
  GN_STAMP(unit, 0);
  GN_STAMP(unit, 8);

  // hyper gather from the workgroup's scenes staged in LDS (block-uniform; the stage shares `part` with the partial
  // sums, which are exchanged behind a barrier at the end)
  const bool hstage = Tb.g[gi].lines == 2;
  int hnode0 = 0;
  if (hstage) {
    const int rpw = fused ? rows_used : 128 / wpr, r0 = min(wg * rpw, rows - 1);
    hnode0 = ori_stage_fill<T>(G, r0, min(rows - 1, r0 + rpw - 1), reinterpret_cast<T*>(part_dyn));
    __syncthreads();
  }
  bool pair_form = false;
  if constexpr (P != 1) pair_form = G.A != nullptr;
  if constexpr (!PAIRF) {
    if (pair_form && !G.node_form) return;      // (the launcher never sends such a group here)
  }
  // (the weight image of this group's form and its flag word)
  const void* img = pair_form ? pick_image<P>(G.W2x, G.W2h) : pick_image<P>(G.W12x, G.W12h);
  if constexpr (P == 2) ovf.wf |= image_flag(img, pair_form ? K * 16 : K * 32);
  if constexpr (P != 1) {
    if (pair_form && G.node_form) {      // (block-uniform; the workgroup's four waves share one 32-node row block)
      agg_node_body<P, T>(G, wg, wave, threadIdx.x & 63, img, reinterpret_cast<float*>(wring), part_dyn, ovf, lwg * 4 + wave);
      return;
    }
  }
  if (PAIRF && pair_form && wpr == 1) {
    // ---- pair form: hid_t = relu(A_i + A_j) * ef_k is VALU work (V_t), its layer-2 slice the matrix work (B_t:
    // one chunk of the stream, 16 sub-steps per type).  wpr == 1, pipelined: V of the NEXT tile (the next type's
    // tile 0 after t == 3) is interleaved with the MFMAs of B_t; the pre-activations it needs were loaded one tile
    // earlier still.
    const int N = G.N, Pn = G.E;
    // The pair-form waves are the launch's critical path (they end ~15 us after the hyper modules' waves at B = 512):
    // where one shares a SIMD with a two-layer wave its instructions issue first.  Measured: 57 -> 52 us.
    __builtin_amdgcn_s_setprio(1);
    int i, j;
    {
      const int b = rb.row_ld / Pn, p = rb.row_ld - b * Pn;
      gn_pair_decode(p, N, i, j);
      i += b * N;
      j += b * N;
    }
    const size_t ldA = (size_t)K * 128;
    const T* Abase = reinterpret_cast<const T*>(G.A);
    ws.begin(img, wring, lane, wave, K * 16 / CH);
    auto hidden = [&](const PreTile& pa, const PreTile& pb, float efk, Parts<P> (&xh)[2]) {
      f32x16 hid;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) hid[4 * q + c] = fmaxf(pa.v[q][c] + pb.v[q][c], 0.f) * efk;
      make_parts<P>(hid, 0, xh[0], ovf);
      make_parts<P>(hid, 1, xh[1], ovf);
    };
    // B_t with the next tile's V in its shadow
    auto slice = [&](int c0, int t, const Parts<P> (&xh)[2], PreTile& pa, PreTile& pb, float ef_next,
                     Parts<P> (&xh_next)[2], auto na, auto nb_) {
      hidden(pa, pb, ef_next, xh_next);
      pa = load_pre(na, h);          // the tile after the next one (LDS)
      pb = load_pre(nb_, h);
#pragma unroll
      for (int u = 0; u < 4; ++u)                    // [W2(0,t) hf0, hf1, W2(1,t) hf0, hf1]
        ws.template step<false>(c0, 4 * t + u, xh[u & 1], out[u >> 1]);
#pragma unroll
      for (int m = 0; m < 4 * kMfmaPerSub<P>; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, P == 2 ? 11 : 7, 0);
        if (m % kMfmaPerSub<P> == 0) __builtin_amdgcn_sched_group_barrier(0x100, P, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    if (staged) {
      // The 4 row blocks of this workgroup touch a short run of consecutive node rows: the workgroup copies that
      // run (one type at a time, coalesced, prefetched in registers a type ahead) into one of two LDS buffers.
      float* stage = &part[0][0][0];
      const int r0 = wg * 128, r1 = min(rows - 1, r0 + 127);
      const int node0 = (r0 / Pn) * N;
      const int nodes = (r1 / Pn + 1) * N - node0;
      const f32x4* Ag = reinterpret_cast<const f32x4*>(Abase + (size_t)node0 * ldA);
      const int total4 = nodes * 32;                      // 16-byte pieces per type (fp32 storage)
      f32x4 pre[kStageLoadsX];
      auto fetch = [&](int kk) {
#pragma unroll
        for (int it = 0; it < kStageLoadsX; ++it) {
          const int idx = min((int)threadIdx.x + it * 256, total4 - 1);
          pre[it] = Ag[(size_t)(idx >> 5) * (ldA / 4) + kk * 32 + (idx & 31)];
        }
      };
      auto commit = [&](int buf) {
#pragma unroll
        for (int it = 0; it < kStageLoadsX; ++it) {
          const int idx = (int)threadIdx.x + it * 256;
          if (idx < total4)
            *reinterpret_cast<f32x4*>(stage + buf * kStageBuf + (idx >> 5) * kStagePitch + (idx & 31) * 4) = pre[it];
        }
      };
      const int oi = (i - node0) * kStagePitch, oj = (j - node0) * kStagePitch;
      fetch(0);
      commit(0);
      fetch(K > 1 ? 1 : 0);
      __syncthreads();
      float efk = efrow[0];
      Parts<P> xh[2];
      PreTile pa = load_pre(stage + oi, h), pb = load_pre(stage + oj, h);
      hidden(pa, pb, efk, xh);                            // V of (type 0, tile 0): the only exposed one
      pa = load_pre(stage + oi + 32, h);
      pb = load_pre(stage + oj + 32, h);
#pragma unroll 1
      for (int k = 0; k < K; ++k) {
        const int kc = k + 1 < K ? k + 1 : k;
        const float efk_next = efrow[kc];
        const float* cb = stage + (k & 1) * kStageBuf;        // this type's rows
        const float* nb = stage + (kc & 1) * kStageBuf;       // the next type's
        if (k + 1 < K) commit((k + 1) & 1);                   // (that buffer was last read during type k - 1)
        if (k + 2 < K) fetch(k + 2);
        add_b2(b2 + k * 64, efk, lane, h, out);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          // pa/pb hold tile t+1 (tile 0 of the next type when t == 3); fetch the one after it.  The stream's chunk
          // boundary at the head of tile 2 is a workgroup barrier behind the commit above: the next type's rows
          // are visible from there on.
          if (t == 2) __syncthreads();
          const float* nsrc = t < 2 ? cb + 32 * (t + 2) : nb + 32 * (t - 2);
          Parts<P> xn[2];
          slice(k * 16 / CH, t, xh, pa, pb, t < 3 ? efk : efk_next, xn, nsrc + oi, nsrc + oj);
          xh[0] = xn[0];
          xh[1] = xn[1];
        }
        efk = efk_next;
      }
    } else if (wpr == 1) {
      // the rows do not fit the LDS stage: pre-activations straight from HBM / L2, same pipeline
      const T* Ai = Abase + (size_t)i * ldA;
      const T* Aj = Abase + (size_t)j * ldA;
      float efk = efrow[0];
      Parts<P> xh[2];
      PreTile pa = load_pre(Ai, h), pb = load_pre(Aj, h);
      hidden(pa, pb, efk, xh);
      pa = load_pre(Ai + 32, h);
      pb = load_pre(Aj + 32, h);
#pragma unroll 1
      for (int k = 0; k < K; ++k) {
        const int kc = k + 1 < K ? k + 1 : k;
        const float efk_next = efrow[kc];
        add_b2(b2 + k * 64, efk, lane, h, out);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int off = t < 2 ? k * 128 + 32 * (t + 2) : kc * 128 + 32 * (t - 2);
          Parts<P> xn[2];
          slice(k * 16 / CH, t, xh, pa, pb, t < 3 ? efk : efk_next, xn, Ai + off, Aj + off);
          xh[0] = xn[0];
          xh[1] = xn[1];
        }
        efk = efk_next;
      }
    }
  } else if (PAIRF && pair_form) {
    // ---- pair form, wpr > 1: types dealt over the waves of a row block, private register rings ----------------------
    if (any_rows && sub < K) {
      const int N = G.N, Pn = G.E;
      int i, j;
      {
        const int b = rb.row_ld / Pn, p = rb.row_ld - b * Pn;
        gn_pair_decode(p, N, i, j);
        i += b * N;
        j += b * N;
      }
      const size_t ldA = (size_t)K * 128;
      const T* Ai = reinterpret_cast<const T*>(G.A) + (size_t)i * ldA;
      const T* Aj = reinterpret_cast<const T*>(G.A) + (size_t)j * ldA;
      const f32x4* Wx = reinterpret_cast<const f32x4*>(img) + lane;     // sub-step s of type k: (k*16 + s)
      XStream<P> xs;
      xs.begin(Wx + (size_t)sub * 16 * P * 64);
#pragma unroll 1
      for (int k = sub; k < K; k += wpr) {
        const int kc = k + wpr < K ? k + wpr : k;
        const float efk = efrow[k];
        xs.segment(Wx + (size_t)k * 16 * P * 64, Wx + (size_t)kc * 16 * P * 64, 16);
        add_b2(b2 + k * 64, efk, lane, h, out);
        PreTile pa = load_pre(Ai + k * 128, h), pb = load_pre(Aj + k * 128, h);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          f32x16 hid;
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) hid[4 * q + c] = fmaxf(pa.v[q][c] + pb.v[q][c], 0.f) * efk;
          if (t < 3) {
            pa = load_pre(Ai + k * 128 + 32 * (t + 1), h);
            pb = load_pre(Aj + k * 128 + 32 * (t + 1), h);
          }
          Parts<P> xh[2];
          make_parts<P>(hid, 0, xh[0], ovf);
          make_parts<P>(hid, 1, xh[1], ovf);
#pragma unroll
          for (int u = 0; u < 4; ++u) xs.step(0, 4 * t + u, xh[u & 1], out[u >> 1]);
        }
      }
    }
  } else if (wpr == 1) {
    // ---- two-layer form, hidden tile by hidden tile (32 sub-steps per type, pipeline order A0 A1 B0 A2 B1 A3 B2 B3:
    // A_t = tile t of layer 1 (4 sub-steps), ReLU * ef_k, its bf16 part(s), B_t = its contribution to both output
    // tiles (4 sub-steps)) — one hidden tile live --------------------------------------------------------------
    f32x16 in[2];
    if (G.eo != nullptr) {
      load_rows<2>(reinterpret_cast<const T*>(G.eo), GN_FEAT, rb.row_ld, h, in);
    } else if (hstage) {
      gather_hyper_staged<T>(G, rb.row_ld, h, reinterpret_cast<const T*>(part_dyn), hnode0, in);
    } else if (Tb.g[gi].lines) {
      gather_rows_lines<T>(G, blk, rows, lane, &part[wave][0][0]);
      __builtin_amdgcn_wave_barrier();
      read_rows_lines(&part[wave][0][0], lane, in);
    } else {
      gather_rows<T>(G, rb.row_ld, h, in);
    }
    Parts<P> xi[2][2];
    make_parts_tiles<P, 2>(in, xi, ovf);
    GN_STAMP(unit, 1);
    ws.begin(img, wring, lane, wave, K * 32 / CH);
    f32x16 hid0 = load_bias_tile(b1, h);
    float efk = efrow[0];
    float bf0 = 0.f, bf1 = 0.f;
    if constexpr (P != 1) bf0 = b2[lane & 31], bf1 = b2[32 + (lane & 31)];
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
      const int kc = k + 1 < K ? k + 1 : k;
      const float efk_next = efrow[kc];
      float bn0 = 0.f, bn1 = 0.f;
      if constexpr (P != 1) bn0 = b2[kc * 64 + (lane & 31)], bn1 = b2[kc * 64 + 32 + (lane & 31)];
      const f32x16 hid0_next = load_bias_tile(b1 + kc * 128, h);
      if constexpr (P == 1) {
        // VALU-lean form: the type's layer 2 accumulates into a temporary that starts at b2_k, ReLU runs on the
        // packed bf16 operands, and feat += ef_k * (W2k relu(..) + b2k) is 32 FMAs per type — instead of scaling every
        // hidden value (64 multiplications) and two extra fp32 MFMAs for the bias
        f32x16 tmp[2];
        tmp[0] = load_bias_tile(b2 + k * 64, h);
        tmp[1] = load_bias_tile(b2 + k * 64 + 32, h);
        layer_pair<P, 2, 2, 4, true>(ws, k * 32 / CH, 0, xi, hid0, b1 + k * 128, h, tmp, ovf, [&](int, f32x16&) {});
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
          for (int r = 0; r < 16; ++r) out[o][r] = fmaf(efk, tmp[o][r], out[o][r]);
      } else {
        add_b2_regs(bf0, bf1, efk, h, out);
        layer_pair<P, 2, 2, 4>(ws, k * 32 / CH, 0, xi, hid0, b1 + k * 128, h, out, ovf,
                               [&](int, f32x16& hid) { relu_scale16(hid, efk); });
      }
      hid0 = hid0_next;
      efk = efk_next, bf0 = bn0, bf1 = bn1;
    }
  } else if (any_rows && sub < K) {
    // ---- two-layer form, wpr > 1: types dealt over the waves of a row block, private register rings ------------------
    if (fused) __builtin_amdgcn_s_setprio(1);
    f32x16 in[2];
    if (G.eo != nullptr) {
      load_rows<2>(reinterpret_cast<const T*>(G.eo), GN_FEAT, rb.row_ld, h, in);
    } else if (hstage) {
      gather_hyper_staged<T>(G, rb.row_ld, h, reinterpret_cast<const T*>(part_dyn), hnode0, in);
    } else if (Tb.g[gi].lines) {
      // (every wave of the row block forms the rows itself, in its own scratch; the partial sums reuse it at the end)
      gather_rows_lines<T>(G, blk, rows, lane, &part[wave][0][0]);
      __builtin_amdgcn_wave_barrier();
      read_rows_lines(&part[wave][0][0], lane, in);
    } else {
      gather_rows<T>(G, rb.row_ld, h, in);
    }
    Parts<P> xi[2][2];
    make_parts_tiles<P, 2>(in, xi, ovf);
    GN_STAMP(unit, 1);
    const f32x4* Wx = reinterpret_cast<const f32x4*>(img) + lane;      // sub-step s of type k: (k*32 + s)
    XStream<P> xs;
    xs.begin(Wx + (size_t)sub * 32 * P * 64);
    f32x16 hid0 = load_bias_tile(b1 + sub * 128, h);
    float efk = efrow[sub], bf0 = b2[sub * 64 + (lane & 31)], bf1 = b2[sub * 64 + 32 + (lane & 31)];
#pragma unroll 1
    for (int k = sub; k < K; k += wpr) {
      const int kc = k + wpr < K ? k + wpr : k;
      const float efk_next = efrow[kc], bn0 = b2[kc * 64 + (lane & 31)], bn1 = b2[kc * 64 + 32 + (lane & 31)];
      const f32x16 hid0_next = load_bias_tile(b1 + kc * 128, h);
      xs.segment(Wx + (size_t)k * 32 * P * 64, Wx + (size_t)kc * 32 * P * 64, 32);
      if constexpr (P == 1) {
        f32x16 tmp[2];
        tmp[0] = load_bias_tile(b2 + k * 64, h);
        tmp[1] = load_bias_tile(b2 + k * 64 + 32, h);
        layer_pair<P, 2, 2, 4, true>(xs, 0, 0, xi, hid0, b1 + k * 128, h, tmp, ovf, [&](int, f32x16&) {});
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
          for (int r = 0; r < 16; ++r) out[o][r] = fmaf(efk, tmp[o][r], out[o][r]);
      } else {
        if (k == sub + wpr) GN_STAMP(unit, 5);
        add_b2_regs(bf0, bf1, efk, h, out);
        if (k == sub + wpr) GN_STAMP(unit, 6);
        layer_pair<P, 2, 2, 4>(xs, 0, 0, xi, hid0, b1 + k * 128, h, out, ovf, [&](int, f32x16& hid) { relu_scale16(hid, efk); });
        if (k == sub + wpr) GN_STAMP(unit, 7);
      }
      hid0 = hid0_next;
      efk = efk_next, bf0 = bn0, bf1 = bn1;
    }
  }
  T* feat = reinterpret_cast<T*>(G.feat);
  GN_STAMP(unit, 2);
  if (wpr == 1) {
    store_rows<2>(feat, GN_FEAT, rb.row, h, rb.live && any_rows, out);
    GN_STAMP(unit, 4);
    GN_STAMP(unit, 9);
    return;
  }
  __syncthreads();      // (the staged node rows share `part`; not used together with wpr > 1, but keep the order explicit)
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][16 * o + r][lane] = out[o][r];
  __syncthreads();
  // the wpr waves of a row block each finish 32/wpr of its registers
  if constexpr (P != 1) {
    if (fused) {
      // ---- fused closing stage: feat of the workgroup's scenes -> LDS, H^T feat and ori of their nodes -> X, closing MLP
      // per 32-node row block (closing_chain).  The summed registers first (all reads of `part` done), then `part`'s LDS
      // is reused: F = feat rows [rows_used][kFPitch], X behind it.
      constexpr int kFPitch = GN_FEAT + 4;
      f32x4 fv[8];
      const int w0 = wave - sub, nreg = 32 / wpr;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (u * 4 < nreg) {
          const int reg0 = sub * nreg + u * 4;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          for (int jw = 0; jw < wpr; ++jw) {
            v[0] += part[w0 + jw][reg0 + 0][lane];
            v[1] += part[w0 + jw][reg0 + 1][lane];
            v[2] += part[w0 + jw][reg0 + 2][lane];
            v[3] += part[w0 + jw][reg0 + 3][lane];
          }
          fv[u] = v;
        }
      }
      __syncthreads();
      float* F = part_dyn;
      float* X = part_dyn + 64 * kFPitch;
      {
        const int lr = (wave / wpr) * 32 + (lane & 31);
        if (lr < rows_used) {
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (u * 4 < nreg) {
              const int reg0 = sub * nreg + u * 4;
              const int o = reg0 >> 4, q = (reg0 & 15) >> 2;
              *reinterpret_cast<f32x4*>(F + lr * kFPitch + 32 * o + 8 * q + 4 * h) = fv[u];
            }
        }
      }
      __syncthreads();
      const int N = G.N, E = G.E;
      const int scene0 = wg * spw, B = rows / E;
      const int nodes = min(spw, B - scene0) * N;                  // node rows of this workgroup (>= 1)
      const T* orib = reinterpret_cast<const T*>(G.ori);
      ClosingStream<P> cs;
      cs.init(pick_image<P>(G.m2x, G.m2h), wave, lane, ovf);
      for (int nb = 0; nb * 32 < nodes; ++nb) {
        cs.begin();
        // X[row][0..63] = (sum_e H[b,e,n] feat[b,e]) / divisor in the order of the fused scatter of gn_mlp2 (e ascending,
        // fmaf), X[row][64..127] = ori[b,n] / divisor; thread (row = t / 8, piece = t % 8) forms two 16-byte pieces of each
        const int row = (int)threadIdx.x >> 3, piece = (int)threadIdx.x & 7;
        const int m = min(nb * 32 + row, nodes - 1);               // node of this workgroup (clamped)
        const int sl = m / N, n = m - sl * N;
        const float* hcol = G.H + ((size_t)(scene0 + sl) * E) * N + n;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        // (the node's incidence column first, all of it in flight at once — E <= 16 here: a load per loop iteration cost
        // one L2 round trip per hyperedge, 8 k cycles per row block in the per-wave stamps)
        float hw[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) hw[e] = e < E ? hcol[(size_t)e * N] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (e < E) {
            const float* fr = F + (sl * E + e) * kFPitch + 4 * piece;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(fr), v1 = *reinterpret_cast<const f32x4*>(fr + 32);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              a0[c] = fmaf(hw[e], v0[c], a0[c]);
              a1[c] = fmaf(hw[e], v1[c], a1[c]);
            }
          }
        const T* orow = orib + ((size_t)scene0 * N + m) * GN_FEAT + 4 * piece;
        f32x4 o0 = ld4(orow), o1 = ld4(orow + 32);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          a0[c] = a0[c] / G.divisor;
          a1[c] = a1[c] / G.divisor;
          o0[c] = o0[c] / G.divisor;
          o1[c] = o1[c] / G.divisor;
        }
        float* xr = X + row * kXPitch + 4 * piece;
        *reinterpret_cast<f32x4*>(xr) = a0;
        *reinterpret_cast<f32x4*>(xr + 32) = a1;
        *reinterpret_cast<f32x4*>(xr + 64) = o0;
        *reinterpret_cast<f32x4*>(xr + 96) = o1;
        __syncthreads();
        closing_chain<P, T>(cs, G.m2bias, reinterpret_cast<T*>(G.y), G.ldy, G.dout, scene0 * N + nb * 32, nodes - nb * 32, X,
                            wring, wave, lane, ovf);
      }
      GN_STAMP(unit, 4);
      GN_STAMP(unit, 9);
      return;
    }
  }
  if (rb.live && any_rows) {
    T* p = feat + (size_t)rb.row * GN_FEAT + 4 * h;
    const int w0 = wave - sub;
    const int nreg = 32 / wpr;
    for (int rr = 0; rr < nreg; rr += 4) {
      const int reg0 = sub * nreg + rr;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      for (int jw = 0; jw < wpr; ++jw) {
        v[0] += part[w0 + jw][reg0 + 0][lane];
        v[1] += part[w0 + jw][reg0 + 1][lane];
        v[2] += part[w0 + jw][reg0 + 2][lane];
        v[3] += part[w0 + jw][reg0 + 3][lane];
      }
      const int o = reg0 >> 4, q = (reg0 & 15) >> 2;
      st4(p + 32 * o + 8 * q, v);
    }
  }
  GN_STAMP(unit, 4);
  GN_STAMP(unit, 9);
}
template <int P, typename T, bool PAIRF>
__global__ __launch_bounds__(256, P == 1 ? GN_OCC_P1 : 2) void agg_x_kernel(GroupTable<AggGroup> Tb) {
  __shared__ f32x4 wring[kRingF4For<P>];
  run_with_fallback<P>([&](auto pc, ovf_t& ovf) { agg_x_body<decltype(pc)::value, T, PAIRF>(Tb, wring, ovf); });
}

// ---- A5 typed MLP, bf16 storage, TWO row blocks per wave (large launches, one wave per row-block pair) --------------
// A P = 1 sub-step is ONE 32-cycle MFMA against one 1-KiB weight operand: with one row block per wave the LDS operand
// read, the chunk boundary (barrier, staged writes) and the scalar bookkeeping are paid per MFMA and the matrix pipe
// idles for more than half of the time (tools/microbench/substep_rate.hip: 52 % at two workgroups per CU).  Here
// every operand read feeds two MFMAs (the same weights against rows [64w, 64w+32) and [64w+32, 64w+64)): 82 % in the
// same microbenchmark at two workgroups per CU — which is why the kernel must fit 256 registers:
//   * no per-type temporary: feat += W2k (relu(W1k eo + b1k) ef_k) with ef_k >= 0 applied BEFORE the packed ReLU, and
//     the b2 term  sum_k ef_k b2k  as fp32 MFMAs (32x32x2: two types per instruction) ahead of the stream;
//   * bias tiles form one sequence over all types (b1 is [K][128]): two tiles ahead, no per-type restart.
// Stream order per type as everywhere (A0 A1 B0 A2 B1 A3 B2 B3, 8 sub-steps = one chunk per pair); V_t (scale,
// convert, ReLU of hidden tile t) runs in the shadow of the MFMA group that precedes B_t: A_{t+1}, or B_2 for t = 3.
template <typename T>
__global__ __launch_bounds__(256, 2) void agg_rb2_kernel(GroupTable<AggGroup> Tb, int stage_bytes) {
  constexpr int RB = 2;
  using WS = WStream<1, 2>;
  ovf_t ovf_unused = {0ull, 0};      // (the range vote belongs to the two-part fp16 path)
  __shared__ f32x4 wring[WS::kRingF4];
  extern __shared__ __align__(16) unsigned char ori_dyn[];      // staged ori rows of the pairwise gather
  const int lwg = gn_uniform(gn_xcd_logical(Tb.xs, blockIdx.x));
  if (lwg < 0) return;
  const int gi = find_group(Tb, lwg);
  const gn_agg_group_t G = Tb.g[gi].a;
  const int rows = G.rows, K = G.K;
  const int wave = wave_id();
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int blk0 = ((lwg - Tb.first_wg[gi]) * 4 + wave) * RB;
  RowBlock rb[RB];
#pragma unroll
  for (int b = 0; b < RB; ++b) rb[b] = row_block(rows, blk0 + b);   // (a block past the rows: clamped loads, no stores)
  WS ws;
  ws.begin(G.W12x, wring, lane, wave, K * 32 / WS::CH);
  // pairwise gather (unordered pairs): the scenes' ori rows from an LDS stage when they fit (block-uniform)
  const int nodes_max = (G.eo == nullptr && G.H == nullptr && G.sym) ? pool_stage_nodes(128 * RB, G.E, G.N) : 0;
  const bool staged = nodes_max > 0 && (size_t)nodes_max * PoolStage<T>::kPitch * sizeof(T) <= (size_t)stage_bytes;
  T* s_ori = reinterpret_cast<T*>(ori_dyn);
  int node0 = 0;
  if (staged) {
    const int r0 = (lwg - Tb.first_wg[gi]) * (128 * RB);
    node0 = ori_stage_fill<T>(G, r0, min(rows - 1, r0 + 128 * RB - 1), s_ori);
    __syncthreads();
  }
  Parts<1> xi[RB][2][2];
  const float* efrow[RB];
  f32x16 out[RB][2];
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    f32x16 in[2];
    if (G.eo != nullptr)
      load_rows<2>(reinterpret_cast<const T*>(G.eo), GN_FEAT, rb[b].row_ld, h, in);
    else if (staged)
      gather_pair_staged<T>(G, rb[b].row_ld, h, s_ori, node0, in);
    else
      gather_rows<T>(G, rb[b].row_ld, h, in);
    make_parts_tiles<1, 2>(in, xi[b], ovf_unused);
    efrow[b] = G.edge_feat + (size_t)rb[b].row_ld * K;
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[b][o][r] = 0.f;
  }
  // sum_k ef_k b2k: lane (i, hh) of the A operand carries b2[k + hh][32 o + i], lane (j, hh) of B ef[row j][k + hh]
  if (G.b2 != nullptr) {
#pragma unroll 1
    for (int k = 0; k < K; k += 2) {
      const int kk = k + h;
      const bool on = kk < K;
      const float a0 = on ? G.b2[kk * 64 + (lane & 31)] : 0.f;
      const float a1 = on ? G.b2[kk * 64 + 32 + (lane & 31)] : 0.f;
#pragma unroll
      for (int b = 0; b < RB; ++b) {
        const float e = on ? efrow[b][kk] : 0.f;
        out[b][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, e, out[b][0], 0, 0, 0);
        out[b][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, e, out[b][1], 0, 0, 0);
      }
    }
  }
  const float* b1 = G.b1;
  const int last_tile = 4 * K - 1;
  f32x16 bias_n = load_bias_tile(b1, h);                         // tile 0 of type 0
  f32x16 bias_nn = load_bias_tile(b1 + 32 * min(1, last_tile), h);
  float ef[RB];
#pragma unroll
  for (int b = 0; b < RB; ++b) ef[b] = efrow[b][0];
  f32x16 hidn[RB];
  // A_t: hidden tile t of the current type, all row blocks (positions inside the type, compile-time)
  auto A = [&](auto fence, int s0, int next_bias_tile) {
#pragma unroll
    for (int b = 0; b < RB; ++b) hidn[b] = bias_n;
    bias_n = bias_nn;
    bias_nn = load_bias_tile(b1 + 32 * min(next_bias_tile, last_tile), h);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      ws.template step_rb<RB, decltype(fence)::value>(
          s0 + u, [&](int b) -> const Parts<1>& { return xi[b][u >> 1][u & 1]; }, [&](int b) -> f32x16& { return hidn[b]; });
  };
  auto Bm = [&](auto fence, int s0, const Parts<1> (&xh)[RB][2]) {
#pragma unroll
    for (int u = 0; u < 4; ++u)                       // [W2(0,t) hf0, hf1, W2(1,t) hf0, hf1]
      ws.template step_rb<RB, decltype(fence)::value>(
          s0 + u, [&](int b) -> const Parts<1>& { return xh[b][u & 1]; }, [&](int b) -> f32x16& { return out[b][u >> 1]; });
  };
  auto V = [&](Parts<1> (&xh)[RB][2]) {               // consumes hidn
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      f32x16 cur = hidn[b];
#pragma unroll
      for (int r = 0; r < 16; ++r) cur[r] *= ef[b];
      make_parts_relu(cur, 0, xh[b][0]);
      make_parts_relu(cur, 1, xh[b][1]);
    }
  };
  auto shadow = [&]() {                               // 4 sub-steps x RB MFMAs with V's VALU spread between them
#pragma unroll
    for (int i = 0; i < 4 * RB; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
      if (i % RB == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  using Fence = std::integral_constant<bool, true>;
  using NoFence = std::integral_constant<bool, false>;
#pragma unroll 1
  for (int k = 0; k < K; ++k) {
    const int kn = k + 1 < K ? k + 1 : k;
    float efn[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) efn[b] = efrow[b][kn];
    const int t0 = 4 * k;
    Parts<1> xa[RB][2], xb[RB][2];
    A(Fence{}, 0, t0 + 2);                // A0 (bias tiles t0+2 .. ride two ahead)
    V(xa);                                // V0 in the shadow of A1
    A(NoFence{}, 4, t0 + 3);
    shadow();
    Bm(Fence{}, 8, xa);                   // B0
    V(xb);                                // V1 | A2
    A(NoFence{}, 12, t0 + 4);
    shadow();
    Bm(Fence{}, 16, xb);                  // B1
    V(xa);                                // V2 | A3
    A(NoFence{}, 20, t0 + 5);
    shadow();
    V(xb);                                // V3 | B2
    Bm(NoFence{}, 24, xa);
    shadow();
    Bm(Fence{}, 28, xb);                  // B3
#pragma unroll
    for (int b = 0; b < RB; ++b) ef[b] = efn[b];
  }
  T* feat = reinterpret_cast<T*>(G.feat);
#pragma unroll
  for (int b = 0; b < RB; ++b) store_rows<2>(feat, GN_FEAT, rb[b].row, h, rb[b].live, out[b]);
}

// ---- A4 on bf16 storage, two row blocks per wave (large launches) — see agg_rb2_kernel for the why -----------------
// Same image and stream order as edge_x_kernel<1>: pair A (64 -> 128 -> 64: z), pair B (64 -> 256 -> logits | factor),
// then the Gumbel-softmax epilogue per row block.  The bias tiles of both pairs ride ONE two-ahead pipeline (the last
// two loads of pair A fetch the first two tiles of pair B).
template <typename T>
__global__ __launch_bounds__(256, 2) void edge_rb2_kernel(GroupTable<gn_edge_group_t> Tb, float tau, unsigned long long seed,
                                                          const unsigned long long* __restrict__ offset_dev, int pool_bytes) {
  constexpr int RB = 2;
  using WS = WStream<1, 2>;
  ovf_t ovf_unused = {0ull, 0};      // (the range vote belongs to the two-part fp16 path)
  __shared__ f32x4 wring[WS::kRingF4];
  extern __shared__ __align__(16) unsigned char pool_dyn[];      // staged x' / pq rows of the pairwise pooling
  const int lwg = gn_uniform(gn_xcd_logical(Tb.xs, blockIdx.x));
  if (lwg < 0) return;
  const int gi = find_group(Tb, lwg);
  const gn_edge_group_t G = Tb.g[gi];
  const int rows = G.rows, K = G.K;
  const int wave = wave_id();
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int blk0 = ((lwg - Tb.first_wg[gi]) * 4 + wave) * RB;
  RowBlock rb[RB];
#pragma unroll
  for (int b = 0; b < RB; ++b) rb[b] = row_block(rows, blk0 + b);
  const float* bi0 = G.bias;
  const float* bi1 = G.bias + 128;
  const float* bd0 = G.bias + 192;
  const float* bd1 = G.bias + 448;
  WS ws;
  ws.begin(G.Wx, wring, lane, wave, 80 / WS::CH);
  // fused node -> edge pooling of unordered pairs: from the scenes' node rows staged in LDS when they fit
  const int nodes_max = G.edges == nullptr && G.pool_H == nullptr && G.sym_N > 0
                            ? pool_stage_nodes(128 * RB, gn_pair_count(G.pool_N), G.pool_N) : 0;
  const bool staged = nodes_max > 0 && PoolStage<T>::bytes(nodes_max) <= (size_t)pool_bytes;   // (block-uniform)
  T* s_xp = reinterpret_cast<T*>(pool_dyn);
  T* s_pq = s_xp + (size_t)nodes_max * PoolStage<T>::kPitch;
  int node0 = 0;
  if (staged) {
    const int r0 = (lwg - Tb.first_wg[gi]) * (128 * RB);
    node0 = pool_stage_fill<T>(G, r0, min(rows - 1, r0 + 128 * RB - 1), s_xp, s_pq);
    __syncthreads();
  }
  Parts<1> xi[RB][2][2];
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    f32x16 in[2];
    if (G.edges != nullptr)
      load_rows<2>(reinterpret_cast<const T*>(G.edges), GN_FEAT, rb[b].row_ld, h, in);
    else if (staged)
      pooled_rows_staged<T>(G, rb[b].row_ld, h, s_xp, s_pq, node0, in);
    else
      pooled_rows<T>(G, rb[b].row_ld, h, in);  // fused node -> edge pooling
    make_parts_tiles<1, 2>(in, xi[b], ovf_unused);
  }
  f32x16 bias_n = load_bias_tile(bi0, h), bias_nn = load_bias_tile(bi0 + 32, h);
  f32x16 hidn[RB];
  // A: one hidden tile for all row blocks at stream position s0; nb: the bias tile loaded for two tiles later
  auto A = [&](auto fence, int s0, const float* nb) {
#pragma unroll
    for (int b = 0; b < RB; ++b) hidn[b] = bias_n;
    bias_n = bias_nn;
    bias_nn = load_bias_tile(nb, h);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      ws.template step_rb<RB, decltype(fence)::value>(
          s0 + u, [&](int b) -> const Parts<1>& { return xi[b][u >> 1][u & 1]; }, [&](int b) -> f32x16& { return hidn[b]; });
  };
  auto V = [&](Parts<1> (&xh)[RB][2]) {               // consumes hidn
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      make_parts_relu(hidn[b], 0, xh[b][0]);
      make_parts_relu(hidn[b], 1, xh[b][1]);
    }
  };
  using Fence = std::integral_constant<bool, true>;
  using NoFence = std::integral_constant<bool, false>;
  // ---- pair A: 64 -> 128 -> 64 (positions 0 .. 31) ----
  {
    f32x16 z[RB][2];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      z[b][0] = load_bias_tile(bi1, h);
      z[b][1] = load_bias_tile(bi1 + 32, h);
    }
    auto Bm = [&](auto fence, int s0, const Parts<1> (&xh)[RB][2]) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        ws.template step_rb<RB, decltype(fence)::value>(
            s0 + u, [&](int b) -> const Parts<1>& { return xh[b][u & 1]; }, [&](int b) -> f32x16& { return z[b][u >> 1]; });
    };
    auto shadow = [&]() {
#pragma unroll
      for (int i = 0; i < 4 * RB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        if (i % RB == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    Parts<1> xa[RB][2], xb[RB][2];
    A(Fence{}, 0, bi0 + 64);
    V(xa);
    A(NoFence{}, 4, bi0 + 96);
    shadow();
    Bm(Fence{}, 8, xa);
    V(xb);
    A(NoFence{}, 12, bd0);               // (bias tiles 0, 1 of pair B)
    shadow();
    Bm(Fence{}, 16, xb);
    V(xa);
    A(NoFence{}, 20, bd0 + 32);
    shadow();
    V(xb);
    Bm(NoFence{}, 24, xa);
    shadow();
    Bm(Fence{}, 28, xb);
#pragma unroll
    for (int b = 0; b < RB; ++b) make_parts_tiles<1, 2>(z[b], xi[b], ovf_unused);
  }
  // ---- pair B: 64 -> 256 -> (logits | factor) (positions 32 .. 79: A_t 4 sub-steps, B_t 2) ----
  f32x16 lg[RB];
#pragma unroll
  for (int b = 0; b < RB; ++b) lg[b] = load_bias_tile(bd1, h);
  {
    auto Bm = [&](auto fence, int s0, const Parts<1> (&xh)[RB][2]) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        ws.template step_rb<RB, decltype(fence)::value>(
            s0 + u, [&](int b) -> const Parts<1>& { return xh[b][u & 1]; }, [&](int b) -> f32x16& { return lg[b]; });
    };
    auto shadowA = [&]() {
#pragma unroll
      for (int i = 0; i < 4 * RB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        if (i % RB == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    auto shadowB = [&]() {
#pragma unroll
      for (int i = 0; i < 2 * RB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);
        if (i % RB == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    Parts<1> xa[RB][2], xb[RB][2];
    auto tile_ptr = [&](int t) { return bd0 + 32 * (t < 7 ? t : 7); };
    A(Fence{}, 32, tile_ptr(2));          // A0
    V(xa);
    A(NoFence{}, 36, tile_ptr(3));        // V0 | A1
    shadowA();
    Bm(Fence{}, 40, xa);                  // B0
    V(xb);
    A(NoFence{}, 42, tile_ptr(4));        // V1 | A2
    shadowA();
    Bm(Fence{}, 46, xb);                  // B1
    V(xa);
    A(NoFence{}, 48, tile_ptr(5));        // V2 | A3
    shadowA();
    Bm(Fence{}, 52, xa);                  // B2
    V(xb);
    A(NoFence{}, 54, tile_ptr(6));        // V3 | A4
    shadowA();
    Bm(Fence{}, 58, xb);                  // B3
    V(xa);
    A(NoFence{}, 60, tile_ptr(7));        // V4 | A5
    shadowA();
    Bm(Fence{}, 64, xa);                  // B4
    V(xb);
    A(NoFence{}, 66, tile_ptr(7));        // V5 | A6
    shadowA();
    Bm(Fence{}, 70, xb);                  // B5
    V(xa);
    A(NoFence{}, 72, tile_ptr(7));        // V6 | A7
    shadowA();
    V(xb);                                // V7 | B6
    Bm(NoFence{}, 76, xa);
    shadowB();
    Bm(Fence{}, 78, xb);                  // B7
  }
  // ---- epilogue, one row block after the other (as edge_x_kernel) ----
  const unsigned long long pbase = G.philox_offset + (offset_dev ? *offset_dev : 0ull);
  auto epilogue = [&](const RowBlock& r, const f32x16& lgb) {
    long long o1 = r.row_ld, o2 = r.row_ld;
    bool diag = true;
    if (G.sym_N > 0) {
      const int N = G.sym_N, Pn = gn_pair_count(N);
      const int bb = r.row_ld / Pn, p = r.row_ld - bb * Pn;
      int i, j;
      gn_pair_decode(p, N, i, j);
      o1 = (long long)bb * N * N + i * N + j;
      o2 = (long long)bb * N * N + j * N + i;
      diag = i == j;
    }
    float u1[8], u2[8];
    fetch_uniforms(G.U, pbase, seed, o1, K, h, u1);
    if (G.sym_N > 0) fetch_uniforms(G.U, pbase, seed, o2, K, h, u2);
    float facv = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (feat_of(q, h) == K) facv = lgb[q];
    facv += __shfl_xor(facv, 32, GN_WAVE);
    const float sig = 1.f / (1.f + expf(-facv));
    float d1[8], d2[8];
    gumbel_softmax_row<true>(lgb, u1, K, tau, h, d1);
    if (G.sym_N > 0) gumbel_softmax_row<true>(lgb, u2, K, tau, h, d2);
    if (r.live) {
      float* frow = G.edge_feat + (size_t)r.row * K;
      T* dist = reinterpret_cast<T*>(G.dist);
      if (G.sym_N == 0) {
        T* drow = dist + (size_t)r.row * K;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int f = feat_of(q, h);
          if (f < K) {
            st1(drow + f, d1[q]);
            frow[f] = sig * d1[q];
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int f = feat_of(q, h);
          if (f < K) {
            if (dist != nullptr) {
              st1(dist + (size_t)o1 * K + f, d1[q]);
              if (!diag) st1(dist + (size_t)o2 * K + f, d2[q]);
            }
            frow[f] = diag ? 2.f * (sig * d1[q]) : sig * d1[q] + sig * d2[q];
          }
        }
      }
    }
  };
  epilogue(rb[0], lg[0]);
  epilogue(rb[1], lg[1]);
}

// ---- A6 / closing MLP on the bf16 cores: y = W1 relu(W0 x + b0) + b1, dout <= 64 ---------------------------------
// Image, hidden-tile-major: per hidden tile t the tiles [W0(t, in 0..IT-1), W1(0..OT-1, t)].  Input rows read from
// x or formed on the fly (fused scatter, IT == 4) exactly as in mlp2_kernel.
// (occupancy: the fused scatter of the three-part kernel keeps ~200 registers in flight — one workgroup per CU; the
// bf16-storage kernel fits two with 128 inputs, three with 64)
template <int P, typename T, int IT, int HT, int OT>
__device__ __forceinline__ void mlp2_x_body(const GroupTable<gn_mlp2_group_t>& Tb, int rows, int dout, int ldy, int N,
                                            float divisor, f32x4* wring, ovf_t& ovf) {
  using WS = WStream<P>;
  const int lwg = gn_uniform(gn_xcd_logical(Tb.xs, blockIdx.x));
  if (lwg < 0) return;
  const int gi = find_group(Tb, lwg);             // (first_wg[g] = g * workgroups per group)
  const int blk = (lwg - Tb.first_wg[gi]) * 4 + wave_id();
  const gn_mlp2_group_t G = Tb.g[gi];
  const RowBlock rb = row_block(rows, blk);      // (a wave past the rows works on a clamped row, stores nothing)
  const int lane = rb.lane, h = rb.h;
  const int unit = lwg * 4 + wave_id();
  GN_STAMP(unit, 0);
  GN_STAMP(unit, 8);
  constexpr int kSub = HT * (2 * IT + 2 * OT);
  static_assert(kSub % WS::CH == 0, "the image must be a whole number of chunks");
  const float* b0 = G.bias;
  const float* b1 = G.bias + 32 * HT;
  const f32x16 hid0 = load_bias_tile(b0, h);
  f32x16 out[OT];
#pragma unroll
  for (int o = 0; o < OT; ++o) out[o] = load_bias_tile(b1 + 32 * o, h);
  WS ws;
  Parts<P> xi[IT][2];
  {
    // the input rows (a gather over the previous kernel's output: long latencies) are requested first; the first
    // chunks of the weight stream arrive in their shadow
    f32x16 in[IT];
    mlp2_rows<IT, T>(G, rb.row_ld, h, N, divisor, in);
    const void* img = pick_image<P>(G.Wx, G.Wh);
    if constexpr (P == 2) ovf.wf |= image_flag(img, kSub);
    ws.begin(img, wring, lane, wave_id(), kSub / WS::CH);
    if (G.in_out != nullptr) store_rows<IT>(G.in_out, 32 * IT, rb.row, h, rb.live, in);   // kept for the backward
    make_parts_tiles<P, IT>(in, xi, ovf);
  }
  GN_STAMP(unit, 1);
  layer_pair<P, IT, OT, HT, P == 1>(ws, 0, 0, xi, hid0, b0, h, out, ovf, [&](int t, f32x16& hid) {
    if constexpr (P != 1) {
      relu16(hid);
      if (G.hid_out != nullptr && rb.live) store_tile(G.hid_out + (size_t)rb.row * (32 * HT) + 32 * t + 4 * h, hid);
    }
  });
  GN_STAMP(unit, 2);
  if (rb.live) {
#pragma unroll
    for (int o = 0; o < OT; ++o) store_out_tile(reinterpret_cast<T*>(G.y), rb.row, ldy, dout, o, h, out[o]);
  }
  GN_STAMP(unit, 4);
  GN_STAMP(unit, 9);
}
template <int P, typename T, int IT, int HT, int OT>
__global__ __launch_bounds__(256, P == 1 ? (IT == 4 ? 2 : 3) : (IT == 4 ? 1 : 2)) void mlp2_x_kernel(GroupTable<gn_mlp2_group_t> Tb, int rows, int dout, int ldy,
                                                        int N, float divisor) {
  __shared__ f32x4 wring[kRingF4For<P>];
  run_with_fallback<P>([&](auto pc, ovf_t& ovf) {
    mlp2_x_body<decltype(pc)::value, T, IT, HT, OT>(Tb, rows, dout, ldy, N, divisor, wring, ovf);
  });
}


// =====================================================================================================================
// Small launches of the closing MLP: the 4 waves of a workgroup SHARE one 32-row block (mlp2_xs_kernel).
// With fewer row blocks than SIMDs (B*N = 5.6 k rows per module at B = 512: 704 blocks for 4 modules on 1024 SIMDs) a
// wave that owns a whole chain sits alone on its SIMD and the launch lasts as long as one wave's dependent chain
// (fused-scatter prologue + HT serial layer-pair slices): 23 us for ~4 us of matrix work.  Here the hidden tiles of the
// layer pair are dealt over the 4 waves of the row block (wave w takes tiles w, w+4, ...), each wave walks only ITS
// slices of the (unchanged, pipeline-ordered) weight image through a private register ring, the input operands are
// formed once per workgroup (wave w forms input tile w) and exchanged through LDS as bf16 parts, and the partial outputs
// meet in LDS: 4x the waves, each with a quarter of the chain, three workgroups per CU so that the launch is one round.
// Measured at B = 512 (MI355X, same-session A/B, GN_MLP2_XS = 0 / 1): launch 22.4 -> 14.0 us, single-stream forward
// 0.158 -> 0.149 ms, overlapped throughput unchanged (5.03 vs 5.00 M scenes/s).
// The same split of the NODE stage (chain workgroups of 4 waves per row block beside the unchanged per-node typed
// layer) was built and measured too and is not kept: launch 23.4 -> 22.4 us, overlapped throughput -3 % (4x the weight
// traffic from L2: a workgroup of one row block streams the whole image for 32 rows instead of 128) — the node stage's
// critical path is its 132 long "A" workgroups, not the chain.
// =====================================================================================================================

template <typename T>
__device__ __forceinline__ void add_row_tile(const T* __restrict__ src, float w, int h, f32x16& a) {   // src: row + 32*tile
  const T* p = src + 4 * h;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = ld4(p + 8 * q);
    a[4 * q + 0] = fmaf(w, v[0], a[4 * q + 0]);
    a[4 * q + 1] = fmaf(w, v[1], a[4 * q + 1]);
    a[4 * q + 2] = fmaf(w, v[2], a[4 * q + 2]);
    a[4 * q + 3] = fmaf(w, v[3], a[4 * q + 3]);
  }
}
template <typename T>
__device__ __forceinline__ void load_row_tile(const T* __restrict__ src, int h, f32x16& a) {          // src: row + 32*tile
  const T* p = src + 4 * h;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = ld4(p + 8 * q);
    a[4 * q + 0] = v[0];
    a[4 * q + 1] = v[1];
    a[4 * q + 2] = v[2];
    a[4 * q + 3] = v[3];
  }
}

// ONE 32-feature tile of the MLP's input row in the MFMA layout, for the cases without a gather worth the name:
// x[row, 32*tile ..]; tile 2/3 of the fused scatter's input = ori[b,n] / divisor; tile 0/1 for ORDERED pairwise edge
// rows (2N members, the C ABI's plain form; the engine uses unordered pairs).  Hyper and unordered-pair groups go through
// scatter_tile_lines below.
template <typename T>
__device__ __forceinline__ void mlp2_rows_tile(const gn_mlp2_group_t& G, int row, int h, int N, float divisor, int IT,
                                               int tile, f32x16& in) {
  if (G.x != nullptr) {
    load_row_tile(reinterpret_cast<const T*>(G.x) + (size_t)row * (IT * 32) + 32 * tile, h, in);
    return;
  }
  if (tile >= 2) {
    load_row_tile(reinterpret_cast<const T*>(G.ori) + (size_t)row * GN_FEAT + 32 * (tile - 2), h, in);
#pragma unroll
    for (int r = 0; r < 16; ++r) in[r] = in[r] / divisor;
    return;
  }
  const int E = G.E;
  if (E == 0) {      // feat already holds H^T feat per node (node form of the typed aggregation)
    load_row_tile(reinterpret_cast<const T*>(G.feat) + (size_t)row * GN_FEAT + 32 * tile, h, in);
#pragma unroll
    for (int r = 0; r < 16; ++r) in[r] = in[r] / divisor;
    return;
  }
  const int b = row / N, n = row - b * N;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const T* fb = reinterpret_cast<const T*>(G.feat) + (size_t)b * E * GN_FEAT + 32 * tile;
  for (int j = 0; j < N; ++j) {
    add_row_tile(fb + (size_t)(n * N + j) * GN_FEAT, 1.f, h, acc);
    add_row_tile(fb + (size_t)(j * N + n) * GN_FEAT, 1.f, h, acc);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) in[r] = acc[r] / divisor;
}

// The fused scatter of one 32-row block, one 128-byte feature tile, in LINE layout: lane L = (sub = L / 8, piece = L % 8)
// accumulates the 16-byte piece `piece` of the rows rg * 8 + sub (rg = 0..3), so that one load instruction covers 8 whole
// 128-byte lines.  In the MFMA layout (lane (j, h) = row j, features 8 q + 4 h + c) every load instruction touches 32
// lines for 16 bytes per lane and every line is looked up by four instructions: the CU's L1 handles one line per clock,
// and the gather of a row block (11 members x 4 loads x 32 lines per tile) kept it busy for ~12 k cycles, longer than
// the block's matrix work.  Same members, same order, same fmaf per (row, feature): the sums are bit-identical; the
// layout changes once, through `scratch` (32 rows x 36 floats, private to the wave: write, barrier, read).
// Returns false (nothing done) for shapes the per-lane form handles (ordered pairwise edges).
constexpr int kLinePitch = 36;
// NRG row groups (of 8 rows) starting at rg0: the whole tile (rg0 = 0, NRG = 4: one wave per tile), or half of it
// (NRG = 2: two waves per tile — half the loads per wave, so ALL members of a row fit one batch of requests: one L2
// round trip instead of two for the N = 11 graphs)
template <typename T, bool hyper, int NRG>
__device__ __forceinline__ void scatter_tile_lines_(const gn_mlp2_group_t& G, int blk, int rows, int N, float divisor,
                                                    int tile, int lane, int rg0, float* __restrict__ scratch) {
  const int E = G.E;
  const int sub = lane >> 3, piece = lane & 7;
  const int cnt = hyper ? E : N;
  unsigned fo[NRG], ho[NRG];                     // feat / H offsets (elements) of this lane's rows
  int nn[NRG];
#pragma unroll
  for (int rg = 0; rg < NRG; ++rg) {
    const int r = min(blk * 32 + (rg0 + rg) * 8 + sub, rows - 1);
    const int b = r / N;
    nn[rg] = r - b * N;
    fo[rg] = (unsigned)b * E * GN_FEAT + 32 * tile + 4 * piece;
    ho[rg] = (unsigned)b * E * N + nn[rg];
  }
  const T* feat = reinterpret_cast<const T*>(G.feat);
  f32x4 acc[NRG];
#pragma unroll
  for (int rg = 0; rg < NRG; ++rg) acc[rg] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int MB = NRG == 4 ? 6 : 11;          // members per batch (their rows are requested together)
  for (int m0 = 0; m0 < cnt; m0 += MB) {
    f32x4 v[MB][NRG];
    float w[MB][hyper ? NRG : 1];
#pragma unroll
    for (int u = 0; u < MB; ++u) {
      const int m = min(m0 + u, cnt - 1);
#pragma unroll
      for (int rg = 0; rg < NRG; ++rg) {
        const int idx = hyper ? m : gn_pair_index(nn[rg], m, N);
        v[u][rg] = ld4(feat + fo[rg] + (size_t)idx * GN_FEAT);
        if constexpr (hyper) w[u][rg] = G.H[ho[rg] + (size_t)m * N];
      }
    }
#pragma unroll
    for (int u = 0; u < MB; ++u)
      if (m0 + u < cnt) {
#pragma unroll
        for (int rg = 0; rg < NRG; ++rg)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[rg][c] = fmaf(hyper ? w[u][hyper ? rg : 0] : 1.f, v[u][rg][c], acc[rg][c]);
      }
  }
#pragma unroll
  for (int rg = 0; rg < NRG; ++rg) {
    f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = acc[rg][c] / divisor;
    *reinterpret_cast<f32x4*>(scratch + ((rg0 + rg) * 8 + sub) * kLinePitch + 4 * piece) = o;
  }
}
template <typename T, int NRG = 4>
__device__ __forceinline__ bool scatter_tile_lines(const gn_mlp2_group_t& G, int blk, int rows, int N, float divisor,
                                                   int tile, int lane, float* __restrict__ scratch, int rg0 = 0) {
  if (G.H != nullptr) scatter_tile_lines_<T, true, NRG>(G, blk, rows, N, divisor, tile, lane, rg0, scratch);
  else if (G.sym) scatter_tile_lines_<T, false, NRG>(G, blk, rows, N, divisor, tile, lane, rg0, scratch);
  else return false;
  return true;
}
// ... and back in the MFMA layout (after a barrier)
__device__ __forceinline__ void read_tile_lines(const float* __restrict__ scratch, int lane, f32x16& in) {
  const int j = lane & 31, h = lane >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(scratch + j * kLinePitch + 8 * q + 4 * h);
    in[4 * q + 0] = v[0];
    in[4 * q + 1] = v[1];
    in[4 * q + 2] = v[2];
    in[4 * q + 3] = v[3];
  }
}

// ---- closing MLP, 4 waves per row block: y = W1 relu(W0 x + b0) + b1, dout <= 64 ----------------------------------------
// Same image, same inputs (incl. the fused scatter) and the same optional kept activations as mlp2_x_kernel.
template <int P, int IT, int OT>
constexpr int kXsLdsF4 = (IT * 2 * P * 64 > 4 * 4 * OT * 64) ? IT * 2 * P * 64 : 4 * 4 * OT * 64;
template <int P, typename T, int IT, int HT, int OT>
__device__ __forceinline__ void mlp2_xs_body(const GroupTable<gn_mlp2_group_t>& Tb, int rows, int dout, int ldy, int N,
                                             float divisor, f32x4* lds, float (*lines)[32 * kLinePitch], ovf_t& ovf) {
  static_assert(HT % 4 == 0, "hidden tiles are dealt over 4 waves");
  constexpr int TPW = HT / 4, NA = 2 * IT, NB = 2 * OT, L = TPW * (NA + NB);
  const int lwg = gn_uniform(gn_xcd_logical(Tb.xs, blockIdx.x));
  if (lwg < 0) return;
  const int gi = find_group(Tb, lwg);
  const gn_mlp2_group_t G = Tb.g[gi];
  const int wave = wave_id();
  const RowBlock rb = row_block(rows, lwg - Tb.first_wg[gi]);
  const int lane = rb.lane, h = rb.h;
  const int unit = lwg * 4 + wave;
  GN_STAMP(unit, 0);
  GN_STAMP(unit, 8);
  const float* b0 = G.bias;
  const float* b1 = G.bias + 32 * HT;
  const void* image = pick_image<P>(G.Wx, G.Wh);
  if constexpr (P == 2) ovf.wf |= image_flag(image, HT * (NA + NB));
  const f32x4* img = reinterpret_cast<const f32x4*>(image) + lane;
  const f32x4* segA[TPW];
  const f32x4* segB[TPW];
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    const int t = wave + 4 * tt;
    segA[tt] = img + (size_t)pipe_off_A(t, NA, NB) * P * 64;
    segB[tt] = img + (size_t)pipe_off_B(t, HT, NA, NB) * P * 64;
  }
  // consumption order of this wave: A of all its tiles, then B of all its tiles
  auto at = [&](int s) -> const f32x4* {
    return s < TPW * NA ? segA[s / NA] + (s % NA) * P * 64 : segB[(s - TPW * NA) / NB] + ((s - TPW * NA) % NB) * P * 64;
  };
  PStream<P, L> ps;
  // fused scatter (IT == 4, waves 0 / 1): accumulated in line layout, handed over through `lines`
  bool via_lines = false;
  f32x16 in;
  if constexpr (IT == 4) {
    if (G.x == nullptr) {
      // all four waves gather: wave w takes rows [16 (w >> 1), 16 (w >> 1) + 16) of feature tile w & 1 (the two halves of a
      // tile meet in `lines`); waves 2 / 3 request their ori tiles first, so those loads fly beside the gather
      via_lines = (G.H != nullptr || G.sym != 0) && G.E > 0;      // (block-uniform: the group's shape decides)
      if (via_lines) {
        if (wave >= 2) mlp2_rows_tile<T>(G, rb.row_ld, h, N, divisor, IT, wave, in);
        scatter_tile_lines<T, 2>(G, lwg - Tb.first_wg[gi], rows, N, divisor, wave & 1, lane, lines[wave & 1], 2 * (wave >> 1));
        __syncthreads();
      }
    }
  }
  if (wave < IT) {
    if (via_lines && wave < 2) read_tile_lines(lines[wave], lane, in);
    else if (!(via_lines && wave >= 2)) mlp2_rows_tile<T>(G, rb.row_ld, h, N, divisor, IT, wave, in);
    ps.begin(at);       // (behind the gather, whose batches of member rows need the registers; in flight across the exchange)
    if (G.in_out != nullptr && rb.live) store_tile(G.in_out + (size_t)rb.row * (32 * IT) + 32 * wave + 4 * h, in);
    put_parts<P>(lds, wave, lane, in, ovf);
  } else {
    ps.begin(at);
  }
  __syncthreads();
  // layer 1: the input operands are read from LDS one sub-step ahead of their MFMAs (not held in registers: IT = 4
  // tiles of three parts would be 96 of them, and the launch wants 3 workgroups per CU to run in one round)
  auto xop = [&](int u, Parts<P>& x) {             // operand of sub-step u of an A phase: input tile u/2, half u%2
#pragma unroll
    for (int p = 0; p < P; ++p) x.p[p] = __builtin_bit_cast(bf16x8, lds[(u * P + p) * 64 + lane]);
  };
  Parts<P> xa[2];
  xop(0, xa[0]);
  GN_STAMP(unit, 1);
  // accumulators start at zero and the bias tiles — requested before a phase, added behind it — arrive in its shadow
  // (held from the start they would be live across the gather and cost the third workgroup per CU)
  f32x16 hid[TPW], bt[TPW];
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    bt[tt] = load_bias_tile(b0 + 32 * (wave + 4 * tt), h);
#pragma unroll
    for (int r = 0; r < 16; ++r) hid[tt][r] = 0.f;
  }
  int s = 0;
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      if (s + 1 < TPW * NA) xop((u + 1) % NA, xa[(s + 1) & 1]);
      ps.step(s, at, xa[s & 1], hid[tt]);
      ++s;
    }
  f32x16 out[OT], bo[OT];
#pragma unroll
  for (int o = 0; o < OT; ++o) {
    if (wave == 0) bo[o] = load_bias_tile(b1 + 32 * o, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) out[o][r] = 0.f;
  }
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) hid[tt][r] += bt[tt][r];
    Parts<P> xh[2];
    if constexpr (P == 1) {
      make_parts_relu(hid[tt], 0, xh[0]);
      make_parts_relu(hid[tt], 1, xh[1]);
    } else {
      relu16(hid[tt]);
      if (G.hid_out != nullptr && rb.live)
        store_tile(G.hid_out + (size_t)rb.row * (32 * HT) + 32 * (wave + 4 * tt) + 4 * h, hid[tt]);
      make_parts<P>(hid[tt], 0, xh[0], ovf);
      make_parts<P>(hid[tt], 1, xh[1], ovf);
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      ps.step(s, at, xh[u & 1], out[u >> 1]);
      ++s;
    }
  }
  if (wave == 0) {
#pragma unroll
    for (int o = 0; o < OT; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[o][r] += bo[o][r];
  }
  GN_STAMP(unit, 2);
  __syncthreads();                                // every wave is past its reads of the exchanged operands
  put_partial<OT>(lds, wave, lane, out);
  __syncthreads();
  GN_STAMP(unit, 3);
  GN_STAMP(unit, 4);
  GN_STAMP(unit, 9);
  if (!rb.live) return;
  T* yrow = reinterpret_cast<T*>(G.y) + (size_t)rb.row * ldy;
  const bool vec = ((dout | ldy) & 3) == 0;
#pragma unroll
  for (int qq = 0; qq < OT; ++qq) {               // wave w finishes quads [w*OT, w*OT + OT) of the 4*OT
    const int q = wave * OT + qq;
    const f32x4 v = sum_partial<OT>(lds, q, lane);
    const int f = 32 * (q >> 2) + 8 * (q & 3) + 4 * h;
    if (vec) {
      if (f < dout) st4(yrow + f, v);
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (f + c < dout) st1(yrow + f + c, v[c]);
    }
  }
}
template <int P, typename T, int IT, int HT, int OT>
__global__ __launch_bounds__(256, HT == 4 ? 3 : 2) void mlp2_xs_kernel(GroupTable<gn_mlp2_group_t> Tb, int rows, int dout, int ldy, int N,
                                                          float divisor) {
  __shared__ f32x4 lds[kXsLdsF4<P == 2 ? 3 : P, IT, OT>];
  __shared__ __align__(16) float lines[2][32 * kLinePitch];     // layout change of the fused scatter (waves 0 / 1)
  run_with_fallback<P>([&](auto pc, ovf_t& ovf) {
    mlp2_xs_body<decltype(pc)::value, T, IT, HT, OT>(Tb, rows, dout, ldy, N, divisor, lds, lines, ovf);
  });
}

}  // namespace
