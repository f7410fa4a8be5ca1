/*
 * groupnet_hip.h — C ABI of libgroupnet_hip.so: the MI355X (gfx950) kernels of the
 * GroupNet MS-HGNN hot path.
 *
 * The reference (TaliMotzkin/GroupNet) is pure Python/PyTorch and has no FFI for this
 * path; each entry point below names the reference lines it replaces (paths relative to
 * the reference root).  A host binds them with ctypes/cffi/cgo/JNI: plain pointers and
 * ints only, no torch types.  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - every pointer is DEVICE memory owned by the caller, fp32, row-major, contiguous,
 *     16-byte aligned (torch allocations are); inputs are never written;
 *   - `stream` is a hipStream_t (NULL = the default stream); launchers enqueue and return,
 *     they never synchronise, allocate or free — safe to capture in a hipGraph;
 *   - return value: GN_OK (0) or a negative GN_ERR_* code; nothing is launched on error;
 *   - B scenes, N agents (nodes), E hyperedges, K edge types, D = 64 feature width.
 *   - "packed" weights are produced by gn_pack_linear_f32 from an nn.Linear weight
 *     (out x in, row-major); the layout is private to the library.  A kernel that applies several
 *     layers takes ONE buffer holding their packed images back to back, in the order stated.
 *   - bf16 twins (BASELINE config 4: bf16 storage, fp32 accumulate): every stage of the forward
 *     has a *_bf16 entry point taking the SAME descriptor structs.  In a twin the fields marked [T] point to
 *     bf16 tensors (2 bytes per element, 8-byte aligned rows) instead of fp32; incidence H, uniforms U,
 *     edge_feat, biases and the small attention vectors stay fp32; weights are given as the one-part bf16
 *     image (`Wx` fields, gn_split_bf16_f32 with parts = 1).  Products run on v_mfma_f32_32x32x16_bf16 with
 *     fp32 accumulation; a layer's fp32 result is rounded to bf16 (nearest even) when it becomes the next
 *     layer's operand or is stored.  The twins are forward-only (the optional training outputs must be NULL).
 *   - `Wx` images: the fp32-accurate bf16-core path of the *_f32 entry points.  A group that carries `Wx`
 *     (gn_split_bf16_f32 with parts = 3: every weight as three bf16 parts) is evaluated with six bf16
 *     part-products per product on the bf16 matrix cores — as accurate as fp32 accumulation, faster than the
 *     fp32 matrix cores; without it the launch falls back to v_mfma_f32_32x32x2_f32 on the plain packed stream.
 *     Either all groups of a launch carry it or none.
 *   - `Wh` images (`Wh`, `WAh`, `W2h`, `W12h`; ABI 30): the SAME tile stream split into TWO fp16 parts per weight
 *     (gn_split_bf16_f32 with parts = 2) — the "f16x3" path: three v_mfma_f32_32x32x16_f16 part-products per product,
 *     fp32-accurate (3e-7 of max|result| at K = 256), about twice the rate of the six-product bf16 path.  fp16 has a
 *     narrow exponent, so a group that carries `Wh` MUST also carry the corresponding `Wx` image: a workgroup that
 *     meets an operand beyond 65000 in magnitude (or a flagged weight image) repeats its rows on the bf16 path inside
 *     the same launch.  Either all groups of a launch carry `Wh` or none.  The image is followed by a 16-byte FLAG
 *     word the caller zero-initialises once (the split raises it when a weight does not fit fp16).
 *   - Order of the tiles in every bf16-/fp16-core image of a LAYER PAIR  out += W1 act(W0 x)  ("pipeline order", the
 *     order the kernels consume them; ops.pipeline_order): with A_t = the first-layer tiles [W0(t, in 0..IT-1)] that
 *     produce hidden tile t and B_t = the second-layer tiles [W1(0..OT-1, t)] that consume it:
 *         A0 A1 B0 A2 B1 ... A(HT-1) B(HT-2) B(HT-1)
 *     (NOT hidden-tile-major [A_t B_t]: A_{t+1} is issued between A_t and B_t so that the splitting of hidden tile t
 *     runs in the shadow of the matrix pipe).  Sub-step offset of A_t: t == 0 ? 0 : NA + (t-1)(NA+NB); of B_t:
 *     t < HT-1 ? 2 NA + t (NA+NB) : HT NA + (HT-1) NB, with NA = 2 IT, NB = 2 OT sub-steps (two per 32x32 tile).
 *   - Node form of the pairwise typed aggregation (gn_agg_group_t.node_form; ABI 33): edge_aggregation.forward reads the
 *     per-edge feature only as H^T feat (MS_HGNN_batch.py:267) and the type weighting and layer 2 are linear, so the
 *     launch can evaluate H^T feat directly, layer 2 once per node instead of once per pair — see the field.  gn_mlp2_*
 *     takes that aggregate with E = 0.
 *   - Closing stage in the aggregation launch (gn_agg_group_t.y ...; ABI 34): the same launch also applies the closing
 *     MLP to cat(H^T feat, ori) / divisor — one launch per message-passing stage fewer, rows bit-identical.
 */
#ifndef GROUPNET_HIP_H
#define GROUPNET_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gn_stream_t; /* hipStream_t */

#define GN_OK 0
#define GN_ERR_NULL -1        /* a required pointer is NULL */
#define GN_ERR_SHAPE -2       /* a size is <= 0 or unsupported (see each function) */
#define GN_ERR_K_RANGE -3     /* top-k: k > N  (torch.topk raises RuntimeError here) */
#define GN_ERR_ALIGN -4       /* a pointer is not 16-byte aligned */
#define GN_ERR_LAUNCH -5      /* hipGetLastError() reported a launch failure */
#define GN_ERR_LDS -6         /* the problem does not fit the 160 KiB LDS tile of the kernel */

#define GN_FEAT 64            /* h_dim == hdim_extend == 64 (MS_HGNN_batch.py:72,292) */
#define GN_MAX_TYPES 16       /* edge types K <= 16 (reference: 6 pairwise, 10 hyper) */
#define GN_MAX_SCALES 8

/* Pitched device-to-device copy of `rows` rows of `width_bytes` (multiples of 16 bytes, 16-byte aligned): the column block of
 * the concatenated feature tensor (model/GroupNet_nba.py:301-309) that a rank ships into its all-gather staging bank. */
int gn_copy_2d(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width_bytes, int rows,
               gn_stream_t stream);

/* ABI version of this header; bumped on any signature change. */
int gn_abi_version(void);
/* Short static description of a GN_ERR_* code. */
const char* gn_strerror(int code);

/* ---- A0: cosine affinity -------------------------------------------------------------
 * corr[b] = q q^T with q = f / max(||f||_2, 1e-12) per agent.
 * Replaces model/GroupNet_nba.py:284-286 (F.normalize + matmul).
 * f (B,N,D) -> corr (B,N,N).  D must be a multiple of 4, D <= 1024. */
int gn_affinity_f32(const float* f, float* corr, int B, int N, int D, gn_stream_t stream);

/* ---- A1: top-k hyperedge incidence -----------------------------------------------------
 * Replaces MS_HGNN_hyper.init_adj_attention, model/MS_HGNN_batch.py:372-388
 * (torch.topk + zeros().scatter()).  For each of n_scales group sizes k_s:
 *   k_s == N : H_s is (B,1,N), all ones;
 *   else     : k = max(k_s,1); H_s is (B,N,N) with H_s[b,i,c] = 1 iff c is among the k
 *              largest entries of corr[b,i,:]  (ties: lowest index wins; NaN ranks first);
 *   k_s > N  : GN_ERR_K_RANGE.
 * H_list / k_list are HOST arrays of n_scales entries (1 <= n_scales <= GN_MAX_SCALES);
 * one pass over corr serves every scale. */
int gn_topk_incidence_f32(const float* corr, float* const* H_list, const int* k_list, int n_scales,
                          int B, int N, gn_stream_t stream);

/* A9: exhaustive hyperedge search — replaces MS_HGNN_hyper.init_adj_attention_listall,
 * model/MS_HGNN_batch.py:390-414 (with its constant candidate table all_combs, :313-326, never
 * materialised): scale == N -> H (B,1,N) all ones; else s = max(scale,1) and H (B,N,N) with row i = the
 * group of s agents containing i that maximises sum_{a,b in group} corr[a][b] over all C(N-1,s-1)
 * candidates (first maximum in the lexicographic candidate order of torch.combinations; NaN ranks first).
 * N <= 64 and C(N-1,s-1) < 2^31, else GN_ERR_SHAPE; scale > N -> GN_ERR_K_RANGE. */
int gn_listall_incidence_f32(const float* corr, float* H, int B, int N, int scale, gn_stream_t stream);

/* A0+A1 fused: f -> corr (may be NULL: not written) and every H_s, without re-reading corr
 * from HBM.  Same contracts as the two functions above; N*(N+D)*4 bytes must fit in LDS.
 * `extras` (may be NULL) lets this first launch of a multiscale forward also produce what the
 * caller's concatenations need, so that no copy kernels follow (model/GroupNet_nba.py:296-311):
 *   f_out   : f is also written to f_out with row stride f_out_ld floats (the first D columns of
 *             the concatenated feature tensor);
 *   H_cat   : every H_s is also written into the (B, sum_s E_s, N) concatenation, scale order;
 *   counter : *counter += counter_add by one thread (advances the device Philox position once per
 *             forward, see gn_edge_mlp_gumbel_f32) — ordered before every later launch of the stream;
 *   x_raw   : the agent embedding itself is computed here: f[b,n] = M x_raw[b,n] + c[n]
 *             with x_raw (B,N,x_dim), M (D,x_dim), c (N,D) — the embedding front-end of
 *             PastEncoder.forward (model/GroupNet_nba.py:269-280), which in eval mode is one affine map
 *             per agent slot; the `f` argument is then ignored and f is ALSO written contiguously to
 *             f_contig (B,N,D), the h_states input of the modules. */
typedef struct {
  float* f_out;
  int f_out_ld;
  float* H_cat;
  unsigned long long* counter;
  unsigned long long counter_add;
  const float* x_raw;
  int x_dim;
  const float* M;
  const float* c;
  float* f_contig;
} gn_block_extras_t;
int gn_affinity_topk_f32(const float* f, float* corr, float* const* H_list, const int* k_list,
                         int n_scales, int B, int N, int D, const gn_block_extras_t* extras,
                         gn_stream_t stream);
/* twin: f, extras->f_out and extras->H_cat are bf16 (H values 0/1 are exact in bf16); the normalisation, the
 * affinity and the ranking run in fp32 on the bf16 inputs, corr (may be NULL) and every H_s stay fp32; the
 * embedding front-end (extras->x_raw) is not part of the twin. */
int gn_affinity_topk_bf16(const void* f, float* corr, float* const* H_list, const int* k_list,
                          int n_scales, int B, int N, int D, const gn_block_extras_t* extras,
                          gn_stream_t stream);

/* ---- weight packing -------------------------------------------------------------------
 * Number of floats of the packed image of an (out x in) nn.Linear weight. */
size_t gn_packed_elems(int out_features, int in_features);
/* W (out x in, row-major) -> Wp (gn_packed_elems floats).  The rows/cols are zero-padded to
 * multiples of 32.  `col_offset`/`ld` let a sub-block of a wider matrix be packed:
 * element (o,i) is read from W[o*ld + col_offset + i]. */
int gn_pack_linear_f32(const float* W, float* Wp, int out_features, int in_features, int ld,
                       int col_offset, gn_stream_t stream);

/* Many weights in one launch: segment i drops the source block src (rows x cols, row-major, ld) scaled by
 * `scale` into the packed image starting at dst, at (place_r, place_c) of the image's virtual matrix whose
 * packed rows hold IT 32-column tiles (the layout of gn_pack_linear_f32); IT == 0 writes a plain row-major
 * destination instead: dst[(place_r + r) * dst_ld + place_c + c] = scale * src[r][c] (vectors: rows = 1;
 * concatenated weights for the backward's GEMMs).  Nothing outside the block is written: zero the
 * destination once.
 * `segs` is a DEVICE array (built once per module; the refresh after an optimizer step is then one fill +
 * one launch, capturable in a hipGraph); max_elems = the largest rows*cols of any segment. */
typedef struct {
  const float* src;
  float* dst;
  int ld, rows, cols, place_r, place_c, IT;
  float scale;
  int dst_ld;
} gn_pack_seg_t;
int gn_pack_segments_f32(const gn_pack_seg_t* segs, int n_segs, int max_elems, gn_stream_t stream);

/* ---- grouped launches --------------------------------------------------------------------
 * The 1+S modules of one multiscale forward (pairwise + one hyper module per scale) are
 * independent and differ only in weights, incidence and edge count.  Every stage below
 * therefore takes an array of `n_groups` descriptors (HOST memory, 1 <= n_groups <=
 * GN_MAX_GROUPS) and serves all of them with ONE launch, so that a launch always carries
 * enough workgroups to fill the 256 CUs; a single module is the n_groups == 1 case.
 * Groups must not alias each other's outputs. */
#define GN_MAX_GROUPS 10

/* ---- A3 (first half): node MLP + attention projections -----------------------------------
 * x' = MLP_{64->256->64}(x)            (node2edge_start_mlp, MS_HGNN_batch.py:125,358)
 * pq = x' Wpq^T + bpq  (64 wide)       the node-side halves of attention_mlp layer 0
 *                                      (MS_HGNN_batch.py:131-134,362-365), see gn_node2edge_f32.
 * W    = packed images [W0 (256x64) | W1 (64x256) | Wpq (64x64)] back to back (one weight stream,
 *        consumed in this order); bias = [b0 (256) | b1 (64) | bpq (64)].
 * x (rows,64) -> xp (rows,64), pq (rows,64) per group. */
typedef struct {
  const float* x;   /* [T] */
  const float* W;
  const float* bias;
  float* xp;        /* [T] */
  float* pq;        /* [T] */
  float* hid_out;   /* optional (training): the hidden activations relu(W0 x + b0) (rows, 256) */
  const void* Wx;   /* optional bf16-core image of the chain (72 sub-steps): the layer pair W0 (256x64) / W1 (64x256)
                       in PIPELINE order (header comment: A_t = [W0(t,in0), W0(t,in1)], B_t = [W1(0,t), W1(1,t)],
                       A0 A1 B0 A2 B1 ... A7 B6 B7), then [Wpq(0,in0), Wpq(0,in1), Wpq(1,in0), Wpq(1,in1)].
                       With it W may be NULL. */
  /* optional, with Wx only: the per-node first layer of the typed aggregation MLP of the pairwise graph in the
   * SAME launch (it reads the same node rows; see gn_node_linear_f32): A = WA x + bA, A (rows, KA*128) [T],
   * WAx = bf16-core image of the packed (KA*128 x 64) matrix, bA (KA*128) fp32.  A == NULL: not computed. */
  const void* WAx;
  const float* bA;
  float* A;         /* [T] */
  int KA;
  const void* Wh;   /* optional fp16 two-part image of the same chain stream as Wx (f16x3 path; needs Wx too) */
  const void* WAh;  /* ... of the same stream as WAx (required with Wh when A != NULL) */
} gn_node_group_t;
int gn_node_mlp_f32(const gn_node_group_t* groups, int n_groups, int rows, gn_stream_t stream);
int gn_node_mlp_bf16(const gn_node_group_t* groups, int n_groups, int rows, gn_stream_t stream);
/* The node stage of a forward's FIRST round needs only the agent features, not the incidences: A0 + A1 of the same
 * forward (gn_affinity_topk_*: model/GroupNet_nba.py:284-286, model/MS_HGNN_batch.py:372-388) can ride in the same
 * launch.  `job` = the arguments of gn_affinity_topk_* (f may be NULL with the embedding front-end in `extras`); its B
 * scenes become the launch's TAIL workgroups, dispatched behind the node stage's own and filling the CUs its short
 * workgroups leave early — one launch and one launch boundary fewer (measured at B = 512, N = 11: 7.3 us + 1.3 us).
 * Results are those of the two separate launches, bit for bit.  Needs the groups' bf16-core images (`Wx`) and a scene
 * tile of at most gn_affinity_tail_lds_limit() bytes (N (D + 4 + x_dim) 4 + N N 8 + 8), else GN_ERR_SHAPE / GN_ERR_LDS. */
typedef struct {
  const void* f;             /* [T] (B, N, D) agent features, or NULL with extras->x_raw */
  float* corr;               /* optional (B, N, N) */
  float* const* H_list;      /* n_scales incidence outputs, as gn_affinity_topk_f32 */
  const int* k_list;
  int n_scales;
  int B, N, D;
  const gn_block_extras_t* extras;   /* optional */
} gn_affinity_job_t;
int gn_node_mlp_affinity_f32(const gn_node_group_t* groups, int n_groups, int rows, const gn_affinity_job_t* job,
                             gn_stream_t stream);
int gn_node_mlp_affinity_bf16(const gn_node_group_t* groups, int n_groups, int rows, const gn_affinity_job_t* job,
                              gn_stream_t stream);
size_t gn_affinity_tail_lds_limit(void);

/* ---- A3 (second half): attention-weighted node -> edge pooling ---------------------------
 * Replaces the rest of node2edge, MS_HGNN_batch.py:127-141 / 359-370:
 *   e0 = H x';  att[e,n] = w2 . relu(P_n + (H Qn)_e) + b2  with P = pq[:, :32] (bias folded),
 *   Qn = pq[:, 32:];  W = softmax_n(att * H) * H;  edges = W x'.
 * H == NULL selects the pairwise graph of MS_HGNN_oridinary (E must be N*N; edge e = i*N + j has
 * weight 1 on i and on j, 2 when i == j; MS_HGNN_batch.py:118,124,143-160) without ever
 * materialising it.  xp, pq (B,N,64); H (B,E,N) or NULL; w2 (32 floats, device); edges (B,E,64).
 * Symmetric pairwise form (H == NULL, sym = 1): edge (i,j) and edge (j,i) of the pairwise graph carry
 * the same pooled feature, so only the N(N+1)/2 unordered pairs are produced: E = N(N+1)/2, row
 * p(i,j) = i*N - i(i-1)/2 + (j-i) for i <= j.  Every pairwise stage below has the matching form. */
typedef struct {
  const float* xp;   /* [T] */
  const float* pq;   /* [T] */
  const float* H;
  const float* w2;
  float* edges;      /* [T] */
  const float* b2;   /* device pointer to the scalar bias of attention layer 1 (the parameter itself: no
                        host read-back, so a training step stays capturable in a hipGraph) */
  int E;
  int sym;
} gn_n2e_group_t;
int gn_node2edge_f32(const gn_n2e_group_t* groups, int n_groups, int B, int N, gn_stream_t stream);
int gn_node2edge_bf16(const gn_n2e_group_t* groups, int n_groups, int B, int N, gn_stream_t stream);

/* ---- A4: per-edge MLPs + Gumbel-softmax edge typing ----------------------------------------
 * Replaces MLP_dict_softmax.forward + gumbel_softmax, MS_HGNN_batch.py:41-53,446-520:
 *   z = MLP_{64->128->64}(edges); logits = MLP_{64->128->K}(z); fac = sigmoid(MLP_{64->128->1}(z));
 *   g = -log(1e-10 - log(U + 1e-10)); dist = softmax((logits + g) / tau); edge_feat = fac * dist.
 * W = the packed images of Wi0 = init_MLP.0 (128x64), Wi1 = init_MLP.1 (64x128), Wd0 (256x64) and
 *   Wd1 (32x256), where Wd0 = [MLP_distribution.layers.0 ; MLP_factor.layers.0] and Wd1 is the block
 *   matrix whose rows 0..K-1 are [MLP_distribution.layers.1, 0] and row K is [0, MLP_factor.layers.1],
 *   cut into hidden tiles T (8 steps of 256 floats) and second-layer slices S and ordered as the kernel
 *   consumes them:  Wi0/Wi1: T0 T1 S0 T2 S1 T3 S2 S3 (S_t = Wi1 tiles (0,t),(1,t): 8 steps);
 *   Wd0/Wd1: T0 T1 S0 T2 S1 ... T7 S6 S7 (S_t = Wd1 tile (0,t): 4 steps); then 8 steps of padding;
 * bias = [128 | 64 | 256 | 32] in the same order (bd1: K logits biases, then the factor bias, zeros).
 * edges (rows,64), U (rows,K) uniforms in [0,1) -> edge_feat (rows,K), dist (rows,K).  K <= 15.
 * U == NULL: the uniforms are generated inside the kernel — element row*K + k is element
 * philox_offset (+ *offset_dev if not NULL) + row*K + k of the Philox stream `seed`, exactly what
 * gn_philox_uniform_f32 would have written into U.
 * Symmetric pairwise form (sym_N = N > 0): rows = B*N(N+1)/2 unordered pairs.  The MLPs run once per
 * pair; the Gumbel softmax runs for BOTH ordered edges (i,j) and (j,i) with their own uniforms
 * (U / Philox positions are those of the ordered (B,N*N,K) tensor).  dist (B*N*N,K) receives the ordered
 * distributions (may be NULL when the caller does not need them); edge_feat (rows,K) receives
 * fac*(dist_ij + dist_ji) (2*fac*dist_ii on the diagonal) — exactly the weight the pair carries in the
 * edge->node sum, where both ordered edges meet the same typed MLP output. */
typedef struct {
  const float* edges;   /* [T] */
  const float* U;
  const float* W;
  const float* bias;
  float* edge_feat;
  float* dist;          /* [T] */
  unsigned long long philox_offset;
  int rows;
  int K;
  int sym_N;
  /* optional (training), all (rows, .): hidden of init_MLP (128), z (64), hidden of MLP_distribution |
   * MLP_factor (256), and the 32-wide (logits | factor pre-activation | 0) tile — what the backward needs and
   * the kernel otherwise keeps in registers */
  float* keep_z1;
  float* keep_z;
  float* keep_dh1;
  float* keep_lgf;
  const void* Wx;    /* optional bf16-core image (gn_split_bf16_f32) of the fp32 tile stream of the four layers, two
                        layer pairs in PIPELINE order (header comment) — pair A (init_MLP, 4 hidden tiles): A_t =
                        [Wi0(t,in0), Wi0(t,in1)], B_t = [Wi1(0,t), Wi1(1,t)]; pair B (8 hidden tiles): A_t =
                        [Wd0(t,in0), Wd0(t,in1)], B_t = [Wd1(0,t)] — 40 tiles = 80 sub-steps.  With it W may be NULL. */
  /* Fused node->edge pooling (bf16-core kernels only, i.e. with Wx): edges == NULL and xp != NULL — every row of
   * `edges` is formed inside the kernel exactly as gn_node2edge_* would have written it (MS_HGNN_batch.py:127-141,
   * 359-370) and never touches HBM.  xp / pq: (B*N, 64) outputs of gn_node_mlp_*; w2 (32), b2 (1): attention layer 1;
   * pool_N = N; pool_H == NULL: the implicit pairwise graph (unordered pairs when sym_N = N, else the N*N ordered
   * edges); pool_H != NULL: (rows, N) incidence of a hyper module with pool_E hyperedges per scene, N <= 16. */
  const float* xp;      /* [T] */
  const float* pq;      /* [T] */
  const float* pool_H;
  const float* w2;
  const float* b2;
  int pool_N;
  int pool_E;
  const void* Wh;    /* optional fp16 two-part image of the same stream as Wx (f16x3 path; needs Wx too) */
} gn_edge_group_t;
int gn_edge_mlp_gumbel_f32(const gn_edge_group_t* groups, int n_groups, float tau, unsigned long long seed,
                           const unsigned long long* offset_dev, gn_stream_t stream);
int gn_edge_mlp_gumbel_bf16(const gn_edge_group_t* groups, int n_groups, float tau, unsigned long long seed,
                            const unsigned long long* offset_dev, gn_stream_t stream);

/* ---- A5: hyperedge aggregation --------------------------------------------------------------
 * gather: eo = H ori            (edge_aggregation.forward, MS_HGNN_batch.py:263)
 * H == NULL: pairwise graph (E = N*N), eo[(i,j)] = ori_i + ori_j; with sym = 1 only the
 * E = N(N+1)/2 unordered pairs.   ori (B,N,64) -> eo (B,E,64). */
typedef struct {
  const float* ori;   /* [T] */
  const float* H;
  float* eo;          /* [T] */
  int E;
  int sym;
} gn_gather_group_t;
int gn_agg_gather_f32(const gn_gather_group_t* groups, int n_groups, int B, int N, gn_stream_t stream);
int gn_agg_gather_bf16(const gn_gather_group_t* groups, int n_groups, int B, int N, gn_stream_t stream);

/* typed MLP: feat = sum_k edge_feat[:,k] * MLP^k_{64->128->64}(eo)   (MS_HGNN_batch.py:262,264-265)
 * W: for each type k the packed images [agg_mlp[k].layers.0 (128x64) | agg_mlp[k].layers.1 (64x128)],
 * types back to back; b1 (K,128); b2 (K,64).  eo (rows,64), edge_feat (rows,K) -> feat (rows,64).
 * Fused gather (eo == NULL): the kernel forms its input rows itself from ori (B,N,64) exactly as
 * gn_agg_gather_f32 would — row r = b*E + e is sum_n H[b,e,n] ori[b,n] (H (B,E,N)), or for the
 * pairwise graph (H == NULL) ori_i + ori_j with (i,j) the ordered edge (E = N*N) or the unordered
 * pair (sym = 1, E = N(N+1)/2) — so eo never exists in HBM.  rows must equal B*E.
 * Pair form (A != NULL; pairwise graph, symmetric rows, E = N(N+1)/2): A (B*N, K*128) holds the
 * per-node first layer W1k ori + b1k/2 (gn_node_linear_f32); row p = (i <= j) computes
 * sum_k edge_feat[p,k] * (W2k relu(A[i,k] + A[j,k]) + b2k).  W then is, per type, the packed (64 x 128)
 * image re-ordered hidden-tile-major ((t, o) instead of (o, t)); b1 is unused. */
typedef struct {
  const float* eo;     /* [T] */
  const float* edge_feat;
  const float* W;
  const float* b1;
  const float* b2;
  float* feat;         /* [T] */
  int rows;
  int K;
  const float* ori;    /* [T] */
  const float* H;
  int E;
  int N;
  int sym;
  const float* A;
  const void* W2x;    /* optional, pair form (fp32 entry point only): bf16-core image of layer 2 of every type
                         (gn_split_bf16_f32 of the hidden-tile-major W image: per type and hidden tile t the tiles
                         [W2k(0,t), W2k(1,t)], 16 sub-steps per type) */
  const void* W12x;   /* optional, two-layer form: bf16-core image of both layers, per type one layer pair in PIPELINE
                         order (header comment): A_t = [W1k(t, in 0), W1k(t, in 1)], B_t = [W2k(out 0, t), W2k(out 1, t)],
                         A0 A1 B0 A2 B1 A3 B2 B3 (32 sub-steps per type).  With W2x / W12x, W may be NULL. */
  const void* W2h;    /* optional fp16 two-part images of the same streams as W2x / W12x (f16x3 path; need those too) */
  const void* W12h;
  int node_form;      /* pair form with W2x, N <= 16, K <= 12 only.  Nothing downstream of edge_aggregation.forward reads the
                         per-edge feature: MS_HGNN_batch.py:267 consumes it as H^T feat, and both the type weighting and
                         layer 2 are linear, so they commute with that sum.  With node_form = 1 the kernel evaluates
                           S[n,k] = sum_j edge_feat[p(n,j),k] relu(A[n,k] + A[j,k]),  c[n,k] = sum_j edge_feat[p(n,j),k]
                           (H^T feat)[n] = sum_k (W2k S[n,k] + b2k c[n,k])
                         i.e. layer 2 once per NODE (B*N rows) instead of once per pair (B*N(N+1)/2 rows), and `feat`
                         receives (H^T feat) (B*N, 64) — what gn_mlp2_f32 takes with E = 0.  rows stays B*E. */
  /* Fused closing stage (ABI 34; fp32 entry point on the fp16/bf16-core images, N <= 16; for EVERY group of a launch or
   * for none): with y != NULL the workgroups that finish a scene's aggregate also apply the closing MLP
   *   y[b*N + n] = W1 relu(W0 cat(H^T feat, ori)[n] / divisor + b0) + b1     (MS_HGNN_batch.py:267, :120 / :355, :220-229)
   * — what gn_mlp2_f32 would have computed from `feat` in a launch of its own — and `feat` is neither written nor
   * read (may be NULL).  m2x / m2h / m2bias: gn_mlp2_group_t.Wx / Wh / bias of a 128 -> 128 -> dout MLP, 32 < dout <= 64; y rows
   * have stride ldy.  Needs ori and, for a hyper group, H / E / N with the fused gather (eo == NULL, E <= 16); a
   * pairwise group must use node_form.  Groups too large for the per-scene workgroup shapes (more than 767 row blocks
   * of edge rows) are refused: use the two launches. */
  const void* m2x;
  const void* m2h;
  const float* m2bias;
  float* y;
  int ldy;
  int dout;
  float divisor;
} gn_agg_group_t;
int gn_agg_mlp_f32(const gn_agg_group_t* groups, int n_groups, gn_stream_t stream);
/* twin: two-layer form only (eo, or the fused gather from ori) — a per-node first layer stored in bf16 would
 * cost as much VALU / LDS work per pair as the layer itself costs on the bf16 cores */
int gn_agg_mlp_bf16(const gn_agg_group_t* groups, int n_groups, gn_stream_t stream);
/* Packed fp32 32x32 weight tiles (gn_pack_linear_f32 layout, n_tiles of 1024 floats) -> bf16-core image: for
 * every tile and each of its two k-halves (one "sub-step") `parts` bf16 parts in the A-operand order of
 * v_mfma_f32_32x32x16_bf16: out[(((tile*2 + half)*parts + part)*64 + lane)*8 + j] (16-bit words), where
 * element j of lane (m, h) is the weight of output m and k-feature 16*half + (j&3) + 8*(j>>2) + 4*h — the
 * order in which a lane's accumulator registers hold those features.  parts = 3: x = p1 + p2 + p3 (8 mantissa
 * bits each, round to nearest: the fp32-accurate path); parts = 1: p1 = x rounded to bf16 (the bf16 twins);
 * parts = 2: TWO fp16 parts x = hi + lo (the `Wh` images of the f16x3 path) — `out` then holds 16 more bytes behind
 * the image, the FLAG word (zero-initialised by the caller once), which the split raises when |x| > 65000. */
int gn_split_bf16_f32(const float* packed, void* out, int n_tiles, int parts, gn_stream_t stream);
/* The same for `n_jobs` images in ONE launch (job table in DEVICE memory; max_tiles = the largest n_tiles): what a
 * training step needs after its optimizer update — every weight image of every module, one launch after the one
 * gn_pack_segments_f32 launch over the concatenated segment tables — instead of one launch per image (38 launches
 * of ~4 us each per step of the multiscale block; the reference re-reads its nn.Parameters directly,
 * train_hyper_nba.py:107-118). */
typedef struct {
  const float* packed;
  void* out;
  int n_tiles;
  int reserved;
} gn_split_job_t;
int gn_split_bf16_batch_f32(const gn_split_job_t* jobs, int n_jobs, int max_tiles, int parts, gn_stream_t stream);

/* ---- A5, pairwise graph, layer 1 hoisted to the nodes -----------------------------------------
 * For the pairwise graph the typed MLP's input row is eo = ori_i + ori_j, so its first layer is
 * linear in the two nodes: W1k eo + b1k = (W1k ori_i + b1k/2) + (W1k ori_j + b1k/2).  The first layer
 * therefore runs once per NODE (N rows per scene instead of N(N+1)/2 pairs):
 *   y = W x + bias, x (rows,64) -> y (rows,dout), dout a multiple of 128 (W = packed (dout x 64)
 *   image; used with W = [W1_0; ...; W1_{K-1}], bias = b1/2 to produce A (B*N, K*128)); the pair form
 *   of gn_agg_mlp_f32 (field A) then applies relu and the second layer per unordered pair. */
int gn_node_linear_f32(const float* x, const float* W, const float* bias, float* y, int rows, int dout,
                       gn_stream_t stream);

/* scatter: out = cat(H^T feat, ori) / divisor     (MS_HGNN_batch.py:267; divisor = N gives the
 * division of edge2node :120,355, divisor = 1 the bare edge_aggregation.forward).
 * feat (B,E,64), ori (B,N,64) -> out (B,N,128).  H == NULL: pairwise (E = N*N); with sym = 1 feat
 * holds the E = N(N+1)/2 pair sums (see gn_edge_mlp_gumbel_f32) and node n adds the N pairs {n,j}. */
typedef struct {
  const float* feat;   /* [T] */
  const float* H;
  const float* ori;    /* [T] */
  float* out;          /* [T] */
  int E;
  int sym;
} gn_scatter_group_t;
int gn_agg_scatter_f32(const gn_scatter_group_t* groups, int n_groups, int B, int N, float divisor,
                       gn_stream_t stream);
int gn_agg_scatter_bf16(const gn_scatter_group_t* groups, int n_groups, int B, int N, float divisor,
                        gn_stream_t stream);

/* ---- A6 / generic two-layer MLP ---------------------------------------------------------------
 * y = W1 relu(W0 x + b0) + b1     (MLP.forward with one hidden layer, MS_HGNN_batch.py:220-229;
 * nmp_mlp_end 128->128->bottleneck and nmp_mlps[even] 128->128->64).
 * din in {64,128}, dh in {128,256}, dout >= 1 (the same for every group).
 * W = packed [W0 (dh x din) | W1 (dout x dh)]; bias = [b0 (dh) | b1 zero-padded to a multiple of 32].
 * x (rows,din) -> y (rows,dout) with row stride ldy >= dout floats (lets the result land in a
 * column block of a wider tensor, e.g. the concatenated per-scale features).
 * Fused scatter (x == NULL, din == 128): the kernel forms its input rows itself exactly as
 * gn_agg_scatter_f32 would — row b*N + n is cat(sum_e H[b,e,n] feat[b,e], ori[b,n]) / divisor
 * (H (B,E,N); H == NULL: the pairwise graph, E = N*N ordered edges or, sym = 1, E = N(N+1)/2 pair
 * sums) — so the (B,N,128) aggregate never exists in HBM.  rows must equal B*N.
 * E == 0 (x == NULL): `feat` already holds H^T feat per node, (rows, 64) — the output of gn_agg_mlp_f32's node form;
 * the input row is cat(feat[row], ori[row]) / divisor. */
typedef struct {
  const float* x;      /* [T] */
  const float* W;
  const float* bias;
  float* y;            /* [T] */
  const float* feat;   /* [T] */
  const float* H;
  const float* ori;    /* [T] */
  int E;
  int sym;
  float* in_out;    /* optional (training): the MLP's input rows as evaluated (rows, din) — with the fused
                       scatter that is cat(H^T feat, ori)/N, which otherwise never exists in memory */
  float* hid_out;   /* optional (training): the hidden activations relu(W0 x + b0) (rows, dh) */
  const void* Wx;   /* optional bf16-core image, one layer pair in PIPELINE order (header comment): A_t =
                       [W0(t, in 0..din/32-1)], B_t = [W1(0..ceil(dout/32)-1, t)]; used when dout <= 64 (else the
                       launch needs W).  With it W may be NULL. */
  const void* Wh;   /* optional fp16 two-part image of the same stream (f16x3 path; needs Wx too) */
} gn_mlp2_group_t;
int gn_mlp2_f32(const gn_mlp2_group_t* groups, int n_groups, int rows, int din, int dh, int dout, int ldy,
                int N, float divisor, gn_stream_t stream);
/* twin: dout <= 64 */
int gn_mlp2_bf16(const gn_mlp2_group_t* groups, int n_groups, int rows, int din, int dh, int dout, int ldy,
                 int N, float divisor, gn_stream_t stream);

/* ---- backward (training) building blocks — SURVEY.md §8f rank 2 -------------------------------
 * train_hyper_nba.py:116 back-propagates through the two modules.  The backward of the path is
 * assembled (groupnet_amd/backward.py) from these generic kernels plus the forward gather / scatter
 * kernels, which are each other's adjoints.  All pointers device fp32, row-major.
 *
 * gn_gemm_f32: C (M x N, ldc) = beta*C + alpha*op(A) op(B) [+ bias[n]] [relu] [zeroed where mask[m][n] <= 0];
 *   op(A) = A (M x K, lda) or, transA, A^T with A stored (K x M, lda); op(B) = B (K x N, ldb) or, transB,
 *   B^T with B stored (N x K, ldb).  With few output tiles and K >= 4096 (weight gradients dW = dY^T X)
 *   K is split over workgroups and partial sums are added atomically (then no bias/relu/mask).
 * gn_gumbel_bwd_f32: back through edge_feat = sigmoid(f) * dist, dist = softmax((logits + g)/tau)
 *   (model/MS_HGNN_batch.py:45-50): dist (rows,K), lgf (rows,ldl) whose column K is f, def = d edge_feat,
 *   gdist = d dist or NULL -> dlgf (rows,ldl): columns 0..K-1 = d logits, column K = d f, rest 0.
 *   sym_N > 0 (here and in gn_gumbel_ef_f32): the rows of lgf / def / dlgf / ef are the N(N+1)/2 unordered
 *   pairs of the pairwise graph per scene, dist / gdist stay per ordered edge (B, N*N, K): a pair row
 *   stands for the ordered edges (i,j) and (j,i) — ef is their sum (self-loop rows times diag_w: 2 gives the
 *   rows the forward's pair-form aggregation consumes, which fold H = 2 into ef), dlgf the sum of their
 *   gradients.
 * gn_node2edge_bwd_f32: back through gn_node2edge_f32 for an explicit H (B,E,N), or H == NULL for the
 *   implicit pairwise graph (E == N*N ordered edges, or with sym its N(N+1)/2 unordered pairs whose dedges
 *   are already summed over the two directions): given dedges (B,E,64) ADDS into dxp (B,N,64),
 *   dpq (B,N,64), dw2 (32), db2 (1) (zero them first).  One workgroup per scene with the scene's rows in
 *   LDS while they fit (N <= ~140); beyond that one wave per hyperedge with global atomics (explicit H only). */
/* Grouped form: n independent problems in one launch (all types of the typed aggregation MLP, all modules
 * of a multiscale block).  rs (optional) scales A's STORED rows: A_eff[r][:] = rs[r*rs_ld] * A[r][:];
 * colsum (optional, GN_GEMM_TRANS_A only) receives colsum[m] += sum_k A_eff[k][m] — the bias gradient next
 * to dW = dY^T X.  GN_GEMM_ACCUM: C += alpha*op(A)op(B) by atomic adds with K split over workgroups (C must
 * hold zeros / the running sum; no bias, relu, mask, beta). */
#define GN_GEMM_TRANS_A 1
#define GN_GEMM_TRANS_B 2
#define GN_GEMM_RELU 4
#define GN_GEMM_ACCUM 8
#define GN_GEMM_TRANS_C 16   /* with GN_GEMM_ACCUM: the product is added into C^T, i.e. C is (N x M, ldc) */
typedef struct {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* mask;
  const float* rs;
  float* colsum;
  int M, N, K, lda, ldb, ldc, ldmask, rs_ld;
  int flags;
  float alpha, beta;
} gn_gemm_desc_t;
int gn_gemm_grouped_f32(const gn_gemm_desc_t* descs, int n, gn_stream_t stream);
int gn_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                int transA, int transB, const float* bias, const float* mask, int ldmask, int relu,
                float alpha, float beta, gn_stream_t stream);
/* out (rows x cols, ldo) = alpha * a (rows x cols, lda) + beta * out — slices, scalings and sums of
 * gradients without leaving the library. */
int gn_axpby2d_f32(float* out, int ldo, const float* a, int lda, long long rows, int cols, float alpha,
                   float beta, gn_stream_t stream);
/* ef[r][k] = sigmoid(lgf[r][K]) * dist[r][k]: edge_feat of MLP_dict_softmax from dist and the factor
 * pre-activation (column K of lgf (rows, ldl)); ef has leading dimension ld_ef >= K (a multiple of 4 keeps
 * the GEMMs that read it on the vector path). */
int gn_gumbel_ef_f32(const float* dist, const float* lgf, float* ef, long long rows, int K, int ldl, int sym_N,
                     float diag_w, int ld_ef, gn_stream_t stream);
/* Middle of the typed aggregation MLP's backward (model/MS_HGNN_batch.py:264-265), all K types of a row at
 * once: T (rows, K*hid) holds dfeat W2cat, Hc (rows, K*hid) the hidden activations; writes
 * def[r][k] = <T[r,k,:], Hc[r,k,:]> + <dfeat[r], b2[k]> and T[r,k,:] <- ef[r][k] * T[r,k,:] * (Hc > 0). */
int gn_typed_bwd_f32(float* T, const float* Hc, const float* ef, int ld_ef, const float* dfeat, const float* b2,
                     float* def, long long rows, int K, int hid, gn_stream_t stream);
int gn_gumbel_bwd_f32(const float* dist, const float* lgf, const float* def, const float* gdist, float* dlgf,
                      long long rows, int K, int ldl, float tau, int sym_N, gn_stream_t stream);
/* The two per-module stages above for every module of a backward round in ONE launch each (descriptor arrays in HOST
 * memory, 1 <= n_groups <= GN_MAX_GROUPS; all modules over the same B scenes and N nodes): the single-module launches
 * take 13-88 us each and ran end to end, four per stage and training step.  gn_node2edge_bwd_grouped_f32 covers the
 * per-scene form only (the scene's rows fit in LDS, else GN_ERR_LDS: go module by module). */
typedef struct {
  const float* dist;
  const float* lgf;
  const float* def;
  const float* gdist;   /* or NULL */
  float* dlgf;
  long long rows;
  int K;
  int sym_N;
} gn_gumbel_bwd_group_t;
int gn_gumbel_bwd_grouped_f32(const gn_gumbel_bwd_group_t* groups, int n_groups, int ldl, float tau, gn_stream_t stream);
typedef struct {
  const float* xp;
  const float* pq;
  const float* H;       /* or NULL: the pairwise graph */
  const float* w2;
  const float* b2;
  const float* dedges;
  float* dxp;
  float* dpq;
  float* dw2;
  float* db2;
  int E;
  int sym;
} gn_n2e_bwd_group_t;
int gn_node2edge_bwd_grouped_f32(const gn_n2e_bwd_group_t* groups, int n_groups, int B, int N, gn_stream_t stream);
int gn_node2edge_bwd_f32(const float* xp, const float* pq, const float* H, const float* w2, const float* b2,
                         const float* dedges, float* dxp, float* dpq, float* dw2, float* db2, int B, int N,
                         int E, int sym, gn_stream_t stream);

/* ---- device noise (build's own; the reference draws torch.rand on the host) --------------------
 * U[i] = Philox4x32-10(counter = (i + offset) / 4, key = seed)[(i + offset) % 4] >> 8, scaled to
 * [0,1).  Lets a sharded run draw exactly the rows of the full-batch stream it owns.
 * offset_dev (may be NULL): a device counter ADDED to `offset` when the kernel runs, so that a
 * captured hipGraph draws fresh noise on every replay; gn_counter_add_u64 advances it in stream
 * order. */
int gn_philox_uniform_f32(float* U, size_t n, unsigned long long seed, unsigned long long offset,
                          const unsigned long long* offset_dev, gn_stream_t stream);
int gn_counter_add_u64(unsigned long long* counter, unsigned long long add, gn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GROUPNET_HIP_H */
