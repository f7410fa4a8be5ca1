/*
 * CPU oracle (plain C) for the index part of the path: top-k hyperedge incidence.
 * TEST INFRASTRUCTURE ONLY — built by oracle/Makefile into oracle/_build/, loaded only by tests.
 *
 * Restates MS_HGNN_hyper.init_adj_attention (model/MS_HGNN_batch.py:372-388):
 *   scale == N          -> H = ones(B, 1, N)
 *   else k = max(scale,1); idx = topk(corr, k, dim=2, largest); H = zeros(B,N,N).scatter(2, idx, 1)
 * with the tie rule the build defines: among equal values the LOWER index wins, and
 * NaN ranks above every number (torch.topk treats NaN as the largest value).  The selection is written
 * as k rounds of arg-max over the not-yet-taken columns — deliberately a different algorithm from the
 * rank count the HIP kernel uses, so the two check each other bit for bit.
 *
 * Pinned by tests/test_oracle_golden.py against every golden H generated from the reference.
 */
#include <math.h>
#include <stdlib.h>

static int better(float a, int ia, float b, int ib) { /* does (a, ia) beat (b, ib)? */
  const int na = isnan(a), nb = isnan(b);
  if (na || nb) return (na && !nb) || (na && nb && ia < ib);
  return a > b || (a == b && ia < ib);
}

/* returns 0, or -1 when scale > N (torch.topk: "selected index k out of range") */
int gn_oracle_topk_incidence(const float* corr, float* H, int B, int N, int scale) {
  if (scale > N) return -1;
  if (scale == N) {
    for (long i = 0; i < (long)B * N; ++i) H[i] = 1.0f;
    return 0;
  }
  const int k = scale < 1 ? 1 : scale;
  char* taken = (char*)malloc((size_t)N);
  for (long r = 0; r < (long)B * N; ++r) {
    const float* row = corr + r * N;
    float* hrow = H + r * N;
    for (int c = 0; c < N; ++c) {
      taken[c] = 0;
      hrow[c] = 0.0f;
    }
    for (int round = 0; round < k; ++round) {
      int best = -1;
      for (int c = 0; c < N; ++c)
        if (!taken[c] && (best < 0 || better(row[c], c, row[best], best))) best = c;
      taken[best] = 1;
      hrow[best] = 1.0f;
    }
  }
  free(taken);
  return 0;
}
