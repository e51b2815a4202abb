/*
 * nsm_oracle.c -- plain scalar C restatement of the per-pair arithmetic (TEST INFRASTRUCTURE).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it (see
 * oracle/__init__.py).  It restates, with deliberately different algorithms from the HIP kernels
 * (nested loops for set intersection, the O(n m) dynamic programme for LCS, every level scored
 * independently), what the reference computes per pair:
 *
 *   jaccard      intersection_vs_union   napkon_string_matching/compare/score_functions.py:6-13
 *   indel_ratio  fuzzy_match             compare/score_functions.py:20-27 (QRatio of rapidfuzz 2.1.x:
 *                                        ((1 - (la+lb-2 LCS)/(la+lb)) * 100) / 100, 0 if an operand is empty)
 *   *_levels     compare_terms           types/comparable_data.py:248-265
 *   category     categories_matching     types/comparable_data.py:464-476 (as bit masks)
 *
 * It is validated against the Python restatement (the .py files of oracle/), which is pinned to the reference's
 * own outputs (tests/golden/).  Operands are CSR arrays; strings are int32 code points.
 */
#include <stdint.h>
#include <stdlib.h>

typedef struct {
  double score;
  int32_t i;
  int32_t j;
} oracle_hit;

static int cat_ok(const uint64_t* lc, const uint64_t* rc, int i, int j, int mode) {
  if (mode == 0) return 1;
  const uint64_t a = lc[i], b = rc[j];
  if (a & b) return 1;
  return mode == 2 && a == 0 && b == 0;
}

/* |A n B| by nested loops; sets hold distinct ids. */
static int intersect(const int32_t* a, int na, const int32_t* b, int nb) {
  int k = 0;
  for (int x = 0; x < na; ++x)
    for (int y = 0; y < nb; ++y)
      if (a[x] == b[y]) {
        ++k;
        break;
      }
  return k;
}

/* returns 0 on 0/0 via *err (the reference raises ZeroDivisionError there) */
static double jaccard(const int32_t* a, int na, const int32_t* b, int nb, int* err) {
  const int k = intersect(a, na, b, nb);
  const int u = na + nb - k;
  if (u == 0) {
    *err = 1;
    return 0.0;
  }
  return (double)k / (double)u;
}

static int lcs_dp(const int32_t* a, int la, const int32_t* b, int lb, int* row) {
  for (int y = 0; y <= lb; ++y) row[y] = 0;
  for (int x = 1; x <= la; ++x) {
    int diag = 0; /* row[x-1][y-1] */
    for (int y = 1; y <= lb; ++y) {
      const int up = row[y];
      if (a[x - 1] == b[y - 1]) row[y] = diag + 1;
      else row[y] = up > row[y - 1] ? up : row[y - 1];
      diag = up;
    }
  }
  return row[lb];
}

static double indel_ratio(const int32_t* a, int la, const int32_t* b, int lb, int* row) {
  if (la == 0 || lb == 0) return 0.0 / 100.0;
  const int lcs = lcs_dp(a, la, b, lb, row);
  const volatile double maximum = (double)(la + lb);
  const volatile double dist = (double)(la + lb - 2 * lcs);
  const volatile double q = dist / maximum;
  const volatile double sim = 1.0 - q;
  const volatile double pct = sim * 100.0;
  return pct / 100.0;
}

#define EMIT(S, I, J)                         \
  do {                                        \
    if (n < cap) {                            \
      out[n].score = (S);                     \
      out[n].i = (int32_t)(I);                \
      out[n].j = (int32_t)(J);                \
    }                                         \
    ++n;                                      \
  } while (0)

/* RAW Jaccard grid.  Returns the number of hits (may exceed cap), or -1 at a 0/0 pair. */
long long oracle_jaccard_raw(const int32_t* lids, const int64_t* loff, int nl, const int32_t* rids,
                             const int64_t* roff, int nr, double thr, oracle_hit* out, long long cap) {
  long long n = 0;
  for (int i = 0; i < nl; ++i)
    for (int j = 0; j < nr; ++j) {
      int err = 0;
      const double s = jaccard(lids + loff[i], (int)(loff[i + 1] - loff[i]), rids + roff[j],
                               (int)(roff[j + 1] - roff[j]), &err);
      if (err) return -1;
      if (s >= thr) EMIT(s, i, j);
    }
  return n;
}

long long oracle_indel_raw(const int32_t* lcp, const int64_t* loff, int nl, const int32_t* rcp,
                           const int64_t* roff, int nr, double thr, oracle_hit* out, long long cap) {
  long long n = 0;
  int maxb = 0;
  for (int j = 0; j < nr; ++j)
    if (roff[j + 1] - roff[j] > maxb) maxb = (int)(roff[j + 1] - roff[j]);
  int* row = (int*)malloc(sizeof(int) * (size_t)(maxb + 1));
  for (int i = 0; i < nl; ++i)
    for (int j = 0; j < nr; ++j) {
      const double s = indel_ratio(lcp + loff[i], (int)(loff[i + 1] - loff[i]), rcp + roff[j],
                                   (int)(roff[j + 1] - roff[j]), row);
      if (s >= thr) EMIT(s, i, j);
    }
  free(row);
  return n;
}

/* Levels: item k owns the level sets/strings llev[k] .. llev[k+1]-1; set/string t spans off[t]..off[t+1].
 * use_indel = 0: Jaccard per level, 1: Indel ratio per level.
 * Returns hits, -1 at a 0/0 level pair, -2 at a zero-level item meeting a non-empty one (IndexError). */
long long oracle_levels(int use_indel, const int32_t* lval, const int64_t* loff, const int64_t* llev, int nl,
                        const int32_t* rval, const int64_t* roff, const int64_t* rlev, int nr,
                        const uint64_t* lcat, const uint64_t* rcat, int cat_mode, double thr,
                        oracle_hit* out, long long cap) {
  long long n = 0;
  int maxb = 0;
  for (int64_t t = 0; t < rlev[nr]; ++t)
    if (roff[t + 1] - roff[t] > maxb) maxb = (int)(roff[t + 1] - roff[t]);
  int* row = (int*)malloc(sizeof(int) * (size_t)(maxb + 1));
  for (int i = 0; i < nl; ++i)
    for (int j = 0; j < nr; ++j) {
      if (!cat_ok(lcat, rcat, i, j, cat_mode)) continue;
      const int ll = (int)(llev[i + 1] - llev[i]), lr = (int)(rlev[j + 1] - rlev[j]);
      const int steps = ll > lr ? ll : lr;
      double score = 0.0, factor = 1.0;
      for (int s = 1; s <= steps; ++s) {
        if (ll == 0 || lr == 0) {
          free(row);
          return -2;
        }
        const int64_t a = llev[i] + (s < ll - 1 ? s : ll - 1);
        const int64_t b = rlev[j] + (s < lr - 1 ? s : lr - 1);
        double part;
        if (use_indel) {
          part = indel_ratio(lval + loff[a], (int)(loff[a + 1] - loff[a]), rval + roff[b],
                             (int)(roff[b + 1] - roff[b]), row);
        } else {
          int err = 0;
          part = jaccard(lval + loff[a], (int)(loff[a + 1] - loff[a]), rval + roff[b],
                         (int)(roff[b + 1] - roff[b]), &err);
          if (err) {
            free(row);
            return -1;
          }
        }
        factor /= 2;
        score += part * factor;
      }
      if (score >= thr) EMIT(score, i, j);
    }
  free(row);
  return n;
}
