// C ABI entry point of the levels-mode Jaccard grid (include/nsm_hip.h).
#include "nsm_common.hpp"

namespace nsm {
template <int W>
int launch_levels(const nsm_set_table* l, const nsm_set_table* r, double threshold, int32_t category_mode,
                  uint32_t flags, nsm_hit* hits, uint64_t capacity, unsigned long long* hit_count,
                  hipStream_t stream);
#define NSM_DECL(W)                                                                                      \
  extern template int launch_levels<W>(const nsm_set_table*, const nsm_set_table*, double, int32_t, uint32_t, \
                                       nsm_hit*, uint64_t, unsigned long long*, hipStream_t);
NSM_DECL(16)
NSM_DECL(32)
NSM_DECL(64)
#undef NSM_DECL
template <int W>
int launch_levels_index(const nsm_set_table* l, const nsm_set_table* r, double threshold, int32_t category_mode, nsm_hit* hits,
                        uint64_t capacity, unsigned long long* hit_count, hipStream_t stream);
extern template int launch_levels_index<16>(const nsm_set_table*, const nsm_set_table*, double, int32_t, nsm_hit*, uint64_t,
                                            unsigned long long*, hipStream_t);
extern template int launch_levels_index<32>(const nsm_set_table*, const nsm_set_table*, double, int32_t, nsm_hit*, uint64_t,
                                            unsigned long long*, hipStream_t);
// candidates from the right table's GLOBAL inverted index (jaccard_levels_global.hip); probe_only: *estimate = posting
// entries the probes would visit, nothing is launched
template <int W>
int launch_levels_global(const nsm_set_table* l, const nsm_set_table* r, double threshold, int32_t category_mode, nsm_hit* hits,
                         uint64_t capacity, unsigned long long* hit_count, hipStream_t stream, bool probe_only, double* estimate);
#define NSM_DECL(W)                                                                                                     \
  extern template int launch_levels_global<W>(const nsm_set_table*, const nsm_set_table*, double, int32_t, nsm_hit*, uint64_t, \
                                              unsigned long long*, hipStream_t, bool, double*);
NSM_DECL(16)
NSM_DECL(32)
NSM_DECL(64)
#undef NSM_DECL
}  // namespace nsm

// Below these thresholds one or two common ids already pass the signature filter of jaccard_levels_kernel for most pairs,
// and candidates come from the per-tile inverted index instead (jaccard_levels_index.hip); NSM_FLAG_INDEX / NSM_FLAG_NO_INDEX
// force / forbid it.  (Sweep: DESIGN.md section 4.2.)
#ifndef NSM_LEV_INDEX_BELOW_16
#define NSM_LEV_INDEX_BELOW_16 0.45
#endif
#ifndef NSM_LEV_INDEX_BELOW_32
#define NSM_LEV_INDEX_BELOW_32 0.65
#endif

extern "C" int nsm_jaccard_levels_grid(const nsm_set_table* left, const nsm_set_table* right, double threshold,
                                       int32_t category_mode, uint32_t flags, nsm_hit* hits, uint64_t capacity,
                                       unsigned long long* hit_count, void* stream) {
  using namespace nsm;
  if (!left || !right || !hit_count || (!hits && capacity)) {
    set_error("nsm_jaccard_levels_grid: null argument");
    return NSM_E_BADARG;
  }
  if (left->width != right->width) {
    set_error("nsm_jaccard_levels_grid: left width %d != right width %d", left->width, right->width);
    return NSM_E_BADARG;
  }
  if (left->n < 0 || right->n < 0 || left->max_levels < 1 || right->max_levels < 1) {
    set_error("nsm_jaccard_levels_grid: bad row count or level stride");
    return NSM_E_BADARG;
  }
  if (category_mode != NSM_CAT_NONE && category_mode != NSM_CAT_INTERSECT &&
      category_mode != NSM_CAT_INTERSECT_OR_BOTH_EMPTY) {
    set_error("nsm_jaccard_levels_grid: unknown category mode %d", category_mode);
    return NSM_E_BADARG;
  }
  if (left->n == 0 || right->n == 0) return 0;
  const nsm_set_table* t[2] = {left, right};
  for (int k = 0; k < 2; ++k) {
    if (!t[k]->ids || !t[k]->cnt || !t[k]->sig || !t[k]->orig || !t[k]->nlev || !t[k]->plen || !t[k]->filt ||
        (category_mode != NSM_CAT_NONE && !t[k]->cat)) {
      set_error("nsm_jaccard_levels_grid: %s table has a null column", k ? "right" : "left");
      return NSM_E_BADARG;
    }
  }
  if ((left->seg == nullptr) != (right->seg == nullptr) ||
      (left->seg && (!left->seg_start || category_mode != NSM_CAT_INTERSECT))) {
    set_error("nsm_jaccard_levels_grid: a category partition needs seg/seg_start on both sides and "
              "NSM_CAT_INTERSECT");
    return NSM_E_BADARG;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  // The right table carries a global inverted index (postings by (category segment, id)): walk posting lists instead of
  // visiting every pair, when the candidates are few -- the kernel's time follows the posting entries visited (~2e-9 ms
  // each) plus the exact score of the candidates that pass the filter record test; the alternatives' follows N x M.
  // (tools/bench_levels.py sweeps, profiles/r04_levels_global_sweep.txt.)
#ifndef NSM_LEV_GLOBAL_DENSITY
#define NSM_LEV_GLOBAL_DENSITY 16.0  // use the global index when (entries visited) x this < N x M
#endif
  if (right->post && right->post_start && right->vocab > 0 && threshold > 0.0 && !(flags & NSM_FLAG_NO_INDEX) &&
      !(flags & NSM_FLAG_TILE_INDEX)) {
    double visited = 0.0;
    const double pairs = static_cast<double>(left->n) * static_cast<double>(right->n);
    const int w = left->width;
    auto go = [&](bool probe, double* est) -> int {
      if (w == 16) return launch_levels_global<16>(left, right, threshold, category_mode, hits, capacity, hit_count, s, probe, est);
      if (w == 32) return launch_levels_global<32>(left, right, threshold, category_mode, hits, capacity, hit_count, s, probe, est);
      return launch_levels_global<64>(left, right, threshold, category_mode, hits, capacity, hit_count, s, probe, est);
    };
    if (w == 16 || w == 32 || w == 64) {
      if (int rc = go(true, &visited)) return rc;
      if ((flags & NSM_FLAG_INDEX) || visited * NSM_LEV_GLOBAL_DENSITY < pairs) return go(false, nullptr);
    }
  }
  if (threshold > 0.0 && !(flags & NSM_FLAG_NO_INDEX) && (left->width == 16 || left->width == 32)) {
    // (a pair without a common id scores exactly 0: with a positive threshold only pairs that share an id can hit)
    const double below = left->width == 16 ? NSM_LEV_INDEX_BELOW_16 : NSM_LEV_INDEX_BELOW_32;
    if ((flags & NSM_FLAG_INDEX) || threshold < below) {
      if (left->width == 16) return launch_levels_index<16>(left, right, threshold, category_mode, hits, capacity, hit_count, s);
      return launch_levels_index<32>(left, right, threshold, category_mode, hits, capacity, hit_count, s);
    }
  }
  switch (left->width) {
    case 16: return launch_levels<16>(left, right, threshold, category_mode, flags, hits, capacity, hit_count, s);
    case 32: return launch_levels<32>(left, right, threshold, category_mode, flags, hits, capacity, hit_count, s);
    case 64: return launch_levels<64>(left, right, threshold, category_mode, flags, hits, capacity, hit_count, s);
    default:
      set_error("nsm_jaccard_levels_grid: width %d not in {16, 32, 64}", left->width);
      return NSM_E_UNSUPPORTED;
  }
}
