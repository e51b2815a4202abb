// C ABI entry point of the RAW Jaccard grid (include/nsm_hip.h).
#include "nsm_common.hpp"

namespace nsm {
template <int W>
int launch_raw(const nsm_set_table* l, const nsm_set_table* r, double threshold, uint32_t flags,
               nsm_hit* hits, uint64_t capacity, unsigned long long* hit_count, hipStream_t stream);
extern template int launch_raw<16>(const nsm_set_table*, const nsm_set_table*, double, uint32_t,
                                   nsm_hit*, uint64_t, unsigned long long*, hipStream_t);
extern template int launch_raw<32>(const nsm_set_table*, const nsm_set_table*, double, uint32_t,
                                   nsm_hit*, uint64_t, unsigned long long*, hipStream_t);
extern template int launch_raw<64>(const nsm_set_table*, const nsm_set_table*, double, uint32_t,
                                   nsm_hit*, uint64_t, unsigned long long*, hipStream_t);
}  // namespace nsm

extern "C" int nsm_jaccard_raw_grid(const nsm_set_table* left, const nsm_set_table* right,
                                    double threshold, uint32_t flags, nsm_hit* hits, uint64_t capacity,
                                    unsigned long long* hit_count, void* stream) {
  using namespace nsm;
  if (!left || !right || !hit_count || (!hits && capacity)) {
    set_error("nsm_jaccard_raw_grid: null argument");
    return NSM_E_BADARG;
  }
  if (left->width != right->width) {
    set_error("nsm_jaccard_raw_grid: left width %d != right width %d", left->width, right->width);
    return NSM_E_BADARG;
  }
  if (left->n < 0 || right->n < 0) {
    set_error("nsm_jaccard_raw_grid: negative row count");
    return NSM_E_BADARG;
  }
  if (left->n == 0 || right->n == 0) return 0;
  if (!left->ids || !left->cnt || !left->orig || !left->size_start || !right->ids || !right->cnt ||
      !right->orig) {
    set_error("nsm_jaccard_raw_grid: table has a null column");
    return NSM_E_BADARG;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (left->width) {
    case 16: return launch_raw<16>(left, right, threshold, flags, hits, capacity, hit_count, s);
    case 32: return launch_raw<32>(left, right, threshold, flags, hits, capacity, hit_count, s);
    case 64: return launch_raw<64>(left, right, threshold, flags, hits, capacity, hit_count, s);
    default:
      set_error("nsm_jaccard_raw_grid: width %d not in {16, 32, 64}", left->width);
      return NSM_E_UNSUPPORTED;
  }
}
