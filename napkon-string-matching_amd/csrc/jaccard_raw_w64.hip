// Explicit instantiation of the RAW Jaccard grid for rows of 64 ids (own TU: compile time).
#include "jaccard_raw_impl.hpp"

namespace nsm {
template int launch_raw<64>(const nsm_set_table*, const nsm_set_table*, double, uint32_t, nsm_hit*,
                             uint64_t, unsigned long long*, hipStream_t);
}
