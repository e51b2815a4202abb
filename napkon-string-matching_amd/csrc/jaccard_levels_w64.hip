// Explicit instantiation of the levels-mode Jaccard grid for rows of 64 ids (own TU: compile time).
#include "jaccard_levels_impl.hpp"

namespace nsm {
template int launch_levels<64>(const nsm_set_table*, const nsm_set_table*, double, int32_t, uint32_t,
                               nsm_hit*, uint64_t, unsigned long long*, hipStream_t);
}
