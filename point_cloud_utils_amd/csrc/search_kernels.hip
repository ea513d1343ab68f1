// csrc/search_kernels.hip -- the k > 1 search kernels (lane-per-query k_search<T, K> and wave-per-query k_search_wave<T, K>), instantiated
// in their own translation unit so that the two halves of the device code compile side by side (__graft_entry__.build runs both hipcc
// jobs at once; these two kernel families are more than half of the library's code). pcu_hip.hip declares the same instantiations
// `extern` (search_inst.h) and launches them; flags are the same for both units (-ffp-contract=off is part of the numerical contract).
#include <hip/hip_runtime.h>
#include "grid.h"
#include "search.h"

namespace pcu {
#define PCU_SEARCH_INST template
#include "search_inst.h"
#undef PCU_SEARCH_INST
}
