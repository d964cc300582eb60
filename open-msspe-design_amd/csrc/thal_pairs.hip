// thal_pairs.hip -- placeholder for the tuned all-pairs kernel (next milestone).
#include "kernels.hpp"

namespace msspe {
int pairs_fast_max_k() { return 0; }
hipError_t launch_pairs_fast(const PairKernelArgs &, hipStream_t) { return hipErrorNotSupported; }
}  // namespace msspe
