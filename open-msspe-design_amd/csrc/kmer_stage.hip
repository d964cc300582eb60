// kmer_stage.hip -- placeholder until the stage-A kernels land (next milestone).
#include "kmer_stage.hpp"

namespace msspe {

int KmerStage::ensure(int, size_t, std::string &) { return MSSPE_OK; }
void KmerStage::release() {}

int KmerStage::run(const uint8_t *, int, size_t, const msspe_kmer_opt &, int, uint64_t *,
                   uint32_t *, int, int *, hipStream_t, std::string &err)
{
    err = "stage A kernels are not built into this library yet";
    return MSSPE_ERR_DEVICE;
}

}  // namespace msspe
