// pool_sort.hip -- groups the column primers of a cross-dimer launch by base composition.
//
// The all-pairs kernel walks the complementary cells of 64 pairs in lock-step, so a wave runs for
// as long as its heaviest lane.  The number of complementary cells of pair (a, b) is
// sum_x count_a[x] * count_b[3 - x]: for a fixed row primer `a` it depends on `b` only through
// b's base composition.  Sorting the columns by composition therefore gives the 64 lanes of a
// wave (same a, consecutive b) nearly identical trip counts.  Results are written through the
// permutation, so callers never see the internal order.
#include <hip/hip_runtime.h>

#include <cstdint>

#include <rocprim/device/device_radix_sort.hpp>

namespace msspe {

namespace {

constexpr int kBins = 33 * 33 * 33;   // (count_A, count_C, count_G) for oligos up to 32 bases

__device__ __forceinline__ int composition_bin(uint64_t w, int k)
{
    int cnt[4] = {0, 0, 0, 0};
    for (int p = 0; p < k; ++p) cnt[(w >> (2 * p)) & 3]++;
    // order bins so that neighbours differ little: T-rich ... A-rich along the major axis, the
    // minor axes walked back and forth (boustrophedon), so that consecutive bins always differ by
    // one base and a wave that straddles a bin boundary still holds near-equal table sizes
    const int c1 = (cnt[0] & 1) ? 32 - cnt[1] : cnt[1];
    const int c2 = ((cnt[0] + c1) & 1) ? 32 - cnt[2] : cnt[2];
    return (cnt[0] * 33 + c1) * 33 + c2;
}

// key of column q = its composition bin (16 bits); value = its pool index
__global__ void k_bin_keys(const uint64_t *pool, int col0, int ncols, int k, uint32_t *keys, uint32_t *vals)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= ncols) return;
    keys[q] = (uint32_t)composition_bin(pool[col0 + q], k);
    vals[q] = (uint32_t)(col0 + q);
}

__global__ void k_gather_sorted(const uint64_t *pool, int ncols, const uint32_t *perm, uint64_t *sorted)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < ncols) sorted[q] = pool[perm[q]];
}

static_assert(kBins <= (1 << 16), "the composition bin is sorted as a 16-bit key");

}  // namespace

// Scratch the sort needs for ncols columns: two key arrays, one value array and rocPRIM's temporary storage.
size_t pool_sort_scratch_bytes(size_t ncols)
{
    size_t tmp = 0;
    uint32_t *nk = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, nk, nk, nk, nk, ncols, 0u, 16u, (hipStream_t) nullptr);
    const size_t arr = (sizeof(uint32_t) * ncols + 255) & ~(size_t)255;
    return 3 * arr + tmp;
}

// sorted[0..ncols) = the primers pool[col0..col0+ncols) grouped by composition, columns of one composition in
// ascending pool order (a STABLE sort: the lane a pair runs in, hence which pairs share a wave, which are
// handed on for their wave's sake and every stage counter, is the same in every run);
// perm[q] = original pool index of sorted[q].  scratch: pool_sort_scratch_bytes(ncols) bytes.
hipError_t sort_columns_by_composition(const uint64_t *pool, int col0, int ncols, int k, void *scratch,
                                       size_t scratch_bytes, uint64_t *sorted, uint32_t *perm,
                                       hipStream_t stream)
{
    const size_t arr = (sizeof(uint32_t) * (size_t)ncols + 255) & ~(size_t)255;
    if (scratch_bytes < 3 * arr) return hipErrorInvalidValue;
    uint32_t *keys_in = (uint32_t *)scratch;
    uint32_t *keys_out = (uint32_t *)((char *)scratch + arr);
    uint32_t *vals_in = (uint32_t *)((char *)scratch + 2 * arr);
    void *tmp = (char *)scratch + 3 * arr;
    size_t tmp_bytes = scratch_bytes - 3 * arr;
    const int grid = (ncols + 255) / 256;
    hipLaunchKernelGGL(k_bin_keys, dim3(grid), dim3(256), 0, stream, pool, col0, ncols, k, keys_in, vals_in);
    hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, perm, (size_t)ncols, 0u,
                                             16u, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_gather_sorted, dim3(grid), dim3(256), 0, stream, pool, ncols, perm, sorted);
    return hipGetLastError();
}

}  // namespace msspe
