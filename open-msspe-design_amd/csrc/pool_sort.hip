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

// g: side of the composition blocks that are kept together (1: every composition on its own).  A large pool has
// dozens of columns per composition and a wave of 64 consecutive columns sees one or two of them; a pool of the
// reference's size (<= 2,000 primers over the 560 compositions of a 13-mer) has three or four, and 64 consecutive
// columns of the fine order run through a whole line of the composition space -- count_G from end to end -- so that the
// wave's rows are padded to the widest lane far beyond what the table holds (2,000 primers: 23 % of the pairs left the
// first stage for that).  Blocks of g x g x g compositions, walked back and forth like the compositions inside them,
// keep a wave's lanes within g of each other in every count.
__device__ __forceinline__ int composition_bin(uint64_t w, int k, int g)
{
    int cnt[4] = {0, 0, 0, 0};
    for (int p = 0; p < k; ++p) cnt[(w >> (2 * p)) & 3]++;
    // order bins so that neighbours differ little: T-rich ... A-rich along the major axis, the
    // minor axes walked back and forth (boustrophedon), so that consecutive bins always differ by
    // one base and a wave that straddles a bin boundary still holds near-equal table sizes
    const int b0 = cnt[0] / g, b1r = cnt[1] / g, b2r = cnt[2] / g;
    const int b1 = (b0 & 1) ? 32 - b1r : b1r;
    const int b2 = ((b0 + b1) & 1) ? 32 - b2r : b2r;
    const int coarse = (b0 * 33 + b1) * 33 + b2;
    if (g == 1) return coarse;
    const int f0 = cnt[0] % g, f1r = cnt[1] % g, f2r = cnt[2] % g;
    const int f1 = (f0 & 1) ? g - 1 - f1r : f1r;
    const int f2 = ((f0 + f1) & 1) ? g - 1 - f2r : f2r;
    return coarse * (g * g * g) + (f0 * g + f1) * g + f2;
}

// key of column q = its composition bin; value = its pool index
__global__ void k_bin_keys(const uint64_t *pool, int col0, int ncols, int k, int g, uint32_t *keys, uint32_t *vals)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= ncols) return;
    keys[q] = (uint32_t)composition_bin(pool[col0 + q], k, g);
    vals[q] = (uint32_t)(col0 + q);
}

__global__ void k_gather_sorted(const uint64_t *pool, int ncols, const uint32_t *perm, uint64_t *sorted)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < ncols) sorted[q] = pool[perm[q]];
}

static_assert(kBins <= (1 << 16), "the composition bin is a 16-bit key (times the block's g^3 <= 64 fine positions: 22 bits)");

// side of the composition blocks for ncols columns of k bases: about 64 columns per block (one wave)
int block_side(int ncols, int k)
{
    const double bins = (double)(k + 1) * (k + 2) * (k + 3) / 6.0;   // compositions of a k-mer
    int g = 1;
    while (g < 4 && (double)ncols / bins * g * g * g < 64.0) ++g;
    return g;
}

}  // namespace

// Scratch the sort needs for ncols columns: two key arrays, one value array and rocPRIM's temporary storage.
size_t pool_sort_scratch_bytes(size_t ncols)
{
    size_t tmp = 0;
    uint32_t *nk = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, nk, nk, nk, nk, ncols, 0u, 22u, (hipStream_t) nullptr);
    const size_t arr = (sizeof(uint32_t) * ncols + 255) & ~(size_t)255;
    return 3 * arr + tmp;
}

// sorted[0..ncols) = the primers pool[col0..col0+ncols) grouped by composition (small pools: by blocks of neighbouring
// compositions, block_side), columns of one composition in ascending pool order (a STABLE sort: the lane a pair runs in, hence which pairs share a wave, which are
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
    const int g = block_side(ncols, k);
    hipLaunchKernelGGL(k_bin_keys, dim3(grid), dim3(256), 0, stream, pool, col0, ncols, k, g, keys_in, vals_in);
    hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, perm, (size_t)ncols, 0u,
                                             g == 1 ? 16u : 22u, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_gather_sorted, dim3(grid), dim3(256), 0, stream, pool, ncols, perm, sorted);
    return hipGetLastError();
}

}  // namespace msspe
