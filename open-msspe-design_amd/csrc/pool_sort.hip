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

__global__ void k_hist(const uint64_t *pool, int col0, int ncols, int k, uint32_t *bins)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < ncols) atomicAdd(&bins[composition_bin(pool[col0 + q], k)], 1u);
}

__global__ void k_scan(uint32_t *bins)   // exclusive scan of kBins counters, one block
{
    __shared__ uint32_t part[1024];
    const int t = threadIdx.x;
    constexpr int per = (kBins + 1023) / 1024;
    uint32_t local[per];
    uint32_t sum = 0;
    for (int e = 0; e < per; ++e) {
        const int idx = t * per + e;
        local[e] = idx < kBins ? bins[idx] : 0u;
        sum += local[e];
    }
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t v = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = t ? part[t - 1] : 0u;
    for (int e = 0; e < per; ++e) {
        const int idx = t * per + e;
        if (idx < kBins) bins[idx] = run;
        run += local[e];
    }
}

__global__ void k_scatter(const uint64_t *pool, int col0, int ncols, int k, uint32_t *cursor,
                          uint64_t *sorted, uint32_t *perm)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= ncols) return;
    const uint64_t w = pool[col0 + q];
    const uint32_t at = atomicAdd(&cursor[composition_bin(w, k)], 1u);
    sorted[at] = w;
    perm[at] = (uint32_t)(col0 + q);
}

}  // namespace

int pool_sort_bins() { return kBins; }

// sorted[0..ncols) = the primers pool[col0..col0+ncols) grouped by composition;
// perm[q] = original pool index of sorted[q].  bins: scratch of pool_sort_bins() uint32.
hipError_t sort_columns_by_composition(const uint64_t *pool, int col0, int ncols, int k,
                                       uint32_t *bins, uint64_t *sorted, uint32_t *perm,
                                       hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(bins, 0, sizeof(uint32_t) * kBins, stream);
    if (e != hipSuccess) return e;
    const int grid = (ncols + 255) / 256;
    hipLaunchKernelGGL(k_hist, dim3(grid), dim3(256), 0, stream, pool, col0, ncols, k, bins);
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, stream, bins);
    hipLaunchKernelGGL(k_scatter, dim3(grid), dim3(256), 0, stream, pool, col0, ncols, k, bins, sorted,
                       perm);
    return hipGetLastError();
}

}  // namespace msspe
