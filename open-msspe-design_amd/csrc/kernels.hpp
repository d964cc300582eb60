// kernels.hpp -- launch interface between the C ABI (capi.cpp) and the HIP kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "fast_tables.hpp"
#include "nn_params.hpp"
#include "split_tables.hpp"

namespace msspe {

// One conflict edge as the kernels emit it: pool indices of the ordered pair and the raw double dG (the
// reference's stored text value is msspe_round_fixed_f32(msspe_round_g_f32(dg), 2), applied by the host).
struct EdgeRecord {
    uint32_t a, b;
    double dg;
};

// Where the results of one ordered pair go (all pointers optional, device memory).
struct PairSinks {
    uint32_t *row_conflicts;   // [n]            += 1 per conflicting column
    uint64_t *bitmap;          // [(row1-row0) * words]
    double *dg;                // [(row1-row0) * (col1-col0)]
    double *tm;                // [(row1-row0) * (col1-col0)]
    int row0, col0, ncols, words;
    EdgeRecord *edges;         // conflict edges (i, j, dG), at most edge_cap of them are stored ...
    unsigned long long *edge_count;   // ... and every one is counted here (a count above the capacity = truncated)
    unsigned long long edge_cap;
};

#ifdef __HIPCC__
__device__ __forceinline__ void sink_edge(const PairSinks &s, int row, int col, double dG)
{
    if (!s.edge_count) return;
    const unsigned long long at = atomicAdd(s.edge_count, 1ull);
    if (s.edges && at < s.edge_cap) {
        EdgeRecord e;
        e.a = (uint32_t)row;
        e.b = (uint32_t)col;
        e.dg = dG;
        s.edges[at] = e;
    }
}
#endif

// Generic dense-DP dimer kernel.
//   list != nullptr : work item w = ordered pair (list[w].x, list[w].y) of pool indices
//   list == nullptr : work item w = (row0 + w / ncols, col0 + w % ncols)  (matrix mode)
//   self_mode       : work item w = (w, w); results go to self_t[w] = max(0, t)
// pt[0] / c[0]: ordinary pairs; pt[1] / c[1]: both oligos self-complementary (RC differs).
struct GenericDimerArgs {
    const PairTables *pt;      // device, 2 entries
    ThalConsts c[2];
    const uint64_t *pool;
    int k;
    int mode;                  // kModeAny / kModeEnd1
    const uint2 *list;
    const uint32_t *list_count;   // optional device counter overriding n_work (overflow lists)
    long n_work;
    int self_mode;
    double *self_t;
    void *detail;              // optional ThalDetail[n_work] (list mode; thal_dense.hpp)
    PairSinks sinks;
    double *wsS, *wsH;         // workspace planes, [cell][lane], lanes = ws_lanes
    size_t ws_lanes;
};
hipError_t launch_dimer_generic(const GenericDimerArgs &a, hipStream_t stream);

// Generic hairpin kernel: one lane per oligo; out_t[w] = max(0, t).
struct HairpinArgs {
    const NNTables *tb;        // device
    ThalConsts c;
    const uint64_t *pool;
    int k;
    long n_work;
    double *out_t;
    double *wsS, *wsH;         // (k+2)^2 planes per lane, [cell][lane]
    size_t ws_lanes;
};
hipError_t launch_hairpin_generic(const HairpinArgs &a, hipStream_t stream);
// The same recurrence with one wave per oligo and the planes in LDS (thal_hairpin_wave.hip), k <= 32.
hipError_t launch_hairpin_wave(const HairpinArgs &a, int n_cu, hipStream_t stream);

// oligotm + GC%: one lane per oligo.
hipError_t launch_oligo_tm(const uint64_t *pool, int n, int k, double dna_conc, double mv,
                           double dv, double dntp, double *tm, double *gc, hipStream_t stream);

// Tuned all-pairs kernel (thal ANY, k <= 16, ordinary pairs).  Pairs whose DP does not fit the
// register-resident table are appended to overflow_list (capacity overflow_cap) and must be
// finished by launch_dimer_generic.
struct PairKernelArgs {
    const FastTables *ft;      // device
    ThalConsts c;
    const uint64_t *pool;
    const uint64_t *cols_sorted;   // column primers grouped by composition (pool_sort.hip)
    const uint32_t *perm;          // cols_sorted[q] == pool[perm[q]]
    int ncols_sorted;
    int n, k;
    int row0, row1, col0, col1;
    PairSinks sinks;
    uint2 *overflow_list;
    uint32_t *overflow_count;
    uint32_t overflow_cap;
    uint32_t *work_counter;        // device word for the integer first stage's work queue (launch_pairs_int)
};
hipError_t launch_pairs_fast(const PairKernelArgs &a, hipStream_t stream);
// The main table over an explicit pair list (what the integer stage handed on); pairs that do not
// fit are appended to a.overflow_list.
hipError_t launch_pairs_main_list(const PairKernelArgs &a, const uint2 *in_list,
                                  const uint32_t *in_count, hipStream_t stream);
// Wide instantiation over an explicit pair list (the overflow list of launch_pairs_fast); pairs
// that still do not fit are appended to a.overflow_list.
hipError_t launch_pairs_wide(const PairKernelArgs &a, const uint2 *in_list,
                             const uint32_t *in_count, hipStream_t stream);
// Exact-integer first stage (thal_pairs_int.hip): same contract as launch_pairs_fast; pairs it does
// not answer (ties, oversized tables) are appended to a.overflow_list.  reasons: optional device
// counters [8] ([0] = pairs handed on because of a tie, [1 + b] = reason bit b, see the kernel).
hipError_t launch_pairs_int(const PairKernelArgs &a, const IntTables *it, unsigned long long *reasons, int n_cu,
                            hipStream_t stream);
// The same stage for oligos of up to pairs_row_max_k() bases (thal_pairs_row.hip): a block works on one
// row primer at a time and folds everything the loop terms take from that primer into its LDS table.
hipError_t launch_pairs_row(const PairKernelArgs &a, const IntTables *it, unsigned long long *reasons, int n_cu,
                            hipStream_t stream);
int pairs_row_max_k();
int pairs_row_oob_max_k();   // oligos up to here run the instance that reads LDS beyond its allocation (pairs_row_lds_reads_zero)
bool pairs_row_tables_ok(const IntTables &it);   // host: may this chemistry run the row kernel?
hipError_t pairs_row_lds_reads_zero(hipStream_t stream, int n_cu, bool *ok);   // does this device read 0 beyond a block's LDS allocation?
// List mode of the integer stage: retries the pairs of in_list that carry no "needs f64" mark (bit
// 31 of .x) with a 64-slot table in lanes sorted by table size; everything else passes through.
hipError_t launch_pairs_int_list(const PairKernelArgs &a, const IntTables *it, const uint2 *in_list,
                                 const uint32_t *in_count, unsigned long long *reasons, int n_cu, hipStream_t stream);
// Long oligos (17 .. SplitTables::max_k bases, thal_pairs_split.hip): exact-integer first stage with
// a pair's table split over 2, 4 or 8 lanes; same contract as launch_pairs_int (a.ft is not used).
hipError_t launch_pairs_split(const PairKernelArgs &a, const SplitTables *st, unsigned long long *reasons,
                              int lanes, hipStream_t stream);
hipError_t launch_pairs_split_list(const PairKernelArgs &a, const SplitTables *st, const uint2 *in_list,
                                   const uint32_t *in_count, int n_cu, hipStream_t stream);
int pairs_split_lanes(int k);
// f64 DP with one wave per pair (thal_pairs_wave.hip), oligos up to st->f64_max_k bases.  in_list:
// explicit pairs (count *in_count, at most a.overflow_cap); nullptr: the block rows [a.row0, a.row1) x
// pool columns [a.col0, a.col1).  Pairs it does not take (two self-complementary oligos, oversized
// tables) are appended to a.overflow_list for launch_dimer_generic.
hipError_t launch_pairs_wave(const PairKernelArgs &a, const SplitTables *st, const uint2 *in_list,
                             const uint32_t *in_count, hipStream_t stream);
// Stage B on the same kernel: thal ANY and / or END1 of oligos with themselves from one fill per oligo,
// self_any[row] / self_end[row] = max(0, t) (either may be null); the oligos are [row0, row1) or, with in_list, the
// entries' .x; what it does not take is appended to list as (row, row).
hipError_t launch_self_wave(const SplitTables *st, const ThalConsts &c, const uint64_t *pool, int k, int row0,
                            int row1, double *self_any, double *self_end, const uint2 *in_list,
                            const uint32_t *in_count, uint2 *list, uint32_t *list_count, uint32_t list_cap,
                            uint32_t *work_counter, hipStream_t stream);
// Stage B for large pools, one LANE per oligo (thal_pairs.hip: the f64 register-table kernels over the list of
// (i, i), k <= pairs_fast_max_k()): one fill, both picks.  list_a / list_b: n entries each; counters[0 .. 3).  What is
// left (self-complementary oligos, tables beyond 72 cells) is in list_a, counters[2] entries: launch_self_wave's.
hipError_t launch_self_lists(const FastTables *ft, const ThalConsts &c, const uint64_t *pool, int n, int k,
                             double *self_any, double *self_end, uint2 *list_a, uint2 *list_b, uint32_t *counters,
                             uint32_t list_cap, hipStream_t stream);
int pairs_int_slots();
int pairs_fast_max_k();
// Stable grouping of the column primers by base composition (pool_sort.hip).
size_t pool_sort_scratch_bytes(size_t ncols);
hipError_t sort_columns_by_composition(const uint64_t *pool, int col0, int ncols, int k, void *scratch,
                                       size_t scratch_bytes, uint64_t *sorted, uint32_t *perm,
                                       hipStream_t stream);
int pairs_fast_main_slots();
int pairs_fast_wide_slots();

}  // namespace msspe
